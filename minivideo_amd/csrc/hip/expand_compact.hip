// expand_compact.hip -- compact pictures (include/minivideo_hotpath.h, "compact pictures": what crosses PCIe) -> the packed
// macroblock records the reconstruction kernels read.  gfx950 only.  Memory-bound: ~140 B read + 800 B written per
// macroblock; the record is assembled in LDS (zero, scatter the levels, copy out in 16-byte pieces) so that HBM sees
// whole-line writes.  Replaces nothing in the reference (its Macroblock_t never leaves the host); it is the GPU half of
// the record hand-over at h264_macroblock.c:278.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "minivideo_hotpath.h"
#include "recon_kernels.h"
#include "recon_device.h"

namespace mvhp {

// 256 threads = 8 macroblocks, 32 lanes each (two macroblocks per wavefront; lanes of one macroblock never wait for
// another wavefront, so wavefront-scope synchronisation is all it takes)
__global__ __launch_bounds__(256) void expand_compact_kernel(ExpandArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[8][200];
    const int lane = threadIdx.x & 31, sub = threadIdx.x >> 5;
    const long long mbg = (long long)blockIdx.x * 8 + sub;
    const bool live = mbg < (long long)a.mbs * a.n_pictures;
    const long long mbc = live ? mbg : 0;
    const int pic = (int)(mbc / a.mbs), mb = (int)(mbc - (long long)pic * a.mbs);
    const uint8_t *pbase = a.compact + (size_t)pic * a.stride;
    const uint32_t off = reinterpret_cast<const uint32_t *>(pbase)[mb];
    const uint32_t *rec = reinterpret_cast<const uint32_t *>(pbase + (size_t)a.mbs * 4 + off);
    uint32_t *t = tile[sub];
    const uint32_t w1 = rec[1];
    const bool dense = ((w1 >> 8) & 1u) != 0;                       // header.flags bit 0
    const uint32_t n = dense ? 0u : min(rec[7], (uint32_t)MVHP_MB_COEFS);   // header.reserved1
    // header (the transfer-only fields cleared) and, for a dense macroblock, its coefficient area
#pragma unroll
    for (int i = lane; i < 200; i += 32) {
        uint32_t v = 0;
        if (i < 8 || dense) v = rec[i];
        if (i == 1) v &= ~0x0000ff00u;
        if (i == 7) v = 0;
        t[i] = v;
    }
    WAVE_SYNC();
    int16_t *coef = reinterpret_cast<int16_t *>(t + 8);
    for (uint32_t i = (uint32_t)lane; i < n; i += 32) {
        const uint32_t e = rec[8 + i];
        const uint32_t pos = e & 0xffffu;
        if (pos < (uint32_t)MVHP_MB_COEFS) coef[pos] = (int16_t)(e >> 16);
    }
    WAVE_SYNC();
    if (live) {
        uint4 *out = reinterpret_cast<uint4 *>(a.packed + (size_t)mbg * MVHP_MB_BYTES);
        const uint4 *src = reinterpret_cast<const uint4 *>(t);
        out[lane] = src[lane];
        if (lane < 18) out[lane + 32] = src[lane + 32];
    }
}

hipError_t launch_expand(const ExpandArgs &a, hipStream_t stream)
{
    const long long total = (long long)a.mbs * a.n_pictures;
    if (total <= 0) return hipSuccess;
    const unsigned groups = (unsigned)((total + 7) / 8);
    hipLaunchKernelGGL(expand_compact_kernel, dim3(groups), dim3(256), 0, stream, a);
    return hipGetLastError();
}

} // namespace mvhp
