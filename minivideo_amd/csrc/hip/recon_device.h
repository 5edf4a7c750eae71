// recon_device.h -- device helpers shared by the reconstruction kernels (tables, the
// prediction tap tables, the inverse transforms).  Included by recon_kernels.hip (one picture
// per workgroup, one wavefront per macroblock row) and recon_quad.hip (four pictures per
// workgroup, 16 lanes per picture).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvhp {

__device__ static const int c_v4x4[18] = {10, 16, 13, 11, 18, 14, 13, 20, 16, 14, 23, 18, 16, 25, 20, 18, 29, 23};
__device__ static const int c_v8x8[36] = {20, 18, 32, 19, 25, 24, 22, 19, 35, 21, 28, 26, 26, 23, 42, 24, 33, 31,
                                          28, 25, 45, 26, 35, 33, 32, 28, 51, 30, 40, 38, 36, 32, 58, 34, 46, 43};
__device__ static const uint8_t c_qpc[22] = {29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36,
                                             36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};

#define WAVE_SYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

__device__ __forceinline__ int clip255(int v) { return min(max(v, 0), 255); }

// Prediction table entry for an n x n block (n = 4 or 8), unified edge array
// EE: EE[0..1] = left[n-1] replicated, EE[2+j] = left[n-1-j], EE[cor] = p[-1,-1],
// EE[top0+i] = top[i] (i < 2n), EE[top0+2n] = top[2n-1] replicated.
// type 0: EE[k]; 1: (EE[k]+EE[k+1]+1)>>1; 2: (EE[k]+2EE[k+1]+EE[k+2]+2)>>2.
// Restates the nine mode functions h264_intra_prediction.c:496-960 (4x4) and
// :1366-1793 (8x8).
static __device__ int mode_entry(int n, int mode, int x, int y)
{
    const int top0 = (n == 4) ? 7 : 11, cor = top0 - 1, left0 = top0 - 2;
    int k = 0, t = 0;
    switch (mode) {
    case 0: k = top0 + x; t = 0; break;
    case 1: k = left0 - y; t = 0; break;
    case 3: k = top0 + x + y; t = 2; break;
    case 4: k = left0 + x - y; t = 2; break;
    case 5: {
        int z = 2 * x - y;
        if (z >= 0) { if ((z & 1) == 0) { k = cor + x - (y >> 1); t = 1; } else { k = left0 + x - (y >> 1); t = 2; } }
        else if (z == -1) { k = left0; t = 2; }
        else { k = cor - y + 2 * x; t = 2; }
        break;
    }
    case 6: {
        int z = 2 * y - x;
        if (z >= 0) { k = left0 - y + (x >> 1); t = ((z & 1) == 0) ? 1 : 2; }
        else if (z == -1) { k = left0; t = 2; }
        else { k = left0 - 1 + x - 2 * y; t = 2; }
        break;
    }
    case 7: k = top0 + x + (y >> 1); t = (y & 1) ? 2 : 1; break;
    case 8: {
        int s = y + (x >> 1), z = x + 2 * y;
        if ((z & 1) == 0) { k = left0 - 1 - s; t = 1; } else { k = left0 - 2 - s; t = 2; }
        if (k < 0) k = 0;
        break;
    }
    default: break;
    }
    return k | (t << 5);
}

// Every directional mode reduces to ONE formula, (a + 2b + c + 2) >> 2, by choosing the taps:
// copy EE[k] = taps (k,k,k); (EE[k]+EE[k+1]+1)>>1 = taps (k,k+1,k) because (2a+2b+2)>>2 == (a+b+1)>>1.
// tap4 entry: three bytes, each the offset (+33) of the tap relative to the tile index of the
// block's top-left sample: top[i] -> -32+i, corner -> -33, left[j] -> 32j-1 (tile rows are 32 bytes).
static __device__ uint32_t tap4_entry(int mode, int x, int y, bool no_upright)
{
    const int e = mode_entry(4, mode, x, y), k = e & 31, t = e >> 5;
    const int idx[3] = {k, (t == 0) ? k : k + 1, (t == 2) ? k + 2 : k};
    uint32_t out = 0;
    for (int q = 0; q < 3; q++) {
        const int i = idx[q];
        int off;
        if (i >= 6) off = -32 + min(i - 7, no_upright ? 3 : 7);
        else off = 32 * min(5 - i, 3) - 1;
        out |= (uint32_t)(off + 33) << (8 * q);
    }
    return out;
}
// tap8 entry: three indices into E8 (EE8 index space of mode_entry(8,...)), clamped to the replicated ends.
static __device__ uint32_t tap8_entry(int mode, int x, int y)
{
    const int e = mode_entry(8, mode, x, y), k = e & 31, t = e >> 5;
    const int i0 = k, i1 = min((t == 0) ? k : k + 1, 27), i2 = min((t == 2) ? k + 2 : k, 27);
    return (uint32_t)i0 | ((uint32_t)i1 << 8) | ((uint32_t)i2 << 16);
}

// ---------------------------------------------------------------------------
// residual arithmetic (h264_transform.c)
// ---------------------------------------------------------------------------

// idct4x4, h264_transform.c:1145-1191, in place on d[row*4+col]
__device__ __forceinline__ void idct4x4(int d[16])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int e0 = d[i * 4 + 0] + d[i * 4 + 2];
        int e1 = d[i * 4 + 0] - d[i * 4 + 2];
        int e2 = (d[i * 4 + 1] >> 1) - d[i * 4 + 3];
        int e3 = d[i * 4 + 1] + (d[i * 4 + 3] >> 1);
        d[i * 4 + 0] = e0 + e3;
        d[i * 4 + 1] = e1 + e2;
        d[i * 4 + 2] = e1 - e2;
        d[i * 4 + 3] = e0 - e3;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int g0 = d[0 + j] + d[8 + j];
        int g1 = d[0 + j] - d[8 + j];
        int g2 = (d[4 + j] >> 1) - d[12 + j];
        int g3 = d[4 + j] + (d[12 + j] >> 1);
        d[0 + j]  = (g0 + g3) >> 6;   // the +32 rounding term was added to d[0] by the caller: every output
        d[4 + j]  = (g1 + g2) >> 6;   // contains d[0] exactly once, unshifted, so this equals (h + 32) >> 6
        d[8 + j]  = (g1 - g2) >> 6;
        d[12 + j] = (g0 - g3) >> 6;
    }
}

// idct4x4 + the final (h + 32) >> 6 on PACKED pairs: h is saturated to int16 first (v_cvt_pk_i16_i32) and the two
// halves are shifted together (v_pk_ashrrev_i16).  Exact for the reconstructed sample: |h| > 32767 means |r| >= 511,
// where clip255(pred + r) no longer depends on r, and the saturated value shifts to +511 / -512 -- same side.
__device__ __forceinline__ void idct4x4_packed(int d[16], int out[8])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int e0 = d[i * 4 + 0] + d[i * 4 + 2];
        int e1 = d[i * 4 + 0] - d[i * 4 + 2];
        int e2 = (d[i * 4 + 1] >> 1) - d[i * 4 + 3];
        int e3 = d[i * 4 + 1] + (d[i * 4 + 3] >> 1);
        d[i * 4 + 0] = e0 + e3;
        d[i * 4 + 1] = e1 + e2;
        d[i * 4 + 2] = e1 - e2;
        d[i * 4 + 3] = e0 - e3;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int g0 = d[0 + j] + d[8 + j];
        int g1 = d[0 + j] - d[8 + j];
        int g2 = (d[4 + j] >> 1) - d[12 + j];
        int g3 = d[4 + j] + (d[12 + j] >> 1);
        d[0 + j]  = g0 + g3;
        d[4 + j]  = g1 + g2;
        d[8 + j]  = g1 - g2;
        d[12 + j] = g0 - g3;
    }
    typedef short short2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const short2_t v = __builtin_amdgcn_cvt_pk_i16(d[2 * i], d[2 * i + 1]);
        out[i] = __builtin_bit_cast(int, (short2_t)(v >> (short)6));
    }
}

// ... with the output packed by row pairs: out[i], i = 4*y + x (y = 0, 1), = residuals of samples (x, y) and (x, y + 2)
__device__ __forceinline__ void idct4x4_ypairs(int d[16], int out[8])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int e0 = d[i * 4 + 0] + d[i * 4 + 2];
        int e1 = d[i * 4 + 0] - d[i * 4 + 2];
        int e2 = (d[i * 4 + 1] >> 1) - d[i * 4 + 3];
        int e3 = d[i * 4 + 1] + (d[i * 4 + 3] >> 1);
        d[i * 4 + 0] = e0 + e3;
        d[i * 4 + 1] = e1 + e2;
        d[i * 4 + 2] = e1 - e2;
        d[i * 4 + 3] = e0 - e3;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int g0 = d[0 + j] + d[8 + j];
        int g1 = d[0 + j] - d[8 + j];
        int g2 = (d[4 + j] >> 1) - d[12 + j];
        int g3 = d[4 + j] + (d[12 + j] >> 1);
        d[0 + j]  = g0 + g3;
        d[4 + j]  = g1 + g2;
        d[8 + j]  = g1 - g2;
        d[12 + j] = g0 - g3;
    }
    typedef short short2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const short2_t v = __builtin_amdgcn_cvt_pk_i16(d[i], d[i + 8]);
        out[i] = __builtin_bit_cast(int, (short2_t)(v >> (short)6));
    }
}

// 8-point butterfly of idct8x8 (h264_transform.c:1308-1342 / :1344-1378), in place.
__device__ __forceinline__ void idct8_1d(int d[8])
{
    int e0 = d[0] + d[4];
    int e1 = -d[3] + d[5] - d[7] - (d[7] >> 1);
    int e2 = d[0] - d[4];
    int e3 = d[1] + d[7] - d[3] - (d[3] >> 1);
    int e4 = (d[2] >> 1) - d[6];
    int e5 = -d[1] + d[7] + d[5] + (d[5] >> 1);
    int e6 = d[2] + (d[6] >> 1);
    int e7 = d[3] + d[5] + d[1] + (d[1] >> 1);
    int f0 = e0 + e6;
    int f1 = e1 + (e7 >> 2);
    int f2 = e2 + e4;
    int f3 = e3 + (e5 >> 2);
    int f4 = e2 - e4;
    int f5 = (e3 >> 2) - e5;
    int f6 = e0 - e6;
    int f7 = e7 - (e1 >> 2);
    d[0] = f0 + f7;
    d[1] = f2 + f5;
    d[2] = f4 + f3;
    d[3] = f6 + f1;
    d[4] = f6 - f1;
    d[5] = f4 - f3;
    d[6] = f2 - f5;
    d[7] = f0 - f7;
}

// sign of idct_dccoeff_4x4[i][k] (h264_transform.c:62-68): rows ++++, ++--, +--+, +-+-
__device__ __forceinline__ bool hneg(int i, int k)
{
    return (i == 1 && k >= 2) || (i == 2 && (k == 1 || k == 2)) || (i == 3 && (k & 1));
}

// Two residuals -> packed int16 with signed saturation (v_cvt_pk_i16_i32).  Saturating is exact for the
// final sample: clip255(pred + r) only depends on r inside [-255, 255].
__device__ __forceinline__ int pack_res(int a, int b)
{
    typedef short short2_t __attribute__((ext_vector_type(2)));
    const short2_t v = __builtin_amdgcn_cvt_pk_i16(a, b);
    return __builtin_bit_cast(int, v);
}

__device__ __forceinline__ void unpack8(const int4 v, int d[8])
{
    d[0] = (int16_t)(v.x & 0xffff); d[1] = v.x >> 16; d[2] = (int16_t)(v.y & 0xffff); d[3] = v.y >> 16;
    d[4] = (int16_t)(v.z & 0xffff); d[5] = v.z >> 16; d[6] = (int16_t)(v.w & 0xffff); d[7] = v.w >> 16;
}

__device__ __forceinline__ int sum4(uint32_t w) { return (int)__builtin_amdgcn_sad_u8(w, 0u, 0u); }

} // namespace mvhp
