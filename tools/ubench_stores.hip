// ubench_stores.hip -- what the memory system makes of the eight-picture kernel's WRITE pattern, without the kernel
// (MEASUREMENT TOOL).  Same grid (one 8-wave workgroup per eight 1080p pictures), same owner-lane mapping, same
// instructions (global_store_dwordx4 / dwordx2 with a scalar base and a 32-bit lane offset), no reads, no arithmetic.
//   pattern 0  as recon_oct.hip writes: 4-macroblock strips; per store instruction the lanes (m, h) of an octet cover 64
//              contiguous bytes of TWO luma rows (32 of two chroma rows; a row's 192 RGB bytes in three instructions)
//   pattern 1  what 8-macroblock strips would give: per instruction the eight lanes of an octet cover 128 contiguous bytes
//              of ONE row (whole cache lines), 64 of a chroma row, a row's 384 RGB bytes in three instructions
//   pattern 2  pattern 0 without the RGB stores (planes only)
//   --gap N    N x 64 dependent v_add between two flushes of a wave (the real kernel computes ~75k cycles per flush)
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_stores tools/ubench_stores.hip     run (GPU box): ./ubench_stores
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int W = 120, H = 68, PITCH = W * 16, CPITCH = W * 8;
constexpr uint32_t PLANE_Y = (uint32_t)W * H * 256, PLANE_C = (uint32_t)W * H * 64;
constexpr uint32_t YUV_PIC = (uint32_t)W * H * 384, RGB_PIC = (uint32_t)W * H * 768;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

// (s_nop 4: a vector-memory read of an SGPR base needs five wait states behind a VALU / SALU write of it, and inline
// assembly hides that hazard from the compiler -- as in recon_oct.hip)
__device__ __forceinline__ void st4(uint32_t off, v4i d, uint8_t *base)
{
    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2" : : "v"(off), "v"(d), "s"(base) : "memory");
}
__device__ __forceinline__ void st2(uint32_t off, v2i d, uint8_t *base)
{
    asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2" : : "v"(off), "v"(d), "s"(base) : "memory");
}

template <int PATTERN>
__global__ __launch_bounds__(512) void store_kernel(uint8_t *yuv, uint8_t *rgb, int gap)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, o = lane >> 3, j = lane & 7;
    uint8_t *gyuv = yuv + (size_t)blockIdx.x * 8 * YUV_PIC;
    uint8_t *grgb = rgb + (size_t)blockIdx.x * 8 * RGB_PIC;
    const uint32_t oy = (uint32_t)o * YUV_PIC, orgb = (uint32_t)o * RGB_PIC;
    v4i d = {lane, wave, 3, 4};
    v2i d2 = {lane, wave};
    uint32_t acc = lane;
    for (int row = wave; row < H; row += 8) {
        if (PATTERN == 1) {
            for (int strip = 0; strip < W / 8; strip++) {   // 8-macroblock strips: 128-byte luma runs
                for (int r = 0; r < 16; r++) st4(oy + (uint32_t)((row * 16 + r) * PITCH + strip * 128 + j * 16), d, gyuv);
                for (int r = 0; r < 8; r++) {
                    st2(oy + PLANE_Y + (uint32_t)((row * 8 + r) * CPITCH + strip * 64 + j * 8), d2, gyuv);
                    st2(oy + PLANE_Y + PLANE_C + (uint32_t)((row * 8 + r) * CPITCH + strip * 64 + j * 8), d2, gyuv);
                }
                for (int r = 0; r < 16; r++)
                    for (int k = 0; k < 3; k++)
                        st4(orgb + (uint32_t)((row * 16 + r) * PITCH + strip * 128) * 3u + (uint32_t)(k * 128 + j * 16), d, grgb);
                for (int g = 0; g < 2 * gap; g++)
#pragma unroll
                    for (int u = 0; u < 64; u++) asm volatile("v_add_u32 %0, %0, %0" : "+v"(acc));
            }
        } else {
            const int m = j & 3, h = j >> 2;
            for (int strip = 0; strip < W / 4; strip++) {
                const uint32_t x0 = (uint32_t)(strip * 64 + m * 16);
                for (int i = 0; i < 4; i++) {
                    const uint32_t p = oy + (uint32_t)((row * 16 + 4 * i + 2 * h) * PITCH) + x0;
                    st4(p, d, gyuv);
                    st4(p + PITCH, d, gyuv);
                }
                for (int i = 0; i < 4; i++) {
                    const uint32_t p = oy + PLANE_Y + (uint32_t)((row * 8 + 2 * i + h) * CPITCH) + (x0 >> 1);
                    st2(p, d2, gyuv);
                    st2(p + PLANE_C, d2, gyuv);
                }
                if (PATTERN == 0) {
                    for (int i = 0; i < 4; i++) {
                        const uint32_t pa = orgb + ((uint32_t)((row * 16 + 4 * i + 2 * h) * PITCH) + x0) * 3u, pb = pa + 3u * PITCH;
                        st4(pa, d, grgb); st4(pa + 16, d, grgb); st4(pa + 32, d, grgb);
                        st4(pb, d, grgb); st4(pb + 16, d, grgb); st4(pb + 32, d, grgb);
                    }
                }
                for (int g = 0; g < gap; g++)
#pragma unroll
                    for (int u = 0; u < 64; u++) asm volatile("v_add_u32 %0, %0, %0" : "+v"(acc));
            }
        }
    }
    if (acc == 0x1234567u) gyuv[0] = 1;
}

int main(int argc, char **argv)
{
    int F = 2048;
    uint8_t *yuv, *rgb;
    CHECK(hipMalloc(&yuv, (size_t)F * YUV_PIC));
    CHECK(hipMalloc(&rgb, (size_t)F * RGB_PIC));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-44s %6s %9s %9s\n", "pattern", "gap", "ms", "GB/s");
    const char *names[3] = {"0: as recon_oct (64-B runs, two rows/instr)", "1: 8-MB strips (128-B runs, whole lines)", "2: as recon_oct, planes only"};
    for (int pat = 0; pat < 3; pat++) {
        for (int gap : {0, 100, 200, 290, 400}) {
            // gap: 64 v_add of 4-5 cycles each per unit; 290 units ~ 75-80k cycles between two flushes, like the real kernel
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0, 0));
                if (pat == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(F / 8), dim3(512), 0, 0, yuv, rgb, gap);
                else if (pat == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(F / 8), dim3(512), 0, 0, yuv, rgb, gap);
                else hipLaunchKernelGGL(store_kernel<2>, dim3(F / 8), dim3(512), 0, 0, yuv, rgb, gap);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            const double bytes = (double)F * W * H * (pat == 2 ? 384.0 : 1152.0);
            printf("%-44s %6d %9.3f %9.1f\n", names[pat], gap, best, bytes / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
