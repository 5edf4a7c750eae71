// ubench_issue.hip -- what one SIMD of an MI355X issues per cycle for the integer instructions the reconstruction kernels
// are made of, at 1 / 2 / 3 / 4 waves per SIMD (MEASUREMENT TOOL, not part of the product).
//   build: hipcc --offload-arch=gfx950 -O3 -o ubench_issue tools/ubench_issue.hip      run (GPU box): ./ubench_issue
// For every instruction: cycles per wave-instruction per SIMD with W waves per SIMD issuing INDEPENDENT instructions
// (8 chains) and DEPENDENT ones (1 chain).  One workgroup per CU (LDS-limited), 256 workgroups, s_memtime around the loop
// of the slowest wave of workgroup 0..255 (max), plus the wall time of the launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 2000, UNROLL = 64;   // 128k instructions per wave

#define OP8(ASM)                                                                                                        \
    asm volatile(ASM(0) "\n\t" ASM(1) "\n\t" ASM(2) "\n\t" ASM(3) "\n\t" ASM(4) "\n\t" ASM(5) "\n\t" ASM(6) "\n\t" ASM(7)       \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k), "v"(k2))
#define OP1(ASM)                                                                                                        \
    asm volatile(ASM(0) "\n\t" ASM(0) "\n\t" ASM(0) "\n\t" ASM(0) "\n\t" ASM(0) "\n\t" ASM(0) "\n\t" ASM(0) "\n\t" ASM(0)       \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k), "v"(k2))

#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8"
#define A_PKADD(i) "v_pk_add_i16 %" #i ", %" #i ", %8"
#define A_PKADDC(i) "v_pk_add_i16 %" #i ", %" #i ", %8 clamp"
#define A_MAD16(i) "v_mad_i32_i16 %" #i ", %" #i ", %8, %9"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9"
#define A_MUL32(i) "v_mul_lo_u32 %" #i ", %" #i ", %8"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9"
#define A_LERP(i) "v_lerp_u8 %" #i ", %" #i ", %8, %9"
#define A_SATPK(i) "v_sat_pk_u8_i16 %" #i ", %" #i
#define A_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define A_ADDDPP(i) "v_add_u32_dpp %" #i ", %" #i ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define A_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc"
#define A_ASHR(i) "v_ashrrev_i32 %" #i ", 1, %" #i
#define A_PKASHR(i) "v_pk_ashrrev_i16 %" #i ", 1, %" #i
#define A_PKMUL(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %8"
#define A_PKMAD(i) "v_pk_mad_i16 %" #i ", %" #i ", %8, %9"
#define A_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9"
#define A_ALIGN(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 1"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8"
#define A_SAD(i) "v_sad_u8 %" #i ", %" #i ", %8, %9"
#define A_ASHRPK(i) "v_ashr_pk_u8_i32 %" #i ", %" #i ", %8, %9"
#define A_CVTPK(i) "v_cvt_pk_i16_i32 %" #i ", %" #i ", %8"
#define A_MOV64(i) "v_lshlrev_b32 %" #i ", 0, %" #i
#define A_PKSUB(i) "v_pk_sub_i16 %" #i ", %" #i ", %8"
#define A_PKMAX(i) "v_pk_max_i16 %" #i ", %" #i ", %8"

template <int OPID, bool DEP>
__global__ __launch_bounds__(1024) void issue_kernel(uint64_t *cyc, uint32_t *sink, uint32_t k, uint32_t k2)
{
    extern __shared__ uint8_t lds[];
    uint32_t r[8];
    for (int i = 0; i < 8; i++) r[i] = threadIdx.x * 8 + i + k;
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(r[0]), "v"(k2) : "vcc");
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL / 8; u++) {
#define RUN(ID, ASM) if (OPID == ID) { if (DEP) OP1(ASM); else OP8(ASM); }
            RUN(0, A_ADD) RUN(1, A_PKADD) RUN(2, A_PKADDC) RUN(3, A_MAD16) RUN(4, A_MAD24) RUN(5, A_MUL32) RUN(6, A_PERM)
            RUN(7, A_LERP) RUN(8, A_SATPK) RUN(9, A_DPP) RUN(10, A_ADDDPP) RUN(11, A_CND) RUN(12, A_ASHR) RUN(13, A_PKASHR)
            RUN(14, A_PKMUL) RUN(15, A_PKMAD) RUN(16, A_DOT4) RUN(17, A_ALIGN) RUN(18, A_BFE) RUN(19, A_AND) RUN(20, A_ADD3)
            RUN(21, A_LSHLADD) RUN(22, A_SAD) RUN(23, A_ASHRPK) RUN(24, A_CVTPK) RUN(25, A_PKSUB) RUN(26, A_PKMAX)
#undef RUN
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= r[i];
    if (s == 0x12345u) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (threadIdx.x == 0 && lds[0] == 77 && k == 12345u) sink[1] = 1;
}

// ---- LDS dependent-read latency: a pointer chase through LDS with byte / dword reads, one lane group active ----
template <int KIND>   // 0 ds_read_u8 chain, 1 ds_read_b32 chain, 2 ds_bpermute chain, 3 u8 write -> read round trip
__global__ __launch_bounds__(1024) void lds_chain_kernel(uint64_t *cyc, uint32_t *sink, int steps)
{
    extern __shared__ uint8_t lds[];
    uint32_t *l32 = reinterpret_cast<uint32_t *>(lds);
    const int wave = threadIdx.x >> 6;
    uint8_t *mine = lds + wave * 4096;
    for (int i = threadIdx.x & 63; i < 1024; i += 64) reinterpret_cast<uint32_t *>(mine)[i] = 0;   // every chase stays at offset 0..3
    __syncthreads();
    uint32_t p = 0, acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int s = 0; s < steps; s++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == 0) p = mine[p + (threadIdx.x & 63)];
            else if (KIND == 1) p = reinterpret_cast<uint32_t *>(mine)[p + (threadIdx.x & 63)];
            else if (KIND == 2) p = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x + 1) & 63) * 4 + (p & 0)), (int)p);
            else { mine[(threadIdx.x & 63) + (p & 3)] = (uint8_t)p; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); p = mine[((threadIdx.x + 1) & 63) + (p & 3)]; }
            acc += p;
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (acc == 0x7777u) sink[2] = acc + l32[0];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

struct Res { double cyc_max, cyc_med, ms; };

template <typename K, typename... Args>
static Res run(K kern, int threads, size_t lds, uint64_t *d_cyc, Args... args)
{
    CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipMemset(d_cyc, 0, 256 * 16 * 8));
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), lds, 0, d_cyc, args...);   // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), lds, 0, d_cyc, args...);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(256 * 16);
    CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<uint64_t> v;
    for (uint64_t x : h) if (x) v.push_back(x);
    std::sort(v.begin(), v.end());
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return Res{(double)v.back(), (double)v[v.size() / 2], ms};
}

template <int OPID>
static void bench_op(const char *name, uint64_t *d_cyc, uint32_t *d_sink)
{
    const size_t lds = 96 * 1024;   // one workgroup per CU
    printf("%-22s", name);
    for (int dep = 0; dep < 2; dep++) {
        for (int wps = 1; wps <= 4; wps++) {
            const int threads = wps * 256;
            Res r = dep ? run(issue_kernel<OPID, true>, threads, lds, d_cyc, d_sink, 3u, 5u)
                        : run(issue_kernel<OPID, false>, threads, lds, d_cyc, d_sink, 3u, 5u);
            // cycles per wave-instruction per SIMD = cycles of the median wave / (instructions per wave * waves per SIMD)
            // s_memtime ticks at 100 MHz on gfx950?  report both the tick-based and the wall-based figure (2.4 GHz assumed)
            const double n = (double)ITER * UNROLL;
            printf(" | %5.2f %5.2f", r.cyc_med / n / wps, r.ms * 1e-3 * 2.4e9 / n / wps);
        }
        printf(dep ? "\n" : "  ||dep");
    }
}

int main()
{
    uint64_t *d_cyc;
    uint32_t *d_sink;
    CHECK(hipMalloc(&d_cyc, 256 * 16 * 8));
    CHECK(hipMalloc(&d_sink, 64));
    printf("cycles per wave-instruction per SIMD: [memtime-based wall@2.4GHz-based] at 1 2 3 4 waves per SIMD; independent || dependent\n");
#define B(ID, NAME) bench_op<ID>(NAME, d_cyc, d_sink);
    B(0, "v_add_u32") B(1, "v_pk_add_i16") B(2, "v_pk_add_i16 clamp") B(3, "v_mad_i32_i16") B(4, "v_mad_u32_u24") B(5, "v_mul_lo_u32")
    B(6, "v_perm_b32") B(7, "v_lerp_u8") B(8, "v_sat_pk_u8_i16") B(9, "v_mov_b32_dpp") B(10, "v_add_u32_dpp") B(11, "v_cndmask_b32")
    B(12, "v_ashrrev_i32") B(13, "v_pk_ashrrev_i16") B(14, "v_pk_mul_lo_u16") B(15, "v_pk_mad_i16") B(16, "v_dot4_u32_u8")
    B(17, "v_alignbyte_b32") B(18, "v_bfe_u32") B(19, "v_and_b32") B(20, "v_add3_u32") B(21, "v_lshl_add_u32") B(22, "v_sad_u8")
    B(23, "v_ashr_pk_u8_i32") B(24, "v_cvt_pk_i16_i32") B(25, "v_pk_sub_i16") B(26, "v_pk_max_i16")
    printf("\nLDS dependent chains, cycles per step (memtime-based | wall-based), 1 / 2 / 4 waves per SIMD\n");
    const char *names[4] = {"ds_read_u8 chain", "ds_read_b32 chain", "ds_bpermute chain", "ds_write_b8 -> ds_read_u8"};
    for (int kind = 0; kind < 4; kind++) {
        printf("%-26s", names[kind]);
        for (int wps : {1, 2, 4}) {
            const int steps = 500;
            Res r = kind == 0 ? run(lds_chain_kernel<0>, wps * 256, (size_t)96 * 1024, d_cyc, d_sink, steps)
                  : kind == 1 ? run(lds_chain_kernel<1>, wps * 256, (size_t)96 * 1024, d_cyc, d_sink, steps)
                  : kind == 2 ? run(lds_chain_kernel<2>, wps * 256, (size_t)96 * 1024, d_cyc, d_sink, steps)
                              : run(lds_chain_kernel<3>, wps * 256, (size_t)96 * 1024, d_cyc, d_sink, steps);
            printf(" | %6.1f %6.1f", r.cyc_med / (steps * 16.0), r.ms * 1e-3 * 2.4e9 / (steps * 16.0));
        }
        printf("\n");
    }
    return 0;
}
