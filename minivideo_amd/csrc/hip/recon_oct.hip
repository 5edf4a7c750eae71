// recon_oct.hip -- the large-batch form of the reconstruction kernel (gfx950 only): EIGHT pictures per wavefront.
//
//   recon_oct_kernel<NW, RGB>   same contract as recon_quad_kernel / recon_rows_kernel: replaces
//                               intra_prediction_process() (decoder/h264/h264_intra_prediction.c:112-145), all of
//                               h264_transform.c, the planar gather of export.c:65-188 and mb_to_rgb()
//                               (export_utils.c:209-324) for whole pictures.
//
// Mapping: as recon_quad.hip, with octets instead of quarters: octet o (8 lanes) of every wavefront works on picture
// 8*blockIdx+o, wave w owns macroblock rows w, w+NW, ... of all eight.  Why eight: the reconstruction is bound by
// VALU issue, and an instruction costs the same for 8 or 64 active lanes.  With 8 lanes per picture
//   * the 16 dependent Intra4x4 block steps serve eight macroblocks (one lane predicts two samples of the block),
//   * chroma (8 blocks of 4x4) keeps all lanes of the octet busy (16 lanes per picture left half of them idle),
//   * everything that is paid once per step (header decoding, the row-above wait, neighbour bookkeeping) is shared
//     by eight macroblocks.
// Lane j of an octet owns luma 4x4 blocks 2j and 2j+1 (64 contiguous bytes of the record: for an Intra8x8 macroblock
// the same bytes are rows 4(j&1)..+3 of 8x8 block j>>1) and chroma block j.  The finished macroblock is written out by
// OWNER lanes: lane (m, h) = (j & 3, j >> 2) keeps eight luma rows of macroblock m of the 4-macroblock output strip (see
// the write-out).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "minivideo_hotpath.h"
#include "recon_kernels.h"
#include "recon_device.h"
#include "recon_batch_device.h"

namespace mvhp {

#ifndef MVHP_LOAD_HINT
#define MVHP_LOAD_HINT ""   // cache-policy suffix of the record loads (measurement builds try " nt")
#endif
#ifndef MVHP_STORE_HINT
#define MVHP_STORE_HINT ""  // the same for the strip stores
#endif
// The eight-picture kernel's LDS block per picture and wave: QLds (recon_batch_device.h) with the filtered Intra8x8
// edge kept as three 28-entry arrays -- G[0][k] = p'[k] (the unified edge EE of recon_device.h mode_entry(): 0-1 left[7]
// replicated, 2-9 left[7..0], 10 corner, 11-26 top[0..15], 27 top[15] replicated), G[1][k] = (p'[k] + p'[k+1] + 1) >> 1,
// G[2][k] = (p'[k] + 2 p'[k+1] + p'[k+2] + 2) >> 2 -- so that a predicted sample is ONE byte read, whatever the mode.
struct __attribute__((aligned(16))) OLds {
    union {
        int32_t scr[128];    // (unused by this kernel: the 8x8 column pass exchanges registers between lane pairs)
        int16_t res[256];    // Intra4x4: [blk][sample pair] ; Intra8x8: [blk8][row][column]
    };
    uint8_t T[17 * 32 + 16]; // luma tile, as QLds
    uint8_t TC[2][9 * 16];   // chroma tiles, as QLds
    uint8_t Lcol[16];        // compact left neighbour column (luma)
    uint8_t LcolC[2][8];     // compact left neighbour columns (Cb, Cr)
    uint8_t G[3][32];        // Intra8x8: filtered edge arrays of the block being predicted
    uint8_t Lc8[8];          // Intra8x8: right column of 8x8 block 0 / 2 = left neighbours of block 1 / 3
    uint8_t pad8[8];
    uint8_t SC[2][8 * 24];   // output strip: [plane][chroma row][parked macroblock 0..2] x 8 bytes
#ifdef MVHP_OLDS_PAD
    uint8_t pad[MVHP_OLDS_PAD];   // measurement builds: bank offset between the pictures of a wavefront
#endif
};
#ifndef MVHP_OLDS_PAD
static_assert(sizeof(OLds) == 1888, "OLds layout");
#endif

struct __attribute__((aligned(16))) OTables {
    int      progress[16];
    int      abort_flag;
    int      pad[3];
    int4     q4[52];         // as QTables
    int      ls0[52];
    int      ls8[36];
    uint8_t  qpc[64];
    uint32_t tap4[2 * 9 * 16];
    uint8_t  tap8b[9 * 64];  // Intra8x8 [mode][y][x]: array * 32 + index into OLds::G
};

// tap8b entry: which of the three edge arrays, and where (mode_entry: type 0 = p'[k], 1 = two taps, 2 = three taps)
static __device__ uint8_t tap8b_entry(int mode, int x, int y)
{
    const int e = mode_entry(8, mode, x, y), k = e & 31, t = e >> 5;
    return (uint8_t)(t * 32 + min(k, 27));
}

// (a >> 6, b >> 6) packed: saturate to int16 first, then one packed shift -- exact for the reconstructed sample, as in
// idct4x4_packed (recon_device.h): a value beyond int16 means |r| >= 511, where clip255(pred + r) no longer depends on r
__device__ __forceinline__ int pack_res_shr6(int a, int b)
{
    typedef short short2_t __attribute__((ext_vector_type(2)));
    const short2_t v = __builtin_amdgcn_cvt_pk_i16(a, b);
    return __builtin_bit_cast(int, (short2_t)(v >> (short)6));
}

__device__ __forceinline__ uint32_t lerp_u8(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_lerp(a, b, c); }


#ifndef MVHP_I8_UNROLL
#define MVHP_I8_UNROLL 4   // the four 8x8 blocks of an Intra8x8 macroblock as four copies: positions become constants (-31 % VALU,
                           // -78 % SALU in that loop; High 12.64 -> 12.05 ms); 1 = one loop body
#endif
#define MVHP_PRAGMA_(x) _Pragma(#x)
#define MVHP_UNROLL(n) MVHP_PRAGMA_(unroll n)

#if defined(MVHP_STAMPS)
// Measurement build (tools/stamp_profile.py; results differ in timing only): the shader clock at every section boundary,
// the elapsed cycles added to the accumulator of the boundary that ENDS the interval -- so the interval that ends at mark
// "x_end" is the body of x, the one that ends at a begin mark is what ran since the previous mark.  Accumulators are 32-bit
// scalars (a launch is ~2*10^7 cycles); the s_memtime result is waited for with lgkmcnt(0), which also drains the wave's
// LDS queue at every boundary: sections are charged their own LDS latency, not their successors'.
constexpr const char *kStampNames[] = {"prefetch_wait", "header", "resid_luma", "r_8x8", "r_8x8_end", "r_4x4_dc", "r_4x4", "r_4x4_end",
                                       "resid_store", "resid_chroma", "wait_up", "wait_up_end", "pred_chroma", "pred_luma", "p_i16",
                                       "p_i16_end", "p_i4_setup", "p_i4_chain", "p_i4_end", "p_i8", "p_i8_end", "writeout",
                                       "wo_flush_setup", "wo_planes", "wo_rgb", "wo_short", "wo_park", "wo_end", "neighbours", "publish",
                                       "step_end"};
constexpr int kStampCount = sizeof(kStampNames) / sizeof(kStampNames[0]);
constexpr bool stamp_streq(const char *a, const char *b) { while (*a && *a == *b) { ++a; ++b; } return *a == *b; }
constexpr int stamp_id(const char *n) { for (int i = 0; i < kStampCount; i++) if (stamp_streq(kStampNames[i], n)) return i; return -1; }
__device__ uint32_t g_stamps[256 * 8 * 32];   // [workgroup < 256][wave < 8][boundary < 32] cycles, summed over the launch
#define MVHP_MARK(name)                                                                           \
    do {                                                                                          \
        constexpr int id_ = stamp_id(name);                                                       \
        static_assert(id_ >= 0, "unknown stamp");                                                 \
        const uint32_t t_ = (uint32_t)__builtin_amdgcn_s_memtime();                               \
        st_acc[id_] += t_ - st_prev;                                                              \
        st_prev = t_;                                                                             \
    } while (0)
#elif defined(MVHP_MARKS)   // measurement builds: section markers that survive into the ISA text (tools/isa_sections.py)
#define MVHP_MARK(name) asm volatile("; MARK " name ::: "memory")
#else
#define MVHP_MARK(name)
#endif
// Wave priorities inside a step (round 4; `tools/ab_same_buffers.py`, profiles/r04l_ab_prio*.log).  Two waves share a SIMD's issue
// port; when both have an instruction ready the port takes the higher priority.  The parts of a step that are chains of LDS
// round trips (the two luma chains above all, then the neighbour fetch, chroma / Intra16x16 prediction, the hand-over of the
// neighbour state and the publication the row below waits for) go first, the throughput parts (residuals, colour conversion)
// fill the gaps: High 2048 x 1080p 11.24 -> 10.48 ms, Baseline 8.07 -> 7.75 ms on well-placed buffers (no change where the
// launch waits for memory).  0 everywhere except 2 in the Intra4x4 chain was round 3's setting.
#ifndef MVHP_PRIO_PRED
#define MVHP_PRIO_PRED 1    // from the wait for the row above to the end of luma prediction (outside the two chains)
#endif
#ifndef MVHP_PRIO_TAIL
#define MVHP_PRIO_TAIL 2    // neighbour state, publication, and on through the next record's header until the residual stage starts
#endif
#ifndef MVHP_CHAIN_PRIO
#define MVHP_CHAIN_PRIO 3   // the Intra4x4 chain (measured in round 2: 0 -> 2 = -2 % / -6 % kernel time with / without RGB)
#endif
#ifndef MVHP_I8_PRIO
#define MVHP_I8_PRIO 3      // the four dependent Intra8x8 blocks
#endif

// The compiler is left to the low registers (256 are available at two waves per SIMD; it needs ~190); v216-v247 are
// the record prefetch registers, named only inside inline assembly (see recon_quad.hip and
// tools/check_prefetch_hazard.py, which checks the ISA of every instantiation).
template <int NW, bool RGB>
__global__ __launch_bounds__(NW * 64) void recon_oct_kernel(ReconArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int W = a.width_mbs, H = a.height_mbs;
    OTables &B = *reinterpret_cast<OTables *>(smem);
    uint8_t *lines = smem + sizeof(OTables);              // [octet][ luma W*16 | Cb W*8 | Cr W*8 ]
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane_c = threadIdx.x & 63;
    uint8_t *wave_lds = lines + (size_t)8 * W * 32 + (size_t)wave * 8 * sizeof(OLds);

    // ---- one-time table setup (as recon_quad.hip) ----
    for (int i = threadIdx.x; i < 52; i += NW * 64) {
        const int m = i % 6, s = i / 6;
        const int shl = max(s - 4, 0), shr = max(4 - s, 0), rnd = (1 << shr) >> 1;
        int4 e;
        e.x = (16 * c_v4x4[m * 3 + 0]) << shl;
        e.y = (16 * c_v4x4[m * 3 + 1]) << shl;
        e.z = (16 * c_v4x4[m * 3 + 2]) << shl;
        e.w = shr | (rnd << 8) | (s << 16) | (m << 24);
        B.q4[i] = e;
        B.ls0[i] = 16 * c_v4x4[m * 3 + 0];
    }
    for (int i = threadIdx.x; i < 36; i += NW * 64) B.ls8[i] = 16 * c_v8x8[i];
    for (int i = threadIdx.x; i < 64; i += NW * 64) B.qpc[i] = (uint8_t)((i < 30) ? i : c_qpc[min(i, 51) - 30]);
    for (int i = threadIdx.x; i < 2 * 9 * 16; i += NW * 64) {   // [up-right missing][mode][lane j][upper / lower sample of the lane]
        const int jj = (i >> 1) & 7, half = i & 1;
        B.tap4[i] = tap4_entry((i >> 4) % 9, jj & 3, (jj >> 2) + 2 * half, i >= 9 * 16);
    }
    for (int i = threadIdx.x; i < 9 * 64; i += NW * 64) B.tap8b[i] = tap8b_entry(i >> 6, i & 7, (i >> 3) & 7);
    if (threadIdx.x < 16) B.progress[threadIdx.x] = 0;
    if (threadIdx.x == 16) B.abort_flag = 0;
    __syncthreads();

    const int pitch = W * 16, cpitch = W * 8;
    const uint32_t plane_y = (uint32_t)W * H * 256, plane_c = (uint32_t)W * H * 64;
    const int up_wave = (wave + NW - 1) % NW;

    // this lane's picture; addresses = a scalar base per workgroup + a 32-bit per-lane offset (eight pictures of the
    // largest supported size exceed 4 GiB of RGB: the launcher falls back to the quad kernel beyond 512 Ki macroblocks)
    const int o_c = lane_c >> 3;
    const int frame_raw = (int)blockIdx.x * 8 + o_c;
#if defined(MVHP_ABL_NO_STORES)   // measurement build: everything but the global stores (pictures are not written)
    const bool valid = false;
#else
    const bool valid = frame_raw < a.n_frames;            // a short last workgroup repeats the last picture, stores off
#endif
    const int frame = min(frame_raw, a.n_frames - 1);
    const uint32_t qf = (uint32_t)(frame - (int)blockIdx.x * 8);
    const uint8_t *gpacked = a.packed + (size_t)blockIdx.x * 8 * W * H * MVHP_MB_BYTES;
    uint8_t *gyuv = a.yuv + (size_t)blockIdx.x * 8 * W * H * 384;
    uint8_t *grgb = a.rgb + (size_t)blockIdx.x * 8 * W * H * 768;
    const uint32_t qmb = qf * (uint32_t)(W * H);
#define OPACKED (__umul24(qmb_v, MVHP_MB_BYTES))
#define OYUV (__umul24(qmb_v, 384u))
#define ORGB (__umul24(qmb_v, 768u))

    // Record prefetch, one macroblock ahead, in v216-v247: header (32 B), the lane's two luma blocks (64 B), its
    // chroma block (32 B).  A step issues no store or exactly VM_STRIP stores behind the eight loads (`n_st`).
    constexpr int VM_STRIP = 16 + (RGB ? 24 : 0);   // a full strip: 8 luma rows + 4 chroma rows x 2 planes (+ 8 rows x 3 RGB pieces)
    auto prefetch = [&](int prow, int px, int lane_p) {
        const int jj = lane_p & 7;
        uint32_t qmb_v = qmb;
        asm volatile("" : "+v"(qmb_v));
#if defined(MVHP_ABL_SAME_RECORD)
        const uint32_t rec = OPACKED + (uint32_t)((prow & 1) * W + (px & 7)) * MVHP_MB_BYTES;   // measurement build: cache-resident input
#else
        const uint32_t rec = OPACKED + (uint32_t)(prow * W + px) * MVHP_MB_BYTES;
#endif
        const uint32_t recL = rec + MVHP_MB_HEADER_BYTES + jj * 64;
        const uint32_t recC = rec + MVHP_MB_HEADER_BYTES + (16 + jj) * 32;
        asm volatile("s_nop 4\n\t"
                     "global_load_dwordx4 v[216:219], %0, %3" MVHP_LOAD_HINT "\n\t"
                     "global_load_dwordx4 v[220:223], %0, %3 offset:16" MVHP_LOAD_HINT "\n\t"
                     "global_load_dwordx4 v[224:227], %1, %3" MVHP_LOAD_HINT "\n\t"
                     "global_load_dwordx4 v[228:231], %1, %3 offset:16" MVHP_LOAD_HINT "\n\t"
                     "global_load_dwordx4 v[232:235], %1, %3 offset:32" MVHP_LOAD_HINT "\n\t"
                     "global_load_dwordx4 v[236:239], %1, %3 offset:48" MVHP_LOAD_HINT "\n\t"
                     "global_load_dwordx4 v[240:243], %2, %3" MVHP_LOAD_HINT "\n\t"
                     "global_load_dwordx4 v[244:247], %2, %3 offset:16" MVHP_LOAD_HINT ""
                     : : "v"(rec), "v"(recL), "v"(recC), "s"(gpacked)
                     : "memory", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227",
                       "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239",
                       "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247");
    };
    if (wave < H) prefetch(wave, 0, lane_c);
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");   // (a wave without rows never reads the registers)

    const int up_adj = __builtin_amdgcn_readfirstlane((wave == 0) ? -1 : 0); // wave 0 follows the last wave's previous pass
    int done = 0;  // macroblocks completed by this wave
    int n_st = 0;  // asm stores the previous step issued behind its prefetch (0 also when the compiler counted them)
#if defined(MVHP_STAMPS)
    uint32_t st_acc[kStampCount];
#pragma unroll
    for (int i = 0; i < kStampCount; i++) st_acc[i] = 0;
    uint32_t st_prev = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
    // output strip (see the write-out): the luma rows this lane owns of "its" macroblock of the strip; chroma rows wait in LDS (Q.SC)
    v4i L0 = {0, 0, 0, 0}, L1 = L0, L2 = L0, L3 = L0, L4 = L0, L5 = L0, L6 = L0, L7 = L0;

    for (int row = wave; row < H; row += NW) {
        const int pass = row / NW;
        const int up_base = (pass + up_adj) * W; // MBs the upper wave finished before its row (row-1)
        const bool Bv = row > 0;
#pragma unroll 1
        for (int mbx = 0; mbx < W; mbx++) {
            int lane = lane_c;
            asm volatile("" : "+v"(lane));   // re-materialised per macroblock: keeps lane-dependent addresses out of registers
            const int o = lane >> 3, j = lane & 7;
            OLds &Q = *reinterpret_cast<OLds *>(wave_lds + o * sizeof(OLds));
            uint8_t *line_y = lines + (size_t)o * W * 32;
            uint8_t *line_cb = line_y + W * 16;
            uint8_t *line_cr = line_cb + W * 8;
            const bool A = mbx > 0, C = Bv && (mbx < W - 1), D = A && Bv;

            MVHP_MARK("prefetch_wait");
            // wait for the prefetched record and move it into compiler-visible registers
            v2i w[16];
#define MVHP_WAIT_PREFETCH(N)                                                                                          \
            asm volatile("s_waitcnt vmcnt(%16)\n\t"                                                                    \
                         "v_mov_b64 %0, v[216:217]\n\tv_mov_b64 %1, v[218:219]\n\tv_mov_b64 %2, v[220:221]\n\t"         \
                         "v_mov_b64 %3, v[222:223]\n\tv_mov_b64 %4, v[224:225]\n\tv_mov_b64 %5, v[226:227]\n\t"         \
                         "v_mov_b64 %6, v[228:229]\n\tv_mov_b64 %7, v[230:231]\n\tv_mov_b64 %8, v[232:233]\n\t"         \
                         "v_mov_b64 %9, v[234:235]\n\tv_mov_b64 %10, v[236:237]\n\tv_mov_b64 %11, v[238:239]\n\t"        \
                         "v_mov_b64 %12, v[240:241]\n\tv_mov_b64 %13, v[242:243]\n\tv_mov_b64 %14, v[244:245]\n\t"       \
                         "v_mov_b64 %15, v[246:247]"                                                                   \
                         : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]),    \
                           "=&v"(w[7]), "=&v"(w[8]), "=&v"(w[9]), "=&v"(w[10]), "=&v"(w[11]), "=&v"(w[12]),             \
                           "=&v"(w[13]), "=&v"(w[14]), "=&v"(w[15])                                                    \
                         : "n"(N)                                                                                      \
                         : "memory")
            if (n_st) MVHP_WAIT_PREFETCH(VM_STRIP);
            else MVHP_WAIT_PREFETCH(0);
#undef MVHP_WAIT_PREFETCH
            {   // the record has been moved out of the prefetch registers: request the next macroblock of this wave at
                // once -- same row, or the first of its next row (none left: this one again)
                int nrow = row, nx = mbx + 1;
                if (nx >= W) { nrow = row + NW; nx = 0; }
                if (nrow >= H) { nrow = row; nx = mbx; }
                prefetch(nrow, nx, lane);
            }
            MVHP_MARK("header");
            const uint32_t h0 = (uint32_t)w[0].x, h1 = (uint32_t)w[0].y, nz = (uint32_t)w[1].x;
            const uint32_t m0 = (uint32_t)w[1].y, m1 = (uint32_t)w[2].x, m2 = (uint32_t)w[2].y, m3 = (uint32_t)w[3].x;
            const int kind = h0 & 255;
            const int qpy = min((int)((h0 >> 8) & 255), 51);
            const int cmode = (h0 >> 24) & 255, i16mode = h1 & 255;
            // Intra16x16 at QP'Y == 36 yields a non-zero DC term even from all-zero levels
            // (h264_transform.c:797-808), so the residual stage cannot be skipped there.
            const bool quirk36 = (kind == MVHP_KIND_I16x16) && (qpy == 36) && (a.dc_shift_from > 36);
            const bool need_l = ((nz & 0xffffu) != 0) || quirk36;
            const bool need_c = (nz & 0xff0000u) != 0;
            const bool any_l = __builtin_amdgcn_ballot_w64(need_l) != 0;
            const bool any_c = __builtin_amdgcn_ballot_w64(need_c) != 0;

            // geometry of the lane's two luma blocks 2j (slot 0) and 2j+1 (slot 1): same rows, columns 4 apart
            const int xO0 = ((j >> 1) & 1) << 3;           // slot s: xO0 + 4*s
            const int yO = ((j >> 2) << 3) | ((j & 1) << 2);
            const int obase4 = (lane & 56) << 2;            // byte address (ds_bpermute) of the octet's lane 0

            // =====================================================================================
            // residuals
            // =====================================================================================
            MVHP_MARK("resid_luma");
            __builtin_amdgcn_s_setprio(0);   // the residual stages are the throughput part of a step (priorities: see above)
            int r2[2][8];   // luma blocks 2j, 2j+1: residuals packed by row pairs (see idct4x4_ypairs)
            int c2[8];      // chroma block j
#pragma unroll
            for (int i = 0; i < 8; i++) { r2[0][i] = 0; r2[1][i] = 0; c2[i] = 0; }
            if (any_l) {
                const int4 qt = B.q4[qpy];
                const int shr = qt.w & 255, rnd = (qt.w >> 8) & 255, s = (qt.w >> 16) & 255, m = (qt.w >> 24) & 255;
                if (kind == MVHP_KIND_I8x8) {
                    // ---- luma 8x8 (transform_8x8_residual, h264_transform.c:1205-1383): lane j holds rows
                    //      4*(j&1) .. +3 of 8x8 block j >> 1; rows in registers, columns after an LDS transpose,
                    //      two blocks at a time ----
MVHP_MARK("r_8x8");
                    const int hh = j & 1;
                    const int *l8 = &B.ls8[m * 6];
                    const int ls0 = l8[0], ls1 = l8[1], ls2 = l8[2], ls3 = l8[3], ls4 = l8[4], ls5 = l8[5];
                    // quant8x8 (:1256-1284) as ONE form for both branches: (level * LS' + rnd') >> shr' with
                    // qP >= 36: LS' = LS << (qP/6 - 6), rnd' = 0, shr' = 0;  else: LS' = LS, rnd' = 2^(5 - qP/6), shr' = 6 - qP/6
                    const int shl8 = max(s - 6, 0), shr8 = max(6 - s, 0), rnd8 = (1 << shr8) >> 1;
                    const int q0 = ls0 << shl8, q1 = ls1 << shl8, q2 = ls2 << shl8, q3 = ls3 << shl8, q4 = ls4 << shl8, q5 = ls5 << shl8;
                    int dr[4][8];
#pragma unroll
                    for (int t = 0; t < 4; t++) {   // row 4*hh + t: its class pattern (h264.c:438-446) depends on t only
                        const int pkw[4] = {w[4 + 2 * t].x, w[4 + 2 * t].y, w[5 + 2 * t].x, w[5 + 2 * t].y};
                        const int k0 = (t == 0) ? q0 : (t == 2) ? q4 : q3;   // columns 0, 4
                        const int k1 = (t == 0) ? q3 : (t == 2) ? q5 : q1;   // odd columns
                        const int k2 = (t == 0) ? q4 : (t == 2) ? q2 : q5;   // columns 2, 6
#pragma unroll
                        for (int c = 0; c < 8; c++) {
                            const int ls = (c & 1) ? k1 : ((c & 3) == 0 ? k0 : k2);
                            dr[t][c] = ((c & 1) ? mad_level<1>(pkw[c >> 1], ls, rnd8) : mad_level<0>(pkw[c >> 1], ls, rnd8)) >> shr8;
                        }
                    }
                    if (hh == 0) dr[0][0] += 32; // rounding term of the final (m + 32) >> 6, see idct4x4
#pragma unroll
                    for (int t = 0; t < 4; t++) idct8_1d(dr[t]);
                    // Column pass: the two lanes of a block (rows 0-3 / rows 4-7) swap half of what they hold (DPP, lane ^ 1),
                    // after which the even lane has columns 0-3 and the odd lane columns 4-7 of all eight rows.
                    int colv[4][8];   // [column of this lane's half][row]
#pragma unroll
                    for (int t = 0; t < 4; t++) {
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            // rows 0-3: the even lane's own columns 0-3, or (odd lane) the partner's columns 4-7;
                            // rows 4-7: the partner's columns 0-3 (even lane), or the odd lane's own columns 4-7
                            // (one select with a DPP operand each)
                            const int from_lo = dpp_quad<DPP_XOR1>(dr[t][c]), from_hi = dpp_quad<DPP_XOR1>(dr[t][c + 4]);
                            colv[c][t] = hh ? from_hi : dr[t][c];
                            colv[c][t + 4] = hh ? dr[t][c + 4] : from_lo;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 4; c++) idct8_1d(colv[c]);
                    // res[blk8][row][column] (int16): this lane's four columns of every row = one 8-byte store per row
                    {
                        int16_t *dst = &Q.res[(j >> 1) * 64 + hh * 4];
#pragma unroll
                        for (int i = 0; i < 8; i++)
                            *reinterpret_cast<int2 *>(dst + i * 8) = make_int2(pack_res_shr6(colv[0][i], colv[1][i]),
                                                                               pack_res_shr6(colv[2][i], colv[3][i]));
                    }
MVHP_MARK("r_8x8_end");
                } else {
                    // ---- luma 4x4 (transform_4x4_residual, h264_transform.c:1049-1191), two blocks per lane ----
MVHP_MARK("r_4x4_dc");
                    int dc0 = 0, dc1 = 0;
                    if (kind == MVHP_KIND_I16x16) {
                        // transform_16x16_lumadc, h264_transform.c:756-812 (incl. the `qP > 36` test).  Block 2j+s sits at
                        // DC-matrix column (j1, s) and row (j2, j0): the column transform is half in-lane (s), half
                        // across lanes j ^ 2; the row transform runs across lanes (j0: quad_perm, j2: ds_bpermute).
                        const int d00 = (int)(short)(w[4].x & 0xffff), d01 = (int)(short)(w[8].x & 0xffff);
                        const int S = d00 + d01, Dd = d00 - d01;
                        const int pS = dpp_quad<DPP_XOR2>(S), pD = dpp_quad<DPP_XOR2>(Dd);
                        const bool x1 = (j & 2) != 0;
                        const int u = x1 ? pD : S, v = x1 ? Dd : pS;
                        const int g0 = x1 ? (u - v) : (u + v), g1 = x1 ? (u + v) : (u - v);
                        const int ci = ((j >> 1) & 2) | (j & 1);
                        const int aP = ((lane & 56) | (j & ~5) | ((j >> 2) & 1)) << 2;
                        const int f0 = had4_lanes(g0, dpp_quad<DPP_XOR1>(g0), ci, aP, aP | (4 << 2));
                        const int f1 = had4_lanes(g1, dpp_quad<DPP_XOR1>(g1), ci, aP, aP | (4 << 2));
                        const int lsA = B.ls0[qpy];
                        if (qpy >= a.dc_shift_from) {
                            dc0 = (int)((unsigned)(f0 * lsA) << ((s - 6) & 31));
                            dc1 = (int)((unsigned)(f1 * lsA) << ((s - 6) & 31));
                        } else {
                            dc0 = (int)((unsigned)(f0 * lsA) + (1u << ((5 - s) & 31))) >> ((6 - s) & 31);
                            dc1 = (int)((unsigned)(f1 * lsA) + (1u << ((5 - s) & 31))) >> ((6 - s) & 31);
                        }
                    }
MVHP_MARK("r_4x4");
                    const bool fast = __builtin_amdgcn_ballot_w64(shr != 0) == 0;   // shr = rnd = 0 from qP 24 up
#pragma unroll
                    for (int sl = 0; sl < 2; sl++) {
                        const int pk[8] = {w[4 + 4 * sl].x, w[4 + 4 * sl].y, w[5 + 4 * sl].x, w[5 + 4 * sl].y,
                                           w[6 + 4 * sl].x, w[6 + 4 * sl].y, w[7 + 4 * sl].x, w[7 + 4 * sl].y};
                        int d[16];
                        // quant4x4, h264_transform.c:1100-1134: ((c*LS + rnd) >> shr) << shl, the left shift folded into LS
                        if (fast) {
#pragma unroll
                            for (int i = 0; i < 16; i++) {
                                const int r = i >> 2, c = i & 3;
                                const int ls = ((r & 1) == 0 && (c & 1) == 0) ? qt.x : (((r & 1) && (c & 1)) ? qt.y : qt.z);
                                d[i] = (i & 1) ? mad_level<1>(pk[i >> 1], ls, 0) : mad_level<0>(pk[i >> 1], ls, 0);
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < 16; i++) {
                                const int r = i >> 2, c = i & 3;
                                const int ls = ((r & 1) == 0 && (c & 1) == 0) ? qt.x : (((r & 1) && (c & 1)) ? qt.y : qt.z);
                                d[i] = ((i & 1) ? mad_level<1>(pk[i >> 1], ls, rnd) : mad_level<0>(pk[i >> 1], ls, rnd)) >> shr;
                            }
                        }
                        if (kind == MVHP_KIND_I16x16) d[0] = sl ? dc1 : dc0;
                        d[0] += 32;
                        idct4x4_ypairs(d, r2[sl]);
                    }
                    if (!need_l) {
#pragma unroll
                        for (int i = 0; i < 8; i++) { r2[0][i] = 0; r2[1][i] = 0; }
                    }
MVHP_MARK("r_4x4_end");
                }
            }
            MVHP_MARK("resid_store");
            if (kind == MVHP_KIND_I4x4) {   // lane-per-block -> lane-per-sample-pair goes through LDS (zeros without residual)
                int32_t *dst = reinterpret_cast<int32_t *>(&Q.res[j * 32]);
                *reinterpret_cast<int4 *>(dst) = make_int4(r2[0][0], r2[0][1], r2[0][2], r2[0][3]);
                *reinterpret_cast<int4 *>(dst + 4) = make_int4(r2[0][4], r2[0][5], r2[0][6], r2[0][7]);
                *reinterpret_cast<int4 *>(dst + 8) = make_int4(r2[1][0], r2[1][1], r2[1][2], r2[1][3]);
                *reinterpret_cast<int4 *>(dst + 12) = make_int4(r2[1][4], r2[1][5], r2[1][6], r2[1][7]);
            }
            MVHP_MARK("resid_chroma");
            if (any_c) {
                // ---- chroma 4x4 + transform_2x2_chromadc (h264_transform.c:827-860, :924-936, :988-1005) ----
                const int pl = j >> 2, k = j & 3;
                const int qpi = min(max(qpy + (pl ? a.cqp_off_cr : a.cqp_off_cb), 0), 51);
                const int qpc = B.qpc[qpi];
                const int4 qt = B.q4[qpc];
                const int shr = qt.w & 255, rnd = (qt.w >> 8) & 255, s = (qt.w >> 16) & 255;
                int d[16];
                const int pk[8] = {w[12].x, w[12].y, w[13].x, w[13].y, w[14].x, w[14].y, w[15].x, w[15].y};
                const int d0 = (int)(short)(pk[0] & 0xffff);
                const int c0 = dpp_quad<0x00>(d0), c1 = dpp_quad<0x55>(d0), c2v = dpp_quad<0xAA>(d0), c3 = dpp_quad<0xFF>(d0);
                const int f = (k == 0) ? (c0 + c1 + c2v + c3) : (k == 1) ? (c0 - c1 + c2v - c3)
                            : (k == 2) ? (c0 + c1 - c2v - c3) : (c0 - c1 - c2v + c3);
                const int dc = (int)((unsigned)(f * B.ls0[qpc]) << s) >> 5;
                if (__builtin_amdgcn_ballot_w64(shr != 0) == 0) {
#pragma unroll
                    for (int i = 1; i < 16; i++) {
                        const int r = i >> 2, c = i & 3;
                        const int ls = ((r & 1) == 0 && (c & 1) == 0) ? qt.x : (((r & 1) && (c & 1)) ? qt.y : qt.z);
                        d[i] = (i & 1) ? mad_level<1>(pk[i >> 1], ls, 0) : mad_level<0>(pk[i >> 1], ls, 0);
                    }
                } else {
#pragma unroll
                    for (int i = 1; i < 16; i++) {
                        const int r = i >> 2, c = i & 3;
                        const int ls = ((r & 1) == 0 && (c & 1) == 0) ? qt.x : (((r & 1) && (c & 1)) ? qt.y : qt.z);
                        d[i] = ((i & 1) ? mad_level<1>(pk[i >> 1], ls, rnd) : mad_level<0>(pk[i >> 1], ls, rnd)) >> shr;
                    }
                }
                d[0] = dc + 32;
                idct4x4_ypairs(d, c2);
                if (!need_c) {
#pragma unroll
                    for (int i = 0; i < 8; i++) c2[i] = 0;
                }
            }

            // =====================================================================================
            // wait for the row above: needs columns <= min(mbx+1, W-1); then fetch the top neighbours
            // =====================================================================================
            MVHP_MARK("wait_up");
            __builtin_amdgcn_s_setprio(MVHP_PRIO_PRED);
            if (Bv) {
                const int need = up_base + min(mbx + 2, W);
                int spins = 0;
                while (__hip_atomic_load(&B.progress[up_wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22) || __hip_atomic_load(&B.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        if (lane == 0) { __hip_atomic_store(&B.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicOr(a.err, 1u); }
                        return;
                    }
                }
                asm volatile("" ::: "memory");
                MVHP_MARK("wait_up_end");
                // ten dwords per picture: lanes 0-3 luma top, 4-5 luma up-right (when C), 6-7 Cb top; then lanes 0-1 Cr top
                if (C || (j >> 1) != 2) {
                    uint8_t *dst;
                    const uint8_t *src;
                    if (j < 4) { dst = &Q.T[16 + j * 4]; src = &line_y[mbx * 16 + j * 4]; }
                    else if (j < 6) { dst = &Q.T[32 + (j - 4) * 4]; src = &line_y[mbx * 16 + 16 + (j - 4) * 4]; }
                    else { dst = &Q.TC[0][8 + (j - 6) * 4]; src = &line_cb[mbx * 8 + (j - 6) * 4]; }
                    *reinterpret_cast<uint32_t *>(dst) = *reinterpret_cast<const uint32_t *>(src);
                }
                if (j < 2)
                    *reinterpret_cast<uint32_t *>(&Q.TC[1][8 + j * 4]) = *reinterpret_cast<const uint32_t *>(&line_cr[mbx * 8 + j * 4]);
            }
            WAVE_SYNC();

            // =====================================================================================
            // chroma prediction (h264_intra_prediction.c:2157-2564 + transform4x4_chroma): lane j predicts chroma
            // block j (plane j >> 2, block j & 3) -- all eight lanes of the octet
            // =====================================================================================
            MVHP_MARK("pred_chroma");
            {
                // Branch-free over the modes (round 3).  The eight pictures of a wavefront rarely agree on the mode, so every
                // mode's code ran anyway -- one after the other, each behind its own LDS round trip (the section cost a wave
                // 1250 cycles per step, of which 165 instructions).  Now: the edges are read once, every mode's words are masked by
                // "this lane's mode, and the neighbours it needs exist" (else the prediction stays 0,
                // h264_intra_prediction.c:2157-2323), and OR-ed; only Plane, the long one, is skipped when no picture wants it.
                const int pl = j >> 2, k = j & 3;
                const int cx = (k & 1) * 4, cy = (k >> 1) * 4;
                uint8_t *TCp = Q.TC[pl];
                const uint2 topv = *reinterpret_cast<const uint2 *>(&TCp[8]);         // p[0..7, -1]
                const uint2 lefv = *reinterpret_cast<const uint2 *>(Q.LcolC[pl]);     // p[-1, 0..7]
                const uint32_t cor = TCp[7];
                const uint32_t sA = A ? ~0u : 0u, sB = Bv ? ~0u : 0u;                 // (scalars)
                const uint32_t mD = (cmode == 0) ? ~0u : 0u;
                const uint32_t mH = ((cmode == 1) ? ~0u : 0u) & sA;
                const uint32_t mV = ((cmode == 2) ? ~0u : 0u) & sB;
                const uint32_t mP = ((cmode == 3) ? ~0u : 0u) & sA & sB;
                const uint32_t topw = (k & 1) ? topv.y : topv.x;
                const uint32_t lefw = (k & 2) ? lefv.y : lefv.x;
                // Intra_Chroma_DC (:2338-2441): which sums a block uses depends on its position and on which neighbours
                // exist -- the latter is the same for the eight pictures (scalar branches)
                const int sH = sum4(topw), sV = sum4(lefw);
                int dcv;
                if (A && Bv) {
                    const bool both = (k == 0) || (k == 3);
                    const int one = (k == 1) ? sH : sV;               // block 1 prefers the top, block 2 the left
                    dcv = both ? ((sH + sV + 4) >> 3) : ((one + 2) >> 2);
                } else if (A) dcv = (sV + 2) >> 2;
                else if (Bv) dcv = (sH + 2) >> 2;
                else dcv = 128;
                const uint32_t fix = (topw & mV) | (((uint32_t)dcv * 0x01010101u) & mD);
                const uint32_t lh = lefw & mH;
                uint32_t pw[4];
#pragma unroll
                for (int y = 0; y < 4; y++) pw[y] = fix | __builtin_amdgcn_perm(lh, lh, 0x01010101u * (uint32_t)y);
                if (__builtin_amdgcn_ballot_w64(mP != 0u) != 0) {   // some picture predicts Plane (:2524-2564)
                    const int Hh = plane_grad8(topv, cor), Vv = plane_grad8(lefv, cor);
                    const int aa = 16 * ((int)(lefv.y >> 24) + (int)(topv.y >> 24));
                    const int bb = ((34 * Hh + 32) >> 6) & (int)mP;
                    const int cc = ((34 * Vv + 32) >> 6) & (int)mP;
                    // the other lanes: gradient 0 and a value far below zero -> every sample clips to 0
                    const int v00 = (int)(((uint32_t)(aa + bb * (cx - 3) + cc * (cy - 3) + 16) & mP) | (0x80000000u & ~mP));
#pragma unroll
                    for (int y = 0; y < 4; y++) pw[y] |= plane_row(v00 + cc * y, bb);
                }
                emit_block_ypairs(&TCp[(cy + 1) * 16 + 8 + cx], 16, pw, c2);
            }

            // =====================================================================================
            // luma prediction
            // =====================================================================================
            MVHP_MARK("pred_luma");
            if (kind == MVHP_KIND_I16x16) {
                // h264_intra_prediction.c:1809-2141 + transform16x16_luma; lane j predicts its own two 4x4 blocks
MVHP_MARK("p_i16");
                // Branch-free over the modes, like the chroma prediction above (this section cost a wave 3100 cycles per step with
                // the four modes as four branches, 1000 when the eight pictures happened to agree): edges once, masked words OR-ed
                // (a mode whose neighbours are missing predicts 0, h264_intra_prediction.c:1868-1932), Plane skipped when unused.
                const uint4 topv = *reinterpret_cast<const uint4 *>(&Q.T[16]);     // p[0..15, -1]
                const uint4 lefv = *reinterpret_cast<const uint4 *>(Q.Lcol);       // p[-1, 0..15]
                const uint32_t lrow = *reinterpret_cast<const uint32_t *>(&Q.Lcol[yO]);
                const uint32_t cor = Q.T[15];
                const uint32_t sA = A ? ~0u : 0u, sB = Bv ? ~0u : 0u;              // (scalars)
                const uint32_t mV = ((i16mode == 0) ? ~0u : 0u) & sB;
                const uint32_t mH = ((i16mode == 1) ? ~0u : 0u) & sA;
                const uint32_t mD = (i16mode == 2) ? ~0u : 0u;
                const uint32_t mP = ((i16mode == 3) ? ~0u : 0u) & sA & sB;
                const bool right = (j & 2) != 0;                                   // the lane's eight columns: 0-7 or 8-15
                int sumH = (int)__builtin_amdgcn_sad_u8(topv.x, 0u, 0u), sumV = (int)__builtin_amdgcn_sad_u8(lefv.x, 0u, 0u);
                sumH = (int)__builtin_amdgcn_sad_u8(topv.y, 0u, (uint32_t)sumH); sumV = (int)__builtin_amdgcn_sad_u8(lefv.y, 0u, (uint32_t)sumV);
                sumH = (int)__builtin_amdgcn_sad_u8(topv.z, 0u, (uint32_t)sumH); sumV = (int)__builtin_amdgcn_sad_u8(lefv.z, 0u, (uint32_t)sumV);
                sumH = (int)__builtin_amdgcn_sad_u8(topv.w, 0u, (uint32_t)sumH); sumV = (int)__builtin_amdgcn_sad_u8(lefv.w, 0u, (uint32_t)sumV);
                int dcv;   // Intra_16x16_DC (:2017-2083): the variant is positional, the same for the eight pictures
                if (A && Bv) dcv = (sumH + sumV + 16) >> 5;
                else if (A) dcv = (sumV + 8) >> 4;
                else if (Bv) dcv = (sumH + 8) >> 4;
                else dcv = 128;
                const uint32_t dcw = ((uint32_t)dcv * 0x01010101u) & mD;
                const uint32_t fix0 = ((right ? topv.z : topv.x) & mV) | dcw, fix1 = ((right ? topv.w : topv.y) & mV) | dcw;
                const uint32_t lh = lrow & mH;
                uint32_t pw[2][4];
#pragma unroll
                for (int y = 0; y < 4; y++) {
                    const uint32_t hrow = __builtin_amdgcn_perm(lh, lh, 0x01010101u * (uint32_t)y);
                    pw[0][y] = fix0 | hrow;
                    pw[1][y] = fix1 | hrow;
                }
                if (__builtin_amdgcn_ballot_w64(mP != 0u) != 0) {   // some picture predicts Plane (:2096-2141)
                    const int Hh = plane_grad16(topv, cor), Vv = plane_grad16(lefv, cor);
                    const int aa = 16 * ((int)(lefv.w >> 24) + (int)(topv.w >> 24));
                    const int bb = ((5 * Hh + 32) >> 6) & (int)mP;
                    const int cc = ((5 * Vv + 32) >> 6) & (int)mP;
                    // the other lanes: gradient 0 and a value far below zero -> every sample clips to 0
                    const int v00 = (int)(((uint32_t)(aa + bb * (xO0 - 7) + cc * (yO - 7) + 16) & mP) | (0x80000000u & ~mP));
#pragma unroll
                    for (int y = 0; y < 4; y++) {
                        pw[0][y] |= plane_row(v00 + cc * y, bb);
                        pw[1][y] |= plane_row(v00 + cc * y + 4 * bb, bb);
                    }
                }
                emit_block_ypairs(&Q.T[(yO + 1) * 32 + 16 + xO0], 32, pw[0], r2[0]);
                emit_block_ypairs(&Q.T[(yO + 1) * 32 + 16 + xO0 + 4], 32, pw[1], r2[1]);
MVHP_MARK("p_i16_end");
            } else if (kind == MVHP_KIND_I4x4) {
                // Intra 4x4: 16 dependent block steps; lane j predicts samples (j&3, j>>2) and (j&3, (j>>2)+2) of the block.
                // h264_intra_prediction.c:161-177, :315-483, :496-960 + transform4x4_luma (h264_transform.c:121-156).
MVHP_MARK("p_i4_setup");
                constexpr uint32_t X0 = (1u << 0) | (1u << 2) | (1u << 8) | (1u << 10);   // blocks with xO == 0
                constexpr uint32_t Y0 = (1u << 0) | (1u << 1) | (1u << 4) | (1u << 5);    // blocks with yO == 0
                const uint32_t av_left = A ? 0xffffu : (0xffffu & ~X0);
                const uint32_t av_up = Bv ? 0xffffu : (0xffffu & ~Y0);
                const uint32_t av_upleft = (0xffffu & ~(X0 | Y0)) | (Bv ? ((1u << 1) | (1u << 4) | (1u << 5)) : 0u) |
                                           (A ? ((1u << 2) | (1u << 8) | (1u << 10)) : 0u) | (D ? 1u : 0u);
                const uint32_t av_upright = ((1u << 2) | (1u << 6) | (1u << 8) | (1u << 9) | (1u << 10) | (1u << 12) | (1u << 14)) |
                                            (Bv ? ((1u << 0) | (1u << 1) | (1u << 4)) : 0u) | (C ? (1u << 5) : 0u);
                constexpr uint32_t REQ = (2u << 0) | (1u << 3) | (0u << 6) | (2u << 9) | (7u << 12) | (7u << 15) | (7u << 18) |
                                         (2u << 21) | (1u << 24);
                // control words of blocks 2j and 2j+1, computed by lane j and broadcast inside the octet at their steps:
                // bit 31 the mode is DC, bit 16 prediction allowed, bits 0-15 tap table row offset (bytes)
                uint32_t info[2];
                const uint32_t mw = (j < 2) ? m0 : (j < 4) ? m1 : (j < 6) ? m2 : m3;
#pragma unroll
                for (int sl = 0; sl < 2; sl++) {
                    const int b = 2 * j + sl;
                    const uint32_t mode = (mw >> ((b & 3) * 8)) & 255u;
                    const uint32_t avail = ((av_left >> b) & 1u) | (((av_up >> b) & 1u) << 1) | (((av_upleft >> b) & 1u) << 2);
                    const uint32_t req = (REQ >> (min(mode, 8u) * 3)) & 7u;
                    const uint32_t ok = (((req & ~avail) == 0u) && (mode < 9u)) ? 1u : 0u; // else the prediction stays 0 (:442)
                    const uint32_t trow = (((av_upright >> b) & 1u) ? 0u : 9u) + min(mode, 8u);
                    info[sl] = ((mode == 2u) ? 0x80000000u : 0u) | (ok << 16) | (trow * 64u);
                }
                const int pix = (j >> 2) * 32 + (j & 3);   // the lane's upper sample inside a block, tile units (lower: +64)
                const uint8_t *T = Q.T;
                const uint8_t *tapb = reinterpret_cast<const uint8_t *>(B.tap4) + j * 8;   // the lane's pair of entries inside a row
                const int32_t *res32 = reinterpret_cast<const int32_t *>(Q.res) + j;
                // The 16 blocks run in TEN dependent steps: block (bx, by) only needs blocks decoded at bx + 2*by - 1 or
                // earlier (left, up, up-left and -- where the standard lets it be used at all -- up-right), so the two
                // blocks of an anti-diagonal go together and their LDS round trips overlap.  Control word, table
                // entries and residuals of the next step are fetched before this step's dependent tile reads.
                __builtin_amdgcn_s_setprio(MVHP_CHAIN_PRIO);   // the dependent chain issues few, latency-critical instructions
                constexpr int SA[10] = {0, 1, 2, 3, 6, 7, 10, 11, 14, 15};
                constexpr int SB[10] = {-1, -1, 4, 5, 8, 9, 12, 13, -1, -1};
                struct Ctl { uint32_t inf, ea, eb; int r; };
                auto fetch = [&](const int blk) {
                    Ctl c;
                    c.inf = (uint32_t)__builtin_amdgcn_ds_bpermute(obase4 + (blk >> 1) * 4, (int)info[blk & 1]);
                    const uint2 e = *reinterpret_cast<const uint2 *>(tapb + (c.inf & 0xffffu));   // one 8-byte read: upper, lower sample
                    c.ea = e.x;
                    c.eb = e.y;
                    c.r = res32[blk * 8];
                    return c;
                };
                auto predict = [&](const int blk, const Ctl &c) -> uint32_t {   // byte 0: upper sample, byte 1: lower
                    const int bxO = (((blk >> 2) & 1) << 3) | ((blk & 1) << 2);
                    const int byO = ((blk >> 3) << 3) | (((blk >> 1) & 1) << 2);
                    const int base = (byO + 1) * 32 + 16 + bxO;     // tile index of the block's top-left sample
                    const uint32_t okmask = (uint32_t)(((int)(c.inf << 15)) >> 31);   // bit 16 -> 0 / all ones
                    const int a0 = T[base - 33 + (int)(c.ea & 255)], b0 = T[base - 33 + (int)((c.ea >> 8) & 255)], c0 = T[base - 33 + (int)(c.ea >> 16)];
                    const int a1 = T[base - 33 + (int)(c.eb & 255)], b1 = T[base - 33 + (int)((c.eb >> 8) & 255)], c1 = T[base - 33 + (int)(c.eb >> 16)];
                    // both samples in one word from here on (one mask, one select instead of two)
                    uint32_t pp = ((uint32_t)((a0 + 2 * b0 + c0 + 2) >> 2) | ((uint32_t)((a1 + 2 * b1 + c1 + 2) >> 2) << 16)) & okmask;
                    const bool isdc = (int)c.inf < 0;
                    if (__builtin_amdgcn_ballot_w64(isdc) != 0) { // some picture predicts DC
                        // which neighbours exist is positional, i.e. the same for the eight pictures: scalar branches
                        const bool bl = (bxO > 0) || A, bu = (byO > 0) || Bv;
                        int dcv = 128;
                        if (bl && bu) {
                            const int sumH = sum4(*reinterpret_cast<const uint32_t *>(&T[base - 32]));
                            const int sumV = T[base - 1] + T[base + 31] + T[base + 63] + T[base + 95];
                            dcv = (sumH + sumV + 4) >> 3;
                        } else if (bl) {
                            dcv = (T[base - 1] + T[base + 31] + T[base + 63] + T[base + 95] + 2) >> 2;
                        } else if (bu) {
                            dcv = (sum4(*reinterpret_cast<const uint32_t *>(&T[base - 32])) + 2) >> 2;
                        }
                        pp = isdc ? (uint32_t)dcv * 0x00010001u : pp;
                    }
                    return sat_pk_u8(pk_add_sat((int)pp, c.r));
                };
                auto put = [&](const int blk, const uint32_t two) {
                    const int bxO = (((blk >> 2) & 1) << 3) | ((blk & 1) << 2);
                    const int byO = ((blk >> 3) << 3) | (((blk >> 1) & 1) << 2);
                    const int base = (byO + 1) * 32 + 16 + bxO;
                    Q.T[base + pix] = (uint8_t)two;
                    Q.T[base + pix + 64] = (uint8_t)(two >> 8);
                };
MVHP_MARK("p_i4_chain");
                Ctl nA = fetch(0), nB = nA;
#pragma unroll
                for (int t = 0; t < 10; t++) {
                    const Ctl cA = nA, cB = nB;
                    if (t < 9) {
                        nA = fetch(SA[t + 1]);
                        if (SB[t + 1] >= 0) nB = fetch(SB[t + 1]);
                    }
                    const uint32_t twoA = predict(SA[t], cA);
                    uint32_t twoB = 0;
                    if (SB[t] >= 0) twoB = predict(SB[t], cB);
                    put(SA[t], twoA);
                    if (SB[t] >= 0) put(SB[t], twoB);
                    WAVE_SYNC();
                }
                __builtin_amdgcn_s_setprio(MVHP_PRIO_PRED);
MVHP_MARK("p_i4_end");
            } else {
                // Intra 8x8: h264_intra_prediction.c:1107-1353 (edge filter) + :1366-1793 + transform8x8_luma;
                // lane j predicts row j of the block
MVHP_MARK("p_i8");
                __builtin_amdgcn_s_setprio(MVHP_I8_PRIO);   // the Intra8x8 blocks are a dependent chain as well
                MVHP_UNROLL(MVHP_I8_UNROLL)
                for (int blk = 0; blk < 4; blk++) {
                    const int bxO = (blk & 1) * 8, byO = (blk >> 1) * 8;
                    const int mode = (int)((m0 >> (blk * 8)) & 255u);
                    const bool left = (bxO > 0) || A;
                    const bool up = (byO > 0) || Bv;
                    const bool upleft = (bxO > 0) ? ((byO > 0) || Bv) : ((byO > 0) ? A : D);
                    const bool upright = (blk == 0) ? Bv : (blk == 1) ? C : (blk == 2);
                    // ---- raw edge, the same for the eight lanes of the picture: 16 samples above (tile row byO), the
                    //      corner, 8 samples to the left (compact columns) ----
                    const uint8_t *Trow = &Q.T[byO * 32 + 16 + bxO];
                    const uint2 tA = *reinterpret_cast<const uint2 *>(Trow);
                    uint2 tB = *reinterpret_cast<const uint2 *>(Trow + 8);
                    const uint32_t cw = *reinterpret_cast<const uint32_t *>(Trow - 4);            // byte 3 = p[-1,-1]
                    const uint2 lf = *reinterpret_cast<const uint2 *>(bxO ? Q.Lc8 : &Q.Lcol[byO]); // p[-1,0..7]
                    if (!upright) tB.x = tB.y = __builtin_amdgcn_perm(0u, tA.y, 0x03030303u);      // p[8..15,-1] = p[7,-1] (:1230-1236)
                    // the unified edge sequence (28 entries, ends replicated) as seven words
                    uint32_t Wd[7];
                    Wd[0] = __builtin_amdgcn_perm(0u, lf.y, 0x02030303u);                          // L7 L7 L7 L6
                    Wd[1] = __builtin_amdgcn_perm(lf.y, lf.x, 0x02030405u);                        // L5 L4 L3 L2
                    Wd[2] = __builtin_amdgcn_perm(cw, __builtin_amdgcn_perm(tA.x, lf.x, 0x040c0001u), 0x03070100u); // L1 L0 C T0
                    Wd[3] = __builtin_amdgcn_alignbyte(tA.y, tA.x, 1);                             // T1..T4
                    Wd[4] = __builtin_amdgcn_alignbyte(tB.x, tA.y, 1);                             // T5..T8
                    Wd[5] = __builtin_amdgcn_alignbyte(tB.y, tB.x, 1);                             // T9..T12
                    Wd[6] = __builtin_amdgcn_perm(0u, tB.y, 0x03030201u);                          // T13 T14 T15 T15
                    // Intra_8x8_sample_filtering (:1295-1353): p'[k] = (s[k-1] + 2 s[k] + s[k+1] + 2) >> 2 on that sequence
                    // = lerp_up(lerp_down(s[k-1], s[k+1]), s[k]) per byte; where the corner or a whole side is missing the
                    // neighbour is the sample itself (the reference's special cases), patched into word 2
                    const uint32_t selP = 0x03020100u + (left ? 0u : 0x00040000u) + (upleft ? 0u : 0x04000000u);
                    const uint32_t selN = 0x03020100u + (upleft ? 0u : 0x00000400u) + (up ? 0u : 0x00040000u);
                    uint32_t Ed[8];
#pragma unroll
                    for (int k = 0; k < 7; k++) {
                        uint32_t Pk = __builtin_amdgcn_alignbyte(Wd[k], Wd[k ? k - 1 : 0], 3);
                        uint32_t Nk = __builtin_amdgcn_alignbyte(Wd[k < 6 ? k + 1 : 6], Wd[k], 1);
                        if (k == 2) {
                            Pk = __builtin_amdgcn_perm(Wd[2], Pk, selP);
                            Nk = __builtin_amdgcn_perm(Wd[2], Nk, selN);
                        }
                        Ed[k] = lerp_u8(lerp_u8(Pk, Nk, 0u), Wd[k], 0x01010101u);
                    }
                    Ed[0] = __builtin_amdgcn_perm(0u, Ed[0], 0x03020202u);   // entries 0, 1 = p'[-1,7] replicated
                    Ed[6] = __builtin_amdgcn_perm(0u, Ed[6], 0x02020100u);   // entry 27 = p'[15,-1] replicated
                    Ed[7] = __builtin_amdgcn_perm(0u, Ed[6], 0x03030303u);
                    // lane k < 7 publishes words k of the three arrays
                    {
                        const bool b0 = (j & 1) != 0, b1 = (j & 2) != 0, b2 = (j & 4) != 0;
                        const uint32_t a0 = b0 ? Ed[1] : Ed[0], a1 = b0 ? Ed[3] : Ed[2], a2 = b0 ? Ed[5] : Ed[4], a3 = b0 ? Ed[7] : Ed[6];
                        const uint32_t n0 = b0 ? Ed[2] : Ed[1], n1 = b0 ? Ed[4] : Ed[3], n2 = b0 ? Ed[6] : Ed[5], n3 = Ed[7];
                        const uint32_t c0 = b1 ? a1 : a0, c1 = b1 ? a3 : a2, d0 = b1 ? n1 : n0, d1 = b1 ? n3 : n2;
                        const uint32_t Ej = b2 ? c1 : c0, En = b2 ? d1 : d0;
                        const uint32_t NE = __builtin_amdgcn_alignbyte(En, Ej, 1), N2 = __builtin_amdgcn_alignbyte(En, Ej, 2);
                        *reinterpret_cast<uint32_t *>(&Q.G[0][j * 4]) = Ej;
                        *reinterpret_cast<uint32_t *>(&Q.G[1][j * 4]) = lerp_u8(Ej, NE, 0x01010101u);
                        *reinterpret_cast<uint32_t *>(&Q.G[2][j * 4]) = lerp_u8(lerp_u8(Ej, N2, 0u), NE, 0x01010101u);
                    }
                    WAVE_SYNC();
                    {
                        const int y = j;
                        // which neighbours a mode needs (bit 0 left, 1 up, 2 up-left), three bits per mode: the same
                        // requirements as Intra4x4 (h264_intra_prediction.c:1366-1793 test them mode by mode)
                        constexpr uint32_t REQ = (2u << 0) | (1u << 3) | (0u << 6) | (2u << 9) | (7u << 12) | (7u << 15) | (7u << 18) |
                                                 (2u << 21) | (1u << 24);
                        const uint32_t avail = (left ? 1u : 0u) | (up ? 2u : 0u) | (upleft ? 4u : 0u);
                        const uint32_t mm = min((uint32_t)mode, 8u);
                        const bool ok = (((REQ >> (mm * 3u)) & 7u & ~avail) == 0u) && ((uint32_t)mode < 9u);   // else the prediction stays 0
                        // Intra_8x8_DC (:1435-1500) on the filtered samples: entries 2..9 (left), 11..18 (top)
                        int dcv = 128;
                        if (__builtin_amdgcn_ballot_w64(mode == 2) != 0)   // some picture predicts DC
                        {
                            const int sumV = sum4(Ed[0] & 0xffff0000u) + sum4(Ed[1]) + sum4(Ed[2] & 0x0000ffffu);
                            const int sumH = sum4(Ed[2] & 0xff000000u) + sum4(Ed[3]) + sum4(Ed[4] & 0x00ffffffu);
                            if (left && up) dcv = (sumH + sumV + 8) >> 4;
                            else if (left) dcv = (sumV + 4) >> 3;
                            else if (up) dcv = (sumH + 4) >> 3;
                        }
                        const uint2 tb = *reinterpret_cast<const uint2 *>(&B.tap8b[mm * 64 + y * 8]);
                        const uint8_t *Gb = &Q.G[0][0];
                        const uint32_t s0 = Gb[tb.x & 255u], s1 = Gb[(tb.x >> 8) & 255u], s2 = Gb[(tb.x >> 16) & 255u], s3 = Gb[tb.x >> 24];
                        const uint32_t s4 = Gb[tb.y & 255u], s5 = Gb[(tb.y >> 8) & 255u], s6 = Gb[(tb.y >> 16) & 255u], s7 = Gb[tb.y >> 24];
                        const uint32_t dcw = (uint32_t)dcv * 0x01010101u;
                        const uint32_t okm = ok ? 0xffffffffu : 0u;
                        const uint32_t pwa = (mode == 2) ? dcw : ((s0 | (s1 << 8) | (s2 << 16) | (s3 << 24)) & okm);   // samples 0..3 of row y
                        const uint32_t pwb = (mode == 2) ? dcw : ((s4 | (s5 << 8) | (s6 << 16) | (s7 << 24)) & okm);   // samples 4..7
                        int4 rr = make_int4(0, 0, 0, 0);
                        if (need_l) rr = *reinterpret_cast<const int4 *>(&Q.res[blk * 64 + y * 8]);
                        const uint32_t oa = sat_pk_u8(pk_add_sat((int)__builtin_amdgcn_perm(0u, pwa, 0x0c010c00u), rr.x)) |
                                            (sat_pk_u8(pk_add_sat((int)__builtin_amdgcn_perm(0u, pwa, 0x0c030c02u), rr.y)) << 16);
                        const uint32_t ob = sat_pk_u8(pk_add_sat((int)__builtin_amdgcn_perm(0u, pwb, 0x0c010c00u), rr.z)) |
                                            (sat_pk_u8(pk_add_sat((int)__builtin_amdgcn_perm(0u, pwb, 0x0c030c02u), rr.w)) << 16);
                        *reinterpret_cast<uint2 *>(&Q.T[(byO + y + 1) * 32 + 16 + bxO]) = make_uint2(oa, ob);
                        if (bxO == 0) Q.Lc8[y] = (uint8_t)(ob >> 24);   // left neighbours of the block to the right
                    }
                    WAVE_SYNC();
                }
                __builtin_amdgcn_s_setprio(MVHP_PRIO_PRED);
MVHP_MARK("p_i8_end");
            }
            WAVE_SYNC();
            // I_PCM (8.3.5; only MVHP_STREAM_SPEC streams carry it): the samples as they are, over whatever
            // the prediction paths above made of such a record.  Record layout (minivideo_hotpath.h): the lane's 64 bytes hold
            // luma rows 2j and 2j+1, Cb row j, Cr row j.
            if (__builtin_amdgcn_ballot_w64(kind == MVHP_KIND_IPCM) != 0) {
                if (kind == MVHP_KIND_IPCM) {
                    *reinterpret_cast<int4 *>(&Q.T[(2 * j + 1) * 32 + 16]) = make_int4(w[4].x, w[4].y, w[5].x, w[5].y);
                    *reinterpret_cast<int4 *>(&Q.T[(2 * j + 2) * 32 + 16]) = make_int4(w[6].x, w[6].y, w[7].x, w[7].y);
                    *reinterpret_cast<int2 *>(&Q.TC[0][(j + 1) * 16 + 8]) = make_int2(w[8].x, w[8].y);
                    *reinterpret_cast<int2 *>(&Q.TC[1][(j + 1) * 16 + 8]) = make_int2(w[9].x, w[9].y);
                }
                WAVE_SYNC();
            }

            // =====================================================================================
            // write-out (mb_to_rgb, export_utils.c:209-324, fused): park, or flush the 4-macroblock strip
            // =====================================================================================
            // Strip ownership by MACROBLOCK: lane (m, h) = (j & 3, j >> 2) of an octet keeps, of macroblock m of the 4-macroblock
            // strip, luma rows 4i + 2h, 4i + 2h + 1 (i < 4) -- pairs that share chroma row 2i + h -- and writes them when the
            // strip is complete: in one store instruction lanes m = 0..3 then cover 64 contiguous bytes of a luma row
            // (32 of a chroma row), and the 192 RGB bytes of a row leave in three consecutive instructions.
            MVHP_MARK("writeout");
            __builtin_amdgcn_s_setprio(0);
            {
                const int mbi = mbx & 3;
                const int m_own = j & 3, h_own = j >> 2;
                n_st = 0;
                if (m_own == mbi) {   // this macroblock's owners take its luma rows out of the tile
                    const uint8_t *t0 = &Q.T[(2 * h_own + 1) * 32 + 16];
                    L0 = *reinterpret_cast<const v4i *>(t0);            L1 = *reinterpret_cast<const v4i *>(t0 + 32);
                    L2 = *reinterpret_cast<const v4i *>(t0 + 4 * 32);   L3 = *reinterpret_cast<const v4i *>(t0 + 5 * 32);
                    L4 = *reinterpret_cast<const v4i *>(t0 + 8 * 32);   L5 = *reinterpret_cast<const v4i *>(t0 + 9 * 32);
                    L6 = *reinterpret_cast<const v4i *>(t0 + 12 * 32);  L7 = *reinterpret_cast<const v4i *>(t0 + 13 * 32);
                }
                if (mbi == 3 || mbx == W - 1) {
                    MVHP_MARK("wo_flush_setup");
                    uint32_t qmb_v = qmb;
                    asm volatile("" : "+v"(qmb_v));
                    // chroma rows 2i + h of the lane's macroblock: parked ones from the strip, the current one from the tile
                    const bool cur = (m_own == mbi);
                    const uint8_t *cb_src = cur ? &Q.TC[0][(h_own + 1) * 16 + 8] : &Q.SC[0][h_own * 24 + m_own * 8];
                    const uint8_t *cr_src = cur ? &Q.TC[1][(h_own + 1) * 16 + 8] : &Q.SC[1][h_own * 24 + m_own * 8];
                    const int cstep = cur ? 32 : 48;
                    const uint2 cb0 = *reinterpret_cast<const uint2 *>(cb_src), cb1 = *reinterpret_cast<const uint2 *>(cb_src + cstep);
                    const uint2 cb2 = *reinterpret_cast<const uint2 *>(cb_src + 2 * cstep), cb3 = *reinterpret_cast<const uint2 *>(cb_src + 3 * cstep);
                    const uint2 cr0 = *reinterpret_cast<const uint2 *>(cr_src), cr1 = *reinterpret_cast<const uint2 *>(cr_src + cstep);
                    const uint2 cr2 = *reinterpret_cast<const uint2 *>(cr_src + 2 * cstep), cr3 = *reinterpret_cast<const uint2 *>(cr_src + 3 * cstep);
                    const uint32_t x0 = (uint32_t)((mbx & ~3) * 16 + m_own * 16);
                    const uint32_t lrow = (uint32_t)((row * 16 + 2 * h_own) * pitch) + x0;     // luma row 2h of the macroblock row
                    const uint32_t pl = OYUV + lrow;
                    const uint32_t pcb = OYUV + plane_y + (uint32_t)((row * 8 + h_own) * cpitch) + (x0 >> 1), pcr = pcb + plane_c;
                    const uint32_t p4 = 4u * (uint32_t)pitch, c2 = 2u * (uint32_t)cpitch;
                    if (mbi == 3) {
                        // ---- full strip: exactly VM_STRIP store instructions ----
#define MVHP_ST(ADDR, DATA, BASE, OFF) MVHP_ST_(ADDR, DATA, BASE, OFF)
#define MVHP_ST_(ADDR, DATA, BASE, OFF) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:" #OFF MVHP_STORE_HINT "\n\ts_nop 1" : : "v"(ADDR), "v"(DATA), "s"(BASE) : "memory")
#define MVHP_ST2(ADDR, DATA, BASE) asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2" MVHP_STORE_HINT : : "v"(ADDR), "v"(DATA), "s"(BASE) : "memory")
                        MVHP_MARK("wo_planes");
#if defined(MVHP_SPREAD_STORES)   // measurement build: the plane stores of a row pair behind its RGB stores instead of all sixteen up front
                        const bool planes_first = !RGB;
#else
                        const bool planes_first = true;
#endif
                        if (valid && planes_first) {
                            MVHP_ST(pl, L0, gyuv, 0);            MVHP_ST(pl + pitch, L1, gyuv, 0);
                            MVHP_ST(pl + p4, L2, gyuv, 0);       MVHP_ST(pl + p4 + pitch, L3, gyuv, 0);
                            MVHP_ST(pl + 2 * p4, L4, gyuv, 0);   MVHP_ST(pl + 2 * p4 + pitch, L5, gyuv, 0);
                            MVHP_ST(pl + 3 * p4, L6, gyuv, 0);   MVHP_ST(pl + 3 * p4 + pitch, L7, gyuv, 0);
                            const v2i b0 = {(int)cb0.x, (int)cb0.y}, b1 = {(int)cb1.x, (int)cb1.y}, b2 = {(int)cb2.x, (int)cb2.y}, b3 = {(int)cb3.x, (int)cb3.y};
                            const v2i q0 = {(int)cr0.x, (int)cr0.y}, q1 = {(int)cr1.x, (int)cr1.y}, q2 = {(int)cr2.x, (int)cr2.y}, q3 = {(int)cr3.x, (int)cr3.y};
                            MVHP_ST2(pcb, b0, gyuv);            MVHP_ST2(pcr, q0, gyuv);
                            MVHP_ST2(pcb + c2, b1, gyuv);       MVHP_ST2(pcr + c2, q1, gyuv);
                            MVHP_ST2(pcb + 2 * c2, b2, gyuv);   MVHP_ST2(pcr + 2 * c2, q2, gyuv);
                            MVHP_ST2(pcb + 3 * c2, b3, gyuv);   MVHP_ST2(pcr + 3 * c2, q3, gyuv);
                        }
                        if (RGB) {
                            MVHP_MARK("wo_rgb");
                            const uint32_t prgb = ORGB + lrow * 3u;
                            // a row pair of the lane's macroblock against the chroma row it shares (export_utils.c:278-279)
#define MVHP_RGB_OUT(YQA, YQB, CB, CR, I)                                                                              \
                            {                                                                                          \
                                v4i a0, a1, a2, c0, c1, c2;                                                            \
                                rgb16x2(make_uint4((uint32_t)(YQA).x, (uint32_t)(YQA).y, (uint32_t)(YQA).z, (uint32_t)(YQA).w), \
                                        make_uint4((uint32_t)(YQB).x, (uint32_t)(YQB).y, (uint32_t)(YQB).z, (uint32_t)(YQB).w), \
                                        CB, CR, a0, a1, a2, c0, c1, c2);                                               \
                                const uint32_t pa = prgb + (I) * 3u * p4, pb = pa + 3u * (uint32_t)pitch;               \
                                if (valid) {                                                                           \
                                    MVHP_ST(pa, a0, grgb, 0); MVHP_ST(pa, a1, grgb, 16); MVHP_ST(pa, a2, grgb, 32);     \
                                    MVHP_ST(pb, c0, grgb, 0); MVHP_ST(pb, c1, grgb, 16); MVHP_ST(pb, c2, grgb, 32);     \
                                } else {                                                                               \
                                    asm volatile("" : : "v"(a0), "v"(a1), "v"(a2), "v"(c0), "v"(c1), "v"(c2));         \
                                }                                                                                      \
                            }
#if defined(MVHP_SPREAD_STORES)
#define MVHP_PLANES_OUT(LA, LB, CBV, CRV, I)                                                                          \
                            if (valid) {                                                                               \
                                const v2i cbq = {(int)(CBV).x, (int)(CBV).y}, crq = {(int)(CRV).x, (int)(CRV).y};      \
                                MVHP_ST(pl + (I) * p4, LA, gyuv, 0); MVHP_ST(pl + (I) * p4 + pitch, LB, gyuv, 0);      \
                                MVHP_ST2(pcb + (I) * c2, cbq, gyuv); MVHP_ST2(pcr + (I) * c2, crq, gyuv);              \
                            }
#else
#define MVHP_PLANES_OUT(LA, LB, CBV, CRV, I)
#endif
                            MVHP_RGB_OUT(L0, L1, cb0, cr0, 0u) MVHP_PLANES_OUT(L0, L1, cb0, cr0, 0u)
                            MVHP_RGB_OUT(L2, L3, cb1, cr1, 1u) MVHP_PLANES_OUT(L2, L3, cb1, cr1, 1u)
                            MVHP_RGB_OUT(L4, L5, cb2, cr2, 2u) MVHP_PLANES_OUT(L4, L5, cb2, cr2, 2u)
                            MVHP_RGB_OUT(L6, L7, cb3, cr3, 3u) MVHP_PLANES_OUT(L6, L7, cb3, cr3, 3u)
#undef MVHP_PLANES_OUT
#undef MVHP_RGB_OUT
                        }
#undef MVHP_ST
#undef MVHP_ST_
#undef MVHP_ST2
#if defined(MVHP_ABL_NO_STORES)
                        n_st = a.n_frames < 0 ? VM_STRIP : 0;   // (0 at run time; both forms of the wait stay in the code)
#else
                        n_st = VM_STRIP;
#endif
                    } else if (m_own <= mbi && valid) {
                        MVHP_MARK("wo_short");
                        // ---- short strip at the right picture edge (W % 4 != 0): compiler-counted stores ----
                        const v4i Lr[8] = {L0, L1, L2, L3, L4, L5, L6, L7};
                        const uint2 cbr[4] = {cb0, cb1, cb2, cb3}, crr[4] = {cr0, cr1, cr2, cr3};
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            *reinterpret_cast<v4i *>(gyuv + pl + i * p4) = Lr[2 * i];
                            *reinterpret_cast<v4i *>(gyuv + pl + i * p4 + pitch) = Lr[2 * i + 1];
                            *reinterpret_cast<uint2 *>(gyuv + pcb + i * c2) = cbr[i];
                            *reinterpret_cast<uint2 *>(gyuv + pcr + i * c2) = crr[i];
                            if (RGB) {
                                v4i a0, a1, a2, c0, c1, c2v;
                                rgb16x2(make_uint4((uint32_t)Lr[2 * i].x, (uint32_t)Lr[2 * i].y, (uint32_t)Lr[2 * i].z, (uint32_t)Lr[2 * i].w),
                                        make_uint4((uint32_t)Lr[2 * i + 1].x, (uint32_t)Lr[2 * i + 1].y, (uint32_t)Lr[2 * i + 1].z, (uint32_t)Lr[2 * i + 1].w),
                                        cbr[i], crr[i], a0, a1, a2, c0, c1, c2v);
                                const uint32_t prgb = ORGB + lrow * 3u;
                                v4i *dst = reinterpret_cast<v4i *>(grgb + prgb + i * 3u * p4);
                                dst[0] = a0; dst[1] = a1; dst[2] = a2;
                                dst = reinterpret_cast<v4i *>(grgb + prgb + i * 3u * p4 + 3u * (uint32_t)pitch);
                                dst[0] = c0; dst[1] = c1; dst[2] = c2v;
                            }
                        }
                    }
                } else {
                    MVHP_MARK("wo_park");
                    // ---- park the chroma rows (lane j: row j of both planes) in the LDS strip ----
                    const uint2 cvb = *reinterpret_cast<const uint2 *>(&Q.TC[0][(j + 1) * 16 + 8]);
                    const uint2 cvr = *reinterpret_cast<const uint2 *>(&Q.TC[1][(j + 1) * 16 + 8]);
                    *reinterpret_cast<uint2 *>(&Q.SC[0][j * 24 + mbi * 8]) = cvb;
                    *reinterpret_cast<uint2 *>(&Q.SC[1][j * 24 + mbi * 8]) = cvr;
                }
            }

            // =====================================================================================
            // neighbour state for the next macroblock / next row, then publish
            // =====================================================================================
            MVHP_MARK("wo_end");
            MVHP_MARK("neighbours");
            __builtin_amdgcn_s_setprio(MVHP_PRIO_TAIL);
            {
                // left columns: lane j luma rows j, j+8 and chroma row j of both planes; corners (old top-right sample) by
                // lanes 0-2; bottom rows -> line buffer: lanes 0-3 luma, 4-5 Cb, 6-7 Cr (one dword each)
                const uint8_t kla = Q.T[(j + 1) * 32 + 31], klb = Q.T[(j + 9) * 32 + 31];
                const uint8_t kcb = Q.TC[0][(j + 1) * 16 + 15], kcr = Q.TC[1][(j + 1) * 16 + 15];
                uint8_t kk = 0;
                uint8_t *kdst = &Q.T[15];
                if (j == 0) kk = Q.T[31];
                else if (j == 1) { kk = Q.TC[0][15]; kdst = &Q.TC[0][7]; }
                else if (j == 2) { kk = Q.TC[1][15]; kdst = &Q.TC[1][7]; }
                uint32_t bot;
                uint8_t *bdst;
                if (j < 4) { bot = *reinterpret_cast<const uint32_t *>(&Q.T[16 * 32 + 16 + j * 4]); bdst = &line_y[mbx * 16 + j * 4]; }
                else if (j < 6) { bot = *reinterpret_cast<const uint32_t *>(&Q.TC[0][8 * 16 + 8 + (j - 4) * 4]); bdst = &line_cb[mbx * 8 + (j - 4) * 4]; }
                else { bot = *reinterpret_cast<const uint32_t *>(&Q.TC[1][8 * 16 + 8 + (j - 6) * 4]); bdst = &line_cr[mbx * 8 + (j - 6) * 4]; }
                WAVE_SYNC();
                Q.T[(j + 1) * 32 + 15] = kla;
                Q.T[(j + 9) * 32 + 15] = klb;
                Q.Lcol[j] = kla;
                Q.Lcol[j + 8] = klb;
                Q.TC[0][(j + 1) * 16 + 7] = kcb;
                Q.TC[1][(j + 1) * 16 + 7] = kcr;
                Q.LcolC[0][j] = kcb;
                Q.LcolC[1][j] = kcr;
                if (j < 3) *kdst = kk;
                *reinterpret_cast<uint32_t *>(bdst) = bot;
            }
            MVHP_MARK("publish");
            done++;
            // LDS operations of one wave complete in order; the explicit wait makes the line-buffer
            // writes land before the counter without waiting for the global plane stores (vmcnt).
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&B.progress[wave], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            WAVE_SYNC();
            MVHP_MARK("step_end");
        }
    }
#if defined(MVHP_STAMPS)
    if (lane_c == 0 && blockIdx.x < 256 && wave < 8) {
#pragma unroll
        for (int i = 0; i < kStampCount; i++) g_stamps[((int)blockIdx.x * 8 + wave) * 32 + i] = st_acc[i];
    }
#endif
}

#undef OPACKED
#undef OYUV
#undef ORGB

#if defined(MVHP_STAMPS)
extern "C" __attribute__((visibility("default"))) int mvhp_debug_read_stamps(uint32_t *out, const char **names, int *count)
{
    if (count) *count = kStampCount;
    if (names) for (int i = 0; i < kStampCount; i++) names[i] = kStampNames[i];
    if (!out) return 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(uint32_t) * 256 * 8 * 32) == hipSuccess ? 1 : 0;
}
#endif

size_t recon_oct_lds_bytes(int width_mbs, int nw)
{
    return sizeof(OTables) + (size_t)8 * width_mbs * 32 + (size_t)nw * 8 * sizeof(OLds);
}

template <int NW, bool RGB>
static hipError_t launch_oct_one(const ReconArgs &a, hipStream_t stream)
{
    const size_t lds = recon_oct_lds_bytes(a.width_mbs, NW);
    const int groups = (a.n_frames + 7) / 8;
    hipError_t e = hipFuncSetAttribute((const void *)recon_oct_kernel<NW, RGB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((recon_oct_kernel<NW, RGB>), dim3(groups), dim3(NW * 64), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_recon_oct(const ReconArgs &a, int nw, hipStream_t stream)
{
    const bool rgb = a.rgb != nullptr;
    switch (nw) {
    case 4: return rgb ? launch_oct_one<4, true>(a, stream) : launch_oct_one<4, false>(a, stream);
    case 6: return rgb ? launch_oct_one<6, true>(a, stream) : launch_oct_one<6, false>(a, stream);
    case 8: return rgb ? launch_oct_one<8, true>(a, stream) : launch_oct_one<8, false>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace mvhp
