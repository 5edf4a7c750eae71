// recon_pipe.hip -- the low-latency form of the reconstruction kernel (gfx950 only): a macroblock row is worked on by THREE
// wavefronts in a pipeline, and a picture's rows are spread over several workgroups.
//
//   recon_pipe_kernel<R, RGB>  same contract as recon_quad_kernel (recon_quad.hip): replaces intra_prediction_process()
//                           (decoder/h264/h264_intra_prediction.c:112-145), all of h264_transform.c, the planar gather of
//                           export.c:65-188 and mb_to_rgb() (export_utils.c:209-324) for whole pictures.
//
// Why.  A picture's critical path is W + 2 (H - 1) macroblock steps (254 for 1080p: a row may start a macroblock when the row
// above is two ahead), so a small batch -- config 5's 64 pictures per GPU, a thumbnail job -- is as fast as ONE step is short,
// however many CUs idle.  In recon_quad_kernel a step is 5.4 us for a wave alone on its SIMD: residual arithmetic, the
// dependent prediction chain, colour conversion and stores one after the other.  Only the prediction depends on the
// neighbours.  Here a row has
//   F  the residual wave: reads the packed records (the only wave that loads from HBM), dequantises, inverse-transforms and
//      leaves header + residuals of macroblock x in slot x % 4 of a ring in LDS -- no dependency on anything but a free slot;
//   K  the luma wave: waits for the row above (or the seam), predicts luma into tile x % 3, adds the residuals, hands the
//      bottom row / right column on, publishes -- the chain every other row waits for, and nothing else;
//   O  the chroma + output wave: chroma prediction is independent of luma and needs no up-right neighbour, so it runs here,
//      one macroblock behind the row above's O; then the finished macroblock is parked in the four-macroblock strip,
//      converted to RGB and stored (the only wave that writes HBM).
// The three run on one SIMD (waves w, w + R, w + 2R of the workgroup), so K's LDS round trips are filled with F's and O's
// arithmetic.  Four pictures per wavefront, 16 lanes each, exactly as recon_quad.hip (lane j owns luma 4x4 block j and, for
// j < 8, chroma block j); a workgroup = one band of R = 1, 2 or 4 rows = 3 R wavefronts; bands, tickets and seams as
// recon_quad_kernel<.., WIDE> (the seam format is the same; a column's granules 0-3 are K's, 4-7 are O's).
// Measured (profiles/r04f_*): 64 x 1080p Baseline 0.97 ms (recon_rows_kernel in bands: 1.28), 16 pictures 0.78, one picture 0.58.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "minivideo_hotpath.h"
#include "recon_kernels.h"
#include "recon_device.h"
#include "recon_batch_device.h"

#ifndef MVHP_PIPE_PRIO_K
#define MVHP_PIPE_PRIO_K 3   // wave priority of the luma wave (K), and of the write-out wave (O); the residual wave (F) stays at 0
                            // (K 2 / O 0 before: K1p1 64 x 1080p High 0.951 -> 0.924 ms, 150 pictures 1.78 -> 1.70; K1p -1 %)
#endif
#ifndef MVHP_PIPE_PRIO_O
#define MVHP_PIPE_PRIO_O 2
#endif

namespace mvhp {

#define MVHP_PRAGMA_(x) _Pragma(#x)
#define MVHP_UNROLL(n) MVHP_PRAGMA_(unroll n)

#ifndef MVHP_PIPE_NAP_F
#define MVHP_PIPE_NAP_F 8   // s_sleep units between polls of the residual wave (it runs ahead) ...
#endif
#ifndef MVHP_PIPE_NAP_O
#define MVHP_PIPE_NAP_O 3   // ... and of the chroma + output wave (it runs behind)
#endif
#ifndef MVHP_WIDE_NAP
#define MVHP_WIDE_NAP 1   // s_sleep units between the luma wave's polls of the row above
#endif
constexpr int PIPE_ROWS_MAX = 4;    // rows per band = rows per workgroup: 1, 2 or 4 (three wavefronts each)
#ifndef MVHP_PIPE_SLOTS
#define MVHP_PIPE_SLOTS 4           // F -> K / O ring: F may run this many macroblocks ahead of the slower of K and O
#endif
#ifndef MVHP_PIPE_TILES
#define MVHP_PIPE_TILES 3           // K -> O ring: K may run this many macroblocks ahead of O (whose strip flush is bursty)
#endif
constexpr int NSLOT = MVHP_PIPE_SLOTS, NTILE = MVHP_PIPE_TILES;

// F -> K: one macroblock of one picture
struct __attribute__((aligned(16))) PSlot {
    uint32_t hdr[8];       // h0, h1, nz_mask, pred modes 0-3 | 4-7 | 8-11 | 12-15, -
    int16_t  res[256];     // luma residuals: Intra4x4 / Intra16x16 [blk][sample pairs row-major] (lane-per-block form), Intra8x8
                           // [blk8][column][row]; I_PCM: the 256 + 128 raw samples as the record holds them (32 B per lane)
    int32_t  c2[64];       // chroma residuals: lane j < 8 -> eight packed pairs
};
static_assert(sizeof(PSlot) == 800, "PSlot layout");

// K -> O: the luma of the macroblock under construction / just finished (as QLds::T)
struct __attribute__((aligned(16))) PTile {
    uint8_t T[17 * 32 + 16];   // row 0 = top neighbours (bytes 16..31, up-right 32..39), byte 15 of rows 0..16 = corner / left column
};
static_assert(sizeof(PTile) == 560, "PTile layout");

struct __attribute__((aligned(16))) PRow {   // one macroblock row of the band, four pictures
    PSlot   slot[NSLOT][4];
    int32_t scr[4][128];       // F: Intra8x8 transpose scratch
    PTile   tile[NTILE][4];
    uint8_t Lcol[4][16];       // K: compact left neighbour column (luma)
    uint8_t E8[4][32];         // K: filtered Intra8x8 edge
    uint8_t TC[4][2][9 * 16];  // O: chroma tiles (as QLds::TC): row 0 = top; byte 7 = corner / left column, 8..15 samples
    uint8_t LcolC[4][2][8];    // O: compact left neighbour columns (Cb, Cr)
    uint8_t SC[4][2][8 * 24];  // O: chroma rows of the three parked macroblocks of a strip
};

struct __attribute__((aligned(16))) PCtl {
    int f_done[PIPE_ROWS_MAX];     // macroblocks whose residuals F has left in the ring
    int k_done[PIPE_ROWS_MAX];     // macroblocks K has finished (what the row below and O wait for; frees F's slot)
    int o_done[PIPE_ROWS_MAX];     // macroblocks O has taken out of their tile and whose ring slot it has read (frees both)
    int c_done[PIPE_ROWS_MAX];     // macroblocks whose chroma O has finished (what the row below's O waits for)
    int abort_flag;
    int unit;
    int pad[2];
};
static_assert(sizeof(PCtl) % 16 == 0, "PCtl layout");

#if defined(MVHP_PIPE_STAMPS)
// measurement build (tools/build_variant.sh pstamps -DMVHP_PIPE_STAMPS; tools/pipe_stamps.py): where a K wave's step goes --
// the shader clock at section boundaries, summed over the launch.  [section][0] = cycles, [section][1] = visits
__device__ unsigned long long g_pipe_stamps[16][2];
#define PSTAMP(k)                                                                                  \
    do {                                                                                           \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        ps_acc[k] += t_ - ps_prev;                                                                 \
        ps_n[k]++;                                                                                 \
        ps_prev = t_;                                                                              \
    } while (0)
#else
#define PSTAMP(k)
#endif

// spin until *ctr >= need; false = give up (error word set)
// NAP: s_sleep units (64 clocks) between polls -- 1 on the luma chain, longer for the waves that run ahead of or behind it: a
// polling wave spends vector-ALU issue slots (compare, branch) that the luma wave on the same SIMD wants
template <int NAP = 1>
__device__ __forceinline__ bool pipe_wait(const int *ctr, int need, PCtl &C, uint32_t *err, int lane)
{
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
        __builtin_amdgcn_s_sleep(NAP);
        if (++spins > (1 << 22) || __hip_atomic_load(&C.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            if (lane == 0) { __hip_atomic_store(&C.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicOr(err, 1u); }
            return false;
        }
    }
    asm volatile("" ::: "memory");
    return true;
}

template <int PIPE_ROWS, bool RGB>
__global__ __launch_bounds__(PIPE_ROWS * 3 * 64) void recon_pipe_kernel(ReconArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int W = a.width_mbs, H = a.height_mbs;
    QTables &B = *reinterpret_cast<QTables *>(smem);
    PCtl &C = *reinterpret_cast<PCtl *>(smem + sizeof(QTables));
    uint8_t *lines = smem + sizeof(QTables) + sizeof(PCtl);   // [quarter][ luma W*16 | Cb W*8 | Cr W*8 ]
    PRow *rows = reinterpret_cast<PRow *>(lines + (size_t)4 * W * 32);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int role = wave / PIPE_ROWS, r = wave - role * PIPE_ROWS;     // role 0 = F, 1 = K, 2 = O of row r of the band
    if (MVHP_PIPE_PRIO_O && role == 2) __builtin_amdgcn_s_setprio(MVHP_PIPE_PRIO_O);
    const int lane_c = threadIdx.x & 63;
    constexpr int NT = PIPE_ROWS * 3 * 64;

    if (threadIdx.x == 0) C.unit = (int)(atomicAdd(a.wide_ticket, 1u) - a.wide_base);
    // ---- one-time table setup (as recon_quad.hip) ----
    for (int i = threadIdx.x; i < 52; i += NT) {
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
    for (int i = threadIdx.x; i < 36; i += NT) B.ls8[i] = 16 * c_v8x8[i];
    for (int i = threadIdx.x; i < 64; i += NT) B.qpc[i] = (uint8_t)((i < 30) ? i : c_qpc[min(i, 51) - 30]);
    for (int i = threadIdx.x; i < 2 * 9 * 16; i += NT)
        B.tap4[i] = tap4_entry((i >> 4) % 9, i & 3, (i >> 2) & 3, i >= 9 * 16);
    for (int i = threadIdx.x; i < 9 * 64; i += NT) B.tap8[i] = tap8_entry(i >> 6, i & 7, (i >> 3) & 7);
    if (threadIdx.x < PIPE_ROWS_MAX) { C.f_done[threadIdx.x] = 0; C.k_done[threadIdx.x] = 0; C.o_done[threadIdx.x] = 0; C.c_done[threadIdx.x] = 0; }
    if (threadIdx.x == 16) C.abort_flag = 0;
    __syncthreads();

    // this workgroup's pictures and band (band-major tickets: see recon_rows_kernel)
    const int groups = (a.n_frames + 3) / 4;
    const int bands = (H + PIPE_ROWS - 1) / PIPE_ROWS;
    const int unit = __builtin_amdgcn_readfirstlane(C.unit);
    const int band = unit / groups;
    const int grp = unit - band * groups;
    if ((unsigned)unit >= (unsigned)(bands * groups)) {   // a ticket outside the launch: the host's bookkeeping of the counter is off
        if (threadIdx.x == 0) atomicOr(a.err, 2u);
        return;
    }
    const int row = band * PIPE_ROWS + r;
    if (row >= H) return;                            // the three waves of a row beyond the picture
    PRow &R = rows[r];

    const int pitch = W * 16, cpitch = W * 8;
    const uint32_t plane_y = (uint32_t)W * H * 256, plane_c = (uint32_t)W * H * 64;
    const int q_c = lane_c >> 4;
    const int frame_raw = grp * 4 + q_c;
    const bool valid = frame_raw < a.n_frames;            // a short last group repeats the last picture, stores off
    const int frame = min(frame_raw, a.n_frames - 1);
    const uint32_t qf = (uint32_t)(frame - grp * 4);
    const uint32_t qmb = qf * (uint32_t)(W * H);
    const bool Bv = row > 0;

    if (role == 0) {
        // =========================================================================================================
        // F: records -> residuals (transform_4x4_residual / transform_8x8_residual / the DC transforms of h264_transform.c)
        // =========================================================================================================
        const uint8_t *gpacked = a.packed + (size_t)grp * 4 * W * H * MVHP_MB_BYTES;
        const int j_c = lane_c & 15;
        auto rec_of = [&](int x) { return __umul24(qmb, MVHP_MB_BYTES) + (uint32_t)(row * W + x) * MVHP_MB_BYTES; };
        int4 nH0, nH1, nLA, nLB, nCA, nCB;
        auto load_rec = [&](int x) {
            const uint32_t rec = rec_of(x);
            const uint8_t *p = gpacked + rec;
            nH0 = *reinterpret_cast<const int4 *>(p);
            nH1 = *reinterpret_cast<const int4 *>(p + 16);
            const uint8_t *pl = p + MVHP_MB_HEADER_BYTES + j_c * 32;
            nLA = *reinterpret_cast<const int4 *>(pl);
            nLB = *reinterpret_cast<const int4 *>(pl + 16);
            const uint8_t *pc = p + MVHP_MB_HEADER_BYTES + ((j_c < 8) ? (16 + j_c) : j_c) * 32;
            nCA = *reinterpret_cast<const int4 *>(pc);
            nCB = *reinterpret_cast<const int4 *>(pc + 16);
        };
        load_rec(0);
#pragma unroll 1
        for (int mbx = 0; mbx < W; mbx++) {
            int lane = lane_c;
            asm volatile("" : "+v"(lane));
            const int q = lane >> 4, j = lane & 15;
            const int4 cH0 = nH0, cH1 = nH1, cLA = nLA, cLB = nLB, cCA = nCA, cCB = nCB;
            if (mbx + 1 < W) load_rec(mbx + 1);
            PSlot &S = R.slot[mbx % NSLOT][q];
            int32_t *scr = R.scr[q];
            const uint32_t h0 = (uint32_t)cH0.x, nz = (uint32_t)cH0.z;
            const int kind = h0 & 255;
            const int qpy = min((int)((h0 >> 8) & 255), 51);
            // Intra16x16 at QP'Y == 36 yields a non-zero DC term even from all-zero levels
            // (h264_transform.c:797-808), so the residual stage cannot be skipped there.
            const bool quirk36 = (kind == MVHP_KIND_I16x16) && (qpy == 36) && (a.dc_shift_from > 36);
            const bool need_l = ((nz & 0xffffu) != 0) || quirk36;
            const bool need_c = (nz & 0xff0000u) != 0;
            const bool any_l = __builtin_amdgcn_ballot_w64(need_l) != 0;
            const bool any_c = __builtin_amdgcn_ballot_w64(need_c) != 0;

            // the slot is free once K and O have finished with macroblock mbx - NSLOT
            if (!pipe_wait<MVHP_PIPE_NAP_F>(&C.k_done[r], mbx - NSLOT + 1, C, a.err, lane)) return;
            if (!pipe_wait<MVHP_PIPE_NAP_F>(&C.o_done[r], mbx - NSLOT + 1, C, a.err, lane)) return;

            int r2[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // luma block j: packed int16 pairs, row-major
            int c2[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // chroma block j (j < 8)
            bool res_written = false;               // Intra8x8 with residual: S.res holds the column form already
            if (any_l) {
                const int4 qt = B.q4[qpy];
                const int shr = qt.w & 255, rnd = (qt.w >> 8) & 255, s = (qt.w >> 16) & 255, m = (qt.w >> 24) & 255;
                if (kind == MVHP_KIND_I8x8) {
                    // ---- luma 8x8 (transform_8x8_residual, h264_transform.c:1205-1383): lane j holds rows
                    //      (2i, 2i+1), i = j & 3, of 8x8 block j >> 2; rows in registers, columns after an LDS
                    //      transpose, two blocks at a time ----
                    const int r0 = (j & 3) * 2;
                    const int *l8 = &B.ls8[m * 6];
                    const int4 l8a = make_int4(l8[0], l8[1], l8[2], l8[3]);
                    const int2 l8b = make_int2(l8[4], l8[5]);
                    // classes (h264.c:438-446) of the even row: r0%4==0 -> (0,3,4) else (4,5,2); odd row: (3,1,5)
                    const bool r4 = (r0 & 2) == 0;
                    const int e0 = r4 ? l8a.x : l8b.x, e1 = r4 ? l8a.w : l8b.y, e2 = r4 ? l8b.x : l8a.z;
                    const int o0 = l8a.w, o1 = l8a.y, o2 = l8b.y;
                    int d0[8], d1[8];
                    unpack8(cLA, d0);
                    unpack8(cLB, d1);
                    if (qpy > 35) {
                        const int sh = (s - 6) & 31;
#pragma unroll
                        for (int c = 0; c < 8; c++) {
                            const int le = (c & 1) ? e1 : ((c & 3) == 0 ? e0 : e2), lo = (c & 1) ? o1 : ((c & 3) == 0 ? o0 : o2);
                            d0[c] = (int)((unsigned)(d0[c] * le) << sh);
                            d1[c] = (int)((unsigned)(d1[c] * lo) << sh);
                        }
                    } else {
                        const int rn = 1 << ((5 - s) & 31), sh = (6 - s) & 31;
#pragma unroll
                        for (int c = 0; c < 8; c++) {
                            const int le = (c & 1) ? e1 : ((c & 3) == 0 ? e0 : e2), lo = (c & 1) ? o1 : ((c & 3) == 0 ? o0 : o2);
                            d0[c] = (d0[c] * le + rn) >> sh;
                            d1[c] = (d1[c] * lo + rn) >> sh;
                        }
                    }
                    if (r0 == 0) d0[0] += 32; // rounding term of the final (m + 32) >> 6, see idct4x4
                    idct8_1d(d0);
                    idct8_1d(d1);
                    int col[2][8];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        if ((j >> 3) == h) {
                            int32_t *dst = &scr[((j >> 2) & 1) * 64 + r0 * 8];
                            *reinterpret_cast<int4 *>(dst) = make_int4(d0[0], d0[1], d0[2], d0[3]);
                            *reinterpret_cast<int4 *>(dst + 4) = make_int4(d0[4], d0[5], d0[6], d0[7]);
                            *reinterpret_cast<int4 *>(dst + 8) = make_int4(d1[0], d1[1], d1[2], d1[3]);
                            *reinterpret_cast<int4 *>(dst + 12) = make_int4(d1[4], d1[5], d1[6], d1[7]);
                        }
                        WAVE_SYNC();
#pragma unroll
                        for (int i = 0; i < 8; i++) col[h][i] = scr[(j >> 3) * 64 + i * 8 + (j & 7)];
                        idct8_1d(col[h]);
                        WAVE_SYNC();
                    }
                    // res[blk8][column][row]: lane j column j & 7 of block 2h + (j >> 3); zeros when the macroblock has no residual
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        int4 o;
                        o.x = need_l ? pack_res(col[h][0] >> 6, col[h][1] >> 6) : 0;
                        o.y = need_l ? pack_res(col[h][2] >> 6, col[h][3] >> 6) : 0;
                        o.z = need_l ? pack_res(col[h][4] >> 6, col[h][5] >> 6) : 0;
                        o.w = need_l ? pack_res(col[h][6] >> 6, col[h][7] >> 6) : 0;
                        *reinterpret_cast<int4 *>(&S.res[(2 * h + (j >> 3)) * 64 + (j & 7) * 8]) = o;
                    }
                    res_written = true;
                } else if (kind != MVHP_KIND_IPCM) {
                    // ---- luma 4x4 (transform_4x4_residual, h264_transform.c:1049-1191) ----
                    int d[16];
                    const int pk[8] = {cLA.x, cLA.y, cLA.z, cLA.w, cLB.x, cLB.y, cLB.z, cLB.w};   // two levels per word
                    int dc = 0;
                    if (kind == MVHP_KIND_I16x16) {
                        // transform_16x16_lumadc, h264_transform.c:756-812 (incl. the `qP > 36` test): the 16 DC
                        // levels sit one per lane; rows/columns of the 4x4 DC matrix are lane bits (3,1) / (2,0)
                        const int d0 = (int)(short)(pk[0] & 0xffff);
                        const int base = (lane & 48);
                        const int cj = ((j >> 1) & 2) | (j & 1), ci = ((j >> 2) & 2) | ((j >> 1) & 1);
                        const int aP = (base | (j & ~5) | ((j >> 2) & 1)) << 2;
                        const int g = had4_lanes(d0, dpp_quad<DPP_XOR1>(d0), cj, aP, aP | (4 << 2));
                        const int bP = (base | (j & ~10) | ((j >> 2) & 2)) << 2;
                        const int f = had4_lanes(g, dpp_quad<DPP_XOR2>(g), ci, bP, bP | (8 << 2));
                        const int lsA = B.ls0[qpy];
                        if (qpy >= a.dc_shift_from) dc = (int)((unsigned)(f * lsA) << ((s - 6) & 31));
                        else dc = (int)((unsigned)(f * lsA) + (1u << ((5 - s) & 31))) >> ((6 - s) & 31);
                    }
                    // quant4x4, h264_transform.c:1100-1134: ((c*LS + rnd) >> shr) << shl, the left shift folded into LS;
                    // shr = rnd = 0 from qP 24 up (checked for the whole wave)
                    if (__builtin_amdgcn_ballot_w64(shr != 0) == 0) {
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const int rr = i >> 2, c = i & 3;
                            const int ls = ((rr & 1) == 0 && (c & 1) == 0) ? qt.x : (((rr & 1) && (c & 1)) ? qt.y : qt.z);
                            d[i] = (i & 1) ? mad_level<1>(pk[i >> 1], ls, 0) : mad_level<0>(pk[i >> 1], ls, 0);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const int rr = i >> 2, c = i & 3;
                            const int ls = ((rr & 1) == 0 && (c & 1) == 0) ? qt.x : (((rr & 1) && (c & 1)) ? qt.y : qt.z);
                            d[i] = ((i & 1) ? mad_level<1>(pk[i >> 1], ls, rnd) : mad_level<0>(pk[i >> 1], ls, rnd)) >> shr;
                        }
                    }
                    if (kind == MVHP_KIND_I16x16) d[0] = dc;
                    d[0] += 32;
                    idct4x4_packed(d, r2);
                    if (!need_l) {
#pragma unroll
                        for (int i = 0; i < 8; i++) r2[i] = 0;
                    }
                }
            }
            if (kind == MVHP_KIND_IPCM) {
                // I_PCM (8.3.5; MVHP_STREAM_SPEC streams only): the lane's 32 record bytes as they are (K copies them into the tile)
                *reinterpret_cast<int4 *>(&S.res[j * 16]) = cLA;
                *reinterpret_cast<int4 *>(&S.res[j * 16 + 8]) = cLB;
            } else if (!res_written) {
                // Intra4x4 / Intra16x16: lane-per-block form (K's Intra4x4 chain reads it sample by sample); an Intra8x8
                // macroblock without residual in a step where nobody has one: zeros
                *reinterpret_cast<int4 *>(&S.res[j * 16]) = make_int4(r2[0], r2[1], r2[2], r2[3]);
                *reinterpret_cast<int4 *>(&S.res[j * 16 + 8]) = make_int4(r2[4], r2[5], r2[6], r2[7]);
            }
            if (any_c) {
                // ---- chroma 4x4 + transform_2x2_chromadc (h264_transform.c:827-860, :924-936, :988-1005) ----
                const int pl = (j >> 2) & 1, k = j & 3;
                const int qpi = min(max(qpy + (pl ? a.cqp_off_cr : a.cqp_off_cb), 0), 51);
                const int qpc = B.qpc[qpi];
                const int4 qt = B.q4[qpc];
                const int shr = qt.w & 255, rnd = (qt.w >> 8) & 255, s = (qt.w >> 16) & 255;
                int d[16];
                const int pk[8] = {cCA.x, cCA.y, cCA.z, cCA.w, cCB.x, cCB.y, cCB.z, cCB.w};   // two levels per word
                const int d0 = (int)(short)(pk[0] & 0xffff);
                const int c0 = dpp_quad<0x00>(d0), c1 = dpp_quad<0x55>(d0), c2v = dpp_quad<0xAA>(d0), c3 = dpp_quad<0xFF>(d0);
                const int f = (k == 0) ? (c0 + c1 + c2v + c3) : (k == 1) ? (c0 - c1 + c2v - c3)
                            : (k == 2) ? (c0 + c1 - c2v - c3) : (c0 - c1 - c2v + c3);
                const int dc = (int)((unsigned)(f * B.ls0[qpc]) << s) >> 5;
                if (__builtin_amdgcn_ballot_w64(shr != 0) == 0) {
#pragma unroll
                    for (int i = 1; i < 16; i++) {
                        const int rr = i >> 2, c = i & 3;
                        const int ls = ((rr & 1) == 0 && (c & 1) == 0) ? qt.x : (((rr & 1) && (c & 1)) ? qt.y : qt.z);
                        d[i] = (i & 1) ? mad_level<1>(pk[i >> 1], ls, 0) : mad_level<0>(pk[i >> 1], ls, 0);
                    }
                } else {
#pragma unroll
                    for (int i = 1; i < 16; i++) {
                        const int rr = i >> 2, c = i & 3;
                        const int ls = ((rr & 1) == 0 && (c & 1) == 0) ? qt.x : (((rr & 1) && (c & 1)) ? qt.y : qt.z);
                        d[i] = ((i & 1) ? mad_level<1>(pk[i >> 1], ls, rnd) : mad_level<0>(pk[i >> 1], ls, rnd)) >> shr;
                    }
                }
                d[0] = dc + 32;
                idct4x4_packed(d, c2);
                if (!need_c || kind == MVHP_KIND_IPCM) {
#pragma unroll
                    for (int i = 0; i < 8; i++) c2[i] = 0;
                }
            }
            if (j < 8) {
                *reinterpret_cast<int4 *>(&S.c2[j * 8]) = make_int4(c2[0], c2[1], c2[2], c2[3]);
                *reinterpret_cast<int4 *>(&S.c2[j * 8 + 4]) = make_int4(c2[4], c2[5], c2[6], c2[7]);
            }
            if (j == 0) {
                *reinterpret_cast<int4 *>(&S.hdr[0]) = cH0;
                *reinterpret_cast<int4 *>(&S.hdr[4]) = cH1;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&C.f_done[r], mbx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            WAVE_SYNC();
        }
        return;
    }

    if (role == 1) {
        // =========================================================================================================
        // K: luma prediction + residual add (h264_intra_prediction.c:112-2141, transform*_luma of h264_transform.c), luma neighbour
        //    state, the row dependency
        // =========================================================================================================
        const bool seam_in = (r == 0) && band > 0;                       // top neighbours of this row come from the seam above
        const bool seam_out = (r == PIPE_ROWS - 1) && (row + 1 < H);     // this row's bottom samples feed the seam below
        const uint32_t seam_pic = (uint32_t)((bands - 1) * W * SEAM_GRANULES);   // granules per picture
        const unsigned long long *seam_rd = seam_in ? a.seam + ((size_t)grp * 4 * (bands - 1) + (band - 1)) * W * SEAM_GRANULES : nullptr;
        unsigned long long *seam_wr = seam_out ? a.seam + ((size_t)grp * 4 * (bands - 1) + band) * W * SEAM_GRANULES : nullptr;
        unsigned long long seam_pend = 0;
        __builtin_amdgcn_s_setprio(MVHP_PIPE_PRIO_K);   // the chain every other row waits for
#if defined(MVHP_PIPE_STAMPS)
        unsigned long long ps_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ps_prev = __builtin_amdgcn_s_memtime();
        unsigned ps_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#pragma unroll 1
        for (int mbx = 0; mbx < W; mbx++) {
            int lane = lane_c;
            asm volatile("" : "+v"(lane));
            const int q = lane >> 4, j = lane & 15;
            const PSlot &S = R.slot[mbx % NSLOT][q];
            PTile &Tl = R.tile[mbx % NTILE][q];
            PTile &Tn = R.tile[(mbx + 1) % NTILE][q];    // where the next macroblock of the row will be built
            uint8_t *T = Tl.T;
            uint8_t *line_y = lines + (size_t)q * W * 32;
            const bool A = mbx > 0, Cav = Bv && (mbx < W - 1), D = A && Bv;

            if (seam_in && (mbx & 1) == 0) {
                // Macroblocks mbx and mbx + 1 read luma columns <= mbx + 2 of the row above.  The first step of a row fetches
                // columns 0..3 now (two rounds); every later even step finds (mbx + 1, mbx + 2) asked for two steps ago, and asks
                // for (mbx + 3, mbx + 4).  Lane j < 8 of a quarter: column c0 + (j >> 2), luma granule j & 3 (the chroma granules
                // 4-7 of a column are O's).
                const int g = j & 3;
                const uint32_t lo = qf * seam_pic + (uint32_t)g;
                int c0 = mbx ? mbx + 1 : 0;
                for (int round = mbx ? 1 : 0; round < 2; round++, c0 += 2) {
                    const int col = c0 + ((j >> 2) & 1);
                    const bool act = (j < 8) && (col < W);
                    const unsigned long long *src = seam_rd + lo + (uint32_t)((act ? col : 0) * SEAM_GRANULES);
                    unsigned long long v = seam_pend;
                    if (mbx == 0) v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    int spins = 0;
                    while (__builtin_amdgcn_ballot_w64(act && (uint32_t)(v >> 32) != a.wide_epoch) != 0) {
                        __builtin_amdgcn_s_sleep(2);
                        // bounded; a failure anywhere in the launch (error word) ends every wait
                        bool stop = ++spins > (1 << 20) || __hip_atomic_load(&C.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (!stop && (spins & 255) == 0) stop = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                        if (stop) {
                            if (lane == 0) { __hip_atomic_store(&C.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicOr(a.err, 1u); }
                            return;
                        }
                        v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (act) *reinterpret_cast<uint32_t *>(&line_y[col * 16 + g * 4]) = (uint32_t)v;
                }
                const int ncol = mbx + 3 + ((j >> 2) & 1);
                if (j < 8 && ncol < W) seam_pend = __hip_atomic_load(seam_rd + lo + (uint32_t)(ncol * SEAM_GRANULES), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                WAVE_SYNC();
            }

            PSTAMP(0);   // seam
            // the tile is free once O has taken macroblock mbx - NTILE out of it; the residuals of mbx must be in the ring
            // (the left column / corner K wrote into this tile one step ago are not O's bytes)
            if (!pipe_wait(&C.o_done[r], mbx - NTILE + 1, C, a.err, lane)) return;
            if (!pipe_wait(&C.f_done[r], mbx + 1, C, a.err, lane)) return;
            PSTAMP(1);   // waits for O and F
            const int4 hA = *reinterpret_cast<const int4 *>(&S.hdr[0]);
            const int4 hB = *reinterpret_cast<const int4 *>(&S.hdr[4]);
            const uint32_t h0 = (uint32_t)hA.x, h1 = (uint32_t)hA.y;
            const uint32_t m0 = (uint32_t)hA.w, m1 = (uint32_t)hB.x, m2 = (uint32_t)hB.y, m3 = (uint32_t)hB.z;
            const int kind = h0 & 255;
            const int i16mode = h1 & 255;
            const int xO = (((j >> 2) & 1) << 3) | ((j & 1) << 2);
            const int yO = ((j >> 3) << 3) | (((j >> 1) & 1) << 2);

            // wait for the row above: needs columns <= min(mbx+1, W-1); then fetch the top neighbours
            if (Bv) {
                if (!seam_in && !pipe_wait<MVHP_WIDE_NAP>(&C.k_done[r - 1], min(mbx + 2, W), C, a.err, lane)) return;
                // lanes 0-3 luma top, 4-5 luma up-right (when C): one dword each
                if (j < 6 && (Cav || j < 4)) {
                    uint8_t *dst = (j < 4) ? &T[16 + j * 4] : &T[32 + (j - 4) * 4];
                    *reinterpret_cast<uint32_t *>(dst) = *reinterpret_cast<const uint32_t *>(&line_y[mbx * 16 + j * 4]);
                }
            }
            WAVE_SYNC();
            PSTAMP(2);   // header, wait for the row above, top fetch

            // ---- luma prediction ----
            if (kind == MVHP_KIND_I16x16) {
                // h264_intra_prediction.c:1809-2141 + transform16x16_luma; lane j predicts its own 4x4 block
                const int4 ra = *reinterpret_cast<const int4 *>(&S.res[j * 16]), rb = *reinterpret_cast<const int4 *>(&S.res[j * 16 + 8]);
                const int r2[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
                uint32_t pw[4] = {0u, 0u, 0u, 0u};
                if (i16mode == 0) {
                    if (Bv) { const uint32_t t = *reinterpret_cast<const uint32_t *>(&T[16 + xO]); pw[0] = pw[1] = pw[2] = pw[3] = t; }
                } else if (i16mode == 1) {
                    if (A) {
                        const uint32_t l = *reinterpret_cast<const uint32_t *>(&R.Lcol[q][yO]);
#pragma unroll
                        for (int y = 0; y < 4; y++) pw[y] = ((l >> (8 * y)) & 255u) * 0x01010101u;
                    }
                } else if (i16mode == 2) {
                    const uint4 topv = *reinterpret_cast<const uint4 *>(&T[16]);
                    const uint4 lefv = *reinterpret_cast<const uint4 *>(R.Lcol[q]);
                    const int sumH = sum4(topv.x) + sum4(topv.y) + sum4(topv.z) + sum4(topv.w);
                    const int sumV = sum4(lefv.x) + sum4(lefv.y) + sum4(lefv.z) + sum4(lefv.w);
                    int v;
                    if (A && Bv) v = (sumH + sumV + 16) >> 5;
                    else if (A) v = (sumV + 8) >> 4;
                    else if (Bv) v = (sumH + 8) >> 4;
                    else v = 128;
                    pw[0] = pw[1] = pw[2] = pw[3] = (uint32_t)v * 0x01010101u;
                } else if (i16mode == 3) {
                    if (A && Bv) {
                        const uint4 topv = *reinterpret_cast<const uint4 *>(&T[16]);
                        const uint4 lefv = *reinterpret_cast<const uint4 *>(R.Lcol[q]);
                        const int cor = T[15];
                        const int Hh = plane_grad16(topv, (uint32_t)cor), Vv = plane_grad16(lefv, (uint32_t)cor);
                        const int aa = 16 * ((int)(lefv.w >> 24) + (int)(topv.w >> 24));
                        const int bb = (5 * Hh + 32) >> 6;
                        const int cc = (5 * Vv + 32) >> 6;
                        const int v00 = aa + bb * (xO - 7) + cc * (yO - 7) + 16;
#pragma unroll
                        for (int y = 0; y < 4; y++) pw[y] = plane_row(v00 + cc * y, bb);
                    }
                }
                emit_block(&T[(yO + 1) * 32 + 16 + xO], 32, pw, r2);
            } else if (kind == MVHP_KIND_I4x4) {
                // Intra 4x4: 16 dependent block steps, lane j = one sample of the block.
                // h264_intra_prediction.c:161-177, :315-483, :496-960 + transform4x4_luma (h264_transform.c:121-156).
                // availability per luma4x4BlkIdx (wave-uniform): deriv_neighbouringlocations, h264_spatial.c:739-786
                constexpr uint32_t X0 = (1u << 0) | (1u << 2) | (1u << 8) | (1u << 10);   // blocks with xO == 0
                constexpr uint32_t Y0 = (1u << 0) | (1u << 1) | (1u << 4) | (1u << 5);    // blocks with yO == 0
                const uint32_t av_left = A ? 0xffffu : (0xffffu & ~X0);
                const uint32_t av_up = Bv ? 0xffffu : (0xffffu & ~Y0);
                const uint32_t av_upleft = (0xffffu & ~(X0 | Y0)) | (Bv ? ((1u << 1) | (1u << 4) | (1u << 5)) : 0u) |
                                           (A ? ((1u << 2) | (1u << 8) | (1u << 10)) : 0u) | (D ? 1u : 0u);
                const uint32_t av_upright = ((1u << 2) | (1u << 6) | (1u << 8) | (1u << 9) | (1u << 10) | (1u << 12) | (1u << 14)) |
                                            (Bv ? ((1u << 0) | (1u << 1) | (1u << 4)) : 0u) | (Cav ? (1u << 5) : 0u);
                // neighbours each mode needs, 3 bits per mode: bit0 left, bit1 up, bit2 up-left (mode 2 = DC apart)
                constexpr uint32_t REQ = (2u << 0) | (1u << 3) | (0u << 6) | (2u << 9) | (7u << 12) | (7u << 15) | (7u << 18) |
                                         (2u << 21) | (1u << 24);
                // control word of block j, computed by lane j and broadcast inside the quarter at step j:
                // bit 31 the mode is DC, bit 16 prediction allowed, bits 0-15 tap table row offset
                uint32_t info;
                {
                    const uint32_t mw = (j < 4) ? m0 : (j < 8) ? m1 : (j < 12) ? m2 : m3;
                    const uint32_t mode = (mw >> ((j & 3) * 8)) & 255u;
                    const uint32_t avail = ((av_left >> j) & 1u) | (((av_up >> j) & 1u) << 1) | (((av_upleft >> j) & 1u) << 2);
                    const uint32_t req = (REQ >> (min(mode, 8u) * 3)) & 7u;
                    const uint32_t ok = (((req & ~avail) == 0u) && (mode < 9u)) ? 1u : 0u; // else the prediction stays 0 (:442)
                    const uint32_t trow = (((av_upright >> j) & 1u) ? 0u : 9u) + min(mode, 8u);
                    info = ((mode == 2u) ? 0x80000000u : 0u) | (ok << 16) | (trow * 64u);
                }
                const int pix = (j >> 2) * 32 + (j & 3);   // this lane's sample inside a block, tile units
                const uint8_t *tapb = reinterpret_cast<const uint8_t *>(B.tap4) + j * 4;
                // software pipeline: control word, table entry and residual of block b+1 are fetched before block
                // b's dependent tile reads (the ring holds zeros when the macroblock has no residual)
                const int qbase4 = (lane & 48) << 2;
                uint32_t inf = quarter_bcast(info, qbase4, 0);
                uint32_t e_nx = *reinterpret_cast<const uint32_t *>(tapb + (inf & 0xffffu));
                int r_nx = (int)S.res[j];
#pragma unroll
                for (int blk = 0; blk < 16; blk++) {
                    const int bxO = (((blk >> 2) & 1) << 3) | ((blk & 1) << 2);
                    const int byO = ((blk >> 3) << 3) | (((blk >> 1) & 1) << 2);
                    const int base = (byO + 1) * 32 + 16 + bxO;     // tile index of the block's top-left sample
                    const uint32_t cur = inf;
                    const uint32_t e = e_nx;
                    const int rr = r_nx;
                    if (blk < 15) {
                        inf = quarter_bcast(info, qbase4, blk + 1);
                        e_nx = *reinterpret_cast<const uint32_t *>(tapb + (inf & 0xffffu));
                        r_nx = (int)S.res[(blk + 1) * 16 + j];
                    }
                    const int okmask = ((int)(cur << 15)) >> 31;   // bit 16 -> 0 / -1
                    const int ta = T[base - 33 + (int)(e & 255)];
                    const int tb = T[base - 33 + (int)((e >> 8) & 255)];
                    const int tc = T[base - 33 + (int)(e >> 16)];
                    int pred = ((ta + 2 * tb + tc + 2) >> 2) & okmask;
                    const bool isdc = (int)cur < 0;
                    if (__builtin_amdgcn_ballot_w64(isdc) != 0) { // some quarter predicts DC
                        // which neighbours exist is positional, i.e. the same for the four pictures: scalar branches
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
                        pred = isdc ? dcv : pred;
                    }
                    T[base + pix] = (uint8_t)clip255(pred + rr);
                    WAVE_SYNC();
                }
            } else if (kind == MVHP_KIND_I8x8) {
                // Intra 8x8: h264_intra_prediction.c:1107-1353 (edge filter) + :1366-1793 + transform8x8_luma;
                // lane j predicts samples (4*(j&1) .. +3, j>>1) of the block
                uint8_t *E8 = R.E8[q];
                MVHP_UNROLL(4)
                for (int blk = 0; blk < 4; blk++) {
                    const int bxO = (blk & 1) * 8, byO = (blk >> 1) * 8;
                    const int mode = (int)((m0 >> (blk * 8)) & 255u);
                    const bool left = (bxO > 0) || A;
                    const bool up = (byO > 0) || Bv;
                    const bool upleft = (bxO > 0) ? ((byO > 0) || Bv) : ((byO > 0) ? A : D);
                    const bool upright = (blk == 0) ? Bv : (blk == 1) ? Cav : (blk == 2);
                    const uint8_t *Trow = &T[byO * 32 + 16 + bxO];
                    const uint8_t *Tcol = &T[(byO + 1) * 32 + 15 + bxO];
#pragma unroll
                    for (int half = 0; half < 2; half++) {
                        const int el = j + 16 * half;
                        if (el < 28) {
                            // raw edge sample for EE8 index e: e<=9: left[9-e] (clamped), 10: corner, >=11: top[e-11]
                            const int e = min(max(el, 2), 26);
                            const int maxi = upright ? 15 : 7;
                            int lo = e - 1, hi = e + 1;
                            if (e == 2 || (e == 11 && !upleft) || (e == 10 && !left)) lo = e;
                            if (e == 26 || (e == 9 && !upleft) || (e == 10 && !up)) hi = e;
                            int v[3];
                            const int idxs[3] = {lo, e, hi};
#pragma unroll
                            for (int t = 0; t < 3; t++) {
                                const int idx = idxs[t];
                                v[t] = (idx >= 10) ? (int)Trow[min(idx - 11, maxi)] : (int)Tcol[(9 - idx) * 32];
                            }
                            E8[el] = (uint8_t)((v[0] + 2 * v[1] + v[2] + 2) >> 2);
                        }
                    }
                    WAVE_SYNC();
                    {
                        const int y = j >> 1, x0 = (j & 1) * 4;
                        uint32_t pwv = 0;
                        if (mode == 2) {
                            const uint32_t *E = reinterpret_cast<const uint32_t *>(E8);
                            const uint32_t w0 = E[0], w1 = E[1], w2 = E[2], w3 = E[3], w4 = E[4];
                            const int sumV = sum4(w0 & 0xffff0000u) + sum4(w1) + sum4(w2 & 0x0000ffffu);       // E8[2..9]
                            const int sumH = sum4(w2 & 0xff000000u) + sum4(w3) + sum4(w4 & 0x00ffffffu);       // E8[11..18]
                            int v;
                            if (left && up) v = (sumH + sumV + 8) >> 4;
                            else if (left) v = (sumV + 4) >> 3;
                            else if (up) v = (sumH + 4) >> 3;
                            else v = 128;
                            pwv = (uint32_t)v * 0x01010101u;
                        } else {
                            bool ok;
                            switch (mode) {
                            case 0: case 3: case 7: ok = up; break;
                            case 1: case 8: ok = left; break;
                            default: ok = left && up && upleft; break;
                            }
                            if (ok && mode < 9) {
                                const uint4 e4 = *reinterpret_cast<const uint4 *>(&B.tap8[mode * 64 + y * 8 + x0]);
                                const uint32_t ee[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                                for (int x = 0; x < 4; x++) {
                                    const int v0 = E8[ee[x] & 255], v1 = E8[(ee[x] >> 8) & 255], v2 = E8[ee[x] >> 16];
                                    pwv |= (uint32_t)((v0 + 2 * v1 + v2 + 2) >> 2) << (8 * x);
                                }
                            }
                        }
                        int rr[4];
#pragma unroll
                        for (int x = 0; x < 4; x++) rr[x] = (int)S.res[blk * 64 + (x0 + x) * 8 + y];   // (zeros without residual)
                        uint32_t out = 0;
#pragma unroll
                        for (int x = 0; x < 4; x++) out |= (uint32_t)clip255((int)((pwv >> (8 * x)) & 255u) + rr[x]) << (8 * x);
                        *reinterpret_cast<uint32_t *>(&T[(byO + y + 1) * 32 + 16 + bxO + x0]) = out;
                    }
                    WAVE_SYNC();
                }
            } else {
                // I_PCM (8.3.5; only MVHP_STREAM_SPEC streams carry it): the samples as they are.  Record layout
                // (minivideo_hotpath.h): the owner of luma block 2i holds luma rows 2i and 2i+1 (the owner of block 2i+1 Cb row
                // i and Cr row i: O's part).
                if ((j & 1) == 0) {
                    const int4 sA = *reinterpret_cast<const int4 *>(&S.res[j * 16]), sB = *reinterpret_cast<const int4 *>(&S.res[j * 16 + 8]);
                    const int jp = j >> 1;
                    *reinterpret_cast<int4 *>(&T[(2 * jp + 1) * 32 + 16]) = sA;
                    *reinterpret_cast<int4 *>(&T[(2 * jp + 2) * 32 + 16]) = sB;
                }
            }
            WAVE_SYNC();
            PSTAMP(4);   // luma (all paths some quarter takes)

            // ---- luma neighbour state for the next macroblock (built in the OTHER tile) / the next row, then publish ----
            {
                // corner (this macroblock's top-right sample) by lane 0, left column: lane j row j; bottom row -> line buffer by
                // lanes 0-3 (one dword each)
                const uint8_t kl = T[(j + 1) * 32 + 31];
                const uint8_t kk = T[31];
                uint32_t bot = 0;
                if (j < 4) bot = *reinterpret_cast<const uint32_t *>(&T[16 * 32 + 16 + j * 4]);
                WAVE_SYNC();
                Tn.T[(j + 1) * 32 + 15] = kl;
                R.Lcol[q][j] = kl;
                if (j == 0) Tn.T[15] = kk;
                if (j < 4) *reinterpret_cast<uint32_t *>(&line_y[mbx * 16 + j * 4]) = bot;
                if (seam_out && j < 4)   // the same four dwords, tagged, to the band below (one write-through store per granule)
                    __hip_atomic_store(seam_wr + qf * seam_pic + (uint32_t)(mbx * SEAM_GRANULES + j), ((unsigned long long)a.wide_epoch << 32) | bot,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // LDS operations of one wave complete in order; the explicit wait makes the tile, the line buffer and the neighbour
            // columns land before the counter
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&C.k_done[r], mbx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            WAVE_SYNC();
            PSTAMP(5);   // neighbour state, publish
        }
#if defined(MVHP_PIPE_STAMPS)
        if (lane_c == 0)
            for (int i = 0; i < 8; i++) { atomicAdd(&g_pipe_stamps[i][0], ps_acc[i]); atomicAdd(&g_pipe_stamps[i][1], (unsigned long long)ps_n[i]); }
#endif
        return;
    }

    // =============================================================================================================
    // O: chroma prediction (h264_intra_prediction.c:2157-2564 + transform4x4_chroma) and write-out (planar gather of
    //    export.c:65-188, mb_to_rgb export_utils.c:209-324 fused).  Chroma needs the macroblock above, not the one above-right:
    //    this wave follows the row above's O by ONE macroblock.  Strip owned by MACROBLOCK as in recon_quad.hip: lane (m, h) =
    //    (j & 3, j >> 2) of a quarter keeps luma rows 4h .. 4h+3 of macroblock m of the strip and writes them when the strip is
    //    complete -- four adjacent lanes then cover 64 contiguous bytes of a luma row (32 of a chroma row) per store
    //    instruction, and the 192 RGB bytes of a row leave in three consecutive instructions.
    // =============================================================================================================
    {
        const bool seam_in = (r == 0) && band > 0;
        const bool seam_out = (r == PIPE_ROWS - 1) && (row + 1 < H);
        const uint32_t seam_pic = (uint32_t)((bands - 1) * W * SEAM_GRANULES);
        const unsigned long long *seam_rd = seam_in ? a.seam + ((size_t)grp * 4 * (bands - 1) + (band - 1)) * W * SEAM_GRANULES : nullptr;
        unsigned long long *seam_wr = seam_out ? a.seam + ((size_t)grp * 4 * (bands - 1) + band) * W * SEAM_GRANULES : nullptr;
        unsigned long long seam_pend = 0;
        uint8_t *gyuv = a.yuv + (size_t)grp * 4 * W * H * 384;
        uint8_t *grgb = a.rgb + (size_t)grp * 4 * W * H * 768;
        v4i L0 = {0, 0, 0, 0}, L1 = L0, L2 = L0, L3 = L0;
#pragma unroll 1
        for (int mbx = 0; mbx < W; mbx++) {
            int lane = lane_c;
            asm volatile("" : "+v"(lane));
            const int q = lane >> 4, j = lane & 15;
            const PSlot &S = R.slot[mbx % NSLOT][q];
            const PTile &Tl = R.tile[mbx % NTILE][q];
            uint8_t (*TC)[9 * 16] = R.TC[q];
            uint8_t (*SC)[8 * 24] = R.SC[q];
            uint8_t *line_cb = lines + (size_t)q * W * 32 + W * 16;
            uint8_t *line_cr = line_cb + W * 8;
            const bool A = mbx > 0;

            if (seam_in && (mbx & 1) == 0) {
                // Macroblocks mbx and mbx + 1 read chroma columns mbx and mbx + 1 of the row above.  The first step of a row fetches
                // (0, 1) now; every later even step finds its two columns asked for two steps ago, and asks for (mbx + 2, mbx + 3).
                // Lane j < 8 of a quarter: column mbx + (j >> 2), granule 4 + (j & 3) (4-5 Cb dwords, 6-7 Cr).
                const int g = 4 + (j & 3);
                const uint32_t lo = qf * seam_pic + (uint32_t)g;
                const int col = mbx + ((j >> 2) & 1);
                const bool act = (j < 8) && (col < W);
                const unsigned long long *src = seam_rd + lo + (uint32_t)((act ? col : 0) * SEAM_GRANULES);
                unsigned long long v = seam_pend;
                if (mbx == 0) v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__builtin_amdgcn_ballot_w64(act && (uint32_t)(v >> 32) != a.wide_epoch) != 0) {
                    __builtin_amdgcn_s_sleep(2);
                    bool stop = ++spins > (1 << 20) || __hip_atomic_load(&C.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (!stop && (spins & 255) == 0) stop = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                    if (stop) {
                        if (lane == 0) { __hip_atomic_store(&C.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicOr(a.err, 1u); }
                        return;
                    }
                    v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (act) {
                    uint8_t *dst = (g < 6) ? &line_cb[col * 8 + (g - 4) * 4] : &line_cr[col * 8 + (g - 6) * 4];
                    *reinterpret_cast<uint32_t *>(dst) = (uint32_t)v;
                }
                const int ncol = mbx + 2 + ((j >> 2) & 1);
                if (j < 8 && ncol < W) seam_pend = __hip_atomic_load(seam_rd + lo + (uint32_t)(ncol * SEAM_GRANULES), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                WAVE_SYNC();
            }

            // header and chroma residuals of this macroblock; the row above's chroma of this column
            if (!pipe_wait<MVHP_PIPE_NAP_O>(&C.f_done[r], mbx + 1, C, a.err, lane)) return;
            const uint32_t h0 = S.hdr[0];
            const int kind = h0 & 255;
            const int cmode = (h0 >> 24) & 255;
            if (Bv) {
                if (!seam_in && !pipe_wait<MVHP_PIPE_NAP_O>(&C.c_done[r - 1], mbx + 1, C, a.err, lane)) return;
                if (j < 4) {   // lanes 0-1 Cb top, 2-3 Cr top: one dword each
                    const uint8_t *src = (j < 2) ? &line_cb[mbx * 8 + j * 4] : &line_cr[mbx * 8 + (j - 2) * 4];
                    *reinterpret_cast<uint32_t *>(&TC[j >> 1][8 + (j & 1) * 4]) = *reinterpret_cast<const uint32_t *>(src);
                }
            }
            WAVE_SYNC();

            // ---- chroma prediction (h264_intra_prediction.c:2157-2564 + transform4x4_chroma): lane j < 8 predicts its own
            //      4x4 block (plane j >> 2, block j & 3) ----
            if (j < 8) {
                const int pl = j >> 2, k = j & 3;
                const int cx = (k & 1) * 4, cy = (k >> 1) * 4;
                uint8_t *TCp = TC[pl];
                const int4 ca = *reinterpret_cast<const int4 *>(&S.c2[j * 8]), cb = *reinterpret_cast<const int4 *>(&S.c2[j * 8 + 4]);
                const int c2[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
                uint32_t pw[4] = {0u, 0u, 0u, 0u};
                const uint32_t topw = *reinterpret_cast<const uint32_t *>(&TCp[8 + cx]);
                const uint32_t lefw = *reinterpret_cast<const uint32_t *>(&R.LcolC[q][pl][cy]);
                if (cmode == 0) {
                    const int bx = k & 1, by = k >> 1;
                    const int sH = sum4(topw), sV = sum4(lefw);
                    int v;
                    if (!A && !Bv) v = 128;
                    else if (bx == by) {
                        if (A && Bv) v = (sH + sV + 4) >> 3;
                        else if (A) v = (sV + 2) >> 2;
                        else v = (sH + 2) >> 2;
                    } else if (bx == 1) { // xO > 0, yO == 0: prefers top
                        v = Bv ? ((sH + 2) >> 2) : ((sV + 2) >> 2);
                    } else {              // xO == 0, yO > 0: prefers left
                        v = A ? ((sV + 2) >> 2) : ((sH + 2) >> 2);
                    }
                    pw[0] = pw[1] = pw[2] = pw[3] = (uint32_t)v * 0x01010101u;
                } else if (cmode == 1) {
                    if (A) {
#pragma unroll
                        for (int y = 0; y < 4; y++) pw[y] = ((lefw >> (8 * y)) & 255u) * 0x01010101u;
                    }
                } else if (cmode == 2) {
                    if (Bv) pw[0] = pw[1] = pw[2] = pw[3] = topw;
                } else if (cmode == 3) {
                    if (A && Bv) {
                        const uint2 topv = *reinterpret_cast<const uint2 *>(&TCp[8]);
                        const uint2 lefv = *reinterpret_cast<const uint2 *>(R.LcolC[q][pl]);
                        const int cor = TCp[7];
                        const int Hh = plane_grad8(topv, (uint32_t)cor), Vv = plane_grad8(lefv, (uint32_t)cor);
                        const int aa = 16 * ((int)(lefv.y >> 24) + (int)(topv.y >> 24));
                        const int bb = (34 * Hh + 32) >> 6;
                        const int cc = (34 * Vv + 32) >> 6;
                        const int v00 = aa + bb * (cx - 3) + cc * (cy - 3) + 16;
#pragma unroll
                        for (int y = 0; y < 4; y++) pw[y] = plane_row(v00 + cc * y, bb);
                    }
                }
                emit_block(&TCp[(cy + 1) * 16 + 8 + cx], 16, pw, c2);
            }

            if (kind == MVHP_KIND_IPCM && (j & 1)) {
                // I_PCM (see K): the owner of luma block 2i + 1 holds Cb row i and Cr row i, over what the prediction made
                const int4 sA = *reinterpret_cast<const int4 *>(&S.res[j * 16]);
                const int jp = j >> 1;
                *reinterpret_cast<int2 *>(&TC[0][(jp + 1) * 16 + 8]) = make_int2(sA.x, sA.y);
                *reinterpret_cast<int2 *>(&TC[1][(jp + 1) * 16 + 8]) = make_int2(sA.z, sA.w);
            }
            WAVE_SYNC();

            // ---- the luma of this macroblock: park, or flush the strip ----
            if (!pipe_wait<MVHP_PIPE_NAP_O>(&C.k_done[r], mbx + 1, C, a.err, lane)) return;
            const int mbi = mbx & 3;
            const int m_own = j & 3, h_own = j >> 2;
            const bool flush = (mbi == 3 || mbx == W - 1);
            if (m_own == mbi) {   // this macroblock's owners take its luma rows out of the tile
                const uint8_t *t0 = &Tl.T[(4 * h_own + 1) * 32 + 16];
                L0 = *reinterpret_cast<const v4i *>(t0);            L1 = *reinterpret_cast<const v4i *>(t0 + 32);
                L2 = *reinterpret_cast<const v4i *>(t0 + 2 * 32);   L3 = *reinterpret_cast<const v4i *>(t0 + 3 * 32);
            }
            uint2 cb0 = make_uint2(0, 0), cb1 = cb0, cr0 = cb0, cr1 = cb0;
            if (flush) {
                // chroma rows 2h, 2h + 1 of the lane's macroblock: parked ones from the strip, the current one from the tile
                const bool cur = (m_own == mbi);
                const uint8_t *cb_src = cur ? &TC[0][(2 * h_own + 1) * 16 + 8] : &SC[0][2 * h_own * 24 + m_own * 8];
                const uint8_t *cr_src = cur ? &TC[1][(2 * h_own + 1) * 16 + 8] : &SC[1][2 * h_own * 24 + m_own * 8];
                const int cstep = cur ? 16 : 24;
                cb0 = *reinterpret_cast<const uint2 *>(cb_src); cb1 = *reinterpret_cast<const uint2 *>(cb_src + cstep);
                cr0 = *reinterpret_cast<const uint2 *>(cr_src); cr1 = *reinterpret_cast<const uint2 *>(cr_src + cstep);
            } else {
                // park the chroma rows (lane j: row j & 7 of plane j >> 3) in the LDS strip
                const uint2 cv = *reinterpret_cast<const uint2 *>(&TC[j >> 3][((j & 7) + 1) * 16 + 8]);
                *reinterpret_cast<uint2 *>(&SC[j >> 3][(j & 7) * 24 + mbi * 8]) = cv;
            }
            // the luma tile and the ring slot have been read: K may build macroblock mbx + 2 in the tile, F may refill the slot
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&C.o_done[r], mbx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

            // ---- chroma neighbour state for the next macroblock / the next row, then publish the chroma ----
            {
                // corners (this macroblock's top-right samples) by lanes 0-1, left columns: lane j row j & 7 of plane j >> 3;
                // bottom rows -> line buffer by lanes 0-3 (one dword each)
                const uint8_t kc = TC[j >> 3][((j & 7) + 1) * 16 + 15];
                uint8_t kk = 0;
                if (j < 2) kk = TC[j][15];
                uint32_t bot = 0;
                uint8_t *bdst = line_cb;
                if (j < 2) { bot = *reinterpret_cast<const uint32_t *>(&TC[0][8 * 16 + 8 + j * 4]); bdst = &line_cb[mbx * 8 + j * 4]; }
                else if (j < 4) { bot = *reinterpret_cast<const uint32_t *>(&TC[1][8 * 16 + 8 + (j - 2) * 4]); bdst = &line_cr[mbx * 8 + (j - 2) * 4]; }
                WAVE_SYNC();
                TC[j >> 3][((j & 7) + 1) * 16 + 7] = kc;
                R.LcolC[q][j >> 3][j & 7] = kc;
                if (j < 2) TC[j][7] = kk;
                if (j < 4) *reinterpret_cast<uint32_t *>(bdst) = bot;
                if (seam_out && j < 4)   // the same four dwords, tagged, to the band below (granules 4-7 of the column)
                    __hip_atomic_store(seam_wr + qf * seam_pic + (uint32_t)(mbx * SEAM_GRANULES + 4 + j), ((unsigned long long)a.wide_epoch << 32) | bot,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&C.c_done[r], mbx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

            if (flush && m_own <= mbi && valid) {
                const uint32_t x0 = (uint32_t)((mbx & ~3) * 16 + m_own * 16);
                const uint32_t lrow = (uint32_t)((row * 16 + 4 * h_own) * pitch) + x0;     // luma row 4h of the macroblock row
                const uint32_t oyuv = __umul24(qmb, 384u);
                const uint32_t pl = oyuv + lrow;
                const uint32_t pcb = oyuv + plane_y + (uint32_t)((row * 8 + 2 * h_own) * cpitch) + (x0 >> 1), pcr = pcb + plane_c;
                *reinterpret_cast<v4i *>(gyuv + pl) = L0;
                *reinterpret_cast<v4i *>(gyuv + pl + pitch) = L1;
                *reinterpret_cast<v4i *>(gyuv + pl + 2 * pitch) = L2;
                *reinterpret_cast<v4i *>(gyuv + pl + 3 * pitch) = L3;
                *reinterpret_cast<uint2 *>(gyuv + pcb) = cb0;
                *reinterpret_cast<uint2 *>(gyuv + pcr) = cr0;
                *reinterpret_cast<uint2 *>(gyuv + pcb + cpitch) = cb1;
                *reinterpret_cast<uint2 *>(gyuv + pcr + cpitch) = cr1;
                if (RGB) {
                    // one luma row of the lane's macroblock against its chroma row (rows 2c, 2c + 1 share row c, export_utils.c:278-279)
                    const uint32_t prgb = __umul24(qmb, 768u) + lrow * 3u;
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        const v4i yk = (rr == 0) ? L0 : (rr == 1) ? L1 : (rr == 2) ? L2 : L3;
                        const uint2 cbq = (rr < 2) ? cb0 : cb1, crq = (rr < 2) ? cr0 : cr1;
                        v4i a0, a1, a2;
                        rgb16(make_uint4((uint32_t)yk.x, (uint32_t)yk.y, (uint32_t)yk.z, (uint32_t)yk.w), cbq, crq, a0, a1, a2);
                        v4i *dst = reinterpret_cast<v4i *>(grgb + prgb + (uint32_t)rr * 3u * (uint32_t)pitch);
                        dst[0] = a0; dst[1] = a1; dst[2] = a2;
                    }
                }
            }
            WAVE_SYNC();
        }
    }
}

size_t recon_pipe_lds_bytes(int width_mbs, int rows)
{
    return sizeof(QTables) + sizeof(PCtl) + (size_t)4 * width_mbs * 32 + (size_t)rows * sizeof(PRow);
}

template <int PIPE_ROWS, bool RGB>
static hipError_t launch_pipe_one(const ReconArgs &a, hipStream_t stream)
{
    const int bands = (a.height_mbs + PIPE_ROWS - 1) / PIPE_ROWS;
    const size_t lds = recon_pipe_lds_bytes(a.width_mbs, PIPE_ROWS);
    const int groups = (a.n_frames + 3) / 4;
    hipError_t e = hipFuncSetAttribute((const void *)recon_pipe_kernel<PIPE_ROWS, RGB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((recon_pipe_kernel<PIPE_ROWS, RGB>), dim3(groups * bands), dim3(PIPE_ROWS * 3 * 64), lds, stream, a);
    return hipGetLastError();
}

// one workgroup per (band of `rows` rows, group of four pictures); a.wide_ticket / wide_base / wide_epoch / seam set by the caller
hipError_t launch_recon_pipe(const ReconArgs &a, int rows, hipStream_t stream)
{
    if (!a.wide_ticket || !a.wide_epoch) return hipErrorInvalidValue;
    if ((a.height_mbs + rows - 1) / rows > 1 && !a.seam) return hipErrorInvalidValue;
    const bool rgb = a.rgb != nullptr;
    switch (rows) {
    case 1: return rgb ? launch_pipe_one<1, true>(a, stream) : launch_pipe_one<1, false>(a, stream);
    case 2: return rgb ? launch_pipe_one<2, true>(a, stream) : launch_pipe_one<2, false>(a, stream);
    case 4: return rgb ? launch_pipe_one<4, true>(a, stream) : launch_pipe_one<4, false>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace mvhp

#if defined(MVHP_PIPE_STAMPS)
extern "C" __attribute__((visibility("default"))) int mvhp_debug_pipe_stamps(unsigned long long *out, int clear)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(mvhp::g_pipe_stamps), sizeof(mvhp::g_pipe_stamps)) != hipSuccess) return 0;
    if (clear) {
        static const unsigned long long zero[16][2] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mvhp::g_pipe_stamps), zero, sizeof(zero)) != hipSuccess) return 0;
    }
    return 1;
}
#endif
