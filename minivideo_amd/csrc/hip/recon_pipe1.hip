// recon_pipe1.hip -- the low-latency form for ONE picture per wavefront (gfx950 only): a macroblock row is worked on by THREE
// wavefronts in a pipeline, a picture's rows are spread over several workgroups, and nothing runs in lock step with another
// picture.
//
//   recon_pipe1_kernel<R, EXT>  same contract as recon_rows_kernel (recon_kernels.hip): replaces intra_prediction_process()
//                               (decoder/h264/h264_intra_prediction.c:112-145), all of h264_transform.c, the planar gather of
//                               export.c:65-188 and mb_to_rgb() (export_utils.c:209-324) for whole pictures.
//
// recon_pipe_kernel (recon_pipe.hip) does this with four pictures per wavefront: its luma wave runs the Intra16x16 path, the
// Intra4x4 chain and (High profile) the four Intra8x8 blocks one after the other whenever its four pictures disagree on the
// macroblock kind, which they nearly always do.  Here a wavefront has ONE picture, so the luma wave's step is the path of that
// macroblock alone, on 64 lanes (the arithmetic and the lane mappings are recon_rows_kernel's):
//   F  records -> dequantised, inverse-transformed residuals of a macroblock PAIR + the two headers, in a ring in LDS;
//   K  wait for the row above (or the luma granules of the seam), luma prediction + residual add into a ring of tiles, luma
//      neighbour state, publish;
//   O  chroma prediction (one macroblock behind the row above's O: no up-right dependency), 4-macroblock output strip, colour
//      conversion, stores.
// Bands of R = 1, 2 or 4 rows per workgroup (3 R wavefronts), band-major tickets and seam granules as the other wide forms
// (granules 0-3 of a column are K's, 4-7 O's).  EXT: pictures of several slices and scaling matrices (MVHP_PARAM_SLICES /
// MVHP_PARAM_SCALING), as recon_rows_kernel<.., EXT>.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "minivideo_hotpath.h"
#include "recon_kernels.h"
#include "recon_device.h"
#include "recon_batch_device.h"
#include "recon_rows_device.h"

#ifndef MVHP_PIPE_PRIO_K
#define MVHP_PIPE_PRIO_K 3   // wave priority of the luma wave (K), and of the write-out wave (O); the residual wave (F) stays at 0
                            // (K 2 / O 0 before: K1p1 64 x 1080p High 0.951 -> 0.924 ms, 150 pictures 1.78 -> 1.70; K1p -1 %)
#endif
#ifndef MVHP_PIPE_PRIO_O
#define MVHP_PIPE_PRIO_O 2
#endif

namespace mvhp {
namespace p1 {

#ifndef MVHP_PIPE_NAP_F
#define MVHP_PIPE_NAP_F 8   // s_sleep units between polls of the residual wave (it runs ahead) ...
#endif
#ifndef MVHP_PIPE_NAP_O
#define MVHP_PIPE_NAP_O 3   // ... and of the chroma + output wave (it runs behind)
#endif
#ifndef MVHP_WIDE_NAP
#define MVHP_WIDE_NAP 1   // s_sleep units between the luma wave's polls of the row above
#endif
constexpr int ROWS_MAX = 4;
constexpr int NPAIR = 3;    // F -> K / O ring, in macroblock PAIRS: F may be this many pairs ahead of the slower of K and O
constexpr int NTILE = 3;    // K -> O ring of luma tiles

struct __attribute__((aligned(16))) P1Tables {   // as BlockLds of recon_kernels.hip, without the counters
    int     ls4[18];       // LevelScale4x4 classes, 16*normAdjust (h264.c:427-435)
    int     ls8[36];       // LevelScale8x8 classes (h264.c:438-446)
    int     pad[2];
    uint8_t cls8[64];      // 8x8 position -> class
    uint8_t w4[3][16];     // weight matrices (raster), 16 = flat: LevelScale = weight * normAdjust (h264_transform.c:645-741)
    uint8_t w8[64];
    uint32_t tap4[2 * 9 * 16]; // Intra4x4: [up-right unavailable][mode][sample] -> three byte offsets into the tile
    uint32_t tap8[9 * 64];     // Intra8x8: [mode][sample] -> three indices into the filtered edge array E8
};

struct __attribute__((aligned(16))) P1Ctl {
    int f_done[ROWS_MAX];      // macroblocks whose residuals + headers F has left in the ring (a multiple of 2, or W)
    int k_done[ROWS_MAX];      // macroblocks whose luma K has finished
    int o_done[ROWS_MAX];      // macroblocks O has taken out of their tile and ring slot
    int c_done[ROWS_MAX];      // macroblocks whose chroma O has finished (what the row below's O waits for)
    int abort_flag;
    int unit;
    int pad[2];
};

struct __attribute__((aligned(16))) P1Row {
    int16_t  res[NPAIR][2][384];   // F -> K / O: residuals of a macroblock pair, MB raster: luma y*16+x | 256+Cb y*8+x | 320+Cr
    uint32_t hdr[NPAIR][2][8];     // ... and their record headers
    int32_t  scr[256];             // F: 8x8 transpose scratch / DC exchange
    uint8_t  T[NTILE][17 * 32 + 16]; // K -> O: luma tiles: row 0 = top neighbours; byte 15 = left/corner, 16..31 samples; row 0 bytes 32..39 = up-right
    uint8_t  Lcol[16];             // K: compact left neighbour column (luma)
    uint8_t  E8[32];               // K: filtered Intra8x8 edge
    uint8_t  TC[2][9 * 16];        // O: chroma tiles: row 0 = top; byte 7 = left/corner, 8..15 samples
    uint8_t  LcolC[2][8];          // O: compact left neighbour columns (Cb, Cr)
    uint8_t  SY[16 * 64];          // O: output strip: 4 macroblocks of reconstructed luma (flushed with wide stores)
    uint8_t  SC[2][8 * 32];        // O: output strip: 4 macroblocks of Cb / Cr
};

// NAP: s_sleep units (64 clocks) between polls -- 1 on the luma chain, longer for the waves that run ahead of or behind it: a
// polling wave spends vector-ALU issue slots (compare, branch) that the luma wave on the same SIMD wants
template <int NAP = 1>
__device__ __forceinline__ bool p1_wait(const int *ctr, int need, P1Ctl &C, uint32_t *err, int lane)
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

// one lane-uniform poll of seam granules: `act` lanes hold granule pointers; spins until every active lane's tag matches
__device__ __forceinline__ bool p1_seam_poll(const unsigned long long *src, bool act, unsigned long long &v, uint32_t epoch, P1Ctl &C,
                                             uint32_t *err, int lane)
{
    int spins = 0;
    while (__builtin_amdgcn_ballot_w64(act && (uint32_t)(v >> 32) != epoch) != 0) {
        __builtin_amdgcn_s_sleep(2);
        bool stop = ++spins > (1 << 20) || __hip_atomic_load(&C.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!stop && (spins & 255) == 0) stop = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (stop) {
            if (lane == 0) { __hip_atomic_store(&C.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicOr(err, 1u); }
            return false;
        }
        v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return true;
}

// ---- the arithmetic of recon_rows_kernel, shared (recon_rows_device.h); the Intra4x4 chain is this kernel's own ----
using namespace rowsdev;

// Intra 4x4 macroblock: TEN dependent steps instead of sixteen -- the blocks of an anti-diagonal do not depend on each other
// ({2,4}, {3,5}, {6,8}, {7,9}, {10,12}, {11,13}: left / up / up-left / up-right of either lie on earlier anti-diagonals, and
// blocks 3, 11, 13 have no up-right by the rule of h264_intra_prediction.c:410-412), so lanes 0..15 predict one block of the
// step and lanes 16..31 the other, one sample each.  (recon_rows_kernel runs the sixteen blocks one after the other on 16
// lanes: there three more waves share the SIMD; here this chain IS the step every other row waits for.)
// h264_intra_prediction.c:161-177, :315-483, :496-960 + transform4x4_luma (h264_transform.c:121-156).
__device__ __forceinline__ void predict_mb_4x4(uint8_t *WT, const P1Tables &B, int lane, uint32_t m0, uint32_t m1,
                                               uint32_t m2, uint32_t m3, bool A, bool Bv, bool C, bool D, bool has_res,
                                               const int16_t *res)
{
    const Avail4 av = avail4(A, Bv, C, D);
    // neighbours each mode needs, 3 bits per mode: bit0 left, bit1 up, bit2 up-left (mode 2 = DC handled apart)
    constexpr uint32_t REQ = (2u << 0) | (1u << 3) | (0u << 6) | (2u << 9) | (7u << 12) | (7u << 15) | (7u << 18) |
                             (2u << 21) | (1u << 24);
    const int rmask = has_res ? -1 : 0;
    // Per-block control word, computed once by lane b for block b (16 lanes in parallel) and handed to the steps with
    // v_readlane: bits 0-1 left/up available, bit 2 mode is DC, bit 3 prediction allowed, bits 8.. byte offset of the block's
    // row in the tap table.
    uint32_t info;
    {
        const int b = lane & 15;
        const uint32_t mw = (b < 4) ? m0 : (b < 8) ? m1 : (b < 12) ? m2 : m3;
        const uint32_t mode = (mw >> ((b & 3) * 8)) & 255u;
        const uint32_t avail = ((av.left >> b) & 1u) | (((av.up >> b) & 1u) << 1) | (((av.upleft >> b) & 1u) << 2);
        const uint32_t req = (REQ >> (min(mode, 8u) * 3)) & 7u;
        const uint32_t ok = (((req & ~avail) == 0u) && (mode < 9u)) ? 1u : 0u; // else the prediction stays 0 (:442)
        const uint32_t trow = (((av.upright >> b) & 1u) ? 0u : 9u) + min(mode, 8u);
        info = (avail & 3u) | ((mode == 2u) ? 4u : 0u) | (ok << 3) | ((trow * 64u) << 8);
    }
    if (lane < 32) {
        const bool second = lane >= 16;                 // which block of the step this lane works on
        const int sl = lane & 15;
        const int pix = (sl >> 2) * 32 + (sl & 3);      // this lane's sample inside a block, tile units
        const int rpix = (sl >> 2) * 16 + (sl & 3);     // same in the residual array
        const uint8_t *T = WT;
        const uint8_t *tapb = reinterpret_cast<const uint8_t *>(B.tap4) + sl * 4;
        constexpr int SA[10] = {0, 1, 2, 3, 6, 7, 10, 11, 14, 15};
        constexpr int SB[10] = {-1, -1, 4, 5, 8, 9, 12, 13, -1, -1};
#pragma unroll
        for (int st = 0; st < 10; st++) {
            constexpr auto XO = [](int b) { return (((b >> 2) & 1) << 3) | ((b & 1) << 2); };
            constexpr auto YO = [](int b) { return ((b >> 3) << 3) | (((b >> 1) & 1) << 2); };
            const int bA = SA[st], bB = SB[st] < 0 ? SA[st] : SB[st];
            const bool on = !second || SB[st] >= 0;
            const uint32_t infA = __builtin_amdgcn_readlane(info, bA), infB = __builtin_amdgcn_readlane(info, bB);
            const uint32_t cur = second ? infB : infA;
            const int base = second ? (YO(bB) + 1) * 32 + 16 + XO(bB) : (YO(bA) + 1) * 32 + 16 + XO(bA);   // tile index of the block's top-left sample
            const int rbase = second ? YO(bB) * 16 + XO(bB) : YO(bA) * 16 + XO(bA);
            if (on) {
                const uint32_t e = *reinterpret_cast<const uint32_t *>(tapb + (cur >> 8));
                const int r = (int)res[rbase + rpix] & rmask;
                int pred;
                if (cur & 4u) { // DC
                    const int sumH = sum4(*reinterpret_cast<const uint32_t *>(&T[base - 32]));
                    const int sumV = T[base - 1] + T[base + 31] + T[base + 63] + T[base + 95];
                    const uint32_t lu = cur & 3u; // 3 both, 1 left only, 2 up only, 0 none
                    const int both = (sumH + sumV + 4) >> 3, l = (sumV + 2) >> 2, u = (sumH + 2) >> 2;
                    pred = (lu == 3u) ? both : (lu == 1u) ? l : (lu == 2u) ? u : 128;
                } else {
                    const int okmask = (cur & 8u) ? -1 : 0;
                    const int a = T[base - 33 + (int)(e & 255)];
                    const int b = T[base - 33 + (int)((e >> 8) & 255)];
                    const int c = T[base - 33 + (int)(e >> 16)];
                    pred = ((a + 2 * b + c + 2) >> 2) & okmask;
                }
                WT[base + pix] = (uint8_t)clip255(pred + r);
            }
            WAVE_SYNC();
        }
    }
    WAVE_SYNC();
}

template <int R, bool EXT>
__global__ __launch_bounds__(R * 3 * 64) void recon_pipe1_kernel(ReconArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int W = a.width_mbs, H = a.height_mbs;
    P1Tables &B = *reinterpret_cast<P1Tables *>(smem);
    P1Ctl &C = *reinterpret_cast<P1Ctl *>(smem + sizeof(P1Tables));
    uint8_t *line_y = smem + sizeof(P1Tables) + sizeof(P1Ctl);
    uint8_t *line_cb = line_y + W * 16;
    uint8_t *line_cr = line_cb + W * 8;
    P1Row *rows = reinterpret_cast<P1Row *>(line_cr + W * 8);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int role = wave / R, r = wave - role * R;     // role 0 = F, 1 = K, 2 = O of row r of the band
    if (MVHP_PIPE_PRIO_O && role == 2) __builtin_amdgcn_s_setprio(MVHP_PIPE_PRIO_O);
    const int lane_c = threadIdx.x & 63;
    constexpr int NT = R * 3 * 64;

    if (threadIdx.x == 0) C.unit = (int)(atomicAdd(a.wide_ticket, 1u) - a.wide_base);
    // ---- one-time table setup (as recon_rows_kernel) ----
    for (int i = threadIdx.x; i < 18; i += NT) B.ls4[i] = 16 * c_v4x4[i];
    for (int i = threadIdx.x; i < 36; i += NT) B.ls8[i] = 16 * c_v8x8[i];
    for (int i = threadIdx.x; i < 64; i += NT) {
        const int rr = i >> 3, c = i & 7;
        int k;
        if ((rr % 4 == 0) && (c % 4 == 0)) k = 0;
        else if ((rr % 2 == 1) && (c % 2 == 1)) k = 1;
        else if ((rr % 4 == 2) && (c % 4 == 2)) k = 2;
        else if (((rr % 4 == 0) && (c % 2 == 1)) || ((rr % 2 == 1) && (c % 4 == 0))) k = 3;
        else if (((rr % 4 == 0) && (c % 4 == 2)) || ((rr % 4 == 2) && (c % 4 == 0))) k = 4;
        else k = 5;
        B.cls8[i] = (uint8_t)k;
    }
    for (int i = threadIdx.x; i < 2 * 9 * 16; i += NT)
        B.tap4[i] = tap4_entry((i >> 4) % 9, i & 3, (i >> 2) & 3, i >= 9 * 16);
    for (int i = threadIdx.x; i < 9 * 64; i += NT) B.tap8[i] = tap8_entry(i >> 6, i & 7, (i >> 3) & 7);
    if (EXT)
        for (int i = threadIdx.x; i < 112; i += NT) (&B.w4[0][0])[i] = a.scaling ? a.weights[i] : (uint8_t)16;   // w4 | w8 are adjacent
    if (threadIdx.x < ROWS_MAX) { C.f_done[threadIdx.x] = 0; C.k_done[threadIdx.x] = 0; C.o_done[threadIdx.x] = 0; C.c_done[threadIdx.x] = 0; }
    if (threadIdx.x == 16) C.abort_flag = 0;
    __syncthreads();

    // this workgroup's picture and band (band-major tickets: see recon_rows_kernel)
    const int bands = (H + R - 1) / R;
    const int unit = __builtin_amdgcn_readfirstlane(C.unit);
    const int band = unit / a.n_frames;
    const int frame = unit - band * a.n_frames;
    if ((unsigned)unit >= (unsigned)(bands * a.n_frames)) {   // a ticket outside the launch: the host's bookkeeping of the counter is off
        if (threadIdx.x == 0) atomicOr(a.err, 2u);
        return;
    }
    const int row = band * R + r;
    if (row >= H) return;                            // the three waves of a row beyond the picture
    P1Row &Rw = rows[r];
    const bool BvG = row > 0;                        // the row above exists: what the waits and the fetches go by
    const bool seam_in = (r == 0) && band > 0;       // top neighbours of this row come from the seam above
    const bool seam_out = (r == R - 1) && (row + 1 < H);
    const unsigned long long *seam_rd = seam_in ? a.seam + (size_t)(frame * (bands - 1) + band - 1) * W * SEAM_GRANULES : nullptr;
    unsigned long long *seam_wr = seam_out ? a.seam + (size_t)(frame * (bands - 1) + band) * W * SEAM_GRANULES : nullptr;
    const unsigned long long seam_tag = (unsigned long long)a.wide_epoch << 32;

    if (role == 0) {
        // =========================================================================================================
        // F: records -> residuals of a macroblock pair (residual_pair: lanes 0-23 own the 24 4x4 blocks of macroblock 0,
        //    lanes 24-47 those of macroblock 1; lanes 48-51 carry the two 32-byte headers)
        // =========================================================================================================
        const uint8_t *fpacked = a.packed + (size_t)frame * W * H * MVHP_MB_BYTES;
        int4 pA = make_int4(0, 0, 0, 0), pB = make_int4(0, 0, 0, 0);
        auto prefetch = [&](int px, int lane_p) {
            pA = make_int4(0, 0, 0, 0);
            pB = make_int4(0, 0, 0, 0);
            if (px >= W) return;
            const uint8_t *rec0 = fpacked + (size_t)(row * W + px) * MVHP_MB_BYTES;
            const bool two = (px + 1) < W;
            const uint8_t *src = nullptr;
            if (lane_p < 24) src = rec0 + MVHP_MB_HEADER_BYTES + lane_p * 32;
            else if (lane_p < 48) { if (two) src = rec0 + MVHP_MB_BYTES + MVHP_MB_HEADER_BYTES + (lane_p - 24) * 32; }
            else if (lane_p < 50) src = rec0 + (lane_p - 48) * 16;
            else if (lane_p < 52) { if (two) src = rec0 + MVHP_MB_BYTES + (lane_p - 50) * 16; }
            if (src) {
                pA = *reinterpret_cast<const int4 *>(src);
                if (lane_p < 48) pB = *reinterpret_cast<const int4 *>(src + 16);
            }
        };
        prefetch(0, lane_c);
#pragma unroll 1
        for (int mbx0 = 0; mbx0 < W; mbx0 += 2) {
            const int npair = min(2, W - mbx0);
            int lane = lane_c;
            asm volatile("" : "+v"(lane));
            const int4 cA = pA, cB = pB;
            prefetch(mbx0 + 2, lane);
            const int slot = (mbx0 >> 1) % NPAIR;
            // the slot is free once K and O have finished with the pair NPAIR pairs back
            if (!p1_wait<MVHP_PIPE_NAP_F>(&C.k_done[r], mbx0 - 2 * NPAIR + 2, C, a.err, lane)) return;
            if (!p1_wait<MVHP_PIPE_NAP_F>(&C.o_done[r], mbx0 - 2 * NPAIR + 2, C, a.err, lane)) return;
            // headers: wave-uniform -> scalars (v_readlane from the header lanes)
            PairCtl pc;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint32_t hh0 = __builtin_amdgcn_readlane((uint32_t)cA.x, 48 + 2 * k);
                const uint32_t hnz = __builtin_amdgcn_readlane((uint32_t)cA.z, 48 + 2 * k);
                pc.kind[k] = hh0 & 255;
                pc.qpy[k] = (hh0 >> 8) & 255;
#pragma unroll
                for (int c = 0; c < 2; c++) { // derivChromaQP, h264_transform.c:598-637
                    int qpi = pc.qpy[k] + (c ? a.cqp_off_cr : a.cqp_off_cb);
                    qpi = min(max(qpi, 0), 51);
                    // Table 8-15 (h264_transform.c:71) as nibbles of (QPC - 29) for qPI = 30..51: scalar arithmetic only
                    const unsigned long long lo = 0x9888776655433210ull, hi = 0xAAAA99ull; // qPI 30..45 | 46..51
                    const int q = qpi - 30;
                    const int nib = (int)(((q < 16) ? (lo >> ((q & 15) * 4)) : (hi >> (((q - 16) & 15) * 4))) & 15ull);
                    const int v = (qpi > 29) ? 29 + nib : qpi;
                    if (c) pc.qpc_cr[k] = v; else pc.qpc_cb[k] = v;
                }
                // Intra16x16 at QP'Y == 36 yields a non-zero DC term even from all-zero levels
                // (h264_transform.c:797-808), so the residual stage cannot be skipped there.
                const bool quirk36 = (pc.kind[k] == MVHP_KIND_I16x16) && (pc.qpy[k] == 36) && (a.dc_shift_from > 36);
                const bool rl = ((hnz & 0xffffu) != 0) || quirk36;
                const bool rc = (hnz & 0xff0000u) != 0;
                pc.need[k] = (rl || rc) && (k < npair) && (pc.kind[k] != MVHP_KIND_IPCM);
            }
            pc.dc_shift_from = a.dc_shift_from;
            if (pc.need[0] || pc.need[1]) residual_pair<EXT>(Rw.res[slot], Rw.scr, B, lane, cA, cB, pc);
            // the headers for K and O (lanes 48-51: 16 bytes each); an I_PCM macroblock's samples travel in the residual area
            // as the record holds them (32 bytes per block lane; K and O copy them out)
            if (lane >= 48 && lane < 52) *reinterpret_cast<int4 *>(&Rw.hdr[slot][(lane - 48) >> 1][((lane - 48) & 1) * 4]) = cA;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (pc.kind[k] == MVHP_KIND_IPCM && k < npair && lane >= 24 * k && lane < 24 * k + 16) {
                    int16_t *dst = &Rw.res[slot][k][(lane - 24 * k) * 16];
                    *reinterpret_cast<int4 *>(dst) = cA;
                    *reinterpret_cast<int4 *>(dst + 8) = cB;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&C.f_done[r], mbx0 + npair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            WAVE_SYNC();
        }
        return;
    }

    if (role == 1) {
        // =========================================================================================================
        // K: luma prediction + residual add (h264_intra_prediction.c:112-2141, transform*_luma), luma neighbour state, the row
        //    dependency -- the chain every other row waits for, and nothing else
        // =========================================================================================================
        unsigned long long seam_pend = 0;
        __builtin_amdgcn_s_setprio(MVHP_PIPE_PRIO_K);
#pragma unroll 1
        for (int mbx = 0; mbx < W; mbx++) {
            int lane = lane_c;
            asm volatile("" : "+v"(lane));
            const int slot = (mbx >> 1) % NPAIR, k = mbx & 1;
            uint8_t *T = Rw.T[mbx % NTILE];
            uint8_t *Tn = Rw.T[(mbx + 1) % NTILE];    // where the next macroblock of the row will be built

            if (seam_in && (mbx & 1) == 0) {
                // Macroblocks mbx and mbx + 1 read luma columns <= mbx + 2 of the row above: the first step of a row fetches columns
                // 0..2 now; every later even step finds (mbx + 1, mbx + 2) asked for two steps ago, and asks for (mbx + 3, mbx + 4).
                // Lane l: column c0 + (l >> 2), luma granule l & 3.
                const int c0 = mbx ? mbx + 1 : 0, ncol = mbx ? 2 : 3;
                const int col = c0 + (lane >> 2), g = lane & 3;
                const bool act = (lane < ncol * 4) && (col < W);
                const unsigned long long *src = seam_rd + (size_t)(act ? col : 0) * SEAM_GRANULES + g;
                unsigned long long v = seam_pend;
                if (mbx == 0) v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!p1_seam_poll(src, act, v, a.wide_epoch, C, a.err, lane)) return;
                if (act) *reinterpret_cast<uint32_t *>(&line_y[col * 16 + g * 4]) = (uint32_t)v;
                const int ncolumn = mbx + 3 + (lane >> 2);
                if (lane < 8 && ncolumn < W)
                    seam_pend = __hip_atomic_load(seam_rd + (size_t)ncolumn * SEAM_GRANULES + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                WAVE_SYNC();
            }

            // the tile is free once O has taken macroblock mbx - NTILE out of it; the residuals of mbx must be in the ring
            if (!p1_wait(&C.o_done[r], mbx - NTILE + 1, C, a.err, lane)) return;
            if (!p1_wait(&C.f_done[r], mbx + 1, C, a.err, lane)) return;
            const int4 hA = *reinterpret_cast<const int4 *>(&Rw.hdr[slot][k][0]);
            const int4 hB = *reinterpret_cast<const int4 *>(&Rw.hdr[slot][k][4]);
            const uint32_t h0 = __builtin_amdgcn_readfirstlane((uint32_t)hA.x), h1 = __builtin_amdgcn_readfirstlane((uint32_t)hA.y);
            const uint32_t hnz = __builtin_amdgcn_readfirstlane((uint32_t)hA.z);
            const uint32_t m0 = __builtin_amdgcn_readfirstlane((uint32_t)hA.w), m1 = __builtin_amdgcn_readfirstlane((uint32_t)hB.x);
            const uint32_t m2 = __builtin_amdgcn_readfirstlane((uint32_t)hB.y), m3 = __builtin_amdgcn_readfirstlane((uint32_t)hB.z);
            const int kind = h0 & 255, qpy = (h0 >> 8) & 255;
            const int i16mode = h1 & 255;
            const bool quirk36 = (kind == MVHP_KIND_I16x16) && (qpy == 36) && (a.dc_shift_from > 36);
            const bool res_luma = ((hnz & 0xffffu) != 0) || quirk36;
            const int16_t *res = Rw.res[slot][k];
            // neighbours: by geometry (h264_spatial.c:333-416), less those in another slice (MVHP_PARAM_SLICES: header byte 6)
            const uint32_t un = (EXT && a.slices) ? ((h1 >> 16) & 255u) : 0u;
            const bool A = (mbx > 0) && !(un & MVHP_UNAVAIL_A), Cav = BvG && (mbx < W - 1) && !(un & MVHP_UNAVAIL_C),
                       D = (mbx > 0) && BvG && !(un & MVHP_UNAVAIL_D);
            const bool Bv = BvG && !(un & MVHP_UNAVAIL_B);

            // wait for the row above: needs columns <= min(mbx+1, W-1); then fetch the top neighbours
            if (BvG) {
                if (!seam_in && !p1_wait<MVHP_WIDE_NAP>(&C.k_done[r - 1], min(mbx + 2, W), C, a.err, lane)) return;
                // lanes 0-3 luma top, 4-5 luma up-right (when the column exists): one dword each
                if (lane < 6 && ((mbx < W - 1) || lane < 4))
                    *reinterpret_cast<uint32_t *>(&T[16 + lane * 4]) = *reinterpret_cast<const uint32_t *>(&line_y[mbx * 16 + lane * 4]);
            }
            WAVE_SYNC();

            if (kind == MVHP_KIND_IPCM) {
                // I_PCM (8.3.5; MVHP_STREAM_SPEC streams only): the owner of luma block 2j holds luma rows 2j and 2j+1
                if (lane < 16 && (lane & 1) == 0) {
                    const int jj = lane >> 1;
                    *reinterpret_cast<int4 *>(&T[(2 * jj + 1) * 32 + 16]) = *reinterpret_cast<const int4 *>(&res[lane * 16]);
                    *reinterpret_cast<int4 *>(&T[(2 * jj + 2) * 32 + 16]) = *reinterpret_cast<const int4 *>(&res[lane * 16 + 8]);
                }
                WAVE_SYNC();
            } else if (kind == MVHP_KIND_I16x16) {
                predict_16x16(T, Rw.Lcol, lane, i16mode, A, Bv, D, res_luma, res);
            } else if (kind == MVHP_KIND_I4x4) {
                predict_mb_4x4(T, B, lane, m0, m1, m2, m3, A, Bv, Cav, D, res_luma, res);
            } else {
                const Edge8 g8 = edge8_of(lane);
                for (int blk = 0; blk < 4; blk++)
                    predict_8x8(T, Rw.E8, B, lane, g8, blk, (m0 >> (blk * 8)) & 255, A, Bv, Cav, D, res_luma, res);
            }

            // ---- luma neighbour state for the next macroblock (built in the NEXT tile) / the next row, then publish ----
            {
                // lane 0: corner (this macroblock's top-right sample); lanes 16-31: right column -> left column of the next tile +
                // Lcol; lanes 48-51: bottom row -> line buffer (+ seam)
                uint32_t keep = 0, bot = 0;
                if (lane == 0) keep = T[31];
                else if (lane >= 16 && lane < 32) keep = T[(lane - 15) * 32 + 31];
                if (lane >= 48 && lane < 52) bot = *reinterpret_cast<const uint32_t *>(&T[16 * 32 + 16 + (lane - 48) * 4]);
                WAVE_SYNC();
                if (lane == 0) Tn[15] = (uint8_t)keep;
                else if (lane >= 16 && lane < 32) { Tn[(lane - 15) * 32 + 15] = (uint8_t)keep; Rw.Lcol[lane - 16] = (uint8_t)keep; }
                if (lane >= 48 && lane < 52) {
                    *reinterpret_cast<uint32_t *>(&line_y[mbx * 16 + (lane - 48) * 4]) = bot;
                    if (seam_out)   // the same four dwords, tagged, to the band below (one write-through store per granule)
                        __hip_atomic_store(seam_wr + (size_t)mbx * SEAM_GRANULES + (lane - 48), seam_tag | bot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&C.k_done[r], mbx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            WAVE_SYNC();
        }
        return;
    }

    // =============================================================================================================
    // O: chroma prediction (h264_intra_prediction.c:2157-2564 + transform4x4_chroma; one macroblock behind the row above's O)
    //    and write-out: the macroblock joins a 4-macroblock output strip in LDS; full strips go to HBM as 64-byte luma /
    //    32-byte chroma row segments plus (fused) the RGB conversion (export.c:65-188, export_utils.c:209-324)
    // =============================================================================================================
    {
        uint8_t *fy = a.yuv + (size_t)frame * W * H * 384;
        uint8_t *fcb = fy + (size_t)W * H * 256;
        uint8_t *fcr = fcb + (size_t)W * H * 64;
        uint8_t *frgb = a.rgb ? a.rgb + (size_t)frame * W * H * 768 : nullptr;
        const int pitch = W * 16, cpitch = W * 8;
        unsigned long long seam_pend = 0;
#pragma unroll 1
        for (int mbx = 0; mbx < W; mbx++) {
            int lane = lane_c;
            asm volatile("" : "+v"(lane));
            const int slot = (mbx >> 1) % NPAIR, k = mbx & 1;
            const uint8_t *T = Rw.T[mbx % NTILE];

            if (seam_in && (mbx & 1) == 0) {
                // Macroblocks mbx and mbx + 1 read chroma columns mbx and mbx + 1 of the row above: the first step fetches (0, 1) now;
                // every later even step finds its two columns asked for two steps ago, and asks for (mbx + 2, mbx + 3).
                // Lane l < 8: column mbx + (l >> 2), granule 4 + (l & 3) (4-5 Cb dwords, 6-7 Cr).
                const int g = 4 + (lane & 3);
                const int col = mbx + ((lane >> 2) & 1);
                const bool act = (lane < 8) && (col < W);
                const unsigned long long *src = seam_rd + (size_t)(act ? col : 0) * SEAM_GRANULES + g;
                unsigned long long v = seam_pend;
                if (mbx == 0) v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!p1_seam_poll(src, act, v, a.wide_epoch, C, a.err, lane)) return;
                if (act) {
                    uint8_t *dst = (g < 6) ? &line_cb[col * 8 + (g - 4) * 4] : &line_cr[col * 8 + (g - 6) * 4];
                    *reinterpret_cast<uint32_t *>(dst) = (uint32_t)v;
                }
                const int ncol = mbx + 2 + ((lane >> 2) & 1);
                if (lane < 8 && ncol < W)
                    seam_pend = __hip_atomic_load(seam_rd + (size_t)ncol * SEAM_GRANULES + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                WAVE_SYNC();
            }

            // header and chroma residuals of this macroblock; the row above's chroma of this column
            if (!p1_wait<MVHP_PIPE_NAP_O>(&C.f_done[r], mbx + 1, C, a.err, lane)) return;
            const uint32_t h0 = __builtin_amdgcn_readfirstlane(Rw.hdr[slot][k][0]), h1 = __builtin_amdgcn_readfirstlane(Rw.hdr[slot][k][1]);
            const uint32_t hnz = __builtin_amdgcn_readfirstlane(Rw.hdr[slot][k][2]);
            const int kind = h0 & 255;
            const int cmode = (h0 >> 24) & 255;
            const bool res_chroma = (hnz & 0xff0000u) != 0 && kind != MVHP_KIND_IPCM;
            const int16_t *res = Rw.res[slot][k];
            const uint32_t un = (EXT && a.slices) ? ((h1 >> 16) & 255u) : 0u;
            const bool A = (mbx > 0) && !(un & MVHP_UNAVAIL_A), D = (mbx > 0) && BvG && !(un & MVHP_UNAVAIL_D);
            const bool Bv = BvG && !(un & MVHP_UNAVAIL_B);
            if (BvG) {
                if (!seam_in && !p1_wait<MVHP_PIPE_NAP_O>(&C.c_done[r - 1], mbx + 1, C, a.err, lane)) return;
                if (lane < 4) {   // lanes 0-1 Cb top, 2-3 Cr top: one dword each
                    const uint8_t *src = (lane < 2) ? &line_cb[mbx * 8 + lane * 4] : &line_cr[mbx * 8 + (lane - 2) * 4];
                    *reinterpret_cast<uint32_t *>(&Rw.TC[lane >> 1][8 + (lane & 1) * 4]) = *reinterpret_cast<const uint32_t *>(src);
                }
            }
            WAVE_SYNC();
            predict_chroma(Rw.TC, Rw.LcolC, lane, cmode, A, Bv, D, res_chroma, res);
            if (kind == MVHP_KIND_IPCM) {
                // I_PCM: the owner of luma block 2j + 1 holds Cb row j and Cr row j, over what the prediction made
                if (lane < 16 && (lane & 1)) {
                    const int jj = lane >> 1;
                    const int4 sA = *reinterpret_cast<const int4 *>(&res[lane * 16]);
                    *reinterpret_cast<int2 *>(&Rw.TC[0][(jj + 1) * 16 + 8]) = make_int2(sA.x, sA.y);
                    *reinterpret_cast<int2 *>(&Rw.TC[1][(jj + 1) * 16 + 8]) = make_int2(sA.z, sA.w);
                }
                WAVE_SYNC();
            }

            // ---- the luma of this macroblock: into the strip ----
            if (!p1_wait<MVHP_PIPE_NAP_O>(&C.k_done[r], mbx + 1, C, a.err, lane)) return;
            const int mbi = mbx & 3;
            {
                const int y = lane >> 2, q = lane & 3;
                *reinterpret_cast<uint32_t *>(&Rw.SY[y * 64 + mbi * 16 + q * 4]) = *reinterpret_cast<const uint32_t *>(&T[(y + 1) * 32 + 16 + q * 4]);
                if (lane < 32) {
                    const int pl = lane >> 4, cy = (lane & 15) >> 1, hf = lane & 1;
                    *reinterpret_cast<uint32_t *>(&Rw.SC[pl][cy * 32 + mbi * 8 + hf * 4]) =
                        *reinterpret_cast<const uint32_t *>(&Rw.TC[pl][(cy + 1) * 16 + 8 + hf * 4]);
                }
            }
            // the luma tile and the ring slot have been read: K may build macroblock mbx + NTILE in the tile, F may refill the slot
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&C.o_done[r], mbx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

            // ---- chroma neighbour state for the next macroblock / the next row, then publish the chroma ----
            {
                // lanes 1-2: corners; lanes 32-47: right columns -> left columns + LcolC; lanes 52-55: bottom rows -> line buffer
                uint32_t keep = 0, bot = 0;
                uint8_t *bdst = line_cb;
                if (lane == 1 || lane == 2) keep = Rw.TC[lane - 1][15];
                else if (lane >= 32 && lane < 48) keep = Rw.TC[(lane - 32) >> 3][(((lane - 32) & 7) + 1) * 16 + 15];
                if (lane >= 52 && lane < 54) { bot = *reinterpret_cast<const uint32_t *>(&Rw.TC[0][8 * 16 + 8 + (lane - 52) * 4]); bdst = &line_cb[mbx * 8 + (lane - 52) * 4]; }
                else if (lane >= 54 && lane < 56) { bot = *reinterpret_cast<const uint32_t *>(&Rw.TC[1][8 * 16 + 8 + (lane - 54) * 4]); bdst = &line_cr[mbx * 8 + (lane - 54) * 4]; }
                WAVE_SYNC();
                if (lane == 1 || lane == 2) Rw.TC[lane - 1][7] = (uint8_t)keep;
                else if (lane >= 32 && lane < 48) {
                    const int pl = (lane - 32) >> 3, cy = (lane - 32) & 7;
                    Rw.TC[pl][(cy + 1) * 16 + 7] = (uint8_t)keep;
                    Rw.LcolC[pl][cy] = (uint8_t)keep;
                }
                if (lane >= 52 && lane < 56) {
                    *reinterpret_cast<uint32_t *>(bdst) = bot;
                    if (seam_out)   // granules 4-7 of the column, tagged, to the band below
                        __hip_atomic_store(seam_wr + (size_t)mbx * SEAM_GRANULES + (lane - 48), seam_tag | bot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&C.c_done[r], mbx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

            if (mbi == 3 || mbx == W - 1) {
                WAVE_SYNC();
                const int x0 = mbx - mbi, nb = (mbi + 1) * 16; // strip origin (MB units), width in samples
                {   // luma: lane -> 16 bytes of one row
                    const int y = lane >> 2, part = (lane & 3) * 16;
                    if (part < nb)
                        *reinterpret_cast<uint4 *>(&fy[(size_t)(row * 16 + y) * pitch + x0 * 16 + part]) =
                            *reinterpret_cast<const uint4 *>(&Rw.SY[y * 64 + part]);
                }
                {   // chroma: lane -> 8 bytes of one row of one plane
                    const int pl = lane >> 5, cy = (lane >> 2) & 7, part = (lane & 3) * 8;
                    if (part < (nb >> 1))
                        *reinterpret_cast<uint2 *>((pl ? fcr : fcb) + (size_t)(row * 8 + cy) * cpitch + x0 * 8 + part) =
                            *reinterpret_cast<const uint2 *>(&Rw.SC[pl][cy * 32 + part]);
                }
                if (frgb) {
                    // mb_to_rgb (export_utils.c:209-324) on the strip: 2x2 nearest chroma, integer formula :300-302
                    const int x4 = (lane & 15) * 4;
                    if (x4 < nb) {
#pragma unroll 2
                        for (int i = 0; i < 4; i++) {
                            const int y = i * 4 + (lane >> 4);
                            const uint32_t yw = *reinterpret_cast<const uint32_t *>(&Rw.SY[y * 64 + x4]);
                            const uint32_t cbw = *reinterpret_cast<const uint16_t *>(&Rw.SC[0][(y >> 1) * 32 + (x4 >> 1)]);
                            const uint32_t crw = *reinterpret_cast<const uint16_t *>(&Rw.SC[1][(y >> 1) * 32 + (x4 >> 1)]);
                            int d0, d1, d2;   // packed 16-bit arithmetic, see recon_batch_device.h rgb4()
                            rgb4(yw, bytes01(cbw), bytes01(crw), d0, d1, d2);
                            // one 12-byte store per lane: the 16 lanes of a row cover its 192 bytes in one instruction
                            typedef int v3i __attribute__((ext_vector_type(3)));
                            typedef v3i v3i_a4 __attribute__((aligned(4)));
                            *reinterpret_cast<v3i_a4 *>(frgb + ((size_t)(row * 16 + y) * pitch + x0 * 16 + x4) * 3) = v3i{d0, d1, d2};
                        }
                    }
                }
            }
            WAVE_SYNC();
        }
    }
}

} // namespace p1

size_t recon_pipe1_lds_bytes(int width_mbs, int rows)
{
    return sizeof(p1::P1Tables) + sizeof(p1::P1Ctl) + (size_t)width_mbs * 32 + (size_t)rows * sizeof(p1::P1Row);
}

template <int R, bool EXT>
static hipError_t launch_pipe1_one(const ReconArgs &a, hipStream_t stream)
{
    const int bands = (a.height_mbs + R - 1) / R;
    const size_t lds = recon_pipe1_lds_bytes(a.width_mbs, R);
    hipError_t e = hipFuncSetAttribute((const void *)p1::recon_pipe1_kernel<R, EXT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((p1::recon_pipe1_kernel<R, EXT>), dim3(a.n_frames * bands), dim3(R * 3 * 64), lds, stream, a);
    return hipGetLastError();
}

// one workgroup per (band of `rows` rows, picture); a.wide_ticket / wide_base / wide_epoch / seam set by the caller
hipError_t launch_recon_pipe1(const ReconArgs &a, int rows, hipStream_t stream)
{
    if (!a.wide_ticket || !a.wide_epoch) return hipErrorInvalidValue;
    if ((a.height_mbs + rows - 1) / rows > 1 && !a.seam) return hipErrorInvalidValue;
    const bool ext = a.slices || a.scaling;
    switch (rows) {
    case 1: return ext ? launch_pipe1_one<1, true>(a, stream) : launch_pipe1_one<1, false>(a, stream);
    case 2: return ext ? launch_pipe1_one<2, true>(a, stream) : launch_pipe1_one<2, false>(a, stream);
    case 4: return ext ? launch_pipe1_one<4, true>(a, stream) : launch_pipe1_one<4, false>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace mvhp
