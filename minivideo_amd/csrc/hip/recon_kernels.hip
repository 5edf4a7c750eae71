// recon_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the H.264 intra
// reconstruction hot path.  Integer stencil work: no MFMA, HBM/LDS-latency bound.
//
//   recon_rows_kernel<NW, EXT> replaces intra_prediction_process()
//                           (decoder/h264/h264_intra_prediction.c:112-145) for a
//                           whole picture: all prediction modes, dequantisation,
//                           4x4/8x8 IDCT, DC transforms, residual add + clip and
//                           the planar gather of export.c:65-188.
//   ycbcr_to_rgb_kernel     replaces mb_to_rgb() (export_utils.c:209-324).
//
// Mapping: one workgroup per picture, NW wavefronts (64 lanes) per workgroup,
// one wavefront per macroblock row (wave w owns rows w, w+NW, ...).  Row r may
// reconstruct macroblock x once row r-1 has published x+2 (left/up/up-left/
// up-right dependencies); publication is a per-wave counter in LDS, so the whole
// dependency protocol stays inside one CU (no agent-scope fences).  The bottom
// sample row of every macroblock row lives in ONE LDS line buffer per picture
// (each row overwrites column x after it has consumed it), the left column and
// the macroblock being built live in a per-wave LDS tile.  Residuals
// (dequantised + inverse-transformed coefficients) are staged through LDS as
// int16; raw coefficients go global -> registers of the lane that owns the 4x4
// block (800 contiguous bytes per macroblock, read exactly once).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "minivideo_hotpath.h"
#include "recon_kernels.h"
#include "recon_device.h"
#include "recon_batch_device.h"
#include "recon_rows_device.h"

// wave priorities inside a step: as recon_oct.hip (the chains of LDS round trips first, residuals and colour conversion fill in)
#ifndef MVHP_ROWS_PRIO
#define MVHP_ROWS_PRIO 1     // 0: everything at priority 0, as rounds 1-3 (K1w, 256 x 1080p: 2.64 -> 2.48 ms High, 2.52 -> 2.36 Baseline)
#endif
#define MVHP_RP_PRED (MVHP_ROWS_PRIO ? 1 : 0)
#define MVHP_RP_TAIL (MVHP_ROWS_PRIO ? 2 : 0)
#define MVHP_RP_CHAIN (MVHP_ROWS_PRIO ? 3 : 0)
#ifndef MVHP_WIDE_NAP
#define MVHP_WIDE_NAP 1   // s_sleep units between polls of the row above in the banded instantiations (see recon_quad.hip)
#endif

namespace mvhp {

// ---------------------------------------------------------------------------
// LDS layout
// ---------------------------------------------------------------------------
struct __attribute__((aligned(16))) WaveLds {
    int16_t res[2][384];   // residuals of a macroblock pair, MB raster: luma y*16+x | 256+Cb y*8+x | 320+Cr
    uint8_t T[17 * 32 + 16]; // luma tile: row 0 = top neighbours; byte 15 = left/corner, 16..31 samples;
                           // row 0 bytes 32..39 = up-right neighbours
    uint8_t TC[2][9 * 16]; // chroma tiles: row 0 = top; byte 7 = left/corner, 8..15 samples
    uint8_t Lcol[16];      // compact left neighbour column (luma)
    uint8_t LcolC[2][8];   // compact left neighbour columns (Cb, Cr)
    uint8_t E8[32];        // filtered Intra8x8 edge: [0..1]=rep left7, [2+j]=left[7-j], [10]=corner, [11+i]=top[i], [27]=rep
    int32_t scr[256];      // 8x8 transpose scratch / DC exchange
    uint8_t SY[16 * 64];   // output strip: 4 macroblocks of reconstructed luma (flushed with wide stores)
    uint8_t SC[2][8 * 32]; // output strip: 4 macroblocks of Cb / Cr
};

struct __attribute__((aligned(16))) BlockLds {
    int     progress[16];  // macroblocks completed by wave w (monotonic over its rows)
    int     abort_flag;
    int     unit;          // WIDE: this workgroup's ticket (picture * bands + band)
    int     pad[2];
    int     ls4[18];       // LevelScale4x4 classes, 16*normAdjust (h264.c:427-435)
    int     ls8[36];       // LevelScale8x8 classes (h264.c:438-446)
    uint8_t cls8[64];      // 8x8 position -> class
    uint8_t w4[3][16];     // weight matrices (raster), 16 = flat: LevelScale = weight * normAdjust (h264_transform.c:645-741)
    uint8_t w8[64];
    uint32_t tap4[2 * 9 * 16]; // Intra4x4: [up-right unavailable][mode][sample] -> three byte offsets into the tile
    uint32_t tap8[9 * 64];     // Intra8x8: [mode][sample] -> three indices into the filtered edge array E8
};

// The residual stage of a macroblock pair and the Intra8x8 / Intra16x16 / chroma prediction of a macroblock: recon_rows_device.h
// (shared with recon_pipe1_kernel, which gives a row three wavefronts).  Here one wavefront does everything for its row, and
// the Intra4x4 chain below keeps its sixteen steps on sixteen lanes: three more waves share the SIMD and fill its waits.
using namespace rowsdev;

// Intra 4x4 macroblock: 16 dependent block steps, lanes 0..15 own one sample each.
// h264_intra_prediction.c:161-177, :315-483, :496-960 + transform4x4_luma (h264_transform.c:121-156).
__device__ __forceinline__ void predict_mb_4x4(WaveLds &W, const BlockLds &B, int lane, uint32_t m0, uint32_t m1,
                                               uint32_t m2, uint32_t m3, bool A, bool Bv, bool C, bool D, bool has_res,
                                               const int16_t *res)
{
    const Avail4 av = avail4(A, Bv, C, D);
    // neighbours each mode needs, 3 bits per mode: bit0 left, bit1 up, bit2 up-left (mode 2 = DC handled apart)
    constexpr uint32_t REQ = (2u << 0) | (1u << 3) | (0u << 6) | (2u << 9) | (7u << 12) | (7u << 15) | (7u << 18) |
                             (2u << 21) | (1u << 24);
    const int rmask = has_res ? -1 : 0;
    // Per-block control word, computed once by lane b for block b (16 lanes in parallel) and handed to the
    // block steps with v_readlane: bits 0-1 left/up available, bit 2 mode is DC, bit 3 prediction allowed,
    // bits 8.. byte offset of the block's row in the tap table.
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
    if (lane < 16) {
        const int pix = (lane >> 2) * 32 + (lane & 3);   // this lane's sample inside a block, tile units
        const int rpix = (lane >> 2) * 16 + (lane & 3);  // same in the residual array
        const uint8_t *T = W.T;
        const uint8_t *tapb = reinterpret_cast<const uint8_t *>(B.tap4) + lane * 4;
        // software pipeline: the table entry and the residual of block b+1 are fetched before block b's
        // dependent tile reads, so only (tile read -> combine -> tile write) sits on the per-block chain
        uint32_t inf = __builtin_amdgcn_readlane(info, 0);
        uint32_t e_nx = *reinterpret_cast<const uint32_t *>(tapb + (inf >> 8));
        int r_nx = (int)res[rpix];
#pragma unroll
        for (int blk = 0; blk < 16; blk++) {
            const int xO = (((blk >> 2) & 1) << 3) | ((blk & 1) << 2);
            const int yO = ((blk >> 3) << 3) | (((blk >> 1) & 1) << 2);
            const int base = (yO + 1) * 32 + 16 + xO;     // tile index of the block's top-left sample
            const uint32_t cur = inf;
            const uint32_t e = e_nx;
            const int r = r_nx & rmask;
            if (blk < 15) {
                const int nb = blk + 1;
                const int nxO = (((nb >> 2) & 1) << 3) | ((nb & 1) << 2), nyO = ((nb >> 3) << 3) | (((nb >> 1) & 1) << 2);
                inf = __builtin_amdgcn_readlane(info, nb);
                e_nx = *reinterpret_cast<const uint32_t *>(tapb + (inf >> 8));
                r_nx = (int)res[nyO * 16 + nxO + rpix];
            }
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
            W.T[base + pix] = (uint8_t)clip255(pred + r);
            WAVE_SYNC();
        }
    }
    WAVE_SYNC();
}

// ---------------------------------------------------------------------------
// the reconstruction kernel
// ---------------------------------------------------------------------------
// EXT: pictures of several slices and / or scaling matrices (MVHP_PARAM_SLICES, MVHP_PARAM_SCALING: MVHP_STREAM_SPEC streams,
// SURVEY 8f row f4) -- an instantiation of its own; the ordinary one is the round-2 kernel plus the I_PCM copy.
// WIDE: a workgroup reconstructs ONE BAND of a picture -- NW consecutive macroblock rows, one per wavefront, a single pass --
// and a picture's bands run on different CUs ("grid = F x PicHeightInMbs wavefronts"): sixteen 1080p
// pictures fill the 256 CUs where the one-workgroup-per-picture form needs 256.  Inside a band nothing changes (progress
// counters and line buffer in LDS).  Across a band boundary ("seam") the bottom samples of the upper band's last row travel
// through global memory as 8-byte granules {dword of samples, tag of this launch}, each written by ONE agent-scope (sc1,
// write-through) store and polled with agent-scope loads: a granule is there or it is not, so there is no separate flag,
// no fence and no wait on the producer's side (MI355X_MICROARCH "handoff-1to1": data-tagged granules).  The consumer asks for
// the columns of the NEXT macroblock pair before it starts on this one, so in the steady state the hand-off's latency hides
// behind a pair's work.  Units (picture, band) are handed out by a ticket counter in the order workgroups start: the band
// above is always held by a workgroup that is already running, whatever order the hardware dispatches in.
template <int NW, bool EXT, bool WIDE>
__global__ __launch_bounds__(NW * 64) void recon_rows_kernel(ReconArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int W = a.width_mbs, H = a.height_mbs;
    BlockLds &B = *reinterpret_cast<BlockLds *>(smem);
    uint8_t *line_y = smem + sizeof(BlockLds);
    uint8_t *line_cb = line_y + W * 16;
    uint8_t *line_cr = line_cb + W * 8;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane_c = threadIdx.x & 63;
    const int lane = lane_c;
    WaveLds &Wv = *reinterpret_cast<WaveLds *>(line_cr + W * 8 + (size_t)wave * sizeof(WaveLds));
    if (WIDE && threadIdx.x == 0) B.unit = (int)(atomicAdd(a.wide_ticket, 1u) - a.wide_base);

    // ---- one-time table setup ----
    for (int i = threadIdx.x; i < 18; i += NW * 64) B.ls4[i] = 16 * c_v4x4[i];
    for (int i = threadIdx.x; i < 36; i += NW * 64) B.ls8[i] = 16 * c_v8x8[i];
    for (int i = threadIdx.x; i < 64; i += NW * 64) {
        const int r = i >> 3, c = i & 7;
        int k;
        if ((r % 4 == 0) && (c % 4 == 0)) k = 0;
        else if ((r % 2 == 1) && (c % 2 == 1)) k = 1;
        else if ((r % 4 == 2) && (c % 4 == 2)) k = 2;
        else if (((r % 4 == 0) && (c % 2 == 1)) || ((r % 2 == 1) && (c % 4 == 0))) k = 3;
        else if (((r % 4 == 0) && (c % 4 == 2)) || ((r % 4 == 2) && (c % 4 == 0))) k = 4;
        else k = 5;
        B.cls8[i] = (uint8_t)k;
    }
    for (int i = threadIdx.x; i < 2 * 9 * 16; i += NW * 64)
        B.tap4[i] = tap4_entry((i >> 4) % 9, i & 3, (i >> 2) & 3, i >= 9 * 16);
    for (int i = threadIdx.x; i < 9 * 64; i += NW * 64) B.tap8[i] = tap8_entry(i >> 6, i & 7, (i >> 3) & 7);
    if (EXT)
        for (int i = threadIdx.x; i < 112; i += NW * 64) (&B.w4[0][0])[i] = a.scaling ? a.weights[i] : (uint8_t)16;   // w4 | w8 are adjacent
    if (threadIdx.x < 16) B.progress[threadIdx.x] = 0;
    if (threadIdx.x == 16) B.abort_flag = 0;
    __syncthreads();

    const int bands = (H + NW - 1) / NW;
    const int unit = WIDE ? __builtin_amdgcn_readfirstlane(B.unit) : 0;
    // band-major: band b of every picture before band b + 1 of any -- when the batch outnumbers the resident workgroups, a
    // band then starts about when the band above it has got ahead, instead of holding a slot idle from the launch on
    const int band = WIDE ? unit / a.n_frames : 0;
    const int frame = WIDE ? unit - band * a.n_frames : (int)blockIdx.x;
    if (WIDE && (unsigned)unit >= (unsigned)(bands * a.n_frames)) {   // a ticket outside the launch: the host's bookkeeping of the counter is off
        if (threadIdx.x == 0) atomicOr(a.err, 2u);
        return;
    }
    const int row_first = WIDE ? band * NW : 0;
    const int row_end = WIDE ? min(H, row_first + NW) : H;
    // seams: the first row of a band below the first takes its top neighbours from the seam above; the last row of a band
    // above the last one feeds the seam below
    const bool seam_in = WIDE && wave == 0 && band > 0;
    const bool seam_out = WIDE && wave == NW - 1 && row_first + NW < H;
    const unsigned long long *seam_rd = seam_in ? a.seam + (size_t)(frame * (bands - 1) + band - 1) * W * SEAM_GRANULES : nullptr;
    unsigned long long *seam_wr = seam_out ? a.seam + (size_t)(frame * (bands - 1) + band) * W * SEAM_GRANULES : nullptr;
    const unsigned long long seam_tag = (unsigned long long)a.wide_epoch << 32;

    const uint8_t *fpacked = a.packed + (size_t)frame * W * H * MVHP_MB_BYTES;
    uint8_t *fy = a.yuv + (size_t)frame * W * H * 384;
    uint8_t *fcb = fy + (size_t)W * H * 256;
    uint8_t *fcr = fcb + (size_t)W * H * 64;
    uint8_t *frgb = a.rgb ? a.rgb + (size_t)frame * W * H * 768 : nullptr;
    const int pitch = W * 16, cpitch = W * 8;
    const int up_wave = (wave + NW - 1) % NW;

    // Per-lane copy plans (fixed for the whole kernel) so that the per-macroblock neighbour traffic is a
    // couple of predicated LDS moves instead of a ladder of lane-range branches.
    //  top fetch : lanes 0-3 luma top, 4-5 luma up-right, 6-7 Cb top, 8-9 Cr top (one dword each)
    //  keep      : lanes 0-2 corners, 16-31 luma right column, 32-47 chroma right columns (one byte each)
    //  bottom    : lanes 48-51 luma, 52-53 Cb, 54-55 Cr bottom rows -> line buffer (one dword each)
    uint8_t *top_dst = Wv.T, *keep_src = Wv.T, *keep_dst = Wv.T, *keep_dst2 = Wv.T, *bot_src = Wv.T, *bot_dst = line_y;
    const uint8_t *top_src = line_y;
    int top_mul = 0, bot_mul = 0;
    if (lane < 4) { top_dst = &Wv.T[16 + lane * 4]; top_src = &line_y[lane * 4]; top_mul = 16; }
    else if (lane < 6) { top_dst = &Wv.T[32 + (lane - 4) * 4]; top_src = &line_y[16 + (lane - 4) * 4]; top_mul = 16; }
    else if (lane < 8) { top_dst = &Wv.TC[0][8 + (lane - 6) * 4]; top_src = &line_cb[(lane - 6) * 4]; top_mul = 8; }
    else if (lane < 10) { top_dst = &Wv.TC[1][8 + (lane - 8) * 4]; top_src = &line_cr[(lane - 8) * 4]; top_mul = 8; }
    const bool keep_act = (lane < 3) || (lane >= 16 && lane < 48);
    if (lane == 0) { keep_src = &Wv.T[31]; keep_dst = keep_dst2 = &Wv.T[15]; }
    else if (lane == 1) { keep_src = &Wv.TC[0][15]; keep_dst = keep_dst2 = &Wv.TC[0][7]; }
    else if (lane == 2) { keep_src = &Wv.TC[1][15]; keep_dst = keep_dst2 = &Wv.TC[1][7]; }
    else if (lane >= 16 && lane < 32) {
        keep_src = &Wv.T[(lane - 15) * 32 + 31]; keep_dst = &Wv.T[(lane - 15) * 32 + 15]; keep_dst2 = &Wv.Lcol[lane - 16];
    } else if (lane >= 32 && lane < 48) {
        const int pl = (lane - 32) >> 3, cy = (lane - 32) & 7;
        keep_src = &Wv.TC[pl][(cy + 1) * 16 + 15]; keep_dst = &Wv.TC[pl][(cy + 1) * 16 + 7]; keep_dst2 = &Wv.LcolC[pl][cy];
    }
    const bool bot_act = lane >= 48 && lane < 56;
    if (lane >= 48 && lane < 52) { bot_src = &Wv.T[16 * 32 + 16 + (lane - 48) * 4]; bot_dst = &line_y[(lane - 48) * 4]; bot_mul = 16; }
    else if (lane >= 52 && lane < 54) { bot_src = &Wv.TC[0][8 * 16 + 8 + (lane - 52) * 4]; bot_dst = &line_cb[(lane - 52) * 4]; bot_mul = 8; }
    else if (lane >= 54 && lane < 56) { bot_src = &Wv.TC[1][8 * 16 + 8 + (lane - 54) * 4]; bot_dst = &line_cr[(lane - 54) * 4]; bot_mul = 8; }

    // Packed records are prefetched one macroblock PAIR ahead, straight into the registers of the lanes that
    // consume them: lane L < 48 owns 4x4 block (L % 24) of macroblock (L / 24) of the pair = 32 bytes
    // (pA, pB); lanes 48-51 carry the two 32-byte headers (16 bytes each, pA).  Each 800-byte record is read
    // exactly once, as two instructions of 16-byte pieces.
    int4 pA = make_int4(0, 0, 0, 0), pB = make_int4(0, 0, 0, 0);
    auto prefetch = [&](int prow, int px, int lane_c) {
        pA = make_int4(0, 0, 0, 0);
        pB = make_int4(0, 0, 0, 0);
        if (prow >= row_end) return;
        const uint8_t *rec0 = fpacked + (size_t)(prow * W + px) * MVHP_MB_BYTES;
        const bool two = (px + 1) < W;
        const uint8_t *src = nullptr;
        if (lane_c < 24) src = rec0 + MVHP_MB_HEADER_BYTES + lane_c * 32;
        else if (lane_c < 48) { if (two) src = rec0 + MVHP_MB_BYTES + MVHP_MB_HEADER_BYTES + (lane_c - 24) * 32; }
        else if (lane_c < 50) src = rec0 + (lane_c - 48) * 16;
        else if (lane_c < 52) { if (two) src = rec0 + MVHP_MB_BYTES + (lane_c - 50) * 16; }
        if (src) {
            pA = *reinterpret_cast<const int4 *>(src);
            if (lane_c < 48) pB = *reinterpret_cast<const int4 *>(src + 16);
        }
    };
    prefetch(row_first + wave, 0, lane_c);

    int done = 0; // macroblocks completed by this wave
    unsigned long long seam_pend = 0;   // WIDE, seam_in: the granule this lane requested for the next macroblock pair
    for (int row = row_first + wave; row < row_end; row += NW) {
        const int pass = WIDE ? 0 : row / NW;
        const int up_base = ((wave == 0) ? (pass - 1) : pass) * W; // MBs the upper wave finished before its row (row-1)
        const bool Bv = row > 0;
        for (int mbx0 = 0; mbx0 < W; mbx0 += 2) {
            const int npair = min(2, W - mbx0);
            int lane_p = lane_c;
            asm volatile("" : "+v"(lane_p)); // see the per-macroblock copy below
            const int4 cA = pA, cB = pB;
            if (WIDE && seam_in) {
                // This pair reads columns <= mbx0 + 2 of the row above.  The first pair of a row asks for columns 0..2 now;
                // every later pair finds (mbx0 + 1, mbx0 + 2) requested one pair ago.  Lane l: column c0 + (l >> 3), granule
                // l & 7 (0-3 luma dwords, 4-5 Cb, 6-7 Cr).  Then the request for the next pair, (mbx0 + 3, mbx0 + 4).
                const int c0 = mbx0 ? mbx0 + 1 : 0, ncol = mbx0 ? 2 : 3;
                const int col = c0 + (lane_p >> 3), g = lane_p & 7;
                const bool act = (lane_p < ncol * 8) && (col < W);
                const unsigned long long *src = seam_rd + (size_t)(act ? col : 0) * SEAM_GRANULES + g;
                unsigned long long v = seam_pend;
                if (mbx0 == 0) v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__builtin_amdgcn_ballot_w64(act && (v >> 32) != (unsigned long long)a.wide_epoch) != 0) {
                    __builtin_amdgcn_s_sleep(2);
                    // bounded: 2^20 polls of ~1 us; a failure anywhere in the launch (error word) ends every wait
                    bool stop = ++spins > (1 << 20) || __hip_atomic_load(&B.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (!stop && (spins & 255) == 0) stop = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                    if (stop) {
                        if (lane_p == 0) { __hip_atomic_store(&B.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicOr(a.err, 1u); }
                        return;
                    }
                    v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (act) {
                    uint8_t *dst = (g < 4) ? &line_y[col * 16 + g * 4] : (g < 6) ? &line_cb[col * 8 + (g - 4) * 4] : &line_cr[col * 8 + (g - 6) * 4];
                    *reinterpret_cast<uint32_t *>(dst) = (uint32_t)v;
                }
                const int ncolumn = mbx0 + 3 + (lane_p >> 3);
                if (lane_p < 16 && ncolumn < W)
                    seam_pend = __hip_atomic_load(seam_rd + (size_t)ncolumn * SEAM_GRANULES + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                WAVE_SYNC();
            }
            {   // next pair of this wave: same row, or the first pair of its next row
                int nrow = row, nx = mbx0 + 2;
                if (nx >= W) { nrow = row + NW; nx = 0; }   // (WIDE: a wave has one row -- prefetch() answers zeros beyond row_end)
                prefetch(nrow, nx, lane_p);
            }
            // headers: wave-uniform -> scalars (v_readlane from the header lanes)
            uint32_t hh0[2], hh1[2], hnz[2], hm0[2], hm1[2], hm2[2], hm3[2];
            PairCtl pc;
            bool rl[2], rc[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                hh0[k] = __builtin_amdgcn_readlane((uint32_t)cA.x, 48 + 2 * k);
                hh1[k] = __builtin_amdgcn_readlane((uint32_t)cA.y, 48 + 2 * k);
                hnz[k] = __builtin_amdgcn_readlane((uint32_t)cA.z, 48 + 2 * k);
                hm0[k] = __builtin_amdgcn_readlane((uint32_t)cA.w, 48 + 2 * k);
                hm1[k] = __builtin_amdgcn_readlane((uint32_t)cA.x, 49 + 2 * k);
                hm2[k] = __builtin_amdgcn_readlane((uint32_t)cA.y, 49 + 2 * k);
                hm3[k] = __builtin_amdgcn_readlane((uint32_t)cA.z, 49 + 2 * k);
                pc.kind[k] = hh0[k] & 255;
                pc.qpy[k] = (hh0[k] >> 8) & 255;
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
                rl[k] = ((hnz[k] & 0xffffu) != 0) || quirk36;
                rc[k] = (hnz[k] & 0xff0000u) != 0;
                pc.need[k] = (rl[k] || rc[k]) && (k < npair);
            }
            pc.dc_shift_from = a.dc_shift_from;
            if (MVHP_ROWS_PRIO) __builtin_amdgcn_s_setprio(0);
            if (pc.need[0] || pc.need[1]) residual_pair<EXT>(Wv.res, Wv.scr, B, lane_p, cA, cB, pc);

#pragma unroll 1
            for (int k = 0; k < npair; k++) {
            const int mbx = mbx0 + k;
            // Re-materialise the lane id every macroblock: it stops the compiler from hoisting hundreds of
            // lane-dependent LDS addresses out of this loop (128+ VGPRs, 4 waves/SIMD) at the price of a few
            // recomputed adds (71 VGPRs, 6-7 waves/SIMD).
            int lane_v = lane_c;
            asm volatile("" : "+v"(lane_v));
            const int lane = lane_v;
            const uint32_t h0 = k ? hh0[1] : hh0[0], h1 = k ? hh1[1] : hh1[0];
            const uint32_t m0 = k ? hm0[1] : hm0[0], m1 = k ? hm1[1] : hm1[0], m2 = k ? hm2[1] : hm2[0], m3 = k ? hm3[1] : hm3[0];
            const int kind = h0 & 255;
            const int cmode = (h0 >> 24) & 255, i16mode = h1 & 255;
            const bool res_luma = k ? rl[1] : rl[0], res_chroma = k ? rc[1] : rc[0];
            const int16_t *res = Wv.res[k];
            // neighbours: by geometry (h264_spatial.c:333-416), less those in another slice (MVHP_PARAM_SLICES: header byte 6)
            const uint32_t un = (EXT && a.slices) ? ((h1 >> 16) & 255u) : 0u;
            const bool BvG = Bv;   // the row above exists: what the wait and the fetch below go by
            const bool A = (mbx > 0) && !(un & MVHP_UNAVAIL_A), C = BvG && (mbx < W - 1) && !(un & MVHP_UNAVAIL_C),
                       D = (mbx > 0) && BvG && !(un & MVHP_UNAVAIL_D);
            const bool Bv = BvG && !(un & MVHP_UNAVAIL_B);

            // ---- wait for the row above: needs columns <= min(mbx+1, W-1) ----
            if (MVHP_ROWS_PRIO) __builtin_amdgcn_s_setprio(MVHP_RP_PRED);
            if (BvG) {
                const int need = (WIDE && seam_in) ? 0 : up_base + min(mbx + 2, W);   // (seam_in: the pair's columns are in the line buffer)
                int spins = 0;
                while (__hip_atomic_load(&B.progress[up_wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                    __builtin_amdgcn_s_sleep(WIDE ? MVHP_WIDE_NAP : 1);
                    if (++spins > (1 << 22) || __hip_atomic_load(&B.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        if (lane == 0) { __hip_atomic_store(&B.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicOr(a.err, 1u); }
                        return;
                    }
                }
                asm volatile("" ::: "memory");
                // top neighbours: luma 16 + 8 up-right (when C), chroma 8 + 8
                if (lane < 10 && ((mbx < W - 1) || (lane >> 1) != 2))
                    *reinterpret_cast<uint32_t *>(top_dst) = *reinterpret_cast<const uint32_t *>(top_src + mbx * top_mul);
            }
            WAVE_SYNC();

            if (kind == MVHP_KIND_IPCM) {
                // I_PCM (8.3.5; MVHP_STREAM_SPEC streams only): the samples as they are.  Record layout (minivideo_hotpath.h):
                // the owner of luma block 2j holds luma rows 2j and 2j+1, the owner of block 2j+1 Cb row j and Cr row j.
                // (read again from the record -- a rare path; keeping the prefetched registers alive for it would cost every
                //  macroblock eight registers and the kernel a wave per SIMD)
                if (lane < 16) {
                    const int jj = lane >> 1;
                    const uint8_t *src = fpacked + (size_t)(row * W + mbx) * MVHP_MB_BYTES + MVHP_MB_HEADER_BYTES + lane * 32;
                    const int4 sA = *reinterpret_cast<const int4 *>(src);
                    if ((lane & 1) == 0) {
                        const int4 sB = *reinterpret_cast<const int4 *>(src + 16);
                        *reinterpret_cast<int4 *>(&Wv.T[(2 * jj + 1) * 32 + 16]) = sA;
                        *reinterpret_cast<int4 *>(&Wv.T[(2 * jj + 2) * 32 + 16]) = sB;
                    } else {
                        *reinterpret_cast<int2 *>(&Wv.TC[0][(jj + 1) * 16 + 8]) = make_int2(sA.x, sA.y);
                        *reinterpret_cast<int2 *>(&Wv.TC[1][(jj + 1) * 16 + 8]) = make_int2(sA.z, sA.w);
                    }
                }
                WAVE_SYNC();
            } else {
            // ---- luma ----
            if (kind == MVHP_KIND_I16x16) {
                predict_16x16(Wv.T, Wv.Lcol, lane, i16mode, A, Bv, D, res_luma, res);
            } else if (kind == MVHP_KIND_I4x4) {
                if (MVHP_ROWS_PRIO) __builtin_amdgcn_s_setprio(MVHP_RP_CHAIN);
                predict_mb_4x4(Wv, B, lane, m0, m1, m2, m3, A, Bv, C, D, res_luma, res);
                if (MVHP_ROWS_PRIO) __builtin_amdgcn_s_setprio(MVHP_RP_PRED);
            } else {
                if (MVHP_ROWS_PRIO) __builtin_amdgcn_s_setprio(MVHP_RP_CHAIN);
                const Edge8 g8 = edge8_of(lane);
                for (int blk = 0; blk < 4; blk++)
                    predict_8x8(Wv.T, Wv.E8, B, lane, g8, blk, (m0 >> (blk * 8)) & 255, A, Bv, C, D, res_luma, res);
                if (MVHP_ROWS_PRIO) __builtin_amdgcn_s_setprio(MVHP_RP_PRED);
            }
            // ---- chroma ----
            predict_chroma(Wv.TC, Wv.LcolC, lane, cmode, A, Bv, D, res_chroma, res);
            }

            // ---- write-out: the macroblock joins a 4-macroblock output strip in LDS; full strips go to HBM
            //      as 64-byte luma / 32-byte chroma row segments plus (fused) the RGB conversion ----
            const int mbi = mbx & 3;
            {
                const int y = lane >> 2, q = lane & 3;
                *reinterpret_cast<uint32_t *>(&Wv.SY[y * 64 + mbi * 16 + q * 4]) =
                    *reinterpret_cast<const uint32_t *>(&Wv.T[(y + 1) * 32 + 16 + q * 4]);
                if (lane < 32) {
                    const int pl = lane >> 4, cy = (lane & 15) >> 1, hf = lane & 1;
                    *reinterpret_cast<uint32_t *>(&Wv.SC[pl][cy * 32 + mbi * 8 + hf * 4]) =
                        *reinterpret_cast<const uint32_t *>(&Wv.TC[pl][(cy + 1) * 16 + 8 + hf * 4]);
                }
            }
            if (mbi == 3 || mbx == W - 1) {
                if (MVHP_ROWS_PRIO) __builtin_amdgcn_s_setprio(0);
                WAVE_SYNC();
                const int x0 = mbx - mbi, nb = (mbi + 1) * 16; // strip origin (MB units), width in samples
                {   // luma: lane -> 16 bytes of one row
                    const int y = lane >> 2, part = (lane & 3) * 16;
                    if (part < nb)
                        *reinterpret_cast<uint4 *>(&fy[(size_t)(row * 16 + y) * pitch + x0 * 16 + part]) =
                            *reinterpret_cast<const uint4 *>(&Wv.SY[y * 64 + part]);
                }
                {   // chroma: lane -> 8 bytes of one row of one plane
                    const int pl = lane >> 5, cy = (lane >> 2) & 7, part = (lane & 3) * 8;
                    if (part < (nb >> 1))
                        *reinterpret_cast<uint2 *>((pl ? fcr : fcb) + (size_t)(row * 8 + cy) * cpitch + x0 * 8 + part) =
                            *reinterpret_cast<const uint2 *>(&Wv.SC[pl][cy * 32 + part]);
                }
                if (frgb) {
                    // mb_to_rgb (export_utils.c:209-324) on the strip: 2x2 nearest chroma, integer formula :300-302
                    const int x4 = (lane & 15) * 4;
                    if (x4 < nb) {
#pragma unroll 2
                        for (int i = 0; i < 4; i++) {
                            const int y = i * 4 + (lane >> 4);
                            const uint32_t yw = *reinterpret_cast<const uint32_t *>(&Wv.SY[y * 64 + x4]);
                            const uint32_t cbw = *reinterpret_cast<const uint16_t *>(&Wv.SC[0][(y >> 1) * 32 + (x4 >> 1)]);
                            const uint32_t crw = *reinterpret_cast<const uint16_t *>(&Wv.SC[1][(y >> 1) * 32 + (x4 >> 1)]);
                            int d0, d1, d2;   // packed 16-bit arithmetic, see recon_batch_device.h rgb4()
                            rgb4(yw, bytes01(cbw), bytes01(crw), d0, d1, d2);
                            // one 12-byte store per lane: the 16 lanes of a row cover its 192 bytes in one instruction
                            typedef int v3i __attribute__((ext_vector_type(3)));
                            typedef v3i v3i_a4 __attribute__((aligned(4)));
                            *reinterpret_cast<v3i_a4 *>(frgb + ((size_t)(row * 16 + y) * pitch + x0 * 16 + x4) * 3) = v3i{d0, d1, d2};
                        }
                    }
                }
                WAVE_SYNC();
            }
            // ---- neighbour state for the next macroblock / next row ----
            // corners first (old top-right sample), then left columns, then the line buffer.
            if (MVHP_ROWS_PRIO) __builtin_amdgcn_s_setprio(MVHP_RP_TAIL);
            uint32_t keep = 0, bot = 0;
            if (keep_act) keep = *keep_src;
            if (bot_act) bot = *reinterpret_cast<const uint32_t *>(bot_src);
            WAVE_SYNC();
            if (keep_act) { *keep_dst = (uint8_t)keep; *keep_dst2 = (uint8_t)keep; }
            if (bot_act) *reinterpret_cast<uint32_t *>(bot_dst + mbx * bot_mul) = bot;
            if (WIDE && seam_out && bot_act)   // the same eight dwords, tagged, to the band below (one write-through store per granule)
                __hip_atomic_store(seam_wr + (size_t)mbx * SEAM_GRANULES + (lane - 48), seam_tag | bot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // ---- publish ----
            done++;
            // LDS operations of one wave complete in order; the explicit wait makes the line-buffer
            // writes land before the counter without waiting for the global plane stores (vmcnt).
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&B.progress[wave], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            WAVE_SYNC();
            } // macroblock of the pair
        }
    }
}

// ---------------------------------------------------------------------------
// colour conversion (export_utils.c:209-324): 2x2 nearest chroma replicate and
// the integer formula of :300-302.  One thread per 4 horizontal samples.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ycbcr_to_rgb_kernel(ColorArgs a)
{
    const int Wp = a.width_mbs * 16, Hp = a.height_mbs * 16;
    const int quads_per_row = Wp >> 2;
    const size_t quads_per_frame = (size_t)quads_per_row * Hp;
    const size_t total = quads_per_frame * a.n_frames;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int frame = (int)(t / quads_per_frame);
        const size_t r = t - (size_t)frame * quads_per_frame;
        const int y = (int)(r / quads_per_row), xq = (int)(r - (size_t)y * quads_per_row);
        const uint8_t *fy = a.yuv + (size_t)frame * Wp * Hp * 3 / 2;
        const uint8_t *fcb = fy + (size_t)Wp * Hp;
        const uint8_t *fcr = fcb + (size_t)(Wp >> 1) * (Hp >> 1);
        const uint32_t yw = *reinterpret_cast<const uint32_t *>(&fy[(size_t)y * Wp + xq * 4]);
        const uint32_t cbw = *reinterpret_cast<const uint16_t *>(&fcb[(size_t)(y >> 1) * (Wp >> 1) + xq * 2]);
        const uint32_t crw = *reinterpret_cast<const uint16_t *>(&fcr[(size_t)(y >> 1) * (Wp >> 1) + xq * 2]);
        uint8_t o[12];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int l = (yw >> (q * 8)) & 255;
            const int cb = (cbw >> ((q >> 1) * 8)) & 255, cr = (crw >> ((q >> 1) * 8)) & 255;
            const int ly = (298 * l) >> 8;
            o[q * 3 + 0] = (uint8_t)clip255(ly + ((408 * cr) >> 8) - 222);
            o[q * 3 + 1] = (uint8_t)clip255(ly - ((100 * cb) >> 8) - ((208 * cr) >> 8) + 135);
            o[q * 3 + 2] = (uint8_t)clip255(ly + ((516 * cb) >> 8) - 276);
        }
        typedef int v3i __attribute__((ext_vector_type(3)));
        typedef v3i v3i_a4 __attribute__((aligned(4)));
        *reinterpret_cast<v3i_a4 *>(a.rgb + (size_t)frame * Wp * Hp * 3 + ((size_t)y * Wp + xq * 4) * 3) =   // one 12-byte store
            v3i{(int)(o[0] | (o[1] << 8) | (o[2] << 16) | ((uint32_t)o[3] << 24)),
                (int)(o[4] | (o[5] << 8) | (o[6] << 16) | ((uint32_t)o[7] << 24)),
                (int)(o[8] | (o[9] << 8) | (o[10] << 16) | ((uint32_t)o[11] << 24))};
    }
}

// ---------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------
size_t recon_lds_bytes(int width_mbs, int nw)
{
    return sizeof(BlockLds) + (size_t)width_mbs * 32 + (size_t)nw * sizeof(WaveLds);
}

template <int NW, bool EXT, bool WIDE = false>
static hipError_t launch_rows_one(const ReconArgs &a, int n_groups, size_t lds, hipStream_t stream)
{
    hipError_t e = hipFuncSetAttribute((const void *)recon_rows_kernel<NW, EXT, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((recon_rows_kernel<NW, EXT, WIDE>), dim3(n_groups), dim3(NW * 64), lds, stream, a);
    return hipGetLastError();
}

// one workgroup per (picture, band of nw rows); a.wide_ticket / wide_base / wide_epoch / seam set by the caller
hipError_t launch_recon_wide(const ReconArgs &a, int n_frames, int nw, hipStream_t stream)
{
    if (nw != 4 || !a.wide_ticket || !a.wide_epoch) return hipErrorInvalidValue;
    const int bands = (a.height_mbs + nw - 1) / nw;
    if (bands > 1 && !a.seam) return hipErrorInvalidValue;
    const size_t lds = recon_lds_bytes(a.width_mbs, nw);
    const bool ext = a.slices || a.scaling;
    return ext ? launch_rows_one<4, true, true>(a, n_frames * bands, lds, stream) : launch_rows_one<4, false, true>(a, n_frames * bands, lds, stream);
}

hipError_t launch_recon(const ReconArgs &a, int n_frames, int nw, hipStream_t stream)
{
    const size_t lds = recon_lds_bytes(a.width_mbs, nw);
    const bool ext = a.slices || a.scaling;
    switch (nw) {
    case 4: return ext ? launch_rows_one<4, true>(a, n_frames, lds, stream) : launch_rows_one<4, false>(a, n_frames, lds, stream);
    case 8: return ext ? launch_rows_one<8, true>(a, n_frames, lds, stream) : launch_rows_one<8, false>(a, n_frames, lds, stream);
    case 16: return ext ? launch_rows_one<16, true>(a, n_frames, lds, stream) : launch_rows_one<16, false>(a, n_frames, lds, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_color(const ColorArgs &a, hipStream_t stream)
{
    const size_t total = (size_t)a.width_mbs * 4 * a.height_mbs * 16 * a.n_frames;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(ycbcr_to_rgb_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

} // namespace mvhp
