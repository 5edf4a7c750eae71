// recon_quad.hip -- the batch form of the reconstruction kernel (gfx950 only).
//
//   recon_quad_kernel<NW>   same contract as recon_rows_kernel (recon_kernels.hip): replaces
//                           intra_prediction_process() (decoder/h264/h264_intra_prediction.c:112-145),
//                           all of h264_transform.c, the planar gather of export.c:65-188 and
//                           mb_to_rgb() (export_utils.c:209-324) for whole pictures.
//
// Mapping: one workgroup reconstructs FOUR pictures in lockstep.  A wavefront is split into four
// quarters of 16 lanes, quarter q works on picture 4*blockIdx+q; wave w owns macroblock rows
// w, w+NW, ... of all four.  The four pictures sit at the same macroblock position at any time, so
// everything positional (neighbour availability, the row-above dependency wait, addresses inside a
// picture) is wave-uniform, while everything the stream decides (macroblock kind, prediction modes,
// QP, coefficients) is per-lane data.  The reason: the Intra4x4 chain is 16 dependent block steps of
// 16 samples each -- with one picture per wavefront 48 of the 64 lanes idle through it, and every
// instruction is paid per wavefront, not per lane.  Here each step serves four macroblocks.
//
// Inside a quarter, lane j owns luma 4x4 block j (luma4x4BlkIdx order) and, for j < 8, chroma block j
// (0-3 Cb, 4-7 Cr): its 16 levels arrive in registers straight from the packed record, are dequantised
// and inverse-transformed there, and for Intra16x16 / chroma the same lane predicts its 16 samples and
// writes them to the tile -- the residual never touches LDS.  Only Intra4x4 / Intra8x8 residuals are
// transposed through LDS (lane-per-block -> lane-per-sample).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "minivideo_hotpath.h"
#include "recon_kernels.h"
#include "recon_device.h"
#include "recon_batch_device.h"

namespace mvhp {

#ifndef MVHP_I8_UNROLL
#define MVHP_I8_UNROLL 4   // the four 8x8 blocks of an Intra8x8 macroblock as four copies (positions become constants: 2160p High
                           // 26.08 -> 24.82 ms, 1080p High 13.70 -> 13.54 ms); 1 = one loop body
#endif
#define MVHP_PRAGMA_(x) _Pragma(#x)
#define MVHP_UNROLL(n) MVHP_PRAGMA_(unroll n)

#if defined(MVHP_MARKS)   // measurement builds: section markers that survive into the ISA text (tools/isa_sections.py)
#define MVHP_MARK(name) asm volatile("; MARK " name ::: "memory")
#else
#define MVHP_MARK(name)
#endif
#ifndef MVHP_WIDE_NAP
#define MVHP_WIDE_NAP 1   // s_sleep units between polls of the row above in the banded instantiations
#endif
// wave priorities inside a step: as recon_oct.hip (the chains of LDS round trips first, residuals and colour conversion fill in)
#ifndef MVHP_PRIO_PRED
#define MVHP_PRIO_PRED 1
#endif
#ifndef MVHP_PRIO_TAIL
#define MVHP_PRIO_TAIL 2
#endif
#ifndef MVHP_CHAIN_PRIO
#define MVHP_CHAIN_PRIO 3   // wave priority inside the Intra4x4 chain
#endif
#ifndef MVHP_I8_PRIO
#define MVHP_I8_PRIO 3
#endif

// 128 VGPRs = four waves per SIMD: two 8-wave workgroups (or four 4-wave ones) per CU; LDS allows as many.
// The compiler gets v0-v99 (plus one register above everything for its SGPR spill lanes); v100-v123 are the
// record prefetch registers, named only inside inline assembly, so
// that nothing the compiler generates (copies, spills, reuse as temporaries) can touch a register a load is
// still writing.
// WIDE: as recon_rows_kernel<.., WIDE> (recon_kernels.hip): a workgroup reconstructs ONE BAND -- NW consecutive macroblock rows,
// one per wavefront, a single pass -- of its four pictures, the bands of a group run on different CUs, units (band, group) come
// from a ticket counter in band-major order, and across a band boundary the bottom samples travel as 8-byte tagged granules
// (one agent-scope store each, polled with agent-scope loads; no flag, no fence).  512 pictures = 128 groups fill the chip
// with 128 * bands workgroups where the one-workgroup-per-group form leaves half the CUs idle.
template <int NW, bool RGB, bool WIDE>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_num_vgpr(100))) void recon_quad_kernel(ReconArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int W = a.width_mbs, H = a.height_mbs;
    QTables &B = *reinterpret_cast<QTables *>(smem);
    uint8_t *lines = smem + sizeof(QTables);              // [quarter][ luma W*16 | Cb W*8 | Cr W*8 ]
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane_c = threadIdx.x & 63;
    uint8_t *wave_lds = lines + (size_t)4 * W * 32 + (size_t)wave * 4 * sizeof(QLds);
    if (WIDE && threadIdx.x == 0) B.pad[0] = (int)(atomicAdd(a.wide_ticket, 1u) - a.wide_base);

    // ---- one-time table setup ----
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
    for (int i = threadIdx.x; i < 2 * 9 * 16; i += NW * 64)
        B.tap4[i] = tap4_entry((i >> 4) % 9, i & 3, (i >> 2) & 3, i >= 9 * 16);
    for (int i = threadIdx.x; i < 9 * 64; i += NW * 64) B.tap8[i] = tap8_entry(i >> 6, i & 7, (i >> 3) & 7);
    if (threadIdx.x < 16) B.progress[threadIdx.x] = 0;
    if (threadIdx.x == 16) B.abort_flag = 0;
    __syncthreads();

    const int pitch = W * 16, cpitch = W * 8;
    const uint32_t plane_y = (uint32_t)W * H * 256, plane_c = (uint32_t)W * H * 64;
    const int up_wave = (wave + NW - 1) % NW;

    // this workgroup's pictures (and, WIDE, its band of rows)
    const int groups = (a.n_frames + 3) / 4;
    const int bands = (H + NW - 1) / NW;
    const int unit = WIDE ? __builtin_amdgcn_readfirstlane(B.pad[0]) : 0;
    const int band = WIDE ? unit / groups : 0;                       // band-major: see recon_rows_kernel
    const int grp = WIDE ? unit - band * groups : (int)blockIdx.x;
    if (WIDE && (unsigned)unit >= (unsigned)(bands * groups)) {   // a ticket outside the launch: the host's bookkeeping of the counter is off
        if (threadIdx.x == 0) atomicOr(a.err, 2u);
        return;
    }
    const int row_first = WIDE ? band * NW : 0;
    const int row_end = WIDE ? min(H, row_first + NW) : H;
    const bool seam_in = WIDE && wave == 0 && band > 0;             // top neighbours of this row come from the seam above
    const bool seam_out = WIDE && wave == NW - 1 && row_first + NW < H;   // this row's bottom samples feed the seam below

    // this lane's picture
    const int q_c = lane_c >> 4;
    const int frame_raw = grp * 4 + q_c;
    const bool valid = frame_raw < a.n_frames;            // a short last workgroup repeats the last picture, stores off
    const int frame = min(frame_raw, a.n_frames - 1);
    // Addresses: a scalar base per workgroup (its first picture) plus a 32-bit per-lane offset -- four pictures of
    // the largest supported size (1024 x 1024 macroblocks) span < 4 GiB in every buffer.
    const uint32_t qf = (uint32_t)(frame - grp * 4);
    const uint8_t *gpacked = a.packed + (size_t)grp * 4 * W * H * MVHP_MB_BYTES;
    uint8_t *gyuv = a.yuv + (size_t)grp * 4 * W * H * 384;
    uint8_t *grgb = a.rgb + (size_t)grp * 4 * W * H * 768;
    // seams of the group's first picture (scalar) + a 32-bit offset per lane: picture qf, granule (j & 7) of column (j >> 3)
    const unsigned long long *seam_rd = seam_in ? a.seam + ((size_t)grp * 4 * (bands - 1) + (band - 1)) * W * SEAM_GRANULES : nullptr;
    unsigned long long *seam_wr = seam_out ? a.seam + ((size_t)grp * 4 * (bands - 1) + band) * W * SEAM_GRANULES : nullptr;
    const uint32_t seam_pic = (uint32_t)((bands - 1) * W * SEAM_GRANULES);   // granules per picture
    // seam_in: the granule a lane has asked for, for the macroblock pair after the current one, lands in v[124:125] -- like
    // the record prefetch, registers that only inline assembly names (the compiler's own are full)
    const uint32_t qmb = qf * (uint32_t)(W * H);   // macroblocks in front of this lane's picture (< 2^22): one register;
                                                   // the three byte offsets are one 24-bit multiply away
#define OPACKED (__umul24(qmb_v, MVHP_MB_BYTES))
#define OYUV (__umul24(qmb_v, 384u))
#define ORGB (__umul24(qmb_v, 768u))

    // Packed records are prefetched one macroblock ahead into the registers of the lanes that consume them:
    // every lane of the quarter reads the 32-byte header (same address: one fetch), lane j the 32 bytes of
    // luma block j and, for j < 8, the 32 bytes of chroma block j (lanes 8-15 repeat their luma address).
    // The loads and the plane stores are inline assembly so that the number of vector-memory operations
    // between a prefetch and its use is fixed: a step issues either no store or exactly VM_STRIP stores after the twelve
    // loads (`n_st`), and the use is guarded by s_waitcnt vmcnt(n_st) -- the loads have landed, the stores of the step are still in flight.  (Left to the compiler, the wait became vmcnt(0) plus an immediate wait on the header.)
    // Every asm load / store is preceded by five wait states: its scalar base may have been reloaded from a spill lane
    // (v_readlane, a VALU write of an SGPR) by the instruction right in front of it, and a vector-memory read of such
    // an SGPR needs that distance.  Each asm store also carries two wait states behind it: a VALU write of the data registers of a >64-bit store right behind it
    // is a hardware hazard the compiler cannot see through inline assembly.
    constexpr int VM_STRIP = 8 + (RGB ? 12 : 0);   // a full strip per lane: 4 luma rows (16 B) + 2 chroma rows x 2 planes (8 B) (+ 4 rows x 3 RGB pieces)
    auto prefetch = [&](int prow, int px, int lane_p) {
        const int jj = lane_p & 15;
        uint32_t qmb_v = qmb;
        asm volatile("" : "+v"(qmb_v));   // recomputed per use: not worth three registers across the loop
        const uint32_t rec = OPACKED + (uint32_t)(prow * W + px) * MVHP_MB_BYTES;
        const uint32_t recL = rec + MVHP_MB_HEADER_BYTES + jj * 32;
        const uint32_t recC = rec + MVHP_MB_HEADER_BYTES + ((jj < 8) ? (16 + jj) : jj) * 32;
        asm volatile("s_nop 4\n\t"
                     "global_load_dwordx4 v[100:103], %0, %3\n\t"
                     "global_load_dwordx4 v[104:107], %0, %3 offset:16\n\t"
                     "global_load_dwordx4 v[108:111], %1, %3\n\t"
                     "global_load_dwordx4 v[112:115], %1, %3 offset:16\n\t"
                     "global_load_dwordx4 v[116:119], %2, %3\n\t"
                     "global_load_dwordx4 v[120:123], %2, %3 offset:16"
                     : : "v"(rec), "v"(recL), "v"(recC), "s"(gpacked)
                     : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
                       "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123");
    };
    if (row_first + wave < row_end) prefetch(row_first + wave, 0, lane_c);
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");   // (a wave without rows never reads the registers)

    const int up_adj = __builtin_amdgcn_readfirstlane((wave == 0) ? -1 : 0); // wave 0 follows the last wave's previous pass
    int done = 0; // macroblocks completed by this wave
    int n_st = 0;  // asm stores the previous step issued after its prefetch (0 also when the compiler counted them)
    // Output strip, owned by MACROBLOCK (as recon_oct.hip): lane (m, h) = (j & 3, j >> 2) of a quarter keeps luma rows
    // 4h .. 4h+3 of macroblock m of the 4-macroblock strip (two row pairs, each sharing a chroma row) and writes them when
    // the strip is complete: four adjacent lanes then cover 64 contiguous bytes of a luma row (32 of a chroma row) per store
    // instruction, and the 192 RGB bytes of a row leave in three consecutive instructions.  (HBM likes long runs, and L2
    // lines that are completed piecemeal leave early: with 16-byte runs the write traffic doubled.)  The chroma rows of the
    // three parked macroblocks wait in LDS (Q.SC).
    v4i L0 = {0, 0, 0, 0}, L1 = L0, L2 = L0, L3 = L0;
    for (int row = row_first + wave; row < row_end; row += NW) {
        const int pass = WIDE ? 0 : row / NW;
        // MBs the upper wave finished before its row (row-1); kept scalar explicitly
        const int up_base = (pass + up_adj) * W;
        const bool Bv = row > 0;
#pragma unroll 1
        for (int mbx = 0; mbx < W; mbx++) {
            // Re-materialise the lane id every macroblock: it keeps the compiler from hoisting hundreds of
            // lane-dependent LDS addresses out of this loop.
            int lane = lane_c;
            asm volatile("" : "+v"(lane));
            const int q = lane >> 4, j = lane & 15;
            QLds &Q = *reinterpret_cast<QLds *>(wave_lds + q * sizeof(QLds));
            uint8_t *line_y = lines + (size_t)q * W * 32;
            uint8_t *line_cb = line_y + W * 16;
            uint8_t *line_cr = line_cb + W * 8;
            const bool A = mbx > 0, C = Bv && (mbx < W - 1), D = A && Bv;

            // wait for the prefetched record and move it into compiler-visible registers (one asm block: nothing can
            // read v100-v123 before the wait); the next prefetch is issued behind the residual stage
            v2i w[12];
#define MVHP_WAIT_PREFETCH(N)                                                                                          \
            asm volatile("s_waitcnt vmcnt(%12)\n\t"                                                                    \
                         "v_mov_b64 %0, v[100:101]\n\tv_mov_b64 %1, v[102:103]\n\tv_mov_b64 %2, v[104:105]\n\t"         \
                         "v_mov_b64 %3, v[106:107]\n\tv_mov_b64 %4, v[108:109]\n\tv_mov_b64 %5, v[110:111]\n\t"         \
                         "v_mov_b64 %6, v[112:113]\n\tv_mov_b64 %7, v[114:115]\n\tv_mov_b64 %8, v[116:117]\n\t"         \
                         "v_mov_b64 %9, v[118:119]\n\tv_mov_b64 %10, v[120:121]\n\tv_mov_b64 %11, v[122:123]"            \
                         : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]),                \
                           "=&v"(w[6]), "=&v"(w[7]), "=&v"(w[8]), "=&v"(w[9]), "=&v"(w[10]), "=&v"(w[11])               \
                         : "n"(N)                                                                                      \
                         : "memory")
            if (WIDE && seam_out) {   // one more store behind the prefetch: the previous step's seam granules
                if (n_st) MVHP_WAIT_PREFETCH(VM_STRIP + 1);
                else MVHP_WAIT_PREFETCH(1);
            } else {
                if (n_st) MVHP_WAIT_PREFETCH(VM_STRIP);
                else MVHP_WAIT_PREFETCH(0);
            }
#undef MVHP_WAIT_PREFETCH
            if (WIDE && seam_in && (mbx & 1) == 0) {
                // Macroblocks mbx and mbx + 1 read columns <= mbx + 2 of the row above.  The first step of a row fetches columns
                // 0..3 now (two rounds); every later even step finds (mbx + 1, mbx + 2) asked for two steps ago, and asks for
                // (mbx + 3, mbx + 4).  Lane j of a quarter: column c0 + (j >> 3), granule j & 7 (0-3 luma dwords, 4-5 Cb, 6-7 Cr).
                // Every vector-memory instruction here is inline assembly: a load the compiler knows of would make it wait for
                // vmcnt(0) all over the step (it cannot see that the asm stores of the strip are not what it waits for).
                // (a lane id of its own: nothing computed here is shared with -- and kept alive for -- the rest of the step)
                int lane_s = lane_c;
                asm volatile("" : "+v"(lane_s));
                const int js = lane_s & 15, g = js & 7;
                uint8_t *sline_y = lines + (size_t)(lane_s >> 4) * W * 32;
                uint8_t *sline_cb = sline_y + W * 16, *sline_cr = sline_cb + W * 8;
                const uint32_t lo = (uint32_t)(min(grp * 4 + (lane_s >> 4), a.n_frames - 1) - grp * 4) * seam_pic + (uint32_t)g;
                auto seam_now = [&](uint32_t off) {   // blocking (first step of a row, and while a granule is not there yet)
                    v2i pv;
                    asm volatile("s_nop 4\n\tglobal_load_dwordx2 v[124:125], %1, %2 sc1\n\ts_waitcnt vmcnt(0)\n\tv_mov_b64 %0, v[124:125]"
                                 : "=v"(pv) : "v"(off), "s"(seam_rd) : "memory", "v124", "v125");
                    return pv;
                };
                int c0 = mbx ? mbx + 1 : 0;
                for (int round = mbx ? 1 : 0; round < 2; round++, c0 += 2) {
                    const int col = c0 + (js >> 3);
                    const bool act = col < W;
                    const uint32_t off = (lo + (uint32_t)((act ? col : 0) * SEAM_GRANULES)) * 8u;
                    v2i pv;
                    if (mbx == 0) pv = seam_now(off);
                    else asm volatile("v_mov_b64 %0, v[124:125]" : "=v"(pv) : : "memory");   // asked for two steps ago, in front of two
                                                                                           // record prefetches: the wait above saw it land
                    int spins = 0;
                    while (__builtin_amdgcn_ballot_w64(act && (uint32_t)pv.y != a.wide_epoch) != 0) {
                        __builtin_amdgcn_s_sleep(2);
                        // bounded; a failure anywhere in the launch (error word) ends every wait
                        bool stop = ++spins > (1 << 20) || __hip_atomic_load(&B.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (!stop && (spins & 255) == 0) {
                            uint32_t ew;
                            asm volatile("s_nop 4\n\tglobal_load_dword v124, %1, %2 sc1\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32 %0, v124"
                                         : "=v"(ew) : "v"(0u), "s"(a.err) : "memory", "v124");
                            stop = __builtin_amdgcn_ballot_w64(ew != 0) != 0;
                        }
                        if (stop) {
                            if (lane_s == 0) { __hip_atomic_store(&B.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicOr(a.err, 1u); }
                            return;
                        }
                        pv = seam_now(off);
                    }
                    if (act) {
                        uint8_t *dst = (g < 4) ? &sline_y[col * 16 + g * 4] : (g < 6) ? &sline_cb[col * 8 + (g - 4) * 4] : &sline_cr[col * 8 + (g - 6) * 4];
                        *reinterpret_cast<uint32_t *>(dst) = (uint32_t)pv.x;
                    }
                }
                {   // the request for the next pair (columns beyond the picture: the last column again, ignored later)
                    const uint32_t off = (lo + (uint32_t)(min(mbx + 3 + (js >> 3), W - 1) * SEAM_GRANULES)) * 8u;
                    asm volatile("s_nop 4\n\tglobal_load_dwordx2 v[124:125], %0, %1 sc1" : : "v"(off), "s"(seam_rd) : "memory", "v124", "v125");
                }
                WAVE_SYNC();
            }
            const int4 cH0 = make_int4(w[0].x, w[0].y, w[1].x, w[1].y), cH1 = make_int4(w[2].x, w[2].y, w[3].x, w[3].y);
            const int4 cLA = make_int4(w[4].x, w[4].y, w[5].x, w[5].y), cLB = make_int4(w[6].x, w[6].y, w[7].x, w[7].y);
            const int4 cCA = make_int4(w[8].x, w[8].y, w[9].x, w[9].y), cCB = make_int4(w[10].x, w[10].y, w[11].x, w[11].y);
            const uint32_t h0 = (uint32_t)cH0.x, h1 = (uint32_t)cH0.y, nz = (uint32_t)cH0.z;
            const uint32_t m0 = (uint32_t)cH0.w, m1 = (uint32_t)cH1.x, m2 = (uint32_t)cH1.y, m3 = (uint32_t)cH1.z;
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

            // luma block geometry (luma4x4BlkIdx j)
            const int xO = (((j >> 2) & 1) << 3) | ((j & 1) << 2);
            const int yO = ((j >> 3) << 3) | (((j >> 1) & 1) << 2);

            // =====================================================================================
            // residuals (no neighbour dependency: done before waiting for the row above)
            // =====================================================================================
            MVHP_MARK("resid_luma");
            if (MVHP_PRIO_TAIL) __builtin_amdgcn_s_setprio(0);
            int r2[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // luma block j: packed int16 pairs, row-major
            int c2[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // chroma block j (j < 8)
            if (any_l) {
                const int4 qt = B.q4[qpy];
                const int shr = qt.w & 255, rnd = (qt.w >> 8) & 255, s = (qt.w >> 16) & 255, m = (qt.w >> 24) & 255;
                if (kind == MVHP_KIND_I8x8) {
                    MVHP_MARK("r_8x8");
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
                            int32_t *dst = &Q.scr[((j >> 2) & 1) * 64 + r0 * 8];
                            *reinterpret_cast<int4 *>(dst) = make_int4(d0[0], d0[1], d0[2], d0[3]);
                            *reinterpret_cast<int4 *>(dst + 4) = make_int4(d0[4], d0[5], d0[6], d0[7]);
                            *reinterpret_cast<int4 *>(dst + 8) = make_int4(d1[0], d1[1], d1[2], d1[3]);
                            *reinterpret_cast<int4 *>(dst + 12) = make_int4(d1[4], d1[5], d1[6], d1[7]);
                        }
                        WAVE_SYNC();
#pragma unroll
                        for (int i = 0; i < 8; i++) col[h][i] = Q.scr[(j >> 3) * 64 + i * 8 + (j & 7)];
                        idct8_1d(col[h]);
                        WAVE_SYNC();
                    }
                    // res[blk8][column][row]: lane j column j & 7 of block 2h + (j >> 3)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        int4 o;
                        o.x = pack_res(col[h][0] >> 6, col[h][1] >> 6);
                        o.y = pack_res(col[h][2] >> 6, col[h][3] >> 6);
                        o.z = pack_res(col[h][4] >> 6, col[h][5] >> 6);
                        o.w = pack_res(col[h][6] >> 6, col[h][7] >> 6);
                        *reinterpret_cast<int4 *>(&Q.res[(2 * h + (j >> 3)) * 64 + (j & 7) * 8]) = o;
                    }
                } else {
                    MVHP_MARK("r_4x4");
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
                    // shr = rnd = 0 from qP 24 up (checked for the whole wave).  The levels are multiplied straight out
                    // of the packed words (mad_level: v_mad_i32_i16 selects the 16-bit half itself).
                    if (__builtin_amdgcn_ballot_w64(shr != 0) == 0) {
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
                    if (kind == MVHP_KIND_I16x16) d[0] = dc;
                    d[0] += 32;
                    idct4x4_packed(d, r2);
                    if (!need_l) {
#pragma unroll
                        for (int i = 0; i < 8; i++) r2[i] = 0;
                    }
                }
            }
            if (kind == MVHP_KIND_I4x4) {   // lane-per-block -> lane-per-sample goes through LDS (zeros when there is no residual)
                *reinterpret_cast<int4 *>(&Q.res[j * 16]) = make_int4(r2[0], r2[1], r2[2], r2[3]);
                *reinterpret_cast<int4 *>(&Q.res[j * 16 + 8]) = make_int4(r2[4], r2[5], r2[6], r2[7]);
            }
            MVHP_MARK("resid_chroma");
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
                idct4x4_packed(d, c2);
                if (!need_c) {
#pragma unroll
                    for (int i = 0; i < 8; i++) c2[i] = 0;
                }
            }

            {   // the record is consumed: prefetch the next macroblock of this wave -- same row, or the first of its
                // next row (none left: this one again)
                int nrow = row, nx = mbx + 1;
                if (nx >= W) { nrow = row + NW; nx = 0; }
                if (nrow >= row_end) { nrow = row; nx = mbx; }
                prefetch(nrow, nx, lane);
            }

            // =====================================================================================
            // wait for the row above: needs columns <= min(mbx+1, W-1); then fetch the top neighbours
            // =====================================================================================
            MVHP_MARK("wait_up");
            if (MVHP_PRIO_PRED) __builtin_amdgcn_s_setprio(MVHP_PRIO_PRED);
#if defined(MVHP_ABL_NO_WAIT)
            if (false) {
#else
            if (Bv) {
#endif
                const int need = (WIDE && seam_in) ? 0 : up_base + min(mbx + 2, W);   // (seam_in: the columns are in the line buffer)
                int spins = 0;
                while (__hip_atomic_load(&B.progress[up_wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                    // (a poll costs vector-ALU issue slots -- the counter lands in a VGPR --; what a longer nap adds to a row's
                    //  phase behind the row above it adds once per row, not per macroblock)
                    __builtin_amdgcn_s_sleep(WIDE ? MVHP_WIDE_NAP : 1);
                    if (++spins > (1 << 22) || __hip_atomic_load(&B.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        if (lane == 0) { __hip_atomic_store(&B.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicOr(a.err, 1u); }
                        return;
                    }
                }
                asm volatile("" ::: "memory");
                // lanes 0-3 luma top, 4-5 luma up-right (when C), 6-7 Cb top, 8-9 Cr top: one dword each
                if (j < 10 && (C || (j >> 1) != 2)) {
                    uint8_t *dst;
                    const uint8_t *src;
                    if (j < 4) { dst = &Q.T[16 + j * 4]; src = &line_y[mbx * 16 + j * 4]; }
                    else if (j < 6) { dst = &Q.T[32 + (j - 4) * 4]; src = &line_y[mbx * 16 + 16 + (j - 4) * 4]; }
                    else if (j < 8) { dst = &Q.TC[0][8 + (j - 6) * 4]; src = &line_cb[mbx * 8 + (j - 6) * 4]; }
                    else { dst = &Q.TC[1][8 + (j - 8) * 4]; src = &line_cr[mbx * 8 + (j - 8) * 4]; }
                    *reinterpret_cast<uint32_t *>(dst) = *reinterpret_cast<const uint32_t *>(src);
                }
            }
            WAVE_SYNC();

            // =====================================================================================
            // chroma prediction (h264_intra_prediction.c:2157-2564 + transform4x4_chroma): lane j < 8 predicts
            // its own 4x4 block (plane j >> 2, block j & 3)
            // =====================================================================================
            MVHP_MARK("pred_chroma");
            if (j < 8) {
                const int pl = j >> 2, k = j & 3;
                const int cx = (k & 1) * 4, cy = (k >> 1) * 4;
                uint8_t *TCp = Q.TC[pl];
                uint32_t pw[4] = {0u, 0u, 0u, 0u};
                const uint32_t topw = *reinterpret_cast<const uint32_t *>(&TCp[8 + cx]);
                const uint32_t lefw = *reinterpret_cast<const uint32_t *>(&Q.LcolC[pl][cy]);
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
                        const uint2 lefv = *reinterpret_cast<const uint2 *>(Q.LcolC[pl]);
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

            // =====================================================================================
            // luma prediction
            // =====================================================================================
            MVHP_MARK("p_i16");
            if (kind == MVHP_KIND_I16x16) {
                // h264_intra_prediction.c:1809-2141 + transform16x16_luma; lane j predicts its own 4x4 block
                uint32_t pw[4] = {0u, 0u, 0u, 0u};
                if (i16mode == 0) {
                    if (Bv) { const uint32_t t = *reinterpret_cast<const uint32_t *>(&Q.T[16 + xO]); pw[0] = pw[1] = pw[2] = pw[3] = t; }
                } else if (i16mode == 1) {
                    if (A) {
                        const uint32_t l = *reinterpret_cast<const uint32_t *>(&Q.Lcol[yO]);
#pragma unroll
                        for (int y = 0; y < 4; y++) pw[y] = ((l >> (8 * y)) & 255u) * 0x01010101u;
                    }
                } else if (i16mode == 2) {
                    const uint4 topv = *reinterpret_cast<const uint4 *>(&Q.T[16]);
                    const uint4 lefv = *reinterpret_cast<const uint4 *>(Q.Lcol);
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
                        const uint4 topv = *reinterpret_cast<const uint4 *>(&Q.T[16]);
                        const uint4 lefv = *reinterpret_cast<const uint4 *>(Q.Lcol);
                        const int cor = Q.T[15];
                        const int Hh = plane_grad16(topv, (uint32_t)cor), Vv = plane_grad16(lefv, (uint32_t)cor);
                        const int aa = 16 * ((int)(lefv.w >> 24) + (int)(topv.w >> 24));
                        const int bb = (5 * Hh + 32) >> 6;
                        const int cc = (5 * Vv + 32) >> 6;
                        const int v00 = aa + bb * (xO - 7) + cc * (yO - 7) + 16;
#pragma unroll
                        for (int y = 0; y < 4; y++) pw[y] = plane_row(v00 + cc * y, bb);
                    }
                }
                emit_block(&Q.T[(yO + 1) * 32 + 16 + xO], 32, pw, r2);
            } else if (kind == MVHP_KIND_I4x4) {
                MVHP_MARK("p_i4");
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
                                            (Bv ? ((1u << 0) | (1u << 1) | (1u << 4)) : 0u) | (C ? (1u << 5) : 0u);
                // neighbours each mode needs, 3 bits per mode: bit0 left, bit1 up, bit2 up-left (mode 2 = DC apart)
                constexpr uint32_t REQ = (2u << 0) | (1u << 3) | (0u << 6) | (2u << 9) | (7u << 12) | (7u << 15) | (7u << 18) |
                                         (2u << 21) | (1u << 24);
                // control word of block j, computed by lane j and broadcast inside the quarter at step j:
                // (control word, below: bit 31 the mode is DC, bit 16 prediction allowed, bits 0-15 tap table row offset)
                uint32_t info;
                {
                    const uint32_t mw = (j < 4) ? m0 : (j < 8) ? m1 : (j < 12) ? m2 : m3;
                    const uint32_t mode = (mw >> ((j & 3) * 8)) & 255u;
                    const uint32_t avail = ((av_left >> j) & 1u) | (((av_up >> j) & 1u) << 1) | (((av_upleft >> j) & 1u) << 2);
                    const uint32_t req = (REQ >> (min(mode, 8u) * 3)) & 7u;
                    const uint32_t ok = (((req & ~avail) == 0u) && (mode < 9u)) ? 1u : 0u; // else the prediction stays 0 (:442)
                    const uint32_t trow = (((av_upright >> j) & 1u) ? 0u : 9u) + min(mode, 8u);
                    info = ((mode == 2u) ? 0x80000000u : 0u) | (ok << 16) | (trow * 64u);   // bit 31 DC, bit 16 allowed, bits 0-15 table row offset
                }
                const int pix = (j >> 2) * 32 + (j & 3);   // this lane's sample inside a block, tile units
                const uint8_t *T = Q.T;
                const uint8_t *tapb = reinterpret_cast<const uint8_t *>(B.tap4) + j * 4;
                // software pipeline: control word, table entry and residual of block b+1 are fetched before block
                // b's dependent tile reads (the residual array holds zeros when the macroblock has none)
                const int qbase4 = (lane & 48) << 2;
                __builtin_amdgcn_s_setprio(MVHP_CHAIN_PRIO);   // the dependent chain issues few, latency-critical instructions
                uint32_t inf = quarter_bcast(info, qbase4, 0);
                uint32_t e_nx = *reinterpret_cast<const uint32_t *>(tapb + (inf & 0xffffu));
                int r_nx = (int)Q.res[j];
#pragma unroll
                for (int blk = 0; blk < 16; blk++) {
                    const int bxO = (((blk >> 2) & 1) << 3) | ((blk & 1) << 2);
                    const int byO = ((blk >> 3) << 3) | (((blk >> 1) & 1) << 2);
                    const int base = (byO + 1) * 32 + 16 + bxO;     // tile index of the block's top-left sample
                    const uint32_t cur = inf;
                    const uint32_t e = e_nx;
                    const int r = r_nx;
                    if (blk < 15) {
                        inf = quarter_bcast(info, qbase4, blk + 1);
                        e_nx = *reinterpret_cast<const uint32_t *>(tapb + (inf & 0xffffu));
                        r_nx = (int)Q.res[(blk + 1) * 16 + j];
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
                    Q.T[base + pix] = (uint8_t)clip255(pred + r);
                    WAVE_SYNC();
                }
                __builtin_amdgcn_s_setprio(MVHP_PRIO_PRED);
            } else {
                // Intra 8x8: h264_intra_prediction.c:1107-1353 (edge filter) + :1366-1793 + transform8x8_luma;
                // lane j predicts samples (4*(j&1) .. +3, j>>1) of the block
                MVHP_MARK("p_i8");
                if (MVHP_I8_PRIO) __builtin_amdgcn_s_setprio(MVHP_I8_PRIO);
                // The lane's two entries of the unified edge (recon_device.h mode_entry: 0-1 left[7] replicated, 2-9 left[7..0],
                // 10 corner, 11-26 top[0..15], 27 replicated) and where their three taps lie RELATIVE to the block's top row in the
                // tile: that does not depend on the block, so it is worked out once per macroblock; a block only picks between the
                // neighbour and the sample itself where a side is missing (scalar conditions) and clamps the taps beyond top[7]
                // when there is no up-right block (:1230-1236).
                int e_of[2], o_e[2], o_lo[2], o_hi[2];
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int e = min(max(j + 16 * half, 2), 26);
                    const int lo = max(e - 1, 2), hi = min(e + 1, 26);
                    e_of[half] = e;
                    o_e[half] = (e >= 10) ? e - 11 : 31 + (9 - e) * 32;
                    o_lo[half] = (lo >= 10) ? lo - 11 : 31 + (9 - lo) * 32;
                    o_hi[half] = (hi >= 10) ? hi - 11 : 31 + (9 - hi) * 32;
                }
                MVHP_UNROLL(MVHP_I8_UNROLL)
                for (int blk = 0; blk < 4; blk++) {
                    const int bxO = (blk & 1) * 8, byO = (blk >> 1) * 8;
                    const int mode = (int)((m0 >> (blk * 8)) & 255u);
                    const bool left = (bxO > 0) || A;
                    const bool up = (byO > 0) || Bv;
                    const bool upleft = (bxO > 0) ? ((byO > 0) || Bv) : ((byO > 0) ? A : D);
                    const bool upright = (blk == 0) ? Bv : (blk == 1) ? C : (blk == 2);
                    const uint8_t *Trow = &Q.T[byO * 32 + 16 + bxO];
#pragma unroll
                    for (int half = 0; half < 2; half++) {
                        const int e = e_of[half];
                        int a_lo = (((e == 11) && !upleft) || ((e == 10) && !left)) ? o_e[half] : o_lo[half];
                        int a_hi = (((e == 9) && !upleft) || ((e == 10) && !up)) ? o_e[half] : o_hi[half];
                        int a_e = o_e[half];
                        if (!upright) {
                            a_lo = (e > 19) ? 7 : a_lo;
                            a_e = (e > 18) ? 7 : a_e;
                            a_hi = (e > 17) ? 7 : a_hi;
                        }
                        const int v0 = Trow[a_lo], v1 = Trow[a_e], v2 = Trow[a_hi];
                        if (half == 0 || j < 12) Q.E8[j + 16 * half] = (uint8_t)((v0 + 2 * v1 + v2 + 2) >> 2);
                    }
                    WAVE_SYNC();
                    {
                        const int y = j >> 1, x0 = (j & 1) * 4;
                        uint32_t pwv = 0;
                        {   // one path for the four pictures (round 4): taps for every lane, DC only when some picture wants it
                            const uint32_t mm = min((uint32_t)mode, 8u);
                            constexpr uint32_t REQ8 = (2u << 0) | (1u << 3) | (0u << 6) | (2u << 9) | (7u << 12) | (7u << 15) | (7u << 18) |
                                                      (2u << 21) | (1u << 24);
                            const uint32_t avail = (left ? 1u : 0u) | (up ? 2u : 0u) | (upleft ? 4u : 0u);
                            const bool ok = (((REQ8 >> (mm * 3u)) & 7u & ~avail) == 0u) && ((uint32_t)mode < 9u);
                            const uint4 e4 = *reinterpret_cast<const uint4 *>(&B.tap8[mm * 64 + y * 8 + x0]);
                            const uint32_t ee[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                            for (int x = 0; x < 4; x++) {
                                const int v0 = Q.E8[ee[x] & 255], v1 = Q.E8[(ee[x] >> 8) & 255], v2 = Q.E8[ee[x] >> 16];
                                pwv |= (uint32_t)((v0 + 2 * v1 + v2 + 2) >> 2) << (8 * x);
                            }
                            pwv &= ok ? 0xffffffffu : 0u;
                            if (__builtin_amdgcn_ballot_w64(mode == 2) != 0) {
                                const uint32_t *E = reinterpret_cast<const uint32_t *>(Q.E8);
                                const uint32_t w0 = E[0], w1 = E[1], w2 = E[2], w3 = E[3], w4 = E[4];
                                const int sumV = sum4(w0 & 0xffff0000u) + sum4(w1) + sum4(w2 & 0x0000ffffu);       // E8[2..9]
                                const int sumH = sum4(w2 & 0xff000000u) + sum4(w3) + sum4(w4 & 0x00ffffffu);       // E8[11..18]
                                int v;
                                if (left && up) v = (sumH + sumV + 8) >> 4;
                                else if (left) v = (sumV + 4) >> 3;
                                else if (up) v = (sumH + 4) >> 3;
                                else v = 128;
                                pwv = (mode == 2) ? (uint32_t)v * 0x01010101u : pwv;
                            }
                        }
                        int rr[4] = {0, 0, 0, 0};
                        if (need_l) {
#pragma unroll
                            for (int x = 0; x < 4; x++) rr[x] = (int)Q.res[blk * 64 + (x0 + x) * 8 + y];
                        }
                        uint32_t out = 0;
#pragma unroll
                        for (int x = 0; x < 4; x++) out |= (uint32_t)clip255((int)((pwv >> (8 * x)) & 255u) + rr[x]) << (8 * x);
                        *reinterpret_cast<uint32_t *>(&Q.T[(byO + y + 1) * 32 + 16 + bxO + x0]) = out;
                    }
                    WAVE_SYNC();
                }
                if (MVHP_I8_PRIO) __builtin_amdgcn_s_setprio(MVHP_PRIO_PRED);
            }

            MVHP_MARK("pred_end");
            WAVE_SYNC();
            // I_PCM (8.3.5; only MVHP_STREAM_SPEC streams carry it): the samples as they are, over whatever
            // the prediction paths above made of such a record.  Record layout (minivideo_hotpath.h): the owner of luma block
            // 2i holds luma rows 2i and 2i+1, the owner of block 2i+1 Cb row i and Cr row i.
            if (__builtin_amdgcn_ballot_w64(kind == MVHP_KIND_IPCM) != 0) {
                if (kind == MVHP_KIND_IPCM) {
                    const int jp = j >> 1;
                    if ((j & 1) == 0) {
                        *reinterpret_cast<int4 *>(&Q.T[(2 * jp + 1) * 32 + 16]) = cLA;
                        *reinterpret_cast<int4 *>(&Q.T[(2 * jp + 2) * 32 + 16]) = cLB;
                    } else {
                        *reinterpret_cast<int2 *>(&Q.TC[0][(jp + 1) * 16 + 8]) = make_int2(cLA.x, cLA.y);
                        *reinterpret_cast<int2 *>(&Q.TC[1][(jp + 1) * 16 + 8]) = make_int2(cLA.z, cLA.w);
                    }
                }
                WAVE_SYNC();
            }

            // =====================================================================================
            // write-out (mb_to_rgb, export_utils.c:209-324, fused): park, or flush the 4-macroblock strip
            // =====================================================================================
            MVHP_MARK("writeout");
            {
                const int mbi = mbx & 3;
                const int m_own = j & 3, h_own = j >> 2;
                n_st = 0;
                if (m_own == mbi) {   // this macroblock's owners take its luma rows out of the tile
                    const uint8_t *t0 = &Q.T[(4 * h_own + 1) * 32 + 16];
                    L0 = *reinterpret_cast<const v4i *>(t0);            L1 = *reinterpret_cast<const v4i *>(t0 + 32);
                    L2 = *reinterpret_cast<const v4i *>(t0 + 2 * 32);   L3 = *reinterpret_cast<const v4i *>(t0 + 3 * 32);
                }
                if (mbi == 3 || mbx == W - 1) {
                    if (MVHP_PRIO_PRED) __builtin_amdgcn_s_setprio(0);   // the flush (colour conversion) is throughput work; taking / parking is not
                    uint32_t qmb_v = qmb;
                    asm volatile("" : "+v"(qmb_v));
                    // chroma rows 2h, 2h + 1 of the lane's macroblock: parked ones from the strip, the current one from the tile
                    const bool cur = (m_own == mbi);
                    const uint8_t *cb_src = cur ? &Q.TC[0][(2 * h_own + 1) * 16 + 8] : &Q.SC[0][2 * h_own * 24 + m_own * 8];
                    const uint8_t *cr_src = cur ? &Q.TC[1][(2 * h_own + 1) * 16 + 8] : &Q.SC[1][2 * h_own * 24 + m_own * 8];
                    const int cstep = cur ? 16 : 24;
                    const uint2 cb0 = *reinterpret_cast<const uint2 *>(cb_src), cb1 = *reinterpret_cast<const uint2 *>(cb_src + cstep);
                    const uint2 cr0 = *reinterpret_cast<const uint2 *>(cr_src), cr1 = *reinterpret_cast<const uint2 *>(cr_src + cstep);
                    const uint32_t x0 = (uint32_t)((mbx & ~3) * 16 + m_own * 16);
                    const uint32_t lrow = (uint32_t)((row * 16 + 4 * h_own) * pitch) + x0;     // luma row 4h of the macroblock row
                    const uint32_t pl = OYUV + lrow;
                    const uint32_t pcb = OYUV + plane_y + (uint32_t)((row * 8 + 2 * h_own) * cpitch) + (x0 >> 1), pcr = pcb + plane_c;
                    if (mbi == 3) {
                        // ---- full strip: exactly VM_STRIP store instructions ----
#define MVHP_ST(ADDR, DATA, BASE, OFF) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:" #OFF "\n\ts_nop 1" : : "v"(ADDR), "v"(DATA), "s"(BASE) : "memory")
#define MVHP_ST2(ADDR, DATA, BASE) asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2" : : "v"(ADDR), "v"(DATA), "s"(BASE) : "memory")
                        if (valid) {
                            MVHP_ST(pl, L0, gyuv, 0);                MVHP_ST(pl + pitch, L1, gyuv, 0);
                            MVHP_ST(pl + 2 * pitch, L2, gyuv, 0);    MVHP_ST(pl + 3 * pitch, L3, gyuv, 0);
                            const v2i b0 = {(int)cb0.x, (int)cb0.y}, b1 = {(int)cb1.x, (int)cb1.y};
                            const v2i q0 = {(int)cr0.x, (int)cr0.y}, q1 = {(int)cr1.x, (int)cr1.y};
                            MVHP_ST2(pcb, b0, gyuv);            MVHP_ST2(pcr, q0, gyuv);
                            MVHP_ST2(pcb + cpitch, b1, gyuv);   MVHP_ST2(pcr + cpitch, q1, gyuv);
                        }
                        if (RGB) {
                            const uint32_t prgb = ORGB + lrow * 3u;
                            // one luma row of the lane's macroblock against its chroma row (rows 2c, 2c + 1 share row c,
                            // export_utils.c:278-279); the colour terms are recomputed per row here -- the row-pair form
                            // of recon_oct.hip needs more registers than this kernel's 100 leave
#define MVHP_RGB_OUT(YQ, CB, CR, R)                                                                                    \
                            {                                                                                          \
                                v4i a0, a1, a2;                                                                        \
                                rgb16(make_uint4((uint32_t)(YQ).x, (uint32_t)(YQ).y, (uint32_t)(YQ).z, (uint32_t)(YQ).w), CB, CR, a0, a1, a2); \
                                const uint32_t pa = prgb + (R) * 3u * (uint32_t)pitch;                                  \
                                if (valid) {                                                                           \
                                    MVHP_ST(pa, a0, grgb, 0); MVHP_ST(pa, a1, grgb, 16); MVHP_ST(pa, a2, grgb, 32);     \
                                } else {                                                                               \
                                    asm volatile("" : : "v"(a0), "v"(a1), "v"(a2));                                    \
                                }                                                                                      \
                            }
                            MVHP_RGB_OUT(L0, cb0, cr0, 0u)
                            MVHP_RGB_OUT(L1, cb0, cr0, 1u)
                            MVHP_RGB_OUT(L2, cb1, cr1, 2u)
                            MVHP_RGB_OUT(L3, cb1, cr1, 3u)
#undef MVHP_RGB_OUT
                        }
#undef MVHP_ST
#undef MVHP_ST2
                        n_st = VM_STRIP;
                    } else if (m_own <= mbi && valid) {
                        // ---- short strip at the right picture edge (W % 4 != 0): compiler-counted stores ----
                        *reinterpret_cast<v4i *>(gyuv + pl) = L0;
                        *reinterpret_cast<v4i *>(gyuv + pl + pitch) = L1;
                        *reinterpret_cast<v4i *>(gyuv + pl + 2 * pitch) = L2;
                        *reinterpret_cast<v4i *>(gyuv + pl + 3 * pitch) = L3;
                        *reinterpret_cast<uint2 *>(gyuv + pcb) = cb0;
                        *reinterpret_cast<uint2 *>(gyuv + pcr) = cr0;
                        *reinterpret_cast<uint2 *>(gyuv + pcb + cpitch) = cb1;
                        *reinterpret_cast<uint2 *>(gyuv + pcr + cpitch) = cr1;
                        if (RGB) {
                            // one row per trip of a loop that is NOT unrolled: with the four rows' colour arithmetic scheduled together
                            // the compiler needs more registers than v0-v99 and reaches into the prefetch registers
                            // (check_prefetch_hazard.py); the row is chosen by scalar selects
                            const uint32_t prgb = ORGB + lrow * 3u;
#pragma unroll 1
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
                } else {
                    // ---- park the chroma rows (lane j: row j & 7 of plane j >> 3) in the LDS strip ----
                    const uint2 cv = *reinterpret_cast<const uint2 *>(&Q.TC[j >> 3][((j & 7) + 1) * 16 + 8]);
                    *reinterpret_cast<uint2 *>(&Q.SC[j >> 3][(j & 7) * 24 + mbi * 8]) = cv;
                }
            }

            // =====================================================================================
            // neighbour state for the next macroblock / next row, then publish
            // =====================================================================================
            MVHP_MARK("neighbours");
            if (MVHP_PRIO_TAIL) __builtin_amdgcn_s_setprio(MVHP_PRIO_TAIL);
            {
                // corners (old top-right sample) by lanes 0-2, left columns: lane j luma row j; lane j chroma
                // row j & 7 of plane j >> 3; bottom rows -> line buffer by lanes 0-7 (one dword each)
                const uint8_t kl = Q.T[(j + 1) * 32 + 31];
                const uint8_t kc = Q.TC[j >> 3][((j & 7) + 1) * 16 + 15];
                uint8_t kk = 0;
                uint8_t *kdst = &Q.T[15];
                if (j == 0) kk = Q.T[31];
                else if (j == 1) { kk = Q.TC[0][15]; kdst = &Q.TC[0][7]; }
                else if (j == 2) { kk = Q.TC[1][15]; kdst = &Q.TC[1][7]; }
                uint32_t bot = 0;
                uint8_t *bdst = line_y;
                if (j < 4) { bot = *reinterpret_cast<const uint32_t *>(&Q.T[16 * 32 + 16 + j * 4]); bdst = &line_y[mbx * 16 + j * 4]; }
                else if (j < 6) { bot = *reinterpret_cast<const uint32_t *>(&Q.TC[0][8 * 16 + 8 + (j - 4) * 4]); bdst = &line_cb[mbx * 8 + (j - 4) * 4]; }
                else if (j < 8) { bot = *reinterpret_cast<const uint32_t *>(&Q.TC[1][8 * 16 + 8 + (j - 6) * 4]); bdst = &line_cr[mbx * 8 + (j - 6) * 4]; }
                WAVE_SYNC();
                Q.T[(j + 1) * 32 + 15] = kl;
                Q.Lcol[j] = kl;
                Q.TC[j >> 3][((j & 7) + 1) * 16 + 7] = kc;
                Q.LcolC[j >> 3][j & 7] = kc;
                if (j < 3) *kdst = kk;
                if (j < 8) *reinterpret_cast<uint32_t *>(bdst) = bot;
                if (WIDE && seam_out) {
                    // the same eight dwords, tagged, to the band below: one write-through store per granule -- inline assembly, see
                    // the seam_in block; the address from a lane id made here (computed ahead, it would sit in registers through
                    // the write-out)
                    int lane_w = lane_c;
                    asm volatile("" : "+v"(lane_w));
                    const uint32_t pic = (uint32_t)(min(grp * 4 + (lane_w >> 4), a.n_frames - 1) - grp * 4);
                    const uint32_t off = (pic * seam_pic + (uint32_t)(mbx * SEAM_GRANULES + (lane_w & 7))) * 8u;
                    const v2i gv = {(int)bot, (int)a.wide_epoch};
                    if ((lane_w & 15) < 8)
                        asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2 sc1" : : "v"(off), "v"(gv), "s"(seam_wr) : "memory");
                }
            }
            done++;
            // LDS operations of one wave complete in order; the explicit wait makes the line-buffer
            // writes land before the counter without waiting for the global plane stores (vmcnt).
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&B.progress[wave], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            WAVE_SYNC();
        }
    }
}

#undef OPACKED
#undef OYUV
#undef ORGB

size_t recon_quad_lds_bytes(int width_mbs, int nw)
{
    return sizeof(QTables) + (size_t)4 * width_mbs * 32 + (size_t)nw * 4 * sizeof(QLds);
}

template <int NW, bool RGB, bool WIDE = false>
static hipError_t launch_quad_one(const ReconArgs &a, hipStream_t stream)
{
    const size_t lds = recon_quad_lds_bytes(a.width_mbs, NW);
    const int groups = (a.n_frames + 3) / 4;
    const int bands = WIDE ? (a.height_mbs + NW - 1) / NW : 1;
    hipError_t e = hipFuncSetAttribute((const void *)recon_quad_kernel<NW, RGB, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((recon_quad_kernel<NW, RGB, WIDE>), dim3(groups * bands), dim3(NW * 64), lds, stream, a);
    return hipGetLastError();
}

// one workgroup per (band of nw rows, group of four pictures); a.wide_ticket / wide_base / wide_epoch / seam set by the caller
hipError_t launch_recon_quad_wide(const ReconArgs &a, int nw, hipStream_t stream)
{
    if (!a.wide_ticket || !a.wide_epoch) return hipErrorInvalidValue;
    if ((a.height_mbs + nw - 1) / nw > 1 && !a.seam) return hipErrorInvalidValue;
    const bool rgb = a.rgb != nullptr;
    switch (nw) {
    case 4: return rgb ? launch_quad_one<4, true, true>(a, stream) : launch_quad_one<4, false, true>(a, stream);
    case 8: return rgb ? launch_quad_one<8, true, true>(a, stream) : launch_quad_one<8, false, true>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_recon_quad(const ReconArgs &a, int nw, hipStream_t stream)
{
    const bool rgb = a.rgb != nullptr;
    switch (nw) {
    case 4: return rgb ? launch_quad_one<4, true>(a, stream) : launch_quad_one<4, false>(a, stream);
    case 6: return rgb ? launch_quad_one<6, true>(a, stream) : launch_quad_one<6, false>(a, stream);
    case 8: return rgb ? launch_quad_one<8, true>(a, stream) : launch_quad_one<8, false>(a, stream);
    case 12: return rgb ? launch_quad_one<12, true>(a, stream) : launch_quad_one<12, false>(a, stream);
    case 16: return rgb ? launch_quad_one<16, true>(a, stream) : launch_quad_one<16, false>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace mvhp
