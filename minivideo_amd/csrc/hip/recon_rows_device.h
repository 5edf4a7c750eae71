// recon_rows_device.h -- the per-macroblock arithmetic of the one-picture-per-wavefront kernels on explicit LDS pointers: the
// residual stage for a macroblock pair and the Intra8x8 / Intra16x16 / chroma prediction of a macroblock on 64 lanes.  Shared by
// recon_rows_kernel (recon_kernels.hip: one wavefront does everything for its macroblock row) and recon_pipe1_kernel
// (recon_pipe1.hip: three wavefronts per row).  `Tables` = the workgroup's LDS tables (ls4, ls8, cls8, w4, w8, tap8).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "minivideo_hotpath.h"
#include "recon_device.h"

namespace mvhp {
namespace rowsdev {

// Residual stage for a PAIR of horizontally adjacent macroblocks (residuals do not depend on neighbours, so
// two macroblocks share one pass): lanes 0-23 own the 24 4x4 blocks of macroblock 0, lanes 24-47 those of
// macroblock 1 (0-15 luma, 16-19 Cb, 20-23 Cr each).  The lane's 16 levels arrive in registers (cA, cB = the
// two 16-byte halves of its block, prefetched straight from the packed record).  Luma 8x8 blocks: the four
// lanes of an 8x8 block hold its rows (2i, 2i+1); rows are transformed in place, columns after an LDS transpose.
struct PairCtl {
    int kind[2], qpy[2], qpc_cb[2], qpc_cr[2];
    bool need[2];
    int dc_shift_from;   // ReconArgs::dc_shift_from
};

// SCALING: the weights of BlockLds::w4 / w8 are not all 16 (MVHP_PARAM_SCALING; its own instantiation, so that the flat
// case keeps its three-class LevelScale in three registers)
template <bool SCALING, class Tables>
__device__ __forceinline__ void residual_pair(int16_t (*Wres)[384], int32_t *Wscr, const Tables &B, int lane, const int4 cA, const int4 cB,
                                              const PairCtl &pc)
{
    const int sel = (lane >= 24) ? 1 : 0;
    const int b = lane - 24 * sel; // block index inside the lane's macroblock (valid for lane < 48)

    // ---- luma 8x8 (transform_8x8_residual, h264_transform.c:1205-1383), one macroblock at a time ----
#pragma unroll
    for (int s8 = 0; s8 < 2; s8++) {
        if (pc.kind[s8] != MVHP_KIND_I8x8 || !pc.need[s8]) continue;
        const int qpy = pc.qpy[s8];
        const int m = qpy % 6, s = qpy / 6;
        if (lane >= 24 * s8 && lane < 24 * s8 + 16) {
            const int blk = b >> 2, r0 = (b & 3) * 2;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int row = r0 + h;
                int d[8];
                unpack8(h ? cB : cA, d);
                // LevelScale8x8: 16 * normAdjust, or weight * normAdjust (SCALING)
                auto ls8 = [&](int j) {
                    const int v = B.ls8[m * 6 + B.cls8[row * 8 + j]];
                    return SCALING ? (v >> 4) * (int)B.w8[row * 8 + j] : v;
                };
                if (qpy > 35) {
#pragma unroll
                    for (int j = 0; j < 8; j++) d[j] = (int)((unsigned)(d[j] * ls8(j)) << ((s - 6) & 31));
                } else {
                    const int rnd = 1 << ((5 - s) & 31), sh = (6 - s) & 31;
#pragma unroll
                    for (int j = 0; j < 8; j++) d[j] = (d[j] * ls8(j) + rnd) >> sh;
                }
                if (row == 0) d[0] += 32; // rounding term of the final (m + 32) >> 6, see idct4x4
                idct8_1d(d);
#pragma unroll
                for (int j = 0; j < 8; j++) Wscr[blk * 64 + row * 8 + j] = d[j];
            }
        }
        WAVE_SYNC();
        if (lane < 32) {
            const int blk = lane >> 3, col = lane & 7;
            int d[8];
#pragma unroll
            for (int i = 0; i < 8; i++) d[i] = Wscr[blk * 64 + i * 8 + col];
            idct8_1d(d);
            const int xO = (blk & 1) * 8, yO = (blk >> 1) * 8;
#pragma unroll
            for (int i = 0; i < 8; i++)
                Wres[s8][(yO + i) * 16 + xO + col] = (int16_t)(pack_res(d[i] >> 6, 0) & 0xffff);
        }
        WAVE_SYNC();
    }

    // ---- 4x4 blocks (transform_4x4_residual, h264_transform.c:1049-1191) ----
    const int kind = sel ? pc.kind[1] : pc.kind[0];
    const int qpy = sel ? pc.qpy[1] : pc.qpy[0];
    const bool need = sel ? pc.need[1] : pc.need[0];
    const int first = (kind == MVHP_KIND_I8x8) ? 16 : 0;
    const bool act = (lane < 48) && (b >= first) && need;
    const bool all_ge24 = (pc.qpy[0] > 23) && (pc.qpc_cb[0] > 23) && (pc.qpc_cr[0] > 23) && (pc.qpy[1] > 23) &&
                          (pc.qpc_cb[1] > 23) && (pc.qpc_cr[1] > 23); // wave-uniform
    int d[16];
    if (act) {
        unpack8(cA, d);
        unpack8(cB, d + 8);
        Wscr[lane] = d[0];
    }
    WAVE_SYNC();
    if (act) {
        const bool chroma = b >= 16;
        const int qpc = (b >= 20) ? (sel ? pc.qpc_cr[1] : pc.qpc_cr[0]) : (sel ? pc.qpc_cb[1] : pc.qpc_cb[0]);
        const int qP = chroma ? qpc : qpy;
        const int m = qP % 6, s = qP / 6;
        int lsA = B.ls4[m * 3 + 0];
        const int lsB = B.ls4[m * 3 + 1], lsC = B.ls4[m * 3 + 2];
        int lsw[16];   // SCALING: LevelScale4x4 per position = weight * normAdjust (plane: Y / Cb / Cr)
        if (SCALING) {
            const uint8_t *w = B.w4[chroma ? ((b >= 20) ? 2 : 1) : 0];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int r = i >> 2, c = i & 3;
                const int ls = ((r & 1) == 0 && (c & 1) == 0) ? lsA : (((r & 1) && (c & 1)) ? lsB : lsC);
                lsw[i] = (ls >> 4) * (int)w[i];
            }
            lsA = lsw[0];   // the DC transforms use LevelScale(qP % 6, 0, 0) of their plane (8.5.10, 8.5.11.2)
        }
        int dc = d[0];
        const bool keep_dc = chroma || (kind == MVHP_KIND_I16x16);
        if (chroma) {
            // transform_2x2_chromadc, h264_transform.c:827-860, :924-936, :988-1005
            const int base = 24 * sel + ((b >= 20) ? 20 : 16), k = b & 3;
            const int c0 = Wscr[base], c1 = Wscr[base + 1], c2 = Wscr[base + 2], c3 = Wscr[base + 3];
            int f = (k == 0) ? (c0 + c1 + c2 + c3) : (k == 1) ? (c0 - c1 + c2 - c3)
                  : (k == 2) ? (c0 + c1 - c2 - c3) : (c0 - c1 - c2 + c3);
            dc = (int)((unsigned)(f * lsA) << s) >> 5;
        } else if (kind == MVHP_KIND_I16x16) {
            // transform_16x16_lumadc, h264_transform.c:756-812 (incl. the `qP > 36` test)
            const int bi = ((b >> 3) << 1) | ((b >> 1) & 1);   // block row of luma4x4BlkIdx
            const int bj = (((b >> 2) & 1) << 1) | (b & 1);    // block column
            int f = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int ri = ((q >> 3) << 1) | ((q >> 1) & 1), rj = (((q >> 2) & 1) << 1) | (q & 1);
                const int v = Wscr[24 * sel + q];            // c[ri][rj]
                f += (hneg(bi, ri) != hneg(rj, bj)) ? -v : v; // H4[bi][ri] * c * H4[rj][bj]
            }
            if (qpy >= pc.dc_shift_from) dc = (int)((unsigned)(f * lsA) << ((s - 6) & 31));
            else dc = (int)((unsigned)(f * lsA) + (1u << ((5 - s) & 31))) >> ((6 - s) & 31);
        }
        // quant4x4, h264_transform.c:1100-1134.  qP differs between lanes, so the two cases are merged:
        // ((c*LS + rnd) >> shr) << shl with (shr, rnd) = (0, 0) when qP > 23.
        if (all_ge24) {
            const int shl = s - 4;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int r = i >> 2, c = i & 3;
                const int ls = SCALING ? lsw[i] : (((r & 1) == 0 && (c & 1) == 0) ? lsA : (((r & 1) && (c & 1)) ? lsB : lsC));
                d[i] = (int)((unsigned)(d[i] * ls) << shl);
            }
        } else {
            const int shl = max(s - 4, 0), shr = max(4 - s, 0), rnd = (1 << shr) >> 1;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int r = i >> 2, c = i & 3;
                const int ls = SCALING ? lsw[i] : (((r & 1) == 0 && (c & 1) == 0) ? lsA : (((r & 1) && (c & 1)) ? lsB : lsC));
                d[i] = (int)((unsigned)((d[i] * ls + rnd) >> shr) << shl);
            }
        }
        if (keep_dc) d[0] = dc;
        d[0] += 32;
        idct4x4(d);
        int base, stride;
        if (!chroma) {
            const int xO = (((b >> 2) & 1) << 3) | ((b & 1) << 2);
            const int yO = ((b >> 3) << 3) | (((b >> 1) & 1) << 2);
            base = yO * 16 + xO; stride = 16;
        } else {
            const int k = b & 3;
            base = 256 + ((b >= 20) ? 64 : 0) + (k >> 1) * 32 + (k & 1) * 4; stride = 8;
        }
        int16_t *res = Wres[sel];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int2 pk;
            pk.x = pack_res(d[i * 4 + 0], d[i * 4 + 1]);
            pk.y = pack_res(d[i * 4 + 2], d[i * 4 + 3]);
            *reinterpret_cast<int2 *>(&res[base + i * stride]) = pk;
        }
    }
    WAVE_SYNC();
}

// ---------------------------------------------------------------------------
// prediction helpers
// ---------------------------------------------------------------------------
// Availability of the neighbours of the 16 luma 4x4 blocks, one bit per luma4x4BlkIdx
// (deriv_neighbouringlocations by geometry, h264_spatial.c:739-786; the blkIdx 3/11 rule of
// h264_intra_prediction.c:410-412).
struct Avail4 { uint32_t left, up, upleft, upright; };
__device__ __forceinline__ Avail4 avail4(bool A, bool Bv, bool C, bool D)
{
    constexpr uint32_t X0 = (1u << 0) | (1u << 2) | (1u << 8) | (1u << 10);   // blocks with xO == 0
    constexpr uint32_t Y0 = (1u << 0) | (1u << 1) | (1u << 4) | (1u << 5);    // blocks with yO == 0
    Avail4 a;
    a.left = A ? 0xffffu : (0xffffu & ~X0);
    a.up = Bv ? 0xffffu : (0xffffu & ~Y0);
    a.upleft = (0xffffu & ~(X0 | Y0)) | (Bv ? ((1u << 1) | (1u << 4) | (1u << 5)) : 0u) |
               (A ? ((1u << 2) | (1u << 8) | (1u << 10)) : 0u) | (D ? 1u : 0u);
    a.upright = ((1u << 2) | (1u << 6) | (1u << 8) | (1u << 9) | (1u << 10) | (1u << 12) | (1u << 14)) |
                (Bv ? ((1u << 0) | (1u << 1) | (1u << 4)) : 0u) | (C ? (1u << 5) : 0u);
    return a;
}

// Intra 8x8 block: edge filtering by lanes 0..27, prediction by all 64 lanes.
// h264_intra_prediction.c:1107-1353 + :1366-1793 + transform8x8_luma.
// A lane's entry of the unified Intra8x8 edge (recon_device.h mode_entry: 0-1 left[7] replicated, 2-9 left[7..0], 10 corner, 11-26
// top[0..15], 27 replicated) and where its three taps lie RELATIVE to the block's top row in the tile: the same for the four blocks
// of a macroblock, so it is worked out once per macroblock (round 4; as recon_quad.hip).
struct Edge8 {
    int e, o_e, o_lo, o_hi;
};
__device__ __forceinline__ Edge8 edge8_of(int lane)
{
    Edge8 g;
    g.e = min(max(lane, 2), 26);
    const int lo = max(g.e - 1, 2), hi = min(g.e + 1, 26);
    g.o_e = (g.e >= 10) ? g.e - 11 : 31 + (9 - g.e) * 32;
    g.o_lo = (lo >= 10) ? lo - 11 : 31 + (9 - lo) * 32;
    g.o_hi = (hi >= 10) ? hi - 11 : 31 + (9 - hi) * 32;
    return g;
}

template <class Tables>
__device__ __forceinline__ void predict_8x8(uint8_t *WT, uint8_t *WE8, const Tables &B, int lane, const Edge8 &g, int blk, int mode,
                                            bool A, bool Bv, bool C, bool D, bool has_res, const int16_t *res)
{
    const int xO = (blk & 1) * 8, yO = (blk >> 1) * 8;
    const bool left = (xO > 0) || A;
    const bool up = (yO > 0) || Bv;
    const bool upleft = (xO > 0) ? ((yO > 0) || Bv) : ((yO > 0) ? A : D);
    const bool upright = (blk == 0) ? Bv : (blk == 1) ? C : (blk == 2);
    const uint8_t *Trow = &WT[yO * 32 + 16 + xO];
    if (lane < 28) {
        // a missing side: the neighbour is the sample itself (h264_intra_prediction.c:1295-1353); no up-right block: the taps
        // beyond top[7] read top[7] (:1230-1236)
        const int e = g.e;
        int a_lo = (((e == 11) && !upleft) || ((e == 10) && !left)) ? g.o_e : g.o_lo;
        int a_hi = (((e == 9) && !upleft) || ((e == 10) && !up)) ? g.o_e : g.o_hi;
        int a_e = g.o_e;
        if (!upright) {
            a_lo = (e > 19) ? 7 : a_lo;
            a_e = (e > 18) ? 7 : a_e;
            a_hi = (e > 17) ? 7 : a_hi;
        }
        const int v0 = Trow[a_lo], v1 = Trow[a_e], v2 = Trow[a_hi];
        WE8[lane] = (uint8_t)((v0 + 2 * v1 + v2 + 2) >> 2);
    }
    WAVE_SYNC();
    {
        const int x = lane & 7, y = lane >> 3;
        int pred = 0;
        if (mode == 2) {
            const uint32_t *E = reinterpret_cast<const uint32_t *>(WE8);
            const uint32_t w0 = E[0], w1 = E[1], w2 = E[2], w3 = E[3], w4 = E[4];
            const int sumV = sum4(w0 & 0xffff0000u) + sum4(w1) + sum4(w2 & 0x0000ffffu);       // E8[2..9]
            const int sumH = sum4(w2 & 0xff000000u) + sum4(w3) + sum4(w4 & 0x00ffffffu);       // E8[11..18]
            if (left && up) pred = (sumH + sumV + 8) >> 4;
            else if (left) pred = (sumV + 4) >> 3;
            else if (up) pred = (sumH + 4) >> 3;
            else pred = 128;
        } else {
            bool ok;
            switch (mode) {
            case 0: case 3: case 7: ok = up; break;
            case 1: case 8: ok = left; break;
            default: ok = left && up && upleft; break;
            }
            if (ok && mode < 9) {
                const uint32_t e = B.tap8[mode * 64 + lane];
                const int v0 = WE8[e & 255], v1 = WE8[(e >> 8) & 255], v2 = WE8[e >> 16];
                pred = (v0 + 2 * v1 + v2 + 2) >> 2;
            }
        }
        const int r = has_res ? (int)res[(yO + y) * 16 + xO + x] : 0;
        WT[(yO + y + 1) * 32 + 16 + xO + x] = (uint8_t)clip255(pred + r);
    }
    WAVE_SYNC();
}

// Intra 16x16: 64 lanes x 4 samples. h264_intra_prediction.c:1809-2141 + transform16x16_luma.
// D: the up-left macroblock is available -- it always is when A and Bv are, except across a slice boundary, where the
// reference's code reads the corner as 0 (h264_intra_prediction.c:1839-1846: phv stays 0); a conforming stream never
// predicts Plane there
__device__ __forceinline__ void predict_16x16(uint8_t *WT, const uint8_t *WLcol, int lane, int mode, bool A, bool Bv, bool D, bool has_res,
                                              const int16_t *res)
{
    const int y = lane >> 2, x0 = (lane & 3) * 4;
    const bool left = A, up = Bv;
    const uint4 topv = *reinterpret_cast<const uint4 *>(&WT[16]);
    const uint4 lefv = *reinterpret_cast<const uint4 *>(WLcol);
    int p[4] = {0, 0, 0, 0};
    if (mode == 0) {
        if (up) {
            const uint32_t w = (lane & 3) == 0 ? topv.x : (lane & 3) == 1 ? topv.y : (lane & 3) == 2 ? topv.z : topv.w;
            p[0] = w & 255; p[1] = (w >> 8) & 255; p[2] = (w >> 16) & 255; p[3] = w >> 24;
        }
    } else if (mode == 1) {
        if (left) { const int v = WLcol[y]; p[0] = p[1] = p[2] = p[3] = v; }
    } else if (mode == 2) {
        const int sumH = sum4(topv.x) + sum4(topv.y) + sum4(topv.z) + sum4(topv.w);
        const int sumV = sum4(lefv.x) + sum4(lefv.y) + sum4(lefv.z) + sum4(lefv.w);
        int v;
        if (left && up) v = (sumH + sumV + 16) >> 5;
        else if (left) v = (sumV + 8) >> 4;
        else if (up) v = (sumH + 8) >> 4;
        else v = 128;
        p[0] = p[1] = p[2] = p[3] = v;
    } else if (mode == 3) {
        if (left && up) {
            const int cor = D ? (int)WT[15] : 0;
            const uint32_t tw[4] = {topv.x, topv.y, topv.z, topv.w};
            const uint32_t lw[4] = {lefv.x, lefv.y, lefv.z, lefv.w};
            int H = 0, V = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int hi = 8 + i, lo = 6 - i;
                const int th = (tw[hi >> 2] >> ((hi & 3) * 8)) & 255;
                const int lh = (lw[hi >> 2] >> ((hi & 3) * 8)) & 255;
                const int tl = (lo < 0) ? cor : (int)((tw[lo >> 2] >> ((lo & 3) * 8)) & 255);
                const int ll = (lo < 0) ? cor : (int)((lw[lo >> 2] >> ((lo & 3) * 8)) & 255);
                H += (i + 1) * (th - tl);
                V += (i + 1) * (lh - ll);
            }
            const int a = 16 * ((int)(lefv.w >> 24) + (int)(topv.w >> 24));
            const int b = (5 * H + 32) >> 6;
            const int c = (5 * V + 32) >> 6;
#pragma unroll
            for (int q = 0; q < 4; q++) p[q] = clip255((a + b * (x0 + q - 7) + c * (y - 7) + 16) >> 5);
        }
    }
    if (has_res) {
        const int2 rr = *reinterpret_cast<const int2 *>(&res[y * 16 + x0]);
        p[0] += (int16_t)(rr.x & 0xffff); p[1] += rr.x >> 16;
        p[2] += (int16_t)(rr.y & 0xffff); p[3] += rr.y >> 16;
    }
    const uint32_t out = (uint32_t)clip255(p[0]) | ((uint32_t)clip255(p[1]) << 8) |
                         ((uint32_t)clip255(p[2]) << 16) | ((uint32_t)clip255(p[3]) << 24);
    *reinterpret_cast<uint32_t *>(&WT[(y + 1) * 32 + 16 + x0]) = out;
    WAVE_SYNC();
}

// Chroma, both planes: lane -> plane = lane>>5, y = (lane&31)>>2, x0 = (lane&3)*2.
// h264_intra_prediction.c:2157-2564 + transform4x4_chroma.
__device__ __forceinline__ void predict_chroma(uint8_t (*WTC)[9 * 16], uint8_t (*WLcolC)[8], int lane, int mode, bool A, bool Bv, bool D, bool has_res,
                                               const int16_t *res)
{
    const int pl = lane >> 5, y = (lane & 31) >> 2, x0 = (lane & 3) * 2;
    const bool left = A, up = Bv;
    const uint8_t *TC = WTC[pl];
    const uint2 topv = *reinterpret_cast<const uint2 *>(&TC[8]);
    const uint2 lefv = *reinterpret_cast<const uint2 *>(WLcolC[pl]);
    int p0 = 0, p1 = 0;
    if (mode == 0) {
        const int bx = x0 >> 2, by = y >> 2;
        const int sH = sum4(bx ? topv.y : topv.x), sV = sum4(by ? lefv.y : lefv.x);
        int v;
        if (!left && !up) v = 128;
        else if (bx == by) {
            if (left && up) v = (sH + sV + 4) >> 3;
            else if (left) v = (sV + 2) >> 2;
            else v = (sH + 2) >> 2;
        } else if (bx == 1) { // xO > 0, yO == 0: prefers top
            v = up ? ((sH + 2) >> 2) : ((sV + 2) >> 2);
        } else {              // xO == 0, yO > 0: prefers left
            v = left ? ((sV + 2) >> 2) : ((sH + 2) >> 2);
        }
        p0 = p1 = v;
    } else if (mode == 1) {
        if (left) p0 = p1 = WLcolC[pl][y];
    } else if (mode == 2) {
        if (up) { p0 = TC[8 + x0]; p1 = TC[8 + x0 + 1]; }
    } else if (mode == 3) {
        if (left && up) {
            const int cor = D ? (int)TC[7] : 0;
            const uint32_t tw[2] = {topv.x, topv.y}, lw[2] = {lefv.x, lefv.y};
            int H = 0, V = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int hi = 4 + i, lo = 2 - i;
                const int th = (tw[hi >> 2] >> ((hi & 3) * 8)) & 255;
                const int lh = (lw[hi >> 2] >> ((hi & 3) * 8)) & 255;
                const int tl = (lo < 0) ? cor : (int)((tw[0] >> (lo * 8)) & 255);
                const int ll = (lo < 0) ? cor : (int)((lw[0] >> (lo * 8)) & 255);
                H += (i + 1) * (th - tl);
                V += (i + 1) * (lh - ll);
            }
            const int a = 16 * ((int)(lefv.y >> 24) + (int)(topv.y >> 24));
            const int b = (34 * H + 32) >> 6;
            const int c = (34 * V + 32) >> 6;
            p0 = clip255((a + b * (x0 - 3) + c * (y - 3) + 16) >> 5);
            p1 = clip255((a + b * (x0 + 1 - 3) + c * (y - 3) + 16) >> 5);
        }
    }
    if (has_res) {
        const int rr = *reinterpret_cast<const int *>(&res[256 + pl * 64 + y * 8 + x0]);
        p0 += (int16_t)(rr & 0xffff); p1 += rr >> 16;
    }
    const uint16_t out = (uint16_t)(clip255(p0) | (clip255(p1) << 8));
    *reinterpret_cast<uint16_t *>(&WTC[pl][(y + 1) * 16 + 8 + x0]) = out;
    WAVE_SYNC();
}



} // namespace rowsdev
} // namespace mvhp
