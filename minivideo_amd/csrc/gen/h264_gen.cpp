// h264_gen.cpp -- synthetic H.264 IDR stream generator (TEST / BENCH INFRASTRUCTURE,
// built into libmvgen.so; the product library does not contain or load it).
//
// Not an encoder of pictures: it draws random *syntax elements* inside the
// envelope in which the reference decoder is a conforming decoder (SURVEY.md
// 8b) and writes them as a legal Annex-B stream -- Baseline/Main/High, CAVLC or
// CABAC, 4x4 and 8x8 transforms.  Alongside the bitstream it emits the packed
// macroblock records a correct front end must produce, computed from the drawn
// syntax elements with a formulation independent of the decoder's (picture-wide
// 4x4-block maps instead of per-macroblock neighbour walks), so that
// "generate -> parse -> compare records" is a genuine two-implementation check.
// mvgen_stream_ex() additionally leaves the reference's envelope on request (what MVHP_STREAM_SPEC decodes
// by the standard): several slices per picture, I_PCM macroblocks, SPS / PPS scaling lists (with the fall-back rules
// evaluated here independently of the decoder's formulation).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "h264_tables.h"
#include "minivideo_hotpath.h"

namespace {

using namespace h264;

#include "h264_cabac_tables.inc"

struct Rng { // splitmix64
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
    int below(int n) { return (int)(next() % (uint64_t)n); }
    int geom(double p, int cap) { int k = 0; while (k < cap && uni() >= p) k++; return k; }
};

struct BitWriter {
    std::vector<uint8_t> bytes;
    int nbits = 0;
    void bit(int b)
    {
        if ((nbits & 7) == 0) bytes.push_back(0);
        if (b) bytes.back() |= (uint8_t)(0x80 >> (nbits & 7));
        nbits++;
    }
    void bits(uint32_t v, int n) { for (int i = n - 1; i >= 0; i--) bit((v >> i) & 1); }
    void ue(uint32_t v)
    {
        const uint32_t x = v + 1;
        int len = 0;
        while ((x >> len) > 1) len++;
        for (int i = 0; i < len; i++) bit(0);
        bits(x, len + 1);
    }
    void se(int v) { ue(v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
    bool aligned() const { return (nbits & 7) == 0; }
    void trailing() { bit(1); while (!aligned()) bit(0); }
};

// Annex-B NAL: 4-byte start code, header, payload with emulation prevention
void emit_nal(std::vector<uint8_t> &out, int ref_idc, int type, const std::vector<uint8_t> &rbsp)
{
    out.push_back(0); out.push_back(0); out.push_back(0); out.push_back(1);
    out.push_back((uint8_t)((ref_idc << 5) | type));
    int zeros = 0;
    for (uint8_t b : rbsp) {
        if (zeros >= 2 && b <= 3) { out.push_back(3); zeros = 0; }
        out.push_back(b);
        zeros = (b == 0) ? zeros + 1 : 0;
    }
}

// ---------------------------------------------------------------------------
// CABAC encoder (9.3.4.2)
// ---------------------------------------------------------------------------
struct CabacEnc {
    BitWriter &bw;
    uint8_t state[460], mps[460];
    uint32_t low = 0, range = 510;
    int outstanding = 0;
    bool first = true;
    explicit CabacEnc(BitWriter &w) : bw(w) {}
    void init(int qp)
    {
        for (int i = 0; i < 460; i++) {
            int pre = ((kCtxInitM[i] * qp) >> 4) + kCtxInitN[i];
            pre = pre < 1 ? 1 : (pre > 126 ? 126 : pre);
            if (pre <= 63) { state[i] = (uint8_t)(63 - pre); mps[i] = 0; }
            else { state[i] = (uint8_t)(pre - 64); mps[i] = 1; }
        }
        low = 0; range = 510; outstanding = 0; first = true;
    }
    void put(int b)
    {
        if (first) first = false; else bw.bit(b);
        while (outstanding > 0) { bw.bit(1 - b); outstanding--; }
    }
    void renorm()
    {
        while (range < 256) {
            if (low < 256) put(0);
            else if (low >= 512) { low -= 512; put(1); }
            else { low -= 256; outstanding++; }
            range <<= 1;
            low <<= 1;
        }
    }
    void decision(int ctx, int bin)
    {
        const uint32_t lps = kRangeLps[(range >> 6) & 3][state[ctx]];
        range -= lps;
        if (bin != mps[ctx]) {
            low += range;
            range = lps;
            if (state[ctx] == 0) mps[ctx] = 1 - mps[ctx];
            state[ctx] = kTransLps[state[ctx]];
        } else {
            state[ctx] = kTransMps[state[ctx]];
        }
        renorm();
    }
    void bypass(int bin)
    {
        low <<= 1;
        if (bin) low += range;
        if (low >= 1024) { put(1); low -= 1024; }
        else if (low < 512) put(0);
        else { low -= 512; outstanding++; }
    }
    void restart() { low = 0; range = 510; outstanding = 0; first = true; }   // 9.3.1.2 behind pcm samples: contexts kept
    void terminate(int bin)
    {
        range -= 2;
        if (bin) {
            low += range;
            range = 2;
            renorm();
            put((low >> 9) & 1);
            bw.bits(((low >> 7) & 3) | 1, 2);
        } else {
            renorm();
        }
    }
};

// ---------------------------------------------------------------------------
// picture-wide maps (independent neighbour formulation)
// ---------------------------------------------------------------------------
struct Maps {
    int W, H; // macroblocks
    std::vector<int8_t> mode4;     // [H*4][W*4]: Intra4x4/8x8 pred mode per 4x4 block, -1 = not NxN ("2" for prediction)
    std::vector<uint8_t> tc;       // [H*4][W*4] luma TotalCoeff
    std::vector<uint8_t> tcc[2];   // [H*2][W*2] chroma AC TotalCoeff
    std::vector<uint8_t> cbf;      // [H*4][W*4] luma coded_block_flag (CABAC)
    std::vector<uint8_t> cbfc[2];  // [H*2][W*2] chroma AC cbf
    // per macroblock
    std::vector<uint8_t> mbtype, cbp_l, cbp_c, cmode, t8, dqp_nz, cbf_dc, cbf_cdc[2], kind;
    std::vector<int> slice;        // per macroblock: index of its slice; -1 = not written yet
    int cur_slice = 0;
    // 6.4.8: a macroblock is available to the one being written when it exists, precedes it and lies in the same slice
    bool mb_avail(int mbx, int mby) const { return mbx >= 0 && mby >= 0 && mbx < W && slice[(size_t)mby * W + mbx] == cur_slice; }
    // the same for the macroblock that holds 4x4 luma block (bx, by) / 4x4 chroma block (cx, cy), picture coordinates
    bool blk_avail(int bx, int by) const { return bx >= 0 && by >= 0 && mb_avail(bx >> 2, by >> 2); }
    bool cblk_avail(int cx, int cy) const { return cx >= 0 && cy >= 0 && mb_avail(cx >> 1, cy >> 1); }
    void init(int w, int h)
    {
        W = w; H = h;
        slice.assign((size_t)W * H, -1);
        cur_slice = 0;
        mode4.assign((size_t)W * H * 16, -1);
        tc.assign((size_t)W * H * 16, 0);
        cbf.assign((size_t)W * H * 16, 0);
        for (int c = 0; c < 2; c++) { tcc[c].assign((size_t)W * H * 4, 0); cbfc[c].assign((size_t)W * H * 4, 0); }
        const size_t n = (size_t)W * H;
        mbtype.assign(n, 0); cbp_l.assign(n, 0); cbp_c.assign(n, 0); cmode.assign(n, 0); t8.assign(n, 0);
        dqp_nz.assign(n, 0); cbf_dc.assign(n, 0); kind.assign(n, 0);
        cbf_cdc[0].assign(n, 0); cbf_cdc[1].assign(n, 0);
    }
};

struct GenCfg {
    int width_mbs, height_mbs, n_frames;
    uint64_t seed;
    int profile_idc;      // 66, 77, 100
    int cabac;            // entropy_coding_mode_flag
    int transform8x8;     // PPS transform_8x8_mode_flag (profile 100 only)
    int dense;            // 1 dense, 0 light
    int cqp_offset[2];
    int sps_pps_every_frame;
    int allow_qp36_i16;
    int qp_min, qp_max;   // slice QP range
    int max_level;        // |level| cap
    // outside the reference's envelope (mvgen_stream_ex; all 0 = inside)
    int n_slices = 1;     // slices per picture
    int pcm_permille = 0; // share of I_PCM macroblocks
    int scaling = 0;      // bit 0: scaling lists in the SPS, bit 1: in the PPS (profile 100 only)
};

struct MbSyntax {
    int pcm;                   // I_PCM (mb_type 25): samples[] = 256 luma (raster), 64 Cb, 64 Cr
    uint8_t samples[384];
    int mb_type;               // 0 I_NxN, 1..24 I16x16
    int t8;
    int pred[16];              // desired final modes (16 or 4)
    int prev_flag[16], rem[16];
    int cmode;
    int cbp_l, cbp_c;
    int dqp;
    int qp;
    // levels in zig-zag order per block
    int dc16[16];
    int luma[16][16];          // 4x4 (I_NxN: 16 coefs; I16: [k] is zig-zag k+1, 15 coefs)
    int luma8[4][64];
    int cdc[2][4];
    int cac[2][4][15];
};

struct Gen {
    GenCfg cfg;
    Rng rng;
    Maps m;
    int W, H;
    explicit Gen(const GenCfg &c) : cfg(c), rng(c.seed), W(c.width_mbs), H(c.height_mbs) {}

    // ---- random syntax ----
    void rand_block(int *lev, int n, double p_coded, double geo, int first = 0)
    {
        for (int i = 0; i < n; i++) lev[i] = 0;
        if (rng.uni() >= p_coded) return;
        int total = 1 + rng.geom(geo, n - 1);
        if (total > n) total = n;
        for (int k = 0; k < total; k++) {
            int pos;
            do { pos = (int)(rng.uni() * rng.uni() * n); } while (pos >= n); // biased to low frequencies
            int v = 1 + rng.geom(0.5, cfg.max_level - 1);
            if (v > cfg.max_level) v = cfg.max_level;
            lev[pos] = (rng.next() & 1) ? -v : v;
        }
        (void)first;
    }

    int pred_mode_of(int bx, int by) const // 4x4-block coordinates in the picture
    {
        const int v = m.mode4[(size_t)by * W * 4 + bx];
        return v < 0 ? 2 : v;
    }
    // predIntraNxNPredMode for the block whose top-left 4x4 is (bx,by); n4 = 1 (4x4) or 2 (8x8)
    int predicted_mode(int bx, int by, int n4) const
    {
        if (!m.blk_avail(bx - 1, by) || !m.blk_avail(bx, by - 1)) return 2; // a neighbour macroblock is unavailable -> DC
        int mA, mB;
        if (n4 == 1) { mA = pred_mode_of(bx - 1, by); mB = pred_mode_of(bx, by - 1); }
        else {
            // 8.3.2.1: A -> the 4x4 block containing (x-1, y) sample row 0 of the 8x8 block: for an
            // Intra4x4 neighbour the standard picks blkA*4+1 (top-right 4x4 of the 8x8 to the left),
            // B picks blkB*4+2 (bottom-left 4x4 of the 8x8 above); 8x8 neighbours store the same
            // mode in all four 4x4 entries, so the map lookup serves both.
            mA = pred_mode_of(bx - 1, by);
            mB = pred_mode_of(bx, by - 1);
        }
        return mA < mB ? mA : mB;
    }

    void draw_mb(int mbx, int mby, int qp_prev, MbSyntax &s)
    {
        memset(&s, 0, sizeof(s));
        const bool A = m.mb_avail(mbx - 1, mby), B = m.mb_avail(mbx, mby - 1), C = m.mb_avail(mbx + 1, mby - 1),
                   D = m.mb_avail(mbx - 1, mby - 1);
        (void)C;
        if (cfg.pcm_permille > 0 && rng.below(1000) < cfg.pcm_permille) {
            s.pcm = 1;
            s.mb_type = 25;
            const int base = rng.below(256), spread = 1 + rng.below(64);
            for (int i = 0; i < 384; i++) {
                int v = base + rng.below(2 * spread + 1) - spread;
                if (rng.below(16) == 0) v = rng.below(2) ? 0 : 255;   // the extremes too (0x00 needs emulation prevention)
                s.samples[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
            s.qp = qp_prev;   // no mb_qp_delta: QP'Y carries over (7.4.5)
            return;
        }
        const bool i16 = rng.uni() < (cfg.dense ? 0.4 : 0.5);
        if (!i16) {
            s.mb_type = 0;
            s.t8 = cfg.transform8x8 ? (rng.uni() < 0.5) : 0;
            const int nb = s.t8 ? 4 : 16, n4 = s.t8 ? 2 : 1;
            for (int b = 0; b < nb; b++) {
                const int xO = s.t8 ? (b & 1) * 8 : blk4_x(b), yO = s.t8 ? (b >> 1) * 8 : blk4_y(b);
                const bool left = A || xO > 0, up = B || yO > 0;
                const bool upleft = xO > 0 ? (B || yO > 0) : (yO > 0 ? A : D);
                int cand[9], nc = 0;
                cand[nc++] = 2;
                if (up) { cand[nc++] = 0; cand[nc++] = 3; cand[nc++] = 7; }
                if (left) { cand[nc++] = 1; cand[nc++] = 8; }
                if (left && up && upleft) { cand[nc++] = 4; cand[nc++] = 5; cand[nc++] = 6; }
                const int bx = mbx * 4 + xO / 4, by = mby * 4 + yO / 4;
                const int pm = predicted_mode(bx, by, n4);
                int want = (cfg.dense || rng.uni() < 0.3) ? cand[rng.below(nc)] : pm;
                bool legal = false;
                for (int i = 0; i < nc; i++) legal |= cand[i] == want;
                if (!legal) want = 2;
                s.pred[b] = want;
                if (want == pm) s.prev_flag[b] = 1;
                else { s.prev_flag[b] = 0; s.rem[b] = want < pm ? want : want - 1; }
                for (int yy = 0; yy < n4; yy++)
                    for (int xx = 0; xx < n4; xx++) m.mode4[(size_t)(by + yy) * W * 4 + bx + xx] = (int8_t)want;
            }
        }
        { // chroma mode
            int cand[4], nc = 0;
            cand[nc++] = 0;
            if (A) cand[nc++] = 1;
            if (B) cand[nc++] = 2;
            if (A && B && D) cand[nc++] = 3;   // Plane reads p[-1,-1] too (with one slice per picture A && B implies D)
            s.cmode = cand[rng.below(nc)];
        }
        // ---- residual ----
        int i16mode = 0;
        if (cfg.dense) {
            const double u = rng.uni();
            s.cbp_c = u < 0.4 ? 0 : (u < 0.7 ? 1 : 2);
            if (i16) {
                s.cbp_l = (rng.uni() < 0.5) ? 15 : 0;
                rand_block(s.dc16, 16, 0.8, 0.35);
                if (s.cbp_l) for (int b = 0; b < 16; b++) rand_block(s.luma[b], 15, 0.5, 0.35);
            } else if (s.t8 && cfg.cabac) {
                for (int k = 0; k < 4; k++) {
                    rand_block(s.luma8[k], 64, 0.6, 0.2);
                    bool any = false;
                    for (int i = 0; i < 64; i++) any |= s.luma8[k][i] != 0;
                    if (any) s.cbp_l |= 1 << k;
                }
            } else {
                for (int k = 0; k < 4; k++) {
                    bool any = false;
                    if (rng.uni() < 0.7)
                        for (int i4 = 0; i4 < 4; i4++) {
                            rand_block(s.luma[k * 4 + i4], 16, 0.6, 0.35);
                            for (int i = 0; i < 16; i++) any |= s.luma[k * 4 + i4][i] != 0;
                        }
                    if (any) s.cbp_l |= 1 << k;
                    else for (int i4 = 0; i4 < 4; i4++) memset(s.luma[k * 4 + i4], 0, sizeof(s.luma[0]));
                }
            }
            if (s.cbp_c) {
                for (int c = 0; c < 2; c++) rand_block(s.cdc[c], 4, 0.8, 0.5);
                if (s.cbp_c == 2)
                    for (int c = 0; c < 2; c++)
                        for (int b = 0; b < 4; b++) rand_block(s.cac[c][b], 15, 0.5, 0.4);
            }
        } else if (i16) {
            static const int vals[4] = {-3, -2, 2, 3};
            s.dc16[rng.below(16)] = vals[rng.below(4)];
        }
        if (i16) {
            int cand[4], nc = 0;
            cand[nc++] = 2;
            if (B) cand[nc++] = 0;
            if (A) cand[nc++] = 1;
            if (A && B && D) cand[nc++] = 3;
            i16mode = cand[rng.below(nc)];
            s.mb_type = 1 + i16mode + 4 * s.cbp_c + (s.cbp_l ? 12 : 0);
        }
        // ---- QP ----
        const bool has_dqp = i16 || s.cbp_l || s.cbp_c;
        s.dqp = 0;
        if (has_dqp && rng.uni() < 0.1) s.dqp = rng.below(5) - 2;
        int qp = qp_prev;
        if (s.dqp) qp = (qp_prev + s.dqp + 52) % 52;
        if (i16 && qp == 36 && !cfg.allow_qp36_i16) { // stay inside the reference's envelope (h264_transform.c:797)
            s.dqp = (qp_prev == 36) ? 1 : 37 - qp_prev;
            qp = (qp_prev + s.dqp + 52) % 52;
        }
        s.qp = qp;
    }

    // ---- expected packed record (independent of the decoder) ----
    uint8_t unavail_bits(int mbx, int mby) const
    {
        uint8_t u = 0;   // neighbours that geometry has but the slice structure takes away
        if (mbx > 0 && !m.mb_avail(mbx - 1, mby)) u |= MVHP_UNAVAIL_A;
        if (mby > 0 && !m.mb_avail(mbx, mby - 1)) u |= MVHP_UNAVAIL_B;
        if (mby > 0 && mbx < W - 1 && !m.mb_avail(mbx + 1, mby - 1)) u |= MVHP_UNAVAIL_C;
        if (mbx > 0 && mby > 0 && !m.mb_avail(mbx - 1, mby - 1)) u |= MVHP_UNAVAIL_D;
        return u;
    }

    void fill_record(const MbSyntax &s, uint8_t *rec, int mbx, int mby)
    {
        memset(rec, 0, MVHP_MB_BYTES);
        if (s.pcm) {
            mvhp_mb_header_t h;
            memset(&h, 0, sizeof(h));
            h.mb_kind = MVHP_KIND_IPCM;
            h.qp_y = (uint8_t)s.qp;
            h.unavail = unavail_bits(mbx, mby);
            memcpy(rec, &h, sizeof(h));
            uint8_t *area = rec + MVHP_MB_HEADER_BYTES;
            for (int y = 0; y < 16; y++) memcpy(area + 64 * (y >> 1) + 16 * (y & 1), s.samples + 16 * y, 16);
            for (int y = 0; y < 8; y++) {
                memcpy(area + 64 * y + 32, s.samples + 256 + 8 * y, 8);
                memcpy(area + 64 * y + 40, s.samples + 320 + 8 * y, 8);
            }
            return;
        }
        int16_t *coef = reinterpret_cast<int16_t *>(rec + MVHP_MB_HEADER_BYTES);
        const bool i16 = s.mb_type != 0;
        const int kind = i16 ? MVHP_KIND_I16x16 : (s.t8 ? MVHP_KIND_I8x8 : MVHP_KIND_I4x4);
        if (i16) {
            for (int i = 0; i < 16; i++) {
                const int r = kZigzag4x4[i] >> 2, c = kZigzag4x4[i] & 3; // DC matrix c1[r][c]
                coef[blk4_from_xy(c * 4, r * 4) * 16] = (int16_t)s.dc16[i];
            }
            if (s.cbp_l)
                for (int b = 0; b < 16; b++)
                    for (int k = 0; k < 15; k++) coef[b * 16 + kZigzag4x4[k + 1]] = (int16_t)s.luma[b][k];
        } else if (s.t8) {
            for (int k8 = 0; k8 < 4; k8++) {
                if (!(s.cbp_l & (1 << k8))) continue;
                if (cfg.cabac) {
                    for (int i = 0; i < 64; i++) coef[k8 * 64 + kZigzag8x8[i]] = (int16_t)s.luma8[k8][i];
                } else {
                    for (int i4 = 0; i4 < 4; i4++)
                        for (int i = 0; i < 16; i++) coef[k8 * 64 + kZigzag8x8[4 * i + i4]] = (int16_t)s.luma[k8 * 4 + i4][i];
                }
            }
        } else {
            for (int b = 0; b < 16; b++)
                if (s.cbp_l & (1 << (b >> 2)))
                    for (int i = 0; i < 16; i++) coef[b * 16 + kZigzag4x4[i]] = (int16_t)s.luma[b][i];
        }
        for (int c = 0; c < 2; c++) {
            if (s.cbp_c) for (int k = 0; k < 4; k++) coef[256 + c * 64 + k * 16] = (int16_t)s.cdc[c][k];
            if (s.cbp_c == 2)
                for (int b = 0; b < 4; b++)
                    for (int k = 0; k < 15; k++) coef[256 + c * 64 + b * 16 + kZigzag4x4[k + 1]] = (int16_t)s.cac[c][b][k];
        }
        mvhp_mb_header_t h;
        memset(&h, 0, sizeof(h));
        h.mb_kind = (uint8_t)kind;
        h.qp_y = (uint8_t)s.qp;
        h.cbp = (uint8_t)(s.cbp_l | (s.cbp_c << 4));
        h.chroma_pred_mode = (uint8_t)s.cmode;
        h.i16_pred_mode = (uint8_t)(i16 ? (s.mb_type - 1) % 4 : 0);
        h.unavail = unavail_bits(mbx, mby);
        if (!i16) for (int b = 0; b < (s.t8 ? 4 : 16); b++) h.pred_mode[b] = (uint8_t)s.pred[b];
        uint32_t nz = 0;
        for (int b = 0; b < 24; b++) {
            bool any = false;
            for (int i = 0; i < 16; i++) any |= coef[b * 16 + i] != 0;
            if (any) nz |= 1u << b;
        }
        if (kind == MVHP_KIND_I8x8)
            for (int k = 0; k < 4; k++) if (nz & (0xfu << (4 * k))) nz |= 0xfu << (4 * k);
        h.nz_mask = nz;
        memcpy(rec, &h, sizeof(h));
    }

    // ---- CAVLC writer (9.2 inverted) ----
    static int count_nz(const int *lev, int n) { int c = 0; for (int i = 0; i < n; i++) c += lev[i] != 0; return c; }

    void write_cavlc_block(BitWriter &bw, const int *lev, int maxNum, int nC)
    {
        int idx[16], total = 0;
        for (int i = 0; i < maxNum; i++) if (lev[i]) idx[total++] = i;
        int t1s = 0;
        for (int k = total - 1; k >= 0 && t1s < 3; k--) { if (abs(lev[idx[k]]) == 1) t1s++; else break; }
        // coeff_token
        if (nC >= 8) {
            bw.bits(total == 0 ? 3u : (uint32_t)(((total - 1) << 2) | t1s), 6);
        } else if (nC == -1) {
            bw.bits(kCoeffTokenChromaDcCode[t1s][total], kCoeffTokenChromaDcLen[t1s][total]);
        } else {
            const int tab = nC < 2 ? 0 : (nC < 4 ? 1 : 2);
            bw.bits(kCoeffTokenCode[tab][t1s][total], kCoeffTokenLen[tab][t1s][total]);
        }
        if (total == 0) return;
        int suffixLength = (total > 10 && t1s < 3) ? 1 : 0;
        for (int i = 0; i < total; i++) {
            const int level = lev[idx[total - 1 - i]];
            if (i < t1s) { bw.bit(level < 0); continue; }
            int levelCode = level > 0 ? 2 * level - 2 : -2 * level - 1;
            if (i == t1s && t1s < 3) levelCode -= 2;
            if (suffixLength == 0) {
                if (levelCode < 14) { for (int z = 0; z < levelCode; z++) bw.bit(0); bw.bit(1); }
                else if (levelCode < 30) { for (int z = 0; z < 14; z++) bw.bit(0); bw.bit(1); bw.bits((uint32_t)(levelCode - 14), 4); }
                else { for (int z = 0; z < 15; z++) bw.bit(0); bw.bit(1); bw.bits((uint32_t)(levelCode - 30), 12); }
            } else {
                if (levelCode < (15 << suffixLength)) {
                    const int prefix = levelCode >> suffixLength;
                    for (int z = 0; z < prefix; z++) bw.bit(0);
                    bw.bit(1);
                    bw.bits((uint32_t)(levelCode & ((1 << suffixLength) - 1)), suffixLength);
                } else {
                    for (int z = 0; z < 15; z++) bw.bit(0);
                    bw.bit(1);
                    bw.bits((uint32_t)(levelCode - (15 << suffixLength)), 12);
                }
            }
            if (suffixLength == 0) suffixLength = 1;
            if (abs(level) > (3 << (suffixLength - 1)) && suffixLength < 6) suffixLength++;
        }
        if (total < maxNum) {
            const int total_zeros = idx[total - 1] + 1 - total;
            if (nC == -1) bw.bits(kTotalZerosChromaDcCode[total - 1][total_zeros], kTotalZerosChromaDcLen[total - 1][total_zeros]);
            else bw.bits(kTotalZerosCode[total - 1][total_zeros], kTotalZerosLen[total - 1][total_zeros]);
            int zerosLeft = total_zeros;
            for (int i = total - 1; i > 0 && zerosLeft > 0; i--) {
                const int run = idx[i] - idx[i - 1] - 1;
                const int v = (zerosLeft - 1 < 6) ? zerosLeft - 1 : 6;
                bw.bits(kRunBeforeCode[v][run], kRunBeforeLen[v][run]);
                zerosLeft -= run;
            }
        }
    }

    int nC_luma(int bx, int by) const
    {
        const bool a = m.blk_avail(bx - 1, by), b = m.blk_avail(bx, by - 1);
        const int nA = a ? m.tc[(size_t)by * W * 4 + bx - 1] : 0, nB = b ? m.tc[(size_t)(by - 1) * W * 4 + bx] : 0;
        if (a && b) return (nA + nB + 1) >> 1;
        return a ? nA : (b ? nB : 0);
    }
    int nC_chroma(int c, int bx, int by) const
    {
        const bool a = m.cblk_avail(bx - 1, by), b = m.cblk_avail(bx, by - 1);
        const int nA = a ? m.tcc[c][(size_t)by * W * 2 + bx - 1] : 0, nB = b ? m.tcc[c][(size_t)(by - 1) * W * 2 + bx] : 0;
        if (a && b) return (nA + nB + 1) >> 1;
        return a ? nA : (b ? nB : 0);
    }

    // what later macroblocks derive from an I_PCM macroblock: nC = 16 (9.2.1), coded_block_flag = 1 (9.3.3.1.1.9), mb_type !=
    // I_NxN, CodedBlockPattern 47 (9.3.3.1.1.4), intra_chroma_pred_mode 0, no transform_size_8x8_flag, mb_qp_delta 0
    void publish_pcm(int mbx, int mby)
    {
        const int addr = mby * W + mbx;
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
                m.tc[(size_t)(mby * 4 + y) * W * 4 + mbx * 4 + x] = 16;
                m.cbf[(size_t)(mby * 4 + y) * W * 4 + mbx * 4 + x] = 1;
                m.mode4[(size_t)(mby * 4 + y) * W * 4 + mbx * 4 + x] = -1;
            }
        for (int c = 0; c < 2; c++)
            for (int y = 0; y < 2; y++)
                for (int x = 0; x < 2; x++) {
                    m.tcc[c][(size_t)(mby * 2 + y) * W * 2 + mbx * 2 + x] = 16;
                    m.cbfc[c][(size_t)(mby * 2 + y) * W * 2 + mbx * 2 + x] = 1;
                }
        m.mbtype[addr] = 25; m.cbp_l[addr] = 15; m.cbp_c[addr] = 2; m.cmode[addr] = 0; m.t8[addr] = 0;
        m.kind[addr] = 3; m.dqp_nz[addr] = 0; m.cbf_dc[addr] = 1; m.cbf_cdc[0][addr] = 1; m.cbf_cdc[1][addr] = 1;
    }
    static void write_pcm_samples(BitWriter &bw, const MbSyntax &s)
    {
        while (!bw.aligned()) bw.bit(0);   // pcm_alignment_zero_bit
        for (int i = 0; i < 384; i++) bw.bits(s.samples[i], 8);
    }

    void write_mb_cavlc(BitWriter &bw, int mbx, int mby, const MbSyntax &s)
    {
        const bool i16 = s.mb_type != 0;
        bw.ue((uint32_t)s.mb_type);
        if (s.pcm) { write_pcm_samples(bw, s); publish_pcm(mbx, mby); return; }
        if (!i16) {
            if (cfg.transform8x8) bw.bit(s.t8);
            for (int b = 0; b < (s.t8 ? 4 : 16); b++) { bw.bit(s.prev_flag[b]); if (!s.prev_flag[b]) bw.bits((uint32_t)s.rem[b], 3); }
        }
        bw.ue((uint32_t)s.cmode);
        if (!i16) {
            const int cbp = s.cbp_l | (s.cbp_c << 4);
            int code = -1;
            for (int i = 0; i < 48; i++) if (kCbpIntraFromCodeNum[i] == cbp) code = i;
            bw.ue((uint32_t)code);
        }
        if (!(i16 || s.cbp_l || s.cbp_c)) return;
        bw.se(s.dqp);
        if (i16) {
            write_cavlc_block(bw, s.dc16, 16, nC_luma(mbx * 4, mby * 4));
        }
        for (int b = 0; b < 16; b++) {
            const int bx = mbx * 4 + blk4_x(b) / 4, by = mby * 4 + blk4_y(b) / 4;
            uint8_t &tc = m.tc[(size_t)by * W * 4 + bx];
            if (!(s.cbp_l & (1 << (b >> 2)))) { tc = 0; continue; }
            const int n = i16 ? 15 : 16;
            write_cavlc_block(bw, s.luma[b], n, nC_luma(bx, by));
            tc = (uint8_t)count_nz(s.luma[b], n);
        }
        if (s.cbp_c) for (int c = 0; c < 2; c++) write_cavlc_block(bw, s.cdc[c], 4, -1);
        if (s.cbp_c == 2)
            for (int c = 0; c < 2; c++)
                for (int b = 0; b < 4; b++) {
                    const int bx = mbx * 2 + (b & 1), by = mby * 2 + (b >> 1);
                    write_cavlc_block(bw, s.cac[c][b], 15, nC_chroma(c, bx, by));
                    m.tcc[c][(size_t)by * W * 2 + bx] = (uint8_t)count_nz(s.cac[c][b], 15);
                }
    }

    // ---- CABAC writer ----
    void cabac_residual(CabacEnc &e, const int *lev, int maxNum, int cat, int cbf_inc, int *cbf_out)
    {
        static const int kCbfOff[8] = {0, 8, 0, 4, 12, 12, 16, 16};
        static const int kSigOff[8] = {0, 29, 0, 15, 44, 44, 47, 47};
        static const int kAbsOff[8] = {0, 20, 0, 10, 30, 30, 39, 39};
        const bool is8 = cat == 0, cdc = (cat == 4 || cat == 5);
        int total = 0, last = -1;
        for (int i = 0; i < maxNum; i++) if (lev[i]) { total++; last = i; }
        if (!is8) { e.decision(85 + kCbfOff[cat] + cbf_inc, total != 0); }
        if (cbf_out) *cbf_out = is8 ? 1 : (total != 0);
        if (total == 0) return; // (an 8x8 block is only written when its cbp bit is set, i.e. non-empty)
        const int sig_base = is8 ? 402 : 105 + kSigOff[cat], last_base = is8 ? 417 : 166 + kSigOff[cat];
        const int abs_base = is8 ? 426 : 227 + kAbsOff[cat];
        for (int i = 0; i < maxNum - 1; i++) {
            const int inc_s = is8 ? kSigInc8x8[i] : (cdc ? (i < 2 ? i : 2) : i);
            e.decision(sig_base + inc_s, lev[i] != 0);
            if (lev[i]) {
                const int inc_l = is8 ? kLastInc8x8[i] : (cdc ? (i < 2 ? i : 2) : i);
                e.decision(last_base + inc_l, i == last);
                if (i == last) break;
            }
        }
        int eq1 = 0, gt1 = 0;
        for (int i = last; i >= 0; i--) {
            if (!lev[i]) continue;
            const int a = abs(lev[i]) - 1;
            const int inc0 = gt1 ? 0 : ((1 + eq1) < 4 ? 1 + eq1 : 4);
            const int lim = 4 - (cdc ? 1 : 0);
            const int incn = 5 + (gt1 < lim ? gt1 : lim);
            if (a == 0) e.decision(abs_base + inc0, 0);
            else {
                e.decision(abs_base + inc0, 1);
                const int pre = a < 14 ? a : 14;
                for (int k = 1; k < pre; k++) e.decision(abs_base + incn, 1);
                if (a < 14) e.decision(abs_base + incn, 0);
                else {
                    int suf = a - 14, k = 0;
                    for (;;) {
                        if (suf >= (1 << k)) { e.bypass(1); suf -= 1 << k; k++; }
                        else { e.bypass(0); while (k--) e.bypass((suf >> k) & 1); break; }
                    }
                }
            }
            e.bypass(lev[i] < 0);
            if (a == 0) eq1++; else gt1++;
        }
    }

    void write_mb_cabac(CabacEnc &e, int mbx, int mby, const MbSyntax &s)
    {
        const int addr = mby * W + mbx, a = m.mb_avail(mbx - 1, mby) ? addr - 1 : -1, b = m.mb_avail(mbx, mby - 1) ? addr - W : -1;
        const bool i16 = s.mb_type != 0;
        if (s.pcm) {   // mb_type I_PCM: prefix bin, terminate bin 1 (= EncodeFlush), alignment, samples, encoder restarted
            const int inc = ((a >= 0 && m.mbtype[a] != 0) ? 1 : 0) + ((b >= 0 && m.mbtype[b] != 0) ? 1 : 0);
            e.decision(3 + inc, 1);
            e.terminate(1);
            write_pcm_samples(e.bw, s);
            e.restart();
            publish_pcm(mbx, mby);
            return;
        }
        { // mb_type
            const int inc = ((a >= 0 && m.mbtype[a] != 0) ? 1 : 0) + ((b >= 0 && m.mbtype[b] != 0) ? 1 : 0);
            if (!i16) e.decision(3 + inc, 0);
            else {
                e.decision(3 + inc, 1);
                e.terminate(0);
                const int t = s.mb_type - 1, pm = t % 4, ch = (t / 4) % 3, lu = t / 12;
                e.decision(3 + 3, lu);
                e.decision(3 + 4, ch != 0);
                if (ch) e.decision(3 + 5, ch == 2);
                e.decision(3 + 6, pm >> 1);
                e.decision(3 + 7, pm & 1);
            }
        }
        if (!i16) {
            if (cfg.transform8x8) {
                const int inc = ((a >= 0 && m.t8[a]) ? 1 : 0) + ((b >= 0 && m.t8[b]) ? 1 : 0);
                e.decision(399 + inc, s.t8);
            }
            for (int k = 0; k < (s.t8 ? 4 : 16); k++) {
                e.decision(68, s.prev_flag[k]);
                if (!s.prev_flag[k]) { e.decision(69, s.rem[k] & 1); e.decision(69, (s.rem[k] >> 1) & 1); e.decision(69, (s.rem[k] >> 2) & 1); }
            }
        }
        { // intra_chroma_pred_mode
            const int inc = ((a >= 0 && m.cmode[a] != 0) ? 1 : 0) + ((b >= 0 && m.cmode[b] != 0) ? 1 : 0);
            e.decision(64 + inc, s.cmode != 0);
            if (s.cmode != 0) { e.decision(67, s.cmode != 1); if (s.cmode != 1) e.decision(67, s.cmode != 2); }
        }
        if (!i16) { // coded_block_pattern
            for (int b8 = 0; b8 < 4; b8++) {
                int cA, cB;
                if (b8 & 1) cA = ((s.cbp_l >> (b8 - 1)) & 1) ? 0 : 1; else cA = a >= 0 ? (((m.cbp_l[a] >> (b8 + 1)) & 1) ? 0 : 1) : 0;
                if (b8 & 2) cB = ((s.cbp_l >> (b8 - 2)) & 1) ? 0 : 1; else cB = b >= 0 ? (((m.cbp_l[b] >> (b8 + 2)) & 1) ? 0 : 1) : 0;
                e.decision(73 + cA + 2 * cB, (s.cbp_l >> b8) & 1);
            }
            const int cA = (a >= 0 && m.cbp_c[a] != 0) ? 1 : 0, cB = (b >= 0 && m.cbp_c[b] != 0) ? 1 : 0;
            e.decision(77 + cA + 2 * cB, s.cbp_c != 0);
            if (s.cbp_c) {
                const int dA = (a >= 0 && m.cbp_c[a] == 2) ? 1 : 0, dB = (b >= 0 && m.cbp_c[b] == 2) ? 1 : 0;
                e.decision(77 + 4 + dA + 2 * dB, s.cbp_c == 2);
            }
        }
        // publish per-MB state BEFORE residual so that in-MB lookups see this MB's cbp
        m.mbtype[addr] = (uint8_t)s.mb_type; m.cbp_l[addr] = (uint8_t)s.cbp_l; m.cbp_c[addr] = (uint8_t)s.cbp_c;
        m.cmode[addr] = (uint8_t)s.cmode; m.t8[addr] = (uint8_t)s.t8;
        m.kind[addr] = (uint8_t)(i16 ? 2 : (s.t8 ? 1 : 0));
        m.dqp_nz[addr] = 0;
        if (!(i16 || s.cbp_l || s.cbp_c)) return;
        { // mb_qp_delta
            int inc = 0;
            if (addr > 0 && m.slice[(size_t)addr - 1] == m.cur_slice) {   // the previous macroblock in decoding order of this slice
                const int p = addr - 1;
                const bool nores = (m.kind[p] != 2) && m.cbp_l[p] == 0 && m.cbp_c[p] == 0;
                inc = (!nores && m.dqp_nz[p]) ? 1 : 0;
            }
            const int k = s.dqp > 0 ? 2 * s.dqp - 1 : -2 * s.dqp;
            if (k == 0) e.decision(60 + inc, 0);
            else {
                e.decision(60 + inc, 1);
                if (k == 1) e.decision(62, 0);
                else {
                    e.decision(62, 1);
                    for (int i = 2; i < k; i++) e.decision(63, 1);
                    e.decision(63, 0);
                }
            }
            m.dqp_nz[addr] = s.dqp != 0;
        }
        // coded_block_flag ctxIdxInc from picture-wide maps: unavailable -> 1 (intra), uncoded 8x8 region -> 0
        auto luma_inc = [&](int bx, int by) {
            const int cA = m.blk_avail(bx - 1, by) ? m.cbf[(size_t)by * W * 4 + bx - 1] : 1;
            const int cB = m.blk_avail(bx, by - 1) ? m.cbf[(size_t)(by - 1) * W * 4 + bx] : 1;
            return cA + 2 * cB;
        };
        if (i16) {
            const int cA = a >= 0 ? (m.kind[a] >= 2 ? m.cbf_dc[a] : 0) : 1;   // (kind 3 = I_PCM: always 1)
            const int cB = b >= 0 ? (m.kind[b] >= 2 ? m.cbf_dc[b] : 0) : 1;
            int f = 0;
            cabac_residual(e, s.dc16, 16, 2, cA + 2 * cB, &f);
            m.cbf_dc[addr] = (uint8_t)f;
        }
        if (!i16 && s.t8) {
            for (int k8 = 0; k8 < 4; k8++) {
                const int bx = mbx * 4 + (k8 & 1) * 2, by = mby * 4 + (k8 >> 1) * 2;
                const int coded = (s.cbp_l >> k8) & 1;
                if (coded) cabac_residual(e, s.luma8[k8], 64, 0, 0, nullptr);
                for (int yy = 0; yy < 2; yy++) for (int xx = 0; xx < 2; xx++) m.cbf[(size_t)(by + yy) * W * 4 + bx + xx] = (uint8_t)coded;
            }
        } else {
            for (int blk = 0; blk < 16; blk++) {
                const int bx = mbx * 4 + blk4_x(blk) / 4, by = mby * 4 + blk4_y(blk) / 4;
                uint8_t &f = m.cbf[(size_t)by * W * 4 + bx];
                if (!(s.cbp_l & (1 << (blk >> 2)))) { f = 0; continue; }
                int fl = 0;
                cabac_residual(e, s.luma[blk], i16 ? 15 : 16, i16 ? 3 : 1, luma_inc(bx, by), &fl);
                f = (uint8_t)fl;
            }
        }
        if (s.cbp_c) {
            for (int c = 0; c < 2; c++) {
                const int cA = a >= 0 ? (m.cbp_c[a] != 0 ? m.cbf_cdc[c][a] : 0) : 1;
                const int cB = b >= 0 ? (m.cbp_c[b] != 0 ? m.cbf_cdc[c][b] : 0) : 1;
                int f = 0;
                cabac_residual(e, s.cdc[c], 4, 4 + c, cA + 2 * cB, &f);
                m.cbf_cdc[c][addr] = (uint8_t)f;
            }
        }
        if (s.cbp_c == 2) {
            for (int c = 0; c < 2; c++)
                for (int blk = 0; blk < 4; blk++) {
                    const int bx = mbx * 2 + (blk & 1), by = mby * 2 + (blk >> 1);
                    const int cA = m.cblk_avail(bx - 1, by) ? m.cbfc[c][(size_t)by * W * 2 + bx - 1] : 1;
                    const int cB = m.cblk_avail(bx, by - 1) ? m.cbfc[c][(size_t)(by - 1) * W * 2 + bx] : 1;
                    int f = 0;
                    cabac_residual(e, s.cac[c][blk], 15, 6 + c, cA + 2 * cB, &f);
                    m.cbfc[c][(size_t)by * W * 2 + bx] = (uint8_t)f;
                }
        }
    }

    // ---- scaling lists (7.3.2.1.1.1; SPS and PPS of profile 100) ----
    // A level's eight lists as drawn: state 0 = not transmitted, 1 = transmitted, 2 = transmitted as "use the default".
    struct ListSet {
        int present = 0;
        int state[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint8_t v[8][64];
    };
    ListSet sps_lists, pps_lists;
    void draw_lists(ListSet &ls, int n_lists)
    {
        ls.present = 1;
        for (int i = 0; i < 8; i++) {
            ls.state[i] = 0;
            if (i >= n_lists) continue;
            const int n = i < 6 ? 16 : 64, r = rng.below(10);
            ls.state[i] = r < 3 ? 0 : (r < 5 ? 2 : 1);
            if (ls.state[i] != 1) continue;
            // a smooth ramp with noise, like real matrices, inside 1..255; now and then the list ends early (the rest repeats)
            int cur = 4 + rng.below(28);
            const int stop = rng.below(4) == 0 ? 1 + rng.below(n - 1) : n;
            for (int j = 0; j < n; j++) {
                if (j < stop) { cur += rng.below(7) - 2; cur = cur < 1 ? 1 : (cur > 255 ? 255 : cur); }
                ls.v[i][j] = (uint8_t)cur;
            }
            if (rng.below(8) == 0) ls.v[i][rng.below(n)] = (uint8_t)(200 + rng.below(56));   // a large weight somewhere
            if (stop < n) for (int j = stop; j < n; j++) ls.v[i][j] = ls.v[i][stop - 1];
        }
    }
    static void write_list(BitWriter &bw, const uint8_t *v, int n, bool use_default)
    {
        if (use_default) { bw.se(-8); return; }   // nextScale = (8 - 8) % 256 = 0 at j = 0: useDefaultScalingMatrixFlag
        int last = 8;
        for (int j = 0; j < n; j++) {
            // the rest of the list equal to the last value: one delta that makes nextScale 0 ends the transmission
            bool rest_equal = j > 0;
            for (int k = j; k < n && rest_equal; k++) rest_equal = v[k] == last;
            if (rest_equal) { int d = -last; if (d < -128) d += 256; bw.se(d); return; }
            int d = (int)v[j] - last;
            if (d > 127) d -= 256;
            if (d < -128) d += 256;
            bw.se(d);
            last = v[j];
        }
    }
    void write_list_set(BitWriter &bw, const ListSet &ls, int n_lists)
    {
        for (int i = 0; i < n_lists; i++) {
            bw.bit(ls.state[i] != 0);
            if (ls.state[i]) write_list(bw, ls.v[i], i < 6 ? 16 : 64, ls.state[i] == 2);
        }
    }
    // The weights an Intra picture ends up with, RASTER order -- the generator's own reading of Table 7-2: walk each list's
    // chain of fall-backs (set A: default, then "the list before"; set B: the sequence level instead of the default).
    void effective_weights(uint8_t w4[3][16], uint8_t w8[64]) const
    {
        static const uint8_t dflt4[16] = {6, 13, 13, 20, 20, 20, 28, 28, 28, 28, 32, 32, 32, 37, 37, 42};
        static const uint8_t dflt8[64] = {6,  10, 10, 13, 11, 13, 16, 16, 16, 16, 18, 18, 18, 18, 18, 23, 23, 23, 23, 23, 23, 25,
                                          25, 25, 25, 25, 25, 25, 27, 27, 27, 27, 27, 27, 27, 27, 29, 29, 29, 29, 29, 29, 29, 31,
                                          31, 31, 31, 31, 31, 33, 33, 33, 33, 33, 36, 36, 36, 36, 38, 38, 38, 40, 40, 42};
        auto seq_level = [&](int i, uint8_t *out) {   // Intra lists only: i = 0, 1, 2 (4x4) or 6 (8x8)
            const int n = i < 6 ? 16 : 64;
            if (!sps_lists.present) { memset(out, 16, (size_t)n); return; }
            int k = i;
            while (k != 0 && k != 6 && sps_lists.state[k] == 0) k--;          // Cb <- Y, Cr <- Cb
            if (sps_lists.state[k] == 1) memcpy(out, sps_lists.v[k], (size_t)n);
            else memcpy(out, i < 6 ? dflt4 : dflt8, (size_t)n);                // "use default", or fall-back rule A at the head
        };
        auto pic_level = [&](int i, uint8_t *out) {
            const int n = i < 6 ? 16 : 64;
            if (!pps_lists.present) { seq_level(i, out); return; }
            int k = i;
            const bool has8 = cfg.transform8x8 != 0;
            auto st = [&](int q) { return (q >= 6 && !has8) ? 0 : pps_lists.state[q]; };
            while (k != 0 && k != 6 && st(k) == 0) k--;
            if (st(k) == 1) memcpy(out, pps_lists.v[k], (size_t)n);
            else if (st(k) == 2) memcpy(out, i < 6 ? dflt4 : dflt8, (size_t)n);
            else if (sps_lists.present) seq_level(k, out);                      // rule B: the sequence-level list of the head
            else memcpy(out, i < 6 ? dflt4 : dflt8, (size_t)n);                // rule A
        };
        uint8_t z[64];
        for (int pl = 0; pl < 3; pl++) {
            pic_level(pl, z);
            for (int k = 0; k < 16; k++) w4[pl][kZigzag4x4[k]] = z[k];
        }
        pic_level(6, z);
        for (int k = 0; k < 64; k++) w8[kZigzag8x8[k]] = z[k];
    }

    // ---- parameter sets / slice ----
    void write_sps(std::vector<uint8_t> &out)
    {
        BitWriter bw;
        bw.bits((uint32_t)cfg.profile_idc, 8);
        bw.bits(0, 8);           // constraint flags + reserved_zero_2bits
        bw.bits(40, 8);          // level_idc
        bw.ue(0);                // seq_parameter_set_id
        if (cfg.profile_idc == 100) {
            bw.ue(1); bw.ue(0); bw.ue(0); bw.bit(0);
            if (cfg.scaling & 1) { bw.bit(1); write_list_set(bw, sps_lists, 8); }   // seq_scaling_matrix_present_flag
            else bw.bit(0);
        }
        bw.ue(0);                // log2_max_frame_num_minus4
        bw.ue(0);                // pic_order_cnt_type
        bw.ue(0);                // log2_max_pic_order_cnt_lsb_minus4
        bw.ue(0);                // max_num_ref_frames
        bw.bit(0);               // gaps_in_frame_num_value_allowed_flag
        bw.ue((uint32_t)W - 1);
        bw.ue((uint32_t)H - 1);
        bw.bit(1);               // frame_mbs_only_flag
        bw.bit(1);               // direct_8x8_inference_flag
        bw.bit(0);               // frame_cropping_flag
        bw.bit(0);               // vui_parameters_present_flag
        bw.trailing();
        emit_nal(out, 3, 7, bw.bytes);
    }
    void write_pps(std::vector<uint8_t> &out)
    {
        BitWriter bw;
        bw.ue(0); bw.ue(0);
        bw.bit(cfg.cabac);
        bw.bit(0);
        bw.ue(0);                // num_slice_groups_minus1
        bw.ue(0); bw.ue(0);
        bw.bit(0); bw.bits(0, 2);
        bw.se(0); bw.se(0);
        bw.se(cfg.cqp_offset[0]);
        bw.bit(0);               // deblocking_filter_control_present_flag
        bw.bit(0);               // constrained_intra_pred_flag
        bw.bit(0);               // redundant_pic_cnt_present_flag
        if (cfg.profile_idc == 100) {
            bw.bit(cfg.transform8x8);
            if (cfg.scaling & 2) { bw.bit(1); write_list_set(bw, pps_lists, 6 + (cfg.transform8x8 ? 2 : 0)); }   // pic_scaling_matrix_present_flag
            else bw.bit(0);
            bw.se(cfg.cqp_offset[1]);
        }
        bw.trailing();
        emit_nal(out, 3, 8, bw.bytes);
    }

    void write_picture(std::vector<uint8_t> &out, int frame, uint8_t *packed)
    {
        m.init(W, H);
        const int N = W * H;
        // slice starts: macroblock 0 and n_slices - 1 further distinct addresses (the reference's envelope: one slice)
        std::vector<int> starts(1, 0);
        const int want = cfg.n_slices < N ? cfg.n_slices : N;
        while ((int)starts.size() < want) {
            const int a = 1 + rng.below(N - 1);
            bool dup = false;
            for (int v : starts) dup |= v == a;
            if (!dup) starts.push_back(a);
        }
        for (size_t i = 1; i < starts.size(); i++)
            for (size_t k = i; k > 0 && starts[k] < starts[k - 1]; k--) { const int t = starts[k]; starts[k] = starts[k - 1]; starts[k - 1] = t; }
        MbSyntax s;
        for (size_t sl = 0; sl < starts.size(); sl++) {
            const int first = starts[sl], end = sl + 1 < starts.size() ? starts[sl + 1] : N;
            m.cur_slice = (int)sl;
            const int slice_qp = cfg.qp_min + rng.below(cfg.qp_max - cfg.qp_min + 1);
            BitWriter bw;
            bw.ue((uint32_t)first);            // first_mb_in_slice
            bw.ue(7);                          // slice_type: I (all slices of the picture)
            bw.ue(0);                          // pic_parameter_set_id
            bw.bits(0, 4);                     // frame_num
            bw.ue((uint32_t)(frame & 0xffff)); // idr_pic_id
            bw.bits(0, 4);                     // pic_order_cnt_lsb
            bw.bit(0); bw.bit(0);              // no_output_of_prior_pics_flag, long_term_reference_flag
            bw.se(slice_qp - 26);
            CabacEnc enc(bw);
            if (cfg.cabac) { while (!bw.aligned()) bw.bit(1); enc.init(slice_qp); }
            int qp_prev = slice_qp;
            for (int addr = first; addr < end; addr++) {
                const int mbx = addr % W, mby = addr / W;
                m.slice[(size_t)addr] = (int)sl;
                draw_mb(mbx, mby, qp_prev, s);
                qp_prev = s.qp;
                if (packed) fill_record(s, packed + (size_t)addr * MVHP_MB_BYTES, mbx, mby);
                if (cfg.cabac) {
                    write_mb_cabac(enc, mbx, mby, s);
                    enc.terminate(addr == end - 1);
                } else {
                    write_mb_cavlc(bw, mbx, mby, s);
                }
            }
            if (!cfg.cabac) bw.trailing();
            // (CABAC: EncodeFlush wrote the stop bit (the final "1") as part of terminate(1); pad to a byte)
            else while (!bw.aligned()) bw.bit(0);
            emit_nal(out, 3, 5, bw.bytes);
        }
    }
};

} // namespace

namespace {
size_t run_generator(const GenCfg &g, uint8_t *out, size_t cap, uint8_t *packed, uint8_t *weights);
}

extern "C" {

typedef struct mvgen_cfg {
    int32_t width_mbs, height_mbs, n_frames;
    uint64_t seed;
    int32_t profile_idc, cabac, transform8x8, dense;
    int32_t cqp_offset_cb, cqp_offset_cr;
    int32_t sps_pps_every_frame, allow_qp36_i16;
    int32_t qp_min, qp_max, max_level;
} mvgen_cfg_t;

static GenCfg to_cfg(const mvgen_cfg_t *c)
{
    GenCfg g;
    g.width_mbs = c->width_mbs; g.height_mbs = c->height_mbs; g.n_frames = c->n_frames; g.seed = c->seed;
    g.profile_idc = c->profile_idc; g.cabac = c->cabac; g.transform8x8 = c->transform8x8; g.dense = c->dense;
    g.cqp_offset[0] = c->cqp_offset_cb;
    g.cqp_offset[1] = (c->profile_idc == 100) ? c->cqp_offset_cr : c->cqp_offset_cb;
    g.sps_pps_every_frame = c->sps_pps_every_frame; g.allow_qp36_i16 = c->allow_qp36_i16;
    g.qp_min = c->qp_min > 0 ? c->qp_min : 24; g.qp_max = c->qp_max >= g.qp_min ? c->qp_max : 32;
    g.max_level = c->max_level > 0 ? c->max_level : 32;
    return g;
}

// Returns the number of stream bytes (0 on bad config). Writes at most `cap` bytes to `out`
// (call with out = NULL to size). `packed` (may be NULL) receives n_frames*W*H*800 bytes of
// expected packed records.
__attribute__((visibility("default")))
size_t mvgen_stream(const mvgen_cfg_t *c, uint8_t *out, size_t cap, uint8_t *packed)
{
    if (!c || c->width_mbs <= 0 || c->height_mbs <= 0 || c->n_frames <= 0) return 0;
    if (c->profile_idc != 66 && c->profile_idc != 77 && c->profile_idc != 100) return 0;
    if (c->cabac && c->profile_idc == 66) return 0;
    if (c->transform8x8 && c->profile_idc != 100) return 0;
    GenCfg g = to_cfg(c);
    return run_generator(g, out, cap, packed, nullptr);
}

// The same outside the reference's envelope (what MVHP_STREAM_SPEC decodes): n_slices slices per picture, pcm_permille / 1000
// of the macroblocks I_PCM, scaling lists in the SPS (scaling & 1) and / or the PPS (scaling & 2; profile 100 only).
// `weights` (may be NULL) receives the 112 bytes a correct decoder must report in mvhp_stream_params_t::scaling4 / scaling8.
__attribute__((visibility("default")))
size_t mvgen_stream_ex(const mvgen_cfg_t *c, int32_t n_slices, int32_t pcm_permille, int32_t scaling, uint8_t *out, size_t cap,
                       uint8_t *packed, uint8_t *weights)
{
    if (!c || c->width_mbs <= 0 || c->height_mbs <= 0 || c->n_frames <= 0) return 0;
    if (c->profile_idc != 66 && c->profile_idc != 77 && c->profile_idc != 100) return 0;
    if (c->cabac && c->profile_idc == 66) return 0;
    if (c->transform8x8 && c->profile_idc != 100) return 0;
    if (scaling && c->profile_idc != 100) return 0;
    if (n_slices < 1 || pcm_permille < 0 || pcm_permille > 1000) return 0;
    GenCfg g = to_cfg(c);
    g.n_slices = n_slices;
    g.pcm_permille = pcm_permille;
    g.scaling = scaling;
    return run_generator(g, out, cap, packed, weights);
}

} // extern "C"

namespace {
size_t run_generator(const GenCfg &g, uint8_t *out, size_t cap, uint8_t *packed, uint8_t *weights)
{
    Gen gen(g);
    if (g.scaling & 1) gen.draw_lists(gen.sps_lists, 8);
    if (g.scaling & 2) gen.draw_lists(gen.pps_lists, 6 + (g.transform8x8 ? 2 : 0));
    if (weights) {
        uint8_t w4[3][16], w8[64];
        gen.effective_weights(w4, w8);
        memcpy(weights, w4, 48);
        memcpy(weights + 48, w8, 64);
    }
    std::vector<uint8_t> s;
    const size_t pf = (size_t)g.width_mbs * g.height_mbs * MVHP_MB_BYTES;
    for (int f = 0; f < g.n_frames; f++) {
        if (f == 0 || g.sps_pps_every_frame) { gen.write_sps(s); gen.write_pps(s); }
        gen.write_picture(s, f, packed ? packed + (size_t)f * pf : nullptr);
    }
    for (int i = 0; i < 64; i++) s.push_back(0); // esparser.c:65 stops scanning 32 bytes before EOF
    if (out) memcpy(out, s.data(), s.size() < cap ? s.size() : cap);
    return s.size();
}
} // namespace
