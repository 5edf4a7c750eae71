// h264_cabac.cpp -- see h264_cabac.h.
#include "h264_cabac.h"

#include <string.h>

#include "h264_frontend.h"
#include "h264_tables.h"

namespace h264 {

#include "h264_cabac_tables.inc"

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

uint8_t CabacEngine::kNext[256], CabacEngine::kRangeLpsQ[64][4];

void CabacEngine::build_tables()
{
    for (int st = 0; st < 64; st++) {
        for (int mps = 0; mps < 2; mps++) {
            kNext[(st * 2 + mps) * 2 + 0] = (uint8_t)(kTransMps[st] * 2 + mps);
            kNext[(st * 2 + mps) * 2 + 1] = (uint8_t)(kTransLps[st] * 2 + (st == 0 ? 1 - mps : mps));   // valMPS flips at state 0
        }
        for (int q = 0; q < 4; q++) kRangeLpsQ[st][q] = kRangeLps[q][st];
    }
}

void CabacEngine::build_tables_once()
{
    static const bool done = (build_tables(), true);
    (void)done;
}

// initCabacContextVariables (:529) + initCabacDecodingEngine (:581); 9.3.1.1 / 9.3.1.2
void CabacEngine::init(int slice_qp)
{
    build_tables_once();
    const int qp = clip3(0, 51, slice_qp);
    for (int i = 0; i < 460; i++) {
        const int pre = clip3(1, 126, ((kCtxInitM[i] * qp) >> 4) + kCtxInitN[i]);
        if (pre <= 63) st_[i] = (uint8_t)((63 - pre) * 2);
        else st_[i] = (uint8_t)((pre - 64) * 2 + 1);
    }
    range_ = 510;
    val_ = pd_.br_.bits(9);
    k_ = 0;
    refill();
}

size_t CabacEngine::standard_bit_position() const { return pd_.br_.pos() - (size_t)k_; }

void CabacEngine::restart()
{
    range_ = 510;
    val_ = pd_.br_.bits(9);
    k_ = 0;
    refill();
}

bool CabacEngine::overrun() const { return pd_.br_.pos() > pd_.br_.size_bits() + (size_t)k_; }

void CabacEngine::refill()
{
    val_ = (val_ << 32) | pd_.br_.bits(32);
    k_ += 32;
}

// mb_type, I slices: binarization Table 9-36, ctxIdxOffset 3; ctxIdxInc 9.3.3.1.1.3 (:1548-1596)
unsigned CabacEngine::mb_type(int addr)
{
    const int a = pd_.mbA(addr), b = pd_.mbB(addr);
    const int inc = ((a >= 0 && pd_.mbs_[a].mb_type != 0) ? 1 : 0) + ((b >= 0 && pd_.mbs_[b].mb_type != 0) ? 1 : 0);
    if (!decode_decision(3 + inc)) return 0;          // I_NxN
    if (decode_terminate()) return 25;                 // I_PCM
    const int luma = decode_decision(3 + 3);          // CodedBlockPatternLuma != 0
    int chroma = 0;
    const int b3 = decode_decision(3 + 4);
    if (b3) chroma = 1 + decode_decision(3 + 5);
    // 9.3.3.1.2: binIdx 4 uses ctxIdxInc 5 or 6 and binIdx 5 uses 6 or 7 depending on b3, i.e.
    // b3 != 0: bins are b4 (inc 5), p1 (inc 6), p0 (inc 7); b3 == 0: p1 (inc 6), p0 (inc 7).
    const int p1 = decode_decision(3 + 6);
    const int p0 = decode_decision(3 + 7);
    return 1 + (unsigned)(p1 * 2 + p0) + 4 * (unsigned)chroma + 12 * (unsigned)luma;
}

int CabacEngine::transform_size_8x8_flag(int addr)
{
    const int a = pd_.mbA(addr), b = pd_.mbB(addr);
    const int inc = ((a >= 0 && pd_.mbs_[a].transform8x8) ? 1 : 0) + ((b >= 0 && pd_.mbs_[b].transform8x8) ? 1 : 0);
    return decode_decision(399 + inc);
}

int CabacEngine::prev_intra_pred_mode_flag() { return decode_decision(68); }

int CabacEngine::rem_intra_pred_mode()
{
    int v = decode_decision(69);
    v |= decode_decision(69) << 1;
    v |= decode_decision(69) << 2;
    return v;
}

// 9.3.3.1.1.8 (:1804-1848)
unsigned CabacEngine::intra_chroma_pred_mode(int addr)
{
    const int a = pd_.mbA(addr), b = pd_.mbB(addr);
    const int inc = ((a >= 0 && pd_.mbs_[a].chroma_pred_mode != 0) ? 1 : 0) +
                    ((b >= 0 && pd_.mbs_[b].chroma_pred_mode != 0) ? 1 : 0);
    if (!decode_decision(64 + inc)) return 0;
    if (!decode_decision(64 + 3)) return 1;
    if (!decode_decision(64 + 3)) return 2;
    return 3;
}

// 9.3.2.6 + 9.3.3.1.1.4 (:1609-1748)
unsigned CabacEngine::coded_block_pattern(int addr)
{
    const int a = pd_.mbA(addr), b = pd_.mbB(addr);
    unsigned luma = 0;
    for (int b8 = 0; b8 < 4; b8++) {
        int condA, condB;
        if (b8 & 1) condA = ((luma >> (b8 - 1)) & 1) ? 0 : 1;
        else condA = (a >= 0) ? (((pd_.mbs_[a].cbp_luma >> (b8 + 1)) & 1) ? 0 : 1) : 0;
        if (b8 & 2) condB = ((luma >> (b8 - 2)) & 1) ? 0 : 1;
        else condB = (b >= 0) ? (((pd_.mbs_[b].cbp_luma >> (b8 + 2)) & 1) ? 0 : 1) : 0;
        luma |= (unsigned)decode_decision(73 + condA + 2 * condB) << b8;
    }
    unsigned chroma = 0;
    {
        const int condA = (a >= 0 && pd_.mbs_[a].cbp_chroma != 0) ? 1 : 0;
        const int condB = (b >= 0 && pd_.mbs_[b].cbp_chroma != 0) ? 1 : 0;
        if (decode_decision(77 + condA + 2 * condB)) {
            const int cA = (a >= 0 && pd_.mbs_[a].cbp_chroma == 2) ? 1 : 0;
            const int cB = (b >= 0 && pd_.mbs_[b].cbp_chroma == 2) ? 1 : 0;
            chroma = 1 + (unsigned)decode_decision(77 + 4 + cA + 2 * cB);
        }
    }
    return luma | (chroma << 4);
}

// 9.3.2.7 + 9.3.3.1.1.5 (:1760-1787)
int CabacEngine::mb_qp_delta(int addr)
{
    int inc = 0;
    if (addr > pd_.slice_first_) {   // the previous macroblock in decoding order of THIS slice
        const MbState &p = pd_.mbs_[addr - 1];
        const bool no_residual = (p.kind != MVHP_KIND_I16x16) && p.cbp_luma == 0 && p.cbp_chroma == 0;
        inc = (!no_residual && p.qp_delta_nonzero) ? 1 : 0;
    }
    if (!decode_decision(60 + inc)) return 0;
    int k = 1;
    if (decode_decision(60 + 2)) {
        k = 2;
        while (decode_decision(60 + 3)) {
            if (++k > 110) break; // malformed stream guard
        }
    }
    // mapped value k -> (-1)^(k+1) * ceil(k/2)
    return (k & 1) ? (k + 1) / 2 : -(k / 2);
}

// 9.3.3.1.1.9 (:1862-2146)
int CabacEngine::cbf_ctx_inc(int addr, int cat, int blkIdx) const
{
    const int W = pd_.W_;
    int condA, condB;
    auto mb_cond = [&](int n, int bit, bool applicable) -> int {
        if (n < 0) return 1;               // unavailable, current MB is intra
        if (pd_.mbs_[n].kind == MVHP_KIND_IPCM) return 1;   // mb_type I_PCM: condTermFlagN = 1 (spec mode only)
        if (!applicable) return 0;         // transBlockN not assigned
        return (pd_.mbs_[n].cbf >> bit) & 1;
    };
    if (cat == CAT_LUMA_16x16_DC) {
        const int a = pd_.mbA(addr), b = pd_.mbB(addr);
        condA = mb_cond(a, 16, a >= 0 && pd_.mbs_[a].kind == MVHP_KIND_I16x16);
        condB = mb_cond(b, 16, b >= 0 && pd_.mbs_[b].kind == MVHP_KIND_I16x16);
    } else if (cat == CAT_LUMA_4x4 || cat == CAT_LUMA_16x16_AC) {
        const uint8_t eA = nb_tables().lumaA[blkIdx], eB = nb_tables().lumaB[blkIdx];
        const int bA = eA & 15, bB = eB & 15;
        const int a = (eA & 0x80) ? pd_.mbA(addr) : addr, b = (eB & 0x80) ? pd_.mbB(addr) : addr;
        (void)W;
        condA = mb_cond(a, bA, a >= 0 && ((pd_.mbs_[a].cbp_luma >> (bA >> 2)) & 1));
        condB = mb_cond(b, bB, b >= 0 && ((pd_.mbs_[b].cbp_luma >> (bB >> 2)) & 1));
    } else if (cat == CAT_CHROMA_DC_CB || cat == CAT_CHROMA_DC_CR) {
        const int a = pd_.mbA(addr), b = pd_.mbB(addr), bit = 25 + (cat - CAT_CHROMA_DC_CB);
        condA = mb_cond(a, bit, a >= 0 && pd_.mbs_[a].cbp_chroma != 0);
        condB = mb_cond(b, bit, b >= 0 && pd_.mbs_[b].cbp_chroma != 0);
    } else { // chroma AC
        const int c = cat - CAT_CHROMA_AC_CB;
        int a, b, bA, bB;
        if (blkIdx & 1) { a = addr; bA = blkIdx - 1; } else { a = pd_.mbA(addr); bA = blkIdx + 1; }
        if (blkIdx & 2) { b = addr; bB = blkIdx - 2; } else { b = pd_.mbB(addr); bB = blkIdx + 2; }
        condA = mb_cond(a, 17 + c * 4 + bA, a >= 0 && pd_.mbs_[a].cbp_chroma == 2);
        condB = mb_cond(b, 17 + c * 4 + bB, b >= 0 && pd_.mbs_[b].cbp_chroma == 2);
    }
    return condA + 2 * condB;
}

// residual_block_cabac, :138-325; ctxIdx assignment 9.3.3.1.3 / Table 9-40 (:2266-2340)
int CabacEngine::residual_block(int addr, int startIdx, int endIdx, int maxNumCoeff, int cat, int blkIdx)
{
    static const int kCbfOff[8] = {0, 8, 0, 4, 12, 12, 16, 16};
    static const int kSigOff[8] = {0, 29, 0, 15, 44, 44, 47, 47};
    static const int kAbsOff[8] = {0, 20, 0, 10, 30, 30, 39, 39};
    MbState &mb = pd_.mbs_[addr];
    const bool is8 = (cat == CAT_LUMA_8x8);
    const bool cdc = (cat == CAT_CHROMA_DC_CB || cat == CAT_CHROMA_DC_CR);
    int cbf = 1;
    if (!is8) { // maxNumCoeff != 64 || ChromaArrayType == 3 (:171)
        cbf = decode_decision(85 + kCbfOff[cat] + cbf_ctx_inc(addr, cat, blkIdx));
    }
    // record for neighbours
    {
        int bit;
        if (cat == CAT_LUMA_16x16_DC) bit = 16;
        else if (cat == CAT_CHROMA_DC_CB) bit = 25;
        else if (cat == CAT_CHROMA_DC_CR) bit = 26;
        else if (cat == CAT_CHROMA_AC_CB) bit = 17 + blkIdx;
        else if (cat == CAT_CHROMA_AC_CR) bit = 21 + blkIdx;
        else bit = blkIdx;
        if (is8) { if (cbf) mb.cbf |= 0xfu << (4 * blkIdx); }
        else if (cbf) mb.cbf |= 1u << bit;
    }
    if (!cbf) return RC_SUCCESS;

    const int sig_base = is8 ? 402 : 105 + kSigOff[cat];
    const int last_base = is8 ? 417 : 166 + kSigOff[cat];
    const int abs_base = is8 ? 426 : 227 + kAbsOff[cat];
    // significance map as a bit mask (bit i = coefficient i is significant), levels in reverse scan order (:236-316)
    uint64_t sigmask = 0;
    int numCoeff = endIdx + 1;
    if (numCoeff > 64) return RC_FAILURE;
    int i = startIdx;
    if (is8) {
        while (i < numCoeff - 1) {
            if (decode_decision(sig_base + kSigInc8x8[i])) {
                sigmask |= 1ull << i;
                if (decode_decision(last_base + kLastInc8x8[i])) numCoeff = i + 1;
            }
            i++;
        }
    } else if (cdc) {
        while (i < numCoeff - 1) {
            const int inc = i < 2 ? i : 2;
            if (decode_decision(sig_base + inc)) {
                sigmask |= 1ull << i;
                if (decode_decision(last_base + inc)) numCoeff = i + 1;
            }
            i++;
        }
    } else {
        while (i < numCoeff - 1) {
            if (decode_decision(sig_base + i)) {
                sigmask |= 1ull << i;
                if (decode_decision(last_base + i)) numCoeff = i + 1;
            }
            i++;
        }
    }
    sigmask |= 1ull << (numCoeff - 1);
    int eq1 = 0, gt1 = 0;
    const int lim = 4 - (cdc ? 1 : 0);
    while (sigmask) {
        i = 63 - __builtin_clzll(sigmask);
        sigmask &= ~(1ull << i);
        // coeff_abs_level_minus1: UEG0, signedValFlag=0, uCoff=14 (9.3.2.3)
        const int inc0 = (gt1 != 0) ? 0 : ((1 + eq1) < 4 ? (1 + eq1) : 4);
        int v = 0;
        if (decode_decision(abs_base + inc0)) {
            const int incn = 5 + (gt1 < lim ? gt1 : lim);
            v = 1;
            while (v < 14 && decode_decision(abs_base + incn)) v++;
            if (v == 14) { // suffix: Exp-Golomb k = 0, bypass
                int k = 0;
                while (decode_bypass()) {
                    v += 1 << k;
                    if (++k > 24) return RC_FAILURE;
                }
                while (k--) v += decode_bypass() << k;
            }
        }
        const int sign = decode_bypass();
        const int lvl = v + 1;
        if (lvl == 1) eq1++; else gt1++;
        pd_.put(i, sign ? -lvl : lvl);
    }
    (void)maxNumCoeff;
    return RC_SUCCESS;
}

} // namespace h264
