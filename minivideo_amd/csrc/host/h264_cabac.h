// h264_cabac.h -- CABAC parsing for I slices (9.3), restating
// decoder/h264/h264_cabac.c:138-2563 without its table-matching binarization
// decoder: the bins are decoded directly against each syntax element's
// binarization, which yields the same values on every conforming stream.
#pragma once
#include <stdint.h>

#include "bitreader.h"

namespace h264 {

// residual block categories, numbered like the reference's BlockType_e
// (h264_macroblock_struct.h:160-171)
enum {
    CAT_LUMA_8x8 = 0,
    CAT_LUMA_4x4 = 1,
    CAT_LUMA_16x16_DC = 2,
    CAT_LUMA_16x16_AC = 3,
    CAT_CHROMA_DC_CB = 4,
    CAT_CHROMA_DC_CR = 5,
    CAT_CHROMA_AC_CB = 6,
    CAT_CHROMA_AC_CR = 7,
};

class PictureDecoder;

struct CabacEngine {
    explicit CabacEngine(PictureDecoder &pd) : pd_(pd) {}
    void init(int slice_qp);                          // :529-615
    bool overrun() const;                             // the standard's 9-bit register would have read past the end
    // The arithmetic decoder of 9.3.3.2 (DecodeDecision :2380-2469, RenormD :2471, DecodeBypass :2506,
    // DecodeTerminate :2542) with codIOffset kept `k_` bits ahead of the standard's 9-bit register: val_ =
    // codIOffset << k_ | the next k_ bits of the slice data, so a comparison against codIRange << k_ is the standard's
    // comparison, renormalisation only lowers k_, and the bit reader is asked for 32 bits at a time.
    int decode_decision(int ctx)
    {
        // branch-free on the bin value (about half of all bins are not predictable): the LPS case is folded in with a mask
        const uint32_t st = st_[ctx];
        const uint32_t lps = kRangeLpsQ[st >> 1][(range_ >> 6) & 3];
        const uint32_t rmps = range_ - lps;
        const uint64_t scaled = (uint64_t)rmps << k_;
        const uint64_t is_lps = (uint64_t)0 - (uint64_t)(val_ >= scaled);   // all ones when the LPS was coded
        val_ -= scaled & is_lps;
        uint32_t range = (uint32_t)((rmps & ~is_lps) | (lps & is_lps));
        st_[ctx] = kNext[(st << 1) | (uint32_t)(is_lps & 1u)];
        const int bin = (int)((st ^ (uint32_t)is_lps) & 1u);
        const int sh = __builtin_clz(range) - 23;   // 0 while range >= 256 (range < 512 always)
        range_ = range << sh;
        k_ -= sh;
        if (k_ < 16) refill();
        return bin;
    }
    int decode_bypass()
    {
        if (--k_ < 16) refill();
        const uint64_t scaled = (uint64_t)range_ << k_;
        const uint64_t one = (uint64_t)0 - (uint64_t)(val_ >= scaled);   // sign bits are coin flips: no branch on them
        val_ -= scaled & one;
        return (int)(one & 1u);
    }
    int decode_terminate()
    {
        range_ -= 2;
        if (val_ >= ((uint64_t)range_ << k_)) return 1;
        if (range_ < 256) {
            const int sh = __builtin_clz(range_) - 23;
            range_ <<= sh;
            k_ -= sh;
            if (k_ < 16) refill();
        }
        return 0;
    }

    // I_PCM (9.3.1.2): where the standard's 9-bit register stands in the slice data (the bit reader runs k_ bits ahead of
    // it), and a fresh start of the arithmetic decoder behind the samples with the context variables kept
    size_t standard_bit_position() const;
    void   restart();

    unsigned mb_type(int addr);
    int      transform_size_8x8_flag(int addr);
    int      prev_intra_pred_mode_flag();
    int      rem_intra_pred_mode();
    unsigned intra_chroma_pred_mode(int addr);
    unsigned coded_block_pattern(int addr);
    int      mb_qp_delta(int addr);
    int      residual_block(int addr, int startIdx, int endIdx, int maxNumCoeff, int cat, int blkIdx);

private:
    int  cbf_ctx_inc(int addr, int cat, int blkIdx) const;
    void refill();
    PictureDecoder &pd_;
    uint8_t  st_[460];            // pStateIdx << 1 | valMPS
    uint32_t range_ = 510;
    uint64_t val_ = 0;
    int      k_ = 0;
    // transitions on the combined state byte, rangeTabLPS as [state][quarter]: built once from Tables 9-44 / 9-45
    static uint8_t kNext[256], kRangeLpsQ[64][4];   // kNext[state byte << 1 | LPS coded]
    static void build_tables();
public:
    static void build_tables_once();
};

} // namespace h264
