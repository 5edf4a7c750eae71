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
    int  decode_decision(int ctxIdx);                 // :2380-2469
    int  decode_bypass();                             // :2506
    int  decode_terminate();                          // :2542

    unsigned mb_type(int addr);
    int      transform_size_8x8_flag(int addr);
    int      prev_intra_pred_mode_flag();
    int      rem_intra_pred_mode();
    unsigned intra_chroma_pred_mode(int addr);
    unsigned coded_block_pattern(int addr);
    int      mb_qp_delta(int addr);
    int      residual_block(int addr, int *coeff, int startIdx, int endIdx, int maxNumCoeff, int cat, int blkIdx);

private:
    int  cbf_ctx_inc(int addr, int cat, int blkIdx) const;
    PictureDecoder &pd_;
    uint8_t  state_[460];
    uint8_t  mps_[460];
    uint32_t range_ = 510, offset_ = 0;
};

} // namespace h264
