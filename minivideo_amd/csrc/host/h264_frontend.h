// h264_frontend.h -- host side of the split decoder: Annex-B index, NAL
// unescaping, SPS/PPS/slice-header parsing, macroblock layer, CAVLC/CABAC
// residual decoding and Intra NxN prediction-mode derivation.  Output: one
// packed macroblock record per macroblock (include/minivideo_hotpath.h).
//
// Reference functions restated here (minivideo/src/...):
//   demuxer/esparser/esparser.c:40-143          -> index_annexb()
//   decoder/h264/h264_nalu.c:109-249            -> Nal header, unescape_rbsp()
//   decoder/h264/h264_parameterset.c:123-397    -> parse_sps()
//   decoder/h264/h264_parameterset.c:812-942    -> parse_pps()
//   decoder/h264/h264_slice.c:156-334,1013-1142 -> PictureDecoder::slice_header/slice_data
//   decoder/h264/h264_macroblock.c:75-313,393-518,766-842,1102-1295 -> PictureDecoder::macroblock
//   decoder/h264/h264_cavlc.c:79-346            -> PictureDecoder::residual_block_cavlc
//   decoder/h264/h264_cabac.c:138-2563          -> h264_cabac.cpp
//   decoder/h264/h264_intra_prediction.c:196-290,977-1083 -> derive_pred_modes
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "bitreader.h"
#include "minivideo_hotpath.h"

namespace h264 {

enum { RC_UNSUPPORTED = -1, RC_FAILURE = 0, RC_SUCCESS = 1 }; // typedef.h:40-42

// scaling_list() data of an SPS or a PPS (7.3.2.1.1.1), as transmitted: list i < 6 is 4x4 (Intra Y, Cb, Cr, Inter Y, Cb, Cr),
// list 6 / 7 is 8x8 (Intra Y / Inter Y), each in zig-zag order.  state: 0 = not present (fall-back rule applies), 1 =
// transmitted, 2 = transmitted as "use the default list" (useDefaultScalingMatrixFlag).
struct ScalingLists {
    bool    present = false;            // seq_ / pic_scaling_matrix_present_flag
    uint8_t state[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint8_t l4[6][16] = {{0}};
    uint8_t l8[2][64] = {{0}};
};
// The weight matrices an Intra picture uses (Intra Y / Cb / Cr 4x4 and Intra Y 8x8), in RASTER order, after the fall-back
// rules of 7.4.2.1.1 (set A, SPS) and 7.4.2.2 (set A or B, PPS).  Returns true when any weight differs from 16.
bool effective_intra_scaling(const ScalingLists &sps, const ScalingLists &pps, bool transform8x8, uint8_t w4[3][16], uint8_t w8[64]);

struct Sps {
    bool     valid = false;
    int      profile_idc = 0, level_idc = 0, sps_id = 0;
    int      chroma_format_idc = 1;
    int      log2_max_frame_num = 4;
    int      poc_type = 0, log2_max_poc_lsb = 4;
    bool     delta_pic_order_always_zero = false;
    int      width_mbs = 0, height_map_units = 0;
    bool     frame_mbs_only = true;
    bool     direct_8x8_inference = false;
    bool     frame_cropping = false;
    int      crop[4] = {0, 0, 0, 0};
    bool     qpprime_y_zero_transform_bypass = false;
    ScalingLists scaling;   // seq_scaling_matrix (only parsed for MVHP_STREAM_SPEC streams; the reference envelope has none)
};

struct Pps {
    bool valid = false;
    int  pps_id = 0, sps_id = 0;
    bool entropy_coding_mode = false;
    bool bottom_field_pic_order_in_frame_present = false;
    int  num_slice_groups_minus1 = 0;
    bool weighted_pred = false;
    int  weighted_bipred_idc = 0;
    int  pic_init_qp_minus26 = 0, pic_init_qs_minus26 = 0;
    int  chroma_qp_index_offset = 0, second_chroma_qp_index_offset = 0;
    bool deblocking_filter_control_present = false;
    bool constrained_intra_pred = false;
    bool redundant_pic_cnt_present = false;
    bool transform_8x8_mode = false;
    ScalingLists scaling;   // pic_scaling_matrix (MVHP_STREAM_SPEC streams only)
};

// One entry of the elementary-stream sample table (bitstream_map_struct.h:46-129,
// as filled by esparser.c:82-124): offset of the NAL header byte and the
// distance to the next indexed NAL header (the reference's sample_size, which
// includes the next start code), plus the true NAL payload end.
struct EsSample {
    size_t offset = 0;       // NAL header byte
    size_t sample_size = 0;  // reference semantics (to next indexed NAL header byte / EOF)
    size_t nal_size = 0;     // header + payload, start code and trailing zero bytes trimmed
    int    nal_unit_type = 0;
    int    nal_ref_idc = 0;
    bool   is_idr = false;
};

int  index_annexb(const uint8_t *data, size_t size, std::vector<EsSample> &out);
// Annex B as the standard defines it (MVHP_STREAM_SPEC): 3- or 4-byte start codes, any nal_ref_idc, scan to the end
int  index_annexb_spec(const uint8_t *data, size_t size, std::vector<EsSample> &out);
void unescape_rbsp(const uint8_t *src, size_t n, std::vector<uint8_t> &dst);
// `spec` (MVHP_STREAM_SPEC): scaling lists are parsed (7.3.2.1.1.1) instead of refused
int  parse_sps(BitReader &br, Sps &sps, std::string &err, bool spec = false);
int  parse_pps(BitReader &br, const Sps *sps_table /*[32]*/, Pps &pps, std::string &err, bool spec = false);

// Per-macroblock state kept for neighbour derivations (nC, ctxIdxInc, pred modes).
struct MbState {
    uint8_t  kind = 0;          // MVHP_KIND_*
    uint8_t  mb_type = 0;       // raw I-slice mb_type 0..25
    uint8_t  cbp_luma = 0, cbp_chroma = 0;
    uint8_t  chroma_pred_mode = 0;
    uint8_t  qp_delta_nonzero = 0;
    uint8_t  transform8x8 = 0;
    uint8_t  pred[16] = {0};    // final Intra4x4PredMode[16] / Intra8x8PredMode[4]
    uint8_t  tc_luma[16] = {0}, tc_c[2][4] = {{0}}; // CAVLC TotalCoeff
    uint32_t cbf = 0;           // CABAC coded_block_flag: bits 0-15 luma, 16 luma DC,
                                // 17-20 Cb AC, 21-24 Cr AC, 25 Cb DC, 26 Cr DC
};

struct CabacEngine;

// one slice NAL of a picture: payload after the NAL header byte, emulation prevention removed
struct SliceRbsp {
    const uint8_t *rbsp = nullptr;
    size_t         n = 0;
    int            nal_ref_idc = 0;
};

class PictureDecoder {
public:
    // spec = MVHP_STREAM_SPEC: the standard's slice semantics (first_mb_in_slice honoured, a slice ends where its data ends,
    // I_PCM accepted); otherwise the reference's (one slice NAL = one picture decoded from macroblock 0, h264_slice.c:1019)
    PictureDecoder(const Sps &sps, const Pps &pps, int nal_ref_idc, bool spec = false);
    ~PictureDecoder();
    // rbsp: slice NAL payload (after the NAL header byte), emulation prevention removed.
    int decode(const uint8_t *rbsp, size_t n, uint8_t *packed, size_t packed_bytes, std::string &err);
    // The same picture in the COMPACT transfer format (include/minivideo_hotpath.h, "compact pictures"): the levels that
    // are zero -- most of an 800-byte record -- never cross the PCIe link; the GPU expands it into packed records.
    int decode_compact(const uint8_t *rbsp, size_t n, uint8_t *buf, size_t cap, size_t *used, std::string &err);
    // A picture of several slices (spec mode): the slices in macroblock order, together covering the picture.
    int decode_slices(const SliceRbsp *slices, int n_slices, uint8_t *packed, size_t packed_bytes, std::string &err);
    int decode_slices_compact(const SliceRbsp *slices, int n_slices, uint8_t *buf, size_t cap, size_t *used, std::string &err);

private:
    friend struct CabacEngine;
    int  slice_header(std::string &err);
    int  slice_data(std::string &err);
    int  macroblock(int addr, std::string &err);
    void derive_pred_modes(int addr, const uint8_t prev_flag[16], const uint8_t rem[16]);
    int  residual(int addr, std::string &err);
    int  residual_block_cavlc(int addr, int startIdx, int endIdx, int maxNumCoeff, int cat, int blkIdx);
    int  nC_for(int addr, int cat, int blkIdx) const;
    // destination of the block being decoded: coefficient index -> int16 slot of the packed record
    struct Sink {
        int            base = 0;              // first int16 slot (0..383) of the block inside the coefficient area
        const uint8_t *tab = nullptr;
        int            scale = 1;
        uint32_t       nz_bit = 0;
        bool           nz_per_coef = false;   // DC blocks: level i marks block tab[i] (its slot 0 holds the level)
    };
    void sink_begin(int addr, int cat, int blkIdx, int part);
    void put(int idx, int v)
    {
        if (v > 32767 || v < -32768) { level_overflow_ = true; v = 0; }
        const int slot = sink_.tab[idx];
        const int pos = sink_.base + slot * sink_.scale;
        if (compact_) {   // one entry per level, in the order the entropy decoder delivers them
            *cwl_++ = ((uint32_t)(uint16_t)v << 16) | (uint32_t)pos;
        } else {
            coef_[pos] = (int16_t)v;
        }
        nz_cur_ |= sink_.nz_per_coef ? (sink_.nz_bit << slot) : sink_.nz_bit;
    }

    // neighbour helpers: address of MB A/B or -1 (cached for the macroblock being parsed: no division per call); a
    // macroblock of an earlier slice is not available (6.4.8; slice_first_ = 0 for the reference's one-slice pictures)
    int mbA(int addr) const { return addr == cur_addr_ ? curA_ : (((addr % W_) > 0 && addr - 1 >= slice_first_) ? addr - 1 : -1); }
    int mbB(int addr) const { return addr - W_ >= slice_first_ ? addr - W_ : -1; }

    const Sps &sps_;
    const Pps &pps_;
    int        nal_ref_idc_;
    int        W_, H_;
    BitReader  br_;
    std::vector<MbState> mbs_;
    uint8_t   *out_ = nullptr;
    int        slice_qp_ = 26, qp_prev_ = 26;
    CabacEngine *cabac_ = nullptr;
    bool       level_overflow_ = false;
    Sink       sink_;
    int16_t   *coef_ = nullptr;  // dense output: coefficient area of the macroblock being parsed
    bool       compact_ = false; // compact output: the writer (cw_ = record of the macroblock being parsed, cwl_ = its next entry)
    uint8_t   *cw_ = nullptr, *cw_base_ = nullptr;
    uint32_t  *cwl_ = nullptr;
    uint32_t   compact_max_ = MVHP_COMPACT_MAX_ENTRIES;
    uint32_t  *mb_off_ = nullptr;
    int  run(const SliceRbsp *slices, int n_slices, std::string &err);
    int  pcm_samples(int addr, std::string &err);
    uint8_t unavail_bits(int addr) const;
    bool       spec_ = false;
    int        slice_first_ = 0;     // address of the first macroblock of the slice being parsed
    int        next_addr_ = 0;       // macroblocks decoded so far (= the address the next slice must start at)
    bool       multi_slice_ = false; // records carry MVHP_UNAVAIL_* bits
    int        cur_addr_ = -1, curA_ = -1, cur_x_ = 0;   // the macroblock being parsed, its left neighbour, its column
    uint32_t   nz_cur_ = 0;      // nz_mask of the macroblock being decoded
};

} // namespace h264
