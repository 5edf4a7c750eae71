// h264_frontend.cpp -- see h264_frontend.h for the reference functions restated here.
#include "h264_frontend.h"

#include <stdlib.h>
#include <string.h>

#include "h264_cabac.h"
#include "h264_tables.h"

namespace h264 {

// ---------------------------------------------------------------------------
// H1: Annex-B elementary-stream index (esparser.c:40-143)
// ---------------------------------------------------------------------------
int index_annexb(const uint8_t *data, size_t size, std::vector<EsSample> &out)
{
    // What the reference's byte-at-a-time scan accepts (esparser.c:40-143), found by hopping from one 0x01 byte to the next
    // (memchr: a 1080p stream has one every ~256 bytes; the byte loop this replaces took 0.2 s on a 200-MB file, twice per
    // mini_thumbnailer run):
    //   * a sample starts behind 00 00 00 01 (at least three zero bytes, esparser.c:78) when the next byte is 0x65 / 0x67 /
    //     0x68 (esparser.c:82) and the 0x01 lies more than 32 bytes before the end of the file (esparser.c:65);
    //   * its NAL unit ends at the first 00 00 01 behind its header byte -- anywhere up to the end of the file -- with the
    //     zero bytes in front of that trimmed; a sample runs to the next sample's first byte.
    out.clear();
    if (!data) return RC_FAILURE;
    const int64_t limit = (int64_t)size - 32;
    bool open = false;   // the last sample of `out` has no end yet
    auto close_at = [&](size_t end) {
        const size_t beg = out.back().offset;
        while (end > beg + 1 && data[end - 1] == 0) end--;
        out.back().nal_size = end - beg;
        open = false;
    };
    size_t from = 0;
    while (from < size) {
        const uint8_t *hit = static_cast<const uint8_t *>(memchr(data + from, 0x01, size - from));
        if (!hit) break;
        const size_t p = (size_t)(hit - data);
        from = p + 1;
        if (p < 2 || data[p - 1] != 0 || data[p - 2] != 0) continue;
        if (open && p - 2 >= out.back().offset + 1) close_at(p - 2);
        if (p >= 3 && data[p - 3] == 0 && (int64_t)p < limit) {
            const uint8_t nb = data[p + 1];
            if (nb == 0x65 || nb == 0x67 || nb == 0x68) {
                EsSample s;
                s.offset = p + 1;
                s.nal_unit_type = nb & 31;
                s.nal_ref_idc = (nb >> 5) & 3;
                s.is_idr = (nb == 0x65);
                if (!out.empty()) out.back().sample_size = s.offset - out.back().offset;
                out.push_back(s);
                open = true;
            }
        }
    }
    if (out.empty()) return RC_FAILURE;
    out.back().sample_size = size - out.back().offset;
    if (open) close_at(size);
    return RC_SUCCESS;
}

// The standard's byte stream format (Annex B.1): a NAL unit starts behind 00 00 01 (any number of zero bytes in front)
// and ends at the next 00 00 00 / 00 00 01 or at the end of the stream; slice (5), SPS (7) and PPS (8) NAL units are
// kept whatever their nal_ref_idc.  SURVEY 8f row f4: outside the parity contract -- the reference needs four-byte start
// codes, nal_ref_idc = 3 and 32 bytes behind the last NAL unit (esparser.c:65-82).
int index_annexb_spec(const uint8_t *data, size_t size, std::vector<EsSample> &out)
{
    out.clear();
    if (!data) return RC_FAILURE;
    size_t i = 0;
    while (i + 3 < size) {
        if (!(data[i] == 0 && data[i + 1] == 0 && data[i + 2] == 1)) { i++; continue; }
        const size_t beg = i + 3;
        size_t end = size;
        for (size_t p = beg; p + 2 < size; p++)
            if (data[p] == 0 && data[p + 1] == 0 && data[p + 2] <= 1) { end = p; break; }
        const uint8_t nb = data[beg];
        const int type = nb & 31;
        if (!(nb & 0x80) && (type == 5 || type == 7 || type == 8) && end > beg) {
            EsSample s;
            s.offset = beg;
            s.nal_unit_type = type;
            s.nal_ref_idc = (nb >> 5) & 3;
            s.is_idr = (type == 5);
            size_t e = end;
            while (e > beg + 1 && data[e - 1] == 0) e--;   // trailing_zero_8bits
            s.nal_size = e - beg;
            if (!out.empty()) out.back().sample_size = s.offset - out.back().offset;
            out.push_back(s);
        }
        i = end > beg ? end : beg;
    }
    if (out.empty()) return RC_FAILURE;
    out.back().sample_size = size - out.back().offset;
    return RC_SUCCESS;
}

// H4: emulation prevention removal (h264_nalu.c:195-249)
void unescape_rbsp(const uint8_t *src, size_t n, std::vector<uint8_t> &dst)
{
    // an emulation prevention byte is a 0x03 behind two zero bytes of the input (a removed byte resets the zero run,
    // and it is not zero itself, so looking back two input bytes is the same test as counting zeros while copying);
    // everything between two of them is copied in one piece
    dst.resize(n);
    uint8_t *d = dst.data();
    size_t from = 0, o = 0, i = 0;
    while (i < n) {
        const uint8_t *hit = static_cast<const uint8_t *>(memchr(src + i, 0x03, n - i));
        if (!hit) break;
        i = (size_t)(hit - src);
        if (i >= 2 && src[i - 1] == 0 && src[i - 2] == 0) {
            memcpy(d + o, src + from, i - from);
            o += i - from;
            from = i + 1;
        }
        i++;
    }
    memcpy(d + o, src + from, n - from);
    o += n - from;
    dst.resize(o);
}

// ---------------------------------------------------------------------------
// H5: parameter sets
// ---------------------------------------------------------------------------
static bool is_frext_profile(int p)
{
    return p == 100 || p == 110 || p == 122 || p == 244 || p == 44 || p == 83 || p == 86 || p == 118 || p == 128;
}

// scaling_list(), 7.3.2.1.1.1 (the reference's scaling_list_4x4 / _8x8, h264_parameterset.c:723-775, parse the same syntax
// but store into sps_array[0] and apply no fall-back rule)
// false: delta_scale outside -128..127 (7.4.2.1.1.1; untrusted input -- a wild se(v) would overflow the sum below) or the
// list runs off the end of the parameter set
static bool parse_scaling_list(BitReader &br, uint8_t *list, int size, uint8_t *state)
{
    int last = 8, next = 8;
    bool use_default = false;
    for (int j = 0; j < size; j++) {
        if (next != 0) {
            const int delta = br.se();
            if (delta < -128 || delta > 127 || br.overrun()) return false;
            next = (last + delta + 256) % 256;
            use_default = (j == 0 && next == 0);
        }
        list[j] = (uint8_t)(next == 0 ? last : next);
        last = list[j];
    }
    *state = use_default ? 2 : 1;
    return true;
}

static bool parse_scaling_matrix(BitReader &br, ScalingLists &sl, int n_lists)
{
    sl.present = true;
    for (int i = 0; i < n_lists; i++) {
        sl.state[i] = 0;
        if (!br.bit()) continue;   // seq_ / pic_scaling_list_present_flag[i]
        if (!(i < 6 ? parse_scaling_list(br, sl.l4[i], 16, &sl.state[i]) : parse_scaling_list(br, sl.l8[i - 6], 64, &sl.state[i]))) return false;
    }
    return true;
}

// Tables 7-3 / 7-4 (zig-zag order, like the transmitted lists)
static const uint8_t kDefault4x4Intra[16] = {6, 13, 13, 20, 20, 20, 28, 28, 28, 28, 32, 32, 32, 37, 37, 42};
static const uint8_t kDefault4x4Inter[16] = {10, 14, 14, 20, 20, 20, 24, 24, 24, 24, 27, 27, 27, 30, 30, 34};
static const uint8_t kDefault8x8Intra[64] = {6,  10, 10, 13, 11, 13, 16, 16, 16, 16, 18, 18, 18, 18, 18, 23, 23, 23, 23, 23, 23, 25,
                                             25, 25, 25, 25, 25, 25, 27, 27, 27, 27, 27, 27, 27, 27, 29, 29, 29, 29, 29, 29, 29, 31,
                                             31, 31, 31, 31, 31, 33, 33, 33, 33, 33, 36, 36, 36, 36, 38, 38, 38, 40, 40, 42};
static const uint8_t kDefault8x8Inter[64] = {9,  13, 13, 15, 13, 15, 17, 17, 17, 17, 19, 19, 19, 19, 19, 21, 21, 21, 21, 21, 21, 22,
                                             22, 22, 22, 22, 22, 22, 24, 24, 24, 24, 24, 24, 24, 24, 25, 25, 25, 25, 25, 25, 25, 27,
                                             27, 27, 27, 27, 27, 28, 28, 28, 28, 28, 30, 30, 30, 30, 32, 32, 32, 33, 33, 35};

// One level of the hierarchy: the eight lists (zig-zag) that result from `sl` with fall-back rule set A (`base` = nullptr: the
// defaults of Table 7-2) or set B (`base` = the sequence-level lists).
static void resolve_lists(const ScalingLists &sl, const uint8_t (*base4)[16], const uint8_t (*base8)[64], uint8_t out4[6][16],
                          uint8_t out8[2][64])
{
    for (int i = 0; i < 6; i++) {
        const uint8_t *def = (i < 3) ? kDefault4x4Intra : kDefault4x4Inter;
        const uint8_t *src;
        if (sl.state[i] == 1) src = sl.l4[i];
        else if (sl.state[i] == 2) src = def;
        else if (i == 0 || i == 3) src = base4 ? base4[i] : def;   // rule A: the default; rule B: the sequence-level list
        else src = out4[i - 1];                                     // both rules: the previous list of the same level
        memcpy(out4[i], src, 16);
    }
    for (int i = 0; i < 2; i++) {
        const uint8_t *def = i ? kDefault8x8Inter : kDefault8x8Intra;
        const uint8_t *src;
        if (sl.state[6 + i] == 1) src = sl.l8[i];
        else if (sl.state[6 + i] == 2) src = def;
        else src = base8 ? base8[i] : def;
        memcpy(out8[i], src, 64);
    }
}

bool effective_intra_scaling(const ScalingLists &sps, const ScalingLists &pps, bool transform8x8, uint8_t w4[3][16], uint8_t w8[64])
{
    uint8_t s4[6][16], s8[2][64], p4[6][16], p8[2][64];
    if (sps.present) resolve_lists(sps, nullptr, nullptr, s4, s8);
    else { memset(s4, 16, sizeof(s4)); memset(s8, 16, sizeof(s8)); }   // Flat_4x4_16 / Flat_8x8_16
    if (pps.present) {
        ScalingLists pl = pps;
        if (!transform8x8) pl.state[6] = pl.state[7] = 0;   // the 8x8 lists are only transmitted with transform_8x8_mode_flag
        // 7.4.2.2: set A when the SPS carries no matrix, else set B
        if (sps.present) resolve_lists(pl, s4, s8, p4, p8);
        else resolve_lists(pl, nullptr, nullptr, p4, p8);
    } else {
        memcpy(p4, s4, sizeof(p4));
        memcpy(p8, s8, sizeof(p8));
    }
    bool nonflat = false;
    for (int pl = 0; pl < 3; pl++)
        for (int k = 0; k < 16; k++) {
            w4[pl][kZigzag4x4[k]] = p4[pl][k];   // zig-zag position k -> raster slot (frame scan, Table 8-13 / utils.h:64)
            nonflat |= p4[pl][k] != 16;
        }
    for (int k = 0; k < 64; k++) {
        w8[kZigzag8x8[k]] = p8[0][k];
        nonflat |= p8[0][k] != 16;
    }
    return nonflat;
}

int parse_sps(BitReader &br, Sps &s, std::string &err, bool spec)
{
    s = Sps();
    s.profile_idc = (int)br.bits(8);
    br.bits(6);                                   // constraint_set0..5
    if (br.bits(2) != 0) { err = "SPS: reserved_zero_2bits != 0"; return RC_FAILURE; }
    s.level_idc = (int)br.bits(8);
    s.sps_id = (int)br.ue();
    if (s.sps_id > 31) { err = "SPS: seq_parameter_set_id out of range"; return RC_FAILURE; }
    if (is_frext_profile(s.profile_idc)) {
        s.chroma_format_idc = (int)br.ue();
        if (s.chroma_format_idc != 1) { err = "SPS: only 4:2:0 is supported"; return RC_UNSUPPORTED; } // :175-199
        const unsigned bdl = br.ue(), bdc = br.ue();
        if (bdl != 0 || bdc != 0) { err = "SPS: only 8-bit samples are supported"; return RC_UNSUPPORTED; }
        s.qpprime_y_zero_transform_bypass = br.bit();
        if (br.bit()) {   // seq_scaling_matrix_present_flag
            // reference envelope: its lists have no fall-back rule and land in sps_array[0] -- refused; by the
            // standard (MVHP_STREAM_SPEC): parsed, h264_parameterset.c:723-736 is the syntax
            if (!spec) { err = "SPS: scaling matrices are not supported"; return RC_UNSUPPORTED; }
            if (!parse_scaling_matrix(br, s.scaling, 8)) { err = "SPS: malformed scaling list"; return RC_FAILURE; }
        }
    }
    // h264_parameterset.c:458-465: only Baseline(66), Main(77), High(100)
    if (s.profile_idc != 66 && s.profile_idc != 77 && s.profile_idc != 100) {
        err = "SPS: unsupported profile_idc";
        return RC_UNSUPPORTED;
    }
    s.log2_max_frame_num = (int)br.ue() + 4;
    if (s.log2_max_frame_num > 16) { err = "SPS: log2_max_frame_num_minus4 out of range"; return RC_FAILURE; }
    s.poc_type = (int)br.ue();
    if (s.poc_type == 0) {
        s.log2_max_poc_lsb = (int)br.ue() + 4;
        if (s.log2_max_poc_lsb > 16) { err = "SPS: log2_max_pic_order_cnt_lsb_minus4 out of range"; return RC_FAILURE; }
    } else if (s.poc_type == 1) {
        s.delta_pic_order_always_zero = br.bit();
        br.se();
        br.se();
        const unsigned n = br.ue();
        if (n > 255) { err = "SPS: num_ref_frames_in_pic_order_cnt_cycle out of range"; return RC_FAILURE; }
        for (unsigned i = 0; i < n; i++) br.se();
    } else if (s.poc_type > 2) {
        err = "SPS: pic_order_cnt_type out of range";
        return RC_FAILURE;
    }
    br.ue();  // max_num_ref_frames
    br.bit(); // gaps_in_frame_num_value_allowed_flag
    s.width_mbs = (int)br.ue() + 1;
    s.height_map_units = (int)br.ue() + 1;
    s.frame_mbs_only = br.bit();
    if (!s.frame_mbs_only) { err = "SPS: interlaced streams are not supported"; return RC_UNSUPPORTED; }
    s.direct_8x8_inference = br.bit();
    s.frame_cropping = br.bit();
    if (s.frame_cropping)
        for (int i = 0; i < 4; i++) s.crop[i] = (int)br.ue(); // parsed, never applied (export.c:80-81)
    br.bit(); // vui_parameters_present_flag: VUI is parse-and-ignore in the reference; nothing after it matters
    if (br.overrun()) { err = "SPS: truncated"; return RC_FAILURE; }
    if (s.width_mbs <= 0 || s.height_map_units <= 0 || s.width_mbs > 1024 || s.height_map_units > 1024) {
        err = "SPS: picture size out of range";
        return RC_FAILURE;
    }
    s.valid = true;
    return RC_SUCCESS;
}

int parse_pps(BitReader &br, const Sps *sps_table, Pps &p, std::string &err, bool spec)
{
    p = Pps();
    p.pps_id = (int)br.ue();
    p.sps_id = (int)br.ue();
    if (p.pps_id > 255 || p.sps_id > 31) { err = "PPS: id out of range"; return RC_FAILURE; }
    p.entropy_coding_mode = br.bit();
    p.bottom_field_pic_order_in_frame_present = br.bit();
    p.num_slice_groups_minus1 = (int)br.ue();
    if (p.num_slice_groups_minus1 > 0) { err = "PPS: slice groups (FMO) are not supported"; return RC_UNSUPPORTED; }
    br.ue(); // num_ref_idx_l0_default_active_minus1
    br.ue(); // num_ref_idx_l1_default_active_minus1
    p.weighted_pred = br.bit();
    p.weighted_bipred_idc = (int)br.bits(2);
    p.pic_init_qp_minus26 = br.se();
    p.pic_init_qs_minus26 = br.se();
    p.chroma_qp_index_offset = br.se();
    p.deblocking_filter_control_present = br.bit();
    p.constrained_intra_pred = br.bit();
    p.redundant_pic_cnt_present = br.bit();
    const Sps &s = sps_table[p.sps_id];
    if (!s.valid) { err = "PPS: refers to an SPS that was not received"; return RC_FAILURE; }
    // h264_parameterset.c:898: extension read when more_rbsp_data() && profile_idc >= 100
    if (br.more_rbsp_data() && s.profile_idc >= 100) {
        p.transform_8x8_mode = br.bit();
        if (br.bit()) {   // pic_scaling_matrix_present_flag (:904-923: the reference answers UNSUPPORTED)
            if (!spec) { err = "PPS: scaling matrices are not supported"; return RC_UNSUPPORTED; }
            if (!parse_scaling_matrix(br, p.scaling, 6 + (p.transform_8x8_mode ? 2 : 0))) { err = "PPS: malformed scaling list"; return RC_FAILURE; }
        }
        p.second_chroma_qp_index_offset = br.se();
    } else {
        p.second_chroma_qp_index_offset = p.chroma_qp_index_offset;
    }
    if (br.overrun()) { err = "PPS: truncated"; return RC_FAILURE; }
    if (p.chroma_qp_index_offset < -12 || p.chroma_qp_index_offset > 12 || p.second_chroma_qp_index_offset < -12 ||
        p.second_chroma_qp_index_offset > 12 || p.pic_init_qp_minus26 < -26 || p.pic_init_qp_minus26 > 25) {
        err = "PPS: QP parameters out of range";
        return RC_FAILURE;
    }
    p.valid = true;
    return RC_SUCCESS;
}

// ---------------------------------------------------------------------------
// picture decoder
// ---------------------------------------------------------------------------
PictureDecoder::PictureDecoder(const Sps &sps, const Pps &pps, int nal_ref_idc, bool spec)
    : sps_(sps), pps_(pps), nal_ref_idc_(nal_ref_idc), W_(sps.width_mbs), H_(sps.height_map_units), spec_(spec)
{
}

PictureDecoder::~PictureDecoder() { delete cabac_; }

int PictureDecoder::run(const SliceRbsp *slices, int n_slices, std::string &err)
{
    mbs_.assign((size_t)W_ * H_, MbState());
    level_overflow_ = false;
    next_addr_ = 0;
    multi_slice_ = n_slices > 1;
    for (int k = 0; k < n_slices; k++) {
        br_ = BitReader(slices[k].rbsp, slices[k].n);
        nal_ref_idc_ = slices[k].nal_ref_idc;
        int rc = slice_header(err);
        if (rc != RC_SUCCESS) return rc;
        rc = slice_data(err);
        if (rc != RC_SUCCESS) return rc;
    }
    if (next_addr_ != W_ * H_) { err = "the slices of the picture do not cover it"; return RC_FAILURE; }
    if (level_overflow_) { err = "transform coefficient level outside int16"; return RC_FAILURE; }
    return RC_SUCCESS;
}

int PictureDecoder::decode_slices(const SliceRbsp *slices, int n_slices, uint8_t *packed, size_t packed_bytes, std::string &err)
{
    if ((size_t)W_ * H_ * MVHP_MB_BYTES != packed_bytes) { err = "packed buffer size mismatch"; return RC_FAILURE; }
    if (n_slices < 1) { err = "no slice"; return RC_FAILURE; }
    out_ = packed;   // (every record is zeroed right before its macroblock is parsed: macroblock())
    compact_ = false;
    const int rc = run(slices, n_slices, err);
    if (rc != RC_SUCCESS) memset(out_, 0, packed_bytes);   // the records behind a failure were never written: no stale bytes
    return rc;
}

int PictureDecoder::decode(const uint8_t *rbsp, size_t n, uint8_t *packed, size_t packed_bytes, std::string &err)
{
    const SliceRbsp one{rbsp, n, nal_ref_idc_};
    return decode_slices(&one, 1, packed, packed_bytes, err);
}

int PictureDecoder::decode_slices_compact(const SliceRbsp *slices, int n_slices, uint8_t *buf, size_t cap, size_t *used, std::string &err)
{
    const size_t mbs = (size_t)W_ * H_;
    if (cap < mbs * MVHP_COMPACT_MB_BYTES_MAX + MVHP_COMPACT_SLACK_BYTES) { err = "compact buffer too small"; return RC_FAILURE; }
    if (n_slices < 1) { err = "no slice"; return RC_FAILURE; }
    out_ = nullptr;
    compact_ = true;
    mb_off_ = reinterpret_cast<uint32_t *>(buf);
    cw_base_ = cw_ = buf + mbs * 4;
    compact_max_ = MVHP_COMPACT_MAX_ENTRIES;
    if (const char *e = getenv("MINIVIDEO_TEST_COMPACT_MAX")) {   // test hook: exercise the dense fallback on ordinary streams
        const int v = atoi(e);
        if (v >= 0 && v < MVHP_COMPACT_MAX_ENTRIES) compact_max_ = (uint32_t)v;
    }
    const int rc = run(slices, n_slices, err);
    if (used) *used = rc == RC_SUCCESS ? (size_t)(cw_ - buf) : 0;
    return rc;
}

int PictureDecoder::decode_compact(const uint8_t *rbsp, size_t n, uint8_t *buf, size_t cap, size_t *used, std::string &err)
{
    const SliceRbsp one{rbsp, n, nal_ref_idc_};
    return decode_slices_compact(&one, 1, buf, cap, used, err);
}

// H6: decodeSliceHeader, h264_slice.c:156-334 (IDR / I slices only)
int PictureDecoder::slice_header(std::string &err)
{
    const unsigned first_mb = br_.ue();        // first_mb_in_slice: the reference ignores it, its MB loop starts at 0 (:1019)
    if (spec_) {                               // by the standard: the slice starts there; slices arrive in macroblock order
        if ((int)first_mb != next_addr_) { err = "slice does not start where the previous one ended (arbitrary slice order is not supported)"; return RC_FAILURE; }
        slice_first_ = (int)first_mb;
    } else {
        slice_first_ = 0;
    }
    const unsigned slice_type = br_.ue();
    br_.ue();                                  // pic_parameter_set_id (resolved by the caller)
    if (slice_type != 2 && slice_type != 7) { err = "slice: IDR slice_type must be I (2 or 7)"; return RC_FAILURE; }
    const unsigned frame_num = br_.bits(sps_.log2_max_frame_num);
    if (frame_num != 0) { err = "slice: IDR frame_num must be 0"; return RC_FAILURE; } // checkSliceHeader :531-537
    const unsigned idr_pic_id = br_.ue();
    if (idr_pic_id > 65535) { err = "slice: idr_pic_id out of range"; return RC_FAILURE; }
    if (sps_.poc_type == 0) {
        br_.bits(sps_.log2_max_poc_lsb);
        if (pps_.bottom_field_pic_order_in_frame_present) br_.se();
    } else if (sps_.poc_type == 1 && !sps_.delta_pic_order_always_zero) {
        br_.se();
        if (pps_.bottom_field_pic_order_in_frame_present) br_.se();
    }
    if (pps_.redundant_pic_cnt_present) br_.ue();
    if (nal_ref_idc_ != 0) { // dec_ref_pic_marking() of an IDR picture (:863-867)
        br_.bit();
        br_.bit();
    }
    const int slice_qp_delta = br_.se();
    slice_qp_ = 26 + pps_.pic_init_qp_minus26 + slice_qp_delta; // :293
    qp_prev_ = slice_qp_;
    if (slice_qp_ < 0 || slice_qp_ > 51) { err = "slice: SliceQPY out of range"; return RC_FAILURE; }
    if (pps_.deblocking_filter_control_present) {
        const unsigned idc = br_.ue();         // parsed and ignored: the reference never deblocks
        if (idc != 1) { br_.se(); br_.se(); }
    }
    if (br_.overrun()) { err = "slice header truncated"; return RC_FAILURE; }
    return RC_SUCCESS;
}

// decodeSliceData, h264_slice.c:1013-1142
int PictureDecoder::slice_data(std::string &err)
{
    const int n_mbs = W_ * H_;
    if (pps_.entropy_coding_mode) {
        while (!br_.byte_aligned()) {
            if (br_.bit() == 0) { err = "slice: cabac_alignment_one_bit is 0"; return RC_FAILURE; }
        }
        delete cabac_;
        cabac_ = new CabacEngine(*this);
        cabac_->init(slice_qp_);
    }
    cur_x_ = slice_first_ % W_;
    int addr = slice_first_;
    for (; addr < n_mbs; addr++) {
        cur_addr_ = addr;
        curA_ = (cur_x_ > 0 && addr - 1 >= slice_first_) ? addr - 1 : -1;
        int rc = macroblock(addr, err);
        if (++cur_x_ == W_) cur_x_ = 0;
        if (rc != RC_SUCCESS) return rc;
        if (pps_.entropy_coding_mode) {
            const int end = cabac_->decode_terminate();
            if (end) {
                // Reference mode: a slice that ends before the last macroblock fails the picture (the reference leaves its loop
                // with SUCCESS there and exports a picture with holes, h264_slice.c:1047-1139 -- documented divergence).
                // By the standard (spec mode) the next slice NAL of the picture continues at addr + 1.
                if (!spec_ && addr != n_mbs - 1) { err = "slice ends before the last macroblock (one slice per picture only)"; return RC_FAILURE; }
                addr++;
                break;
            }
            if (cabac_->overrun()) { err = "slice data truncated"; return RC_FAILURE; }
        } else {
            if (br_.overrun()) { err = "slice data truncated"; return RC_FAILURE; }
            // CAVLC: the reference's more_rbsp_data() is true to the very end of its sample (H12), its loop ends with the
            // picture; by the standard the slice ends where its data does
            if (spec_ && !br_.more_rbsp_data()) { addr++; break; }
        }
    }
    next_addr_ = addr;
    return RC_SUCCESS;
}

// neighbouring 4x4 blocks (6.4.11.4 via h264_spatial.c:559 luma, :631 chroma): inside the macroblock or in A / B
static inline int luma_neighbour_A(int addr, int addrA, int blk, int *blkN)
{
    const uint8_t e = nb_tables().lumaA[blk];
    *blkN = e & 15;
    return (e & 0x80) ? addrA : addr;
}
static inline int luma_neighbour_B(int addr, int addrB, int blk, int *blkN)
{
    const uint8_t e = nb_tables().lumaB[blk];
    *blkN = e & 15;
    return (e & 0x80) ? addrB : addr;
}
static inline int chroma_neighbour_A(int addr, int addrA, int blk, int *blkN)
{
    if (blk & 1) { *blkN = blk - 1; return addr; }
    *blkN = blk + 1;
    return addrA;
}
static inline int chroma_neighbour_B(int addr, int addrB, int blk, int *blkN)
{
    if (blk & 2) { *blkN = blk - 2; return addr; }
    *blkN = blk + 2;
    return addrB;
}

// G1: Intra_4x4_deriv_PredMode (h264_intra_prediction.c:196-290) and
// Intra_8x8_deriv_PredMode (:977-1083), constrained_intra_pred irrelevant in I pictures.
void PictureDecoder::derive_pred_modes(int addr, const uint8_t prev_flag[16], const uint8_t rem[16])
{
    MbState &mb = mbs_[addr];
    if (mb.kind == MVHP_KIND_I4x4) {
        for (int blk = 0; blk < 16; blk++) {
            int bA, bB;
            const int aA = luma_neighbour_A(addr, mbA(addr), blk, &bA), aB = luma_neighbour_B(addr, mbB(addr), blk, &bB);
            int mA = 2, mB = 2;
            if (aA >= 0 && aB >= 0) {
                const MbState &A = mbs_[aA], &B = mbs_[aB];
                if (A.kind == MVHP_KIND_I4x4) mA = A.pred[bA];
                else if (A.kind == MVHP_KIND_I8x8) mA = A.pred[bA >> 2];
                if (B.kind == MVHP_KIND_I4x4) mB = B.pred[bB];
                else if (B.kind == MVHP_KIND_I8x8) mB = B.pred[bB >> 2];
            }
            const int pm = mA < mB ? mA : mB;
            mb.pred[blk] = (uint8_t)(prev_flag[blk] ? pm : (rem[blk] < pm ? rem[blk] : rem[blk] + 1));
        }
    } else if (mb.kind == MVHP_KIND_I8x8) {
        for (int blk = 0; blk < 4; blk++) {
            // deriv_8x8lumablocks, h264_spatial.c:461
            int aA, bA, aB, bB;
            if (blk & 1) { aA = addr; bA = blk - 1; } else { aA = mbA(addr); bA = blk + 1; }
            if (blk & 2) { aB = addr; bB = blk - 2; } else { aB = mbB(addr); bB = blk + 2; }
            int mA = 2, mB = 2;
            if (aA >= 0 && aB >= 0) {
                const MbState &A = mbs_[aA], &B = mbs_[aB];
                if (A.kind == MVHP_KIND_I8x8) mA = A.pred[bA];
                else if (A.kind == MVHP_KIND_I4x4) mA = A.pred[bA * 4 + 1];
                if (B.kind == MVHP_KIND_I8x8) mB = B.pred[bB];
                else if (B.kind == MVHP_KIND_I4x4) mB = B.pred[bB * 4 + 2];
            }
            const int pm = mA < mB ? mA : mB;
            mb.pred[blk] = (uint8_t)(prev_flag[blk] ? pm : (rem[blk] < pm ? rem[blk] : rem[blk] + 1));
        }
    }
}

// MVHP_UNAVAIL_*: neighbours that geometry has but that belong to an earlier slice (6.4.8; h264_spatial.c:333-416 knows
// one slice only).  0 for every macroblock of a one-slice picture.
uint8_t PictureDecoder::unavail_bits(int addr) const
{
    if (slice_first_ == 0) return 0;
    const bool hasA = cur_x_ > 0, hasB = addr >= W_, hasC = hasB && cur_x_ < W_ - 1, hasD = hasA && hasB;
    // (cur_x_ is the column of `addr`: called before slice_data advances it)
    uint8_t u = 0;
    if (hasA && addr - 1 < slice_first_) u |= MVHP_UNAVAIL_A;
    if (hasB && addr - W_ < slice_first_) u |= MVHP_UNAVAIL_B;
    if (hasC && addr - W_ + 1 < slice_first_) u |= MVHP_UNAVAIL_C;
    if (hasD && addr - W_ - 1 < slice_first_) u |= MVHP_UNAVAIL_D;
    return u;
}

// I_PCM (mb_type 25, spec mode only): pcm_alignment_zero_bits, then 256 + 2 * 64 samples of 8 bits (7.3.5).  Under CABAC
// the arithmetic decoder has just decoded the terminate bin (9.3.1.2): the samples start at the next byte boundary behind
// the last bit the standard's 9-bit register has read, and the engine is initialised again behind them (contexts kept).
// Neighbour state (what later macroblocks derive from this one): nC = 16 (9.2.1), coded_block_flag = 1 (9.3.3.1.1.9),
// CodedBlockPattern read as 47, intra_chroma_pred_mode 0, mb_qp_delta 0 -- QP'Y carries over unchanged (7.4.5).
int PictureDecoder::pcm_samples(int addr, std::string &err)
{
    MbState &mb = mbs_[addr];
    if (pps_.entropy_coding_mode) br_.seek(cabac_->standard_bit_position());
    while (!br_.byte_aligned()) {
        if (br_.bit() != 0) { err = "pcm_alignment_zero_bit is 1"; return RC_FAILURE; }
    }
    if (br_.bits_left() < 384 * 8) { err = "I_PCM samples truncated"; return RC_FAILURE; }
    const uint8_t *smp = br_.data() + (br_.pos() >> 3);   // 256 luma (raster), 64 Cb, 64 Cr
    br_.skip(384 * 8);
    if (pps_.entropy_coding_mode) cabac_->restart();
    mb = MbState();
    mb.kind = MVHP_KIND_IPCM;
    mb.mb_type = 25;
    mb.cbp_luma = 15;
    mb.cbp_chroma = 2;
    memset(mb.tc_luma, 16, sizeof(mb.tc_luma));
    memset(mb.tc_c, 16, sizeof(mb.tc_c));
    mb.cbf = 0x7ffffffu;
    for (int b = 0; b < 16; b++) mb.pred[b] = 2;
    // the record: header + the samples in the layout of MVHP_KIND_IPCM (include/minivideo_hotpath.h)
    uint8_t area[MVHP_MB_COEFS * 2];
    memset(area, 0, sizeof(area));
    for (int j = 0; j < 8; j++) {
        memcpy(area + 64 * j, smp + 32 * j, 32);             // luma rows 2j, 2j+1
        memcpy(area + 64 * j + 32, smp + 256 + 8 * j, 8);    // Cb row j
        memcpy(area + 64 * j + 40, smp + 320 + 8 * j, 8);    // Cr row j
    }
    mvhp_mb_header_t h;
    memset(&h, 0, sizeof(h));
    h.mb_kind = MVHP_KIND_IPCM;
    h.qp_y = (uint8_t)qp_prev_;
    h.unavail = unavail_bits(addr);
    if (!compact_) {
        uint8_t *rec = out_ + (size_t)addr * MVHP_MB_BYTES;
        memcpy(rec, &h, sizeof(h));
        memcpy(rec + MVHP_MB_HEADER_BYTES, area, sizeof(area));
        return RC_SUCCESS;
    }
    uint8_t *rec = cw_;   // compact record in its dense form (flags bit 0): header + the 768-byte area
    mb_off_[addr] = (uint32_t)(rec - cw_base_);
    h.flags = 1;
    memcpy(rec, &h, sizeof(h));
    memcpy(rec + MVHP_MB_HEADER_BYTES, area, sizeof(area));
    cw_ = rec + MVHP_MB_HEADER_BYTES + sizeof(area);
    return RC_SUCCESS;
}

// H7: macroblock_layer, h264_macroblock.c:75-313
int PictureDecoder::macroblock(int addr, std::string &err)
{
    MbState &mb = mbs_[addr];
    const bool cabac = pps_.entropy_coding_mode;
    if (compact_) cwl_ = reinterpret_cast<uint32_t *>(cw_ + MVHP_MB_HEADER_BYTES);   // entries follow the header (written last)
    else {
        memset(out_ + (size_t)addr * MVHP_MB_BYTES, 0, MVHP_MB_BYTES);   // levels are written sparsely into a zero record
        coef_ = reinterpret_cast<int16_t *>(out_ + (size_t)addr * MVHP_MB_BYTES + MVHP_MB_HEADER_BYTES);
    }
    const unsigned mb_type = cabac ? cabac_->mb_type(addr) : br_.ue();
    if (mb_type == 25) {   // I_PCM: the reference answers UNSUPPORTED (:151-154); by the standard (spec mode) 7.3.5 / 8.3.5
        if (!spec_) { err = "I_PCM macroblocks are not supported"; return RC_UNSUPPORTED; }
        return pcm_samples(addr, err);
    }
    if (mb_type > 25) { err = "invalid mb_type in an I slice"; return RC_FAILURE; }
    mb.mb_type = (uint8_t)mb_type;
    uint8_t prev_flag[16] = {0}, rem[16] = {0};
    int i16_mode = 0;
    if (mb_type == 0) {
        mb.kind = MVHP_KIND_I4x4;
        if (pps_.transform_8x8_mode) {
            const int t8 = cabac ? cabac_->transform_size_8x8_flag(addr) : (int)br_.bit();
            mb.transform8x8 = (uint8_t)t8;
            if (t8) mb.kind = MVHP_KIND_I8x8;
        }
        const int n = (mb.kind == MVHP_KIND_I8x8) ? 4 : 16;
        for (int b = 0; b < n; b++) { // mb_pred, :393-450
            if (cabac) {
                prev_flag[b] = (uint8_t)cabac_->prev_intra_pred_mode_flag();
                if (!prev_flag[b]) rem[b] = (uint8_t)cabac_->rem_intra_pred_mode();
            } else {
                prev_flag[b] = (uint8_t)br_.bit();
                if (!prev_flag[b]) rem[b] = (uint8_t)br_.bits(3);
            }
        }
        derive_pred_modes(addr, prev_flag, rem);
    } else {
        // Table 7-11 (MbPartPredMode, :766-842)
        mb.kind = MVHP_KIND_I16x16;
        i16_mode = (int)(mb_type - 1) % 4;
        mb.cbp_chroma = (uint8_t)(((mb_type - 1) / 4) % 3);
        mb.cbp_luma = (mb_type > 12) ? 15 : 0;
    }
    {
        const unsigned cm = cabac ? cabac_->intra_chroma_pred_mode(addr) : br_.ue();
        if (cm > 3) { err = "intra_chroma_pred_mode out of range"; return RC_FAILURE; }
        mb.chroma_pred_mode = (uint8_t)cm;
    }
    if (mb.kind != MVHP_KIND_I16x16) {
        unsigned cbp;
        if (cabac) cbp = cabac_->coded_block_pattern(addr);
        else {
            const unsigned code = br_.ue(); // me(v), h264_expgolomb.c:130, Table 9-4 intra column
            if (code > 47) { err = "coded_block_pattern codeNum out of range"; return RC_FAILURE; }
            cbp = kCbpIntraFromCodeNum[code];
        }
        mb.cbp_luma = (uint8_t)(cbp % 16);
        mb.cbp_chroma = (uint8_t)(cbp / 16);
    }
    int mb_qp_delta = 0;
    nz_cur_ = 0;
    if (mb.cbp_luma > 0 || mb.cbp_chroma > 0 || mb.kind == MVHP_KIND_I16x16) {
        mb_qp_delta = cabac ? cabac_->mb_qp_delta(addr) : br_.se();
        mb.qp_delta_nonzero = mb_qp_delta != 0;
        int rc = residual(addr, err);
        if (rc != RC_SUCCESS) return rc;
    }
    // :263-269 (QpBdOffsetY = 0)
    int qp = qp_prev_;
    if (mb_qp_delta) qp = (qp_prev_ + mb_qp_delta + 52) % 52;
    if (qp < 0 || qp > 51) { err = "mb_qp_delta out of range"; return RC_FAILURE; }
    qp_prev_ = qp;

    // header of the packed record
    mvhp_mb_header_t h;
    memset(&h, 0, sizeof(h));
    h.mb_kind = mb.kind;
    h.qp_y = (uint8_t)qp;
    h.cbp = (uint8_t)(mb.cbp_luma | (mb.cbp_chroma << 4));
    h.chroma_pred_mode = mb.chroma_pred_mode;
    h.i16_pred_mode = (uint8_t)i16_mode;
    h.unavail = unavail_bits(addr);
    memcpy(h.pred_mode, mb.pred, 16);
    uint32_t nz = nz_cur_;   // collected while the blocks were decoded: every decoded level is non-zero
    if (mb.kind == MVHP_KIND_I8x8)
        for (int k = 0; k < 4; k++)
            if (nz & (0xfu << (4 * k))) nz |= 0xfu << (4 * k);
    h.nz_mask = nz;
    if (!compact_) {
        memcpy(out_ + (size_t)addr * MVHP_MB_BYTES, &h, sizeof(h));
        return RC_SUCCESS;
    }
    // compact record: header (reserved1 = number of entries) + one 32-bit entry per level; a macroblock with more than
    // MVHP_COMPACT_MAX_ENTRIES levels is sent as its dense coefficient area instead (flags bit 0)
    uint8_t *rec = cw_;
    mb_off_[addr] = (uint32_t)(rec - cw_base_);
    uint32_t *ent = reinterpret_cast<uint32_t *>(rec + MVHP_MB_HEADER_BYTES);
    const uint32_t n = (uint32_t)(cwl_ - ent);
    if (n > compact_max_) {
        int16_t dense[MVHP_MB_COEFS];
        memset(dense, 0, sizeof(dense));
        for (uint32_t i = 0; i < n; i++) dense[ent[i] & 0xffffu] = (int16_t)(ent[i] >> 16);
        memcpy(ent, dense, sizeof(dense));
        h.flags = 1;
        h.reserved1 = 0;
        cw_ = rec + MVHP_MB_HEADER_BYTES + sizeof(dense);
    } else {
        h.reserved1 = n;
        cw_ = reinterpret_cast<uint8_t *>(cwl_);
    }
    memcpy(rec, &h, sizeof(h));
    return RC_SUCCESS;
}

// Where a decoded level goes: entropy decoders hand (coefficient index, level) pairs to put(), which applies the
// inverse scan (h264_transform.c:440-480, a pure permutation) and the record layout of include/minivideo_hotpath.h in
// one table lookup, and notes which blocks hold a level (nz_mask).
namespace {
struct SinkTables {
    uint8_t l4[16];        // 4x4 block, coefficient i -> raster slot
    uint8_t ac[16];        // Intra16x16 / chroma AC: coefficient k is zig-zag position k + 1
    uint8_t dc16[16];      // Intra16x16 DC: coefficient i -> slot 0 of the block at that raster position (/16)
    uint8_t l8[64];        // 8x8 block
    uint8_t l8i[4][16];    // CAVLC 8x8: four interleaved 4x4 blocks, coefficient i of part p = 8x8 index 4i + p
    SinkTables()
    {
        for (int i = 0; i < 16; i++) {
            l4[i] = kZigzag4x4[i];
            ac[i] = kZigzag4x4[i < 15 ? i + 1 : 15];
            const int rc = kZigzag4x4[i];
            dc16[i] = (uint8_t)blk4_from_xy((rc & 3) * 4, (rc >> 2) * 4);
        }
        for (int i = 0; i < 64; i++) l8[i] = kZigzag8x8[i];
        for (int p = 0; p < 4; p++)
            for (int i = 0; i < 16; i++) l8i[p][i] = kZigzag8x8[4 * i + p];   // h264_macroblock.c:1175-1184
    }
};
const SinkTables g_sink;
} // namespace

void PictureDecoder::sink_begin(int addr, int cat, int blkIdx, int part)
{
    (void)addr;
    sink_.scale = 1;
    sink_.nz_per_coef = false;
    switch (cat) {
    case CAT_LUMA_8x8:
        if (part >= 0) { sink_.tab = g_sink.l8i[part]; }
        else sink_.tab = g_sink.l8;
        sink_.base = blkIdx * 64;
        sink_.nz_bit = 0xfu << (4 * blkIdx);
        break;
    case CAT_LUMA_4x4: sink_.tab = g_sink.l4; sink_.base = blkIdx * 16; sink_.nz_bit = 1u << blkIdx; break;
    case CAT_LUMA_16x16_DC:   // c1[row][col] -> slot 0 of the block at that raster position
        sink_.tab = g_sink.dc16; sink_.base = 0; sink_.scale = 16; sink_.nz_bit = 1u; sink_.nz_per_coef = true;
        break;
    case CAT_LUMA_16x16_AC: sink_.tab = g_sink.ac; sink_.base = blkIdx * 16; sink_.nz_bit = 1u << blkIdx; break;
    case CAT_CHROMA_DC_CB:
    case CAT_CHROMA_DC_CR: {   // DC level k -> slot 0 of chroma block k
        static const uint8_t ident[4] = {0, 1, 2, 3};
        sink_.tab = ident; sink_.base = 256 + (cat - CAT_CHROMA_DC_CB) * 64; sink_.scale = 16;
        sink_.nz_bit = 1u << (16 + 4 * (cat - CAT_CHROMA_DC_CB)); sink_.nz_per_coef = true;
        break;
    }
    default:   // chroma AC
        sink_.tab = g_sink.ac; sink_.base = 256 + (cat - CAT_CHROMA_AC_CB) * 64 + blkIdx * 16;
        sink_.nz_bit = 1u << (16 + 4 * (cat - CAT_CHROMA_AC_CB) + blkIdx);
        break;
    }
}

// residual_luma + residual_chroma, h264_macroblock.c:1102-1295
int PictureDecoder::residual(int addr, std::string &err)
{
    MbState &mb = mbs_[addr];
    const bool cabac = pps_.entropy_coding_mode;
    auto block = [&](int cat, int blkIdx, int part, int endIdx, int maxNum) -> int {
        sink_begin(addr, cat, blkIdx, part);
        const int cat_e = (part >= 0) ? CAT_LUMA_4x4 : cat;   // a CAVLC 8x8 block is parsed as four 4x4 blocks
        const int blk_e = (part >= 0) ? blkIdx * 4 + part : blkIdx;
        return cabac ? cabac_->residual_block(addr, 0, endIdx, maxNum, cat_e, blk_e)
                     : residual_block_cavlc(addr, 0, endIdx, maxNum, cat_e, blk_e);
    };
    if (mb.kind == MVHP_KIND_I16x16) {
        if (block(CAT_LUMA_16x16_DC, 0, -1, 15, 16) != RC_SUCCESS) { err = "residual: Intra16x16 DC block"; return RC_FAILURE; }
    }
    for (int i8 = 0; i8 < 4; i8++) {
        if (!mb.transform8x8 || !cabac) {
            for (int i4 = 0; i4 < 4; i4++) {
                const int blk = i8 * 4 + i4;
                if (mb.cbp_luma & (1 << i8)) {
                    if (mb.kind == MVHP_KIND_I16x16) {
                        if (block(CAT_LUMA_16x16_AC, blk, -1, 14, 15) != RC_SUCCESS) { err = "residual: Intra16x16 AC block"; return RC_FAILURE; }
                    } else if (mb.transform8x8) {   // :1175-1184: coefficient i of 4x4 block i4 is 8x8 coefficient 4i + i4
                        if (block(CAT_LUMA_8x8, i8, i4, 15, 16) != RC_SUCCESS) { err = "residual: luma 4x4 block"; return RC_FAILURE; }
                    } else {
                        if (block(CAT_LUMA_4x4, blk, -1, 15, 16) != RC_SUCCESS) { err = "residual: luma 4x4 block"; return RC_FAILURE; }
                    }
                } else {
                    mb.tc_luma[blk] = 0;
                }
            }
        } else if (mb.cbp_luma & (1 << i8)) {
            if (block(CAT_LUMA_8x8, i8, -1, 63, 64) != RC_SUCCESS) { err = "residual: luma 8x8 block"; return RC_FAILURE; }
        }
    }
    for (int c = 0; c < 2; c++) {
        if (mb.cbp_chroma & 3) {
            if (block(CAT_CHROMA_DC_CB + c, 0, -1, 3, 4) != RC_SUCCESS) { err = "residual: chroma DC block"; return RC_FAILURE; }
        }
    }
    for (int c = 0; c < 2; c++) {
        for (int blk = 0; blk < 4; blk++) {
            if (mb.cbp_chroma & 2) {
                if (block(CAT_CHROMA_AC_CB + c, blk, -1, 14, 15) != RC_SUCCESS) { err = "residual: chroma AC block"; return RC_FAILURE; }
            }
        }
    }
    return RC_SUCCESS;
}

// ---------------------------------------------------------------------------
// H8: CAVLC (h264_cavlc.c:79-346)
// ---------------------------------------------------------------------------
int PictureDecoder::nC_for(int addr, int cat, int blkIdx) const
{
    if (cat == CAT_CHROMA_DC_CB || cat == CAT_CHROMA_DC_CR) return -1;
    int aA, aB, bA = 0, bB = 0, nA = 0, nB = 0;
    if (cat == CAT_CHROMA_AC_CB || cat == CAT_CHROMA_AC_CR) {
        const int c = cat - CAT_CHROMA_AC_CB;
        aA = chroma_neighbour_A(addr, mbA(addr), blkIdx, &bA);
        aB = chroma_neighbour_B(addr, mbB(addr), blkIdx, &bB);
        if (aA >= 0) nA = mbs_[aA].tc_c[c][bA];
        if (aB >= 0) nB = mbs_[aB].tc_c[c][bB];
    } else {
        aA = luma_neighbour_A(addr, mbA(addr), blkIdx, &bA);
        aB = luma_neighbour_B(addr, mbB(addr), blkIdx, &bB);
        if (aA >= 0) nA = mbs_[aA].tc_luma[bA];
        if (aB >= 0) nB = mbs_[aB].tc_luma[bB];
    }
    if (aA >= 0 && aB >= 0) return (nA + nB + 1) >> 1;
    if (aA >= 0) return nA;
    if (aB >= 0) return nB;
    return 0;
}

// ---- decoding tables built once from the (length, code) tables of h264_tables.h ----
// coeff_token for 0 <= nC < 8: every code is `lz` zeros, a one, then at most 3 more bits, so the pair
// (leading zeros, next three bits) identifies it.  Entry: len | total << 8 | trailing_ones << 16 (0 = invalid).
struct CavlcTables {
    uint32_t coeff_token[3][17][8];
    uint16_t total_zeros[15][512];   // index: next 9 bits -> len | value << 8
    uint16_t coeff_token_cdc[256];   // chroma DC (nC = -1): next 8 bits -> len | total << 4 | trailing_ones << 8
    uint8_t  total_zeros_cdc[3][8];  // next 3 bits -> len | value << 4
    uint8_t  run_before[6][8];       // zerosLeft 1..6: next 3 bits -> len | value << 4
    // a level whose prefix and suffix fit the next 8 bits (9.2.2.1 with level_prefix < 14: levelCode = (prefix << suffixLength)
    // + suffix), per suffixLength 0..6: len | levelCode << 8; 0 = longer than 8 bits or an escape, decoded the long way
    uint16_t level8[7][256];
    CavlcTables()
    {
        memset(this, 0, sizeof(*this));
        for (int sl = 0; sl < 7; sl++)
            for (int v = 0; v < 256; v++) {
                int prefix = 0;
                while (prefix < 8 && !(v & (0x80 >> prefix))) prefix++;
                const int len = prefix + 1 + sl;
                if (prefix >= 8 || len > 8) continue;
                const int suffix = (v >> (8 - len)) & ((1 << sl) - 1);
                level8[sl][v] = (uint16_t)(len | (((prefix << sl) + suffix) << 8));
            }
        for (int tab = 0; tab < 3; tab++)
            for (int t = 0; t < 4; t++)
                for (int n = 0; n < 17; n++) {
                    const int len = kCoeffTokenLen[tab][t][n], code = kCoeffTokenCode[tab][t][n];
                    if (!len) continue;
                    int width = 0;
                    while ((code >> width) != 0) width++;        // position of the leading one
                    const int lz = len - width, sfx = width - 1; // bits after the one
                    const int sfx_bits = code & ((1 << sfx) - 1);
                    for (int fill = 0; fill < (1 << (3 - sfx)); fill++)
                        coeff_token[tab][lz][(sfx_bits << (3 - sfx)) | fill] = (uint32_t)len | (n << 8) | (t << 16);
                }
        for (int t = 0; t < 4; t++)
            for (int n = 0; n < 5; n++) {
                const int len = kCoeffTokenChromaDcLen[t][n], code = kCoeffTokenChromaDcCode[t][n];
                if (!len) continue;
                for (int fill = 0; fill < (1 << (8 - len)); fill++)
                    coeff_token_cdc[(code << (8 - len)) | fill] = (uint16_t)(len | (n << 4) | (t << 8));
            }
        for (int v = 0; v < 15; v++)
            for (int z = 0; z < 16; z++) {
                const int len = kTotalZerosLen[v][z], code = kTotalZerosCode[v][z];
                if (!len) continue;
                for (int fill = 0; fill < (1 << (9 - len)); fill++)
                    total_zeros[v][(code << (9 - len)) | fill] = (uint16_t)(len | (z << 8));
            }
        for (int v = 0; v < 3; v++)
            for (int z = 0; z < 4; z++) {
                const int len = kTotalZerosChromaDcLen[v][z], code = kTotalZerosChromaDcCode[v][z];
                if (!len) continue;
                for (int fill = 0; fill < (1 << (3 - len)); fill++)
                    total_zeros_cdc[v][(code << (3 - len)) | fill] = (uint8_t)(len | (z << 4));
            }
        for (int v = 0; v < 6; v++)
            for (int r = 0; r < 15; r++) {
                const int len = kRunBeforeLen[v][r], code = kRunBeforeCode[v][r];
                if (!len) continue;
                for (int fill = 0; fill < (1 << (3 - len)); fill++)
                    run_before[v][(code << (3 - len)) | fill] = (uint8_t)(len | (r << 4));
            }
    }
};
static const CavlcTables g_cavlc;

int PictureDecoder::residual_block_cavlc(int addr, int startIdx, int endIdx, int maxNumCoeff, int cat, int blkIdx)
{
    MbState &mb = mbs_[addr];
    const int nC = nC_for(addr, cat, blkIdx);
    BitWindow bw(br_);   // (the member reader is not used below: its position is updated when bw ends)
    int total = -1, t1s = 0;
    // 9.2.1 coeff_token
    if (nC >= 8) {
        const unsigned v = bw.bits(6);
        if (v == 3) { total = 0; t1s = 0; }
        else { total = (int)(v >> 2) + 1; t1s = (int)(v & 3); if (t1s > total) return RC_FAILURE; }
    } else if (nC == -1) {
        const uint16_t e = g_cavlc.coeff_token_cdc[bw.peek(8)];
        if (e) { bw.skip((int)(e & 15u)); total = (e >> 4) & 15; t1s = e >> 8; }
    } else {
        const int tab = nC < 2 ? 0 : (nC < 4 ? 1 : 2);
        const int lz = bw.leading_zeros32();
        if (lz <= 16) {
            const uint32_t e = g_cavlc.coeff_token[tab][lz][(bw.peek(lz + 4) & 7u)];
            if (e) { bw.skip((int)(e & 255u)); total = (int)((e >> 8) & 255u); t1s = (int)(e >> 16); }
        }
    }
    if (total < 0) return RC_FAILURE;
    // h264_cavlc.c:207-212: the count is recorded for neighbours (luma categories share tc_luma;
    // the Intra16x16 DC count lands in slot blkIdx = 0 and is overwritten by AC block 0)
    if (cat <= CAT_LUMA_16x16_AC) mb.tc_luma[blkIdx] = (uint8_t)total;
    else if (cat == CAT_CHROMA_AC_CB) mb.tc_c[0][blkIdx] = (uint8_t)total;
    else if (cat == CAT_CHROMA_AC_CR) mb.tc_c[1][blkIdx] = (uint8_t)total;
    if (total == 0) return RC_SUCCESS;
    if (total > maxNumCoeff) return RC_SUCCESS; // silently skipped (:218)

    int level[16 + 3];
    int suffixLength = (total > 10 && t1s < 3) ? 1 : 0;
    {   // trailing ones: up to three sign bits, read at once
        const uint32_t sgn = bw.peek(3);
        level[0] = 1 - 2 * (int)((sgn >> 2) & 1u);
        level[1] = 1 - 2 * (int)((sgn >> 1) & 1u);
        level[2] = 1 - 2 * (int)(sgn & 1u);
        bw.skip(t1s);
    }
    int first_adjust = (t1s < 3) ? 2 : 0;   // the first level behind fewer than three trailing ones: levelCode += 2
    for (int i = t1s; i < total; i++) {
        int levelCode;
        const uint16_t e = g_cavlc.level8[suffixLength][bw.peek(8)];
        if (e) {
            bw.skip((int)(e & 255u));
            levelCode = (int)(e >> 8);
        } else {
            const int level_prefix = bw.leading_zeros32();
            if (level_prefix > 28 || bw.overrun()) return RC_FAILURE;
            bw.skip(level_prefix + 1);
            levelCode = (level_prefix < 15 ? level_prefix : 15) << suffixLength;
            if (suffixLength > 0 || level_prefix >= 14) {
                int size = suffixLength;
                if (level_prefix == 14 && suffixLength == 0) size = 4;
                else if (level_prefix > 14) size = level_prefix - 3;
                if (size > 0) levelCode += (int)bw.bits(size);
            }
            if (level_prefix >= 15 && suffixLength == 0) levelCode += 15;
            if (level_prefix >= 16) levelCode += (1 << (level_prefix - 3)) - 4096;
        }
        levelCode += first_adjust;
        first_adjust = 0;
        const int sign = -(levelCode & 1), mag = (levelCode + 2) >> 1;   // even: (levelCode + 2) >> 1, odd: (-levelCode - 1) >> 1
        level[i] = (mag ^ sign) - sign;
        suffixLength += (suffixLength == 0);
        suffixLength += (mag > (3 << (suffixLength - 1))) & (suffixLength < 6);
    }
    int zerosLeft = 0;
    if (total < endIdx - startIdx + 1) {
        int tz;
        if (nC == -1) {
            const uint8_t e = g_cavlc.total_zeros_cdc[total - 1][bw.peek(3)];
            if (!e) return RC_FAILURE;
            bw.skip((int)(e & 15u));
            tz = e >> 4;
        } else {
            const uint16_t e = g_cavlc.total_zeros[total - 1][bw.peek(9)];
            if (!e) return RC_FAILURE;
            bw.skip((int)(e & 255u));
            tz = e >> 8;
        }
        zerosLeft = tz;
    }
    // run_before (9.2.3): level[0] is the highest-frequency coefficient, at index total - 1 + total_zeros; every run_before
    // moves the next one down.  Positions first, then the levels are handed over in ascending order as before.
    int pos[16];
    int idx = total - 1 + zerosLeft;
    if (startIdx + idx > endIdx || startIdx + idx >= 64) return RC_FAILURE;
    int i = 0;
    for (; i < total - 1 && zerosLeft > 0; i++) {
        int rb;
        if (zerosLeft <= 6) {
            const uint8_t e = g_cavlc.run_before[zerosLeft - 1][bw.peek(3)];
            if (!e) return RC_FAILURE;
            bw.skip((int)(e & 15u));
            rb = e >> 4;
        } else { // Table 9-10, zerosLeft > 6: 3-bit codes 111..001 = 0..6, then 0001 = 7, 00001 = 8, ...
            const uint32_t v3 = bw.peek(3);
            if (v3) { bw.skip(3); rb = 7 - (int)v3; }
            else {
                const int lz = bw.leading_zeros32();
                if (lz > 10) return RC_FAILURE;
                bw.skip(lz + 1);
                rb = lz + 4;
            }
        }
        zerosLeft -= rb;
        if (zerosLeft < 0) return RC_FAILURE;
        pos[i] = idx;
        idx -= rb + 1;
    }
    for (; i < total; i++) pos[i] = idx--;   // no zeros left: the rest are adjacent (the last one sits on what is left)
    for (i = total - 1; i >= 0; i--) put(startIdx + pos[i], level[i]);
    return RC_SUCCESS;
}

} // namespace h264
