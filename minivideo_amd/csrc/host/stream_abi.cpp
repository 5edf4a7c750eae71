// stream_abi.cpp -- mvhp_stream_* entry points of include/minivideo_hotpath.h:
// Annex-B buffer -> sample table -> parameter sets -> packed pictures (host only).
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "h264_frontend.h"
#include "mp4_demux.h"
#include "stream_internal.h"

using namespace h264;

// A slice cannot be smaller than its picture allows: a CAVLC macroblock takes at least one bit, a CABAC one rarely
// less than a quarter of a bit -- eight macroblocks per payload bit is far outside anything an encoder emits.  Checked
// before any picture-sized buffer is reserved, so a 100-byte file announcing 1024 x 1024 macroblocks costs nothing.
static bool slice_can_hold_picture(const Sps &sps, size_t nal_size)
{
    return (uint64_t)nal_size * 64u >= (uint64_t)sps.width_mbs * (uint64_t)sps.height_map_units;
}

int mvhp_stream::build(std::string &err)
{
    if ((spec ? index_annexb_spec(data, size, samples) : index_annexb(data, size, samples)) != RC_SUCCESS) { err = "no NAL unit found in the bitstream"; return RC_FAILURE; }
    Sps sps_tab[32];
    Pps pps_tab[256];
    std::vector<uint8_t> rbsp;
    for (size_t i = 0; i < samples.size(); i++) {
        const EsSample &s = samples[i];
        if (s.nal_size < 2) continue;
        // (of a slice only first_mb_in_slice, slice_type and pic_parameter_set_id are read here: three ue(v) of at most 65
        // bits each -- the picture's 200 KB are unescaped by the thread that decodes it, not by this loop)
        unescape_rbsp(data + s.offset + 1, s.nal_unit_type == 5 ? std::min<size_t>(s.nal_size - 1, 64) : s.nal_size - 1, rbsp);
        BitReader br(rbsp.data(), rbsp.size());
        std::string e;
        if (s.nal_unit_type == 7) {
            Sps sps;
            if (parse_sps(br, sps, e, spec) == RC_SUCCESS) sps_tab[sps.sps_id] = sps;
            else param_errors++;
        } else if (s.nal_unit_type == 8) {
            Pps pps;
            if (parse_pps(br, sps_tab, pps, e, spec) == RC_SUCCESS) pps_tab[pps.pps_id] = pps;
            else param_errors++;
        } else if (s.nal_unit_type == 5) {
            Idr idr;
            idr.sample = i;
            const unsigned first_mb = br.ue(); // first_mb_in_slice (the reference ignores it, h264_slice.c:1019)
            br.ue(); // slice_type
            const unsigned pid = br.ue();
            if (pid < 256 && pps_tab[pid].valid && sps_tab[pps_tab[pid].sps_id].valid) {
                idr.pps = pps_tab[pid];
                idr.sps = sps_tab[idr.pps.sps_id];
                // reference mode: one slice NAL = one picture, and it must be able to hold it.  MVHP_STREAM_SPEC: a picture may
                // come in many small slices (one per macroblock row of flat content: ten bytes each) -- the guard goes by the
                // bytes of ALL its slices, once the last one is known (below)
                idr.nal_bytes = s.nal_size;
                idr.ok = spec || slice_can_hold_picture(idr.sps, s.nal_size);
                if (!idr.ok) idr.why = "slice NAL too small for the picture size of its SPS";
                if (spec && first_mb != 0) {   // a further slice of the previous picture: not a picture of its own
                    if (idrs.empty()) continue;                       // (a stream that starts in the middle of a picture)
                    Idr &pic = idrs.back();
                    if (pic.ok && pic.pps.pps_id != (int)pid) { pic.ok = false; pic.why = "the slices of a picture refer to different picture parameter sets"; }
                    pic.more.push_back(i);
                    pic.nal_bytes += s.nal_size;
                    continue;
                }
            } else {
                idr.why = "slice refers to a parameter set that was not (successfully) received";
            }
            idrs.push_back(idr);
        }
    }
    if (spec)
        for (Idr &pic : idrs)
            if (pic.ok && !slice_can_hold_picture(pic.sps, pic.nal_bytes)) {
                pic.ok = false;
                pic.why = "the slices of the picture are too small, together, for the picture size of its SPS";
            }
    return RC_SUCCESS;
}

// MP4: parameter sets come from the avcC box, pictures from the sync samples of the first
// video track; a sample is a sequence of length-prefixed NAL units.  The decode target is "the same pictures as the
// elementary-stream path" -- in-band SPS/PPS are honoured, the first IDR slice NAL of a sync sample is the picture.
int mvhp_stream::build_mp4(std::string &err)
{
    mp4::VideoTrack trk;
    if (!mp4::parse(data, size, trk, err)) return RC_FAILURE;
    Sps sps_tab[32];
    Pps pps_tab[256];
    std::vector<uint8_t> rbsp;
    auto add_param = [&](size_t off, size_t len) {
        if (len < 2 || off + len > size) return;
        const int type = data[off] & 31;
        unescape_rbsp(data + off + 1, len - 1, rbsp);
        BitReader br(rbsp.data(), rbsp.size());
        std::string e;
        if (type == 7) { Sps sps; if (parse_sps(br, sps, e) == RC_SUCCESS) sps_tab[sps.sps_id] = sps; else param_errors++; }
        else if (type == 8) { Pps pps; if (parse_pps(br, sps_tab, pps, e) == RC_SUCCESS) pps_tab[pps.pps_id] = pps; else param_errors++; }
    };
    for (const mp4::NalRef &n : trk.sps) add_param(n.offset, n.size);
    for (const mp4::NalRef &n : trk.pps) add_param(n.offset, n.size);
    for (const mp4::Sample &smp : trk.samples) {
        if (!smp.sync || smp.size == 0) continue;
        size_t p = smp.offset;
        const size_t end = smp.offset + smp.size;
        bool took = false;
        while (p + (size_t)trk.nal_length_size < end) {
            size_t len = 0;
            for (int i = 0; i < trk.nal_length_size; i++) len = (len << 8) | data[p + i];
            p += (size_t)trk.nal_length_size;
            if (len == 0 || len > end - p) break;
            const int type = data[p] & 31;
            if (type == 7 || type == 8) add_param(p, len);
            else if (type == 5 && !took) {
                EsSample s;
                s.offset = p;
                s.sample_size = s.nal_size = len;
                s.nal_unit_type = 5;
                s.nal_ref_idc = (data[p] >> 5) & 3;
                s.is_idr = true;
                samples.push_back(s);
                Idr idr;
                idr.sample = samples.size() - 1;
                unescape_rbsp(data + p + 1, len - 1 < 16 ? len - 1 : 16, rbsp);
                BitReader br(rbsp.data(), rbsp.size());
                br.ue();
                br.ue();
                const unsigned pid = br.ue();
                if (pid < 256 && pps_tab[pid].valid && sps_tab[pps_tab[pid].sps_id].valid) {
                    idr.pps = pps_tab[pid];
                    idr.sps = sps_tab[idr.pps.sps_id];
                    idr.ok = slice_can_hold_picture(idr.sps, len);
                    if (!idr.ok) idr.why = "slice NAL too small for the picture size of its SPS";
                } else {
                    idr.why = "slice refers to a parameter set that was not (successfully) received";
                }
                idrs.push_back(idr);
                took = true;
            }
            p += len;
        }
    }
    if (idrs.empty()) { err = "MP4: no IDR picture in the sync samples"; return RC_FAILURE; }
    return RC_SUCCESS;
}

// the slice NAL units of picture `idr`, unescaped: one in the reference's world, several for MVHP_STREAM_SPEC streams
static void picture_slices(const mvhp_stream &st, const mvhp_stream::Idr &idr, std::vector<std::vector<uint8_t>> &store,
                           std::vector<SliceRbsp> &slices)
{
    store.resize(1 + idr.more.size());
    slices.resize(store.size());
    for (size_t k = 0; k < store.size(); k++) {
        const EsSample &s = st.samples[k == 0 ? idr.sample : idr.more[k - 1]];
        unescape_rbsp(st.data + s.offset + 1, s.nal_size - 1, store[k]);
        slices[k].rbsp = store[k].data();
        slices[k].n = store[k].size();
        slices[k].nal_ref_idc = s.nal_ref_idc;
    }
}

int mvhp_stream::decode_packed(int k, void *packed, size_t bytes, std::string &err) const
{
    if (k < 0 || (size_t)k >= idrs.size()) { err = "IDR index out of range"; return RC_FAILURE; }
    const Idr &idr = idrs[k];
    if (!idr.ok) { err = idr.why; return RC_FAILURE; }
    std::vector<std::vector<uint8_t>> store;
    std::vector<SliceRbsp> slices;
    picture_slices(*this, idr, store, slices);
    PictureDecoder pd(idr.sps, idr.pps, slices[0].nal_ref_idc, spec);
    return pd.decode_slices(slices.data(), (int)slices.size(), (uint8_t *)packed, bytes, err);
}

int mvhp_stream::decode_compact(int k, void *buf, size_t cap, size_t *used, std::string &err) const
{
    if (used) *used = 0;
    if (k < 0 || (size_t)k >= idrs.size()) { err = "IDR index out of range"; return RC_FAILURE; }
    const Idr &idr = idrs[k];
    if (!idr.ok) { err = idr.why; return RC_FAILURE; }
    std::vector<std::vector<uint8_t>> store;
    std::vector<SliceRbsp> slices;
    picture_slices(*this, idr, store, slices);
    PictureDecoder pd(idr.sps, idr.pps, slices[0].nal_ref_idc, spec);
    return pd.decode_slices_compact(slices.data(), (int)slices.size(), (uint8_t *)buf, cap, used, err);
}

static thread_local std::string g_stream_err;

extern "C" {

MVHP_EXPORT const char *mvhp_stream_last_error(void) { return g_stream_err.c_str(); }

MVHP_EXPORT int mvhp_stream_open(const uint8_t *data, size_t size, mvhp_stream_t **out)
{
    if (!out || !data) return MVHP_FAILURE;
    *out = nullptr;
    mvhp_stream *s = new mvhp_stream();
    s->data = data;
    s->size = size;
    if (s->build(g_stream_err) != RC_SUCCESS) { delete s; return MVHP_FAILURE; }
    *out = s;
    return MVHP_SUCCESS;
}

MVHP_EXPORT int mvhp_stream_open_ex(const uint8_t *data, size_t size, uint32_t flags, mvhp_stream_t **out)
{
    if (!out || !data || (flags & ~MVHP_STREAM_SPEC)) return MVHP_FAILURE;
    *out = nullptr;
    mvhp_stream *s = new mvhp_stream();
    s->data = data;
    s->size = size;
    s->spec = (flags & MVHP_STREAM_SPEC) != 0;
    if (s->build(g_stream_err) != RC_SUCCESS) { delete s; return MVHP_FAILURE; }
    *out = s;
    return MVHP_SUCCESS;
}

MVHP_EXPORT int mvhp_stream_open_mp4(const uint8_t *data, size_t size, mvhp_stream_t **out)
{
    if (!out || !data) return MVHP_FAILURE;
    *out = nullptr;
    mvhp_stream *s = new mvhp_stream();
    s->data = data;
    s->size = size;
    if (s->build_mp4(g_stream_err) != RC_SUCCESS) { delete s; return MVHP_FAILURE; }
    *out = s;
    return MVHP_SUCCESS;
}

MVHP_EXPORT void mvhp_stream_close(mvhp_stream_t *s) { delete s; }

MVHP_EXPORT int mvhp_stream_idr_count(const mvhp_stream_t *s) { return s ? (int)s->idrs.size() : 0; }

MVHP_EXPORT int mvhp_stream_params(const mvhp_stream_t *s, int idr, mvhp_stream_params_t *out)
{
    if (!s || !out || idr < 0 || (size_t)idr >= s->idrs.size() || !s->idrs[idr].ok) return MVHP_FAILURE;
    const mvhp_stream::Idr &i = s->idrs[idr];
    out->width_mbs = (uint32_t)i.sps.width_mbs;
    out->height_mbs = (uint32_t)i.sps.height_map_units;
    out->chroma_qp_index_offset = i.pps.chroma_qp_index_offset;
    out->second_chroma_qp_index_offset = i.pps.second_chroma_qp_index_offset;
    out->flags = (i.pps.transform_8x8_mode ? MVHP_PARAM_MAY_HAVE_8X8 : 0u) | (s->spec ? MVHP_PARAM_SPEC_LUMA_DC : 0u);
    memset(out->scaling4, 16, sizeof(out->scaling4));
    memset(out->scaling8, 16, sizeof(out->scaling8));
    if (s->spec) {   // SURVEY 8f row f4 (outside parity): several slices, scaling matrices
        if (!i.more.empty()) out->flags |= MVHP_PARAM_SLICES;
        if ((i.sps.scaling.present || i.pps.scaling.present) &&
            effective_intra_scaling(i.sps.scaling, i.pps.scaling, i.pps.transform_8x8_mode, out->scaling4, out->scaling8))
            out->flags |= MVHP_PARAM_SCALING;
    }
    return MVHP_SUCCESS;
}

MVHP_EXPORT int mvhp_stream_decode_packed(const mvhp_stream_t *s, int idr, void *packed, size_t packed_bytes)
{
    if (!s || !packed) return MVHP_FAILURE;
    std::string err;
    const int rc = s->decode_packed(idr, packed, packed_bytes, err);
    if (rc != RC_SUCCESS) g_stream_err = err;
    return rc;
}

MVHP_EXPORT int mvhp_stream_decode_compact(const mvhp_stream_t *s, int idr, void *buf, size_t cap, size_t *used)
{
    if (!s || !buf) return MVHP_FAILURE;
    std::string err;
    const int rc = s->decode_compact(idr, buf, cap, used, err);
    if (rc != RC_SUCCESS) g_stream_err = err;
    return rc;
}

} // extern "C"
