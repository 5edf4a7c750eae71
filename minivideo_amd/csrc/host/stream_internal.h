// stream_internal.h -- the parsed elementary stream behind mvhp_stream_t.
#pragma once
#include <string>
#include <vector>

#include "h264_frontend.h"

struct mvhp_stream {
    struct Idr {
        size_t      sample = 0;   // index into samples
        std::vector<size_t> more; // MVHP_STREAM_SPEC: the further slice NAL units of the picture (first_mb_in_slice > 0)
        h264::Sps   sps;          // parameter sets in force when this picture was reached
        h264::Pps   pps;
        bool        ok = false;
        std::string why;
        size_t      nal_bytes = 0; // MVHP_STREAM_SPEC: NAL bytes of all the picture's slices together (the size guard goes by them)
    };
    const uint8_t *data = nullptr;
    size_t size = 0;
    std::vector<h264::EsSample> samples;
    std::vector<Idr> idrs;
    int param_errors = 0;
    bool spec = false;            // MVHP_STREAM_SPEC: standard-conformant index + luma-DC rule (opt-in, outside parity)

    int build(std::string &err);      // Annex-B elementary stream
    int build_mp4(std::string &err);  // ISO-BMFF: avcC parameter sets + length-prefixed NAL units of the sync samples
    int decode_packed(int idr, void *packed, size_t bytes, std::string &err) const;
    int decode_compact(int idr, void *buf, size_t cap, size_t *used, std::string &err) const;
};
