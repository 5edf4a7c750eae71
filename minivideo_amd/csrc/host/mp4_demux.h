// mp4_demux.h -- minimal ISO-BMFF reader for the IDR thumbnail path (SURVEY.md 8f "next" row f1):
// finds the first H.264 video track, its avcC parameter sets and its sample table.  Restates the parts of
// demuxer/mp4/mp4.c the decode path depends on: box walk (:633-692, :895-949, :1244-1310, :1377-1427, :1503-1625),
// stsd/avc1/avcC (:1627-1939), stss (:2301), stsc (:2362), stsz (:2448), stco/co64 (:2527) and the sample-map
// construction of convertTrack (:150-500).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace mp4 {

struct NalRef { size_t offset = 0, size = 0; };      // NAL header byte .. end of NAL

struct Sample {
    size_t offset = 0, size = 0;
    bool   sync = false;
};

struct VideoTrack {
    bool found = false;
    bool is_h264 = false;
    int  nal_length_size = 4;                         // avcC lengthSizeMinusOne + 1
    unsigned width = 0, height = 0;
    uint32_t timescale = 0;
    uint64_t duration = 0;
    std::vector<NalRef> sps, pps;                     // inside the avcC box
    std::vector<Sample> samples;                      // decoding order
};

// Returns true when a video track with a sample table was found.
bool parse(const uint8_t *data, size_t size, VideoTrack &out, std::string &err);

} // namespace mp4
