// mp4_demux.cpp -- see mp4_demux.h.
#include "mp4_demux.h"

#include <string.h>

namespace mp4 {

namespace {

struct Reader {
    const uint8_t *d;
    size_t n;
    bool ok(size_t off, size_t len) const { return off <= n && len <= n - off; }
    uint32_t u8(size_t o) const { return d[o]; }
    uint32_t u16(size_t o) const { return ((uint32_t)d[o] << 8) | d[o + 1]; }
    uint32_t u32(size_t o) const { return ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]; }
    uint64_t u64(size_t o) const { return ((uint64_t)u32(o) << 32) | u32(o + 4); }
};

struct Box {
    uint32_t type = 0;
    size_t   start = 0, body = 0, end = 0; // header start, payload start, end (exclusive)
};

constexpr uint32_t fourcc(const char (&s)[5]) { return ((uint32_t)s[0] << 24) | ((uint32_t)s[1] << 16) | ((uint32_t)s[2] << 8) | (uint32_t)s[3]; }

// parse_box_header, mp4.c:633-692 (size 1 = 64-bit largesize, size 0 = to the end of the enclosing space)
bool next_box(const Reader &r, size_t pos, size_t limit, Box &b)
{
    if (pos + 8 > limit) return false;
    uint64_t size = r.u32(pos);
    b.type = r.u32(pos + 4);
    b.start = pos;
    b.body = pos + 8;
    if (size == 1) {
        if (pos + 16 > limit) return false;
        size = r.u64(pos + 8);
        b.body = pos + 16;
    } else if (size == 0) {
        size = limit - pos;
    }
    if (size < (uint64_t)(b.body - pos) || size > (uint64_t)(limit - pos)) return false;
    b.end = pos + (size_t)size;
    return true;
}

struct Tables {
    std::vector<uint32_t> stsz;            // per-sample sizes (or empty with constant size)
    uint32_t const_size = 0, sample_count = 0;
    std::vector<uint64_t> chunk_offset;    // stco / co64
    struct Stsc { uint32_t first_chunk, samples_per_chunk, desc; };
    std::vector<Stsc> stsc;
    std::vector<uint32_t> stss;            // 1-based sync sample numbers; empty box absent = every sample is sync
    bool have_stss = false;
};

void parse_avcC(const Reader &r, const Box &b, VideoTrack &t)
{
    size_t p = b.body;
    if (p + 6 > b.end) return;
    t.nal_length_size = (int)(r.u8(p + 4) & 3) + 1;
    const unsigned n_sps = r.u8(p + 5) & 31;
    p += 6;
    for (unsigned i = 0; i < n_sps && p + 2 <= b.end; i++) {
        const size_t len = r.u16(p);
        p += 2;
        if (p + len > b.end) return;
        NalRef n;
        n.offset = p;
        n.size = len;
        t.sps.push_back(n);
        p += len;
    }
    if (p + 1 > b.end) return;
    const unsigned n_pps = r.u8(p);
    p += 1;
    for (unsigned i = 0; i < n_pps && p + 2 <= b.end; i++) {
        const size_t len = r.u16(p);
        p += 2;
        if (p + len > b.end) return;
        NalRef n;
        n.offset = p;
        n.size = len;
        t.pps.push_back(n);
        p += len;
    }
    t.is_h264 = true;
}

void parse_stsd(const Reader &r, const Box &b, VideoTrack &t)
{
    if (b.body + 8 > b.end) return;
    const uint32_t entries = r.u32(b.body + 4);
    size_t p = b.body + 8;
    for (uint32_t e = 0; e < entries; e++) {
        Box entry;
        if (!next_box(r, p, b.end, entry)) return;
        if (entry.type == fourcc("avc1") || entry.type == fourcc("avc3")) {
            // VisualSampleEntry: 78 bytes of fixed fields, then child boxes (mp4.c:1700-1850)
            if (entry.body + 78 <= entry.end) {
                t.width = r.u16(entry.body + 24);
                t.height = r.u16(entry.body + 26);
                size_t q = entry.body + 78;
                Box child;
                while (next_box(r, q, entry.end, child)) {
                    if (child.type == fourcc("avcC")) parse_avcC(r, child, t);
                    q = child.end;
                }
            }
        }
        p = entry.end;
    }
}

void parse_stbl(const Reader &r, const Box &stbl, VideoTrack &t, Tables &tb)
{
    size_t p = stbl.body;
    Box b;
    while (next_box(r, p, stbl.end, b)) {
        const size_t q = b.body;
        if (b.type == fourcc("stsd")) parse_stsd(r, b, t);
        else if (b.type == fourcc("stsz") && q + 12 <= b.end) {
            tb.const_size = r.u32(q + 4);
            tb.sample_count = r.u32(q + 8);
            if (tb.const_size == 0) {
                const size_t avail = (b.end - (q + 12)) / 4;
                const size_t n = tb.sample_count < avail ? tb.sample_count : avail;
                tb.stsz.resize(n);
                for (size_t i = 0; i < n; i++) tb.stsz[i] = r.u32(q + 12 + 4 * i);
                tb.sample_count = (uint32_t)n;
            }
        } else if (b.type == fourcc("stco") && q + 8 <= b.end) {
            const size_t avail = (b.end - (q + 8)) / 4;
            size_t n = r.u32(q + 4);
            if (n > avail) n = avail;
            tb.chunk_offset.resize(n);
            for (size_t i = 0; i < n; i++) tb.chunk_offset[i] = r.u32(q + 8 + 4 * i);
        } else if (b.type == fourcc("co64") && q + 8 <= b.end) {
            const size_t avail = (b.end - (q + 8)) / 8;
            size_t n = r.u32(q + 4);
            if (n > avail) n = avail;
            tb.chunk_offset.resize(n);
            for (size_t i = 0; i < n; i++) tb.chunk_offset[i] = r.u64(q + 8 + 8 * i);
        } else if (b.type == fourcc("stsc") && q + 8 <= b.end) {
            const size_t avail = (b.end - (q + 8)) / 12;
            size_t n = r.u32(q + 4);
            if (n > avail) n = avail;
            tb.stsc.resize(n);
            for (size_t i = 0; i < n; i++)
                tb.stsc[i] = {r.u32(q + 8 + 12 * i), r.u32(q + 12 + 12 * i), r.u32(q + 16 + 12 * i)};
        } else if (b.type == fourcc("stss") && q + 8 <= b.end) {
            const size_t avail = (b.end - (q + 8)) / 4;
            size_t n = r.u32(q + 4);
            if (n > avail) n = avail;
            tb.stss.resize(n);
            for (size_t i = 0; i < n; i++) tb.stss[i] = r.u32(q + 8 + 4 * i);
            tb.have_stss = true;
        }
        p = b.end;
    }
}

// sample offsets from chunk offsets + sample-to-chunk runs (convertTrack, mp4.c:420-500)
void build_samples(const Tables &tb, size_t file_size, VideoTrack &t)
{
    t.samples.clear();
    if (tb.sample_count == 0 || tb.chunk_offset.empty() || tb.stsc.empty()) return;
    // a sample occupies at least one byte of the file: bound the count before trusting it for an allocation
    const uint32_t n_samples = (uint64_t)tb.sample_count < (uint64_t)file_size ? tb.sample_count : (uint32_t)file_size;
    t.samples.reserve(n_samples);
    uint32_t sample = 0;
    size_t run = 0;                                          // stsc runs are ordered by first_chunk
    for (size_t c = 0; c < tb.chunk_offset.size() && sample < n_samples; c++) {
        while (run + 1 < tb.stsc.size() && tb.stsc[run + 1].first_chunk >= 1 && (size_t)(tb.stsc[run + 1].first_chunk - 1) <= c) run++;
        const uint32_t per_chunk = tb.stsc[run].samples_per_chunk;
        uint64_t off = tb.chunk_offset[c];
        for (uint32_t i = 0; i < per_chunk && sample < n_samples; i++, sample++) {
            const uint32_t sz = tb.const_size ? tb.const_size : tb.stsz[sample];
            Sample s;
            s.offset = (size_t)off;
            s.size = sz;
            s.sync = !tb.have_stss;
            if (off > file_size || sz > file_size - off) { s.size = 0; }
            t.samples.push_back(s);
            off += sz;
        }
    }
    for (uint32_t n : tb.stss)
        if (n >= 1 && (size_t)(n - 1) < t.samples.size()) t.samples[n - 1].sync = true;
}

void parse_trak(const Reader &r, const Box &trak, VideoTrack &out)
{
    VideoTrack t;
    Tables tb;
    bool is_video = false;
    Box b;
    size_t p = trak.body;
    while (next_box(r, p, trak.end, b)) {
        if (b.type == fourcc("mdia")) {
            Box m;
            size_t q = b.body;
            while (next_box(r, q, b.end, m)) {
                if (m.type == fourcc("hdlr") && m.body + 12 <= m.end) is_video = r.u32(m.body + 8) == fourcc("vide");
                else if (m.type == fourcc("mdhd") && m.body + 4 <= m.end) {
                    const unsigned version = r.u8(m.body);
                    if (version == 1 && m.body + 32 <= m.end) { t.timescale = r.u32(m.body + 20); t.duration = r.u64(m.body + 24); }
                    else if (m.body + 20 <= m.end) { t.timescale = r.u32(m.body + 12); t.duration = r.u32(m.body + 16); }
                } else if (m.type == fourcc("minf")) {
                    Box f;
                    size_t w = m.body;
                    while (next_box(r, w, m.end, f)) {
                        if (f.type == fourcc("stbl")) parse_stbl(r, f, t, tb);
                        w = f.end;
                    }
                }
                q = m.end;
            }
        }
        p = b.end;
    }
    if (!is_video || out.found) return;
    build_samples(tb, r.n, t);
    t.found = true;
    out = t;
}

} // namespace

bool parse(const uint8_t *data, size_t size, VideoTrack &out, std::string &err)
{
    out = VideoTrack();
    Reader r{data, size};
    Box b;
    size_t p = 0;
    bool saw_moov = false;
    while (next_box(r, p, size, b)) {
        if (b.type == fourcc("moov")) {
            saw_moov = true;
            Box c;
            size_t q = b.body;
            while (next_box(r, q, b.end, c)) {
                if (c.type == fourcc("trak")) parse_trak(r, c, out);
                q = c.end;
            }
        }
        p = b.end;
    }
    if (!saw_moov) { err = "MP4: no moov box"; return false; }
    if (!out.found) { err = "MP4: no video track"; return false; }
    if (!out.is_h264) { err = "MP4: the video track is not H.264 (avc1/avcC)"; return false; }
    if (out.samples.empty()) { err = "MP4: empty sample table"; return false; }
    return true;
}

} // namespace mp4
