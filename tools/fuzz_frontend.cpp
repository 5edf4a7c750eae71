// tools/fuzz_frontend.cpp -- sanitizer harness for the host front end (CPU only):
//   g++ -O1 -g -fsanitize=address,undefined -std=c++17 -Iinclude -Iminivideo_amd/csrc/host \
//       tools/fuzz_frontend.cpp minivideo_amd/csrc/host/{h264_frontend,h264_cabac,stream_abi,mp4_demux}.cpp -o /tmp/fuzz_frontend
//   /tmp/fuzz_frontend stream.264|clip.mp4 [iterations] [spec]   (a .mp4/.mov name goes through mvhp_stream_open_mp4; `spec` opens
//   the stream with MVHP_STREAM_SPEC)
// Mutates the stream (bit flips, byte splats, truncations) and parses every picture; any memory error aborts.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "minivideo_hotpath.h"

static uint64_t rng_state = 0x1234567;
static uint64_t rnd()
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> base;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) base.insert(base.end(), buf, buf + n);
    fclose(f);
    const size_t nl = strlen(argv[1]);
    const bool is_mp4 = nl > 4 && (!strcmp(argv[1] + nl - 4, ".mp4") || !strcmp(argv[1] + nl - 4, ".mov"));
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    const bool spec = argc > 3 && !strcmp(argv[3], "spec");   // MVHP_STREAM_SPEC: several slices, scaling lists, I_PCM are parsed
    long ok = 0, bad = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> d = base;
        const int kind = (int)(rnd() % 4);
        const int edits = 1 + (int)(rnd() % 8);
        for (int e = 0; e < edits; e++) {
            const size_t pos = (size_t)(rnd() % d.size());
            if (kind == 0) d[pos] ^= (uint8_t)(1u << (rnd() % 8));
            else if (kind == 1) d[pos] = (uint8_t)rnd();
            else if (kind == 2) { const size_t len = 1 + (size_t)(rnd() % 16); for (size_t i = pos; i < pos + len && i < d.size(); i++) d[i] = 0xff; }
            else { d.resize(pos + 1); d.insert(d.end(), 64, 0); break; }
        }
        mvhp_stream_t *s = nullptr;
        if ((is_mp4 ? mvhp_stream_open_mp4(d.data(), d.size(), &s)
                    : spec ? mvhp_stream_open_ex(d.data(), d.size(), MVHP_STREAM_SPEC, &s) : mvhp_stream_open(d.data(), d.size(), &s)) != MVHP_SUCCESS) continue;
        const int cnt = mvhp_stream_idr_count(s);
        for (int k = 0; k < cnt; k++) {
            mvhp_stream_params_t p;
            if (mvhp_stream_params(s, k, &p) != MVHP_SUCCESS) continue;
            std::vector<uint8_t> packed((size_t)p.width_mbs * p.height_mbs * MVHP_MB_BYTES);
            if (mvhp_stream_decode_packed(s, k, packed.data(), packed.size()) == MVHP_SUCCESS) ok++; else bad++;
            // the transfer format too (its own writer: offsets table, entry lists, dense fallback, I_PCM records)
            std::vector<uint8_t> compact((size_t)p.width_mbs * p.height_mbs * MVHP_COMPACT_MB_BYTES_MAX + MVHP_COMPACT_SLACK_BYTES);
            size_t used = 0;
            (void)mvhp_stream_decode_compact(s, k, compact.data(), compact.size(), &used);
            if (used > compact.size()) { fprintf(stderr, "compact picture overran its buffer\n"); return 1; }
        }
        mvhp_stream_close(s);
    }
    printf("fuzz: %d iterations, %ld pictures parsed, %ld rejected, no memory errors\n", iters, ok, bad);
    return 0;
}
