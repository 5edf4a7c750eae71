// tools/index_check.cpp -- TEST TOOL (CPU): the Annex-B index of the product (h264::index_annexb, which hops between 0x01
// bytes with memchr) against the byte-at-a-time statement of the reference's scan (esparser.c:40-143: a zero counter, a
// sample behind >= 3 zero bytes + 0x01 when the next byte is 0x65 / 0x67 / 0x68, the scan stopping 32 bytes before the end;
// the NAL unit ending at the next 00 00 01 with trailing zero bytes trimmed), on random strings over a small alphabet that
// is dense in start codes, partial start codes and runs of zeros, of every length from 0 up.
//   g++ -O2 -std=c++17 -Iinclude -Iminivideo_amd/csrc/host tools/index_check.cpp minivideo_amd/csrc/host/h264_frontend.cpp \
//       minivideo_amd/csrc/host/h264_cabac.cpp -o /tmp/index_check && /tmp/index_check [cases]
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "h264_frontend.h"

struct Ref { size_t offset, sample_size, nal_size; int type; };

static bool by_bytes(const uint8_t *data, size_t size, std::vector<Ref> &out)
{
    out.clear();
    const long long limit = (long long)size - 32;
    long long off = 0;
    int zeros = 0;
    while (off < limit) {
        const uint8_t b = data[off++];
        if (b == 0) { zeros++; continue; }
        if (b == 1 && zeros > 2) {
            const uint8_t nb = data[off];
            if (nb == 0x65 || nb == 0x67 || nb == 0x68) {
                if (!out.empty()) out.back().sample_size = (size_t)off - out.back().offset;
                out.push_back(Ref{(size_t)off, 0, 0, nb & 31});
            }
        }
        zeros = 0;
    }
    if (out.empty()) return false;
    out.back().sample_size = size - out.back().offset;
    for (Ref &r : out) {
        const size_t beg = r.offset, lim = beg + r.sample_size;
        size_t end = lim;
        for (size_t p = beg + 1; p + 2 < lim; p++)
            if (data[p] == 0 && data[p + 1] == 0 && data[p + 2] == 1) { end = p; break; }
        while (end > beg + 1 && data[end - 1] == 0) end--;
        r.nal_size = end - beg;
    }
    return true;
}

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 200000;
    static const uint8_t alphabet[] = {0, 0, 0, 0, 1, 1, 0x65, 0x67, 0x68, 0x03, 0x41, 0xff};
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&] { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    long samples = 0;
    for (int c = 0; c < cases; c++) {
        const size_t n = (c < 200) ? (size_t)c : (size_t)(rnd() % 400);
        std::vector<uint8_t> d(n);
        const int zero_bias = (int)(rnd() % 3);
        for (auto &b : d) b = (zero_bias == 2 && rnd() % 4 == 0) ? (uint8_t)rnd() : alphabet[rnd() % (zero_bias ? sizeof(alphabet) : 9)];
        std::vector<Ref> want;
        std::vector<h264::EsSample> got;
        const bool w = by_bytes(d.data(), n, want);
        const bool g = h264::index_annexb(d.data(), n, got) == h264::RC_SUCCESS;
        bool same = (w == g) && (!w || want.size() == got.size());
        for (size_t i = 0; same && w && i < want.size(); i++)
            same = want[i].offset == got[i].offset && want[i].sample_size == got[i].sample_size && want[i].nal_size == got[i].nal_size &&
                   want[i].type == got[i].nal_unit_type && got[i].is_idr == (want[i].type == 5);
        if (!same) {
            fprintf(stderr, "case %d (length %zu): the index differs from the byte-at-a-time scan\n", c, n);
            return 1;
        }
        samples += (long)want.size();
    }
    printf("index_check: %d strings, %ld samples, identical\n", cases, samples);
    return 0;
}
