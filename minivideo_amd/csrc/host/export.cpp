// export.cpp -- see export.h.  File formats restated from the reference's calls into
// stb_image_write v1.01 (bundled at minivideo/src/stb_image_write.h); BMP and TGA are
// byte-exact with what that library emits for 3-component input, PNG is pixel-exact
// (a valid PNG with stored deflate blocks; the reference's zlib stream differs).
#include "export.h"

#include <stdio.h>
#include <string.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#if defined(__x86_64__)
#include <tmmintrin.h>
#endif

#include <algorithm>
#include <vector>

namespace mvexport {

namespace {

struct File {
    FILE *f;
    bool bad = false;   // a short write (disk full, quota): the writers report it instead of leaving a truncated picture behind silently
    explicit File(const std::string &p) : f(fopen(p.c_str(), "wb"))
    {
        if (f) setvbuf(f, nullptr, _IOFBF, 1 << 20);   // (row-sized puts of a 6-MB picture: a system call per megabyte, not per row)
    }
    ~File() { if (f) fclose(f); }
    bool ok() const { return f != nullptr; }
    void put(const void *p, size_t n) { if (n && fwrite(p, 1, n, f) != n) bad = true; }
    int finish()   // 1: every byte reached the file
    {
        const bool closed = f && fclose(f) == 0;
        f = nullptr;
        return (closed && !bad) ? 1 : 0;
    }
    void u8(unsigned v) { uint8_t b = (uint8_t)v; put(&b, 1); }
    void u16(unsigned v) { u8(v & 255); u8((v >> 8) & 255); }
    void u32(unsigned v) { u16(v & 0xffff); u16(v >> 16); }
};

} // namespace

int write_yuv420(const std::string &path, const uint8_t *yuv, int width, int height)
{
    File o(path);
    if (!o.ok()) return 0;
    o.put(yuv, (size_t)width * height * 3 / 2);
    return o.finish();
}

int write_yuv444(const std::string &path, const uint8_t *yuv, int width, int height)
{
    File o(path);
    if (!o.ok()) return 0;
    const size_t n = (size_t)width * height;
    o.put(yuv, n);
    std::vector<uint8_t> up(n);
    for (int c = 0; c < 2; c++) {
        const uint8_t *src = yuv + n + (size_t)c * (n / 4);
        memset(up.data(), 0, n);
        for (int y = 0; y < height / 2; y++)
            for (int x = 0; x < width / 2; x++) {
                const uint8_t v = src[(size_t)y * (width / 2) + x];
                up[(size_t)(2 * y) * width + 2 * x] = v;
                up[(size_t)(2 * y) * width + 2 * x + 1] = v;
                up[(size_t)(2 * y + 1) * width + 2 * x] = v;
                // (2y+1, 2x+1) is never written by the reference (export.c:267-268): stays 0
            }
        o.put(up.data(), n);
    }
    return o.finish();
}

// R G B -> B G R for n pixels (dst and src do not overlap); five pixels per 16-byte shuffle where the CPU has SSSE3
namespace {
#if defined(__x86_64__)
__attribute__((target("ssse3"))) void swap_rb_ssse3(uint8_t *dst, const uint8_t *src, size_t n)
{
    const __m128i sh = _mm_setr_epi8(2, 1, 0, 5, 4, 3, 8, 7, 6, 11, 10, 9, 14, 13, 12, 15);
    size_t i = 0;
    for (; i + 6 <= n; i += 5) {   // (a 16-byte load / store stays inside the row: one pixel of slack)
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i * 3));
        _mm_storeu_si128(reinterpret_cast<__m128i *>(dst + i * 3), _mm_shuffle_epi8(v, sh));
    }
    for (; i < n; i++) { dst[i * 3] = src[i * 3 + 2]; dst[i * 3 + 1] = src[i * 3 + 1]; dst[i * 3 + 2] = src[i * 3]; }
}
#endif
void swap_rb(uint8_t *dst, const uint8_t *src, size_t n)
{
#if defined(__x86_64__)
    static const bool ssse3 = __builtin_cpu_supports("ssse3");
    if (ssse3) { swap_rb_ssse3(dst, src, n); return; }
#endif
    for (size_t i = 0; i < n; i++) { dst[i * 3] = src[i * 3 + 2]; dst[i * 3 + 1] = src[i * 3 + 1]; dst[i * 3 + 2] = src[i * 3]; }
}
} // namespace

// stbi_write_bmp, 3 components: bottom-up rows, B G R, rows padded to 4 bytes
int write_bmp(const std::string &path, const uint8_t *rgb, int width, int height)
{
    File o(path);
    if (!o.ok()) return 0;
    const int pad = (-width * 3) & 3;
    o.u8('B'); o.u8('M');
    o.u32(14 + 40 + (unsigned)(width * 3 + pad) * height);
    o.u16(0); o.u16(0);
    o.u32(14 + 40);
    o.u32(40); o.u32(width); o.u32(height);
    o.u16(1); o.u16(24);
    for (int i = 0; i < 6; i++) o.u32(0);
    std::vector<uint8_t> row((size_t)width * 3 + pad, 0);
    for (int y = height - 1; y >= 0; y--) {
        const uint8_t *s = rgb + (size_t)y * width * 3;
        swap_rb(row.data(), s, (size_t)width);
        o.put(row.data(), row.size());
    }
    return o.finish();
}

// TGA, type 10 (run-length encoded true colour), bottom row first -- the file stbi_write_tga() makes with its default
// stbi_write_tga_with_rle = 1 (export.c:726-733 calls it with 3 components).  Byte-exact output needs the library's packet
// boundaries, which follow from two per-pixel facts of a row, both found here in ONE backward pre-pass:
//   same[i]  how many pixels from i on equal pixel i                      -> a run packet at i covers min(same[i], 128) pixels
//   echo[i]  the first k >= i with pixel k == pixel k - 2 (or `width`)    -> a raw packet that starts at i (pixel i + 1
//            differs from pixel i) ends one pixel BEFORE the first such k >= i + 2 inside its 128-pixel window: the library
//            compares every new pixel with the one two back, not with its neighbour, and gives the last pixel back
// The last pixel of a row on its own is a raw packet of one.
int write_tga(const std::string &path, const uint8_t *rgb, int width, int height)
{
    File o(path);
    if (!o.ok()) return 0;
    o.u8(0); o.u8(0); o.u8(2 + 8);
    o.u16(0); o.u16(0); o.u8(0);
    o.u16(0); o.u16(0); o.u16(width); o.u16(height);
    o.u8(24); o.u8(0);
    std::vector<uint8_t> out;
    out.reserve((size_t)width * 4);
    std::vector<int> same((size_t)width + 2), echo((size_t)width + 3);
    for (int j = height - 1; j >= 0; j--) {
        const uint8_t *row = rgb + (size_t)j * width * 3;
        auto equal = [row](int p, int q) { return row[3 * p] == row[3 * q] && row[3 * p + 1] == row[3 * q + 1] && row[3 * p + 2] == row[3 * q + 2]; };
        same[width] = 0;
        echo[width] = echo[width + 1] = echo[width + 2] = width;
        for (int i = width - 1; i >= 0; i--) {
            same[i] = (i + 1 < width && equal(i, i + 1)) ? same[i + 1] + 1 : 1;
            echo[i] = (i >= 2 && equal(i, i - 2)) ? i : echo[i + 1];
        }
        out.clear();
        for (int i = 0; i < width;) {
            const int window = std::min(width - i, 128);
            if (same[i] >= 2) {                       // run packet: count - 129, then the pixel once (B G R)
                const int n = std::min(same[i], 128);
                out.push_back((uint8_t)(n - 129));
                out.push_back(row[3 * i + 2]); out.push_back(row[3 * i + 1]); out.push_back(row[3 * i]);
                i += n;
            } else {                                  // raw packet: count - 1, then the pixels
                const int k = echo[i + 2];
                const int n = (k < i + window) ? k - i - 1 : window;
                const size_t at = out.size();
                out.resize(at + 1 + (size_t)n * 3);
                out[at] = (uint8_t)(n - 1);
                swap_rb(&out[at + 1], row + (size_t)i * 3, (size_t)n);
                i += n;
            }
        }
        o.put(out.data(), out.size());
    }
    return o.finish();
}

// ---- PNG (RGB8, filter 0, stored deflate blocks) ----
// Written by several threads at once since round 3 (minivideo_decode's file writers) and the default format of
// mini_thumbnailer, so: the CRC tables are built when the library is loaded (no lazy flag to race on); CRC-32 runs eight
// bytes per step (slicing by 8), Adler-32 takes its modulo once per 5552 bytes instead of twice per byte, and the IDAT
// payload is assembled from the picture's rows in ONE pass.  43 -> 6 ms per 1080p picture on the container's core; the bytes
// of the file are what the byte-at-a-time writer produced.
namespace {
struct CrcTables {
    uint32_t t[8][256];
    CrcTables()
    {
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            t[0][n] = c;
        }
        for (uint32_t n = 0; n < 256; n++)
            for (int k = 1; k < 8; k++) t[k][n] = t[0][t[k - 1][n] & 255] ^ (t[k - 1][n] >> 8);
    }
};
const CrcTables g_crc;

uint32_t crc_update(uint32_t c, const uint8_t *p, size_t n)
{
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;   // (little-endian host: x86-64 / the only target of this library)
        c = g_crc.t[7][lo & 255] ^ g_crc.t[6][(lo >> 8) & 255] ^ g_crc.t[5][(lo >> 16) & 255] ^ g_crc.t[4][lo >> 24] ^
            g_crc.t[3][hi & 255] ^ g_crc.t[2][(hi >> 8) & 255] ^ g_crc.t[1][(hi >> 16) & 255] ^ g_crc.t[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    for (size_t i = 0; i < n; i++) c = g_crc.t[0][(c ^ p[i]) & 255] ^ (c >> 8);
    return c;
}

struct Adler {
    uint32_t a = 1, b = 0;
    void update(const uint8_t *p, size_t n)
    {
        while (n) {
            size_t m = n < 5552 ? n : 5552;   // the largest run for which b cannot overflow 32 bits (zlib's NMAX)
            n -= m;
#if defined(__SSE2__)
            // sixteen bytes per step: a += sum(p[i]), b += 16 * a_before + sum((16 - i) * p[i])
            const __m128i zero = _mm_setzero_si128();
            const __m128i w_lo = _mm_set_epi16(9, 10, 11, 12, 13, 14, 15, 16), w_hi = _mm_set_epi16(1, 2, 3, 4, 5, 6, 7, 8);
            while (m >= 16) {
                const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p));
                const __m128i sad = _mm_sad_epu8(v, zero);                                   // two 64-bit halves: byte sums
                const uint32_t s = (uint32_t)_mm_cvtsi128_si32(sad) + (uint32_t)_mm_extract_epi16(sad, 4);
                const __m128i lo = _mm_madd_epi16(_mm_unpacklo_epi8(v, zero), w_lo), hi = _mm_madd_epi16(_mm_unpackhi_epi8(v, zero), w_hi);
                __m128i w = _mm_add_epi32(lo, hi);
                w = _mm_add_epi32(w, _mm_shuffle_epi32(w, 0x4e));
                w = _mm_add_epi32(w, _mm_shuffle_epi32(w, 0xb1));
                b += 16 * a + (uint32_t)_mm_cvtsi128_si32(w);
                a += s;
                p += 16;
                m -= 16;
            }
#endif
            for (size_t i = 0; i < m; i++) { a += p[i]; b += a; }
            p += m;
            a %= 65521;
            b %= 65521;
        }
    }
    uint32_t value() const { return (b << 16) | a; }
};

void put_be32(uint8_t *d, uint32_t x) { d[0] = (uint8_t)(x >> 24); d[1] = (uint8_t)(x >> 16); d[2] = (uint8_t)(x >> 8); d[3] = (uint8_t)x; }

// length | type | data | CRC(type + data)
void chunk(File &o, const char *type, const uint8_t *data, size_t n)
{
    uint8_t w[4];
    put_be32(w, (uint32_t)n);
    o.put(w, 4);
    o.put(type, 4);
    if (n) o.put(data, n);
    uint32_t c = crc_update(0xffffffffu, (const uint8_t *)type, 4);
    c = crc_update(c, data, n) ^ 0xffffffffu;
    put_be32(w, c);
    o.put(w, 4);
}
} // namespace

int write_png(const std::string &path, const uint8_t *rgb, int width, int height)
{
    File o(path);
    if (!o.ok()) return 0;
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    o.put(sig, 8);
    uint8_t ihdr[13];
    put_be32(ihdr, (uint32_t)width);
    put_be32(ihdr + 4, (uint32_t)height);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk(o, "IHDR", ihdr, sizeof(ihdr));
    // IDAT: zlib header, the scanlines (filter byte 0 + the row) cut into stored blocks of at most 65535 bytes, Adler-32
    const size_t row = (size_t)width * 3, raw_size = (row + 1) * (size_t)height;
    const size_t n_blocks = raw_size ? (raw_size + 65534) / 65535 : 1;
    std::vector<uint8_t> z(2 + raw_size + 5 * n_blocks + 4);
    uint8_t *d = z.data();
    *d++ = 0x78; *d++ = 0x01;
    size_t left_in_block = 0, raw_left = raw_size;
    auto emit = [&](const uint8_t *p, size_t n) {   // n raw bytes; a block header wherever one is due
        while (n) {
            if (left_in_block == 0) {
                const size_t len = raw_left < 65535 ? raw_left : 65535;
                *d++ = (raw_left <= 65535) ? 1 : 0;
                *d++ = (uint8_t)(len & 255); *d++ = (uint8_t)(len >> 8);
                *d++ = (uint8_t)(~len & 255); *d++ = (uint8_t)((~len >> 8) & 255);
                left_in_block = len;
            }
            const size_t m = n < left_in_block ? n : left_in_block;
            memcpy(d, p, m);
            d += m; p += m; n -= m;
            left_in_block -= m;
            raw_left -= m;
        }
    };
    Adler ad;
    static const uint8_t filter0 = 0;
    if (raw_size == 0) { *d++ = 1; *d++ = 0; *d++ = 0; *d++ = 0xff; *d++ = 0xff; }   // (an empty picture: one empty final block)
    for (int y = 0; y < height; y++) {
        emit(&filter0, 1);
        emit(rgb + (size_t)y * row, row);
        ad.update(&filter0, 1);
        ad.update(rgb + (size_t)y * row, row);
    }
    put_be32(d, ad.value());
    d += 4;
    chunk(o, "IDAT", z.data(), (size_t)(d - z.data()));
    chunk(o, "IEND", nullptr, 0);
    return o.finish();
}

} // namespace mvexport
