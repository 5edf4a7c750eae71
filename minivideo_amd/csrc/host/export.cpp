// export.cpp -- see export.h.  File formats restated from the reference's calls into
// stb_image_write v1.01 (bundled at minivideo/src/stb_image_write.h); BMP and TGA are
// byte-exact with what that library emits for 3-component input, PNG is pixel-exact
// (a valid PNG with stored deflate blocks; the reference's zlib stream differs).
#include "export.h"

#include <stdio.h>
#include <string.h>

#include <vector>

namespace mvexport {

namespace {

struct File {
    FILE *f;
    explicit File(const std::string &p) : f(fopen(p.c_str(), "wb")) {}
    ~File() { if (f) fclose(f); }
    bool ok() const { return f != nullptr; }
    void put(const void *p, size_t n) { fwrite(p, 1, n, f); }
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
    return 1;
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
    return 1;
}

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
        for (int x = 0; x < width; x++) { row[x * 3] = s[x * 3 + 2]; row[x * 3 + 1] = s[x * 3 + 1]; row[x * 3 + 2] = s[x * 3]; }
        o.put(row.data(), row.size());
    }
    return 1;
}

// stbi_write_tga with stbi_write_tga_with_rle = 1 (the library default), 3 components
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
    auto same = [](const uint8_t *a, const uint8_t *b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; };
    for (int j = height - 1; j >= 0; j--) {
        const uint8_t *row = rgb + (size_t)j * width * 3;
        out.clear();
        int len;
        for (int i = 0; i < width; i += len) {
            const uint8_t *begin = row + (size_t)i * 3;
            bool diff = true;
            len = 1;
            if (i < width - 1) {
                ++len;
                diff = !same(begin, row + (size_t)(i + 1) * 3);
                if (diff) {
                    // raw packet: extend while pixel k differs from pixel k-2 (the library's own rule)
                    const uint8_t *prev = begin;
                    for (int k = i + 2; k < width && len < 128; ++k) {
                        if (!same(prev, row + (size_t)k * 3)) { prev += 3; ++len; }
                        else { --len; break; }
                    }
                } else {
                    for (int k = i + 2; k < width && len < 128; ++k) {
                        if (same(begin, row + (size_t)k * 3)) ++len; else break;
                    }
                }
            }
            if (diff) {
                out.push_back((uint8_t)(len - 1));
                for (int k = 0; k < len; ++k) { const uint8_t *p = begin + k * 3; out.push_back(p[2]); out.push_back(p[1]); out.push_back(p[0]); }
            } else {
                out.push_back((uint8_t)(len - 129));
                out.push_back(begin[2]); out.push_back(begin[1]); out.push_back(begin[0]);
            }
        }
        o.put(out.data(), out.size());
    }
    return 1;
}

// ---- PNG (RGB8, filter 0, stored deflate blocks) ----
namespace {
uint32_t crc_table[256];
bool crc_ready = false;
void crc_init()
{
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
    crc_ready = true;
}
uint32_t crc_update(uint32_t c, const uint8_t *p, size_t n)
{
    for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 255] ^ (c >> 8);
    return c;
}
void be32(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x >> 24); v.push_back((x >> 16) & 255); v.push_back((x >> 8) & 255); v.push_back(x & 255); }
void chunk(File &o, const char *type, const std::vector<uint8_t> &data)
{
    std::vector<uint8_t> hdr;
    be32(hdr, (uint32_t)data.size());
    o.put(hdr.data(), 4);
    o.put(type, 4);
    if (!data.empty()) o.put(data.data(), data.size());
    uint32_t c = crc_update(0xffffffffu, (const uint8_t *)type, 4);
    c = crc_update(c, data.data(), data.size()) ^ 0xffffffffu;
    std::vector<uint8_t> t;
    be32(t, c);
    o.put(t.data(), 4);
}
} // namespace

int write_png(const std::string &path, const uint8_t *rgb, int width, int height)
{
    if (!crc_ready) crc_init();
    File o(path);
    if (!o.ok()) return 0;
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    o.put(sig, 8);
    std::vector<uint8_t> ihdr;
    be32(ihdr, (uint32_t)width);
    be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(o, "IHDR", ihdr);
    // raw scanlines with filter byte 0
    const size_t stride = (size_t)width * 3 + 1;
    std::vector<uint8_t> raw(stride * height);
    for (int y = 0; y < height; y++) {
        raw[y * stride] = 0;
        memcpy(&raw[y * stride + 1], rgb + (size_t)y * width * 3, (size_t)width * 3);
    }
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    size_t pos = 0;
    while (pos < raw.size() || raw.empty()) {
        const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        const bool last = pos + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back(n & 255); z.push_back((n >> 8) & 255);
        z.push_back((~n) & 255); z.push_back(((~n) >> 8) & 255);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; i++) { a = (a + raw[pos + i]) % 65521; b = (b + a) % 65521; }
        pos += n;
        if (last) break;
    }
    be32(z, (b << 16) | a);
    chunk(o, "IDAT", z);
    chunk(o, "IEND", std::vector<uint8_t>());
    return 1;
}

} // namespace mvexport
