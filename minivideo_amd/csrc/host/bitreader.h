// bitreader.h -- MSB-first bit reader over an in-memory RBSP (replaces the
// file-backed reader of bitstream.c:382-539 and the Exp-Golomb readers of
// decoder/h264/h264_expgolomb.c:92-172 for the IDR decode path).
//
// Stateless apart from the bit position: every read is one unaligned big-endian 64-bit load at the current byte,
// shifted by the bit offset inside it (57 usable bits >= the 32 a read may ask for).  The last 7 bytes of the buffer
// take a byte-wise path; reads past the end return zero bits and are detected through the position (overrun()).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace h264 {

class BitReader {
public:
    BitReader() : p_(nullptr), n_bytes_(0), n_bits_(0), pos_(0) {}
    BitReader(const uint8_t *p, size_t n_bytes) : p_(p), n_bytes_(n_bytes), n_bits_(n_bytes * 8), pos_(0) {}

    size_t pos() const { return pos_; }
    size_t size_bits() const { return n_bits_; }
    size_t bits_left() const { return pos_ < n_bits_ ? n_bits_ - pos_ : 0; }
    bool   overrun() const { return pos_ > n_bits_; }
    bool   byte_aligned() const { return (pos_ & 7) == 0; }
    const uint8_t *data() const { return p_; }

    // next n bits (n <= 32) without consuming them; zero-padded past the end
    uint32_t peek(int n) const { return n ? (uint32_t)(window() >> (64 - n)) : 0u; }
    void skip(size_t n) { pos_ += n; }
    uint32_t bits(int n) // n <= 32
    {
        const uint32_t v = peek(n);
        pos_ += (size_t)n;
        return v;
    }
    uint32_t bit()
    {
        const size_t byte = pos_ >> 3;
        const uint32_t v = byte < n_bytes_ ? (uint32_t)(p_[byte] >> (7 - (pos_ & 7))) & 1u : 0u;
        pos_++;
        return v;
    }
    void seek(size_t bitpos) { pos_ = bitpos; }

    // number of leading zero bits in the next 32 bits (32 if they are all zero)
    int leading_zeros32() const
    {
        const uint32_t v = peek(32);
        return v ? __builtin_clz(v) : 32;
    }

    // ue(v), 9.1 (h264_expgolomb.c:92)
    uint32_t ue()
    {
        const int lz = leading_zeros32();
        if (lz >= 32 || overrun()) { skip(32); return 0xffffffffu; }
        skip((size_t)lz + 1);
        if (lz == 0) return 0;
        return ((1u << lz) - 1u) + bits(lz);
    }
    // se(v), 9.1.1 (h264_expgolomb.c:107)
    int32_t se()
    {
        const uint32_t k = ue();
        if (k == 0xffffffffu) return 0;
        return (k & 1) ? (int32_t)((k + 1) >> 1) : -(int32_t)(k >> 1);
    }

    // more_rbsp_data(), 7.2: true while there is something before the
    // rbsp_stop_one_bit (the last 1 bit of the RBSP).
    bool more_rbsp_data() const
    {
        if (pos_ >= n_bits_) return false;
        size_t nb = n_bytes_;
        while (nb > 0 && p_[nb - 1] == 0) nb--;
        if (nb == 0) return false;
        const uint8_t b = p_[nb - 1];
        int tz = 0;
        while (((b >> tz) & 1) == 0) tz++;
        const size_t last = (nb - 1) * 8 + (size_t)(7 - tz); // bit position of the stop bit
        return pos_ < last;
    }

    // the bits from the current position on, left-aligned (at least 57 of them valid)
    uint64_t window() const
    {
        const size_t byte = pos_ >> 3;
        uint64_t w;
        if (byte + 8 <= n_bytes_) {
            memcpy(&w, p_ + byte, 8);
            w = __builtin_bswap64(w);
        } else {
            w = 0;
            for (size_t i = byte; i < n_bytes_; i++) w |= (uint64_t)p_[i] << (56 - 8 * (i - byte));
        }
        return w << (pos_ & 7);
    }

private:
    const uint8_t *p_;
    size_t n_bytes_, n_bits_;
    size_t pos_;
};

// A local view for hot loops: the 64-bit window is fetched once and consumed from a register; it is fetched again only
// when a read would run past its 57 valid bits.  The reader's position is brought up to date when the view ends.
class BitWindow {
public:
    explicit BitWindow(BitReader &br) : br_(br), w_(br.window()), used_(0) {}
    ~BitWindow() { br_.skip((size_t)used_); }
    BitWindow(const BitWindow &) = delete;
    BitWindow &operator=(const BitWindow &) = delete;
    uint32_t peek(int n)   // n <= 32
    {
        if (used_ + n > 57) { br_.skip((size_t)used_); w_ = br_.window(); used_ = 0; }
        return n ? (uint32_t)((w_ << used_) >> (64 - n)) : 0u;
    }
    void skip(int n) { used_ += n; }   // (behind a peek of at least n bits, or up to 32 bits further)
    uint32_t bits(int n) { const uint32_t v = peek(n); used_ += n; return v; }
    uint32_t bit() { return bits(1); }
    int leading_zeros32() { const uint32_t v = peek(32); return v ? __builtin_clz(v) : 32; }
    bool overrun() const { return br_.pos() + (size_t)used_ > br_.size_bits(); }

private:
    BitReader &br_;
    uint64_t w_;
    int used_;
};

} // namespace h264
