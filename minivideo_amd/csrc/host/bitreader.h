// bitreader.h -- MSB-first bit reader over an in-memory RBSP (replaces the
// file-backed reader of bitstream.c:382-539 and the Exp-Golomb readers of
// decoder/h264/h264_expgolomb.c:92-172 for the IDR decode path).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace h264 {

class BitReader {
public:
    BitReader() : p_(nullptr), n_bits_(0), pos_(0) {}
    BitReader(const uint8_t *p, size_t n_bytes) : p_(p), n_bits_(n_bytes * 8), pos_(0) {}

    size_t pos() const { return pos_; }
    size_t size_bits() const { return n_bits_; }
    size_t bits_left() const { return pos_ < n_bits_ ? n_bits_ - pos_ : 0; }
    bool   overrun() const { return pos_ > n_bits_; }
    bool   byte_aligned() const { return (pos_ & 7) == 0; }
    const uint8_t *data() const { return p_; }

    // Reads past the end return zero bits and set overrun().
    uint32_t bit()
    {
        uint32_t v = 0;
        if (pos_ < n_bits_) v = (p_[pos_ >> 3] >> (7 - (pos_ & 7))) & 1u;
        pos_++;
        return v;
    }
    uint32_t bits(int n) // n <= 32
    {
        uint32_t v = 0;
        if (n > 0 && pos_ + (size_t)n <= n_bits_) {
            // fast path: gather up to 5 bytes
            size_t byte = pos_ >> 3;
            int off = (int)(pos_ & 7);
            uint64_t acc = 0;
            int need = (off + n + 7) >> 3;
            for (int i = 0; i < need; i++) acc = (acc << 8) | p_[byte + i];
            acc >>= (need * 8 - off - n);
            v = (uint32_t)(acc & ((n == 32) ? 0xffffffffull : ((1ull << n) - 1)));
            pos_ += n;
            return v;
        }
        for (int i = 0; i < n; i++) v = (v << 1) | bit();
        return v;
    }
    uint32_t peek(int n)
    {
        size_t save = pos_;
        uint32_t v = bits(n);
        pos_ = save;
        return v;
    }
    void skip(size_t n) { pos_ += n; }
    void seek(size_t bitpos) { pos_ = bitpos; }

    // ue(v), 9.1 (h264_expgolomb.c:92)
    uint32_t ue()
    {
        int lz = 0;
        while (bit() == 0) {
            if (++lz > 32 || overrun()) return 0xffffffffu;
        }
        if (lz == 0) return 0;
        if (lz == 32) return 0xffffffffu;
        return ((1u << lz) - 1u) + bits(lz);
    }
    // se(v), 9.1.1 (h264_expgolomb.c:107)
    int32_t se()
    {
        uint32_t k = ue();
        if (k == 0xffffffffu) return 0;
        return (k & 1) ? (int32_t)((k + 1) >> 1) : -(int32_t)(k >> 1);
    }

    // more_rbsp_data(), 7.2: true while there is something before the
    // rbsp_stop_one_bit (the last 1 bit of the RBSP).
    bool more_rbsp_data() const
    {
        if (pos_ >= n_bits_) return false;
        // find last set bit
        size_t last = n_bits_;
        size_t nb = n_bits_ >> 3;
        while (nb > 0 && p_[nb - 1] == 0) nb--;
        if (nb == 0) return false;
        uint8_t b = p_[nb - 1];
        int tz = 0;
        while (((b >> tz) & 1) == 0) tz++;
        last = (nb - 1) * 8 + (7 - tz); // bit position of the stop bit
        return pos_ < last;
    }

private:
    const uint8_t *p_;
    size_t n_bits_;
    size_t pos_;
};

} // namespace h264
