// range_coder.cpp — host-side binary arithmetic coder for the ±1 embeddings (libcnc_codec.so).
// See include/cnc_codec.h for the interface it replaces and the bitstream definition.
//
// Structure: the interval update is the classic 32-bit low/high scheme; output bits are packed
// through a 64-bit accumulator and pending (underflow) bits are emitted as runs, so the inner
// loop does no per-bit byte handling.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "cnc_codec.h"

namespace {

constexpr uint32_t kTop = 0x80000000u, kQ1 = 0x40000000u, kQ3 = 0xC0000000u;

inline uint32_t cdf_one(float p_one)
{
    // float32 arithmetic exactly as torch does it: (1 - p) * 65534, round half to even, + 1
    const float u = 1.0f - p_one;
    const float r = std::nearbyintf(u * 65534.0f);
    return (static_cast<uint32_t>(static_cast<int32_t>(r)) + 1u) & 0xFFFFu;
}

class BitSink {
public:
    BitSink(uint8_t* buf, int64_t cap) : buf_(buf), cap_(cap) {}

    // `count` copies of `bit`
    void run(uint32_t bit, uint64_t count)
    {
        while (count > 0) {
            const unsigned room = 64 - fill_;
            const unsigned take = count < room ? static_cast<unsigned>(count) : room;
            if (take == 64) acc_ = bit ? ~0ull : 0ull;
            else acc_ = (acc_ << take) | (bit ? ((1ull << take) - 1ull) : 0ull);
            fill_ += take;
            count -= take;
            if (fill_ == 64) drain();
        }
    }

    void bit_then_pending(uint32_t bit, uint64_t& pending)
    {
        run(bit, 1);
        run(bit ^ 1u, pending);
        pending = 0;
    }

    int64_t finish()
    {
        // flush whole bytes, then zero-pad the last partial byte
        while (fill_ >= 8) {
            put(static_cast<uint8_t>(acc_ >> (fill_ - 8)));
            fill_ -= 8;
        }
        if (fill_ > 0) {
            put(static_cast<uint8_t>((acc_ << (8 - fill_)) & 0xFF));
            fill_ = 0;
        }
        return overflow_ ? -1 : n_;
    }

private:
    void drain()
    {
        for (int s = 56; s >= 0; s -= 8) put(static_cast<uint8_t>(acc_ >> s));
        fill_ = 0;
        acc_ = 0;
    }
    void put(uint8_t b)
    {
        if (n_ < cap_) buf_[n_] = b; else overflow_ = true;
        ++n_;
    }
    uint8_t* buf_;
    int64_t  cap_;
    int64_t  n_ = 0;
    uint64_t acc_ = 0;
    unsigned fill_ = 0;
    bool     overflow_ = false;
};

class BitSource {
public:
    BitSource(const uint8_t* buf, int64_t len) : buf_(buf), len_(len) {}
    // next bit, 0 past the end of the stream
    uint32_t next()
    {
        if (left_ == 0) {
            if (pos_ >= len_) return 0;
            cur_ = buf_[pos_++];
            left_ = 8;
        }
        --left_;
        return (cur_ >> left_) & 1u;
    }
private:
    const uint8_t* buf_;
    int64_t        len_;
    int64_t        pos_ = 0;
    uint8_t        cur_ = 0;
    int            left_ = 0;
};

struct Interval {
    uint32_t low = 0, high = 0xFFFFFFFFu;
    void narrow(uint32_t c_lo, uint32_t c_hi)
    {
        const uint64_t span = static_cast<uint64_t>(high) - low + 1;
        high = low - 1 + static_cast<uint32_t>((span * c_hi) >> 16);
        low = low + static_cast<uint32_t>((span * c_lo) >> 16);
    }
};

}  // namespace

extern "C" int64_t cnc_rc_bound(int64_t n) { return 2 * n + 16; }

extern "C" int64_t cnc_rc_encode_pm1(const float* p_one, const float* x_pm1, int64_t n,
                                     uint8_t* out, int64_t cap)
{
    BitSink  sink(out, cap);
    Interval iv;
    uint64_t pending = 0;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t c1 = cdf_one(p_one[i]);
        if (x_pm1[i] > 0) iv.narrow(c1, 0x10000u); else iv.narrow(0u, c1);
        for (;;) {
            if (iv.high < kTop) {
                sink.bit_then_pending(0, pending);
            } else if (iv.low >= kTop) {
                sink.bit_then_pending(1, pending);
            } else if (iv.low >= kQ1 && iv.high < kQ3) {
                ++pending;
                iv.low &= ~kQ1;          // after the shift below: low = (low<<1) & 0x7FFFFFFF
                iv.high |= kQ1;          //                        high = (high<<1) | 0x80000001
            } else {
                break;
            }
            iv.low <<= 1;
            iv.high = (iv.high << 1) | 1u;
        }
    }
    ++pending;
    sink.bit_then_pending(iv.low < kQ1 ? 0u : 1u, pending);
    return sink.finish();
}

// ---- general alphabets: torchac's `encode_int16_normalized_cdf` / `decode_int16_normalized_cdf` ----------------
// cdf: [n, Lp] 16-bit integers, cdf[i][0] = 0 <= cdf[i][1] <= ... ; the last entry stands for 2^16 whatever it
// holds (torchac stores 65536 wrapped to 0 there); symbols 0 .. Lp-2.
extern "C" int64_t cnc_rc_encode_cdf16(const uint16_t* cdf, const int16_t* sym, int64_t n, int32_t Lp,
                                       uint8_t* out, int64_t cap)
{
    if (Lp < 2) return -2;
    BitSink  sink(out, cap);
    Interval iv;
    uint64_t pending = 0;
    const int32_t last = Lp - 2;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t s = sym[i];
        if (s < 0 || s > last) return -2;
        const uint16_t* c = cdf + i * Lp;
        iv.narrow(c[s], s == last ? 0x10000u : static_cast<uint32_t>(c[s + 1]));
        for (;;) {
            if (iv.high < kTop) {
                sink.bit_then_pending(0, pending);
            } else if (iv.low >= kTop) {
                sink.bit_then_pending(1, pending);
            } else if (iv.low >= kQ1 && iv.high < kQ3) {
                ++pending;
                iv.low &= ~kQ1;
                iv.high |= kQ1;
            } else {
                break;
            }
            iv.low <<= 1;
            iv.high = (iv.high << 1) | 1u;
        }
    }
    ++pending;
    sink.bit_then_pending(iv.low < kQ1 ? 0u : 1u, pending);
    return sink.finish();
}

extern "C" int cnc_rc_decode_cdf16(const uint16_t* cdf, int64_t n, int32_t Lp, const uint8_t* in, int64_t len,
                                   int16_t* sym)
{
    if (Lp < 2) return -2;
    BitSource src(in, len);
    Interval  iv;
    uint32_t  value = 0;
    const int32_t last = Lp - 2;
    for (int i = 0; i < 32; ++i) value = (value << 1) | src.next();
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t span = static_cast<uint64_t>(iv.high) - iv.low + 1;
        const uint32_t count = static_cast<uint32_t>(
            (((static_cast<uint64_t>(value) - iv.low + 1) << 16) - 1) / span) & 0xFFFFu;
        const uint16_t* c = cdf + i * Lp;
        // the largest s in [0, last] with c[s] <= count (c[0] = 0): bisection over the row
        int32_t lo = 0, hi = last;
        while (lo < hi) {
            const int32_t mid = (lo + hi + 1) >> 1;
            if (c[mid] <= count) lo = mid; else hi = mid - 1;
        }
        sym[i] = static_cast<int16_t>(lo);
        if (i == n - 1) break;
        iv.narrow(c[lo], lo == last ? 0x10000u : static_cast<uint32_t>(c[lo + 1]));
        for (;;) {
            if (iv.low >= kTop || iv.high < kTop) {
            } else if (iv.low >= kQ1 && iv.high < kQ3) {
                iv.low &= ~kQ1;
                iv.high |= kQ1;
                value -= kQ1;
            } else {
                break;
            }
            iv.low <<= 1;
            iv.high = (iv.high << 1) | 1u;
            value = (value << 1) | src.next();
        }
    }
    return 0;
}

extern "C" int cnc_rc_decode_pm1(const float* p_one, int64_t n, const uint8_t* in, int64_t len,
                                 float* x_pm1)
{
    BitSource src(in, len);
    Interval  iv;
    uint32_t  value = 0;
    for (int i = 0; i < 32; ++i) value = (value << 1) | src.next();
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t span = static_cast<uint64_t>(iv.high) - iv.low + 1;
        const uint32_t count = static_cast<uint32_t>(
            (((static_cast<uint64_t>(value) - iv.low + 1) << 16) - 1) / span) & 0xFFFFu;
        const uint32_t c1 = cdf_one(p_one[i]);
        const bool     one = c1 <= count;
        x_pm1[i] = one ? 1.0f : -1.0f;
        if (i == n - 1) break;
        if (one) iv.narrow(c1, 0x10000u); else iv.narrow(0u, c1);
        for (;;) {
            if (iv.low >= kTop || iv.high < kTop) {
                // nothing to subtract: the shift drops the common top bit
            } else if (iv.low >= kQ1 && iv.high < kQ3) {
                iv.low &= ~kQ1;
                iv.high |= kQ1;
                value -= kQ1;
            } else {
                break;
            }
            iv.low <<= 1;
            iv.high = (iv.high << 1) | 1u;
            value = (value << 1) | src.next();
        }
    }
    return 0;
}
