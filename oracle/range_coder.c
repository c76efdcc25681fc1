/*
 * range_coder.c — TEST INFRASTRUCTURE ONLY (part of the CPU oracle).
 *
 * The reference codes the ±1 embeddings with the third-party package torchac==0.9.3
 * (requirements.txt:32; call sites examples/utils_bpp_acc.py:87 encode_float_cdf and :108
 * decode_float_cdf).  torchac is NOT vendored under the reference tree and is not installed in
 * this image, so this file restates its published algorithm:
 *   - float CDF -> 16-bit integer CDF:  c = round(cdf * (2^16 - (Lp-1))) + arange(Lp)   (Lp = 3)
 *     i.e. for the CNC binary alphabet  c1 = round_half_even((1 - p) * 65534) + 1,
 *     c0 = 0, c2 = 2^16 (hard-wired for the last symbol);
 *   - 32-bit low/high interval coder with pending (underflow) bits, MSB-first bit packing,
 *     one terminating bit + pending, zero-padded to a byte.
 * PARITY UNPINNED at the byte level: no torchac binary or bitstream fixture exists to check
 * against.  What is held: exact encode->decode round trips and size vs. the entropy estimate.
 */
#include <math.h>
#include <stdint.h>

/* is_u != 0: the caller passes cdf[...,1] = 1 - p directly (what torchac itself receives) */
static int g_unused;
static inline uint32_t orc_cdf1_(float prob, int is_u)
{
    /* torch: p_u = 1 - p (float32); cdf.mul(65534.0f).round() (half-to-even); +1 (arange) */
    float u = is_u ? prob : 1.0f - prob;
    float s = u * 65534.0f;
    float r = nearbyintf(s);              /* default rounding mode = half-to-even == torch.round */
    return ((uint32_t)(int32_t)r + 1u) & 0xFFFFu;
}

typedef struct { uint8_t* buf; int64_t cap; int64_t n; uint8_t cache; int count; int overflow; } orc_bw_t;

static void bw_put(orc_bw_t* w, int bit)
{
    w->cache = (uint8_t)((w->cache << 1) | (bit & 1));
    if (++w->count == 8) {
        if (w->n < w->cap) w->buf[w->n] = w->cache; else w->overflow = 1;
        w->n++;
        w->count = 0;
        w->cache = 0;
    }
}

static void bw_put_pending(orc_bw_t* w, int bit, uint64_t* pending)
{
    bw_put(w, bit);
    while (*pending > 0) { bw_put(w, !bit); (*pending)--; }
}

/* returns number of bytes, or -1 if `cap` was too small */
int64_t orc_rc_encode2(const float* p_one, int is_u, const int16_t* sym, int64_t n, uint8_t* out, int64_t cap)
{
    orc_bw_t w = {out, cap, 0, 0, 0, 0};
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint64_t pending = 0;
    for (int64_t i = 0; i < n; i++) {
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        const uint32_t c1 = orc_cdf1_(p_one[i], is_u);
        const uint32_t c_low = sym[i] ? c1 : 0u;
        const uint32_t c_high = sym[i] ? 0x10000u : c1;
        high = (low - 1) + (uint32_t)((span * (uint64_t)c_high) >> 16);
        low = low + (uint32_t)((span * (uint64_t)c_low) >> 16);
        for (;;) {
            if (high < 0x80000000u) {
                bw_put_pending(&w, 0, &pending);
                low <<= 1; high <<= 1; high |= 1;
            } else if (low >= 0x80000000u) {
                bw_put_pending(&w, 1, &pending);
                low <<= 1; high <<= 1; high |= 1;
            } else if (low >= 0x40000000u && high < 0xC0000000u) {
                pending++;
                low <<= 1; low &= 0x7FFFFFFFu;
                high <<= 1; high |= 0x80000001u;
            } else break;
        }
    }
    pending += 1;
    bw_put_pending(&w, low < 0x40000000u ? 0 : 1, &pending);
    while (w.count != 0) bw_put(&w, 0);
    return w.overflow ? -1 : w.n;
}

typedef struct { const uint8_t* buf; int64_t len; int64_t pos; uint8_t cache; int bits; } orc_br_t;

static void br_get(orc_br_t* r, uint32_t* value)
{
    if (r->bits == 0) {
        if (r->pos == r->len) { *value <<= 1; return; }
        r->cache = r->buf[r->pos++];
        r->bits = 8;
    }
    *value <<= 1;
    *value |= (uint32_t)((r->cache >> (r->bits - 1)) & 1);
    r->bits--;
}

int orc_rc_decode2(const float* p_one, int is_u, int64_t n, const uint8_t* in, int64_t len, int16_t* out)
{
    orc_br_t r = {in, len, 0, 0, 0};
    uint32_t low = 0, high = 0xFFFFFFFFu, value = 0;
    for (int i = 0; i < 32; i++) br_get(&r, &value);
    for (int64_t i = 0; i < n; i++) {
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        const uint16_t count = (uint16_t)((((uint64_t)value - (uint64_t)low + 1) * 0x10000u - 1) / span);
        const uint32_t c1 = orc_cdf1_(p_one[i], is_u);
        /* binary search over {0, c1}: largest m with cdf[m] <= count */
        const int s = (c1 <= count) ? 1 : 0;
        out[i] = (int16_t)s;
        if (i == n - 1) break;
        const uint32_t c_low = s ? c1 : 0u;
        const uint32_t c_high = s ? 0x10000u : c1;
        high = (low - 1) + (uint32_t)((span * (uint64_t)c_high) >> 16);
        low = low + (uint32_t)((span * (uint64_t)c_low) >> 16);
        for (;;) {
            if (low >= 0x80000000u || high < 0x80000000u) {
                low <<= 1; high <<= 1; high |= 1;
                br_get(&r, &value);
            } else if (low >= 0x40000000u && high < 0xC0000000u) {
                low <<= 1; low &= 0x7FFFFFFFu;
                high <<= 1; high |= 0x80000001u;
                value -= 0x40000000u;
                br_get(&r, &value);
            } else break;
        }
    }
    return 0;
}

int64_t orc_rc_encode(const float* p_one, const int16_t* sym, int64_t n, uint8_t* out, int64_t cap)
{
    (void)g_unused;
    return orc_rc_encode2(p_one, 0, sym, n, out, cap);
}

int orc_rc_decode(const float* p_one, int64_t n, const uint8_t* in, int64_t len, int16_t* out)
{
    return orc_rc_decode2(p_one, 0, n, in, len, out);
}
