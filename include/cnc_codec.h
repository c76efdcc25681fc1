/*
 * cnc_codec.h — C ABI of libcnc_codec.so: the CPU entropy coder for the ±1 hash-grid embeddings.
 *
 * Replaces the reference's use of torchac==0.9.3 (requirements.txt:32) behind
 * `encoder(x, p, file_name)` / `decoder(p, file_name)` (examples/utils_bpp_acc.py:77-110):
 *     torchac.encode_float_cdf(cat([0, 1-p, 1]), sym=(x+1)//2)   ->  cnc_rc_encode_pm1
 *     torchac.decode_float_cdf(cat([0, 1-p, 1]), bytes) * 2 - 1   ->  cnc_rc_decode_pm1
 * The coder is sequential and runs on the host in the reference as well (torchac is a CPU
 * extension fed by .cpu() copies); it is not part of the GPU data path.
 *
 * Bitstream: torchac's published format — 16-bit CDF c1 = round_half_even((1-p)*65534)+1,
 * 32-bit low/high interval coder with pending-bit carry resolution, MSB-first, one terminating
 * bit (+pending), zero padded to a byte.  torchac itself is absent from the reference tree, so
 * byte-level parity with it is UNPINNED; round-trip exactness and size-vs-entropy are tested.
 */
#ifndef CNC_CODEC_H
#define CNC_CODEC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Upper bound of the encoded size in bytes for n symbols (worst case 16 bits/symbol + tail). */
int64_t cnc_rc_bound(int64_t n);

/* x_pm1[i] in {-1,+1} (any value > 0 codes as +1), p_one[i] = P(x=+1) in (0,1).
 * Returns the number of bytes written, or -1 if cap < cnc_rc_bound(n) was too small. */
int64_t cnc_rc_encode_pm1(const float* p_one, const float* x_pm1, int64_t n, uint8_t* out,
                          int64_t cap);

/* Inverse: fills x_pm1[0..n) with -1.0f / +1.0f.  Returns 0. */
int cnc_rc_decode_pm1(const float* p_one, int64_t n, const uint8_t* in, int64_t len,
                      float* x_pm1);

/* General alphabets — what `torchac.encode_int16_normalized_cdf(cdf_int, sym)` / `decode_int16_normalized_cdf`
 * bind (and, after the float -> 16-bit conversion done by the Python shim cnc_amd/backends/torchac.py exactly as
 * torchac's `_convert_to_int_and_normalize` publishes it, `encode_float_cdf` / `decode_float_cdf`, the two calls of
 * examples/utils_bpp_acc.py:87,108).  cdf: [n, Lp] uint16 rows, row[0] = 0, non-decreasing, the LAST entry stands
 * for 2^16 whatever it holds (torchac stores 65536 wrapped to 0); sym[i] in [0, Lp-2].
 * encode: bytes written, -1 if cap < cnc_rc_bound(n), -2 for a symbol outside the alphabet.  decode: 0 / -2. */
int64_t cnc_rc_encode_cdf16(const uint16_t* cdf, const int16_t* sym, int64_t n, int32_t Lp, uint8_t* out,
                            int64_t cap);
int cnc_rc_decode_cdf16(const uint16_t* cdf, int64_t n, int32_t Lp, const uint8_t* in, int64_t len,
                        int16_t* sym);

#ifdef __cplusplus
}
#endif
#endif
