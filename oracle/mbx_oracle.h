/*
 * mbx_oracle.h -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's (arancormonk/mbelib-neo v2.0.0) hot path, used
 * to check the HIP kernels.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it; the product (mbelib-neo_amd/) never links, imports or
 * falls back to anything in oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference_golden.py checks this file
 * against (a) every golden vector the reference's own tests hold for the path
 * (tests/test_golden_pcm.c:78-84 FNV hashes, tests/test_ecc.c known answers,
 * tests/test_params.c known answers, tests/test_floattoshort_parity.c edge values) and
 * (b) fixtures produced by the real reference compiled here (oracle/_ref, see Makefile).
 * Integer stages are bit-exact; float stages are bit-exact too except the values that
 * pass through the 256-point FFT (the reference uses the vendored PFFFT, this file uses
 * a double-precision FFT): those agree to ~1e-7 relative.
 */
#ifndef MBX_ORACLE_H
#define MBX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "mbx_tables.h"
#include "mbx_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* tables -------------------------------------------------------------------------- */
void mbxo_set_preclip_peaks(float* out); /* diagnostic: when non-NULL, mbxo_process_batch[_soft] writes per frame the largest |sample| before the soft clip (0 for frames that do not run the synthesiser); calling thread only */
void mbxo_set_tones(int on); /* 0: the reference's NOTONES build (tone frames = silence, phases untouched); 1 (default): tones synthesised */
void mbxo_set_fft_float(int on); /* 1: the unvoiced FFT as FFTPACK's float real transform (= the reference's PFFFT, bit for bit); 0 (default): double precision */
int mbxo_load_tables(const void* blob, size_t n); /* 0, or -1 on bad magic/size/checksum */

/* frame packing (host side of the boundary) ---------------------------------------- */
int mbxo_pack_imbe_frame(const char fr[8][23], uint8_t out[MBX_IMBE_FRAME_BYTES]); /* 0 / -1 / -2 */
int mbxo_pack_ambe_frame(const char fr[4][24], uint8_t out[MBX_AMBE_FRAME_BYTES]);
void mbxo_unpack_imbe_frame(const uint8_t in[MBX_IMBE_FRAME_BYTES], char fr[8][23]);
void mbxo_unpack_ambe_frame(const uint8_t in[MBX_AMBE_FRAME_BYTES], char fr[4][24]);
void mbxo_record_to_bits(const mbx_param_record* rec, int nbits, char* bits);
void mbxo_record_to_result(const mbx_param_record* rec, mbe_process_result* result);

/* FEC stage (stateless) -------------------------------------------------------------- */
int mbxo_golay2312_word(uint32_t cw, uint32_t* fixed);   /* returns corrected data bits */
int mbxo_hamming1511_word(uint32_t cw, uint32_t* fixed); /* returns 0/1 */
int mbxo_golay2312(const char* in, char* out);           /* char-array forms of the two above */
int mbxo_hamming1511(const char* in, char* out);
int mbxo_fec_imbe7200x4400(const uint8_t frame[MBX_IMBE_FRAME_BYTES], mbx_param_record* rec);
int mbxo_fec_ambe3600x2450(const uint8_t frame[MBX_AMBE_FRAME_BYTES], mbx_param_record* rec);
/* the reference's char-array entry points, same return/validation behaviour */
/* IMBE 7100x4400 front end -- ref src/imbe/imbe7100x4400.c:100-122 (C0), :153-212
 * (data ECC), :292-334 (demodulation), :381-438 (mbe_convertImbe7100to7200), :440-479 (frame decode),
 * src/ecc/ecc.c:422-464 (mbe_7100x4400hamming1511).  The record holds the 88 bits AFTER the conversion. */
int mbxo_pack_imbe7100_frame(const char fr[7][24], uint8_t out[MBX_IMBE7100_FRAME_BYTES]); /* 0 / -1 / -2 */
void mbxo_unpack_imbe7100_frame(const uint8_t in[MBX_IMBE7100_FRAME_BYTES], char fr[7][24]);
int mbxo_hamming1511_7100_word(uint32_t cw, uint32_t* fixed);
int mbxo_hamming1511_7100(const char* in, char* out);
int mbxo_convert_imbe7100to7200(char* imbe_d);
int mbxo_fec_imbe7100x4400(const uint8_t frame[MBX_IMBE7100_FRAME_BYTES], mbx_param_record* rec);
int mbxo_decode_imbe7100x4400_frame(const char fr[7][24], char imbe_d[88], mbe_process_result* result);

/* AMBE 3600x2400 (D-STAR) -- FEC / demodulation are the AMBE+2 ones (ambe_common.c), the
 * parameter decode and the frame policy are its own: ref src/ambe/ambe3600x2400.c:164-551 (decode), :629-763
 * (policy), src/core/mbelib.c:813-856 (D-STAR tones).  mbxo_decode returns 0 voice, 3 tone/silence class, or the
 * tone index 5..122. */
int mbxo_decode_ambe2400_parms(const char* ambe_d, mbe_parms* cur, mbe_parms* prev);
int mbxo_process_ambe2400_dataf(float* out, mbe_process_result* result, const char ambe_d[49], mbe_parms* cur,
                                mbe_parms* prev, mbe_parms* prev_enh, mbx_stream_rng* rng);
int mbxo_process_ambe3600x2400_framef(float* out, mbe_process_result* result, const char fr[4][24], char ambe_d[49],
                                      mbe_parms* cur, mbe_parms* prev, mbe_parms* prev_enh, mbx_stream_rng* rng);
void mbxo_tone_dstarf(float* out, mbe_parms* cur, int id1);

/* soft-decision front end -- ref src/ecc/ecc.c:138-215,303-357,410-413,
 * src/imbe/imbe7200x4400.c:445-459,517-560,675-707,746-778, src/ambe/ambe_common.c:48-73,102-124,159-190,
 * src/ambe/ambe3600x2450.c:684-714, src/core/mbelib.c:107-158 */
mbe_soft_bit mbxo_soft_bit_from_hard(int bit, uint8_t reliability);
mbe_soft_bit mbxo_soft_bit_from_llr(int16_t llr);
int mbxo_soft_bits_from_hard(const char* bits, mbe_soft_bit* soft, size_t count, uint8_t reliability);
int mbxo_soft_bits_from_llr(const int16_t* llr, mbe_soft_bit* soft, size_t count);
int mbxo_golay2312_soft(const mbe_soft_bit* in, char* out);   /* returns data-bit differences, -1/-2 on bad input */
int mbxo_hamming1511_soft(const mbe_soft_bit* in, char* out); /* returns bit differences */
int mbxo_hamming1511_7100_soft(const mbe_soft_bit* in, char* out); /* 7100x4400 bit mapping, src/ecc/ecc.c:466-469 */
int mbxo_fec_imbe7100x4400_soft(const mbe_soft_bit fr[7][24], mbx_param_record* rec); /* src/imbe/imbe7100x4400.c:124-150,214-274,336-378,481-525 */
int mbxo_decode_imbe7100x4400_soft_frame(const mbe_soft_bit fr[7][24], char imbe_d[88], mbe_process_result* result);
int mbxo_fec_imbe7200x4400_soft(const mbe_soft_bit fr[8][23], mbx_param_record* rec);
int mbxo_fec_ambe3600x2450_soft(const mbe_soft_bit fr[4][24], mbx_param_record* rec);
int mbxo_decode_imbe7200x4400_soft_frame(const mbe_soft_bit fr[8][23], char imbe_d[88], mbe_process_result* result);
int mbxo_decode_ambe3600x2450_soft_frame(const mbe_soft_bit fr[4][24], char ambe_d[49], mbe_process_result* result);
int mbxo_fec_soft_batch(int codec, size_t n, const mbe_soft_bit* soft, mbx_param_record* records);
int mbxo_process_batch_soft(int codec, int S, int T, const mbe_soft_bit* soft, mbe_parms* state, mbx_stream_rng* rng,
                            int16_t* pcm16, float* pcmf, mbe_process_result* results, mbx_param_record* records);

int mbxo_decode_imbe7200x4400_frame(const char fr[8][23], char imbe_d[88], mbe_process_result* result);
int mbxo_decode_ambe3600x2450_frame(const char fr[4][24], char ambe_d[49], mbe_process_result* result);

/* stream stage ------------------------------------------------------------------------ */
void mbxo_init_parms(mbe_parms* cur, mbe_parms* prev, mbe_parms* prev_enh);
void mbxo_rng_default(mbx_stream_rng* rng);
void mbxo_rng_seed(mbx_stream_rng* rng, uint32_t seed);
int mbxo_decode_imbe4400_parms(const char* imbe_d, mbe_parms* cur, mbe_parms* prev);
int mbxo_decode_ambe2450_parms(const char* ambe_d, mbe_parms* cur, mbe_parms* prev, int total_errors);
int mbxo_process_imbe4400_dataf(float* out, mbe_process_result* result, const char imbe_d[88], mbe_parms* cur,
                                mbe_parms* prev, mbe_parms* prev_enh, mbx_stream_rng* rng);
int mbxo_process_ambe2450_dataf(float* out, mbe_process_result* result, const char ambe_d[49], mbe_parms* cur,
                                mbe_parms* prev, mbe_parms* prev_enh, mbx_stream_rng* rng);
int mbxo_process_imbe7200x4400_framef(float* out, mbe_process_result* result, const char fr[8][23], char imbe_d[88],
                                      mbe_parms* cur, mbe_parms* prev, mbe_parms* prev_enh, mbx_stream_rng* rng);
int mbxo_process_ambe3600x2450_framef(float* out, mbe_process_result* result, const char fr[4][24], char ambe_d[49],
                                      mbe_parms* cur, mbe_parms* prev, mbe_parms* prev_enh, mbx_stream_rng* rng);

/* synthesis pieces -------------------------------------------------------------------- */
float mbxo_spectral_amp_enhance(mbe_parms* cur); /* returns pre-enhancement Rm0 */
void mbxo_adaptive_smoothing(mbe_parms* cur, const mbe_parms* prev);
void mbxo_synthesize_speechf(float* out, mbe_parms* cur, mbe_parms* prev, mbx_stream_rng* rng);
void mbxo_comfort_noisef(float* out, mbx_stream_rng* rng);
void mbxo_tonef(float* out, const char* ambe_d, mbe_parms* cur);
void mbxo_floattoshort(const float* in, int16_t* out);
void mbxo_noise_next(float buffer[256], float* seed, float overlap[96], mbx_stream_rng* rng);

/* batch driver: same contract as the HIP launcher's mbx_process_batch (include/mbx.h) */
int mbxo_process_batch(int codec, int S, int T, const uint8_t* frames, mbe_parms* state, mbx_stream_rng* rng,
                       int16_t* pcm16, float* pcmf, mbe_process_result* results, mbx_param_record* records);
int mbxo_fec_batch(int codec, size_t n, const uint8_t* frames, mbx_param_record* records);
void mbxo_floattoshort_batch(const float* in, int16_t* out, size_t nframes);
void mbxo_synthesize_speech_batch(int S, mbe_parms* cur, mbe_parms* prev, mbx_stream_rng* rng, float* pcmf);

uint32_t mbxo_fnv1a32(const void* data, size_t len);

#ifdef __cplusplus
}
#endif
#endif
