/*
 * mbx_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see mbx_oracle.h).
 *
 * Restates, in plain C99 and in this project's own structure (packed code words, one
 * table blob, explicit per-stream RNG state), what the reference computes on the hot path.
 * Every function names the reference location it follows; `ref:` paths are relative to
 * /root/reference.  Float expressions keep the reference's operand order and types so the
 * results are bit-identical under an IEEE build (-ffp-contract=off, no fast-math).  The 256-point FFT of the unvoiced
 * path comes in two forms: a double-precision one (the default: what the HIP path is compared with, a precision ABOVE both
 * float implementations) and, after mbxo_set_fft_float(1), FFTPACK's float real transform as the reference's PFFFT runs it --
 * with that one the float PCM is the reference's bit for bit (its float golden hash is reproduced).
 */
#define _GNU_SOURCE 1 /* sincosf */
#include "mbx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_SQRT2
#define M_SQRT2 1.41421356237309504880
#endif

static mbx_tables g_tab_storage;
static const mbx_tables* T = NULL;

uint32_t
mbxo_fnv1a32(const void* data, size_t len) {
    const uint8_t* p = (const uint8_t*)data;
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < len; ++i) {
        h = (h ^ p[i]) * 16777619u;
    }
    return h;
}

int
mbxo_load_tables(const void* blob, size_t n) {
    if (!blob || n != sizeof(mbx_tables)) {
        return -1;
    }
    memcpy(&g_tab_storage, blob, sizeof(g_tab_storage));
    if (g_tab_storage.magic != MBX_TABLES_MAGIC || g_tab_storage.version != MBX_TABLES_VERSION
        || g_tab_storage.total_bytes != sizeof(mbx_tables)) {
        return -1;
    }
    if (mbxo_fnv1a32((const uint8_t*)&g_tab_storage.checksum + 4, sizeof(mbx_tables) - 16) != g_tab_storage.checksum) {
        return -1;
    }
    T = &g_tab_storage;
    return 0;
}

/* =====================================================================================
 * Bit plumbing
 * ===================================================================================== */

static const int imbe_row_width[8] = {23, 23, 23, 23, 15, 15, 15, 7};
static const int ambe_row_width[4] = {24, 23, 11, 14};
static const int imbe7100_row_width[7] = {19, 24, 23, 23, 15, 15, 23};

/* ref: src/internal/mbe_result.h:18-29 (mbe_validate_bits) */
static int
validate_bits(const char* bits, size_t count) {
    if (!bits) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    for (size_t i = 0; i < count; ++i) {
        if (bits[i] != 0 && bits[i] != 1) {
            return MBE_STATUS_INVALID_BITS;
        }
    }
    return 0;
}

static void
put_bit(uint8_t* buf, int pos, int bit) {
    if (bit) {
        buf[pos >> 3] |= (uint8_t)(0x80u >> (pos & 7));
    }
}

static int
get_bit(const uint8_t* buf, int pos) {
    return (buf[pos >> 3] >> (7 - (pos & 7))) & 1;
}

int
mbxo_pack_imbe_frame(const char fr[8][23], uint8_t out[MBX_IMBE_FRAME_BYTES]) {
    int rc = validate_bits((const char*)fr, 8u * 23u); /* the whole array, like the reference */
    if (rc < 0) {
        return rc;
    }
    if (!out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    memset(out, 0, MBX_IMBE_FRAME_BYTES);
    int pos = 0;
    for (int r = 0; r < 8; ++r) {
        for (int j = imbe_row_width[r] - 1; j >= 0; --j) {
            put_bit(out, pos++, fr[r][j]);
        }
    }
    return 0;
}

int
mbxo_pack_ambe_frame(const char fr[4][24], uint8_t out[MBX_AMBE_FRAME_BYTES]) {
    int rc = validate_bits((const char*)fr, 4u * 24u);
    if (rc < 0) {
        return rc;
    }
    if (!out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    memset(out, 0, MBX_AMBE_FRAME_BYTES);
    int pos = 0;
    for (int r = 0; r < 4; ++r) {
        for (int j = ambe_row_width[r] - 1; j >= 0; --j) {
            put_bit(out, pos++, fr[r][j]);
        }
    }
    return 0;
}

void
mbxo_unpack_imbe_frame(const uint8_t in[MBX_IMBE_FRAME_BYTES], char fr[8][23]) {
    memset(fr, 0, 8 * 23);
    int pos = 0;
    for (int r = 0; r < 8; ++r) {
        for (int j = imbe_row_width[r] - 1; j >= 0; --j) {
            fr[r][j] = (char)get_bit(in, pos++);
        }
    }
}

void
mbxo_unpack_ambe_frame(const uint8_t in[MBX_AMBE_FRAME_BYTES], char fr[4][24]) {
    memset(fr, 0, 4 * 24);
    int pos = 0;
    for (int r = 0; r < 4; ++r) {
        for (int j = ambe_row_width[r] - 1; j >= 0; --j) {
            fr[r][j] = (char)get_bit(in, pos++);
        }
    }
}

/* read `width` bits starting at stream position `pos` as an integer, first bit = MSB */
static uint32_t
take_bits(const uint8_t* buf, int pos, int width) {
    uint32_t v = 0;
    for (int i = 0; i < width; ++i) {
        v = (v << 1) | (uint32_t)get_bit(buf, pos + i);
    }
    return v;
}

static void
rec_put(mbx_param_record* rec, int i, int bit) {
    if (bit) {
        rec->w[i >> 5] |= 0x80000000u >> (i & 31);
    }
}

static int
rec_get(const mbx_param_record* rec, int i) {
    return (int)((rec->w[i >> 5] >> (31 - (i & 31))) & 1u);
}

/* append the top `count` bits of a `width`-bit value (bit width-1 first) */
static int
rec_append(mbx_param_record* rec, int at, uint32_t value, int width, int count) {
    for (int k = 0; k < count; ++k) {
        rec_put(rec, at++, (int)((value >> (width - 1 - k)) & 1u));
    }
    return at;
}

void
mbxo_record_to_bits(const mbx_param_record* rec, int nbits, char* bits) {
    for (int i = 0; i < nbits; ++i) {
        bits[i] = (char)rec_get(rec, i);
    }
}

void
mbxo_record_to_result(const mbx_param_record* rec, mbe_process_result* result) {
    result->c0_errors = (int)(rec->w[3] & 0xffu);
    result->protected_errors = (int)((rec->w[3] >> 8) & 0xffu);
    result->c4_errors = (int)((rec->w[3] >> 16) & 0xffu);
    result->total_errors = result->c0_errors + result->protected_errors;
    result->flags = (rec->w[3] >> 24) & 0xffu;
}

/* =====================================================================================
 * FEC stage  (a4..a7)
 * ===================================================================================== */

/* ref: src/ecc/ecc.c:221-301 (mbe_checkGolayBlock + mbe_golay2312).  cw bit j is in[j];
 * the 11 parity bits pass through untouched, the return value counts changed data bits. */
int
mbxo_golay2312_word(uint32_t cw, uint32_t* fixed) {
    uint32_t expect = 0;
    for (int i = 0; i < 12; ++i) {
        if (cw & (0x400000u >> i)) {
            expect ^= T->golay_gen[i];
        }
    }
    uint32_t syndrome = expect ^ (cw & 0x7ffu);
    uint32_t fix = T->golay_matrix[syndrome];
    *fixed = cw ^ (fix << 11);
    return __builtin_popcount(fix);
}

/* ref: src/ecc/ecc.c:366-408 (mbe_hamming1511) */
int
mbxo_hamming1511_word(uint32_t cw, uint32_t* fixed) {
    int syndrome = 0;
    for (int i = 0; i < 4; ++i) {
        syndrome |= (__builtin_popcount(cw & T->hamming_gen[i]) & 1) << i;
    }
    *fixed = (syndrome > 0) ? (cw ^ T->hamming_fix[syndrome]) : cw;
    return syndrome > 0;
}

static uint32_t
chars_to_word(const char* in, int n) {
    uint32_t w = 0;
    for (int j = n - 1; j >= 0; --j) {
        w = (w << 1) | (uint32_t)(in[j] & 1);
    }
    return w;
}

static void
word_to_chars(uint32_t w, char* out, int n) {
    for (int j = 0; j < n; ++j) {
        out[j] = (char)((w >> j) & 1u);
    }
}

int
mbxo_golay2312(const char* in, char* out) {
    if (!out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_bits(in, 23u);
    if (rc < 0) {
        return rc;
    }
    uint32_t fixed;
    int errs = mbxo_golay2312_word(chars_to_word(in, 23), &fixed);
    word_to_chars(fixed, out, 23);
    return errs;
}

int
mbxo_hamming1511(const char* in, char* out) {
    if (!out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_bits(in, 15u);
    if (rc < 0) {
        return rc;
    }
    uint32_t fixed;
    int errs = mbxo_hamming1511_word(chars_to_word(in, 15), &fixed);
    word_to_chars(fixed, out, 15);
    return errs;
}

/* pseudo-random demodulation bit k (k >= 1) of the sequence seeded by the 12 data bits of
 * C0; ref: src/imbe/imbe7200x4400.c:648-656, src/ambe/ambe_common.c:81-91 */
static void
pr_bits(uint32_t seed12, int count, uint8_t* bits /* [count+1], bits[0] unused */) {
    uint32_t x = (16u * seed12) & 0xffffu;
    for (int k = 1; k <= count; ++k) {
        x = (173u * x + 13849u) & 0xffffu;
        bits[k] = (uint8_t)(x >> 15);
    }
}

/* ref: src/imbe/imbe7200x4400.c:424-443 (C0), :636-673 (demod), :469-515 (data ECC),
 *      :709-744 (mbe_decodeImbe7200x4400Frame) */
int
mbxo_fec_imbe7200x4400(const uint8_t frame[MBX_IMBE_FRAME_BYTES], mbx_param_record* rec) {
    uint32_t row[8];
    int pos = 0;
    for (int r = 0; r < 8; ++r) {
        row[r] = take_bits(frame, pos, imbe_row_width[r]);
        pos += imbe_row_width[r];
    }

    int c0 = mbxo_golay2312_word(row[0], &row[0]);

    uint8_t pr[115];
    pr_bits(row[0] >> 11, 114, pr);
    int k = 1;
    for (int r = 1; r < 7; ++r) {
        for (int j = imbe_row_width[r] - 1; j >= 0; --j) {
            row[r] ^= (uint32_t)pr[k++] << j;
        }
    }

    int prot = 0, c4 = 0;
    memset(rec, 0, sizeof(*rec));
    int at = rec_append(rec, 0, row[0], 23, 12);
    for (int r = 1; r < 4; ++r) {
        prot += mbxo_golay2312_word(row[r], &row[r]);
        at = rec_append(rec, at, row[r], 23, 12);
    }
    for (int r = 4; r < 7; ++r) {
        int e = mbxo_hamming1511_word(row[r], &row[r]);
        prot += e;
        if (r == 4) {
            c4 = e;
        }
        at = rec_append(rec, at, row[r], 15, 11);
    }
    at = rec_append(rec, at, row[7], 7, 7);
    rec->w[3] = (uint32_t)c0 | ((uint32_t)prot << 8) | ((uint32_t)c4 << 16)
                | ((MBE_PROCESS_FLAG_C0_VALID | MBE_PROCESS_FLAG_C4_VALID) << 24);
    return c0 + prot;
}

/* ref: src/ambe/ambe_common.c:22-46 (C0 + Golay24 parity), :75-100 (demod), :127-157 (data),
 *      src/ambe/ambe3600x2450.c:649-682 (mbe_decodeAmbe3600x2450Frame) */
int
mbxo_fec_ambe3600x2450(const uint8_t frame[MBX_AMBE_FRAME_BYTES], mbx_param_record* rec) {
    uint32_t row[4];
    int pos = 0;
    for (int r = 0; r < 4; ++r) {
        row[r] = take_bits(frame, pos, ambe_row_width[r]);
        pos += ambe_row_width[r];
    }

    uint32_t cw;
    int c0 = mbxo_golay2312_word(row[0] >> 1, &cw);
    row[0] = (cw << 1) | (row[0] & 1u);
    if (c0 == 0 && (__builtin_popcount(row[0]) & 1)) {
        row[0] ^= 1u;
        c0 = 1;
    }

    uint8_t pr[24];
    pr_bits((row[0] >> 12) & 0xfffu, 23, pr);
    int k = 1;
    for (int j = 22; j >= 0; --j) {
        row[1] ^= (uint32_t)pr[k++] << j;
    }

    memset(rec, 0, sizeof(*rec));
    int at = rec_append(rec, 0, row[0], 24, 12);
    int prot = mbxo_golay2312_word(row[1], &row[1]);
    at = rec_append(rec, at, row[1], 23, 12);
    at = rec_append(rec, at, row[2], 11, 11);
    at = rec_append(rec, at, row[3], 14, 14);
    rec->w[3] = (uint32_t)c0 | ((uint32_t)prot << 8) | (MBE_PROCESS_FLAG_C0_VALID << 24);
    return c0 + prot;
}

/* =====================================================================================
 * IMBE 7100x4400 front end (§8(f) row 4): its own C0 (Golay shortened to 18 cells), demodulation
 * seed (7 bits), Hamming bit mapping and parameter-bit order; after mbe_convertImbe7100to7200 the
 * 88 bits are those of the 7200x4400 path.
 * ===================================================================================== */
int
mbxo_pack_imbe7100_frame(const char fr[7][24], uint8_t out[MBX_IMBE7100_FRAME_BYTES]) {
    int rc = validate_bits((const char*)fr, 7u * 24u);
    if (rc < 0) {
        return rc;
    }
    if (!out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    memset(out, 0, MBX_IMBE7100_FRAME_BYTES);
    int pos = 0;
    for (int r = 0; r < 7; ++r) {
        for (int j = imbe7100_row_width[r] - 1; j >= 0; --j) {
            put_bit(out, pos++, fr[r][j]);
        }
    }
    return 0;
}

void
mbxo_unpack_imbe7100_frame(const uint8_t in[MBX_IMBE7100_FRAME_BYTES], char fr[7][24]) {
    memset(fr, 0, 7 * 24);
    int pos = 0;
    for (int r = 0; r < 7; ++r) {
        for (int j = imbe7100_row_width[r] - 1; j >= 0; --j) {
            fr[r][j] = (char)get_bit(in, pos++);
        }
    }
}

/* ref: src/ecc/ecc.c:422-464 */
int
mbxo_hamming1511_7100_word(uint32_t cw, uint32_t* fixed) {
    int syndrome = 0;
    for (int i = 0; i < 4; ++i) {
        syndrome |= (__builtin_popcount(cw & T->hamming7100_gen[i]) & 1) << i;
    }
    *fixed = (syndrome > 0) ? (cw ^ T->hamming7100_fix[syndrome]) : cw;
    return syndrome > 0;
}

int
mbxo_hamming1511_7100(const char* in, char* out) {
    if (!out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_bits(in, 15u);
    if (rc < 0) {
        return rc;
    }
    uint32_t fixed;
    int errs = mbxo_hamming1511_7100_word(chars_to_word(in, 15), &fixed);
    word_to_chars(fixed, out, 15);
    return errs;
}

/* ref: src/imbe/imbe7100x4400.c:381-438 -- a permutation of the 88 bits that depends on K(b0) */
int
mbxo_convert_imbe7100to7200(char* d) {
    int rc = validate_bits(d, 88u);
    if (rc < 0) {
        return rc;
    }
    static const int b0_index[8] = {1, 2, 3, 4, 5, 6, 86, 87};
    int b0 = 0;
    for (int i = 0; i < 8; ++i) {
        b0 = (b0 << 1) | d[b0_index[i]];
    }
    float w0 = ((float)(4 * M_PI) / (float)((float)b0 + 39.5));
    int L = (int)(0.9254 * (int)((M_PI / w0) + 0.25));
    int K = (L < 37) ? (int)((float)(L + 2) / (float)3) : 12;
    char tmp[88];
    tmp[87] = d[0];
    tmp[48 + K] = d[42];
    tmp[49 + K] = d[43];
    for (int i = 0; i < K; ++i) {
        tmp[48 + i] = d[44 + i];
    }
    int j = 0, k = 1;
    while (j < 87) {
        tmp[j] = d[k];
        if (++j == 48) {
            j += K + 2;
        }
        if (++k == 42) {
            k += K + 2;
        }
    }
    memcpy(d, tmp, 88);
    return 0;
}

int
mbxo_fec_imbe7100x4400(const uint8_t frame[MBX_IMBE7100_FRAME_BYTES], mbx_param_record* rec) {
    uint32_t row[7];
    int pos = 0;
    for (int r = 0; r < 7; ++r) {
        row[r] = take_bits(frame, pos, imbe7100_row_width[r]);
        pos += imbe7100_row_width[r];
    }
    /* C0: cells 1..18 are the low 18 positions of a Golay block whose top five positions are zero */
    uint32_t fixed;
    int c0 = mbxo_golay2312_word((row[0] >> 1) & 0x3ffffu, &fixed);
    row[0] = ((fixed & 0x3ffffu) << 1) | (row[0] & 1u);

    uint8_t pr[101];
    pr_bits((row[0] >> 12) & 0x7fu, 100, pr);
    int k = 1;
    for (int r = 1; r < 6; ++r) {
        for (int j = imbe7100_row_width[r] - 1; j >= 0; --j) {
            row[r] ^= (uint32_t)pr[k++] << j;
        }
    }

    char d[88];
    int at = 0, prot = 0, c4 = 0;
    for (int j = 18; j > 11; --j) {
        d[at++] = (char)((row[0] >> j) & 1u);
    }
    uint32_t w;
    prot += mbxo_golay2312_word(row[1] >> 1, &w); /* C1: cells 1..23 */
    for (int j = 22; j > 10; --j) {
        d[at++] = (char)((w >> j) & 1u);
    }
    for (int r = 2; r < 4; ++r) {
        prot += mbxo_golay2312_word(row[r], &w);
        for (int j = 22; j > 10; --j) {
            d[at++] = (char)((w >> j) & 1u);
        }
    }
    for (int r = 4; r < 6; ++r) {
        int e = mbxo_hamming1511_7100_word(row[r], &w);
        prot += e;
        if (r == 4) {
            c4 = e;
        }
        for (int j = 14; j >= 4; --j) {
            d[at++] = (char)((w >> j) & 1u);
        }
    }
    for (int j = 22; j >= 0; --j) {
        d[at++] = (char)((row[6] >> j) & 1u);
    }
    mbxo_convert_imbe7100to7200(d);

    memset(rec, 0, sizeof(*rec));
    for (int i = 0; i < 88; ++i) {
        rec_put(rec, i, d[i]);
    }
    rec->w[3] = (uint32_t)c0 | ((uint32_t)prot << 8) | ((uint32_t)c4 << 16)
                | ((MBE_PROCESS_FLAG_C0_VALID | MBE_PROCESS_FLAG_C4_VALID) << 24);
    return c0 + prot;
}

int
mbxo_decode_imbe7100x4400_frame(const char fr[7][24], char imbe_d[88], mbe_process_result* result) {
    if (result) {
        memset(result, 0, sizeof(*result));
    }
    if (!imbe_d) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    uint8_t packed[MBX_IMBE7100_FRAME_BYTES];
    int rc = mbxo_pack_imbe7100_frame(fr, packed);
    if (rc < 0) {
        return rc;
    }
    mbx_param_record rec;
    int total = mbxo_fec_imbe7100x4400(packed, &rec);
    mbxo_record_to_bits(&rec, 88, imbe_d);
    if (result) {
        mbxo_record_to_result(&rec, result);
    }
    return total;
}

/* =====================================================================================
 * Soft-decision front end (§8(f) row 1).  Word-based restatement: a candidate is a packed
 * code word, its cost the sum of the reliabilities of the positions where it disagrees with
 * the hard decisions.  The reference scans the candidates in ascending data order and keeps
 * the current best unless the new one is strictly better under
 *     (lower cost) > (data equal to the hard decoder's output) > (fewer differing bits),
 * src/ecc/ecc.c:50-63; that scan returns the minimum of the key
 *     (cost, !matches_hard, differing bits, data word)           -- see soft_key().
 * ===================================================================================== */
mbe_soft_bit
mbxo_soft_bit_from_hard(int bit, uint8_t reliability) { /* src/core/mbelib.c:116-122 */
    mbe_soft_bit s;
    s.bit = (uint8_t)(bit ? 1u : 0u);
    s.reliability = reliability;
    return s;
}

mbe_soft_bit
mbxo_soft_bit_from_llr(int16_t llr) { /* src/core/mbelib.c:124-131 */
    int mag = (llr < 0) ? -(int)llr : (int)llr;
    mbe_soft_bit s;
    s.bit = (uint8_t)((llr > 0) ? 1u : 0u);
    s.reliability = (uint8_t)(mag > 255 ? 255 : mag);
    return s;
}

int
mbxo_soft_bits_from_hard(const char* bits, mbe_soft_bit* soft, size_t count, uint8_t reliability) {
    if (!soft) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_bits(bits, count);
    if (rc < 0) {
        return rc;
    }
    for (size_t i = 0; i < count; ++i) {
        soft[i] = mbxo_soft_bit_from_hard(bits[i], reliability);
    }
    return 0;
}

int
mbxo_soft_bits_from_llr(const int16_t* llr, mbe_soft_bit* soft, size_t count) {
    if (!llr || !soft) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    for (size_t i = 0; i < count; ++i) {
        soft[i] = mbxo_soft_bit_from_llr(llr[i]);
    }
    return 0;
}

static int
validate_soft(const mbe_soft_bit* bits, size_t count) { /* src/internal/mbe_result.h:31-42 */
    if (!bits) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    for (size_t i = 0; i < count; ++i) {
        if (bits[i].bit > 1u) {
            return MBE_STATUS_INVALID_BITS;
        }
    }
    return 0;
}

static uint32_t
soft_hard_word(const mbe_soft_bit* in, int n) {
    uint32_t w = 0;
    for (int j = n - 1; j >= 0; --j) {
        w = (w << 1) | (uint32_t)(in[j].bit & 1u);
    }
    return w;
}

static uint32_t
soft_cost(const mbe_soft_bit* in, uint32_t diff) {
    uint32_t cost = 0;
    while (diff) {
        int j = __builtin_ctz(diff);
        cost += in[j].reliability;
        diff &= diff - 1;
    }
    return cost;
}

static uint64_t
soft_key(uint32_t cost, int matches_hard, int diffs, uint32_t data) {
    return ((uint64_t)cost << 32) | ((uint64_t)(matches_hard ? 0u : 1u) << 24) | ((uint64_t)diffs << 16) | data;
}

static uint32_t
golay_encode(uint32_t data) { /* src/ecc/ecc.c:65-80 */
    uint32_t ecc = 0;
    for (int i = 0; i < 12; ++i) {
        if ((data >> (11 - i)) & 1u) {
            ecc ^= T->golay_gen[i];
        }
    }
    return (data << 11) | ecc;
}

/* soft Golay on a packed word; *out = chosen data bits over the HARD parity bits (ecc.c:354-356) */
static int
golay2312_soft_word(const mbe_soft_bit* in, uint32_t* out) {
    const uint32_t hard = soft_hard_word(in, 23);
    uint32_t hard_fixed;
    (void)mbxo_golay2312_word(hard, &hard_fixed);
    const uint32_t hard_data = hard_fixed >> 11;
    uint64_t best = ~(uint64_t)0;
    for (uint32_t data = 0; data < 4096u; ++data) {
        const uint32_t diff = golay_encode(data) ^ hard;
        const uint64_t key = soft_key(soft_cost(in, diff), data == hard_data, __builtin_popcount(diff >> 11), data);
        if (key < best) {
            best = key;
        }
    }
    const uint32_t data = (uint32_t)(best & 0xffffu);
    *out = (data << 11) | (hard & 0x7ffu);
    return (int)((best >> 16) & 0xffu);
}

int
mbxo_golay2312_soft(const mbe_soft_bit* in, char* out) {
    if (!out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_soft(in, 23u);
    if (rc < 0) {
        return rc;
    }
    uint32_t w;
    int diffs = golay2312_soft_word(in, &w);
    word_to_chars(w, out, 23);
    return diffs;
}

/* Hamming(15,11) code word of an 11-bit data word: data bit i sits at position pos[i], the four
 * parity positions are the unique values with a zero syndrome (src/ecc/ecc.c:128-155).  variant7100
 * selects the IMBE 7100x4400 bit mapping and generator. */
static uint32_t
hamming_encode(uint32_t data, int variant7100) {
    static const int std_data[11] = {2, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14}, std_parity[4] = {0, 1, 3, 7};
    static const int v71_data[11] = {4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14}, v71_parity[4] = {0, 1, 2, 3};
    const int* data_pos = variant7100 ? v71_data : std_data;
    const int* parity_pos = variant7100 ? v71_parity : std_parity;
    const uint16_t* gen = variant7100 ? T->hamming7100_gen : T->hamming_gen;
    uint32_t cw = 0;
    for (int i = 0; i < 11; ++i) {
        cw |= ((data >> i) & 1u) << data_pos[i];
    }
    for (uint32_t p = 0; p < 16u; ++p) {
        uint32_t c = cw;
        for (int i = 0; i < 4; ++i) {
            c |= ((p >> i) & 1u) << parity_pos[i];
        }
        int syndrome = 0;
        for (int i = 0; i < 4; ++i) {
            syndrome |= (__builtin_popcount(c & gen[i]) & 1) << i;
        }
        if (syndrome == 0) {
            return c;
        }
    }
    return 0xffffffffu; /* not reached: every data word has a code word */
}

static int
hamming1511_soft_word(const mbe_soft_bit* in, uint32_t* out, int variant7100) {
    const uint32_t hard = soft_hard_word(in, 15);
    uint32_t hard_fixed;
    if (variant7100) {
        (void)mbxo_hamming1511_7100_word(hard, &hard_fixed);
    } else {
        (void)mbxo_hamming1511_word(hard, &hard_fixed);
    }
    uint64_t best = ~(uint64_t)0;
    uint32_t best_cw = hard_fixed;
    for (uint32_t data = 0; data < 2048u; ++data) {
        const uint32_t cw = hamming_encode(data, variant7100);
        if (cw == 0xffffffffu) {
            continue;
        }
        const uint32_t diff = cw ^ hard;
        const uint64_t key = soft_key(soft_cost(in, diff), cw == hard_fixed, __builtin_popcount(diff), data);
        if (key < best) {
            best = key;
            best_cw = cw;
        }
    }
    *out = best_cw;
    return (best == ~(uint64_t)0) ? __builtin_popcount(best_cw ^ hard) : (int)((best >> 16) & 0xffu);
}

static int
hamming1511_soft_chars(const mbe_soft_bit* in, char* out, int variant7100) {
    if (!out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_soft(in, 15u);
    if (rc < 0) {
        return rc;
    }
    uint32_t w;
    int diffs = hamming1511_soft_word(in, &w, variant7100);
    word_to_chars(w, out, 15);
    return diffs;
}

int
mbxo_hamming1511_7100_soft(const mbe_soft_bit* in, char* out) {
    return hamming1511_soft_chars(in, out, 1);
}

int
mbxo_hamming1511_soft(const mbe_soft_bit* in, char* out) {
    return hamming1511_soft_chars(in, out, 0);
}

/* ref: src/imbe/imbe7200x4400.c:445-459 (C0), :675-707 (demod), :517-560 (data ECC), :746-778 */
int
mbxo_fec_imbe7200x4400_soft(const mbe_soft_bit fr[8][23], mbx_param_record* rec) {
    uint32_t row[8];
    int c0 = golay2312_soft_word(fr[0], &row[0]);

    uint8_t pr[115];
    pr_bits(row[0] >> 11, 114, pr);
    mbe_soft_bit work[7][23];
    int k = 1;
    for (int r = 1; r < 7; ++r) {
        for (int j = imbe_row_width[r] - 1; j >= 0; --j) {
            work[r][j].bit = (uint8_t)((fr[r][j].bit & 1u) ^ pr[k++]);
            work[r][j].reliability = fr[r][j].reliability;
        }
    }

    int prot = 0, c4 = 0;
    memset(rec, 0, sizeof(*rec));
    int at = rec_append(rec, 0, row[0], 23, 12);
    for (int r = 1; r < 4; ++r) {
        prot += golay2312_soft_word(work[r], &row[r]);
        at = rec_append(rec, at, row[r], 23, 12);
    }
    for (int r = 4; r < 7; ++r) {
        int e = hamming1511_soft_word(work[r], &row[r], 0);
        prot += e;
        if (r == 4) {
            c4 = e;
        }
        at = rec_append(rec, at, row[r], 15, 11);
    }
    at = rec_append(rec, at, soft_hard_word(fr[7], 7), 7, 7);
    rec->w[3] = (uint32_t)c0 | ((uint32_t)prot << 8) | ((uint32_t)c4 << 16)
                | ((MBE_PROCESS_FLAG_SOFT_INPUT | MBE_PROCESS_FLAG_C0_VALID | MBE_PROCESS_FLAG_C4_VALID) << 24);
    return c0 + prot;
}

/* ref: src/ambe/ambe_common.c:48-73 (C0 + parity), :102-124 (demod), :159-190 (data),
 *      src/ambe/ambe3600x2450.c:684-714 */
int
mbxo_fec_ambe3600x2450_soft(const mbe_soft_bit fr[4][24], mbx_param_record* rec) {
    uint32_t cw;
    int c0 = golay2312_soft_word(&fr[0][1], &cw);
    uint32_t row0 = (cw << 1) | (uint32_t)(fr[0][0].bit & 1u);
    if (c0 == 0 && (__builtin_popcount(row0) & 1)) {
        row0 ^= 1u;
        c0 = 1;
    }

    uint8_t pr[24];
    pr_bits((row0 >> 12) & 0xfffu, 23, pr);
    mbe_soft_bit c1[23];
    int k = 1;
    for (int j = 22; j >= 0; --j) {
        c1[j].bit = (uint8_t)((fr[1][j].bit & 1u) ^ pr[k++]);
        c1[j].reliability = fr[1][j].reliability;
    }

    memset(rec, 0, sizeof(*rec));
    int at = rec_append(rec, 0, row0, 24, 12);
    uint32_t row1;
    int prot = golay2312_soft_word(c1, &row1);
    at = rec_append(rec, at, row1, 23, 12);
    at = rec_append(rec, at, soft_hard_word(fr[2], 11), 11, 11);
    at = rec_append(rec, at, soft_hard_word(fr[3], 14), 14, 14);
    rec->w[3] = (uint32_t)c0 | ((uint32_t)prot << 8) | ((MBE_PROCESS_FLAG_SOFT_INPUT | MBE_PROCESS_FLAG_C0_VALID) << 24);
    return c0 + prot;
}

/* ref: src/imbe/imbe7100x4400.c:124-150 (C0: the five missing positions are certain zeros), :336-378
 *      (demodulation), :214-274 (data ECC), :481-525 (frame decode) */
int
mbxo_fec_imbe7100x4400_soft(const mbe_soft_bit fr[7][24], mbx_param_record* rec) {
    mbe_soft_bit blk[23];
    for (int j = 0; j < 18; ++j) {
        blk[j] = fr[0][j + 1];
    }
    for (int j = 18; j < 23; ++j) {
        blk[j] = mbxo_soft_bit_from_hard(0, 255u);
    }
    uint32_t w;
    int c0 = golay2312_soft_word(blk, &w);
    uint32_t row0 = ((w & 0x3ffffu) << 1) | (uint32_t)(fr[0][0].bit & 1u);

    uint8_t pr[101];
    pr_bits((row0 >> 12) & 0x7fu, 100, pr);
    mbe_soft_bit work[6][24];
    int k = 1;
    for (int r = 1; r < 6; ++r) {
        for (int j = imbe7100_row_width[r] - 1; j >= 0; --j) {
            work[r][j].bit = (uint8_t)((fr[r][j].bit & 1u) ^ pr[k++]);
            work[r][j].reliability = fr[r][j].reliability;
        }
    }

    char d[88];
    int at = 0, prot = 0, c4 = 0;
    for (int j = 18; j > 11; --j) {
        d[at++] = (char)((row0 >> j) & 1u);
    }
    prot += golay2312_soft_word(&work[1][1], &w); /* C1: cells 1..23 */
    for (int j = 22; j > 10; --j) {
        d[at++] = (char)((w >> j) & 1u);
    }
    for (int r = 2; r < 4; ++r) {
        prot += golay2312_soft_word(work[r], &w);
        for (int j = 22; j > 10; --j) {
            d[at++] = (char)((w >> j) & 1u);
        }
    }
    for (int r = 4; r < 6; ++r) {
        int e = hamming1511_soft_word(work[r], &w, 1);
        prot += e;
        if (r == 4) {
            c4 = e;
        }
        for (int j = 14; j >= 4; --j) {
            d[at++] = (char)((w >> j) & 1u);
        }
    }
    for (int j = 22; j >= 0; --j) {
        d[at++] = (char)(fr[6][j].bit & 1u);
    }
    mbxo_convert_imbe7100to7200(d);
    memset(rec, 0, sizeof(*rec));
    for (int i = 0; i < 88; ++i) {
        rec_put(rec, i, d[i]);
    }
    rec->w[3] = (uint32_t)c0 | ((uint32_t)prot << 8) | ((uint32_t)c4 << 16)
                | ((MBE_PROCESS_FLAG_SOFT_INPUT | MBE_PROCESS_FLAG_C0_VALID | MBE_PROCESS_FLAG_C4_VALID) << 24);
    return c0 + prot;
}

int
mbxo_decode_imbe7100x4400_soft_frame(const mbe_soft_bit fr[7][24], char imbe_d[88], mbe_process_result* result) {
    if (result) {
        memset(result, 0, sizeof(*result));
    }
    if (!imbe_d) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_soft(&fr[0][0], 7u * 24u);
    if (rc < 0) {
        return rc;
    }
    mbx_param_record rec;
    int total = mbxo_fec_imbe7100x4400_soft(fr, &rec);
    mbxo_record_to_bits(&rec, 88, imbe_d);
    if (result) {
        mbxo_record_to_result(&rec, result);
    }
    return total;
}

int
mbxo_decode_imbe7200x4400_soft_frame(const mbe_soft_bit fr[8][23], char imbe_d[88], mbe_process_result* result) {
    if (result) {
        memset(result, 0, sizeof(*result));
    }
    if (!imbe_d) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_soft(&fr[0][0], 8u * 23u);
    if (rc < 0) {
        return rc;
    }
    mbx_param_record rec;
    int total = mbxo_fec_imbe7200x4400_soft(fr, &rec);
    mbxo_record_to_bits(&rec, 88, imbe_d);
    if (result) {
        mbxo_record_to_result(&rec, result);
    }
    return total;
}

int
mbxo_decode_ambe3600x2450_soft_frame(const mbe_soft_bit fr[4][24], char ambe_d[49], mbe_process_result* result) {
    if (result) {
        memset(result, 0, sizeof(*result));
    }
    if (!ambe_d) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_soft(&fr[0][0], 4u * 24u);
    if (rc < 0) {
        return rc;
    }
    mbx_param_record rec;
    int total = mbxo_fec_ambe3600x2450_soft(fr, &rec);
    mbxo_record_to_bits(&rec, 49, ambe_d);
    if (result) {
        mbxo_record_to_result(&rec, result);
    }
    return total;
}

int
mbxo_fec_soft_batch(int codec, size_t n, const mbe_soft_bit* soft, mbx_param_record* records) {
    for (size_t i = 0; i < n; ++i) {
        if (codec == MBX_CODEC_IMBE7200X4400) {
            mbxo_fec_imbe7200x4400_soft((const mbe_soft_bit(*)[23])(soft + i * MBX_IMBE_SOFT_BITS), &records[i]);
        } else if (codec == MBX_CODEC_IMBE7100X4400) {
            mbxo_fec_imbe7100x4400_soft((const mbe_soft_bit(*)[24])(soft + i * MBX_IMBE7100_SOFT_BITS), &records[i]);
        } else {
            mbxo_fec_ambe3600x2450_soft((const mbe_soft_bit(*)[24])(soft + i * MBX_AMBE_SOFT_BITS), &records[i]);
        }
    }
    return 0;
}

int
mbxo_decode_imbe7200x4400_frame(const char fr[8][23], char imbe_d[88], mbe_process_result* result) {
    if (result) {
        memset(result, 0, sizeof(*result));
    }
    if (!imbe_d) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    uint8_t packed[MBX_IMBE_FRAME_BYTES];
    int rc = mbxo_pack_imbe_frame(fr, packed);
    if (rc < 0) {
        return rc;
    }
    mbx_param_record rec;
    int total = mbxo_fec_imbe7200x4400(packed, &rec);
    mbxo_record_to_bits(&rec, 88, imbe_d);
    if (result) {
        mbxo_record_to_result(&rec, result);
    }
    return total;
}

int
mbxo_decode_ambe3600x2450_frame(const char fr[4][24], char ambe_d[49], mbe_process_result* result) {
    if (result) {
        memset(result, 0, sizeof(*result));
    }
    if (!ambe_d) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    uint8_t packed[MBX_AMBE_FRAME_BYTES];
    int rc = mbxo_pack_ambe_frame(fr, packed);
    if (rc < 0) {
        return rc;
    }
    mbx_param_record rec;
    int total = mbxo_fec_ambe3600x2450(packed, &rec);
    mbxo_record_to_bits(&rec, 49, ambe_d);
    if (result) {
        mbxo_record_to_result(&rec, result);
    }
    return total;
}

/* =====================================================================================
 * State helpers (a12, a22)
 * ===================================================================================== */

static int
valid_L(int L) {
    return L >= 1 && L <= MBX_MAX_BANDS;
}

/* ref: src/core/mbelib.c:367-410 (mbe_initMbeParms) */
void
mbxo_init_parms(mbe_parms* cur, mbe_parms* prev, mbe_parms* prev_enh) {
    if (!cur || !prev || !prev_enh) {
        return;
    }
    prev->swn = 0;
    prev->tonePhase = 0;
    prev->w0 = (float)((4.0 * M_PI) / (134.0 + 39.5));
    prev->L = (int)(0.9254 * (int)((M_PI / prev->w0) + 0.25));
    prev->K = 12;
    prev->gamma = 0.0f;
    for (int l = 0; l <= 56; ++l) {
        prev->Ml[l] = 1.0f;
        prev->Vl[l] = 0;
        prev->log2Ml[l] = 0.0f;
        prev->PHIl[l] = 0.0f;
        prev->PSIl[l] = 0.0f;
    }
    prev->localEnergy = 75000.0f;
    prev->amplitudeThreshold = 20480;
    prev->errorRate = 0.0f;
    prev->errorCountTotal = 0;
    prev->errorCount4 = 0;
    prev->repeatCount = 0;
    prev->mutingThreshold = MBE_MUTING_THRESHOLD_IMBE;
    prev->noiseSeed = -1.0f;
    memset(prev->noiseOverlap, 0, sizeof(prev->noiseOverlap));
    memset(prev->previousUw, 0, sizeof(prev->previousUw));
    *cur = *prev;
    *prev_enh = *prev;
}

/* ref: src/ambe/ambe_common.c:191-229 (mbe_initAmbeParms_common) */
static void
init_ambe_parms(mbe_parms* cur, mbe_parms* prev, mbe_parms* prev_enh) {
    prev->swn = 0;
    prev->tonePhase = 0;
    prev->w0 = (float)((M_PI / 32.0) * (2.0 * M_PI));
    prev->L = 15;
    prev->K = 0;
    prev->gamma = 0.0f;
    for (int l = 0; l <= 56; ++l) {
        prev->Ml[l] = 1.0f;
        prev->Vl[l] = 0;
        prev->log2Ml[l] = 0.0f;
        prev->PHIl[l] = 0.0f;
        prev->PSIl[l] = 0.0f;
    }
    prev->localEnergy = 75000.0f;
    prev->amplitudeThreshold = 20480;
    prev->errorRate = 0.0f;
    prev->errorCountTotal = 0;
    prev->errorCount4 = 0;
    prev->repeatCount = 0;
    prev->mutingThreshold = MBE_MUTING_THRESHOLD_AMBE;
    prev->noiseSeed = -1.0f;
    memset(prev->noiseOverlap, 0, sizeof(prev->noiseOverlap));
    memset(prev->previousUw, 0, sizeof(prev->previousUw));
    *cur = *prev;
    *prev_enh = *prev;
}

/* ref: src/ambe/ambe_common.c:231-260 (mbe_setAmbeErasureParms_common) */
static void
set_ambe_erasure_parms(mbe_parms* mp, const mbe_parms* keep) {
    mp->swn = 0;
    mp->tonePhase = 0;
    mp->w0 = 0.0f;
    mp->L = 9;
    mp->K = 0;
    mp->gamma = 0.0f;
    for (int l = 0; l <= 56; ++l) {
        mp->Ml[l] = 1.0f;
        mp->Vl[l] = 0;
        mp->log2Ml[l] = 0.0f;
        mp->PHIl[l] = keep->PHIl[l];
        mp->PSIl[l] = keep->PSIl[l];
    }
    mp->localEnergy = 75000.0f;
    mp->amplitudeThreshold = 20480;
    mp->noiseSeed = keep->noiseSeed;
    memmove(mp->noiseOverlap, keep->noiseOverlap, sizeof(mp->noiseOverlap));
    memmove(mp->previousUw, keep->previousUw, sizeof(mp->previousUw));
}

/* the reference's thread-local defaults: src/core/mbe_adaptive.c:29-30,
 * src/core/mbe_unvoiced_fft.c:29-30 */
void
mbxo_rng_default(mbx_stream_rng* rng) {
    memset(rng, 0, sizeof(*rng));
    rng->unvoiced_seed_state = 3147u;
}

/* ref: src/core/mbelib.c:173-181, src/core/mbe_adaptive.c:32-39, src/core/mbe_unvoiced_fft.c:295-302 */
void
mbxo_rng_seed(mbx_stream_rng* rng, uint32_t seed) {
    if (seed == 0u) {
        seed = 0x6d25357bu;
    }
    rng->cn_seed48 = (((uint64_t)seed) ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1ULL);
    rng->cn_seeded = 1;
    rng->unvoiced_seed_state = seed % 53125u;
    rng->unvoiced_seed_override = 1;
}

/* =====================================================================================
 * Result bookkeeping (a2)  ref: src/internal/mbe_result.h:45-121
 * ===================================================================================== */

#define CONTEXT_FLAGS (MBE_PROCESS_FLAG_SOFT_INPUT | MBE_PROCESS_FLAG_C0_VALID | MBE_PROCESS_FLAG_C4_VALID)
#define STATUS_FLAGS  (MBE_PROCESS_FLAG_TONE | MBE_PROCESS_FLAG_ERASURE | MBE_PROCESS_FLAG_REPEAT | MBE_PROCESS_FLAG_MUTE)

static int
count_ok(int c) {
    return c >= 0 && c <= 184;
}

static int
resolve_total_errors(const mbe_process_result* r, int* total_out) {
    if (!r) {
        *total_out = 0;
        return 0;
    }
    if ((r->flags & ~(CONTEXT_FLAGS | STATUS_FLAGS)) != 0u) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (!count_ok(r->c0_errors) || !count_ok(r->protected_errors) || !count_ok(r->c4_errors)
        || !count_ok(r->total_errors) || r->c0_errors > 184 - r->protected_errors) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int parts = r->c0_errors + r->protected_errors;
    if (!count_ok(parts)) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int total = (r->total_errors == 0 && parts != 0) ? parts : r->total_errors;
    int c0v = (r->flags & MBE_PROCESS_FLAG_C0_VALID) != 0u;
    int c4v = (r->flags & MBE_PROCESS_FLAG_C4_VALID) != 0u;
    if (!((parts == 0 || total == parts) && (!c0v || total >= r->c0_errors) && (!c4v || total >= r->c4_errors))) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    *total_out = total;
    return 0;
}

static void
result_prepare_synthesis(mbe_process_result* r, int total) {
    if (!r) {
        return;
    }
    unsigned ctx = r->flags & CONTEXT_FLAGS;
    int c0 = (ctx & MBE_PROCESS_FLAG_C0_VALID) ? r->c0_errors : 0;
    int c4 = (ctx & MBE_PROCESS_FLAG_C4_VALID) ? r->c4_errors : 0;
    r->flags = ctx;
    r->c0_errors = c0;
    r->c4_errors = c4;
    r->total_errors = total;
    r->protected_errors = total - c0;
}

/* =====================================================================================
 * IMBE 7200x4400 parameter decode (a8)
 * ===================================================================================== */

/* ref: src/imbe/imbe7200x4400.c:272-281 */
static float
imbe_rho(int L) {
    if (L <= 15) {
        return 0.4f;
    }
    if (L <= 24) {
        return (0.03f * (float)L) - 0.05f;
    }
    return 0.7f;
}

static int
clampL(int L) {
    return L < 1 ? 1 : (L > 56 ? 56 : L);
}

static int
field(const int* word, int nbits) { /* low nbits of a parameter word; nbits <= 0 reads as 0 */
    return nbits > 0 ? (*word & ((1 << nbits) - 1)) : 0;
}

/* ref: src/imbe/imbe7200x4400.c:589-630 (mbe_decodeImbe4400Parms) and its helpers :117-354 */
int
mbxo_decode_imbe4400_parms(const char* imbe_d, mbe_parms* cur, mbe_parms* prev) {
    if (!cur || !prev) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_bits(imbe_d, 88u);
    if (rc < 0) {
        return rc;
    }

    /* fundamental (:117-154): b0 from bits 0..5, 85, 86 */
    int b0 = 0;
    for (int i = 0; i < 6; ++i) {
        b0 = (b0 << 1) | imbe_d[i];
    }
    b0 = (b0 << 1) | imbe_d[85];
    b0 = (b0 << 1) | imbe_d[86];
    if (b0 > 207) {
        return 1;
    }
    cur->w0 = T->imbe_w0[b0];
    if (T->imbe_L[b0] == 0) {
        return 1;
    }
    int L = T->imbe_L[b0];
    cur->L = L;
    cur->K = T->imbe_K[b0];
    const int L9 = L - 9;

    /* bit layout (:156-168): parameter word m collects bit positions as an integer */
    int word[58];
    memset(word, 0, sizeof(word));
    for (int i = 6; i < 85; ++i) {
        const uint8_t* e = T->imbe_bo[L9][i - 6];
        word[e[0]] |= (int)imbe_d[i] << e[1];
    }

    /* voicing (:170-188): three harmonics share one band decision, band K-1 first */
    for (int l = 1; l <= L; ++l) {
        int band = (cur->K - 1) - ((l - 1) / 3);
        if (band < 0) {
            band = 0;
        }
        cur->Vl[l] = (word[1] >> band) & 1;
    }

    /* gains (:190-209) */
    float Gm[7];
    Gm[1] = T->imbe_B2[field(&word[2], 6)];
    for (int g = 2; g < 7; ++g) {
        float nb = T->imbe_ba[L9][g - 2][0];
        float step = T->imbe_ba[L9][g - 2][1];
        int bm = field(&word[g + 1], (int)nb);
        Gm[g] = (step * ((float)bm - exp2f(nb - 1.0f) + 0.5f));
    }

    /* 6-point inverse DCT of the gains (:211-231) */
    float Ri[7];
    for (int i = 1; i <= 6; ++i) {
        float sum = 0;
        for (int m = 1; m <= 6; ++m) {
            int am = (m == 1) ? 1 : 2;
            sum = sum + ((float)am * Gm[m] * T->imbe_ri_cos[m][i]);
        }
        Ri[i] = sum;
    }

    /* higher-order coefficients (:233-249) */
    float Cik[7][11];
    memset(Cik, 0, sizeof(Cik));
    int m = 8;
    for (int i = 1; i <= 6; ++i) {
        Cik[i][1] = Ri[i];
        for (int k = 2; k <= T->imbe_ji[L9][i - 1]; ++k) {
            int Bm = T->imbe_hoba[L9][m - 8];
            if (Bm <= 0) {
                Cik[i][k] = 0;
            } else {
                int bm = field(&word[m], Bm);
                Cik[i][k] = ((T->imbe_quantstep[Bm - 1] * T->imbe_standdev[k - 2])
                             * (((float)bm - exp2f((float)Bm - 1.0f)) + 0.5f));
            }
            ++m;
        }
    }

    /* per-block inverse DCT (:251-270) */
    float Tl[57];
    memset(Tl, 0, sizeof(Tl));
    int l = 1;
    for (int i = 1; i <= 6; ++i) {
        int ji = T->imbe_ji[L9][i - 1];
        for (int j = 1; j <= ji; ++j) {
            float sum = 0;
            for (int k = 1; k <= ji; ++k) {
                int ak = (k == 1) ? 1 : 2;
                sum = sum + ((float)ak * Cik[i][k] * T->imbe_idct_cos[ji][j][k]);
            }
            Tl[l++] = sum;
        }
    }

    /* log-magnitude prediction (:294-354) -- also pads the PREVIOUS model */
    const float rho = imbe_rho(cur->L);
    const int cur_L = clampL(cur->L);
    const int prev_L = clampL(prev->L);
    cur->L = cur_L;
    if (cur_L > prev_L) {
        for (int q = prev_L + 1; q <= cur_L; ++q) {
            prev->Ml[q] = prev->Ml[prev_L];
            prev->log2Ml[q] = prev->log2Ml[prev_L];
        }
    }
    prev->log2Ml[0] = prev->log2Ml[1];
    prev->Ml[0] = prev->Ml[1];

    int lo[57];
    float frac[57];
    float Sum77 = 0;
    for (int q = 1; q <= cur_L; ++q) {
        float pos = ((float)prev_L / (float)cur_L) * (float)q;
        lo[q] = (int)pos;
        if (lo[q] < 0) {
            lo[q] = 0;
        } else if (lo[q] > 56) {
            lo[q] = 56;
        }
        frac[q] = pos - (float)lo[q];
        int hi = lo[q] + 1 > 56 ? 56 : lo[q] + 1;
        Sum77 = Sum77 + ((((float)1 - frac[q]) * prev->log2Ml[lo[q]]) + (frac[q] * prev->log2Ml[hi]));
    }
    Sum77 = ((rho / (float)cur_L) * Sum77);
    for (int q = 1; q <= cur_L; ++q) {
        int hi = lo[q] + 1 > 56 ? 56 : lo[q] + 1;
        float c1 = (rho * ((float)1 - frac[q]) * prev->log2Ml[lo[q]]);
        float c2 = (rho * frac[q] * prev->log2Ml[hi]);
        cur->log2Ml[q] = Tl[q] + c1 + c2 - Sum77;
        cur->Ml[q] = exp2f(cur->log2Ml[q]);
    }
    return 0;
}

/* =====================================================================================
 * AMBE+2 3600x2450 parameter decode (a9)
 * ===================================================================================== */

static int
bits_msb(const char* d, const int* idx, int n) {
    int v = 0;
    for (int i = 0; i < n; ++i) {
        v = (v << 1) | d[idx[i]];
    }
    return v;
}

/* ref: src/internal/mbe_tone.h:14-59 */
static const float dual_tone_hz[36][2] = {
    {1336, 941}, {1209, 697}, {1336, 697}, {1477, 697}, {1209, 770}, {1336, 770}, {1477, 770}, {1209, 852}, {1336, 852},
    {1477, 852}, {1633, 697}, {1633, 770}, {1633, 852}, {1633, 941}, {1209, 941}, {1477, 941}, {1162, 820}, {1052, 606},
    {1162, 606}, {1279, 606}, {1052, 672}, {1162, 672}, {1279, 672}, {1052, 743}, {1162, 743}, {1279, 743}, {1430, 606},
    {1430, 672}, {1430, 743}, {1430, 820}, {1052, 820}, {1279, 820}, {440, 350},  {480, 440},  {620, 480},  {490, 350},
};

static int
tone_freqs(int id, float* f1, float* f2) {
    *f1 = 0.0f;
    *f2 = 0.0f;
    if (id == 5) {
        *f1 = *f2 = 156.25f;
        return 1;
    }
    if (id == 6) {
        *f1 = *f2 = 187.5f;
        return 1;
    }
    if (id >= 7 && id <= 122) {
        *f1 = *f2 = 31.25f * (float)id;
        return 1;
    }
    if (id >= 128 && id <= 163) {
        *f1 = dual_tone_hz[id - 128][0];
        *f2 = dual_tone_hz[id - 128][1];
        return 1;
    }
    return 0;
}

/* prev->log2Ml[idx] where the reference may index one past the array (idx == 57 aliases
 * PHIl[0] in the struct); ref: src/ambe/ambe3600x2450.c:427,444 */
static float
log2ml_at(const mbe_parms* p, int idx) {
    return idx <= 56 ? p->log2Ml[idx] : p->PHIl[idx - 57];
}

/* ref: src/ambe/ambe3600x2450.c:564-621 (mbe_decodeAmbe2450ParmsInternal) and helpers :176-553.
 * Returns 0 voice, 2 erasure, 7 tone. */
int
mbxo_decode_ambe2450_parms(const char* d, mbe_parms* cur, mbe_parms* prev, int total_errors) {
    if (!cur || !prev) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_bits(d, 49u);
    if (rc < 0) {
        return rc;
    }

    /* tone classification (:474-519) */
    int u0 = 0, u1 = 0, u3 = 0;
    for (int i = 0; i < 12; ++i) {
        u0 = (u0 << 1) | d[i];
    }
    for (int i = 12; i < 24; ++i) {
        u1 = (u1 << 1) | d[i];
    }
    for (int i = 35; i < 49; ++i) {
        u3 = (u3 << 1) | d[i];
    }
    int tone_sig = (((u0 >> 6) & 0x3f) == 63) && (((u3 & 0xf) == 0) || (((u1 >> 8) & 0xf) == (u1 & 0xf)));
    if (tone_sig && total_errors < 6) {
        return 7;
    }

    static const int ib0[7] = {0, 1, 2, 3, 37, 38, 39};
    int b0 = bits_msb(d, ib0, 7);
    int L;
    float f0;
    int silence = 0;
    if (b0 >= 120 && b0 <= 123) {
        return 2;
    }
    if (b0 == 124 || b0 == 125) { /* silence model (:493-504) */
        silence = 1;
        f0 = (float)M_PI / 32.0f;
        cur->w0 = f0 * (float)(2.0 * M_PI);
        L = (b0 == 124) ? 15 : 14;
        cur->L = L;
        for (int l = 1; l <= L; ++l) {
            cur->Vl[l] = 0;
        }
    } else if (b0 == 126 || b0 == 127) {
        return 2;
    } else {
        f0 = T->ambe_w0[b0];
        cur->w0 = f0 * (float)2 * M_PI;
        L = T->ambe_L[b0];
        cur->L = L;
    }

    float unvc = (float)0.2046 / sqrtf(cur->w0);

    /* V/UV (:197-219) */
    static const int ib1[5] = {4, 5, 6, 7, 35};
    int b1 = bits_msb(d, ib1, 5);
    if (!silence) {
        for (int l = 1; l <= L; ++l) {
            int jl = (int)((float)l * (float)16.0 * f0);
            cur->Vl[l] = T->ambe_vuv[b1][jl];
        }
    }

    /* gain (:598-607) */
    static const int ib2[5] = {8, 9, 10, 11, 36};
    int b2 = bits_msb(d, ib2, 5);
    cur->gamma = T->ambe_dg[b2] + ((float)0.5 * prev->gamma);

    /* PRBA -> Ri by 8-point inverse DCT (:221-273) */
    static const int ib3[9] = {12, 13, 14, 15, 16, 17, 18, 19, 40};
    static const int ib4[7] = {20, 21, 22, 23, 41, 42, 43};
    int b3 = bits_msb(d, ib3, 9);
    int b4 = bits_msb(d, ib4, 7);
    float Gm[9];
    Gm[1] = 0;
    Gm[2] = T->ambe_prba24[b3][0];
    Gm[3] = T->ambe_prba24[b3][1];
    Gm[4] = T->ambe_prba24[b3][2];
    Gm[5] = T->ambe_prba58[b4][0];
    Gm[6] = T->ambe_prba58[b4][1];
    Gm[7] = T->ambe_prba58[b4][2];
    Gm[8] = T->ambe_prba58[b4][3];
    float Ri[9];
    for (int i = 1; i <= 8; ++i) {
        float sum = 0;
        for (int m = 1; m <= 8; ++m) {
            int am = (m == 1) ? 1 : 2;
            sum = sum + ((float)am * Gm[m] * T->ambe_ri_cos[m][i]);
        }
        Ri[i] = sum;
    }

    /* block coefficients (:275-363) */
    float Cik[5][18];
    memset(Cik, 0, sizeof(Cik));
    const float rconst = ((float)1 / ((float)2 * M_SQRT2));
    for (int i = 1; i <= 4; ++i) {
        Cik[i][1] = (float)0.5 * (Ri[2 * i - 1] + Ri[2 * i]);
        Cik[i][2] = rconst * (Ri[2 * i - 1] - Ri[2 * i]);
    }
    static const int ib5[5] = {24, 25, 26, 27, 44};
    static const int ib6[4] = {28, 29, 30, 45};
    static const int ib7[4] = {31, 32, 33, 46};
    static const int ib8[3] = {34, 47, 48};
    const float* hoc[5] = {NULL, T->ambe_hoc_b5[bits_msb(d, ib5, 5)], T->ambe_hoc_b6[bits_msb(d, ib6, 4)],
                           T->ambe_hoc_b7[bits_msb(d, ib7, 4)], T->ambe_hoc_b8[bits_msb(d, ib8, 3)]};
    int Ji[5];
    for (int i = 1; i <= 4; ++i) {
        Ji[i] = T->ambe_lmprbl[L][i - 1];
        for (int k = 3; k <= Ji[i]; ++k) {
            Cik[i][k] = (k > 6) ? 0.0f : hoc[i][k - 3];
        }
    }

    /* per-block inverse DCT (:365-387) */
    float Tl[57];
    memset(Tl, 0, sizeof(Tl));
    int l = 1;
    for (int i = 1; i <= 4; ++i) {
        int ji = Ji[i];
        for (int j = 1; j <= ji; ++j) {
            float sum = 0;
            for (int k = 1; k <= ji; ++k) {
                int ak = (k == 1) ? 1 : 2;
                sum = sum + ((float)ak * Cik[i][k] * T->ambe_idct_cos[ji][j][k]);
            }
            Tl[l++] = sum;
        }
    }

    /* log-magnitude prediction (:389-459) */
    int prev_L = clampL(prev->L);
    cur->L = clampL(cur->L);
    if (cur->L > prev_L) {
        for (int q = prev_L + 1; q <= cur->L; ++q) {
            prev->Ml[q] = prev->Ml[prev_L];
            prev->log2Ml[q] = prev->log2Ml[prev_L];
        }
    }
    prev->log2Ml[0] = prev->log2Ml[1];
    prev->Ml[0] = prev->Ml[1];

    int lo[57];
    float frac[57];
    float Sum43 = 0;
    for (int q = 1; q <= cur->L; ++q) {
        float pos = ((float)prev_L / (float)cur->L) * (float)q;
        lo[q] = (int)pos;
        frac[q] = pos - (float)lo[q];
        Sum43 = Sum43 + ((((float)1 - frac[q]) * log2ml_at(prev, lo[q])) + (frac[q] * log2ml_at(prev, lo[q] + 1)));
    }
    Sum43 = (((float)0.65 / (float)cur->L) * Sum43);
    float Sum42 = 0;
    for (int q = 1; q <= cur->L; ++q) {
        Sum42 += Tl[q];
    }
    Sum42 = Sum42 / (float)cur->L;
    float BigGamma = cur->gamma - (0.5f * log2f((float)cur->L)) - Sum42;
    for (int q = 1; q <= cur->L; ++q) {
        float c1 = ((float)0.65 * ((float)1 - frac[q]) * log2ml_at(prev, lo[q]));
        float c2 = ((float)0.65 * frac[q] * log2ml_at(prev, lo[q] + 1));
        cur->log2Ml[q] = Tl[q] + c1 + c2 - Sum43 + BigGamma;
        if (cur->Vl[q] == 1) {
            cur->Ml[q] = exp2f(cur->log2Ml[q]);
        } else {
            cur->Ml[q] = unvc * exp2f(cur->log2Ml[q]);
        }
    }
    return 0;
}

/* =====================================================================================
 * AMBE 3600x2400 (D-STAR) parameter decode  ref: src/ambe/ambe3600x2400.c:164-551
 * Returns 0 voice, 3 tone-class frame without a usable index (a silence model may have been set),
 * or the tone index 5..122.
 * ===================================================================================== */
int
mbxo_decode_ambe2400_parms(const char* d, mbe_parms* cur, mbe_parms* prev) {
    if (!cur || !prev) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_bits(d, 49u);
    if (rc < 0) {
        return rc;
    }
    static const int ib0[7] = {0, 1, 2, 3, 4, 5, 48};
    const int b0 = bits_msb(d, ib0, 7);
    int L = 0;

    if ((b0 & 0x7E) == 0x7E) { /* tone class (:212-234) */
        static const int t7[8] = {1, 0, 0, 0, 0, 1, 1, 1}, t6[8] = {0, 0, 0, 1, 1, 1, 1, 0}, t5[8] = {0, 0, 1, 0, 1, 1, 0, 1};
        const int def = (d[6] << 2) | (d[7] << 1) | d[8];
        const int tone = (t7[def] << 7) | (t6[def] << 6) | (t5[def] << 5) | (d[9] << 4) | (d[42] << 3) | (d[43] << 2)
                         | (d[10] << 1) | d[11];
        if (tone >= 5 && tone <= 122) {
            return tone;
        }
        if (!(tone >= 128 && tone <= 163)) { /* silence model (:202-210) */
            cur->w0 = ((float)2 * M_PI) / (float)32;
            cur->L = 14;
            for (int l = 1; l <= 14; ++l) {
                cur->Vl[l] = 0;
            }
        }
        return 3;
    }

    const float f0 = exp2f(-4.311767578125f - (2.1336e-2f * ((float)b0 + 0.5f)));
    cur->w0 = f0 * (float)2 * M_PI;
    L = T->ambep_L[b0];
    cur->L = L;
    const float unvc = (float)0.2046 / sqrtf(cur->w0);

    static const int ib1[4] = {38, 39, 40, 41};
    const int b1 = bits_msb(d, ib1, 4);
    for (int l = 1; l <= L; ++l) {
        int jl = (int)((float)l * (float)16.0 * f0);
        cur->Vl[l] = T->ambep_vuv[b1][jl];
    }

    static const int ib2[6] = {6, 7, 8, 9, 42, 43};
    cur->gamma = T->ambep_dg[bits_msb(d, ib2, 6)] + ((float)0.5 * prev->gamma);

    static const int ib3[9] = {10, 11, 12, 13, 14, 15, 16, 44, 45};
    static const int ib4[7] = {17, 18, 19, 20, 21, 46, 47};
    const int b3 = bits_msb(d, ib3, 9), b4 = bits_msb(d, ib4, 7);
    float Gm[9];
    Gm[1] = 0;
    Gm[2] = T->ambep_prba24[b3][0];
    Gm[3] = T->ambep_prba24[b3][1];
    Gm[4] = T->ambep_prba24[b3][2];
    Gm[5] = T->ambep_prba58[b4][0];
    Gm[6] = T->ambep_prba58[b4][1];
    Gm[7] = T->ambep_prba58[b4][2];
    Gm[8] = T->ambep_prba58[b4][3];
    float Ri[9];
    for (int i = 1; i <= 8; ++i) {
        float sum = 0;
        for (int m = 1; m <= 8; ++m) {
            int am = (m == 1) ? 1 : 2;
            sum = sum + ((float)am * Gm[m] * T->ambe_ri_cos[m][i]);
        }
        Ri[i] = sum;
    }

    float Cik[5][18];
    memset(Cik, 0, sizeof(Cik));
    const float rconst = ((float)1 / ((float)2 * M_SQRT2));
    for (int i = 1; i <= 4; ++i) {
        Cik[i][1] = (float)0.5 * (Ri[2 * i - 1] + Ri[2 * i]);
        Cik[i][2] = rconst * (Ri[2 * i - 1] - Ri[2 * i]);
    }
    static const int ib5[4] = {22, 23, 25, 26};
    static const int ib6[4] = {27, 28, 29, 30};
    static const int ib7[4] = {31, 32, 33, 34};
    const int b8 = (d[35] << 3) | (d[36] << 2) | (d[37] << 1);
    const float* hoc[5] = {NULL, T->ambep_hoc_b5[bits_msb(d, ib5, 4)], T->ambep_hoc_b6[bits_msb(d, ib6, 4)],
                           T->ambep_hoc_b7[bits_msb(d, ib7, 4)], T->ambep_hoc_b8[b8]};
    int Ji[5];
    for (int i = 1; i <= 4; ++i) {
        Ji[i] = T->ambep_lmprbl[L][i - 1];
        for (int k = 3; k <= Ji[i]; ++k) {
            Cik[i][k] = (k > 6) ? 0.0f : hoc[i][k - 3];
        }
    }

    float Tl[57];
    memset(Tl, 0, sizeof(Tl));
    int l = 1;
    for (int i = 1; i <= 4; ++i) {
        int ji = Ji[i];
        for (int j = 1; j <= ji; ++j) {
            float sum = 0;
            for (int k = 1; k <= ji; ++k) {
                int ak = (k == 1) ? 1 : 2;
                sum = sum + ((float)ak * Cik[i][k] * T->ambe_idct_cos[ji][j][k]);
            }
            Tl[l++] = sum;
        }
    }

    /* log-magnitude prediction (:427-497): identical to AMBE+2 */
    int prev_L = clampL(prev->L);
    cur->L = clampL(cur->L);
    if (cur->L > prev_L) {
        for (int q = prev_L + 1; q <= cur->L; ++q) {
            prev->Ml[q] = prev->Ml[prev_L];
            prev->log2Ml[q] = prev->log2Ml[prev_L];
        }
    }
    prev->log2Ml[0] = prev->log2Ml[1];
    prev->Ml[0] = prev->Ml[1];
    int lo[57];
    float frac[57];
    float Sum43 = 0;
    for (int q = 1; q <= cur->L; ++q) {
        float pos = ((float)prev_L / (float)cur->L) * (float)q;
        lo[q] = (int)pos;
        frac[q] = pos - (float)lo[q];
        Sum43 = Sum43 + ((((float)1 - frac[q]) * log2ml_at(prev, lo[q])) + (frac[q] * log2ml_at(prev, lo[q] + 1)));
    }
    Sum43 = (((float)0.65 / (float)cur->L) * Sum43);
    float Sum42 = 0;
    for (int q = 1; q <= cur->L; ++q) {
        Sum42 += Tl[q];
    }
    Sum42 = Sum42 / (float)cur->L;
    float BigGamma = cur->gamma - (0.5f * log2f((float)cur->L)) - Sum42;
    for (int q = 1; q <= cur->L; ++q) {
        float c1 = ((float)0.65 * ((float)1 - frac[q]) * log2ml_at(prev, lo[q]));
        float c2 = ((float)0.65 * frac[q] * log2ml_at(prev, lo[q] + 1));
        cur->log2Ml[q] = Tl[q] + c1 + c2 - Sum43 + BigGamma;
        if (cur->Vl[q] == 1) {
            cur->Ml[q] = exp2f(cur->log2Ml[q]);
        } else {
            cur->Ml[q] = unvc * exp2f(cur->log2Ml[q]);
        }
    }
    return 0;
}

/* =====================================================================================
 * Spectral amplitude enhancement (a13)  ref: src/core/mbelib.c:412-666
 * ===================================================================================== */
float
mbxo_spectral_amp_enhance(mbe_parms* cur) {
    if (!cur || !valid_L(cur->L)) {
        return 0.0f;
    }
    const int L = cur->L;
    float cw[57];
    float s_step, c_step;
    sincosf(cur->w0, &s_step, &c_step);
    float c = 1.0f, s = 0.0f;
    for (int l = 1; l <= L; ++l) { /* cos(l*w0) by rotation, as the reference does */
        float cn = (c * c_step) - (s * s_step);
        float sn = (s * c_step) + (c * s_step);
        c = cn;
        s = sn;
        cw[l] = c;
    }
    float Rm0 = 0.0f, Rm1 = 0.0f;
    for (int l = 1; l <= L; ++l) {
        const float Ml2 = cur->Ml[l] * cur->Ml[l];
        Rm0 += Ml2;
        Rm1 += Ml2 * cw[l];
    }
    const float R2m0 = Rm0 * Rm0;
    const float R2m1 = Rm1 * Rm1;
    for (int l = 1; l <= L; ++l) {
        if (cur->Ml[l] != 0.0f) {
            float Wl = sqrtf(cur->Ml[l])
                       * sqrtf(sqrtf(((float)0.96 * (float)M_PI * ((R2m0 + R2m1) - ((float)2 * Rm0 * Rm1 * cw[l])))
                                     / (cur->w0 * Rm0 * (R2m0 - R2m1))));
            if ((8 * l) <= L) {
                /* low harmonics are left alone */
            } else if (Wl > 1.2f) {
                cur->Ml[l] = 1.2f * cur->Ml[l];
            } else if (Wl < 0.5f) {
                cur->Ml[l] = 0.5f * cur->Ml[l];
            } else {
                cur->Ml[l] = Wl * cur->Ml[l];
            }
        }
    }
    float sum = 0.0f;
    for (int l = 1; l <= L; ++l) {
        float M = cur->Ml[l];
        if (M < 0.0f) {
            M = -M;
        }
        sum += M * M;
    }
    float gamma = (sum == 0.0f) ? 1.0f : sqrtf(Rm0 / sum);
    for (int l = 1; l <= L; ++l) {
        cur->Ml[l] = gamma * cur->Ml[l];
    }
    return Rm0;
}

/* =====================================================================================
 * Adaptive smoothing (a14)  ref: src/core/mbe_adaptive.c:151-276
 * ===================================================================================== */
static void
adaptive_smoothing_core(mbe_parms* cur, const mbe_parms* prev, float RM0) {
    const int L = cur->L;
    const float er = cur->errorRate;
    const int et = cur->errorCountTotal;
    const int e4 = cur->errorCount4;

    float pe = prev->localEnergy;
    if (pe < 10000.0f) {
        pe = 75000.0f;
    }
    float le = 0.95f * pe + 0.05f * RM0;
    if (le < 10000.0f) {
        le = 10000.0f;
    }
    cur->localEnergy = le;

    float VM;
    if (er <= 0.005f && et <= 4) {
        VM = __FLT_MAX__;
    } else {
        float x8 = sqrtf(sqrtf(sqrtf(le)));
        float energy = x8 * x8 * x8;
        if (er <= 0.0125f && e4 == 0) {
            VM = (45.255f * energy) / expf(277.26f * er);
        } else {
            VM = 1.414f * energy;
        }
    }
    for (int l = 1; l <= L; ++l) {
        if (cur->Ml[l] > VM) {
            cur->Vl[l] = 1;
        }
    }
    float Am = 0.0f;
    for (int l = 1; l <= L; ++l) {
        Am += cur->Ml[l];
    }
    int pt = prev->amplitudeThreshold;
    if (pt <= 0) {
        pt = 20480;
    }
    int Tm = (er <= 0.005f && et <= 6) ? 20480 : (6000 - (300 * et) + pt);
    cur->amplitudeThreshold = Tm;
    if (Am > (float)Tm && Am > 0.0f) {
        float scale = (float)Tm / Am;
        for (int l = 1; l <= L; ++l) {
            cur->Ml[l] *= scale;
        }
    }
}

void
mbxo_adaptive_smoothing(mbe_parms* cur, const mbe_parms* prev) {
    if (!cur || !prev || !valid_L(cur->L) || !valid_L(prev->L)) {
        return;
    }
    float rm0 = 0.0f;
    for (int l = 1; l <= cur->L; ++l) {
        rm0 += cur->Ml[l] * cur->Ml[l];
    }
    adaptive_smoothing_core(cur, prev, rm0);
}

/* =====================================================================================
 * Noise sources (a15, a20)
 * ===================================================================================== */

/* ref: src/core/mbe_unvoiced_fft.c:304-341 (mbe_generate_noise_with_overlap) */
void
mbxo_noise_next(float buffer[256], float* seed, float overlap[96], mbx_stream_rng* rng) {
    if (*seed < 0.0f) {
        memset(buffer, 0, 256 * sizeof(float));
        memset(overlap, 0, 96 * sizeof(float));
        if (rng->unvoiced_seed_override) {
            *seed = (float)rng->unvoiced_seed_state;
            rng->unvoiced_seed_override = 0;
        } else {
            *seed = 3147.0f;
        }
        return;
    }
    memcpy(buffer, overlap, 96 * sizeof(float));
    unsigned int x = ((unsigned int)(*seed)) % 53125u;
    for (int i = 96; i < 256; ++i) {
        buffer[i] = (float)x;
        x = (171u * x + 11213u) % 53125u;
    }
    *seed = (float)x;
    memcpy(overlap, buffer + 160, 96 * sizeof(float));
}

/* ref: src/core/mbe_adaptive.c:50-60, :116-131 */
void
mbxo_comfort_noisef(float* out, mbx_stream_rng* rng) {
    if (!out) {
        return;
    }
    const uint64_t mask = (1ULL << 48) - 1ULL;
    const float gain = (0.003f * 32767.0f) / 7.0f;
    for (int i = 0; i < 160; ++i) {
        if (!rng->cn_seeded) {
            rng->cn_seed48 = (0x12345678ULL ^ 0x5DEECE66DULL) & mask;
            rng->cn_seeded = 1;
        }
        rng->cn_seed48 = (rng->cn_seed48 * 0x5DEECE66DULL + 0xBULL) & mask;
        uint32_t r24 = (uint32_t)(rng->cn_seed48 >> 24);
        float u = ((float)r24 / 16777216.0f) * 2.0f - 1.0f;
        out[i] = u * gain;
    }
}

/* =====================================================================================
 * 256-point real FFT pair used by the unvoiced path (stands in for PFFFT; any correct
 * transform meets the tolerance).  Double precision, results rounded to float.
 * ===================================================================================== */
static void
fft256(double* re, double* im, int inverse) {
    static int ready = 0;
    static double wr[128], wi[128];
    static int rev[256];
    if (!ready) {
        for (int i = 0; i < 128; ++i) {
            wr[i] = cos(2.0 * M_PI * i / 256.0);
            wi[i] = -sin(2.0 * M_PI * i / 256.0);
        }
        for (int i = 0; i < 256; ++i) {
            int r = 0;
            for (int b = 0; b < 8; ++b) {
                r |= ((i >> b) & 1) << (7 - b);
            }
            rev[i] = r;
        }
        ready = 1;
    }
    for (int i = 0; i < 256; ++i) {
        if (rev[i] > i) {
            double t = re[i];
            re[i] = re[rev[i]];
            re[rev[i]] = t;
            t = im[i];
            im[i] = im[rev[i]];
            im[rev[i]] = t;
        }
    }
    for (int len = 2; len <= 256; len <<= 1) {
        int half = len >> 1, step = 256 / len;
        for (int base = 0; base < 256; base += len) {
            for (int k = 0; k < half; ++k) {
                double c = wr[k * step], s = inverse ? -wi[k * step] : wi[k * step];
                int a = base + k, b = a + half;
                double tr = re[b] * c - im[b] * s;
                double ti = re[b] * s + im[b] * c;
                re[b] = re[a] - tr;
                im[b] = im[a] - ti;
                re[a] += tr;
                im[a] += ti;
            }
        }
    }
}

/* =====================================================================================
 * The reference's OWN 256-point real transform, restated in float: mbxo_set_fft_float(1).
 *
 * The reference calls PFFFT (ref src/core/mbe_unvoiced_fft.c:738,751: pffft_transform_ordered), a third-party library
 * vendored under src/external/pffft; its scalar build (what oracle/_ref compiles, PFFFT_SIMD_DISABLE) is FFTPACK's real
 * transform -- P. N. Swarztrauber's published algorithm (rfftf / rfftb with the radix-4 passes radf4 / radb4) -- followed by
 * a re-ordering of the half-complex result.  256 = 4^4, so four radix-4 passes each way and nothing else.  The arithmetic
 * below follows that algorithm operation for operation (every product and sum rounded to float in the same order, no
 * contraction), which is what makes the oracle's float PCM equal to the reference's to the last bit
 * (tests/test_oracle_golden.py: FNV hash 0x59741032 of ref tests/test_golden_pcm.c:78).  Structure and names are this
 * file's: x(i, k, j) / y(i, j, k) are the three-index views FFTPACK describes as cc(ido, l1, 4) / ch(ido, 4, l1).
 *   ref: src/external/pffft/pffft.c:749-833 (forward pass), :837-924 (backward pass), :1109-1199 (pass schedules),
 *        :1230-1262 (twiddles), :2018-2040 (ordering)
 * ===================================================================================== */
static int g_fft_float = 0;
/* Diagnostic for the parity checker (tests/parity.py int16_bound): the largest |sample| of the frame's sum BEFORE the soft clip, so that
 * the int16 bound of a frame driven far beyond the output range can be stated relative to the amplitude it was computed at.  The
 * reference keeps no such figure; nothing in the decode reads it. */
static __thread float g_preclip_peak = 0.0f;
static __thread float* g_preclip_peaks_out = 0;

void
mbxo_set_fft_float(int on) {
    g_fft_float = on ? 1 : 0;
}

static float g_rtw[256]; /* FFTPACK's real-transform twiddles for n = 256 */
static int g_rtw_ready = 0;

static void
real_twiddles_256(void) {
    const int n = 256;
    const float argh = (float)((2 * M_PI) / n);
    int is = 0, l1 = 1;
    for (int pass = 0; pass < 3; ++pass) { /* all factors but the last */
        const int l2 = l1 * 4, ido = n / l2;
        int ld = 0;
        for (int j = 1; j <= 3; ++j) {
            ld += l1;
            const float argld = (float)ld * argh;
            int i = is;
            for (int fi = 1; 2 * fi + 1 <= ido; ++fi) {
                i += 2;
                g_rtw[i - 2] = (float)cos((double)((float)fi * argld));
                g_rtw[i - 1] = (float)sin((double)((float)fi * argld));
            }
            is += ido;
        }
        l1 = l2;
    }
    g_rtw_ready = 1;
}

/* (ar + i ai) (wr - i wi) and (ar + i ai) (wr + i wi), in the library's operation order */
#define CMUL_CONJ(ar, ai, wr, wi)            \
    do {                                     \
        const float t_ = (ar) * (wi);        \
        (ar) = (ar) * (wr);                  \
        (ar) = (ar) + ((ai) * (wi));         \
        (ai) = (ai) * (wr);                  \
        (ai) = (ai) - t_;                    \
    } while (0)
#define CMUL_FWD(ar, ai, wr, wi)             \
    do {                                     \
        const float t_ = (ar) * (wi);        \
        (ar) = (ar) * (wr);                  \
        (ar) = (ar) - ((ai) * (wi));         \
        (ai) = (ai) * (wr);                  \
        (ai) = (ai) + t_;                    \
    } while (0)

/* one forward radix-4 pass: x(i, k, j), i < ido, k < l1, j < 4  ->  y(i, j, k) */
static void
real_pass4_forward(int ido, int l1, const float* x, float* y, const float* w1, const float* w2, const float* w3) {
#define X_(i, k, j) x[(i) + ido * ((k) + l1 * (j))]
#define Y_(i, j, k) y[(i) + ido * ((j) + 4 * (k))]
    const float minus_half_sqrt2 = (float)-0.7071067811865475;
    for (int k = 0; k < l1; ++k) {
        const float a0 = X_(0, k, 0), a1 = X_(0, k, 1), a2 = X_(0, k, 2), a3 = X_(0, k, 3);
        const float s13 = a1 + a3, s02 = a0 + a2;
        Y_(ido - 1, 1, k) = a0 - a2;
        Y_(0, 2, k) = a3 - a1;
        Y_(0, 0, k) = s13 + s02;
        Y_(ido - 1, 3, k) = s02 - s13;
    }
    if (ido < 2) {
        return;
    }
    if (ido != 2) {
        for (int k = 0; k < l1; ++k) {
            for (int i = 2; i < ido; i += 2) {
                const int ic = ido - i;
                float r2 = X_(i - 1, k, 1), q2 = X_(i, k, 1);
                CMUL_CONJ(r2, q2, w1[i - 2], w1[i - 1]);
                float r3 = X_(i - 1, k, 2), q3 = X_(i, k, 2);
                CMUL_CONJ(r3, q3, w2[i - 2], w2[i - 1]);
                float r4 = X_(i - 1, k, 3), q4 = X_(i, k, 3);
                CMUL_CONJ(r4, q4, w3[i - 2], w3[i - 1]);
                const float tr1 = r2 + r4, tr4 = r4 - r2;
                const float tr2 = X_(i - 1, k, 0) + r3, tr3 = X_(i - 1, k, 0) - r3;
                Y_(i - 1, 0, k) = tr1 + tr2;
                Y_(ic - 1, 3, k) = tr2 - tr1;
                const float ti1 = q2 + q4, ti4 = q2 - q4;
                Y_(i - 1, 2, k) = ti4 + tr3;
                Y_(ic - 1, 1, k) = tr3 - ti4;
                const float ti2 = X_(i, k, 0) + q3, ti3 = X_(i, k, 0) - q3;
                Y_(i, 0, k) = ti1 + ti2;
                Y_(ic, 3, k) = ti1 - ti2;
                Y_(i, 2, k) = tr4 + ti3;
                Y_(ic, 1, k) = tr4 - ti3;
            }
        }
    }
    for (int k = 0; k < l1; ++k) { /* ido is even: the middle column */
        const float a = X_(ido - 1, k, 1), b = X_(ido - 1, k, 3), c = X_(ido - 1, k, 0), d = X_(ido - 1, k, 2);
        const float ti1 = minus_half_sqrt2 * (a + b), tr1 = minus_half_sqrt2 * (b - a);
        Y_(ido - 1, 0, k) = tr1 + c;
        Y_(ido - 1, 2, k) = c - tr1;
        Y_(0, 1, k) = ti1 - d;
        Y_(0, 3, k) = ti1 + d;
    }
#undef X_
#undef Y_
}

/* one backward radix-4 pass: c(i, j, k), j < 4  ->  h(i, k, j) */
static void
real_pass4_backward(int ido, int l1, const float* c, float* h, const float* w1, const float* w2, const float* w3) {
#define C_(i, j, k) c[(i) + ido * ((j) + 4 * (k))]
#define H_(i, k, j) h[(i) + ido * ((k) + l1 * (j))]
    const float minus_sqrt2 = (float)-1.414213562373095;
    const float two = 2.f;
    for (int k = 0; k < l1; ++k) {
        const float a = C_(0, 0, k), b = C_(ido - 1, 3, k), cc = C_(0, 2, k), d = C_(ido - 1, 1, k);
        const float tr3 = two * d, tr2 = a + b, tr1 = a - b, tr4 = two * cc;
        H_(0, k, 0) = tr2 + tr3;
        H_(0, k, 2) = tr2 - tr3;
        H_(0, k, 1) = tr1 - tr4;
        H_(0, k, 3) = tr1 + tr4;
    }
    if (ido < 2) {
        return;
    }
    if (ido != 2) {
        for (int k = 0; k < l1; ++k) {
            for (int i = 2; i < ido; i += 2) {
                const int ic = ido - i;
                const float tr1 = C_(i - 1, 0, k) - C_(ic - 1, 3, k), tr2 = C_(i - 1, 0, k) + C_(ic - 1, 3, k);
                const float ti4 = C_(i - 1, 2, k) - C_(ic - 1, 1, k), tr3 = C_(i - 1, 2, k) + C_(ic - 1, 1, k);
                H_(i - 1, k, 0) = tr2 + tr3;
                float r3 = tr2 - tr3;
                const float ti3 = C_(i, 2, k) - C_(ic, 1, k), tr4 = C_(i, 2, k) + C_(ic, 1, k);
                float r2 = tr1 - tr4, r4 = tr1 + tr4;
                const float ti1 = C_(i, 0, k) + C_(ic, 3, k), ti2 = C_(i, 0, k) - C_(ic, 3, k);
                H_(i, k, 0) = ti2 + ti3;
                float q3 = ti2 - ti3, q2 = ti1 + ti4, q4 = ti1 - ti4;
                CMUL_FWD(r2, q2, w1[i - 2], w1[i - 1]);
                H_(i - 1, k, 1) = r2;
                H_(i, k, 1) = q2;
                CMUL_FWD(r3, q3, w2[i - 2], w2[i - 1]);
                H_(i - 1, k, 2) = r3;
                H_(i, k, 2) = q3;
                CMUL_FWD(r4, q4, w3[i - 2], w3[i - 1]);
                H_(i - 1, k, 3) = r4;
                H_(i, k, 3) = q4;
            }
        }
    }
    for (int k = 0; k < l1; ++k) {
        const float cc = C_(ido - 1, 0, k), d = C_(ido - 1, 2, k), a = C_(0, 1, k), b = C_(0, 3, k);
        const float tr1 = cc - d, tr2 = cc + d, ti1 = b + a, ti2 = b - a;
        H_(ido - 1, k, 0) = tr2 + tr2;
        H_(ido - 1, k, 1) = minus_sqrt2 * (ti1 - tr1);
        H_(ido - 1, k, 2) = ti2 + ti2;
        H_(ido - 1, k, 3) = minus_sqrt2 * (ti1 + tr1);
    }
#undef C_
#undef H_
}

/* forward: 256 real samples -> [X0.re, X128.re, X1.re, X1.im, ..., X127.re, X127.im] (the library's "ordered" layout) */
static void
real_fft256_forward(const float in[256], float out[256]) {
    if (!g_rtw_ready) {
        real_twiddles_256();
    }
    float a[256], b[256];
    /* (l1, ido, twiddle offset): 64,1,252 -> 16,4,240 -> 4,16,192 -> 1,64,0 */
    real_pass4_forward(1, 64, in, a, &g_rtw[252], &g_rtw[253], &g_rtw[254]);
    real_pass4_forward(4, 16, a, b, &g_rtw[240], &g_rtw[244], &g_rtw[248]);
    real_pass4_forward(16, 4, b, a, &g_rtw[192], &g_rtw[208], &g_rtw[224]);
    real_pass4_forward(64, 1, a, b, &g_rtw[0], &g_rtw[64], &g_rtw[128]);
    out[0] = b[0];
    out[1] = b[255];
    for (int k = 255; k > 1; --k) {
        out[k] = b[k - 1];
    }
}

/* backward of the above, unnormalised (backward(forward(x)) = 256 x) */
static void
real_fft256_backward(const float in[256], float out[256]) {
    if (!g_rtw_ready) {
        real_twiddles_256();
    }
    float a[256], b[256];
    a[0] = in[0];
    a[255] = in[1];
    for (int k = 1; k < 255; ++k) {
        a[k] = in[k + 1];
    }
    real_pass4_backward(64, 1, a, b, &g_rtw[0], &g_rtw[64], &g_rtw[128]);
    real_pass4_backward(16, 4, b, a, &g_rtw[192], &g_rtw[208], &g_rtw[224]);
    real_pass4_backward(4, 16, a, b, &g_rtw[240], &g_rtw[244], &g_rtw[248]);
    real_pass4_backward(1, 64, b, out, &g_rtw[252], &g_rtw[253], &g_rtw[254]);
}

/* =====================================================================================
 * Unvoiced synthesis (a18)  ref: src/core/mbe_unvoiced_fft.c:643-761
 * ===================================================================================== */
static void
synth_unvoiced(float* out, mbe_parms* cur, const mbe_parms* prev, const float noise[256]) {
    if (!valid_L(cur->L) || !valid_L(prev->L)) {
        return;
    }
    /* window, forward transform */
    double re[256], im[256];
    float Xr[129], Xi[129];
    float fbuf[256], fspec[256];
    if (g_fft_float) { /* the reference's own float transform (see above) */
        for (int i = 0; i < 256; ++i) {
            fbuf[i] = noise[i] * T->uv_window[i];
        }
        real_fft256_forward(fbuf, fspec);
        Xr[0] = fspec[0];
        Xi[0] = 0.0f;
        Xr[128] = fspec[1];
        Xi[128] = 0.0f;
        for (int k = 1; k < 128; ++k) {
            Xr[k] = fspec[2 * k];
            Xi[k] = fspec[2 * k + 1];
        }
    } else {
        for (int i = 0; i < 256; ++i) {
            re[i] = (double)(noise[i] * T->uv_window[i]);
            im[i] = 0.0;
        }
        fft256(re, im, 0);
        for (int k = 0; k <= 128; ++k) {
            Xr[k] = (float)re[k];
            Xi[k] = (k == 0 || k == 128) ? 0.0f : (float)im[k];
        }
    }

    /* band edges (:643-661) and per-band scale (:663-686); voiced/uncovered bins stay 0 */
    float scale[129];
    memset(scale, 0, sizeof(scale));
    const float mult = (256.0f / (2.0f * 3.14159265358979323846f)) * cur->w0;
    for (int l = 1; l <= cur->L; ++l) {
        int a = (int)ceilf((l - 0.5f) * mult);
        int b = (int)ceilf((l + 0.5f) * mult);
        if (a < 0) {
            a = 0;
        }
        if (b > 128) {
            b = 128;
        }
        if (cur->Vl[l] != 0) {
            continue;
        }
        /* energy of bins [a,b) in the reference's order: DC first, interior ascending,
         * Nyquist last (:546-641) */
        float num = 0.0f;
        if (b > a) {
            int s = a, e = b;
            if (s == 0) {
                num += Xr[0] * Xr[0];
                s = 1;
            }
            int nyq = (e > 128);
            if (nyq) {
                e = 128;
            }
            for (int k = s; k < e; ++k) {
                num += (Xr[k] * Xr[k]) + (Xi[k] * Xi[k]);
            }
            if (nyq) {
                num += Xr[128] * Xr[128];
            }
        }
        int count = b - a;
        if (count > 0 && num > 1e-10f) {
            float sc = 146.17696f * cur->Ml[l] / sqrtf(num / (float)count);
            for (int k = a; k < b; ++k) {
                scale[k] = sc;
            }
        }
    }

    /* scale, inverse transform, 1/256 */
    float Uw[256];
    if (g_fft_float) { /* ref src/core/mbe_unvoiced_fft.c:236-272 (bins scaled in the ordered layout), :689-712 (1/256) */
        fspec[0] *= scale[0];
        for (int k = 1; k < 128; ++k) {
            fspec[2 * k] *= scale[k];
            fspec[2 * k + 1] *= scale[k];
        }
        fspec[1] *= scale[128];
        real_fft256_backward(fspec, Uw);
        const float norm = 1.0f / (float)256;
        for (int i = 0; i < 256; ++i) {
            Uw[i] *= norm;
        }
    } else {
        for (int k = 0; k <= 128; ++k) {
            re[k] = (double)(Xr[k] * scale[k]);
            im[k] = (double)(Xi[k] * scale[k]);
        }
        for (int k = 1; k < 128; ++k) {
            re[256 - k] = re[k];
            im[256 - k] = -im[k];
        }
        im[0] = 0.0;
        im[128] = 0.0;
        fft256(re, im, 1);
        for (int i = 0; i < 256; ++i) {
            Uw[i] = (float)re[i] * (1.0f / 256.0f);
        }
    }

    /* weighted overlap-add with the previous block (:385-411, :511-530) */
    for (int n = 0; n < 160; ++n) {
        float ps = (n + 128 < 256) ? prev->previousUw[n + 128] : 0.0f;
        float cs = (n - 32 >= 0) ? Uw[n - 32] : 0.0f;
        float d = T->wola_denom[n];
        if (d > 1e-10f) {
            out[n] += ((T->wola_w_prev[n] * ps) + (T->wola_w_curr[n] * cs)) / d;
        }
    }
    memcpy(cur->previousUw, Uw, sizeof(Uw));
}

/* =====================================================================================
 * Speech synthesis (a16, a17, a19)  ref: src/core/mbelib.c:895-1115
 * ===================================================================================== */
static void
synth_core(float* out, mbe_parms* cur, mbe_parms* prev, int have_rm0, float rm0, mbx_stream_rng* rng) {
    const int N = 160;
    const float TWO_PI = 2.0f * (float)M_PI;
    if (!out) {
        return;
    }
    if (!cur || !prev || !valid_L(cur->L) || !valid_L(prev->L)) {
        memset(out, 0, 160 * sizeof(float));
        return;
    }

    if (have_rm0) {
        adaptive_smoothing_core(cur, prev, rm0);
    } else {
        mbxo_adaptive_smoothing(cur, prev);
    }

    /* mute (:895-899): max repeats, or (non-AMBE state only) error rate over threshold */
    int rate_mutes = (fabsf(cur->mutingThreshold - MBE_MUTING_THRESHOLD_AMBE) > 1e-6f);
    if (cur->repeatCount >= MBE_MAX_FRAME_REPEATS || (rate_mutes && cur->errorRate > cur->mutingThreshold)) {
        mbxo_comfort_noisef(out, rng);
        return;
    }

    float noise[256];
    mbxo_noise_next(noise, &cur->noiseSeed, cur->noiseOverlap, rng);
    memset(out, 0, 160 * sizeof(float));

    /* reconcile model lengths (:912-929) */
    int maxl;
    if (cur->L > prev->L) {
        maxl = cur->L;
        for (int l = prev->L + 1; l <= maxl; ++l) {
            prev->Ml[l] = 0.0f;
            prev->Vl[l] = 1;
        }
    } else {
        maxl = prev->L;
        for (int l = cur->L + 1; l <= maxl; ++l) {
            cur->Ml[l] = 0.0f;
            cur->Vl[l] = 1;
        }
    }

    /* phases (:901-951); the unvoiced count includes slot 0 */
    int numUv = 0;
    for (int l = 0; l <= cur->L; ++l) {
        if (cur->Vl[l] == 0) {
            numUv++;
        }
    }
    const float cw0 = cur->w0;
    const float pw0 = prev->w0;
    for (int l = 1; l <= 56; ++l) {
        float wrapped = fmodf(prev->PSIl[l], TWO_PI);
        if (wrapped < 0.0f) {
            wrapped += TWO_PI;
        }
        prev->PSIl[l] = wrapped;
        cur->PSIl[l] = wrapped + ((pw0 + cw0) * ((float)(l * N) / 2.0f));
        if (l <= (cur->L / 4)) {
            cur->PHIl[l] = cur->PSIl[l];
        } else {
            float pl = ((2.0f * (float)M_PI / 53125.0f) * noise[l]) - (float)M_PI;
            cur->PHIl[l] = cur->PSIl[l] + (((float)numUv * pl) / (float)cur->L);
        }
    }

    /* voiced bank (:953-1040) */
    const float* Ws = T->ws;
    for (int l = 1; l <= maxl; ++l) {
        const float cw0l = cw0 * (float)l;
        const float pw0l = pw0 * (float)l;
        const int cv = (cur->Vl[l] == 1);
        const int pv = (prev->Vl[l] == 1);
        if (!cv && !pv) {
            continue;
        }
        if ((l < 8) && cv && pv && (fabsf(cw0 - pw0) < (0.1f * cw0))) {
            /* low harmonics with a stable pitch: interpolated amplitude, quadratic phase */
            float dphi = cur->PHIl[l] - prev->PHIl[l] - (((pw0 + cw0) * (float)(l * N)) / 2.0f);
            float dw = (1.0f / (float)N)
                       * (dphi - (2.0f * (float)M_PI * floorf((dphi + (float)M_PI) / (2.0f * (float)M_PI))));
            for (int n = 0; n < N; ++n) {
                float theta =
                    prev->PHIl[l] + ((pw0l + dw) * (float)n) + (((cw0 - pw0) * (float)(l * n * n)) / (float)(2 * N));
                float a = prev->Ml[l] + (((float)n / (float)N) * (cur->Ml[l] - prev->Ml[l]));
                out[n] += 2.0f * a * cosf(theta);
            }
            continue;
        }
        /* windowed oscillators advanced by plane rotation */
        float gp = 0, sdp = 0, cdp = 0, sp = 0, cp = 0;
        float gc = 0, sdc = 0, cdc = 0, sc = 0, cc = 0;
        if (pv) {
            gp = 2.0f * prev->Ml[l];
            sincosf(pw0l, &sdp, &cdp);
            sincosf(prev->PHIl[l], &sp, &cp);
        }
        if (cv) {
            gc = 2.0f * cur->Ml[l];
            sincosf(cw0l, &sdc, &cdc);
            sincosf(cur->PHIl[l] - (cw0l * (float)N), &sc, &cc);
        }
        for (int n = 0; n < N; ++n) {
            if (pv) {
                out[n] += gp * Ws[n + N] * cp;
                float c2 = (cp * cdp) - (sp * sdp);
                float s2 = (sp * cdp) + (cp * sdp);
                cp = c2;
                sp = s2;
            }
            if (cv) {
                out[n] += gc * Ws[n] * cc;
                float c2 = (cc * cdc) - (sc * sdc);
                float s2 = (sc * cdc) + (cc * sdc);
                cc = c2;
                sc = s2;
            }
        }
    }

    synth_unvoiced(out, cur, prev, noise);

    /* soft clip (:669-689) */
    const float clip = (32767.0f * 0.95f) / 7.0f;
    for (int n = 0; n < N; ++n) {
        float a = fabsf(out[n]);
        if (a > g_preclip_peak) { /* diagnostic only: how far beyond the output range the frame's sum went (mbxo_set_preclip_peaks) */
            g_preclip_peak = a;
        }
        if (out[n] > clip) {
            out[n] = clip;
        } else if (out[n] < -clip) {
            out[n] = -clip;
        }
    }
}

void
mbxo_synthesize_speechf(float* out, mbe_parms* cur, mbe_parms* prev, mbx_stream_rng* rng) {
    synth_core(out, cur, prev, 0, 0.0f, rng);
}

/* =====================================================================================
 * float -> int16 (a21)  ref: src/core/mbelib.c:1148-1177, 1312-1320
 * ===================================================================================== */
void
mbxo_floattoshort(const float* in, int16_t* out) {
    if (!in || !out) {
        return;
    }
    const float top = 32767.0f * 0.95f;
    for (int i = 0; i < 160; ++i) {
        uint32_t bits;
        memcpy(&bits, &in[i], 4);
        uint32_t mag = bits & 0x7FFFFFFFu;
        float v;
        if (mag > 0x7F800000u) {
            v = 0.0f;
        } else if (mag == 0x7F800000u) {
            v = (bits & 0x80000000u) ? -top : top;
        } else {
            v = 7.0f * in[i];
            if (v > top) {
                v = top;
            } else if (v < -top) {
                v = -top;
            }
        }
        out[i] = (int16_t)v;
    }
}

/* =====================================================================================
 * AMBE tone frames  ref: src/core/mbelib.c:691-804
 * ===================================================================================== */
static uint32_t
tone_step(double hz) {
    double step = (hz / 8000.0) * 4294967296.0;
    return step <= 0.0 ? 0u : (uint32_t)(step + 0.5);
}

static float
tone_sample(uint32_t phase) {
    float angle = (float)(((double)phase * ((2.0 * M_PI) / 4294967296.0)) - (M_PI / 2.0));
    return sinf(angle);
}

/* the reference's NOTONES build option (-DDISABLE_AMBE_TONES, ref CMakeLists.txt:330-337) as a switch of the restatement:
 * off = tone frames are 160 zeros and the tone phases stay (ref src/core/mbelib.c:747-751, 815-819) */
static int g_tones_on = 1;
void
mbxo_set_tones(int on) {
    g_tones_on = on ? 1 : 0;
}

void
mbxo_tonef(float* out, const char* d, mbe_parms* cur) {
    if (!out) {
        return;
    }
    memset(out, 0, 160 * sizeof(float));
    if (!g_tones_on || !cur || validate_bits(d, 49u) < 0) {
        return;
    }
    int u0 = 0, u1 = 0, u3 = 0;
    for (int i = 0; i < 12; ++i) {
        u0 = (u0 << 1) | d[i];
    }
    for (int i = 12; i < 24; ++i) {
        u1 = (u1 << 1) | d[i];
    }
    for (int i = 35; i < 49; ++i) {
        u3 = (u3 << 1) | d[i];
    }
    int AD = ((u0 & 0x3f) << 1) + ((u3 >> 4) & 0x1);
    int ID1 = ((u1 & 0xfff) >> 4);
    float f1, f2;
    if (!tone_freqs(ID1, &f1, &f2) || f1 <= 0.0f) {
        return;
    }
    const int dual = (f2 > 0.0f) && (fabsf(f2 - f1) > 1e-6f);
    const float clip = (32767.0f * 0.95f) / 7.0f;
    const float gain = (((AD < 0) ? 0.0f : (float)AD) / 127.0f) * clip;
    const uint32_t s1 = tone_step((double)f1);
    const uint32_t s2 = dual ? tone_step((double)f2) : 0u;
    uint32_t p1 = (uint32_t)cur->swn;
    uint32_t p2 = cur->tonePhase;
    for (int n = 0; n < 160; ++n) {
        p1 += s1;
        float a = tone_sample(p1);
        if (dual) {
            p2 += s2;
            float b = tone_sample(p2);
            out[n] = (0.5f * gain * a) + (0.5f * gain * b);
        } else {
            out[n] = gain * a;
        }
    }
    cur->swn = (int)p1;
    cur->tonePhase = p2;
}

/* ref: src/core/mbelib.c:813-856 (mbe_synthesizeTonefdstar) + :708-736 (mbe_renderTonef): a single tone of
 * 156.25 Hz (index 5), 187.5 Hz (6) or 31.25 Hz x index (7..122) at the fixed amplitude 103 */
void
mbxo_tone_dstarf(float* out, mbe_parms* cur, int id1) {
    if (!out) {
        return;
    }
    memset(out, 0, 160 * sizeof(float));
    if (!g_tones_on || !cur) {
        return;
    }
    float f1 = 0.0f;
    if (id1 == 5) {
        f1 = 156.25f;
    } else if (id1 == 6) {
        f1 = 187.5f;
    } else if (id1 >= 7 && id1 <= 122) {
        f1 = 31.25f * (float)id1;
    }
    if (f1 <= 0.0f) {
        return;
    }
    const float clip = (32767.0f * 0.95f) / 7.0f;
    const float gain = ((float)103 / 127.0f) * clip;
    const uint32_t s1 = tone_step((double)f1);
    uint32_t p1 = (uint32_t)cur->swn;
    for (int n = 0; n < 160; ++n) {
        p1 += s1;
        out[n] = gain * tone_sample(p1);
    }
    cur->swn = (int)p1;
}

/* =====================================================================================
 * IMBE stream stage (a10)  ref: src/imbe/imbe7200x4400.c:56-81, 780-909
 * ===================================================================================== */
static void
imbe_headroom_reset(mbe_parms* mp) {
    mp->swn = 0;
    mp->tonePhase = 0;
    mp->w0 = (float)((4.0 * M_PI) / (134.0 + 39.5));
    mp->L = (int)(0.9254 * (int)((M_PI / mp->w0) + 0.25));
    mp->K = 12;
    mp->gamma = 0.0f;
    for (int l = 0; l <= 56; ++l) {
        mp->Vl[l] = 0;
        mp->Ml[l] = 1.0f;
        mp->log2Ml[l] = 0.0f;
    }
    mp->repeatCount = 0;
    mp->localEnergy = 75000.0f;
    mp->amplitudeThreshold = 20480;
    mp->mutingThreshold = MBE_MUTING_THRESHOLD_IMBE;
}

int
mbxo_process_imbe4400_dataf(float* out, mbe_process_result* result, const char imbe_d[88], mbe_parms* cur,
                            mbe_parms* prev, mbe_parms* prev_enh, mbx_stream_rng* rng) {
    mbe_process_result local;
    if (!result) {
        memset(&local, 0, sizeof(local));
        result = &local;
    }
    if (!out || !cur || !prev || !prev_enh) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int total;
    int rc = resolve_total_errors(result, &total);
    if (rc < 0) {
        return rc;
    }
    rc = validate_bits(imbe_d, 88u);
    if (rc < 0) {
        return rc;
    }
    const int c0_valid = (result->flags & MBE_PROCESS_FLAG_C0_VALID) != 0u;
    const int c4_valid = (result->flags & MBE_PROCESS_FLAG_C4_VALID) != 0u;
    const int c0 = c0_valid ? result->c0_errors : 0;
    cur->errorCount4 = c4_valid ? result->c4_errors : 0;
    result_prepare_synthesis(result, total);

    cur->mutingThreshold = MBE_MUTING_THRESHOLD_IMBE;
    cur->errorCountTotal = total;
    cur->errorRate = (0.95f * prev->errorRate) + (0.000365f * (float)total);

    int bad = mbxo_decode_imbe4400_parms(imbe_d, cur, prev);
    if (bad < 0) {
        return bad;
    }
    const float repeat_threshold = 10.0f + (40.0f * cur->errorRate);
    int repeat;
    if (bad == 1) {
        repeat = 1;
    } else if (c0_valid) {
        repeat = (c0 >= 2) && ((float)total >= repeat_threshold);
    } else {
        repeat = (total > 5);
    }
    if (!repeat) {
        cur->repeatCount = 0;
    } else {
        if (prev->repeatCount > (MBE_MAX_FRAME_REPEATS - 1)) {
            imbe_headroom_reset(cur);
        } else {
            *cur = *prev;
            cur->repeatCount++;
        }
        result->flags |= MBE_PROCESS_FLAG_REPEAT;
    }

    /* synthesize (:842-856) */
    int muted = (cur->repeatCount >= MBE_MAX_FRAME_REPEATS) || (cur->errorRate > cur->mutingThreshold);
    *prev = *cur;
    float rm0 = mbxo_spectral_amp_enhance(cur);
    synth_core(out, cur, prev_enh, 1, rm0, rng);
    if (muted) {
        result->flags |= MBE_PROCESS_FLAG_MUTE;
    }
    *prev_enh = *cur;
    return result->total_errors;
}

/* =====================================================================================
 * AMBE+2 stream stage (a11)  ref: src/ambe/ambe3600x2450.c:716-898
 * ===================================================================================== */
int
mbxo_process_ambe2450_dataf(float* out, mbe_process_result* result, const char d[49], mbe_parms* cur, mbe_parms* prev,
                            mbe_parms* prev_enh, mbx_stream_rng* rng) {
    mbe_process_result local;
    if (!result) {
        memset(&local, 0, sizeof(local));
        result = &local;
    }
    if (!out || !cur || !prev || !prev_enh) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int total;
    int rc = resolve_total_errors(result, &total);
    if (rc < 0) {
        return rc;
    }
    rc = validate_bits(d, 49u);
    if (rc < 0) {
        return rc;
    }
    const int c0_valid = (result->flags & MBE_PROCESS_FLAG_C0_VALID) != 0u;
    const int c0 = c0_valid ? result->c0_errors : 0;
    result_prepare_synthesis(result, total);

    if (fabsf(prev->mutingThreshold - MBE_MUTING_THRESHOLD_AMBE) > 1e-6f) {
        init_ambe_parms(cur, prev, prev_enh); /* state came from the generic initialiser */
    }
    cur->mutingThreshold = MBE_MUTING_THRESHOLD_AMBE;
    cur->errorCountTotal = total;
    cur->errorCount4 = 0;
    cur->errorRate = (0.95f * prev->errorRate) + (0.001064f * (float)cur->errorCountTotal);

    int bad = mbxo_decode_ambe2450_parms(d, cur, prev, total);
    if (bad < 0) {
        return bad;
    }

    /* decode state (:760-783) */
    if (bad == 2) {
        result->flags |= MBE_PROCESS_FLAG_ERASURE;
        cur->repeatCount = 0;
        set_ambe_erasure_parms(cur, prev);
    } else if (bad == 7) {
        result->flags |= MBE_PROCESS_FLAG_TONE;
        cur->repeatCount = 0;
    } else {
        int repeat = c0_valid ? ((c0 >= 4) || ((c0 >= 2) && (total >= 6))) : (total > 3);
        if (repeat) {
            *cur = *prev;
            cur->repeatCount++;
            result->flags |= MBE_PROCESS_FLAG_REPEAT;
        } else {
            cur->repeatCount = 0;
        }
    }

    /* synthesize (:785-849) */
    if (bad == 0) {
        if (cur->repeatCount < MBE_MAX_FRAME_REPEATS) {
            *prev = *cur;
            float rm0 = mbxo_spectral_amp_enhance(cur);
            synth_core(out, cur, prev_enh, 1, rm0, rng);
            *prev_enh = *cur;
        } else {
            result->flags |= MBE_PROCESS_FLAG_MUTE;
            mbxo_comfort_noisef(out, rng);
            init_ambe_parms(cur, prev, prev_enh);
        }
    } else if (bad == 7) {
        int id1 = 0;
        for (int i = 12; i < 20; ++i) {
            id1 = (id1 << 1) | d[i];
        }
        float f1, f2;
        if (tone_freqs(id1, &f1, &f2)) {
            mbxo_tonef(out, d, cur);
        } else if (!(prev->repeatCount >= MBE_MAX_FRAME_REPEATS)) {
            mbe_parms tmp = *prev_enh;
            synth_core(out, &tmp, prev_enh, 0, 0.0f, rng);
            *prev_enh = tmp;
        } else {
            mbxo_comfort_noisef(out, rng);
            init_ambe_parms(cur, prev, prev_enh);
        }
    } else { /* erasure */
        mbxo_comfort_noisef(out, rng);
        *prev = *cur;
        *prev_enh = *cur;
    }
    return result->total_errors;
}

/* =====================================================================================
 * AMBE 3600x2400 stream stage  ref: src/ambe/ambe3600x2400.c:629-763
 * ===================================================================================== */
int
mbxo_process_ambe2400_dataf(float* out, mbe_process_result* result, const char d[49], mbe_parms* cur, mbe_parms* prev,
                            mbe_parms* prev_enh, mbx_stream_rng* rng) {
    mbe_process_result local;
    if (!result) {
        memset(&local, 0, sizeof(local));
        result = &local;
    }
    if (!out || !cur || !prev || !prev_enh) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int total;
    int rc = resolve_total_errors(result, &total);
    if (rc < 0) {
        return rc;
    }
    rc = validate_bits(d, 49u);
    if (rc < 0) {
        return rc;
    }
    const int c0 = ((result->flags & MBE_PROCESS_FLAG_C0_VALID) != 0u) ? result->c0_errors : 0;
    result_prepare_synthesis(result, total);

    if (fabsf(prev->mutingThreshold - MBE_MUTING_THRESHOLD_AMBE) > 1e-6f) {
        init_ambe_parms(cur, prev, prev_enh);
    }
    cur->mutingThreshold = MBE_MUTING_THRESHOLD_AMBE;
    cur->errorCountTotal = total;
    cur->errorCount4 = 0;
    cur->errorRate = (0.95f * prev->errorRate) + (0.001064f * (float)cur->errorCountTotal);

    int bad = mbxo_decode_ambe2400_parms(d, cur, prev);
    if (bad < 0) {
        return bad;
    }
    const int valid_tone = (bad >= 7) && (bad <= 122) && (c0 < 2) && (total < 3);

    /* decode state (:661-686) */
    if (bad == 3) {
        result->flags |= MBE_PROCESS_FLAG_TONE;
        cur->repeatCount = 0;
    } else if (valid_tone) {
        /* nothing */
    } else if (total > 3) {
        *cur = *prev;
        cur->repeatCount++;
        result->flags |= MBE_PROCESS_FLAG_REPEAT;
    } else {
        cur->repeatCount = 0;
    }

    /* synthesize (:688-731) */
    if (valid_tone) {
        mbxo_tone_dstarf(out, cur, bad);
        *prev = *cur;
    } else if (bad == 0) {
        if (cur->repeatCount < MBE_MAX_FRAME_REPEATS) {
            *prev = *cur;
            float rm0 = mbxo_spectral_amp_enhance(cur);
            synth_core(out, cur, prev_enh, 1, rm0, rng);
            *prev_enh = *cur;
        } else {
            result->flags |= MBE_PROCESS_FLAG_MUTE;
            mbxo_comfort_noisef(out, rng);
            init_ambe_parms(cur, prev, prev_enh);
        }
    } else {
        mbxo_comfort_noisef(out, rng);
        init_ambe_parms(cur, prev, prev_enh);
    }
    return result->total_errors;
}

/* frame-level entries  ref: src/imbe/imbe7200x4400.c:935-948, src/ambe/ambe3600x2450.c:924-937 */
int
mbxo_process_imbe7200x4400_framef(float* out, mbe_process_result* result, const char fr[8][23], char imbe_d[88],
                                  mbe_parms* cur, mbe_parms* prev, mbe_parms* prev_enh, mbx_stream_rng* rng) {
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    int rc = mbxo_decode_imbe7200x4400_frame(fr, imbe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbxo_process_imbe4400_dataf(out, result, imbe_d, cur, prev, prev_enh, rng);
}

int
mbxo_process_ambe3600x2450_framef(float* out, mbe_process_result* result, const char fr[4][24], char ambe_d[49],
                                  mbe_parms* cur, mbe_parms* prev, mbe_parms* prev_enh, mbx_stream_rng* rng) {
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    int rc = mbxo_decode_ambe3600x2450_frame(fr, ambe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbxo_process_ambe2450_dataf(out, result, ambe_d, cur, prev, prev_enh, rng);
}

/* ref: src/ambe/ambe3600x2400.c:563-596, 788-802: the AMBE FEC front end is shared with 3600x2450 */
int
mbxo_process_ambe3600x2400_framef(float* out, mbe_process_result* result, const char fr[4][24], char ambe_d[49],
                                  mbe_parms* cur, mbe_parms* prev, mbe_parms* prev_enh, mbx_stream_rng* rng) {
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    int rc = mbxo_decode_ambe3600x2450_frame(fr, ambe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbxo_process_ambe2400_dataf(out, result, ambe_d, cur, prev, prev_enh, rng);
}

/* =====================================================================================
 * Batch drivers -- same contract as the HIP launcher (include/mbx.h): S streams x T frames,
 * stream-major; state[3*s + {0,1,2}] = {cur, prev, prev_enhanced}.
 * ===================================================================================== */
int
mbxo_fec_batch(int codec, size_t n, const uint8_t* frames, mbx_param_record* records) {
    for (size_t i = 0; i < n; ++i) {
        if (codec == MBX_CODEC_IMBE7200X4400) {
            mbxo_fec_imbe7200x4400(frames + i * MBX_IMBE_FRAME_BYTES, &records[i]);
        } else if (codec == MBX_CODEC_IMBE7100X4400) {
            mbxo_fec_imbe7100x4400(frames + i * MBX_IMBE7100_FRAME_BYTES, &records[i]);
        } else { /* both AMBE codecs share the FEC front end */
            mbxo_fec_ambe3600x2450(frames + i * MBX_AMBE_FRAME_BYTES, &records[i]);
        }
    }
    return 0;
}

void
mbxo_set_preclip_peaks(float* out) {
    g_preclip_peaks_out = out;
}

/* `soft` != 0: frames are mbe_soft_bit arrays (184 | 96 per frame) through the soft-decision FEC */
static int
process_batch_impl(int codec, int S, int Tn, const void* frames, int soft, mbe_parms* state, mbx_stream_rng* rng,
                   int16_t* pcm16, float* pcmf, mbe_process_result* results, mbx_param_record* records) {
    if (!T || !frames || !state || !rng || S < 0 || Tn < 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    const int imbe = (codec == MBX_CODEC_IMBE7200X4400 || codec == MBX_CODEC_IMBE7100X4400);
    const size_t cells = (codec == MBX_CODEC_IMBE7200X4400)   ? MBX_IMBE_SOFT_BITS
                         : (codec == MBX_CODEC_IMBE7100X4400) ? MBX_IMBE7100_SOFT_BITS
                                                              : MBX_AMBE_SOFT_BITS;
    const size_t fb = soft ? cells * sizeof(mbe_soft_bit) : (size_t)(imbe ? MBX_IMBE_FRAME_BYTES : MBX_AMBE_FRAME_BYTES);
    for (int s = 0; s < S; ++s) {
        mbe_parms* cur = &state[3 * (size_t)s];
        mbe_parms* prev = cur + 1;
        mbe_parms* enh = cur + 2;
        for (int t = 0; t < Tn; ++t) {
            size_t f = (size_t)s * (size_t)Tn + (size_t)t;
            const uint8_t* fr = (const uint8_t*)frames + f * fb;
            mbx_param_record rec;
            mbe_process_result res;
            float pcm[160];
            char bits[88];
            g_preclip_peak = 0.0f;
            if (imbe) {
                if (soft && codec == MBX_CODEC_IMBE7100X4400) {
                    mbxo_fec_imbe7100x4400_soft((const mbe_soft_bit(*)[24])fr, &rec);
                } else if (soft) {
                    mbxo_fec_imbe7200x4400_soft((const mbe_soft_bit(*)[23])fr, &rec);
                } else if (codec == MBX_CODEC_IMBE7100X4400) {
                    mbxo_fec_imbe7100x4400(fr, &rec);
                } else {
                    mbxo_fec_imbe7200x4400(fr, &rec);
                }
                mbxo_record_to_bits(&rec, 88, bits);
                mbxo_record_to_result(&rec, &res);
                mbxo_process_imbe4400_dataf(pcm, &res, bits, cur, prev, enh, &rng[s]);
            } else {
                if (soft) {
                    mbxo_fec_ambe3600x2450_soft((const mbe_soft_bit(*)[24])fr, &rec);
                } else {
                    mbxo_fec_ambe3600x2450(fr, &rec);
                }
                mbxo_record_to_bits(&rec, 49, bits);
                mbxo_record_to_result(&rec, &res);
                if (codec == MBX_CODEC_AMBE3600X2400) {
                    mbxo_process_ambe2400_dataf(pcm, &res, bits, cur, prev, enh, &rng[s]);
                } else {
                    mbxo_process_ambe2450_dataf(pcm, &res, bits, cur, prev, enh, &rng[s]);
                }
            }
            if (g_preclip_peaks_out) {
                g_preclip_peaks_out[f] = g_preclip_peak;
            }
            if (records) {
                records[f] = rec;
            }
            if (results) {
                results[f] = res;
            }
            if (pcmf) {
                memcpy(pcmf + f * 160, pcm, sizeof(pcm));
            }
            if (pcm16) {
                mbxo_floattoshort(pcm, pcm16 + f * 160);
            }
        }
    }
    return 0;
}

int
mbxo_process_batch(int codec, int S, int Tn, const uint8_t* frames, mbe_parms* state, mbx_stream_rng* rng,
                   int16_t* pcm16, float* pcmf, mbe_process_result* results, mbx_param_record* records) {
    return process_batch_impl(codec, S, Tn, frames, 0, state, rng, pcm16, pcmf, results, records);
}

int
mbxo_process_batch_soft(int codec, int S, int Tn, const mbe_soft_bit* soft, mbe_parms* state, mbx_stream_rng* rng,
                        int16_t* pcm16, float* pcmf, mbe_process_result* results, mbx_param_record* records) {
    return process_batch_impl(codec, S, Tn, soft, 1, state, rng, pcm16, pcmf, results, records);
}

void
mbxo_floattoshort_batch(const float* in, int16_t* out, size_t nframes) {
    for (size_t i = 0; i < nframes; ++i) {
        mbxo_floattoshort(in + i * 160, out + i * 160);
    }
}

void
mbxo_synthesize_speech_batch(int S, mbe_parms* cur, mbe_parms* prev, mbx_stream_rng* rng, float* pcmf) {
    for (int s = 0; s < S; ++s) {
        mbxo_synthesize_speechf(pcmf + (size_t)s * 160, &cur[s], &prev[s], &rng[s]);
    }
}
