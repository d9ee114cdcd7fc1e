/*
 * dump_tables.c -- authoring-container tool (TEST/DATA INFRASTRUCTURE, not product code).
 *
 * Builds the constant-table blob described in include/mbx_tables.h.  The reference's
 * tables are read where they lie under /root/reference (this file #includes the
 * reference's private const headers at build time and links oracle/_ref/libmbe_ref.so);
 * nothing but the resulting binary DATA file enters the repository.
 *
 *   usage: oracle/_ref/dump_tables mbelib-neo_amd/data/mbx_tables.bin
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ambe3600x2400_const.h"  /* reference: src/internal */
#include "ambe3600x2450_const.h"  /* reference: src/internal */
#include "ecc_const.h"            /* reference: src/internal (extern; defined in libmbe_ref) */
#include "imbe7200x4400_const.h"  /* reference: src/internal */
#include "mbe_unvoiced_fft.h"     /* reference: mbe_synthesisWindow() */
#include "mbelib_const.h"         /* reference: Ws[321] */

#include "mbx_tables.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static uint32_t
fnv1a(const uint8_t* p, size_t n) {
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < n; ++i) {
        h = (h ^ p[i]) * 16777619u;
    }
    return h;
}

#define NARROW(dst, src, count, maxv)                                                                                  \
    do {                                                                                                               \
        for (size_t i_ = 0; i_ < (size_t)(count); ++i_) {                                                              \
            long v_ = (long)(src)[i_];                                                                                 \
            if (v_ < 0 || v_ > (maxv)) {                                                                               \
                fprintf(stderr, "value out of range in %s[%zu]: %ld\n", #src, i_, v_);                                 \
                exit(1);                                                                                               \
            }                                                                                                          \
            (dst)[i_] = (__typeof__((dst)[0]))v_;                                                                      \
        }                                                                                                              \
    } while (0)

static int
hamming_syndrome(unsigned block, const int gen[4]) {
    int s = 0;
    for (int i = 0; i < 4; ++i) {
        s |= (__builtin_popcount(block & (unsigned)gen[i]) & 1) << i;
    }
    return s;
}

int
main(int argc, char** argv) {
    if (argc != 2) {
        fprintf(stderr, "usage: %s out.bin\n", argv[0]);
        return 2;
    }
    mbx_tables* t = calloc(1, sizeof(*t));
    t->magic = MBX_TABLES_MAGIC;
    t->version = MBX_TABLES_VERSION;
    t->total_bytes = (uint32_t)sizeof(*t);

    /* FEC */
    NARROW(t->golay_matrix, golayMatrix, 2048, 4095);
    NARROW(t->golay_gen, golayGenerator, 12, 2047);
    NARROW(t->hamming_gen, hammingGenerator, 4, 32767);
    for (int b = 0; b < 15; ++b) { /* syndrome of each single-bit error -> the bit to flip */
        t->hamming_fix[hamming_syndrome(1u << b, hammingGenerator)] = (uint16_t)(1u << b);
    }

    NARROW(t->hamming7100_gen, imbe7100x4400hammingGenerator, 4, 32767);
    for (int b = 0; b < 15; ++b) {
        t->hamming7100_fix[hamming_syndrome(1u << b, imbe7100x4400hammingGenerator)] = (uint16_t)(1u << b);
    }

    /* IMBE fundamental: same expressions as src/imbe/imbe7200x4400.c:132-148 */
    for (int b0 = 0; b0 < 208; ++b0) {
        float w0 = ((float)(4 * M_PI) / (float)((float)b0 + 39.5));
        int L = (int)(0.9254 * (int)((M_PI / w0) + 0.25));
        int K = (L < 37) ? (int)((float)(L + 2) / (float)3) : 12;
        t->imbe_w0[b0] = w0;
        t->imbe_L[b0] = (uint8_t)((L > 56 || L < 9) ? 0 : L);
        t->imbe_K[b0] = (uint8_t)K;
    }
    NARROW(&t->imbe_bo[0][0][0], &bo[0][0][0], 48 * 79 * 2, 255);
    NARROW(&t->imbe_hoba[0][0], &hoba[0][0], 48 * 50, 255);
    NARROW(&t->imbe_ji[0][0], &ImbeJi[0][0], 48 * 6, 255);
    memcpy(t->imbe_ba, ba, sizeof(t->imbe_ba));
    memcpy(t->imbe_B2, B2, sizeof(t->imbe_B2));
    memcpy(t->imbe_quantstep, quantstep, sizeof(t->imbe_quantstep));
    memcpy(t->imbe_standdev, standdev, sizeof(t->imbe_standdev));
    for (int m = 1; m <= 6; m++) {
        for (int i = 1; i <= 6; i++) {
            t->imbe_ri_cos[m][i] = cosf((M_PI * (float)(m - 1) * ((float)i - 0.5f)) / 6.0f);
        }
    }
    for (int ji = 1; ji <= 10; ji++) {
        for (int j = 1; j <= ji; j++) {
            for (int k = 1; k <= ji; k++) {
                t->imbe_idct_cos[ji][j][k] = cosf((M_PI * (float)(k - 1) * ((float)j - 0.5f)) / (float)ji);
            }
        }
    }

    /* AMBE+2 */
    memcpy(t->ambe_w0, AmbeW0table, sizeof(t->ambe_w0));
    NARROW(t->ambe_L, AmbeLtable, 120, 56);
    NARROW(&t->ambe_vuv[0][0], &AmbeVuv[0][0], 32 * 8, 1);
    NARROW(&t->ambe_lmprbl[0][0], &AmbeLmprbl[0][0], 57 * 4, 255);
    memcpy(t->ambe_dg, AmbeDg, sizeof(t->ambe_dg));
    memcpy(t->ambe_prba24, AmbePRBA24, sizeof(t->ambe_prba24));
    memcpy(t->ambe_prba58, AmbePRBA58, sizeof(t->ambe_prba58));
    memcpy(t->ambe_hoc_b5, AmbeHOCb5, sizeof(t->ambe_hoc_b5));
    memcpy(t->ambe_hoc_b6, AmbeHOCb6, sizeof(t->ambe_hoc_b6));
    memcpy(t->ambe_hoc_b7, AmbeHOCb7, sizeof(t->ambe_hoc_b7));
    memcpy(t->ambe_hoc_b8, AmbeHOCb8, sizeof(t->ambe_hoc_b8));
    /* AMBE 3600x2400 (D-STAR) */
    memcpy(t->ambep_dg, AmbePlusDg, sizeof(t->ambep_dg));
    memcpy(t->ambep_prba24, AmbePlusPRBA24, sizeof(t->ambep_prba24));
    memcpy(t->ambep_prba58, AmbePlusPRBA58, sizeof(t->ambep_prba58));
    memcpy(t->ambep_hoc_b5, AmbePlusHOCb5, sizeof(t->ambep_hoc_b5));
    memcpy(t->ambep_hoc_b6, AmbePlusHOCb6, sizeof(t->ambep_hoc_b6));
    memcpy(t->ambep_hoc_b7, AmbePlusHOCb7, sizeof(t->ambep_hoc_b7));
    memcpy(t->ambep_hoc_b8, AmbePlusHOCb8, sizeof(t->ambep_hoc_b8));
    NARROW(t->ambep_L, AmbePlusLtable, 126, 56);
    NARROW(&t->ambep_vuv[0][0], &AmbePlusVuv[0][0], 16 * 8, 1);
    NARROW(&t->ambep_lmprbl[0][0], &AmbePlusLmprbl[0][0], 57 * 4, 255);

    for (int m = 1; m <= 8; m++) {
        for (int i = 1; i <= 8; i++) {
            t->ambe_ri_cos[m][i] = cosf((M_PI * (float)(m - 1) * ((float)i - 0.5f)) / 8.0f);
        }
    }
    for (int ji = 1; ji <= 17; ji++) {
        for (int j = 1; j <= ji; j++) {
            for (int k = 1; k <= ji; k++) {
                t->ambe_idct_cos[ji][j][k] = cosf((M_PI * (float)(k - 1) * ((float)j - 0.5f)) / (float)ji);
            }
        }
    }

    /* synthesis windows */
    memcpy(t->ws, Ws, sizeof(t->ws));
    for (int i = 0; i < 256; ++i) {
        t->uv_window[i] = mbe_synthesisWindow(i - 128);
    }
    for (int n = 0; n < 160; ++n) {
        float wp = mbe_synthesisWindow(n);
        float wc = mbe_synthesisWindow(n - 160);
        t->wola_w_prev[n] = wp;
        t->wola_w_curr[n] = wc;
        t->wola_denom[n] = (wp * wp) + (wc * wc);
    }

    t->checksum = fnv1a((const uint8_t*)&t->checksum + 4, sizeof(*t) - 16);

    FILE* f = fopen(argv[1], "wb");
    if (!f || fwrite(t, sizeof(*t), 1, f) != 1) {
        perror("write");
        return 1;
    }
    fclose(f);
    printf("wrote %s: %zu bytes, checksum 0x%08X\n", argv[1], sizeof(*t), t->checksum);
    return 0;
}
