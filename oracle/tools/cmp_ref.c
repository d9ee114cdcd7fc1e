/*
 * cmp_ref.c -- authoring-container check (TEST INFRASTRUCTURE): drives the real reference
 * (oracle/_ref/libmbe_ref.so) and the restatement (oracle/liboracle.so) with the same
 * random-bit streams and reports how far apart they are.
 *
 *   usage: oracle/_ref/cmp_ref tables.bin [codec 0|1] [S] [T] [ber_percent (-1 = raw random bits)]
 *
 * Integer outputs (parameter bits, error counts, flags, return codes, every integer field of
 * the three state structs) must match exactly; float outputs are compared bitwise and by
 * relative RMS.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mbx_oracle.h"

/* Entry points of the real reference (include/mbelib-neo/mbelib.h:194,429,505,596,615,675).
 * mbx_types.h declares ABI-identical structs, so the prototypes are restated with them
 * instead of including both headers (which would define the same tags twice). */
extern void mbe_initMbeParms(mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced);
extern void mbe_setThreadRngSeed(uint32_t seed);
extern void mbe_floattoshort(const float* float_buf, short* aout_buf);
extern int mbe_processImbe7200x4400Framef(float* aout_buf, mbe_process_result* result, const char imbe_fr[8][23],
                                          char imbe_d[88], mbe_parms* cur_mp, mbe_parms* prev_mp,
                                          mbe_parms* prev_mp_enhanced);
extern int mbe_processAmbe3600x2450Framef(float* aout_buf, mbe_process_result* result, const char ambe_fr[4][24],
                                          char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp,
                                          mbe_parms* prev_mp_enhanced);

static uint64_t sm_state;
static uint64_t
splitmix64(void) {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static int
float_same(float a, float b) {
    uint32_t x, y;
    memcpy(&x, &a, 4);
    memcpy(&y, &b, 4);
    return x == y || (a != a && b != b);
}

typedef struct {
    long n, bitdiff;
    double num, den, worst;
} fstat;

static void
facc(fstat* s, const float* a, const float* b, int n) {
    double num = 0, den = 0;
    for (int i = 0; i < n; ++i) {
        s->n++;
        if (!float_same(a[i], b[i])) {
            s->bitdiff++;
        }
        if (a[i] == a[i] && b[i] == b[i]) {
            double d = (double)a[i] - (double)b[i];
            num += d * d;
            den += (double)a[i] * (double)a[i];
        }
    }
    s->num += num;
    s->den += den;
    if (den > 0 && sqrt(num / den) > s->worst) {
        s->worst = sqrt(num / den);
    }
}

static void
frep(const char* name, const fstat* s) {
    printf("  %-14s n=%-9ld bit-different=%-8ld rel-rms=%.3e worst-frame=%.3e\n", name, s->n, s->bitdiff,
           s->den > 0 ? sqrt(s->num / s->den) : 0.0, s->worst);
}

static long
int_fields_differ(const mbe_parms* a, const mbe_parms* b) {
    long d = 0;
    d += a->L != b->L;
    d += a->K != b->K;
    d += memcmp(a->Vl, b->Vl, sizeof(a->Vl)) != 0;
    d += a->tonePhase != b->tonePhase;
    d += a->swn != b->swn;
    d += a->amplitudeThreshold != b->amplitudeThreshold;
    d += a->errorCountTotal != b->errorCountTotal;
    d += a->errorCount4 != b->errorCount4;
    d += a->repeatCount != b->repeatCount;
    d += !float_same(a->noiseSeed, b->noiseSeed);
    for (int i = 0; i < 96; ++i) {
        d += !float_same(a->noiseOverlap[i], b->noiseOverlap[i]);
    }
    return d;
}

int
main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s tables.bin [codec] [S] [T] [ber%%]\n", argv[0]);
        return 2;
    }
    int codec = argc > 2 ? atoi(argv[2]) : 0;
    int S = argc > 3 ? atoi(argv[3]) : 256;
    int Tn = argc > 4 ? atoi(argv[4]) : 32;
    double ber = argc > 5 ? atof(argv[5]) : -1.0;

    FILE* f = fopen(argv[1], "rb");
    static uint8_t blob[1 << 17];
    size_t nb = f ? fread(blob, 1, sizeof(blob), f) : 0;
    if (mbxo_load_tables(blob, nb) != 0) {
        fprintf(stderr, "bad table blob\n");
        return 1;
    }

    long int_mismatch = 0, state_int_mismatch = 0, frames = 0;
    long n_repeat = 0, n_mute = 0, n_erasure = 0, n_tone = 0;
    double errsum = 0;
    fstat pcm = {0}, ml = {0}, l2 = {0}, phi = {0}, psi = {0}, uw = {0}, sc = {0};
    long pcm16_off1 = 0, pcm16_offmore = 0;

    sm_state = 0x9E3779B97F4A7C15ULL ^ (uint64_t)codec;
    for (int s = 0; s < S; ++s) {
        mbe_parms rc, rp, re, oc, op, oe;
        mbx_stream_rng rng;
        mbe_initMbeParms(&rc, &rp, &re);
        mbxo_init_parms(&oc, &op, &oe);
        if (memcmp(&rc, &oc, sizeof(rc)) || memcmp(&rp, &op, sizeof(rp))) {
            int_mismatch++;
        }
        mbe_setThreadRngSeed(1234u + (uint32_t)s);
        mbxo_rng_default(&rng);
        mbxo_rng_seed(&rng, 1234u + (uint32_t)s);
        for (int t = 0; t < Tn; ++t) {
            char fr[8][23];
            memset(fr, 0, sizeof(fr));
            char(*afr)[24] = (char(*)[24])fr; /* 4x24 view of the same 96 leading bytes */
            int ncell = codec == 0 ? 184 : 96;
            char* flat = (char*)fr;
            if (ber < 0) {
                for (int i = 0; i < ncell; ++i) {
                    flat[i] = (char)(splitmix64() & 1);
                }
            } else {
                /* start from an all-zero (valid code word) frame with a random b0 field, then flip */
                for (int i = 0; i < ncell; ++i) {
                    flat[i] = (char)(((double)(splitmix64() >> 11) / 9007199254740992.0) < ber / 100.0);
                }
            }
            float rf[160], of[160];
            short rs[160];
            int16_t os[160];
            char rd[88], od[88];
            mbe_process_result rr, orr;
            int r1, r2;
            if (codec == 0) {
                r1 = mbe_processImbe7200x4400Framef(rf, &rr, (const char(*)[23])fr, rd, &rc, &rp, &re);
                r2 = mbxo_process_imbe7200x4400_framef(of, &orr, (const char(*)[23])fr, od, &oc, &op, &oe, &rng);
                int_mismatch += memcmp(rd, od, 88) != 0;
            } else {
                r1 = mbe_processAmbe3600x2450Framef(rf, &rr, (const char(*)[24])afr, rd, &rc, &rp, &re);
                r2 = mbxo_process_ambe3600x2450_framef(of, &orr, (const char(*)[24])afr, od, &oc, &op, &oe, &rng);
                int_mismatch += memcmp(rd, od, 49) != 0;
            }
            mbe_floattoshort(rf, rs);
            mbxo_floattoshort(of, os);
            frames++;
            int_mismatch += (r1 != r2);
            int_mismatch += memcmp(&rr, &orr, sizeof(rr)) != 0;
            n_repeat += (rr.flags & MBE_PROCESS_FLAG_REPEAT) != 0;
            n_mute += (rr.flags & MBE_PROCESS_FLAG_MUTE) != 0;
            n_erasure += (rr.flags & MBE_PROCESS_FLAG_ERASURE) != 0;
            n_tone += (rr.flags & MBE_PROCESS_FLAG_TONE) != 0;
            errsum += r1;
            facc(&pcm, rf, of, 160);
            for (int i = 0; i < 160; ++i) {
                int d = abs((int)rs[i] - (int)os[i]);
                pcm16_off1 += d == 1;
                pcm16_offmore += d > 1;
            }
            const mbe_parms* R[3] = {&rc, &rp, &re};
            const mbe_parms* O[3] = {&oc, &op, &oe};
            for (int q = 0; q < 3; ++q) {
                long d = int_fields_differ(R[q], O[q]);
                if (d && state_int_mismatch < 5) {
                    printf("  state int mismatch s=%d t=%d struct=%d L=%d/%d flags=%x\n", s, t, q, R[q]->L, O[q]->L,
                           rr.flags);
                }
                state_int_mismatch += d;
                facc(&ml, R[q]->Ml, O[q]->Ml, 57);
                facc(&l2, R[q]->log2Ml, O[q]->log2Ml, 57);
                facc(&phi, R[q]->PHIl, O[q]->PHIl, 57);
                facc(&psi, R[q]->PSIl, O[q]->PSIl, 57);
                facc(&uw, R[q]->previousUw, O[q]->previousUw, 256);
                float ra[6] = {R[q]->w0, R[q]->gamma, R[q]->localEnergy, R[q]->errorRate, R[q]->mutingThreshold, 0};
                float oa[6] = {O[q]->w0, O[q]->gamma, O[q]->localEnergy, O[q]->errorRate, O[q]->mutingThreshold, 0};
                facc(&sc, ra, oa, 6);
            }
        }
    }
    printf("codec=%d S=%d T=%d ber=%g frames=%ld  mean errs/frame=%.2f repeat=%.1f%% mute=%.1f%% erasure=%.1f%% "
           "tone=%.1f%%\n",
           codec, S, Tn, ber, frames, errsum / frames, 100.0 * n_repeat / frames, 100.0 * n_mute / frames,
           100.0 * n_erasure / frames, 100.0 * n_tone / frames);
    printf("  integer mismatches (bits/ret/result): %ld   state integer/noise mismatches: %ld\n", int_mismatch,
           state_int_mismatch);
    frep("pcm float", &pcm);
    printf("  pcm int16: off-by-1=%ld  off-by->1=%ld of %ld\n", pcm16_off1, pcm16_offmore, frames * 160);
    frep("state Ml", &ml);
    frep("state log2Ml", &l2);
    frep("state PHIl", &phi);
    frep("state PSIl", &psi);
    frep("state prevUw", &uw);
    frep("state scalars", &sc);
    return (int_mismatch || state_int_mismatch) ? 1 : 0;
}
