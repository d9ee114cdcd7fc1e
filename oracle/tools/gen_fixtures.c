/*
 * gen_fixtures.c -- authoring-container tool (TEST INFRASTRUCTURE).
 *
 * Links the REAL reference (oracle/_ref/libmbe_ref.so, built by oracle/Makefile from the
 * sources where they lie) and writes golden input/output vectors -- DATA only -- under
 * tests/golden/.  The layouts are described in tests/golden/README.md and parsed by
 * tests/golden_io.py.
 *
 *   usage: oracle/_ref/gen_fixtures tests/golden [soft]
 *
 * All inputs are seeded (splitmix64, seed 0x9E3779B97F4A7C15 ^ tag) so the files are
 * reproducible.  Per-stream RNG seeds follow SURVEY.md §8(d): mbe_setThreadRngSeed(1234 + s).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mbx_types.h" /* ABI-identical structs; reference prototypes restated below */

/* reference entry points: include/mbelib-neo/mbelib.h:194-253, 395, 429, 471, 505, 596-675 */
extern void mbe_initMbeParms(mbe_parms*, mbe_parms*, mbe_parms*);
extern void mbe_setThreadRngSeed(uint32_t);
extern void mbe_floattoshort(const float*, short*);
extern void mbe_synthesizeSpeechf(float*, mbe_parms*, mbe_parms*);
extern void mbe_spectralAmpEnhance(mbe_parms*);
extern void mbe_applyAdaptiveSmoothing(mbe_parms*, const mbe_parms*);
extern void mbe_synthesizeComfortNoisef(float*);
extern int mbe_golay2312(const char*, char*);
extern int mbe_hamming1511(const char*, char*);
extern int mbe_checkGolayBlock(long int*);
extern int mbe_decodeImbe7200x4400Frame(const char[8][23], char[88], mbe_process_result*);
extern int mbe_decodeAmbe3600x2450Frame(const char[4][24], char[49], mbe_process_result*);
extern int mbe_decodeImbe4400Parms(const char*, mbe_parms*, mbe_parms*);
extern int mbe_decodeAmbe2450Parms(const char*, mbe_parms*, mbe_parms*);
extern int mbe_processImbe7200x4400Framef(float*, mbe_process_result*, const char[8][23], char[88], mbe_parms*,
                                          mbe_parms*, mbe_parms*);
extern int mbe_processAmbe3600x2450Framef(float*, mbe_process_result*, const char[4][24], char[49], mbe_parms*,
                                          mbe_parms*, mbe_parms*);
extern int mbe_processImbe4400Dataf(float*, mbe_process_result*, const char[88], mbe_parms*, mbe_parms*, mbe_parms*);
extern int mbe_processAmbe2450Dataf(float*, mbe_process_result*, const char[49], mbe_parms*, mbe_parms*, mbe_parms*); /* :415 */
/* tones: include/mbelib-neo/mbelib.h:630, 638 */
extern void mbe_synthesizeTonef(float*, const char*, mbe_parms*);
extern void mbe_synthesizeTonefdstar(float*, const char*, mbe_parms*, int);
/* AMBE 3600x2400 (D-STAR): include/mbelib-neo/mbelib.h:315-375 */
extern int mbe_processAmbe3600x2400Framef(float*, mbe_process_result*, const char[4][24], char[49], mbe_parms*, mbe_parms*,
                                          mbe_parms*);
extern int mbe_processAmbe2400Dataf(float*, mbe_process_result*, const char[49], mbe_parms*, mbe_parms*, mbe_parms*);
extern void mbe_initProcessResult(mbe_process_result*);
/* IMBE 7100x4400: include/mbelib-neo/mbelib.h:267, 533-590 */
extern int mbe_7100x4400hamming1511(const char*, char*);
extern int mbe_convertImbe7100to7200(char*);
extern int mbe_decodeImbe7100x4400Frame(const char[7][24], char[88], mbe_process_result*);
extern int mbe_7100x4400hamming1511Soft(const mbe_soft_bit*, char*);
extern int mbe_decodeImbe7100x4400SoftFrame(const mbe_soft_bit[7][24], char[88], mbe_process_result*);
extern int mbe_processImbe7100x4400Framef(float*, mbe_process_result*, const char[7][24], char[88], mbe_parms*, mbe_parms*,
                                          mbe_parms*);
/* soft-decision front end: include/mbelib-neo/mbelib.h:208-224, 246, 260, 437-447, 513-523 */
extern mbe_soft_bit mbe_softBitFromLlr(int16_t);
extern int mbe_softBitsFromLlr(const int16_t*, mbe_soft_bit*, size_t);
extern int mbe_softBitsFromHard(const char*, mbe_soft_bit*, size_t, uint8_t);
extern int mbe_golay2312Soft(const mbe_soft_bit*, char*);
extern int mbe_hamming1511Soft(const mbe_soft_bit*, char*);
extern int mbe_decodeImbe7200x4400SoftFrame(const mbe_soft_bit[8][23], char[88], mbe_process_result*);
extern int mbe_decodeAmbe3600x2450SoftFrame(const mbe_soft_bit[4][24], char[49], mbe_process_result*);
extern int mbe_processImbe7200x4400SoftFramef(float*, mbe_process_result*, const mbe_soft_bit[8][23], char[88], mbe_parms*,
                                              mbe_parms*, mbe_parms*);

static uint64_t sm_state;
static uint64_t
splitmix64(void) {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static uint32_t
fnv1a32(const void* data, size_t len) {
    const uint8_t* p = (const uint8_t*)data;
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < len; ++i) {
        h = (h ^ p[i]) * 16777619u;
    }
    return h;
}

static FILE*
open_out(const char* dir, const char* name) {
    char path[512];
    snprintf(path, sizeof(path), "%s/%s", dir, name);
    FILE* f = fopen(path, "wb");
    if (!f) {
        perror(path);
        exit(1);
    }
    return f;
}

#define W(f, p, n) fwrite((p), 1, (n), (f))

/* ------------------------------------------------------------------------------------ */
/* ecc_kat.bin: Golay -- 4 data words x 2048 parity-error patterns (every syndrome) plus
 * 4096 random 23-bit words; Hamming -- all 32768 15-bit words.
 * record: u32 in, u32 out, i32 errs                                                    */
static void
gen_ecc(const char* dir) {
    FILE* f = open_out(dir, "ecc_kat.bin");
    const uint32_t data_words[4] = {0x000, 0xA55, 0xFFF, 0x123};
    /* a code word for data d = the word the decoder leaves unchanged with zero syndrome:
     * find its parity by decoding (d<<11) and re-reading which parity gives syndrome 0 is not
     * exposed, so use the decoder itself: flip parity patterns over the word (d<<11). */
    uint32_t n = 0;
    long pos = ftell(f);
    W(f, &n, 4);
    for (int w = 0; w < 4; ++w) {
        for (uint32_t s = 0; s < 2048; ++s) {
            uint32_t in = (data_words[w] << 11) | s, out = 0;
            char a[23], b[23];
            for (int j = 0; j < 23; ++j) {
                a[j] = (char)((in >> j) & 1);
            }
            int32_t errs = mbe_golay2312(a, b);
            for (int j = 0; j < 23; ++j) {
                out |= (uint32_t)(b[j] & 1) << j;
            }
            W(f, &in, 4);
            W(f, &out, 4);
            W(f, &errs, 4);
            ++n;
        }
    }
    sm_state = 0x9E3779B97F4A7C15ULL ^ 0xECC;
    for (int i = 0; i < 4096; ++i) {
        uint32_t in = (uint32_t)(splitmix64() & 0x7fffff), out = 0;
        char a[23], b[23];
        for (int j = 0; j < 23; ++j) {
            a[j] = (char)((in >> j) & 1);
        }
        int32_t errs = mbe_golay2312(a, b);
        for (int j = 0; j < 23; ++j) {
            out |= (uint32_t)(b[j] & 1) << j;
        }
        W(f, &in, 4);
        W(f, &out, 4);
        W(f, &errs, 4);
        ++n;
    }
    /* the reference's own known answer, tests/test_ecc.c:356-375: data 0xA55, bit 5 flipped */
    {
        long blk;
        uint32_t cw = 0;
        /* build the code word: parity that yields syndrome 0 for data 0xA55 */
        for (uint32_t s = 0; s < 2048; ++s) {
            blk = (long)((0xA55u << 11) | s);
            long t = blk;
            mbe_checkGolayBlock(&t);
            char a[23], b[23];
            for (int j = 0; j < 23; ++j) {
                a[j] = (char)((blk >> j) & 1);
            }
            if (mbe_golay2312(a, b) == 0 && t == 0xA55) {
                /* zero data correction; pick the pattern whose single-bit flips all decode back */
                cw = (uint32_t)blk;
                uint32_t bad = cw ^ (1u << (11 + 5));
                long t2 = (long)bad;
                mbe_checkGolayBlock(&t2);
                if (t2 == 0xA55) {
                    break;
                }
            }
        }
        uint32_t in = cw ^ (1u << (11 + 5)), out = 0;
        char a[23], b[23];
        for (int j = 0; j < 23; ++j) {
            a[j] = (char)((in >> j) & 1);
        }
        int32_t errs = mbe_golay2312(a, b);
        for (int j = 0; j < 23; ++j) {
            out |= (uint32_t)(b[j] & 1) << j;
        }
        W(f, &in, 4);
        W(f, &out, 4);
        W(f, &errs, 4);
        ++n;
    }
    uint32_t nh = 32768;
    W(f, &nh, 4);
    for (uint32_t in = 0; in < 32768; ++in) {
        char a[15], b[15];
        uint32_t out = 0;
        for (int j = 0; j < 15; ++j) {
            a[j] = (char)((in >> j) & 1);
        }
        int32_t errs = mbe_hamming1511(a, b);
        for (int j = 0; j < 15; ++j) {
            out |= (uint32_t)(b[j] & 1) << j;
        }
        W(f, &in, 4);
        W(f, &out, 4);
        W(f, &errs, 4);
    }
    fseek(f, pos, SEEK_SET);
    W(f, &n, 4);
    fclose(f);
    printf("ecc_kat.bin: %u golay + %u hamming records\n", n, nh);
}

/* ------------------------------------------------------------------------------------ */
/* fec_{imbe,ambe}.bin: N random-bit frames.
 * record: chars[184|96] frame, chars[88|49] params, i32 ret, result(20 B)              */
static void
gen_fec(const char* dir, int codec, int N) {
    FILE* f = open_out(dir, codec == 0 ? "fec_imbe.bin" : "fec_ambe.bin");
    uint32_t n = (uint32_t)N;
    W(f, &n, 4);
    sm_state = 0x9E3779B97F4A7C15ULL ^ (0xFEC0 + (uint64_t)codec);
    const int ncell = codec == 0 ? 184 : 96, nd = codec == 0 ? 88 : 49;
    for (int i = 0; i < N; ++i) {
        char fr[184], d[88];
        mbe_process_result r;
        /* one frame in eight keeps a clean C0 (few flips) so low-error paths are covered too */
        int sparse = (i % 8) == 7;
        for (int c = 0; c < ncell; ++c) {
            uint64_t v = splitmix64();
            fr[c] = sparse ? (char)((v % 29) == 0) : (char)(v & 1);
        }
        int32_t ret = codec == 0 ? mbe_decodeImbe7200x4400Frame((const char(*)[23])fr, d, &r)
                                 : mbe_decodeAmbe3600x2450Frame((const char(*)[24])fr, d, &r);
        W(f, fr, ncell);
        W(f, d, nd);
        W(f, &ret, 4);
        W(f, &r, sizeof(r));
    }
    fclose(f);
    printf("%s: %d frames\n", codec == 0 ? "fec_imbe.bin" : "fec_ambe.bin", N);
}

/* ------------------------------------------------------------------------------------ */
/* stream_{imbe,ambe}.bin: S streams x T random-bit frames through the full frame entry.
 * header: u32 S, u32 T
 * per frame: chars[184|96], chars[88|49], i32 ret, result(20), float pcm[160], i16 pcm[160],
 *            u32 fnv(cur), u32 fnv(prev), u32 fnv(prev_enh)   [hash of the integer-exact fields]
 * per stream (after its T frames): cur, prev, prev_enh (3 x 2604 B)                     */
static uint32_t
state_digest(const mbe_parms* p) { /* integer-valued fields only */
    uint32_t h = 2166136261u;
#define MIX(x) h = (h ^ fnv1a32(&(x), sizeof(x))) * 16777619u
    MIX(p->L);
    MIX(p->K);
    MIX(p->Vl);
    MIX(p->tonePhase);
    MIX(p->swn);
    MIX(p->amplitudeThreshold);
    MIX(p->errorCountTotal);
    MIX(p->errorCount4);
    MIX(p->repeatCount);
    MIX(p->noiseSeed);
    MIX(p->noiseOverlap);
#undef MIX
    return h;
}

static void
gen_stream(const char* dir, int codec, int S, int T) {
    FILE* f = open_out(dir, codec == 0 ? "stream_imbe.bin" : "stream_ambe.bin");
    uint32_t hdr[2] = {(uint32_t)S, (uint32_t)T};
    W(f, hdr, 8);
    sm_state = 0x9E3779B97F4A7C15ULL ^ (0x57E0 + (uint64_t)codec);
    const int ncell = codec == 0 ? 184 : 96, nd = codec == 0 ? 88 : 49;
    long rep = 0, mute = 0, era = 0, tone = 0, w0zero = 0;
    for (int s = 0; s < S; ++s) {
        mbe_parms cur, prev, enh;
        mbe_initMbeParms(&cur, &prev, &enh);
        mbe_setThreadRngSeed(1234u + (uint32_t)s);
        /* stream classes: 0 raw random bits; 1 light errors (few set cells); 2 bursts */
        int cls = s % 4;
        for (int t = 0; t < T; ++t) {
            char fr[184], d[88];
            memset(fr, 0, sizeof(fr));
            for (int c = 0; c < ncell; ++c) {
                uint64_t v = splitmix64();
                if (cls == 1) {
                    fr[c] = (char)((v % 23) == 0);
                } else if (cls == 2 && (t % 7) < 3) {
                    fr[c] = (char)((v % 11) == 0);
                } else {
                    fr[c] = (char)(v & 1);
                }
            }
            float pcm[160];
            short s16[160];
            mbe_process_result r;
            int32_t ret = codec == 0
                              ? mbe_processImbe7200x4400Framef(pcm, &r, (const char(*)[23])fr, d, &cur, &prev, &enh)
                              : mbe_processAmbe3600x2450Framef(pcm, &r, (const char(*)[24])fr, d, &cur, &prev, &enh);
            mbe_floattoshort(pcm, s16);
            rep += (r.flags & MBE_PROCESS_FLAG_REPEAT) != 0;
            mute += (r.flags & MBE_PROCESS_FLAG_MUTE) != 0;
            era += (r.flags & MBE_PROCESS_FLAG_ERASURE) != 0;
            tone += (r.flags & MBE_PROCESS_FLAG_TONE) != 0;
            w0zero += (cur.w0 == 0.0f) && (r.flags & MBE_PROCESS_FLAG_REPEAT);
            uint32_t dg[3] = {state_digest(&cur), state_digest(&prev), state_digest(&enh)};
            W(f, fr, ncell);
            W(f, d, nd);
            W(f, &ret, 4);
            W(f, &r, sizeof(r));
            W(f, pcm, sizeof(pcm));
            W(f, s16, sizeof(s16));
            W(f, dg, sizeof(dg));
        }
        W(f, &cur, sizeof(cur));
        W(f, &prev, sizeof(prev));
        W(f, &enh, sizeof(enh));
    }
    fclose(f);
    printf("%s: %dx%d frames, repeat=%ld mute=%ld erasure=%ld tone=%ld erasure->repeat(w0=0)=%ld\n",
           codec == 0 ? "stream_imbe.bin" : "stream_ambe.bin", S, T, rep, mute, era, tone, w0zero);
}

/* ------------------------------------------------------------------------------------ */
/* golden_synth.bin: the reference's tools/gen_golden.c scenario (:38-64).
 * cur(2604) prev(2604) as passed in, float pcm[160], i16 pcm[160], u32 hash_f32, u32 hash_s16,
 * then cur/prev after the call.                                                          */
static void
gen_golden_synth(const char* dir) {
    FILE* f = open_out(dir, "golden_synth.bin");
    mbe_parms cur, prev, enh;
    mbe_setThreadRngSeed(0xC0FFEEu);
    mbe_initMbeParms(&cur, &prev, &enh);
    cur.w0 = 0.105f;
    cur.L = 36;
    for (int l = 1; l <= cur.L; ++l) {
        cur.Vl[l] = (l % 4) ? 1 : 0;
        cur.Ml[l] = 0.035f + 0.0015f * (float)l;
        cur.PHIl[l] = (float)l * 0.03f;
        cur.PSIl[l] = (float)l * 0.02f;
    }
    prev = cur;
    W(f, &cur, sizeof(cur));
    W(f, &prev, sizeof(prev));
    float pcm[160];
    short s16[160];
    mbe_synthesizeSpeechf(pcm, &cur, &prev);
    mbe_floattoshort(pcm, s16);
    uint32_t h[2] = {fnv1a32(pcm, sizeof(pcm)), fnv1a32(s16, sizeof(s16))};
    W(f, pcm, sizeof(pcm));
    W(f, s16, sizeof(s16));
    W(f, h, sizeof(h));
    W(f, &cur, sizeof(cur));
    W(f, &prev, sizeof(prev));
    fclose(f);
    printf("golden_synth.bin: F32 0x%08X S16 0x%08X\n", h[0], h[1]);
}

/* ------------------------------------------------------------------------------------ */
/* synth_seq.bin: the reference's bench recipes as multi-frame known answers:
 *   recipe 0 = bench/bench_synth.c:40-67 (L=40, mixed), recipe 1 = bench/bench_unvoiced.c:33-52,87
 * header u32 nrec, u32 frames; per recipe per frame: float pcm[160]; final cur, prev.    */
static void
gen_synth_seq(const char* dir, int frames) {
    FILE* f = open_out(dir, "synth_seq.bin");
    uint32_t hdr[2] = {2, (uint32_t)frames};
    W(f, hdr, 8);
    for (int recipe = 0; recipe < 2; ++recipe) {
        mbe_parms cur, prev, enh;
        float out[160];
        mbe_setThreadRngSeed(recipe == 0 ? 0x123456u : 0xBEEFu);
        mbe_initMbeParms(&cur, &prev, &enh);
        if (recipe == 0) {
            cur.w0 = 0.09378f;
            cur.L = 40;
            for (int l = 1; l <= cur.L; ++l) {
                cur.Vl[l] = (l % 3) != 0;
                cur.Ml[l] = 0.05f + 0.002f * l;
                cur.log2Ml[l] = 0.0f;
                cur.PHIl[l] = (float)l * 0.1f;
                cur.PSIl[l] = (float)l * 0.05f;
            }
        } else {
            cur.w0 = 0.11f;
            cur.L = 36;
            for (int l = 1; l <= cur.L; ++l) {
                cur.Vl[l] = 0;
                cur.Ml[l] = 0.03f + 0.002f * (float)(l & 7);
                cur.PHIl[l] = 0.0f;
                cur.PSIl[l] = 0.0f;
            }
        }
        prev = cur;
        for (int i = 0; i < frames; ++i) {
            if (recipe == 0) {
                cur.w0 = (i & 1) ? 0.09f : 0.11f;
                for (int l = 1; l <= cur.L; ++l) {
                    cur.Vl[l] = ((i + l) % 5) ? 1 : 0;
                    cur.Ml[l] = 0.04f + 0.003f * (float)((i + l) % 7);
                }
            } else {
                cur.w0 = (i & 1) ? 0.10f : 0.12f;
            }
            mbe_synthesizeSpeechf(out, &cur, &prev);
            prev = cur;
            W(f, out, sizeof(out));
        }
        W(f, &cur, sizeof(cur));
        W(f, &prev, sizeof(prev));
    }
    fclose(f);
    printf("synth_seq.bin: 2 recipes x %d frames\n", frames);
}

/* ------------------------------------------------------------------------------------ */
/* f2s_kat.bin: tests/test_floattoshort_parity.c:36-68 inputs (4 seeds x 160, 12 edge values
 * each) and the reference's outputs.  u32 n; then per case float in[160], i16 out[160].   */
static void
gen_f2s(const char* dir) {
    FILE* f = open_out(dir, "f2s_kat.bin");
    const uint32_t seeds[4] = {0x00000001u, 0x12345678u, 0x00C0FFEEu, 0xFFFFFFFFu};
    uint32_t n = 4;
    W(f, &n, 4);
    const float clip_point = (32767.0f * 0.95f) / 7.0f;
    for (int s = 0; s < 4; ++s) {
        float in[160];
        short out[160];
        uint32_t state = seeds[s];
        for (int i = 0; i < 160; ++i) {
            state = (state * 1664525u) + 1013904223u;
            int32_t v = (int32_t)(state >> 8) - 0x007FFFFF;
            in[i] = (float)v / 65536.0f;
        }
        in[0] = 0.0f;
        in[1] = clip_point;
        in[2] = clip_point + (1.0f / 32768.0f);
        in[3] = clip_point - (1.0f / 32768.0f);
        in[4] = -clip_point;
        in[5] = -clip_point - (1.0f / 32768.0f);
        in[6] = -clip_point + (1.0f / 32768.0f);
        in[7] = 1.0f / 7.0f;
        in[8] = -1.0f / 7.0f;
        in[9] = NAN;
        in[10] = INFINITY;
        in[11] = -INFINITY;
        mbe_floattoshort(in, out);
        W(f, in, sizeof(in));
        W(f, out, sizeof(out));
    }
    fclose(f);
    printf("f2s_kat.bin: 4 cases\n");
}

/* ------------------------------------------------------------------------------------ */
/* params_kat.bin: parameter-decode known answers over every fundamental index.
 * IMBE: for b0 in 0..255: i32 rc, float w0, i32 L, i32 K (zero parameter bits otherwise,
 * state from mbe_initMbeParms) -- generalises tests/test_params.c:227-253.
 * AMBE: for b0 in 0..127: i32 rc, float w0, i32 L -- generalises tests/test_params.c:303-341.
 * Then two full decodes with random parameter bits per codec x 64: chars bits, cur after.  */
static void
gen_params(const char* dir) {
    FILE* f = open_out(dir, "params_kat.bin");
    for (int b0 = 0; b0 < 256; ++b0) {
        char d[88];
        memset(d, 0, sizeof(d));
        for (int i = 0; i < 8; ++i) {
            int bit = (b0 >> (7 - i)) & 1;
            d[i < 6 ? i : (i == 6 ? 85 : 86)] = (char)bit;
        }
        mbe_parms cur, prev, enh;
        mbe_initMbeParms(&cur, &prev, &enh);
        int32_t rc = mbe_decodeImbe4400Parms(d, &cur, &prev);
        W(f, &rc, 4);
        W(f, &cur.w0, 4);
        W(f, &cur.L, 4);
        W(f, &cur.K, 4);
    }
    for (int b0 = 0; b0 < 128; ++b0) {
        char d[49];
        memset(d, 0, sizeof(d));
        d[0] = (char)((b0 >> 6) & 1);
        d[1] = (char)((b0 >> 5) & 1);
        d[2] = (char)((b0 >> 4) & 1);
        d[3] = (char)((b0 >> 3) & 1);
        d[37] = (char)((b0 >> 2) & 1);
        d[38] = (char)((b0 >> 1) & 1);
        d[39] = (char)(b0 & 1);
        mbe_parms cur, prev, enh;
        mbe_initMbeParms(&cur, &prev, &enh);
        int32_t rc = mbe_decodeAmbe2450Parms(d, &cur, &prev);
        W(f, &rc, 4);
        W(f, &cur.w0, 4);
        W(f, &cur.L, 4);
    }
    sm_state = 0x9E3779B97F4A7C15ULL ^ 0x9A7A;
    for (int codec = 0; codec < 2; ++codec) {
        for (int i = 0; i < 64; ++i) {
            char d[88];
            int nd = codec == 0 ? 88 : 49;
            for (int c = 0; c < nd; ++c) {
                d[c] = (char)(splitmix64() & 1);
            }
            mbe_parms cur, prev, enh;
            mbe_initMbeParms(&cur, &prev, &enh);
            /* give the prediction memory some shape */
            for (int l = 0; l <= 56; ++l) {
                prev.log2Ml[l] = 0.25f * (float)((l * 7) % 11) - 1.0f;
                prev.Ml[l] = exp2f(prev.log2Ml[l]);
            }
            prev.L = 9 + (int)(splitmix64() % 48);
            prev.gamma = 1.5f;
            int32_t rc = codec == 0 ? mbe_decodeImbe4400Parms(d, &cur, &prev) : mbe_decodeAmbe2450Parms(d, &cur, &prev);
            W(f, d, nd);
            W(f, &prev.L, 4);
            W(f, &rc, 4);
            W(f, &cur, sizeof(cur));
        }
    }
    fclose(f);
    printf("params_kat.bin written\n");
}

/* ------------------------------------------------------------------------------------ */
/* misc_kat.bin: scalar behaviours pinned by tests/test_params.c:
 *  (1) :573-594  Tm may go negative: amplitudeThreshold, Ml[1] after mbe_applyAdaptiveSmoothing
 *  (2) :596-618  seeded comfort noise (seed 0x12345678): float[160]; cold-start seed 0x1234
 *  (3) :717-740  IMBE repeat headroom reset: cur after the call (L must be 39)            */
static void
gen_misc(const char* dir) {
    FILE* f = open_out(dir, "misc_kat.bin");
    {
        mbe_parms cur, prev, enh;
        mbe_initMbeParms(&cur, &prev, &enh);
        cur.L = 4;
        for (int l = 1; l <= cur.L; ++l) {
            cur.Ml[l] = 10.0f;
            cur.Vl[l] = 0;
        }
        cur.errorRate = 0.5f;
        cur.errorCountTotal = 30;
        cur.errorCount4 = 2;
        prev.amplitudeThreshold = 1;
        mbe_applyAdaptiveSmoothing(&cur, &prev);
        W(f, &cur.amplitudeThreshold, 4);
        W(f, &cur.Ml[1], 4);
        W(f, &cur.localEnergy, 4);
    }
    {
        float n[160];
        mbe_setThreadRngSeed(0x12345678u);
        mbe_synthesizeComfortNoisef(n);
        W(f, n, sizeof(n));
        mbe_parms cur, prev, enh;
        mbe_initMbeParms(&cur, &prev, &enh);
        cur.w0 = 0.10f;
        cur.L = 12;
        for (int l = 1; l <= cur.L; ++l) {
            cur.Vl[l] = (l % 3) ? 1 : 0;
            cur.Ml[l] = 0.03f + (0.001f * (float)l);
        }
        prev = cur;
        mbe_setThreadRngSeed(0x1234u);
        mbe_synthesizeSpeechf(n, &cur, &prev);
        W(f, &cur.noiseSeed, 4);
    }
    {
        char d[88];
        float out[160];
        memset(d, 0, sizeof(d));
        mbe_process_result r;
        memset(&r, 0, sizeof(r));
        r.total_errors = 6;
        mbe_parms cur, prev, enh;
        mbe_initMbeParms(&cur, &prev, &enh);
        prev.repeatCount = 4;
        cur = prev;
        mbe_setThreadRngSeed(77u);
        int32_t rc = mbe_processImbe4400Dataf(out, &r, d, &cur, &prev, &enh);
        W(f, &rc, 4);
        W(f, &r, sizeof(r));
        W(f, &cur, sizeof(cur));
        W(f, out, sizeof(out));
    }
    fclose(f);
    printf("misc_kat.bin written\n");
}

/* ------------------------------------------------------------------------------------ */
/* soft_kat.bin: the soft-decision front end.
 *   u32 NG, NG x { soft[23] (bit, reliability), char out[23], i32 ret }      mbe_golay2312Soft
 *   u32 NH, NH x { soft[15], char out[15], i32 ret }                          mbe_hamming1511Soft
 *   u32 NI, NI x { soft[8][23], char d[88], i32 ret, result(20) }             mbe_decodeImbe7200x4400SoftFrame
 *   u32 NA, NA x { soft[4][24], char d[49], i32 ret, result(20) }             mbe_decodeAmbe3600x2450SoftFrame
 *   u32 NL, NL x { i16 llr, soft }                                            mbe_softBitFromLlr
 *   u32 NP, NP x { soft[8][23], i32 ret, result(20), float pcm[160] }         mbe_processImbe7200x4400SoftFramef,
 *                                                                             one stream, mbe_setThreadRngSeed(4242)
 * Reliability profiles rotate per case: 0 uniform random, 1 all equal (tie-breaks decide),
 * 2 two-level {3, 200}, 3 all zero, 4 random with the flipped positions weak (a clean code word
 * with a few low-confidence errors -- the realistic case).                                  */
static void
soft_fill(mbe_soft_bit* s, int n, int profile) {
    for (int i = 0; i < n; ++i) {
        uint64_t v = splitmix64();
        s[i].bit = (uint8_t)(v & 1u);
        switch (profile) {
            case 0: s[i].reliability = (uint8_t)(v >> 8); break;
            case 1: s[i].reliability = 77; break;
            case 2: s[i].reliability = ((v >> 8) & 1u) ? 200 : 3; break;
            case 3: s[i].reliability = 0; break;
            default: s[i].reliability = (uint8_t)(128u + ((v >> 8) & 127u)); break;
        }
    }
}

/* profile 4: overwrite a block with a valid code word (through the hard decoder) and damage it */
static void
soft_damage_golay(mbe_soft_bit* s) {
    char in[23], out[23];
    for (int i = 0; i < 23; ++i) {
        in[i] = (char)s[i].bit;
    }
    mbe_golay2312(in, out);
    /* out = corrected data + original parity: re-encode by decoding once more is not needed, a word
     * within distance 3 of a code word is what a receiver sees */
    for (int i = 0; i < 23; ++i) {
        s[i].bit = (uint8_t)out[i];
    }
    int flips = (int)(splitmix64() % 5u);
    for (int k = 0; k < flips; ++k) {
        int at = (int)(splitmix64() % 23u);
        s[at].bit ^= 1u;
        s[at].reliability = (uint8_t)(splitmix64() % 40u);
    }
}

static void
gen_soft(const char* dir) {
    FILE* f = open_out(dir, "soft_kat.bin");
    sm_state = 0x9E3779B97F4A7C15ULL ^ 0x50F7ULL;
    uint32_t n = 640;
    W(f, &n, 4);
    for (uint32_t i = 0; i < n; ++i) {
        mbe_soft_bit s[23];
        char out[23];
        soft_fill(s, 23, (int)(i % 5u));
        if ((i % 5u) == 4u) {
            soft_damage_golay(s);
        }
        int32_t ret = mbe_golay2312Soft(s, out);
        W(f, s, sizeof(s));
        W(f, out, 23);
        W(f, &ret, 4);
    }
    n = 640;
    W(f, &n, 4);
    for (uint32_t i = 0; i < n; ++i) {
        mbe_soft_bit s[15];
        char out[15];
        soft_fill(s, 15, (int)(i % 5u));
        int32_t ret = mbe_hamming1511Soft(s, out);
        W(f, s, sizeof(s));
        W(f, out, 15);
        W(f, &ret, 4);
    }
    n = 160;
    W(f, &n, 4);
    for (uint32_t i = 0; i < n; ++i) {
        mbe_soft_bit fr[8][23];
        char d[88];
        mbe_process_result r;
        soft_fill(&fr[0][0], 184, (int)(i % 5u));
        if ((i % 5u) == 4u) {
            for (int row = 0; row < 4; ++row) {
                soft_damage_golay(fr[row]);
            }
        }
        int32_t ret = mbe_decodeImbe7200x4400SoftFrame((const mbe_soft_bit(*)[23])fr, d, &r);
        W(f, fr, sizeof(fr));
        W(f, d, 88);
        W(f, &ret, 4);
        W(f, &r, sizeof(r));
    }
    n = 320;
    W(f, &n, 4);
    for (uint32_t i = 0; i < n; ++i) {
        mbe_soft_bit fr[4][24];
        char d[49];
        mbe_process_result r;
        soft_fill(&fr[0][0], 96, (int)(i % 5u));
        if ((i % 5u) == 4u) {
            soft_damage_golay(&fr[0][1]);
            soft_damage_golay(&fr[1][0]);
        }
        int32_t ret = mbe_decodeAmbe3600x2450SoftFrame((const mbe_soft_bit(*)[24])fr, d, &r);
        W(f, fr, sizeof(fr));
        W(f, d, 49);
        W(f, &ret, 4);
        W(f, &r, sizeof(r));
    }
    {
        static const int16_t llr[] = {0, 1, -1, 2, -2, 100, -100, 254, -254, 255, -255, 256, -256, 1000, -1000, 32767, -32767, -32768};
        n = (uint32_t)(sizeof(llr) / sizeof(llr[0]));
        W(f, &n, 4);
        for (uint32_t i = 0; i < n; ++i) {
            mbe_soft_bit s = mbe_softBitFromLlr(llr[i]);
            W(f, &llr[i], 2);
            W(f, &s, sizeof(s));
        }
    }
    {
        n = 12;
        W(f, &n, 4);
        mbe_parms cur, prev, enh;
        mbe_initMbeParms(&cur, &prev, &enh);
        mbe_setThreadRngSeed(4242u);
        for (uint32_t i = 0; i < n; ++i) {
            mbe_soft_bit fr[8][23];
            char d[88];
            float out[160];
            mbe_process_result r;
            soft_fill(&fr[0][0], 184, (i % 3u) ? 4 : 0);
            if (i % 3u) {
                for (int row = 0; row < 4; ++row) {
                    soft_damage_golay(fr[row]);
                }
            }
            int32_t ret = mbe_processImbe7200x4400SoftFramef(out, &r, (const mbe_soft_bit(*)[23])fr, d, &cur, &prev, &enh);
            W(f, fr, sizeof(fr));
            W(f, &ret, 4);
            W(f, &r, sizeof(r));
            W(f, out, sizeof(out));
        }
    }
    fclose(f);
    printf("soft_kat.bin written\n");
}

/* ------------------------------------------------------------------------------------ */
/* imbe7100_kat.bin: the IMBE 7100x4400 front end.
 *   u32 NH, NH x { u32 in, u32 out, i32 errs }                 mbe_7100x4400hamming1511, all 32768 words
 *   u32 NC, NC x { char in[88], char out[88] }                 mbe_convertImbe7100to7200
 *   u32 NF, NF x { char fr[7][24], char d[88], i32 ret, result(20) }   mbe_decodeImbe7100x4400Frame
 *   u32 S, u32 T, per stream T x { char fr[7][24], i32 ret, result(20), float pcm[160] }, then cur (2604 B)
 *                                                              mbe_processImbe7100x4400Framef, seeds 1234 + s */
static void
gen_imbe7100(const char* dir) {
    FILE* f = open_out(dir, "imbe7100_kat.bin");
    sm_state = 0x9E3779B97F4A7C15ULL ^ 0x7100ULL;
    uint32_t n = 32768;
    W(f, &n, 4);
    for (uint32_t w = 0; w < n; ++w) {
        char in[15], out[15];
        for (int j = 0; j < 15; ++j) {
            in[j] = (char)((w >> j) & 1u);
        }
        int32_t errs = mbe_7100x4400hamming1511(in, out);
        uint32_t o = 0;
        for (int j = 0; j < 15; ++j) {
            o |= (uint32_t)(out[j] & 1) << j;
        }
        W(f, &w, 4);
        W(f, &o, 4);
        W(f, &errs, 4);
    }
    n = 512;
    W(f, &n, 4);
    for (uint32_t i = 0; i < n; ++i) {
        char in[88], out[88];
        for (int c = 0; c < 88; ++c) {
            in[c] = (char)(splitmix64() & 1u);
        }
        memcpy(out, in, 88);
        mbe_convertImbe7100to7200(out);
        W(f, in, 88);
        W(f, out, 88);
    }
    n = 2048;
    W(f, &n, 4);
    for (uint32_t i = 0; i < n; ++i) {
        char fr[7][24], d[88];
        mbe_process_result r;
        int sparse = (i % 8u) == 7u;
        for (int c = 0; c < 7 * 24; ++c) {
            uint64_t v = splitmix64();
            ((char*)fr)[c] = sparse ? (char)((v % 29) == 0) : (char)(v & 1);
        }
        int32_t ret = mbe_decodeImbe7100x4400Frame((const char(*)[24])fr, d, &r);
        W(f, fr, sizeof(fr));
        W(f, d, 88);
        W(f, &ret, 4);
        W(f, &r, sizeof(r));
    }
    uint32_t S = 16, T = 12;
    W(f, &S, 4);
    W(f, &T, 4);
    for (uint32_t s = 0; s < S; ++s) {
        mbe_parms cur, prev, enh;
        mbe_initMbeParms(&cur, &prev, &enh);
        mbe_setThreadRngSeed(1234u + s);
        for (uint32_t t = 0; t < T; ++t) {
            char fr[7][24], d[88];
            float out[160];
            mbe_process_result r;
            for (int c = 0; c < 7 * 24; ++c) {
                ((char*)fr)[c] = (char)(splitmix64() & 1u);
            }
            int32_t ret = mbe_processImbe7100x4400Framef(out, &r, (const char(*)[24])fr, d, &cur, &prev, &enh);
            W(f, fr, sizeof(fr));
            W(f, &ret, 4);
            W(f, &r, sizeof(r));
            W(f, out, sizeof(out));
        }
        W(f, &cur, sizeof(cur));
    }
    /* soft-decision 7100x4400: u32 NHS, NHS x { soft[15], char out[15], i32 ret }   mbe_7100x4400hamming1511Soft
     *                           u32 NFS, NFS x { soft[7][24], char d[88], i32 ret, result(20) }  mbe_decodeImbe7100x4400SoftFrame */
    n = 320;
    W(f, &n, 4);
    for (uint32_t i = 0; i < n; ++i) {
        mbe_soft_bit sb[15];
        char out[15];
        soft_fill(sb, 15, (int)(i % 5u));
        int32_t ret = mbe_7100x4400hamming1511Soft(sb, out);
        W(f, sb, sizeof(sb));
        W(f, out, 15);
        W(f, &ret, 4);
    }
    n = 160;
    W(f, &n, 4);
    for (uint32_t i = 0; i < n; ++i) {
        mbe_soft_bit fr[7][24];
        char d[88];
        mbe_process_result r;
        soft_fill(&fr[0][0], 168, (int)(i % 5u));
        if ((i % 5u) == 4u) {
            soft_damage_golay(&fr[1][1]);
            soft_damage_golay(&fr[2][0]);
            soft_damage_golay(&fr[3][0]);
        }
        int32_t ret = mbe_decodeImbe7100x4400SoftFrame((const mbe_soft_bit(*)[24])fr, d, &r);
        W(f, fr, sizeof(fr));
        W(f, d, 88);
        W(f, &ret, 4);
        W(f, &r, sizeof(r));
    }
    fclose(f);
    printf("imbe7100_kat.bin written\n");
}

/* ------------------------------------------------------------------------------------ */
/* ambe2400_kat.bin: AMBE 3600x2400 / D-STAR.
 *   u32 S, u32 T, per stream T x { char fr[4][24], char d[49], i32 ret, result(20), float pcm[160] },
 *                 then cur, prev, prev_enh (3 x 2604 B)        mbe_processAmbe3600x2400Framef, random cells,
 *                                                              seeds 1234 + s
 *   u32 S2, u32 T2, per stream T2 x { char d[49], i32 total_in, i32 ret, result(20), float pcm[160] },
 *                 then cur, prev, prev_enh                      mbe_processAmbe2400Dataf on scripted parameter
 *                                                              bits: voice, valid tones, silence / invalid tone
 *                                                              classes, with a given total error count; seeds 5000 + s */
static void
put_bits_msb(char* d, const int* idx, int n, int value) {
    for (int i = 0; i < n; ++i) {
        d[idx[i]] = (char)((value >> (n - 1 - i)) & 1);
    }
}

static void
gen_ambe2400(const char* dir) {
    FILE* f = open_out(dir, "ambe2400_kat.bin");
    sm_state = 0x9E3779B97F4A7C15ULL ^ 0x2400ULL;
    uint32_t S = 32, T = 24;
    W(f, &S, 4);
    W(f, &T, 4);
    for (uint32_t s = 0; s < S; ++s) {
        mbe_parms cur, prev, enh;
        mbe_initMbeParms(&cur, &prev, &enh);
        mbe_setThreadRngSeed(1234u + s);
        for (uint32_t t = 0; t < T; ++t) {
            char fr[4][24], d[49];
            float out[160];
            mbe_process_result r;
            int clean = (s % 4u) == 3u; /* a quarter of the streams: few channel errors */
            for (int c = 0; c < 96; ++c) {
                uint64_t v = splitmix64();
                ((char*)fr)[c] = clean ? (char)((v % 23) == 0) : (char)(v & 1);
            }
            int32_t ret = mbe_processAmbe3600x2400Framef(out, &r, (const char(*)[24])fr, d, &cur, &prev, &enh);
            W(f, fr, sizeof(fr));
            W(f, d, 49);
            W(f, &ret, 4);
            W(f, &r, sizeof(r));
            W(f, out, sizeof(out));
        }
        W(f, &cur, sizeof(cur));
        W(f, &prev, sizeof(prev));
        W(f, &enh, sizeof(enh));
    }
    S = 24;
    T = 24;
    W(f, &S, 4);
    W(f, &T, 4);
    static const int tone_list[12] = {5, 6, 7, 20, 64, 122, 123, 4, 130, 163, 164, 255};
    static const int totals[8] = {0, 0, 0, 0, 1, 2, 4, 5};
    static const int i_tone_hi[3] = {6, 7, 8};
    for (uint32_t s = 0; s < S; ++s) {
        mbe_parms cur, prev, enh;
        mbe_initMbeParms(&cur, &prev, &enh);
        mbe_setThreadRngSeed(5000u + s);
        for (uint32_t t = 0; t < T; ++t) {
            char d[49];
            float out[160];
            for (int c = 0; c < 49; ++c) {
                d[c] = (char)(splitmix64() & 1u);
            }
            uint64_t pick = splitmix64() % 10u;
            if (pick < 7) { /* voice: b0 (bits 0..5, 48) must not be 126/127 */
                if (d[0] && d[1] && d[2] && d[3] && d[4] && d[5]) {
                    d[(int)(splitmix64() % 6u)] = 0;
                }
            } else { /* tone class with a chosen index */
                for (int c = 0; c < 6; ++c) {
                    d[c] = 1;
                }
                int idx = tone_list[splitmix64() % 12u];
                /* index bits: 7..5 through the def tables (inverse: def with (t7,t6,t5) = top three bits) */
                static const int def_of[8] = {1, 2, 3, 4, 0, 7, 6, 5}; /* (t7 t6 t5) as a number -> def */
                put_bits_msb(d, i_tone_hi, 3, def_of[(idx >> 5) & 7]);
                d[9] = (char)((idx >> 4) & 1);
                d[42] = (char)((idx >> 3) & 1);
                d[43] = (char)((idx >> 2) & 1);
                d[10] = (char)((idx >> 1) & 1);
                d[11] = (char)(idx & 1);
            }
            mbe_process_result r;
            mbe_initProcessResult(&r);
            int32_t total_in = totals[splitmix64() % 8u];
            r.total_errors = total_in;
            int32_t ret = mbe_processAmbe2400Dataf(out, &r, d, &cur, &prev, &enh);
            W(f, d, 49);
            W(f, &total_in, 4);
            W(f, &ret, 4);
            W(f, &r, sizeof(r));
            W(f, out, sizeof(out));
        }
        W(f, &cur, sizeof(cur));
        W(f, &prev, sizeof(prev));
        W(f, &enh, sizeof(enh));
    }
    fclose(f);
    printf("ambe2400_kat.bin written\n");
}

/* ------------------------------------------------------------------------------------ */
/* tone_kat.bin: mbe_synthesizeTonef / mbe_synthesizeTonefdstar on one evolving mbe_parms each.
 *   u32 N1, N1 x { char d[49], float pcm[160], i32 swn, u32 tonePhase }     AMBE+2 tones from parameter bits
 *   u32 N2, N2 x { i32 id, float pcm[160], i32 swn, u32 tonePhase }         D-STAR tone indices              */
static void
gen_tones(const char* dir) {
    FILE* f = open_out(dir, "tone_kat.bin");
    static const int ids[16] = {5, 6, 7, 64, 122, 123, 128, 129, 143, 144, 150, 163, 164, 255, 4, 0};
    static const int ads[3] = {127, 64, 3};
    mbe_parms cur, prev, enh;
    mbe_initMbeParms(&cur, &prev, &enh);
    uint32_t n = 16 * 3;
    W(f, &n, 4);
    for (int i = 0; i < 16; ++i) {
        for (int a = 0; a < 3; ++a) {
            char d[49];
            float out[160];
            memset(d, 0, sizeof(d));
            const int AD = ads[a], id = ids[i];
            for (int b = 0; b < 6; ++b) {
                d[b] = 1; /* tone signature in the top six bits of u0 */
            }
            for (int b = 0; b < 6; ++b) {
                d[6 + b] = (char)(((AD >> 1) >> (5 - b)) & 1); /* AD bits 6..1 */
            }
            for (int b = 0; b < 8; ++b) {
                d[12 + b] = (char)((id >> (7 - b)) & 1); /* ID1 = bits 11..4 of u1 */
            }
            d[35 + 9] = (char)(AD & 1); /* u3 bit 4 = AD bit 0 (u3 = d[35..48], bit 13 first) */
            mbe_synthesizeTonef(out, d, &cur);
            W(f, d, 49);
            W(f, out, sizeof(out));
            W(f, &cur.swn, 4);
            W(f, &cur.tonePhase, 4);
        }
    }
    static const int dids[10] = {5, 6, 7, 50, 50, 122, 123, 0, 4, 100};
    mbe_initMbeParms(&cur, &prev, &enh);
    n = 10;
    W(f, &n, 4);
    for (int i = 0; i < 10; ++i) {
        float out[160];
        int32_t id = dids[i];
        mbe_synthesizeTonefdstar(out, NULL, &cur, id);
        W(f, &id, 4);
        W(f, out, sizeof(out));
        W(f, &cur.swn, 4);
        W(f, &cur.tonePhase, 4);
    }
    fclose(f);
    printf("tone_kat.bin written\n");
}

/* ------------------------------------------------------------------------------------ */
/* notones_kat.bin: written by THIS program linked against the reference built with its NOTONES option
 * (-DDISABLE_AMBE_TONES, ref CMakeLists.txt:330-337; `make -C oracle notones`): tone frames synthesise silence and leave the
 * tone phases alone (ref src/core/mbelib.c:747-751, 815-819), everything else is the ordinary build.
 *   part A  the layout of tone_kat.bin (gen_tones): every output 160 zeros, swn / tonePhase unchanged
 *   part B  u32 S, u32 T, per stream T x { char d[49], i32 total_in, i32 ret, result(20), float pcm[160] }, then cur, prev,
 *           prev_enh: mbe_processAmbe2450Dataf (AMBE+2) on scripted parameter bits -- voice, tone frames (valid and invalid
 *           indices), with a given total error count; seeds 7000 + s
 *   part C  the same through mbe_processAmbe2400Dataf (D-STAR tone indices), seeds 8000 + s                                   */
static void
scripted_tone_streams(FILE* f, int dstar, uint32_t S, uint32_t T, uint32_t seed0) {
    static const int tone_list[12] = {5, 6, 7, 20, 64, 122, 123, 4, 130, 163, 164, 255};
    static const int totals[8] = {0, 0, 0, 0, 1, 2, 4, 5};
    static const int i_tone_hi[3] = {6, 7, 8};
    static const int def_of[8] = {1, 2, 3, 4, 0, 7, 6, 5};
    W(f, &S, 4);
    W(f, &T, 4);
    long tones = 0;
    for (uint32_t s = 0; s < S; ++s) {
        mbe_parms cur, prev, enh;
        mbe_initMbeParms(&cur, &prev, &enh);
        mbe_setThreadRngSeed(seed0 + s);
        for (uint32_t t = 0; t < T; ++t) {
            char d[49];
            float out[160];
            for (int c = 0; c < 49; ++c) {
                d[c] = (char)(splitmix64() & 1u);
            }
            if (splitmix64() % 10u < 6) { /* voice: the tone signature (bits 0..5 all set) must not appear */
                if (d[0] && d[1] && d[2] && d[3] && d[4] && d[5]) {
                    d[(int)(splitmix64() % 6u)] = 0;
                }
            } else { /* tone class with a chosen index */
                for (int c = 0; c < 6; ++c) {
                    d[c] = 1;
                }
                const int idx = tone_list[splitmix64() % 12u];
                if (dstar) { /* as gen_ambe2400 scripts them */
                    put_bits_msb(d, i_tone_hi, 3, def_of[(idx >> 5) & 7]);
                    d[9] = (char)((idx >> 4) & 1);
                    d[42] = (char)((idx >> 3) & 1);
                    d[43] = (char)((idx >> 2) & 1);
                    d[10] = (char)((idx >> 1) & 1);
                    d[11] = (char)(idx & 1);
                } else { /* AMBE+2: ID1 = bits 11..4 of u1 = d[12..19] (gen_tones) */
                    for (int b = 0; b < 8; ++b) {
                        d[12 + b] = (char)((idx >> (7 - b)) & 1);
                    }
                    if (splitmix64() % 4u) { /* verified tone (ref src/ambe/ambe3600x2450.c:475-490: low four bits of u3 zero); a
                                                quarter stay unverified: tone fundamental -> erasure (:536-541) */
                        d[45] = d[46] = d[47] = d[48] = 0;
                    }
                }
            }
            mbe_process_result r;
            mbe_initProcessResult(&r);
            int32_t total_in = totals[splitmix64() % 8u];
            r.total_errors = total_in;
            int32_t ret = dstar ? mbe_processAmbe2400Dataf(out, &r, d, &cur, &prev, &enh) : mbe_processAmbe2450Dataf(out, &r, d, &cur, &prev, &enh);
            tones += (r.flags & MBE_PROCESS_FLAG_TONE) != 0;
            W(f, d, 49);
            W(f, &total_in, 4);
            W(f, &ret, 4);
            W(f, &r, sizeof(r));
            W(f, out, sizeof(out));
        }
        W(f, &cur, sizeof(cur));
        W(f, &prev, sizeof(prev));
        W(f, &enh, sizeof(enh));
    }
    printf("  scripted %s streams: %u x %u frames, %ld flagged TONE\n", dstar ? "AMBE 3600x2400" : "AMBE+2 3600x2450", S, T, tones);
}

static void
gen_notones(const char* dir) {
    /* refuse to write the fixture from an ordinary build: a tone frame must come out silent */
    {
        mbe_parms cur, prev, enh;
        float out[160];
        mbe_initMbeParms(&cur, &prev, &enh);
        mbe_synthesizeTonefdstar(out, NULL, &cur, 50);
        for (int n = 0; n < 160; ++n) {
            if (out[n] != 0.0f) {
                fprintf(stderr, "gen_fixtures notones: this reference build synthesises tones (link the -DDISABLE_AMBE_TONES build)\n");
                exit(3);
            }
        }
    }
    gen_tones(dir); /* part A, into tone_kat.bin of `dir` ... */
    char a[1024], b[1024];
    snprintf(a, sizeof(a), "%s/tone_kat.bin", dir);
    snprintf(b, sizeof(b), "%s/notones_kat.bin", dir);
    FILE* in = fopen(a, "rb");
    FILE* f = fopen(b, "wb");
    if (!in || !f) {
        perror("notones_kat.bin");
        exit(2);
    }
    int ch;
    while ((ch = fgetc(in)) != EOF) { /* ... moved to the head of notones_kat.bin */
        fputc(ch, f);
    }
    fclose(in);
    remove(a);
    sm_state = 0x9E3779B97F4A7C15ULL ^ 0x707E5ULL;
    scripted_tone_streams(f, 0, 12, 24, 7000u);
    scripted_tone_streams(f, 1, 12, 24, 8000u);
    fclose(f);
    printf("notones_kat.bin written\n");
}

int
main(int argc, char** argv) {
    if (argc != 2 && argc != 3) {
        fprintf(stderr, "usage: %s outdir [soft|imbe7100|ambe2400|tones|notones]\n", argv[0]);
        return 2;
    }
    const char* dir = argv[1];
    if (argc == 3 && strcmp(argv[2], "soft") == 0) {   /* only the soft-decision file */
        gen_soft(dir);
        return 0;
    }
    if (argc == 3 && strcmp(argv[2], "imbe7100") == 0) {
        gen_imbe7100(dir);
        return 0;
    }
    if (argc == 3 && strcmp(argv[2], "ambe2400") == 0) {
        gen_ambe2400(dir);
        return 0;
    }
    if (argc == 3 && strcmp(argv[2], "tones") == 0) {
        gen_tones(dir);
        return 0;
    }
    if (argc == 3 && strcmp(argv[2], "notones") == 0) { /* only with the NOTONES build of the reference (oracle/Makefile `notones`) */
        gen_notones(dir);
        return 0;
    }
    gen_ecc(dir);
    gen_fec(dir, 0, 2048);
    gen_fec(dir, 1, 2048);
    gen_stream(dir, 0, 32, 24);
    gen_stream(dir, 1, 48, 24);
    gen_golden_synth(dir);
    gen_synth_seq(dir, 24);
    gen_f2s(dir);
    gen_params(dir);
    gen_misc(dir);
    gen_soft(dir);
    gen_imbe7100(dir);
    gen_ambe2400(dir);
    gen_tones(dir);
    return 0;
}
