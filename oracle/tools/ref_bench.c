/*
 * ref_bench.c -- TEST INFRASTRUCTURE: a batch driver around the REAL reference (oracle/_ref/libmbe_ref.so), so that
 * bench.py's cpu_baseline leg can time the reference itself on the host cores of the GPU box without a Python call
 * per frame.  Built by `make -C oracle ref` into oracle/_ref/libref_bench.so (git-ignored, travels with the snapshot).
 * Streams run one after the other on the calling thread; the reference keeps its RNG per thread, so every stream is
 * re-seeded with mbe_setThreadRngSeed(seed0 + s) exactly as the fixtures do -- state parity is not the point here,
 * only the time per frame of the reference's own code path.
 */
#include <stddef.h>
#include <stdint.h>

#include "mbx_types.h" /* ABI-identical structs; reference prototypes restated below */

/* include/mbelib-neo/mbelib.h:433, 509, 568, 359, 596, 615 */
extern int mbe_processImbe7200x4400Frame(short*, mbe_process_result*, const char[8][23], char[88], mbe_parms*, mbe_parms*, mbe_parms*);
extern int mbe_processAmbe3600x2450Frame(short*, mbe_process_result*, const char[4][24], char[49], mbe_parms*, mbe_parms*, mbe_parms*);
extern int mbe_processImbe7100x4400Frame(short*, mbe_process_result*, const char[7][24], char[88], mbe_parms*, mbe_parms*, mbe_parms*);
extern int mbe_processAmbe3600x2400Frame(short*, mbe_process_result*, const char[4][24], char[49], mbe_parms*, mbe_parms*, mbe_parms*);
extern void mbe_setThreadRngSeed(uint32_t);

/* cells: S*T frames of `ncell` chars, stream-major; state: S*3 mbe_parms (in/out); pcm16: S*T*160 */
int
ref_process_batch(int codec, int S, int T, const char* cells, int ncell, mbe_parms* state, uint32_t seed0, int16_t* pcm16) {
    char bits[88];
    for (int s = 0; s < S; ++s) {
        mbe_parms* cur = &state[3 * (size_t)s];
        mbe_setThreadRngSeed(seed0 + (uint32_t)s);
        for (int t = 0; t < T; ++t) {
            const size_t f = (size_t)s * (size_t)T + (size_t)t;
            const char* fr = cells + f * (size_t)ncell;
            short* out = (short*)(pcm16 + f * 160);
            int rc;
            switch (codec) {
                case MBX_CODEC_IMBE7200X4400: rc = mbe_processImbe7200x4400Frame(out, NULL, (const char(*)[23])fr, bits, cur, cur + 1, cur + 2); break;
                case MBX_CODEC_AMBE3600X2450: rc = mbe_processAmbe3600x2450Frame(out, NULL, (const char(*)[24])fr, bits, cur, cur + 1, cur + 2); break;
                case MBX_CODEC_IMBE7100X4400: rc = mbe_processImbe7100x4400Frame(out, NULL, (const char(*)[24])fr, bits, cur, cur + 1, cur + 2); break;
                default: rc = mbe_processAmbe3600x2400Frame(out, NULL, (const char(*)[24])fr, bits, cur, cur + 1, cur + 2); break;
            }
            if (rc < 0) {
                return rc;
            }
        }
    }
    return 0;
}

/* ---- all host cores: the streams split over `nthreads` POSIX threads (the reference is re-entrant per stream with
 *      thread-local helper state, include/mbelib-neo/mbelib.h:28-30) ---------------------------------------------- */
#include <pthread.h>
#include <time.h>

struct mt_job {
    int codec, S, T, ncell, rc;
    const char* cells;
    mbe_parms* state;
    uint32_t seed0;
    int16_t* pcm16;
};

static void*
mt_worker(void* arg) {
    struct mt_job* j = (struct mt_job*)arg;
    j->rc = ref_process_batch(j->codec, j->S, j->T, j->cells, j->ncell, j->state, j->seed0, j->pcm16);
    return NULL;
}

int
ref_process_batch_mt(int codec, int S, int T, const char* cells, int ncell, mbe_parms* state, uint32_t seed0, int16_t* pcm16,
                     int nthreads) {
    if (nthreads < 1) {
        nthreads = 1;
    }
    if (nthreads > 256) {
        nthreads = 256;
    }
    pthread_t th[256];
    struct mt_job job[256];
    int first = 0;
    for (int k = 0; k < nthreads; ++k) {
        const int count = S / nthreads + (k < S % nthreads ? 1 : 0);
        job[k] = (struct mt_job){codec, count, T, ncell, 0, cells + (size_t)first * (size_t)T * (size_t)ncell, state + 3 * (size_t)first,
                                 seed0 + (uint32_t)first, pcm16 + (size_t)first * (size_t)T * 160};
        first += count;
        if (pthread_create(&th[k], NULL, mt_worker, &job[k]) != 0) {
            return -1;
        }
    }
    int rc = 0;
    for (int k = 0; k < nthreads; ++k) {
        pthread_join(th[k], NULL);
        if (job[k].rc < 0) {
            rc = job[k].rc;
        }
    }
    return rc;
}

/* ---- the reference's own micro-benchmark recipes, as functions (drivers of the reference API written here; the
 *      workloads are those of bench/bench_synth.c:40-67 and bench/bench_unvoiced.c:33-52,87-100) --------------------- */
extern void mbe_initMbeParms(mbe_parms*, mbe_parms*, mbe_parms*);
extern void mbe_moveMbeParms(const mbe_parms*, mbe_parms*);
extern void mbe_synthesizeSpeechf(float*, mbe_parms*, mbe_parms*);
extern void mbe_floattoshort(const float*, short*);

static double
wall(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* recipe 0 = bench_synth (L = 40, w0 alternating 0.09 / 0.11, mixed voicing), 1 = bench_unvoiced (L = 36, all unvoiced):
 * returns seconds for `frames` calls of mbe_synthesizeSpeechf (best of `runs`), checksum of the output in *sink;
 * recipe 2 = bench_convert (bench/bench_convert.c:33-50: one 160-sample ramp converted `frames` times by mbe_floattoshort) */
double
ref_bench_recipe(int recipe, int frames, int runs, float* sink) {
    float out[160];
    mbe_parms cur, prev, enh;
    double best = 1e30;
    float acc = 0.0f;
    if (recipe == 2) {
        short s16[160];
        for (int r = 0; r < runs; ++r) {
            for (int i = 0; i < 160; ++i) {
                out[i] = (float)i * 0.01f - 0.8f;
            }
            const double t0 = wall();
            for (int i = 0; i < frames; ++i) {
                mbe_floattoshort(out, s16);
                out[0] += (float)(s16[0] & 1) * 1e-6f;   /* keeps the call alive, as the reference's bench does */
            }
            const double dt = wall() - t0;
            if (dt < best) {
                best = dt;
            }
            acc += (float)s16[0];
        }
        if (sink) {
            *sink = acc;
        }
        return best;
    }
    for (int r = 0; r < runs; ++r) {
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
        const double t0 = wall();
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
            mbe_moveMbeParms(&cur, &prev);
            acc += out[i % 160];
        }
        const double dt = wall() - t0;
        if (dt < best) {
            best = dt;
        }
    }
    if (sink) {
        *sink = acc;
    }
    return best;
}
