/* host_bench.c -- what the HOST sees: end-to-end rates of the three ways a C program can drive the MI355X path, from
 * host memory to host memory (bench.py's `host_path`; built by mbelib-neo_amd/csrc/Makefile into mbelib-neo_amd/host_bench).
 *
 *   sync     the reference's per-frame call as it is (mbe_processImbe7200x4400Frame): one GPU round trip per frame
 *   queue    the same calls in queue mode (mbe_batchBegin / mbe_flush), state resident or written back every flush
 *   session  mbx_session_submit: whole batches of wire frames, state resident, pinned or pageable buffers
 *
 *   threads  queue mode and sessions again from N = 1, 4, 16 host threads at once (the reference is re-entrant per stream,
 *            ref include/mbelib-neo/mbelib.h:28-30; a real host runs one decoder thread per core): every thread owns its
 *            channels / its session, the rate is all frames over the wall time of the slowest thread
 *   devices  sessions on every visible device at once, one host thread per device (per-device contexts)
 *
 * usage: host_bench <tables.bin> <frames.bin: n x 18-byte IMBE 7200x4400 wire frames> [device]
 * prints one JSON object.  No CPU decode path here either: everything goes through the two libraries.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mbe_neo_amd.h"
#include "mbx.h"

extern int hipSetDevice(int);   /* libamdhip64, already a dependency of libmbx_hip.so; 0 = hipSuccess */
extern int hipGetDeviceCount(int*);
static int hipSetDevice_like(int d) { return hipSetDevice(d); }

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void unpack_cells(const uint8_t* packed, char cells[8][23]) {   /* inverse of mbx_pack_imbe7200x4400 */
    static const int width[8] = {23, 23, 23, 23, 15, 15, 15, 7};
    int pos = 0;
    memset(cells, 0, 184);
    for (int r = 0; r < 8; ++r) {
        for (int j = width[r] - 1; j >= 0; --j, ++pos) {
            cells[r][j] = (char)((packed[pos >> 3] >> (7 - (pos & 7))) & 1);
        }
    }
}


/* ---- N host threads ---------------------------------------------------------------------------------------------- */
struct job {
    int kind;              /* 0 queue mode (resident), 1 session (pinned buffers) */
    int device;
    int n;                 /* channels / streams of this thread */
    int ticks;
    const uint8_t* frames; /* n x 18 bytes */
    pthread_barrier_t* go;
    double seconds;        /* out: timed region of this thread */
    int rc;
};

static void* worker(void* arg) {
    struct job* j = (struct job*)arg;
    j->rc = 0;
    if (j->kind == 0) {
        const int C = j->n;
        mbe_parms* st = (mbe_parms*)malloc((size_t)C * 3 * sizeof(mbe_parms));
        short* pcm = (short*)malloc((size_t)C * 160 * sizeof(short));
        char(*cells)[8][23] = (char(*)[8][23])malloc((size_t)C * 184);
        char(*d)[88] = (char(*)[88])malloc((size_t)C * 88);
        for (int c = 0; c < C; ++c) {
            unpack_cells(j->frames + 18 * (size_t)c, cells[c]);
            mbe_initMbeParms(&st[3 * c], &st[3 * c + 1], &st[3 * c + 2]);
        }
        mbe_batchBegin(MBE_BATCH_STATE_RESIDENT);
        double t0 = 0;
        for (int k = 0; k < j->ticks + 2; ++k) {
            if (k == 2) {
                pthread_barrier_wait(j->go);
                t0 = now();
            }
            for (int c = 0; c < C; ++c) {
                mbe_processImbe7200x4400Frame(pcm + 160 * (size_t)c, NULL, (const char(*)[23])cells[c], d[c], &st[3 * c], &st[3 * c + 1],
                                              &st[3 * c + 2]);
            }
            if (mbe_flush() < 0) {
                j->rc = -1;
            }
        }
        j->seconds = now() - t0;
        mbe_batchEnd();
        free(st);
        free(pcm);
        free(cells);
        free(d);
    } else {
        const int S = j->n;
        mbx_session* h = NULL;
        if (hipSetDevice_like(j->device) != 0
            || mbx_session_create(&h, MBX_CODEC_IMBE7200X4400, S, (size_t)S, MBX_SESSION_PCM16) != 0) {
            j->rc = -1;
            pthread_barrier_wait(j->go);
            return NULL;
        }
        uint8_t* in = (uint8_t*)mbx_host_alloc((size_t)S * 18);
        int16_t* out[3];
        for (int i = 0; i < 3; ++i) {
            out[i] = (int16_t*)mbx_host_alloc((size_t)S * 320);
        }
        memcpy(in, j->frames, (size_t)S * 18);
        for (int i = 0; i < 3; ++i) {
            mbx_session_submit(h, 1, in, out[i], NULL, NULL);
        }
        mbx_session_wait(h);
        pthread_barrier_wait(j->go);
        const double t0 = now();
        for (int k = 0; k < j->ticks; ++k) {
            if (mbx_session_submit(h, 1, in, out[k % 3], NULL, NULL) < 0) {
                j->rc = -1;
            }
        }
        mbx_session_wait(h);
        j->seconds = now() - t0;
        mbx_session_destroy(h);
        mbx_host_free(in);
        for (int i = 0; i < 3; ++i) {
            mbx_host_free(out[i]);
        }
    }
    return NULL;
}

/* all frames of `nthreads` workers over the wall time of the slowest; device < 0: thread t on device t */
static double run_threads(int kind, int nthreads, int per_thread, int ticks, const uint8_t* frames, int S, int device) {
    pthread_t th[64];
    struct job jobs[64];
    pthread_barrier_t go;
    pthread_barrier_init(&go, NULL, (unsigned)nthreads);
    for (int t = 0; t < nthreads; ++t) {
        jobs[t].kind = kind;
        jobs[t].device = device < 0 ? t : device;
        jobs[t].n = per_thread;
        jobs[t].ticks = ticks;
        jobs[t].frames = frames + 18 * (size_t)(((size_t)t * (size_t)per_thread) % (size_t)(S - per_thread + 1));
        jobs[t].go = &go;
        jobs[t].seconds = 0;
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    double slowest = 0;
    int bad = 0;
    for (int t = 0; t < nthreads; ++t) {
        pthread_join(th[t], NULL);
        if (jobs[t].seconds > slowest) {
            slowest = jobs[t].seconds;
        }
        bad |= jobs[t].rc;
    }
    pthread_barrier_destroy(&go);
    return (bad || slowest <= 0) ? 0.0 : (double)nthreads * per_thread * ticks / slowest;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: host_bench tables.bin frames.bin [device]\n");
        return 2;
    }
    const int device = argc > 3 ? atoi(argv[3]) : 0;
    FILE* f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 2;
    }
    static unsigned char blob[sizeof(mbx_tables)];
    const size_t nb = fread(blob, 1, sizeof(blob), f);
    fclose(f);
    if (mbx_init(device, blob, nb) != 0) {
        fprintf(stderr, "mbx_init: %s\n", mbx_last_error());
        return 3;
    }
    f = fopen(argv[2], "rb");
    if (!f) {
        perror(argv[2]);
        return 2;
    }
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    const int S = (int)(bytes / 18);
    uint8_t* frames = (uint8_t*)malloc((size_t)S * 18);
    if (fread(frames, 18, (size_t)S, f) != (size_t)S) {
        return 2;
    }
    fclose(f);

    /* ---- sync: one channel, the per-frame call as the reference defines it ---- */
    double sync_us;
    {
        mbe_parms st[3];
        mbe_initMbeParms(&st[0], &st[1], &st[2]);
        char cells[8][23], d[88];
        short pcm[160];
        const int warm = 200, n = 2000;
        for (int i = 0; i < warm; ++i) {
            unpack_cells(frames + 18 * (size_t)(i % S), cells);
            mbe_processImbe7200x4400Frame(pcm, NULL, (const char(*)[23])cells, d, &st[0], &st[1], &st[2]);
        }
        const double t0 = now();
        for (int i = 0; i < n; ++i) {
            unpack_cells(frames + 18 * (size_t)(i % S), cells);
            mbe_processImbe7200x4400Frame(pcm, NULL, (const char(*)[23])cells, d, &st[0], &st[1], &st[2]);
        }
        sync_us = (now() - t0) / n * 1e6;
    }
    if (getenv("MBX_HOST_BENCH_SYNC_ONLY")) {   /* bench.py: the same figure again under another switch of the per-frame library */
        printf("{\"streams\": %d, \"sync_call_us\": %.2f}\n", S, sync_us);
        return 0;
    }

    /* ---- queue mode: C channels, one frame per channel and tick ---- */
    const int C = S < 16384 ? S : 16384, ticks = 12;
    double queue_rate[2] = {0, 0}, queue_call_ns[2] = {0, 0}, queue_flush_ms[2] = {0, 0};
    const int threads_only = getenv("HB_THREADS_ONLY") != NULL;   /* development: only the N-thread sections (with HB_THREADS) */
    if (!threads_only) {
        mbe_parms* st = (mbe_parms*)malloc((size_t)C * 3 * sizeof(mbe_parms));
        short* pcm = (short*)malloc((size_t)C * 160 * sizeof(short));
        char(*cells)[8][23] = (char(*)[8][23])malloc((size_t)C * 184);
        char(*d)[88] = (char(*)[88])malloc((size_t)C * 88);
        mbe_process_result* res = (mbe_process_result*)malloc((size_t)C * sizeof(mbe_process_result));
        for (int c = 0; c < C; ++c) {
            unpack_cells(frames + 18 * (size_t)c, cells[c]);
        }
        for (int mode = 0; mode < 2; ++mode) {
            for (int c = 0; c < C; ++c) {
                mbe_initMbeParms(&st[3 * c], &st[3 * c + 1], &st[3 * c + 2]);
            }
            mbe_batchBegin(mode == 0 ? MBE_BATCH_STATE_RESIDENT : MBE_BATCH_STATE_WRITEBACK);
            double t_calls = 0, t_flush = 0, t0 = 0;
            for (int k = 0; k < ticks + 2; ++k) {
                if (k == 2) {
                    t0 = now();
                    t_calls = t_flush = 0;
                }
                const double a = now();
                for (int c = 0; c < C; ++c) {
                    mbe_processImbe7200x4400Frame(pcm + 160 * (size_t)c, &res[c], (const char(*)[23])cells[c], d[c], &st[3 * c],
                                                  &st[3 * c + 1], &st[3 * c + 2]);
                }
                const double b = now();
                mbe_flush();
                const double e = now();
                t_calls += b - a;
                t_flush += e - b;
            }
            const double dt = now() - t0;
            mbe_batchEnd();
            queue_rate[mode] = (double)C * ticks / dt;
            queue_call_ns[mode] = t_calls / ((double)C * ticks) * 1e9;
            queue_flush_ms[mode] = t_flush / ticks * 1e3;
        }
        free(st);
        free(pcm);
        free(cells);
        free(d);
        free(res);
    }

    /* ---- sessions: S streams x T = 1 per submit ---- */
    double sess_rate[2] = {0, 0}, sess_rate_res = 0;
    if (!threads_only) {
        const int K = 40;
        for (int pinned = 1; pinned >= 0; --pinned) {
            mbx_session* h = NULL;
            if (mbx_session_create(&h, MBX_CODEC_IMBE7200X4400, S, (size_t)S, MBX_SESSION_PCM16 | MBX_SESSION_RESULTS) != 0) {
                fprintf(stderr, "mbx_session_create: %s\n", mbx_last_error());
                return 3;
            }
            /* three output buffers in rotation, like a host that consumes batch k-2 while k is in flight */
            uint8_t* in = pinned ? (uint8_t*)mbx_host_alloc((size_t)S * 18) : (uint8_t*)malloc((size_t)S * 18);
            int16_t* out[3];
            mbe_process_result* res[3];
            for (int i = 0; i < 3; ++i) {
                out[i] = pinned ? (int16_t*)mbx_host_alloc((size_t)S * 320) : (int16_t*)malloc((size_t)S * 320);
                res[i] = pinned ? (mbe_process_result*)mbx_host_alloc((size_t)S * 20) : (mbe_process_result*)malloc((size_t)S * 20);
            }
            memcpy(in, frames, (size_t)S * 18);
            for (int i = 0; i < 3; ++i) {
                mbx_session_submit(h, 1, in, out[i], NULL, NULL);
            }
            mbx_session_wait(h);
            double t0 = now();
            for (int k = 0; k < K; ++k) {
                mbx_session_submit(h, 1, in, out[k % 3], NULL, NULL);
            }
            mbx_session_wait(h);
            sess_rate[pinned] = (double)S * K / (now() - t0);
            if (pinned) {
                t0 = now();
                for (int k = 0; k < K; ++k) {
                    mbx_session_submit(h, 1, in, out[k % 3], NULL, res[k % 3]);
                }
                mbx_session_wait(h);
                sess_rate_res = (double)S * K / (now() - t0);
            }
            mbx_session_destroy(h);
            if (pinned) {
                mbx_host_free(in);
                for (int i = 0; i < 3; ++i) {
                    mbx_host_free(out[i]);
                    mbx_host_free(res[i]);
                }
            } else {
                free(in);
                for (int i = 0; i < 3; ++i) {
                    free(out[i]);
                    free(res[i]);
                }
            }
        }
    }
    /* ---- the same from N host threads (each with its own channels / session), and sessions on every device ---- */
    int tn[3] = {1, 4, 16};
    if (getenv("HB_THREADS")) {   /* development: other thread counts, e.g. HB_THREADS=2,8,12 */
        sscanf(getenv("HB_THREADS"), "%d,%d,%d", &tn[0], &tn[1], &tn[2]);
    }
    double tq[3], ts[3];
    for (int i = 0; i < 3; ++i) {
        const int per_q = (16384 / tn[i]) < S ? (16384 / tn[i]) : S, per_s = (S / tn[i]) > 0 ? (S / tn[i]) : 1;
        /* the same number of frames per THREAD whatever the thread count (12 / 40 ticks of the whole set from one thread): with a fixed
         * tick count sixteen threads were timed over 4 ms, and one late wake-up at the barrier decided the figure */
        tq[i] = run_threads(0, tn[i], per_q, 12 * tn[i], frames, S, device);
        ts[i] = run_threads(1, tn[i], per_s, 40 * tn[i], frames, S, device);
    }
    int ndev = 1;
    (void)hipGetDeviceCount(&ndev);
    double all_dev = 0;
    if (ndev > 1 && ndev <= 16) {
        int ok = 1;
        for (int dv = 0; dv < ndev; ++dv) {
            ok &= (mbx_init(dv, blob, nb) == 0);
        }
        hipSetDevice(device);
        all_dev = ok ? run_threads(1, ndev, S, 40, frames, S, -1) : 0.0;
    }
    printf("{\"streams\": %d, \"sync_call_us\": %.2f, "
           "\"threads\": [%d, %d, %d], \"queue_resident_frames_per_s_by_threads\": [%.0f, %.0f, %.0f], "
           "\"session_pinned_frames_per_s_by_threads\": [%.0f, %.0f, %.0f], \"devices\": %d, \"session_all_devices_frames_per_s\": %.0f, "
           "\"queue_channels\": %d, \"queue_resident_frames_per_s\": %.0f, \"queue_resident_call_ns\": %.1f, \"queue_resident_flush_ms\": %.3f, "
           "\"queue_writeback_frames_per_s\": %.0f, \"queue_writeback_call_ns\": %.1f, \"queue_writeback_flush_ms\": %.3f, "
           "\"session_pinned_frames_per_s\": %.0f, \"session_pinned_with_results_frames_per_s\": %.0f, \"session_pageable_frames_per_s\": %.0f}\n",
           S, sync_us, tn[0], tn[1], tn[2], tq[0], tq[1], tq[2], ts[0], ts[1], ts[2], ndev, all_dev, C, queue_rate[0], queue_call_ns[0], queue_flush_ms[0], queue_rate[1], queue_call_ns[1], queue_flush_ms[1],
           sess_rate[1], sess_rate_res, sess_rate[0]);
    return 0;
}
