/* host_bench.c -- what the HOST sees: end-to-end rates of the three ways a C program can drive the MI355X path, from
 * host memory to host memory (bench.py's `host_path`; built by mbelib-neo_amd/csrc/Makefile into mbelib-neo_amd/host_bench).
 *
 *   sync     the reference's per-frame call as it is (mbe_processImbe7200x4400Frame): one GPU round trip per frame
 *   queue    the same calls in queue mode (mbe_batchBegin / mbe_flush), state resident or written back every flush
 *   session  mbx_session_submit: whole batches of wire frames, state resident, pinned or pageable buffers
 *
 * usage: host_bench <tables.bin> <frames.bin: n x 18-byte IMBE 7200x4400 wire frames> [device]
 * prints one JSON object.  No CPU decode path here either: everything goes through the two libraries.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mbe_neo_amd.h"
#include "mbx.h"

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

    /* ---- queue mode: C channels, one frame per channel and tick ---- */
    const int C = S < 16384 ? S : 16384, ticks = 12;
    double queue_rate[2], queue_call_ns[2], queue_flush_ms[2];
    {
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
    double sess_rate[2], sess_rate_res = 0;
    {
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
    printf("{\"streams\": %d, \"sync_call_us\": %.2f, "
           "\"queue_channels\": %d, \"queue_resident_frames_per_s\": %.0f, \"queue_resident_call_ns\": %.1f, \"queue_resident_flush_ms\": %.3f, "
           "\"queue_writeback_frames_per_s\": %.0f, \"queue_writeback_call_ns\": %.1f, \"queue_writeback_flush_ms\": %.3f, "
           "\"session_pinned_frames_per_s\": %.0f, \"session_pinned_with_results_frames_per_s\": %.0f, \"session_pageable_frames_per_s\": %.0f}\n",
           S, sync_us, C, queue_rate[0], queue_call_ns[0], queue_flush_ms[0], queue_rate[1], queue_call_ns[1], queue_flush_ms[1],
           sess_rate[1], sess_rate_res, sess_rate[0]);
    return 0;
}
