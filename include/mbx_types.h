/*
 * mbx_types.h -- plain-C data types shared by the host shim, the C-ABI HIP launcher,
 * the CPU oracle and the tests.
 *
 * The per-stream model state and the per-frame status record are binary-compatible with
 * the reference's public structs so a caller can hand the very same memory to either
 * library:
 *   struct mbe_parameters / mbe_parms   <- reference include/mbelib-neo/mbelib.h:88-139
 *   mbe_process_result                  <- reference include/mbelib-neo/mbelib.h:180-191
 *   MBE_PROCESS_FLAG_* / MBE_STATUS_*   <- reference include/mbelib-neo/mbelib.h:154-171
 * Layout (x86-64 / amdgcn, 4-byte scalars, no padding): sizeof == 2604, offsets asserted below.
 */
#ifndef MBX_TYPES_H
#define MBX_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MBX_MAX_BANDS      56 /* harmonics are indexed 1..56, slot 0 is a scratch/edge slot */
#define MBX_BAND_SLOTS     57
#define MBX_FRAME_SAMPLES  160 /* 20 ms at 8 kHz */
#define MBX_FFT_N          256
#define MBX_NOISE_OVERLAP  96

struct mbe_parameters {
    float    w0;                       /* fundamental, rad/sample                         */
    int      L;                        /* number of harmonics                              */
    int      K;                        /* number of voicing bands (IMBE)                   */
    int      Vl[MBX_BAND_SLOTS];       /* 1 = voiced                                       */
    float    Ml[MBX_BAND_SLOTS];       /* spectral amplitude                               */
    float    log2Ml[MBX_BAND_SLOTS];   /* log2 amplitude (prediction memory)               */
    float    PHIl[MBX_BAND_SLOTS];     /* synthesis phase                                  */
    float    PSIl[MBX_BAND_SLOTS];     /* smoothed phase                                   */
    float    gamma;                    /* AMBE gain memory                                 */
    uint32_t tonePhase;                /* tone oscillator 2 phase                          */
    int      swn;                      /* tone oscillator 1 phase                          */
    float    localEnergy;              /* adaptive smoothing IIR                           */
    int      amplitudeThreshold;       /* adaptive smoothing Tm                            */
    float    errorRate;                /* error-rate IIR                                   */
    int      errorCountTotal;
    int      errorCount4;
    int      repeatCount;
    float    mutingThreshold;          /* 0.0875 marks IMBE state, 0.096 AMBE state        */
    float    previousUw[MBX_FFT_N];    /* last inverse-FFT block, for the overlap-add      */
    float    noiseSeed;                /* LCG state (<0: cold start)                       */
    float    noiseOverlap[MBX_NOISE_OVERLAP];
};
typedef struct mbe_parameters mbe_parms;

typedef struct mbe_process_result {
    int      c0_errors;
    int      protected_errors;
    int      c4_errors;
    int      total_errors;
    unsigned flags;
} mbe_process_result;

/* One soft-decision bit (ref include/mbelib-neo/mbelib.h:148-151): hard decision + confidence in it
 * (0 = unknown / erasure-like, 255 = highly reliable).  ABI-identical to the reference's struct. */
typedef struct mbe_soft_bit {
    uint8_t bit;
    uint8_t reliability;
} mbe_soft_bit;

#define MBE_PROCESS_FLAG_SOFT_INPUT 0x0001u
#define MBE_PROCESS_FLAG_C0_VALID   0x0002u
#define MBE_PROCESS_FLAG_C4_VALID   0x0004u
#define MBE_PROCESS_FLAG_TONE       0x0010u
#define MBE_PROCESS_FLAG_ERASURE    0x0020u
#define MBE_PROCESS_FLAG_REPEAT     0x0040u
#define MBE_PROCESS_FLAG_MUTE       0x0080u

#define MBE_STATUS_INVALID_ARGUMENT (-1)
#define MBE_STATUS_INVALID_BITS     (-2)

#define MBE_MAX_FRAME_REPEATS     4
#define MBE_MUTING_THRESHOLD_IMBE 0.0875f
#define MBE_MUTING_THRESHOLD_AMBE 0.096f

/* ---- batch-side additions (no counterpart in the reference) ------------------------ */

/* Codec selector for the batch entry points. */
enum {
    MBX_CODEC_IMBE7200X4400 = 0,
    MBX_CODEC_AMBE3600X2450 = 1,
    MBX_CODEC_IMBE7100X4400 = 2, /* own FEC/demodulation front end, then the 7200x4400 path */
    MBX_CODEC_AMBE3600X2400 = 3  /* D-STAR: the AMBE FEC front end of 3600x2450, its own parameter decode and frame policy */
};

/* Channel-frame wire size handed to the launcher: the 0/1 chars of the reference's
 * imbe_fr[8][23] / ambe_fr[4][24] with the meaningful cells packed MSB-first.
 *   IMBE: rows 0-3 cells 22..0, rows 4-6 cells 14..0, row 7 cells 6..0  = 144 bits
 *   AMBE: row 0 cells 23..0, row 1 cells 22..0, row 2 cells 10..0, row 3 cells 13..0 = 72 bits */
#define MBX_IMBE_FRAME_BYTES 18
#define MBX_AMBE_FRAME_BYTES 9
#define MBX_IMBE7100_FRAME_BYTES 18 /* 142 channel bits: rows of 19, 24, 23, 23, 15, 15, 23 cells, MSB-first */
/* soft-decision frames keep the reference's own array shapes: mbe_soft_bit[8][23] / [4][24] */
#define MBX_IMBE_SOFT_BITS 184
#define MBX_AMBE_SOFT_BITS 96
#define MBX_IMBE7100_SOFT_BITS 168 /* mbe_soft_bit[7][24] */

/* One decoded-parameter record per frame, produced by the FEC stage and consumed by the
 * stream stage: 88 (IMBE) or 49 (AMBE) parameter bits, bit i at word i/32, bit 31-(i%32);
 * w[3] = c0 | protected<<8 | c4<<16 | flags<<24. */
typedef struct mbx_param_record {
    uint32_t w[4];
} mbx_param_record;

/* The reference keeps three pieces of synthesis state in thread-local storage
 * (src/core/mbe_adaptive.c:29-30, src/core/mbe_unvoiced_fft.c:29-30).  A batch has no
 * "calling thread", so they become per-stream state. */
typedef struct mbx_stream_rng {
    uint64_t cn_seed48;             /* Java-LCG state of the comfort-noise generator      */
    uint32_t cn_seeded;             /* 0: seed lazily with the reference's default         */
    uint32_t unvoiced_seed_state;   /* next cold-start seed of the unvoiced LCG            */
    uint32_t unvoiced_seed_override;/* 1: a seed is pending for the next cold start        */
    uint32_t reserved;
} mbx_stream_rng;

#ifdef __cplusplus
}
#endif

#if defined(__cplusplus)
#define MBX_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define MBX_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif
MBX_STATIC_ASSERT(sizeof(mbe_parms) == 2604, "mbe_parms must stay ABI compatible");
MBX_STATIC_ASSERT(offsetof(mbe_parms, Vl) == 12, "Vl");
MBX_STATIC_ASSERT(offsetof(mbe_parms, Ml) == 240, "Ml");
MBX_STATIC_ASSERT(offsetof(mbe_parms, log2Ml) == 468, "log2Ml");
MBX_STATIC_ASSERT(offsetof(mbe_parms, PHIl) == 696, "PHIl");
MBX_STATIC_ASSERT(offsetof(mbe_parms, PSIl) == 924, "PSIl");
MBX_STATIC_ASSERT(offsetof(mbe_parms, gamma) == 1152, "gamma");
MBX_STATIC_ASSERT(offsetof(mbe_parms, localEnergy) == 1164, "localEnergy");
MBX_STATIC_ASSERT(offsetof(mbe_parms, mutingThreshold) == 1188, "mutingThreshold");
MBX_STATIC_ASSERT(offsetof(mbe_parms, previousUw) == 1192, "previousUw");
MBX_STATIC_ASSERT(offsetof(mbe_parms, noiseSeed) == 2216, "noiseSeed");
MBX_STATIC_ASSERT(offsetof(mbe_parms, noiseOverlap) == 2220, "noiseOverlap");
MBX_STATIC_ASSERT(sizeof(mbe_process_result) == 20, "mbe_process_result");
MBX_STATIC_ASSERT(sizeof(mbx_stream_rng) == 24, "mbx_stream_rng");
MBX_STATIC_ASSERT(sizeof(mbx_param_record) == 16, "mbx_param_record");

/* request mailbox of a frame server (include/mbx.h, mbx_frame_server_start): pinned, coherent host memory */
typedef struct mbx_frame_mailbox {
    /* the REQUEST: one 64-byte line, written by the host -- fields first, seq_in last -- and read by the server with ONE load,
     * so a snapshot that shows the new seq_in shows the whole request */
    uint32_t seq_in, quit;
    int32_t  codec;
    uint32_t want;                                   /* bit 0: int16 PCM, bit 1: float PCM, bit 2 (MBX_FRAME_WANT_SHADOW): state from the device copy */
    uint8_t  frame[24];                              /* the wire frame (18 | 9 bytes) */
    uint32_t pad0[6];
    uint32_t seq_out, alive, served, pad1[13];       /* written by the server (its own line); served: requests since its start */
} mbx_frame_mailbox;
MBX_STATIC_ASSERT(offsetof(mbx_frame_mailbox, seq_out) == 64 && offsetof(mbx_frame_mailbox, frame) == 16 && sizeof(mbx_frame_mailbox) == 128, "mbx_frame_mailbox");

#define MBX_FRAME_WANT_PCM16  1u
#define MBX_FRAME_WANT_PCMF   2u
#define MBX_FRAME_WANT_SHADOW 4u

#endif /* MBX_TYPES_H */
