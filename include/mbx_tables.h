/*
 * mbx_tables.h -- the constant-table blob ("codebook tables") shared by every rank.
 *
 * One flat little-endian struct.  It is DATA: produced in the authoring container by
 * oracle/tools/dump_tables.c, which reads the reference's constant tables where they lie
 * (src/internal/imbe7200x4400_const.h, src/internal/ambe3600x2450_const.h,
 * src/internal/mbelib_const.h, src/ecc/ecc_const.c, mbe_synthesisWindow() of
 * src/core/mbe_unvoiced_fft.c:202-208) and evaluates the reference's table-building
 * formulas with the host libm (DCT cosine caches: src/imbe/imbe7200x4400.c:91-115,
 * src/ambe/ambe3600x2450.c:54-78; b0 -> w0/L/K: src/imbe/imbe7200x4400.c:117-154;
 * WOLA weights: src/core/mbe_unvoiced_fft.c:159-175).  Integer tables are narrowed to
 * u8/u16 where the value range allows.  The committed file is
 * mbelib-neo_amd/data/mbx_tables.bin; rank 0 reads it and broadcasts it (RCCL) to the
 * other ranks, every rank uploads it with mbx_init() and verifies `checksum`.
 */
#ifndef MBX_TABLES_H
#define MBX_TABLES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MBX_TABLES_MAGIC   0x3154584Du /* "MXT1" */
#define MBX_TABLES_VERSION 4u

typedef struct mbx_tables {
    uint32_t magic;
    uint32_t version;
    uint32_t total_bytes; /* sizeof(mbx_tables) */
    uint32_t checksum;    /* FNV-1a-32 over every byte after this field */

    /* ---- FEC (a4, a5) ---- */
    uint16_t golay_matrix[2048];   /* syndrome -> 12-bit data correction mask            */
    uint16_t golay_gen[12];        /* parity contribution of data bit 11-i                */
    uint16_t hamming_gen[4];       /* parity-check row masks over the 15-bit block        */
    uint16_t hamming_fix[16];      /* syndrome -> single-bit flip mask                    */

    /* ---- IMBE 7200x4400 parameter decode (a8) ---- */
    float    imbe_w0[208];         /* b0 -> w0                                            */
    uint8_t  imbe_L[208];          /* b0 -> L (0 marks an invalid L)                      */
    uint8_t  imbe_K[208];          /* b0 -> K                                             */
    uint8_t  imbe_bo[48][79][2];   /* [L-9][i-6] -> (parameter index, bit position)       */
    uint8_t  imbe_hoba[48][50];    /* [L-9][m-8] -> bit count of higher-order coefficient */
    uint8_t  imbe_ji[48][6];       /* [L-9][block] -> block length                        */
    float    imbe_ba[48][5][2];    /* [L-9][g-2] -> (bit count, step size) of gain g      */
    float    imbe_B2[64];          /* gain codebook                                       */
    float    imbe_quantstep[11];
    float    imbe_standdev[9];
    float    imbe_ri_cos[7][7];
    float    imbe_idct_cos[11][11][11];

    /* ---- AMBE+2 3600x2450 parameter decode (a9) ---- */
    float    ambe_w0[120];         /* b0 -> f0 (cycles/sample)                            */
    uint8_t  ambe_L[120];
    uint8_t  ambe_vuv[32][8];
    uint8_t  ambe_lmprbl[57][4];
    float    ambe_dg[32];
    float    ambe_prba24[512][3];
    float    ambe_prba58[128][4];
    float    ambe_hoc_b5[32][4];
    float    ambe_hoc_b6[16][4];
    float    ambe_hoc_b7[16][4];
    float    ambe_hoc_b8[8][4];
    float    ambe_ri_cos[9][9];
    float    ambe_idct_cos[18][18][18];

    /* ---- synthesis (a17, a18) ---- */
    float    ws[321];              /* voiced synthesis window                              */
    float    uv_window[256];       /* 211-tap unvoiced synthesis window centred at 128     */
    float    wola_w_prev[160];
    float    wola_w_curr[160];
    float    wola_denom[160];

    /* ---- IMBE 7100x4400 (§8(f) row 4): its own Hamming(15,11) bit mapping, ref src/ecc/ecc.c:422-464 ---- */
    uint16_t hamming7100_gen[4];   /* parity-check row masks                              */
    uint16_t hamming7100_fix[16];  /* syndrome -> single-bit flip mask                    */

    /* ---- AMBE 3600x2400 parameter decode (§8(f) row 4), ref src/internal/ambe3600x2400_const.h.
     *      The 8-point / per-block DCT cosines are those of AMBE+2 (ambe_ri_cos, ambe_idct_cos). ---- */
    float    ambep_dg[64];
    float    ambep_prba24[512][3];
    float    ambep_prba58[128][4];
    float    ambep_hoc_b5[16][4];
    float    ambep_hoc_b6[16][4];
    float    ambep_hoc_b7[16][4];
    float    ambep_hoc_b8[16][4];
    uint8_t  ambep_L[128];         /* b0 -> L (b0 <= 125)                                   */
    uint8_t  ambep_vuv[16][8];
    uint8_t  ambep_lmprbl[57][4];
    uint8_t  pad_[4];
} mbx_tables;

#ifdef __cplusplus
}
#endif
#endif /* MBX_TABLES_H */
