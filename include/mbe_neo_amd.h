/*
 * mbe_neo_amd.h -- the per-frame `mbe_*` entry points exported by libmbe_neo_amd.so.
 *
 * Same names, argument meaning, return codes and in/out behaviour as the reference's public
 * header (arancormonk/mbelib-neo v2.0.0, include/mbelib-neo/mbelib.h; the line of each
 * reference declaration is given).  A program written against that header keeps calling the
 * same functions; behind them every frame goes through the MI355X launcher (include/mbx.h).
 * Only the hot path of SURVEY.md §8 is exported -- soft-decision decoding, the D-STAR and
 * ProVoice codecs and the debug dump helpers are not part of this library.
 *
 * There is no CPU compute path: the first call initialises the GPU (device $MBX_DEVICE or 0,
 * tables from $MBX_TABLES or <library dir>/data/mbx_tables.bin) and the process aborts with a
 * message on stderr if that fails.
 *
 * Throughput note: one call = one 20 ms frame = a few tiny transfers and launches (tens of
 * microseconds).  Hosts that decode many streams should hand whole batches to
 * mbx_process_batch() (include/mbx.h) -- see INTEGRATION.md.
 */
#ifndef MBE_NEO_AMD_H
#define MBE_NEO_AMD_H

#include <stddef.h>

#include "mbx_types.h"

#ifdef __cplusplus
extern "C" {
#endif

void mbe_initProcessResult(mbe_process_result* result);                                   /* mbelib.h:194 */
void mbe_formatProcessResult(char* str, size_t size, const mbe_process_result* result);    /* :202 */
int mbe_checkGolayBlock(long int* block);                                                 /* :231 */
int mbe_golay2312(const char* in, char* out);                                             /* :238 */
int mbe_hamming1511(const char* in, char* out);                                           /* :253 */

int mbe_decodeAmbe3600x2450Frame(const char ambe_fr[4][24], char ambe_d[49], mbe_process_result* result); /* :395 */
int mbe_processAmbe2450Dataf(float* aout_buf, mbe_process_result* result, const char ambe_d[49], mbe_parms* cur_mp,
                             mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced);            /* :415 */
int mbe_processAmbe2450Data(short* aout_buf, mbe_process_result* result, const char ambe_d[49], mbe_parms* cur_mp,
                            mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced);             /* :418 */
int mbe_processAmbe3600x2450Framef(float* aout_buf, mbe_process_result* result, const char ambe_fr[4][24],
                                   char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp,
                                   mbe_parms* prev_mp_enhanced);                          /* :429 */
int mbe_processAmbe3600x2450Frame(short* aout_buf, mbe_process_result* result, const char ambe_fr[4][24],
                                  char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp,
                                  mbe_parms* prev_mp_enhanced);                           /* :433 */

int mbe_decodeImbe7200x4400Frame(const char imbe_fr[8][23], char imbe_d[88], mbe_process_result* result); /* :471 */
int mbe_processImbe4400Dataf(float* aout_buf, mbe_process_result* result, const char imbe_d[88], mbe_parms* cur_mp,
                             mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced);            /* :491 */
int mbe_processImbe4400Data(short* aout_buf, mbe_process_result* result, const char imbe_d[88], mbe_parms* cur_mp,
                            mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced);             /* :494 */
int mbe_processImbe7200x4400Framef(float* aout_buf, mbe_process_result* result, const char imbe_fr[8][23],
                                   char imbe_d[88], mbe_parms* cur_mp, mbe_parms* prev_mp,
                                   mbe_parms* prev_mp_enhanced);                          /* :505 */
int mbe_processImbe7200x4400Frame(short* aout_buf, mbe_process_result* result, const char imbe_fr[8][23],
                                  char imbe_d[88], mbe_parms* cur_mp, mbe_parms* prev_mp,
                                  mbe_parms* prev_mp_enhanced);                           /* :509 */

/* AMBE 3600x2400 / D-STAR */
int mbe_decodeAmbe3600x2400Frame(const char ambe_fr[4][24], char ambe_d[49], mbe_process_result* result);   /* :315 */
int mbe_decodeAmbe3600x2400SoftFrame(const mbe_soft_bit ambe_fr[4][24], char ambe_d[49], mbe_process_result* result); /* :323 */
int mbe_processAmbe2400Dataf(float* aout_buf, mbe_process_result* result, const char ambe_d[49], mbe_parms* cur_mp,
                             mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced);            /* :335 */
int mbe_processAmbe2400Data(short* aout_buf, mbe_process_result* result, const char ambe_d[49], mbe_parms* cur_mp,
                            mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced);             /* :341 */
int mbe_processAmbe3600x2400Framef(float* aout_buf, mbe_process_result* result, const char ambe_fr[4][24],
                                   char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp,
                                   mbe_parms* prev_mp_enhanced);                          /* :352 */
int mbe_processAmbe3600x2400Frame(short* aout_buf, mbe_process_result* result, const char ambe_fr[4][24],
                                  char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp,
                                  mbe_parms* prev_mp_enhanced);                           /* :359 */
int mbe_processAmbe3600x2400SoftFramef(float* aout_buf, mbe_process_result* result, const mbe_soft_bit ambe_fr[4][24],
                                       char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced); /* :367 */
int mbe_processAmbe3600x2400SoftFrame(short* aout_buf, mbe_process_result* result, const mbe_soft_bit ambe_fr[4][24],
                                      char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced); /* :371 */

/* IMBE 7100x4400 */
int mbe_7100x4400hamming1511(const char* in, char* out);                                  /* :267 */
int mbe_decodeImbe7100x4400Frame(const char imbe_fr[7][24], char imbe_d[88], mbe_process_result* result); /* :545 */
int mbe_processImbe7100x4400Framef(float* aout_buf, mbe_process_result* result, const char imbe_fr[7][24],
                                   char imbe_d[88], mbe_parms* cur_mp, mbe_parms* prev_mp,
                                   mbe_parms* prev_mp_enhanced);                          /* :564 */
int mbe_processImbe7100x4400Frame(short* aout_buf, mbe_process_result* result, const char imbe_fr[7][24],
                                  char imbe_d[88], mbe_parms* cur_mp, mbe_parms* prev_mp,
                                  mbe_parms* prev_mp_enhanced);                           /* :568 */

int mbe_7100x4400hamming1511Soft(const mbe_soft_bit* in, char* out);                      /* :274 */
int mbe_decodeImbe7100x4400SoftFrame(const mbe_soft_bit imbe_fr[7][24], char imbe_d[88], mbe_process_result* result); /* :553 */
int mbe_processImbe7100x4400SoftFramef(float* aout_buf, mbe_process_result* result, const mbe_soft_bit imbe_fr[7][24],
                                       char imbe_d[88], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced); /* :576 */
int mbe_processImbe7100x4400SoftFrame(short* aout_buf, mbe_process_result* result, const mbe_soft_bit imbe_fr[7][24],
                                      char imbe_d[88], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced); /* :580 */

/* soft-decision entry points */
mbe_soft_bit mbe_softBitFromHard(int bit, uint8_t reliability);                           /* :208 */
mbe_soft_bit mbe_softBitFromLlr(int16_t llr);                                             /* :214 */
int mbe_softBitsFromHard(const char* bits, mbe_soft_bit* soft, size_t count, uint8_t reliability); /* :219 */
int mbe_softBitsFromLlr(const int16_t* llr, mbe_soft_bit* soft, size_t count);            /* :224 */
int mbe_golay2312Soft(const mbe_soft_bit* in, char* out);                                 /* :246 */
int mbe_hamming1511Soft(const mbe_soft_bit* in, char* out);                               /* :260 */
int mbe_decodeAmbe3600x2450SoftFrame(const mbe_soft_bit ambe_fr[4][24], char ambe_d[49], mbe_process_result* result); /* :403 */
int mbe_processAmbe3600x2450SoftFramef(float* aout_buf, mbe_process_result* result, const mbe_soft_bit ambe_fr[4][24],
                                       char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced); /* :441 */
int mbe_processAmbe3600x2450SoftFrame(short* aout_buf, mbe_process_result* result, const mbe_soft_bit ambe_fr[4][24],
                                      char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced); /* :445 */
int mbe_decodeImbe7200x4400SoftFrame(const mbe_soft_bit imbe_fr[8][23], char imbe_d[88], mbe_process_result* result); /* :479 */
int mbe_processImbe7200x4400SoftFramef(float* aout_buf, mbe_process_result* result, const mbe_soft_bit imbe_fr[8][23],
                                       char imbe_d[88], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced); /* :517 */
int mbe_processImbe7200x4400SoftFrame(short* aout_buf, mbe_process_result* result, const mbe_soft_bit imbe_fr[8][23],
                                      char imbe_d[88], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced); /* :521 */

const char* mbe_versionString(void);                                                      /* :588 */
void mbe_setThreadRngSeed(uint32_t seed);                                                 /* :596 */
void mbe_moveMbeParms(const mbe_parms* source_mp, mbe_parms* destination_mp);             /* :602 */
void mbe_useLastMbeParms(mbe_parms* cur_mp, const mbe_parms* prev_mp);                    /* :608 */
void mbe_initMbeParms(mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced);/* :615 */
void mbe_spectralAmpEnhance(mbe_parms* cur_mp);                                           /* :623 */
void mbe_synthesizeTonef(float* aout_buf, const char* ambe_d, mbe_parms* cur_mp);          /* :630 */
void mbe_synthesizeTonefdstar(float* aout_buf, const char* ambe_d, mbe_parms* cur_mp, int ID1); /* :638 */
void mbe_synthesizeSilencef(float* aout_buf);                                             /* :640 */
void mbe_synthesizeSilence(short* aout_buf);                                              /* :642 */
void mbe_synthesizeSpeechf(float* aout_buf, mbe_parms* cur_mp, mbe_parms* prev_mp);       /* :652 */
void mbe_synthesizeSpeech(short* aout_buf, mbe_parms* cur_mp, mbe_parms* prev_mp);        /* :662 */
void mbe_floattoshort(const float* float_buf, short* aout_buf);                           /* :675 */
int mbe_requiresMuting(const mbe_parms* mp);                                              /* :693 */
int mbe_isMaxFrameRepeat(const mbe_parms* mp);                                            /* :700 */
void mbe_synthesizeComfortNoisef(float* aout_buf);                                        /* :706 */
void mbe_synthesizeComfortNoise(short* aout_buf);                                         /* :712 */
void mbe_applyAdaptiveSmoothing(mbe_parms* cur_mp, const mbe_parms* prev_mp);             /* :725 */
int mbe_requiresAdaptiveSmoothing(const mbe_parms* mp);                                   /* :732 */

#ifdef __cplusplus
}
#endif
#endif /* MBE_NEO_AMD_H */
