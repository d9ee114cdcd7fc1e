/*
 * mbe_neo_amd.h -- the per-frame `mbe_*` entry points exported by libmbe_neo_amd.so.
 *
 * Same names, argument meaning, return codes and in/out behaviour as the reference's public
 * header (arancormonk/mbelib-neo v2.0.0, include/mbelib-neo/mbelib.h; the line of each
 * reference declaration is given).  A program written against that header keeps calling the
 * same functions; behind them every frame goes through the MI355X launcher (include/mbx.h).
 * All 87 functions of the reference header are exported -- the four codecs (IMBE 7200x4400, IMBE 7100x4400,
 * AMBE+2 3600x2450, AMBE 3600x2400 / D-STAR), hard- and soft-decision, the stage-by-stage helpers of the classic
 * ecc -> demodulate -> ecc -> process*Data sequence, tones, and the stderr dump helpers -- so the library can stand
 * in for libmbe-neo.so.2 at link or load time.
 *
 * There is no CPU compute path: the first call initialises the GPU (device $MBX_DEVICE or 0,
 * tables from $MBX_TABLES or <library dir>/data/mbx_tables.bin) and the process aborts with a
 * message on stderr if that fails.  Like the reference (mbelib.h:28-30) the processing functions are re-entrant per
 * stream: each host thread has its own HIP stream, device scratch and RNG state.
 *
 * Throughput.  A synchronous call is one 20 ms frame = one launch of one wavefront and one device round trip: 16.97 us per call
 * in the driver's round-4 run (16.6-19.8 us by box), against 17.44 us of CPU time for the reference's own call on the same host.
 * A host that decodes many channels has two better options, both measured in bench.py's `host_path`:
 *   * queue mode (below): keep calling the per-frame functions, mbe_flush() runs everything queued as batched launches;
 *   * sessions (include/mbx.h): hand whole batches of wire frames over, state stays on the device.
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

/* stage-by-stage helpers of the classic call sequence and the parameter decode alone */
int mbe_eccImbe7200x4400C0(char imbe_fr[8][23]);                                          /* :457 */
int mbe_eccImbe7200x4400Data(char imbe_fr[8][23], char* imbe_d);                          /* :459 */
int mbe_decodeImbe4400Parms(const char* imbe_d, mbe_parms* cur_mp, mbe_parms* prev_mp);   /* :461 */
int mbe_demodulateImbe7200x4400Data(char imbe[8][23]);                                    /* :463 */
int mbe_eccAmbe3600x2450C0(char ambe_fr[4][24]);                                          /* :381 */
int mbe_eccAmbe3600x2450Data(char ambe_fr[4][24], char* ambe_d);                          /* :383 */
int mbe_decodeAmbe2450Parms(const char* ambe_d, mbe_parms* cur_mp, mbe_parms* prev_mp);   /* :385 */
int mbe_demodulateAmbe3600x2450Data(char ambe_fr[4][24]);                                 /* :387 */
int mbe_eccAmbe3600x2400C0(char ambe_fr[4][24]);                                          /* :286 */
int mbe_eccAmbe3600x2400Data(char ambe_fr[4][24], char* ambe_d);                          /* :293 */
int mbe_decodeAmbe2400Parms(const char* ambe_d, mbe_parms* cur_mp, mbe_parms* prev_mp);   /* :301 */
int mbe_demodulateAmbe3600x2400Data(char ambe_fr[4][24]);                                 /* :307 */
int mbe_eccImbe7100x4400C0(char imbe_fr[7][24]);                                          /* :531 */
int mbe_eccImbe7100x4400Data(char imbe_fr[7][24], char* imbe_d);                          /* :533 */
int mbe_demodulateImbe7100x4400Data(char imbe[7][24]);                                    /* :535 */
int mbe_convertImbe7100to7200(char* imbe_d);                                              /* :537 */

/* stderr dump helpers (host text, the reference's formats) */
void mbe_dumpAmbe2400Data(const char* ambe_d);                                            /* :278 */
void mbe_dumpAmbe3600x2400Frame(const char ambe_fr[4][24]);                               /* :280 */
void mbe_dumpAmbe2450Data(const char* ambe_d);                                            /* :377 */
void mbe_dumpAmbe3600x2450Frame(const char ambe_fr[4][24]);                               /* :379 */
void mbe_dumpImbe4400Data(const char* imbe_d);                                            /* :451 */
void mbe_dumpImbe7200x4400Data(const char* imbe_d);                                       /* :453 */
void mbe_dumpImbe7200x4400Frame(const char imbe_fr[8][23]);                               /* :455 */
void mbe_dumpImbe7100x4400Data(const char* imbe_d);                                       /* :527 */
void mbe_dumpImbe7100x4400Frame(const char imbe_fr[7][24]);                               /* :529 */

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

/* ---- queue mode: the per-frame API fanning frames into batched launches (not in the reference) ------------------
 * The reference's per-frame calls are synchronous by contract (PCM valid on return), which on a GPU means one round
 * trip per 20 ms frame.  A host that serves many channels can keep its per-frame code and defer the work instead:
 *
 *     mbe_batchBegin(MBE_BATCH_STATE_RESIDENT);
 *     every 20 ms:  for each channel with a frame:  mbe_processImbe7200x4400Frame(pcm[ch], &res[ch], fr[ch], d[ch],
 *                                                                                &cur[ch], &prev[ch], &enh[ch]);   // queued, returns 0
 *                   mbe_flush();        // all queued frames run as a few batched launches; pcm / res / d are filled now
 *     mbe_batchEnd();                   // flush, bring the model state back to the host structs, leave queue mode
 *
 * Queued are the hard-decision mbe_process*Frame / mbe_process*Framef calls of the four codecs (every other function
 * keeps running synchronously).  A queued call validates its arguments and the frame bits exactly like the synchronous
 * one (negative return, nothing queued, nothing written), then returns 0; its outputs -- aout_buf, *result (incl. the
 * total error count the synchronous call would have returned), imbe_d / ambe_d and the three mbe_parms -- are written by
 * mbe_flush(), so those pointers must stay valid until then.  A channel is identified by its cur_mp pointer; frames
 * queued for one channel are decoded in call order.  The reference's thread-local RNG state becomes per channel: a
 * channel takes a copy of the calling thread's state (mbe_setThreadRngSeed) at its first queued frame.
 * Queue mode is per host thread, like the reference's helper state.
 *
 *   MBE_BATCH_STATE_WRITEBACK  after every mbe_flush() the three structs of every channel that had frames are current on
 *                              the host, exactly as after synchronous calls (15.6 KB of PCIe per channel and flush)
 *   MBE_BATCH_STATE_RESIDENT   the structs are uploaded at a channel's first queued frame and then live on the device
 *                              until mbe_batchEnd() / mbe_batchRelease(cur_mp); the host must not read or change them
 *                              in between (338 B of PCIe per frame).  A synchronous mbe_process* call on a resident
 *                              channel first flushes and releases it, so mixing the two is safe, just slow. */
#define MBE_BATCH_STATE_WRITEBACK 0
#define MBE_BATCH_STATE_RESIDENT 1
int mbe_batchBegin(int state_mode);          /* 0, or MBE_STATUS_INVALID_ARGUMENT (bad mode, already in queue mode) */
int mbe_flush(void);                         /* frames run (>= 0) */
int mbe_batchPending(void);                  /* frames queued and not yet flushed by this thread */
int mbe_batchRelease(mbe_parms* cur_mp);     /* flush, then write one resident channel's state back and forget it */
int mbe_batchEnd(void);                      /* flush, write every resident channel back, leave queue mode */

#ifdef __cplusplus
}
#endif
#endif /* MBE_NEO_AMD_H */
