/*
 * mbx.h -- C-ABI of the MI355X batch launcher (libmbx_hip.so).
 *
 * This is the drop-in boundary for the hot path: plain pointers and sizes, no C++ or torch
 * types.  Every entry point names the reference interface it stands in for; `ref:` paths are
 * relative to the reference repository (arancormonk/mbelib-neo v2.0.0).  The reference is a
 * per-frame, per-thread C library (include/mbelib-neo/mbelib.h); a batch is S independent
 * streams x T consecutive frames per stream, and what the reference keeps per *thread* is
 * kept here per *stream* (mbx_stream_rng).
 *
 * Pointers whose name starts with d_ are DEVICE pointers (hipMalloc / torch CUDA tensors);
 * `stream` is a hipStream_t passed as void* (NULL = default stream).  All launchers are
 * asynchronous on `stream`, never synchronise, never allocate (after mbx_reserve()), and return 0 or a negative
 * MBE_STATUS_* / MBX_E* code.  There is no CPU fallback: without a HIP device every launcher
 * fails with MBX_ENODEVICE.
 *
 * Devices and threads.  mbx_init(device, ...) creates the context of ONE device; call it once per device the process
 * uses.  Every launcher then works on the context of the calling thread's CURRENT device (hipSetDevice), like any HIP
 * call, and its pointers / stream must belong to that device.  The library is re-entrant the way the reference is
 * (ref include/mbelib-neo/mbelib.h:28-30: re-entrant per stream, mutable helper state thread-local): any number of host
 * threads may call any launcher concurrently, on separate hipStream_t's or on a shared one; the only mutable launcher
 * state -- the expand workspace below -- is owned per (device, hipStream_t) and guarded internally.  What the caller
 * owns (state, rng, outputs of one stream set) must of course not be handed to two launches that can overlap.
 * mbx_last_error() is per thread.
 */
#ifndef MBX_H
#define MBX_H

#include <stddef.h>
#include <stdint.h>

#include "mbx_tables.h"
#include "mbx_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MBX_ENODEVICE (-100) /* no HIP device / HIP call failed */
#define MBX_ENOTINIT  (-101) /* mbx_init() has not been called on this device */
#define MBX_EBADTABLE (-102) /* table blob has the wrong magic, size or checksum */

/* ---- lifetime ------------------------------------------------------------------------ */

/* Upload the constant tables (include/mbx_tables.h) to `device` and build the derived
 * device tables (LCG jump-ahead, FFT twiddles).  Replaces the reference's lazily built
 * thread-local plan/caches: ref src/core/mbelib.c:164-171 (mbe_get_fft_plan),
 * src/imbe/imbe7200x4400.c:91-115, src/ambe/ambe3600x2450.c:54-78.
 * Makes `device` the calling thread's current device.  Idempotent: a second call for a device that already holds the
 * same tables returns 0 without touching it (any thread may call it). */
int mbx_init(int device, const void* table_blob, size_t table_bytes);
/* frees the contexts of all devices (the caller must have no launches in flight) */
void mbx_shutdown(void);
/* 1 when mbx_init() has completed for `device` */
int mbx_device_ready(int device);
/* Walking order of successive stream-kernel launches over the same state, process-wide: alternate != 0 (the default;
 * MBX_NO_REVERSE=1 in the environment starts with 0) = every other launch of a (device, stream) slot / session walks the
 * streams backwards, so a launch that directly follows another over the same state finds the tail of it in the Infinity
 * Cache; 0 = always forwards.  Results do not depend on it.  Returns the previous setting.  (No reference counterpart:
 * the reference decodes one stream per call.) */
int mbx_set_stream_order(int alternate);
/* Tone synthesis on (the default) or off, process-wide and for every initialised device: the run-time form of the reference's
 * NOTONES build option (-DDISABLE_AMBE_TONES: ref CMakeLists.txt:41,330-337), the one compile-time option that changes what this
 * path computes.  Off: AMBE+2 / D-STAR tone frames (mbe_synthesizeTonef, mbe_synthesizeTonefdstar and the tone frames inside
 * mbe_processAmbe*) synthesise 160 samples of silence and leave swn / tonePhase alone (ref src/core/mbelib.c:747-751,815-819);
 * flags, results and every other frame are unchanged.  MBX_DISABLE_TONES=1 in the environment starts with it off.  Takes effect
 * for launches issued after the call.  Returns the previous setting (1 = on). */
int mbx_set_tone_synthesis(int enabled);
/* FNV-1a-32 of the table blob resident on the current device (for the per-rank checksum after the broadcast). */
uint32_t mbx_table_checksum(void);
/* message of the last failure on the calling thread */
const char* mbx_last_error(void);

/* The expand workspace of the stream stage: 256 B per frame for the expanded parameters of the AMBE codecs and of
 * one-frame-per-stream IMBE launches of more than 256 streams (other IMBE launches expand inside the stream kernel and
 * need none).
 * Each (device, hipStream_t) owns one, grown on demand -- growth waits for that stream and allocates, so size it up
 * front where that matters (required before stream capture):
 *   mbx_reserve(n)            the default stream's workspace holds >= n frames now, and every other stream's at least
 *                             that much from the next time it has to grow (n = the largest S*T of one launch); streams
 *                             seen earlier are not touched -- their handles may be gone
 *   mbx_reserve_stream(s, n)  the same for one stream, creating its slot
 *   mbx_release_stream(s)     waits for `s` and frees its slot (call before hipStreamDestroy on long-lived processes)
 * Alternatively the caller owns the workspace: the *_ws launchers take a device buffer of mbx_workspace_bytes(S*T)
 * bytes and touch no internal buffer at all. */
int mbx_reserve(size_t max_frames);
/* 1 when mbx_process_records / mbx_process_batch issue the parameter expansion of an S x T batch of `codec` as a separate
 * launch through the workspace (one-frame-per-stream IMBE launches of more than 256 streams, AMBE launches with fewer than
 * four frames per stream), 0 when the stream kernel expands the records itself. */
int mbx_uses_expand_launch(int codec, int S, int T);
int mbx_reserve_stream(void* stream, size_t max_frames);
int mbx_release_stream(void* stream);
size_t mbx_workspace_bytes(size_t max_frames);

/* ---- multi-GPU start-up: the ONE collective of the path --------------------------------- */

/* Streams shard across the GPUs of a node with no data-path collective (frames of a stream stay on one wavefront); what
 * the ranks share is the constant-table blob.  SURVEY.md §8(e) / north_star: "RCCL broadcast of the shared codebook
 * tables over xGMI only".  The reference has no counterpart (single-process CPU library; ref include/mbelib-neo/mbelib.h
 * has no communication API).  RCCL is bound at run time (librccl.so.1): a host that never calls these needs none.
 *
 *   mbx_init_broadcast(comm, root, device, blob, bytes, minmax, stream)
 *       comm = an ncclComm_t of the caller's (RCCL / NCCL API) or one made by mbx_comm_init() below; every rank calls it.
 *       Rank `root` passes the blob (mbx_tables.bin, sizeof(mbx_tables) bytes); every other rank passes a buffer of the
 *       same size that RECEIVES it.  One ncclBroadcast moves the bytes GPU to GPU, every rank then runs
 *       mbx_init(device, blob, bytes) on its copy, and an ncclAllReduce (min, max) of the per-rank mbx_table_checksum()
 *       must coincide -- MBX_EBADTABLE otherwise.  minmax (may be NULL) receives {min, max}.
 *       A rank whose copy fails that validation still takes part in the agreement (with a pair that cannot agree), so every
 *       rank returns -- the failed one with its own error, the others with MBX_EBADTABLE -- instead of waiting for ever.
 *   mbx_comm_agree(comm, value, minmax, stream)
 *       the agreement step alone: 0 when every rank passed the same 32-bit value, MBX_EBADTABLE on every rank otherwise.
 *   mbx_comm_unique_id(id)  rank 0 makes the 128-byte id and hands it to the other ranks by whatever channel the host has
 *   mbx_comm_init(&comm, nranks, id, rank, device) / mbx_comm_destroy(comm)
 *       ncclCommInitRank / ncclCommDestroy for hosts that do not want the RCCL headers; one rank per GPU. */
#define MBX_COMM_ID_BYTES 128
int mbx_comm_unique_id(void* id128);
int mbx_comm_init(void** comm, int nranks, const void* id128, int rank, int device);
int mbx_comm_destroy(void* comm);
int mbx_comm_agree(void* comm, uint32_t value, uint32_t* min_max, void* stream);
int mbx_init_broadcast(void* comm, int root, int device, void* table_blob, size_t table_bytes, uint32_t* checksums_min_max,
                       void* stream);

/* ---- host-side frame packing (no device work) ----------------------------------------- */

/* imbe_fr[8][23] / ambe_fr[4][24] arrays of 0/1 chars -> 18 / 9 byte wire frames.
 * Validation follows ref src/internal/mbe_result.h:18-29: the WHOLE array is checked,
 * NULL -> MBE_STATUS_INVALID_ARGUMENT, any cell outside {0,1} -> MBE_STATUS_INVALID_BITS
 * and nothing is written. */
int mbx_pack_imbe7200x4400(const char* frames /* n*8*23 */, size_t n, uint8_t* packed /* n*18 */);
/* IMBE 7100x4400: char[7][24] cells, rows of 19, 24, 23, 23, 15, 15, 23 -> 142 bits in 18 bytes */
int mbx_pack_imbe7100x4400(const char* frames /* n*7*24 */, size_t n, uint8_t* packed /* n*18 */);
int mbx_pack_ambe3600x2450(const char* frames /* n*4*24 */, size_t n, uint8_t* packed /* n*9 */);
/* parameter record -> the reference's imbe_d[88] / ambe_d[49] chars and mbe_process_result */
void mbx_unpack_records(const mbx_param_record* rec, size_t n, int nbits /* 88|49 */, char* bits /* n*nbits, or NULL */,
                        mbe_process_result* results /* n, or NULL */);

/* The same packing on the DEVICE, for hosts that keep whole batches of bursts as the reference's
 * cell arrays: d_cells = n x (8*23 | 4*24 | 7*24) chars as uploaded, d_packed = n wire frames, d_status[i] (may be NULL)
 * = 0 or MBE_STATUS_INVALID_BITS when any cell of frame i -- unused cells included, like mbe_validate_bits -- is outside
 * {0, 1}; such a frame must not be decoded (mask it out of the batch or drop the stream's tick).  184 B instead of 18 B
 * of PCIe per IMBE frame, but no per-frame host loop.
 * Wire order = row after row (C0 first), inside a row from the highest cell down to cell 0: the order in which a
 * deinterleaver that fills imbe_fr[r][j] / ambe_fr[r][j] per the air-interface tables would emit the bits of row r. */
int mbx_pack_cells(int codec, const char* d_cells, size_t n, uint8_t* d_packed, int32_t* d_status, void* stream);

/* From an air-interface burst straight to the wire frame.  The reference's callers deinterleave a voice
 * burst into imbe_fr[r][j] / ambe_fr[r][j] with tables of their own (the reference holds none: ref include/mbelib-neo/mbelib.h:429,505
 * take the filled arrays); such a host can fold its tables into ONE permutation that writes the packed frame directly, without the
 * 184 / 96-byte cell array in between:
 *     mbx_wire_bit_of_cell(codec, r, j)   where cell [r][j] sits in the wire frame: bit index from the frame's first bit (index 0 = bit 7
 *                                         of byte 0; rows of 23,23,23,23,15,15,15,7 | 19,24,23,23,15,15,23 | 24,23,11,14 cells, row after
 *                                         row, inside a row from the highest cell down to cell 0), or -1 for a cell the codec does not use
 *     mbx_wire_permutation(codec, cell_row, cell_col, n, wire_bit)
 *                                         for a burst of n = 144 | 142 | 72 channel bits whose i-th received bit the caller's tables send to
 *                                         cell (cell_row[i], cell_col[i]): wire_bit[i] = its index in the wire frame.  Returns
 *                                         MBE_STATUS_INVALID_ARGUMENT unless the schedule is a bijection onto the codec's cells.
 * A frame is then `for i: if (bit[i]) frame[wire_bit[i] >> 3] |= 0x80 >> (wire_bit[i] & 7)` on a zeroed 18 / 9 byte buffer -- the bytes
 * mbx_pack_* would have produced from the filled array (tests/test_host_logic.py holds that for random schedules).  Host-only. */
int mbx_wire_bit_of_cell(int codec, int row, int col);
int mbx_wire_permutation(int codec, const int* cell_row, const int* cell_col, int n, int* wire_bit);

/* ---- FEC stage: frames -> parameter records (stateless, one thread per frame) ---------- */

/* ref: mbe_decodeImbe7200x4400Frame  include/mbelib-neo/mbelib.h:471, src/imbe/imbe7200x4400.c:709-744
 *      (Golay(23,12) src/ecc/ecc.c:221-301, Hamming(15,11) src/ecc/ecc.c:366-408,
 *       PR demodulation src/imbe/imbe7200x4400.c:636-673) */
int mbx_fec_imbe7200x4400(const uint8_t* d_frames /* n*18 */, size_t n, mbx_param_record* d_records /* n */,
                          void* stream);
/* ref: mbe_decodeAmbe3600x2450Frame  include/mbelib-neo/mbelib.h:395, src/ambe/ambe3600x2450.c:649-682,
 *      src/ambe/ambe_common.c:22-46, 75-100, 127-157 */
int mbx_fec_ambe3600x2450(const uint8_t* d_frames /* n*9 */, size_t n, mbx_param_record* d_records /* n */,
                          void* stream);

/* ref: mbe_decodeImbe7100x4400Frame  include/mbelib-neo/mbelib.h:545, src/imbe/imbe7100x4400.c:440-479 (C0 :100-122,
 *      demodulation :292-334, data ECC :153-212, mbe_convertImbe7100to7200 :381-438; mbe_7100x4400hamming1511
 *      src/ecc/ecc.c:422-464).  The records hold the converted bits: feed them to mbx_process_records with
 *      MBX_CODEC_IMBE7200X4400, or call mbx_process_batch with MBX_CODEC_IMBE7100X4400. */
int mbx_fec_imbe7100x4400(const uint8_t* d_frames /* n*18 */, size_t n, mbx_param_record* d_records /* n */,
                          void* stream);

/* The sub-stages of the frame decode one by one, as the reference exposes them for the classic
 * ecc -> demodulate -> ecc call sequence (packed frames in, packed frames out; see mbx_fec.hip):
 *   MBX_STAGE_C0          ref mbe_ecc{Imbe7200x4400,Imbe7100x4400,Ambe3600x2450,Ambe3600x2400}C0   mbelib.h:457,531,381,286
 *   MBX_STAGE_DEMODULATE  ref mbe_demodulate*Data                                                  mbelib.h:463,535,387,307
 *   MBX_STAGE_DATA        ref mbe_ecc*Data (parameter bits to d_out; 7100x4400 in its own bit order) mbelib.h:459,533,383,293
 *   MBX_STAGE_CONVERT7100 ref mbe_convertImbe7100to7200 (d_in and d_out are parameter records)       mbelib.h:537
 * d_out[i].w[3] = the reference's return value (corrected errors) in bits 0..7, the C4 count in bits 16..23. */
#define MBX_STAGE_C0 1
#define MBX_STAGE_DEMODULATE 2
#define MBX_STAGE_DATA 4
#define MBX_STAGE_CONVERT7100 8
int mbx_fec_stage(int codec, int stage, const void* d_in, size_t n, uint8_t* d_frames_out /* C0, DEMODULATE */,
                  mbx_param_record* d_out /* DATA, CONVERT7100 */, void* stream);

/* ---- soft-decision FEC stage: soft frames -> parameter records ------
 * One wavefront per frame; exhaustive maximum-likelihood Golay/Hamming decode with the reference's
 * tie rules, bit-exact.  Frames keep the reference's own shapes: n x mbe_soft_bit[8][23] (IMBE) or
 * n x mbe_soft_bit[4][24] (AMBE+2).  Hard decisions must be 0/1 (mbx_validate_soft_bits, host); the
 * records carry MBE_PROCESS_FLAG_SOFT_INPUT.
 * ref: mbe_decodeImbe7200x4400SoftFrame / mbe_decodeAmbe3600x2450SoftFrame
 *      include/mbelib-neo/mbelib.h:437-447, 513-523; src/imbe/imbe7200x4400.c:445-459,517-560,675-707,746-778;
 *      src/ambe/ambe_common.c:48-73,102-124,159-190; src/ambe/ambe3600x2450.c:684-714;
 *      mbe_golay2312Soft src/ecc/ecc.c:303-357, mbe_hamming1511Soft src/ecc/ecc.c:157-215,410-413 */
int mbx_fec_soft(int codec, const mbe_soft_bit* d_soft /* n*184 | n*96 */, size_t n, mbx_param_record* d_records,
                 void* stream);
/* ref: mbe_processImbe7200x4400SoftFrame[f] / mbe_processAmbe3600x2450SoftFrame[f]
 *      src/imbe/imbe7200x4400.c:950-980, src/ambe/ambe3600x2450.c:939-969: soft FEC, then mbx_process_records */
int mbx_process_batch_soft(int codec, int S, int T, const mbe_soft_bit* d_soft, mbe_parms* d_state, mbx_stream_rng* d_rng,
                           int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results, mbx_param_record* d_records,
                           void* stream);
/* ref: mbe_golay2312Soft (kind 0, 23 soft bits per block) / mbe_hamming1511Soft (kind 1, 15 soft bits)
 *      include/mbelib-neo/mbelib.h:246, 260.  out[i] bit j = corrected cell j, errs[i] = the reference's return value */
int mbx_ecc_soft_words(int kind, const mbe_soft_bit* d_in, size_t n, uint32_t* d_out, int32_t* d_errs, void* stream);
/* host helpers -- ref mbe_validate_soft_bits src/internal/mbe_result.h:31-42, mbe_softBitsFromHard / mbe_softBitsFromLlr
 * include/mbelib-neo/mbelib.h:219-224, src/core/mbelib.c:133-158 */
int mbx_validate_soft_bits(const mbe_soft_bit* soft, size_t count); /* 0, -1 (NULL), -2 (a hard decision > 1) */
int mbx_soft_bits_from_hard(const char* bits, mbe_soft_bit* soft, size_t count, uint8_t reliability);
int mbx_soft_bits_from_llr(const int16_t* llr, mbe_soft_bit* soft, size_t count);

/* ---- stream stage: parameter records + per-stream state -> PCM (one wavefront per stream) */

/* ref: mbe_processImbe4400Dataf / mbe_processAmbe2450Dataf
 *      include/mbelib-neo/mbelib.h:491, 415; src/imbe/imbe7200x4400.c:858-909;
 *      src/ambe/ambe3600x2450.c:851-898 -- applied to frame t = 0..T-1 of every stream in order.
 * d_records is stream-major (stream s, frame t at s*T + t) and carries the C0/C4 error context
 * exactly as the frame-level entry points hand it over.
 * d_state: 3 structs per stream {cur_mp, prev_mp, prev_mp_enhanced}, read and written.
 * Outputs (each may be NULL): int16 PCM (mbe_floattoshort applied, ref src/core/mbelib.c:1296),
 * float PCM, per-frame mbe_process_result. */
int mbx_process_records(int codec, int S, int T, const mbx_param_record* d_records, mbe_parms* d_state,
                        mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results,
                        void* stream);

/* The same with a caller-owned expand workspace (mbx_workspace_bytes(S*T) bytes; may be NULL for IMBE with T > 1). */
int mbx_process_records_ws(int codec, int S, int T, const mbx_param_record* d_records, mbe_parms* d_state,
                           mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results,
                           void* d_workspace, size_t workspace_bytes, void* stream);

/* The two halves of mbx_process_records(), exposed so that a caller (bench.py) can time them apart:
 * mbx_expand_records() runs the frame-parallel, stateless half of the parameter decode (one thread per
 * frame: fundamental, voicing, dequantisation, block inverse DCTs; ref src/imbe/imbe7200x4400.c:117-270,
 * src/ambe/ambe3600x2450.c:176-387) into the workspace of `stream`; mbx_stream_expanded() runs the
 * stateful half (prediction, policy, synthesis) on it, one wavefront per stream, and fails unless the last
 * mbx_expand_records() on the same `stream` was for the same codec, frame count and record array. */
int mbx_expand_records(int codec, const mbx_param_record* d_records, size_t n, void* stream);
int mbx_stream_expanded(int codec, int S, int T, const mbx_param_record* d_records, mbe_parms* d_state,
                        mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results,
                        void* stream);

/* pinned host memory -> device memory by a kernel on `stream` (16-byte aligned buffers): what the sessions use for the wire
 * frames of a batch instead of a DMA copy, which would queue behind the previous batch's PCM on the copy engine */
int mbx_stage_in(void* d_dst, const void* pinned_src, size_t bytes, void* stream);

/* A FRAME SERVER: one wavefront that stays on the device and serves mbx_process_frame requests from a mailbox in pinned,
 * coherent host memory (hipHostMalloc(..., hipHostMallocCoherent), 64-byte aligned) -- the path of a host that calls the
 * synchronous mbe_process*Frame[f] back to back.  A launch per call costs the launch itself (~5 us) and an instruction cache
 * that every dispatch starts cold (~3 us of a 16 us kernel); a served request costs neither.
 *   the buffers (state triplet, rng, PCM, result, record: the arguments of mbx_process_frame, device-accessible) are fixed when
 *   the server starts; a request names the codec, the outputs wanted and carries the wire frame.
 *   host, per request:  write codec / want / frame into the mailbox, then store ++seq_in with release order (everything in
 *                       ONE 64-byte line: the server reads it with one load); complete when seq_out == seq_in (acquire load).
 *   server:             polls that line; exits -- after storing alive = 0, touching nothing afterwards -- when quit != 0, when
 *                       no request has arrived for idle_us microseconds, or after 2 s (it cannot outlive its host by more, and
 *                       a device-wide synchronisation elsewhere in the process is held up by idle_us at most).  A request that
 *                       arrives while alive == 0 is NOT served: the host sets alive = 1 and starts a server again (same
 *                       mailbox, same stream: it queues behind the one that is leaving), which finds seq_in != seq_out.
 * mbx_frame_server_start launches the wavefront on `stream` (a stream of its own: it occupies the stream while it lives).
 * (struct mbx_frame_mailbox: include/mbx_types.h) */
int mbx_frame_server_start(mbx_frame_mailbox* mailbox, unsigned idle_us, mbe_parms* d_state, mbx_stream_rng* d_rng, int16_t* d_pcm16,
                           float* d_pcmf, mbe_process_result* d_result, mbx_param_record* d_record, mbe_parms* d_shadow_state,
                           mbx_stream_rng* d_shadow_rng, uint32_t* d_shadow_ok, void* stream);
/*   d_shadow_*: the device copy of the state as in mbx_process_frame_shadow (all three NULL: none); a request whose `want` has
 *   MBX_FRAME_WANT_SHADOW set reads its state from the copy. */

/* ONE frame of ONE stream as one launch of one wavefront (FEC + parameter decode + policy + synthesis + float->int16):
 * what the synchronous mbe_process*Frame[f] of libmbe_neo_amd.so issue.
 *   ref: mbe_processImbe7200x4400Frame[f]  include/mbelib-neo/mbelib.h:429-441, mbe_processAmbe3600x2450Frame[f] :505-517,
 *        mbe_processImbe7100x4400Frame[f] :564-576, mbe_processAmbe3600x2400Frame[f] :352-364
 * Every pointer must be device-accessible -- device memory, or PINNED host memory (hipHostMalloc / mbx_host_alloc): the
 * kernel reads the three structs at its start and writes them at its end, nothing in between touches them, so with the
 * caller's structs in pinned memory the call needs no copy in either direction.  d_frame: one wire frame (18 | 9 bytes);
 * d_state: {cur, prev, prev_enhanced}; d_pcm16 / d_pcmf: 160 samples, either may be NULL; d_result may be NULL; d_record:
 * the 16-byte parameter record (out).  d_done (may be NULL): after everything else has been written the kernel stores
 * `token` there with system scope -- a host that polls a word of pinned memory for the token needs no stream
 * synchronisation to know that the outputs are complete. */
int mbx_process_frame(int codec, const uint8_t* d_frame, mbe_parms* d_state, mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf,
                      mbe_process_result* d_result, mbx_param_record* d_record, uint32_t* d_done, uint32_t token, void* stream);

/* The same with a DEVICE COPY of the state kept between calls: d_shadow_state (3 structs) and d_shadow_rng in device memory,
 * d_shadow_ok a word of pinned memory.  The frame's results always go to d_state / d_rng AND to the copy; with use_shadow != 0
 * the frame READS its state from the copy instead of d_state / d_rng -- for a caller that knows (by comparing on the host) that
 * the structs it passes still are what the previous call returned, and *d_shadow_ok was 1 after that call: the 7.8 KB then
 * come from HBM (~1 us) instead of across PCIe (~3 us).  The kernel stores 1 to *d_shadow_ok when the copy is complete after
 * this frame and 0 when it is not (an AMBE tone frame, which leaves prev_mp_enhanced alone).
 * h_frame (may be NULL): the frame's bytes where the HOST can read them (for pinned memory: d_frame itself) -- they then travel
 * with the launch as kernel arguments and the kernel does not fetch d_frame at all. */
int mbx_process_frame_shadow(int codec, const uint8_t* d_frame, mbe_parms* d_state, mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf,
                             mbe_process_result* d_result, mbx_param_record* d_record, uint32_t* d_done, uint32_t token,
                             mbe_parms* d_shadow_state, mbx_stream_rng* d_shadow_rng, uint32_t* d_shadow_ok, int use_shadow,
                             const uint8_t* h_frame, void* stream);

/* FEC stage + stream stage back to back:
 * ref: mbe_processImbe7200x4400Frame[f] include/mbelib-neo/mbelib.h:505-511,
 *      mbe_processAmbe3600x2450Frame[f] include/mbelib-neo/mbelib.h:429-435.
 * d_records is a caller-provided S*T workspace (it also returns imbe_d / ambe_d). */
int mbx_process_batch(int codec, int S, int T, const uint8_t* d_frames, mbe_parms* d_state, mbx_stream_rng* d_rng,
                      int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results, mbx_param_record* d_records,
                      void* stream);
int mbx_process_batch_ws(int codec, int S, int T, const uint8_t* d_frames, mbe_parms* d_state, mbx_stream_rng* d_rng,
                         int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results, mbx_param_record* d_records,
                         void* d_workspace, size_t workspace_bytes, void* stream);

/* ref: mbe_decodeImbe4400Parms / mbe_decodeAmbe2450Parms / mbe_decodeAmbe2400Parms  include/mbelib-neo/mbelib.h:461, 385, 301
 * (src/imbe/imbe7200x4400.c:589-630, src/ambe/ambe3600x2450.c:555-634, src/ambe/ambe3600x2400.c:427-561): the parameter
 * decode alone -- no frame policy, no synthesis -- for n (cur_mp, prev_mp) pairs.  d_rc[i] = the reference's return value. */
int mbx_decode_parms(int codec, const mbx_param_record* d_records, size_t n, mbe_parms* d_cur, mbe_parms* d_prev, int32_t* d_rc,
                     void* stream);

/* The same for a SUBSET of a larger resident pool of streams: batch row s (its T frames, records, PCM and results at
 * rows s*T .. s*T+T-1) belongs to the stream whose state / rng live in slot d_stream_index[s] of the pools.  The index
 * must not name a slot twice.  This is what a host with many open channels needs each 20 ms tick: only the channels
 * that received a frame are launched, the state of the others stays where it is (used by the queue mode of the
 * per-frame API, include/mbe_neo_amd.h, and by mbx_session_submit_indexed). */
int mbx_process_batch_indexed(int codec, int S, int T, const int32_t* d_stream_index, const uint8_t* d_frames,
                              mbe_parms* d_state_pool, mbx_stream_rng* d_rng_pool, int16_t* d_pcm16, float* d_pcmf,
                              mbe_process_result* d_results, mbx_param_record* d_records, void* stream);

/* The two halves with a caller-owned workspace (mbx_workspace_bytes(S*T) bytes), free to run on DIFFERENT streams: the
 * front end of batch k + 1 (mbx_fec_* + mbx_expand_records_ws) depends only on its frames, not on the stream stage of batch k,
 * so a host that decodes batch after batch MAY overlap them and order them with its own events (front end on one stream into
 * alternating record / workspace buffers, stream stage on another).  The pair allows it; it is not a recommendation: on MI355X
 * at T = 1 it measured SLOWER than a single stream (bench.py uses it only with --overlap-front-end: 255 against 265 M frames/s),
 * and the IMBE codecs' T = 1 step is one fused launch in mbx_process_batch anyway.  d_resident: NULL, or the resident words of
 * mbx_process_batch_resident. */
int mbx_expand_records_ws(int codec, const mbx_param_record* d_records, size_t n, void* d_workspace, size_t workspace_bytes,
                          void* stream);
int mbx_stream_expanded_ws(int codec, int S, int T, const mbx_param_record* d_records, mbe_parms* d_state, uint32_t* d_resident,
                           mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results,
                           const void* d_workspace, size_t workspace_bytes, void* stream);

/* Resident state -- for a caller that OWNS the state of its streams between launches (sessions; the queue mode's device
 * pool).  d_state_pool is the same [slots][3] array of ABI structs, d_resident one word per slot (zero-initialised by the
 * caller).  After every ordinary frame the reference leaves prev_mp_enhanced a field-for-field copy of cur_mp
 * (ref src/imbe/imbe7200x4400.c:842-856, src/ambe/ambe3600x2450.c:790-800); while d_resident[slot] != 0 that struct is
 * ELIDED: the kernels neither write nor read it, and of prev_mp they fetch only what the decode reads (the rest on demand:
 * repeats, erasures).  PCM, results and every later frame are bit-identical to mbx_process_batch_indexed on the same
 * inputs; the three structs are in their ABI form again after mbx_resident_materialize() (which clears the words).
 * d_stream_index may be NULL (row s = slot s).  ref (what the state is): include/mbelib-neo/mbelib.h:88-139. */
int mbx_process_batch_resident(int codec, int S, int T, const int32_t* d_stream_index, const uint8_t* d_frames,
                               mbe_parms* d_state_pool, uint32_t* d_resident, mbx_stream_rng* d_rng_pool, int16_t* d_pcm16,
                               float* d_pcmf, mbe_process_result* d_results, mbx_param_record* d_records, void* stream);
/* the stream stage alone on resident state, after mbx_expand_records() on the same stream (cf. mbx_stream_expanded) */
int mbx_stream_expanded_resident(int codec, int S, int T, const mbx_param_record* d_records, mbe_parms* d_state, uint32_t* d_resident,
                                 mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results,
                                 void* stream);
/* prev_mp_enhanced := cur_mp for the n listed slots (d_stream_index NULL: slots 0..n-1) whose struct is elided */
int mbx_resident_materialize(int n, const int32_t* d_stream_index, mbe_parms* d_state_pool, uint32_t* d_resident, void* stream);

/* ref: mbe_synthesizeSpeechf  include/mbelib-neo/mbelib.h:652, src/core/mbelib.c:1112-1115.
 * One frame for each of S (cur, prev) pairs; both structs are updated like the reference does. */
int mbx_synthesize_speech(int S, mbe_parms* d_cur, mbe_parms* d_prev, mbx_stream_rng* d_rng, float* d_pcmf,
                          int16_t* d_pcm16, void* stream);

/* ref: mbe_floattoshort  include/mbelib-neo/mbelib.h:675, src/core/mbelib.c:1148-1177 */
int mbx_floattoshort(const float* d_in, int16_t* d_out, size_t nframes, void* stream);

/* Per-batch counters over mbe_process_result (ref include/mbelib-neo/mbelib.h:154-166: what a host tallies from the flags and
 * the error counts of every frame), formed on the device: 20 bytes read per frame instead of a copy of every result to the
 * host.  The call ADDS the n results at d_results to *d_hist (device memory; zero it, e.g. with hipMemsetAsync, to start a new
 * tally; any number of launches may accumulate into one).  Asynchronous on `stream`. */
typedef struct mbx_result_hist {
    uint64_t frames;               /* results counted */
    uint64_t flag[8];              /* frames with bit k of `flags` set: 0 SOFT_INPUT, 1 C0_VALID, 2 C4_VALID, 4 TONE, 5 ERASURE, 6 REPEAT, 7 MUTE */
    uint64_t c0_errors;            /* sums of the three counts and of their total ... */
    uint64_t protected_errors;
    uint64_t c4_errors;
    uint64_t total_errors;
    uint64_t frames_with_errors;   /* ... and the number of frames with total_errors > 0 */
} mbx_result_hist;
int mbx_result_histogram(const mbe_process_result* d_results, size_t n, mbx_result_hist* d_hist, void* stream);

/* ---- single stages of the public API, batched (one wavefront per struct) ------------------- */

/* ref: mbe_spectralAmpEnhance  include/mbelib-neo/mbelib.h:623, src/core/mbelib.c:641-666 */
int mbx_spectral_amp_enhance(int S, mbe_parms* d_parms, void* stream);
/* ref: mbe_applyAdaptiveSmoothing  include/mbelib-neo/mbelib.h:725, src/core/mbe_adaptive.c:268-276 */
int mbx_adaptive_smoothing(int S, mbe_parms* d_cur, const mbe_parms* d_prev, void* stream);
/* ref: mbe_synthesizeComfortNoisef / mbe_synthesizeComfortNoise  include/mbelib-neo/mbelib.h:706-712 */
int mbx_comfort_noise(int S, mbx_stream_rng* d_rng, float* d_pcmf, int16_t* d_pcm16, void* stream);
/* ref: mbe_golay2312 / mbe_checkGolayBlock (kind 0), mbe_hamming1511 (kind 1), mbe_7100x4400hamming1511 (kind 2)
 *      include/mbelib-neo/mbelib.h:231-267.
 * One code word per element: bit j of in[i] is cell j; out[i] is the corrected word (Golay parity bits
 * pass through like the reference), errs[i] the corrected-bit count (may be NULL). */
int mbx_ecc_words(int kind, const uint32_t* d_in, size_t n, uint32_t* d_out, int32_t* d_errs, void* stream);
/* ref: mbe_synthesizeTonef (d_dstar_ids == NULL: AMBE+2 tone from the parameter bits of d_records[s]) /
 *      mbe_synthesizeTonefdstar (d_dstar_ids[s] = tone index)  include/mbelib-neo/mbelib.h:630, 638;
 *      src/core/mbelib.c:691-856.  One struct per element; the tone phases in d_cur[s] advance; invalid tones give silence. */
int mbx_synthesize_tone(int S, const mbx_param_record* d_records, const int32_t* d_dstar_ids, mbe_parms* d_cur, float* d_pcmf,
                        int16_t* d_pcm16, void* stream);
/* Loads and stores the state triplet of S streams without touching it: the HBM-traffic floor of the
 * stream stage (used by bench.py --calibrate to price the access pattern; not a reference function). */
int mbx_state_copy(int S, mbe_parms* d_state, void* stream);

/* ---- sessions: device-resident state, frames in from / PCM out to HOST memory (mbx_session.hip) -------------------
 * The fan-in path for a C host that keeps the reference's per-frame bookkeeping but hands frames over in batches: the
 * state triplets ({cur_mp, prev_mp, prev_mp_enhanced} of ref mbe_process*Frame, include/mbelib-neo/mbelib.h:429,505)
 * and the per-stream RNG (the reference's thread-local helper state) of `streams` streams live on the device for the
 * life of the session, so one 20 ms frame costs 18 B (9 B) of PCIe in and 320 B out (+ 20 B with results) instead of
 * the 15.6 KB of state mbx_process_batch_host() moves.  Submissions are asynchronous and pipelined three deep (copy-in,
 * kernels and copy-out of consecutive batches overlap on three HIP streams); outputs are valid after
 * mbx_session_wait().  Pinned host buffers (mbx_host_alloc, hipHostMalloc, hipHostRegister) are used in place; pageable
 * ones are staged through pinned memory owned by the session (frames are copied out before submit returns, outputs are
 * copied in by the wait / by the submit that reuses the pipeline slot).  A session belongs to the device that was
 * current when it was created and may be driven by one host thread at a time; any number of sessions may run
 * concurrently.  Streams start from mbe_initMbeParms defaults and the default RNG. */
typedef struct mbx_session mbx_session;
#define MBX_SESSION_PCM16   1u   /* int16 PCM (mbe_floattoshort applied); the default when `outputs` is 0 */
#define MBX_SESSION_PCMF    2u   /* float PCM */
#define MBX_SESSION_RESULTS 4u   /* mbe_process_result per frame */
int mbx_session_create(mbx_session** out, int codec, int streams, size_t max_frames_per_submit, unsigned outputs);
int mbx_session_destroy(mbx_session* s);
int mbx_session_streams(const mbx_session* s);
/* frames: [streams][T] wire frames, stream-major; every stream advances by T frames.  Output pointers may be NULL. */
int mbx_session_submit(mbx_session* s, int T, const uint8_t* frames, int16_t* pcm16, float* pcmf, mbe_process_result* results);
/* the same for n of the session's streams: batch row i belongs to stream stream_index[i] (no stream twice); the
 * others keep their state.  `records` (may be NULL) returns imbe_d / ambe_d + error context (mbx_unpack_records). */
int mbx_session_submit_indexed(mbx_session* s, int n, int T, const int32_t* stream_index, const uint8_t* frames, int16_t* pcm16,
                               float* pcmf, mbe_process_result* results, mbx_param_record* records);
/* every batch submitted so far is complete and its outputs are in the caller's buffers */
int mbx_session_wait(mbx_session* s);
/* state of streams [first, first + count): ref mbe_initMbeParms / mbe_setThreadRngSeed / direct access.  These wait. */
int mbx_session_reset(mbx_session* s, int first, int count);
int mbx_session_seed(mbx_session* s, int first, int count, const uint32_t* seeds);
int mbx_session_get_state(mbx_session* s, int first, int count, mbe_parms* state /* count*3, or NULL */, mbx_stream_rng* rng /* or NULL */);
int mbx_session_set_state(mbx_session* s, int first, int count, const mbe_parms* state, const mbx_stream_rng* rng);
/* pinned host memory for zero-copy DMA of frames / PCM */
void* mbx_host_alloc(size_t bytes);
void mbx_host_free(void* p);

/* ---- convenience: same calls on HOST buffers (stages through device memory, synchronous) -- */
int mbx_process_batch_host(int codec, int S, int T, const uint8_t* frames, mbe_parms* state, mbx_stream_rng* rng,
                           int16_t* pcm16, float* pcmf, mbe_process_result* results, mbx_param_record* records);
int mbx_synthesize_speech_host(int S, mbe_parms* cur, mbe_parms* prev, mbx_stream_rng* rng, float* pcmf,
                               int16_t* pcm16);
int mbx_floattoshort_host(const float* in, int16_t* out, size_t nframes);
int mbx_fec_host(int codec, const uint8_t* frames, size_t n, mbx_param_record* records);
int mbx_fec_soft_host(int codec, const mbe_soft_bit* soft, size_t n, mbx_param_record* records);
int mbx_process_batch_soft_host(int codec, int S, int T, const mbe_soft_bit* soft, mbe_parms* state, mbx_stream_rng* rng,
                                int16_t* pcm16, float* pcmf, mbe_process_result* results, mbx_param_record* records);
int mbx_ecc_soft_words_host(int kind, const mbe_soft_bit* in, size_t n, uint32_t* out, int32_t* errs);

/* per-stream RNG helpers (host): ref mbe_setThreadRngSeed src/core/mbelib.c:173-181 */
void mbx_rng_default(mbx_stream_rng* rng);
void mbx_rng_seed(mbx_stream_rng* rng, uint32_t seed);

/* name of the stream kernel a launch with T frames per stream takes (for the bench / profile summaries): with T >= 4 the
 * instance that keeps prev_mp / prev_mp_enhanced in LDS for the whole launch (*_lds), otherwise the HBM-slot one; T < 0:
 * the instance a resident launch (mbx_process_batch_resident) with -T frames per stream takes (*_res, *_res1) */
const char* mbx_stream_kernel_name(int codec, int T);
/* the dominant kernel of mbx_process_batch (resident != 0: of mbx_process_batch_resident) for a batch of S streams x T frames:
 * the IMBE codecs at T = 1 and S > 256 take ONE launch (ref src/imbe/imbe7200x4400.c:935-948 -- decode and process in one
 * call): 7200x4400 imbe_one_launch_kernel (front blocks -- FEC + parameter expansion of eight frames per wave -- and stream blocks
 * in one grid), 7100x4400 imbe7100_stream_kernel_one_fused (the front end in the stream's own wave); every other shape the
 * kernel mbx_stream_kernel_name() names, behind its FEC (and expansion) launches.  MBX_FUSE_ONE in the environment (A/B timing):
 * 0 = the staged launches everywhere, 1 = the in-wave form for 7200x4400 too. */
const char* mbx_batch_kernel_name(int codec, int S, int T, int resident);
/* diagnostics of the one-launch kernel: the number of its stream blocks that did not find their front block's rows in time and
 * expanded their own frame instead, since the workspace of `stream` was allocated (expected 0; results are the same either way).
 * Synchronises the stream.  -1: the stream has no workspace yet. */
long long mbx_front_fallbacks(void* stream);
/* (The fault-injection hook that FORCES that fall-back path is not part of this library: it exists only in the -DMBX_TESTING build,
 * libmbx_hip_testing.so -- `make -C mbelib-neo_amd/csrc testing` -- which tests/ load in a child process through MBX_HIP_LIBRARY.) */
/* Sliced launches.  A launch of S streams x T >= 32 frames whose S does not fill the device's resident wave slots evenly (the
 * last round of waves would run part-empty: 8,192 streams on 5,120 slots are 1.6 rounds) is cut into three groups of streams x
 * slices of 16 frames, issued in order on three internal HIP streams that are forked from and joined to the caller's stream with
 * events: the groups' kernels fill each other's empty slots.  Results are bit-identical to the plain launch (a slice IS a launch
 * of 16 frames per stream).  Not taken under stream capture.  mbx_launch_slices: the slice length in frames for a shape (0: not
 * sliced); MBX_SLICE=0 / =n in the environment switches it off / sets the slice length. */
int mbx_launch_slices(int codec, int S, int T);

#ifdef __cplusplus
}
#endif
#endif /* MBX_H */
