// mbx_fec.hip -- FEC stage: wire frames -> parameter records.  One thread per frame; frames
// are independent here (no stream state), so the grid is simply ceil(n / 256) workgroups.
//
// Replaces (ref = arancormonk/mbelib-neo v2.0.0):
//   IMBE  src/imbe/imbe7200x4400.c:424-443 (C0 Golay), :636-673 (PR demodulation),
//         :469-515 (3x Golay + 3x Hamming + raw row), :709-744 (frame decode + status)
//   AMBE  src/ambe/ambe_common.c:22-46 (C0 Golay + overall parity), :75-100 (demodulation),
//         :127-157 (Golay on C1, raw C2/C3); src/ambe/ambe3600x2450.c:649-682
//   ECC   src/ecc/ecc.c:221-301 (Golay(23,12) by syndrome table), :366-408 (Hamming(15,11))
// Further down: the IMBE 7100x4400 front end (its own C0 / seed / Hamming mapping / bit order) and the soft-decision
// front end (one wavefront per frame, exhaustive maximum-likelihood decode of every block).
// Integer work only: results are bit-exact.
#include "mbx_device.h"
#include "mbx_fec_frame.h"

namespace mbx {

__global__ void __launch_bounds__(256)
fec_imbe7200x4400_kernel(const uint8_t* __restrict__ frames, size_t n, mbx_param_record* __restrict__ out,
                         DeviceTables tabs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    *reinterpret_cast<uint4*>(&out[i]) = fec_imbe7200x4400_frame(tabs.t, frames + i * MBX_IMBE_FRAME_BYTES);
}

__global__ void __launch_bounds__(256)
fec_ambe3600x2450_kernel(const uint8_t* __restrict__ frames, size_t n, mbx_param_record* __restrict__ out,
                         DeviceTables tabs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    *reinterpret_cast<uint4*>(&out[i]) = fec_ambe3600x2450_frame(tabs.t, frames + i * MBX_AMBE_FRAME_BYTES);   // 9-byte frames: byte loads
}

// ------------------------------------------------------------------------------------------
// IMBE 7100x4400 front end: rows of 19, 24, 23, 23, 15, 15, 23 cells.
//   ref src/imbe/imbe7100x4400.c:100-122 (C0: Golay shortened to 18 cells), :292-334 (demodulation, 7-bit
//       seed), :153-212 (Golay on C1..C3, Hamming with the 7100 bit mapping on C4/C5, raw C6),
//       :381-438 (mbe_convertImbe7100to7200: a permutation of the 88 bits that depends on K(b0)),
//       :440-479 (frame decode); src/ecc/ecc.c:422-464 (mbe_7100x4400hamming1511)
// The record holds the 88 bits AFTER the conversion, i.e. in 7200x4400 order: the stream stage is the
// 7200x4400 one.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
fec_imbe7100x4400_kernel(const uint8_t* __restrict__ frames, size_t n, mbx_param_record* __restrict__ out,
                         DeviceTables tabs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    *reinterpret_cast<uint4*>(&out[i]) = fec_imbe7100x4400_frame(tabs.t, frames + i * MBX_IMBE7100_FRAME_BYTES);
}

// float -> int16 (a21): ref src/core/mbelib.c:1148-1177.  One thread per sample.
__device__ __forceinline__ int16_t float_to_pcm16(float x) {
    const float top = 32767.0f * 0.95f;
    const uint32_t bits = __float_as_uint(x);
    const uint32_t mag = bits & 0x7FFFFFFFu;
    float v;
    if (mag > 0x7F800000u) {
        v = 0.0f;
    } else if (mag == 0x7F800000u) {
        v = (bits & 0x80000000u) ? -top : top;
    } else {
        v = 7.0f * x;
        v = (v > top) ? top : ((v < -top) ? -top : v);
    }
    return (int16_t)(int)v;   // C cast: truncate toward zero
}

// Host -> device staging of a batch's wire frames by the GPU itself: `src` is PINNED host memory, read over PCIe in 16-byte
// pieces (every line once), `dst` device memory.  Used by the sessions instead of a DMA copy: a host-to-device DMA of batch
// k + 1 queues behind the device-to-host DMA of batch k's PCM on the copy engine, which serialises the next batch's kernels
// behind that transfer (measured: 0.257 ms of kernels + 0.379 ms of PCM per 65,536 frames back to back instead of
// overlapped); 1.2 MB read by a kernel on the compute stream costs 40 us and leaves the engine to the PCM.
__global__ void __launch_bounds__(256)
stage_in_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t bytes) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n16 = bytes >> 4;
    if (i < n16) {
        reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    } else if (i == n16) {
        for (size_t b = n16 << 4; b < bytes; ++b) {
            dst[b] = src[b];
        }
    }
}

__global__ void __launch_bounds__(256)
floattoshort_kernel(const float* __restrict__ in, int16_t* __restrict__ out, size_t nsamples) {
    // two samples per thread so every lane stores a full dword
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < nsamples) {
        const float2 v = *reinterpret_cast<const float2*>(in + i);
        const uint32_t lo = (uint16_t)float_to_pcm16(v.x), hi = (uint16_t)float_to_pcm16(v.y);
        *reinterpret_cast<uint32_t*>(out + i) = lo | (hi << 16);
    } else if (i < nsamples) {
        out[i] = float_to_pcm16(in[i]);
    }
}


// What a host tallies from the mbe_process_result of every frame (ref include/mbelib-neo/mbelib.h:154-166: the flag bits and
// the three error counts), for a whole batch on the device: 20 bytes read per frame, one set of atomics per workgroup.
// Integer sums only: the histogram of a batch does not depend on the launch geometry.
__global__ void __launch_bounds__(256)
result_histogram_kernel(const mbe_process_result* __restrict__ results, size_t n, unsigned long long* __restrict__ hist) {
    unsigned v[kResultHistWords];
#pragma unroll
    for (int k = 0; k < kResultHistWords; ++k) {
        v[k] = 0u;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const mbe_process_result r = results[i];
        v[0] += 1u;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            v[1 + b] += (r.flags >> b) & 1u;
        }
        v[9] += (unsigned)r.c0_errors;
        v[10] += (unsigned)r.protected_errors;
        v[11] += (unsigned)r.c4_errors;
        v[12] += (unsigned)r.total_errors;
        v[13] += r.total_errors > 0 ? 1u : 0u;
    }
    __shared__ unsigned part[4][kResultHistWords];
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int k = 0; k < kResultHistWords; ++k) {   // (a thread holds at most n / (grid x 256) + 1 frames: the 32-bit partial sums cannot wrap
        unsigned s = v[k];                          //  for the grid mbx_result_histogram launches)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            s += (unsigned)__shfl_xor((int)s, d, 64);
        }
        if (lane == 0) {
            part[wave][k] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < kResultHistWords) {
        const unsigned long long s = (unsigned long long)part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (s != 0ULL) {
            atomicAdd(&hist[threadIdx.x], s);
        }
    }
}


// ------------------------------------------------------------------------------------------
// The sub-stages of the frame decode as the reference exposes them one by one (in-place helpers of the classic
// ecc -> demodulate -> ecc call sequence), batched: one thread per frame, packed frames in, packed frames out.
//   stage 1  C0 ECC        mbe_eccImbe7200x4400C0 src/imbe/imbe7200x4400.c:424-443, mbe_eccAmbe3600C0_common
//                          src/ambe/ambe_common.c:22-46, mbe_eccImbe7100x4400C0 src/imbe/imbe7100x4400.c:100-122
//   stage 2  demodulation  mbe_demodulateImbe7200x4400Data :636-673, mbe_demodulateAmbe3600Data_common :75-100,
//                          mbe_demodulateImbe7100x4400Data :292-334  (seed = the C0 data bits AS THEY ARE in the frame)
//   stage 4  data ECC      mbe_eccImbe7200x4400Data :469-515,563-578, mbe_eccAmbe3600Data_common :127-157,
//                          mbe_eccImbe7100x4400Data :153-212 (bits in 7100 order: no conversion here)
//   stage 8  mbe_convertImbe7100to7200 src/imbe/imbe7100x4400.c:381-438 on a record (frames = records in, out = records)
// out record: parameter bits of stage 4 / 8; w[3] = corrected-error count of the stage (bits 0..7), C4 errors (16..23).
// These are slices of the frame kernels above, kept apart from them so that the hot kernels stay as they are.
// ------------------------------------------------------------------------------------------
struct BitWriter160 {
    uint32_t w[5] = {0, 0, 0, 0, 0};
    int      n = 0;
    __device__ void push(uint32_t value, int width) {   // `width` bits, first = MSB of the field
        for (int j = width - 1; j >= 0; --j, ++n) {
            w[n >> 5] |= ((value >> j) & 1u) << (31 - (n & 31));
        }
    }
    __device__ void store(uint8_t* p, int nbytes) const {
        for (int b = 0; b < nbytes; ++b) {
            p[b] = (uint8_t)(w[b >> 2] >> (24 - 8 * (b & 3)));
        }
    }
};

__device__ void convert_7100_to_7200(const mbx_tables* T, const Bits88& d, Bits88& t) {
    const int b0 = (int)(((d.hi >> 56) & 0x7eull) << 1) | (d.get(86) << 1) | d.get(87);   // bits 1..6, 86, 87
    const int K = (b0 < 208) ? (int)T->imbe_K[b0] : 12;
    t.put(87, d.get(0));
    t.put(48 + K, d.get(42));
    t.put(49 + K, d.get(43));
    for (int q = 0; q < K; ++q) {
        t.put(48 + q, d.get(44 + q));
    }
    int j = 0, k = 1;
    while (j < 87) {
        t.put(j, d.get(k));
        if (++j == 48) {
            j += K + 2;
        }
        if (++k == 42) {
            k += K + 2;
        }
    }
}

__global__ void __launch_bounds__(256)
fec_stage_kernel(int codec, int stage, const uint8_t* __restrict__ frames, size_t n, uint8_t* __restrict__ frames_out,
                 mbx_param_record* __restrict__ out, DeviceTables tabs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const mbx_tables* T = tabs.t;
    if (stage == 8) {   // records in, records out
        const uint4 r = reinterpret_cast<const uint4*>(frames)[i];
        Bits88 d, t;
        d.hi = ((uint64_t)r.x << 32) | r.y;
        d.lo = (uint64_t)r.z << 32;
        convert_7100_to_7200(T, d, t);
        reinterpret_cast<uint4*>(out)[i] = make_uint4((uint32_t)(t.hi >> 32), (uint32_t)t.hi, (uint32_t)(t.lo >> 32), 0u);
        return;
    }
    const bool ambe = codec == MBX_CODEC_AMBE3600X2450 || codec == MBX_CODEC_AMBE3600X2400;
    const int fbytes = ambe ? MBX_AMBE_FRAME_BYTES : MBX_IMBE_FRAME_BYTES;
    const uint8_t* f = frames + i * (size_t)fbytes;
    BitReader br;
    for (int k = 0; k < 5; ++k) {
        br.w[k] = 0;
    }
    for (int b = 0; b < fbytes; ++b) {
        br.w[b >> 2] |= (uint32_t)f[b] << (24 - 8 * (b & 3));
    }
    int errs = 0, c4 = 0;
    RecordWriter rw;
    BitWriter160 bw;
    uint32_t w;
    if (codec == MBX_CODEC_IMBE7200X4400) {
        const int width[8] = {23, 23, 23, 23, 15, 15, 15, 7};
        uint32_t row[8];
        for (int r = 0, pos = 0; r < 8; pos += width[r], ++r) {
            row[r] = br.take(pos, width[r]);
        }
        if (stage == 1) {
            errs = golay2312(T, row[0], row[0]);
        } else if (stage == 2) {
            PrSequence pr(row[0] >> 11);
            for (int r = 1; r < 7; ++r) {
                row[r] ^= pr.mask_for(width[r]);
            }
        } else {
            rw.push(row[0], 23, 12);
            for (int r = 1; r < 4; ++r) {
                errs += golay2312(T, row[r], w);
                rw.push(w, 23, 12);
            }
            for (int r = 4; r < 7; ++r) {
                const int e = hamming1511(T, row[r], w);
                errs += e;
                c4 = (r == 4) ? e : c4;
                rw.push(w, 15, 11);
            }
            rw.push(row[7], 7, 7);
        }
        for (int r = 0; r < 8; ++r) {
            bw.push(row[r], width[r]);
        }
    } else if (codec == MBX_CODEC_IMBE7100X4400) {
        const int width[7] = {19, 24, 23, 23, 15, 15, 23};
        uint32_t row[7];
        for (int r = 0, pos = 0; r < 7; pos += width[r], ++r) {
            row[r] = br.take(pos, width[r]);
        }
        if (stage == 1) {   // cells 1..18 are the code word's low 18 positions, the five missing ones are zeros
            errs = golay2312(T, (row[0] >> 1) & 0x3ffffu, w);
            row[0] = ((w & 0x3ffffu) << 1) | (row[0] & 1u);
        } else if (stage == 2) {
            PrSequence pr((row[0] >> 12) & 0x7fu);
            for (int r = 1; r < 6; ++r) {
                row[r] ^= pr.mask_for(width[r]);
            }
        } else {
            rw.push(row[0] >> 12, 7, 7);
            errs += golay2312(T, row[1] >> 1, w);
            rw.push(w, 23, 12);
            errs += golay2312(T, row[2], w);
            rw.push(w, 23, 12);
            errs += golay2312(T, row[3], w);
            rw.push(w, 23, 12);
            c4 = hamming1511_7100(T, row[4], w);
            errs += c4;
            rw.push(w, 15, 11);
            errs += hamming1511_7100(T, row[5], w);
            rw.push(w, 15, 11);
            rw.push(row[6], 23, 23);
        }
        for (int r = 0; r < 7; ++r) {
            bw.push(row[r], width[r]);
        }
    } else {
        uint32_t row0 = br.take(0, 24), row1 = br.take(24, 23), row2 = br.take(47, 11), row3 = br.take(58, 14);
        if (stage == 1) {
            errs = golay2312(T, row0 >> 1, w);
            row0 = (w << 1) | (row0 & 1u);
            if (errs == 0 && (__popc(row0) & 1)) {
                row0 ^= 1u;
                errs = 1;
            }
        } else if (stage == 2) {
            PrSequence pr((row0 >> 12) & 0xfffu);
            row1 ^= pr.mask_for(23);
        } else {
            rw.push(row0, 24, 12);
            errs = golay2312(T, row1, w);
            rw.push(w, 23, 12);
            rw.push(row2, 11, 11);
            rw.push(row3, 14, 14);
        }
        bw.push(row0, 24);
        bw.push(row1, 23);
        bw.push(row2, 11);
        bw.push(row3, 14);
    }
    if (frames_out) {
        bw.store(frames_out + i * (size_t)fbytes, fbytes);
    }
    if (out) {
        reinterpret_cast<uint4*>(out)[i] = make_uint4((uint32_t)(rw.hi >> 32), (uint32_t)rw.hi, (uint32_t)(rw.lo >> 32),
                                                      (uint32_t)errs | ((uint32_t)c4 << 16));
    }
}

// ------------------------------------------------------------------------------------------
// Cell arrays -> wire frames on the device: hosts that hold their bursts as the reference's
// char arrays (imbe_fr[8][23] / ambe_fr[4][24] / imbe_fr[7][24], one char per bit) can upload them as they are; this
// kernel validates every cell like the reference does (mbe_validate_bits, src/internal/mbe_result.h:18-29: the WHOLE
// array, unused cells included) and packs the rows in wire order (row r, cells width[r]-1 .. 0).
// One workgroup of 256 threads takes 32 frames: the cells are staged through LDS with coalesced 4-byte loads, then one
// thread per output byte gathers its eight cells.  status[i] = 0 or MBE_STATUS_INVALID_BITS (the frame is then packed
// from the low bit of each cell and must be ignored by the caller, like the reference's "nothing is written").
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pack_cells_kernel(int codec, const char* __restrict__ cells, size_t n, uint8_t* __restrict__ packed, int32_t* __restrict__ status) {
    constexpr int kFrames = 32;
    __shared__ uint32_t stage[kFrames * 184 / 4];
    __shared__ int bad[kFrames];
    const bool ambe = codec == MBX_CODEC_AMBE3600X2450 || codec == MBX_CODEC_AMBE3600X2400;
    const int ncell = codec == MBX_CODEC_IMBE7200X4400 ? 184 : (ambe ? 96 : 168);
    const int stride = codec == MBX_CODEC_IMBE7200X4400 ? 23 : 24;
    const int fbytes = ambe ? MBX_AMBE_FRAME_BYTES : MBX_IMBE_FRAME_BYTES;
    const int rows = codec == MBX_CODEC_IMBE7200X4400 ? 8 : (ambe ? 4 : 7);
    const int w0[8] = {23, 23, 23, 23, 15, 15, 15, 7}, w1[8] = {24, 23, 11, 14, 0, 0, 0, 0}, w2[8] = {19, 24, 23, 23, 15, 15, 23, 0};
    const size_t first = (size_t)blockIdx.x * kFrames;
    const int here = (int)((n - first) < (size_t)kFrames ? (n - first) : (size_t)kFrames);
    if (threadIdx.x < kFrames) {
        bad[threadIdx.x] = 0;
    }
    __syncthreads();
    const int words = here * ncell / 4;   // ncell is a multiple of 4 for all three shapes
    const uint32_t* src = reinterpret_cast<const uint32_t*>(cells + first * (size_t)ncell);
    for (int i = (int)threadIdx.x; i < words; i += 256) {
        const uint32_t v = src[i];
        stage[i] = v;
        if (v & 0xfefefefeu) {   // a cell outside {0, 1}
            bad[(i * 4) / ncell] = 1;   // the four cells of a word may straddle two frames: mark both
            bad[(i * 4 + 3) / ncell] = 1;
        }
    }
    __syncthreads();
    const char* c = reinterpret_cast<const char*>(stage);
    for (int o = (int)threadIdx.x; o < here * fbytes; o += 256) {
        const int f = o / fbytes, b = o % fbytes;
        uint32_t byte = 0;
        for (int k = 0; k < 8; ++k) {
            int pos = 8 * b + k, r = 0;   // wire position -> (row, cell)
            for (; r < rows; ++r) {
                const int w = codec == MBX_CODEC_IMBE7200X4400 ? w0[r] : (ambe ? w1[r] : w2[r]);
                if (pos < w) {
                    byte |= (uint32_t)(c[f * ncell + r * stride + (w - 1 - pos)] & 1) << (7 - k);
                    break;
                }
                pos -= w;
            }
        }
        packed[(first + (size_t)f) * (size_t)fbytes + (size_t)b] = (uint8_t)byte;
    }
    if ((int)threadIdx.x < here && status) {
        status[first + threadIdx.x] = bad[threadIdx.x] ? MBE_STATUS_INVALID_BITS : 0;
    }
}

// Code-word level ECC (the public per-word helpers, batched): kind 0 = Golay(23,12), 1 = Hamming(15,11),
// 2 = Hamming(15,11) with the IMBE 7100x4400 bit mapping (mbe_7100x4400hamming1511, src/ecc/ecc.c:422-464).
//   ref mbe_golay2312 / mbe_checkGolayBlock src/ecc/ecc.c:221-301, mbe_hamming1511 src/ecc/ecc.c:366-408
__global__ void __launch_bounds__(256)
ecc_words_kernel(int kind, const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ out, int32_t* __restrict__ errs,
                 DeviceTables tabs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    uint32_t fixed;
    const int e = (kind == 0)   ? golay2312(tabs.t, in[i] & 0x7fffffu, fixed)
                  : (kind == 1) ? hamming1511(tabs.t, in[i] & 0x7fffu, fixed)
                                : hamming1511_7100(tabs.t, in[i] & 0x7fffu, fixed);
    out[i] = fixed;
    if (errs) {
        errs[i] = e;
    }
}


// ------------------------------------------------------------------------------------------
// Soft-decision front end.  ONE WAVEFRONT PER FRAME (or per code word).
//
//   ref src/ecc/ecc.c:36-63 (cost, tie rules), :65-80 (Golay encoder), :303-357 (mbe_golay2312Soft),
//       :128-215 (Hamming candidates, mbe_hamming1511Soft);
//       src/imbe/imbe7200x4400.c:445-459, :517-560, :675-707, :746-778;
//       src/ambe/ambe_common.c:48-73, :102-124, :159-190; src/ambe/ambe3600x2450.c:684-714
//
// The reference scores all 4096 (2048) code words against the 23 (15) soft bits -- the cost of a
// candidate is the sum of the reliabilities of the positions where it disagrees with the hard
// decisions -- scanning data words in ascending order and replacing the best only when the new one
// is strictly better under (lower cost) > (equals the hard decoder's output) > (fewer differing
// bits).  That scan returns the minimum of the packed key
//     cost << 17 | !matches_hard << 16 | differing_bits << 12 | data          (Golay; cost <= 5865)
// so here every lane scores 64 (32) candidates with three (two) LDS byte-tables of partial costs and
// the wave takes the minimum key.  Integer work: results are bit-exact.
// ------------------------------------------------------------------------------------------
struct SoftScratch {
    // 4,992 B: eight waves per SIMD fit the CU's 160 KB (the allocation granule is 1,280 B; <= 5,120 B is what it takes)
    uint16_t parity[2048];   // Golay: cost of parity pattern p at index rot(p) (golay_soft_wave); Hamming: 16 key dwords
    uint2    round[64];      // per-round constants: (table offset, key contribution) of the high bits j of u
    uint8_t  bit[192];       // the frame's hard decisions
    uint8_t  rel[192];       // and reliabilities (read wave-uniformly: LDS broadcasts, no v_readlane)
};

// Minimum over the wave, returned wave-uniform in scalar registers (what follows it runs on the scalar unit):
// four DPP steps inside each row of 16, the four row minima by v_readlane.
template <int kCtrl>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, kCtrl, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    uint32_t o = dpp_u32<kDppXor1>(v);
    v = o < v ? o : v;
    o = dpp_u32<kDppXor2>(v);
    v = o < v ? o : v;
    o = dpp_u32<kDppHalfMirror>(v);
    v = o < v ? o : v;
    o = dpp_u32<kDppMirror>(v);
    v = o < v ? o : v;
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
    return a < b ? a : b;
}

__device__ __forceinline__ uint32_t rl(uint32_t v, int lane_index) {   // wave-uniform lane index
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane_index);
}

// How the candidate search is organised (both codes).  Let hd / hp be the data / parity part of the
// hard decisions.  The candidates are walked in u = data ^ hd instead of data:
//   * the data-part cost and the number of differing data bits depend on u only and split into a
//     lane part (six low bits of u = lane) and a round part (high bits of u = j);
//   * the code is linear, parity(data) = parity(u) ^ parity(hd), so the parity-part cost is a table
//     look-up at parity_lo(lane) ^ parity_hi(j) ^ s with s = parity(hd) ^ hp -- the table holds the
//     cost already shifted into its key position;
//   * data = u ^ hd splits into lane and round bits as well.
// Every field of the key is therefore a SUM of a per-lane constant, a per-round constant (held one per
// lane, fetched with v_readlane) and one LDS word: per candidate one XOR, one ds_read_b32, one
// v_add3 and one v_min.  All candidates are scored as "does not match the hard decoder"; the single
// one that does is re-scored after the loop with that bit cleared (it can only win then).

// Soft Golay(23,12).  `hard` = the 23 hard decisions (bit j = cell j, wave-uniform), lane j holds
// reliability j.  Returns the chosen data bits over the HARD parity bits (ecc.c:354-356); `diffs` =
// data-bit differences between the hard decisions and the chosen code word (the return value of
// mbe_golay2312Soft).  Key: cost << 17 | !matches_hard << 16 | differing data bits << 12 | data.
// The parity-cost table holds 16-bit entries, pattern p at index rot(p) = p with its low seven bits rotated left by one:
// the six parity bits that are an invertible function of the lane's six data bits then select the LDS bank (index bits
// 1..6, 64 banks of four bytes), so the 64 lanes of a ds_read_u16 never collide.  rot is linear over XOR: the
// generator rows come pre-rotated from the host (DerivedTables::golay_rot), the hard parity bits are rotated here.
__device__ __forceinline__ uint32_t rot_parity(uint32_t p) { return ((p & 0x3fu) << 1) | ((p >> 6) & 1u) | (p & 0x780u); }

// What depends on the lane alone (not on the block): the rotated parity patterns of the lane's six low / six high
// candidate bits.  Computed once per frame.
struct SoftLane {
    uint32_t golay_lo, golay_hi;   // Golay(23,12)
    uint32_t ham_lo, ham_hi;       // Hamming(15,11), 4-bit parity patterns
};

__device__ __forceinline__ void golay_lane_patterns(const DeviceTables& tabs, int lane, SoftLane& L) {
    const uint32_t* grot = tabs.d->golay_rot;
    L.golay_lo = 0;
    L.golay_hi = 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const uint32_t bit = (uint32_t)(lane >> b) & 1u;
        L.golay_lo ^= bit ? grot[11 - b] : 0u;   // row i <-> data bit 11 - i
        L.golay_hi ^= bit ? grot[5 - b] : 0u;
    }
}

__device__ uint32_t golay_soft_wave(const DeviceTables& tabs, const SoftLane& L, uint32_t hard, int first, SoftScratch& S, int lane,
                                    int& diffs) {
    uint32_t hard_fixed;
    (void)golay2312(tabs.t, hard, hard_fixed);
    const uint32_t hd = hard >> 11, hp = hard & 0x7ffu;
    const uint8_t* rel = &S.rel[first];   // reliability of cell j of the block, read wave-uniformly
    const uint32_t* grot = tabs.d->golay_rot;
    // per-lane pieces: bit b of the lane index selects position ...
    const uint32_t par_lo = L.golay_lo, par_hi = L.golay_hi;   // rotated parity of data bits 0..5 / 6..11 (pattern = lane)
    uint32_t par_hd = 0;                            // ... and of hd (wave-uniform: scalar unit)
    uint32_t a_lo = 0, a_hi = 0;                    // data-part cost: cells 11..16 / 17..22
    uint32_t b_lo = 0, b_hi = 0;                    // parity-part cost: cells 0..5 / 6..10 (pattern = lane, < 32)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const uint32_t bit = (uint32_t)(lane >> b) & 1u;
        par_hd ^= ((hd >> b) & 1u) ? grot[11 - b] : 0u;
        par_hd ^= ((hd >> (b + 6)) & 1u) ? grot[5 - b] : 0u;
        a_lo = __umul24(bit, (uint32_t)rel[11 + b]) + a_lo;   // one v_mad_u32_u24 each
        a_hi = __umul24(bit, (uint32_t)rel[17 + b]) + a_hi;
        b_lo = __umul24(bit, (uint32_t)rel[b]) + b_lo;
        if (b < 5) {
            b_hi = __umul24(bit, (uint32_t)rel[6 + b]) + b_hi;
        }
    }
    const uint32_t s = par_hd ^ rot_parity(hp);
    wave_lds_sync();
    // parity table: pattern lane + 64 k sits at rot(...) = 2 lane + (k & 1) + 128 (k >> 1), i.e. the patterns k = 2 q
    // and 2 q + 1 of a lane share dword lane + 64 q -- written as one packed sum (no carry: a cost is < 2^12)
    uint16_t* const stage = reinterpret_cast<uint16_t*>(S.round);   // b_hi of pattern k, k < 32, read back in pairs
    stage[lane] = (uint16_t)b_hi;
    wave_lds_sync();
    {
        const uint32_t* pairs = reinterpret_cast<const uint32_t*>(S.round);
        uint32_t* table32 = reinterpret_cast<uint32_t*>(S.parity);
        const uint32_t b_lo2 = b_lo | (b_lo << 16);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            table32[lane + 64 * q] = b_lo2 + pairs[q];
        }
    }
    wave_lds_sync();   // the staging words become the round constants below
    const uint32_t addr_lane = par_lo << 1;                                         // byte offsets into S.parity
    const uint32_t addr_round = (par_hi ^ s) << 1;                                  // for round j = lane
    const uint32_t key_lane = (a_lo << 17) + 0x10000u + ((uint32_t)__popc(lane) << 12) + ((uint32_t)lane ^ (hd & 63u));
    const uint32_t key_round = (a_hi << 17) + ((uint32_t)__popc(lane) << 12) + ((((uint32_t)lane ^ (hd >> 6)) & 63u) << 6);
    S.round[lane] = make_uint2(addr_round, key_round);   // read back wave-uniformly: LDS broadcasts, no VALU
    wave_lds_sync();
    const char* table = reinterpret_cast<const char*>(S.parity);
    auto score_round = [&](int j, uint32_t& best) {   // the lane's candidate of round j (u_hi = j); j is wave-uniform
        const uint2 r = S.round[j];
        const uint32_t val = *reinterpret_cast<const uint16_t*>(table + (addr_lane ^ r.x));
        const uint32_t key = (val << 17) + r.y;   // v_lshl_add_u32; the lane's own part is added after the loop
        best = key < best ? key : best;
    };
    // Exact pruning.  The cost of a candidate is at least the cost of the high data bits it flips, a_hi(j), which is the
    // same for the 64 candidates of round j.  Two candidates give an upper bound on the winning cost before the search:
    // the hard decoder's own code word (scored anyway, for its cleared tie bit) and the best candidate of round 0
    // (no high data bit flipped).  A round with a_hi(j) above that bound cannot hold the minimum -- keys order by cost
    // first, and rounds that merely TIE the bound are kept, so the reference's tie rules see every candidate they could
    // pick.  On noisy random words 10 to 30 of the 64 rounds survive, on clean code words a handful.
    uint32_t key_hard;
    {   // the candidate whose data equals the hard decoder's output
        const uint32_t ut = (hard_fixed >> 11) ^ hd;
        const int lt = (int)(ut & 63u), jt = (int)(ut >> 6);
        const uint32_t val = *reinterpret_cast<const uint16_t*>(table + (rl(addr_lane, lt) ^ rl(addr_round, jt)));
        key_hard = (val << 17) + rl(key_lane, lt) + rl(key_round, jt) - 0x10000u;
    }
    uint32_t best = 0xffffffffu;
    score_round(0, best);
    const uint32_t round0 = wave_min_u32(best + key_lane);
    const uint32_t limit = (round0 < key_hard ? round0 : key_hard) >> 17;   // a cost no winner can exceed
    unsigned long long live = __ballot(a_hi <= limit) & ~1ULL;           // lane j speaks for round j
    while (live) {   // four rounds per trip (the two LDS reads of a round are a dependent pair: give the scheduler several)
        const int j0 = __ffsll((long long)live) - 1;
        live &= live - 1;
        const int j1 = live ? (__ffsll((long long)live) - 1) : j0;   // a repeated round changes nothing: min is idempotent
        live &= live - 1;
        const int j2 = live ? (__ffsll((long long)live) - 1) : j0;
        live &= live - 1;
        const int j3 = live ? (__ffsll((long long)live) - 1) : j0;
        live &= live - 1;
        score_round(j0, best);
        score_round(j1, best);
        score_round(j2, best);
        score_round(j3, best);
    }
    best += key_lane;
    best = wave_min_u32(best);
    best = key_hard < best ? key_hard : best;
    diffs = (int)((best >> 12) & 0xfu);
    return ((best & 0xfffu) << 11) | hp;
}

// Soft Hamming(15,11): returns the chosen code word, `diffs` = differing bits over all 15 positions.
// Data bit i sits at cell kHamData[i], parity bit q at cell kHamParity[q] (ecc.c:128-131).
// Key: cost << 16 | !matches_hard << 15 | differing bits << 11 | data.
// k7100: the IMBE 7100x4400 bit mapping (data at cells 4..14, parity at 0..3), ecc.c:130-131
template <bool k7100>
__device__ __forceinline__ void hamming_lane_patterns(const DeviceTables& tabs, int lane, SoftLane& L) {
    constexpr int kHamParity[4] = {0, 1, k7100 ? 2 : 3, k7100 ? 3 : 7};
    const uint32_t* basis = k7100 ? tabs.d->ham7100_basis : tabs.d->ham_basis;
    auto gather_parity = [&](uint32_t cw) {
        uint32_t q = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            q |= ((cw >> kHamParity[i]) & 1u) << i;
        }
        return q;
    };
    L.ham_lo = 0;
    L.ham_hi = 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const uint32_t bit = (uint32_t)(lane >> b) & 1u;
        L.ham_lo ^= bit ? gather_parity(basis[b]) : 0u;
        if (b < 5) {
            L.ham_hi ^= bit ? gather_parity(basis[6 + b]) : 0u;
        }
    }
}

template <bool k7100>
__device__ uint32_t hamming_soft_wave(const DeviceTables& tabs, const SoftLane& L, uint32_t hard, int first, SoftScratch& S, int lane,
                                      int& diffs) {
    constexpr int kHamData[11] = {k7100 ? 4 : 2, k7100 ? 5 : 4, k7100 ? 6 : 5, k7100 ? 7 : 6, 8, 9, 10, 11, 12, 13, 14};
    constexpr int kHamParity[4] = {0, 1, k7100 ? 2 : 3, k7100 ? 3 : 7};
    const uint32_t* basis = k7100 ? tabs.d->ham7100_basis : tabs.d->ham_basis;
    uint32_t hard_fixed;
    if (k7100) {
        (void)hamming1511_7100(tabs.t, hard, hard_fixed);
    } else {
        (void)hamming1511(tabs.t, hard, hard_fixed);
    }
    const uint8_t* rel = &S.rel[first];   // reliability of cell j of the block, read wave-uniformly
    auto gather_data = [&](uint32_t cw) {
        uint32_t d = 0;
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            d |= ((cw >> kHamData[i]) & 1u) << i;
        }
        return d;
    };
    auto gather_parity = [&](uint32_t cw) {
        uint32_t q = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            q |= ((cw >> kHamParity[i]) & 1u) << i;
        }
        return q;
    };
    const uint32_t hd = gather_data(hard), hp = gather_parity(hard);
    const uint32_t par_lo = L.ham_lo, par_hi = L.ham_hi;
    uint32_t par_hd = 0, a_lo = 0, a_hi = 0, p_cost = 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const uint32_t bit = (uint32_t)(lane >> b) & 1u;
        par_hd ^= ((hd >> b) & 1u) ? gather_parity(basis[b]) : 0u;
        a_lo = __umul24(bit, (uint32_t)rel[kHamData[b]]) + a_lo;
        if (b < 5) {
            par_hd ^= ((hd >> (b + 6)) & 1u) ? gather_parity(basis[6 + b]) : 0u;
            a_hi = __umul24(bit, (uint32_t)rel[kHamData[6 + b]]) + a_hi;
        }
        if (b < 4) {
            p_cost = __umul24(bit, (uint32_t)rel[kHamParity[b]]) + p_cost;
        }
    }
    const uint32_t s = par_hd ^ hp;
    wave_lds_sync();
    if (lane < 16) {   // parity pattern = lane: its cost and its differing-bit count, in key position
        reinterpret_cast<uint32_t*>(S.parity)[lane] = (p_cost << 16) + ((uint32_t)__popc(lane) << 11);
    }
    const uint32_t addr_lane = par_lo << 2, addr_round = (par_hi ^ s) << 2;
    const uint32_t key_lane = (a_lo << 16) + 0x8000u + ((uint32_t)__popc(lane) << 11) + ((uint32_t)lane ^ (hd & 63u));
    const uint32_t key_round = (a_hi << 16) + ((uint32_t)__popc(lane & 31) << 11) + ((((uint32_t)lane ^ (hd >> 6)) & 31u) << 6);
    S.round[lane] = make_uint2(addr_round, key_round);
    wave_lds_sync();
    const char* table = reinterpret_cast<const char*>(S.parity);
    auto score_round = [&](int j, uint32_t& best) {
        const uint2 r = S.round[j];
        const uint32_t val = *reinterpret_cast<const uint32_t*>(table + (addr_lane ^ r.x));
        const uint32_t key = val + r.y;   // the lane's own part is added after the loop
        best = key < best ? key : best;
    };
    // exact pruning of whole rounds by the cost of their high data bits, as in golay_soft_wave
    uint32_t key_hard;
    {   // the candidate that equals the hard decoder's output
        const uint32_t ut = gather_data(hard_fixed) ^ hd;
        const int lt = (int)(ut & 63u), jt = (int)(ut >> 6);
        const uint32_t val = *reinterpret_cast<const uint32_t*>(table + (rl(addr_lane, lt) ^ rl(addr_round, jt)));
        key_hard = val + rl(key_lane, lt) + rl(key_round, jt) - 0x8000u;
    }
    uint32_t best = 0xffffffffu;
    score_round(0, best);
    const uint32_t round0 = wave_min_u32(best + key_lane);
    const uint32_t limit = (round0 < key_hard ? round0 : key_hard) >> 16;
    unsigned long long live = __ballot(lane < 32 && a_hi <= limit) & ~1ULL;   // lane j < 32 speaks for round j
    while (live) {
        const int j0 = __ffsll((long long)live) - 1;
        live &= live - 1;
        const int j1 = live ? (__ffsll((long long)live) - 1) : j0;
        live &= live - 1;
        score_round(j0, best);
        score_round(j1, best);
    }
    best += key_lane;
    best = wave_min_u32(best);
    best = key_hard < best ? key_hard : best;
    diffs = (int)((best >> 11) & 0xfu);
    const uint32_t data = best & 0x7ffu;
    uint32_t cw = 0;
#pragma unroll
    for (int b = 0; b < 11; ++b) {
        cw ^= ((data >> b) & 1u) ? basis[b] : 0u;
    }
    return cw;
}

// cells [first, first + width) of the frame as a block: hard word (bit j = cell first + j, optionally
// demodulated with the PR sequence, whose bit for cell j is number k_first + (width - 1 - j)) and the
// reliability of cell j in lane j
__device__ __forceinline__ uint32_t soft_block(const SoftScratch& S, const DeviceTables& tabs, int first, int width, int k_first,
                                               uint32_t pr_x0, int lane, int& rel_lane) {
    uint32_t bit = 0;
    rel_lane = 0;
    if (lane < width) {
        bit = S.bit[first + lane];
        rel_lane = (int)S.rel[first + lane];
        if (k_first > 0) {
            const int k = k_first + (width - 1 - lane);
            const uint32_t x = (tabs.d->pr_mul[k] * pr_x0 + tabs.d->pr_add[k]) & 0xffffu;
            bit ^= x >> 15;
        }
    }
    return (uint32_t)__ballot(bit != 0u);
}

__device__ __forceinline__ void load_soft_cells(SoftScratch& S, const mbe_soft_bit* frame, int count, int lane) {
    const uint16_t* src = reinterpret_cast<const uint16_t*>(frame);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int idx = lane + 64 * k;
        if (idx < count) {
            const uint32_t v = src[idx];
            S.bit[idx] = (uint8_t)(v & 1u);   // the reference masks the hard decision with & 1
            S.rel[idx] = (uint8_t)(v >> 8);
        }
    }
    wave_lds_sync();
}

__global__ void __launch_bounds__(64, 8)
fec_imbe7200x4400_soft_kernel(const mbe_soft_bit* __restrict__ soft, size_t n, mbx_param_record* __restrict__ out,
                              DeviceTables tabs) {
    __shared__ SoftScratch S;
    const size_t i = blockIdx.x;
    if (i >= n) {
        return;
    }
    const int lane = lane_id();
    load_soft_cells(S, soft + i * MBX_IMBE_SOFT_BITS, MBX_IMBE_SOFT_BITS, lane);
    SoftLane L;
    golay_lane_patterns(tabs, lane, L);
    hamming_lane_patterns<false>(tabs, lane, L);
    int rel, diffs;
    uint32_t row[8];
    {
        const uint32_t hard = soft_block(S, tabs, 0, 23, 0, 0u, lane, rel);
        row[0] = golay_soft_wave(tabs, L, hard, 0, S, lane, diffs);
    }
    const int c0 = diffs;
    const uint32_t x0 = (16u * (row[0] >> 11)) & 0xffffu;
    int prot = 0, c4 = 0, k = 1;
#pragma unroll 1
    for (int r = 1; r < 4; ++r) {
        const uint32_t hard = soft_block(S, tabs, 23 * r, 23, k, x0, lane, rel);
        row[r] = golay_soft_wave(tabs, L, hard, 23 * r, S, lane, diffs);
        prot += diffs;
        k += 23;
    }
#pragma unroll 1
    for (int r = 4; r < 7; ++r) {
        const uint32_t hard = soft_block(S, tabs, 23 * r, 15, k, x0, lane, rel);
        row[r] = hamming_soft_wave<false>(tabs, L, hard, 23 * r, S, lane, diffs);
        prot += diffs;
        if (r == 4) {
            c4 = diffs;
        }
        k += 15;
    }
    row[7] = soft_block(S, tabs, 23 * 7, 7, 0, 0u, lane, rel);
    if (lane == 0) {
        RecordWriter rw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rw.push(row[r], 23, 12);
        }
#pragma unroll
        for (int r = 4; r < 7; ++r) {
            rw.push(row[r], 15, 11);
        }
        rw.push(row[7], 7, 7);
        *reinterpret_cast<uint4*>(&out[i]) =
            make_uint4((uint32_t)(rw.hi >> 32), (uint32_t)rw.hi, (uint32_t)(rw.lo >> 32),
                       (uint32_t)c0 | ((uint32_t)prot << 8) | ((uint32_t)c4 << 16)
                           | ((MBE_PROCESS_FLAG_SOFT_INPUT | MBE_PROCESS_FLAG_C0_VALID | MBE_PROCESS_FLAG_C4_VALID) << 24));
    }
}

__global__ void __launch_bounds__(64, 8)
fec_ambe3600x2450_soft_kernel(const mbe_soft_bit* __restrict__ soft, size_t n, mbx_param_record* __restrict__ out,
                              DeviceTables tabs) {
    __shared__ SoftScratch S;
    const size_t i = blockIdx.x;
    if (i >= n) {
        return;
    }
    const int lane = lane_id();
    load_soft_cells(S, soft + i * MBX_AMBE_SOFT_BITS, MBX_AMBE_SOFT_BITS, lane);
    SoftLane L;
    golay_lane_patterns(tabs, lane, L);
    L.ham_lo = L.ham_hi = 0;
    int rel, diffs;
    // C0: cells 1..23 of row 0 are the Golay block, cell 0 the overall parity bit
    uint32_t hard = soft_block(S, tabs, 1, 23, 0, 0u, lane, rel);
    const uint32_t cw = golay_soft_wave(tabs, L, hard, 1, S, lane, diffs);
    int c0 = diffs;
    uint32_t row0 = (cw << 1) | (uint32_t)S.bit[0];
    if (c0 == 0 && (__popc(row0) & 1)) {
        row0 ^= 1u;
        c0 = 1;
    }
    const uint32_t x0 = (16u * ((row0 >> 12) & 0xfffu)) & 0xffffu;
    hard = soft_block(S, tabs, 24, 23, 1, x0, lane, rel);
    const uint32_t row1 = golay_soft_wave(tabs, L, hard, 24, S, lane, diffs);
    const int prot = diffs;
    const uint32_t row2 = soft_block(S, tabs, 48, 11, 0, 0u, lane, rel);
    const uint32_t row3 = soft_block(S, tabs, 72, 14, 0, 0u, lane, rel);
    if (lane == 0) {
        RecordWriter rw;
        rw.push(row0, 24, 12);
        rw.push(row1, 23, 12);
        rw.push(row2, 11, 11);
        rw.push(row3, 14, 14);
        *reinterpret_cast<uint4*>(&out[i]) =
            make_uint4((uint32_t)(rw.hi >> 32), (uint32_t)rw.hi, (uint32_t)(rw.lo >> 32),
                       (uint32_t)c0 | ((uint32_t)prot << 8) | ((MBE_PROCESS_FLAG_SOFT_INPUT | MBE_PROCESS_FLAG_C0_VALID) << 24));
    }
}

// IMBE 7100x4400 soft frames, mbe_soft_bit[7][24] (ref src/imbe/imbe7100x4400.c:124-150, 214-274, 336-378, 481-525):
// C0 = cells 1..18 of row 0 completed by five certain zeros, 7-bit demodulation seed, C1 = cells 1..23 of
// row 1, the 7100 Hamming mapping on rows 4/5, then mbe_convertImbe7100to7200.
__global__ void __launch_bounds__(64, 8)
fec_imbe7100x4400_soft_kernel(const mbe_soft_bit* __restrict__ soft, size_t n, mbx_param_record* __restrict__ out,
                              DeviceTables tabs) {
    __shared__ SoftScratch S;
    const size_t i = blockIdx.x;
    if (i >= n) {
        return;
    }
    const int lane = lane_id();
    load_soft_cells(S, soft + i * MBX_IMBE7100_SOFT_BITS, MBX_IMBE7100_SOFT_BITS, lane);
    SoftLane L;
    golay_lane_patterns(tabs, lane, L);
    hamming_lane_patterns<true>(tabs, lane, L);
    // C0 is cells 1..18 of row 0 completed by five certain zeros, mbe_softBitFromHard(0, 255): they go into
    // the (unused) cells 19..23 of the row, so the block is simply cells 1..23
    if (lane >= 19 && lane < 24) {
        S.bit[lane] = 0;
        S.rel[lane] = 255;
    }
    wave_lds_sync();
    int rel, diffs;
    uint32_t hard = soft_block(S, tabs, 1, 23, 0, 0u, lane, rel);
    uint32_t w = golay_soft_wave(tabs, L, hard, 1, S, lane, diffs);
    const int c0 = diffs;
    const uint32_t row0 = ((w & 0x3ffffu) << 1) | (uint32_t)S.bit[0];
    const uint32_t x0 = (16u * ((row0 >> 12) & 0x7fu)) & 0xffffu;
    // demodulation bit numbers: row 1 (24 cells) uses 1..24 with cell j <- 1 + (23 - j); its Golay block is
    // cells 1..23, i.e. a 23-wide block whose cell c uses number 1 + (22 - c)
    int prot = 0, c4 = 0;
    Bits88 d;
    int at = 0;
    auto push = [&](uint32_t value, int width, int count) {   // top `count` bits of a `width`-bit value
        for (int q = 0; q < count; ++q) {
            d.put(at++, (int)((value >> (width - 1 - q)) & 1u));
        }
    };
    push(row0 >> 12, 7, 7);
    hard = soft_block(S, tabs, 24 + 1, 23, 1, x0, lane, rel);
    w = golay_soft_wave(tabs, L, hard, 24 + 1, S, lane, diffs);
    prot += diffs;
    push(w, 23, 12);
    int k = 25;
#pragma unroll 1
    for (int r = 2; r < 4; ++r) {
        hard = soft_block(S, tabs, 24 * r, 23, k, x0, lane, rel);
        w = golay_soft_wave(tabs, L, hard, 24 * r, S, lane, diffs);
        prot += diffs;
        push(w, 23, 12);
        k += 23;
    }
#pragma unroll 1
    for (int r = 4; r < 6; ++r) {
        hard = soft_block(S, tabs, 24 * r, 15, k, x0, lane, rel);
        w = hamming_soft_wave<true>(tabs, L, hard, 24 * r, S, lane, diffs);
        prot += diffs;
        if (r == 4) {
            c4 = diffs;
        }
        push(w, 15, 11);
        k += 15;
    }
    push(soft_block(S, tabs, 24 * 6, 23, 0, 0u, lane, rel), 23, 23);
    if (lane == 0) {
        Bits88 t;
        const int b0 = (int)(((d.hi >> 56) & 0x7eull) << 1) | (d.get(86) << 1) | d.get(87);
        const int K = (b0 < 208) ? (int)tabs.t->imbe_K[b0] : 12;
        t.put(87, d.get(0));
        t.put(48 + K, d.get(42));
        t.put(49 + K, d.get(43));
        for (int q = 0; q < K; ++q) {
            t.put(48 + q, d.get(44 + q));
        }
        int j = 0, kk = 1;
        while (j < 87) {
            t.put(j, d.get(kk));
            if (++j == 48) {
                j += K + 2;
            }
            if (++kk == 42) {
                kk += K + 2;
            }
        }
        *reinterpret_cast<uint4*>(&out[i]) =
            make_uint4((uint32_t)(t.hi >> 32), (uint32_t)t.hi, (uint32_t)(t.lo >> 32),
                       (uint32_t)c0 | ((uint32_t)prot << 8) | ((uint32_t)c4 << 16)
                           | ((MBE_PROCESS_FLAG_SOFT_INPUT | MBE_PROCESS_FLAG_C0_VALID | MBE_PROCESS_FLAG_C4_VALID) << 24));
    }
}

// Code-word level soft ECC, batched: kind 0 = Golay (23 soft bits per block), 1 = Hamming (15), 2 = Hamming
// with the IMBE 7100x4400 bit mapping (mbe_7100x4400hamming1511Soft).
// out = corrected word in the cell order of the hard helpers, errs = the reference's return value.
__global__ void __launch_bounds__(64, 8)
ecc_soft_words_kernel(int kind, const mbe_soft_bit* __restrict__ in, size_t n, uint32_t* __restrict__ out, int32_t* __restrict__ errs,
                      DeviceTables tabs) {
    __shared__ SoftScratch S;
    const size_t i = blockIdx.x;
    if (i >= n) {
        return;
    }
    const int lane = lane_id();
    const int width = (kind == 0) ? 23 : 15;
    load_soft_cells(S, in + i * (size_t)width, width, lane);
    SoftLane L;
    golay_lane_patterns(tabs, lane, L);
    if (kind == 2) {
        hamming_lane_patterns<true>(tabs, lane, L);
    } else {
        hamming_lane_patterns<false>(tabs, lane, L);
    }
    int rel, diffs;
    const uint32_t hard = soft_block(S, tabs, 0, width, 0, 0u, lane, rel);
    const uint32_t w = (kind == 0)   ? golay_soft_wave(tabs, L, hard, 0, S, lane, diffs)
                       : (kind == 1) ? hamming_soft_wave<false>(tabs, L, hard, 0, S, lane, diffs)
                                     : hamming_soft_wave<true>(tabs, L, hard, 0, S, lane, diffs);
    if (lane == 0) {
        out[i] = w;
        if (errs) {
            errs[i] = diffs;
        }
    }
}

}  // namespace mbx
