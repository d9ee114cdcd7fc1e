// mbx_fec.hip -- FEC stage: wire frames -> parameter records.  One thread per frame; frames
// are independent here (no stream state), so the grid is simply ceil(n / 256) workgroups.
//
// Replaces (ref = arancormonk/mbelib-neo v2.0.0):
//   IMBE  src/imbe/imbe7200x4400.c:424-443 (C0 Golay), :636-673 (PR demodulation),
//         :469-515 (3x Golay + 3x Hamming + raw row), :709-744 (frame decode + status)
//   AMBE  src/ambe/ambe_common.c:22-46 (C0 Golay + overall parity), :75-100 (demodulation),
//         :127-157 (Golay on C1, raw C2/C3); src/ambe/ambe3600x2450.c:649-682
//   ECC   src/ecc/ecc.c:221-301 (Golay(23,12) by syndrome table), :366-408 (Hamming(15,11))
// Integer work only: results are bit-exact.
#include "mbx_device.h"

namespace mbx {

struct BitReader {            // 160 bits, big-endian bit order
    uint32_t w[5];
    __device__ uint32_t take(int pos, int width) const {   // bits [pos, pos+width), first = MSB
        const int wi = pos >> 5, sh = pos & 31;
        uint64_t two = ((uint64_t)w[wi] << 32) | (uint64_t)(wi + 1 < 5 ? w[wi + 1] : 0u);
        return (uint32_t)((two << sh) >> (64 - width));
    }
};

__device__ __forceinline__ uint32_t load_be16(const uint8_t* p) {
    const uint16_t v = *reinterpret_cast<const uint16_t*>(p);   // frames start on even addresses
    return (uint32_t)((v >> 8) | ((v & 0xffu) << 8));
}

// Golay(23,12): cw bit j = cell j.  Data bits 22..11, parity 10..0 (parity passes through).
__device__ __forceinline__ int golay2312(const mbx_tables* T, uint32_t cw, uint32_t& fixed) {
    uint32_t expect = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        expect ^= (cw & (0x400000u >> i)) ? (uint32_t)T->golay_gen[i] : 0u;
    }
    const uint32_t fix = T->golay_matrix[expect ^ (cw & 0x7ffu)];
    fixed = cw ^ (fix << 11);
    return __popc(fix);
}

__device__ __forceinline__ int hamming1511(const mbx_tables* T, uint32_t cw, uint32_t& fixed) {
    int syndrome = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        syndrome |= (__popc(cw & (uint32_t)T->hamming_gen[i]) & 1) << i;
    }
    fixed = syndrome ? (cw ^ (uint32_t)T->hamming_fix[syndrome]) : cw;
    return syndrome != 0;
}

// The demodulation sequence: x0 = 16*seed, x_k = 173*x_{k-1} + 13849 mod 2^16, bit_k = x_k >> 15.
// `mask_for(width)` returns the next `width` bits, first bit aligned to bit width-1.
struct PrSequence {
    uint32_t x;
    __device__ explicit PrSequence(uint32_t seed12) : x((16u * seed12) & 0xffffu) {}
    __device__ uint32_t mask_for(int width) {
        uint32_t m = 0;
        for (int j = 0; j < width; ++j) {
            x = (173u * x + 13849u) & 0xffffu;
            m = (m << 1) | (x >> 15);
        }
        return m;
    }
};

struct RecordWriter {
    uint64_t hi = 0, lo = 0;   // 128-bit shift register, only the first 96 bits are used
    int      n = 0;
    __device__ void push(uint32_t value, int width, int count) {   // top `count` bits of a `width`-bit value
        const uint32_t v = (value >> (width - count)) & ((1u << count) - 1u);
        // append below the bits written so far
        const int pos = n;            // bit index of the first new bit
        n += count;
        // place so that record bit i lives at (127 - i)
        const int shift = 128 - pos - count;
        if (shift >= 64) {
            hi |= (uint64_t)v << (shift - 64);
        } else if (shift + count <= 64) {
            lo |= (uint64_t)v << shift;
        } else {                       // straddles the two halves
            hi |= (uint64_t)v >> (64 - shift);
            lo |= (uint64_t)v << shift;
        }
    }
};

__global__ void __launch_bounds__(256)
fec_imbe7200x4400_kernel(const uint8_t* __restrict__ frames, size_t n, mbx_param_record* __restrict__ out,
                         DeviceTables tabs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const mbx_tables* T = tabs.t;
    const uint8_t* f = frames + i * MBX_IMBE_FRAME_BYTES;
    BitReader br;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        br.w[k] = (load_be16(f + 4 * k) << 16) | load_be16(f + 4 * k + 2);
    }
    br.w[4] = load_be16(f + 16) << 16;

    uint32_t row[8];
    row[0] = br.take(0, 23);
    row[1] = br.take(23, 23);
    row[2] = br.take(46, 23);
    row[3] = br.take(69, 23);
    row[4] = br.take(92, 15);
    row[5] = br.take(107, 15);
    row[6] = br.take(122, 15);
    row[7] = br.take(137, 7);

    const int c0 = golay2312(T, row[0], row[0]);
    PrSequence pr(row[0] >> 11);
    row[1] ^= pr.mask_for(23);
    row[2] ^= pr.mask_for(23);
    row[3] ^= pr.mask_for(23);
    row[4] ^= pr.mask_for(15);
    row[5] ^= pr.mask_for(15);
    row[6] ^= pr.mask_for(15);

    int prot = 0;
    RecordWriter rw;
    rw.push(row[0], 23, 12);
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        prot += golay2312(T, row[r], row[r]);
        rw.push(row[r], 23, 12);
    }
    int c4 = 0;
#pragma unroll
    for (int r = 4; r < 7; ++r) {
        const int e = hamming1511(T, row[r], row[r]);
        prot += e;
        if (r == 4) {
            c4 = e;
        }
        rw.push(row[r], 15, 11);
    }
    rw.push(row[7], 7, 7);

    mbx_param_record rec;
    rec.w[0] = (uint32_t)(rw.hi >> 32);
    rec.w[1] = (uint32_t)rw.hi;
    rec.w[2] = (uint32_t)(rw.lo >> 32);
    rec.w[3] = (uint32_t)c0 | ((uint32_t)prot << 8) | ((uint32_t)c4 << 16)
               | ((MBE_PROCESS_FLAG_C0_VALID | MBE_PROCESS_FLAG_C4_VALID) << 24);
    *reinterpret_cast<uint4*>(&out[i]) = make_uint4(rec.w[0], rec.w[1], rec.w[2], rec.w[3]);
}

__global__ void __launch_bounds__(256)
fec_ambe3600x2450_kernel(const uint8_t* __restrict__ frames, size_t n, mbx_param_record* __restrict__ out,
                         DeviceTables tabs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const mbx_tables* T = tabs.t;
    const uint8_t* f = frames + i * MBX_AMBE_FRAME_BYTES;   // 9-byte frames: byte loads
    BitReader br;
    br.w[0] = ((uint32_t)f[0] << 24) | ((uint32_t)f[1] << 16) | ((uint32_t)f[2] << 8) | f[3];
    br.w[1] = ((uint32_t)f[4] << 24) | ((uint32_t)f[5] << 16) | ((uint32_t)f[6] << 8) | f[7];
    br.w[2] = (uint32_t)f[8] << 24;
    br.w[3] = br.w[4] = 0;

    uint32_t row0 = br.take(0, 24), row1 = br.take(24, 23), row2 = br.take(47, 11), row3 = br.take(58, 14);

    uint32_t cw;
    int c0 = golay2312(T, row0 >> 1, cw);
    row0 = (cw << 1) | (row0 & 1u);
    if (c0 == 0 && (__popc(row0) & 1)) {   // Golay24 overall parity, only when no data bit was fixed
        row0 ^= 1u;
        c0 = 1;
    }
    PrSequence pr((row0 >> 12) & 0xfffu);
    row1 ^= pr.mask_for(23);
    const int prot = golay2312(T, row1, row1);

    RecordWriter rw;
    rw.push(row0, 24, 12);
    rw.push(row1, 23, 12);
    rw.push(row2, 11, 11);
    rw.push(row3, 14, 14);
    *reinterpret_cast<uint4*>(&out[i]) =
        make_uint4((uint32_t)(rw.hi >> 32), (uint32_t)rw.hi, (uint32_t)(rw.lo >> 32),
                   (uint32_t)c0 | ((uint32_t)prot << 8) | (MBE_PROCESS_FLAG_C0_VALID << 24));
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


// Code-word level ECC (the public per-word helpers, batched): kind 0 = Golay(23,12), 1 = Hamming(15,11).
//   ref mbe_golay2312 / mbe_checkGolayBlock src/ecc/ecc.c:221-301, mbe_hamming1511 src/ecc/ecc.c:366-408
__global__ void __launch_bounds__(256)
ecc_words_kernel(int kind, const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ out, int32_t* __restrict__ errs,
                 DeviceTables tabs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    uint32_t fixed;
    const int e = (kind == 0) ? golay2312(tabs.t, in[i] & 0x7fffffu, fixed) : hamming1511(tabs.t, in[i] & 0x7fffu, fixed);
    out[i] = fixed;
    if (errs) {
        errs[i] = e;
    }
}

}  // namespace mbx
