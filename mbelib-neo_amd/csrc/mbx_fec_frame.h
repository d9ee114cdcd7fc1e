// mbx_fec_frame.h -- hard-decision FEC of ONE frame by one thread: wire frame -> 16-byte parameter record.  Shared by the
// batch kernels (mbx_fec.hip: one thread per frame) and the single-frame kernels behind the synchronous per-frame API
// (mbx_stream.hip: lane 0 of the wave decodes the frame, the record stays in registers).
//   IMBE 7200x4400  ref src/imbe/imbe7200x4400.c:424-443, 636-673, 469-515, 709-744
//   AMBE 3600x24x0  ref src/ambe/ambe_common.c:22-46, 75-100, 127-157; src/ambe/ambe3600x2450.c:649-682
//   IMBE 7100x4400  ref src/imbe/imbe7100x4400.c:100-122, 292-334, 153-212, 381-479; src/ecc/ecc.c:422-464
//   ECC             ref src/ecc/ecc.c:221-301 (Golay(23,12) by syndrome table), :366-408 (Hamming(15,11))
// Integer work only: bit-exact.
// Two table readers (the FEC is templated on them): TabVector reads mbx_tables with ordinary per-thread loads (batch kernels);
// TabScalar reads it with s_load (single-frame kernels: the whole FEC is wave-uniform there and runs on the scalar unit, and --
// the point -- scalar loads do not queue behind the wave's vector loads, which return IN ORDER: the frame's FEC really runs
// while the three structs are still on their way over PCIe instead of waiting for them at its first table read).
#pragma once
#include "mbx_device.h"

namespace mbx {

struct TabVector {
    const mbx_tables* T;
    static constexpr bool kUniform = false;
    __device__ uint32_t golay_gen(int i) const { return T->golay_gen[i]; }
    __device__ uint32_t golay_fix(uint32_t syndrome) const { return T->golay_matrix[syndrome]; }
    __device__ uint32_t hamming_gen(bool v7100, int i) const { return v7100 ? T->hamming7100_gen[i] : T->hamming_gen[i]; }
    __device__ uint32_t hamming_fix(bool v7100, int syndrome) const { return v7100 ? T->hamming7100_fix[syndrome] : T->hamming_fix[syndrome]; }
    __device__ uint32_t imbe_K(int b0) const { return T->imbe_K[b0]; }
    // the corrections of up to three Golay and three Hamming syndromes at once (the scalar reader turns them into ONE round trip)
    template <bool v7100>
    __device__ void fixes(const uint32_t gs[3], const int hs[3], uint32_t gf[3], uint32_t hf[3]) const {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            gf[r] = golay_fix(gs[r]);
            hf[r] = hamming_fix(v7100, hs[r]);
        }
    }
    template <typename Seq, typename Wav>
    __device__ Seq pr(uint32_t seed12) const { return Seq(seed12); }
};

// wave-uniform reads of the table blob through the scalar cache.  `T` must be uniform (a kernel argument), offsets too.
__device__ __forceinline__ uint32_t sload_dword(const void* base, uint32_t byte_off) {
    uint32_t v;
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(base), "s"(byte_off & ~3u) : "memory");
    return v;
}
template <typename E>   // element `index` of an array of 8- or 16-bit E at byte offset `array_off` (a multiple of 4) of the blob
__device__ __forceinline__ uint32_t sload_elem(const void* base, uint32_t array_off, uint32_t index) {
    const uint32_t byte = index * (uint32_t)sizeof(E);
    const uint32_t w = sload_dword(base, array_off + byte);
    return (w >> (8u * (byte & 3u))) & ((1u << (8u * (uint32_t)sizeof(E))) - 1u);
}

struct TabScalar {
    const mbx_tables* T;
    uint32_t gen[6];    // golay_gen[12], two per dword, fetched once
    uint32_t hgen[4];   // hamming_gen[4] | hamming7100_gen[4], likewise
    static constexpr bool kUniform = true;
    __device__ TabScalar() : T(nullptr) {}
    __device__ explicit TabScalar(const mbx_tables* t) : T(t) {
        static_assert(offsetof(mbx_tables, golay_gen) % 4 == 0 && offsetof(mbx_tables, golay_matrix) % 4 == 0
                          && offsetof(mbx_tables, hamming_gen) % 4 == 0 && offsetof(mbx_tables, hamming_fix) % 4 == 0
                          && offsetof(mbx_tables, hamming7100_gen) % 4 == 0 && offsetof(mbx_tables, hamming7100_fix) % 4 == 0
                          && offsetof(mbx_tables, imbe_K) % 4 == 0,
                      "scalar loads read whole dwords");
        // plain scalar loads (constant address space): the compiler places the wait at the first USE of a generator row, so a caller
        // that constructs the reader at the start of its wave has them in flight together with everything else it asks for
        const __attribute__((address_space(4))) uint32_t* g =
            (const __attribute__((address_space(4))) uint32_t*)((const __attribute__((address_space(4))) char*)t + offsetof(mbx_tables, golay_gen));
        const __attribute__((address_space(4))) uint32_t* h0 =
            (const __attribute__((address_space(4))) uint32_t*)((const __attribute__((address_space(4))) char*)t + offsetof(mbx_tables, hamming_gen));
        const __attribute__((address_space(4))) uint32_t* h1 =
            (const __attribute__((address_space(4))) uint32_t*)((const __attribute__((address_space(4))) char*)t + offsetof(mbx_tables, hamming7100_gen));
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            gen[i] = g[i];
        }
        hgen[0] = h0[0];
        hgen[1] = h0[1];
        hgen[2] = h1[0];
        hgen[3] = h1[1];
    }
    __device__ uint32_t golay_gen(int i) const { return (gen[i >> 1] >> (16 * (i & 1))) & 0xffffu; }
    __device__ uint32_t golay_fix(uint32_t syndrome) const { return sload_elem<uint16_t>(T, offsetof(mbx_tables, golay_matrix), syndrome); }
    __device__ uint32_t hamming_gen(bool v7100, int i) const { return (hgen[(v7100 ? 2 : 0) + (i >> 1)] >> (16 * (i & 1))) & 0xffffu; }
    template <bool v7100>
    __device__ void fixes(const uint32_t gs[3], const int hs[3], uint32_t gf[3], uint32_t hf[3]) const {
        const uint32_t hbase = v7100 ? (uint32_t)offsetof(mbx_tables, hamming7100_fix) : (uint32_t)offsetof(mbx_tables, hamming_fix);
        const uint32_t gbase = (uint32_t)offsetof(mbx_tables, golay_matrix);
        uint32_t w[6];
        asm volatile("s_load_dword %0, %6, %7\n\ts_load_dword %1, %6, %8\n\ts_load_dword %2, %6, %9\n\t"
                     "s_load_dword %3, %6, %10\n\ts_load_dword %4, %6, %11\n\ts_load_dword %5, %6, %12\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(w[0]), "=&s"(w[1]), "=&s"(w[2]), "=&s"(w[3]), "=&s"(w[4]), "=&s"(w[5])
                     : "s"(T), "s"((gbase + 2u * gs[0]) & ~3u), "s"((gbase + 2u * gs[1]) & ~3u), "s"((gbase + 2u * gs[2]) & ~3u),
                       "s"((hbase + 2u * (uint32_t)hs[0]) & ~3u), "s"((hbase + 2u * (uint32_t)hs[1]) & ~3u),
                       "s"((hbase + 2u * (uint32_t)hs[2]) & ~3u)
                     : "memory");
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            gf[r] = (w[r] >> (16u * (gs[r] & 1u))) & 0xffffu;
            hf[r] = (w[3 + r] >> (16u * ((uint32_t)hs[r] & 1u))) & 0xffffu;
        }
    }
    __device__ uint32_t hamming_fix(bool v7100, int syndrome) const {
        return sload_elem<uint16_t>(T, v7100 ? offsetof(mbx_tables, hamming7100_fix) : offsetof(mbx_tables, hamming_fix), (uint32_t)syndrome);
    }
    __device__ uint32_t imbe_K(int b0) const { return sload_elem<uint8_t>(T, offsetof(mbx_tables, imbe_K), (uint32_t)b0); }
    template <typename Seq, typename Wav>
    __device__ Wav pr(uint32_t seed12) const { return Wav(*lanes, seed12); }
    const struct PrLane* lanes = nullptr;   // the wave's demodulation-sequence constants (set by the caller)
};

// The wire frame as values.  IMBE (18 bytes): nine big-endian 16-bit halves; AMBE (9 bytes): the nine bytes.
struct Wire {
    uint32_t h[9];
};

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
template <typename Tab>
__device__ __forceinline__ int golay2312_t(const Tab& tab, uint32_t cw, uint32_t& fixed) {
    uint32_t expect = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        expect ^= (cw & (0x400000u >> i)) ? tab.golay_gen(i) : 0u;
    }
    const uint32_t fix = tab.golay_fix(expect ^ (cw & 0x7ffu));
    fixed = cw ^ (fix << 11);
    return __popc(fix);
}
__device__ __forceinline__ int golay2312(const mbx_tables* T, uint32_t cw, uint32_t& fixed) {
    return golay2312_t(TabVector{T}, cw, fixed);
}

template <bool v7100, typename Tab>
__device__ __forceinline__ int hamming1511_t(const Tab& tab, uint32_t cw, uint32_t& fixed) {
    int syndrome = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        syndrome |= (__popc(cw & tab.hamming_gen(v7100, i)) & 1) << i;
    }
    fixed = syndrome ? (cw ^ tab.hamming_fix(v7100, syndrome)) : cw;
    return syndrome != 0;
}
__device__ __forceinline__ int hamming1511(const mbx_tables* T, uint32_t cw, uint32_t& fixed) {
    return hamming1511_t<false>(TabVector{T}, cw, fixed);
}

// the same two decoders in two steps, so that a frame's table lookups can go out together (Tab::fixes)
template <typename Tab>
__device__ __forceinline__ uint32_t golay_syndrome(const Tab& tab, uint32_t cw) {
    uint32_t expect = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        expect ^= (cw & (0x400000u >> i)) ? tab.golay_gen(i) : 0u;
    }
    return expect ^ (cw & 0x7ffu);
}
__device__ __forceinline__ int golay_apply(uint32_t cw, uint32_t fix, uint32_t& fixed) {
    fixed = cw ^ (fix << 11);
    return __popc(fix);
}
template <bool v7100, typename Tab>
__device__ __forceinline__ int hamming_syndrome(const Tab& tab, uint32_t cw) {
    int syndrome = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        syndrome |= (__popc(cw & tab.hamming_gen(v7100, i)) & 1) << i;
    }
    return syndrome;
}
__device__ __forceinline__ int hamming_apply(uint32_t cw, int syndrome, uint32_t fix, uint32_t& fixed) {
    fixed = syndrome ? (cw ^ fix) : cw;
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

// The same sequence for a whole wave at once (single-frame kernels): x_k = A_k x_0 + C_k mod 2^16 in closed form, lane j
// holding (A, C) of k = j + 1 and k = j + 65 -- constants of the lane, computed BEFORE the frame's bytes have arrived -- so the
// 114 bits are two multiply-adds and two ballots instead of 114 dependent steps (1,231 scalar instructions, 2.7 us of a lone
// wave's time).  mask_for() then cuts the masks out of the 128-bit ballot pair.
struct PrLane {
    uint32_t a1, c1, a2, c2;   // k = lane + 1, k = lane + 65
    __device__ explicit PrLane(int lane) {
        // (P, Q) of 2^b steps: P_0 = 173, Q_0 = 13849; P_{b+1} = P_b^2, Q_{b+1} = Q_b (P_b + 1)   (all mod 2^16)
        uint32_t P = 173u, Q = 13849u, a = 1u, c = 0u;
        const uint32_t k = (uint32_t)lane + 1u;
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            if (k & (1u << b)) {   // after the map so far: x -> P (a x + c) + Q
                a = (a * P) & 0xffffu;
                c = (c * P + Q) & 0xffffu;
            }
            Q = (Q * (P + 1u)) & 0xffffu;
            P = (P * P) & 0xffffu;
        }
        a1 = a;
        c1 = c;
        // 64 more steps: after the seven doublings (P, Q) is the map of 128 steps; the 64-step map is one doubling earlier
        uint32_t P64 = 173u, Q64 = 13849u;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            Q64 = (Q64 * (P64 + 1u)) & 0xffffu;
            P64 = (P64 * P64) & 0xffffu;
        }
        a2 = (a * P64) & 0xffffu;
        c2 = (c * P64 + Q64) & 0xffffu;
    }
};
struct PrWave {
    uint64_t lo, hi;   // bit j of lo: the sequence bit of step j + 1; of hi: step j + 65
    int      pos = 0;
    __device__ PrWave(const PrLane& L, uint32_t seed12) {
        const uint32_t x0 = (16u * seed12) & 0xffffu;
        lo = __ballot((((L.a1 * x0 + L.c1) >> 15) & 1u) != 0u);
        hi = __ballot((((L.a2 * x0 + L.c2) >> 15) & 1u) != 0u);
    }
    __device__ uint32_t mask_for(int width) {   // the next `width` bits, the first of them at bit width-1
        const int p = pos;
        pos += width;
        const uint64_t cut = (p < 64) ? ((lo >> p) | ((p > 0) ? (hi << (64 - p)) : 0ull)) : (hi >> (p - 64));
        return __brev((uint32_t)cut) >> (32 - width);
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

// f: the frame's 18 wire bytes (even address)
__device__ __forceinline__ Wire load_wire_imbe(const uint8_t* f) {
    Wire w;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        w.h[k] = load_be16(f + 2 * k);
    }
    return w;
}
__device__ __forceinline__ Wire load_wire_ambe(const uint8_t* f) {   // 9-byte frames: byte loads
    Wire w;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        w.h[k] = f[k];
    }
    return w;
}
__device__ __forceinline__ BitReader imbe_bits(const Wire& wire) {
    BitReader br;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        br.w[k] = (wire.h[2 * k] << 16) | wire.h[2 * k + 1];
    }
    br.w[4] = wire.h[8] << 16;
    return br;
}

// The IMBE 7200x4400 FEC in two halves.  The HEAD (rows out of the wire frame, Golay on C0) is all the fundamental needs:
// b0's six high bits are C0's first data bits, its two low bits sit in the unprotected row 7 -- so the one-launch T = 1 kernels
// know L, and can request every table value of the parameter expansion, before the TAIL (demodulation, three Golay and three
// Hamming words, the record) has run.
struct ImbeFecHead {
    uint32_t row[8];
    int      c0;
    __device__ int b0() const { return (int)((((row[0] >> 17) & 0x3fu) << 2) | ((row[7] >> 1) & 3u)); }   // record bits 0..5, 85, 86
};
template <typename Tab>
__device__ __forceinline__ ImbeFecHead fec_imbe7200x4400_head(const Tab& tab, const Wire& wire) {
    const BitReader br = imbe_bits(wire);
    ImbeFecHead h;
    h.row[0] = br.take(0, 23);
    h.row[1] = br.take(23, 23);
    h.row[2] = br.take(46, 23);
    h.row[3] = br.take(69, 23);
    h.row[4] = br.take(92, 15);
    h.row[5] = br.take(107, 15);
    h.row[6] = br.take(122, 15);
    h.row[7] = br.take(137, 7);
    h.c0 = golay2312_t(tab, h.row[0], h.row[0]);
    return h;
}
template <typename Tab>
__device__ __forceinline__ uint4 fec_imbe7200x4400_tail(const Tab& tab, const ImbeFecHead& h) {
    uint32_t row[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        row[r] = h.row[r];
    }
    const int c0 = h.c0;
    auto pr = tab.template pr<PrSequence, PrWave>(row[0] >> 11);
    row[1] ^= pr.mask_for(23);
    row[2] ^= pr.mask_for(23);
    row[3] ^= pr.mask_for(23);
    row[4] ^= pr.mask_for(15);
    row[5] ^= pr.mask_for(15);
    row[6] ^= pr.mask_for(15);

    int prot = 0;
    RecordWriter rw;
    rw.push(row[0], 23, 12);
    uint32_t gs[3], gf[3], hf[3];
    int hs[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        gs[r] = golay_syndrome(tab, row[1 + r]);
        hs[r] = hamming_syndrome<false>(tab, row[4 + r]);
    }
    tab.template fixes<false>(gs, hs, gf, hf);
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        prot += golay_apply(row[r], gf[r - 1], row[r]);
        rw.push(row[r], 23, 12);
    }
    int c4 = 0;
#pragma unroll
    for (int r = 4; r < 7; ++r) {
        const int e = hamming_apply(row[r], hs[r - 4], hf[r - 4], row[r]);
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
    return make_uint4(rec.w[0], rec.w[1], rec.w[2], rec.w[3]);
}
template <typename Tab>
__device__ __forceinline__ uint4 fec_imbe7200x4400_wire(const Tab& tab, const Wire& wire) {
    return fec_imbe7200x4400_tail(tab, fec_imbe7200x4400_head(tab, wire));
}
// ------------------------------------------------------------------------------------------------------------------------
// LaneFec: the IMBE 7200x4400 FEC of ONE frame by ONE WAVE with LANE r = ROW r of the frame (r = 0..7; the other lanes idle
// along as copies of row 7) -- the front end of the one-launch T = 1 kernels (imbe_stream_kernel_one_fused).
// Why not the scalar-unit form the single-frame kernels use (TabScalar above): there a lone wave hides a PCIe round trip behind
// it; here 28 waves per CU run it at once, and ~700 scalar instructions per wave took the CU's one scalar ALU from 34 % to 70 %
// busy (tools/stage_times.py: 6 us of a 23 us wave life before the expansion's table reads had even been requested).  By lanes
// the same work is ~100 vector + ~40 scalar instructions:
//   * rows: one 64-bit funnel shift per lane out of the frame's five words (byte order and the frame's 2-byte alignment in a
//     byte permute with a wave-uniform selector);
//   * Golay syndromes: the parity of a 12-bit data word is half_hi[data >> 6] ^ half_lo[data & 63], two 64-entry tables held one
//     entry per lane (DerivedTables::golay_half_syn) and read with ds_bpermute -- three rows at once;
//   * Hamming syndromes on lanes 4..6 by four popcounts, at the same time;
//   * the demodulation sequence in closed form across the lanes (DerivedTables::pr_lane), two ballots, and every row cuts its own
//     mask out of the 128-bit pair with its own shift;
//   * the corrections: scalar loads (C0's first; then the other six in one round trip) -- NOT per-lane vector loads, which would
//     return behind the wave's state loads.
// Bit-exact with fec_imbe7200x4400_wire (integer work only); ref src/imbe/imbe7200x4400.c:424-443, 636-673, 469-515, 709-744,
// src/ecc/ecc.c:221-301, 366-408.
// ------------------------------------------------------------------------------------------------------------------------
struct FrameWords {   // the five dwords that hold an 18-byte frame, as loaded (little-endian) from the 4-byte boundary at or below it
    uint32_t d[5];
};
struct LaneFecTables {   // one entry per lane, requested at the start of the wave
    uint32_t half_syn;   // golay_half_syn[lane]
    uint32_t pr1, pr2;   // pr_lane[lane]
};
__device__ __forceinline__ LaneFecTables lane_fec_request(const DerivedTables* D, int lane) {
    typedef const __attribute__((address_space(1))) char* G;
    LaneFecTables t;
    const G base = (G)D;
    t.half_syn = *(const __attribute__((address_space(1))) uint32_t*)(base + offsetof(DerivedTables, golay_half_syn) + 4u * (uint32_t)lane);
    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
    const u2v pr = *(const __attribute__((address_space(1))) u2v*)(base + offsetof(DerivedTables, pr_lane) + 8u * (uint32_t)lane);
    t.pr1 = pr.x;
    t.pr2 = pr.y;
    return t;
}

struct LaneFecHead {
    uint32_t row;   // lane r: row r as received (lane 0: after the Golay correction of C0)
    int      c0;    // errors corrected in C0 (wave-uniform)
    int      b0;    // the fundamental's index (wave-uniform): C0's six high data bits, two bits of the unprotected row 7
};

__device__ __forceinline__ uint32_t lane_golay_syndrome(uint32_t cw, uint32_t half_syn) {
    const uint32_t data = cw >> 11;
    const uint32_t e_hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((data >> 6) << 2), (int)half_syn) >> 16;
    const uint32_t e_lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((data & 63u) << 2), (int)half_syn) & 0xffffu;
    return (e_hi ^ e_lo ^ cw) & 0x7ffu;
}

// `misaligned`: the frame starts two bytes into fw.d[0] (wave-uniform)
__device__ __forceinline__ LaneFecHead lane_fec_imbe_head(const FrameWords& fw, bool misaligned, const LaneFecTables& lt, const mbx_tables* T,
                                                          int lane) {
    // the frame as five big-endian words (BitReader's layout): bytes 3,2,1,0 of a dword, or bytes 1,0 of the next and 3,2 of this one
    const uint32_t sel = misaligned ? 0x02030405u : 0x00010203u;   // v_perm_b32 selectors: byte 0..3 = src1, 4..7 = src0
    uint32_t w[6];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        w[k] = __builtin_amdgcn_perm(k < 4 ? fw.d[k + 1] : 0u, fw.d[k], sel);
    }
    if (misaligned) {
        w[4] &= 0xffff0000u;   // (only two bytes of the fifth word belong to the frame; never read past bit 144 anyway)
    }
    w[5] = 0u;
    const int r = lane < 7 ? lane : 7;
    const int start = r < 4 ? 23 * r : 32 + 15 * r;          // 0, 23, 46, 69, 92, 107, 122, 137
    const int width = r < 4 ? 23 : (r < 7 ? 15 : 7);
    const int wi = start >> 5;
    uint32_t hi = w[4], lo = w[5];
    hi = wi == 3 ? w[3] : hi, lo = wi == 3 ? w[4] : lo;
    hi = wi == 2 ? w[2] : hi, lo = wi == 2 ? w[3] : lo;
    hi = wi == 1 ? w[1] : hi, lo = wi == 1 ? w[2] : lo;
    hi = wi == 0 ? w[0] : hi, lo = wi == 0 ? w[1] : lo;
    const uint64_t two = ((uint64_t)hi << 32) | lo;
    LaneFecHead h;
    h.row = (uint32_t)((two << (start & 31)) >> (64 - width));
    // C0: Golay(23,12) on row 0.  Every lane forms the syndrome of its own row and reads a correction (rows 1..3 are still
    // scrambled, rows 4..7 are no Golay words: their values are dropped) -- instructions are per wave either way.
    const uint32_t syn = lane_golay_syndrome(h.row, lt.half_syn);
    // the correction by a SCALAR load (own path, own counter): a vector load would come back behind the wave's forty state loads,
    // which return in order -- measured, 2.5 us instead of 1 for this one look-up
    const uint32_t fix0 = sload_elem<uint16_t>(T, offsetof(mbx_tables, golay_matrix), (uint32_t)__builtin_amdgcn_readlane((int)syn, 0));
    if (lane == 0) {
        h.row ^= fix0 << 11;
    }
    h.c0 = __popc(fix0);
    const uint32_t row0 = (uint32_t)__builtin_amdgcn_readlane((int)h.row, 0), row7 = (uint32_t)__builtin_amdgcn_readlane((int)h.row, 7);
    h.b0 = (int)((((row0 >> 17) & 0x3fu) << 2) | ((row7 >> 1) & 3u));
    return h;
}

// the rest: demodulation, Golay on rows 1..3, Hamming on rows 4..6 (syndromes by lanes, corrections by one batch of scalar loads),
// the record.  hgen: hamming_gen[0..3], two per dword.
__device__ __forceinline__ uint4 lane_fec_imbe_tail(const LaneFecHead& h, const LaneFecTables& lt, const uint32_t hgen[2], const mbx_tables* T,
                                                    int lane) {
    const uint32_t row0 = (uint32_t)__builtin_amdgcn_readlane((int)h.row, 0);
    const uint32_t x0 = (16u * (row0 >> 11)) & 0xffffu;
    // bit j of plo: the sequence bit of step j + 1; of phi: of step j + 65
    const uint64_t plo = __ballot(((((lt.pr1 & 0xffffu) * x0 + (lt.pr1 >> 16)) >> 15) & 1u) != 0u);
    const uint64_t phi = __ballot(((((lt.pr2 & 0xffffu) * x0 + (lt.pr2 >> 16)) >> 15) & 1u) != 0u);
    const int r = lane < 7 ? lane : 7;
    const bool golay = r >= 1 && r <= 3, hamming = r >= 4 && r <= 6;
    const int width = r < 4 ? 23 : 15;
    const int p = golay ? 23 * (r - 1) : (15 * r + 9);   // where the row's mask starts in the sequence: 0, 23, 46 / 69, 84, 99
    uint32_t cut = (p < 64) ? (uint32_t)((plo >> (p & 63)) | ((phi << 1) << (63 - (p & 63)))) : (uint32_t)(phi >> ((p - 64) & 63));
    uint32_t row = h.row;
    if (golay || hamming) {
        row ^= __brev(cut) >> (32 - width);
    }
    // syndromes: Golay on lanes 1..3, Hamming(15,11) on lanes 4..6
    const uint32_t gsyn = lane_golay_syndrome(row, lt.half_syn);
    uint32_t hsyn = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t g = (hgen[i >> 1] >> (16 * (i & 1))) & 0xffffu;
        hsyn |= ((uint32_t)__popc(row & g) & 1u) << i;
    }
    // the six corrections in ONE scalar round trip (see the head), applied on the scalar unit: the rows are wave-uniform from here on
    uint32_t gs[3], gf[3], hf[3], rw[8];
    int hs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gs[k] = (uint32_t)__builtin_amdgcn_readlane((int)gsyn, 1 + k);
        hs[k] = __builtin_amdgcn_readlane((int)hsyn, 4 + k);
    }
    {
        const uint32_t hbase = (uint32_t)offsetof(mbx_tables, hamming_fix), gbase = (uint32_t)offsetof(mbx_tables, golay_matrix);
        uint32_t wd[6];
        asm volatile("s_load_dword %0, %6, %7\n\ts_load_dword %1, %6, %8\n\ts_load_dword %2, %6, %9\n\t"
                     "s_load_dword %3, %6, %10\n\ts_load_dword %4, %6, %11\n\ts_load_dword %5, %6, %12\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(wd[0]), "=&s"(wd[1]), "=&s"(wd[2]), "=&s"(wd[3]), "=&s"(wd[4]), "=&s"(wd[5])
                     : "s"(T), "s"((gbase + 2u * gs[0]) & ~3u), "s"((gbase + 2u * gs[1]) & ~3u), "s"((gbase + 2u * gs[2]) & ~3u),
                       "s"((hbase + 2u * (uint32_t)hs[0]) & ~3u), "s"((hbase + 2u * (uint32_t)hs[1]) & ~3u),
                       "s"((hbase + 2u * (uint32_t)hs[2]) & ~3u)
                     : "memory");
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gf[k] = (wd[k] >> (16u * (gs[k] & 1u))) & 0xffffu;
            hf[k] = (wd[3 + k] >> (16u * ((uint32_t)hs[k] & 1u))) & 0xffffu;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        rw[k] = (uint32_t)__builtin_amdgcn_readlane((int)row, k);
    }
    int prot = 0, c4 = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rw[1 + k] ^= gf[k] << 11;
        prot += __popc(gf[k]);
        if (hs[k] != 0) {
            rw[4 + k] ^= hf[k];
            prot += 1;
            c4 = (k == 0) ? 1 : c4;
        }
    }
    // the record: 12 data bits of rows 0..3, 11 of rows 4..6, the 7 bits of row 7, in this order (88 bits, big-endian in three words)
    uint32_t d[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        d[k] = k < 4 ? rw[k] >> 11 : (k < 7 ? rw[k] >> 4 : rw[k]);
    }
    const uint32_t x = (d[0] << 20) | (d[1] << 8) | (d[2] >> 4);
    const uint32_t y = (d[2] << 28) | (d[3] << 16) | (d[4] << 5) | (d[5] >> 6);
    const uint32_t z = (d[5] << 26) | (d[6] << 15) | (d[7] << 8);
    return make_uint4(x, y, z, (uint32_t)h.c0 | ((uint32_t)prot << 8) | ((uint32_t)c4 << 16)
                                   | ((MBE_PROCESS_FLAG_C0_VALID | MBE_PROCESS_FLAG_C4_VALID) << 24));
}

__device__ __forceinline__ uint4 fec_imbe7200x4400_frame(const mbx_tables* T, const uint8_t* f) {
    return fec_imbe7200x4400_wire(TabVector{T}, load_wire_imbe(f));
}


// the frame's 9 wire bytes
template <typename Tab>
__device__ __forceinline__ uint4 fec_ambe3600x2450_wire(const Tab& tab, const Wire& f) {

    BitReader br;
    br.w[0] = (f.h[0] << 24) | (f.h[1] << 16) | (f.h[2] << 8) | f.h[3];
    br.w[1] = (f.h[4] << 24) | (f.h[5] << 16) | (f.h[6] << 8) | f.h[7];
    br.w[2] = f.h[8] << 24;
    br.w[3] = br.w[4] = 0;

    uint32_t row0 = br.take(0, 24), row1 = br.take(24, 23), row2 = br.take(47, 11), row3 = br.take(58, 14);

    uint32_t cw;
    int c0 = golay2312_t(tab, row0 >> 1, cw);
    row0 = (cw << 1) | (row0 & 1u);
    if (c0 == 0 && (__popc(row0) & 1)) {   // Golay24 overall parity, only when no data bit was fixed
        row0 ^= 1u;
        c0 = 1;
    }
    auto pr = tab.template pr<PrSequence, PrWave>((row0 >> 12) & 0xfffu);
    row1 ^= pr.mask_for(23);
    const int prot = golay2312_t(tab, row1, row1);

    RecordWriter rw;
    rw.push(row0, 24, 12);
    rw.push(row1, 23, 12);
    rw.push(row2, 11, 11);
    rw.push(row3, 14, 14);
    return make_uint4((uint32_t)(rw.hi >> 32), (uint32_t)rw.hi, (uint32_t)(rw.lo >> 32),
                   (uint32_t)c0 | ((uint32_t)prot << 8) | (MBE_PROCESS_FLAG_C0_VALID << 24));
}
__device__ __forceinline__ uint4 fec_ambe3600x2450_frame(const mbx_tables* T, const uint8_t* f) {
    return fec_ambe3600x2450_wire(TabVector{T}, load_wire_ambe(f));
}


__device__ __forceinline__ int hamming1511_7100(const mbx_tables* T, uint32_t cw, uint32_t& fixed) {
    return hamming1511_t<true>(TabVector{T}, cw, fixed);
}

struct Bits88 {   // bit i (0 = first parameter bit) at bit 127 - i of hi:lo
    uint64_t hi = 0, lo = 0;
    __device__ int get(int i) const { return (int)(((i < 64) ? (hi >> (63 - i)) : (lo >> (127 - i))) & 1ull); }
    __device__ void put(int i, int b) {
        if (i < 64) {
            hi |= (uint64_t)b << (63 - i);
        } else {
            lo |= (uint64_t)b << (127 - i);
        }
    }
};

// the frame's 18 wire bytes; the record holds the 88 bits in 7200x4400 order
template <typename Tab>
__device__ __forceinline__ uint4 fec_imbe7100x4400_wire(const Tab& tab, const Wire& wire) {
    const BitReader br = imbe_bits(wire);

    uint32_t row[7];
    row[0] = br.take(0, 19);
    row[1] = br.take(19, 24);
    row[2] = br.take(43, 23);
    row[3] = br.take(66, 23);
    row[4] = br.take(89, 15);
    row[5] = br.take(104, 15);
    row[6] = br.take(119, 23);

    uint32_t w;
    const int c0 = golay2312_t(tab, (row[0] >> 1) & 0x3ffffu, w);   // the five missing positions are zeros
    row[0] = ((w & 0x3ffffu) << 1) | (row[0] & 1u);
    auto pr = tab.template pr<PrSequence, PrWave>((row[0] >> 12) & 0x7fu);
    row[1] ^= pr.mask_for(24);
    row[2] ^= pr.mask_for(23);
    row[3] ^= pr.mask_for(23);
    row[4] ^= pr.mask_for(15);
    row[5] ^= pr.mask_for(15);

    int prot = 0, c4 = 0;
    RecordWriter rw;                       // 7100 order: 7 + 12 + 12 + 12 + 11 + 11 + 23 bits
    rw.push(row[0] >> 12, 7, 7);
    const uint32_t gcw[3] = {row[1] >> 1, row[2], row[3]};   // C1 = cells 1..23
    uint32_t gs[3], gf[3], hf[3];
    int hs[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        gs[r] = golay_syndrome(tab, gcw[r]);
    }
    hs[0] = hamming_syndrome<true>(tab, row[4]);
    hs[1] = hamming_syndrome<true>(tab, row[5]);
    hs[2] = 0;
    tab.template fixes<true>(gs, hs, gf, hf);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        prot += golay_apply(gcw[r], gf[r], w);
        rw.push(w, 23, 12);
    }
    c4 = hamming_apply(row[4], hs[0], hf[0], w);
    prot += c4;
    rw.push(w, 15, 11);
    prot += hamming_apply(row[5], hs[1], hf[1], w);
    rw.push(w, 15, 11);
    rw.push(row[6], 23, 23);

    Bits88 d, t;
    d.hi = rw.hi;
    d.lo = rw.lo;
    // mbe_convertImbe7100to7200
    const int b0 = (int)(((d.hi >> 56) & 0x7eull) << 1) | (d.get(86) << 1) | d.get(87);   // bits 1..6, 86, 87
    const int K = (b0 < 208) ? (int)tab.imbe_K(b0) : 12;   // the reference's expression gives 12 for every b0 >= 208
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
    return make_uint4((uint32_t)(t.hi >> 32), (uint32_t)t.hi, (uint32_t)(t.lo >> 32),
                   (uint32_t)c0 | ((uint32_t)prot << 8) | ((uint32_t)c4 << 16)
                       | ((MBE_PROCESS_FLAG_C0_VALID | MBE_PROCESS_FLAG_C4_VALID) << 24));
}
__device__ __forceinline__ uint4 fec_imbe7100x4400_frame(const mbx_tables* T, const uint8_t* f) {
    return fec_imbe7100x4400_wire(TabVector{T}, load_wire_imbe(f));
}

}  // namespace mbx
