// mbx_front_imbe.h -- the FRONT END of the one-launch T = 1 step for EIGHT IMBE 7200x4400 frames by ONE WAVE:
// lane = (frame fi = lane >> 3, row r = lane & 7) for the FEC, (frame, sub-lane) for the parameter expansion that follows it
// (mbx_expand_imbe.h) -- wire frames in, parameter records + FrameParams rows out.
//
// Why eight frames per wave.  The first one-launch kernel of round 5 ran the front end of a stream's frame in the stream's own
// wave (scalar-unit FEC, in-wave expansion).  Measured (tools/stage_times.py, EXPERIMENTS.md 5.1): 8 us of a 23 us wave life
// before the frame was expanded, and a kernel 17-26 us slower than the stream stage alone -- not latency but INSTRUCTION ISSUE:
// one frame per wave-instruction is 8x (expansion) to 64x (FEC) the issue slots of the frame-parallel kernels, on SIMDs that
// are already ~60 % busy.  Here the work keeps the frame-parallel mappings and moves INTO the stream kernel's launch instead:
// front blocks (this file) and stream blocks are workgroups of ONE grid, the front blocks first, and a front block hands its
// eight rows to the stream blocks that consume them through the workspace and a flag word (see imbe_one_launch_kernel and
// FrontLink, mbx_stream.hip).
//
// The FEC by lanes (bit-exact with fec_imbe7200x4400_wire, mbx_fec_frame.h; ref src/imbe/imbe7200x4400.c:424-443, 636-673,
// 469-515, 709-744; src/ecc/ecc.c:221-301, 366-408):
//   * lane (fi, r) cuts row r out of three 16-bit words of frame fi (rows start at bits 0, 23, 46, 69, 92, 107, 122, 137);
//   * Golay syndromes: parity(data) = half_hi[data >> 6] ^ half_lo[data & 63], two 64-entry tables held one entry per lane
//     (DerivedTables::golay_half_syn) and read with ds_bpermute -- all 64 rows of the wave at once;
//   * C0 first: its correction (one table read by the r = 0 lanes) gives the frame's demodulation seed and b0;
//   * the demodulation mask of a row is a 15- or 23-bit window of a sequence that depends on the 12-bit seed only:
//     DerivedTables::pr_bits holds all 4,096 sequences (114 bits each), a lane reads the two words its window lies in;
//   * rows 1..3 Golay, rows 4..6 Hamming(15,11) (four popcounts), one table read per lane for the correction;
//   * the record: every lane shifts its data bits to their place in the 88-bit record, the eight lanes of a frame add up
//     (disjoint fields: add = or) with three DPP steps per word; the error counts ride in the fourth word the same way.
#pragma once
#include "mbx_device.h"
#include "mbx_expand_imbe.h"
#ifndef MBX_FTS
#define MBX_FTS(i, v) do { } while (0)
#endif

namespace mbx {

__device__ __forceinline__ uint32_t front_golay_syndrome(uint32_t cw, uint32_t half_syn) {
    const uint32_t data = cw >> 11;
    const uint32_t e_hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((data >> 6) << 2), (int)half_syn) >> 16;
    const uint32_t e_lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((data & 63u) << 2), (int)half_syn) & 0xffffu;
    return (e_hi ^ e_lo ^ cw) & 0x7ffu;
}

// have: lane's frame exists; frame: its 18 wire bytes (2-byte aligned).  Returns the frame's parameter record in ALL eight
// lanes of the frame (zeros when !have).  All 64 lanes call this together.
__device__ __forceinline__ uint4 front8_fec_imbe(bool have, const uint8_t* frame, const DeviceTables& tabs, int lane) {
    typedef const __attribute__((address_space(1))) char* G;
    const G T = (G)tabs.t;
    const G D = (G)tabs.d;
    const int r = lane & 7;
    const int start = r < 4 ? 23 * r : 32 + 15 * r;   // 0, 23, 46, 69, 92, 107, 122, 137
    const int width = r < 4 ? 23 : (r < 7 ? 15 : 7);
    // lane-held tables (depend on the lane only: requested with the frame)
    const uint32_t half_syn = *(const __attribute__((address_space(1))) uint32_t*)(D + offsetof(DerivedTables, golay_half_syn) + 4u * (uint32_t)lane);
    const uint32_t hg01 = *(const __attribute__((address_space(1))) uint32_t*)(T + offsetof(mbx_tables, hamming_gen));
    const uint32_t hg23 = *(const __attribute__((address_space(1))) uint32_t*)(T + offsetof(mbx_tables, hamming_gen) + 4u);
    uint32_t row = 0u;
    if (have) {   // three big-endian 16-bit words from the one the row starts in (the last one clamped to the frame: what lies beyond is shifted out)
        const int k0 = start >> 4;
        const __attribute__((address_space(1))) uint16_t* f = (const __attribute__((address_space(1))) uint16_t*)frame;
        const uint32_t a = f[k0], b = f[k0 + 1 < 8 ? k0 + 1 : 8], c = f[k0 + 2 < 8 ? k0 + 2 : 8];
        auto be = [](uint32_t v) { return ((v & 0xffu) << 8) | (v >> 8); };
        const uint64_t two = ((uint64_t)be(a) << 48) | ((uint64_t)be(b) << 32) | ((uint64_t)be(c) << 16);
        row = (uint32_t)((two << (start & 15)) >> (64 - width));
    }
    MBX_FTS(1, row);   // the frame's words are there
    // C0: Golay(23,12) on row 0 (every lane forms the syndrome of its own row; only r = 0 uses it here)
    {
        const uint32_t syn = front_golay_syndrome(row, half_syn);
        uint32_t fix = 0u;
        if (r == 0) {
            fix = *(const __attribute__((address_space(1))) uint16_t*)(T + offsetof(mbx_tables, golay_matrix) + 2u * syn);
        }
        row ^= fix << 11;   // (zero in the other lanes)
        MBX_FTS(2, row);   // C0 corrected
        // c0 errors: kept in lane r = 0 as the low byte of the fourth record word (below)
        const uint32_t c0 = (uint32_t)__popc(fix);
        const uint32_t row0 = (uint32_t)__builtin_amdgcn_ds_bpermute((lane & ~7) << 2, (int)row);
        // demodulation: rows 1..6 XOR a window of the sequence of seed = C0's twelve data bits
        const bool golay = r >= 1 && r <= 3, hamming = r >= 4 && r <= 6;
        if (golay || hamming) {
            const uint32_t seed = row0 >> 11;
            const int p = golay ? 23 * (r - 1) : (15 * r + 9);   // 0, 23, 46 / 69, 84, 99
            typedef uint32_t u2v __attribute__((ext_vector_type(2), aligned(4)));
            const u2v w = *(const __attribute__((address_space(1))) u2v*)(D + offsetof(DerivedTables, pr_bits) + 16u * seed + 4u * (uint32_t)(p >> 5));
            const uint64_t two = ((uint64_t)w.x << 32) | w.y;
            row ^= (uint32_t)((two << (p & 31)) >> (64 - width));
        }
        MBX_FTS(3, row);   // demodulated
        // syndromes: Golay on rows 1..3, Hamming(15,11) on rows 4..6; one correction read per lane
        const uint32_t gsyn = front_golay_syndrome(row, half_syn);
        uint32_t hsyn = 0u;
        hsyn |= ((uint32_t)__popc(row & (hg01 & 0xffffu)) & 1u);
        hsyn |= ((uint32_t)__popc(row & (hg01 >> 16)) & 1u) << 1;
        hsyn |= ((uint32_t)__popc(row & (hg23 & 0xffffu)) & 1u) << 2;
        hsyn |= ((uint32_t)__popc(row & (hg23 >> 16)) & 1u) << 3;
        uint32_t err = 0u;
        if (golay || hamming) {
            const uint32_t off = golay ? (uint32_t)offsetof(mbx_tables, golay_matrix) + 2u * gsyn
                                       : (uint32_t)offsetof(mbx_tables, hamming_fix) + 2u * hsyn;
            const uint32_t fx = *(const __attribute__((address_space(1))) uint16_t*)(T + off);
            if (golay) {
                row ^= fx << 11;
                err = (uint32_t)__popc(fx);
            } else if (hsyn != 0u) {
                row ^= fx;
                err = 1u;
            }
        }
        MBX_FTS(4, row);   // corrections there
        // the record: 12 data bits of rows 0..3, 11 of rows 4..6, the 7 bits of row 7, in this order (88 bits, big-endian)
        const int count = r < 4 ? 12 : (r < 7 ? 11 : 7);
        const int off = r < 4 ? 12 * r : (r < 7 ? 4 + 11 * r : 81);   // 0, 12, 24, 36, 48, 59, 70, 81
        const uint32_t data = row >> (width - count);
        const uint64_t placed = (uint64_t)data << (64 - count - (off & 31));
        const uint32_t hi = (uint32_t)(placed >> 32), lo = (uint32_t)placed;
        const int wi = off >> 5;
        const uint32_t x = sum8(wi == 0 ? hi : 0u);
        const uint32_t y = sum8(wi == 0 ? lo : (wi == 1 ? hi : 0u));
        const uint32_t z = sum8(wi == 1 ? lo : (wi == 2 ? hi : 0u));
        // c0 | protected errors << 8 | c4 << 16 | flags << 24: disjoint fields, the protected errors really add up
        const uint32_t w3 = sum8(r == 0 ? c0 : (r == 7 ? ((MBE_PROCESS_FLAG_C0_VALID | MBE_PROCESS_FLAG_C4_VALID) << 24)
                                                       : ((err << 8) | (r == 4 ? err << 16 : 0u))));
        return have ? make_uint4(x, y, z, w3) : make_uint4(0u, 0u, 0u, 0u);
    }
}

}  // namespace mbx
