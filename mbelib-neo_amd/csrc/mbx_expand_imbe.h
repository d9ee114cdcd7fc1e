// mbx_expand_imbe.h -- the frame-parallel half of the IMBE 7200x4400 parameter decode for ONE (frame, sub-lane) pair: eight
// consecutive lanes expand one frame into its 64-dword FrameParams row in LDS (sub-lane 0 the header and the voicing bits,
// sub-lanes 1..6 one inverse-DCT block each).  Shared by the expand kernel (mbx_expand.hip: records from memory, rows to the
// HBM workspace) and the front blocks of the one-launch T = 1 kernel (mbx_stream.hip: records straight from the lane-parallel
// FEC of the same wave).  ref src/imbe/imbe7200x4400.c:117-270.  Row layout: mbx_expand.hip.
#pragma once
#include "mbx_device.h"
#include "mbx_expand_ambe.h"   // xp::rbit
#ifndef MBX_FTS
#define MBX_FTS(i, v) do { } while (0)   // stage marks of the front blocks (mbx_stream.hip, MBX_STAGE_TIMES builds)
#endif

namespace mbx {
namespace xp {

__device__ __forceinline__ uint32_t low_bits_x(uint32_t v, int n) { return v & ((1u << n) - 1u); }

// `have`: this frame exists; rec: its parameter record (ignored when !have); row: the frame's 64 (+ pad) dwords of LDS; words: 64
// dwords of LDS for the frame's parameter words b_0..b_57; gains: 8 floats of LDS (G_1..G_6); sub: 0..7.  All 64 lanes of the wave
// call this together (wavefront-scope LDS fences between the stages).
__device__ __forceinline__ void expand_imbe_frame_rec(bool have, uint4 rec, float* row, uint32_t* words, float* gains, int sub,
                                                      const DeviceTables& tabs) {
    const mbx_tables* T = tabs.t;
    {
        uint4* z = reinterpret_cast<uint4*>(&words[8 * sub]);
        z[0] = make_uint4(0u, 0u, 0u, 0u);
        z[1] = make_uint4(0u, 0u, 0u, 0u);
    }
    int bad = 1, L = 0, K = 0;
    float w0 = 0.0f;
    if (have) {
        const uint32_t w[3] = {rec.x, rec.y, rec.z};
        int b0 = (int)(w[0] >> 26);
        b0 = (b0 << 1) | rbit(w, 85);
        b0 = (b0 << 1) | rbit(w, 86);
        if (b0 <= 207) {
            const uint2 q = tabs.d->imbe_b0[b0];
            w0 = __uint_as_float(q.x);
            L = (int)(q.y & 0xffu);
            K = (int)(q.y >> 8);
            bad = (L == 0) ? 1 : 0;   // the reference has stored w0 but not L in this case
        }
    } else {
        rec = make_uint4(0u, 0u, 0u, 0u);
    }
    MBX_FTS(5, L);   // b0 entry there: L known
    const int L9 = bad ? 0 : L - 9;
    const bool live = have && !bad;
    const bool block_lane = live && sub >= 1 && sub <= 6;
    // ---- every table value this lane will need is requested HERE, in one round: the addresses depend only on L and
    // the lane (host-made per-block tables, mbx_init), not on each other.  All 8,192 waves of a 65,536-frame launch are
    // resident at once and move in step, so a chain of dependent look-ups would be paid in full.
    const DerivedTables* D = tabs.d;
    uint32_t e[10];                       // bit-layout entries of this lane's ten payload bits
    {
        const uint16_t* bo = reinterpret_cast<const uint16_t*>(&T->imbe_bo[L9][0][0]);
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            const int idx = 10 * sub + t;
            e[t] = bo[idx < 79 ? idx : 78];
        }
    }
    const int blk = block_lane ? sub : 1;
    const uint32_t info = D->imbe_blk_info[L9][blk];
    const int g = (sub >= 2 && sub <= 6) ? sub - 2 : 0;
    const float nb = T->imbe_ba[L9][g][0], step = T->imbe_ba[L9][g][1];
    float ric[7];
#pragma unroll
    for (int m = 1; m <= 6; ++m) {
        ric[m] = T->imbe_ri_cos[m][blk];
    }
    float qstep[11];
    uint32_t bmw[3];                      // the block's bit counts, k = 0..11 as three dwords
    {
        const uint32_t* bp = reinterpret_cast<const uint32_t*>(&D->imbe_blk_bm[L9][blk][0]);
        bmw[0] = bp[0];
        bmw[1] = bp[1];
        bmw[2] = bp[2];
#pragma unroll
        for (int k = 2; k <= 10; ++k) {
            qstep[k] = D->imbe_blk_step[L9][blk][k];
        }
    }
    // cosine rows of the block's outputs, from the table indexed by the block's LENGTH (one round trip later than everything else:
    // the length comes with `info` -- but 4.4 KB that stay in L1 instead of rows scattered over 154 KB; the first one now)
    const int ji_rows = (int)((info >> 16) & 0xffu);
    const float* rows = &D->imbe_len_rows[ji_rows <= 10 ? ji_rows : 10][0][0];
    float cosr[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        cosr[k] = rows[k];
    }
    MBX_FTS(6, cosr[0]);   // every table request of the L round issued; block info and the first cosine row there
    wave_lds_sync();
    // Bit layout (ref src/imbe/imbe7200x4400.c:156-168): payload bit i feeds bit e[1] of word e[0].  The eight lanes of a frame
    // scatter ten payload bits each with LDS atomic ORs -- WITHOUT branches: the lane's ten bits are one 10-bit window of the record
    // (record bits 10 sub + 6 .. 10 sub + 15), a bit that is zero ORs nothing (a missing frame's record is zero; a frame without an
    // L scatters into words nobody reads), and mbx_init has checked every entry's range.
    {
        const int s0 = 10 * sub + 6;   // 6 .. 76: the window starts in word s0 >> 5 and ends in it or in the next (never past word 2)
        const uint32_t hi = s0 < 32 ? rec.x : (s0 < 64 ? rec.y : rec.z), lo = s0 < 32 ? rec.y : (s0 < 64 ? rec.z : 0u);
        uint32_t window = (uint32_t)(((((uint64_t)hi << 32) | lo) << (s0 & 31)) >> 54);
        window &= (sub == 7) ? ~1u : ~0u;   // (payload bit 79 does not exist)
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            const uint32_t m = e[t] & 0xffu, pos = e[t] >> 8;
            atomicOr(&words[m], ((window >> (9 - t)) & 1u) << pos);
        }
    }
    wave_lds_sync();
    MBX_FTS(7, e[0]);   // bit layout there, words scattered
    float b2v = 0.0f;
    if (live && sub == 1) {
        b2v = T->imbe_B2[low_bits_x(words[2], 6)];   // the one look-up that depends on the frame's own bits
    }
    if (block_lane) {   // gain G_sub (:190-209)
        float G = b2v;
        if (sub != 1) {
            const int inb = (int)nb;
            const int bm = (int)low_bits_x(words[sub + 1], inb);
            G = (step * ((float)bm - ldexpf(1.0f, inb - 1) + 0.5f));
        }
        gains[sub] = G;
    }
    wave_lds_sync();
    MBX_FTS(8, b2v);   // B2 there, gains
    // voicing: three harmonics per band, band K-1 first -- harmonic l takes bit max(K-1 - (l-1)/3, 0) of b1: band bit K-1-k covers
    // l = 3k+1..3k+3, every harmonic past 3K shares bit 0 (src/imbe/imbe7200x4400.c:170-188).  The frame's eight lanes take bands
    // sub and sub + 8 and add their (disjoint) masks up.
    const uint32_t b1 = low_bits_x(words[1], 12);
    uint32_t band_lo = 0u, band_hi = 0u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = sub + 8 * h;
        if (k < 12 && k < K && ((b1 >> (K - 1 - k)) & 1u)) {
            const unsigned long long m = 7ULL << (3 * k);
            band_lo |= (uint32_t)m;
            band_hi |= (uint32_t)(m >> 32);
        }
    }
    band_lo = sum8(band_lo);
    band_hi = sum8(band_hi);
    if (have) {
        if (sub == 0) {
            uint32_t vlo = 0, vhi = 0;
            if (!bad) {
                unsigned long long v = ((unsigned long long)band_hi << 32) | band_lo;
                if (b1 & 1u) {
                    v |= ~0ULL << (3 * K);
                }
                v &= (L >= 64) ? ~0ULL : ((1ULL << L) - 1ULL);   // bit l-1 = harmonic l, l <= L
                vlo = (uint32_t)v;
                vhi = (uint32_t)(v >> 32);
            }
            row[0] = 0.0f;
            row[57] = __uint_as_float(vlo);
            row[58] = __uint_as_float(vhi);
            row[59] = w0;
            row[60] = __int_as_float(L);
            row[61] = __int_as_float(K);
            row[62] = __uint_as_float(rec.w);
            row[63] = __int_as_float(bad);
        } else if (block_lane) {
            const int m0 = (int)(info & 0xffu), l0 = (int)((info >> 8) & 0xffu), ji = (int)((info >> 16) & 0xffu);
            float C2[11];   // a_k * C[k]: the doubling is exact
            {   // this block's mean: one output of the 6-point inverse DCT of the gains (:211-231)
                float sum = 0;
#pragma unroll
                for (int m = 1; m <= 6; ++m) {
                    const float am = (m == 1) ? 1.0f : 2.0f;
                    sum = sum + (am * gains[m] * ric[m]);
                }
                C2[1] = sum;
            }
#pragma unroll
            for (int k = 2; k <= 10; ++k) {   // higher-order coefficients (:233-249); zero past the block length
                // (no branch on Bm: with Bm = 0 the field is empty, (0 - 2^-1) + 0.5 = 0 and the step is 0; the word read may then lie
                //  past the frame's 58 -- inside this wave's LDS either way)
                const int Bm = (int)((bmw[k >> 2] >> (8 * (k & 3))) & 0xffu);
                const int bm = (int)low_bits_x(words[m0 + k - 2], Bm);
                const float v = (qstep[k] * (((float)bm - ldexpf(1.0f, Bm - 1)) + 0.5f));
                C2[k] = 2.0f * v;
            }
            // per-block inverse DCT (:251-270).  All ten terms are added unconditionally (x + 0*c == x for every x this
            // sum can take); the cosines of harmonic l are one contiguous row of a host-made table, fetched one
            // output ahead of the arithmetic.
            // (two outputs per trip with the two row buffers swapping roles: no register copies)
            float next[10];
            auto fetch = [&](float (&dst)[10], int j) {   // the row of output j + 1 (clamped: the last fetch is never used)
                const float* nr = rows + 10 * (j < ji ? j : ji - 1);
#pragma unroll
                for (int k = 0; k < 10; ++k) {
                    dst[k] = nr[k];
                }
            };
            auto output = [&](const float (&c)[10], int j) {
                float sum = 0;
#pragma unroll
                for (int k = 1; k <= 10; ++k) {
                    sum = sum + (C2[k] * c[k - 1]);
                }
                row[l0 + j - 1] = sum;
            };
            for (int j = 1; j <= ji; j += 2) {
                fetch(next, j);
                output(cosr, j);
                if (j + 1 <= ji) {
                    fetch(cosr, j + 1);
                    output(next, j + 1);
                }
            }
        }
    }
}

}  // namespace xp
}  // namespace mbx
