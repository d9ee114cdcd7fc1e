// mbx_expand.hip -- stateless half of the parameter decode, EIGHT LANES PER FRAME.
//
// Everything in mbe_decodeImbe4400Parms / mbe_decodeAmbe2450Parms that depends only on the 88 / 49
// parameter bits -- fundamental, voicing decisions, gain / PRBA / higher-order coefficient
// dequantisation, the block inverse DCTs that yield the prediction residuals T_l -- is independent
// between frames.  It runs here, 8 frames per wavefront: sub-lane 0 of a frame does the header and
// the voicing bits; sub-lanes 1..6 own one inverse-DCT block each (IMBE), or all eight lanes own half
// a block each (AMBE, four blocks of up to 17 coefficients).  Sums keep the
// reference's sequential order inside a block, so the residuals are bit-identical to the CPU path.
// What is left for the stream kernel is the part that needs the previous frame: the prediction.
//
// Replaces (ref = arancormonk/mbelib-neo v2.0.0):
//   IMBE  src/imbe/imbe7200x4400.c:117-270  (fundamental, bit layout, voicing, gains, Ri, HOC, IDCT)
//   AMBE  src/ambe/ambe3600x2450.c:176-387, 461-553  (classification, V/UV, gain, PRBA, HOC, IDCT)
//         src/ambe/ambe3600x2400.c:164-425            (the same for D-STAR: expand_ambe_body<true>)
// The IMBE kernel serves launches with one frame per stream; with more the stream kernel expands the record itself
// (expand_imbe_wave, mbx_stream.hip).  Workgroups are four independent waves.
//
// Output record (FrameParams, 64 dwords): v[1..56] T_l, v[57..58] voicing bits, v[59] w0, v[60] L,
// v[61] K (IMBE) / mean residual Sum42 (AMBE), v[62] error-context word, v[63] frame class
// (0 voice, 1 invalid IMBE fundamental, 2 AMBE+2 erasure, 7 AMBE+2 tone; D-STAR: 3 tone class without a usable index,
// 5..122 tone index), v[0] AMBE gain increment.
#include "mbx_device.h"

namespace mbx {

constexpr int kFramesPerWave = 8;
constexpr int kWavesPerBlock = 4;    // 256-thread workgroups: a quarter of the workgroup launches of single-wave blocks
constexpr int kFramesPerBlock = kFramesPerWave * kWavesPerBlock;
constexpr int kRow = 65;   // 64 dwords + 1 pad

__device__ __forceinline__ int rbit(const uint32_t w[3], int i) { return (int)((w[i >> 5] >> (31 - (i & 31))) & 1u); }

// every wave writes the rows of its own eight frames (nothing crosses waves in these kernels)
__device__ __forceinline__ void write_out(const float (*tile)[kRow], FrameParams* out, size_t first, size_t n) {
    wave_lds_sync();
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int q = 0; q < kFramesPerWave; ++q) {
        const int r = kFramesPerWave * wave + q;
        if (first + r < n) {
            reinterpret_cast<float*>(&out[first + r])[lane] = tile[r][lane];   // one coalesced 256-byte row
        }
    }
}

__device__ __forceinline__ uint32_t low_bits(uint32_t v, int n) { return v & ((1u << n) - 1u); }

// 8 waves per SIMD (<= 64 VGPRs): every wave of a 65,536-frame launch (8,192 of them) is then resident at once; at 7 per
// SIMD the last eighth waits for a second round, which doubles the duration of a kernel this short.
__global__ void __launch_bounds__(64 * kWavesPerBlock, 8)
expand_imbe_kernel(const mbx_param_record* __restrict__ recs, size_t n, FrameParams* __restrict__ out, DeviceTables tabs) {
    __shared__ float tile[kFramesPerBlock][kRow];
    __shared__ uint32_t words[kFramesPerBlock][64];   // parameter words b_0..b_57 of each frame
    __shared__ float gains[kFramesPerBlock][8];        // G_1..G_6
    const mbx_tables* T = tabs.t;
    const size_t first = (size_t)blockIdx.x * kFramesPerBlock;
    const int fi = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const size_t i = first + fi;
    float* row = tile[fi];
    {
        uint4* z = reinterpret_cast<uint4*>(&words[fi][8 * sub]);
        z[0] = make_uint4(0u, 0u, 0u, 0u);
        z[1] = make_uint4(0u, 0u, 0u, 0u);
    }
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    int bad = 1, L = 0, K = 0;
    float w0 = 0.0f;
    if (i < n) {
        rec = *reinterpret_cast<const uint4*>(&recs[i]);
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
    }
    const int L9 = bad ? 0 : L - 9;
    const bool live = (i < n) && !bad;
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
    const float* rows = &D->imbe_blk_rows[L9][blk][0][0];   // cosine rows of the block's outputs; the first one now
    float cosr[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        cosr[k] = rows[k];
    }
    wave_lds_sync();
    // Bit layout (ref src/imbe/imbe7200x4400.c:156-168): payload bit i feeds bit e[1] of word e[0].  The
    // eight lanes of a frame scatter ten payload bits each with LDS atomic ORs.
    if (live) {
        const uint32_t w[3] = {rec.x, rec.y, rec.z};
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            const int idx = 10 * sub + t;   // payload bit idx + 6
            if (idx < 79) {
                const uint32_t m = e[t] & 0xffu, pos = e[t] >> 8;
                if (m < 58u && pos < 12u) {
                    atomicOr(&words[fi][m], (uint32_t)rbit(w, idx + 6) << pos);
                }
            }
        }
    }
    wave_lds_sync();
    float b2v = 0.0f;
    if (live && sub == 1) {
        b2v = T->imbe_B2[low_bits(words[fi][2], 6)];   // the one look-up that depends on the frame's own bits
    }
    if (block_lane) {   // gain G_sub (:190-209)
        float G = b2v;
        if (sub != 1) {
            const int inb = (int)nb;
            const int bm = (int)low_bits(words[fi][sub + 1], inb);
            G = (step * ((float)bm - ldexpf(1.0f, inb - 1) + 0.5f));
        }
        gains[fi][sub] = G;
    }
    wave_lds_sync();
    if (i < n) {
        if (sub == 0) {
            uint32_t vlo = 0, vhi = 0;
            if (!bad) {   // voicing: three harmonics per band, band K-1 first
                // harmonic l takes bit max(K-1 - (l-1)/3, 0) of b1: band bit K-1-k covers l = 3k+1..3k+3,
                // every harmonic past 3K shares bit 0 (src/imbe/imbe7200x4400.c:170-188)
                const uint32_t b1 = low_bits(words[fi][1], 12);
                unsigned long long v = 0ULL;
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    if (k < K && ((b1 >> (K - 1 - k)) & 1u)) {
                        v |= 7ULL << (3 * k);
                    }
                }
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
                    sum = sum + (am * gains[fi][m] * ric[m]);
                }
                C2[1] = sum;
            }
#pragma unroll
            for (int k = 2; k <= 10; ++k) {   // higher-order coefficients (:233-249); zero past the block length
                const int Bm = (int)((bmw[k >> 2] >> (8 * (k & 3))) & 0xffu);
                float v = 0.0f;
                if (Bm > 0) {
                    const int bm = (int)low_bits(words[fi][m0 + k - 2], Bm);
                    v = (qstep[k] * (((float)bm - ldexpf(1.0f, Bm - 1)) + 0.5f));
                }
                C2[k] = 2.0f * v;
            }
            // per-block inverse DCT (:251-270).  All ten terms are added unconditionally (x + 0*c == x for every x this
            // sum can take); the cosines of harmonic l are one contiguous row of a host-made table, fetched one
            // output ahead of the arithmetic.
            float next[10];
            for (int j = 1; j <= ji; ++j) {
                const float* nr = rows + 10 * (j < ji ? j : j - 1);
#pragma unroll
                for (int k = 0; k < 10; ++k) {
                    next[k] = nr[k];
                }
                float sum = 0;
#pragma unroll
                for (int k = 1; k <= 10; ++k) {
                    sum = sum + (C2[k] * cosr[k - 1]);
                }
                row[l0 + j - 1] = sum;
#pragma unroll
                for (int k = 0; k < 10; ++k) {
                    cosr[k] = next[k];
                }
            }
        }
    }
    write_out(tile, out, first, n);
}

__device__ __forceinline__ int pick(const uint32_t w[3], int i0, int i1, int i2, int i3 = -1, int i4 = -1, int i5 = -1,
                                    int i6 = -1, int i7 = -1, int i8 = -1) {
    const int idx[9] = {i0, i1, i2, i3, i4, i5, i6, i7, i8};
    int v = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (idx[k] >= 0) {
            v = (v << 1) | rbit(w, idx[k]);
        }
    }
    return v;
}

// k2400: AMBE 3600x2400 (D-STAR, ref src/ambe/ambe3600x2400.c:164-425): other bit positions, codebooks and frame
// classes (0 voice, 3 tone class without a usable index, 5..122 tone index); the arithmetic is the same.
template <bool k2400>
__device__ __forceinline__ void expand_ambe_body(const mbx_param_record* __restrict__ recs, size_t n, FrameParams* __restrict__ out,
                                                 const DeviceTables& tabs) {
    __shared__ float tile[kFramesPerBlock][kRow];
    const mbx_tables* T = tabs.t;
    const size_t first = (size_t)blockIdx.x * kFramesPerBlock;
    const int fi = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const size_t i = first + fi;
    float* row = tile[fi];
    int bad = 0, L = 0;
    float w0 = 0.0f, f0 = 0.0f;
    bool silence = false;
    uint32_t w[3] = {0, 0, 0}, errw = 0;
    if (i < n) {
        const uint4 rec = *reinterpret_cast<const uint4*>(&recs[i]);
        w[0] = rec.x;
        w[1] = rec.y;
        w[2] = rec.z;
        errw = rec.w;
        const int total_errors = (int)(rec.w & 0xffu) + (int)((rec.w >> 8) & 0xffu);
        const int u0 = (int)(w[0] >> 20);
        const int u1 = (int)((w[0] >> 8) & 0xfffu);
        const unsigned long long two = ((unsigned long long)w[0] << 32) | w[1];
        const int u3 = (int)((two >> 15) & 0x3fffu);
        const bool tone_sig = (((u0 >> 6) & 0x3f) == 63) && (((u3 & 0xf) == 0) || (((u1 >> 8) & 0xf) == (u1 & 0xf)));
        if (k2400) {
            const int b0 = pick(w, 0, 1, 2, 3, 4, 5, 48);
            if ((b0 & 0x7E) == 0x7E) {   // tone class (:212-234); a silence model set here is wiped by the policy
                const uint32_t t7 = 0xE1u, t6 = 0x78u, t5 = 0xB4u;   // the three 8-entry tables as bit masks
                const int def = pick(w, 6, 7, 8);
                const int tone = (int)(((t7 >> def) & 1u) << 7 | ((t6 >> def) & 1u) << 6 | ((t5 >> def) & 1u) << 5)
                                 | (rbit(w, 9) << 4) | (rbit(w, 42) << 3) | (rbit(w, 43) << 2) | (rbit(w, 10) << 1) | rbit(w, 11);
                bad = (tone >= 5 && tone <= 122) ? tone : 3;
            } else {
                f0 = tabs.d->ambep_f0[b0];
                w0 = (float)((double)(f0 * (float)2) * M_PI);
                L = T->ambep_L[b0];
            }
        } else {
            const int b0 = pick(w, 0, 1, 2, 3, 37, 38, 39);
            if (tone_sig && total_errors < 6) {
                bad = 7;
            } else if ((b0 >= 120 && b0 <= 123) || b0 == 126 || b0 == 127) {
                bad = 2;
            } else if (b0 == 124 || b0 == 125) {
                silence = true;
                f0 = (float)M_PI / 32.0f;
                w0 = f0 * (float)(2.0 * M_PI);
                L = (b0 == 124) ? 15 : 14;
            } else {
                f0 = T->ambe_w0[b0];
                w0 = (float)((double)(f0 * (float)2) * M_PI);
                L = T->ambe_L[b0];
            }
        }
        if (bad == 0) {   // two lanes per inverse-DCT block: each takes half of the block's outputs
            const int blk = (sub >> 1) + 1, half = sub & 1;
            const int b3 = k2400 ? pick(w, 10, 11, 12, 13, 14, 15, 16, 44, 45) : pick(w, 12, 13, 14, 15, 16, 17, 18, 19, 40);
            const int b4 = k2400 ? pick(w, 17, 18, 19, 20, 21, 46, 47) : pick(w, 20, 21, 22, 23, 41, 42, 43);
            const float(*prba24)[3] = k2400 ? T->ambep_prba24 : T->ambe_prba24;
            const float(*prba58)[4] = k2400 ? T->ambep_prba58 : T->ambe_prba58;
            float Gm[9];
            Gm[1] = 0.0f;
            Gm[2] = prba24[b3][0];
            Gm[3] = prba24[b3][1];
            Gm[4] = prba24[b3][2];
            Gm[5] = prba58[b4][0];
            Gm[6] = prba58[b4][1];
            Gm[7] = prba58[b4][2];
            Gm[8] = prba58[b4][3];
            float Ra = 0, Rb = 0;   // Ri[2*blk-1], Ri[2*blk]
#pragma unroll
            for (int m = 1; m <= 8; ++m) {
                const float am = (m == 1) ? 1.0f : 2.0f;
                Ra = Ra + (am * Gm[m] * T->ambe_ri_cos[m][2 * blk - 1]);
                Rb = Rb + (am * Gm[m] * T->ambe_ri_cos[m][2 * blk]);
            }
            const float rconst = (float)(1.0 / (2.0 * M_SQRT2));
            int hbits;
            const float* hoc;
            if (k2400) {   // (:362-401); bit 24 is not used, b8 is the three bits 35..37 shifted up by one
                hbits = (blk == 1) ? pick(w, 22, 23, 25, 26)
                                   : ((blk == 2) ? pick(w, 27, 28, 29, 30) : ((blk == 3) ? pick(w, 31, 32, 33, 34) : (pick(w, 35, 36, 37) << 1)));
                hoc = (blk == 1) ? T->ambep_hoc_b5[hbits]
                                 : ((blk == 2) ? T->ambep_hoc_b6[hbits] : ((blk == 3) ? T->ambep_hoc_b7[hbits] : T->ambep_hoc_b8[hbits]));
            } else {
                hbits = (blk == 1) ? pick(w, 24, 25, 26, 27, 44)
                                   : ((blk == 2) ? pick(w, 28, 29, 30, 45) : ((blk == 3) ? pick(w, 31, 32, 33, 46) : pick(w, 34, 47, 48)));
                hoc = (blk == 1) ? T->ambe_hoc_b5[hbits]
                                 : ((blk == 2) ? T->ambe_hoc_b6[hbits] : ((blk == 3) ? T->ambe_hoc_b7[hbits] : T->ambe_hoc_b8[hbits]));
            }
            const uint8_t(*lmprbl)[4] = k2400 ? T->ambep_lmprbl : T->ambe_lmprbl;
            int l = 1;
            for (int q = 1; q < blk; ++q) {
                l += lmprbl[L][q - 1];
            }
            const int ji = lmprbl[L][blk - 1];
            // Coefficients 7..17 of a block are zero (:337-348); the reference still adds their products, which
            // leaves every partial sum unchanged (x + 0*c == x for every x this sum can take), so only k <= 6 is done.
            float C[7];
            C[1] = (float)0.5 * (Ra + Rb);
            C[2] = rconst * (Ra - Rb);
#pragma unroll
            for (int k = 3; k <= 6; ++k) {
                C[k] = (k <= ji) ? hoc[k - 3] : 0.0f;
            }
            const int jsplit = (ji + 1) >> 1;
            const int j0 = half ? jsplit + 1 : 1, j1 = half ? ji : jsplit;
            // the six cosines of an output are fetched one output ahead of the arithmetic (all waves of a launch move in
            // step, so a look-up inside the sum would be paid in full)
            float cosr[7], next[7];
            if (j0 <= j1) {
#pragma unroll
                for (int k = 1; k <= 6; ++k) {
                    cosr[k] = T->ambe_idct_cos[ji][j0][k];
                }
            }
            for (int j = j0; j <= j1; ++j) {
                const int jn = j < j1 ? j + 1 : j;
#pragma unroll
                for (int k = 1; k <= 6; ++k) {
                    next[k] = T->ambe_idct_cos[ji][jn][k];
                }
                float sum = 0;
#pragma unroll
                for (int k = 1; k <= 6; ++k) {
                    if (k <= ji) {
                        const float ak = (k == 1) ? 1.0f : 2.0f;
                        sum = sum + (ak * C[k] * cosr[k]);
                    }
                }
                row[l + j - 1] = sum;
#pragma unroll
                for (int k = 1; k <= 6; ++k) {
                    cosr[k] = next[k];
                }
            }
        }
    }
    // Voicing decisions: harmonic l takes entry (int)(l * 16 f0) & 7 of its codebook row.  The eight lanes of a frame
    // share the harmonics (l = sub + 1, sub + 9, ...: at most seven each) and OR their bits together inside the group of
    // eight lanes; every lane of the frame executes this, so the exchange needs no predicate.
    uint32_t vlo = 0, vhi = 0;
    if (i < n && bad == 0 && !silence) {
        const int b1 = k2400 ? pick(w, 38, 39, 40, 41) : pick(w, 4, 5, 6, 7, 35);
        // the eight decisions of this codebook row as one bit mask (one 8-byte load, not L loads)
        const uint2 vq = *reinterpret_cast<const uint2*>(k2400 ? &T->ambep_vuv[b1][0] : &T->ambe_vuv[b1][0]);
        uint32_t vmask = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t byte = ((q < 4 ? vq.x : vq.y) >> (8 * (q & 3))) & 0xffu;
            vmask |= (byte & 1u) << q;
        }
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const int l = sub + 1 + 8 * q;
            if (l <= L) {
                const int jl = (int)((float)l * (float)16.0 * f0);
                const uint32_t v = (vmask >> (jl & 7)) & 1u;
                if (l <= 32) {
                    vlo |= v << (l - 1);
                } else {
                    vhi |= v << (l - 33);
                }
            }
        }
    }
    vlo |= (uint32_t)__shfl_xor((int)vlo, 1, kWave);
    vhi |= (uint32_t)__shfl_xor((int)vhi, 1, kWave);
    vlo |= (uint32_t)__shfl_xor((int)vlo, 2, kWave);
    vhi |= (uint32_t)__shfl_xor((int)vhi, 2, kWave);
    vlo |= (uint32_t)__shfl_xor((int)vlo, 4, kWave);
    vhi |= (uint32_t)__shfl_xor((int)vhi, 4, kWave);
    wave_lds_sync();   // the block lanes and the summary lane of a frame are in the same wave
    if (i < n && sub == 0) {
        float dg = 0.0f, sum42 = 0.0f;
        if (bad == 0) {
            dg = k2400 ? T->ambep_dg[pick(w, 6, 7, 8, 9, 42, 43)] : T->ambe_dg[pick(w, 8, 9, 10, 11, 36)];
            float tsum = 0.0f;   // Sum42 in the reference's order (l ascending); rows past L hold zeros
            const int L4 = (L + 3) & ~3;
            for (int l = 1; l <= L4; l += 4) {
                const float t0 = row[l], t1 = row[l + 1], t2 = row[l + 2], t3 = row[l + 3];   // four reads in flight
                tsum += t0;
                tsum += (l + 1 <= L) ? t1 : 0.0f;
                tsum += (l + 2 <= L) ? t2 : 0.0f;
                tsum += (l + 3 <= L) ? t3 : 0.0f;
            }
            sum42 = tsum / (float)L;
        }
        row[0] = dg;
        row[57] = __uint_as_float(vlo);
        row[58] = __uint_as_float(vhi);
        row[59] = w0;
        row[60] = __int_as_float(L);
        row[61] = sum42;
        row[62] = __uint_as_float(errw);
        row[63] = __int_as_float(bad);
    }
    write_out(tile, out, first, n);
}

__global__ void __launch_bounds__(64 * kWavesPerBlock)
expand_ambe_kernel(const mbx_param_record* __restrict__ recs, size_t n, FrameParams* __restrict__ out, DeviceTables tabs) {
    expand_ambe_body<false>(recs, n, out, tabs);
}

__global__ void __launch_bounds__(64 * kWavesPerBlock)
expand_ambe2400_kernel(const mbx_param_record* __restrict__ recs, size_t n, FrameParams* __restrict__ out, DeviceTables tabs) {
    expand_ambe_body<true>(recs, n, out, tabs);
}

}  // namespace mbx
