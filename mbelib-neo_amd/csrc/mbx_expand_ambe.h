// mbx_expand_ambe.h -- the frame-parallel half of the AMBE parameter decode for ONE (frame, sub-lane) pair: eight
// consecutive lanes expand one frame into its 64-dword FrameParams row in LDS.  Shared by the expand kernels
// (mbx_expand.hip: rows -> HBM workspace) and the LDS-resident AMBE stream kernels (mbx_stream.hip: the wave expands the
// next eight frames of its own stream into eight LDS rows, no launch and no HBM round trip).
//   ref src/ambe/ambe3600x2450.c:176-387, 461-553; src/ambe/ambe3600x2400.c:164-425 (k2400)
// Row layout: mbx_expand.hip.
#pragma once
#include "mbx_device.h"

#ifndef MBX_GLOBAL
#define MBX_GLOBAL __attribute__((address_space(1)))
#endif

namespace mbx {
namespace xp {

__device__ __forceinline__ int rbit(const uint32_t w[3], int i) { return (int)((w[i >> 5] >> (31 - (i & 31))) & 1u); }

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


// `have`: this frame exists (the lanes of a missing frame still take part in the exchanges).  rec_ptr: the frame's
// parameter record; row: the frame's 64 (+ pad) dwords in LDS; sub: 0..7, the lane's place in its group of eight.
// All 64 lanes of the wave must call this together (it exchanges data inside each group of eight lanes and uses a
// wavefront-scope LDS fence between the block lanes and the summary lane of a frame).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 class has no address-space-aware copy)

// the record by value (wherever it came from); `rec` is ignored when !have
template <bool k2400>
__device__ __forceinline__ void expand_ambe_frame_rec(bool have, const u32x4 rec, float* row, int sub, const DeviceTables& tabs) {
    // explicit global address space: inside the stream kernels the table pointers have passed an asm barrier and would
    // otherwise be read with FLAT loads, which also count against the LDS counter
    const MBX_GLOBAL mbx_tables* T = (const MBX_GLOBAL mbx_tables*)tabs.t;
    const MBX_GLOBAL DerivedTables* D = (const MBX_GLOBAL DerivedTables*)tabs.d;
    int bad = 0, L = 0;
    float w0 = 0.0f, f0 = 0.0f;
    bool silence = false;
    uint32_t w[3] = {0, 0, 0}, errw = 0;
    if (have) {
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
                f0 = D->ambep_f0[b0];
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
            const MBX_GLOBAL float(*prba24)[3] = k2400 ? T->ambep_prba24 : T->ambe_prba24;
            const MBX_GLOBAL float(*prba58)[4] = k2400 ? T->ambep_prba58 : T->ambe_prba58;
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
            const MBX_GLOBAL float* hoc;
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
            const MBX_GLOBAL uint8_t(*lmprbl)[4] = k2400 ? T->ambep_lmprbl : T->ambe_lmprbl;
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
    if (have && bad == 0 && !silence) {
        const int b1 = k2400 ? pick(w, 38, 39, 40, 41) : pick(w, 4, 5, 6, 7, 35);
        // the eight decisions of this codebook row as one bit mask (one 8-byte load, not L loads)
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 vq = *reinterpret_cast<const MBX_GLOBAL u32x2*>(k2400 ? &T->ambep_vuv[b1][0] : &T->ambe_vuv[b1][0]);
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
    if (have && sub == 0) {
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
        // v[60]: the unvoiced-band factor 0.2046 / sqrt(w0) of the prediction (ref src/ambe/ambe3600x2450.c:389-459) -- an IEEE
        // square root and an IEEE division, ~25 dependent instructions, here once per frame on one lane instead of in every
        // wave of the stream kernel; L moves into the class word (v[63] = class | L << 8)
        row[60] = (bad == 0) ? ((float)0.2046 / sqrtf(w0)) : 0.0f;
        row[61] = sum42;
        row[62] = __uint_as_float(errw);
        row[63] = __int_as_float(bad | (L << 8));
    }
}

// the record from memory (rec_ptr must be dereferenceable even when !have)
template <bool k2400>
__device__ __forceinline__ void expand_ambe_frame(bool have, const mbx_param_record* __restrict__ rec_ptr, float* row, int sub,
                                                  const DeviceTables& tabs) {
    u32x4 rec = {0u, 0u, 0u, 0u};
    if (have) {
        rec = *reinterpret_cast<const MBX_GLOBAL u32x4*>((const MBX_GLOBAL mbx_param_record*)rec_ptr);
    }
    expand_ambe_frame_rec<k2400>(have, rec, row, sub, tabs);
}

}  // namespace xp
}  // namespace mbx
