// mbx_device.h -- wave64 device helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mbx_tables.h"
#include "mbx_types.h"

// Results must track an IEEE CPU build of the reference: no FMA contraction, no fast-math.
#pragma clang fp contract(off)

namespace mbx {

constexpr int kWave = 64;

// Tables derived on the host at mbx_init() and kept next to the blob in HBM.
struct DerivedTables {
    uint32_t lcg_mul[161];    // 171^k mod 53125           (unvoiced-noise LCG jump-ahead)
    uint32_t lcg_add[161];    // additive term after k steps
    uint32_t lcg_pack[161];   // both in one word (each is below 53,125 < 2^16): lcg_mul[k] | lcg_add[k] << 16 -- one load per sample
    float    log2_int[64];    // log2f((float)L) from the host libm (AMBE gain term)
    // Wave-uniform quotients of the parameter decode: one scalar load each instead of a 12-instruction IEEE
    // division executed by all 64 lanes.  Made on the host with the same float expressions (correctly rounded
    // division, no contraction), so the values are the ones the device would compute.
    float    l_ratio[57][57];       // (float)prev_L / (float)cur_L
    float    imbe_rho_over_l[57];   // rho(L) / (float)L, rho = 0.4 | 0.03 L - 0.05 | 0.7 (imbe7200x4400.c log-magnitude prediction)
    float    ambe_pred_over_l[57];  // 0.65f / (float)L
    float    nfrac[192];            // (float)n / 160.0f, the interpolated branch's amplitude ramp
    uint32_t pr_mul[116];     // 173^k mod 2^16             (demodulation sequence jump-ahead, k = 0..115)
    uint32_t pr_add[116];     // additive term after k steps
    uint32_t ham_basis[11];   // Hamming(15,11) code word of data bit i (soft-decision candidates)
    uint32_t golay_rot[12];   // golay_gen[i] with its low seven bits rotated left by one (soft Golay table index, mbx_fec.hip)
    uint32_t ham7100_basis[11];   // the same for the IMBE 7100x4400 bit mapping
    // IMBE parameter expansion, per L (index L - 9) and lane: which block / position a lane owns
    uint32_t imbe_lane_map[48][64];       // hoc block | hoc index k << 3 | harmonic's block << 7 | block length << 10 | index j << 14
    float    imbe_hoc_sd[48][64];         // standdev[k - 2] of the higher-order coefficient in word lane + 8
    float    imbe_idct_rows[48][64][10];  // idct_cos[ji][j][1..10] of harmonic `lane`
    // the same per inverse-DCT block (index blk = 1..6) for the 8-lanes-per-frame expand kernel
    uint32_t imbe_blk_info[48][8];        // first word m | first harmonic l << 8 | block length << 16
    uint8_t  imbe_blk_bm[48][8][12];      // bit count of the higher-order coefficient k = 2..10 of the block (0 = none)
    float    imbe_blk_step[48][8][12];    // quantstep[Bm - 1] * standdev[k - 2] of that coefficient
    float    imbe_len_rows[11][10][10];    // idct_cos[ji][j][1..10] of output j = 1..10 of a block of LENGTH ji = 0..10 (zero past the length).
                                           // (Until round 4 a copy per (L, block): 154 KB of gathers that did not stay in L1 -- the expansion
                                           //  kernel's time was their volume.  4.4 KB by length: 24.6 -> 21.1 us for 65,536 frames.)
    uint32_t pad_len_rows[52];             // (to 18 x 256 B: the tables behind keep their alignment to cache lines)
    uint2    imbe_b0[208];                // b0 -> (w0 bits, L | K << 8): one look-up instead of three
    float    wola_inv[160];       // 1 / wola_denom[n] (0 where the reference skips the sample: denom <= 1e-10)
    float    ambep_f0[128];       // AMBE 3600x2400: exp2f(-4.311767578125f - 2.1336e-2f * (b0 + 0.5f)) from the host libm
    // Lane-parallel FEC of the one-launch T = 1 kernels (mbx_fec_frame.h, LaneFec): tables of 64 entries held ONE ENTRY PER LANE
    uint32_t golay_half_syn[64];  // entry j: (parity of the data word whose six HIGH bits are j) << 16 | (... whose six LOW bits are j):
                                  // the Golay syndrome of a code word is half_hi[data >> 6] ^ half_lo[data & 63] ^ its parity bits
    uint2    pr_lane[64];         // demodulation sequence in closed form, x_k = A_k x_0 + C_k mod 2^16: entry j = (A | C << 16) of
                                  // k = j + 1 (.x) and of k = j + 65 (.y)
    uint32_t pr_bits[4096 + 1][4];   // the demodulation sequence of every 12-bit seed, 114 bits each, big-endian: bit 31 - (k & 31) of word
                                     // k >> 5 = the sequence bit of step k + 1 (x_0 = 16 seed; bit = x >> 15); + one entry of padding
    uint32_t imbe_L_lanes[64];    // byte k of entry j: IMBE L of b0 = j + 64 k (0: no such b0 / invalid L) -- ONE dword per lane holds the
                                  // whole b0 -> L law, so a wave that asks for it before it knows b0 has L without a memory round trip
};

// Resident waves per SIMD of the LDS-resident (T >= 4) stream kernels: the launch bounds of those kernels (mbx_stream.hip) AND the
// slot count the launcher's slicing heuristic prices a launch against (mbx_api.hip) -- one definition, so the two cannot drift apart.
#ifndef MBX_IMBE_LDS_WAVES_PER_SIMD
#define MBX_IMBE_LDS_WAVES_PER_SIMD 6   // 5,244 B of LDS per wave = 30 waves per CU: the register file decides (80 VGPRs)
#endif
#ifndef MBX_AMBE_LDS_WAVES_PER_SIMD
#define MBX_AMBE_LDS_WAVES_PER_SIMD 5   // 7,328 B of LDS per wave = 21 waves per CU; 80 registers (six waves) would spill
#endif
constexpr int kResultHistWords = 14;   // mbx_result_hist (include/mbx.h) as 64-bit words

// Output of the expand stage, input of the stream stage: 64 dwords per frame (layout in mbx_expand.hip).
struct FrameParams {
    float v[64];
};

struct DeviceTables {
    const mbx_tables*    t;
    const DerivedTables* d;
    int                  ablate;   // timing-only stage mask; read only by the -DMBX_ABLATE development build (tools/)
    int                  reverse;  // stream kernels: workgroup b takes stream S - 1 - b (see launch_stream, mbx_api.hip)
    int                  front_skip;   // read only by the -DMBX_TESTING build (mbx_testing_set_front_skip): 2^k > 0 -- the front blocks of chunks 0, 2^k,
                                       // 2 x 2^k, ... of the one-launch kernels do nothing, so that their stream blocks take the fall-back path; always 0 in the product
    int                  tones_off;    // != 0: AMBE tone frames synthesise silence and leave the tone phases alone -- the reference's NOTONES build
                                       // (mbx_set_tone_synthesis; ref CMakeLists.txt:330-337, src/core/mbelib.c:747-751,815-819)
    const int32_t*       stream_map;   // stream kernels: batch row s works on state / rng slot stream_map[s] (nullptr: slot s)
    // Resident state (sessions, queue mode; nullptr: the ABI triplet is kept whole, what every mbx_process_* entry point does).
    // resident[slot] != 0 says "prev_mp_enhanced of this stream is elided: it equals cur_mp field for field" -- true after every
    // frame that ends with prev_mp_enhanced := cur_mp (all IMBE frames, AMBE voice / erasure / re-initialisation frames).  The
    // LDS-resident stream kernels then neither read nor write that struct, and they fetch from prev_mp only what the decode
    // reads (the rest on demand: repeats, erasures).  mbx_resident_materialize() writes the elided structs out.
    uint32_t*            resident;
};

// Single-frame kernels: a copy of the caller's three structs (and RNG state) in DEVICE memory, kept by the per-frame library
// between synchronous calls.  When the caller's structs still are what the previous call returned (the library compares them on
// the host), the frame reads its state from this copy -- HBM, ~1 us -- instead of pulling 7.8 KB across PCIe (~3 us); results
// always go to both.  `ok` (pinned) tells the host whether the copy is complete after this frame.
struct FrameShadow {
    mbe_parms*      state = nullptr;   // 3 structs: cur_mp, prev_mp, prev_mp_enhanced
    mbx_stream_rng* rng = nullptr;
    uint32_t*       ok = nullptr;
    uint32_t        use = 0u;          // read the state from the copy
    // the wire frame BY VALUE (kernel arguments): a launch whose host could read the frame's bytes passes them here, and the FEC
    // starts with the kernel instead of after a PCIe round trip for 18 bytes
    uint32_t        have_frame = 0u;
    uint32_t        frame_words[6] = {0u, 0u, 0u, 0u, 0u, 0u};   // the bytes, little-endian packed
};

// Stage masks for timing experiments exist only in the development build (make ablate -> libmbx_hip_ablate.so, used by
// tools/); in the product library the tests are compile-time zeros and no entry point can switch a stage off.
#ifdef MBX_ABLATE
#define MBX_ABL(tabs, mask) (((tabs).ablate & (mask)) != 0)
#else
#define MBX_ABL(tabs, mask) false
#endif

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }

// One wavefront per workgroup: lanes exchange data through LDS, and LDS operations of one wave are
// executed in issue order, so all that is needed is that the compiler keeps the program order of the
// LDS accesses.  A wavefront-scope fence does exactly that and -- unlike __syncthreads(), whose
// workgroup-scope fence waits for vmcnt(0) -- does not drain outstanding global stores / loads.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// DPP lane exchange inside a row of 16 lanes (no LDS round trip)
template <int kCtrl>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), kCtrl, 0xf, 0xf, true));
}
constexpr int kDppXor1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;   // lane i <-> 7-i inside each group of 8
constexpr int kDppMirror = 0x140;       // lane i <-> 15-i inside each row of 16

// sum over each group of eight consecutive lanes, returned in all eight (DPP: xor 1, xor 2, mirror within 8); all lanes call it
__device__ __forceinline__ uint32_t sum8(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, kDppXor1, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, kDppXor2, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, kDppHalfMirror, 0xf, 0xf, true);
    return v;
}

// Sum over the 64 lanes, returned wave-uniform (the same bits in every lane): four DPP steps reduce
// each row of 16, the four row totals are read back with v_readlane and added in a fixed order.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<kDppXor1>(v);
    v += dpp_f32<kDppXor2>(v);
    v += dpp_f32<kDppHalfMirror>(v);
    v += dpp_f32<kDppMirror>(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return ((r0 + r1) + r2) + r3;
}

__device__ __forceinline__ float lane_get(float v, int src) { return __shfl(v, src, kWave); }
__device__ __forceinline__ int lane_get(int v, int src) { return __shfl(v, src, kWave); }

// the value of lane `src` for a WAVE-UNIFORM src: one v_readlane_b32 (the result is a scalar), no LDS round trip
__device__ __forceinline__ float lane_read(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(src) & 63));
}

__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }

// base-4 digit reversal of an 8-bit index (radix-4 FFT ordering)
__device__ __forceinline__ int rev4(int p) {
    return ((p & 3) << 6) | ((p & 12) << 2) | ((p & 48) >> 2) | ((p & 192) >> 6);
}

}  // namespace mbx
