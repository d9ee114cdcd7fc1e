// mbx_stream.hip -- stream stage: parameter records + per-stream model state -> PCM.
//
// Layout: ONE STREAM PER WAVEFRONT (one 64-lane wave per workgroup).  Lane l holds element l of every per-harmonic
// array (Vl, Ml, log2Ml, PHIl, PSIl; L <= 56 < 64), previousUw[256] is four registers per lane, noiseOverlap[96] two.
// The T frames of the stream are processed in order inside the kernel.  `cur` stays in registers for the whole launch;
// `prev` and `prev_enhanced` are live only where the reference reads them and are parked in between: in their own HBM/L2
// slots (the loads and stores a T = 1 launch needs anyway; launches with one to three frames per stream), or -- the *_lds
// kernel instances, T >= 4, and the single-frame kernels -- `prev` in LDS for the whole launch and the ten fields of
// `prev_enhanced` that synthesis reads in registers (see ParkedPrevOnly); what the snapshot already holds is read back
// from it instead of being carried across the synthesiser.  A struct is 651 consecutive dwords: coalesced dword accesses.
//
// Stages and the reference code they replace (ref = arancormonk/mbelib-neo v2.0.0):
//   expand_imbe_wave src/imbe/imbe7200x4400.c:117-270               (a8, stateless half; T > 1 launches)
//   decode_imbe      src/imbe/imbe7200x4400.c:294-354               (a8, prediction)
//   decode_ambe      src/ambe/ambe3600x2450.c:389-459, ambe3600x2400.c:427-497   (a9, prediction)
//   imbe policy      src/imbe/imbe7200x4400.c:56-81, 780-888        (a10)
//   ambe policies    src/ambe/ambe3600x2450.c:716-877, ambe3600x2400.c:629-763   (a11)
//   enhance          src/core/mbelib.c:412-661                      (a13)
//   smooth           src/core/mbe_adaptive.c:151-266                (a14)
//   comfort noise    src/core/mbe_adaptive.c:50-60, 116-131         (a20)
//   noise            src/core/mbe_unvoiced_fft.c:304-341            (a15)
//   phases           src/core/mbelib.c:901-951                      (a16)
//   voiced bank      src/core/mbelib.c:208-319, 953-1040            (a17)
//   unvoiced         src/core/mbe_unvoiced_fft.c:210-275, 546-761   (a18)
//   tones            src/core/mbelib.c:691-856                      (f2)
//   clip / convert   src/core/mbelib.c:669-689, 1148-1177           (a19, a21)
//
// Numerics: integer decisions are reproduced exactly.  Float expressions that feed decisions keep the reference's
// operand order with FMA contraction off.  The synthesiser does not replay the reference's recurrences, it tracks them
// (see the voiced bank); measured against the CPU oracle: PCM relative RMS <= 3e-6, int16 within 1 LSB on 99.9998 % of
// samples (DESIGN.md section 4 for the tail).  Sums that feed state later frames build on are formed in the reference's
// index order (seq_sum4), so the decision state (log2Ml, localEnergy) is bit-identical to the reference's; a frame whose
// smoothing decision is within 2e-5 of its threshold is decided again with the reference's own arithmetic (enhance_exact).
#include <type_traits>

#include "mbx_device.h"
#include "mbx_expand_ambe.h"
// Development instrumentation (-DMBX_STAGE_TIMES, never in the product build; tools/stage_times.py): where the waves of a LAUNCH
// spend their lives -- every wave adds the 100 MHz wall-clock time between consecutive marks of the one-frame IMBE body and of
// the synthesiser to device-wide accumulators.
#ifdef MBX_STAGE_TIMES
namespace mbx {
constexpr int kStageWaves = 65536;
__device__ unsigned g_stage_buf[kStageWaves * 16];   // per workgroup (= wave = stream of the launch) and mark: ticks since the mark before
__shared__ unsigned long long s_stage_prev;          // one wave per workgroup
}
#define MBX_TS(i)                                                                          \
    do {                                                                                   \
        const unsigned long long t_ = wall_clock64();                                      \
        if ((threadIdx.x & 63) == 0) {                                                     \
            if ((i) > 0 && blockIdx.x < (unsigned)mbx::kStageWaves) {                      \
                mbx::g_stage_buf[blockIdx.x * 16u + (unsigned)(i)] = (unsigned)(t_ - mbx::s_stage_prev);   \
            }                                                                              \
            mbx::s_stage_prev = t_;                                                        \
        }                                                                                  \
    } while (0)
#else
#define MBX_TS(i) do { } while (0)
#endif
// finer marks inside the synthesiser's first stage (MBX_STAGE_TIMES builds with -DMBX_STAGE_SUB: they reuse the marks of the in-wave front end,
// which a one-launch kernel's stream blocks do not pass): 14 after smoothing + noise, 15 after the phases, 10 after the interpolated branch
#if defined(MBX_STAGE_TIMES) && defined(MBX_STAGE_SUB)
#define MBX_TSX(i) MBX_TS(i)
#else
#define MBX_TSX(i) do { } while (0)
#endif
#ifdef MBX_FRAME_STAMPS
namespace mbx { __device__ unsigned long long g_frame_stamps[16]; }
#define MBX_FSTAMP(i) do { g_frame_stamps[i] = wall_clock64(); } while (0)
#endif
// stage marks inside the front blocks (mbx_front_imbe.h, mbx_expand_imbe.h), MBX_STAGE_TIMES builds only; `v`: a value the mark waits for
#ifdef MBX_STAGE_TIMES
#define MBX_FTS(i, v) do { asm volatile("" :: "v"(v)); MBX_TS(i); } while (0)
#else
#define MBX_FTS(i, v) do { } while (0)
#endif
#include "mbx_fec_frame.h"
#include "mbx_front_imbe.h"

// Wave priorities (s_setprio): a wave raises its priority while it is in one of the two VALU-dense stretches of a frame -- the
// voiced bank's harmonic loop and the unvoiced transform pair -- and (IMBE launches with several frames per stream) in the front
// part of a frame, whose table look-ups should go out as early as possible.  With five or six waves per SIMD in different
// phases of their frames the arbiter otherwise serves them oldest first, whatever they are doing.  Measured by interleaved
// A/B on one box (round 4): 65,536 x 16 IMBE -1.7 %, 8,192 x 128 AMBE+2 -3.6 %; the one-frame-per-stream launches do not care
// (and lose 1-6 % when a new wave's loads or a finishing wave's stores are given priority: tried, dropped); the AMBE bodies
// lose 1-4 % with a raised priority in their front part or in the eight-frame expansion pass: not there.  Raising the rest of the
// synthesiser (noise / phases, between bank and transform, overlap-add) as well costs 0.2-3.6 %: the point is the contrast.
#ifndef MBX_AMBE_GATHER_STORES
#define MBX_AMBE_GATHER_STORES 1   // gathered header stores in the one-frame AMBE instances too: 78 VGPRs = six waves, and still 1.7 % faster than seven waves with fourteen one-lane stores per struct (with a 16 B spill at seven: 2.4 % slower)
#endif
#ifndef MBX_PRIO_BANK
#define MBX_PRIO_BANK 2
#endif
#ifndef MBX_PRIO_FFT
#define MBX_PRIO_FFT 1
#endif
#ifndef MBX_PRIO_FRONT_IMBE
#define MBX_PRIO_FRONT_IMBE 3
#endif
#ifndef MBX_FLAT_LOADS
#define MBX_FLAT_LOADS 1   // one-launch T = 1 kernels: unconditional struct loads (see load_parms_arrays)
#endif
#ifndef MBX_EARLY_NOISE
#define MBX_EARLY_NOISE 1   // the next overlap's jump-ahead constants requested ahead of the voiced bank (synth_core)
#endif
#ifndef MBX_PRIO_FRONT_BLOCK
#define MBX_PRIO_FRONT_BLOCK 3
#endif
#ifndef MBX_AMBE_EARLY_NOISE
#define MBX_AMBE_EARLY_NOISE 0     // 1: the AMBE one-frame instances request the next overlap's jump-ahead constants ahead of the bank too (kEarly);
                                   // off since round 6: its two registers across the bank are what kept those instances at six waves per SIMD
#endif
#ifndef MBX_RES_VIEW_FROM_CUR
#define MBX_RES_VIEW_FROM_CUR 1    // 0 (A/B builds only): resident one-frame instances load prev_mp_enhanced's view separately (see enh_view_of)
#endif
#ifndef MBX_BANK_TRIM
#define MBX_BANK_TRIM 1            // 0 (A/B builds only): the voiced bank's loop always starts at harmonic 1
#endif
#ifndef MBX_BANK_ALWAYS_DRIFT
#define MBX_BANK_ALWAYS_DRIFT 0    // 1 (A/B builds only): the voiced bank always forms its first-order drift sums, as up to round 5
#endif
#ifndef MBX_PARK_N
#define MBX_PARK_N 8   // how many per-lane values wait in LDS across the unvoiced transform pair (synth_core)
#endif
#ifndef MBX_STREAM_WAVES_PER_SIMD
#define MBX_STREAM_WAVES_PER_SIMD 7   // occupancy target of the IMBE stream kernel (caps VGPRs at 72; 4,624 B of LDS per wave)
#endif

namespace mbx {

// ------------------------------------------------------------------------------------------
// Register image of one mbe_parms.
// ------------------------------------------------------------------------------------------
struct Parms {
    float    w0;
    int      L, K;
    int      Vl;       // lane l: Vl[l]
    float    Ml, log2Ml, PHIl, PSIl;
    float    gamma;
    uint32_t tonePhase;
    int      swn;
    float    localEnergy;
    int      amplitudeThreshold;
    float    errorRate;
    int      errorCountTotal, errorCount4, repeatCount;
    float    mutingThreshold;
    float    uw[4];    // previousUw[lane + 64*j]
    float    noiseSeed;
    float    ov[2];    // noiseOverlap[lane + 64*j]  (j = 1: lanes 0..31)
};

// Wave-uniform values (every scalar field of mbe_parms) are pinned to SGPRs: the VGPR file is what
// limits occupancy here, and a scalar costs 1/64 of a VGPR even when it has to be spilled.
__device__ __forceinline__ float uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// dword offsets inside mbe_parms (include/mbx_types.h asserts the byte offsets)
enum : int {
    O_W0 = 0, O_L = 1, O_K = 2, O_VL = 3, O_ML = 60, O_LOG2ML = 117, O_PHI = 174, O_PSI = 231, O_GAMMA = 288,
    O_TONEPHASE = 289, O_SWN = 290, O_LOCALENERGY = 291, O_AMPTHR = 292, O_ERRORRATE = 293, O_ERRTOTAL = 294,
    O_ERR4 = 295, O_REPEAT = 296, O_MUTETHR = 297, O_UW = 298, O_NOISESEED = 554, O_OVERLAP = 555, PARMS_DWORDS = 651
};

// The fourteen wave-uniform fields of a struct come in ONE vector load: lane j fetches header field j (H_*), and the values
// are read out of that register with v_readlane once they are needed.  Written as `uni(f[O_W0])` etc. every field was a load
// of its own followed by a v_readfirstlane -- and because vector memory returns in order and the read-out waits for its load,
// every group of them was a full memory round trip in the middle of the struct's other requests: six to seven serialised
// round trips at the start of every wave of a one-frame launch (and as many PCIe round trips in the single-frame kernels).
// Now every request of a launch's first frame is issued before the first wait.
enum : int {
    H_W0 = 0, H_L = 1, H_K = 2, H_GAMMA = 3, H_TONEPHASE = 4, H_SWN = 5, H_LOCALENERGY = 6, H_AMPTHR = 7, H_ERRORRATE = 8,
    H_ERRTOTAL = 9, H_ERR4 = 10, H_REPEAT = 11, H_MUTETHR = 12, H_NOISESEED = 13
};
__device__ __forceinline__ uint32_t load_header(const mbe_parms* __restrict__ p, int lane) {
    const int j = lane & 15;   // (the lanes above 13 repeat a field: same cache lines, no mask to set up)
    const int idx = (j < 3) ? j : ((j < 13) ? (O_GAMMA - H_GAMMA) + j : O_NOISESEED);
    static_assert(O_MUTETHR - O_GAMMA == H_MUTETHR - H_GAMMA, "fields 3..12 of the header are contiguous in mbe_parms");
    return reinterpret_cast<const uint32_t*>(p)[idx];
}
__device__ __forceinline__ int hdr_i(uint32_t h, int k) { return __builtin_amdgcn_readlane((int)h, k); }
__device__ __forceinline__ float hdr_f(uint32_t h, int k) { return __int_as_float(__builtin_amdgcn_readlane((int)h, k)); }

// kFlat: every load unconditional (lanes past the 57 band slots read the dwords that follow inside the same struct and drop them;
// the short second half of the noise overlap reads lane & 31).  A load under `band ? ... : 0` is compiled as a branch around it, and
// the compiler's s_waitcnt bookkeeping must then assume the path that skipped it: a wait for an OLDER load is given a count that
// leaves only the unconditional younger ones outstanding -- with forty-odd loads in flight at the start of a one-launch wave that is
// the difference between waiting for the table reads and waiting for nearly the whole state.
template <bool kFlat = false>
__device__ __forceinline__ void load_parms_arrays(Parms& r, const mbe_parms* __restrict__ p, int lane) {
    const float* f = reinterpret_cast<const float*>(p);
    const int* i = reinterpret_cast<const int*>(p);
    const bool band = lane < MBX_BAND_SLOTS;
    if constexpr (kFlat) {
        const int vl = i[O_VL + lane];
        const float ml = f[O_ML + lane], l2 = f[O_LOG2ML + lane], ph = f[O_PHI + lane], ps = f[O_PSI + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r.uw[j] = f[O_UW + lane + 64 * j];
        }
        r.ov[0] = f[O_OVERLAP + lane];
        const float o1 = f[O_OVERLAP + 64 + (lane & 31)];
        r.Vl = band ? vl : 0;
        r.Ml = band ? ml : 0.0f;
        r.log2Ml = band ? l2 : 0.0f;
        r.PHIl = band ? ph : 0.0f;
        r.PSIl = band ? ps : 0.0f;
        r.ov[1] = (lane < 32) ? o1 : 0.0f;
        return;
    }
    r.Vl = band ? i[O_VL + lane] : 0;
    r.Ml = band ? f[O_ML + lane] : 0.0f;
    r.log2Ml = band ? f[O_LOG2ML + lane] : 0.0f;
    r.PHIl = band ? f[O_PHI + lane] : 0.0f;
    r.PSIl = band ? f[O_PSI + lane] : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r.uw[j] = f[O_UW + lane + 64 * j];
    }
    r.ov[0] = f[O_OVERLAP + lane];
    r.ov[1] = (lane < 32) ? f[O_OVERLAP + 64 + lane] : 0.0f;
}
__device__ __forceinline__ void set_parms_header(Parms& r, uint32_t h) {
    r.w0 = hdr_f(h, H_W0);
    r.L = hdr_i(h, H_L);
    r.K = hdr_i(h, H_K);
    r.gamma = hdr_f(h, H_GAMMA);
    r.tonePhase = (uint32_t)hdr_i(h, H_TONEPHASE);
    r.swn = hdr_i(h, H_SWN);
    r.localEnergy = hdr_f(h, H_LOCALENERGY);
    r.amplitudeThreshold = hdr_i(h, H_AMPTHR);
    r.errorRate = hdr_f(h, H_ERRORRATE);
    r.errorCountTotal = hdr_i(h, H_ERRTOTAL);
    r.errorCount4 = hdr_i(h, H_ERR4);
    r.repeatCount = hdr_i(h, H_REPEAT);
    r.mutingThreshold = hdr_f(h, H_MUTETHR);
    r.noiseSeed = hdr_f(h, H_NOISESEED);
}
// The whole struct in one call, every scalar field by a load of its own: for the places that are not at the start of a launch
// (a repeat copying prev_mp, copies between slots, the other kernels).  Measured: with the gathered header here too, and in the
// start of the LDS-resident AMBE+2 instances, ambe_stream_kernel_lds is 6.7 % SLOWER over a 128-frame launch (8,192 x 128:
// 2.63 -> 2.81 ms; same instruction counts -- register allocation of the frame loop), so those keep this form.
template <bool kGather = false>
__device__ __forceinline__ void load_parms(Parms& r, const mbe_parms* __restrict__ p, int lane) {
    if constexpr (kGather) {   // the one-frame instances: their register budget (72) is met with this form and not with the other
        const uint32_t h = load_header(p, lane);
        load_parms_arrays(r, p, lane);
        set_parms_header(r, h);
        return;
    }
    const float* f = reinterpret_cast<const float*>(p);
    const int* i = reinterpret_cast<const int*>(p);
    r.w0 = uni(f[O_W0]);
    r.L = uni(i[O_L]);
    r.K = uni(i[O_K]);
    load_parms_arrays(r, p, lane);
    r.gamma = uni(f[O_GAMMA]);
    r.tonePhase = (uint32_t)uni(i[O_TONEPHASE]);
    r.swn = uni(i[O_SWN]);
    r.localEnergy = uni(f[O_LOCALENERGY]);
    r.amplitudeThreshold = uni(i[O_AMPTHR]);
    r.errorRate = uni(f[O_ERRORRATE]);
    r.errorCountTotal = uni(i[O_ERRTOTAL]);
    r.errorCount4 = uni(i[O_ERR4]);
    r.repeatCount = uni(i[O_REPEAT]);
    r.mutingThreshold = uni(f[O_MUTETHR]);
    r.noiseSeed = uni(f[O_NOISESEED]);
}

__device__ __forceinline__ int write_lane(int v, int uniform_value, int k) {   // v with lane k replaced (v_writelane_b32)
    asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane(uniform_value)), "n"(k));
    return v;
}
// kGather: the fourteen scalar fields leave in ONE store (lane j writes header field j, as load_header reads them) instead of
// fourteen one-lane stores -- for the instances whose structs live in HBM (a one-lane store occupies the memory pipeline like any other)
// kNt: NON-TEMPORAL stores (`nt`: the lines are not kept in L2 / the Infinity Cache).  A one-frame launch writes 7.8 KB of state per
// stream that nothing reads again before the next launch has swept a gigabyte: written as ordinary stores those lines push the
// lines the launch is still READING out of the caches.  Measured, interleaved A/B (EXPERIMENTS.md 5.7): 65,536 x 1 IMBE -5.1 ... -9.9 % by box,
// on resident state -1.9 ... -2.5 %, AMBE+2 -2.1 ... -5.2 %; long launches unchanged, SLICED launches +0.8 % (a slice finds the state of the
// slice before it in the caches) -- so only the one-frame instances (= the gathered-header ones) store this way.  Non-temporal
// LOADS of the state: +5 % (kept plain).  Stores to a caller's pinned host structs (the per-frame API) stay plain.
template <bool kNt, class T>
__device__ __forceinline__ void st_state(T* p, T v) {
    if constexpr (kNt) {
        __builtin_nontemporal_store(v, p);
    } else {
        *p = v;
    }
}
template <bool kGather = false, bool kNt = kGather>
__device__ __forceinline__ void store_parms(const Parms& r, mbe_parms* __restrict__ p, int lane) {
    float* f = reinterpret_cast<float*>(p);
    int* i = reinterpret_cast<int*>(p);
    if constexpr (kGather) {
        int h = 0;
        h = write_lane(h, __float_as_int(r.w0), H_W0);
        h = write_lane(h, r.L, H_L);
        h = write_lane(h, r.K, H_K);
        h = write_lane(h, __float_as_int(r.gamma), H_GAMMA);
        h = write_lane(h, (int)r.tonePhase, H_TONEPHASE);
        h = write_lane(h, r.swn, H_SWN);
        h = write_lane(h, __float_as_int(r.localEnergy), H_LOCALENERGY);
        h = write_lane(h, r.amplitudeThreshold, H_AMPTHR);
        h = write_lane(h, __float_as_int(r.errorRate), H_ERRORRATE);
        h = write_lane(h, r.errorCountTotal, H_ERRTOTAL);
        h = write_lane(h, r.errorCount4, H_ERR4);
        h = write_lane(h, r.repeatCount, H_REPEAT);
        h = write_lane(h, __float_as_int(r.mutingThreshold), H_MUTETHR);
        h = write_lane(h, __float_as_int(r.noiseSeed), H_NOISESEED);
        if (lane < 14) {
            const int idx = (lane < 3) ? lane : ((lane < 13) ? (O_GAMMA - H_GAMMA) + lane : O_NOISESEED);
            st_state<kNt>(&i[idx], (int)(h));
        }
    } else if (lane == 0) {
        st_state<kNt>(&f[O_W0], (float)(r.w0));
        st_state<kNt>(&i[O_L], (int)(r.L));
        st_state<kNt>(&i[O_K], (int)(r.K));
        st_state<kNt>(&f[O_GAMMA], (float)(r.gamma));
        st_state<kNt>(&i[O_TONEPHASE], (int)((int)r.tonePhase));
        st_state<kNt>(&i[O_SWN], (int)(r.swn));
        st_state<kNt>(&f[O_LOCALENERGY], (float)(r.localEnergy));
        st_state<kNt>(&i[O_AMPTHR], (int)(r.amplitudeThreshold));
        st_state<kNt>(&f[O_ERRORRATE], (float)(r.errorRate));
        st_state<kNt>(&i[O_ERRTOTAL], (int)(r.errorCountTotal));
        st_state<kNt>(&i[O_ERR4], (int)(r.errorCount4));
        st_state<kNt>(&i[O_REPEAT], (int)(r.repeatCount));
        st_state<kNt>(&f[O_MUTETHR], (float)(r.mutingThreshold));
        st_state<kNt>(&f[O_NOISESEED], (float)(r.noiseSeed));
    }
    if (lane < MBX_BAND_SLOTS) {
        st_state<kNt>(&i[O_VL + lane], (int)(r.Vl));
        st_state<kNt>(&f[O_ML + lane], (float)(r.Ml));
        st_state<kNt>(&f[O_LOG2ML + lane], (float)(r.log2Ml));
        st_state<kNt>(&f[O_PHI + lane], (float)(r.PHIl));
        st_state<kNt>(&f[O_PSI + lane], (float)(r.PSIl));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        st_state<kNt>(&f[O_UW + lane + 64 * j], (float)(r.uw[j]));
    }
    st_state<kNt>(&f[O_OVERLAP + lane], (float)(r.ov[0]));
    if (lane < 32) {
        st_state<kNt>(&f[O_OVERLAP + 64 + lane], (float)(r.ov[1]));
    }
}

// Partial loads.  The stream kernels do not need every field of the previous structs:
//   * decode reads from prev_mp only the prediction memory (Ml, log2Ml), L, errorRate, repeatCount,
//     gamma, mutingThreshold (+ PHIl[0], which AMBE's log2Ml[57] aliases); the rest of prev_mp
//     matters only on a repeat / erasure, where it is fetched on demand;
//   * synthesis reads from prev_mp_enhanced only w0, L, Vl, Ml, PHIl, PSIl, the second half of
//     previousUw and the two smoothing memories.
// Unread fields stay zero and are optimised away: fewer registers, and ~40 % less state read traffic.
template <bool kFlat = false>
__device__ __forceinline__ void load_prev_arrays(Parms& r, const mbe_parms* __restrict__ p, int lane) {
    const float* f = reinterpret_cast<const float*>(p);
    const bool band = lane < MBX_BAND_SLOTS;
    r = Parms{};
    if constexpr (kFlat) {
        const float ml = f[O_ML + lane], l2 = f[O_LOG2ML + lane], ph = f[O_PHI + lane];
        r.Ml = band ? ml : 0.0f;
        r.log2Ml = band ? l2 : 0.0f;
        r.PHIl = band ? ph : 0.0f;
        return;
    }
    r.Ml = band ? f[O_ML + lane] : 0.0f;
    r.log2Ml = band ? f[O_LOG2ML + lane] : 0.0f;
    r.PHIl = band ? f[O_PHI + lane] : 0.0f;
}
__device__ __forceinline__ void set_prev_header(Parms& r, uint32_t h) {
    r.L = hdr_i(h, H_L);
    r.gamma = hdr_f(h, H_GAMMA);
    r.errorRate = hdr_f(h, H_ERRORRATE);
    r.repeatCount = hdr_i(h, H_REPEAT);
    r.mutingThreshold = hdr_f(h, H_MUTETHR);
}
__device__ __forceinline__ void load_prev_view(Parms& r, const mbe_parms* __restrict__ p, int lane) {
    const uint32_t h = load_header(p, lane);
    load_prev_arrays(r, p, lane);
    set_prev_header(r, h);
}

// the same from the LDS copy of prev_mp (LDS-resident instances, once per frame): broadcast reads, nothing to reorder
__device__ __forceinline__ void load_prev_view_lds(Parms& r, const mbe_parms* __restrict__ p, int lane) {
    const float* f = reinterpret_cast<const float*>(p);
    const int* i = reinterpret_cast<const int*>(p);
    load_prev_arrays(r, p, lane);
    r.L = uni(i[O_L]);
    r.gamma = uni(f[O_GAMMA]);
    r.errorRate = uni(f[O_ERRORRATE]);
    r.repeatCount = uni(i[O_REPEAT]);
    r.mutingThreshold = uni(f[O_MUTETHR]);
}

template <bool kFlat = false>
__device__ __forceinline__ void load_enh_arrays(Parms& r, const mbe_parms* __restrict__ p, int lane) {
    const float* f = reinterpret_cast<const float*>(p);
    const int* i = reinterpret_cast<const int*>(p);
    const bool band = lane < MBX_BAND_SLOTS;
    r = Parms{};
    if constexpr (kFlat) {
        const int vl = i[O_VL + lane];
        const float ml = f[O_ML + lane], ph = f[O_PHI + lane], ps = f[O_PSI + lane];
        r.uw[2] = f[O_UW + lane + 128];
        r.uw[3] = f[O_UW + lane + 192];
        r.Vl = band ? vl : 0;
        r.Ml = band ? ml : 0.0f;
        r.PHIl = band ? ph : 0.0f;
        r.PSIl = band ? ps : 0.0f;
        return;
    }
    r.Vl = band ? i[O_VL + lane] : 0;
    r.Ml = band ? f[O_ML + lane] : 0.0f;
    r.PHIl = band ? f[O_PHI + lane] : 0.0f;
    r.PSIl = band ? f[O_PSI + lane] : 0.0f;
    r.uw[2] = f[O_UW + lane + 128];
    r.uw[3] = f[O_UW + lane + 192];
}
__device__ __forceinline__ void set_enh_header(Parms& r, uint32_t h) {
    r.w0 = hdr_f(h, H_W0);
    r.L = hdr_i(h, H_L);
    r.localEnergy = hdr_f(h, H_LOCALENERGY);
    r.amplitudeThreshold = hdr_i(h, H_AMPTHR);
}
__device__ __forceinline__ void load_enh_view(Parms& r, const mbe_parms* __restrict__ p, int lane) {
    const uint32_t h = load_header(p, lane);
    load_enh_arrays(r, p, lane);
    set_enh_header(r, h);
}

// Resident one-frame instances: while prev_mp_enhanced is elided it EQUALS cur_mp, and cur_mp is being loaded whole anyway -- the
// enhanced model's view is then taken from cur_mp's registers instead of a second set of loads from the same struct (seven vector
// loads per wave less), and the resident flag, a scalar load at the very top of the wave, is first needed AFTER every other request
// of the wave has gone out (it used to pick the address of the third load: one exposed round trip before the other twenty).
// ... and so do the five scalars the decode reads of prev_mp: while the flag is set, prev_mp is the snapshot the last frame took of
// cur_mp after its decode (voice frames: `mbe_moveMbeParms(cur, prev)` before the enhancement, ref src/imbe/imbe7200x4400.c:835,
// src/ambe/ambe3600x2450.c:789; erasures and re-initialisations copy the whole struct), and nothing after the snapshot touches L, gamma,
// errorRate, repeatCount or mutingThreshold -- so they are cur_mp's, and prev_mp's two header lines are not fetched at all.
__device__ __forceinline__ void prev_header_of(Parms& r, const Parms& cur) {
    r.L = cur.L;
    r.gamma = cur.gamma;
    r.errorRate = cur.errorRate;
    r.repeatCount = cur.repeatCount;
    r.mutingThreshold = cur.mutingThreshold;
}
__device__ __forceinline__ void enh_view_of(Parms& r, const Parms& cur) {
    r = Parms{};
    r.Vl = cur.Vl;
    r.Ml = cur.Ml;
    r.PHIl = cur.PHIl;
    r.PSIl = cur.PSIl;
    r.uw[2] = cur.uw[2];
    r.uw[3] = cur.uw[3];
}

// the synthesis-continuity fields an AMBE erasure keeps from prev_mp
__device__ __forceinline__ void load_continuity(Parms& r, const mbe_parms* __restrict__ p, int lane) {
    const float* f = reinterpret_cast<const float*>(p);
    const bool band = lane < MBX_BAND_SLOTS;
    r.PHIl = band ? f[O_PHI + lane] : 0.0f;
    r.PSIl = band ? f[O_PSI + lane] : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r.uw[j] = f[O_UW + lane + 64 * j];
    }
    r.noiseSeed = uni(f[O_NOISESEED]);
    r.ov[0] = f[O_OVERLAP + lane];
    r.ov[1] = (lane < 32) ? f[O_OVERLAP + 64 + lane] : 0.0f;
}

// ------------------------------------------------------------------------------------------
// Per-wave LDS scratch.
// ------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));   // two samples per lane: v_pk_{mul,fma}_f32
__device__ __forceinline__ v2f splat(float x) { return v2f{x, x}; }

template <int kParkColsT>
struct WaveScratchT {
    static constexpr int kParkCols = kParkColsT;
    union {
        struct {                       // voiced bank, harmonic l: g (cos psi, sin psi) and d g (sin psi, cos psi),
            alignas(16) float4 coef_amp[64];     //   each as (prev, cur) pairs = the operands of the packed FMAs
            alignas(16) float4 coef_drift[64];
            alignas(16) float4 icoef[8];   // interpolated low harmonics: (phi_prev, w0 l + dw, 2 M_prev, 2 dM)
        };                                 // (the windowed bank output passes through coef_amp's bytes afterwards)
        struct {                       // in-wave parameter expansion (dead before synthesis starts)
            uint32_t words[64];        //   parameter words b_0..b_57
            float    gains[12];        //   IMBE G_1..G_6 / AMBE+2 R_1..R_8
            float    C[84];            //   DCT coefficients C[block][k] at block * 12 + k
            float    fp[64];           //   the frame's FrameParams row
        } x;
        struct {                       // unvoiced path (the coefficients are dead by then)
            float2 fft[256];           //   in-place radix-4 FFT
            float  bins[144];          //   |X(k)|^2, k = 0..128, then the per-bin scale in the same place (+ slack: see the band sums)
        };
    };
    // per-lane values that only cross the unvoiced transform pair (see synth_core); none in the LDS-resident kernel
    // instances: they run at the occupancy LDS allows and have the registers to carry those values themselves
    float park[kParkColsT > 0 ? kParkColsT : 1][kParkColsT > 0 ? 64 : 4];
};
using WaveScratch = WaveScratchT<MBX_PARK_N>;

// Sum of the per-lane terms v_1 .. v_L in INDEX ORDER with a float rounding after every addition -- the reference's
// `for (l = 1; l <= L; l++) sum += v[l]`, bit for bit, returned wave-uniform.  Used where a sum feeds state that later
// frames build on (the log-magnitude prediction memory, the local-energy filter): a wave-parallel tree sum is a few ulp
// away from the sequential one, the difference is carried from frame to frame, and float-threshold decisions downstream
// could then fall on the other side than the reference's.
// The fold runs in registers: every step is ONE v_add_f32 whose first operand comes from the lane below (DPP wave_shr:1),
// acc[l] <- acc[l-1] + v[l], so after t steps lane l holds the left fold of v[l-t .. l]; lane L after L - 1 steps holds
// ((v_1 + v_2) + v_3) + ... + v_L (lanes below 1 contribute +0.0f, and 0 + v_1 is v_1), read back with one v_readlane.
// L - 1 dependent additions like the reference's loop, and no LDS traffic at all (round 2 sent the terms through LDS and
// read them back as broadcast ds_read_b128: 4 LDS-array cycles per 4 terms in a kernel whose LDS pipe is its busiest unit).
// v must be 0.0f in lanes outside 1..L.  `tmp` (unused) keeps the call sites' shape.
// the unvoiced-noise generator x -> 171 x + 11213 mod 53125 (ref src/core/mbe_unvoiced_fft.c), n steps at once: x_n = mul x_0 + add
constexpr uint32_t lcg_mul_steps(int n) {
    uint64_t a = 1;
    for (int k = 0; k < n; ++k) {
        a = (a * 171u) % 53125u;
    }
    return (uint32_t)a;
}
constexpr uint32_t lcg_add_steps(int n) {
    uint64_t c = 0;
    for (int k = 0; k < n; ++k) {
        c = (c * 171u + 11213u) % 53125u;
    }
    return (uint32_t)c;
}
constexpr int kDppWaveShr1 = 0x138;   // gfx9 DPP: whole-wave shift right by one lane
__device__ __forceinline__ float seq_sum4(float v, int L, float* /*tmp*/, int /*lane*/) {
    float acc = v;
    const int n = uni(L);
    auto step = [&]() {
        const float below = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), kDppWaveShr1, 0xf, 0xf, true));
        acc = below + v;
    };
    int t = n - 1;
    for (; t >= 4; t -= 4) {   // four steps per trip: the loop control is scalar work in the shadow of the DPP hazard slots
        step();
        step();
        step();
        step();
    }
    for (; t > 0; --t) {
        step();
    }
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), n & 63));   // every caller has 1 <= L <= 56
}

struct StreamRng {   // register copy of mbx_stream_rng (wave-uniform)
    unsigned long long cn_seed48;
    uint32_t           cn_seeded, unv_state, unv_override;
};

// ------------------------------------------------------------------------------------------
// IMBE 7200x4400, stateless half of the parameter decode for ONE frame on the whole wave
// (ref src/imbe/imbe7200x4400.c:117-270): fundamental, bit layout, voicing, gains, higher-order
// coefficients, per-block inverse DCT.  Writes the FrameParams row (layout: mbx_expand.hip) to
// S.x.fp.  lane = payload bit for the scatter, = gain / coefficient index for the dequantisation,
// = harmonic for the inverse DCT; every sum keeps the reference's sequential order.
//
// The record and every wave-uniform table entry (w0/L/K of b0, the block lengths) are read with SCALAR
// loads (constant address space): they do not queue behind the state loads the caller has in flight.
// Every per-lane table address then depends only on those scalars and the lane, and the three small
// codebooks (B2, quantstep) are held one entry per lane, so there is a single round of vector-memory
// latency instead of a chain of dependent lookups.
// ------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) mbx_tables* ConstTables;
typedef const __attribute__((address_space(4))) DerivedTables* ConstDerived;
// The table pointers pass through an asm barrier once per frame (see the kernels), after which the compiler no longer
// knows that they point to global memory and would emit FLAT loads -- which count against the LDS counter as well and
// make every LDS wait also a wait for outstanding table reads.  Vector table reads therefore go through pointers that
// say "global" explicitly.
#define MBX_GLOBAL __attribute__((address_space(1)))
typedef const MBX_GLOBAL mbx_tables* GlobalTables;
typedef const MBX_GLOBAL DerivedTables* GlobalDerived;
typedef const MBX_GLOBAL float* GlobalFloats;
__device__ __forceinline__ uint32_t low_bits(uint32_t v, int n) { return v & ((1u << n) - 1u); }

// One table entry at (wave-uniform byte address) + (per-lane 32-bit byte offset): written this way the load is a
// global_load with an SGPR base and a 32-bit VGPR offset.  Indexing a multi-dimensional table member with a uniform row and a
// per-lane column made the compiler form 64-bit per-lane addresses instead (v_lshl_add_u64, v_add_co / v_addc, v_mad_i64_i32:
// ~30 VALU instructions a frame in the IMBE expansion alone, several of them double-rate ones).
template <class T>
__device__ __forceinline__ T tab_at(const void* uniform_base, size_t uniform_bytes, uint32_t lane_bytes) {
    const MBX_GLOBAL char* p = (const MBX_GLOBAL char*)uniform_base + uniform_bytes;
    asm("" : "+s"(p));   // the uniform sum stays ONE scalar value: otherwise a member offset too large for the instruction's
                         // immediate field is added to the per-lane part again, in 64-bit vector arithmetic
    return *(const MBX_GLOBAL T*)(p + lane_bytes);
}

// a frame's parameter record by SCALAR loads (wave-uniform address)
__device__ __forceinline__ uint4 load_record_scalar(const mbx_param_record* rp) {
    const __attribute__((address_space(4))) uint32_t* rq = (const __attribute__((address_space(4))) uint32_t*)rp;
    return make_uint4(rq[0], rq[1], rq[2], rq[3]);
}

// The expansion in two halves: expand_imbe_request() issues every table read (they depend on b0 / L and the lane only, not on
// the rest of the record), expand_imbe_finish() does the arithmetic once the record is there.  The one-launch T = 1 kernels put
// the second half of the frame's FEC between the two.
struct ImbeExpandReq {
    int      bad, L;
    uint32_t w0_bits, lk;   // the b0 entry as loaded (w0; L | K << 8): decoded in the second half
    uint32_t e0, e1, own;
    v2f      bas;   // (bit count, step) of the lane's gain
    float    b2, qs, sd;
    int      Bm;
    float    cosr[11], ric[7];
};
// L_known >= 0: the caller has L already (from the lane-held b0 -> L law, DerivedTables::imbe_L_lanes) -- the table reads then do
// not wait for the scalar load of (w0, L, K), which is only needed by the second half.
__device__ __forceinline__ void expand_imbe_request(ImbeExpandReq& q, int b0, const mbx_tables* Tgen, const DerivedTables* Dgen, int lane,
                                                    int L_known = -1) {
    q.bad = 1;
    q.L = 0;
    q.w0_bits = 0u;
    q.lk = 0u;
    if (b0 <= 207) {   // ONE scalar load (host-made table: w0 bits, L | K << 8); byte-wide table entries would be vector loads,
        const __attribute__((address_space(4))) uint32_t* e =   // which queue behind whatever the wave has in flight
            (const __attribute__((address_space(4))) uint32_t*)&((ConstDerived)Dgen)->imbe_b0[b0];
        q.w0_bits = e[0];
        q.lk = e[1];
        q.L = (L_known >= 0) ? L_known : (int)(q.lk & 0xffu);
        q.bad = (q.L == 0) ? 1 : 0;   // the reference has stored w0 but not L in this case
    }
    if (!q.bad) {
        const int L9 = q.L - 9;
        // ---- every per-lane table value is requested here ----
        const uint32_t ul = (uint32_t)lane;
        const size_t uL9 = (size_t)(uint32_t)L9;
        q.e0 = tab_at<uint16_t>(Tgen, offsetof(mbx_tables, imbe_bo) + uL9 * sizeof(Tgen->imbe_bo[0]), 2u * ul);
        q.e1 = tab_at<uint16_t>(Tgen, offsetof(mbx_tables, imbe_bo) + uL9 * sizeof(Tgen->imbe_bo[0]), 2u * (lane < 15 ? ul + 64u : 78u));
        const uint32_t g = (lane >= 2 && lane <= 6) ? ul - 2u : 0u;
        q.bas = tab_at<v2f>(Tgen, offsetof(mbx_tables, imbe_ba) + uL9 * sizeof(Tgen->imbe_ba[0]), 8u * g);
        q.b2 = tab_at<float>(Tgen, offsetof(mbx_tables, imbe_B2), 4u * ul);
        q.qs = tab_at<float>(Tgen, offsetof(mbx_tables, imbe_quantstep), 4u * (lane < 11 ? ul : 0u));
        // which higher-order coefficient / harmonic a lane owns depends on L only: host-made tables (mbx_init)
        q.own = tab_at<uint32_t>(Dgen, offsetof(DerivedTables, imbe_lane_map) + uL9 * sizeof(Dgen->imbe_lane_map[0]), 4u * ul);
        q.Bm = tab_at<uint8_t>(Tgen, offsetof(mbx_tables, imbe_hoba) + uL9 * sizeof(Tgen->imbe_hoba[0]), lane < 50 ? ul : 0u);
        q.sd = tab_at<float>(Dgen, offsetof(DerivedTables, imbe_hoc_sd) + uL9 * sizeof(Dgen->imbe_hoc_sd[0]), 4u * ul);
        {   // fetched now: loads cannot move up across the LDS fences of the second half
            const size_t rows = offsetof(DerivedTables, imbe_idct_rows) + uL9 * sizeof(Dgen->imbe_idct_rows[0]);
            typedef float v4f_u __attribute__((ext_vector_type(4), aligned(8)));   // a row is 40 bytes: 8-byte aligned
            const v4f_u r0 = tab_at<v4f_u>(Dgen, rows, 40u * ul), r1 = tab_at<v4f_u>(Dgen, rows + 16u, 40u * ul);
            const v2f r2 = tab_at<v2f>(Dgen, rows + 32u, 40u * ul);
            q.cosr[1] = r0.x, q.cosr[2] = r0.y, q.cosr[3] = r0.z, q.cosr[4] = r0.w;
            q.cosr[5] = r1.x, q.cosr[6] = r1.y, q.cosr[7] = r1.z, q.cosr[8] = r1.w;
            q.cosr[9] = r2.x, q.cosr[10] = r2.y;
            const uint32_t col = 4u * ((lane >= 1 && lane <= 6) ? ul : 0u);
#pragma unroll
            for (int m = 1; m <= 6; ++m) {
                q.ric[m] = tab_at<float>(Tgen, offsetof(mbx_tables, imbe_ri_cos) + 28u * (size_t)m, col);
            }
        }
    }
}
__device__ __forceinline__ int imbe_record_b0(const uint4 rec) {
    return (int)(((rec.x >> 26) << 2) | ((rec.z >> 9) & 3u));   // payload bits 0..5, 85, 86
}

template <class Scratch>
__device__ __forceinline__ void expand_imbe_finish(const uint4 rec, const ImbeExpandReq& q, Scratch& S, int lane) {
    const int bad = q.bad, L = q.L;
    uint32_t w0_bits = q.w0_bits, lk = q.lk;
    asm volatile("" : "+s"(w0_bits), "+s"(lk));   // (first touched here: see below)
    const int K = (int)(lk >> 8);
    const float w0 = __uint_as_float(w0_bits);
    S.x.words[lane] = 0u;
    S.x.fp[lane] = 0.0f;
    S.x.C[lane] = 0.0f;   // coefficients past a block's length stay zero: the inverse DCT below adds
    if (lane < 20) {      // them unconditionally (x + 0*c == x for every x this sum can take)
        S.x.C[64 + lane] = 0.0f;
    }
    if (!bad) {
        uint32_t e0 = q.e0, e1 = q.e1, own = q.own;
        v2f bas = q.bas;
        float b2 = q.b2, qs = q.qs, sd = q.sd;
        int Bm = q.Bm;
        // the requested values are first touched HERE: arithmetic on them hoisted into the request half would wait for them there,
        // in front of everything the caller issues between the halves
        asm volatile("" : "+v"(e0), "+v"(e1), "+v"(own), "+v"(bas), "+v"(b2), "+v"(qs), "+v"(sd), "+v"(Bm));
        const float nb = bas.x, step = bas.y;
        const int hblk = (int)(own & 7u), hk = (int)((own >> 3) & 15u);
        const int iblk = (int)((own >> 7) & 7u);
        const bool harm = lane >= 1 && lane <= L;
        wave_lds_sync();
        {   // payload bit j + 6 feeds bit e[1] of word e[0] (:156-168)
            const int i0 = lane + 6, i1 = lane + 70;
            const uint32_t word0 = (i0 < 32) ? rec.x : ((i0 < 64) ? rec.y : rec.z);
            const uint32_t m0 = e0 & 0xffu, p0 = e0 >> 8;
            if (m0 < 58u && p0 < 12u) {
                atomicOr(&S.x.words[m0], ((word0 >> (31 - (i0 & 31))) & 1u) << p0);
            }
            const uint32_t word1 = (i1 < 96) ? rec.z : 0u;
            const uint32_t m1 = e1 & 0xffu, p1 = e1 >> 8;
            if (lane < 15 && m1 < 58u && p1 < 12u) {
                atomicOr(&S.x.words[m1], ((word1 >> (31 - (i1 & 31))) & 1u) << p1);
            }
        }
        wave_lds_sync();
        {   // gain G_lane (:190-209); B2 is held one entry per lane
            const float g1 = lane_get(b2, (int)low_bits(S.x.words[2], 6));
            if (lane >= 1 && lane <= 6) {
                float G = g1;
                if (lane != 1) {
                    const int inb = (int)nb;
                    const int bm = (int)low_bits(S.x.words[lane + 1], inb);
                    G = (step * ((float)bm - ldexpf(1.0f, inb - 1) + 0.5f));
                }
                S.x.gains[lane] = G;
            }
        }
        wave_lds_sync();
        if (lane >= 1 && lane <= 6) {   // block mean R_lane: 6-point inverse DCT of the gains (:211-231)
            float sum = 0;
#pragma unroll
            for (int m = 1; m <= 6; ++m) {
                const float am = (m == 1) ? 1.0f : 2.0f;
                sum = sum + (am * S.x.gains[m] * q.ric[m]);
            }
            S.x.C[lane * 12 + 1] = sum;
        }
        {   // higher-order coefficient of word m = lane + 8 (:233-249)
            const float qv = lane_get(qs, Bm > 0 ? Bm - 1 : 0);
            if (lane < L - 6) {
                float v = 0.0f;
                if (Bm > 0) {
                    const int bm = (int)low_bits(S.x.words[lane + 8], Bm);
                    v = ((qv * sd) * (((float)bm - ldexpf(1.0f, Bm - 1)) + 0.5f));
                }
                S.x.C[hblk * 12 + hk] = v;
            }
        }
        wave_lds_sync();
        if (harm) {   // per-block inverse DCT (:251-270), lane = harmonic
            const float* C = &S.x.C[iblk * 12];
            float sum = 0;
#pragma unroll
            for (int k = 1; k <= 10; ++k) {
                const float ak = (k == 1) ? 1.0f : 2.0f;
                sum = sum + (ak * C[k] * q.cosr[k]);
            }
            S.x.fp[lane] = sum;
        }
    } else {
        wave_lds_sync();
    }
    // voicing (:170-188): harmonic l = lane takes bit max(K-1 - (l-1)/3, 0) of b1; the mask is a ballot
    unsigned long long v = 0ULL;
    if (!bad) {
        const uint32_t b1 = low_bits(S.x.words[1], 12);
        const int third = ((lane - 1) * 171) >> 9;   // (lane - 1) / 3 for 1 <= lane <= 64
        const int band = (K - 1 - third) < 0 ? 0 : (K - 1 - third);
        v = __ballot(lane >= 1 && lane <= L && ((b1 >> band) & 1u)) >> 1;   // bit l-1 = harmonic l
    }
    if (lane == 0) {
        S.x.fp[57] = __uint_as_float((uint32_t)v);
        S.x.fp[58] = __uint_as_float((uint32_t)(v >> 32));
        S.x.fp[59] = w0;
        S.x.fp[60] = __int_as_float(L);
        S.x.fp[61] = __int_as_float(K);
        S.x.fp[62] = __uint_as_float(rec.w);
        S.x.fp[63] = __int_as_float(bad);
    }
    wave_lds_sync();
}

template <class Scratch>
__device__ void expand_imbe_wave(const uint4 rec, Scratch& S, const mbx_tables* Tgen, const DerivedTables* Dgen, int lane) {
    ImbeExpandReq q;
    expand_imbe_request(q, imbe_record_b0(rec), Tgen, Dgen, lane);
    expand_imbe_finish(rec, q, S, lane);
}

// ------------------------------------------------------------------------------------------
// IMBE 7200x4400 parameter decode.  Returns 0 (voice) or 1 (invalid fundamental).
// Mutates `prev` exactly like the reference (padding of the prediction memory).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int rec_bit(const uint32_t w[3], int i) { return (int)((w[i >> 5] >> (31 - (i & 31))) & 1u); }

// Frame parameters from the expand stage (mbx_expand.hip): v[1..56] prediction residuals T_l,
// v[57..58] voicing bits, v[59] w0, v[60] L, v[61] K (IMBE) / mean residual (AMBE), v[62] error
// context word, v[63] frame class, v[0] AMBE gain increment.  AMBE rows: v[60] = 0.2046 / sqrt(w0), v[63] = class | L << 8.
__device__ int decode_imbe(const float* __restrict__ fp, Parms& cur, Parms& prev, const DerivedTables* Dg, int lane, float* tmp) {
    const int bad = uni(__float_as_int(fp[63]));
    if (bad != 0) {
        if (fp[59] != 0.0f) {
            cur.w0 = fp[59];   // valid b0 with an out-of-range L: the reference has already stored w0
        }
        return 1;
    }
    cur.w0 = uni(fp[59]);
    const int L = uni(__float_as_int(fp[60]));
    cur.L = L;
    cur.K = uni(__float_as_int(fp[61]));
    const unsigned long long vbits = ((unsigned long long)__float_as_uint(fp[58]) << 32) | __float_as_uint(fp[57]);
    float Tl = 0.0f;
    if (lane >= 1 && lane <= L) {
        cur.Vl = (int)((vbits >> (lane - 1)) & 1ULL);
        Tl = fp[lane];
    }

    // log-magnitude prediction
    const float rho = (L <= 15) ? 0.4f : ((L <= 24) ? ((0.03f * (float)L) - 0.05f) : 0.7f);
    const int cur_L = L;   // 9..56
    const int prev_L = prev.L < 1 ? 1 : (prev.L > 56 ? 56 : prev.L);
    {
        const float padM = lane_read(prev.Ml, prev_L), padL = lane_read(prev.log2Ml, prev_L);   // wave-uniform index: v_readlane, no LDS trip
        if (lane > prev_L && lane <= cur_L) {
            prev.Ml = padM;
            prev.log2Ml = padL;
        }
        const float m1 = lane_read(prev.Ml, 1), l1 = lane_read(prev.log2Ml, 1);
        if (lane == 0) {
            prev.Ml = m1;
            prev.log2Ml = l1;
        }
    }
    const ConstDerived D = (ConstDerived)Dg;   // scalar loads
    const float pos = D->l_ratio[uni(prev_L)][cur_L] * (float)lane;
    int lo = (int)pos;
    lo = lo < 0 ? 0 : (lo > 56 ? 56 : lo);
    const float frac = pos - (float)lo;
    const int hi = (lo + 1 > 56) ? 56 : lo + 1;
    const float a = lane_get(prev.log2Ml, lo), b = lane_get(prev.log2Ml, hi);
    const bool in = lane >= 1 && lane <= cur_L;
    float Sum77 = seq_sum4(in ? ((((float)1 - frac) * a) + (frac * b)) : 0.0f, cur_L, tmp, lane);   // feeds the prediction memory
    Sum77 = (D->imbe_rho_over_l[cur_L] * Sum77);
    if (in) {
        const float c1 = (rho * ((float)1 - frac) * a);
        const float c2 = (rho * frac * b);
        cur.log2Ml = Tl + c1 + c2 - Sum77;
        cur.Ml = exp2f(cur.log2Ml);
    }
    return 0;
}

__device__ __forceinline__ void imbe_headroom_reset(Parms& mp, int lane) {
    mp.swn = 0;
    mp.tonePhase = 0;
    mp.w0 = (float)((4.0 * M_PI) / (134.0 + 39.5));
    mp.L = 39;
    mp.K = 12;
    mp.gamma = 0.0f;
    if (lane < MBX_BAND_SLOTS) {
        mp.Vl = 0;
        mp.Ml = 1.0f;
        mp.log2Ml = 0.0f;
    }
    mp.repeatCount = 0;
    mp.localEnergy = 75000.0f;
    mp.amplitudeThreshold = 20480;
    mp.mutingThreshold = MBE_MUTING_THRESHOLD_IMBE;
}

// e^{2 pi i rev}: double-precision reduction, hardware v_cos/v_sin (phase error <= 2.8e-7, mostly the
// rounding of the reduced argument to float; tools/phasor_accuracy.hip)
__device__ __forceinline__ void unit_phasor_hw(double rev, float& c, float& s) {
    const float r = (float)(rev - floor(rev));
    c = __builtin_amdgcn_cosf(r);
    s = __builtin_amdgcn_sinf(r);
}

// The same with the rounding of the reduced argument taken back out: hardware value at the rounded argument,
// first-order correction for the residual (phase error <= 1.3e-7, rms 5.2e-8; tools/phasor_accuracy.hip).
// Used where the error is amplified afterwards.
__device__ __forceinline__ void unit_phasor(double rev, float& c, float& s) {
    const double rr = rev - rint(rev);
    const float r = (float)rr;
    const float lo = (float)((rr - (double)r) * 6.283185307179586);
    const float c0 = __builtin_amdgcn_cosf(r), s0 = __builtin_amdgcn_sinf(r);
    c = fmaf(-lo, s0, c0);
    s = fmaf(lo, c0, s0);
}

// ------------------------------------------------------------------------------------------
// Spectral amplitude enhancement; returns the pre-enhancement Rm0.
// ------------------------------------------------------------------------------------------
__device__ float enhance(Parms& cur, int lane, float* tmp) {
    const int L = cur.L;
    if (L < 1 || L > 56) {
        return 0.0f;
    }
    // cos(l*w0) for lane = l.  The reference rotates (1, 0) l times by w0 (src/core/mbelib.c:412-424, error ~ l * 6e-8);
    // here it is one evaluation per lane: double-precision argument reduction + v_cos_f32 (2.8e-7).
    float cw, sw_unused;
    unit_phasor_hw(((double)cur.w0 * 0.15915494309189533577) * (double)lane, cw, sw_unused);
    const bool in = lane >= 1 && lane <= L;
    const float Ml2 = cur.Ml * cur.Ml;
    const float Rm0 = seq_sum4(in ? Ml2 : 0.0f, L, tmp, lane);   // returned to the caller: feeds the local-energy filter (state)
    const float Rm1 = wave_sum(in ? (Ml2 * cw) : 0.0f);
    const float R2m0 = Rm0 * Rm0;
    const float R2m1 = Rm1 * Rm1;
    if (in && cur.Ml != 0.0f) {
        // The weight is continuous across its own thresholds (the clamp values equal the weight there),
        // so v_sqrt_f32 / v_rcp_f32 (1 ulp) are accurate enough: no decision can flip visibly.
        const float Wl = __builtin_amdgcn_sqrtf(cur.Ml)
                         * __builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(
                             ((float)0.96 * (float)M_PI * ((R2m0 + R2m1) - ((float)2 * Rm0 * Rm1 * cw)))
                             * __builtin_amdgcn_rcpf(cur.w0 * Rm0 * (R2m0 - R2m1))));
        if ((8 * lane) <= L) {
        } else if (Wl > 1.2f) {
            cur.Ml = 1.2f * cur.Ml;
        } else if (Wl < 0.5f) {
            cur.Ml = 0.5f * cur.Ml;
        } else {
            cur.Ml = Wl * cur.Ml;
        }
    }
    float M = cur.Ml;
    M = (M < 0.0f) ? -M : M;
    const float sum = wave_sum(in ? (M * M) : 0.0f);
    // v_rcp / v_sqrt (1 ulp each): the weights above are already that accurate
    const float gamma = (sum == 0.0f) ? 1.0f : __builtin_amdgcn_sqrtf(Rm0 * __builtin_amdgcn_rcpf(sum));
    if (in) {
        cur.Ml = gamma * cur.Ml;
    }
    return Rm0;
}

// ------------------------------------------------------------------------------------------
// Exact replay of the reference's float arithmetic around ONE threshold decision.
//
// The adaptive smoothing forces a band voiced when its ENHANCED amplitude exceeds a threshold VM derived from the
// pre-enhancement energy Rm0 (src/core/mbe_adaptive.c:217-233).  enhance() / smooth() reach both numbers by wave-parallel
// sums and 1-ulp hardware sqrt / rcp, i.e. within a few ulp of the reference's values but not bit for bit, so an
// amplitude that lands within ~1e-6 of VM could fall on the other side (3 such frames in 49.8 M).
// Frames that have an amplitude within 2e-5 of VM -- about one in a thousand of the frames whose smoothing is active --
// are therefore decided again here with the reference's own sequence of float operations:
//   cos(l w0) by L rotations of (1, 0) by (cosf w0, sinf w0)                 src/core/mbelib.c:412-424
//   Rm0, Rm1 and the sum of squares as sequential sums over l = 1..L          :426-483 (scalar path), :532-590
//   weights with IEEE sqrtf and division, the [0.5, 1.2] clamp, gamma         :485-512, :641-661
//   localEnergy and VM from that Rm0                                          src/core/mbe_adaptive.c:164-195
// and the frame continues with those amplitudes.  (sinf / cosf of w0: double-precision series, rounded to float; the
// reference's libm is within 0.56 ulp of the same value.  expf: the device's, as in the fast path.)
// ------------------------------------------------------------------------------------------
// sinf / cosf for 0 <= x < 1 (every fundamental of the codec tables is below 0.51): series in double, rounded once
__device__ __forceinline__ void small_sincosf(float xf, float& sn, float& cs) {
    const double x = (double)xf, x2 = x * x;
    double ps = -1.0 / 121645100408832000.0, pc = 1.0 / 2432902008176640000.0;   // -1/19!, 1/20!
    const double ks[9] = {1.0 / 355687428096000.0, -1.0 / 1307674368000.0, 1.0 / 6227020800.0, -1.0 / 39916800.0, 1.0 / 362880.0,
                          -1.0 / 5040.0, 1.0 / 120.0, -1.0 / 6.0, 1.0};
    const double kc[10] = {-1.0 / 6402373705728000.0, 1.0 / 20922789888000.0, -1.0 / 87178291200.0, 1.0 / 479001600.0, -1.0 / 3628800.0,
                           1.0 / 40320.0, -1.0 / 720.0, 1.0 / 24.0, -0.5, 1.0};
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        ps = fma(ps, x2, ks[i]);
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        pc = fma(pc, x2, kc[i]);
    }
    sn = (float)(ps * x);
    cs = (float)pc;
}

// pre: lane l holds Ml[l] BEFORE enhancement.  Returns lane l's enhanced Ml[l] exactly as the reference computes it and
// the reference's Rm0 (wave-uniform).
__device__ __forceinline__ float enhance_exact(float pre, float w0, int L, float* tmp, int lane, float& Rm0_out) {
    float s_step, c_step;
    small_sincosf(w0, s_step, c_step);
    float c = 1.0f, sn = 0.0f, cosl = 0.0f;
    for (int l = 1; l <= L; ++l) {
        const float cn = (c * c_step) - (sn * s_step);
        const float sq = (sn * c_step) + (c * s_step);
        c = cn;
        sn = sq;
        cosl = (l == lane) ? c : cosl;
    }
    const float Ml2 = pre * pre;
    const bool in = lane >= 1 && lane <= L;
    const float Rm0 = seq_sum4(in ? Ml2 : 0.0f, L, tmp, lane);
    const float Rm1 = seq_sum4(in ? (Ml2 * cosl) : 0.0f, L, tmp, lane);
    const float R2m0 = Rm0 * Rm0, R2m1 = Rm1 * Rm1;
    float M = pre;
    if (in && M != 0.0f) {
        const float Wl = sqrtf(M)
                         * sqrtf(sqrtf(((float)0.96 * (float)M_PI * ((R2m0 + R2m1) - ((float)2 * Rm0 * Rm1 * cosl)))
                                       / (w0 * Rm0 * (R2m0 - R2m1))));
        if ((8 * lane) <= L) {
        } else if (Wl > 1.2f) {
            M = 1.2f * M;
        } else if (Wl < 0.5f) {
            M = 0.5f * M;
        } else {
            M = Wl * M;
        }
    }
    const float A = (M < 0.0f) ? -M : M;
    const float sum = seq_sum4(in ? (A * A) : 0.0f, L, tmp, lane);
    const float gamma = (sum == 0.0f) ? 1.0f : sqrtf(Rm0 / sum);
    Rm0_out = Rm0;
    return in ? (gamma * M) : M;
}

// ------------------------------------------------------------------------------------------
// Adaptive smoothing.  `pre_ml`: where the amplitudes from before this frame's enhancement can be read back (Ml[0..56] of
// the snapshot), or nullptr when `cur` was not enhanced in this frame -- only touched by the rare exact replay; `tmp`: 64
// floats of LDS.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float smoothing_threshold(float le, float er, int e4) {   // VM for er / et beyond the low thresholds
    const float x8 = sqrtf(sqrtf(sqrtf(le)));
    const float energy = x8 * x8 * x8;
    return (er <= 0.0125f && e4 == 0) ? ((45.255f * energy) / expf(277.26f * er)) : (1.414f * energy);
}

__device__ __forceinline__ float local_energy(float prev_le, float RM0) {
    float pe = prev_le;
    if (pe < 10000.0f) {
        pe = 75000.0f;
    }
    float le = 0.95f * pe + 0.05f * RM0;
    if (le < 10000.0f) {
        le = 10000.0f;
    }
    return le;
}

// The same threshold from hardware sqrt / exp / rcp (1-2 ulp each, relative error of the result < 3e-6): three IEEE square
// roots, an expf and a division are ~80 VALU instructions for ONE wave-uniform number, every frame.  It only has to say
// whether any amplitude is NEAR the threshold (2e-5, an order of magnitude wider than its own error); frames where one is
// are decided with the exact value (and the exact replay), all others compare against a threshold none of their amplitudes
// is close to -- the decision is the reference's either way.
__device__ __forceinline__ float smoothing_threshold_fast(float le, float er, int e4) {
    const float x8 = __builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(le)));
    const float energy = x8 * x8 * x8;
    return (er <= 0.0125f && e4 == 0)
               ? ((45.255f * energy) * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f((277.26f * 1.4426950408889634f) * er)))
               : (1.414f * energy);
}

__device__ __forceinline__ void smooth(Parms& cur, const Parms& prev, float RM0, int lane, const float* pre_ml, float* tmp) {
    const int L = cur.L;
    const float er = cur.errorRate;
    const int et = cur.errorCountTotal, e4 = cur.errorCount4;
    float le = local_energy(prev.localEnergy, RM0);
    const bool in = lane >= 1 && lane <= L;
    bool force = false;
    if (!(er <= 0.005f && et <= 4)) {   // otherwise VM = FLT_MAX and nothing can exceed it
        const float VMf = smoothing_threshold_fast(le, er, e4);
        force = in && cur.Ml > VMf;
        // (a NaN / infinite threshold compares false here: such states take the exact path below)
        const bool near_fast = in && !(fabsf(cur.Ml - VMf) > 4e-5f * VMf);
        if (__ballot(near_fast) != 0ULL) {
            float VM = smoothing_threshold(le, er, e4);
            force = in && cur.Ml > VM;
            const bool near = in && fabsf(cur.Ml - VM) <= 2e-5f * VM;
            if (__ballot(near) != 0ULL && cur.w0 >= 0.0f && cur.w0 < 1.0f) {   // rare: decide again with the reference's own arithmetic
                float rm0e;
                if (pre_ml) {
                    __threadfence_block();   // the snapshot was stored by this wave
                    const float pre = (lane < MBX_BAND_SLOTS) ? pre_ml[lane] : 0.0f;
                    cur.Ml = enhance_exact(pre, cur.w0, L, tmp, lane, rm0e);
                } else {
                    rm0e = RM0;   // the caller's sequential sum already
                }
                le = local_energy(prev.localEnergy, rm0e);
                VM = smoothing_threshold(le, er, e4);
                force = in && cur.Ml > VM;
            }
        }
    } else {
        force = in && cur.Ml > __FLT_MAX__;   // +inf amplitude still compares greater than FLT_MAX
    }
    if (force) {
        cur.Vl = 1;
    }
    cur.localEnergy = uni(le);
    const float Am = wave_sum(in ? cur.Ml : 0.0f);
    int pt = prev.amplitudeThreshold;
    if (pt <= 0) {
        pt = 20480;
    }
    const int Tm = (er <= 0.005f && et <= 6) ? 20480 : (6000 - (300 * et) + pt);
    cur.amplitudeThreshold = Tm;
    if (Am > (float)Tm && Am > 0.0f) {
        const float scale = (float)Tm / Am;
        if (in) {
            cur.Ml *= scale;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Comfort noise: 160 steps of the 48-bit Java LCG, evaluated by affine-map jump-ahead so that
// lane j produces samples j, j+64, j+128.  Output in the lane = sample layout.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long java_lcg_jump(unsigned long long s, int k) {
    const unsigned long long mask = (1ULL << 48) - 1ULL;
    unsigned long long A = 1, C = 0, a = 0x5DEECE66DULL, c = 0xBULL;
    for (int bit = 0; bit < 8; ++bit) {
        if ((k >> bit) & 1) {
            C = (a * C + c) & mask;
            A = (A * a) & mask;
        }
        c = (a * c + c) & mask;
        a = (a * a) & mask;
    }
    return (A * s + C) & mask;
}

__device__ void comfort_noise(float out[3], StreamRng& rng, int lane) {
    if (!rng.cn_seeded) {
        rng.cn_seed48 = (0x12345678ULL ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1ULL);
        rng.cn_seeded = 1;
    }
    const float gain = (0.003f * 32767.0f) / 7.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int n = lane + 64 * j;
        const unsigned long long s = java_lcg_jump(rng.cn_seed48, n + 1);
        const float u = ((float)(uint32_t)(s >> 24) / 16777216.0f) * 2.0f - 1.0f;
        out[j] = u * gain;
    }
    rng.cn_seed48 = java_lcg_jump(rng.cn_seed48, 160);
}

// Packed complex arithmetic with the swaps and signs as OPERAND MODIFIERS (op_sel / neg of the VOP3P encoding).  Written as
// inline assembly because the compiler materialises a shuffled operand such as (-w.y, w.x) with a v_xor and a v_mov before
// every use: a complex product came out as four VALU instructions, a multiplication by -+i as two.  The arithmetic is the
// same as the vector-builtin formulation bit for bit (one packed multiply, one packed FMA).
//   cmul_pk(u, w)  = u w          cmul_cw(u, w) = u conj(w)          cmul_cu(u, w) = conj(u) w
//   add_mi(a, b)   = a - i b      sub_mi(a, b)  = a + i b
__device__ __forceinline__ v2f cmul_pk(v2f u, v2f w) {
    v2f t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(u), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(u), "v"(w), "v"(t));
    return d;
}
__device__ __forceinline__ v2f cmul_cw(v2f u, v2f w) {
    v2f t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(u), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(u), "v"(w), "v"(t));
    return d;
}
__device__ __forceinline__ v2f cmul_cu(v2f u, v2f w) {
    v2f t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(u), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "=v"(d) : "v"(u), "v"(w), "v"(t));
    return d;
}
__device__ __forceinline__ v2f add_mi(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f sub_mi(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// fmodf(x, 2 pi) brought into [0, 2 pi) -- the reference's `wrapped = fmodf(PSIl, 2 pi); if (wrapped < 0) wrapped += 2 pi`
// (src/core/mbelib.c:901-912), bit for bit: fmodf is exact, and so is this for 0 <= x < 4e6.  n is the truncated
// quotient or one more (the product is biased upwards by 1e-6, five times its own rounding error), x - n y is then a
// multiple of ulp(2 pi) of magnitude below 2 pi, i.e. a float: the FMA returns it unrounded, and adding 2 pi to a negative
// one gives the true remainder, again a float.  OCML's fmodf is ~36 VALU instructions with a loop; anything outside the
// range (negative, huge, NaN: caller-made states) still goes through it, for the whole wave.
__device__ __forceinline__ float wrap_two_pi(float x, bool active) {
    const float y = 2.0f * (float)M_PI;
    if (__ballot(active && !(x >= 0.0f && x < 4.0e6f)) != 0ULL) {
        float w = fmodf(x, y);
        if (w < 0.0f) {
            w += y;
        }
        return w;
    }
    const float n = truncf(x * (0.15915494f * 1.000001f));
    float r = fmaf(-n, y, x);
    if (r < 0.0f) {
        r += y;
    }
    return r;
}

// a / b, correctly rounded, for a wave-uniform divisor whose correctly rounded reciprocal is at hand (Markstein: with
// y = RN(1 / b), q = RN(a y) and the exact residual r = a - b q, RN(q + r y) = RN(a / b)): three instructions instead of the
// ~12 of an IEEE division.  |a| is either 0 or far from the overflow / underflow ends where the theorem needs care.
__device__ __forceinline__ float div_by_uniform(float a, float b, float rcp_b) {
    const float q = a * rcp_b;
    const float r = fmaf(-q, b, a);
    return fmaf(r, rcp_b, q);
}

// ------------------------------------------------------------------------------------------
// Speech synthesis core (mbe_synthesizeSpeechCore).  `prev` is the enhanced previous model.
// Output: out[j] = sample lane + 64*j (j = 0..2, sample < 160).
// ------------------------------------------------------------------------------------------
//
// `snap`: the caller has just written `cur` to *snap (the snapshot the reference takes before enhancement), or
// nullptr.  With a snapshot the noise samples taken over from the previous frame are NOT kept in registers
// across the voiced bank: they are read back from the snapshot where the FFT needs them, and the fresh LCG
// samples are recomputed from the seed.  Returns true when the frame was synthesised (cur.uw / cur.ov / cur.PHIl /
// cur.PSIl are new), false when it left early (silence, comfort noise): the per-lane state is then still what
// the snapshot holds.
// kPark: the snapshot lives in LDS (the launch-resident copies of the T >= 4 kernel instances) instead of the stream's HBM slot.
// kEarly: see kEarlyNoise below -- the one-frame instances only (A/B: 65,536 x 1 IMBE -0.9 %, resident -3.5 %; the looped HBM-slot
// instances spill with it, the LDS-resident AMBE+2 one is 1.9 % slower over 128 frames)
// kDriftUnrolled: the voiced bank's rarer loop form unrolled too (the IMBE one-frame instances: see bank_loop)
template <bool kSnap, bool kPark = false, class Scratch = WaveScratch, bool kEarly = false, bool kDriftUnrolled = false>
__device__ bool synth_core(float out[3], Parms& cur, Parms& prev, bool have_rm0, float rm0, StreamRng& rng,
                           Scratch& S, const DeviceTables& tabs, int lane, const mbe_parms* snap_ptr = nullptr) {
    constexpr int kParkN = Scratch::kParkCols;
    // The HBM snapshot is read back through a pointer the compiler cannot trace to the stores that wrote it: otherwise it
    // forwards the stored registers to the loads and carries them across the voiced bank -- the very thing the read-back avoids.
    // (In the LDS-resident instances the snapshot is in LDS and read with ds_read: nothing to hide from the compiler.)
    if (kSnap && !kPark) {
        asm volatile("" : "+s"(snap_ptr));
    }
    const mbe_parms* const snap = kSnap ? snap_ptr : nullptr;
    auto snap_f = [&](int dword) -> float {   // one float of the snapshot
        if constexpr (kPark) {
            return reinterpret_cast<const float*>(snap)[dword];   // LDS
        } else {
            return ((GlobalFloats)snap)[dword];
        }
    };
    GlobalDerived D = (GlobalDerived)tabs.d;
    constexpr int N = 160;
    out[0] = out[1] = out[2] = 0.0f;
    __builtin_amdgcn_s_setprio(0);   // (the front part of an IMBE frame runs at a raised priority, see MBX_PRIO_FRONT_IMBE)
    if (cur.L < 1 || cur.L > 56 || prev.L < 1 || prev.L > 56) {
        return false;   // silence
    }
    if (!have_rm0) {
        const bool in = lane >= 1 && lane <= cur.L;
        rm0 = seq_sum4(in ? (cur.Ml * cur.Ml) : 0.0f, cur.L, reinterpret_cast<float*>(S.fft), lane);
    }
    // the amplitudes before this frame's enhancement are in the snapshot (read only by the rare exact replay, see smooth)
    smooth(cur, prev, rm0, lane, (kSnap && have_rm0) ? reinterpret_cast<const float*>(snap) + O_ML : nullptr,
           reinterpret_cast<float*>(S.fft));

    const bool rate_mutes = fabsf(cur.mutingThreshold - MBE_MUTING_THRESHOLD_AMBE) > 1e-6f;
    if (cur.repeatCount >= MBE_MAX_FRAME_REPEATS || (rate_mutes && cur.errorRate > cur.mutingThreshold)) {
        comfort_noise(out, rng, lane);
        return false;
    }

    // ---- white noise block: noise[i], i = lane + 64*j -------------------------------------
    // noise[0..95] is the overlap kept from the previous frame, noise[96..255] are fresh LCG samples, of
    // which the last 96 become the next overlap (ref src/core/mbe_unvoiced_fft.c).
    float nz[4];
    const bool cold = cur.noiseSeed < 0.0f;   // cold start: a block of zeros, then prime the generator
    const uint32_t x0 = cold ? 0u : (((uint32_t)cur.noiseSeed) % 53125u);
    // k-th value of the LCG started at x0: x_k = (mul_k x_0 + add_k) mod 53125 with (mul_k, add_k) from a host table, ONE packed
    // word per k.  Request (lcg_req) and arithmetic (lcg_val) are separate so that the five samples a frame needs after the voiced
    // bank go out together: written as five look-ups in a row they were five L2 round trips in a row (the compiler keeps each
    // load next to its use), 2-3 us of a wave that lives for 20.
    auto lcg_req = [&](int k) -> uint32_t {
        k = k < 0 ? 0 : k;
        return tab_at<uint32_t>(tabs.d, offsetof(DerivedTables, lcg_pack), 4u * (uint32_t)k);
    };
    auto lcg_val = [&](uint32_t w) -> float {
        return (float)((__umul24(w & 0xffffu, x0) + (w >> 16)) % 53125u);   // both factors are below 53,125: a 24-bit multiply is exact
    };
    auto at = [&](int k) -> float { return lcg_val(lcg_req(k)); };
    // the seed of the next frame is 160 steps on: constants of the generator, no table
    constexpr uint32_t kMul160 = lcg_mul_steps(160), kAdd160 = lcg_add_steps(160);
    if (cold) {
        nz[0] = nz[1] = nz[2] = nz[3] = 0.0f;
        cur.ov[0] = cur.ov[1] = 0.0f;
        if (rng.unv_override) {
            cur.noiseSeed = uni((float)rng.unv_state);
            rng.unv_override = 0;
        } else {
            cur.noiseSeed = 3147.0f;
        }
    } else {
        nz[0] = cur.ov[0];
        if (!snap) {
            nz[1] = (lane < 32) ? cur.ov[1] : at(lane - 32);
            nz[2] = at(lane + 32);
            nz[3] = at(lane + 96);
            cur.ov[0] = at(lane + 64);
            cur.ov[1] = (lane < 32) ? at(lane + 128) : 0.0f;
        }
        cur.noiseSeed = (float)((kMul160 * x0 + kAdd160) % 53125u);   // x0 is wave-uniform: scalar arithmetic
    }
    if (snap) {   // only nz[0] is needed before the FFT (phase randomisation); the rest is rebuilt there
        nz[1] = nz[2] = nz[3] = 0.0f;
        cur.ov[0] = cur.ov[1] = 0.0f;
    }

    MBX_TSX(14);
    // ---- reconcile the two model lengths ---------------------------------------------------
    int maxl;
    if (cur.L > prev.L) {
        maxl = cur.L;
        if (lane > prev.L && lane <= maxl) {
            prev.Ml = 0.0f;
            prev.Vl = 1;
        }
    } else {
        maxl = prev.L;
        if (lane > cur.L && lane <= maxl) {
            cur.Ml = 0.0f;
            cur.Vl = 1;
        }
    }

    // ---- phases ----------------------------------------------------------------------------
    const int numUv = popc64(__ballot(lane <= cur.L && cur.Vl == 0));
    const float cw0 = cur.w0, pw0 = prev.w0;
    const float inv_L = ((ConstDerived)tabs.d)->l_ratio[1][cur.L];   // (float)1 / (float)L from the host (scalar load)
    if (lane >= 1 && lane <= 56 && !MBX_ABL(tabs, 64)) {
        float wrapped = wrap_two_pi(prev.PSIl, lane >= 1 && lane <= 56);
        prev.PSIl = wrapped;
        cur.PSIl = wrapped + ((pw0 + cw0) * ((float)(lane * N) / 2.0f));
        if (lane <= (cur.L / 4)) {
            cur.PHIl = cur.PSIl;
        } else {
            const float pl = ((2.0f * (float)M_PI / 53125.0f) * nz[0]) - (float)M_PI;
            cur.PHIl = cur.PSIl + div_by_uniform((float)numUv * pl, (float)cur.L, inv_L);
        }
    }

    // The two jump-ahead constants of the NEXT frame's noise overlap depend on the lane only: requested here, ahead of the voiced
    // bank, they are there when the bank is done -- an all-voiced frame then has no memory round trip between the bank and its
    // overlap-add (tools/stage_times.py: that round trip was 1.05 us of a one-frame wave's 20).  Two more registers across the bank.
    uint32_t q64_early = 0u, q128_early = 0u;
    constexpr bool kEarlyNoise = MBX_EARLY_NOISE != 0 && kSnap && kEarly;
    if (kEarlyNoise && snap && !cold) {
        q64_early = lcg_req(lane + 64);
        q128_early = lcg_req(lane < 32 ? lane + 128 : 160);
        asm volatile("" ::: "memory");   // (requested HERE: not sunk to the use)
    }

    MBX_TSX(15);
    // ---- voiced bank -----------------------------------------------------------------------
    const bool band = lane >= 1 && lane <= maxl;
    const bool cv = band && (cur.Vl == 1);
    const bool pv = band && (prev.Vl == 1);
    const bool stable = fabsf(cw0 - pw0) < (0.1f * cw0);
    const bool interp = (lane < 8) && cv && pv && stable;
    float acc[3] = {0.0f, 0.0f, 0.0f};

    // (1) low harmonics with a stable pitch (src/core/mbelib.c interpolated branch): amplitude linear and
    //     phase quadratic in n, evaluated directly.  lane = sample; samples (lane, lane + 64) are packed in
    //     one v2f, the short third block (128 + lane, lane < 32) packs two harmonics instead.  The phase
    //     is formed with the reference's own operation order (bit-identical float theta); its cosine uses
    //     a two-float reduction to revolutions and v_cos_f32.
    unsigned long long imask = MBX_ABL(tabs, 16) ? 0ULL : __ballot(interp);
    if (imask) {
        if (lane < 8) {   // per-harmonic constants from the harmonic's own lane, broadcast through LDS
            float4 k = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (interp) {
                const float dphi = cur.PHIl - prev.PHIl - (((pw0 + cw0) * (float)(lane * N)) / 2.0f);
                const float dw = (1.0f / (float)N)
                                 * (dphi - (2.0f * (float)M_PI * floorf((dphi + (float)M_PI) / (2.0f * (float)M_PI))));
                k = make_float4(prev.PHIl, (pw0 * (float)lane) + dw, 2.0f * prev.Ml, 2.0f * (cur.Ml - prev.Ml));
            }
            S.icoef[lane] = k;
        }
        wave_lds_sync();
        const float dw0 = cw0 - pw0;
        // 2 amp cos(theta) for two (harmonic, sample) combinations at once.  kFlat: the two frames have the same
        // fundamental (dw0 == 0, e.g. a held pitch), so the quadratic phase term is an exact zero and is not formed.
        auto term = [&](auto flat, v2f phi, v2f a, v2f fl, v2f amp0, v2f damp, v2f nf, v2f nfrac) -> v2f {
            // (float)(l*n*n) is exact below 2^24, so l * n^2 in float is the same value; the division by
            // 2N = 320 becomes a multiplication by its rounded reciprocal (<= 1 ulp of a phase term < 16 rad)
            v2f theta = phi + (a * nf);
            if constexpr (!decltype(flat)::value) {
                // n^2 is formed here, per term (exact: n^2 < 2^24), not held in three registers across the loop: one more instruction per
                // term of the varying-pitch form buys the seven-wave AMBE one-frame instances their last registers
                v2f nq = nf;
                asm volatile("" : "+v"(nq));   // (or the optimiser hoists the square out of the loop again)
                theta = theta + ((splat(dw0) * (fl * (nq * nq))) * splat(1.0f / 320.0f));
            }
            const v2f hi = theta * splat(0.15915494f);   // revolutions: theta / 2 pi as hi + lo
            const v2f lo = __builtin_elementwise_fma(theta, splat(6.4206382e-9f),
                                                     __builtin_elementwise_fma(theta, splat(0.15915494f), -hi));
            const v2f r = v2f{__builtin_amdgcn_fractf(hi.x), __builtin_amdgcn_fractf(hi.y)} + lo;
            const v2f cs = {__builtin_amdgcn_cosf(r.x), __builtin_amdgcn_cosf(r.y)};
            return __builtin_elementwise_fma(nfrac, damp, amp0) * cs;
        };
        const float nf0 = (float)lane, nf2 = nf0 + 128.0f;
        const v2f nf01 = {nf0, nf0 + 64.0f};
        const v2f nfrac01 = {D->nfrac[lane], D->nfrac[lane + 64]};   // (float)n / (float)N, the reference's quotient (host table)
        const float nfrac2 = D->nfrac[lane + 128];
        v2f acc01 = {0.0f, 0.0f}, acc2 = {0.0f, 0.0f};
        auto run = [&](auto flat) {
            while (imask) {
                const int la = __ffsll((long long)imask) - 1;
                imask &= imask - 1;
                const int lb = imask ? (__ffsll((long long)imask) - 1) : 0;   // slot 0 holds zero amplitudes
                imask &= imask - 1;
                const float4 ka = S.icoef[la], kb = S.icoef[lb];
                acc01 += term(flat, splat(ka.x), splat(ka.y), splat((float)la), splat(ka.z), splat(ka.w), nf01, nfrac01);
                if (lb) {
                    acc01 += term(flat, splat(kb.x), splat(kb.y), splat((float)lb), splat(kb.z), splat(kb.w), nf01, nfrac01);
                }
                acc2 += term(flat, v2f{ka.x, kb.x}, v2f{ka.y, kb.y}, v2f{(float)la, (float)lb}, v2f{ka.z, kb.z}, v2f{ka.w, kb.w},
                             splat(nf2), splat(nfrac2));
            }
        };
        if (dw0 == 0.0f) {
            run(std::true_type{});
        } else {
            run(std::false_type{});
        }
        acc[0] += acc01.x;
        acc[1] += acc01.y;
        if (lane < 32) {
            acc[2] += acc2.x + acc2.y;
        }
    }

    MBX_TSX(10);
    // (2) windowed oscillators.  The reference advances one oscillator per harmonic sample by sample
    //     (src/core/mbelib.c:213-218).  Here the recurrence runs along the OTHER axis: loop over the harmonics
    //     l = 1..maxl, and a sample's phasor e^{i l w0 m} is advanced from harmonic to harmonic by its own step
    //     e^{i w0 m}.  The window leaves the prev model on n = 0..104 and the cur model on n = 56..159, so each
    //     is taken about the CENTRE of its support (52 / 108): lane = distance k = 0..52, and one phasor
    //     e^{i l w0 k} serves the two samples centre +- k,
    //         g cos(psi + l w0 (+-k)) = g cos psi cos(l w0 k) -+ g sin psi sin(l w0 k),
    //     an even sum and an odd sum.  The two models are the two halves of one packed register, so a harmonic
    //     costs 8 packed instructions for all 209 samples.  psi_l = (phase at n = 0) + theta_l * centre, lane =
    //     harmonic, broadcast from LDS (two ds_read_b128 per harmonic); no cross-lane sum.  The synthesis window
    //     depends on the sample only and is applied once, after the sum.
    //     The reference's oscillator turns by theta_l = fl(w0 * l) per sample, not by l * w0; the difference
    //     d_l (exact from one FMA, |d_l| < 1.2e-7) is exact in psi and carried to first order over the +-52
    //     samples around it: cos(x + d m) = cos x - d m sin x, a second pair of sums weighted by k.
    {
        const bool wv_p = pv && !interp && !MBX_ABL(tabs, 4), wv_c = cv && !interp && !MBX_ABL(tabs, 4);
        const bool any = (__ballot(wv_p || wv_c) != 0ULL) && !MBX_ABL(tabs, 8);
        if (any) {
            constexpr double kInv2Pi = 0.15915494309189533577;
            constexpr int kMidPrev = 52, kMidCur = 108;
            float drift_w;   // >= |d_l| g_l of this lane's harmonic, both models
            {   // lane = harmonic: g (cos psi, sin psi, d sin psi, d cos psi) for the prev and the cur model
                const float fl = (float)lane;
                const float pw0l = pw0 * fl, cw0l = cw0 * fl;
                float4 ca = make_float4(0.0f, 0.0f, 0.0f, 0.0f), cd = ca;
                float c, d;
                unit_phasor_hw(fma((double)pw0l, (double)kMidPrev * kInv2Pi, (double)prev.PHIl * kInv2Pi), c, d);
                if (wv_p) {
                    const float g = 2.0f * prev.Ml, dl = -fmaf(pw0, fl, -pw0l);
                    const float A = g * c, B = g * d;
                    ca.x = A;
                    ca.z = B;
                    cd.x = dl * B;
                    cd.z = dl * A;
                }
                unit_phasor_hw(fma((double)cw0l, (double)kMidCur * kInv2Pi, (double)(cur.PHIl - (cw0l * (float)N)) * kInv2Pi), c, d);
                if (wv_c) {
                    const float g = 2.0f * cur.Ml, dl = -fmaf(cw0, fl, -cw0l);
                    const float A = g * c, B = g * d;
                    ca.y = A;
                    ca.w = B;
                    cd.y = dl * B;
                    cd.w = dl * A;
                }
                S.coef_amp[lane] = ca;
                S.coef_drift[lane] = cd;
                // |d_l| g_l <= |d_l g sin psi| + |d_l g cos psi| (<= sqrt 2 of it): the bound from the coefficients themselves, no extra live value
                drift_w = (fabsf(cd.x) + fabsf(cd.z)) + (fabsf(cd.y) + fabsf(cd.w));
            }
            // Whether the harmonic loop forms its drift sums is decided per frame, wave-uniformly (round 6): they correct a
            // sample by k sum_l d_l g_l sin(..), at most 52 sum_l |d_l| g_l float units = 364 sum_l |d_l| g_l int16 LSB.  Where that BOUND
            // is below 1/8 LSB they are not formed -- two of the loop's eight packed instructions and one of its two LDS reads: 82 % of the
            // frames of random-bit IMBE streams, 98 % of AMBE+2's, 61 % of the headline's (counted with the oracle on the bench workloads);
            // frames driven far into the clip keep them, and so does a non-finite amplitude (the compare is false on NaN).  Measured,
            // interleaved A/B (profiles/r06/ab_bank_variants.log): 65,536 x 16 IMBE -6.1 %, 8,192 x 128 AMBE+2 -5.5 %, 65,536 x 1 -2.1 / -1.6 %.
            // (Also tried there and NOT kept: running the loop on one model's half alone, in plain VOP2 instructions, when the other model
            // has no windowed-voiced harmonic -- 37 % of AMBE+2's random-bit frames: +0.9 ... +1.9 %.)
            const bool need_drift = MBX_BANK_ALWAYS_DRIFT || !(wave_sum(drift_w) * (float)(kMidPrev * 7) < 0.125f);
            v2f Ec, Es;   // per-distance steps e^{i w0 k} (.x prev model, .y cur model); their error is amplified by l
            {
                float c, d;
                unit_phasor(((double)pw0 * kInv2Pi) * (double)lane, c, d);
                Ec.x = c;
                Es.x = d;
                unit_phasor(((double)cw0 * kInv2Pi) * (double)lane, c, d);
                Ec.y = c;
                Es.y = d;
            }
            const int kk = lane <= kMidPrev ? lane : kMidPrev;   // lanes 53..63 idle along
            auto Ws = [&](int i) -> float { return tab_at<float>(tabs.t, offsetof(mbx_tables, ws), 4u * (uint32_t)i); };
            const v2f w_plus = {Ws(N + kMidPrev + kk), Ws(kMidCur + (kk < 52 ? kk : 51))};
            const v2f w_minus = {Ws(N + kMidPrev - kk), Ws(kMidCur - kk)};
            wave_lds_sync();
            MBX_TS(5);   // smoothing, phases, bank coefficients
            __builtin_amdgcn_s_setprio(MBX_PRIO_BANK);
            v2f even = {0.0f, 0.0f}, odd = {0.0f, 0.0f}, even_d = {0.0f, 0.0f}, odd_d = {0.0f, 0.0f};
            // The loop runs over the harmonics that HAVE a coefficient in either model, not over 1..maxl (round 6): it ends at the highest
            // such harmonic -- random-bit AMBE+2 frames carry their voiced bands low: 20.4 of 32.6 harmonics on average, IMBE 34.1 of 38.6
            // (counted with the oracle on the bench workloads) -- and where harmonics 1..7 have none (they went to the interpolated
            // branch above: every frame of a held pitch, the headline workload) it starts at harmonic 8 with e^{i 8 w0 k} from three
            // squarings of the step (the same first-order error growth as the seven rotations they replace).
            const unsigned long long live = __ballot(wv_p || wv_c);   // != 0 here
            const int last = 63 - __clzll((long long)live);
            const bool from8 = MBX_BANK_TRIM && (live & 0xfeULL) == 0ULL;
            const int first = from8 ? 8 : 1;
            v2f Qc0 = Ec, Qs0 = Es;   // harmonic 1
            if (from8) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const v2f cc = __builtin_elementwise_fma(Qc0, Qc0, -(Qs0 * Qs0));
                    const v2f cs = Qc0 * Qs0;
                    Qs0 = cs + cs;
                    Qc0 = cc;
                }
            }
            // (the loops count from zero over a base pointer: written as `for (l = first; l <= last; ++l)` the compiler gave up the
            // unrolled body that asks for both harmonics' coefficients before it waits -- one exposed LDS round trip per harmonic,
            // +2 ... +3 % on the long launches with FEWER trips)
            const float4* const amp = &S.coef_amp[first];
            const float4* const drift = &S.coef_drift[first];
            const int trips = last - first + 1;
            auto bank_loop = [&](auto drift_c) {
                constexpr bool kDrift = decltype(drift_c)::value;
                v2f Qc = Qc0, Qs = Qs0;
                auto harmonic = [&](int i) {
                    const float4 a = amp[i];   // wave-uniform address: LDS broadcasts
                    even = __builtin_elementwise_fma(Qc, v2f{a.x, a.y}, even);
                    odd = __builtin_elementwise_fma(Qs, v2f{a.z, a.w}, odd);
                    if constexpr (kDrift) {
                        const float4 b = drift[i];
                        even_d = __builtin_elementwise_fma(Qc, v2f{b.x, b.y}, even_d);
                        odd_d = __builtin_elementwise_fma(Qs, v2f{b.z, b.w}, odd_d);
                    }
                    const v2f nq = __builtin_elementwise_fma(Qc, Ec, -(Qs * Es));
                    Qs = __builtin_elementwise_fma(Qs, Ec, Qc * Es);
                    Qc = nq;
                };
                if constexpr (kDrift && !kDriftUnrolled) {   // the rarer form (2 ... 18 % of the frames of the long launches and of AMBE): not
#pragma unroll 1                                             // unrolled, so that it never costs the common form a register (unrolled, two long
                    for (int i = 0; i < trips; ++i) {        // instances spilled and three AMBE one-frame instances lost a wave per SIMD); the
                        harmonic(i);                         // IMBE one-frame instances have the registers, and 39 % of the headline's frames
                    }                                        // take this form
                } else {
#pragma unroll 2
                    for (int i = 0; i < trips; ++i) {
                        harmonic(i);
                    }
                }
            };
            if (need_drift) {
                bank_loop(std::true_type{});
            } else {
                bank_loop(std::false_type{});
            }
            // sample centre + k: even - odd - k (even_d + odd_d); centre - k: even + odd + k (even_d - odd_d)
            __builtin_amdgcn_s_setprio(0);
            const v2f kf = splat((float)lane);
            const v2f v_plus = ((even - odd) - (kf * (even_d + odd_d))) * w_plus;
            const v2f v_minus = ((even + odd) + (kf * (even_d - odd_d))) * w_minus;
            wave_lds_sync();   // every lane is done with the coefficients: their bytes now carry the samples
            float* const bank_prev = reinterpret_cast<float*>(S.coef_amp);   // samples 0..104
            float* const bank_cur = bank_prev + 112;                          // samples 56..159 at index n - 56
            if (lane <= kMidPrev) {
                bank_prev[kMidPrev + lane] = v_plus.x;   // k = 0: both stores carry the same value (sin 0 = 0)
                bank_prev[kMidPrev - lane] = v_minus.x;
                bank_cur[kMidCur - 56 - lane] = v_minus.y;
                if (lane < 52) {
                    bank_cur[kMidCur - 56 + lane] = v_plus.y;
                }
            }
            wave_lds_sync();
            acc[0] += bank_prev[lane] + ((lane >= 56) ? bank_cur[lane - 56] : 0.0f);
            acc[1] += ((lane <= 40) ? bank_prev[lane + 64] : 0.0f) + bank_cur[lane + 8];
            acc[2] += (lane < 32) ? bank_cur[lane + 72] : 0.0f;
            wave_lds_sync();
        }
    }

    // ---- unvoiced: window, 256-point FFT, per-band scaling, inverse FFT, overlap-add ----------
    // Radix-4, 4 points per lane.  Element i = lane + 64*r lives in lane `lane`, so the first forward
    // stage (span 64) and the last inverse stage run straight from / into registers; the three
    // middle stages of each transform exchange through LDS.  Forward is decimation in frequency
    // (natural in, base-4 digit-reversed out), inverse is decimation in time (digit-reversed in,
    // natural out): no reordering pass.  Twiddles e^{-2 pi i m/256} come from v_sin/v_cos (their
    // argument is in revolutions, m/256 is exact), not from memory.
    // With no unvoiced band every bin is scaled by zero, so the transform pair returns exact zeros:
    // only the overlap with the previous frame's Uw remains.
    const bool any_unvoiced = __ballot(lane >= 1 && lane <= cur.L && cur.Vl == 0) != 0ULL;
    if (!any_unvoiced) {
        cur.uw[0] = cur.uw[1] = cur.uw[2] = cur.uw[3] = 0.0f;
    }
    if (snap && !cold) {
        if (kPark) {
            wave_lds_sync();
        } else {
            __threadfence_block();   // the snapshot was stored by this wave; its stores have long been issued
        }
        MBX_TS(6);   // voiced bank + its output through LDS
        const bool fft_noise = !MBX_ABL(tabs, 32) && any_unvoiced;   // the fresh samples of the transform's input (a voiced frame never needs them)
        float old0 = 0.0f, old1 = 0.0f;
        if (!kEarlyNoise || fft_noise) {   // (the old overlap is the transform's input: an all-voiced frame does not read it back)
            old0 = snap_f(O_OVERLAP + lane);
            old1 = (lane < 32) ? snap_f(O_OVERLAP + 64 + lane) : 0.0f;
        }
        // every request of this stage first (see lcg_req), then ONE wait
        uint32_t q64 = kEarlyNoise ? q64_early : lcg_req(lane + 64), q128 = kEarlyNoise ? q128_early : lcg_req(lane < 32 ? lane + 128 : 160);
        uint32_t qm32 = 0u, q32 = 0u, q96 = 0u;
        if (fft_noise) {
            qm32 = lcg_req(lane - 32);
            q32 = lcg_req(lane + 32);
            q96 = lcg_req(lane + 96);
        }
        asm volatile("" : "+v"(old0), "+v"(old1), "+v"(q64), "+v"(q128), "+v"(qm32), "+v"(q32), "+v"(q96));
        nz[0] = old0;
        nz[1] = old1;
        cur.ov[0] = lcg_val(q64);
        cur.ov[1] = (lane < 32) ? lcg_val(q128) : 0.0f;
        if (fft_noise) {
            nz[1] = (lane < 32) ? nz[1] : lcg_val(qm32);
            nz[2] = lcg_val(q32);
            nz[3] = lcg_val(q96);
        }
    }
    if (!MBX_ABL(tabs, 32) && any_unvoiced) {
        // Register diet: the transform pair is the kernel's register peak, and everything that merely crosses it
        // would cost a wave of occupancy for ALL frames.  Those values wait in the lane's own LDS column instead
        // (same lane writes and reads: no synchronisation).
        if constexpr (kParkN > 2) {
            S.park[2][lane] = acc[0];
        }
        if constexpr (kParkN > 3) {
            S.park[3][lane] = acc[1];
        }
        if constexpr (kParkN > 4) {
            if (lane < 32) {
                S.park[4][lane] = acc[2];
                S.park[4][lane + 32] = cur.ov[1];
            }
        }
        if constexpr (kParkN > 5) {
            S.park[5][lane] = cur.PHIl;
        }
        if constexpr (kParkN > 6) {
            S.park[6][lane] = cur.PSIl;
        }
        if constexpr (kParkN > 7) {
            S.park[7][lane] = cur.ov[0];
        }
        if constexpr (kParkN > 0) {
            S.park[0][lane] = cur.Ml;
        }
        if constexpr (kParkN > 1) {
            S.park[1][lane] = __int_as_float(cur.Vl);
        }
        asm volatile("" ::: "memory");
        // Complex values are (re, im) register pairs: a complex add is one v_pk_add_f32, a multiplication by -+i a
        // swap + sign the packed instructions take as operand modifiers, a complex product two packed instructions.
        // Element e of the transform lives at F[fsw(e)].  Unswizzled, the middle stages collide in the LDS banks (a b64 read is
        // served 32 lanes at a time from 64 dword banks, a b64 write 16 lanes at a time from 32: strides of 4 and 16 elements
        // put 4 and 2 lanes on one bank -- measured on configs[3]: 302 of 1,308 LDS-array cycles per frame were conflicts).
        // fsw is linear over GF(2) (e ^ two shifted bit fields of e), a bijection of 0..255, and conflict-free for EVERY
        // access pattern below (first / last stage and the bin passes: lane + 64 r; stages with spans 16, 4, 1); being
        // linear, fsw(base + r q) = fsw(base) ^ fsw(r q) whenever base has no bit in r q's digit: one swizzle per stage
        // and three XORs with literals.  (tools/fft_swizzle.py searches the family and prints the conflict counts.)
        MBX_TS(7);   // noise samples (table round trip)
        __builtin_amdgcn_s_setprio(MBX_PRIO_FFT);
        v2f* const F = reinterpret_cast<v2f*>(S.fft);
        // (Round 3 kept plain indices in the HBM-slot instances: three more address registers across a butterfly were a spill under
        // their 72 / 80-register caps.  With the products and rotations in two / one instruction the registers are there.)
        constexpr bool kSwz = true;
        constexpr auto fsw = [](int e) constexpr -> int { return kSwz ? (e ^ ((e >> 2) & 3) ^ (((e >> 4) & 7) << 2)) : e; };
        constexpr auto elem = [fsw](int base_sw, int c) constexpr -> int { return kSwz ? (base_sw ^ fsw(c)) : (base_sw + c); };   // element base + c
        const int lane_sw = fsw(lane);   // lane + 64 r  ->  elem(lane_sw, 64 r)
        // w^1, w^2, w^3 of one stage: one hardware evaluation, two complex products (each 2-3 ulp, like the butterflies);
        // the inverse transform asks for the conjugates directly
        auto twiddles = [&](int m, bool conj, v2f& w1, v2f& w2, v2f& w3) {
            const float rev = (float)(m & 255) * (1.0f / 256.0f);
            const float c = __builtin_amdgcn_cosf(rev), sn = __builtin_amdgcn_sinf(rev);
            w1 = conj ? v2f{c, sn} : v2f{c, -sn};
            w2 = cmul_pk(w1, w1);
            w3 = cmul_pk(w1, w2);
        };
        float win[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            win[r] = tab_at<float>(tabs.t, offsetof(mbx_tables, uv_window) + 256u * (size_t)r, 4u * (uint32_t)lane);
        }
        {   // forward stage 1 (span 64): real inputs straight from registers
            const float a0 = nz[0] * win[0], a1 = nz[1] * win[1], a2 = nz[2] * win[2], a3 = nz[3] * win[3];
            const float s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
            v2f w1, w2, w3;
            twiddles(lane, false, w1, w2, w3);
            const v2f dd = {d02, d13};
            F[lane_sw] = v2f{s02 + s13, 0.0f};
            F[elem(lane_sw, 64)] = cmul_cu(dd, w1);    // (d02 - i d13) w
            F[elem(lane_sw, 128)] = splat(s02 - s13) * w2;
            F[elem(lane_sw, 192)] = cmul_pk(dd, w3);   // (d02 + i d13) w^3
        }
        wave_lds_sync();
#pragma unroll
        for (int q = 16; q >= 1; q >>= 2) {
            const int g = lane / q, jj = lane % q;
            const int base = fsw(g * 4 * q + jj);
            const int i1 = elem(base, q), i2 = elem(base, 2 * q), i3 = elem(base, 3 * q);
            const int tstep = 64 / q;   // 256 / (4q)
            const v2f a0 = F[base], a1 = F[i1], a2 = F[i2], a3 = F[i3];
            const v2f s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
            wave_lds_sync();
            F[base] = s02 + s13;
            if (q == 1) {   // last forward stage: all twiddles are 1
                F[i1] = add_mi(d02, d13);   // d02 - i*d13
                F[i2] = s02 - s13;
                F[i3] = sub_mi(d02, d13);   // d02 + i*d13
            } else {
                v2f w1, w2, w3;
                twiddles(jj * tstep, false, w1, w2, w3);
                F[i1] = cmul_pk(add_mi(d02, d13), w1);
                F[i2] = cmul_pk(s02 - s13, w2);
                F[i3] = cmul_pk(sub_mi(d02, d13), w3);
            }
            wave_lds_sync();
        }
        // |X(k)|^2 for k = 0..128 in natural order (position p holds bin rev4(p); rev4(lane + 64 r) = rev4(lane) + r)
        const int kbase = rev4(lane & 63);
        {   // all four reads first: a branch around each read-modify-write made four LDS round trips in series
            float m2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const v2f X = F[elem(lane_sw, 64 * r)];
                const v2f sq = X * X;
                m2[r] = (kbase + r == 0) ? sq.x : (sq.x + sq.y);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (kbase + r <= 128) {
                    S.bins[kbase + r] = m2[r];
                }
            }
        }
        wave_lds_sync();
        // per-band scale: lane = band; a band spans at most 14 bins (w0 <= 4 pi / 39.5)
        bool band_unvoiced = cur.Vl == 0;
        float band_M = cur.Ml;
        if constexpr (kParkN > 1) {
            band_unvoiced = __float_as_int(S.park[1][lane]) == 0;
        }
        if constexpr (kParkN > 0) {
            band_M = S.park[0][lane];
        }
        float band_sc = 0.0f;
        int band_a = 0, band_b = 0;
        const float mult = (256.0f / (2.0f * 3.14159265358979323846f)) * cur.w0;
        // b - a = ceil((l + .5) mult) - ceil((l - .5) mult) <= floor(mult) + 1 in exact arithmetic; a and b are ceilings of two
        // separately ROUNDED float products, so with frac(mult) within ~1e-5 of 1 a band can be one bin wider still: + 2 (wave-uniform;
        // bins[] is padded for it, and the tail loop below only starts at a + 14)
        const int span = uni((mult >= 0.0f && mult < 12.0f) ? ((int)mult + 2) : 14);
        if (lane >= 1 && lane <= cur.L && band_unvoiced) {
            int a = (int)ceilf(((float)lane - 0.5f) * mult);
            int b = (int)ceilf(((float)lane + 0.5f) * mult);
            a = a < 0 ? 0 : a;
            b = b > 128 ? 128 : b;
            float num = 0.0f;
            const int count = b - a;
            const float* const bp = &S.bins[a];   // one address, the bin index is the instruction's offset (bins is padded past 128 + 13)
#pragma unroll
            for (int c = 0; c < 14; ++c) {
                if (c >= span) {   // wave-uniform: no band is wider (a random-bit frame averages four bins per band, not fourteen)
                    break;
                }
                const float m2 = bp[c];
                if (c < count) {
                    num += m2;
                }
            }
            for (int k = a + 14; k < b; ++k) {   // not reached for valid w0; keeps odd states exact
                num += S.bins[k];
            }
            if (count > 0 && num > 1e-10f) {
                // v_rcp / v_rsq (1 ulp each): the bin energies come from a float FFT that differs from the reference's by more
                band_sc = (146.17696f * band_M) * __builtin_amdgcn_rsqf(num * __builtin_amdgcn_rcpf((float)count));
                band_a = a;
                band_b = b;
            }
        }
        wave_lds_sync();   // every band has its energy: the bins turn into scales, zero outside the unvoiced bands
        S.bins[lane] = 0.0f;
        S.bins[lane + 64] = 0.0f;
        if (lane < 4) {
            S.bins[128 + lane] = 0.0f;
        }
        wave_lds_sync();
        for (int k = band_a; k < band_b; ++k) {
            S.bins[k] = band_sc;
        }
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pidx = elem(lane_sw, 64 * r);
            const int k = kbase + r;
            const float sc = S.bins[k > 128 ? 256 - k : k];
            F[pidx] = F[pidx] * splat(sc);
        }
        wave_lds_sync();
        // inverse stages with spans 1, 4, 16 through LDS
#pragma unroll
        for (int q = 1; q <= 16; q <<= 2) {
            const int g = lane / q, jj = lane % q;
            const int base = fsw(g * 4 * q + jj);
            const int i1 = elem(base, q), i2 = elem(base, 2 * q), i3 = elem(base, 3 * q);
            const int tstep = 64 / q;
            const v2f x0 = F[base];
            v2f z1 = F[i1], z2 = F[i2], z3 = F[i3];
            if (q != 1) {   // multiply by the conjugate twiddles
                v2f w1, w2, w3;
                twiddles(jj * tstep, true, w1, w2, w3);
                z1 = cmul_pk(z1, w1);
                z2 = cmul_pk(z2, w2);
                z3 = cmul_pk(z3, w3);
            }
            const v2f s02 = x0 + z2, d02 = x0 - z2, s13 = z1 + z3, d13 = z1 - z3;
            wave_lds_sync();
            F[base] = s02 + s13;
            F[i1] = sub_mi(d02, d13);   // d02 + i*d13
            F[i2] = s02 - s13;
            F[i3] = add_mi(d02, d13);   // d02 - i*d13
            wave_lds_sync();
        }
        {   // last inverse stage (span 64): only the real parts are needed, results stay in registers
            v2f w1, w2, w3;
            twiddles(lane, true, w1, w2, w3);
            const v2f x0 = F[lane_sw];
            const v2f z1 = cmul_pk(F[elem(lane_sw, 64)], w1);
            const v2f z2 = cmul_pk(F[elem(lane_sw, 128)], w2);
            const v2f z3 = cmul_pk(F[elem(lane_sw, 192)], w3);
            const float s02 = x0.x + z2.x, d02 = x0.x - z2.x, s13 = z1.x + z3.x, d13y = z1.y - z3.y;
            cur.uw[0] = (s02 + s13) * (1.0f / 256.0f);
            cur.uw[1] = (d02 - d13y) * (1.0f / 256.0f);   // Re(d02 + i*d13)
            cur.uw[2] = (s02 - s13) * (1.0f / 256.0f);
            cur.uw[3] = (d02 + d13y) * (1.0f / 256.0f);   // Re(d02 - i*d13)
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("" ::: "memory");
        if constexpr (kParkN > 2) {
            acc[0] = S.park[2][lane];
        }
        if constexpr (kParkN > 3) {
            acc[1] = S.park[3][lane];
        }
        if constexpr (kParkN > 4) {
            acc[2] = (lane < 32) ? S.park[4][lane] : 0.0f;
            cur.ov[1] = (lane < 32) ? S.park[4][lane + 32] : 0.0f;
        }
        if constexpr (kParkN > 5) {
            cur.PHIl = S.park[5][lane];
        }
        if constexpr (kParkN > 6) {
            cur.PSIl = S.park[6][lane];
        }
        if constexpr (kParkN > 7) {
            cur.ov[0] = S.park[7][lane];
        }
        if constexpr (kParkN > 0) {
            cur.Ml = S.park[0][lane];
        }
        if constexpr (kParkN > 1) {
            cur.Vl = __float_as_int(S.park[1][lane]);
        }
    }
    MBX_TS(8);   // transform pair (or nothing)
    if (!MBX_ABL(tabs, 32)) {
        // weighted overlap-add: out[n] += (w(n) prevUw[n+128] + w(n-160) Uw[n-32]) / (w(n)^2 + w(n-160)^2);
        // Uw[n-32] sits 32 lanes away: lanes >= 32 take slot j of lane-32, lanes < 32 slot j-1 of lane+32
        // the division by w(n)^2 + w(n-160)^2 is a multiplication by its rounded reciprocal (<= 1 ulp of the unvoiced part)
        const uint32_t lb = 4u * (uint32_t)lane;
        const float wprev[2] = {tab_at<float>(tabs.t, offsetof(mbx_tables, wola_w_prev), lb),
                                tab_at<float>(tabs.t, offsetof(mbx_tables, wola_w_prev) + 256u, lb)};   // w(n) is 0 from n = 106 on
        if (any_unvoiced) {
            const float winv[3] = {tab_at<float>(tabs.d, offsetof(DerivedTables, wola_inv), lb), tab_at<float>(tabs.d, offsetof(DerivedTables, wola_inv) + 256u, lb),
                                   (lane < 32) ? tab_at<float>(tabs.d, offsetof(DerivedTables, wola_inv) + 512u, lb) : 0.0f};
            const float wcurr[3] = {tab_at<float>(tabs.t, offsetof(mbx_tables, wola_w_curr), lb), tab_at<float>(tabs.t, offsetof(mbx_tables, wola_w_curr) + 256u, lb),
                                    (lane < 32) ? tab_at<float>(tabs.t, offsetof(mbx_tables, wola_w_curr) + 512u, lb) : 0.0f};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float give = (lane < 32) ? cur.uw[j] : ((j == 0) ? 0.0f : cur.uw[j - 1]);
                const float cs = __shfl_xor(give, 32, kWave);
                const float ps = (j == 0) ? prev.uw[2] : ((j == 1) ? prev.uw[3] : 0.0f);
                const float wp = (j < 2) ? wprev[j] : 0.0f;
                if ((lane + 64 * j) < N) {
                    acc[j] += ((wp * ps) + (wcurr[j] * cs)) * winv[j];
                }
            }
        } else {   // this frame's Uw is all zeros: only the previous frame's half of the overlap remains
            acc[0] += (wprev[0] * prev.uw[2]) * tab_at<float>(tabs.d, offsetof(DerivedTables, wola_inv), lb);
            acc[1] += (wprev[1] * prev.uw[3]) * tab_at<float>(tabs.d, offsetof(DerivedTables, wola_inv) + 256u, lb);
        }
        wave_lds_sync();
    }

    MBX_TS(9);   // overlap-add
    // ---- soft clip ---------------------------------------------------------------------------
    const float clip = (32767.0f * 0.95f) / 7.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float v = acc[j];
        v = (v > clip) ? clip : ((v < -clip) ? -clip : v);
        out[j] = v;
    }
    return true;
}

// mbe_floattoshort (src/core/mbelib.c:1148-1321): trunc(clamp(7 x, +-31128.65)), NaN -> 0, +-Inf -> +-31128.  The two
// special cases need no test of their own here: 7 * (+-Inf) is clamped like any large value, the comparisons are false
// for NaN, and v_cvt_i32_f32 converts NaN to 0.
__device__ __forceinline__ int16_t to_pcm16(float x) {
    const float top = 32767.0f * 0.95f;
    float v = 7.0f * x;
    v = (v > top) ? top : ((v < -top) ? -top : v);
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return (int16_t)r;
}

__device__ __forceinline__ void store_pcm(const float out[3], size_t frame, int16_t* pcm16, float* pcmf, int lane) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int n = lane + 64 * j;
        if (n < 160) {
            if (pcmf) {
                pcmf[frame * 160 + n] = out[j];
            }
            if (pcm16) {
                pcm16[frame * 160 + n] = to_pcm16(out[j]);
            }
        }
    }
}

__device__ __forceinline__ void load_rng(StreamRng& r, const mbx_stream_rng* p) {
    r.cn_seed48 = p->cn_seed48;
    r.cn_seeded = p->cn_seeded;
    r.unv_state = p->unvoiced_seed_state;
    r.unv_override = p->unvoiced_seed_override;
}

__device__ __forceinline__ void store_rng(const StreamRng& r, mbx_stream_rng* p, int lane) {
    if (lane == 0) {
        p->cn_seed48 = r.cn_seed48;
        p->cn_seeded = r.cn_seeded;
        p->unvoiced_seed_state = r.unv_state;
        p->unvoiced_seed_override = r.unv_override;
    }
}

// ------------------------------------------------------------------------------------------
// Where prev_mp and prev_mp_enhanced live between the frames of a launch.
//   kPark = false (launches with few frames per stream): in their HBM slots -- the loads and stores a one-frame launch
//           needs anyway; seven / six waves per SIMD.
//   kPark = true  (T >= 4, and the single-frame kernels): prev_mp in LDS for the whole launch, copied in once and written back
//           once, so a launch moves each struct over HBM exactly twice whatever T is (at T = 16 the HBM-slot scheme wrote
//           5.2 KB per FRAME: 5.8x the algorithmic bytes); prev_mp_enhanced not in LDS at all (below).  Round 2 kept both
//           structs in LDS: 9,840 B per wave = four waves per SIMD, and the counters showed a kernel starved of waves.
// ------------------------------------------------------------------------------------------
template <bool kPark>
struct ParkedState {};   // kPark = false: nothing in LDS
// prev_mp_enhanced needs no home inside the launch at all.  Every voice frame ends with prev_mp_enhanced := cur_mp
// (ref src/imbe/imbe7200x4400.c:780-888, src/ambe/ambe3600x2450.c:790-800), `cur` never leaves the registers, and synthesis
// reads only ten fields of the enhanced model: they are copied register to register at the end of a frame (enh_keep), the
// struct is loaded from HBM once and written back once from `cur` (AMBE tone-class frames: see `synced` in ambe_stream_body).
// 2,604 B of LDS less per wave: 5,200 B = five allocation granules instead of eight.
struct ParkedPrevOnly {
    mbe_parms prev;
#ifdef MBX_EXP_LDS_PAD
    char pad[MBX_EXP_LDS_PAD];
#endif
};

template <bool kPark>
__device__ __forceinline__ void slot_fence() {   // the wave is about to re-read what it stored to its slots
    if (kPark) {
        wave_lds_sync();          // LDS: program order is enough, and outstanding PCM stores are not waited for
    } else {
        __threadfence_block();
    }
}

template <bool kPark>
__device__ __forceinline__ float slot_read(const float* f, int dword) {   // one float of a parked struct (after an asm barrier on f)
    if constexpr (kPark) {
        return f[dword];                    // LDS: the compiler still knows
    } else {
        return ((GlobalFloats)f)[dword];    // HBM slot: a global load, not a flat one
    }
}

// what load_prev_view reads, and nothing else: resident launches bring only this much of prev_mp into LDS at their start
// (0.7 KB instead of 2.6 KB); a frame that needs the rest (a repeat copies prev_mp whole, an AMBE erasure takes its phases and
// noise state) completes the copy first -- see `prev_partial` in the stream bodies.
__device__ __forceinline__ void copy_prev_view(mbe_parms* dst, const mbe_parms* src, int lane) {
    const float* f = reinterpret_cast<const float*>(src);
    float* g = reinterpret_cast<float*>(dst);
    const bool band = lane < MBX_BAND_SLOTS;
    const float ml = band ? f[O_ML + lane] : 0.0f, l2 = band ? f[O_LOG2ML + lane] : 0.0f, ph = band ? f[O_PHI + lane] : 0.0f;
    const int sc = (lane == 0) ? O_L : ((lane == 1) ? O_GAMMA : ((lane == 2) ? O_ERRORRATE : ((lane == 3) ? O_REPEAT : O_MUTETHR)));
    const float sv = (lane < 5) ? f[sc] : 0.0f;
    if (band) {
        g[O_ML + lane] = ml;
        g[O_LOG2ML + lane] = l2;
        g[O_PHI + lane] = ph;
    }
    if (lane < 5) {
        g[sc] = sv;
    }
}

template <bool kGather = false>
__device__ __forceinline__ void copy_parms(mbe_parms* dst, const mbe_parms* src, int lane) {
    Parms t;
    load_parms<kGather>(t, src, lane);
    store_parms<kGather>(t, dst, lane);
}

// ------------------------------------------------------------------------------------------
// IMBE 7200x4400 stream kernel: grid = S workgroups of one wave.
// ------------------------------------------------------------------------------------------
// Development instrumentation (-DMBX_FRAME_STAMPS, never in the product build): where a synchronous single-frame call spends
// its time on the device -- 100 MHz wall-clock stamps of the single-frame bodies, read back through mbx_debug_frame_stamps().
#ifdef MBX_FRAME_STAMPS
#define MBX_STAMP(i, drain)                                                     \
    do {                                                                        \
        if constexpr (kFrame) {                                                 \
            if (drain) {                                                        \
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
            }                                                                   \
            if (lane_id() == 0) {                                               \
                g_frame_stamps[i] = wall_clock64();                             \
            }                                                                   \
        }                                                                       \
    } while (0)
#else
#define MBX_STAMP(i, drain) do { } while (0)
#endif

__device__ __forceinline__ void frame_done(uint32_t* done, uint32_t token, int lane) {
    __threadfence_system();   // the wave's stores (one wave: s_waitcnt vmcnt(0) covers every lane) are visible to the host ...
    if (done && lane == 0) {
        __hip_atomic_store(done, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... before the flag is
    }
}

__device__ __forceinline__ uint4 broadcast_record(uint4 r, mbx_param_record* out, int lane) {
    if (lane == 0) {
        *reinterpret_cast<uint4*>(out) = r;
    }
    return make_uint4(uni(r.x), uni(r.y), uni(r.z), uni(r.w));
}

// Single-frame kernels: the frame's wire bytes are REQUESTED first of all (frame_fetch: vector memory returns in order, so
// whatever is requested before the state is not held up by it) and its FEC runs wave-uniformly on the scalar unit with scalar
// table loads (frame_record: TabScalar, mbx_fec_frame.h) while the three structs are on their way.
// fec_codec: MBX_CODEC_* of the FRONT END (both AMBE codecs share one)
__device__ __forceinline__ Wire frame_fetch(bool ambe, const uint8_t* frame) {
    return ambe ? load_wire_ambe(frame) : load_wire_imbe(frame);
}
__device__ __forceinline__ Wire frame_from_args(bool ambe, const FrameShadow& x) {   // the same from bytes that came as kernel arguments
    Wire w;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const uint32_t b0 = (x.frame_words[(2 * k) >> 2] >> (8 * ((2 * k) & 3))) & 0xffu;
        const uint32_t b1 = (x.frame_words[(2 * k + 1) >> 2] >> (8 * ((2 * k + 1) & 3))) & 0xffu;
        const uint32_t bk = (x.frame_words[k >> 2] >> (8 * (k & 3))) & 0xffu;
        w.h[k] = ambe ? bk : ((b0 << 8) | b1);
    }
    return w;
}
// the 18 wire bytes of an IMBE frame at a WAVE-UNIFORM address by scalar loads (fused one-frame launches: one wave per stream, so
// the frame's address depends on the workgroup only): they have a counter of their own and do not queue behind the state's vector
// loads.  Frames are 18 bytes apart, i.e. 2-byte aligned: five dwords from the 4-byte boundary at or below the frame, shifted by
// a uniform 0 or 16 bits.  (20 bytes from that boundary end at most 2 bytes past the frame, and never past the end of a batch whose
// base is 4-byte aligned: the last frame of an even batch starts at 2 mod 4, and 18 S of an odd batch is not a page multiple.)
__device__ __forceinline__ FrameWords frame_fetch_scalar_imbe(const uint8_t* frame) {   // the request ...
    const uintptr_t a = reinterpret_cast<uintptr_t>(frame);
    const __attribute__((address_space(4))) uint32_t* q =
        reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(a & ~(uintptr_t)3);
    FrameWords w;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        w.d[i] = q[i];
    }
    return w;
}
__device__ __forceinline__ Wire frame_words_to_wire(const uint8_t* frame, FrameWords w) {   // ... and its first use: a wait for the bytes
    // (the caller issues the wave's vector loads in between: written as one function, the shift below was scheduled -- and the frame
    //  waited for -- in front of every one of them)
    asm volatile("" : "+s"(w.d[0]), "+s"(w.d[1]), "+s"(w.d[2]), "+s"(w.d[3]), "+s"(w.d[4])::"memory");
    uint32_t d[5] = {w.d[0], w.d[1], w.d[2], w.d[3], w.d[4]};
    if (reinterpret_cast<uintptr_t>(frame) & 2u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d[i] = (d[i] >> 16) | (d[i + 1] << 16);
        }
        d[4] >>= 16;
    }
    Wire r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const uint32_t half = (d[k >> 1] >> (16 * (k & 1))) & 0xffffu;   // little-endian half: first wire byte in the low bits
        r.h[k] = ((half & 0xffu) << 8) | (half >> 8);
    }
    return r;
}
__device__ __forceinline__ uint4 frame_record(int fec_codec, Wire wire, mbx_param_record* record, const mbx_tables* T, int lane) {
    const PrLane pr_lanes(lane);   // (before the first use of the frame's bytes: work for the time they are still on their way)
    TabScalar tab(T);
    tab.lanes = &pr_lanes;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        wire.h[k] = uni(wire.h[k]);   // every lane loaded the same bytes
    }
    const uint4 r = (fec_codec == MBX_CODEC_IMBE7200X4400)   ? fec_imbe7200x4400_wire(tab, wire)
                    : (fec_codec == MBX_CODEC_IMBE7100X4400) ? fec_imbe7100x4400_wire(tab, wire)
                                                             : fec_ambe3600x2450_wire(tab, wire);
    return broadcast_record(r, record, lane);
}

// kFrame: the single-frame kernels behind the synchronous per-frame API (S = T = 1): the record comes in registers from
// the FEC the same wave has just run (rec_in), not from memory.
// kOne (HBM-slot instances only): the launch has ONE frame per stream -- no frame loop, and the views of prev_mp / prev_mp_enhanced
// are requested together with cur_mp.  (With a loop the pre-requested views stay allocated through every frame: spills.)
// kFuse (with kOne): the launch IS the whole T = 1 step -- the wave fetches its stream's wire frame by scalar loads, runs the FEC
// wave-uniformly on the scalar unit while the structs are on their way (frame_record, as the single-frame kernels do) and expands
// the record itself: no FEC launch, no expansion launch, no FrameParams row through HBM.  `frame_in` = the batch's frames.
// kFuse = 1: IMBE 7200x4400 frames -- the FEC in two halves (mbx_fec_frame.h): after the head (C0) the fundamental is known, every
// table read of the expansion goes out, then the rest of the state is requested, and the FEC's tail runs while all of that is on
// its way.  kFuse = 2: IMBE 7100x4400 frames (own front end, whole; then the same).
// kFuse = 3: the stream blocks of imbe_one_launch_kernel -- the frame's FrameParams row comes from a FRONT BLOCK of the same launch
// (mbx_front_imbe.h: FEC + expansion of eight frames by one wave), handed over through a flag word (FrontLink); should the row not
// be there in time the wave runs the kFuse = 1 front end itself, so nothing here depends on the order in which blocks are dispatched.
// A launch may hand a workgroup only a SLICE of its stream's frames (the time-sliced launches, *_stream_kernel_lds_sliced): frames
// t0 .. t0 + Tn - 1 of the `stride` the stream has in this call, for the stream at position `pos` of the walk.
struct FrameSlice {
    int stride = 0;   // frames per stream in the batch arrays (0: Tn, the whole stream)
    int t0 = 0;       // first frame of the slice
    int pos = -1;     // the workgroup's position among the streams (its blockIdx.x when < 0)
};
// INVARIANT of the hand-over: it uses RELAXED agent-scope atomics, a hand-written `s_waitcnt vmcnt(0)` between the
// producer's row stores and its flag store, and compiler barriers -- no release / acquire pair.  That is sound ONLY because every word
// that crosses (the 64 dwords of a FrameParams row, the flag, an AMBE record) is written with an sc1 (write-through) atomic store and
// read with an sc1 atomic load, which go past the XCD's L2 on gfx942 / gfx950 (MI355X_MICROARCH.md, inter-workgroup visibility).
// A PLAIN load of handed-over data -- the rest of a row, `records[]` in a stream block -- would read a stale line and no bit-equality
// test would catch it deterministically: anything a stream block reads from its front block must go through the same atomics.
static_assert(sizeof(FrameParams) == 64 * sizeof(uint32_t), "a row is exactly the 64 dwords one wave hands over with one sc1 store per lane");
struct FrontLink {
    const uint32_t* flag = nullptr;   // the ready word of this stream's chunk of eight: == epoch once the chunk's rows are in `params`
    uint32_t        epoch = 0u;
    int             pos = -1;         // the block's position among the stream blocks (its blockIdx.x when < 0)
    void*           lds = nullptr;    // the workgroup's LDS block (shared with the front blocks' arrays: one allocation for both kinds)
    uint32_t*       fallbacks = nullptr;   // counts the stream blocks that expanded their frame themselves (diagnostics: expected 0)
};
#ifndef MBX_FRONT_SPIN
#define MBX_FRONT_SPIN 160   // polls, ~0.25 us apart (s_sleep 8 = 512 cycles), before a stream block gives up on its front block and
#endif                       // expands its frame itself: ~40 us, four times a front block's life
template <bool kPark, bool kFrame = false, bool kRes = false, bool kOne = false, int kFuse = 0>
__device__ __forceinline__ void
imbe_stream_body(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                 mbe_parms* __restrict__ state,
                 mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16, float* __restrict__ pcmf,
                 mbe_process_result* __restrict__ results, DeviceTables tabs_in, const uint8_t* frame_in = nullptr,
                 int fec_codec = 0, FrameShadow shadow = FrameShadow{}, FrontLink link = FrontLink{}, FrameSlice slice = FrameSlice{}) {
    using ScratchT = WaveScratchT<kPark ? 0 : MBX_PARK_N>;
    ScratchT* scratch_ptr;
#ifdef MBX_EXP_PAIR   // EXPERIMENT (never the product build): two streams per 128-thread workgroup, see imbe_stream_kernel_lds_pairexp
    __shared__ ScratchT scratch_pair[2];
    __shared__ std::conditional_t<kPark, ParkedPrevOnly, ParkedState<false>> park_pair[2];
    const int wave_in_group = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) & 1;
    scratch_ptr = &scratch_pair[wave_in_group];
    auto& park = park_pair[wave_in_group];
#else
    if constexpr (kFuse == 3) {
        scratch_ptr = reinterpret_cast<ScratchT*>(link.lds);
    } else {
        __shared__ ScratchT scratch_own;
        scratch_ptr = &scratch_own;
    }
    __shared__ std::conditional_t<kPark, ParkedPrevOnly, ParkedState<false>> park;
#endif
    ScratchT& scratch = *scratch_ptr;
    uint4 rec_in = make_uint4(0u, 0u, 0u, 0u);
    const int bpos = (kFuse == 3) ? link.pos : (slice.pos >= 0 ? slice.pos : (int)blockIdx.x);
    if (bpos >= S) {
        return;
    }
    const int s = tabs_in.reverse ? (S - 1 - bpos) : bpos;
    const size_t fbase = (size_t)s * (size_t)(slice.stride ? slice.stride : Tn) + (size_t)slice.t0;   // the batch index of the slice's first frame
    const int lane_in = lane_id();
    MBX_STAMP(0, false);
    if constexpr (kOne) { MBX_TS(0); }
    Wire wire_in = {};
    if constexpr (kFrame) {
        wire_in = shadow.have_frame ? frame_from_args(false, shadow) : frame_fetch(false, frame_in);   // ahead of every state load
    }

    // Register budget: at most TWO of the three structs are live at any time.  `cur` stays in
    // registers for the whole launch; `prev` is only needed from the start of a frame to the
    // snapshot and `enh` only from the snapshot to the end of synthesis, so both are parked in
    // their own HBM/L2 slots in between (exactly the loads and stores a T = 1 launch needs anyway).
    // batch row s (frames, records, PCM, results) belongs to state / rng slot `slot`: the same number unless the caller
    // passed an index (mbx_process_batch_indexed: the streams that have frames this tick, out of a larger resident pool)
    const size_t slot = tabs_in.stream_map ? (size_t)uni(tabs_in.stream_map[s]) : (size_t)s;
    mbe_parms* const slot_cur = &state[3 * slot + 0];
    mbe_parms* const home_prev = &state[3 * slot + 1];
    mbe_parms* const home_enh = &state[3 * slot + 2];
    mbe_parms *slot_prev, *slot_enh;
    Parms enh_keep;   // kPark: the fields of prev_mp_enhanced that synthesis reads, carried from frame to frame in registers
    Parms cur;
    StreamRng rng;
    // resident launches (DeviceTables::resident): prev_mp_enhanced elided while it equals cur_mp, prev_mp fetched lazily
    uint32_t* const res = (kPark && kRes) ? tabs_in.resident : nullptr;   // (kRes: own kernel instances, the others carry none of this)
    // kRes without kPark: the HBM-slot instance for resident launches of ONE frame per stream.  It reads of prev_mp only the
    // decode's view anyway (load_prev_view) and writes the snapshot straight to its home; what it adds is the elision: the
    // enhanced model's view comes from cur_mp's struct when elided, and prev_mp_enhanced is not written.  (One frame only: with
    // more, the next frame would look for the view in a struct this launch has not written.)
    uint32_t* const res1 = (!kPark && kRes) ? tabs_in.resident : nullptr;
    const bool elided1 = res1 && (uni(res1[slot]) != 0u);
    bool prev_partial = false;   // wave-uniform: the LDS copy of prev_mp holds only the decode's view of it
    // Every request of the launch's first frame goes out before anything waits (see load_header): the three headers first
    // (vector memory returns in order: what is asked for first is there first), then the per-lane arrays.
    Parms enh_first, prev_first;       // !kPark: the first frame's views of prev_mp_enhanced / prev_mp ...
    uint32_t h_enh_first = 0u, h_prev_first = 0u;   // ... and their headers, read out where the frame loop needs them
    float row_first = 0.0f;
    bool have_row = true;   // kFuse == 3: the frame's row came from a front block (else the wave expanded the frame itself)
    ImbeExpandReq xreq;   // kFuse: the expansion's table values, requested at the start
    if constexpr (kPark) {
        slot_prev = &park.prev;
        slot_enh = nullptr;
        const bool elided = res && (uni(res[slot]) != 0u);
        // single-frame kernels: the state comes from the device copy when the host says it is current (FrameShadow)
        const bool from_shadow = kFrame && shadow.use != 0u;
        const mbe_parms* const in_cur = from_shadow ? &shadow.state[0] : slot_cur;
        const mbe_parms* const in_prev = from_shadow ? &shadow.state[1] : home_prev;
        const mbe_parms* const enh_src = from_shadow ? &shadow.state[2] : (elided ? slot_cur : home_enh);
        const uint32_t h_enh = load_header(enh_src, lane_in);
        const uint32_t h_cur = load_header(in_cur, lane_in);
        if (res) {
            load_enh_arrays(enh_keep, enh_src, lane_in);
            copy_prev_view(slot_prev, home_prev, lane_in);
            prev_partial = true;
            load_parms_arrays(cur, slot_cur, lane_in);
            load_rng(rng, &rngs[slot]);
        } else {
            Parms home;
            const uint32_t h_home = load_header(in_prev, lane_in);
            load_enh_arrays(enh_keep, enh_src, lane_in);
            load_parms_arrays(home, in_prev, lane_in);
            load_parms_arrays(cur, in_cur, lane_in);
            load_rng(rng, from_shadow ? shadow.rng : &rngs[slot]);
            if constexpr (kFrame) {   // the FEC of the frame runs while the three structs are on their way (pinned host memory: PCIe)
                rec_in = frame_record(fec_codec, wire_in, const_cast<mbx_param_record*>(records), tabs_in.t, lane_in);
                MBX_STAMP(1, false);
            }
            set_parms_header(home, h_home);
            store_parms<kOne>(home, slot_prev, lane_in);
        }
        set_enh_header(enh_keep, h_enh);
        set_parms_header(cur, h_cur);
        wave_lds_sync();
        MBX_STAMP(2, true);
    } else if constexpr (kOne) {
        slot_prev = home_prev;
        slot_enh = home_enh;
        const mbe_parms* const enh_src = elided1 ? slot_cur : slot_enh;
        uint32_t l_lanes = 0u;
        FrameWords frame_words = {};
        TabScalar fused_tab;       // kFuse == 2 (7100x4400): the scalar-unit FEC
        LaneFecTables lane_tabs;   // kFuse == 1 (7200x4400): the lane-parallel FEC
        uint32_t hgen[2] = {0u, 0u};
        uint32_t flag_v = 0u;
        const uint8_t* const frame_ptr = frame_in + 18u * (size_t)s;
        auto fetch_hgen = [&]() {
            const __attribute__((address_space(4))) uint32_t* hg =
                (const __attribute__((address_space(4))) uint32_t*)((const __attribute__((address_space(4))) char*)tabs_in.t
                                                                    + offsetof(mbx_tables, hamming_gen));
            hgen[0] = hg[0];
            hgen[1] = hg[1];
        };
        if constexpr (kFuse == 3) {
            flag_v = __hip_atomic_load(link.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the wave's first vector load (bypasses L1)
            asm volatile("" ::: "memory");
        } else if constexpr (kFuse != 0) {
            frame_words = frame_fetch_scalar_imbe(frame_ptr);   // scalar loads: first of all, own counter
            if constexpr (kFuse == 1) {
                fetch_hgen();
                lane_tabs = lane_fec_request(tabs_in.d, lane_in);   // the first vector loads of the wave: the FEC's lane-held tables ...
            } else {
                fused_tab = TabScalar(tabs_in.t);   // (the generator rows of the two codes with the frame)
            }
            l_lanes = tab_at<uint32_t>(tabs_in.d, offsetof(DerivedTables, imbe_L_lanes), 4u * (uint32_t)lane_in);   // ... and b0 -> L for every b0
            asm volatile("" ::: "memory");   // (what is asked for first is there first)
        } else if (params) {   // the frame's FrameParams row is what the frame needs first: requested first (tools/stage_times.py: asked for
            row_first = params[fbase].v[lane_in];   // after the scalars of cur_mp had arrived, it cost 2.5 us of a wave's 22)
        }
        constexpr bool kFlat = MBX_FLAT_LOADS && kFuse != 0;
        const uint32_t h_cur = load_header(slot_cur, lane_in);
        if constexpr (!(kRes && MBX_RES_VIEW_FROM_CUR)) {
            h_prev_first = load_header(slot_prev, lane_in);
            h_enh_first = load_header(enh_src, lane_in);
        }
        load_rng(rng, &rngs[slot]);
        load_prev_arrays<kFlat>(prev_first, slot_prev, lane_in);
        load_parms_arrays<kFlat>(cur, slot_cur, lane_in);
        if constexpr (kRes && MBX_RES_VIEW_FROM_CUR) {
            asm volatile("" ::: "memory");   // (everything above is requested before the flag is looked at)
            if (!elided1) {   // (elided: both come from cur_mp's registers where the frame loop starts, see enh_view_of / prev_header_of)
                h_prev_first = load_header(slot_prev, lane_in);
                h_enh_first = load_header(slot_enh, lane_in);
                load_enh_arrays<kFlat>(enh_first, slot_enh, lane_in);
            }
        } else {
            load_enh_arrays<kFlat>(enh_first, enh_src, lane_in);
        }
        mbx_param_record* const rec_out = const_cast<mbx_param_record*>(&records[fbase]);
        auto L_of = [&](int b0) -> int {   // wave-uniform b0
            return (b0 <= 207) ? (int)((__builtin_amdgcn_readlane((int)l_lanes, b0 & 63) >> (8 * (b0 >> 6))) & 0xff) : 0;
        };
        auto front_in_wave_7200 = [&]() {   // FEC by lanes (mbx_fec_frame.h, LaneFec) + in-wave expansion of this stream's frame
            asm volatile("" : "+s"(frame_words.d[0]), "+s"(frame_words.d[1]), "+s"(frame_words.d[2]), "+s"(frame_words.d[3]),
                         "+s"(frame_words.d[4])::"memory");   // (the first use of the frame's bytes: not scheduled in front of the loads above)
            MBX_TS(14);   // (fused) the frame's bytes are there
            const LaneFecHead head = lane_fec_imbe_head(frame_words, (reinterpret_cast<uintptr_t>(frame_ptr) & 2u) != 0u, lane_tabs, tabs_in.t,
                                                        lane_in);
            expand_imbe_request(xreq, head.b0, tabs_in.t, tabs_in.d, lane_in, L_of(head.b0));
            asm volatile("" ::: "memory");
            MBX_TS(15);   // (fused) C0 corrected, L known, the expansion's table reads requested
            rec_in = broadcast_record(lane_fec_imbe_tail(head, lane_tabs, hgen, tabs_in.t, lane_in), rec_out, lane_in);
            MBX_TS(10);   // (fused) frame fetched, FEC done, every request issued
            expand_imbe_finish(rec_in, xreq, scratch, lane_in);   // before anything reads the state out
            MBX_TS(11);   // (fused) table values there, expanded
        };
        if constexpr (kFuse == 3) {
            // The row of this stream's frame is written by a front block of the same launch, dispatched thousands of blocks earlier:
            // normally the flag read at the very start already says so.  Hand-over per MI355X_MICROARCH.md (inter-workgroup
            // visibility): the producer stores the rows with sc1 stores, drains them (s_waitcnt vmcnt(0)) and then stores the flag
            // sc1; the consumer polls the flag with sc1 loads and, once it matches, reads the row with sc1 loads -- no L1 invalidate.
            asm volatile("" ::: "memory");
            bool ready = uni(flag_v) == link.epoch;
            for (int tries = 0; !ready && tries < MBX_FRONT_SPIN; ++tries) {
                __builtin_amdgcn_s_sleep(8);
                ready = uni(__hip_atomic_load(link.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == link.epoch;
            }
            if (ready) {
                const uint32_t* const rowp = reinterpret_cast<const uint32_t*>(&params[fbase].v[0]);
                row_first = __uint_as_float(__hip_atomic_load(rowp + lane_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                have_row = true;
            } else {   // (never observed: the dispatch order is not a contract, so the wave can also do without its front block)
                frame_words = frame_fetch_scalar_imbe(frame_ptr);
                fetch_hgen();
                lane_tabs = lane_fec_request(tabs_in.d, lane_in);
                l_lanes = tab_at<uint32_t>(tabs_in.d, offsetof(DerivedTables, imbe_L_lanes), 4u * (uint32_t)lane_in);
                front_in_wave_7200();
                have_row = false;
                if (link.fallbacks && lane_in == 0) {
                    atomicAdd(link.fallbacks, 1u);
                }
            }
        } else if constexpr (kFuse != 0) {
            // The front end of the frame while the state is on its way.  Its chain of dependent memory round trips is what a fused wave
            // pays for (tools/stage_times.py): the frame's bytes -> C0's Golay correction -> [b0 -> L: from l_lanes, no trip] -> the
            // expansion's table reads, with the FEC's tail (one more table trip) in their shadow.
            asm volatile("" ::: "memory");
            if constexpr (kFuse == 1) {
                front_in_wave_7200();
            } else {
                wire_in = frame_words_to_wire(frame_ptr, frame_words);
                const PrLane pr_lanes(lane_in);   // (lane constants of the demodulation sequence)
                fused_tab.lanes = &pr_lanes;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    wire_in.h[k] = uni(wire_in.h[k]);
                }
                rec_in = broadcast_record(fec_imbe7100x4400_wire(fused_tab, wire_in), rec_out, lane_in);
                const int b0 = imbe_record_b0(rec_in);
                expand_imbe_request(xreq, b0, tabs_in.t, tabs_in.d, lane_in, L_of(b0));
                MBX_TS(10);   // (fused) frame fetched, FEC done, every request issued
                expand_imbe_finish(rec_in, xreq, scratch, lane_in);   // before anything reads the state out
                MBX_TS(11);   // (fused) table values there, expanded
            }
        }
        set_parms_header(cur, h_cur);
        MBX_TS(1);   // cur_mp's scalars are there (first round trip)
    } else {
        slot_prev = home_prev;
        slot_enh = home_enh;
        load_parms<kOne>(cur, slot_cur, lane_in);
        load_rng(rng, &rngs[slot]);
    }

    const int frames = (kOne && Tn > 1) ? 1 : Tn;
    for (int t = 0; t < frames; ++t) {
        const size_t f = fbase + (size_t)t;
        // Keep per-frame table values out of the loop-carried register set: without this the compiler
        // hoists ~100 VGPRs of lane-dependent values (twiddles, windows, jump-ahead constants, indices)
        // across the frame loop, which halves the occupancy.
        DeviceTables ft = tabs_in;
        int lane = lane_in;
        asm volatile("" : "+s"(ft.t), "+s"(ft.d), "+v"(lane));
        if constexpr (kPark) {
            // Tell the compiler again that the lane index is 0..63: struct fields are then addressed as SGPR base + 32-bit
            // lane offset instead of 64-bit per-lane addresses (two VALU and a register pair each).  Only where registers are
            // plentiful: under the 72 / 80-register caps of the HBM-slot instances the extra freedom ends in spills.
            lane &= 63;
        }
        const DeviceTables& tabs = ft;
#if defined(MBX_EXP_PAIR) && MBX_EXP_PAIR > 0   // the rendezvous a paired transform would need: MBX_EXP_PAIR barriers per frame
#pragma unroll
        for (int bq = 0; bq < MBX_EXP_PAIR; ++bq) {
            __builtin_amdgcn_s_barrier();
        }
#endif
        if constexpr (kPark && !kFrame) {
            __builtin_amdgcn_s_setprio(MBX_PRIO_FRONT_IMBE);
        }
        // Frame parameters: expanded here from the FEC record (the normal path), or taken from the
        // workspace row a separate mbx_expand_records() launch has written.  Either way decode reads LDS.
        // The table requests go out BEFORE the state loads (vector memory returns in order, and the
        // tables are L2 hits), the arithmetic runs while the state is in flight.
        // The parts of prev_mp_enhanced that synthesis reads are requested now, together with prev_mp,
        // so that one memory latency covers both (they are first used after the decode).
        Parms enh;
        Parms prev;
        if constexpr (kPark) {
            enh = enh_keep;
            load_prev_view_lds(prev, slot_prev, lane);
        } else if constexpr (kOne) {   // requested at the start, together with cur_mp
            prev = prev_first;
            if (kRes && MBX_RES_VIEW_FROM_CUR && elided1) {   // cur_mp still holds what the last frame left: that IS prev_mp_enhanced
                prev_header_of(prev, cur);
                enh_view_of(enh, cur);
                enh.w0 = cur.w0;
                enh.L = cur.L;
                enh.localEnergy = cur.localEnergy;
                enh.amplitudeThreshold = cur.amplitudeThreshold;
            } else {
                set_prev_header(prev, h_prev_first);
                enh = enh_first;
                set_enh_header(enh, h_enh_first);
            }
        } else {
            load_enh_view(enh, elided1 ? slot_cur : slot_enh, lane);
            load_prev_view(prev, slot_prev, lane);
        }
        if constexpr (kFuse == 3) {
            if (have_row) {
                scratch.x.fp[lane] = row_first;
                wave_lds_sync();
            }
        } else if (params) {
            scratch.x.fp[lane] = kOne ? row_first : params[f].v[lane];
            wave_lds_sync();
        } else if constexpr (kFuse != 0) {
            // (expanded at the start of the wave)
        } else {
            expand_imbe_wave(kFrame ? rec_in : load_record_scalar(&records[f]), scratch, tabs.t, tabs.d, lane);
        }
        MBX_STAMP(3, false);
        if constexpr (kOne) { MBX_TS(2); }   // the frame's parameters are in LDS (row or expansion)
        const float* fp = scratch.x.fp;
        const uint32_t errw = uni(__float_as_uint(fp[62]));
        const int c0 = (int)(errw & 0xffu), prot = (int)((errw >> 8) & 0xffu), c4 = (int)((errw >> 16) & 0xffu);
        unsigned flags = (errw >> 24) & 0xffu;   // C0_VALID | C4_VALID from the FEC stage
        const int total = c0 + prot;
        bool muted;
        {

            // prepare (imbe4400_prepare_process)
            cur.errorCount4 = c4;
            cur.mutingThreshold = MBE_MUTING_THRESHOLD_IMBE;
            cur.errorCountTotal = total;
            cur.errorRate = uni((0.95f * prev.errorRate) + (0.000365f * (float)total));

            const int bad = MBX_ABL(tabs, 1) ? 0 : decode_imbe(fp, cur, prev, tabs.d, lane, scratch.x.C);
            const float repeat_threshold = 10.0f + (40.0f * cur.errorRate);
            const bool c0_valid = (flags & MBE_PROCESS_FLAG_C0_VALID) != 0u;
            const bool repeat =
                (bad == 1) || (c0_valid ? ((c0 >= 2) && ((float)total >= repeat_threshold)) : (total > 5));
            if (!repeat) {
                cur.repeatCount = 0;
            } else {
                if (prev.repeatCount > (MBE_MAX_FRAME_REPEATS - 1)) {
                    imbe_headroom_reset(cur, lane);
                } else {
                    // cur_mp := prev_mp.  Only the prediction memory was loaded (and padded by the
                    // decode); everything else is fetched from the slot now.
                    if (prev_partial) {   // resident launch: the rest of prev_mp has not been brought in yet
                        copy_parms<kOne>(slot_prev, home_prev, lane);
                        slot_fence<kPark>();
                    }
                    load_parms<kOne>(cur, slot_prev, lane);
                    cur.Ml = prev.Ml;
                    cur.log2Ml = prev.log2Ml;
                    cur.repeatCount++;
                }
                flags |= MBE_PROCESS_FLAG_REPEAT;
            }
            muted = (cur.repeatCount >= MBE_MAX_FRAME_REPEATS) || (cur.errorRate > cur.mutingThreshold);
        }
        // prev_mp := cur_mp (snapshot before enhancement).  The scheduling barriers keep the 14 stores in one piece: mixed
        // into the decode before them or the enhancement after them they stretch live ranges past the 72-register budget.
        MBX_STAMP(4, false);
        if constexpr (kOne) { MBX_TS(3); }   // decoded, policy applied
        if (!MBX_ABL(tabs, 256)) store_parms<kOne>(cur, slot_prev, lane);
        prev_partial = false;
        // Register diet for the synthesiser: what the snapshot holds and the synthesiser does not change (log2Ml) or
        // replaces only at its end (previousUw, the noise overlap) is dropped here and read back from the snapshot
        // afterwards -- seven VGPRs less across the voiced bank.
        cur.log2Ml = 0.0f;
        cur.uw[0] = cur.uw[1] = cur.uw[2] = cur.uw[3] = 0.0f;
        float out[3] = {0.0f, 0.0f, 0.0f};
        bool fresh = false;
        {
            const float rm0 = MBX_ABL(tabs, 2) ? 1.0f : enhance(cur, lane, scratch.x.C);
            MBX_STAMP(5, false);
            if constexpr (kOne) { MBX_TS(4); }   // snapshot stored, enhanced
            if (!MBX_ABL(tabs, 128)) {
                fresh = synth_core<true, kPark, ScratchT, kOne, kOne>(out, cur, enh, true, rm0, rng, scratch, tabs, lane, slot_prev);
            }
            MBX_STAMP(6, false);
            if constexpr (kOne) { MBX_TS(12); }   // synthesised (soft clip)
        }
        {
            slot_fence<kPark>();
            const float* f = reinterpret_cast<const float*>(slot_prev);
            if constexpr (!kPark) {
                asm volatile("" : "+s"(f));   // a real load, not the stored registers carried across the synthesiser (see synth_core)
            }
            if (lane < MBX_BAND_SLOTS) {
                cur.log2Ml = slot_read<kPark>(f, O_LOG2ML + lane);
            }
            if (!fresh) {   // silence / comfort noise: previousUw and the noise overlap are unchanged
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cur.uw[j] = slot_read<kPark>(f, O_UW + lane + 64 * j);
                }
                cur.ov[0] = slot_read<kPark>(f, O_OVERLAP + lane);
                cur.ov[1] = (lane < 32) ? slot_read<kPark>(f, O_OVERLAP + 64 + lane) : 0.0f;
            }
        }
        if (muted) {
            flags |= MBE_PROCESS_FLAG_MUTE;
        }
        if constexpr (kPark) {   // prev_mp_enhanced := cur_mp, as far as the next frame's synthesis reads it
            enh_keep = Parms{};
            enh_keep.w0 = cur.w0;
            enh_keep.L = cur.L;
            enh_keep.Vl = cur.Vl;
            enh_keep.Ml = cur.Ml;
            enh_keep.PHIl = cur.PHIl;
            enh_keep.PSIl = cur.PSIl;
            enh_keep.localEnergy = cur.localEnergy;
            enh_keep.amplitudeThreshold = cur.amplitudeThreshold;
            enh_keep.uw[2] = cur.uw[2];
            enh_keep.uw[3] = cur.uw[3];
        } else if (res1) {
            if (lane == 0) {
                res1[slot] = 1u;   // prev_mp_enhanced := cur_mp, elided
            }
        } else {
            if (!MBX_ABL(tabs, 512)) store_parms<kOne>(cur, slot_enh, lane);    // prev_mp_enhanced := cur_mp
        }
        if (t + 1 < Tn) {
            slot_fence<kPark>();             // the next frame of this wave reloads both slots
        }

        store_pcm(out, f, pcm16, pcmf, lane);
        if (results && lane == 0) {
            mbe_process_result r;
            r.c0_errors = c0;
            r.protected_errors = total - c0;
            r.c4_errors = c4;
            r.total_errors = total;
            r.flags = flags;
            results[f] = r;
        }
    }

    if (!MBX_ABL(tabs_in, 1024)) store_parms<kOne>(cur, slot_cur, lane_in);
    store_rng(rng, &rngs[slot], lane_in);
    if constexpr (kPark) {   // prev_mp goes home from LDS; prev_mp_enhanced IS cur_mp after the last frame
        if (res) {
            if (Tn > 0 && lane_in == 0) {
                res[slot] = 1u;   // ... and is not written at all by a resident launch
            }
        } else {
            store_parms<kOne>(cur, home_enh, lane_in);
        }
        wave_lds_sync();
        if (!prev_partial) {
            copy_parms<kOne>(home_prev, slot_prev, lane_in);
        }
        if constexpr (kFrame) {
            if (shadow.state) {   // the device copy of what has just gone to the caller (prev_mp_enhanced is cur_mp after an IMBE frame)
                store_parms<kOne, false>(cur, &shadow.state[0], lane_in);
                store_parms<kOne, false>(cur, &shadow.state[2], lane_in);
                copy_parms(&shadow.state[1], slot_prev, lane_in);
                store_rng(rng, shadow.rng, lane_in);
                if (lane_in == 0) {
                    *shadow.ok = 1u;
                }
            }
        }
    }
    MBX_STAMP(7, false);
    if constexpr (kOne) { MBX_TS(13); }   // every store issued
    MBX_STAMP(8, true);
}

__global__ void __launch_bounds__(64, MBX_STREAM_WAVES_PER_SIMD)
imbe_stream_kernel(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                   mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                   float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    imbe_stream_body<false>(S, Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in);
}

// T = 1 (a tick of the per-tick API, BASELINE configs[1]): the HBM-slot body without a frame loop, every request of the frame
// issued before the first wait.
__global__ void __launch_bounds__(64, MBX_STREAM_WAVES_PER_SIMD)
imbe_stream_kernel_one(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                       mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                       float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    imbe_stream_body<false, false, false, true>(S, Tn > 1 ? 1 : Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in);
}

// The whole T = 1 step as ONE launch (kFuse): wire frames in, PCM / results / records / state out.
// fec_codec: MBX_CODEC_IMBE7200X4400 or MBX_CODEC_IMBE7100X4400 (own front end, the same stream stage).
__global__ void __launch_bounds__(64, MBX_STREAM_WAVES_PER_SIMD)
imbe_stream_kernel_one_fused(int S, int fec_codec, const uint8_t* __restrict__ frames, mbx_param_record* __restrict__ records,
                             mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                             float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    imbe_stream_body<false, false, false, true, 1>(S, 1, records, nullptr, state, rngs, pcm16, pcmf, results, tabs_in, frames, fec_codec);
}
__global__ void __launch_bounds__(64, MBX_STREAM_WAVES_PER_SIMD)
imbe_stream_kernel_res1_fused(int S, int fec_codec, const uint8_t* __restrict__ frames, mbx_param_record* __restrict__ records,
                              mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                              float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    imbe_stream_body<false, false, true, true, 1>(S, 1, records, nullptr, state, rngs, pcm16, pcmf, results, tabs_in, frames, fec_codec);
}
// IMBE 7100x4400 frames: own front end (whole, then the expansion's requests), the same stream stage
__global__ void __launch_bounds__(64, MBX_STREAM_WAVES_PER_SIMD)
imbe7100_stream_kernel_one_fused(int S, int fec_codec, const uint8_t* __restrict__ frames, mbx_param_record* __restrict__ records,
                                 mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                                 float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    imbe_stream_body<false, false, false, true, 2>(S, 1, records, nullptr, state, rngs, pcm16, pcmf, results, tabs_in, frames, fec_codec);
}
__global__ void __launch_bounds__(64, MBX_STREAM_WAVES_PER_SIMD)
imbe7100_stream_kernel_res1_fused(int S, int fec_codec, const uint8_t* __restrict__ frames, mbx_param_record* __restrict__ records,
                                  mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                                  float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    imbe_stream_body<false, false, true, true, 2>(S, 1, records, nullptr, state, rngs, pcm16, pcmf, results, tabs_in, frames, fec_codec);
}

// ------------------------------------------------------------------------------------------
// imbe_one_launch_kernel: the whole T = 1 step of the IMBE 7200x4400 codec as ONE launch whose grid holds two kinds of
// one-wave workgroups: FRONT blocks (FEC + parameter expansion of eight frames each, mbx_front_imbe.h) and STREAM blocks
// (the one-frame stream stage, imbe_stream_body kFuse = 3).  Order in the grid (C = chunks of eight streams, D = lead chunks):
//   front 0 .. D-1 | front D, stream 0..7 | front D+1, stream 8..15 | ... | front C-1, ... | the last 8 D stream blocks
// so the front block of a chunk sits 9 D blocks ahead of its stream blocks: by the time those start, its rows are in the
// workspace, and its work -- table gathers and integer arithmetic -- has run in the shadow of stream blocks that wait for HBM.
// A stream block that does not find its row (its flag word != this launch's epoch) after ~25 us expands its frame itself.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void front_block_imbe(int chunk, int S, const uint8_t* __restrict__ frames, mbx_param_record* __restrict__ records,
                                                 FrameParams* __restrict__ rows, uint32_t* __restrict__ flags, uint32_t epoch,
                                                 const DeviceTables& tabs, void* lds) {
    constexpr int kRowPad = 65;
    float (*tile)[kRowPad] = reinterpret_cast<float (*)[kRowPad]>(lds);                                   // 8 x 65 floats
    uint32_t (*words)[64] = reinterpret_cast<uint32_t (*)[64]>(reinterpret_cast<char*>(lds) + 8 * kRowPad * 4);   // 8 x 64 words
    float (*gains)[8] = reinterpret_cast<float (*)[8]>(reinterpret_cast<char*>(lds) + 8 * kRowPad * 4 + 8 * 64 * 4);   // 8 x 8 floats
    const int lane = lane_id();
    const int fi = lane >> 3, sub = lane & 7;
    const int j = 8 * chunk + fi;
    const bool have = j < S;
    MBX_TS(0);
    __builtin_amdgcn_s_setprio(MBX_PRIO_FRONT_BLOCK);   // a front block holds a wave slot for as long as its chain of table reads takes: first in line
    const int sj = have ? (tabs.reverse ? (S - 1 - j) : j) : 0;
    const uint4 rec = front8_fec_imbe(have, frames + 18u * (size_t)sj, tabs, lane);
    if (have && sub == 0) {
        *reinterpret_cast<uint4*>(&records[sj]) = rec;   // (an output of the call, not read by the stream blocks)
    }
    xp::expand_imbe_frame_rec(have, rec, tile[fi], words[fi], gains[fi], sub, tabs);
    wave_lds_sync();
    MBX_TS(9);   // inverse DCTs done, rows in LDS
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int jq = 8 * chunk + q;
        if (jq < S) {   // one 256-byte row = two whole 128-byte lines by ONE store instruction of the wave, written through (sc1)
            const int sq = tabs.reverse ? (S - 1 - jq) : jq;
            __hip_atomic_store(reinterpret_cast<uint32_t*>(&rows[sq].v[0]) + lane, __float_as_uint(tile[q][lane]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every row store of the wave has left ...
    MBX_TS(10);   // rows written through
    if (lane == 0) {
        __hip_atomic_store(&flags[chunk], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before the flag does
    }
}

template <bool kRes>
__device__ __forceinline__ void imbe_one_launch_body(int S, int lead, const uint8_t* __restrict__ frames, mbx_param_record* __restrict__ records,
                                                     FrameParams* __restrict__ rows, uint32_t* __restrict__ flags, uint32_t* __restrict__ fallbacks,
                                                     uint32_t epoch, mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                                                     float* __restrict__ pcmf, mbe_process_result* __restrict__ results, const DeviceTables& tabs_in) {
    const int C = (S + 7) >> 3;
    const int D = lead < C ? (lead < 0 ? 0 : lead) : C;
    const int G = C - D;
    const int bid = (int)blockIdx.x;   // 9 C blocks
    int chunk = -1, pos = -1;
    if (bid < D) {
        chunk = bid;
    } else {
        const int q = bid - D;
        if (q < 9 * G) {
            const int g = q / 9, r = q - 9 * g;
            if (r == 0) {
                chunk = D + g;
            } else {
                pos = 8 * g + r - 1;
            }
        } else {
            pos = 8 * G + (q - 9 * G);
        }
    }
    constexpr size_t kFrontLds = 8 * 65 * 4 + 8 * 64 * 4 + 8 * 8 * 4, kStreamLds = sizeof(WaveScratchT<MBX_PARK_N>);
    __shared__ alignas(16) char lds[kFrontLds > kStreamLds ? kFrontLds : kStreamLds];   // ONE block of LDS for either kind of workgroup
    if (chunk >= 0) {
#ifdef MBX_TESTING   // libmbx_hip_testing.so only (mbx_testing_set_front_skip): this chunk's stream blocks will not find their rows
        if (tabs_in.front_skip > 0 && (chunk & (tabs_in.front_skip - 1)) == 0) {
            return;
        }
#endif
        front_block_imbe(chunk, S, frames, records, rows, flags, epoch, tabs_in, lds);
        return;
    }
    FrontLink link;
    link.lds = lds;
    link.fallbacks = fallbacks;
    link.flag = &flags[pos >> 3];
    link.epoch = epoch;
    link.pos = pos;
    imbe_stream_body<false, false, kRes, true, 3>(S, 1, records, rows, state, rngs, pcm16, pcmf, results, tabs_in, frames, MBX_CODEC_IMBE7200X4400,
                                                  FrameShadow{}, link);
}

__global__ void __launch_bounds__(64, MBX_STREAM_WAVES_PER_SIMD)
imbe_one_launch_kernel(int S, int lead, const uint8_t* __restrict__ frames, mbx_param_record* __restrict__ records, FrameParams* __restrict__ rows,
                       uint32_t* __restrict__ flags, uint32_t* __restrict__ fallbacks, uint32_t epoch, mbe_parms* __restrict__ state,
                       mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16, float* __restrict__ pcmf,
                       mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    imbe_one_launch_body<false>(S, lead, frames, records, rows, flags, fallbacks, epoch, state, rngs, pcm16, pcmf, results, tabs_in);
}
__global__ void __launch_bounds__(64, MBX_STREAM_WAVES_PER_SIMD)
imbe_one_launch_kernel_res(int S, int lead, const uint8_t* __restrict__ frames, mbx_param_record* __restrict__ records, FrameParams* __restrict__ rows,
                           uint32_t* __restrict__ flags, uint32_t* __restrict__ fallbacks, uint32_t epoch, mbe_parms* __restrict__ state,
                           mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16, float* __restrict__ pcmf,
                           mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    imbe_one_launch_body<true>(S, lead, frames, records, rows, flags, fallbacks, epoch, state, rngs, pcm16, pcmf, results, tabs_in);
}

// T >= 4: prev_mp resident in LDS (see ParkedPrevOnly).  5,200 B of LDS per wave allow 25 waves per CU, so the register
// file decides the occupancy: 79 VGPRs = six waves per SIMD.  Measured on configs[3] (65,536 x T=16) with padded LDS:
// two waves per SIMD 5.15 ms, three 3.91, four (round 2) 3.40, five 2.86, six 2.76.
#define MBX_LDS_KERNEL_ATTR(waves) __launch_bounds__(64, waves)
// (MBX_IMBE_LDS_WAVES_PER_SIMD / MBX_AMBE_LDS_WAVES_PER_SIMD: mbx_device.h, shared with the launcher's slicing heuristic)
__global__ void MBX_LDS_KERNEL_ATTR(MBX_IMBE_LDS_WAVES_PER_SIMD)
imbe_stream_kernel_lds(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                       mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                       float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    imbe_stream_body<true>(S, Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in);
}

// ------------------------------------------------------------------------------------------
// AMBE+2 3600x2450: parameter decode (a9), frame policy (a11), tone frames.
//   ref src/ambe/ambe3600x2450.c:176-621 (decode), :716-877 (policy);
//       src/ambe/ambe_common.c:191-271 (defaults / erasure model); src/core/mbelib.c:691-804 (tones)
// ------------------------------------------------------------------------------------------
__device__ void init_ambe_parms(Parms& p, int lane) {
    p.swn = 0;
    p.tonePhase = 0;
    p.w0 = (float)((M_PI / 32.0) * (2.0 * M_PI));
    p.L = 15;
    p.K = 0;
    p.gamma = 0.0f;
    p.Ml = (lane < MBX_BAND_SLOTS) ? 1.0f : 0.0f;
    p.Vl = 0;
    p.log2Ml = 0.0f;
    p.PHIl = 0.0f;
    p.PSIl = 0.0f;
    p.localEnergy = 75000.0f;
    p.amplitudeThreshold = 20480;
    p.errorRate = 0.0f;
    p.errorCountTotal = 0;
    p.errorCount4 = 0;
    p.repeatCount = 0;
    p.mutingThreshold = MBE_MUTING_THRESHOLD_AMBE;
    p.noiseSeed = -1.0f;
    p.ov[0] = p.ov[1] = 0.0f;
    p.uw[0] = p.uw[1] = p.uw[2] = p.uw[3] = 0.0f;
}

__device__ void set_ambe_erasure_parms(Parms& mp, const Parms& keep, int lane) {
    mp.swn = 0;
    mp.tonePhase = 0;
    mp.w0 = 0.0f;
    mp.L = 9;
    mp.K = 0;
    mp.gamma = 0.0f;
    mp.Ml = (lane < MBX_BAND_SLOTS) ? 1.0f : 0.0f;
    mp.Vl = 0;
    mp.log2Ml = 0.0f;
    mp.PHIl = keep.PHIl;
    mp.PSIl = keep.PSIl;
    mp.localEnergy = 75000.0f;
    mp.amplitudeThreshold = 20480;
    mp.noiseSeed = keep.noiseSeed;
    mp.ov[0] = keep.ov[0];
    mp.ov[1] = keep.ov[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mp.uw[j] = keep.uw[j];
    }
}

__device__ __forceinline__ int pick_bits(const uint32_t w[3], int i0, int i1, int i2, int i3 = -1, int i4 = -1, int i5 = -1,
                                         int i6 = -1, int i7 = -1, int i8 = -1) {
    const int idx[9] = {i0, i1, i2, i3, i4, i5, i6, i7, i8};
    int v = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (idx[k] >= 0) {
            v = (v << 1) | rec_bit(w, idx[k]);
        }
    }
    return v;
}

// Returns 0 voice, 2 erasure, 7 tone (classified by the expand stage).
__device__ int decode_ambe(const float* __restrict__ fp, Parms& cur, Parms& prev, const DeviceTables& tabs, int lane, float* tmp) {
    const int word = uni(__float_as_int(fp[63]));   // AMBE rows: frame class | L << 8 (mbx_expand_ambe.h)
    const int bad = word & 0xff;
    if (bad != 0) {
        return bad;
    }
    cur.w0 = uni(fp[59]);
    const int L = word >> 8;
    cur.L = L;
    const float unvc = uni(fp[60]);   // 0.2046 / sqrtf(w0), formed by the expand stage with the reference's IEEE operations
    const unsigned long long vbits = ((unsigned long long)__float_as_uint(fp[58]) << 32) | __float_as_uint(fp[57]);
    cur.gamma = uni(fp[0] + ((float)0.5 * prev.gamma));
    float Tl = 0.0f;
    if (lane >= 1 && lane <= L) {
        cur.Vl = (int)((vbits >> (lane - 1)) & 1ULL);
        Tl = fp[lane];
    }

    const int prev_L = prev.L < 1 ? 1 : (prev.L > 56 ? 56 : prev.L);
    const int cur_L = L;
    {
        const float padM = lane_read(prev.Ml, prev_L), padL = lane_read(prev.log2Ml, prev_L);   // wave-uniform index: v_readlane, no LDS trip
        if (lane > prev_L && lane <= cur_L) {
            prev.Ml = padM;
            prev.log2Ml = padL;
        }
        const float m1 = lane_read(prev.Ml, 1), l1 = lane_read(prev.log2Ml, 1);
        if (lane == 0) {
            prev.Ml = m1;
            prev.log2Ml = l1;
        }
    }
    const ConstDerived D = (ConstDerived)tabs.d;   // scalar loads
    const float pos = D->l_ratio[uni(prev_L)][cur_L] * (float)lane;
    int lo = (int)pos;
    lo = lo > 56 ? 56 : lo;
    const float frac = pos - (float)lo;
    const float phi0 = lane_read(prev.PHIl, 0);   // the reference's log2Ml[57] aliases PHIl[0]
    const float a = lane_get(prev.log2Ml, lo);
    const float bnext = lane_get(prev.log2Ml, (lo + 1) & 63);
    const float b = (lo + 1 > 56) ? phi0 : bnext;
    const bool in = lane >= 1 && lane <= cur_L;
    float Sum43 = seq_sum4(in ? ((((float)1 - frac) * a) + (frac * b)) : 0.0f, cur_L, tmp, lane);   // feeds the prediction memory
    Sum43 = (D->ambe_pred_over_l[cur_L] * Sum43);
    const float Sum42 = fp[61];   // mean residual, summed in the reference's order by the expand stage
    const float BigGamma = cur.gamma - (0.5f * D->log2_int[cur_L]) - Sum42;
    if (in) {
        const float c1 = ((float)0.65 * ((float)1 - frac) * a);
        const float c2 = ((float)0.65 * frac * b);
        cur.log2Ml = Tl + c1 + c2 - Sum43 + BigGamma;
        const float e = exp2f(cur.log2Ml);
        cur.Ml = (cur.Vl == 1) ? e : (unvc * e);
    }
    return 0;
}

__constant__ float kDualToneHz[36][2] = {
    {1336, 941}, {1209, 697}, {1336, 697}, {1477, 697}, {1209, 770}, {1336, 770}, {1477, 770}, {1209, 852}, {1336, 852},
    {1477, 852}, {1633, 697}, {1633, 770}, {1633, 852}, {1633, 941}, {1209, 941}, {1477, 941}, {1162, 820}, {1052, 606},
    {1162, 606}, {1279, 606}, {1052, 672}, {1162, 672}, {1279, 672}, {1052, 743}, {1162, 743}, {1279, 743}, {1430, 606},
    {1430, 672}, {1430, 743}, {1430, 820}, {1052, 820}, {1279, 820}, {440, 350},  {480, 440},  {620, 480},  {490, 350},
};

__device__ bool tone_freqs(int id, float& f1, float& f2) {   // ref src/internal/mbe_tone.h:14-53
    f1 = f2 = 0.0f;
    if (id == 5) {
        f1 = f2 = 156.25f;
        return true;
    }
    if (id == 6) {
        f1 = f2 = 187.5f;
        return true;
    }
    if (id >= 7 && id <= 122) {
        f1 = f2 = 31.25f * (float)id;
        return true;
    }
    if (id >= 128 && id <= 163) {
        f1 = kDualToneHz[id - 128][0];
        f2 = kDualToneHz[id - 128][1];
        return true;
    }
    return false;
}

__device__ __forceinline__ uint32_t tone_step(double hz) {
    const double step = (hz / 8000.0) * 4294967296.0;
    return step <= 0.0 ? 0u : (uint32_t)(step + 0.5);
}

// sin(2 pi phase / 2^32 - pi/2) = -cos(2 pi phase / 2^32) (ref src/core/mbelib.c:697-706 takes sinf of the float
// angle).  The 32-bit phase splits exactly into a 24-bit part, which v_cos/v_sin take as revolutions, and an 8-bit
// residual carried to first order: ~2e-7 absolute, and none of libm sinf's registers in the rare tone paths.
__device__ __forceinline__ float tone_sample(uint32_t phase) {
    const float r = (float)(phase >> 8) * (1.0f / 16777216.0f);
    const float lo = (float)(phase & 255u) * (float)(6.283185307179586 / 4294967296.0);
    const float c0 = __builtin_amdgcn_cosf(r), s0 = __builtin_amdgcn_sinf(r);
    return -fmaf(-lo, s0, c0);
}

// mbe_synthesizeTonef with a tone id already known to be valid
// tones_off (DeviceTables): the reference built with NOTONES -- silence, tone phases untouched (ref src/core/mbelib.c:747-751)
__device__ void tone_frame(float out[3], const uint32_t w[3], Parms& cur, int lane, int tones_off) {
    if (tones_off) {
        out[0] = out[1] = out[2] = 0.0f;
        return;
    }
    const int u0 = (int)(w[0] >> 20);
    const int u1 = (int)((w[0] >> 8) & 0xfffu);
    const unsigned long long two = ((unsigned long long)w[0] << 32) | w[1];
    const int u3 = (int)((two >> 15) & 0x3fffu);
    const int AD = ((u0 & 0x3f) << 1) + ((u3 >> 4) & 0x1);
    const int ID1 = ((u1 & 0xfff) >> 4);
    float f1, f2;
    out[0] = out[1] = out[2] = 0.0f;
    if (!tone_freqs(ID1, f1, f2) || f1 <= 0.0f) {
        return;
    }
    const bool dual = (f2 > 0.0f) && (fabsf(f2 - f1) > 1e-6f);
    const float clip = (32767.0f * 0.95f) / 7.0f;
    const float gain = (((AD < 0) ? 0.0f : (float)AD) / 127.0f) * clip;
    const uint32_t s1 = tone_step((double)f1);
    const uint32_t s2 = dual ? tone_step((double)f2) : 0u;
    const uint32_t p1 = (uint32_t)cur.swn, p2 = cur.tonePhase;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint32_t n1 = (uint32_t)(lane + 64 * j + 1);
        const float a = tone_sample(p1 + n1 * s1);
        if (dual) {
            const float b = tone_sample(p2 + n1 * s2);
            out[j] = (0.5f * gain * a) + (0.5f * gain * b);
        } else {
            out[j] = gain * a;
        }
    }
    cur.swn = (int)(p1 + 160u * s1);
    cur.tonePhase = dual ? (p2 + 160u * s2) : p2;
}

#ifndef MBX_AMBE_WAVES_PER_SIMD
#define MBX_AMBE_WAVES_PER_SIMD 6       // AMBE+2 3600x2450: 80 VGPRs and 12 bytes of scratch (tools/kres.sh); five waves without the spill are 8 % slower
#endif
#ifndef MBX_AMBE2400_WAVES_PER_SIMD
#define MBX_AMBE2400_WAVES_PER_SIMD 6   // AMBE 3600x2400: 77 VGPRs, no scratch
#endif
// D-STAR single tone (ref src/core/mbelib.c:813-856 + :708-736): 156.25 Hz (index 5), 187.5 Hz (6) or 31.25 Hz x index
// (7..122) at the fixed amplitude 103
__device__ void tone_dstar_frame(float out[3], int id1, Parms& cur, int lane, int tones_off) {
    if (tones_off) {   // (ref src/core/mbelib.c:815-819)
        out[0] = out[1] = out[2] = 0.0f;
        return;
    }
    out[0] = out[1] = out[2] = 0.0f;
    float f1 = 0.0f;
    if (id1 == 5) {
        f1 = 156.25f;
    } else if (id1 == 6) {
        f1 = 187.5f;
    } else if (id1 >= 7 && id1 <= 122) {
        f1 = 31.25f * (float)id1;
    }
    if (f1 <= 0.0f) {
        return;
    }
    const float clip = (32767.0f * 0.95f) / 7.0f;
    const float gain = ((float)103 / 127.0f) * clip;
    const uint32_t s1 = tone_step((double)f1);
    const uint32_t p1 = (uint32_t)cur.swn;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        out[j] = gain * tone_sample(p1 + (uint32_t)(lane + 64 * j + 1) * s1);
    }
    cur.swn = (int)(p1 + 160u * s1);
}

// k2400: AMBE 3600x2400 (D-STAR) frame policy, ref src/ambe/ambe3600x2400.c:629-763 -- no erasure class, D-STAR
// tones, repeats decided by the total error count alone.  The prediction (decode_ambe) is common.
// kFuse = 3 (with kOne): the stream blocks of ambe_one_launch_kernel -- see imbe_stream_body and FrontLink
template <bool k2400, bool kPark, bool kFrame = false, bool kRes = false, bool kOne = false, int kFuse = 0>   // kOne: see imbe_stream_body
__device__ __forceinline__ void
ambe_stream_body(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                 mbe_parms* __restrict__ state,
                 mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16, float* __restrict__ pcmf,
                 mbe_process_result* __restrict__ results, DeviceTables tabs_in, const uint8_t* frame_in = nullptr,
                 FrameShadow shadow = FrameShadow{}, FrameSlice slice = FrameSlice{}, FrontLink link = FrontLink{}) {
    uint4 rec_in = make_uint4(0u, 0u, 0u, 0u);
    using ScratchT = WaveScratchT<kPark ? 0 : MBX_PARK_N>;
    ScratchT* scratch_ptr;
    if constexpr (kFuse == 3) {
        scratch_ptr = reinterpret_cast<ScratchT*>(link.lds);
    } else {
        __shared__ ScratchT scratch_own;
        scratch_ptr = &scratch_own;
    }
    ScratchT& scratch = *scratch_ptr;
    __shared__ std::conditional_t<kPark, ParkedPrevOnly, ParkedState<false>> park;   // T >= 4: prev_mp resident in LDS
    // kPark without a workspace (params == nullptr, the normal case): the wave expands the records of its next EIGHT frames
    // itself, eight lanes per frame exactly like the expand kernels (mbx_expand_ambe.h), into eight LDS rows -- no
    // expand launch, no 256-byte row per frame through HBM.  (One frame at a time fills 8 of 64 lanes: round 2 measured
    // that slower than the separate launch.)  Row stride 65 dwords: the eight frames write the same columns at once.
#ifndef MBX_AMBE_XROWS
#define MBX_AMBE_XROWS 8
#endif
    constexpr int kXRows = MBX_AMBE_XROWS, kXStride = 65;
    __shared__ float xrows[kPark ? kXRows : 1][kPark ? kXStride : 1];
    const int bpos = (kFuse == 3) ? link.pos : (slice.pos >= 0 ? slice.pos : (int)blockIdx.x);
    if (bpos >= S) {
        return;
    }
    const int s = tabs_in.reverse ? (S - 1 - bpos) : bpos;
    const size_t fbase = (size_t)s * (size_t)(slice.stride ? slice.stride : Tn) + (size_t)slice.t0;   // the batch index of the slice's first frame
    const int lane_in = lane_id();
    Wire wire_in = {};
    if constexpr (kFrame) {
        wire_in = shadow.have_frame ? frame_from_args(true, shadow) : frame_fetch(true, frame_in);   // ahead of every state load
    }

    // Same register discipline as the IMBE kernel: `cur` resident, `prev` / `enh` parked in their slots.
    // batch row s (frames, records, PCM, results) belongs to state / rng slot `slot`: the same number unless the caller
    // passed an index (mbx_process_batch_indexed: the streams that have frames this tick, out of a larger resident pool)
    const size_t slot = tabs_in.stream_map ? (size_t)uni(tabs_in.stream_map[s]) : (size_t)s;
    mbe_parms* const slot_cur = &state[3 * slot + 0];
    mbe_parms* const home_prev = &state[3 * slot + 1];
    mbe_parms* const home_enh = &state[3 * slot + 2];
    mbe_parms* const slot_enh = home_enh;
    mbe_parms* slot_prev;
    // kPark: prev_mp_enhanced has no copy in LDS.  The fields synthesis reads travel from frame to frame in registers
    // (enh_keep), and the struct itself is only ever needed whole by the rare invalid-tone replay.  `synced` says that
    // prev_mp_enhanced == cur_mp field for field (true after every frame that ends with prev_mp_enhanced := cur_mp: voice,
    // erasure, re-initialisation); while it holds, the HBM home of prev_mp_enhanced is stale and `cur` is its only copy.
    // A tone frame leaves prev_mp_enhanced alone while it changes cur_mp, so a synced wave first writes `cur` to the home
    // (before the frame touches it); from then on the home is current until the next frame that syncs again.
    Parms enh_keep;
    bool synced = false;
    Parms cur;
    StreamRng rng;
    // resident launches (DeviceTables::resident): `synced` is carried from launch to launch in resident[slot], the struct of
    // prev_mp_enhanced is neither read nor written while it holds, and prev_mp comes into LDS lazily (see the IMBE kernel)
    uint32_t* const res = (kPark && kRes) ? tabs_in.resident : nullptr;
    // kRes without kPark (with kOne): the HBM-slot instance for resident launches of ONE frame per stream (cf. imbe_stream_kernel_res1).
    // `elided`: prev_mp_enhanced of the stream is elided (== cur_mp field for field): its view comes from cur_mp's struct, a frame that
    // ends with prev_mp_enhanced := cur_mp stores nothing, and a frame that will leave prev_mp_enhanced alone while it changes cur_mp
    // (tone class) first writes the still-unchanged cur_mp out as prev_mp_enhanced.
    uint32_t* const res1 = (!kPark && kRes) ? tabs_in.resident : nullptr;
    bool elided = res1 && (uni(res1[slot]) != 0u);
    bool prev_partial = false;
    // every request of the launch's first frame goes out before anything waits: see the IMBE body and load_header
    Parms enh_first, prev_first;
    uint32_t h_enh_first = 0u, h_prev_first = 0u;
    bool enh_from_cur = false;   // resident one-frame instances: the enhanced view is cur_mp's own registers (enh_view_of)
    float row_first = 0.0f;
    if constexpr (kPark) {
        slot_prev = &park.prev;
        synced = res && (uni(res[slot]) != 0u);
        if constexpr (!kFrame) {   // (the long launches keep the struct-by-struct start: see load_parms)
            load_enh_view(enh_keep, synced ? slot_cur : home_enh, lane_in);
            if (res) {
                copy_prev_view(slot_prev, home_prev, lane_in);
                prev_partial = true;
                load_parms<kOne>(cur, slot_cur, lane_in);
                load_rng(rng, &rngs[slot]);
            } else {
                Parms home;
                load_parms<kOne>(home, home_prev, lane_in);
                load_parms<kOne>(cur, slot_cur, lane_in);
                load_rng(rng, &rngs[slot]);
                store_parms<kOne && MBX_AMBE_GATHER_STORES>(home, slot_prev, lane_in);
            }
            wave_lds_sync();
        } else {
        const bool from_shadow = shadow.use != 0u;   // (single-frame kernels: see FrameShadow)
        const mbe_parms* const in_cur = from_shadow ? &shadow.state[0] : slot_cur;
        const mbe_parms* const in_prev = from_shadow ? &shadow.state[1] : home_prev;
        const mbe_parms* const enh_src = from_shadow ? &shadow.state[2] : (synced ? slot_cur : home_enh);
        const uint32_t h_enh = load_header(enh_src, lane_in);
        const uint32_t h_cur = load_header(in_cur, lane_in);
        if (res) {
            load_enh_arrays(enh_keep, enh_src, lane_in);
            copy_prev_view(slot_prev, home_prev, lane_in);
            prev_partial = true;
            load_parms_arrays(cur, slot_cur, lane_in);
            load_rng(rng, &rngs[slot]);
        } else {
            Parms home;
            const uint32_t h_home = load_header(in_prev, lane_in);
            load_enh_arrays(enh_keep, enh_src, lane_in);
            load_parms_arrays(home, in_prev, lane_in);
            load_parms_arrays(cur, in_cur, lane_in);
            load_rng(rng, from_shadow ? shadow.rng : &rngs[slot]);
            if constexpr (kFrame) {   // the FEC of the frame runs while the three structs are on their way (pinned host memory: PCIe)
                rec_in = frame_record(MBX_CODEC_AMBE3600X2450, wire_in, const_cast<mbx_param_record*>(records), tabs_in.t, lane_in);
            }
            set_parms_header(home, h_home);
            store_parms<kOne && MBX_AMBE_GATHER_STORES>(home, slot_prev, lane_in);
        }
        set_enh_header(enh_keep, h_enh);
        set_parms_header(cur, h_cur);
        wave_lds_sync();
        }
    } else if constexpr (kOne) {
        slot_prev = home_prev;
        uint32_t flag_v = 0u;
        if constexpr (kFuse == 3) {
            flag_v = __hip_atomic_load(link.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the wave's first vector load (bypasses L1)
            asm volatile("" ::: "memory");
        } else {
            row_first = params[fbase].v[lane_in];   // the frame's FrameParams row: needed first, requested first
        }
        const mbe_parms* const enh_src = elided ? slot_cur : slot_enh;
        const uint32_t h_cur = load_header(slot_cur, lane_in);
        if constexpr (!(kRes && MBX_RES_VIEW_FROM_CUR)) {
            h_prev_first = load_header(slot_prev, lane_in);
            h_enh_first = load_header(enh_src, lane_in);
        }
        load_rng(rng, &rngs[slot]);
        load_prev_arrays(prev_first, slot_prev, lane_in);
        load_parms_arrays(cur, slot_cur, lane_in);
        if constexpr (kRes && MBX_RES_VIEW_FROM_CUR) {
            asm volatile("" ::: "memory");   // (everything above is requested before the flag is looked at)
            enh_from_cur = elided;
            if (!elided) {   // (elided: both come from cur_mp's registers where the frame loop starts, see enh_view_of / prev_header_of)
                h_prev_first = load_header(slot_prev, lane_in);
                h_enh_first = load_header(slot_enh, lane_in);
                load_enh_arrays(enh_first, slot_enh, lane_in);
            }
        } else {
            load_enh_arrays(enh_first, enh_src, lane_in);
        }
        if constexpr (kFuse == 3) {   // the row comes from a front block of the same launch: see the IMBE body
            asm volatile("" ::: "memory");
            bool ready = uni(flag_v) == link.epoch;
            for (int tries = 0; !ready && tries < MBX_FRONT_SPIN; ++tries) {
                __builtin_amdgcn_s_sleep(8);
                ready = uni(__hip_atomic_load(link.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == link.epoch;
            }
            if (ready) {
                const uint32_t* const rowp = reinterpret_cast<const uint32_t*>(&params[fbase].v[0]);
                row_first = __uint_as_float(__hip_atomic_load(rowp + lane_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            } else {   // (never observed) the wave does its frame's front end itself: scalar-unit FEC, expansion by its first eight lanes
                const Wire wire = frame_fetch(true, frame_in + 9u * (size_t)s);
                rec_in = frame_record(MBX_CODEC_AMBE3600X2450, wire, const_cast<mbx_param_record*>(&records[fbase]), tabs_in.t, lane_in);
                float (*tile)[65] = reinterpret_cast<float (*)[65]>(link.lds);   // eight rows: every group of eight lanes takes part
                xp::expand_ambe_frame_rec<k2400>((lane_in >> 3) == 0, xp::u32x4{rec_in.x, rec_in.y, rec_in.z, rec_in.w}, tile[lane_in >> 3], lane_in & 7,
                                                 tabs_in);
                wave_lds_sync();
                row_first = tile[0][lane_in];
                wave_lds_sync();
                if (link.fallbacks && lane_in == 0) {
                    atomicAdd(link.fallbacks, 1u);
                }
            }
        }
        set_parms_header(cur, h_cur);
    } else {
        slot_prev = home_prev;
        load_parms<kOne>(cur, slot_cur, lane_in);
        load_rng(rng, &rngs[slot]);
    }
    auto keep_enh_view = [&](const Parms& from) {
        enh_keep = Parms{};
        enh_keep.w0 = from.w0;
        enh_keep.L = from.L;
        enh_keep.Vl = from.Vl;
        enh_keep.Ml = from.Ml;
        enh_keep.PHIl = from.PHIl;
        enh_keep.PSIl = from.PSIl;
        enh_keep.localEnergy = from.localEnergy;
        enh_keep.amplitudeThreshold = from.amplitudeThreshold;
        enh_keep.uw[2] = from.uw[2];
        enh_keep.uw[3] = from.uw[3];
    };
    auto complete_prev = [&](int lane) {   // resident launch: a frame is about to read more of prev_mp than the decode's view
        if (prev_partial) {
            copy_parms<kOne>(slot_prev, home_prev, lane);
            slot_fence<kPark>();
            prev_partial = false;
        }
    };
    float row_now = 0.0f;
    if constexpr (kPark) {
        if (params) {
            row_now = params[fbase].v[lane_in];
        }
    }

    const int frames = (kOne && Tn > 1) ? 1 : Tn;
    for (int t = 0; t < frames; ++t) {
        const size_t f = fbase + (size_t)t;
        // Keep per-frame table values out of the loop-carried register set: without this the compiler
        // hoists ~100 VGPRs of lane-dependent values (twiddles, windows, jump-ahead constants, indices)
        // across the frame loop, which halves the occupancy.
        DeviceTables ft = tabs_in;
        int lane = lane_in;
        asm volatile("" : "+s"(ft.t), "+s"(ft.d), "+v"(lane));
        if constexpr (kPark) {
            // Tell the compiler again that the lane index is 0..63: struct fields are then addressed as SGPR base + 32-bit
            // lane offset instead of 64-bit per-lane addresses (two VALU and a register pair each).  Only where registers are
            // plentiful: under the 72 / 80-register caps of the HBM-slot instances the extra freedom ends in spills.
            lane &= 63;
        }
        const DeviceTables& tabs = ft;
        const float* fp;
        if constexpr (kPark) {
            if (params) {
                // rows from a workspace (mbx_stream_expanded): the frame's row goes through LDS, and the NEXT frame's row is
                // requested now -- with few waves per SIMD a global load in front of every decode is not hidden
                scratch.x.fp[lane] = row_now;
                wave_lds_sync();
                fp = scratch.x.fp;
                if (t + 1 < Tn) {
                    row_now = params[f + 1].v[lane];
                }
            } else {
                if ((t & (kXRows - 1)) == 0) {
                    const int q = lane >> 3;
                    const bool have = (t + q) < Tn;
                    wave_lds_sync();   // the previous eight rows have been read
                    if (kXRows == 8 || q < kXRows) {   // (fewer rows than frame slots in the wave: the upper lanes sit the pass out)
                        if constexpr (kFrame) {
                            xp::expand_ambe_frame_rec<k2400>(q == 0, xp::u32x4{rec_in.x, rec_in.y, rec_in.z, rec_in.w}, xrows[q], lane & 7, tabs);
                        } else {
                            xp::expand_ambe_frame<k2400>(have, &records[have ? f + (size_t)q : f], xrows[q], lane & 7, tabs);
                        }
                    }
                    wave_lds_sync();
                }
                fp = xrows[t & (kXRows - 1)];

            }
        }
        // The parts of prev_mp_enhanced that synthesis reads are requested together with prev_mp, so that one
        // memory latency covers both (as in the IMBE kernel) -- and, in the HBM-slot instances, together with the frame's
        // FrameParams row: the row is requested FIRST, the two views right behind it, and the row then goes through LDS, so the
        // first wave-uniform read of it (the error word) waits for one load, not for a round trip before the views are even
        // requested.
        Parms enh;
        Parms prev;
        if constexpr (kPark) {
            enh = enh_keep;
            load_prev_view_lds(prev, slot_prev, lane);
        } else if constexpr (kOne) {   // requested at the start, together with cur_mp
            prev = prev_first;
            scratch.x.fp[lane] = row_first;
            if (kRes && MBX_RES_VIEW_FROM_CUR && enh_from_cur) {   // cur_mp still holds what the last frame left: that IS prev_mp_enhanced
                prev_header_of(prev, cur);
                enh_view_of(enh, cur);
                enh.w0 = cur.w0;
                enh.L = cur.L;
                enh.localEnergy = cur.localEnergy;
                enh.amplitudeThreshold = cur.amplitudeThreshold;
            } else {
                set_prev_header(prev, h_prev_first);
                enh = enh_first;
                set_enh_header(enh, h_enh_first);
            }
            wave_lds_sync();
            fp = scratch.x.fp;
        } else {
            const float row = params[f].v[lane];
            load_enh_view(enh, slot_enh, lane);
            load_prev_view(prev, slot_prev, lane);
            scratch.x.fp[lane] = row;
            wave_lds_sync();
            fp = scratch.x.fp;
        }
        const uint32_t errw = uni(__float_as_uint(fp[62]));
        const int c0 = (int)(errw & 0xffu), prot = (int)((errw >> 8) & 0xffu);
        unsigned flags = (errw >> 24) & 0xffu;   // C0_VALID
        const int total = c0 + prot;
        int bad;
        bool prev_max_repeat, valid_tone = false;
        {
            // prepare (ambe2450_prepare_process): state that came from the generic initialiser is
            // replaced by the AMBE defaults in all three structs
            if (fabsf(prev.mutingThreshold - MBE_MUTING_THRESHOLD_AMBE) > 1e-6f) {
                init_ambe_parms(prev, lane);
                cur = prev;
                store_parms<kOne && MBX_AMBE_GATHER_STORES>(prev, slot_prev, lane);
                prev_partial = false;
                if constexpr (kPark) {
                    keep_enh_view(prev);
                    enh = enh_keep;
                    synced = true;
                } else {
                    store_parms<kOne && MBX_AMBE_GATHER_STORES>(prev, slot_enh, lane);
                    elided = false;   // (all three structs have just been written whole)
                    slot_fence<kPark>();
                    load_enh_view(enh, slot_enh, lane);
                }
                slot_fence<kPark>();
            }
            if constexpr (kPark) {
                // a frame that will leave prev_mp_enhanced alone (AMBE+2 tone class, valid D-STAR tone) while it changes cur_mp:
                // `cur` is still prev_mp_enhanced field for field here (see `synced` above) -- write it home first
                const int cls = uni(__float_as_int(fp[63])) & 0xff;
                const int c0v_early = ((flags & MBE_PROCESS_FLAG_C0_VALID) != 0u) ? c0 : 0;
                const bool keeps_enh = k2400 ? ((cls >= 7) && (cls <= 122) && (c0v_early < 2) && (total < 3)) : (cls == 7);
                if (keeps_enh && synced) {
                    store_parms<kOne && MBX_AMBE_GATHER_STORES>(cur, home_enh, lane);
                    synced = false;
                }
            }
            if constexpr (!kPark && kRes) {   // the same for the one-frame resident instance: prev_mp_enhanced leaves its elision
                const int cls = uni(__float_as_int(fp[63])) & 0xff;
                const int c0v_early = ((flags & MBE_PROCESS_FLAG_C0_VALID) != 0u) ? c0 : 0;
                const bool keeps_enh = k2400 ? ((cls >= 7) && (cls <= 122) && (c0v_early < 2) && (total < 3)) : (cls == 7);
                if (keeps_enh && elided) {
                    store_parms<kOne && MBX_AMBE_GATHER_STORES>(cur, slot_enh, lane);
                    elided = false;
                }
            }
            cur.mutingThreshold = MBE_MUTING_THRESHOLD_AMBE;
            cur.errorCountTotal = total;
            cur.errorCount4 = 0;
            cur.errorRate = uni((0.95f * prev.errorRate) + (0.001064f * (float)total));

            bad = decode_ambe(fp, cur, prev, tabs, lane, scratch.x.C);
            prev_max_repeat = prev.repeatCount >= MBE_MAX_FRAME_REPEATS;
            if (k2400) {   // ambe2400_update_decode_state (:661-686)
                const int c0v = ((flags & MBE_PROCESS_FLAG_C0_VALID) != 0u) ? c0 : 0;
                valid_tone = (bad >= 7) && (bad <= 122) && (c0v < 2) && (total < 3);
                if (bad == 3) {
                    flags |= MBE_PROCESS_FLAG_TONE;
                    cur.repeatCount = 0;
                } else if (valid_tone) {
                } else if (total > 3) {
                    complete_prev(lane);
                    load_parms<kOne>(cur, slot_prev, lane);   // cur_mp := prev_mp
                    if (bad == 0) {                     // (the decode padded the prediction memory in registers)
                        cur.Ml = prev.Ml;
                        cur.log2Ml = prev.log2Ml;
                    }
                    cur.repeatCount++;
                    flags |= MBE_PROCESS_FLAG_REPEAT;
                } else {
                    cur.repeatCount = 0;
                }
            } else if (bad == 2) {
                flags |= MBE_PROCESS_FLAG_ERASURE;
                cur.repeatCount = 0;
                complete_prev(lane);
                load_continuity(prev, slot_prev, lane);   // phases, overlap-add and noise state of prev_mp
                set_ambe_erasure_parms(cur, prev, lane);
            } else if (bad == 7) {
                flags |= MBE_PROCESS_FLAG_TONE;
                cur.repeatCount = 0;
            } else if (((flags & MBE_PROCESS_FLAG_C0_VALID) != 0u) ? ((c0 >= 4) || ((c0 >= 2) && (total >= 6))) : (total > 3)) {
                complete_prev(lane);
                load_parms<kOne>(cur, slot_prev, lane);   // cur_mp := prev_mp (see the IMBE kernel)
                cur.Ml = prev.Ml;
                cur.log2Ml = prev.log2Ml;
                cur.repeatCount++;
                flags |= MBE_PROCESS_FLAG_REPEAT;
            } else {
                cur.repeatCount = 0;
            }
        }

        // One call site each for the synthesiser and the noise generator (code size: the synthesiser is
        // ~20 KB of instructions and the instruction cache holds 64 KB).
        enum { kVoice, kToneFallback, kTone, kToneDstar, kNoiseReinit, kNoiseErasure } action;
        uint32_t tw[3] = {0u, 0u, 0u};
        if (k2400 && bad != 0) {   // ambe2400_synthesize_frame (:711-731)
            action = valid_tone ? kToneDstar : kNoiseReinit;
        } else if (bad == 0) {
            action = (cur.repeatCount < MBE_MAX_FRAME_REPEATS) ? kVoice : kNoiseReinit;
            if (action == kNoiseReinit) {
                flags |= MBE_PROCESS_FLAG_MUTE;
            }
        } else if (bad == 7) {
            uint4 rec;
            if constexpr (kFrame) {
                rec = rec_in;
            } else if constexpr (kFuse == 3) {   // written by a front block of this launch: read past the L1, like the row (see FrontLink)
                const uint32_t* const rp = reinterpret_cast<const uint32_t*>(&records[f]);
                rec = make_uint4(__hip_atomic_load(rp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                 __hip_atomic_load(rp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                 __hip_atomic_load(rp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                 __hip_atomic_load(rp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            } else {
                rec = *reinterpret_cast<const uint4*>(&records[f]);
            }
            tw[0] = rec.x;
            tw[1] = rec.y;
            tw[2] = rec.z;
            const int id1 = (int)((tw[0] >> 12) & 0xffu);   // parameter bits 12..19
            float f1, f2;
            action = tone_freqs(id1, f1, f2) ? kTone : (!prev_max_repeat ? kToneFallback : kNoiseReinit);
        } else {
            action = kNoiseErasure;
        }

        float out[3];
        float rm0 = 0.0f;
        if (action == kVoice) {
            store_parms<kOne && MBX_AMBE_GATHER_STORES>(cur, slot_prev, lane);   // prev_mp := cur_mp
            prev_partial = false;
            cur.log2Ml = 0.0f;                   // read back from the snapshot after the synthesiser (see the IMBE kernel)
            cur.uw[0] = cur.uw[1] = cur.uw[2] = cur.uw[3] = 0.0f;
            rm0 = enhance(cur, lane, scratch.x.C);
        } else if (action == kToneFallback) {
            // invalid tone id: run the synthesiser on a copy of the enhanced model.  `cur` is parked in
            // its slot meanwhile so that still only two structs are live.
            store_parms<kOne && MBX_AMBE_GATHER_STORES>(cur, slot_cur, lane);
            if constexpr (kPark) {
                // The copy takes the LDS home of prev_mp for the duration (it is the synthesiser's snapshot), prev_mp waits in
                // its HBM home.  prev_mp_enhanced is current in ITS home: this is a tone-class frame (see `synced`).
                wave_lds_sync();
                if (!prev_partial) {   // (a resident launch that has only the view in LDS: the home is current as it is)
                    copy_parms<kOne>(home_prev, slot_prev, lane);
                }
                __threadfence_block();
                copy_parms<kOne>(slot_prev, home_enh, lane);
                slot_fence<kPark>();
                load_parms<kOne>(cur, slot_prev, lane);
            } else {
                __threadfence_block();               // (slot_cur is always the HBM slot)
                load_parms<kOne>(cur, slot_enh, lane);     // the copy that is synthesised ...
                load_enh_view(enh, slot_enh, lane);  // ... against the enhanced model itself (only the fields synthesis reads)
            }
            cur.log2Ml = 0.0f;   // slot_enh itself is the snapshot of this copy
            cur.uw[0] = cur.uw[1] = cur.uw[2] = cur.uw[3] = 0.0f;
        }
        if (action == kVoice || action == kToneFallback) {
            const mbe_parms* snap = (kPark || action == kVoice) ? slot_prev : slot_enh;
            const bool fresh = synth_core<true, kPark, ScratchT, kOne && MBX_AMBE_EARLY_NOISE>(out, cur, enh, action == kVoice, rm0, rng, scratch, tabs, lane, snap);
            {
                slot_fence<kPark>();
                const float* f = reinterpret_cast<const float*>(snap);
                if constexpr (!kPark) {
                    asm volatile("" : "+s"(f));   // a real load (see the IMBE kernel)
                }
                if (lane < MBX_BAND_SLOTS) {
                    cur.log2Ml = slot_read<kPark>(f, O_LOG2ML + lane);
                }
                if (!fresh) {   // silence / comfort noise: previousUw and the noise overlap are unchanged
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        cur.uw[j] = slot_read<kPark>(f, O_UW + lane + 64 * j);
                    }
                    cur.ov[0] = slot_read<kPark>(f, O_OVERLAP + lane);
                    cur.ov[1] = (lane < 32) ? slot_read<kPark>(f, O_OVERLAP + 64 + lane) : 0.0f;
                }
            }
            if constexpr (kPark) {   // prev_mp_enhanced := synthesised model
                keep_enh_view(cur);
                if (action == kVoice) {
                    synced = true;
                } else {
                    store_parms<kOne && MBX_AMBE_GATHER_STORES>(cur, home_enh, lane);    // the replayed copy is NOT cur_mp: it goes to the home (synced stays false)
                    wave_lds_sync();
                    __threadfence_block();
                    copy_parms<kOne>(slot_prev, home_prev, lane);   // prev_mp returns to LDS
                    prev_partial = false;
                }
            } else if (res1 && action == kVoice) {
                elided = true;   // prev_mp_enhanced := cur_mp, not written
            } else {
                store_parms<kOne && MBX_AMBE_GATHER_STORES>(cur, slot_enh, lane);   // (the replayed copy of an invalid tone is NOT cur_mp: it is stored)
            }
            if (action == kToneFallback) {
                __threadfence_block();
                load_parms<kOne>(cur, slot_cur, lane);
            }
        } else if (action == kTone) {
            tone_frame(out, tw, cur, lane, tabs.tones_off);
        } else if (action == kToneDstar) {
            tone_dstar_frame(out, bad, cur, lane, tabs.tones_off);
            store_parms<kOne && MBX_AMBE_GATHER_STORES>(cur, slot_prev, lane);   // mbe_moveMbeParms(cur_mp, prev_mp)
            prev_partial = false;
        } else {
            comfort_noise(out, rng, lane);
            if (action == kNoiseReinit) {   // mbe_initAmbeParms_common(cur, prev, prev_enhanced)
                init_ambe_parms(cur, lane);
            }
            store_parms<kOne && MBX_AMBE_GATHER_STORES>(cur, slot_prev, lane);
            prev_partial = false;
            if constexpr (kPark) {
                keep_enh_view(cur);
                synced = true;
            } else if (res1) {
                elided = true;
            } else {
                store_parms<kOne && MBX_AMBE_GATHER_STORES>(cur, slot_enh, lane);
            }
        }
        if (t + 1 < Tn) {
            slot_fence<kPark>();   // the next frame of this wave reloads the parked structs
        }

        store_pcm(out, f, pcm16, pcmf, lane);
        if (results && lane == 0) {
            mbe_process_result r;
            r.c0_errors = c0;
            r.protected_errors = total - c0;
            r.c4_errors = 0;
            r.total_errors = total;
            r.flags = flags;
            results[f] = r;
        }
    }

    {
        // The struct's per-lane address (slot + 4 lane, a VGPR pair) was formed for the loads at the top of the wave; kept for this store it
        // was the T = 2, 3 instance's 12-byte spill.  Laundering the wave-uniform base through the scalar file makes
        // the store form its address again from the SGPR pair: one v_lshl_add_u64 instead of a scratch round trip.
        mbe_parms* slot_cur_again = slot_cur;
        asm volatile("" : "+s"(slot_cur_again));
        store_parms<kOne && MBX_AMBE_GATHER_STORES>(cur, slot_cur_again, lane_in);
    }
    store_rng(rng, &rngs[slot], lane_in);
    if (res1 && lane_in == 0) {
        res1[slot] = elided ? 1u : 0u;
    }
    if constexpr (kPark) {   // prev_mp goes home from LDS; prev_mp_enhanced from `cur` unless its home is already current
        if (res) {
            if (lane_in == 0) {
                res[slot] = synced ? 1u : 0u;   // a resident launch leaves the struct elided while it equals cur_mp
            }
        } else if (synced) {
            store_parms<kOne && MBX_AMBE_GATHER_STORES>(cur, home_enh, lane_in);
        }
        wave_lds_sync();
        if (!prev_partial) {
            copy_parms<kOne>(home_prev, slot_prev, lane_in);
        }
        if constexpr (kFrame) {
            if (shadow.state) {   // the device copy of what has just gone to the caller; complete only if prev_mp_enhanced is cur_mp
                store_parms<kOne && MBX_AMBE_GATHER_STORES, false>(cur, &shadow.state[0], lane_in);   // (a tone frame leaves prev_mp_enhanced at its pinned home alone: the next
                if (synced) {                                  //  call then takes the state from the caller's structs again)
                    store_parms<kOne && MBX_AMBE_GATHER_STORES, false>(cur, &shadow.state[2], lane_in);
                }
                copy_parms(&shadow.state[1], slot_prev, lane_in);
                store_rng(rng, shadow.rng, lane_in);
                if (lane_in == 0) {
                    *shadow.ok = synced ? 1u : 0u;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Single-frame kernels: mbe_process*Frame[f] of the synchronous per-frame API as ONE launch of ONE wavefront.
//   FEC by lane 0 (mbx_fec_frame.h; the record stays in registers and is also written out), then the LDS-resident stream
//   body with S = T = 1: every read of the three structs is issued at the start and every write at the end, nothing in
//   between waits on the state's memory -- which for this path is pinned HOST memory behind PCIe (libmbe_neo_amd.so hands
//   the caller's structs over in place).  The last thing the wave does is store `token` to *done with system scope after a
//   system-scope fence: the host polls that word instead of paying a stream synchronisation.
//   ref include/mbelib-neo/mbelib.h:429,505,564,352 (mbe_process*Frame[f]); the two calls they make,
//       mbe_decode*Frame + mbe_process*Dataf: src/imbe/imbe7200x4400.c:709-744,780-888 and counterparts.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
imbe_frame_kernel(int codec, const uint8_t* __restrict__ frame, mbx_param_record* __restrict__ record, mbe_parms* __restrict__ state,
                  mbx_stream_rng* __restrict__ rng, int16_t* __restrict__ pcm16, float* __restrict__ pcmf,
                  mbe_process_result* __restrict__ result, uint32_t* done, uint32_t token, DeviceTables tabs, FrameShadow shadow) {
    imbe_stream_body<true, true>(1, 1, record, nullptr, state, rng, pcm16, pcmf, result, tabs, frame, codec, shadow);
    frame_done(done, token, lane_id());
#ifdef MBX_FRAME_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (lane_id() == 0) {
        g_frame_stamps[9] = wall_clock64();
    }
#endif
}

template <bool k2400>
__device__ __forceinline__ void ambe_frame_body(const uint8_t* __restrict__ frame, mbx_param_record* __restrict__ record,
                                                mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rng,
                                                int16_t* __restrict__ pcm16, float* __restrict__ pcmf,
                                                mbe_process_result* __restrict__ result, uint32_t* done, uint32_t token,
                                                const DeviceTables& tabs, const FrameShadow& shadow) {
    ambe_stream_body<k2400, true, true>(1, 1, record, nullptr, state, rng, pcm16, pcmf, result, tabs, frame, shadow);
    frame_done(done, token, lane_id());
}

__global__ void __launch_bounds__(64)
ambe_frame_kernel(const uint8_t* __restrict__ frame, mbx_param_record* __restrict__ record, mbe_parms* __restrict__ state,
                  mbx_stream_rng* __restrict__ rng, int16_t* __restrict__ pcm16, float* __restrict__ pcmf,
                  mbe_process_result* __restrict__ result, uint32_t* done, uint32_t token, DeviceTables tabs, FrameShadow shadow) {
    ambe_frame_body<false>(frame, record, state, rng, pcm16, pcmf, result, done, token, tabs, shadow);
}

__global__ void __launch_bounds__(64)
ambe2400_frame_kernel(const uint8_t* __restrict__ frame, mbx_param_record* __restrict__ record, mbe_parms* __restrict__ state,
                      mbx_stream_rng* __restrict__ rng, int16_t* __restrict__ pcm16, float* __restrict__ pcmf,
                      mbe_process_result* __restrict__ result, uint32_t* done, uint32_t token, DeviceTables tabs, FrameShadow shadow) {
    ambe_frame_body<true>(frame, record, state, rng, pcm16, pcmf, result, done, token, tabs, shadow);
}

// ------------------------------------------------------------------------------------------
// Frame server (include/mbx.h, mbx_frame_mailbox): the single-frame kernels as ONE wavefront that stays on the device and takes
// its requests from pinned host memory.  What a request saves against a launch of its own: the launch (~5 us of host + command
// processor) and the instruction fetch of a kernel whose instruction cache starts cold at every dispatch.
// Exits: quit flag, idle time-out, a hard life-time cap, and -- should the clock ever misbehave -- a poll count; after storing
// alive = 0 it touches nothing, so the host may start a successor at once.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
frame_server_kernel(mbx_frame_mailbox* mb, unsigned idle_ticks /* of the 100 MHz wall clock */, mbe_parms* state, mbx_stream_rng* rng,
                    int16_t* pcm16, float* pcmf, mbe_process_result* result, mbx_param_record* record, DeviceTables tabs,
                    FrameShadow shadow_in) {
    __shared__ uint32_t frame_words[8];
    const int lane = lane_id();
    const uint32_t* const line = reinterpret_cast<const uint32_t*>(mb);   // the request line: dword `lane` of it, lanes 0..15
    auto poll = [&]() -> uint32_t {   // ONE 64-byte read of host memory, system-coherent, never from the scalar cache
        return (lane < 16) ? __hip_atomic_load(line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
    };
    uint32_t last = uni(__hip_atomic_load(&mb->seq_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    uint32_t served = 0u;
    const unsigned long long t_start = wall_clock64();
    unsigned long long t_last = t_start;
    unsigned polls = 0u;
    constexpr unsigned long long kMaxLifeTicks = 200000000ULL;   // 2 s
    constexpr unsigned kMaxIdlePolls = 1u << 24;                 // (backstop should the clock misbehave)
    for (;;) {
        const uint32_t v = poll();
        const uint32_t in = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
        if (in != last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // the caller's structs: not from stale cache lines
            __builtin_amdgcn_s_dcache_inv();                //   (the scalar cache is not covered by the fence)
            // The request line is read AGAIN after the acquire: nothing guarantees that the 64-byte poll above was one atomic read
            // across PCIe -- split into 32-byte sectors it could pair a new seq_in (offset 0) with stale frame bytes 16.. (offset 32+).
            // The host writes codec, want and the frame with plain stores and seq_in LAST with release, so every byte read after the
            // new seq_in has been seen (and after the fence) is the request's.
            const uint32_t v2 = poll();
            const int codec = __builtin_amdgcn_readlane((int)v2, 2);
            const uint32_t want = (uint32_t)__builtin_amdgcn_readlane((int)v2, 3);
            if (lane >= 4 && lane < 10) {
                frame_words[lane - 4] = v2;   // the wire frame came with the request: the FEC starts without another trip to the host
            }
            wave_lds_sync();
            const uint8_t* frame = reinterpret_cast<const uint8_t*>(frame_words);
            int16_t* const o16 = (want & 1u) ? pcm16 : nullptr;
            float* const of = (want & 2u) ? pcmf : nullptr;
            FrameShadow shadow = shadow_in;
            shadow.use = (shadow_in.state && (want & 4u)) ? 1u : 0u;   // MBX_FRAME_WANT_SHADOW: the device copy of the state is current
            if (codec == MBX_CODEC_IMBE7200X4400 || codec == MBX_CODEC_IMBE7100X4400) {
                imbe_stream_body<true, true>(1, 1, record, nullptr, state, rng, o16, of, result, tabs, frame, codec, shadow);
            } else if (codec == MBX_CODEC_AMBE3600X2400) {
                ambe_stream_body<true, true, true>(1, 1, record, nullptr, state, rng, o16, of, result, tabs, frame, shadow);
            } else {
                ambe_stream_body<false, true, true>(1, 1, record, nullptr, state, rng, o16, of, result, tabs, frame, shadow);
            }
            last = in;
            ++served;
            if (lane == 0) {
                __hip_atomic_store(&mb->served, served, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            frame_done(&mb->seq_out, in, lane);   // system-scope fence, then the completion word
            t_last = wall_clock64();
#ifdef MBX_FRAME_STAMPS
            if (lane == 0) {
                g_frame_stamps[9] = t_last;
            }
#endif
            polls = 0u;
            continue;
        }
        const unsigned long long now = wall_clock64();
        const bool quit = __builtin_amdgcn_readlane((int)v, 1) != 0;
        if (quit || (now - t_last) > (unsigned long long)idle_ticks || (now - t_start) > kMaxLifeTicks || ++polls > kMaxIdlePolls) {
            if (lane == 0) {
                __hip_atomic_store(&mb->alive, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// mbe_synthesizeSpeechf for S independent (cur, prev) pairs.

// T = 1: the HBM-slot bodies without a frame loop (see imbe_stream_kernel_one)
#ifndef MBX_AMBE_ONE_WAVES_PER_SIMD
#define MBX_AMBE_ONE_WAVES_PER_SIMD 7   // round 6: SEVEN waves (72 VGPRs, no scratch) once the one-frame AMBE instances stopped requesting the next
#endif                                  // overlap's noise constants ahead of the bank (MBX_AMBE_EARLY_NOISE 0: two registers across the bank) and the
                                        // interpolated branch stopped holding n^2 across its loop: 65,536 x 1 AMBE+2 -1.7 %, resident -2.5 %
                                        // (profiles/r06/ab_ambe_seven_waves.log; round 5 forced seven with 44 B of scratch: +7.8 %)
#ifndef MBX_AMBE_ONE_RES_WAVES_PER_SIMD
#define MBX_AMBE_ONE_RES_WAVES_PER_SIMD 6   // ambe_one_launch_kernel_res alone still needs 12 B of scratch at seven (the transform pair's peak)
#endif
__global__ void __launch_bounds__(64, MBX_AMBE_ONE_WAVES_PER_SIMD)
ambe_stream_kernel_one(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                       mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                       float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    ambe_stream_body<false, false, false, false, true>(S, Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in);
}
__global__ void __launch_bounds__(64, MBX_AMBE2400_WAVES_PER_SIMD)
ambe2400_stream_kernel_one(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                           mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                           float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    ambe_stream_body<true, false, false, false, true>(S, Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in);
}

// ambe_one_launch_kernel / ambe2400_one_launch_kernel: the T = 1 step of the AMBE codecs as ONE launch -- front blocks (the frame's
// FEC by the first lane of its eight, then the eight-lane expansion of mbx_expand_ambe.h) and stream blocks in one grid, exactly as
// imbe_one_launch_kernel (FrontLink; all front blocks first in the grid).
template <bool k2400>
__device__ __forceinline__ void front_block_ambe(int chunk, int S, const uint8_t* __restrict__ frames, mbx_param_record* __restrict__ records,
                                                 FrameParams* __restrict__ rows, uint32_t* __restrict__ flags, uint32_t epoch,
                                                 const DeviceTables& tabs, void* lds) {
    float (*tile)[65] = reinterpret_cast<float (*)[65]>(lds);   // 8 x 65 floats
    const int lane = lane_id();
    const int fi = lane >> 3, sub = lane & 7;
    const int j = 8 * chunk + fi;
    const bool have = j < S;
    const int sj = have ? (tabs.reverse ? (S - 1 - j) : j) : 0;
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (have && sub == 0) {   // two Golay words and a 23-step sequence: one lane per frame (ref src/ambe/ambe_common.c:22-157)
        rec = fec_ambe3600x2450_frame(tabs.t, frames + 9u * (size_t)sj);
        // (the stream block of a tone frame reads its record: written through like the rows, and before the flag)
        uint32_t* const rp = reinterpret_cast<uint32_t*>(&records[sj]);
        __hip_atomic_store(rp + 0, rec.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(rp + 1, rec.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(rp + 2, rec.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(rp + 3, rec.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int src = (lane & ~7) << 2;
    rec.x = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)rec.x);
    rec.y = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)rec.y);
    rec.z = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)rec.z);
    rec.w = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)rec.w);
    xp::expand_ambe_frame_rec<k2400>(have, xp::u32x4{rec.x, rec.y, rec.z, rec.w}, tile[fi], sub, tabs);
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int jq = 8 * chunk + q;
        if (jq < S) {
            const int sq = tabs.reverse ? (S - 1 - jq) : jq;
            __hip_atomic_store(reinterpret_cast<uint32_t*>(&rows[sq].v[0]) + lane, __float_as_uint(tile[q][lane]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        __hip_atomic_store(&flags[chunk], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <bool k2400, bool kRes = false>
__device__ __forceinline__ void ambe_one_launch_body(int S, const uint8_t* __restrict__ frames, mbx_param_record* __restrict__ records,
                                                     FrameParams* __restrict__ rows, uint32_t* __restrict__ flags, uint32_t* __restrict__ fallbacks,
                                                     uint32_t epoch, mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs,
                                                     int16_t* __restrict__ pcm16, float* __restrict__ pcmf, mbe_process_result* __restrict__ results,
                                                     const DeviceTables& tabs_in) {
    const int C = (S + 7) >> 3;
    const int bid = (int)blockIdx.x;   // C front blocks, then 8 C stream blocks
    constexpr size_t kFrontLds = 8 * 65 * 4, kStreamLds = sizeof(WaveScratchT<MBX_PARK_N>);
    __shared__ alignas(16) char lds[kFrontLds > kStreamLds ? kFrontLds : kStreamLds];
    if (bid < C) {
#ifdef MBX_TESTING   // libmbx_hip_testing.so only (mbx_testing_set_front_skip)
        if (tabs_in.front_skip > 0 && (bid & (tabs_in.front_skip - 1)) == 0) {
            return;
        }
#endif
        front_block_ambe<k2400>(bid, S, frames, records, rows, flags, epoch, tabs_in, lds);
        return;
    }
    FrontLink link;
    link.pos = bid - C;
    link.flag = &flags[link.pos >> 3];
    link.epoch = epoch;
    link.lds = lds;
    link.fallbacks = fallbacks;
    ambe_stream_body<k2400, false, false, kRes, true, 3>(S, 1, records, rows, state, rngs, pcm16, pcmf, results, tabs_in, frames, FrameShadow{},
                                                         FrameSlice{}, link);
}
__global__ void __launch_bounds__(64, MBX_AMBE_ONE_WAVES_PER_SIMD)
ambe_one_launch_kernel(int S, const uint8_t* __restrict__ frames, mbx_param_record* __restrict__ records, FrameParams* __restrict__ rows,
                       uint32_t* __restrict__ flags, uint32_t* __restrict__ fallbacks, uint32_t epoch, mbe_parms* __restrict__ state,
                       mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16, float* __restrict__ pcmf,
                       mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    ambe_one_launch_body<false>(S, frames, records, rows, flags, fallbacks, epoch, state, rngs, pcm16, pcmf, results, tabs_in);
}
__global__ void __launch_bounds__(64, MBX_AMBE_ONE_RES_WAVES_PER_SIMD)
ambe_one_launch_kernel_res(int S, const uint8_t* __restrict__ frames, mbx_param_record* __restrict__ records, FrameParams* __restrict__ rows,
                           uint32_t* __restrict__ flags, uint32_t* __restrict__ fallbacks, uint32_t epoch, mbe_parms* __restrict__ state,
                           mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16, float* __restrict__ pcmf,
                           mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    ambe_one_launch_body<false, true>(S, frames, records, rows, flags, fallbacks, epoch, state, rngs, pcm16, pcmf, results, tabs_in);
}
__global__ void __launch_bounds__(64, MBX_AMBE2400_WAVES_PER_SIMD)
ambe2400_one_launch_kernel_res(int S, const uint8_t* __restrict__ frames, mbx_param_record* __restrict__ records, FrameParams* __restrict__ rows,
                               uint32_t* __restrict__ flags, uint32_t* __restrict__ fallbacks, uint32_t epoch, mbe_parms* __restrict__ state,
                               mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16, float* __restrict__ pcmf,
                               mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    ambe_one_launch_body<true, true>(S, frames, records, rows, flags, fallbacks, epoch, state, rngs, pcm16, pcmf, results, tabs_in);
}
__global__ void __launch_bounds__(64, MBX_AMBE2400_WAVES_PER_SIMD)
ambe2400_one_launch_kernel(int S, const uint8_t* __restrict__ frames, mbx_param_record* __restrict__ records, FrameParams* __restrict__ rows,
                           uint32_t* __restrict__ flags, uint32_t* __restrict__ fallbacks, uint32_t epoch, mbe_parms* __restrict__ state,
                           mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16, float* __restrict__ pcmf,
                           mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    ambe_one_launch_body<true>(S, frames, records, rows, flags, fallbacks, epoch, state, rngs, pcm16, pcmf, results, tabs_in);
}

__global__ void __launch_bounds__(64, MBX_AMBE_WAVES_PER_SIMD)
ambe_stream_kernel(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                   mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                   float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    ambe_stream_body<false, false>(S, Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in);
}

__global__ void __launch_bounds__(64, MBX_AMBE2400_WAVES_PER_SIMD)
ambe2400_stream_kernel(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                       mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                       float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    ambe_stream_body<true, false>(S, Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in);
}

// T >= 4: prev_mp / prev_mp_enhanced resident in LDS (see the IMBE kernel)
__global__ void MBX_LDS_KERNEL_ATTR(MBX_AMBE_LDS_WAVES_PER_SIMD)
ambe_stream_kernel_lds(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                       mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                       float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    ambe_stream_body<false, true>(S, Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in);
}

__global__ void MBX_LDS_KERNEL_ATTR(MBX_AMBE_LDS_WAVES_PER_SIMD)
ambe2400_stream_kernel_lds(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                           mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                           float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    ambe_stream_body<true, true>(S, Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in);
}

#ifdef MBX_EXP_PAIR
// EXPERIMENT ("two streams per wave for the transform pair"): what does it cost two streams to share a
// workgroup and meet at MBX_EXP_PAIR barriers per frame -- the rendezvous any pairing of their unvoiced transforms needs -- BEFORE
// any of the pairing's arithmetic is saved?  Two waves per workgroup, wave w decodes stream 2 b + w with the unchanged body.
// (An odd stream count would leave a wave without a partner at the barriers: the experiment runs even S only.)
__global__ void __launch_bounds__(128, MBX_IMBE_LDS_WAVES_PER_SIMD)
imbe_stream_kernel_lds_pairexp(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                               mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                               float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    FrameSlice sl;
    sl.pos = 2 * (int)blockIdx.x + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform: say so)
    imbe_stream_body<true>(S, Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in, nullptr, 0, FrameShadow{}, FrontLink{}, sl);
}
#endif

// ------------------------------------------------------------------------------------------
// Slice kernels (*_stream_kernel_lds_slice): the LDS-resident instances on a SLICE of every stream's frames -- frames t0 ..
// t0 + n - 1 of the `stride` frames a stream has in the batch arrays.  A slice IS a launch of n frames per stream (state in from
// HBM, state out to HBM); the launcher (mbx_api.hip, sliced_launch) cuts a launch whose stream count does not fill the device's
// wave slots evenly into groups of streams x slices of frames and issues them on the caller's stream and two internal HIP streams (three groups), so that the slots one
// group's slice leaves empty are taken by the other group's.
// ------------------------------------------------------------------------------------------
__global__ void MBX_LDS_KERNEL_ATTR(MBX_IMBE_LDS_WAVES_PER_SIMD)
imbe_stream_kernel_lds_slice(int S, int stride, int t0, int n, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                             mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                             float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    FrameSlice sl;
    sl.stride = stride;
    sl.t0 = t0;
    imbe_stream_body<true>(S, n, records, params, state, rngs, pcm16, pcmf, results, tabs_in, nullptr, 0, FrameShadow{}, FrontLink{}, sl);
}
__global__ void MBX_LDS_KERNEL_ATTR(MBX_AMBE_LDS_WAVES_PER_SIMD)
ambe_stream_kernel_lds_slice(int S, int stride, int t0, int n, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                             mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                             float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    FrameSlice sl;
    sl.stride = stride;
    sl.t0 = t0;
    ambe_stream_body<false, true>(S, n, records, params, state, rngs, pcm16, pcmf, results, tabs_in, nullptr, FrameShadow{}, sl);
}
__global__ void MBX_LDS_KERNEL_ATTR(MBX_AMBE_LDS_WAVES_PER_SIMD)
ambe2400_stream_kernel_lds_slice(int S, int stride, int t0, int n, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                                 mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                                 float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    FrameSlice sl;
    sl.stride = stride;
    sl.t0 = t0;
    ambe_stream_body<true, true>(S, n, records, params, state, rngs, pcm16, pcmf, results, tabs_in, nullptr, FrameShadow{}, sl);
}

// Resident state (DeviceTables::resident, mbx_process_batch_resident): the LDS-resident bodies with prev_mp_enhanced elided
// while it equals cur_mp and prev_mp fetched lazily -- at every T, T = 1 included (sessions, queue mode).
#ifndef MBX_IMBE_RES_WAVES_PER_SIMD
#define MBX_IMBE_RES_WAVES_PER_SIMD MBX_IMBE_LDS_WAVES_PER_SIMD
#endif
__global__ void MBX_LDS_KERNEL_ATTR(MBX_IMBE_RES_WAVES_PER_SIMD)
imbe_stream_kernel_res(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                       mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                       float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    imbe_stream_body<true, false, true>(S, Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in);
}

// one frame per stream on resident state: the HBM-slot body (seven waves per SIMD, no LDS copy of prev_mp)
__global__ void __launch_bounds__(64, MBX_STREAM_WAVES_PER_SIMD)
imbe_stream_kernel_res1(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                        mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                        float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    imbe_stream_body<false, false, true, true>(S, Tn > 1 ? 1 : Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in);
}

// one frame per stream on resident state: the HBM-slot bodies with the elision (see imbe_stream_kernel_res1)
__global__ void __launch_bounds__(64, MBX_AMBE_ONE_WAVES_PER_SIMD)
ambe_stream_kernel_res1(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                        mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                        float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    ambe_stream_body<false, false, false, true, true>(S, Tn > 1 ? 1 : Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in);
}
__global__ void __launch_bounds__(64, MBX_AMBE2400_WAVES_PER_SIMD)
ambe2400_stream_kernel_res1(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                            mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                            float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    ambe_stream_body<true, false, false, true, true>(S, Tn > 1 ? 1 : Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in);
}

__global__ void MBX_LDS_KERNEL_ATTR(MBX_AMBE_LDS_WAVES_PER_SIMD)
ambe_stream_kernel_res(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                       mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                       float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    ambe_stream_body<false, true, false, true>(S, Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in);
}

__global__ void MBX_LDS_KERNEL_ATTR(MBX_AMBE_LDS_WAVES_PER_SIMD)
ambe2400_stream_kernel_res(int S, int Tn, const mbx_param_record* __restrict__ records, const FrameParams* __restrict__ params,
                           mbe_parms* __restrict__ state, mbx_stream_rng* __restrict__ rngs, int16_t* __restrict__ pcm16,
                           float* __restrict__ pcmf, mbe_process_result* __restrict__ results, DeviceTables tabs_in) {
    ambe_stream_body<true, true, false, true>(S, Tn, records, params, state, rngs, pcm16, pcmf, results, tabs_in);
}

__global__ void __launch_bounds__(64)
synth_speech_kernel(int S, mbe_parms* __restrict__ curs, mbe_parms* __restrict__ prevs, mbx_stream_rng* __restrict__ rngs,
                    float* __restrict__ pcmf, int16_t* __restrict__ pcm16, DeviceTables tabs) {
    __shared__ WaveScratch scratch;
    const int s = blockIdx.x;
    if (s >= S) {
        return;
    }
    const int lane = lane_id();
    Parms cur, prev;
    load_parms(cur, &curs[s], lane);
    load_parms(prev, &prevs[s], lane);
    StreamRng rng;
    load_rng(rng, &rngs[s]);
    float out[3];
    synth_core<false>(out, cur, prev, false, 0.0f, rng, scratch, tabs, lane);
    store_pcm(out, (size_t)s, pcm16, pcmf, lane);
    store_parms(cur, &curs[s], lane);
    store_parms(prev, &prevs[s], lane);
    store_rng(rng, &rngs[s], lane);
}


// mbe_synthesizeTonef (ids == nullptr: AMBE+2 tone from the 49 parameter bits of the record, ref src/core/mbelib.c:745-804)
// / mbe_synthesizeTonefdstar (ids[s] = D-STAR tone index, ref :813-856), batched: one wavefront per struct
__global__ void __launch_bounds__(64)
tone_kernel(int S, const mbx_param_record* __restrict__ records, const int32_t* __restrict__ ids, mbe_parms* __restrict__ curs,
            float* __restrict__ pcmf, int16_t* __restrict__ pcm16, int tones_off) {
    const int s = blockIdx.x;
    if (s >= S) {
        return;
    }
    const int lane = lane_id();
    Parms cur;
    load_parms(cur, &curs[s], lane);
    float out[3];
    if (ids) {
        tone_dstar_frame(out, ids[s], cur, lane, tones_off);
    } else {
        const uint4 rec = *reinterpret_cast<const uint4*>(&records[s]);
        const uint32_t w[3] = {rec.x, rec.y, rec.z};
        tone_frame(out, w, cur, lane, tones_off);
    }
    store_pcm(out, (size_t)s, pcm16, pcmf, lane);
    store_parms(cur, &curs[s], lane);
}

// ------------------------------------------------------------------------------------------
// Single-stage entry points of the public API, batched: one wavefront per struct.
//   mbe_spectralAmpEnhance       ref src/core/mbelib.c:663-666
//   mbe_applyAdaptiveSmoothing   ref src/core/mbe_adaptive.c:268-276
//   mbe_synthesizeComfortNoisef  ref src/core/mbe_adaptive.c:116-131
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
enhance_kernel(int S, mbe_parms* __restrict__ parms) {
    const int s = blockIdx.x;
    if (s >= S) {
        return;
    }
    const int lane = lane_id();
    __shared__ alignas(16) float tmp[64];
    Parms cur;
    load_parms(cur, &parms[s], lane);
    (void)enhance(cur, lane, tmp);
    store_parms(cur, &parms[s], lane);
}

__global__ void __launch_bounds__(64)
smoothing_kernel(int S, mbe_parms* __restrict__ curs, const mbe_parms* __restrict__ prevs) {
    const int s = blockIdx.x;
    if (s >= S) {
        return;
    }
    const int lane = lane_id();
    Parms cur, prev;
    load_parms(cur, &curs[s], lane);
    load_parms(prev, &prevs[s], lane);
    if (cur.L >= 1 && cur.L <= 56 && prev.L >= 1 && prev.L <= 56) {
        const bool in = lane >= 1 && lane <= cur.L;
        __shared__ alignas(16) float tmp[64];
        const float rm0 = seq_sum4(in ? (cur.Ml * cur.Ml) : 0.0f, cur.L, tmp, lane);
        smooth(cur, prev, rm0, lane, nullptr, tmp);
        store_parms(cur, &curs[s], lane);
    }
}

__global__ void __launch_bounds__(64)
comfort_noise_kernel(int S, mbx_stream_rng* __restrict__ rngs, float* __restrict__ pcmf, int16_t* __restrict__ pcm16) {
    const int s = blockIdx.x;
    if (s >= S) {
        return;
    }
    const int lane = lane_id();
    StreamRng rng;
    load_rng(rng, &rngs[s]);
    float out[3];
    comfort_noise(out, rng, lane);
    store_pcm(out, (size_t)s, pcm16, pcmf, lane);
    store_rng(rng, &rngs[s], lane);
}

// mbe_decodeImbe4400Parms / mbe_decodeAmbe2450Parms / mbe_decodeAmbe2400Parms, batched: the stateful half of the parameter
// decode (log-magnitude prediction from prev_mp) WITHOUT policy or synthesis, on FrameParams rows made by the expand
// kernels.  One wavefront per (cur_mp, prev_mp) pair; both structs are updated the way the reference updates them (the
// prediction memory of prev_mp is padded in place) and rc[i] is the reference's return value (0 voice, 1 invalid IMBE
// fundamental, 2 erasure, 7 / tone index for tone frames).
//   ref src/imbe/imbe7200x4400.c:589-630, src/ambe/ambe3600x2450.c:555-634, src/ambe/ambe3600x2400.c:427-561
__global__ void __launch_bounds__(64)
decode_parms_kernel(int codec, int n, const FrameParams* __restrict__ params, mbe_parms* __restrict__ curs,
                    mbe_parms* __restrict__ prevs, int32_t* __restrict__ rc, DeviceTables tabs) {
    __shared__ float fp[64];
    __shared__ alignas(16) float tmp[64];
    const int i = blockIdx.x;
    if (i >= n) {
        return;
    }
    const int lane = lane_id();
    Parms cur, prev;
    load_parms(cur, &curs[i], lane);
    load_parms(prev, &prevs[i], lane);
    fp[lane] = params[i].v[lane];
    wave_lds_sync();
    const int bad = (codec == MBX_CODEC_IMBE7200X4400) ? decode_imbe(fp, cur, prev, tabs.d, lane, tmp) : decode_ambe(fp, cur, prev, tabs, lane, tmp);
    store_parms(cur, &curs[i], lane);
    store_parms(prev, &prevs[i], lane);
    if (lane == 0) {
        rc[i] = bad;
    }
}

// The state-I/O floor of the stream kernels: load the three structs and store them back.
// Used by bench.py (--calibrate) to price the HBM traffic of the access pattern and to calibrate
// the FETCH_SIZE / WRITE_SIZE counters on a known byte count.
__global__ void __launch_bounds__(64)
state_copy_kernel(int S, mbe_parms* __restrict__ state) {
    const int s = blockIdx.x;
    if (s >= S) {
        return;
    }
    const int lane = lane_id();
    Parms a, b, c;
    load_parms(a, &state[3 * (size_t)s + 0], lane);
    load_parms(b, &state[3 * (size_t)s + 1], lane);
    load_parms(c, &state[3 * (size_t)s + 2], lane);
    // (non-temporal stores, like the one-frame stream instances this kernel is the no-arithmetic floor of)
    store_parms<false, true>(a, &state[3 * (size_t)s + 0], lane);
    store_parms<false, true>(b, &state[3 * (size_t)s + 1], lane);
    store_parms<false, true>(c, &state[3 * (size_t)s + 2], lane);
}

}  // namespace mbx

#ifdef MBX_FRAME_STAMPS
// development builds only (tools/frame_stamps.py): the stamps of the last single-frame call, in 10 ns ticks
extern "C" int mbx_debug_frame_stamps(unsigned long long* out16) {
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(mbx::g_frame_stamps), 16 * sizeof(unsigned long long));
}
#endif

#ifdef MBX_STAGE_TIMES
// development builds only (tools/stage_times.py): the marks of the LAST launch, 16 per wave, in 10 ns ticks
extern "C" int mbx_debug_stage_times(unsigned* out, int waves) {
    if (waves > mbx::kStageWaves) {
        waves = mbx::kStageWaves;
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mbx::g_stage_buf), (size_t)waves * 16 * sizeof(unsigned));
}
#endif
