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
// (AMBE rows: v[60] = 0.2046 / sqrt(w0), L in bits 8.. of v[63]),
// v[61] K (IMBE) / mean residual Sum42 (AMBE), v[62] error-context word, v[63] frame class
// (0 voice, 1 invalid IMBE fundamental, 2 AMBE+2 erasure, 7 AMBE+2 tone; D-STAR: 3 tone class without a usable index,
// 5..122 tone index), v[0] AMBE gain increment.
#include "mbx_device.h"
#include "mbx_expand_ambe.h"
#include "mbx_expand_imbe.h"

namespace mbx {

constexpr int kFramesPerWave = 8;
constexpr int kWavesPerBlock = 4;    // 256-thread workgroups: a quarter of the workgroup launches of single-wave blocks
constexpr int kFramesPerBlock = kFramesPerWave * kWavesPerBlock;
constexpr int kRow = 65;   // 64 dwords + 1 pad

using xp::rbit;

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
    const size_t first = (size_t)blockIdx.x * kFramesPerBlock;
    const int fi = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const size_t i = first + fi;
    float* row = tile[fi];
    const uint4 rec = (i < n) ? *reinterpret_cast<const uint4*>(&recs[i]) : make_uint4(0u, 0u, 0u, 0u);
    xp::expand_imbe_frame_rec(i < n, rec, row, words[fi], gains[fi], sub, tabs);
    write_out(tile, out, first, n);
}

// k2400: AMBE 3600x2400 (D-STAR, ref src/ambe/ambe3600x2400.c:164-425): other bit positions, codebooks and frame
// classes (0 voice, 3 tone class without a usable index, 5..122 tone index); the arithmetic is the same.  The per-frame
// work is expand_ambe_frame (mbx_expand_ambe.h), shared with the LDS-resident stream kernels.
template <bool k2400>
__device__ __forceinline__ void expand_ambe_body(const mbx_param_record* __restrict__ recs, size_t n, FrameParams* __restrict__ out,
                                                 const DeviceTables& tabs) {
    __shared__ float tile[kFramesPerBlock][kRow];
    const size_t first = (size_t)blockIdx.x * kFramesPerBlock;
    const int fi = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const size_t i = first + fi;
    xp::expand_ambe_frame<k2400>(i < n, &recs[i < n ? i : 0], tile[fi], sub, tabs);
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
