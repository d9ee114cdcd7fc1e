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
    float2   twiddle[256];    // exp(-2*pi*i*k/256)
    float    log2_int[64];    // log2f((float)L) from the host libm (AMBE gain term)
    uint8_t  imbe_inv_bo[48][58][12];   // inverse of imbe_bo: [L-9][word][bit] -> payload bit (255 = none)
};

// Output of the expand stage, input of the stream stage: 64 dwords per frame (layout in mbx_expand.hip).
struct FrameParams {
    float v[64];
};

struct DeviceTables {
    const mbx_tables*    t;
    const DerivedTables* d;
    int                  ablate;   // timing-only stage mask (mbx_debug_set_ablation); 0 in normal use
};

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }

// One wavefront per workgroup: the barrier only orders this wave's LDS traffic.
__device__ __forceinline__ void wave_lds_sync() { __syncthreads(); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        v += __shfl_xor(v, m, kWave);
    }
    return v;
}

__device__ __forceinline__ float lane_get(float v, int src) { return __shfl(v, src, kWave); }
__device__ __forceinline__ int lane_get(int v, int src) { return __shfl(v, src, kWave); }

__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }

// base-4 digit reversal of an 8-bit index (radix-4 FFT ordering)
__device__ __forceinline__ int rev4(int p) {
    return ((p & 3) << 6) | ((p & 12) << 2) | ((p & 48) >> 2) | ((p & 192) >> 6);
}

}  // namespace mbx
