// copy_patterns.hip -- how fast can one wave per stream move the 7,812-byte state triplet (3 x mbe_parms) in and out of HBM?
// The stream kernels read and write it with one dword per lane (the reference's struct layout is only 4-byte aligned);
// this program times that pattern next to 16-byte-per-lane accesses at the same (unaligned) addresses, 65,536 streams
// (512 MB each way, larger than the Infinity Cache), to see whether the access width is what separates the kernels'
// 5.3 TB/s from the 6.3 TB/s the guide calls achievable.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/copy_patterns tools/copy_patterns.hip && tools/bin/copy_patterns
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kDwords = 1953;   // 3 x 651

__global__ void __launch_bounds__(64) copy_dword(int S, float* state, float bias) {
    const int s = blockIdx.x;
    if (s >= S) return;
    float* p = state + (size_t)s * kDwords;
    const int lane = threadIdx.x;
    float v[31];
#pragma unroll
    for (int i = 0; i < 31; ++i) {
        const int k = lane + 64 * i;
        v[i] = k < kDwords ? p[k] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 31; ++i) {
        const int k = lane + 64 * i;
        if (k < kDwords) p[k] = v[i] + bias;
    }
}

// the same with NON-TEMPORAL stores (what the one-frame stream instances use since round 5)
__global__ void __launch_bounds__(64) copy_dword_nt(int S, float* state, float bias) {
    const int s = blockIdx.x;
    if (s >= S) return;
    float* p = state + (size_t)s * kDwords;
    const int lane = threadIdx.x;
    float v[31];
#pragma unroll
    for (int i = 0; i < 31; ++i) {
        const int k = lane + 64 * i;
        v[i] = k < kDwords ? p[k] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 31; ++i) {
        const int k = lane + 64 * i;
        if (k < kDwords) __builtin_nontemporal_store(v[i] + bias, &p[k]);
    }
}

typedef float f4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) U4 { f4 v; };   // 16-byte access at a 4-byte-aligned address

__global__ void __launch_bounds__(64) copy_x4(int S, float* state, float bias) {
    const int s = blockIdx.x;
    if (s >= S) return;
    float* p = state + (size_t)s * kDwords;
    const int lane = threadIdx.x;
    f4 v[8];
    float tail = 0.0f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {   // 7 x 1 KB = 1,792 dwords
        v[i] = reinterpret_cast<const U4*>(p + 4 * (lane + 64 * i))->v;
    }
    const int t0 = 1792;            // 161 dwords left: three dword rounds
    float tl[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int k = t0 + lane + 64 * i;
        tl[i] = k < kDwords ? p[k] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        U4 o;
        o.v = v[i] + bias;
        *reinterpret_cast<U4*>(p + 4 * (lane + 64 * i)) = o;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int k = t0 + lane + 64 * i;
        if (k < kDwords) p[k] = tl[i] + bias;
    }
    (void)tail;
}

// the same bytes with NO per-stream structure: a plain streaming copy, 16 B per lane, aligned (the ceiling)
__global__ void __launch_bounds__(256) copy_flat(size_t n4, f4* p, float bias) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) p[i] = p[i] + bias;
}

__global__ void __launch_bounds__(256) copy_flat_nt(size_t n4, f4* p, float bias) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) __builtin_nontemporal_store(p[i] + bias, &p[i]);
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 20; ++i) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 20;
}

int main() {
    const int S = 65536;
    const size_t bytes = (size_t)S * kDwords * 4;
    float* d;
    CHECK(hipMalloc(&d, bytes + 64));
    CHECK(hipMemset(d, 0, bytes));
    const double t1 = time_ms([&] { hipLaunchKernelGGL(copy_dword, dim3(S), dim3(64), 0, 0, S, d, 1.0f); });
    const double t2 = time_ms([&] { hipLaunchKernelGGL(copy_x4, dim3(S), dim3(64), 0, 0, S, d, 1.0f); });
    const size_t n4 = bytes / 16;
    const double t3 = time_ms([&] { hipLaunchKernelGGL(copy_flat, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, n4, (f4*)d, 1.0f); });
    const double t4 = time_ms([&] { hipLaunchKernelGGL(copy_dword_nt, dim3(S), dim3(64), 0, 0, S, d, 1.0f); });
    const double t5 = time_ms([&] { hipLaunchKernelGGL(copy_flat_nt, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, n4, (f4*)d, 1.0f); });
    printf("{\"dword_per_lane_nt_stores_ms\": %.4f, \"dword_per_lane_nt_stores_TBps\": %.3f, \"flat_aligned_x4_nt_stores_ms\": %.4f, \"flat_aligned_x4_nt_stores_TBps\": %.3f}\n",
           t4, 2 * bytes / t4 / 1e9, t5, 2 * bytes / t5 / 1e9);
    printf("{\"bytes_each_way\": %zu, \"dword_per_lane_ms\": %.4f, \"dword_per_lane_TBps\": %.3f, \"x4_per_lane_unaligned_ms\": %.4f, \"x4_per_lane_unaligned_TBps\": %.3f, "
           "\"flat_aligned_x4_ms\": %.4f, \"flat_aligned_x4_TBps\": %.3f}\n",
           bytes, t1, 2 * bytes / t1 / 1e9, t2, 2 * bytes / t2 / 1e9, t3, 2 * bytes / t3 / 1e9);
    return 0;
}
