// copy_patterns2.hip -- round 6, VERDICT r5 item 3: does fetching / storing the state structs as ALIGNED 16-byte spans through an LDS
// hop beat the dword-per-lane accesses of the stream kernels, with ONE WAVE PER STREAM (the structure the kernels have)?
// profiles/r05/copy_patterns.json compared the dword-per-lane pattern with a structureless flat copy (5.43 vs 5.87 TB/s); that leaves open
// whether the 8 % belong to the access width / alignment or to the flat kernel's shape.  Variants, all 65,536 streams x 7,812 B each way,
// non-temporal stores, lane = element layout in registers between load and store (what the stream kernels compute in):
//   k0  dword per lane in, dword per lane out                                   (the product's pattern; copy_patterns.hip copy_dword_nt)
//   k1  dword per lane in; out per STRUCT: registers -> LDS -> aligned dwordx4 stores (+ one masked dword store for the <= 6 edge dwords)
//   k2  in per STRUCT: aligned dwordx4 loads (+ edge dwords) -> LDS -> registers; dword per lane out
//   k3  k2's loads + k1's stores
//   k4  k3 with the whole 7,812-byte triplet as ONE span (8 KB of LDS per wave)
//   k5  round 5's "dword per lane" (the triplet as one run, 31 dwords per lane)      } the two ends of profiles/r05/copy_patterns.json,
//   k6  round 5's flat copy: no per-stream structure, 16 B per lane, aligned         } timed HERE like everything else
//   k7  exactly the one-frame stream instances' accesses: gathered header + 57-dword band arrays + previousUw + overlap, per struct
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/copy_patterns2 tools/copy_patterns2.hip && tools/bin/copy_patterns2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kStruct = 651, kTriplet = 3 * kStruct;
typedef float f4 __attribute__((ext_vector_type(4)));

// a span of N dwords at global dword address g <-> LDS buffer l (16-byte aligned, N + 8 dwords): element k lives at l[phase + k], so that
// 16-byte units of the global span are 16-byte units of the buffer
template <int N>
__device__ __forceinline__ void span_to_lds(const float* __restrict__ g, float* l, int lane) {
    const int phase = (int)((reinterpret_cast<uintptr_t>(g) >> 2) & 3), head = (4 - phase) & 3;
    const int units = (N - head) >> 2, tail = (N - head) & 3;
    constexpr int kRounds = (N / 4 + 63) / 64;
    f4 v[kRounds];
#pragma unroll
    for (int j = 0; j < kRounds; ++j) {
        const int u = lane + 64 * j;
        v[j] = (u < units) ? *reinterpret_cast<const f4*>(g + head + 4 * u) : f4{0, 0, 0, 0};
    }
    float e = 0.0f;
    const int ek = (lane < head) ? lane : (N - tail + (lane - head));
    const bool edge = lane < head + tail;
    if (edge) e = g[ek];
#pragma unroll
    for (int j = 0; j < kRounds; ++j) {
        const int u = lane + 64 * j;
        if (u < units) *reinterpret_cast<f4*>(l + phase + head + 4 * u) = v[j];
    }
    if (edge) l[phase + ek] = e;
}
template <int N, bool kNt = true>
__device__ __forceinline__ void lds_to_span(float* __restrict__ g, const float* l, int lane) {
    const int phase = (int)((reinterpret_cast<uintptr_t>(g) >> 2) & 3), head = (4 - phase) & 3;
    const int units = (N - head) >> 2, tail = (N - head) & 3;
    constexpr int kRounds = (N / 4 + 63) / 64;
#pragma unroll
    for (int j = 0; j < kRounds; ++j) {
        const int u = lane + 64 * j;
        if (u < units) {
            const f4 v = *reinterpret_cast<const f4*>(l + phase + head + 4 * u);
            __builtin_nontemporal_store(v, reinterpret_cast<f4*>(g + head + 4 * u));
        }
    }
    const int ek = (lane < head) ? lane : (N - tail + (lane - head));
    if (lane < head + tail) __builtin_nontemporal_store(l[phase + ek], g + ek);
}

template <int N>   // registers (lane = element, ceil(N / 64) per lane) <-> LDS span buffer
__device__ __forceinline__ void regs_from_lds(float* v, const float* l, int phase, int lane) {
#pragma unroll
    for (int i = 0; i < (N + 63) / 64; ++i) {
        const int k = lane + 64 * i;
        v[i] = k < N ? l[phase + k] : 0.0f;
    }
}
template <int N>
__device__ __forceinline__ void regs_to_lds(const float* v, float* l, int phase, int lane) {
#pragma unroll
    for (int i = 0; i < (N + 63) / 64; ++i) {
        const int k = lane + 64 * i;
        if (k < N) l[phase + k] = v[i];
    }
}

template <bool kX4Loads, bool kX4Stores>
__global__ void __launch_bounds__(64) copy_struct_spans(int S, float* state, float bias) {
    __shared__ alignas(16) float lds[kStruct + 8];
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= S) return;
    float* p = state + (size_t)s * kTriplet;
    float v[3][11];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        float* g = p + q * kStruct;
        if constexpr (kX4Loads) {
            const int phase = (int)((reinterpret_cast<uintptr_t>(g) >> 2) & 3);
            span_to_lds<kStruct>(g, lds, lane);
            regs_from_lds<kStruct>(v[q], lds, phase, lane);
        } else {
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                const int k = lane + 64 * i;
                v[q][i] = k < kStruct ? g[k] : 0.0f;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        float* g = p + q * kStruct;
#pragma unroll
        for (int i = 0; i < 11; ++i) v[q][i] += bias;
        if constexpr (kX4Stores) {
            const int phase = (int)((reinterpret_cast<uintptr_t>(g) >> 2) & 3);
            regs_to_lds<kStruct>(v[q], lds, phase, lane);
            lds_to_span<kStruct>(g, lds, lane);
        } else {
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                const int k = lane + 64 * i;
                if (k < kStruct) __builtin_nontemporal_store(v[q][i], g + k);
            }
        }
    }
}

__global__ void __launch_bounds__(64) copy_triplet_span(int S, float* state, float bias) {
    __shared__ alignas(16) float lds[kTriplet + 8];
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= S) return;
    float* g = state + (size_t)s * kTriplet;
    const int phase = (int)((reinterpret_cast<uintptr_t>(g) >> 2) & 3);
    float v[31];
    span_to_lds<kTriplet>(g, lds, lane);
    regs_from_lds<kTriplet>(v, lds, phase, lane);
#pragma unroll
    for (int i = 0; i < 31; ++i) v[i] += bias;
    regs_to_lds<kTriplet>(v, lds, phase, lane);
    lds_to_span<kTriplet>(g, lds, lane);
}

// the two patterns of copy_patterns.hip (round 5) in THIS harness: the triplet as one run of 31 dwords per lane, and the flat copy
__global__ void __launch_bounds__(64) copy_triplet_dword(int S, float* state, float bias) {
    const int s = blockIdx.x;
    if (s >= S) return;
    float* p = state + (size_t)s * kTriplet;
    const int lane = threadIdx.x;
    float v[31];
#pragma unroll
    for (int i = 0; i < 31; ++i) {
        const int k = lane + 64 * i;
        v[i] = k < kTriplet ? p[k] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 31; ++i) {
        const int k = lane + 64 * i;
        if (k < kTriplet) __builtin_nontemporal_store(v[i] + bias, &p[k]);
    }
}
__global__ void __launch_bounds__(256) copy_flat_x4(size_t n4, f4* p, float bias) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) __builtin_nontemporal_store(p[i] + bias, &p[i]);
}

// k7: EXACTLY the one-frame stream instances' accesses (mbx_stream.hip load_header / load_parms_arrays<kFlat> / store_parms<kGather, kNt>):
// per struct one gathered header load (lanes 0..13: dwords 0, 1, 2, 288..297, 554), five 57-dword band arrays, previousUw in four
// rows, the noise overlap in a row and a half -- twelve loads and twelve (masked, non-temporal) stores per struct
__global__ void __launch_bounds__(64) copy_fieldwise(int S, float* state, float bias) {
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= S) return;
    const int j = lane & 15;
    const int hidx = (j < 3) ? j : ((j < 13) ? 285 + j : 554);
    float h[3], band[3][5], uw[3][4], ov0[3], ov1[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float* p = state + (size_t)s * kTriplet + q * kStruct;
        h[q] = p[hidx];
#pragma unroll
        for (int a = 0; a < 5; ++a) band[q][a] = p[3 + 57 * a + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) uw[q][r] = p[298 + lane + 64 * r];
        ov0[q] = p[555 + lane];
        ov1[q] = p[619 + (lane & 31)];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        float* p = state + (size_t)s * kTriplet + q * kStruct;
        if (lane < 14) __builtin_nontemporal_store(h[q] + bias, &p[hidx]);
        if (lane < 57) {
#pragma unroll
            for (int a = 0; a < 5; ++a) __builtin_nontemporal_store(band[q][a] + bias, &p[3 + 57 * a + lane]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(uw[q][r] + bias, &p[298 + lane + 64 * r]);
        __builtin_nontemporal_store(ov0[q] + bias, &p[555 + lane]);
        if (lane < 32) __builtin_nontemporal_store(ov1[q] + bias, &p[619 + lane]);
    }
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) launch();
    double best = 1e9, sum = 0;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(a));
        for (int i = 0; i < 40; ++i) launch();
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        sum += ms / 40;
        if (ms / 40 < best) best = ms / 40;
    }
    return sum / 5;
}

int main() {
    const int S = 65536;
    const size_t n = (size_t)S * kTriplet, bytes = n * 4;
    float* d;
    CHECK(hipMalloc(&d, bytes + 64));
    // correctness first: every variant must add exactly `bias` to every dword and touch nothing else
    float* h = (float*)malloc(bytes + 64);
    const char* names[8] = {"k0_dword_in_dword_out", "k1_dword_in_x4_out", "k2_x4_in_dword_out", "k3_x4_in_x4_out", "k4_triplet_span",
                            "k5_triplet_dword_31_per_lane", "k6_flat_aligned_x4", "k7_fieldwise_as_the_one_frame_kernels"};
    const size_t n4 = (size_t)S * kTriplet / 4;
    auto launch = [&](int k, float bias) {
        switch (k) {
            case 0: hipLaunchKernelGGL((copy_struct_spans<false, false>), dim3(S), dim3(64), 0, 0, S, d, bias); break;
            case 1: hipLaunchKernelGGL((copy_struct_spans<false, true>), dim3(S), dim3(64), 0, 0, S, d, bias); break;
            case 2: hipLaunchKernelGGL((copy_struct_spans<true, false>), dim3(S), dim3(64), 0, 0, S, d, bias); break;
            case 3: hipLaunchKernelGGL((copy_struct_spans<true, true>), dim3(S), dim3(64), 0, 0, S, d, bias); break;
            case 4: hipLaunchKernelGGL(copy_triplet_span, dim3(S), dim3(64), 0, 0, S, d, bias); break;
            case 5: hipLaunchKernelGGL(copy_triplet_dword, dim3(S), dim3(64), 0, 0, S, d, bias); break;
            case 6: hipLaunchKernelGGL(copy_flat_x4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, n4, (f4*)d, bias); break;
            default: hipLaunchKernelGGL(copy_fieldwise, dim3(S), dim3(64), 0, 0, S, d, bias); break;
        }
    };
    for (int k = 0; k < 8; ++k) {
        for (size_t i = 0; i < n + 16; ++i) h[i] = (float)(i % 4093);
        CHECK(hipMemcpy(d, h, bytes + 64, hipMemcpyHostToDevice));
        launch(k, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h, d, bytes + 64, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < n + 16; ++i) {
            const float want = (float)(i % 4093) + (i < n ? 1.0f : 0.0f);
            if (h[i] != want) ++bad;
        }
        if (bad) {
            fprintf(stderr, "%s: %zu wrong dwords\n", names[k], bad);
            return 1;
        }
    }
    CHECK(hipMemset(d, 0, bytes));
    printf("{\"bytes_each_way\": %zu", bytes);
    for (int round = 0; round < 2; ++round) {   // twice, interleaved: box drift shows as a difference between the rounds
        for (int k = 0; k < 8; ++k) {
            const double t = time_ms([&] { launch(k, 0.0f); });
            printf(", \"%s_ms_%d\": %.4f, \"%s_TBps_%d\": %.3f", names[k], round, t, names[k], round, 2 * bytes / t / 1e9);
        }
    }
    printf("}\n");
    return 0;
}
