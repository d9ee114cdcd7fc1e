// development aid (GPU box): how long does the GPU take just to START and retire 65,536 one-wave workgroups with the
// stream kernel's resources (LDS per workgroup, VGPR allocation)?  An empty kernel per configuration.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/dispatch_rate tools/dispatch_rate.hip && tools/bin/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int kLds>
__device__ __forceinline__ void body(float* out) {
    __shared__ float lds[kLds / 4];
    lds[threadIdx.x] = (float)threadIdx.x;
    if (lds[(threadIdx.x + 1) & 63] < -1.0f) {
        out[blockIdx.x] = 1.0f;   // never
    }
}
__global__ void __launch_bounds__(64) k_small(float* out) { body<256>(out); }
__global__ void __launch_bounds__(64) k_lds(float* out) { body<4624>(out); }
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 7))) k_lds_w7(float* out) { body<4624>(out); }
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_lds_w4(float* out) { body<4624>(out); }
__global__ void __launch_bounds__(256) k_256(float* out) { body<4624>(out); }

template <typename K>
static void run(const char* name, K k, int grid, int block, float* d) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d);
    hipEventRecord(a, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s grid %6d x %3d: %7.2f us per launch\n", name, grid, block, ms * 1000.0f / 20.0f);
}

int main() {
    float* d;
    hipMalloc(&d, 65536 * 4);
    run("tiny (256 B LDS)", k_small, 65536, 64, d);
    run("4624 B LDS", k_lds, 65536, 64, d);
    run("4624 B LDS, 7 waves/SIMD regs", k_lds_w7, 65536, 64, d);
    run("4624 B LDS, 4 waves/SIMD regs", k_lds_w4, 65536, 64, d);
    run("4624 B LDS", k_lds, 7168, 64, d);
    run("256-thread groups", k_256, 16384, 256, d);
    return 0;
}
