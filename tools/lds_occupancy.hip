// development aid (GPU box): how many one-wave workgroups are resident per CU for a given LDS size per workgroup?
// (the occupancy API prices LDS against 64 KB on this stack; this measures it).  Every wave counts itself in,
// spins, counts itself out; the high-water mark / CU count is the residency.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/lds_occupancy tools/lds_occupancy.hip && tools/bin/lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(64) probe(int* now, int* high, long long spin_ticks) {
    extern __shared__ char lds[];
    lds[threadIdx.x] = (char)threadIdx.x;
    if (threadIdx.x == 0) {
        const int n = atomicAdd(now, 1) + 1;
        atomicMax(high, n);
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) {
        __builtin_amdgcn_s_sleep(8);
    }
    if (threadIdx.x == 0) {
        atomicSub(now, 1);
    }
    if (lds[threadIdx.x] == 77 && spin_ticks < 0) {
        high[1] = 1;
    }
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    int* d;
    hipMalloc(&d, 16);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("CUs %d, LDS per CU %zu\n", cus, (size_t)prop.maxSharedMemoryPerMultiProcessor);
    for (int bytes = 3072; bytes <= 8192; bytes += 128) {
        hipMemset(d, 0, 16);
        probe<<<cus * 48, 64, bytes>>>(d, d + 1, 20000000LL / 10);   // 100 MHz clock: 20 ms
        hipDeviceSynchronize();
        int h[2];
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("LDS %5d B/wave: %.2f waves per CU\n", bytes, (double)h[1] / cus);
    }
    return 0;
}
