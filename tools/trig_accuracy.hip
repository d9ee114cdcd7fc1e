// development aid: accuracy of candidate cos/sin evaluations on gfx950 against double precision
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

__device__ inline float cos_hw(float x) {   // double-precision range reduction + v_cos_f32
    const double rev = (double)x * 0.15915494309189533577;   // 1/(2*pi)
    const float fr = (float)(rev - floor(rev));
    return __builtin_amdgcn_cosf(fr);
}
__device__ inline float sin_hw(float x) {
    const double rev = (double)x * 0.15915494309189533577;
    const float fr = (float)(rev - floor(rev));
    return __builtin_amdgcn_sinf(fr);
}
// Cody-Waite to [-pi/4, pi/4] + minimax polynomials (float only)
__device__ inline void sincos_cw(float x, float* s, float* c) {
    const float k = rintf(x * 0.63661977236758134308f);   // 2/pi
    float r = fmaf(-k, 1.5703125f, x);                     // pi/2 split: high part has 7 mantissa bits
    r = fmaf(-k, 4.83751296997e-4f, r);
    r = fmaf(-k, 7.5497899549e-8f, r);
    r = fmaf(-k, 2.5579538487363607e-12f, r);
    const float r2 = r * r;
    float sp = fmaf(r2, 2.7183114939898219064e-6f, -1.9839334836096632576e-4f);
    sp = fmaf(sp, r2, 8.3333293858894631756e-3f);
    sp = fmaf(sp, r2, -1.6666666641626524106e-1f);
    sp = fmaf(sp * r2, r, r);
    float cp = fmaf(r2, 2.4433157117e-5f, -1.3887316255e-3f);
    cp = fmaf(cp, r2, 4.1666645683e-2f);
    cp = fmaf(cp, r2, -0.5f);
    cp = fmaf(cp, r2, 1.0f);
    const int q = (int)k;
    const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}

__global__ void k(const float* x, float* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    sincos_cw(x[i], &s, &c);
    o[i] = cosf(x[i]);
    o[n + i] = cos_hw(x[i]);
    o[2 * n + i] = c;
    o[3 * n + i] = sinf(x[i]);
    o[4 * n + i] = sin_hw(x[i]);
    o[5 * n + i] = s;
    o[6 * n + i] = exp2f(x[i] * 0.004f);
}

int main() {
    const int n = 1 << 20;
    std::vector<float> x(n), o(7 * n);
    std::mt19937 g(1);
    std::uniform_real_distribution<float> d(-4000.f, 4000.f);
    for (auto& v : x) v = d(g);
    float *dx, *dout;
    hipMalloc(&dx, n * 4);
    hipMalloc(&dout, 7 * n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, 7 * n * 4, hipMemcpyDeviceToHost);
    const char* names[7] = {"ocml cosf", "f64-reduce + v_cos", "cody-waite cos", "ocml sinf", "f64-reduce + v_sin", "cody-waite sin", "ocml exp2f(rel)"};
    for (int v = 0; v < 7; ++v) {
        double mx = 0, sum = 0;
        for (int i = 0; i < n; ++i) {
            double ref = v < 3 ? cos((double)x[i]) : (v < 6 ? sin((double)x[i]) : exp2((double)(x[i] * 0.004f)));
            double e = fabs((double)o[v * n + i] - ref);
            if (v == 6) e /= ref;
            mx = e > mx ? e : mx;
            sum += e * e;
        }
        printf("%-22s max abs err %.3e  rms %.3e\n", names[v], mx, sqrt(sum / n));
    }
    // glibc on the host for comparison
    double mx = 0;
    for (int i = 0; i < n; ++i) mx = fmax(mx, fabs((double)cosf(x[i]) - cos((double)x[i])));
    printf("%-22s max abs err %.3e\n", "host glibc cosf", mx);
    return 0;
}
