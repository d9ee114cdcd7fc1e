// development aid: accuracy of candidate unit-phasor evaluations e^{2 pi i rev} (rev in double) on gfx950
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

__device__ inline void ph_hw(double rev, float* c, float* s) {
    const float r = (float)(rev - floor(rev));
    *c = __builtin_amdgcn_cosf(r);
    *s = __builtin_amdgcn_sinf(r);
}
__device__ inline void ph_hw_signed(double rev, float* c, float* s) {
    const float r = (float)(rev - rint(rev));
    *c = __builtin_amdgcn_cosf(r);
    *s = __builtin_amdgcn_sinf(r);
}
// hardware value at the rounded argument, first-order correction for the rounding residual
__device__ inline void ph_hw_split(double rev, float* c, float* s) {
    const double rr = rev - rint(rev);
    const float r = (float)rr;
    const float lo = (float)((rr - (double)r) * 6.283185307179586);
    const float c0 = __builtin_amdgcn_cosf(r), s0 = __builtin_amdgcn_sinf(r);
    *c = fmaf(-lo, s0, c0);
    *s = fmaf(lo, c0, s0);
}
// quadrant reduction in double, Taylor polynomials in float on [-pi/4, pi/4]
__device__ inline void ph_poly(double rev, float* c, float* s) {
    const double rr = rev - floor(rev);
    const double q = rint(rr * 4.0);
    const float x = (float)((rr - q * 0.25) * 6.283185307179586);
    const float x2 = x * x;
    float sp = fmaf(x2, 2.7557319e-6f, -1.9841270e-4f);
    sp = fmaf(sp, x2, 8.3333333e-3f);
    sp = fmaf(sp, x2, -1.6666667e-1f);
    sp = fmaf(sp * x2, x, x);
    float cp = fmaf(x2, -2.7557319e-7f, 2.4801587e-5f);
    cp = fmaf(cp, x2, -1.3888889e-3f);
    cp = fmaf(cp, x2, 4.1666667e-2f);
    cp = fmaf(cp, x2, -0.5f);
    cp = fmaf(cp, x2, 1.0f);
    const int k = (int)q;
    const float ss = (k & 1) ? cp : sp, cc = (k & 1) ? sp : cp;
    *s = (k & 2) ? -ss : ss;
    *c = ((k + 1) & 2) ? -cc : cc;
}

__global__ void k(const double* x, float* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ph_hw(x[i], &o[i], &o[n + i]);
    ph_hw_signed(x[i], &o[2 * n + i], &o[3 * n + i]);
    ph_hw_split(x[i], &o[4 * n + i], &o[5 * n + i]);
    ph_poly(x[i], &o[6 * n + i], &o[7 * n + i]);
}

int main() {
    const int n = 1 << 20;
    std::vector<double> x(n);
    std::vector<float> o(8 * n);
    std::mt19937 g(1);
    std::uniform_real_distribution<double> d(-9.0, 9.0);
    for (auto& v : x) v = d(g);
    double* dx;
    float* dout;
    hipMalloc(&dx, n * 8);
    hipMalloc(&dout, 8 * n * 4);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, 8 * n * 4, hipMemcpyDeviceToHost);
    const char* names[4] = {"hw, rev in [0,1)", "hw, rev in [-.5,.5)", "hw + residual", "f64 quadrant + poly"};
    for (int v = 0; v < 4; ++v) {
        double mx = 0, sum = 0, mxang = 0;
        for (int i = 0; i < n; ++i) {
            const double a = 6.283185307179586 * x[i];
            const double ec = (double)o[(2 * v) * n + i] - cos(a), es = (double)o[(2 * v + 1) * n + i] - sin(a);
            const double e = sqrt(ec * ec + es * es);
            const double ang = fabs(-sin(a) * ec + cos(a) * es);   // tangential (phase) component
            mx = e > mx ? e : mx;
            mxang = ang > mxang ? ang : mxang;
            sum += e * e;
        }
        printf("%-22s max |err| %.3e  max phase err %.3e  rms %.3e\n", names[v], mx, mxang, sqrt(sum / n));
    }
    return 0;
}
