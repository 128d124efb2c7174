// accuracy / cost of v_rcp_f64 (+ Newton steps) against the IEEE division, for the pivots of the Gauss-Jordan
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double a = x[i];
    double r = __builtin_amdgcn_rcp(a);
    r0[i] = r;
    double e = fma(-a, r, 1.0); r = fma(r, e, r);
    r1[i] = r;
    e = fma(-a, r, 1.0); r = fma(r, e, r);
    r2[i] = r;
}
int main() {
    const int N = 1 << 20;
    std::vector<double> x(N);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < N; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); x[i] = std::ldexp(0.5 + u, (int)(s % 80) - 40) * ((s & 1) ? 1 : -1); }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, N * 8); hipMalloc(&d0, N * 8); hipMalloc(&d1, N * 8); hipMalloc(&d2, N * 8);
    hipMemcpy(dx, x.data(), N * 8, hipMemcpyHostToDevice);
    k<<<N / 256, 256>>>(dx, d0, d1, d2);
    std::vector<double> r0(N), r1(N), r2(N);
    hipMemcpy(r0.data(), d0, N * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, N * 8, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), d2, N * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < N; ++i) { double t = 1.0 / x[i]; e0 = fmax(e0, fabs(r0[i] - t) / fabs(t)); e1 = fmax(e1, fabs(r1[i] - t) / fabs(t)); e2 = fmax(e2, fabs(r2[i] - t) / fabs(t)); }
    printf("max relative error vs 1/x: v_rcp_f64 %.3e, +1 Newton %.3e, +2 Newton %.3e (eps = %.3e)\n", e0, e1, e2, 2.22e-16);
    return 0;
}
