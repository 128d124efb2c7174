// accuracy of v_rsq_f64 + Goldschmidt iterations (dj::tsqrt on the device: the backend's f64 sqrt expansion without its
// range scaling and special-case selects) against the IEEE square root
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ double fast_sqrt(double a) {
    double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5); g = fma(g, r, g); h = fma(h, r, h);
    double d = fma(-g, g, a); g = fma(d, h, g);
    d = fma(-g, g, a); g = fma(d, h, g);
    return a == 0.0 ? 0.0 : g;
}
__global__ void k(const double* x, double* r) { int i = blockIdx.x * blockDim.x + threadIdx.x; r[i] = fast_sqrt(x[i]); }
int main() {
    const int N = 1 << 20;
    std::vector<double> x(N);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < N; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); x[i] = std::ldexp(0.5 + u, (int)(s % 160) - 100); }
    x[0] = 0.0;
    double *dx, *dr;
    hipMalloc(&dx, N * 8); hipMalloc(&dr, N * 8);
    hipMemcpy(dx, x.data(), N * 8, hipMemcpyHostToDevice);
    k<<<N / 256, 256>>>(dx, dr);
    std::vector<double> r(N);
    hipMemcpy(r.data(), dr, N * 8, hipMemcpyDeviceToHost);
    double e = 0; int nex = 0;
    for (int i = 1; i < N; ++i) { double t = std::sqrt(x[i]); e = fmax(e, fabs(r[i] - t) / t); nex += (r[i] == t); }
    printf("sqrt(0) = %g; max relative error vs sqrt over 2^20 inputs in [2^-100, 2^60]: %.3e (eps = 2.22e-16); bit-equal on %.4f %%\n", r[0], e, 100.0 * nex / (N - 1));
    return 0;
}
