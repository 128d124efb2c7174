#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include "dojo_math.hpp"
__global__ void k(const double* x, double* a, double* b, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { a[i] = dj::tatan(x[i]); b[i] = atan(x[i]); } }
int main() {
    const int n = 1 << 20; double *x, *a, *b; hipMallocManaged(&x, n * 8); hipMallocManaged(&a, n * 8); hipMallocManaged(&b, n * 8);
    for (int i = 0; i < n; ++i) { double u = (double)rand() / RAND_MAX; x[i] = (i & 1 ? -1 : 1) * exp(40 * (u - 0.5)); }
    x[0] = 0; x[1] = -0.0; x[2] = 1; x[3] = -1; x[4] = INFINITY; x[5] = 1e-320;
    k<<<n / 256, 256>>>(x, a, b, n); hipDeviceSynchronize();
    long bad = 0; for (int i = 0; i < n; ++i) if (memcmp(&a[i], &b[i], 8)) { if (bad++ < 5) printf("x %.17g table %.17g lib %.17g\n", x[i], a[i], b[i]); }
    printf("atan: %ld of %d differ bitwise from the device library\n", bad, n); return bad != 0;
}
