// Micro-benchmark for a level-major factorization (review item 3): the cost of ONE Gauss-Jordan pass over a bordered 12 x 18
// supernode system [S | U] with one wave per SIMD, in three layouts:
//   A  the shipped one: 4 lanes per supernode, 3 rows per lane, pivot row broadcast inside the quad with DPP (16 supernodes per wavefront)
//   B  16 lanes per supernode (12 busy), 1 row per lane, pivot row through ds_bpermute (4 supernodes per wavefront)
//   C  the same, pivot row through LDS (the owner writes its row, everybody reads it back: broadcast reads)
// At a tree level of the Ant 4 of the 16 supernodes are active, so layout A spends a pass of 16 on 4; B / C would do those 4 in one pass of 4
// (the rows of a supernode would have to reach the twelve lanes first: not measured here).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 200
__device__ __forceinline__ double rcpd(double a) { double r = __builtin_amdgcn_rcp(a); double e = fma(-a, r, 1.0); r = fma(r, e, r); e = fma(-a, r, 1.0); return fma(r, e, r); }
template <int CTRL> __device__ __forceinline__ double dppd(double v) {
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ double qb(double v, int o) { switch (o) { case 0: return dppd<0x00>(v); case 1: return dppd<0x55>(v); case 2: return dppd<0xAA>(v); default: return dppd<0xFF>(v); } }
__device__ __forceinline__ double bp(double v, int srclane) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(srclane << 2, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(srclane << 2, __double2loint(v)));
}

template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, unsigned long long* cyc, double seed) {
    __shared__ double lds[4 * 20];
    const int lane = threadIdx.x;
    unsigned long long t0 = 0, t1 = 0;
    if (MODE == 0) {
        const int q = lane & 3;
        double A[3][18];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 18; ++j) A[i][j] = seed * (0.01 * (i + 3 * q) + 0.001 * j) + ((3 * q + i) == j ? 4.0 : 0.0);
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < N_IT; ++it) {
#pragma unroll
            for (int p = 0; p < 12; ++p) {
                const int o = p / 3, ro = p % 3;
                double prow[18];
#pragma unroll
                for (int c = 0; c < 18; ++c) prow[c] = qb(A[ro][c], o);
                const double ip = rcpd(prow[p]);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const bool own = (q == o) && (r == ro);
                    const double f = own ? 0.0 : A[r][p] * ip;
#pragma unroll
                    for (int c = 0; c < 18; ++c) if (c != p) A[r][c] = fma(-f, prow[c], A[r][c]);
                    A[r][p] = own ? A[r][p] : -f;
                }
            }
        }
        t1 = __builtin_readcyclecounter();
        double s = 0; for (int i = 0; i < 3; ++i) for (int j = 0; j < 18; ++j) s += A[i][j];
        out[blockIdx.x * 64 + lane] = s;
    } else {
        const int g = lane >> 4, i = lane & 15;        // group of 16 lanes, row i (rows 12..15: idle)
        double A[18];
        for (int j = 0; j < 18; ++j) A[j] = seed * (0.01 * i + 0.001 * j) + (i == j ? 4.0 : 0.0);
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < N_IT; ++it) {
#pragma unroll
            for (int p = 0; p < 12; ++p) {
                double prow[18];
                if (MODE == 1) {
#pragma unroll
                    for (int c = 0; c < 18; ++c) prow[c] = bp(A[c], 16 * g + p);
                } else {
                    if (i == p) {
#pragma unroll
                        for (int c = 0; c < 18; ++c) lds[20 * g + c] = A[c];
                    }
                    __asm__ volatile("" ::: "memory");
#pragma unroll
                    for (int c = 0; c < 18; ++c) prow[c] = lds[20 * g + c];
                    __asm__ volatile("" ::: "memory");
                }
                const double ip = rcpd(prow[p]);
                const bool own = (i == p);
                const double f = own ? 0.0 : A[p] * ip;
#pragma unroll
                for (int c = 0; c < 18; ++c) if (c != p) A[c] = fma(-f, prow[c], A[c]);
                A[p] = own ? A[p] : -f;
            }
        }
        t1 = __builtin_readcyclecounter();
        double s = 0; for (int j = 0; j < 18; ++j) s += A[j];
        out[blockIdx.x * 64 + lane] = s;
    }
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int waves) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, waves * 64 * 8); hipMalloc(&cyc, waves * 8);
    k<MODE><<<waves, 64>>>(out, cyc, 1.0); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<MODE><<<waves, 64>>>(out, cyc, 1.0); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(waves); hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= waves;
    printf("%-64s waves %5d  %.3f ms   cycles per 12-pivot pass %.0f\n", name, waves, ms, avg / N_IT);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("A quad: 3 rows per lane, DPP quad broadcast (16 supernodes/wave)", 1024);
    run<1>("B 16 lanes: 1 row per lane, ds_bpermute broadcast (4 supernodes/wave)", 1024);
    run<2>("C 16 lanes: 1 row per lane, LDS broadcast (4 supernodes/wave)", 1024);
    return 0;
}
