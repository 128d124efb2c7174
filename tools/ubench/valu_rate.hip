// Micro-benchmark: issue cost (cycles per wave-instruction) of the instruction kinds the lane program
// is made of, with ONE wave per SIMD (the step kernel's occupancy).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 2000
template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, unsigned long long* cyc, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT; ++i) {
        if (MODE == 0) {            // 8 independent fp64 FMA chains
            a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
            a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
        } else if (MODE == 1) {     // 1 dependent fp64 FMA chain (8 per iteration)
            a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c);
            a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c);
        } else if (MODE == 2) {     // 8 independent fp32 FMA chains
            float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
            f0 = fmaf(f0, 1.0001f, 1e-3f); f1 = fmaf(f1, 1.0001f, 1e-3f); f2 = fmaf(f2, 1.0001f, 1e-3f); f3 = fmaf(f3, 1.0001f, 1e-3f);
            f4 = fmaf(f4, 1.0001f, 1e-3f); f5 = fmaf(f5, 1.0001f, 1e-3f); f6 = fmaf(f6, 1.0001f, 1e-3f); f7 = fmaf(f7, 1.0001f, 1e-3f);
            a0 = f0; a1 = f1; a2 = f2; a3 = f3; a4 = f4; a5 = f5; a6 = f6; a7 = f7;
        } else if (MODE == 3) {     // quad broadcast of 8 doubles through DPP (16 v_mov_dpp) + 8 fp64 adds
            #define BC(x) __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(x), 0x55, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(__double2loint(x), 0x55, 0xF, 0xF, true))
            a0 += BC(a1); a1 += BC(a2); a2 += BC(a3); a3 += BC(a4); a4 += BC(a5); a5 += BC(a6); a6 += BC(a7); a7 += BC(a0);
        } else if (MODE == 4) {     // 8 ds_bpermute (4 doubles) + adds
            int src = ((threadIdx.x + 4) & 63) << 2;
            #define BP(x) __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(x)), __builtin_amdgcn_ds_bpermute(src, __double2loint(x)))
            a0 += BP(a1); a1 += BP(a2); a2 += BP(a3); a3 += BP(a0);
        } else if (MODE == 5) {     // 8 independent fp64 mul + add (unfused)
            a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a4 = a4 + c; a5 = a5 + c; a6 = a6 + c; a7 = a7 + c;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int per_iter, int waves) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, waves * 64 * 8); hipMalloc(&cyc, waves * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<waves, 64>>>(out, cyc, 1.0); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<waves, 64>>>(out, cyc, 1.0); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(waves); hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= waves;
    printf("%-34s waves %5d  %.3f ms  counter cycles/instr %.2f   wall ns/instr %.2f\n", name, waves, ms, avg / (double)(N_IT * per_iter), ms * 1e6 / (double)(N_IT * per_iter) / ((waves + 1023) / 1024));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int waves : {1024, 2048}) {
        run<0>("fp64 fma x8 independent", 8, waves);
        run<1>("fp64 fma x8 dependent", 8, waves);
        run<2>("fp32 fma x8 (+cvt)", 8, waves);
        run<3>("dpp bcast double x8 + add", 8, waves);
        run<4>("ds_bpermute double x4 + add", 4, waves);
        run<5>("fp64 mul x4 + add x4", 8, waves);
    }
    return 0;
}
