// What does an LDS store cost a wavefront on gfx950 when only some lanes carry data?  (round 6: the row-layout level passes move the rows of the four
// supernodes of a level through LDS; 16 of 64 lanes hold data and the stage-in phase took 3.2 k cycles for 45 ds_write2_b64 + 27 reads.)
// Variants: active lanes 64 / 16 (four quads, one per 16-lane group), instruction ds_write_b64 / ds_write2_b64 (two adjacent doubles) / ds_write_b128,
// waves per CU 4 (grid = 1024: every SIMD busy, all waves storing at once) or 1 (grid = 256).  N stores back to back, then a full wait.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_store_cost.hip -o /tmp/lds_store_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define NST 48
#define REP 50
template <int KIND, int LANES>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) k(unsigned long long* cyc, double seed) {
    __shared__ double lds[64 * (2 * NST + 1)];
    const int lane = threadIdx.x;
    const bool on = LANES == 64 || ((lane >> 2) & 3) == 1;          // 16 lanes: quad 1 of every 16-lane group
    double v[2 * NST];
    for (int i = 0; i < 2 * NST; ++i) v[i] = seed * (i + lane);
    double* base = lds + lane * (2 * NST + 1);
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int rep = 0; rep < REP; ++rep) {
        if (on) {
            if (KIND == 0) {
#pragma unroll
                for (int i = 0; i < NST; ++i) __asm__ volatile("ds_write_b64 %0, %1 offset:%2" :: "v"((unsigned)(size_t)base), "v"(v[i]), "n"(0) : "memory");
            } else if (KIND == 1) {
#pragma unroll
                for (int i = 0; i < NST; ++i) __asm__ volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:1" :: "v"((unsigned)(size_t)base), "v"(v[2 * i]), "v"(v[2 * i + 1]) : "memory");
            } else {
#pragma unroll
                for (int i = 0; i < NST; ++i) { typedef double d2 __attribute__((ext_vector_type(2))); d2 x = {v[2 * i], v[2 * i + 1]};
                    __asm__ volatile("ds_write_b128 %0, %1" :: "v"((unsigned)((size_t)base & ~15u)), "v"(x) : "memory"); }
            }
        }
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    if (seed == 12345.0) cyc[blockIdx.x] += (unsigned long long)lds[lane];
}
template <int KIND, int LANES> void run(const char* name, int waves) {
    unsigned long long* cyc; hipMalloc(&cyc, waves * 8);
    k<KIND, LANES><<<waves, 64>>>(cyc, 1.0); hipDeviceSynchronize();
    k<KIND, LANES><<<waves, 64>>>(cyc, 1.0); hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves); hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto x : h) avg += x; avg /= waves;
    printf("%-22s %2d lanes, %4d waves (%d per CU): %.1f cycles per store instruction\n", name, LANES, waves, waves / 256, avg / REP / NST);
    hipFree(cyc);
}
int main() {
    for (int waves : {1024, 256}) {
        run<0, 64>("ds_write_b64", waves); run<0, 16>("ds_write_b64", waves);
        run<1, 64>("ds_write2_b64", waves); run<1, 16>("ds_write2_b64", waves);
        run<2, 64>("ds_write_b128", waves); run<2, 16>("ds_write_b128", waves);
    }
    return 0;
}
