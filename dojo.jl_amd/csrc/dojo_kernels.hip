// dojo_kernels.hip -- the HIP kernels, compiled once per (ABI scalar type, contacts-per-body bound)
// with -DDJ_TIO=float|double -DDJ_MAXC=1|4|8 (see __graft_entry__.build_hip) so the heavy, fully
// unrolled lane program builds in parallel.  Each object exports one C launcher.
//
// One wavefront = 64/S environments x S supernodes (dojo_device.hpp); one workgroup = one wavefront.
#include <hip/hip_runtime.h>
#include "dojo_device.hpp"

#ifndef DJ_TIO
#define DJ_TIO float
#endif
#ifndef DJ_MAXC
#define DJ_MAXC 1
#endif

namespace {

struct GpuWave {
    __device__ __forceinline__ int lane() const { return (int)(threadIdx.x & 63u); }
    __device__ __forceinline__ int width() const { return 64; }
    __device__ __forceinline__ float  shfl(float v, int src) const { return __shfl(v, src, 64); }
    __device__ __forceinline__ double shfl(double v, int src) const { return __shfl(v, src, 64); }
    __device__ __forceinline__ int    shfl(int v, int src) const { return __shfl(v, src, 64); }
    __device__ __forceinline__ bool   any(bool p) const { return __any(p ? 1 : 0) != 0; }
};

// TIO = ABI scalar type, TS = state / residual precision, TL = factorization precision.
// dtype f64: <double,double,double>.  dtype f32: fp32 buffers at the ABI with fp64 internals
// <float,double,double>: the interior-point iteration drives s·γ to ~1e-9 and the condensed KKT
// then has entries ~γ/s that fp32 cannot resolve (measured in tests/emu).
// GRAD = false compiles the IFT back-solves out (forward-only launches: step!, simulate!).
template <class TIO, class TS, class TL, int MAXC, bool GRAD>
__global__ void __launch_bounds__(64) dojo_step_kernel(dj::KernelArgs<TIO, TS> A) {
    GpuWave w;
    dj::step_entry<TIO, TS, TL, MAXC, GRAD, GpuWave>(w, A, (int)blockIdx.x);
}

} // namespace

#define DJ_CAT2(a, b, c) a##b##_##c
#define DJ_CAT(a, b, c) DJ_CAT2(a, b, c)
#define DJ_LAUNCHER DJ_CAT(dojo_launch_, DJ_TIO, DJ_MAXC)

extern "C" int DJ_LAUNCHER(const void* args, int grid, void* stream, int grad) {
    const dj::KernelArgs<DJ_TIO, double>& A = *(const dj::KernelArgs<DJ_TIO, double>*)args;
    if (grad) hipLaunchKernelGGL((dojo_step_kernel<DJ_TIO, double, double, DJ_MAXC, true>), dim3(grid), dim3(64), 0, (hipStream_t)stream, A);
    else      hipLaunchKernelGGL((dojo_step_kernel<DJ_TIO, double, double, DJ_MAXC, false>), dim3(grid), dim3(64), 0, (hipStream_t)stream, A);
    return (int)hipGetLastError();
}
