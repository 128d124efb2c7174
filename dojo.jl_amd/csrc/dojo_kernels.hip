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
#ifndef DJ_QUAD
#define DJ_QUAD 1      // 1: four lanes per supernode, one wavefront per workgroup (<= 16 bodies); 2: the same over two wavefronts (<= 32 bodies);
                       // 0: one lane per supernode (<= 64 bodies)
#endif

namespace {

// NW = wavefronts per workgroup.  NW = 1: the workgroup is one wavefront (<= 16 supernodes x 4 roles).  NW = 2: one
// environment of 17..32 bodies spans two wavefronts; quads never straddle a wave, so everything quad-local is unchanged,
// and the exchanges between quads (LDS mailbox, reductions, votes) go through LDS with real barriers.
// RF = the refining build of the kernels (DJ_REFINE, dojo_device.hpp): re-solves the environments the plain kernels deferred
template <int NW, bool RF = false>
struct GpuWave {
    static constexpr bool kRefine = RF;
    static constexpr bool kLockstep = true;     // the 64 lanes of a wave (so the 4 lanes of a quad) execute every instruction together
    static constexpr int kWaves = NW;
    static constexpr int kWidth = 64 * NW;      // lanes of the workgroup, for strides that must be compile-time constants (0: ask width())
    static constexpr int kReplicas = 1;         // (GpuWaveRep: wavefronts that hold the same environment, dojo_stepc_kernel)
    __device__ __forceinline__ int atomic_inc(int* p) const { return atomicAdd(p, 1); }
    void* lds_;
    double* red_;                               // reduction / vote scratch at the end of the LDS block (NW > 1)
#ifdef DJ_PROF2
    void* p2_;                                  // 24 cycle counters (LaneProgram::p2)
#endif
    __device__ __forceinline__ void* lds() const { return lds_; }
    // NW = 1: LDS instructions of a wave execute in issue order, so lanes only need the compiler to keep LDS accesses in
    // program order.  (__syncthreads() would also drain the outstanding global stores -- vmcnt(0) -- which costs the IFT
    // sweeps a memory round trip per pipeline step.)  NW > 1: wait for this wave's LDS traffic, then the hardware barrier.
    __device__ __forceinline__ void sync() const {
        if (NW == 1) __asm__ volatile("" ::: "memory");
        else __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // ... and when lanes read GLOBAL memory other lanes of the workgroup wrote: this wave's stores have left it (vmcnt(0)); the vector L1 is
    // shared by the wavefronts of a workgroup, so no invalidation is needed at workgroup scope
    __device__ __forceinline__ void sync_mem() const {
        if (NW == 1) __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __device__ __forceinline__ int lane() const { return (int)threadIdx.x; }              // lane index inside the workgroup
    __device__ __forceinline__ int width() const { return 64 * NW; }
    __device__ __forceinline__ unsigned long long clock() const { return __builtin_readcyclecounter(); }
    // 1/x: v_rcp_f64 (4.6e-8 relative) + two Newton steps = the IEEE quotient to the last bit on 2^20 random inputs
    // (tools/ubench/rcp_test.hip), 5 instructions instead of the ~11 of the division expansion
    static __device__ __forceinline__ double rcp(double a) { double r = __builtin_amdgcn_rcp(a); double e = fma(-a, r, 1.0); r = fma(r, e, r); e = fma(-a, r, 1.0); return fma(r, e, r); }
    static __device__ __forceinline__ float rcp(float a) { return 1.0f / a; }
    // shuffles inside the wavefront only (the quad mapping with NW > 1 does not use them across quads)
    __device__ __forceinline__ float  shfl(float v, int src) const { return __shfl(v, src & 63, 64); }
    __device__ __forceinline__ double shfl(double v, int src) const { return __shfl(v, src & 63, 64); }
    __device__ __forceinline__ int    shfl(int v, int src) const { return __shfl(v, src & 63, 64); }
    __device__ __forceinline__ bool   any(bool p) const {
        bool a = __any(p ? 1 : 0) != 0;
        if (NW == 1) return a;
        int* r = (int*)red_;
        sync(); if ((threadIdx.x & 63u) == 0) r[threadIdx.x >> 6] = a ? 1 : 0; sync();
        int o = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) o |= r[w];
        return o != 0;
    }
    // workgroup-wide reductions: butterfly inside the wave, then NW partial results through LDS
    template <class OP> __device__ __forceinline__ double wg_reduce(double v, OP op) const {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o, 64));
        if (NW == 1) return v;
        sync(); if ((threadIdx.x & 63u) == 0) red_[threadIdx.x >> 6] = v; sync();
        double r = red_[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) r = op(r, red_[w]);
        return r;
    }
    __device__ __forceinline__ double wg_max(double v) const { return wg_reduce(v, [](double a, double b) { return a > b ? a : b; }); }
    __device__ __forceinline__ double wg_min(double v) const { return wg_reduce(v, [](double a, double b) { return a < b ? a : b; }); }
    __device__ __forceinline__ double wg_sum(double v) const { return wg_reduce(v, [](double a, double b) { return a + b; }); }
    __device__ __forceinline__ int wg_or(int v) const { return any(v != 0) ? 1 : 0; }
    // quad-local data movement on the DPP path (v_mov_b32 quad_perm): no LDS traffic, VALU latency
    template <int CTRL> static __device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
    template <int CTRL> static __device__ __forceinline__ float dppf(float v) { return __int_as_float(dpp<CTRL>(__float_as_int(v))); }
    template <int CTRL> static __device__ __forceinline__ double dppd(double v) { return __hiloint2double(dpp<CTRL>(__double2hiint(v)), dpp<CTRL>(__double2loint(v))); }
    template <int CTRL> static __device__ __forceinline__ int    dppx(int v) { return dpp<CTRL>(v); }
    template <int CTRL> static __device__ __forceinline__ float  dppx(float v) { return dppf<CTRL>(v); }
    template <int CTRL> static __device__ __forceinline__ double dppx(double v) { return dppd<CTRL>(v); }
    template <class V> __device__ __forceinline__ V quad_bcast(V v, int o) const {      // value of lane o of my quad (o folds to a constant after unrolling)
        switch (o) { case 0: return dppx<0x00>(v); case 1: return dppx<0x55>(v); case 2: return dppx<0xAA>(v); default: return dppx<0xFF>(v); }
    }
    template <class V> __device__ __forceinline__ V quad_xor(V v, int m) const {        // value of lane (l ^ m) inside my quad, m = 1 or 2
        return m == 1 ? dppx<0xB1>(v) : dppx<0x4E>(v);                                  // quad_perm [1,0,3,2] / [2,3,0,1]
    }
    // Reduction over the 16 supernode slots of a wavefront that is ONE environment (LaneProgram::env_reduce_quad; the four lanes of a quad hold
    // the same value): two rotations inside every 16-lane row on the DPP path, then the four rows through v_readlane -- about twenty
    // instructions and no LDS round trip (the LDS form: a write, then eight dependent ds_read2_b64, each behind a full wait, per value).
    // Every lane ends with the same bits: the result is formed from the four row values in SGPRs, in row order.
    static constexpr bool kWaveReduce = true;   // (NW = 2: every wavefront reduces its 16 slots, the two results cross through red_)
    static __device__ __forceinline__ double rdlane(double v, int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); }
    static __device__ __forceinline__ float rdlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
    template <class V, class OP> static __device__ __forceinline__ V reduce_quads16(V v, OP op) {
        v = op(v, dppx<0x124>(v));          // row_ror:4
        v = op(v, dppx<0x128>(v));          // row_ror:8: every lane holds its row's four supernodes
        const V r0 = rdlane(v, 0), r1 = rdlane(v, 16), r2 = rdlane(v, 32), r3 = rdlane(v, 48);
        return op(op(op(r0, r1), r2), r3);
    }
    template <int NV, class V, class OP> __device__ __forceinline__ void reduce_slots(V (&v)[NV], OP op) const {
#pragma unroll
        for (int n = 0; n < NV; ++n) v[n] = reduce_quads16(v[n], op);
        if (NW == 1) return;
        sync();                                     // (whoever used red_ before has read it)
        if ((threadIdx.x & 63u) == 0) {
#pragma unroll
            for (int n = 0; n < NV; ++n) red_[(threadIdx.x >> 6) * NV + n] = (double)v[n];
        }
        sync();
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            V r = V(red_[n]);
#pragma unroll
            for (int w = 1; w < NW; ++w) r = op(r, V(red_[w * NV + n]));
            v[n] = r;
        }
    }
    // Row layout of the level passes (LaneProgram::factorize_rows): 16 lanes per supernode, one matrix row per lane.  gfx950's DP ALU
    // knows exactly one DPP control, row_newbcast:P (lane P of every 16-lane row), and takes it INSIDE the fp64 multiply-add: a pivot-row
    // entry reaches the twelve rows of its supernode in the instruction that uses it -- no separate broadcast, no LDS traffic.
    static constexpr bool kRows = NW <= 2;      // (two wavefronts: each runs the row passes of its own sixteen slots)
    template <int P> static __device__ __forceinline__ double row_bcast_(double v) {    // v_mov_b64_dpp row_newbcast:P
        const long long x = __builtin_bit_cast(long long, v);
        const long long y = __builtin_amdgcn_mov_dpp(x, 0x150 + P, 0xF, 0xF, false);
        return __builtin_bit_cast(double, y);
    }
    __device__ __forceinline__ double row_bcast(double v, int p) const {                // (p folds to a constant after unrolling)
        switch (p) { case 0: return row_bcast_<0>(v); case 1: return row_bcast_<1>(v); case 2: return row_bcast_<2>(v); case 3: return row_bcast_<3>(v);
                     case 4: return row_bcast_<4>(v); case 5: return row_bcast_<5>(v); case 6: return row_bcast_<6>(v); case 7: return row_bcast_<7>(v);
                     case 8: return row_bcast_<8>(v); case 9: return row_bcast_<9>(v); case 10: return row_bcast_<10>(v); default: return row_bcast_<11>(v); }
    }
    // acc += (src of lane P of my row) * f.  The compiler's DPP combiner does not fold a 64-bit v_mov_dpp into the multiply-add, hence
    // the asm statement; its hazard recognizer does not see a VALU write inside one either, so callers keep two instructions between
    // a row_fmac that writes a register and the next DPP read of that register (CDNA3 ISA 4.5: VALU write -> DPP read, 2 wait states).
    template <int P> static __device__ __forceinline__ void row_fmac_(double& acc, double src, double f) {
        __asm__("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(f), "n"(P));
    }
    __device__ __forceinline__ void row_fmac(double& acc, double src, double f, int p) const {
        switch (p) { case 0: row_fmac_<0>(acc, src, f); break; case 1: row_fmac_<1>(acc, src, f); break; case 2: row_fmac_<2>(acc, src, f); break; case 3: row_fmac_<3>(acc, src, f); break;
                     case 4: row_fmac_<4>(acc, src, f); break; case 5: row_fmac_<5>(acc, src, f); break; case 6: row_fmac_<6>(acc, src, f); break; case 7: row_fmac_<7>(acc, src, f); break;
                     case 8: row_fmac_<8>(acc, src, f); break; case 9: row_fmac_<9>(acc, src, f); break; case 10: row_fmac_<10>(acc, src, f); break; default: row_fmac_<11>(acc, src, f); break; }
    }
    template <int P> __device__ __forceinline__ void row_fmac_c(double& acc, double src, double f) const { row_fmac_<P>(acc, src, f); }
    template <int P> __device__ __forceinline__ double row_bcast_c(double v) const { return row_bcast_<P>(v); }
    template <int P> __device__ __forceinline__ void row_fmac_c(float& acc, float src, float f) const { row_fmac(acc, src, f, P); }
    template <int P> __device__ __forceinline__ float row_bcast_c(float v) const { return row_bcast(v, P); }
    __device__ __forceinline__ void row_fmac(float& acc, float src, float f, int p) const { acc += __shfl(src, (int)((threadIdx.x & 48u) + p), 64) * f; }   // (no fp32 factorization build exists on the GPU)
    __device__ __forceinline__ float row_bcast(float v, int p) const { return __shfl(v, (int)((threadIdx.x & 48u) + p), 64); }
    static __device__ __forceinline__ void dpp_settle() { __asm__ volatile("s_nop 1"); }     // the two wait states, where the order of the statements does not give them
};

// The continuation kernel's wavefronts (Globals::iter_cap, dojo_stepc_kernel): R wavefronts of one workgroup carry the SAME
// environment(s), each in an LDS block of its own, and run the same Newton loop on identical values; they only meet in the line
// search, where replica r evaluates trial ls0 + r and the verdicts cross through `xchg_` between two workgroup barriers.
template <int R>
struct GpuWaveRep : GpuWave<1> {
    static constexpr int kReplicas = R;
    double* xchg_;                              // [R][16 supernode slots][4]
    __device__ __forceinline__ int lane() const { return (int)(threadIdx.x & 63u); }
    __device__ __forceinline__ int replica() const { return (int)(threadIdx.x >> 6); }
    __device__ __forceinline__ double* replica_xchg() const { return xchg_; }
    __device__ __forceinline__ void replica_sync() const { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
};
// replicas per workgroup: what fits the 160 KB of LDS next to the exchange block
template <class TIO, class TS, int MAXC> constexpr int cont_replicas() {
    constexpr int per = (dj::StepLds<TIO, TS, MAXC, 0, true, true, 1>::bytes + 15) / 16 * 16;
    constexpr int r = (160 * 1024 - 4 * 16 * 4 * 8) / per;
    return r > 4 ? 4 : r;
}

// TIO = ABI scalar type, TS = state / residual precision, TL = factorization precision.
// dtype f64: <double,double,double>.  dtype f32: fp32 buffers at the ABI with fp64 internals
// <float,double,double>: the interior-point iteration drives s·γ to ~1e-9 and the condensed KKT
// then has entries ~γ/s that fp32 cannot resolve (measured in tests/emu).
// GRAD = false compiles the IFT back-solves out (forward-only launches: step!, simulate!).
// Two kernels, two register allocations: the Newton loop (dojo_step_kernel) and the IFT gradients
// (dojo_grad_kernel: re-linearization at the converged solution + pipelined column sweeps).
// DJ_FWD_WAVES / DJ_GRAD_WAVES = resident waves per SIMD the compiler must allow for
// (512 unified VGPRs / waves; LDS must fit as well).
#ifndef DJ_FWD_WAVES
#define DJ_FWD_WAVES 1
#endif
#ifndef DJ_GRAD_WAVES
#define DJ_GRAD_WAVES 1
#endif
#ifdef DJ_PROF2
#define DJ_P2_DECL(w) __shared__ unsigned long long p2_buf[24]; (w).p2_ = (void*)p2_buf;
#else
#define DJ_P2_DECL(w)
#endif
template <class TIO, class TS, class TL, int MAXC, bool QUAD, int NW>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(DJ_FWD_WAVES, DJ_FWD_WAVES)))
dojo_step_kernel(dj::KernelArgs<TIO, TS> A) {
    typedef dj::StepLds<TIO, TS, MAXC, 0, QUAD, true, NW> LY;
    __shared__ double lds_buf[(LY::bytes + 7) / 8];
    GpuWave<NW> w;
    w.lds_ = (void*)lds_buf; w.red_ = (double*)((char*)lds_buf + LY::red_off);
    DJ_P2_DECL(w)
    // (a launch lasts as long as its last wavefront: the workgroups whose solves were long in the previous step are handed out first)
    dj::step_entry<TIO, TS, TL, MAXC, QUAD, GpuWave<NW>>(w, A, A.dispatch ? A.dispatch[blockIdx.x] : (int)blockIdx.x);
}
// Globals::iter_cap: the rest of the Newton loops the step kernel left unfinished, one listed workgroup of the step kernel per
// pass of a workgroup here, the line-search trials side by side on R replicas (quad mapping, one wavefront per environment set)
template <class TIO, class TS, class TL, int MAXC, int R>
__global__ void __launch_bounds__(64 * R) __attribute__((amdgpu_waves_per_eu(1, 1)))
dojo_stepc_kernel(dj::KernelArgs<TIO, TS> A) {
    typedef dj::StepLds<TIO, TS, MAXC, 0, true, true, 1> LY;
    constexpr int PER = (LY::bytes + 15) / 16 * 16;
    __shared__ double lds_buf[(R * PER + R * 16 * 4 * 8) / 8];
    GpuWaveRep<R> w;
    const int rep = (int)(threadIdx.x >> 6);
    w.lds_ = (void*)((char*)lds_buf + rep * PER); w.red_ = (double*)((char*)w.lds_ + LY::red_off);
    DJ_P2_DECL(w)
    w.xchg_ = (double*)((char*)lds_buf + R * PER);
    const int n = *(volatile int*)A.cont_count;
    for (int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) {
        dj::step_entry<TIO, TS, TL, MAXC, true, GpuWaveRep<R>, true>(w, A, A.cont_list[i]);
        w.replica_sync();                       // (a replica's LDS block and the exchange slots are reused by the next listed workgroup)
    }
}
// the Newton loop once more, with every linear solve refined once the cones are stiff, for the environments the plain kernel
// deferred (DJ_STATUS_DEFERRED); workgroups without one leave at once
template <class TIO, class TS, class TL, int MAXC, bool QUAD, int NW>
__global__ void __launch_bounds__(64 * NW) dojo_stepp_kernel(dj::KernelArgs<TIO, TS> A) {
    typedef dj::StepLds<TIO, TS, MAXC, 0, QUAD, true, NW> LY;
    __shared__ double lds_buf[(LY::bytes + 7) / 8];
    GpuWave<NW, true> w;
    w.lds_ = (void*)lds_buf; w.red_ = (double*)((char*)lds_buf + LY::red_off);
    DJ_P2_DECL(w)
    dj::step_entry<TIO, TS, TL, MAXC, QUAD, GpuWave<NW, true>>(w, A, (int)blockIdx.x);
}
template <class TIO, class TS, class TL, int MAXC, bool QUAD, int NW>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(DJ_GRAD_WAVES, DJ_GRAD_WAVES)))
dojo_grad_kernel(dj::KernelArgs<TIO, TS> A) {
    typedef dj::StepLds<TIO, TS, MAXC, 1, QUAD, true, NW> LY;
    __shared__ double lds_buf[(LY::bytes + 7) / 8];
    GpuWave<NW> w;
    w.lds_ = (void*)lds_buf; w.red_ = (double*)((char*)lds_buf + LY::red_off);
    DJ_P2_DECL(w)
    dj::grad_entry<TIO, TS, TL, MAXC, QUAD, GpuWave<NW>>(w, A, (int)blockIdx.x);
}
// Globals::iter_cap: the IFT of the workgroups on the continuation list, behind dojo_stepc_kernel (dojo_grad_kernel skips them)
template <class TIO, class TS, class TL, int MAXC>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DJ_GRAD_WAVES, DJ_GRAD_WAVES)))
dojo_gradc_kernel(dj::KernelArgs<TIO, TS> A) {
    typedef dj::StepLds<TIO, TS, MAXC, 1, true, true, 1> LY;
    __shared__ double lds_buf[(LY::bytes + 7) / 8];
    GpuWave<1> w;
    w.lds_ = (void*)lds_buf; w.red_ = (double*)((char*)lds_buf + LY::red_off);
    DJ_P2_DECL(w)
    const int n = *(volatile int*)A.cont_count;
    for (int i = (int)blockIdx.x; i < n; i += (int)gridDim.x)
        dj::grad_entry<TIO, TS, TL, MAXC, true, GpuWave<1>, 0, true>(w, A, A.cont_list[i]);
}
// the IFT kernel of the environments whose linear solves were being refined (DJ_REFINE: stiff cones, max γ/s beyond
// Globals::refine_w): column by column through the refined general solve; workgroups without such an environment leave
// at once.  LDS layout of the step kernel (the lane program's state stays alive).  Quad mappings only.
template <class TIO, class TS, class TL, int MAXC, bool QUAD, int NW>
__global__ void __launch_bounds__(64 * NW) dojo_gradp_kernel(dj::KernelArgs<TIO, TS> A) {
    typedef dj::StepLds<TIO, TS, MAXC, 0, QUAD, true, NW> LY;
    __shared__ double lds_buf[(LY::bytes + 7) / 8];
    GpuWave<NW, true> w;
    w.lds_ = (void*)lds_buf; w.red_ = (double*)((char*)lds_buf + LY::red_off);
    DJ_P2_DECL(w)
    dj::grad_entry<TIO, TS, TL, MAXC, QUAD, GpuWave<NW, true>, 2>(w, A, (int)blockIdx.x);
}
// the IFT kernel for the contact-data columns (get_contact_gradients); quad mappings only
template <class TIO, class TS, class TL, int MAXC, bool QUAD, int NW>
__global__ void __launch_bounds__(64 * NW) dojo_cgrad_kernel(dj::KernelArgs<TIO, TS> A) {
    typedef dj::StepLds<TIO, TS, MAXC, 2, QUAD, true, NW> LY;
    __shared__ double lds_buf[(LY::bytes + 7) / 8];
    GpuWave<NW> w;
    w.lds_ = (void*)lds_buf; w.red_ = (double*)((char*)lds_buf + LY::red_off);
    DJ_P2_DECL(w)
    dj::grad_entry<TIO, TS, TL, MAXC, QUAD, GpuWave<NW>, 1>(w, A, (int)blockIdx.x);
}

} // namespace

#define DJ_CAT2(a, b, c, d) a##b##_##c##_##d
#define DJ_CAT(a, b, c, d) DJ_CAT2(a, b, c, d)
#if DJ_MLIM     // the general lane-mapping builds: joint limits on several coordinates / both halves (DJ_MLIM), kinematic loops (DJ_CUT), translational springs / dampers (DJ_TSD)
#define DJ_LAUNCHER DJ_CAT(dojo_launch_gen_, DJ_TIO, DJ_MAXC, DJ_QUAD)
#elif DJ_LINEAR   // LinearContact builds: the step kernel alone (forward only, like the reference; no refinement, no IFT)
#define DJ_LAUNCHER DJ_CAT(dojo_launch_lin_, DJ_TIO, DJ_MAXC, DJ_QUAD)
#elif DJ_SS     // builds with body-body contacts: the step kernel alone (forward only)
#define DJ_LAUNCHER DJ_CAT(dojo_launch_ss_, DJ_TIO, DJ_MAXC, DJ_QUAD)
#elif DJ_TSD    // builds that evaluate translational springs / dampers (KernelArgs::tsd)
#define DJ_LAUNCHER DJ_CAT(dojo_launch_tsd_, DJ_TIO, DJ_MAXC, DJ_QUAD)
#define DJ_CLAUNCHER DJ_CAT(dojo_launch_cgrad_tsd_, DJ_TIO, DJ_MAXC, DJ_QUAD)
#else
#define DJ_LAUNCHER DJ_CAT(dojo_launch_, DJ_TIO, DJ_MAXC, DJ_QUAD)
#define DJ_CLAUNCHER DJ_CAT(dojo_launch_cgrad_, DJ_TIO, DJ_MAXC, DJ_QUAD)
#endif

// `phases`: 1 = the step kernel (+ the refining one), 2 = the IFT kernel (+ the refining one), 4 = the continuation of the step kernel
// (Globals::iter_cap; `grid` = its workgroups, each loops over the list), 8 = the IFT of the continued workgroups.  mid_event (may be null)
// is recorded behind the step kernels, so that each kernel can be timed on its own.
extern "C" int DJ_LAUNCHER(const void* args, int grid, void* stream, int phases, void* mid_event) {
    const dj::KernelArgs<DJ_TIO, double>& A = *(const dj::KernelArgs<DJ_TIO, double>*)args;
    constexpr int NW = DJ_QUAD == 2 ? 2 : 1;
    if (phases & 1) {
        hipLaunchKernelGGL((dojo_step_kernel<DJ_TIO, double, double, DJ_MAXC, DJ_QUAD != 0, NW>), dim3(grid), dim3(64 * NW), 0, (hipStream_t)stream, A);
#ifdef DJ_ONLY_STEP      // register-allocation experiments (tools/step_only.sh)
        return (int)hipGetLastError();
#endif
#if DJ_QUAD != 0 && DJ_REFINE && !DJ_SS
        if (A.flag != nullptr) hipLaunchKernelGGL((dojo_stepp_kernel<DJ_TIO, double, double, DJ_MAXC, true, NW>), dim3(grid), dim3(64 * NW), 0, (hipStream_t)stream, A);
#endif
        if (mid_event) (void)hipEventRecord((hipEvent_t)mid_event, (hipStream_t)stream);
    }
#if DJ_QUAD == 1
    if ((phases & 4) && A.G.iter_cap > 0 && A.cont_count != nullptr) {      // the unfinished solves go on with their line-search trials side by side
        constexpr int R = cont_replicas<DJ_TIO, double, DJ_MAXC>();
        static_assert(R >= 2, "no room for two replicas");
        hipLaunchKernelGGL((dojo_stepc_kernel<DJ_TIO, double, double, DJ_MAXC, R>), dim3(grid), dim3(64 * R), 0, (hipStream_t)stream, A);
    }
#endif
#if DJ_LINEAR || (DJ_SS && !DJ_MLIM)      // forward-only builds (the general builds carry the body-body contact code too, and an IFT kernel: the host refuses gradients per mechanism)
    return (int)hipGetLastError();
#else
    if (phases & 2) {
        hipLaunchKernelGGL((dojo_grad_kernel<DJ_TIO, double, double, DJ_MAXC, DJ_QUAD != 0, NW>), dim3(grid), dim3(64 * NW), 0, (hipStream_t)stream, A);
#if DJ_QUAD != 0 && DJ_REFINE
        if (A.flag != nullptr) hipLaunchKernelGGL((dojo_gradp_kernel<DJ_TIO, double, double, DJ_MAXC, true, NW>), dim3(grid), dim3(64 * NW), 0, (hipStream_t)stream, A);
#endif
    }
#if DJ_QUAD == 1
    if ((phases & 8) && A.cont_count != nullptr)
        hipLaunchKernelGGL((dojo_gradc_kernel<DJ_TIO, double, double, DJ_MAXC>), dim3(grid), dim3(64), 0, (hipStream_t)stream, A);
#endif
    return (int)hipGetLastError();
#endif
}

#if DJ_QUAD != 0 && !DJ_LINEAR && !DJ_SS && !DJ_MLIM
// the contact-data IFT kernel alone (the step kernel of the same inputs must have run with its hand-off enabled)
extern "C" int DJ_CLAUNCHER(const void* args, int grid, void* stream) {
    const dj::KernelArgs<DJ_TIO, double>& A = *(const dj::KernelArgs<DJ_TIO, double>*)args;
    constexpr int NW = DJ_QUAD == 2 ? 2 : 1;
    hipLaunchKernelGGL((dojo_cgrad_kernel<DJ_TIO, double, double, DJ_MAXC, true, NW>), dim3(grid), dim3(64 * NW), 0, (hipStream_t)stream, A);
    return (int)hipGetLastError();
}
#endif
