// dojo_math.hpp -- tiny fixed-size linear algebra + quaternion helpers for the device code.
// Everything is fully unrollable (compile-time extents) so per-lane data stays in VGPRs.
// Conventions follow the reference: Hamilton quaternions q = (s, v), rotation of a body-frame
// vector p into the world frame is R(q) p, attitude perturbation δq = q ⊗ (0, φ)
// (src/orientation/quaternion.jl `LVᵀmat`, so the rotation angle is 2|φ|).
#pragma once
#include <math.h>
#include <type_traits>
#include <utility>
#ifndef DJ_FAST_SQRT
#define DJ_FAST_SQRT 1
#endif

#if defined(__HIPCC__)
#define DJ_HD __host__ __device__ __forceinline__
#define DJ_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define DJ_HD inline
#define DJ_HD_NOINLINE inline
#endif

// Tell the compiler that a pointer taken from the kernel arguments points to GLOBAL memory: otherwise selects between
// such pointers decay to generic pointers and the accesses become FLAT instructions, which count on lgkmcnt as well
// as vmcnt and make every later LDS wait sit behind the HBM round trip.
#if defined(__HIP_DEVICE_COMPILE__)
template <class X> __device__ __forceinline__ X* dj_assume_global(X* p) {
    __builtin_assume(!__builtin_amdgcn_is_shared((const void*)p));
    __builtin_assume(!__builtin_amdgcn_is_private((const void*)p));
    return p;
}
#define DJ_GLOBAL_PTR(T, p) dj_assume_global<T>(p)
// a value the optimizer must take as unknown from here on (keeps cheap address arithmetic next to its use instead of hoisted into long-lived registers)
#define DJ_OPAQUE(x) __asm__ volatile("" : "+v"(x))
#else
#define DJ_GLOBAL_PTR(T, p) (p)
#define DJ_OPAQUE(x) ((void)0)
#endif

namespace dj {

// compile-time loop: f(std::integral_constant<int, I>()) for I = I0 .. I1-1.  Where a loop index must be a constant of the instruction
// (the lane of a DPP control), `#pragma unroll` is not enough: past its size threshold the unroller leaves a run-time loop behind.
template <int I0, int I1, class F> DJ_HD void static_for(F&& f) {
    if constexpr (I0 < I1) { f(std::integral_constant<int, I0>()); static_for<I0 + 1, I1>(f); }
}

template <class T> DJ_HD T tmax(T a, T b) { return a > b ? a : b; }
template <class T> DJ_HD T tmin(T a, T b) { return a < b ? a : b; }
template <class T> DJ_HD T tabs(T a) { return a < T(0) ? -a : a; }
// 1/a.  Device, fp64: v_rcp_f64 + two Newton steps (the IEEE quotient to the last bit on 2^20 random inputs,
// tools/ubench/rcp_test.hip; 5 instructions instead of the ~11 of the division expansion).  Host / emulator: 1/a.
DJ_HD double trcp(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(a); double e = fma(-a, r, 1.0); r = fma(r, e, r); e = fma(-a, r, 1.0); return fma(r, e, r);
#else
    return 1.0 / a;
#endif
}
DJ_HD float trcp(float a) { return 1.0f / a; }
DJ_HD float  tsqrt(float a)  { return sqrtf(a); }
// sqrt(a).  Device, fp64: v_rsq_f64 + the Goldschmidt iterations of the backend's own f64 sqrt expansion, without its range
// scaling and special-case selects (the arguments here are squared lengths and 4/Δt² − ω·ω; tools/ubench/sqrt_test.hip)
DJ_HD double tsqrt(double a) {
#if defined(__HIP_DEVICE_COMPILE__) && DJ_FAST_SQRT
    double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5); g = fma(g, r, g); h = fma(h, r, h);
    double d = fma(-g, g, a); g = fma(d, h, g);
    d = fma(-g, g, a); g = fma(d, h, g);
    return a == 0.0 ? 0.0 : g;
#else
    return sqrt(a);
#endif
}
DJ_HD float  tatan(float a)  { return atanf(a); }
// atan(a).  Device, fp64: the device library's algorithm (argument reduction to [0, 1] by a division, a degree-19 polynomial in v^2, pi/2 as a
// product of two doubles) operation for operation -- the results are its results bit for bit (tools/ubench/atan_table.hip) -- but the twenty
// coefficients come from a table in constant memory (scalar loads) and every multiply-add reads its coefficient as an SGPR operand.  The
// library form builds each coefficient in the VGPR pair a two-address v_fmac_f64 accumulates into (two v_mov_b32 with literals); inlined
// three times per evaluation of a joint with limits or a damper (rotvec, rotvec_jac_row), the compiler kept those forty moves out of the
// Newton loop and then SPILLED them: sixty dependent scratch loads per evaluation, each behind a full wait (round 6: 103 -> 8 spilled
// registers in the step kernel, 58 -> 12 scratch instructions in its Newton loop).  Measured next to it: the coefficients as s_mov_b32
// literal pairs at the point of use (no table) -- no gain over the library form (profiles/r06_b_ab.txt).
#ifndef DJ_ATAN_TABLE
#define DJ_ATAN_TABLE 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && DJ_ATAN_TABLE
static __constant__ unsigned long long dj_atan_bits[20] = {
    0x3EEBA404B5E68A13ull, 0xBF23E260BD3237F4ull, 0x3F4B2BB069EFB384ull, 0xBF67952DAF56DE9Bull, 0x3F7D6D43A595C56Full,
    0xBF8C6EA4A57D9582ull, 0x3F967E295F08B19Full, 0xBF9E9AE6FC27006Aull, 0x3FA2C15B5711927Aull, 0xBFA59976E82D3FF0ull,
    0x3FA82D5D6EF28734ull, 0xBFAAE5CE6A214619ull, 0x3FAE1BB48427B883ull, 0xBFB110E48B207F05ull, 0x3FB3B13657B87036ull,
    0xBFB745D119378E4Full, 0x3FBC71C717E1913Cull, 0xBFC2492492376B7Dull, 0x3FC99999999952CCull, 0xBFD5555555555523ull};
DJ_HD double tatan(double a) {
    const double v0 = fabs(a);
    const bool g = v0 > 1.0;
    const double v = g ? 1.0 / v0 : v0;
    const double t = v * v;
    // p = fma(t, p, c_i), c_i read from its SGPR pair by the instruction itself (left to the compiler, every coefficient is first copied
    // into the VGPR pair of a v_fmac_f64 -- the forty moves again)
    double p = __longlong_as_double((long long)dj_atan_bits[0]);
#pragma unroll
    for (int i = 1; i < 20; ++i) {
        const double c = __longlong_as_double((long long)dj_atan_bits[i]);
        double pn;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(pn) : "v"(t), "v"(p), "s"(c));
        p = pn;
    }
    const double r = fma(v, t * p, v);
    const double hi = fma(__longlong_as_double(0x3FEDD9AD336A0500ll), __longlong_as_double(0x3FFAF154EEB562D6ll), -r);
    return copysign(g ? hi : r, a);
}
#else
DJ_HD double tatan(double a) { return atan(a); }
#endif

// ---- 3-vectors -------------------------------------------------------------------------------
template <class T> DJ_HD void v3set(T* a, T x, T y, T z) { a[0] = x; a[1] = y; a[2] = z; }
template <class T> DJ_HD void v3cpy(T* a, const T* b) { a[0] = b[0]; a[1] = b[1]; a[2] = b[2]; }
template <class T> DJ_HD T    v3dot(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class T> DJ_HD void v3cross(T* c, const T* a, const T* b) {
    T x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    c[0] = x; c[1] = y; c[2] = z;
}
// ---- 3x3 (row-major) -------------------------------------------------------------------------
template <class T> DJ_HD void m3vec(T* y, const T* M, const T* x) {       // y = M x
    T a = M[0] * x[0] + M[1] * x[1] + M[2] * x[2], b = M[3] * x[0] + M[4] * x[1] + M[5] * x[2], c = M[6] * x[0] + M[7] * x[1] + M[8] * x[2];
    y[0] = a; y[1] = b; y[2] = c;
}
template <class T> DJ_HD void m3tvec(T* y, const T* M, const T* x) {      // y = Mᵀ x
    T a = M[0] * x[0] + M[3] * x[1] + M[6] * x[2], b = M[1] * x[0] + M[4] * x[1] + M[7] * x[2], c = M[2] * x[0] + M[5] * x[1] + M[8] * x[2];
    y[0] = a; y[1] = b; y[2] = c;
}
template <class T> DJ_HD void m3mul(T* C, const T* A, const T* B) {       // C = A B   (C must not alias)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
template <class T> DJ_HD void m3tmul(T* C, const T* A, const T* B) {      // C = Aᵀ B
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
template <class T> DJ_HD void m3mult(T* C, const T* A, const T* B) {      // C = A Bᵀ
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
template <class T> DJ_HD void m3skew(T* S, const T* p) {                  // [p]x
    S[0] = 0; S[1] = -p[2]; S[2] = p[1]; S[3] = p[2]; S[4] = 0; S[5] = -p[0]; S[6] = -p[1]; S[7] = p[0]; S[8] = 0;
}
// s I + [v]x
template <class T> DJ_HD void m3sIpskew(T* S, T s, const T* v) {
    S[0] = s; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = s; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = s;
}
// ---- quaternions (s, v1, v2, v3) -------------------------------------------------------------
template <class T> DJ_HD void qmul(T* c, const T* a, const T* b) {
    T s = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    T x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    T y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    T z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    c[0] = s; c[1] = x; c[2] = y; c[3] = z;
}
template <class T> DJ_HD void qconj(T* c, const T* a) { c[0] = a[0]; c[1] = -a[1]; c[2] = -a[2]; c[3] = -a[3]; }
// c = conj(a) ⊗ b
template <class T> DJ_HD void qcmul(T* c, const T* a, const T* b) { T ac[4]; qconj(ac, a); qmul(c, ac, b); }
// rotation_matrix(q) = VRᵀmat(q) LVᵀmat(q)  (src/orientation/rotate.jl:22); for unit q the usual rotation matrix
template <class T> DJ_HD void qrot(T* R, const T* q) {
    T s = q[0], x = q[1], y = q[2], z = q[3];
    T ss = s * s, xx = x * x, yy = y * y, zz = z * z;
    R[0] = ss + xx - yy - zz; R[1] = 2 * (x * y - s * z);  R[2] = 2 * (x * z + s * y);
    R[3] = 2 * (x * y + s * z);  R[4] = ss - xx + yy - zz; R[5] = 2 * (y * z - s * x);
    R[6] = 2 * (x * z - s * y);  R[7] = 2 * (y * z + s * x);  R[8] = ss - xx - yy + zz;
}
// quaternion_map(ω, Δt)·Δt/2 = ξ(ω): unit quaternion of the step (src/orientation/mapping.jl:1-3)
template <class T> DJ_HD void qstep(T* xi, const T* w, T dt, T* c_out) {
    const T idt = trcp(dt);
    T c = tsqrt(T(4) * idt * idt - v3dot(w, w));
    T h = dt * T(0.5);
    xi[0] = c * h; xi[1] = w[0] * h; xi[2] = w[1] * h; xi[3] = w[2] * h;
    *c_out = c;
}
// Φ(ω): δφ3 = Φ δω where q3 = q2 ⊗ ξ(ω), δq3 = q3 ⊗ (0, δφ3):  Φ = Δt²/4 (c I + ω ωᵀ / c − [ω]x)
// (= LVᵀmat(q3)ᵀ · rotational_integrator_jacobian_velocity(q2, ω, Δt), src/integrators/integrator.jl:64-66)
template <class T> DJ_HD void phi_of(T* P, const T* w, T c, T dt) {
    T k = dt * dt * T(0.25), ic = trcp(c);
    P[0] = k * (c + w[0] * w[0] * ic); P[1] = k * (w[0] * w[1] * ic + w[2]); P[2] = k * (w[0] * w[2] * ic - w[1]);
    P[3] = k * (w[1] * w[0] * ic - w[2]); P[4] = k * (c + w[1] * w[1] * ic); P[5] = k * (w[1] * w[2] * ic + w[0]);
    P[6] = k * (w[2] * w[0] * ic + w[1]); P[7] = k * (w[2] * w[1] * ic - w[0]); P[8] = k * (c + w[2] * w[2] * ic);
}
// rotation_vector(q) = 4 atan(|m|) m/|m|, m = v/(1+s)   (src/orientation/mrp.jl:61-64)
template <class T> DJ_HD void rotvec(T* r, const T* q) {
    T d = trcp(q[0] + T(1));
    T m[3] = {q[1] * d, q[2] * d, q[3] * d};
    T mag = tsqrt(v3dot(m, m));
    if (mag > T(0)) { T f = T(4) * tatan(mag) * trcp(mag); r[0] = f * m[0]; r[1] = f * m[1]; r[2] = f * m[2]; }
    else { r[0] = r[1] = r[2] = T(0); }
}
// aᵀ · drotation_vectordq(q): the 1x4 row  a·∂rotvec/∂q   (src/orientation/mrp.jl:66-78)
template <class T> DJ_HD void rotvec_jac_row(T* row, const T* a, const T* q) {
    T s1 = q[0] + T(1), di = trcp(s1), d1 = di * di;
    T m[3] = {q[1] * di, q[2] * di, q[3] * di};
    T n2 = v3dot(m, m);
    if (n2 > T(0)) {
        T n = tsqrt(n2), th = T(4) * tatan(n);
        // d rotvec / d m = (4/(1+n²)) m̂ m̂ᵀ + (θ/n)(I − m̂ m̂ᵀ)
        T am = v3dot(a, m) * trcp(n2);            // (a·m̂)/n ... times m gives (a·m̂) m̂
        T k1 = T(4) * trcp(T(1) + n2), k2 = th * trcp(n);
        T g[3];                              // g = aᵀ · d rotvec/d m
        for (int i = 0; i < 3; ++i) g[i] = k2 * a[i] + (k1 - k2) * am * m[i];
        // d m / d q = [ −v/(1+s)² | I/(1+s) ]
        row[0] = -(g[0] * q[1] + g[1] * q[2] + g[2] * q[3]) * d1;
        row[1] = g[0] * di; row[2] = g[1] * di; row[3] = g[2] * di;
    } else { row[0] = T(0); row[1] = T(2) * a[0]; row[2] = T(2) * a[1]; row[3] = T(2) * a[2]; }
}

// ---- generic small dense helpers (compile-time sizes, row-major) ------------------------------
template <int R, int K, int C, class T> DJ_HD void mm(T* O, const T* A, const T* B) {          // O(RxC) = A(RxK) B(KxC)
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < C; ++j) {
            T s = T(0);
#pragma unroll
            for (int k = 0; k < K; ++k) s += A[i * K + k] * B[k * C + j];
            O[i * C + j] = s;
        }
}
template <int R, int K, int C, class T> DJ_HD void mm_sub(T* O, const T* A, const T* B) {      // O -= A B
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < C; ++j) {
            T s = T(0);
#pragma unroll
            for (int k = 0; k < K; ++k) s += A[i * K + k] * B[k * C + j];
            O[i * C + j] -= s;
        }
}
template <int R, int C, class T> DJ_HD void mv(T* y, const T* A, const T* x) {                  // y = A x
#pragma unroll
    for (int i = 0; i < R; ++i) {
        T s = T(0);
#pragma unroll
        for (int j = 0; j < C; ++j) s += A[i * C + j] * x[j];
        y[i] = s;
    }
}
// in-place inverse of an NxN matrix by Gauss-Jordan WITHOUT pivoting (pivot order is fixed by the
// elimination order chosen on the host: body block first, then its parent joint's multipliers, so
// every pivot is a Schur complement of a well-conditioned block -- SURVEY.md §7 H3)
template <int N, class T> DJ_HD void gj_inverse(T* A) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        T ip = T(1) / A[k * N + k];
        A[k * N + k] = T(1);
#pragma unroll
        for (int j = 0; j < N; ++j) A[k * N + j] *= ip;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i == k) continue;
            T f = A[i * N + k];
            A[i * N + k] = T(0);
#pragma unroll
            for (int j = 0; j < N; ++j) A[i * N + j] -= f * A[k * N + j];
        }
    }
}

} // namespace dj
