// oracle_math.hpp -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the elementary algebra Dojo.jl's hot path is written in:
// small dense matrices (StaticArrays in the reference), Hamilton quaternions
// (Quaternions.jl) and the quaternion matrix helpers of
//   /root/reference/src/orientation/quaternion.jl:13-223
//   /root/reference/src/orientation/mapping.jl:1-8
//   /root/reference/src/orientation/rotate.jl:1-39
//   /root/reference/src/orientation/mrp.jl:1-80
//   /root/reference/src/orientation/axis_angle.jl:1-15
//   /root/reference/src/utilities/normalize.jl:1-28
// Nothing in the shipped product (dojo.jl_amd/) includes this file.
#pragma once
#include <cmath>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <initializer_list>

namespace orc {

// ---------------------------------------------------------------------------
// Small dense matrix with inline storage (no heap), dynamic shape.
// Largest local object on the path: joint x body-data block, 9 x 19 = 171 for a joint with one limited coordinate, 24 x 19 = 456 for a
// joint with limits on all six coordinates (src/joints/limits.jl with Nb½ = 3 on both halves).  The capacity is CHECKED (the checker is
// built with -DNDEBUG: an assert would let a larger block run over the stack).
// ---------------------------------------------------------------------------
template <class T>
struct SM {
    static constexpr int CAP = 512;
    int r = 0, c = 0;
    T a[CAP];
    SM() {}
    SM(int r_, int c_) : r(r_), c(c_) { if (r * c > CAP) { std::fprintf(stderr, "oracle: a %d x %d block exceeds SM::CAP\n", r, c); std::abort(); } for (int i = 0; i < r * c; ++i) a[i] = T(0); }
    SM(const SM& o) : r(o.r), c(o.c) { for (int i = 0; i < r * c; ++i) a[i] = o.a[i]; }
    SM& operator=(const SM& o) { r = o.r; c = o.c; for (int i = 0; i < r * c; ++i) a[i] = o.a[i]; return *this; }
    // row-major literal
    SM(int r_, int c_, std::initializer_list<T> v) : r(r_), c(c_) {
        assert((int)v.size() == r * c); int i = 0; for (T x : v) a[i++] = x;
    }
    T& operator()(int i, int j) { return a[i * c + j]; }
    const T& operator()(int i, int j) const { return a[i * c + j]; }
    T& operator[](int i) { return a[i]; }               // vector access (column vector r x 1 or flat)
    const T& operator[](int i) const { return a[i]; }
    int size() const { return r * c; }
    static SM eye(int n) { SM m(n, n); for (int i = 0; i < n; ++i) m(i, i) = T(1); return m; }
    static SM vec(std::initializer_list<T> v) { SM m((int)v.size(), 1); int i = 0; for (T x : v) m.a[i++] = x; return m; }
    SM t() const { SM m(c, r); for (int i = 0; i < r; ++i) for (int j = 0; j < c; ++j) m(j, i) = (*this)(i, j); return m; }
    SM block(int i0, int j0, int nr, int nc) const {
        SM m(nr, nc); for (int i = 0; i < nr; ++i) for (int j = 0; j < nc; ++j) m(i, j) = (*this)(i0 + i, j0 + j); return m;
    }
    SM rows(int i0, int nr) const { return block(i0, 0, nr, c); }
    SM cols(int j0, int nc) const { return block(0, j0, r, nc); }
    void set_block(int i0, int j0, const SM& b) { for (int i = 0; i < b.r; ++i) for (int j = 0; j < b.c; ++j) (*this)(i0 + i, j0 + j) = b(i, j); }
    T norm_inf() const { T m = 0; for (int i = 0; i < r * c; ++i) m = std::fmax(m, std::fabs(a[i])); return m; }
};

template <class T> SM<T> operator*(const SM<T>& A, const SM<T>& B) {
    assert(A.c == B.r); SM<T> C(A.r, B.c);
    for (int i = 0; i < A.r; ++i) for (int k = 0; k < A.c; ++k) { T aik = A(i, k); for (int j = 0; j < B.c; ++j) C(i, j) += aik * B(k, j); }
    return C;
}
template <class T> SM<T> operator+(const SM<T>& A, const SM<T>& B) { assert(A.r == B.r && A.c == B.c); SM<T> C(A.r, A.c); for (int i = 0; i < A.size(); ++i) C.a[i] = A.a[i] + B.a[i]; return C; }
template <class T> SM<T> operator-(const SM<T>& A, const SM<T>& B) { assert(A.r == B.r && A.c == B.c); SM<T> C(A.r, A.c); for (int i = 0; i < A.size(); ++i) C.a[i] = A.a[i] - B.a[i]; return C; }
template <class T> SM<T> operator-(const SM<T>& A) { SM<T> C(A.r, A.c); for (int i = 0; i < A.size(); ++i) C.a[i] = -A.a[i]; return C; }
template <class T> SM<T> operator*(T s, const SM<T>& A) { SM<T> C(A.r, A.c); for (int i = 0; i < A.size(); ++i) C.a[i] = s * A.a[i]; return C; }
template <class T> SM<T> operator*(const SM<T>& A, T s) { return s * A; }
template <class T> SM<T>& operator+=(SM<T>& A, const SM<T>& B) { assert(A.r == B.r && A.c == B.c); for (int i = 0; i < A.size(); ++i) A.a[i] += B.a[i]; return A; }
template <class T> SM<T>& operator-=(SM<T>& A, const SM<T>& B) { assert(A.r == B.r && A.c == B.c); for (int i = 0; i < A.size(); ++i) A.a[i] -= B.a[i]; return A; }
template <class T> SM<T> hcat(const SM<T>& A, const SM<T>& B) {
    if (A.c == 0 && A.r == 0) return B; if (B.c == 0 && B.r == 0) return A;
    assert(A.r == B.r); SM<T> C(A.r, A.c + B.c); C.set_block(0, 0, A); C.set_block(0, A.c, B); return C;
}
template <class T> SM<T> vcat(const SM<T>& A, const SM<T>& B) {
    if (A.r == 0) { if (A.c == 0 || A.c == B.c) return B; }
    if (B.r == 0) { if (B.c == 0 || A.c == B.c) return A; }
    assert(A.c == B.c); SM<T> C(A.r + B.r, A.c); C.set_block(0, 0, A); C.set_block(A.r, 0, B); return C;
}
template <class T> T dot(const SM<T>& a, const SM<T>& b) { assert(a.size() == b.size()); T s = 0; for (int i = 0; i < a.size(); ++i) s += a.a[i] * b.a[i]; return s; }
template <class T> T norm2(const SM<T>& a) { return std::sqrt(dot(a, a)); }
template <class T> SM<T> zeros(int r, int c) { return SM<T>(r, c); }
template <class T> SM<T> diag3(T a, T b, T c) { SM<T> m(3, 3); m(0, 0) = a; m(1, 1) = b; m(2, 2) = c; return m; }

// ---------------------------------------------------------------------------
// Quaternion (Hamilton product, Quaternions.jl semantics)
// ---------------------------------------------------------------------------
template <class T>
struct Quat {
    T s = 1, v1 = 0, v2 = 0, v3 = 0;
    Quat() {}
    Quat(T s_, T a, T b, T c) : s(s_), v1(a), v2(b), v3(c) {}
};
// Quaternions.Quaternion(v::AbstractVector) = Quaternion(0, v...)   quaternion.jl:1
template <class T> Quat<T> pure(const SM<T>& v) { return Quat<T>(0, v[0], v[1], v[2]); }
template <class T> SM<T> vector(const Quat<T>& q) { return SM<T>::vec({q.s, q.v1, q.v2, q.v3}); }   // quaternion.jl:13
template <class T> Quat<T> qfromvec(const SM<T>& v) { return Quat<T>(v[0], v[1], v[2], v[3]); }
template <class T> Quat<T> operator*(const Quat<T>& a, const Quat<T>& b) {
    return Quat<T>(a.s * b.s - a.v1 * b.v1 - a.v2 * b.v2 - a.v3 * b.v3,
                   a.s * b.v1 + a.v1 * b.s + a.v2 * b.v3 - a.v3 * b.v2,
                   a.s * b.v2 - a.v1 * b.v3 + a.v2 * b.s + a.v3 * b.v1,
                   a.s * b.v3 + a.v1 * b.v2 - a.v2 * b.v1 + a.v3 * b.s);
}
template <class T> Quat<T> operator*(const Quat<T>& a, T s) { return Quat<T>(a.s * s, a.v1 * s, a.v2 * s, a.v3 * s); }
template <class T> Quat<T> operator/(const Quat<T>& a, T s) { return Quat<T>(a.s / s, a.v1 / s, a.v2 / s, a.v3 / s); }
template <class T> Quat<T> conj(const Quat<T>& q) { return Quat<T>(q.s, -q.v1, -q.v2, -q.v3); }
template <class T> T abs2(const Quat<T>& q) { return q.s * q.s + q.v1 * q.v1 + q.v2 * q.v2 + q.v3 * q.v3; }
template <class T> Quat<T> inv(const Quat<T>& q) { return conj(q) / abs2(q); }   // Quaternions.jl inv

// quaternion.jl:16-129
template <class T> SM<T> Lmat(const Quat<T>& q) {
    return SM<T>(4, 4, {q.s, -q.v1, -q.v2, -q.v3,  q.v1, q.s, -q.v3, q.v2,  q.v2, q.v3, q.s, -q.v1,  q.v3, -q.v2, q.v1, q.s});
}
template <class T> SM<T> Rmat(const Quat<T>& q) {
    return SM<T>(4, 4, {q.s, -q.v1, -q.v2, -q.v3,  q.v1, q.s, q.v3, -q.v2,  q.v2, -q.v3, q.s, q.v1,  q.v3, q.v2, -q.v1, q.s});
}
template <class T> SM<T> LTmat(const Quat<T>& q) { return Lmat(q).t(); }
template <class T> SM<T> RTmat(const Quat<T>& q) { return Rmat(q).t(); }
template <class T> SM<T> Tmat() { return SM<T>(4, 4, {1, 0, 0, 0,  0, -1, 0, 0,  0, 0, -1, 0,  0, 0, 0, -1}); }
template <class T> SM<T> Vmat() { return SM<T>(3, 4, {0, 1, 0, 0,  0, 0, 1, 0,  0, 0, 0, 1}); }
template <class T> SM<T> VTmat() { return SM<T>(4, 3, {0, 0, 0,  1, 0, 0,  0, 1, 0,  0, 0, 1}); }
template <class T> SM<T> Vmat(const Quat<T>& q) { return SM<T>::vec({q.v1, q.v2, q.v3}); }
template <class T> SM<T> VLmat(const Quat<T>& q) {
    return SM<T>(3, 4, {q.v1, q.s, -q.v3, q.v2,  q.v2, q.v3, q.s, -q.v1,  q.v3, -q.v2, q.v1, q.s});
}
template <class T> SM<T> VLTmat(const Quat<T>& q) {   // VLᵀmat
    return SM<T>(3, 4, {-q.v1, q.s, q.v3, -q.v2,  -q.v2, -q.v3, q.s, q.v1,  -q.v3, q.v2, -q.v1, q.s});
}
template <class T> SM<T> VRmat(const Quat<T>& q) {
    return SM<T>(3, 4, {q.v1, q.s, q.v3, -q.v2,  q.v2, -q.v3, q.s, q.v1,  q.v3, q.v2, -q.v1, q.s});
}
template <class T> SM<T> VRTmat(const Quat<T>& q) {   // VRᵀmat
    return SM<T>(3, 4, {-q.v1, q.s, -q.v3, q.v2,  -q.v2, q.v3, q.s, -q.v1,  -q.v3, -q.v2, q.v1, q.s});
}
template <class T> SM<T> LVTmat(const Quat<T>& q) {   // LVᵀmat
    return SM<T>(4, 3, {-q.v1, -q.v2, -q.v3,  q.s, -q.v3, q.v2,  q.v3, q.s, -q.v1,  -q.v2, q.v1, q.s});
}
template <class T> SM<T> LTVTmat(const Quat<T>& q) {  // LᵀVᵀmat
    return SM<T>(4, 3, {q.v1, q.v2, q.v3,  q.s, q.v3, -q.v2,  -q.v3, q.s, q.v1,  q.v2, -q.v1, q.s});
}
template <class T> SM<T> RVTmat(const Quat<T>& q) {   // RVᵀmat
    return SM<T>(4, 3, {-q.v1, -q.v2, -q.v3,  q.s, q.v3, -q.v2,  -q.v3, q.s, q.v1,  q.v2, -q.v1, q.s});
}
template <class T> SM<T> RTVTmat(const Quat<T>& q) {  // RᵀVᵀmat
    return SM<T>(4, 3, {q.v1, q.v2, q.v3,  q.s, -q.v3, q.v2,  q.v3, q.s, -q.v1,  -q.v2, q.v1, q.s});
}

// matrix-vector product Jacobians, quaternion.jl:134-214
template <class T> SM<T> dVLmat_dq(const SM<T>& p) {     // ∂VLmat∂q, p in R3
    return SM<T>(4, 4, {0, p[0], p[1], p[2],  p[0], 0, p[2], -p[1],  p[1], -p[2], 0, p[0],  p[2], p[1], -p[0], 0});
}
template <class T> SM<T> dLVTmat_dq(const SM<T>& p) {    // ∂LVᵀmat∂q, p in R3
    return SM<T>(4, 4, {0, -p[0], -p[1], -p[2],  p[0], 0, p[2], -p[1],  p[1], -p[2], 0, p[0],  p[2], p[1], -p[0], 0});
}
template <class T> SM<T> dVLTmat_dq(const SM<T>& p) {    // ∂VLᵀmat∂q, p in R4
    return SM<T>(3, 4, {p[1], -p[0], -p[3], p[2],  p[2], p[3], -p[0], -p[1],  p[3], -p[2], p[1], -p[0]});
}
template <class T> SM<T> dLTVTmat_dq(const SM<T>& p) {   // ∂LᵀVᵀmat∂q, p in R3
    return SM<T>(4, 4, {0, p[0], p[1], p[2],  p[0], 0, -p[2], p[1],  p[1], p[2], 0, -p[0],  p[2], -p[1], p[0], 0});
}
template <class T> SM<T> dVRmat_dq(const SM<T>& p) {     // ∂VRmat∂q, p in R4
    return SM<T>(3, 4, {p[1], p[0], -p[3], p[2],  p[2], p[3], p[0], -p[1],  p[3], -p[2], p[1], p[0]});
}
template <class T> SM<T> dRTVTmat_dq(const SM<T>& p) {   // ∂RᵀVᵀmat∂q, p in R4 (as written: 3x4)
    return SM<T>(3, 4, {p[1], p[0], p[3], -p[2],  p[2], -p[3], p[0], p[1],  p[3], p[2], -p[1], p[0]});
}
template <class T> SM<T> dVRTmat_dq(const SM<T>& p) {    // ∂VRᵀmat∂q, p in R4
    return SM<T>(3, 4, {p[1], -p[0], p[3], -p[2],  p[2], -p[3], -p[0], p[1],  p[3], p[2], -p[1], -p[0]});
}
template <class T> SM<T> dRTmat_dq(const SM<T>& p) {     // ∂Rᵀmat∂q, p in R4
    return SM<T>(4, 4, {p[0], p[1], p[2], p[3],  p[1], -p[0], p[3], -p[2],  p[2], -p[3], -p[0], p[1],  p[3], p[2], -p[1], -p[0]});
}
template <class T> SM<T> dLmat_dq(const SM<T>& p) {      // ∂Lmat∂q, p in R4
    return SM<T>(4, 4, {p[0], -p[1], -p[2], -p[3],  p[1], p[0], p[3], -p[2],  p[2], -p[3], p[0], p[1],  p[3], p[2], -p[1], p[0]});
}
template <class T> SM<T> skew(const SM<T>& p) {
    return SM<T>(3, 3, {0, -p[2], p[1],  p[2], 0, -p[0],  -p[1], p[0], 0});
}
template <class T> SM<T> dskew_dp(const SM<T>& l) { return skew(-l); }   // ∂skew∂p(λ) = skew(-λ)

// mapping.jl:1-8
template <class T> Quat<T> quaternion_map(const SM<T>& w, T dt) {
    return Quat<T>(std::sqrt(T(4) / (dt * dt) - dot(w, w)), w[0], w[1], w[2]);
}
template <class T> SM<T> quaternion_map_jacobian(const SM<T>& w, T dt) {
    T msq = -std::sqrt(T(4) / (dt * dt) - dot(w, w));
    SM<T> m(4, 3);
    for (int j = 0; j < 3; ++j) m(0, j) = w[j] / msq;
    for (int j = 0; j < 3; ++j) m(1 + j, j) = T(1);
    return m;
}

// rotate.jl:2-39
template <class T> Quat<T> quaternion_rotate(const Quat<T>& q1, const Quat<T>& q2) { return q2 * q1 * inv(q2); }   // q2 * q1 / q2
template <class T> SM<T> vector_rotate(const SM<T>& v, const Quat<T>& q) { return Vmat(quaternion_rotate(pure(v), q)); }
template <class T> SM<T> dvector_rotate_dq(const SM<T>& p, const Quat<T>& q) {   // ∂vector_rotate∂q
    return VLmat(q) * Lmat(pure(p)) * Tmat<T>() + VRTmat(q) * Rmat(pure(p));
}
template <class T> SM<T> rotation_matrix(const Quat<T>& q) { return VRTmat(q) * LVTmat(q); }
template <class T> SM<T> drotation_matrix_dq(const Quat<T>& q, const SM<T>& p) {   // ∂rotation_matrix∂q(q, p): 3x4
    return dVRTmat_dq(LVTmat(q) * p) + VRTmat(q) * dLVTmat_dq(p);
}
template <class T> SM<T> drotation_matrix_inv_dq(const Quat<T>& q, const SM<T>& p) {   // ∂rotation_matrix_inv∂q
    return drotation_matrix_dq(inv(q), p) * Tmat<T>();
}

// mrp.jl:1-80 (rotation vector = 4 atan(|mrp|) * axis)
template <class T> SM<T> mrp(const SM<T>& q) { T d = q[0] + T(1); return SM<T>::vec({q[1] / d, q[2] / d, q[3] / d}); }
template <class T> SM<T> dmrpdq(const SM<T>& q) {
    T s = q[0]; T d1 = T(1) / ((s + 1) * (s + 1)); T di = T(1) / (s + 1);
    return SM<T>(3, 4, {-q[1] * d1, di, 0, 0,  -q[2] * d1, 0, di, 0,  -q[3] * d1, 0, 0, di});
}
template <class T> SM<T> axis_of(const SM<T>& q) {
    SM<T> m = mrp(q); T mag = norm2(m);
    if (mag > 0) return (T(1) / mag) * m;
    return SM<T>::vec({1, 0, 0});
}
template <class T> T angle_of(const SM<T>& q) {
    SM<T> m = mrp(q); T mag = norm2(m);
    return mag > 0 ? T(4) * std::atan(mag) : T(0);
}
template <class T> SM<T> daxisdq(const SM<T>& q) {
    SM<T> m = mrp(q); T n = norm2(m); SM<T> D = dmrpdq(q);
    SM<T> mh = (T(1) / n) * m;
    return (T(1) / n) * D - ((T(1) / (n * n)) * m) * mh.t() * D;
}
template <class T> SM<T> dangledq(const SM<T>& q) {
    SM<T> m = mrp(q); T n = norm2(m);
    return (T(4) / (T(1) + n * n)) * (((T(1) / n) * m).t() * dmrpdq(q));
}
template <class T> SM<T> rotation_vector(const Quat<T>& q) { SM<T> v = vector(q); return angle_of(v) * axis_of(v); }
template <class T> SM<T> drotation_vectordq(const Quat<T>& qq) {
    SM<T> q = vector(qq); T th = angle_of(q);
    if (th != T(0)) return axis_of(q) * dangledq(q) + th * daxisdq(q);
    return SM<T>(3, 4, {0, 2, 0, 0,  0, 0, 2, 0,  0, 0, 0, 2});
}
// axis_angle.jl:1-11
template <class T> Quat<T> axis_angle_to_quaternion(const SM<T>& x) {
    T th = norm2(x);
    if (th > 0) { T s = std::sin(T(0.5) * th) / th; return Quat<T>(std::cos(T(0.5) * th), s * x[0], s * x[1], s * x[2]); }
    return Quat<T>(1, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// dense LU with partial pivoting (stand-in for `solmat \ datamat` and for the
// result -- not the internals -- of GraphBasedSystems' ldu_factorization! /
// ldu_backsubstitution!, an exact direct solve)
// ---------------------------------------------------------------------------
template <class T>
struct DenseLU {
    int n = 0; std::vector<T> lu; std::vector<int> piv;
    std::vector<T> a0;            // the matrix itself, kept for solve_refined
    mutable std::vector<T> w_rhs, w_r; mutable std::vector<long double> w_acc;     // solve_refined's workspaces
    bool factor(const std::vector<T>& A, int n_) {
        n = n_; lu = A; a0 = A; piv.resize(n);
        for (int k = 0; k < n; ++k) {
            int p = k; T best = std::fabs(lu[k * n + k]);
            for (int i = k + 1; i < n; ++i) { T v = std::fabs(lu[i * n + k]); if (v > best) { best = v; p = i; } }
            piv[k] = p;
            if (p != k) for (int j = 0; j < n; ++j) std::swap(lu[k * n + j], lu[p * n + j]);
            T d = lu[k * n + k]; if (d == T(0)) return false;
            for (int i = k + 1; i < n; ++i) {
                T f = lu[i * n + k] / d; lu[i * n + k] = f;
                if (f != T(0)) { const T* rk = &lu[k * n]; T* ri = &lu[i * n]; for (int j = k + 1; j < n; ++j) ri[j] -= f * rk[j]; }
            }
        }
        return true;
    }
    // solve for nrhs right-hand sides stored row-major B[n][nrhs], in place
    void solve(T* B, int nrhs) const {
        for (int k = 0; k < n; ++k) if (piv[k] != k) for (int j = 0; j < nrhs; ++j) std::swap(B[k * nrhs + j], B[piv[k] * nrhs + j]);
        for (int i = 0; i < n; ++i) for (int k = 0; k < i; ++k) { T f = lu[i * n + k]; if (f != T(0)) for (int j = 0; j < nrhs; ++j) B[i * nrhs + j] -= f * B[k * nrhs + j]; }
        for (int i = n - 1; i >= 0; --i) {
            for (int k = i + 1; k < n; ++k) { T f = lu[i * n + k]; if (f != T(0)) for (int j = 0; j < nrhs; ++j) B[i * nrhs + j] -= f * B[k * nrhs + j]; }
            T d = lu[i * n + i]; for (int j = 0; j < nrhs; ++j) B[i * nrhs + j] /= d;
        }
    }
    // The same solve followed by `steps` rounds of iterative refinement with the residual accumulated in long double: the
    // checker's stand-in for exact arithmetic.  (The reference's LDU is an exact direct method; what the GPU path is compared
    // with should not carry the cond(A)·ε ~ 1e-6 forward error a single fp64 LU solve has on contact-rich steps, where
    // cond(A) reaches 1e10 -- the iterate path of a 20..40-iteration solve amplifies it to the parity bound.)
    void solve_refined(T* B, int nrhs, int steps) const {
        if (steps <= 0) { solve(B, nrhs); return; }
        // (workspaces live with the factorization: a 200 x 170 block is past malloc's mmap threshold, and an mmap / page-fault /
        //  munmap round per solve serializes the threads of a multi-threaded batch in the kernel)
        std::vector<T>& rhs = w_rhs; std::vector<T>& r = w_r; std::vector<long double>& acc = w_acc;
        rhs.assign(B, B + (size_t)n * nrhs); r.resize((size_t)n * nrhs); acc.resize(nrhs);
        solve(B, nrhs);
        for (int s = 0; s < steps; ++s) {
            for (int i = 0; i < n; ++i) {
                for (int j = 0; j < nrhs; ++j) acc[j] = (long double)rhs[(size_t)i * nrhs + j];
                const T* ai = &a0[(size_t)i * n];
                for (int k = 0; k < n; ++k) { const long double a = (long double)ai[k]; if (a != 0.0L) { const T* bk = &B[(size_t)k * nrhs]; for (int j = 0; j < nrhs; ++j) acc[j] -= a * (long double)bk[j]; } }
                for (int j = 0; j < nrhs; ++j) r[(size_t)i * nrhs + j] = (T)acc[j];
            }
            solve(r.data(), nrhs);
            for (size_t i = 0; i < r.size(); ++i) B[i] += r[i];
        }
    }
};

// ---------------------------------------------------------------------------
// Sparse LU WITHOUT pivoting on a row- and column-permuted matrix with a static fill pattern: what a block-sparse CPU
// implementation of the reference's solve costs (GraphBasedSystems' LDU eliminates the graph leaves -> root without
// pivoting; SURVEY.md §8d asks for this variant next to the dense one in bench.py's cpu_baseline).  Timing
// infrastructure: the checker itself stays on DenseLU::solve_refined.
//   A' = A[prow][:, pcol];  pattern = entries of A' that were ever non-zero, closed under elimination (symbolic phase,
//   redone when a later matrix has an entry outside it); numeric phase and solves visit pattern entries only.
// ---------------------------------------------------------------------------
template <class T>
struct SparseLU {
    int n = 0; std::vector<int> prow, pcol;           // A'(i, j) = A(prow[i], pcol[j])
    std::vector<char> pat;                            // n x n, closed under elimination
    std::vector<std::vector<int>> lrows, ucols;       // per pivot k: rows i > k with (i,k) in the pattern, columns j > k with (k,j)
    std::vector<T> w;                                 // n x n working copy: L (unit, multipliers) and U in place
    bool analyzed = false; long long flops_factor = 0, flops_solve = 0;     // per factorization / per right-hand side (multiply-adds counted as 2)
    void set_permutation(const std::vector<int>& pr, const std::vector<int>& pc) { prow = pr; pcol = pc; n = (int)pr.size(); analyzed = false; }
    void analyze(const std::vector<T>& A) {
        pat.assign((size_t)n * n, 0);
        for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) if (A[(size_t)prow[i] * n + pcol[j]] != T(0)) pat[(size_t)i * n + j] = 1; pat[(size_t)i * n + i] = 1; }
        lrows.assign(n, {}); ucols.assign(n, {}); flops_factor = 0;
        for (int k = 0; k < n; ++k) {
            for (int i = k + 1; i < n; ++i) if (pat[(size_t)i * n + k]) lrows[k].push_back(i);
            for (int j = k + 1; j < n; ++j) if (pat[(size_t)k * n + j]) ucols[k].push_back(j);
            for (int i : lrows[k]) for (int j : ucols[k]) pat[(size_t)i * n + j] = 1;              // fill
            flops_factor += 2LL * (long long)lrows[k].size() * (long long)ucols[k].size();
        }
        flops_solve = n; for (int k = 0; k < n; ++k) flops_solve += 2LL * (long long)(lrows[k].size() + ucols[k].size());
        w.assign((size_t)n * n, T(0)); analyzed = true;
    }
    bool factor(const std::vector<T>& A) {
        if (!analyzed) analyze(A);
        for (int i = 0; i < n; ++i) { const T* ai = &A[(size_t)prow[i] * n]; T* wi = &w[(size_t)i * n]; const char* pi = &pat[(size_t)i * n];
            for (int j = 0; j < n; ++j) { const T v = ai[pcol[j]]; if (v != T(0) && !pi[j]) { analyze_union(A); return factor(A); } wi[j] = pi[j] ? v : T(0); } }
        for (int k = 0; k < n; ++k) {
            const T piv = w[(size_t)k * n + k]; if (piv == T(0)) return false;
            const T ip = T(1) / piv; const T* wk = &w[(size_t)k * n];
            for (int i : lrows[k]) { T* wi = &w[(size_t)i * n]; const T f = wi[k] * ip; wi[k] = f; if (f != T(0)) for (int j : ucols[k]) wi[j] -= f * wk[j]; }
        }
        return true;
    }
    void analyze_union(const std::vector<T>& A) {        // a matrix with entries outside the pattern: widen it
        std::vector<char> old = pat; const bool had = analyzed;
        analyze(A);
        if (had) { for (size_t i = 0; i < old.size(); ++i) if (old[i]) pat[i] = 1;               // keep the old entries, close again
            std::vector<T> ones((size_t)n * n, T(0)); for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) if (pat[(size_t)i * n + j]) ones[(size_t)prow[i] * n + pcol[j]] = T(1);
            analyze(ones); }
    }
    // B[n][nrhs] row-major, rows in the ORIGINAL row order; the solution comes back in the original column (unknown) order
    mutable std::vector<T> w_y;                           // solve's workspace (kept: see DenseLU::solve_refined)
    void solve(T* B, int nrhs) const {
        std::vector<T>& y = w_y; y.resize((size_t)n * nrhs);
        for (int i = 0; i < n; ++i) std::memcpy(&y[(size_t)i * nrhs], &B[(size_t)prow[i] * nrhs], sizeof(T) * nrhs);
        for (int k = 0; k < n; ++k) { const T* yk = &y[(size_t)k * nrhs]; for (int i : lrows[k]) { const T f = w[(size_t)i * n + k]; if (f != T(0)) { T* yi = &y[(size_t)i * nrhs]; for (int j = 0; j < nrhs; ++j) yi[j] -= f * yk[j]; } } }
        for (int k = n - 1; k >= 0; --k) {
            T* yk = &y[(size_t)k * nrhs];
            for (int c : ucols[k]) { const T f = w[(size_t)k * n + c]; if (f != T(0)) { const T* yc = &y[(size_t)c * nrhs]; for (int j = 0; j < nrhs; ++j) yk[j] -= f * yc[j]; } }
            const T ip = T(1) / w[(size_t)k * n + k]; for (int j = 0; j < nrhs; ++j) yk[j] *= ip;
        }
        for (int i = 0; i < n; ++i) std::memcpy(&B[(size_t)pcol[i] * nrhs], &y[(size_t)i * nrhs], sizeof(T) * nrhs);
    }
};

} // namespace orc
