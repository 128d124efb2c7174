// dojo_coords.hpp -- minimal <-> maximal coordinate maps of one joint, written once for any scalar type:
// double (the value maps) and Dual<N> (forward-mode derivatives: the chain-rule Jacobians of get_minimal_gradients!).
//
//   minimal_to_maximal  src/mechanism/state.jl:9-22  + set_minimal_coordinates/velocities!  src/joints/minimal.jl:160-232
//   maximal_to_minimal  src/mechanism/state.jl:44-66 + translational/minimal.jl:56-113, rotational/minimal.jl:62-118
//   minimal_to_maximal_jacobian / maximal_to_minimal_jacobian  src/gradients/state.jl:9-56,136-181 (attjac convention:
//   a unit quaternion q is perturbed as q (x) (1, phi), and d(q_out) is read as phi_out = V (q_out^-1 (x) dq_out))
#pragma once
#include "dojo_math.hpp"
#include "dojo_device.hpp"

namespace dj {
namespace coords {

// ---- forward-mode dual numbers: value + N directional derivatives ----
template <int N>
struct Dual {
    double v; double d[N];
    DJ_HD Dual() : v(0) { for (int i = 0; i < N; ++i) d[i] = 0; }
    DJ_HD Dual(double a) : v(a) { for (int i = 0; i < N; ++i) d[i] = 0; }
    DJ_HD Dual(int a) : v((double)a) { for (int i = 0; i < N; ++i) d[i] = 0; }
    DJ_HD static Dual seed(double a, int k) { Dual r(a); r.d[k] = 1.0; return r; }
};
template <int N> DJ_HD Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> DJ_HD Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> DJ_HD Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> DJ_HD Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> DJ_HD Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; double ib = 1.0 / b.v; r.v = a.v * ib; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
template <int N> DJ_HD Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> DJ_HD Dual<N> operator+(double b, const Dual<N>& a) { Dual<N> r = a; r.v += b; return r; }
template <int N> DJ_HD Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> DJ_HD Dual<N> operator-(double b, const Dual<N>& a) { Dual<N> r = -a; r.v += b; return r; }
template <int N> DJ_HD Dual<N> operator*(const Dual<N>& a, double b) { Dual<N> r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }
template <int N> DJ_HD Dual<N> operator*(double b, const Dual<N>& a) { return a * b; }
template <int N> DJ_HD Dual<N> operator*(int b, const Dual<N>& a) { return a * (double)b; }
template <int N> DJ_HD Dual<N> operator*(const Dual<N>& a, int b) { return a * (double)b; }
template <int N> DJ_HD Dual<N> operator/(const Dual<N>& a, double b) { return a * (1.0 / b); }
template <int N> DJ_HD Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }
template <int N> DJ_HD Dual<N>& operator+=(Dual<N>& a, const Dual<N>& b) { a = a + b; return a; }
template <int N> DJ_HD Dual<N>& operator-=(Dual<N>& a, const Dual<N>& b) { a = a - b; return a; }
template <int N> DJ_HD Dual<N>& operator*=(Dual<N>& a, const Dual<N>& b) { a = a * b; return a; }
template <int N> DJ_HD Dual<N>& operator*=(Dual<N>& a, double b) { a = a * b; return a; }
template <int N> DJ_HD bool operator>(const Dual<N>& a, const Dual<N>& b) { return a.v > b.v; }
template <int N> DJ_HD bool operator<(const Dual<N>& a, const Dual<N>& b) { return a.v < b.v; }
template <int N> DJ_HD bool operator>(const Dual<N>& a, double b) { return a.v > b; }
template <int N> DJ_HD bool operator<(const Dual<N>& a, double b) { return a.v < b; }
template <int N> DJ_HD Dual<N> dsqrt(const Dual<N>& a) { Dual<N> r; r.v = sqrt(a.v); double k = 0.5 / r.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k; return r; }
template <int N> DJ_HD Dual<N> dsin(const Dual<N>& a) { Dual<N> r; r.v = sin(a.v); double k = cos(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k; return r; }
template <int N> DJ_HD Dual<N> dcos(const Dual<N>& a) { Dual<N> r; r.v = cos(a.v); double k = -sin(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k; return r; }
template <int N> DJ_HD Dual<N> datan(const Dual<N>& a) { Dual<N> r; r.v = atan(a.v); double k = 1.0 / (1.0 + a.v * a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k; return r; }
DJ_HD double dsqrt(double a) { return sqrt(a); }
DJ_HD double dsin(double a) { return sin(a); }
DJ_HD double dcos(double a) { return cos(a); }
DJ_HD double datan(double a) { return atan(a); }
DJ_HD double val(double a) { return a; }
template <int N> DJ_HD double val(const Dual<N>& a) { return a.v; }

// ---- small quaternion / vector helpers on any scalar S (constants of the joint are plain doubles) ----
template <class S> DJ_HD void qmulS(S* c, const S* a, const S* b) {
    c[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    c[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    c[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    c[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
template <class S> DJ_HD void qconjS(S* c, const S* a) { c[0] = a[0]; c[1] = -a[1]; c[2] = -a[2]; c[3] = -a[3]; }
// vector_rotate(v, q) = V (q (0,v) q^-1), q^-1 = conj / |q|^2   (rotate.jl:2-5)
template <class S> DJ_HD void vrotS(S* o, const S* v, const S* q) {
    S p[4] = {S(0.0), v[0], v[1], v[2]}, t[4], qc[4], r[4];
    qmulS(t, q, p); qconjS(qc, q); qmulS(r, t, qc);
    S n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    o[0] = r[1] / n2; o[1] = r[2] / n2; o[2] = r[3] / n2;
}
template <class S> DJ_HD void vrot_invS(S* o, const S* v, const S* q) { S qc[4]; qconjS(qc, q); vrotS(o, v, qc); }
// axis_angle_to_quaternion (axis_angle.jl:1-11); series around 0 so that derivatives exist there
template <class S> DJ_HD void aa2qS(S* q, const S* r) {
    S t2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    if (val(t2) > 1e-12) { S th = dsqrt(t2); S s = dsin(th * 0.5) / th; q[0] = dcos(th * 0.5); q[1] = s * r[0]; q[2] = s * r[1]; q[3] = s * r[2]; }
    else { S s = 0.5 - t2 * (1.0 / 48.0); q[0] = 1.0 - t2 * 0.125; q[1] = s * r[0]; q[2] = s * r[1]; q[3] = s * r[2]; }
}
// rotation_vector(q) = 4 atan(|m|) m / |m|, m = v / (1 + s)   (mrp.jl:61-64); series around the identity
template <class S> DJ_HD void rotvecS(S* r, const S* q) {
    S d = 1.0 / (q[0] + 1.0);
    S m[3] = {q[1] * d, q[2] * d, q[3] * d};
    S m2 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
    S f;
    if (val(m2) > 1e-12) { S mag = dsqrt(m2); f = 4.0 * datan(mag) / mag; } else f = 4.0 - m2 * (4.0 / 3.0);
    r[0] = f * m[0]; r[1] = f * m[1]; r[2] = f * m[2];
}
// next_orientation(q, w, dt) = q (x) [sqrt(4/dt^2 - w.w), w] dt/2   (integrator.jl:15)
template <class S> DJ_HD void next_qS(S* o, const S* q, const S* w, double dt) {
    S xi[4] = {dsqrt(4.0 / (dt * dt) - (w[0] * w[0] + w[1] * w[1] + w[2] * w[2])), w[0], w[1], w[2]};
    qmulS(o, q, xi);
    for (int i = 0; i < 4; ++i) o[i] = o[i] * (0.5 * dt);
}
template <class S> DJ_HD void mask_tS(S* o, const double* A, int n, const S* c) {       // o = A' c  (A: n rows of 3)
    o[0] = S(0.0); o[1] = S(0.0); o[2] = S(0.0);
    for (int i = 0; i < 3; ++i) if (i < n) for (int j = 0; j < 3; ++j) o[j] = o[j] + c[i] * A[3 * i + j];
}

template <class S> struct PoseVel { S x[3], v[3], q[4], w[3]; };

// child body state from the parent's state and the joint's minimal coordinates / velocities (minimal.jl:205-232)
template <class S>
DJ_HD void joint_min2max(PoseVel<S>& b, const NodeP<double>& P, double dt, const PoseVel<S>& a, const S* dx, const S* dth, const S* dv, const S* dw) {
    const int nt = P.nu_t, nr = P.nu_r;
    S qoff[4] = {S(P.qoff[0]), S(P.qoff[1]), S(P.qoff[2]), S(P.qoff[3])}, pb[3] = {S(P.pb[0]), S(P.pb[1]), S(P.pb[2])};
    S r[3], dq[4], t[4], e[3], u[3], s1[3], s2[3];
    mask_tS(r, P.Ar, nr, dth); aa2qS(dq, r);
    qmulS(t, a.q, qoff); qmulS(b.q, t, dq);
    mask_tS(e, P.At, nt, dx); for (int i = 0; i < 3; ++i) u[i] = e[i] + P.pa[i];
    vrotS(s1, u, a.q); vrotS(s2, pb, b.q);
    for (int i = 0; i < 3; ++i) b.x[i] = a.x[i] + s1[i] - s2[i];
    // previous configuration
    S xa1[3], qa1[4], nw[3] = {-a.w[0], -a.w[1], -a.w[2]}, dx1[3], dwt[3], rw[3], dqw[4], dqwc[4], dq1[4], qb1[4], xb1[3];
    for (int i = 0; i < 3; ++i) { xa1[i] = a.x[i] - a.v[i] * dt; dx1[i] = dx[i] - dv[i] * dt; dwt[i] = dw[i] * dt; }
    next_qS(qa1, a.q, nw, dt);
    mask_tS(rw, P.Ar, nr, dwt); aa2qS(dqw, rw); qconjS(dqwc, dqw);          // unit quaternion: inverse = conjugate
    qmulS(dq1, dq, dqwc);
    qmulS(t, qa1, qoff); qmulS(qb1, t, dq1);
    mask_tS(e, P.At, nt, dx1); for (int i = 0; i < 3; ++i) u[i] = e[i] + P.pa[i];
    vrotS(s1, u, qa1); vrotS(s2, pb, qb1);
    for (int i = 0; i < 3; ++i) xb1[i] = xa1[i] + s1[i] - s2[i];
    S qc[4], qd[4]; qconjS(qc, qb1); qmulS(qd, qc, b.q);                       // angular_velocity(q1, q2) = 2/dt V(q1' q2), integrator.jl:25-27
    for (int i = 0; i < 3; ++i) { b.v[i] = (b.x[i] - xb1[i]) / dt; b.w[i] = qd[1 + i] * (2.0 / dt); }
}

// the joint's minimal coordinates and velocities from the two body states (translational/rotational minimal.jl)
template <class S>
DJ_HD void joint_max2min(S* ct, S* cr, S* vt, S* vr, const NodeP<double>& P, double dt, const PoseVel<S>& a, const PoseVel<S>& b) {
    S qoff[4] = {S(P.qoff[0]), S(P.qoff[1]), S(P.qoff[2]), S(P.qoff[3])}, pa[3] = {S(P.pa[0]), S(P.pa[1]), S(P.pa[2])}, pb[3] = {S(P.pb[0]), S(P.pb[1]), S(P.pb[2])};
    S xa1[3], xb1[3], qa1[4], qb1[4], nwa[3] = {-a.w[0], -a.w[1], -a.w[2]}, nwb[3] = {-b.w[0], -b.w[1], -b.w[2]};
    for (int i = 0; i < 3; ++i) { xa1[i] = a.x[i] - a.v[i] * dt; xb1[i] = b.x[i] - b.v[i] * dt; }
    next_qS(qa1, a.q, nwa, dt); next_qS(qb1, b.q, nwb, dt);
    auto qinv_mul = [](S* o, const S* p, const S* q_) { S pc[4]; qconjS(pc, p); S n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3]; S t[4]; qmulS(t, pc, q_); for (int i = 0; i < 4; ++i) o[i] = t[i] / n2; };
    auto disp = [&](S* o, const S* xa_, const S* qa_, const S* xb_, const S* qb_) {
        S s1[3], s2[3], d[3];
        vrotS(s1, pb, qb_); vrotS(s2, pa, qa_);
        for (int i = 0; i < 3; ++i) d[i] = xb_[i] + s1[i] - (xa_[i] + s2[i]);
        vrot_invS(o, d, qa_);
    };
    S t[4], q[4], q1[4], d2[3], d1[3], rv[3], qd[4], rvd[3];
    qinv_mul(t, a.q, b.q); qinv_mul(q, qoff, t);                            // qoff^-1 (x) qa^-1 (x) qb
    qinv_mul(t, qa1, qb1); qinv_mul(q1, qoff, t);
    disp(d2, a.x, a.q, b.x, b.q); disp(d1, xa1, qa1, xb1, qb1);
    rotvecS(rv, q);
    qinv_mul(qd, q1, q); rotvecS(rvd, qd);
    for (int i = 0; i < 3; ++i) {
        S c = S(0.0), v = S(0.0), c2 = S(0.0), v2 = S(0.0);
        for (int j = 0; j < 3; ++j) { c = c + d2[j] * P.At[3 * i + j]; v = v + (d2[j] - d1[j]) * P.At[3 * i + j]; c2 = c2 + rv[j] * P.Ar[3 * i + j]; v2 = v2 + rvd[j] * P.Ar[3 * i + j]; }
        ct[i] = c; vt[i] = v / dt; cr[i] = c2; vr[i] = v2 / dt;
    }
}

} // namespace coords
} // namespace dj
