// dojo_oracle.hpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement ("oracle") of Dojo.jl's per-timestep contact-implicit forward /
// backward path: variational integrator -> Mehrotra interior-point NCP solve ->
// implicit-function-theorem gradients.  Single environment, dense KKT, dense LU.
//
// The reference is 100 % Julia and Julia is not installed in the build container, so
// the reference itself cannot be executed here ("oracle/_ref" does not exist).  This
// file follows the reference function by function; every routine cites the
// /root/reference file:line it restates.
// PINNED against reference-COMPUTED numbers on one contact configuration: the ten 100-step
// Storage trajectories of the reference's own simulate! that it ships in
// examples/system_identification/data/datasets/synthetic_sphere.jld2 (one body, NonlinearContact
// + SphereHalfSpaceCollision, default SolverOptions): step!(row k) == row k+1 to 3.8e-14 over
// all 990 pairs, equal Newton iteration counts (tests/test_reference_sphere.py,
// tests/golden/reference_sphere.npz, tools/jld2_reader.py).  For joints, limits, springs /
// dampers and the IFT gradients no reference-computed numbers exist in the tree (PARITY
// UNPINNED there in the strict sense); they are pinned by re-stating the reference's
// own property tests (tests/test_oracle_*.py):
//   * test/jacobian.jl:1-64     FD(residual wrt solution) == -full_matrix(system)
//   * test/data.jl:82-126       FD(residual wrt data)·attjac == jacobian_data!
//   * test/behaviors.jl:21-55   box toss rests at z = 0.25, 1 N on 1 kg -> v = 0.5
//   * test/mechanism.jl:92-121  hovering block (force vs impulse input scaling)
//   * test/joint_limits.jl      pendulum settles on its limit
//   * test/momentum.jl          momentum conservation without gravity
// The block LDU of the un-vendored GraphBasedSystems package (compat 1.0/1.2, no
// Manifest) is NOT mimicked: it is an exact direct solve of M Δ = r, so the oracle
// solves the same dense system with partial-pivot LU.  At the LDU-internal level:
// PARITY UNPINNED; at the level of the solve result it is pinned by the tests above.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
#pragma once
#include <functional>
#include "counted.hpp"
#include "oracle_math.hpp"
#include "../include/dojo_hip.h"
#include <limits>
#include <string>
#include <cstdio>
#include <cstdlib>

namespace orc {

constexpr double REG = 1e-10;   // src/Dojo.jl:4

// ---------------------------------------------------------------------------
// State{T}  src/bodies/state.jl:25-69
// ---------------------------------------------------------------------------
template <class T>
struct State {
    using M = SM<T>;
    M x1 = M(3, 1); Quat<T> q1; M v15 = M(3, 1), w15 = M(3, 1);
    M x2 = M(3, 1); Quat<T> q2; M JF2 = M(3, 1), Jt2 = M(3, 1), Fext = M(3, 1), text = M(3, 1);
    M vsol[2] = {M(3, 1), M(3, 1)}; M wsol[2] = {M(3, 1), M(3, 1)};
    M d = M(6, 1); M D = M(6, 6);
};
template <class T> struct Body { T mass; SM<T> inertia; State<T> st; };

// Translational / Rotational half of a JointConstraint
// src/joints/translational/constructor.jl:18-67, src/joints/rotational/constructor.jl:18-63
template <class T>
struct Half {
    using M = SM<T>;
    bool is_rot = false;
    int nl = 0, nlim = 0;          // Nλ, Nb½
    M C, A;                        // constraint_mask (nl x 3), nullspace_mask (3-nl x 3)   joint.jl:56-64
    T spring = 0, damper = 0;
    M spring_offset, lo, hi;       // (3-nl), (nlim), (nlim)
    M input = M(3, 1);
    int Nb() const { return 2 * nlim; }
    int N() const { return nl + 2 * Nb(); }       // impulses_length
    int nu() const { return 3 - nl; }             // input_dimension
};
template <class T>
struct Joint {
    using M = SM<T>;
    int parent = -1, child = 0;   // body indices, -1 = origin
    bool spring = false, damper = false;
    M vp, vc;                      // translational.vertices
    Quat<T> qoff;                  // rotational.orientation_offset
    Half<T> tra, rot;
    M imp[2];                      // impulses[1], impulses[2] (current / candidate)
    int N() const { return tra.N() + rot.N(); }
    int nu() const { return tra.nu() + rot.nu(); }
};
template <class T>
struct Contact {
    using M = SM<T>;
    int body = 0;
    int model = 0;                 // 0: NonlinearContact (src/contacts/nonlinear.jl), 1: ImpactContact (src/contacts/impact.jl), 2: LinearContact (src/contacts/linear.jl)
    int nh() const { return model == 1 ? 1 : model == 2 ? 6 : 4; }     // N½: γ and s each (LinearContact: [γ ψ β1..β4])
    T mu = 0;                      // friction_coefficient
    M normal, tangent, origin, offset; T radius = 0;   // 1x3, 2x3, 3, 3
    // SphereSphereCollision (kind 1, sphere_sphere.jl:11-16): `body` = parent_id with (origin, radius) = (origin_parent, radius_parent),
    // body2 = child_id with (origin_c, radius_c)
    int kind = 0; int body2 = -1; M origin_c = M(3, 1); T radius_c = 0;
    M gam[2] = {M(4, 1), M(4, 1)}; // impulses       (γ)
    M s[2] = {M(4, 1), M(4, 1)};   // impulses_dual  (s)
};

struct Options {                    // src/solver/options.jl:16-26
    double rtol = 1e-6, btol = 1e-4, undercut = std::numeric_limits<double>::infinity(), no_progress_undercut = 10.0;
    int max_iter = 50, max_ls = 10, no_progress_max = 3;
};

// ---------------------------------------------------------------------------
// integrator  src/integrators/integrator.jl:14-67
// ---------------------------------------------------------------------------
template <class T> SM<T> next_position(const SM<T>& x2, const SM<T>& v25, T dt) { return x2 + dt * v25; }
template <class T> Quat<T> next_orientation(const Quat<T>& q2, const SM<T>& w25, T dt) { return q2 * quaternion_map(w25, dt) * (dt / T(2)); }
template <class T> SM<T> rotational_integrator_jacobian_velocity(const Quat<T>& q2, const SM<T>& w25, T dt) {
    return Lmat(q2) * quaternion_map_jacobian(w25, dt) * (dt / T(2));
}
template <class T> SM<T> rotational_integrator_jacobian_orientation(const Quat<T>& q2, const SM<T>& w25, T dt, bool attjac) {
    SM<T> M_ = Rmat(quaternion_map(w25, dt) * (dt / T(2)));
    if (attjac) M_ = M_ * LVTmat(q2);
    return M_;
}
template <class T> SM<T> integrator_jacobian_velocity(const Quat<T>& q2, const SM<T>& w25, T dt) {   // 7x6
    SM<T> J(7, 6);
    for (int i = 0; i < 3; ++i) J(i, i) = dt;
    J.set_block(3, 3, rotational_integrator_jacobian_velocity(q2, w25, dt));
    return J;
}
template <class T> SM<T> integrator_jacobian_configuration(const Quat<T>& q2, const SM<T>& w25, T dt, bool attjac) {   // 7x6 / 7x7
    SM<T> J(7, attjac ? 6 : 7);
    for (int i = 0; i < 3; ++i) J(i, i) = T(1);
    J.set_block(3, 3, rotational_integrator_jacobian_orientation(q2, w25, dt, attjac));
    return J;
}

template <class T>
struct Mechanism {
    using M = SM<T>;
    using Q = Quat<T>;
    std::vector<Body<T>> bodies;
    std::vector<Joint<T>> joints;
    std::vector<Contact<T>> contacts;
    State<T> origin;               // get_body(mechanism, 0): identity / zero state, src/bodies/origin.jl
    T dt = T(0.01), input_scaling = T(0.01);
    M gravity = M(3, 1);
    T mu = 0;                      // mechanism.μ (central-path parameter)
    Options opts;
    // dense system in node order [joints; bodies; contacts] (ids 1..Ne, Ne+1..Ne+Nb, ...)
    int n = 0;
    std::vector<int> joff, boff, coff;       // row/col offsets of each node
    std::vector<T> A, b, rcache;             // full_matrix(system), full_vector(system), residual_entries
    std::vector<T> A_solved;                 // matrix of the last set_entries! before any factorization (kept for gradients)
    int last_iters = 0;
    bool verbose = false; T last_alpha = 1;
    int refine_steps = 2;         // rounds of iterative refinement of every linear solve (DenseLU::solve_refined); 0 = plain LU, as timed by bench.py's cpu_baseline
    bool sparse_solver = false;   // timing variant: SparseLU (no pivoting, elimination order of the mechanism graph) instead of DenseLU
    mutable SparseLU<T> splu;
    mutable std::vector<T> w_Dm, w_R; mutable DenseLU<T> w_lu;      // get_maximal_gradients' workspaces
    bool excessive_w = false;

    // ---------------- construction from the C-POD topology ----------------
    explicit Mechanism(const DojoTopology& tp) {
        dt = T(tp.timestep); input_scaling = T(tp.input_scaling);
        for (int i = 0; i < 3; ++i) gravity[i] = T(tp.gravity[i]);
        for (int i = 0; i < tp.n_bodies; ++i) {
            Body<T> B; B.mass = T(tp.bodies[i].mass); B.inertia = M(3, 3);
            for (int k = 0; k < 9; ++k) B.inertia.a[k] = T(tp.bodies[i].inertia[k]);
            bodies.push_back(B);
        }
        for (int i = 0; i < tp.n_joints; ++i) {
            const DojoJoint& s = tp.joints[i]; Joint<T> J;
            J.parent = s.parent; J.child = s.child; J.spring = s.spring_on != 0; J.damper = s.damper_on != 0;
            J.vp = M(3, 1); J.vc = M(3, 1);
            for (int k = 0; k < 3; ++k) { J.vp[k] = T(s.vertex_parent[k]); J.vc[k] = T(s.vertex_child[k]); }
            J.qoff = Q(T(s.orientation_offset[0]), T(s.orientation_offset[1]), T(s.orientation_offset[2]), T(s.orientation_offset[3]));
            auto fill = [](Half<T>& h, const DojoJointHalf& g, bool rot) {
                h.is_rot = rot; h.nl = g.nl; h.nlim = g.nlim;
                h.C = M(g.nl, 3); h.A = M(3 - g.nl, 3);
                for (int r = 0; r < g.nl; ++r) for (int c = 0; c < 3; ++c) h.C(r, c) = T(g.cmask[3 * r + c]);
                for (int r = 0; r < 3 - g.nl; ++r) for (int c = 0; c < 3; ++c) h.A(r, c) = T(g.amask[3 * r + c]);
                h.spring = T(g.spring); h.damper = T(g.damper);
                h.spring_offset = M(3 - g.nl, 1); for (int r = 0; r < 3 - g.nl; ++r) h.spring_offset[r] = T(g.spring_offset[r]);
                h.lo = M(g.nlim, 1); h.hi = M(g.nlim, 1);
                for (int r = 0; r < g.nlim; ++r) { h.lo[r] = T(g.limit_lo[r]); h.hi[r] = T(g.limit_hi[r]); }
            };
            fill(J.tra, s.tra, false); fill(J.rot, s.rot, true);
            J.imp[0] = M(J.N(), 1); J.imp[1] = M(J.N(), 1);
            joints.push_back(J);
        }
        for (int i = 0; i < tp.n_contacts; ++i) {
            const DojoContact& s = tp.contacts[i]; Contact<T> c;
            c.body = s.body; c.model = s.model; c.mu = T(s.friction_coefficient); c.radius = T(s.radius);
            c.normal = M(1, 3); c.tangent = M(2, 3); c.origin = M(3, 1); c.offset = M(3, 1);
            for (int k = 0; k < 3; ++k) { c.normal.a[k] = T(s.normal[k]); c.origin[k] = T(s.origin[k]); c.offset[k] = T(s.offset[k]); }
            for (int k = 0; k < 6; ++k) c.tangent.a[k] = T(s.tangent[k]);
            c.kind = s.collision; c.body2 = s.collision == 1 ? s.child_body : -1; c.radius_c = T(s.child_radius);
            for (int k = 0; k < 3; ++k) c.origin_c[k] = T(s.child_origin[k]);
            contacts.push_back(c);
        }
        verbose = std::getenv("ORC_VERBOSE") != nullptr;
        if (const char* rs_ = std::getenv("ORC_REFINE")) refine_steps = std::atoi(rs_);
        int off = 0;
        for (auto& J : joints) { joff.push_back(off); off += J.N(); }
        for (size_t i = 0; i < bodies.size(); ++i) { boff.push_back(off); off += 6; }
        for (size_t i = 0; i < contacts.size(); ++i) { coff.push_back(off); off += 2 * contacts[i].nh(); }
        n = off; A.assign((size_t)n * n, T(0)); b.assign(n, T(0)); rcache.assign(n, T(0));
    }

    int nu() const { int s = 0; for (auto& J : joints) s += J.nu(); return s; }
    State<T>& bstate(int i) { return i < 0 ? origin : bodies[i].st; }
    const State<T>& bstate(int i) const { return i < 0 ? origin : bodies[i].st; }

    // next_configuration(state, timestep)   integrator.jl:20-21
    M x3(const State<T>& s) const { return next_position(s.x2, s.vsol[1], dt); }
    Q q3(const State<T>& s) const { return next_orientation(s.q2, s.wsol[1], dt); }

    // =====================================================================
    // JOINT HALVES
    // =====================================================================
    // displacement   translational/minimal.jl:4-12, rotational/minimal.jl:4-11
    M tra_displacement(const Joint<T>& J, const M& xa, const Q& qa, const M& xb, const Q& qb, bool rotate = true) const {
        M d = xb + vector_rotate(J.vc, qb) - (xa + vector_rotate(J.vp, qa));
        return rotate ? vector_rotate(d, inv(qa)) : d;
    }
    Q rot_displacement_q(const Joint<T>& J, const Q& qa, const Q& qb) const { return inv(J.qoff) * inv(qa) * qb; }

    // displacement_jacobian_configuration (raw: X 3x3, Q 3x4)
    // translational/minimal.jl:14-30, rotational/minimal.jl:27-38
    void disp_jac_raw(bool parent, const Joint<T>& J, const Half<T>& h, const M& xa, const Q& qa, const M& xb, const Q& qb, M& X, M& Qm) const {
        if (!h.is_rot) {
            if (parent) {
                M d = xb + vector_rotate(J.vc, qb) - (xa + vector_rotate(J.vp, qa));
                X = -rotation_matrix(inv(qa));
                Qm = (-rotation_matrix(inv(qa))) * drotation_matrix_dq(qa, J.vp);
                Qm += drotation_matrix_inv_dq(qa, d);
            } else {
                X = rotation_matrix(inv(qa));
                Qm = rotation_matrix(inv(qa)) * drotation_matrix_dq(qb, J.vc);
            }
        } else {
            X = M(3, 3);
            if (parent) Qm = Vmat<T>() * (LTmat(J.qoff) * Rmat(qb) * Tmat<T>());
            else        Qm = Vmat<T>() * (LTmat(J.qoff) * LTmat(qa));
        }
    }
    // displacement_jacobian_configuration(...; attjac)   joints/joint.jl:141-153
    void disp_jac(bool parent, const Joint<T>& J, const Half<T>& h, const M& xa, const Q& qa, const M& xb, const Q& qb, bool attjac, M& X, M& Qm) const {
        disp_jac_raw(parent, J, h, xa, qa, xb, qb, X, Qm);
        if (attjac) Qm = Qm * LVTmat(parent ? qa : qb);
    }
    // displacement_jacobian_configuration_unstable (vmat=false)   rotational/minimal.jl:13-25
    M rot_disp_jac_unstable_novmat(bool parent, const Joint<T>& J, const Q& qa, const Q& qb, bool attjac) const {
        M Qm;
        if (parent) { Qm = LTmat(J.qoff) * Rmat(qb) * Tmat<T>(); if (attjac) Qm = Qm * LVTmat(qa); }
        else        { Qm = LTmat(J.qoff) * LTmat(qa);            if (attjac) Qm = Qm * LVTmat(qb); }
        return Qm;
    }
    // minimal_coordinates   translational/minimal.jl:56-58, rotational/minimal.jl:62-67
    M minimal_coordinates(const Joint<T>& J, const Half<T>& h, const M& xa, const Q& qa, const M& xb, const Q& qb) const {
        if (!h.is_rot) return h.A * tra_displacement(J, xa, qa, xb, qb);
        return h.A * rotation_vector(rot_displacement_q(J, qa, qb));
    }
    // minimal_coordinates_jacobian_configuration   translational/minimal.jl:60-64, rotational/minimal.jl:69-80
    M minimal_coordinates_jacobian_configuration(bool parent, const Joint<T>& J, const Half<T>& h, const M& xa, const Q& qa, const M& xb, const Q& qb, bool attjac) const {
        if (!h.is_rot) { M X, Qm; disp_jac(parent, J, h, xa, qa, xb, qb, attjac, X, Qm); return h.A * hcat(X, Qm); }
        Q q = rot_displacement_q(J, qa, qb);
        M Qm = rot_disp_jac_unstable_novmat(parent, J, qa, qb, attjac);
        return h.A * hcat(M(3, 3), drotation_vectordq(q) * Qm);
    }
    // joint_constraint   joints/joint.jl:9-12
    M joint_constraint(const Joint<T>& J, const Half<T>& h, const M& xa, const Q& qa, const M& xb, const Q& qb) const {
        if (!h.is_rot) return h.C * tra_displacement(J, xa, qa, xb, qb);
        return h.C * Vmat(rot_displacement_q(J, qa, qb));
    }
    // joint_constraint_jacobian_configuration   joints/joint.jl:14-19  (attjac=false: nl x 7)
    M joint_constraint_jacobian_configuration(bool parent, const Joint<T>& J, const Half<T>& h, const M& xa, const Q& qa, const M& xb, const Q& qb) const {
        M X, Qm; disp_jac(parent, J, h, xa, qa, xb, qb, false, X, Qm);
        return h.C * hcat(X, Qm);
    }
    static void split_impulses(const Half<T>& h, const M& eta, M& s, M& g) {   // joint.jl:129-133
        int Nb = h.Nb(); s = M(Nb, 1); g = M(Nb, 1);
        for (int i = 0; i < Nb; ++i) { s[i] = eta[i]; g[i] = eta[Nb + i]; }
    }
    // constraint(joint half, xa,qa,xb,qb, η, μ)   joints/joint.jl:22-26 (Nb=0), joints/limits.jl:1-17
    M half_constraint(const Joint<T>& J, const Half<T>& h, const M& xa, const Q& qa, const M& xb, const Q& qb, const M& eta, T mu_) const {
        M e1 = joint_constraint(J, h, xa, qa, xb, qb);
        if (h.nlim == 0) return e1;
        M e2 = minimal_coordinates(J, h, xa, qa, xb, qb);
        M s, g; split_impulses(h, eta, s, g);
        int Nb = h.Nb(), Nh = h.nlim; M out(h.N(), 1);
        for (int i = 0; i < Nb; ++i) out[i] = s[i] * g[i] - mu_;
        for (int i = 0; i < Nh; ++i) out[Nb + i] = s[i] - (h.hi[i] - e2[i]);
        for (int i = 0; i < Nh; ++i) out[Nb + Nh + i] = s[Nh + i] - (e2[i] - h.lo[i]);
        for (int i = 0; i < h.nl; ++i) out[2 * Nb + i] = e1[i];
        return out;
    }
    // constraint_jacobian(joint half, η)   joints/joint.jl:33-43
    M half_constraint_jacobian(const Half<T>& h, const M& eta) const {
        int N = h.N(), Nb = h.Nb(); M D(N, N);
        if (h.nlim == 0) { for (int i = 0; i < h.nl; ++i) D(i, i) = T(REG); return D; }
        M s, g; split_impulses(h, eta, s, g);
        for (int i = 0; i < Nb; ++i) { D(i, i) = g[i] + T(REG); D(Nb + i, i) = T(1); D(i, Nb + i) = s[i] + T(REG); }
        for (int i = 0; i < h.nl; ++i) D(2 * Nb + i, 2 * Nb + i) = T(REG);
        return D;
    }
    // constraint_jacobian_configuration(relative, joint half, xa,qa,xb,qb, η): N x 7   joint.jl:49-53, limits.jl:19-29
    M half_constraint_jacobian_configuration(bool parent, const Joint<T>& J, const Half<T>& h, const M& xa, const Q& qa, const M& xb, const Q& qb) const {
        M unl = joint_constraint_jacobian_configuration(parent, J, h, xa, qa, xb, qb);
        if (h.nlim == 0) return unl;
        M mc = minimal_coordinates_jacobian_configuration(parent, J, h, xa, qa, xb, qb, false);
        return vcat(vcat(vcat(M(h.Nb(), 7), mc), -mc), unl);
    }
    // impulse_transform   joints/impulses.jl:4-8
    M impulse_transform(bool parent, const Joint<T>& J, const Half<T>& h, const M& xa, const Q& qa, const M& xb, const Q& qb) const {
        M X, Qm; disp_jac(parent, J, h, xa, qa, xb, qb, true, X, Qm);
        M Tm = hcat(X, Qm).t();   // 6x3
        for (int i = 3; i < 6; ++i) for (int j = 0; j < 3; ++j) Tm(i, j) *= T(0.5);
        return Tm;
    }
    // impulse_projector   joints/joint.jl:88-94
    M impulse_projector(const Half<T>& h) const {
        if (h.nlim == 0) return h.C.t();
        return vcat(vcat(vcat(M(h.Nb(), 3), -h.A), h.A), h.C).t();
    }
    // impulse_map(relative, joint half, xa,qa,xb,qb, η): 6 x N   joints/joint.jl:67-86 (at the CURRENT configuration)
    M half_impulse_map(bool parent, const Joint<T>& J, const Half<T>& h, const M& xa, const Q& qa, const M& xb, const Q& qb) const {
        if (h.nlim > 0) return impulse_transform(parent, J, h, xa, qa, xb, qb) * impulse_projector(h);
        M Jc = half_constraint_jacobian_configuration(parent, J, h, xa, qa, xb, qb);   // nl x 7
        M G(7, 6); for (int i = 0; i < 3; ++i) G(i, i) = T(1);
        G.set_block(3, 3, LVTmat(parent ? qa : qb));
        M out = (Jc * G).t();
        for (int i = 3; i < 6; ++i) for (int j = 0; j < out.c; ++j) out(i, j) *= T(0.5);
        return out;
    }
    // impulse_transform_jacobian   translational/impulses.jl:9-46, rotational/impulses.jl:9-39   (6x6, attjac)
    M impulse_transform_jacobian(bool rel_parent, bool jac_parent, const Joint<T>& J, const Half<T>& h, const M& xa, const Q& qa, const M& xb, const Q& qb, const M& p) const {
        M out(6, 6);
        if (!h.is_rot) {
            if (rel_parent) {
                if (jac_parent) {
                    M dXqa = (-drotation_matrix_dq(qa, p)) * LVTmat(qa);
                    M dQxa = dskew_dp(p) * rotation_matrix(inv(qa));
                    M dQqa = (-dskew_dp(p)) * drotation_matrix_inv_dq(qa, xb - xa + rotation_matrix(qb) * J.vc) * LVTmat(qa);
                    out.set_block(0, 3, dXqa); out.set_block(3, 0, dQxa); out.set_block(3, 3, dQqa);
                } else {
                    M dQxb = (-dskew_dp(p)) * rotation_matrix(inv(qa));
                    M dQqb = (-dskew_dp(p)) * rotation_matrix(inv(qa)) * drotation_matrix_dq(qb, J.vc) * LVTmat(qb);
                    out.set_block(3, 0, dQxb); out.set_block(3, 3, dQqb);
                }
            } else {
                M cbpb_w = rotation_matrix(qb) * J.vc;
                if (jac_parent) {
                    M dXqa = drotation_matrix_dq(qa, p) * LVTmat(qa);
                    M dQqa = rotation_matrix(inv(qb)) * skew(cbpb_w) * drotation_matrix_dq(qa, p) * LVTmat(qa);
                    out.set_block(0, 3, dXqa); out.set_block(3, 3, dQqa);
                } else {
                    M dQqb = drotation_matrix_inv_dq(qb, skew(cbpb_w) * rotation_matrix(qa) * p);
                    dQqb += rotation_matrix(inv(qb)) * dskew_dp(rotation_matrix(qa) * p) * drotation_matrix_dq(qb, J.vc);
                    dQqb = dQqb * LVTmat(qb);
                    out.set_block(3, 3, dQqb);
                }
            }
        } else {
            M blk;
            if (rel_parent) {
                if (jac_parent) blk = dVLTmat_dq(Tmat<T>() * RTmat(qb) * LVTmat(J.qoff) * p) * LVTmat(qa);
                else            blk = VLTmat(qa) * Tmat<T>() * dRTmat_dq(LVTmat(J.qoff) * p) * LVTmat(qb);
            } else {
                if (jac_parent) blk = VLTmat(qb) * dLmat_dq(LVTmat(J.qoff) * p) * LVTmat(qa);
                else            blk = dVLTmat_dq(Lmat(qa) * LVTmat(J.qoff) * p) * LVTmat(qb);
            }
            out.set_block(3, 3, T(0.5) * blk);
        }
        return out;
    }
    // impulse_map_jacobian(relative, jacobian, joint half, pbody, cbody, λ)   joints/impulses.jl:13-21
    M half_impulse_map_jacobian(bool rel_parent, bool jac_parent, const Joint<T>& J, const Half<T>& h, const State<T>& pa, const State<T>& ch, const M& lam) const {
        M p = impulse_projector(h) * lam;
        return impulse_transform_jacobian(rel_parent, jac_parent, J, h, pa.x2, pa.q2, ch.x2, ch.q2, p);
    }

    // ---------------- minimal velocities (used by dampers) ----------------
    // translational/minimal.jl:93-113, rotational/minimal.jl:103-118
    M minimal_velocities(const Joint<T>& J, const Half<T>& h, const M& xa, const M& va, const Q& qa, const M& wa,
                         const M& xb, const M& vb, const Q& qb, const M& wb) const {
        Q qa1 = next_orientation(qa, -wa, dt), qb1 = next_orientation(qb, -wb, dt);
        if (!h.is_rot) {
            M xa1 = next_position(xa, -va, dt), xb1 = next_position(xb, -vb, dt);
            M dx = h.A * tra_displacement(J, xa, qa, xb, qb);
            M dx1 = h.A * tra_displacement(J, xa1, qa1, xb1, qb1);
            return (T(1) / dt) * (dx - dx1);
        }
        Q q = inv(J.qoff) * inv(qa) * qb, q1 = inv(J.qoff) * inv(qa1) * qb1;
        return (T(1) / dt) * (h.A * rotation_vector(inv(q1) * q));
    }
    // minimal_velocities_jacobian_configuration   translational/minimal.jl:115-157, rotational/minimal.jl:120-149  -> (3-nl) x 6
    M minimal_velocities_jacobian_configuration(bool parent, const Joint<T>& J, const Half<T>& h, const M& xa, const M& va, const Q& qa, const M& wa,
                                                const M& xb, const M& vb, const Q& qb, const M& wb) const {
        Q qa1 = next_orientation(qa, -wa, dt), qb1 = next_orientation(qb, -wb, dt);
        if (!h.is_rot) {
            M xa1 = next_position(xa, -va, dt), xb1 = next_position(xb, -vb, dt);
            M X, Qm, X1, Q1;
            disp_jac(parent, J, h, xa, qa, xb, qb, false, X, Qm);
            disp_jac(parent, J, h, xa1, qa1, xb1, qb1, false, X1, Q1);
            X1 = -X1;
            Q1 = (-Q1) * rotational_integrator_jacobian_orientation(parent ? qa : qb, parent ? -wa : -wb, dt, false);
            Qm = Qm * LVTmat(parent ? qa : qb);
            Q1 = Q1 * LVTmat(parent ? qa : qb);
            M Jm = (T(1) / dt) * (h.A * hcat(X, Qm));
            Jm += (T(1) / dt) * (h.A * hcat(X1, Q1));
            return Jm;
        }
        Q io = inv(J.qoff);
        Q q = io * inv(qa) * qb, q1 = io * inv(qa1) * qb1;
        M dr = drotation_vectordq(inv(q1) * q);
        M Qm;
        if (parent) {
            Qm = (T(1) / dt) * (h.A * dr * Rmat(q) * Tmat<T>() * Rmat(qb1) * Lmat(io) * Tmat<T>() * rotational_integrator_jacobian_orientation(qa, -wa, dt, false));
            Qm += (T(1) / dt) * (h.A * dr * Lmat(inv(q1)) * Rmat(qb) * Lmat(io) * Tmat<T>());
            Qm = Qm * LVTmat(qa);
        } else {
            Qm = (T(1) / dt) * (h.A * dr * Rmat(q) * Tmat<T>() * Lmat(io * inv(qa1)) * rotational_integrator_jacobian_orientation(qb, -wb, dt, false));
            Qm += (T(1) / dt) * (h.A * dr * Lmat(inv(q1) * io * inv(qa)));
            Qm = Qm * LVTmat(qb);
        }
        return hcat(M(h.nu(), 3), Qm);
    }
    // minimal_velocities_jacobian_velocity   translational/minimal.jl:159-193, rotational/minimal.jl:151-174  -> (3-nl) x 6
    M minimal_velocities_jacobian_velocity(bool parent, const Joint<T>& J, const Half<T>& h, const M& xa, const M& va, const Q& qa, const M& wa,
                                           const M& xb, const M& vb, const Q& qb, const M& wb) const {
        Q qa1 = next_orientation(qa, -wa, dt), qb1 = next_orientation(qb, -wb, dt);
        if (!h.is_rot) {
            M xa1 = next_position(xa, -va, dt), xb1 = next_position(xb, -vb, dt);
            M X1, Q1; disp_jac(parent, J, h, xa1, qa1, xb1, qb1, false, X1, Q1);
            X1 = (-dt) * X1;
            Q1 = (-Q1) * rotational_integrator_jacobian_velocity(parent ? qa : qb, parent ? -wa : -wb, dt);
            return (T(-1) / dt) * (h.A * hcat(X1, Q1));
        }
        Q io = inv(J.qoff);
        Q q = io * inv(qa) * qb, q1 = io * inv(qa1) * qb1;
        M dr = drotation_vectordq(inv(q1) * q);
        M Om;
        if (parent) Om = (T(1) / dt) * (h.A * dr * Rmat(q) * Tmat<T>() * Lmat(io) * Rmat(qb1) * Tmat<T>() * (-rotational_integrator_jacobian_velocity(qa, -wa, dt)));
        else        Om = (T(1) / dt) * (h.A * dr * Rmat(q) * Tmat<T>() * Lmat(io * inv(qa1)) * (-rotational_integrator_jacobian_velocity(qb, -wb, dt)));
        return hcat(M(h.nu(), 3), Om);
    }

    // ---------------- springs ----------------
    // spring_impulses(relative, joint half, pbody, cbody, timestep; unitary)  at the current configuration
    // translational/springs.jl:5-31, rotational/springs.jl:5-40
    M rot_spring_force(bool parent, const Joint<T>& J, const Half<T>& h, const M& xa, const Q& qa, const M& xb, const Q& qb, bool rotate, bool unitary) const {
        T k = unitary ? T(1) : h.spring;
        M distance = h.spring_offset - minimal_coordinates(J, h, xa, qa, xb, qb);
        M force = (-k) * (h.A.t() * distance);
        M out3;
        if (parent) out3 = rotate ? vector_rotate(force, J.qoff) : force;
        else        out3 = rotate ? vector_rotate(-force, inv(qb) * qa * J.qoff) : -force;
        return vcat(M(3, 1), out3);
    }
    M half_spring_impulses(bool parent, const Joint<T>& J, const Half<T>& h, const State<T>& pa, const State<T>& ch, bool unitary) const {
        if (h.nl == 3) return M(6, 1);
        if (!h.is_rot) {
            T k = unitary ? T(1) : h.spring;
            M distance = h.spring_offset - minimal_coordinates(J, h, pa.x2, pa.q2, ch.x2, ch.q2);
            M force = k * (h.A.t() * distance);
            return dt * (impulse_transform(parent, J, h, pa.x2, pa.q2, ch.x2, ch.q2) * force);
        }
        return dt * rot_spring_force(parent, J, h, pa.x2, pa.q2, ch.x2, ch.q2, true, unitary);
    }
    // spring_jacobian_configuration(relative, jacobian, joint half, pbody, cbody, timestep): 6x6 (attjac)
    // translational/springs.jl:37-76 (NB: the Node method passes unitary=false), rotational/springs.jl:46-94
    M half_spring_jacobian_configuration(bool rel_parent, bool jac_parent, const Joint<T>& J, const Half<T>& h, const State<T>& pa, const State<T>& ch) const {
        if (h.nl == 3) return M(6, 6);
        const M& xa = pa.x2; const Q& qa = pa.q2; const M& xb = ch.x2; const Q& qb = ch.q2;
        if (!h.is_rot) {
            M distance = h.spring_offset - minimal_coordinates(J, h, xa, qa, xb, qb);
            M force = h.spring * (h.A.t() * distance);
            M dforce = (-h.spring) * (h.A.t() * minimal_coordinates_jacobian_configuration(jac_parent, J, h, xa, qa, xb, qb, true));
            M Jm = impulse_transform(rel_parent, J, h, xa, qa, xb, qb) * dforce;
            Jm += impulse_transform_jacobian(rel_parent, jac_parent, J, h, xa, qa, xb, qb, force);
            return dt * Jm;
        }
        M force6 = rot_spring_force(rel_parent, J, h, xa, qa, xb, qb, false, false);
        M force = force6.rows(3, 3);
        M mcj = minimal_coordinates_jacobian_configuration(jac_parent, J, h, xa, qa, xb, qb, true);   // nu x 6
        M Jm;
        if (rel_parent) {
            Jm = dt * (rotation_matrix(J.qoff) * (h.spring * (h.A.t() * mcj)));
        } else {
            Q qrel = inv(qb) * qa * J.qoff;
            M J1 = dt * (rotation_matrix(qrel) * ((-h.spring) * (h.A.t() * mcj)));
            M Q2;
            if (jac_parent) { Q2 = dt * (dvector_rotate_dq(force, qrel) * Rmat(J.qoff) * Lmat(inv(qb))); Q2 = Q2 * LVTmat(qa); }
            else            { Q2 = dt * (dvector_rotate_dq(force, qrel) * Rmat(qa * J.qoff) * Tmat<T>()); Q2 = Q2 * LVTmat(qb); }
            Jm = J1 + hcat(M(3, 3), Q2);
        }
        return vcat(M(3, 6), Jm);
    }

    // ---------------- dampers ----------------
    // damper_impulses(relative, joint half, pbody, cbody, timestep; unitary): uses the CANDIDATE velocities vsol[2], ωsol[2]
    // translational/dampers.jl:5-37, rotational/dampers.jl:4-30
    M rot_damper_force(bool parent, const Joint<T>& J, const Half<T>& h, const Q& qa, const M& wa, const Q& qb, const M& wb, bool rotate, bool unitary) const {
        T c = unitary ? T(1) : h.damper;
        M z = M(3, 1);
        M velocity = minimal_velocities(J, h, z, z, qa, wa, z, z, qb, wb);
        M force;
        if (parent) { force = c * (h.A.t() * velocity); if (rotate) force = vector_rotate(force, J.qoff); }
        else        { force = (-c) * (h.A.t() * velocity); if (rotate) force = vector_rotate(force, inv(qb) * qa * J.qoff); }
        return vcat(M(3, 1), force);
    }
    M tra_damper_force3(const Joint<T>& J, const Half<T>& h, const State<T>& pa, const State<T>& ch, bool unitary) const {
        T c = unitary ? T(1) : h.damper;
        return c * (h.A.t() * (-minimal_velocities(J, h, pa.x2, pa.vsol[1], pa.q2, pa.wsol[1], ch.x2, ch.vsol[1], ch.q2, ch.wsol[1])));
    }
    M half_damper_impulses(bool parent, const Joint<T>& J, const Half<T>& h, const State<T>& pa, const State<T>& ch, bool unitary) const {
        if (h.nl == 3) return M(6, 1);
        if (!h.is_rot) return dt * (impulse_transform(parent, J, h, pa.x2, pa.q2, ch.x2, ch.q2) * tra_damper_force3(J, h, pa, ch, unitary));
        return dt * rot_damper_force(parent, J, h, pa.q2, pa.wsol[1], ch.q2, ch.wsol[1], true, unitary);
    }
    // damper_jacobian_velocity(relative, jacobian, joint half, pbody, cbody, timestep): 6x6
    // translational/dampers.jl:99-123, rotational/dampers.jl:66-84
    M half_damper_jacobian_velocity(bool rel_parent, bool jac_parent, const Joint<T>& J, const Half<T>& h, const State<T>& pa, const State<T>& ch) const {
        if (h.nl == 3) return M(6, 6);
        M dvel = minimal_velocities_jacobian_velocity(jac_parent, J, h, pa.x2, pa.vsol[1], pa.q2, pa.wsol[1], ch.x2, ch.vsol[1], ch.q2, ch.wsol[1]);
        if (!h.is_rot) {
            M dinput = h.damper * (h.A.t() * (-dvel));
            return dt * (impulse_transform(rel_parent, J, h, pa.x2, pa.q2, ch.x2, ch.q2) * dinput);
        }
        M VO;
        if (rel_parent) VO = rotation_matrix(J.qoff) * (h.damper * (h.A.t() * dvel));
        else            VO = rotation_matrix(inv(ch.q2) * pa.q2 * J.qoff) * ((-h.damper) * (h.A.t() * dvel));
        return dt * vcat(M(3, 6), VO);
    }
    // damper_jacobian_configuration(relative, jacobian, joint half, pbody, cbody, timestep): 6x6
    // translational/dampers.jl:70-97, rotational/dampers.jl:36-64
    M half_damper_jacobian_configuration(bool rel_parent, bool jac_parent, const Joint<T>& J, const Half<T>& h, const State<T>& pa, const State<T>& ch) const {
        if (h.nl == 3) return M(6, 6);
        const M& xa = pa.x2; const Q& qa = pa.q2; const M& xb = ch.x2; const Q& qb = ch.q2;
        M dvel = minimal_velocities_jacobian_configuration(jac_parent, J, h, xa, pa.vsol[1], qa, pa.wsol[1], xb, ch.vsol[1], qb, ch.wsol[1]);
        if (!h.is_rot) {
            M input = tra_damper_force3(J, h, pa, ch, false);
            M dinput = h.damper * (h.A.t() * (-dvel));
            M dxq = impulse_transform(rel_parent, J, h, xa, qa, xb, qb) * dinput;
            dxq += impulse_transform_jacobian(rel_parent, jac_parent, J, h, xa, qa, xb, qb, input);
            return dt * dxq;
        }
        M force = rot_damper_force(rel_parent, J, h, qa, pa.wsol[1], qb, ch.wsol[1], false, false).rows(3, 3);
        M dv = dvel.cols(3, 3);
        M Qm;
        if (rel_parent) {
            Qm = rotation_matrix(J.qoff) * (h.damper * (h.A.t() * dv));
        } else {
            Qm = rotation_matrix(inv(qb) * qa * J.qoff) * ((-h.damper) * (h.A.t() * dv));
            if (jac_parent) Qm += rotation_matrix(inv(qb)) * drotation_matrix_dq(qa, vector_rotate(force, J.qoff)) * LVTmat(qa);
            else            Qm += drotation_matrix_inv_dq(qb, vector_rotate(force, qa * J.qoff)) * LVTmat(qb);
        }
        return dt * vcat(M(3, 6), hcat(M(3, 3), Qm));
    }

    // ---------------- control input ----------------
    // set_input!(joint, u)  joints/constraints.jl:352-361, joints/joint.jl:96-99
    void set_input(Joint<T>& J, const T* u) {
        M ut(J.tra.nu(), 1), ur(J.rot.nu(), 1);
        for (int i = 0; i < J.tra.nu(); ++i) ut[i] = u[i];
        for (int i = 0; i < J.rot.nu(); ++i) ur[i] = u[J.tra.nu() + i];
        J.tra.input = J.tra.nu() > 0 ? J.tra.A.t() * ut : M(3, 1);
        J.rot.input = J.rot.nu() > 0 ? J.rot.A.t() * ur : M(3, 1);
        // NB: the reference only calls set_input!(half) when the index range is non-empty; an
        // empty range leaves the previous (cleared) input, i.e. zero.
    }
    // input_impulse!(joint, mechanism, clear)  joints/constraints.jl:379-385, translational/input.jl:5-27, rotational/input.jl:5-17
    void input_impulse(Joint<T>& J, bool clear) {
        State<T>& pa = bstate(J.parent); State<T>& ch = bstate(J.child);
        {   // translational
            M input = input_scaling * J.tra.input;
            M Ta = impulse_transform(true, J, J.tra, pa.x2, pa.q2, ch.x2, ch.q2);
            M Tb = impulse_transform(false, J, J.tra, pa.x2, pa.q2, ch.x2, ch.q2);
            pa.JF2 += Ta.rows(0, 3) * input; pa.Jt2 += T(0.5) * (Ta.rows(3, 3) * input);
            ch.JF2 += Tb.rows(0, 3) * input; ch.Jt2 += T(0.5) * (Tb.rows(3, 3) * input);
            if (clear) J.tra.input = M(3, 1);
        }
        {   // rotational
            M tau = input_scaling * J.rot.input;
            pa.Jt2 += vector_rotate(-tau, J.qoff);
            ch.Jt2 += vector_rotate(tau, inv(ch.q2) * pa.q2 * J.qoff);
            if (clear) J.rot.input = M(3, 1);
        }
    }
    // input_jacobian_control(relative, joint half, ...)·nullspace_maskᵀ: 6 x nu_half
    // joints/joint.jl:106-109, translational/input.jl:33-44, rotational/input.jl:23-40
    M half_input_jacobian_control(bool parent, const Joint<T>& J, const Half<T>& h, const State<T>& pa, const State<T>& ch) const {
        M B;
        if (!h.is_rot) {
            M Ta = impulse_transform(parent, J, h, pa.x2, pa.q2, ch.x2, ch.q2);
            B = input_scaling * vcat(Ta.rows(0, 3), T(0.5) * Ta.rows(3, 3));
        } else {
            M Bt = parent ? -rotation_matrix(J.qoff) : rotation_matrix(inv(ch.q2) * pa.q2 * J.qoff);
            B = input_scaling * vcat(M(3, 3), Bt);
        }
        if (h.nu() == 0) return M(6, 0);
        return B * h.A.t();
    }

    // =====================================================================
    // JOINT CONSTRAINT (both halves)   src/joints/constraints.jl
    // =====================================================================
    static M sub(const M& v, int off, int len) { M o(len, 1); for (int i = 0; i < len; ++i) o[i] = v[off + i]; return o; }
    // constraint(mechanism, joint)   constraints.jl:114-120: evaluated at (x3,q3) of parent and child
    M joint_constraint_full(const Joint<T>& J) const {
        const State<T>& pa = bstate(J.parent); const State<T>& ch = bstate(J.child);
        M xa = x3(pa), xb = x3(ch); Q qa = q3(pa), qb = q3(ch);
        M tr = half_constraint(J, J.tra, xa, qa, xb, qb, sub(J.imp[1], 0, J.tra.N()), mu);
        M ro = half_constraint(J, J.rot, xa, qa, xb, qb, sub(J.imp[1], J.tra.N(), J.rot.N()), mu);
        return vcat(tr, ro);
    }
    // constraint_jacobian(joint)   constraints.jl:128-132
    M joint_constraint_jacobian(const Joint<T>& J) const {
        int N = J.N(); M D(N, N);
        D.set_block(0, 0, half_constraint_jacobian(J.tra, sub(J.imp[1], 0, J.tra.N())));
        D.set_block(J.tra.N(), J.tra.N(), half_constraint_jacobian(J.rot, sub(J.imp[1], J.tra.N(), J.rot.N())));
        return D;
    }
    // constraint_jacobian_configuration(mechanism, joint, body): N x 7 at (x3,q3)   constraints.jl:134-147
    M joint_constraint_jacobian_configuration_full(const Joint<T>& J, bool parent) const {
        const State<T>& pa = bstate(J.parent); const State<T>& ch = bstate(J.child);
        M xa = x3(pa), xb = x3(ch); Q qa = q3(pa), qb = q3(ch);
        return vcat(half_constraint_jacobian_configuration(parent, J, J.tra, xa, qa, xb, qb),
                    half_constraint_jacobian_configuration(parent, J, J.rot, xa, qa, xb, qb));
    }
    // impulse_map(mechanism, joint, body): 6 x N at (x2,q2)   constraints.jl:157-168
    M joint_impulse_map(const Joint<T>& J, bool parent) const {
        const State<T>& pa = bstate(J.parent); const State<T>& ch = bstate(J.child);
        return hcat(half_impulse_map(parent, J, J.tra, pa.x2, pa.q2, ch.x2, ch.q2),
                    half_impulse_map(parent, J, J.rot, pa.x2, pa.q2, ch.x2, ch.q2));
    }
    M joint_spring_impulses(const Joint<T>& J, bool parent, bool unitary) const {   // constraints.jl:302-325
        const State<T>& pa = bstate(J.parent); const State<T>& ch = bstate(J.child);
        return half_spring_impulses(parent, J, J.tra, pa, ch, unitary) + half_spring_impulses(parent, J, J.rot, pa, ch, unitary);
    }
    M joint_damper_impulses(const Joint<T>& J, bool parent, bool unitary) const {   // constraints.jl:328-349
        const State<T>& pa = bstate(J.parent); const State<T>& ch = bstate(J.child);
        return half_damper_impulses(parent, J, J.tra, pa, ch, unitary) + half_damper_impulses(parent, J, J.rot, pa, ch, unitary);
    }
    void reset_joint(Joint<T>& J) {   // constraints.jl:440-448
        M l(J.N(), 1); int o = 0;
        for (int i = 0; i < 2 * J.tra.Nb(); ++i) l[o++] = T(1);
        o += J.tra.nl;
        for (int i = 0; i < 2 * J.rot.Nb(); ++i) l[o++] = T(1);
        J.imp[0] = l; J.imp[1] = l;
    }

    // =====================================================================
    // CONTACT (NonlinearContact + SphereHalfSpaceCollision, child = origin)
    // =====================================================================
    // friction_parameterization of LinearContact  linear.jl:33-38
    static M friction_parameterization() { return M(4, 2, {0, 1,  0, -1,  1, 0,  -1, 0}); }
    // force_mapping(:parent, model, ...): [n' 0 T'P'] (contact.jl:141-154; P = I for NonlinearContact), n' for ImpactContact (impact.jl:106-118)
    M force_mapping(const Contact<T>& c) const {
        if (c.model == 1) return c.normal.t();
        if (c.model == 2) return hcat(hcat(c.normal.t(), M(3, 1)), c.tangent.t() * friction_parameterization().t());
        return hcat(hcat(c.normal.t(), M(3, 1)), c.tangent.t());
    }
    // contact_point(:parent, ...)  sphere_halfspace.jl:55-62
    M contact_point_parent(const Contact<T>& c, const M& xp, const Q& qp) const {
        return xp + vector_rotate(c.origin, qp) - c.offset - c.radius * c.normal.t();
    }
    // distance  sphere_halfspace.jl:34-36
    T distance(const Contact<T>& c, const M& xp, const Q& qp) const {
        return (c.normal * (xp + vector_rotate(c.origin, qp) - c.offset))[0] - c.radius;
    }
    // relative_tangential_velocity  velocity.jl:27-38 (child = origin: zero state => v_child_point = 0)
    M relative_tangential_velocity(const Contact<T>& c, const M& xp, const Q& qp, const M& vp, const M& wp) const {
        M cp = contact_point_parent(c, xp, qp);
        M vpt = vp + skew(vector_rotate(wp, qp)) * (cp - xp);   // contact_point_velocity, velocity.jl:2-4
        return c.tangent * vpt;
    }
    // constraint(mechanism, contact)  nonlinear.jl:50-76
    M contact_constraint(const Contact<T>& c) const {
        const State<T>& st = bodies[c.body].st;
        M xp = x3(st); Q qp = q3(st);
        T d = distance(c, xp, qp);
        M vt = relative_tangential_velocity(c, xp, qp, st.vsol[1], st.wsol[1]);
        const M& g = c.gam[1]; const M& s = c.s[1];
        if (c.model == 1) return M::vec({d - s[0]});                      // impact.jl:41-54
        if (c.model == 2) {                                               // linear.jl:71-102: [d − sγ; μγ − Σβ − sψ; P vt + ψ 1 − sβ]
            M pv = friction_parameterization() * vt;
            return M::vec({d - s[0], c.mu * g[0] - (g[2] + g[3] + g[4] + g[5]) - s[1],
                           pv[0] + g[1] - s[2], pv[1] + g[1] - s[3], pv[2] + g[1] - s[4], pv[3] + g[1] - s[5]});
        }
        return M::vec({d - s[0], c.mu * g[0] - g[1], vt[0] - s[2], vt[1] - s[3]});
    }
    static M cone_product(const M& u, const M& v) {   // cone.jl:6-8 (3-vectors)
        return M::vec({u[0] * v[0] + u[1] * v[1] + u[2] * v[2], u[0] * v[1] + v[0] * u[1], u[0] * v[2] + v[0] * u[2]});
    }
    static M cone_product_jacobian(const M& u) {     // cone.jl:10-12 (column-major literal in the reference)
        return M(3, 3, {u[0], u[1], u[2],  u[1], u[0], 0,  u[2], 0, u[0]});
    }
    // complementarity(mechanism, contact)  complementarity.jl:16-22
    M contact_complementarity(const Contact<T>& c) const {
        const M& g = c.gam[1]; const M& s = c.s[1];
        if (c.model == 1) return M::vec({g[0] * s[0]});                    // complementarity.jl:16 (γ .* s)
        if (c.model == 2) { M r(6, 1); for (int i = 0; i < 6; ++i) r[i] = g[i] * s[i]; return r; }   // complementarity.jl:16
        M cp = cone_product(sub(g, 1, 3), sub(s, 1, 3));
        return M::vec({g[0] * s[0], cp[0], cp[1], cp[2]});
    }
    // constraint_jacobian(contact): 8x8  nonlinear.jl:78-97
    M contact_constraint_jacobian(const Contact<T>& c) const {
        M g = c.gam[1], s = c.s[1];
        if (c.model == 1) {   // impact.jl:56-62: [γ s; −1 0] (columns s, γ) with γ, s + REG·neutral_vector
            M D2(2, 2); D2(0, 0) = g[0] + T(REG); D2(0, 1) = s[0] + T(REG); D2(1, 0) = T(-1); D2(1, 1) = T(0);
            return D2;
        }
        if (c.model == 2) {   // linear.jl:49-69: [∇s ∇γ], ∇s = [Diag(γ); −I], ∇γ = [Diag(s); M(μ)], with γ, s + REG·ones(6)
            M D12(12, 12);
            for (int i = 0; i < 6; ++i) { D12(i, i) = g[i] + T(REG); D12(6 + i, i) = T(-1); D12(i, 6 + i) = s[i] + T(REG); }
            D12(7, 6) = c.mu; for (int i = 2; i < 6; ++i) { D12(7, 6 + i) = T(-1); D12(6 + i, 7) = T(1); }
            return D12;
        }
        g[0] += T(REG); g[1] += T(REG); s[0] += T(REG); s[1] += T(REG);   // + REG * neutral_vector = [1,1,0,0]
        M D(8, 8);
        // ∇s
        D(0, 0) = g[0];
        D.set_block(1, 1, cone_product_jacobian(sub(g, 1, 3)));
        D(4, 0) = T(-1); D(5, 1) = T(0); D(6, 2) = T(-1); D(7, 3) = T(-1);
        // ∇γ
        D(0, 4) = s[0];
        D.set_block(1, 5, cone_product_jacobian(sub(s, 1, 3)));
        D(5, 4) = c.mu; D(5, 5) = T(-1);
        return D;
    }
    // impulse_map(:parent, model, pbody, cbody, timestep): 6x4 at (x3,q3)  contact.jl:79-100,141-154
    M contact_impulse_map(const Contact<T>& c) const {
        const State<T>& st = bodies[c.body].st;
        M xp = x3(st); Q qp = q3(st);
        M X = force_mapping(c);
        M cp = contact_point_parent(c, xp, qp);
        M Qm = rotation_matrix(inv(qp)) * skew(cp - xp) * X;
        return vcat(X, Qm);
    }
    // ∂contact_point∂q(:parent,:parent) = ∂vector_rotate∂q(origin, qp)  sphere_halfspace.jl:82-96
    // constraint_jacobian_velocity(:parent, model, x3,v25,q3,ω25, ..., timestep): 4x6  contact.jl:37-77
    M contact_constraint_jacobian_velocity(const Contact<T>& c) const {
        const State<T>& st = bodies[c.body].st;
        M xp = x3(st); Q qp = q3(st); const M& vp = st.vsol[1]; const M& wp = st.wsol[1];
        M dd_dx = c.normal;                                              // ∂distance∂x
        M dd_dq = c.normal * dvector_rotate_dq(c.origin, qp);             // ∂distance∂q
        M dvt_dq = dvt_dq_parent(c, xp, qp, vp, wp);
        M dvt_dv = c.tangent;                                            // ∂relative_tangential_velocity∂v
        M cp = contact_point_parent(c, xp, qp);
        M dvt_dw = c.tangent * ((-skew(cp - xp)) * rotation_matrix(qp)); // ∂relative_tangential_velocity∂ω
        // "recover current orientation"
        Q q = next_orientation(qp, -wp, dt);
        M dq_dw = rotational_integrator_jacobian_velocity(q, wp, dt);
        if (c.model == 1) return hcat(dt * dd_dx, dd_dq * dq_dw);   // impact.jl:76-104
        if (c.model == 2) {                                          // contact.jl:37-77 with the 4x2 friction_parameterization
            M P = friction_parameterization();
            return hcat(vcat(vcat(dt * dd_dx, M(1, 3)), P * dvt_dv), vcat(vcat(dd_dq * dq_dw, M(1, 3)), P * (dvt_dw + dvt_dq * dq_dw)));
        }
        M V = vcat(vcat(dt * dd_dx, M(1, 3)), dvt_dv);
        M Om = vcat(vcat(dd_dq * dq_dw, M(1, 3)), dvt_dw + dvt_dq * dq_dw);
        return hcat(V, Om);
    }
    // ∂relative_tangential_velocity∂q(:parent, ...)  velocity.jl:71-99 (tangent derivative terms vanish for a half-space)
    M dvt_dq_parent(const Contact<T>& c, const M& xp, const Q& qp, const M& vp, const M& wp) const {
        M cp = contact_point_parent(c, xp, qp);
        M X = c.tangent * ((-skew(cp - xp)) * dvector_rotate_dq(wp, qp));                       // ∂contact_point_velocity∂q
        X += c.tangent * skew(vector_rotate(wp, qp)) * dvector_rotate_dq(c.origin, qp);          // ∂cpv∂c · ∂contact_point∂q(:parent)
        // child term: ∂contact_point_velocity∂c(child) = skew(vector_rotate(ωc, qc)) = 0 for the origin
        return X;
    }
    // ∂relative_tangential_velocity∂x(:parent, ...)  velocity.jl:40-69
    M dvt_dx_parent(const Contact<T>& c, const M& xp, const Q& qp, const M& vp, const M& wp) const {
        M S = skew(vector_rotate(wp, qp));
        M X = c.tangent * (-S);            // ∂contact_point_velocity∂x
        X += c.tangent * S;                // ∂cpv∂c · ∂contact_point∂x(:parent,:parent) = I
        return X;                          // child term is zero (ωc = 0)
    }
    // constraint_jacobian_configuration(:parent, model, x3,v25,q3,ω25,...): 4x7  contact.jl:9-35
    M contact_constraint_jacobian_configuration(const Contact<T>& c) const {
        const State<T>& st = bodies[c.body].st;
        M xp = x3(st); Q qp = q3(st); const M& vp = st.vsol[1]; const M& wp = st.wsol[1];
        M Pm = c.model == 2 ? friction_parameterization() : M::eye(2);      // contact.jl:9-35
        M X = vcat(vcat(c.normal, M(1, 3)), Pm * dvt_dx_parent(c, xp, qp, vp, wp));
        M Qm = vcat(vcat(c.normal * dvector_rotate_dq(c.origin, qp), M(1, 4)), Pm * dvt_dq_parent(c, xp, qp, vp, wp));
        return hcat(X, Qm);
    }
    // impulse_map_jacobian(:parent,:parent, model, pbody, cbody, λ, timestep): 6x7  contact.jl:102-138
    M contact_impulse_map_jacobian(const Contact<T>& c) const {
        const State<T>& st = bodies[c.body].st;
        M xp = x3(st); Q qp = q3(st); const M& lam = c.gam[1];
        M X = force_mapping(c);
        M Xx(3, 3), Xq(3, 4);     // ∂force_mapping_jvp∂x / ∂q vanish for a half-space
        M cp = contact_point_parent(c, xp, qp);
        M r = cp - xp; Q qi = inv(qp);
        M Qx = rotation_matrix(qi) * skew(r) * Xx;
        Qx -= rotation_matrix(qi) * skew(X * lam) * (M::eye(3) - M::eye(3));     // ∂contact_point∂x(:parent,:parent) − I = 0
        M Qq = rotation_matrix(qi) * skew(r) * Xq;
        Qq -= rotation_matrix(qi) * skew(X * lam) * dvector_rotate_dq(c.origin, qp);
        Qq += drotation_matrix_dq(qi, skew(r) * (X * lam)) * Tmat<T>();
        return vcat(hcat(Xx, Xq), hcat(Qx, Qq));
    }
    // =====================================================================
    // BODY-BODY CONTACT: SphereSphereCollision between c.body (parent) and c.body2 (child)
    // src/contacts/collisions/{collision,sphere_sphere}.jl, src/contacts/{contact,velocity}.jl for two bodies.
    // The reference RETURNS FiniteDiff approximations from ∂distance∂x/∂q and ∂contact_point∂x/∂q (sphere_sphere.jl:57-63,84-92,
    // 144-152,186-196) right after computing the analytic expressions; this restatement returns the analytic ones.
    // =====================================================================
    static M normalize_(const M& x) { return (T(1) / norm2(x)) * x; }
    static M dnormalize_dx(const M& x) {                                  // utilities/normalize.jl:10-19
        T mag = norm2(x);
        if (!(mag > T(0))) return M::eye(3);
        return (T(1) / mag) * M::eye(3) - (T(1) / (mag * mag * mag)) * (x * x.t());
    }
    struct SS { M xp, xc; Q qp, qc; };
    SS ss_next(const Contact<T>& c) const { const State<T>& a = bodies[c.body].st; const State<T>& b = bodies[c.body2].st; return SS{x3(a), x3(b), q3(a), q3(b)}; }
    M ss_cop(const Contact<T>& c, const SS& k) const { return k.xp + vector_rotate(c.origin, k.qp); }      // contact_point_origin  collision.jl:9-11
    M ss_coc(const Contact<T>& c, const SS& k) const { return k.xc + vector_rotate(c.origin_c, k.qc); }
    T ss_distance(const Contact<T>& c, const SS& k) const { return norm2(ss_cop(c, k) - ss_coc(c, k)) - (c.radius + c.radius_c); }   // sphere_sphere.jl:28-38
    M ss_contact_point(bool rel_parent, const Contact<T>& c, const SS& k) const {                         // :96-111
        M cop = ss_cop(c, k), coc = ss_coc(c, k); M dir = normalize_(cop - coc);
        return rel_parent ? cop - c.radius * dir : coc + c.radius_c * dir;
    }
    M ss_dd_dx(bool jac_parent, const Contact<T>& c, const SS& k) const {                                 // :40-66 (the analytic D)
        M dn = (T(1) / norm2(ss_cop(c, k) - ss_coc(c, k))) * (ss_cop(c, k) - ss_coc(c, k)).t();         // ∂norm∂x
        return jac_parent ? dn : T(-1) * dn;
    }
    M ss_dd_dq(bool jac_parent, const Contact<T>& c, const SS& k) const {                                 // :68-93
        M dn = (T(1) / norm2(ss_cop(c, k) - ss_coc(c, k))) * (ss_cop(c, k) - ss_coc(c, k)).t();
        return jac_parent ? dn * dvector_rotate_dq(c.origin, k.qp) : T(-1) * (dn * dvector_rotate_dq(c.origin_c, k.qc));
    }
    M ss_dcp_dx(bool rel_parent, bool jac_parent, const Contact<T>& c, const SS& k) const {               // :113-153 (the analytic X)
        M N = dnormalize_dx(ss_cop(c, k) - ss_coc(c, k));
        if (rel_parent) return jac_parent ? M::eye(3) - c.radius * N : c.radius * N;
        return jac_parent ? c.radius_c * N : M::eye(3) - c.radius_c * N;
    }
    M ss_dcp_dq(bool rel_parent, bool jac_parent, const Contact<T>& c, const SS& k) const {               // :155-199 (the analytic Q)
        M N = dnormalize_dx(ss_cop(c, k) - ss_coc(c, k));
        M dp = dvector_rotate_dq(c.origin, k.qp), dc = dvector_rotate_dq(c.origin_c, k.qc);
        if (rel_parent) return jac_parent ? dp - c.radius * (N * dp) : c.radius * (N * dc);
        return jac_parent ? c.radius_c * (N * dp) : dc - c.radius_c * (N * dc);
    }
    // contact_normal (1x3, child -> parent, flipped in penetration)  collision.jl:27-45
    M ss_normal(const Contact<T>& c, const SS& k) const {
        M dir = ss_contact_point(true, c, k) - ss_contact_point(false, c, k);
        M n = normalize_(dir).t();
        return ss_distance(c, k) >= T(0) ? n : T(-1) * n;
    }
    M ss_dnT_dx(bool jac_parent, const Contact<T>& c, const SS& k) const {                                // collision.jl:47-68
        M dir = ss_contact_point(true, c, k) - ss_contact_point(false, c, k);
        M X = dnormalize_dx(dir) * (ss_dcp_dx(true, jac_parent, c, k) - ss_dcp_dx(false, jac_parent, c, k));
        return ss_distance(c, k) >= T(0) ? X : T(-1) * X;
    }
    M ss_dnT_dq(bool jac_parent, const Contact<T>& c, const SS& k) const {                                // collision.jl:70-100
        M dir = ss_contact_point(true, c, k) - ss_contact_point(false, c, k);
        M X = dnormalize_dx(dir) * (ss_dcp_dq(true, jac_parent, c, k) - ss_dcp_dq(false, jac_parent, c, k));
        return ss_distance(c, k) >= T(0) ? X : T(-1) * X;
    }
    // the candidate axis of the first tangent (collision.jl:102-116; the Jacobians test 1e-5 instead of 1e-6, :160-167)
    M ss_w(const Contact<T>& c, const SS& k, T tol) const {
        M n = ss_normal(c, k); M w = M::vec({1, 0, 0});
        if (!(norm2(skew(w) * n.t()) > tol)) w = M::vec({0, 1, 0});
        return w;
    }
    M ss_t1(const Contact<T>& c, const SS& k) const { return (skew(ss_w(c, k, T(1e-6))) * ss_normal(c, k).t()).t(); }    // 1x3, not normalized
    M ss_t2(const Contact<T>& c, const SS& k) const { return (skew(ss_t1(c, k).t()) * ss_normal(c, k).t()).t(); }          // collision.jl:118-129
    M ss_tangent(const Contact<T>& c, const SS& k) const { return vcat(ss_t1(c, k), ss_t2(c, k)); }                       // 2x3
    M ss_dt1T_dx(bool jp, const Contact<T>& c, const SS& k) const { return skew(ss_w(c, k, T(1e-5))) * ss_dnT_dx(jp, c, k); }   // :149-174
    M ss_dt2T_dx(bool jp, const Contact<T>& c, const SS& k) const {                                                          // :176-194
        return skew(ss_t1(c, k).t()) * ss_dnT_dx(jp, c, k) + skew(T(-1) * ss_normal(c, k).t()) * ss_dt1T_dx(jp, c, k);
    }
    // ∂contact_tangent_one_transpose∂q multiplies by skew(t1) where the ∂x version has skew(w) (collision.jl:207): literal
    M ss_dt1T_dq(bool jp, const Contact<T>& c, const SS& k) const {
        M w = ss_w(c, k, T(1e-5)); M t1 = skew(w) * ss_normal(c, k).t();
        return skew(t1) * ss_dnT_dq(jp, c, k);
    }
    M ss_dt2T_dq(bool jp, const Contact<T>& c, const SS& k) const {                                                          // :222-247
        M w = ss_w(c, k, T(1e-5)); M t1 = skew(w) * ss_normal(c, k).t();
        M dt1 = skew(w) * ss_dnT_dq(jp, c, k);
        return skew(t1) * ss_dnT_dq(jp, c, k) + skew(T(-1) * ss_normal(c, k).t()) * dt1;
    }
    // contact point velocities and Δv = vp − vc  velocity.jl:2-38
    struct SSV { M cp, cc, vp, vc, dv; };
    SSV ss_velocities(const Contact<T>& c, const SS& k) const {
        const State<T>& a = bodies[c.body].st; const State<T>& b = bodies[c.body2].st;
        SSV r; r.cp = ss_contact_point(true, c, k); r.cc = ss_contact_point(false, c, k);
        r.vp = a.vsol[1] + skew(vector_rotate(a.wsol[1], k.qp)) * (r.cp - k.xp);
        r.vc = b.vsol[1] + skew(vector_rotate(b.wsol[1], k.qc)) * (r.cc - k.xc);
        r.dv = r.vp - r.vc;
        return r;
    }
    M ss_vt(const Contact<T>& c, const SS& k) const { return ss_tangent(c, k) * ss_velocities(c, k).dv; }
    // ∂relative_tangential_velocity∂x / ∂q / ∂v / ∂ω (jacobian = :parent | :child)  velocity.jl:40-143
    M ss_dvt_dx(bool jp, const Contact<T>& c, const SS& k) const {
        const State<T>& a = bodies[c.body].st; const State<T>& b = bodies[c.body2].st;
        SSV v = ss_velocities(c, k); M Tm = ss_tangent(c, k);
        M Sp = skew(vector_rotate(a.wsol[1], k.qp)), Sc = skew(vector_rotate(b.wsol[1], k.qc));     // ∂cpv∂c; ∂cpv∂x = −that
        M X;
        if (jp) { X = Tm * (T(-1) * Sp); X += Tm * Sp * ss_dcp_dx(true, true, c, k); X -= Tm * Sc * ss_dcp_dx(false, true, c, k); }
        else    { X = Tm * Sp * ss_dcp_dx(true, false, c, k); X -= Tm * (T(-1) * Sc); X -= Tm * Sc * ss_dcp_dx(false, false, c, k); }
        X += vcat(v.dv.t() * ss_dt1T_dx(jp, c, k), v.dv.t() * ss_dt2T_dx(jp, c, k));
        return X;
    }
    M ss_dvt_dq(bool jp, const Contact<T>& c, const SS& k) const {
        const State<T>& a = bodies[c.body].st; const State<T>& b = bodies[c.body2].st;
        SSV v = ss_velocities(c, k); M Tm = ss_tangent(c, k);
        M Sp = skew(vector_rotate(a.wsol[1], k.qp)), Sc = skew(vector_rotate(b.wsol[1], k.qc));
        M X;
        if (jp) { X = Tm * ((T(-1) * skew(v.cp - k.xp)) * dvector_rotate_dq(a.wsol[1], k.qp)); X += Tm * Sp * ss_dcp_dq(true, true, c, k); X -= Tm * Sc * ss_dcp_dq(false, true, c, k); }
        else    { X = Tm * Sp * ss_dcp_dq(true, false, c, k); X -= Tm * ((T(-1) * skew(v.cc - k.xc)) * dvector_rotate_dq(b.wsol[1], k.qc)); X -= Tm * Sc * ss_dcp_dq(false, false, c, k); }
        X += vcat(v.dv.t() * ss_dt1T_dq(jp, c, k), v.dv.t() * ss_dt2T_dq(jp, c, k));
        return X;
    }
    M ss_dvt_dv(bool jp, const Contact<T>& c, const SS& k) const { return jp ? ss_tangent(c, k) : T(-1) * ss_tangent(c, k); }
    M ss_dvt_dw(bool jp, const Contact<T>& c, const SS& k) const {
        SSV v = ss_velocities(c, k); M Tm = ss_tangent(c, k);
        return jp ? Tm * ((T(-1) * skew(v.cp - k.xp)) * rotation_matrix(k.qp)) : T(-1) * (Tm * ((T(-1) * skew(v.cc - k.xc)) * rotation_matrix(k.qc)));
    }
    // constraint(mechanism, contact)  nonlinear.jl:50-76 / impact.jl:41-54 with a two-body collision
    M ss_constraint(const Contact<T>& c) const {
        SS k = ss_next(c); T d = ss_distance(c, k);
        const M& g = c.gam[1]; const M& s = c.s[1];
        if (c.model == 1) return M::vec({d - s[0]});
        M vt = ss_vt(c, k);
        if (c.model == 2) {                                               // linear.jl:71-102
            M pv = friction_parameterization() * vt;
            return M::vec({d - s[0], c.mu * g[0] - (g[2] + g[3] + g[4] + g[5]) - s[1],
                           pv[0] + g[1] - s[2], pv[1] + g[1] - s[3], pv[2] + g[1] - s[4], pv[3] + g[1] - s[5]});
        }
        return M::vec({d - s[0], c.mu * g[0] - g[1], vt[0] - s[2], vt[1] - s[3]});
    }
    // constraint_jacobian_velocity(relative, model, ...)  contact.jl:37-77 (impact.jl:76-104)
    M ss_constraint_jacobian_velocity(bool rel_parent, const Contact<T>& c) const {
        SS k = ss_next(c); const State<T>& st = bodies[rel_parent ? c.body : c.body2].st;
        M dd_dx = ss_dd_dx(rel_parent, c, k), dd_dq = ss_dd_dq(rel_parent, c, k);
        // "recover current orientation": next_orientation(q3, −ω, Δt) and the integrator Jacobian there
        Q q = next_orientation(rel_parent ? k.qp : k.qc, T(-1) * st.wsol[1], dt);
        M dq_dw = rotational_integrator_jacobian_velocity(q, st.wsol[1], dt);
        if (c.model == 1) return hcat(dt * dd_dx, dd_dq * dq_dw);
        M Pm = c.model == 2 ? friction_parameterization() : M::eye(2);
        M V = vcat(vcat(dt * dd_dx, M(1, 3)), Pm * ss_dvt_dv(rel_parent, c, k));
        M Om = vcat(vcat(dd_dq * dq_dw, M(1, 3)), Pm * (ss_dvt_dw(rel_parent, c, k) + ss_dvt_dq(rel_parent, c, k) * dq_dw));
        return hcat(V, Om);
    }
    // force_mapping(relative, model, ...)  contact.jl:141-154
    M ss_force_mapping(bool rel_parent, const Contact<T>& c, const SS& k) const {
        M X = c.model == 1 ? ss_normal(c, k).t() : hcat(hcat(ss_normal(c, k).t(), M(3, 1)), c.model == 2 ? ss_tangent(c, k).t() * friction_parameterization().t() : ss_tangent(c, k).t());
        return rel_parent ? X : T(-1) * X;
    }
    // impulse_map(relative, model, pbody, cbody, timestep)  contact.jl:79-100
    M ss_impulse_map(bool rel_parent, const Contact<T>& c) const {
        SS k = ss_next(c);
        M X = ss_force_mapping(rel_parent, c, k);
        M r = ss_contact_point(rel_parent, c, k) - (rel_parent ? k.xp : k.xc);
        return vcat(X, rotation_matrix(inv(rel_parent ? k.qp : k.qc)) * skew(r) * X);
    }
    // impulse_map_jacobian(relative, jacobian = relative, model, pbody, cbody, λ, timestep): 6x7  contact.jl:102-138
    M ss_impulse_map_jacobian(bool rel_parent, const Contact<T>& c) const {
        SS k = ss_next(c); const M& lam = c.gam[1]; const bool jp = rel_parent;
        M X = ss_force_mapping(rel_parent, c, k);
        M Xx = lam[0] * ss_dnT_dx(jp, c, k), Xq = lam[0] * ss_dnT_dq(jp, c, k);                         // ∂force_mapping_jvp∂x / ∂q  :157-199
        if (c.model != 1) {
            // λ_tangent = friction_parameterization' * λ[3:end]  (contact.jl:165,188): (β3 − β4, β1 − β2) for the LinearContact pyramid
            const T l1 = c.model == 2 ? lam[4] - lam[5] : lam[2], l2 = c.model == 2 ? lam[2] - lam[3] : lam[3];
            Xx += l1 * ss_dt1T_dx(jp, c, k) + l2 * ss_dt2T_dx(jp, c, k); Xq += l1 * ss_dt1T_dq(jp, c, k) + l2 * ss_dt2T_dq(jp, c, k);
        }
        if (!rel_parent) { Xx = T(-1) * Xx; Xq = T(-1) * Xq; }
        M r = ss_contact_point(rel_parent, c, k) - (rel_parent ? k.xp : k.xc); Q qi = inv(rel_parent ? k.qp : k.qc);
        M Qx = rotation_matrix(qi) * skew(r) * Xx;
        Qx -= rotation_matrix(qi) * skew(X * lam) * (ss_dcp_dx(rel_parent, jp, c, k) - M::eye(3));
        M Qq = rotation_matrix(qi) * skew(r) * Xq;
        Qq -= rotation_matrix(qi) * skew(X * lam) * ss_dcp_dq(rel_parent, jp, c, k);
        Qq += drotation_matrix_dq(qi, skew(r) * (X * lam)) * Tmat<T>();
        return vcat(hcat(Xx, Xq), hcat(Qx, Qq));
    }

    void reset_contact(Contact<T>& c) {   // contacts/constraints.jl:79-86, neutral_vector nonlinear.jl:99
        M nv = c.model == 1 ? M::vec({1}) : c.model == 2 ? M::vec({1, 1, 1, 1, 1, 1}) : M::vec({1, 1, 0, 0});       // neutral_vector: contact.jl:202 (ones(N½)) / nonlinear.jl:99
        c.gam[0] = nv; c.gam[1] = nv; c.s[0] = nv; c.s[1] = nv;
    }
    // initialize!(contact)   solver/initialization.jl:7-49
    static void initialize_positive_orthant(T& g, T& s) {
        const T eps = T(1e-20);
        T ds = std::fmax(T(-1.5) * s, T(0)), dg = std::fmax(T(-1.5) * g, T(0));
        T sh = s + ds, gh = g + dg;
        T dhs = T(0.5) * sh * gh / (gh + eps), dhg = T(0.5) * sh * gh / (sh + eps);
        s = sh + dhs; g = gh + dhg;
    }
    static void initialize_second_order_cone(M& g, M& s) {   // 3-vectors
        const T eps = T(1e-20);
        auto n2 = [](const M& v) { return std::sqrt(v[1] * v[1] + v[2] * v[2]); };
        T ds = std::fmax(T(-1.5) * (s[0] - n2(s)), T(0)), dg = std::fmax(T(-1.5) * (g[0] - n2(g)), T(0));
        M sh = s, gh = g; sh[0] += ds; gh[0] += dg;
        T sg = dot(sh, gh);
        T dhs = T(0.5) * sg / ((gh[0] + n2(gh)) + eps), dhg = T(0.5) * sg / ((sh[0] + n2(sh)) + eps);
        s = sh; g = gh; s[0] += dhs; g[0] += dhg;
    }
    void initialize_contact(Contact<T>& c) {
        // the generic initialize! (initialization.jl:1-5) discards what initialize_positive_orthant! returns: for an
        // ImpactContact the variables stay at the neutral vector of reset!
        if (c.model != 0) return;         // (ImpactContact and LinearContact take the generic method)
        for (int k = 0; k < 2; ++k) {
            T g0 = c.gam[k][0], s0 = c.s[k][0]; initialize_positive_orthant(g0, s0);
            M gs = sub(c.gam[k], 1, 3), ss = sub(c.s[k], 1, 3); initialize_second_order_cone(gs, ss);
            c.gam[k] = M::vec({g0, gs[0], gs[1], gs[2]}); c.s[k] = M::vec({s0, ss[0], ss[1], ss[2]});
        }
    }

    // =====================================================================
    // BODY   src/integrators/constraint.jl
    // =====================================================================
    // constraint(mechanism, body)  constraint.jl:1-34
    M body_constraint(int ib) {
        Body<T>& B = bodies[ib]; State<T>& s = B.st;
        M x3_ = x3(s); Q q3_ = q3(s);
        M D1x = (T(-1) / dt * B.mass) * (s.x2 - s.x1) - (T(0.5) * dt) * (B.mass * gravity + s.Fext);
        M D2x = (T(1) / dt * B.mass) * (x3_ - s.x2) - (T(0.5) * dt) * (B.mass * gravity + s.Fext);
        M D1q = (T(-2) / dt) * (LVTmat(s.q2).t() * Lmat(s.q1) * VTmat<T>() * B.inertia * Vmat<T>() * Lmat(s.q1).t() * vector(s.q2)) - (T(0.5) * dt) * s.text;
        M D2q = (T(-2) / dt) * (LVTmat(s.q2).t() * Tmat<T>() * Rmat(q3_).t() * VTmat<T>() * B.inertia * Vmat<T>() * Lmat(s.q2).t() * vector(q3_)) - (T(0.5) * dt) * s.text;
        M d = vcat(D2x + D1x, D2q + D1q);
        d -= vcat(s.JF2, s.Jt2);
        // impulses!  joints/constraints.jl:150-155, contacts/constraints.jl:37-40
        for (auto& J : joints) {
            bool parent = (J.parent == ib), child = (J.child == ib);
            if (!parent && !child) continue;
            if (J.N() > 0) d -= joint_impulse_map(J, parent) * J.imp[1];
            if (J.spring) d -= joint_spring_impulses(J, parent, false);
            if (J.damper) d -= joint_damper_impulses(J, parent, false);
        }
        for (auto& c : contacts) {
            if (c.kind == 1) { if (c.body == ib || c.body2 == ib) d -= ss_impulse_map(c.body == ib, c) * c.gam[1]; continue; }
            if (c.body == ib) d -= contact_impulse_map(c) * c.gam[1];
        }
        s.d = d;
        return d;
    }
    // constraint_jacobian_configuration(mechanism, body)  constraint.jl:36-66
    M body_constraint_jacobian(int ib) {
        Body<T>& B = bodies[ib]; State<T>& s = B.st;
        Q q3_ = q3(s);
        M dynR = (T(-2) / dt) * (LVTmat(s.q2).t() * Tmat<T>() * (dRTmat_dq(VTmat<T>() * B.inertia * Vmat<T>() * Lmat(s.q2).t() * vector(q3_)) + Rmat(q3_).t() * VTmat<T>() * B.inertia * Vmat<T>() * Lmat(s.q2).t()));
        M lhs(6, 7);
        for (int i = 0; i < 3; ++i) lhs(i, i) = B.mass / dt;
        lhs.set_block(3, 3, dynR);
        M ijv = integrator_jacobian_velocity(s.q2, s.wsol[1], dt);
        M D = lhs * ijv;
        for (int i = 0; i < 6; ++i) D(i, i) += T(REG);
        // impulses_jacobian_velocity!  joints/constraints.jl:187-205, contacts/constraints.jl:54-57
        for (auto& J : joints) {
            bool parent = (J.parent == ib), child = (J.child == ib);
            if (!parent && !child) continue;
            const State<T>& pa = bstate(J.parent); const State<T>& ch = bstate(J.child);
            // spring_jacobian_velocity == 0 (translational/springs.jl:78, rotational/springs.jl:96)
            if (J.damper) {
                D -= half_damper_jacobian_velocity(parent, parent, J, J.tra, pa, ch);
                D -= half_damper_jacobian_velocity(parent, parent, J, J.rot, pa, ch);
            }
        }
        for (auto& c : contacts) {
            if (c.kind == 1) { if (c.body == ib || c.body2 == ib) D -= ss_impulse_map_jacobian(c.body == ib, c) * ijv; continue; }
            if (c.body == ib) D -= contact_impulse_map_jacobian(c) * ijv;
        }
        s.D = D;
        return D;
    }

    // =====================================================================
    // LINEAR SYSTEM   src/solver/linear_system.jl:1-17 (dense restatement of set_entries!)
    // =====================================================================
    void put(int r0, int c0, const M& blk) { for (int i = 0; i < blk.r; ++i) for (int j = 0; j < blk.c; ++j) A[(size_t)(r0 + i) * n + c0 + j] = blk(i, j); }
    void putv(int r0, const M& v) { for (int i = 0; i < v.size(); ++i) b[r0 + i] = v.a[i]; }
    void set_entries() {
        std::fill(A.begin(), A.end(), T(0)); std::fill(b.begin(), b.end(), T(0));
        // joints: diagonal + vector (joints/constraints.jl:296-299), off-diagonals (constraints.jl:208-214)
        for (size_t j = 0; j < joints.size(); ++j) {
            Joint<T>& J = joints[j];
            if (J.N() == 0) continue;
            put(joff[j], joff[j], joint_constraint_jacobian(J));
            putv(joff[j], -joint_constraint_full(J));
            for (int side = 0; side < 2; ++side) {
                bool parent = side == 0; int ib = parent ? J.parent : J.child;
                if (ib < 0) continue;
                const State<T>& s = bodies[ib].st;
                put(boff[ib], joff[j], -joint_impulse_map(J, parent));
                put(joff[j], boff[ib], joint_constraint_jacobian_configuration_full(J, parent) * integrator_jacobian_velocity(s.q2, s.wsol[1], dt));
            }
        }
        // bodies: diagonal + vector (integrators/constraint.jl:82-85)
        for (size_t i = 0; i < bodies.size(); ++i) {
            put(boff[i], boff[i], body_constraint_jacobian((int)i));
            putv(boff[i], -body_constraint((int)i));
        }
        // body-body damper blocks (joints/constraints.jl:216-249)
        for (auto& J : joints) {
            if (J.parent < 0 || !J.damper) continue;
            const State<T>& pa = bodies[J.parent].st; const State<T>& ch = bodies[J.child].st;
            M pc(6, 6), cp(6, 6);
            pc -= half_damper_jacobian_velocity(true, false, J, J.tra, pa, ch); pc -= half_damper_jacobian_velocity(true, false, J, J.rot, pa, ch);
            cp -= half_damper_jacobian_velocity(false, true, J, J.tra, pa, ch); cp -= half_damper_jacobian_velocity(false, true, J, J.rot, pa, ch);
            // two bodies may be linked by one joint only (tree), so '=' is '+='
            put(boff[J.parent], boff[J.child], pc);
            put(boff[J.child], boff[J.parent], cp);
        }
        // contacts (contacts/constraints.jl:60-76)
        for (size_t k = 0; k < contacts.size(); ++k) {
            Contact<T>& c = contacts[k];
            put(coff[k], coff[k], contact_constraint_jacobian(c));
            M comp = contact_complementarity(c); comp[0] -= mu; if (c.model == 0) comp[1] -= mu;   // complementarityμ: − μ·neutral_vector
            if (c.model == 2) for (int i = 1; i < 6; ++i) comp[i] -= mu;
            if (c.kind == 1) {      // two bodies: contacts/constraints.jl:60-70 for the parent and for the child (no body-body entry: system.jl:33-40)
                putv(coff[k], vcat(-comp, -ss_constraint(c)));
                for (int side = 0; side < 2; ++side) {
                    const int ib = side == 0 ? c.body : c.body2;
                    put(boff[ib], coff[k], hcat(M(6, c.nh()), -ss_impulse_map(side == 0, c)));
                    put(coff[k], boff[ib], vcat(M(c.nh(), 6), ss_constraint_jacobian_velocity(side == 0, c)));
                }
                continue;
            }
            putv(coff[k], vcat(-comp, -contact_constraint(c)));
            put(boff[c.body], coff[k], hcat(M(6, c.nh()), -contact_impulse_map(c)));
            put(coff[k], boff[c.body], vcat(M(c.nh(), 6), contact_constraint_jacobian_velocity(c)));
        }
    }

    // =====================================================================
    // VIOLATIONS   src/solver/violations.jl
    // =====================================================================
    T residual_violation() {
        T v = 0;
        for (auto& J : joints) {
            if (J.N() == 0) continue;
            M res = joint_constraint_full(J);
            int o = 2 * J.tra.Nb();
            for (int i = 0; i < J.tra.nl; ++i) v = std::fmax(v, std::fabs(res[o + i]));
            o = J.tra.N() + 2 * J.rot.Nb();
            for (int i = 0; i < J.rot.nl; ++i) v = std::fmax(v, std::fabs(res[o + i]));
        }
        for (size_t i = 0; i < bodies.size(); ++i) v = std::fmax(v, body_constraint((int)i).norm_inf());
        for (auto& c : contacts) v = std::fmax(v, (c.kind == 1 ? ss_constraint(c) : contact_constraint(c)).norm_inf());
        return v;
    }
    T bilinear_violation() {
        T v = 0;
        for (auto& J : joints) {   // complementarity(mechanism, joint)  complementarity.jl:2-13
            int o = 0;
            for (const Half<T>* h : {&J.tra, &J.rot}) {
                int Nb = h->Nb();
                for (int i = 0; i < Nb; ++i) v = std::fmax(v, std::fabs(J.imp[1][o + i] * J.imp[1][o + Nb + i]));
                o += h->N();
            }
        }
        for (auto& c : contacts) v = std::fmax(v, contact_complementarity(c).norm_inf());
        return v;
    }

    // =====================================================================
    // LINE SEARCHES / CENTERING / CORRECTION
    // =====================================================================
    static T positive_orthant_step_length(T lam, T dl, T tau) {   // line_search.jl:98-110 (scalar entries)
        return dl < 0 ? std::fmin(T(1), -tau * lam / dl) : T(1);
    }
    static T second_order_cone_step_length(const M& l, const M& d, T tau) {   // line_search.jl:112-139
        const T eps = T(1e-14);
        T l0 = l[0];
        T ll = std::fmax(l0 * l0 - (l[1] * l[1] + l[2] * l[2]), T(1e-25));
        ll += eps;
        T ld = l0 * d[0] - (l[1] * d[1] + l[2] * d[2]) + eps;
        T rs = ld / ll;
        T sq = std::sqrt(ll);
        T f = (ld / sq + d[0]) / (l0 / sq + T(1));
        T rv1 = d[1] / sq - f * l[1] / ll, rv2 = d[2] / sq - f * l[2] / ll;
        T nr = std::sqrt(rv1 * rv1 + rv2 * rv2);
        T a = T(1);
        if (nr - rs > T(0)) a = std::fmin(a, tau / (nr - rs));
        return a;
    }
    // cone_line_search!(mechanism; τort, τsoc)  line_search.jl:36-96  (Δ in b)
    T cone_line_search(T tort, T tsoc) {
        T a = T(1);
        for (size_t k = 0; k < contacts.size(); ++k) {
            Contact<T>& c = contacts[k]; const T* D = &b[coff[k]];
            const M& s = c.s[1]; const M& g = c.gam[1];
            if (c.model != 0) {   // line_search.jl:68-83: ImpactContact and LinearContact, all N½ pairs on the positive orthant
                const int nh = c.nh();
                for (int i = 0; i < nh; ++i) a = std::fmin(std::fmin(a, positive_orthant_step_length(s[i], D[i], tort)), positive_orthant_step_length(g[i], D[nh + i], tort));
                continue;
            }
            T as_ort = positive_orthant_step_length(s[0], D[0], tort);
            T ag_ort = positive_orthant_step_length(g[0], D[4], tort);
            T as_soc = second_order_cone_step_length(sub(s, 1, 3), M::vec({D[1], D[2], D[3]}), tsoc);
            T ag_soc = second_order_cone_step_length(sub(g, 1, 3), M::vec({D[5], D[6], D[7]}), tsoc);
            a = std::fmin(std::fmin(std::fmin(std::fmin(a, as_soc), ag_soc), as_ort), ag_ort);
        }
        for (size_t j = 0; j < joints.size(); ++j) {
            Joint<T>& J = joints[j]; int o = 0;
            for (const Half<T>* h : {&J.tra, &J.rot}) {
                int Nb = h->Nb();
                for (int i = 0; i < Nb; ++i) {
                    a = std::fmin(a, positive_orthant_step_length(J.imp[1][o + i], b[joff[j] + o + i], tort));
                    a = std::fmin(a, positive_orthant_step_length(J.imp[1][o + Nb + i], b[joff[j] + o + Nb + i], tort));
                }
                o += h->N();
            }
        }
        return a;
    }
    // centering!  centering.jl:1-49
    void centering(T aaff, T& nu_, T& nuaff) {
        T p0 = 0, p1 = 0, p2 = 0;
        for (size_t k = 0; k < contacts.size(); ++k) {
            Contact<T>& c = contacts[k]; const T* D = &b[coff[k]];
            const int nh = c.nh();
            for (int i = 0; i < nh; ++i) { p0 += c.s[1][i] * c.gam[1][i]; p1 += (c.s[1][i] + aaff * D[i]) * (c.gam[1][i] + aaff * D[nh + i]); }
            p2 += c.model == 0 ? T(2) : T(nh);  // cone_degree: N½ (contact.jl:203) / 2 for NonlinearContact (nonlinear.jl:101)
        }
        for (size_t j = 0; j < joints.size(); ++j) {
            Joint<T>& J = joints[j]; int o = 0;
            for (const Half<T>* h : {&J.tra, &J.rot}) {
                int Nb = h->Nb();
                for (int i = 0; i < Nb; ++i) {
                    T s = J.imp[1][o + i], g = J.imp[1][o + Nb + i];
                    p0 += s * g; p1 += (s + aaff * b[joff[j] + o + i]) * (g + aaff * b[joff[j] + o + Nb + i]);
                }
                p2 += T(Nb);
                o += h->N();
            }
        }
        nu_ = p0 / p2; nuaff = p1 / p2;   // NaN when there is no cone at all, exactly as in the reference
    }
    // correction!  correction.jl:1-45 (adds to the cached residual)
    void correction() {
        for (size_t k = 0; k < contacts.size(); ++k) {
            const T* D = &b[coff[k]]; T* r = &rcache[coff[k]];
            if (contacts[k].model != 0) { const int nh = contacts[k].nh(); for (int i = 0; i < nh; ++i) r[i] += -D[i] * D[nh + i] + mu; continue; }   // correction.jl:13-19
            M cp = cone_product(M::vec({D[1], D[2], D[3]}), M::vec({D[5], D[6], D[7]}));
            r[0] += -D[0] * D[4] + mu; r[1] += -cp[0] + mu; r[2] += -cp[1]; r[3] += -cp[2];
        }
        for (size_t j = 0; j < joints.size(); ++j) {
            Joint<T>& J = joints[j]; int o = 0;
            for (const Half<T>* h : {&J.tra, &J.rot}) {
                int Nb = h->Nb();
                for (int i = 0; i < Nb; ++i) rcache[joff[j] + o + i] += -b[joff[j] + o + i] * b[joff[j] + o + Nb + i] + mu;
                o += h->N();
            }
        }
    }
    // candidate_step!  line_search.jl:141-163
    void candidate_step(T alpha, int scale) {
        T f = T(1) / std::pow(T(2), T(scale)) * alpha;
        for (size_t k = 0; k < contacts.size(); ++k) {
            Contact<T>& c = contacts[k];
            const int nh = c.nh();
            for (int i = 0; i < nh; ++i) { c.s[1][i] = c.s[0][i] + f * b[coff[k] + i]; c.gam[1][i] = c.gam[0][i] + f * b[coff[k] + nh + i]; }
        }
        for (size_t j = 0; j < joints.size(); ++j) { Joint<T>& J = joints[j]; for (int i = 0; i < J.N(); ++i) J.imp[1][i] = J.imp[0][i] + f * b[joff[j] + i]; }
        T wmax = T(3.9) / (dt * dt);
        for (size_t i = 0; i < bodies.size(); ++i) {
            State<T>& s = bodies[i].st;
            for (int k = 0; k < 3; ++k) { s.vsol[1][k] = s.vsol[0][k] + f * b[boff[i] + k]; s.wsol[1][k] = s.wsol[0][k] + f * b[boff[i] + 3 + k]; }
            T wd = dot(s.wsol[1], s.wsol[1]);
            if (wd > wmax) s.wsol[1] = (wmax / wd) * s.wsol[1];
            if (dot(s.wsol[1], s.wsol[1]) > T(3.91) / (dt * dt)) excessive_w = true;   // error("Excessive angular velocity") line_search.jl:18-20
        }
    }
    // line_search!  line_search.jl:1-34
    long stat_ls_calls = 0, stat_ls_trials = 0;        // statistics for tools/ (how many residual evaluations a line search takes)
    void line_search(T alpha, T rvio, T bvio, T& rc, T& bc) {
        int scale = 0; rc = std::numeric_limits<T>::infinity(); bc = rc;
        ++stat_ls_calls;
        for (int it = 0; it < opts.max_ls; ++it) {
            ++stat_ls_trials;
            candidate_step(alpha, scale);
            rc = residual_violation(); bc = bilinear_violation();
            if (rc > rvio && bc > bvio) scale += 1; else return;
        }
    }

    // Elimination order of the block-sparse timing variant: the cone pairs of every contact and joint limit first (each
    // complementarity row paired with its γ column, each slack / constraint row with its s column, so that no pivot is a
    // structural zero), then the kinematic tree leaves -> root with every joint's equality rows right after its child body --
    // the order GraphBasedSystems' DFS gives the reference (contacts are leaves of the graph).
    void build_sparse_order() const {
        std::vector<int> pr, pc;
        for (size_t j = 0; j < joints.size(); ++j) { int o = joff[j];
            for (const Half<T>* h : {&joints[j].tra, &joints[j].rot}) { const int Nb = h->Nb();
                for (int i = 0; i < Nb; ++i) { pr.push_back(o + Nb + i); pc.push_back(o + i); }          // slack row  <-> s
                for (int i = 0; i < Nb; ++i) { pr.push_back(o + i); pc.push_back(o + Nb + i); }          // comp row   <-> γ
                o += h->N(); } }
        for (size_t k = 0; k < contacts.size(); ++k) { const int o = coff[k], nh = contacts[k].nh();
            if (nh == 1) { pr.push_back(o + 1); pc.push_back(o); pr.push_back(o); pc.push_back(o + 1); continue; }
            const int rws[8] = {4, 6, 7, 5, 0, 2, 3, 1}, cls[8] = {0, 2, 3, 5, 4, 6, 7, 1};              // (k1,s1) (k3,s3) (k4,s4) (k2,γ2) (c1,γ1) (c3,γ3) (c4,γ4) (c2,s2)
            for (int i = 0; i < 8; ++i) { pr.push_back(o + rws[i]); pc.push_back(o + cls[i]); } }
        std::vector<std::vector<int>> ch(bodies.size() + 1); std::vector<int> jof(bodies.size(), -1);
        for (size_t j = 0; j < joints.size(); ++j) { ch[joints[j].parent + 1].push_back(joints[j].child); jof[joints[j].child] = (int)j; }
        std::vector<int> stack, post; std::vector<char> seen(bodies.size() + 1, 0);
        std::function<void(int)> dfs = [&](int b) { for (int c : ch[b + 1]) dfs(c); if (b >= 0) post.push_back(b); };
        dfs(-1);
        for (int b : post) { for (int i = 0; i < 6; ++i) { pr.push_back(boff[b] + i); pc.push_back(boff[b] + i); }
            const int j = jof[b]; if (j < 0) continue; int o = joff[j];
            for (const Half<T>* h : {&joints[j].tra, &joints[j].rot}) { const int Nb = h->Nb(); for (int i = 0; i < h->nl; ++i) { pr.push_back(o + 2 * Nb + i); pc.push_back(o + 2 * Nb + i); } o += h->N(); } }
        splu.set_permutation(pr, pc);
    }
    // =====================================================================
    // mehrotra!   src/solver/mehrotra.jl:9-73
    // =====================================================================
    int mehrotra() {
        for (auto& c : contacts) reset_contact(c);
        for (auto& J : joints) reset_joint(J);
        int status = DOJO_STATUS_FAILED; mu = 0; T mutarget = 0; int no_progress = 0; T undercut = T(opts.undercut);
        excessive_w = false;
        for (auto& c : contacts) initialize_contact(c);
        set_entries();
        T bvio = bilinear_violation(), rvio = residual_violation();
        DenseLU<T> lu; int it = 0;
        for (it = 1; it <= opts.max_iter; ++it) {
            if (verbose) std::printf("%3d  bvio %.3e  rvio %.3e  alpha %.3e  mu %.3e\n", it, (double)bvio, (double)rvio, (double)last_alpha, (double)mutarget);
            if (rvio < T(opts.rtol) && bvio < T(opts.btol)) { status = DOJO_STATUS_SUCCESS; break; }
            rcache = b;                                   // pull_residual!
            {
            OpPause<T> pause_;                            // (oracle/counted.hpp: the linear solves are not counted with the assembly)
            if (sparse_solver) { if (splu.n != n) build_sparse_order(); splu.factor(A); splu.solve(b.data(), 1); }
            else {
            lu.factor(A, n);                              // ldu_factorization!
            lu.solve_refined(b.data(), 1, refine_steps);  // ldu_backsubstitution!  -> Δaff in b
            }
            }
            T aaff = cone_line_search(T(0.95), T(0.95));
            T nu_, nuaff; centering(aaff, nu_, nuaff);
            T ratio = nuaff / (nu_ + T(1e-20));
            T sc = std::fmin(std::fmax(ratio, T(0)), T(1)); sc = sc * sc * sc;   // clamp(...)^3 (NaN propagates like Julia's clamp)
            if (ratio != ratio) sc = ratio;
            mutarget = std::fmax(sc * nu_, T(opts.btol) / undercut);
            if (sc != sc) mutarget = T(opts.btol) / undercut;   // Julia: max(NaN, x) = NaN; only reachable with no cones, where μ is unused
            mu = mutarget;
            correction();
            b = rcache;                                   // push_residual!
            { OpPause<T> pause_; if (sparse_solver) splu.solve(b.data(), 1); else lu.solve_refined(b.data(), 1, refine_steps); }
            T tau = std::fmax(T(0.95), T(1) - std::fmax(rvio, bvio) * std::fmax(rvio, bvio));
            T alpha = cone_line_search(tau, std::fmin(tau, T(0.95)));
            last_alpha = alpha;
            T rv, bv; line_search(alpha, rvio, bvio, rv, bv);
            bool made = (!(rv < T(opts.rtol)) && rv < T(0.8) * rvio) || (!(bv < T(opts.btol)) && bv < T(0.8) * bvio);
            if (made) no_progress = std::max(no_progress - 1, 0); else no_progress += 1;
            rvio = rv; bvio = bv;
            if (no_progress >= opts.no_progress_max) undercut *= T(opts.no_progress_undercut);
            for (auto& B : bodies) { B.st.vsol[0] = B.st.vsol[1]; B.st.wsol[0] = B.st.wsol[1]; }   // update!
            for (auto& J : joints) J.imp[0] = J.imp[1];
            for (auto& c : contacts) { c.s[0] = c.s[1]; c.gam[0] = c.gam[1]; }
            set_entries();
        }
        last_iters = it > opts.max_iter ? opts.max_iter : it - 1;
        if (excessive_w) status = DOJO_STATUS_EXCESSIVE_W;
        return status;
    }

    // =====================================================================
    // SIMULATION API   src/simulation/step.jl, src/mechanism/set.jl, src/bodies/set.jl
    // =====================================================================
    void set_previous_configuration(State<T>& s) {   // bodies/set.jl:9-13
        s.x1 = next_position(s.x2, -s.v15, dt); s.q1 = next_orientation(s.q2, -s.w15, dt);
    }
    void set_maximal_state(const T* z) {   // mechanism/set.jl:10-25
        for (size_t i = 0; i < bodies.size(); ++i) {
            State<T>& s = bodies[i].st; const T* p = z + 13 * i;
            for (int k = 0; k < 3; ++k) { s.x2[k] = p[k]; s.v15[k] = p[3 + k]; s.w15[k] = p[10 + k]; }
            s.q2 = Q(p[6], p[7], p[8], p[9]);
        }
        for (auto& B : bodies) { set_previous_configuration(B.st); B.st.JF2 = M(3, 1); B.st.Jt2 = M(3, 1); }   // initialize_state!
        for (auto& B : bodies) { B.st.vsol[0] = B.st.v15; B.st.vsol[1] = B.st.v15; B.st.wsol[0] = B.st.w15; B.st.wsol[1] = B.st.w15; }   // set_velocity_solution!
    }
    void set_input_all(const T* u) {   // mechanism/set.jl:40-53
        int off = 0;
        for (auto& J : joints) { set_input(J, u + off); off += J.nu(); }
        for (auto& J : joints) input_impulse(J, true);
    }
    void update_state() {   // bodies/set.jl:22-36
        for (auto& B : bodies) {
            State<T>& s = B.st;
            s.x1 = s.x2; s.q1 = s.q2; s.v15 = s.vsol[1]; s.w15 = s.wsol[1];
            s.x2 = next_position(s.x2, s.vsol[1], dt); s.q2 = next_orientation(s.q2, s.wsol[1], dt);
            s.JF2 = M(3, 1); s.Jt2 = M(3, 1);
        }
    }
    // ---------------- minimal <-> maximal coordinates ----------------
    // maximal_to_minimal   src/mechanism/state.jl:44-66: per joint (mechanism.joints order) [c_tra; c_rot; v_tra; v_rot]
    void maximal_to_minimal(const T* z, T* x) const {
        int o = 0;
        for (const Joint<T>& J : joints) {
            M xa(3, 1), va(3, 1), wa(3, 1), xb(3, 1), vb(3, 1), wb(3, 1); Q qa, qb;
            auto unpack = [&](int i, M& x_, M& v_, Q& q_, M& w_) {          // unpack_maximal_state  state.jl:68-76
                const T* zi = z + 13 * i;
                for (int k = 0; k < 3; ++k) { x_[k] = zi[k]; v_[k] = zi[3 + k]; w_[k] = zi[10 + k]; }
                q_ = Q(zi[6], zi[7], zi[8], zi[9]);
            };
            unpack(J.child, xb, vb, qb, wb);
            if (J.parent >= 0) unpack(J.parent, xa, va, qa, wa);           // (the origin: zero position / velocity, identity attitude)
            const int nu = J.tra.nu() + J.rot.nu(); int oc = 0, ov = 0;
            for (const Half<T>* h : {&J.tra, &J.rot}) {
                const int n = h->nu();
                if (n == 0) continue;
                M c = minimal_coordinates(J, *h, xa, qa, xb, qb), v = minimal_velocities(J, *h, xa, va, qa, wa, xb, vb, qb, wb);
                for (int k = 0; k < n; ++k) { x[o + oc + k] = c[k]; x[o + nu + ov + k] = v[k]; }
                oc += n; ov += n;
            }
            o += 2 * nu;
        }
    }
    // minimal_to_maximal   src/mechanism/state.jl:9-22 -> set_minimal_coordinates_velocities!  src/joints/minimal.jl:134-196,
    // joints visited root -> leaves (the parent's maximal state first)
    void minimal_to_maximal(const T* x, T* z) const {
        const int Nb = (int)bodies.size();
        std::vector<int> xoff(joints.size()); { int o = 0; for (size_t j = 0; j < joints.size(); ++j) { xoff[j] = o; o += 2 * (joints[j].tra.nu() + joints[j].rot.nu()); } }
        std::vector<char> done(Nb, 0);
        for (int pass = 0, left = (int)joints.size(); left > 0 && pass <= (int)joints.size(); ++pass)
            for (size_t j = 0; j < joints.size(); ++j) {
                const Joint<T>& J = joints[j];
                if (done[J.child] || (J.parent >= 0 && !done[J.parent])) continue;
                const T* xm = x + xoff[j];
                const int nt = J.tra.nu(), nr = J.rot.nu(), nu = nt + nr;
                M dx(nt, 1), dth(nr, 1), dv(nt, 1), dw(nr, 1);
                for (int k = 0; k < nt; ++k) { dx[k] = xm[k]; dv[k] = xm[nu + k]; }
                for (int k = 0; k < nr; ++k) { dth[k] = xm[nt + k]; dw[k] = xm[nu + nt + k]; }
                M xa(3, 1), va(3, 1), wa(3, 1); Q qa;
                if (J.parent >= 0) { const T* zp = z + 13 * J.parent; for (int k = 0; k < 3; ++k) { xa[k] = zp[k]; va[k] = zp[3 + k]; wa[k] = zp[10 + k]; } qa = Q(zp[6], zp[7], zp[8], zp[9]); }
                M Arot = J.rot.A.t(), Atra = J.tra.A.t();                       // zerodimstaticadjoint(nullspace_mask(...))
                M ax = nr > 0 ? Arot * dth : M(3, 1), tx = nt > 0 ? Atra * dx : M(3, 1);
                // positions
                Q dq = axis_angle_to_quaternion(ax);
                Q qb = qa * J.qoff * dq;
                M xb = xa + vector_rotate(J.vp + tx, qa) - vector_rotate(J.vc, qb);
                // previous configuration, finite-difference configuration
                M xa1 = next_position(xa, -va, dt); Q qa1 = next_orientation(qa, -wa, dt);
                M tx1 = nt > 0 ? Atra * (dx - dt * dv) : M(3, 1);
                M axw = nr > 0 ? dt * (Arot * dw) : M(3, 1);
                Q dq1 = dq * inv(axis_angle_to_quaternion(axw));
                Q qb1 = qa1 * J.qoff * dq1;
                M xb1 = xa1 + vector_rotate(J.vp + tx1, qa1) - vector_rotate(J.vc, qb1);
                // finite-difference velocity: (xb − xb1)/Δt, angular_velocity(qb1, qb, Δt) = 2/Δt V Lᵀ(qb1) qb  (integrator.jl:22-24)
                M vb = (T(1) / dt) * (xb - xb1);
                M wb = (T(2) / dt) * (Vmat<T>() * (LTmat(qb1) * vector(qb)));
                T* zc = z + 13 * J.child;
                for (int k = 0; k < 3; ++k) { zc[k] = xb[k]; zc[3 + k] = vb[k]; zc[10 + k] = wb[k]; }
                zc[6] = qb.s; zc[7] = qb.v1; zc[8] = qb.v2; zc[9] = qb.v3;
                done[J.child] = 1; --left;
            }
    }
    void get_maximal_state(T* z) const {   // mechanism/get.jl:107-117
        for (size_t i = 0; i < bodies.size(); ++i) {
            const State<T>& s = bodies[i].st; T* p = z + 13 * i;
            for (int k = 0; k < 3; ++k) { p[k] = s.x2[k]; p[3 + k] = s.v15[k]; p[10 + k] = s.w15[k]; }
            p[6] = s.q2.s; p[7] = s.q2.v1; p[8] = s.q2.v2; p[9] = s.q2.v3;
        }
    }
    void get_next_state(T* z) const {   // mechanism/get.jl:126-134
        for (size_t i = 0; i < bodies.size(); ++i) {
            const State<T>& s = bodies[i].st; T* p = z + 13 * i;
            M x = x3(s); Q q = q3(s);
            for (int k = 0; k < 3; ++k) { p[k] = x[k]; p[3 + k] = s.vsol[1][k]; p[10 + k] = s.wsol[1][k]; }
            p[6] = q.s; p[7] = q.v1; p[8] = q.v2; p[9] = q.v3;
        }
    }
    // step!(mechanism, z, u)  simulation/step.jl:11-30.  z_state = internal state after update_state!
    // (the parity contract); z_return = the reference's literal (doubly advanced) return value.
    int step(const T* z, const T* u, T* z_state, T* z_return) {
        set_maximal_state(z);
        std::vector<T> zero(nu(), T(0));
        set_input_all(u ? u : zero.data());
        int status = mehrotra();
        update_state();
        if (z_state) get_maximal_state(z_state);
        if (z_return) get_next_state(z_return);
        return status;
    }
    // one iteration of simulate!'s loop  simulation/simulate.jl:25-33 (control already set through set_input / Fext)
    int simulate_step(const T* u, bool last, T* record = nullptr) {
        if (u) { int off = 0; for (auto& J : joints) { set_input(J, u + off); off += J.nu(); } }
        for (auto& J : joints) input_impulse(J, true);
        int status = mehrotra();
        for (auto& B : bodies) { B.st.Fext = M(3, 1); B.st.text = M(3, 1); }
        if (record) save_to_storage(record);                  // record && save_to_storage!  simulate.jl:31
        if (!last) update_state();
        return status;
    }
    // momentum(mechanism, body)  src/mechanics/momentum.jl:17-41: linear and (world-frame) angular momentum of one body
    // at the solved step, from the discrete Legendre transform D2 minus half of the input and joint impulses
    void body_momentum(int ib, M& p_lin, M& p_ang_world) const {
        const Body<T>& B = bodies[ib]; const State<T>& s = B.st;
        M x3_ = x3(s); Q q3_ = q3(s);
        M D2x = (T(1) / dt * B.mass) * (x3_ - s.x2) - (T(0.5) * dt) * (B.mass * gravity + s.Fext);
        M D2q = (T(-2) / dt) * (LVTmat(s.q2).t() * Tmat<T>() * Rmat(q3_).t() * VTmat<T>() * B.inertia * Vmat<T>() * Lmat(s.q2).t() * vector(q3_)) - (T(0.5) * dt) * s.text;
        p_lin = D2x - T(0.5) * s.JF2;
        M p_ang = D2q - T(0.5) * s.Jt2;
        for (auto& J : joints) {                              // joint_impulses  momentum.jl:43-53
            bool parent = (J.parent == ib), child = (J.child == ib);
            if (!parent && !child) continue;
            M f(6, 1);
            if (J.N() > 0) f += joint_impulse_map(J, parent) * J.imp[1];
            if (J.spring) f += joint_spring_impulses(J, parent, false);
            if (J.damper) f += joint_damper_impulses(J, parent, false);
            p_lin -= T(0.5) * sub(f, 0, 3); p_ang -= T(0.5) * sub(f, 3, 3);
        }
        p_ang_world = vector_rotate(p_ang, s.q2);
    }
    // save_to_storage!  src/simulation/storage.jl:50-67: one Storage row per body =
    // [x2(3) q2(4) v15(3) ω15(3) px(3) pq(3) vl(3) ωl(3)]  (25 scalars), from the body states as they are now
    void save_to_storage(T* out) const {
        for (size_t i = 0; i < bodies.size(); ++i) {
            const Body<T>& B = bodies[i]; const State<T>& s = B.st; T* p = out + 25 * i;
            M pl, pq; body_momentum((int)i, pl, pq);
            M vl = (T(1) / B.mass) * pl;
            M wb = vector_rotate(pq, inv(s.q2));
            std::vector<T> A(9); for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[3 * r + c] = B.inertia(r, c);
            T rhs[3] = {wb[0], wb[1], wb[2]};
            DenseLU<T> lu; lu.factor(A, 3); lu.solve(rhs, 1);                  // inertia \ (…)
            for (int k = 0; k < 3; ++k) { p[k] = s.x2[k]; p[7 + k] = s.v15[k]; p[10 + k] = s.w15[k]; p[13 + k] = pl[k]; p[16 + k] = pq[k]; p[19 + k] = vl[k]; p[22 + k] = rhs[k]; }
            p[3] = s.q2.s; p[4] = s.q2.v1; p[5] = s.q2.v2; p[6] = s.q2.v3;
        }
    }
    // kinetic_energy / potential_energy of one Storage row (src/mechanics/energy.jl:24-92): row = save_to_storage's [Nb][25].
    //   ke = Σ ½ m vl·vl + ½ ωlᵀ J ωl                               (:33-41, the momentum-derived velocities vl, ωl of the row)
    //   pe = −Σ m g·x  +  Σ_{joints with springs, halves with spring > 0} ½ |spring_force(:parent, ...)|² / spring     (:61-90)
    void energy_of_storage_row(const T* row, T& ke, T& pe) const {
        ke = T(0); pe = T(0);
        for (size_t i = 0; i < bodies.size(); ++i) {
            const Body<T>& B = bodies[i]; const T* p = row + 25 * i;
            M vl = M::vec({p[19], p[20], p[21]}), wl = M::vec({p[22], p[23], p[24]}), x = M::vec({p[0], p[1], p[2]});
            ke += T(0.5) * B.mass * dot(vl, vl) + T(0.5) * dot(wl, B.inertia * wl);
            pe -= B.mass * dot(gravity, x);
        }
        auto cfg = [&](int b, M& x, Q& q) {
            if (b < 0) { x = origin.x2; q = origin.q2; return; }       // current_configuration(mechanism.origin.state)
            const T* p = row + 25 * b; x = M::vec({p[0], p[1], p[2]}); q = Q(p[3], p[4], p[5], p[6]);
        };
        for (auto& J : joints) {
            if (!J.spring) continue;
            M xa, xb; Q qa, qb; cfg(J.parent, xa, qa); cfg(J.child, xb, qb);
            for (const Half<T>* h : {&J.tra, &J.rot}) {
                if (!(h->spring > T(0))) continue;
                M force;
                if (!h->is_rot) { M distance = h->spring_offset - minimal_coordinates(J, *h, xa, qa, xb, qb); force = h->spring * (h->A.t() * distance); }   // translational/springs.jl:5-15
                else force = rot_spring_force(true, J, *h, xa, qa, xb, qb, true, false);                                                                          // rotational/springs.jl:5-24
                pe += T(0.5) * dot(force, force) / h->spring;
            }
        }
    }
    void initialize_simulation() {   // simulate.jl:53-58
        for (auto& B : bodies) { set_previous_configuration(B.st); B.st.JF2 = M(3, 1); B.st.Jt2 = M(3, 1); }
        for (auto& B : bodies) { B.st.vsol[0] = B.st.v15; B.st.vsol[1] = B.st.v15; B.st.wsol[0] = B.st.w15; B.st.wsol[1] = B.st.w15; }
    }

    // get_solution / set_solution!   gradients/finite_difference.jl:1-41
    void get_solution(T* sol) const {
        int o = 0;
        for (auto& J : joints) for (int i = 0; i < J.N(); ++i) sol[o++] = J.imp[1][i];
        for (auto& B : bodies) { for (int k = 0; k < 3; ++k) sol[o++] = B.st.vsol[1][k]; for (int k = 0; k < 3; ++k) sol[o++] = B.st.wsol[1][k]; }
        for (auto& c : contacts) { for (int k = 0; k < c.nh(); ++k) sol[o++] = c.s[1][k]; for (int k = 0; k < c.nh(); ++k) sol[o++] = c.gam[1][k]; }
    }
    void set_solution(const T* sol) {
        int o = 0;
        for (auto& J : joints) for (int i = 0; i < J.N(); ++i) J.imp[1][i] = sol[o++];
        for (auto& B : bodies) { for (int k = 0; k < 3; ++k) B.st.vsol[1][k] = sol[o++]; for (int k = 0; k < 3; ++k) B.st.wsol[1][k] = sol[o++]; }
        for (auto& c : contacts) { for (int k = 0; k < c.nh(); ++k) c.s[1][k] = sol[o++]; for (int k = 0; k < c.nh(); ++k) c.gam[1][k] = sol[o++]; }
    }

    // =====================================================================
    // DATA (θ) get / set   src/mechanism/data.jl
    // =====================================================================
    int data_dim(bool attjac) const {
        int d = 0;
        for (auto& J : joints) d += J.nu() + 2;
        d += (int)bodies.size() * (attjac ? 19 : 20);
        d += (int)contacts.size() * 5;
        return d;
    }
    void get_data(T* data) const {   // data.jl:56-86
        int o = 0;
        for (auto& J : joints) {
            M ut = J.tra.nu() > 0 ? J.tra.A * J.tra.input : M(0, 1);
            M ur = J.rot.nu() > 0 ? J.rot.A * J.rot.input : M(0, 1);
            for (int i = 0; i < ut.size(); ++i) data[o++] = ut[i];
            for (int i = 0; i < ur.size(); ++i) data[o++] = ur[i];
            data[o++] = J.tra.spring; data[o++] = J.tra.damper;
        }
        for (auto& B : bodies) {
            data[o++] = B.mass;
            const M& I = B.inertia;
            data[o++] = I(0, 0); data[o++] = I(0, 1); data[o++] = I(0, 2); data[o++] = I(1, 1); data[o++] = I(1, 2); data[o++] = I(2, 2);
            for (int k = 0; k < 3; ++k) data[o++] = B.st.v15[k];
            for (int k = 0; k < 3; ++k) data[o++] = B.st.w15[k];
            for (int k = 0; k < 3; ++k) data[o++] = B.st.x2[k];
            data[o++] = B.st.q2.s; data[o++] = B.st.q2.v1; data[o++] = B.st.q2.v2; data[o++] = B.st.q2.v3;
        }
        for (auto& c : contacts) { data[o++] = c.mu; data[o++] = c.radius; for (int k = 0; k < 3; ++k) data[o++] = c.origin[k]; }
    }
    void set_data(const T* data) {   // data.jl:93-190
        int o = 0;
        for (auto& J : joints) {
            set_input(J, data + o); o += J.nu();
            T sp = data[o++], da = data[o++];
            J.tra.spring = sp; J.rot.spring = sp; J.tra.damper = da; J.rot.damper = da;
        }
        for (auto& B : bodies) {
            B.mass = data[o++];
            T j[6]; for (int k = 0; k < 6; ++k) j[k] = data[o++];
            B.inertia = M(3, 3, {j[0], j[1], j[2],  j[1], j[3], j[4],  j[2], j[4], j[5]});
            State<T>& s = B.st;
            for (int k = 0; k < 3; ++k) s.v15[k] = data[o++];
            for (int k = 0; k < 3; ++k) s.w15[k] = data[o++];
            for (int k = 0; k < 3; ++k) s.x2[k] = data[o++];
            s.q2 = Q(data[o], data[o + 1], data[o + 2], data[o + 3]); o += 4;
            s.x1 = next_position(s.x2, -s.v15, dt); s.q1 = next_orientation(s.q2, -s.w15, dt);
            s.JF2 = M(3, 1); s.Jt2 = M(3, 1);
        }
        for (auto& c : contacts) { c.mu = data[o++]; c.radius = data[o++]; for (int k = 0; k < 3; ++k) c.origin[k] = data[o++]; }
        for (auto& J : joints) input_impulse(J, false);
    }
    // evaluate_residual!  finite_difference.jl:43-49
    void evaluate_residual(const T* data, const T* sol, T* out) {
        set_data(data); set_solution(sol); set_entries();
        for (int i = 0; i < n; ++i) out[i] = b[i];
    }
    // data_attitude_jacobian  data.jl:27-49: (data_dim(attjac=false)) x (data_dim(attjac=true))
    void data_attitude_jacobian(std::vector<T>& G, int& nr, int& nc) const {
        nr = data_dim(false); nc = data_dim(true); G.assign((size_t)nr * nc, T(0));
        int r = 0, c = 0;
        for (auto& J : joints) for (int i = 0; i < J.nu() + 2; ++i) { G[(size_t)r * nc + c] = 1; ++r; ++c; }
        for (auto& B : bodies) {
            for (int i = 0; i < 16; ++i) { G[(size_t)r * nc + c] = 1; ++r; ++c; }
            M L = LVTmat(B.st.q2);
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) G[(size_t)(r + i) * nc + c + j] = L(i, j);
            r += 4; c += 3;
        }
        for (size_t k = 0; k < contacts.size(); ++k) for (int i = 0; i < 5; ++i) { G[(size_t)r * nc + c] = 1; ++r; ++c; }
    }

    // =====================================================================
    // DATA JACOBIANS   src/gradients/data.jl
    // =====================================================================
    static M dJp_dJ(const M& p) {   // ∂Jp∂J  gradients/utilities.jl:17-23
        return M(3, 6, {p[0], p[1], p[2], 0, 0, 0,   0, p[0], 0, p[1], p[2], 0,   0, 0, p[0], 0, p[1], p[2]});
    }
    // joint_constraint_jacobian_body_data  data.jl:4-14: N x 19
    M joint_constraint_jacobian_body_data(const Joint<T>& J, bool parent) const {
        const State<T>& s = bstate(parent ? J.parent : J.child);
        M dz2 = -(joint_constraint_jacobian_configuration_full(J, parent) * integrator_jacobian_configuration(s.q2, s.wsol[1], dt, true));
        return hcat(M(J.N(), 13), dz2);
    }
    // body_constraint_jacobian_body_data(mechanism, body)  data.jl:16-55: 6 x 19
    M body_constraint_jacobian_body_data(int ib) const {
        const Body<T>& B = bodies[ib]; const State<T>& s = B.st;
        M x3_ = x3(s); Q q3_ = q3(s);
        M dm = vcat((T(1) / dt) * (s.x2 - s.x1) + (dt / T(2)) * gravity - (T(1) / dt) * (x3_ - s.x2) + (dt / T(2)) * gravity, M(3, 1));
        M dJ = (T(2) / dt) * (LVTmat(s.q2).t() * LVTmat(s.q1) * dJp_dJ(VLTmat(s.q1) * vector(s.q2)));
        dJ += (T(2) / dt) * (LVTmat(s.q2).t() * Tmat<T>() * RTVTmat(q3_) * dJp_dJ(VLTmat(s.q2) * vector(q3_)));
        dJ = vcat(M(3, 6), dJ);
        M dv15 = B.mass * M::eye(3);
        M dq1 = (T(-2) / dt) * (LVTmat(s.q2).t() * dLVTmat_dq(B.inertia * VLTmat(s.q1) * vector(s.q2)));
        dq1 += (T(-2) / dt) * (LVTmat(s.q2).t() * LVTmat(s.q1) * B.inertia * dVLTmat_dq(vector(s.q2)));
        M dw15 = dq1 * rotational_integrator_jacobian_velocity(s.q2, -s.w15, dt);
        M d15(6, 6); d15.set_block(0, 0, dv15); d15.set_block(3, 3, dw15);
        // ∇z2 is multiplied by 0.0 in the reference (data.jl:54)
        return hcat(hcat(hcat(dm, dJ), d15), M(6, 6));
    }
    // body_constraint_jacobian_body_data(mechanism, pbody, cbody, joint)  data.jl:57-124: (∇11, ∇12), each 6 x 19
    // `self` is the body whose residual rows are differentiated; `self_is_parent` says which end of the joint it is.
    void body_constraint_jacobian_body_data_joint(const Joint<T>& J, bool self_is_parent, M& d11, M& d12) const {
        const State<T>& pa = bstate(J.parent); const State<T>& ch = bstate(J.child);
        M aa(6, 6), ab(6, 6);
        const Half<T>* hs[2] = {&J.tra, &J.rot}; int o = 0;
        for (int i = 0; i < 2; ++i) {
            M lam = sub(J.imp[1], o, hs[i]->N()); o += hs[i]->N();   // get_joint_impulses
            aa += half_impulse_map_jacobian(self_is_parent, self_is_parent, J, *hs[i], pa, ch, lam);
            ab += half_impulse_map_jacobian(self_is_parent, !self_is_parent, J, *hs[i], pa, ch, lam);
        }
        if (J.spring) for (int i = 0; i < 2; ++i) {
            aa += half_spring_jacobian_configuration(self_is_parent, self_is_parent, J, *hs[i], pa, ch);
            ab += half_spring_jacobian_configuration(self_is_parent, !self_is_parent, J, *hs[i], pa, ch);
        }
        if (J.damper) for (int i = 0; i < 2; ++i) {
            aa += half_damper_jacobian_configuration(self_is_parent, self_is_parent, J, *hs[i], pa, ch);
            ab += half_damper_jacobian_configuration(self_is_parent, !self_is_parent, J, *hs[i], pa, ch);
        }
        d11 = hcat(M(6, 13), aa); d12 = hcat(M(6, 13), ab);
    }
    // body_constraint_jacobian_body_data(mechanism, body, contact)  data.jl:126-135: 6 x 19
    M body_constraint_jacobian_body_data_contact(const Contact<T>& c) const {
        const State<T>& s = bodies[c.body].st;
        M dz3 = contact_impulse_map_jacobian(c);
        return hcat(M(6, 13), dz3 * integrator_jacobian_configuration(s.q2, s.wsol[1], dt, true));
    }
    // body_constraint_jacobian_joint_data  data.jl:137-150: 6 x (nu_j + 2)
    M body_constraint_jacobian_joint_data(const Joint<T>& J, bool parent) const {
        const State<T>& pa = bstate(J.parent); const State<T>& ch = bstate(J.child);
        M du = hcat(half_input_jacobian_control(parent, J, J.tra, pa, ch), half_input_jacobian_control(parent, J, J.rot, pa, ch));
        M dsp = J.spring ? joint_spring_impulses(J, parent, true) : M(6, 1);
        M dda = J.damper ? joint_damper_impulses(J, parent, true) : M(6, 1);
        return hcat(hcat(du, dsp), dda);
    }
    // body_constraint_jacobian_contact_data  data.jl:152-171: 6 x 5
    M body_constraint_jacobian_contact_data(const Contact<T>& c) const {
        const State<T>& s = bodies[c.body].st;
        Q qp3 = q3(s); const M& g = c.gam[1];
        M X = hcat(hcat(c.normal.t(), M(3, 1)), c.tangent.t());
        M dp = -dskew_dp(VRmat(qp3) * LTVTmat(qp3) * X * g);
        M drad = (-dskew_dp(VRmat(qp3) * LTVTmat(qp3) * X * g)) * ((-rotation_matrix(inv(qp3))) * c.normal.t());
        M dQ = -hcat(hcat(M(3, 1), drad), dp);
        return vcat(M(3, 5), dQ);
    }
    // contact_constraint_jacobian_contact_data  data.jl:173-192: 8 x 5
    M contact_constraint_jacobian_contact_data(const Contact<T>& c) const {
        const State<T>& s = bodies[c.body].st;
        Q qp3 = q3(s); const M& wp = s.wsol[1]; const M& g = c.gam[1];
        M dmu = M::vec({0, g[0], 0, 0});
        M Sw = skew(vector_rotate(wp, qp3));
        M drad = vcat(vcat(-c.normal, M(1, 3)), (-c.tangent) * Sw) * c.normal.t();
        M dp = vcat(vcat(c.normal * rotation_matrix(qp3), M(1, 3)), c.tangent * Sw * rotation_matrix(qp3));
        M dg = -hcat(hcat(dmu, drad), dp);
        return vcat(M(4, 5), dg);
    }
    // contact_constraint_jacobian_body_data  data.jl:194-205: 8 x 19
    M contact_constraint_jacobian_body_data(const Contact<T>& c) const {
        const State<T>& s = bodies[c.body].st;
        M dz3 = -contact_constraint_jacobian_configuration(c);
        M dz2 = dz3 * integrator_jacobian_configuration(s.q2, s.wsol[1], dt, true);
        return vcat(M(4, 19), hcat(M(4, 13), dz2));
    }
    // jacobian_data! + dense export  data.jl:281-355, state.jl:95: n x data_dim(attjac=true)
    void jacobian_data(std::vector<T>& Dm, int& nd) const {
        nd = data_dim(true); Dm.assign((size_t)n * nd, T(0));
        std::vector<int> jc, bc, cc; int o = 0;
        for (auto& J : joints) { jc.push_back(o); o += J.nu() + 2; }
        for (size_t i = 0; i < bodies.size(); ++i) { bc.push_back(o); o += 19; }
        for (size_t i = 0; i < contacts.size(); ++i) { cc.push_back(o); o += 5; }
        auto add = [&](int r0, int c0, const M& blk) { for (int i = 0; i < blk.r; ++i) for (int j = 0; j < blk.c; ++j) Dm[(size_t)(r0 + i) * nd + c0 + j] += blk(i, j); };
        // jacobian_contact_data!
        for (size_t k = 0; k < contacts.size(); ++k) {
            add(boff[contacts[k].body], cc[k], body_constraint_jacobian_contact_data(contacts[k]));
            add(coff[k], cc[k], contact_constraint_jacobian_contact_data(contacts[k]));
        }
        // jacobian_body_data!
        for (size_t j = 0; j < joints.size(); ++j) {
            const Joint<T>& J = joints[j];
            if (J.parent >= 0) add(joff[j], bc[J.parent], joint_constraint_jacobian_body_data(J, true));
            add(joff[j], bc[J.child], joint_constraint_jacobian_body_data(J, false));
        }
        for (size_t i = 0; i < bodies.size(); ++i) {
            add(boff[i], bc[i], body_constraint_jacobian_body_data((int)i));
            for (auto& J : joints) {
                bool isp = (J.parent == (int)i), isc = (J.child == (int)i);
                if (!isp && !isc) continue;
                M d11, d12; body_constraint_jacobian_body_data_joint(J, isp, d11, d12);
                add(boff[i], bc[i], d11);
                int other = isp ? J.child : J.parent;
                if (other >= 0) add(boff[i], bc[other], d12);
            }
        }
        for (auto& c : contacts) add(boff[c.body], bc[c.body], body_constraint_jacobian_body_data_contact(c));
        for (size_t k = 0; k < contacts.size(); ++k) add(coff[k], bc[contacts[k].body], contact_constraint_jacobian_body_data(contacts[k]));
        // jacobian_joint_data!
        for (size_t j = 0; j < joints.size(); ++j) {
            const Joint<T>& J = joints[j];
            if (J.parent >= 0) add(boff[J.parent], jc[j], body_constraint_jacobian_joint_data(J, true));
            add(boff[J.child], jc[j], body_constraint_jacobian_joint_data(J, false));
        }
    }

    // =====================================================================
    // get_maximal_gradients   src/gradients/state.jl:78-126
    // The caller decides which body states the data blocks see:
    //   reference mode : call after step() (post-update_state! states), literal reference behaviour
    //   consistent mode: call with pre-update states (see capi.cpp)
    // `solmat` is mechanism.system's matrix from the last set_entries! of mehrotra! (pre-update).
    // =====================================================================
    void get_maximal_gradients(const std::vector<T>& solmat, T* jac_state, T* jac_control) const {
        int Nb = (int)bodies.size(), nx = 12 * Nb, nu_ = nu(), nd;
        std::vector<T>& Dm = w_Dm; jacobian_data(Dm, nd);      // (workspaces kept between calls: oracle_math.hpp, DenseLU::solve_refined)
        // columns: state [x2(14:16) v15(8:10) φ2(17:19) ω15(11:13)] per body, control 1:nu_j per joint
        std::vector<int> cols;
        int o = 0; std::vector<int> jc, bc;
        for (auto& J : joints) { jc.push_back(o); o += J.nu() + 2; }
        for (int i = 0; i < Nb; ++i) { bc.push_back(o); o += 19; }
        for (int i = 0; i < Nb; ++i) { for (int k : {13, 14, 15, 7, 8, 9, 16, 17, 18, 10, 11, 12}) cols.push_back(bc[i] + k); }
        for (size_t j = 0; j < joints.size(); ++j) for (int k = 0; k < joints[j].nu(); ++k) cols.push_back(jc[j] + k);
        int nc = (int)cols.size();
        std::vector<T>& R = w_R; R.resize((size_t)n * nc);
        for (int r = 0; r < n; ++r) for (int c = 0; c < nc; ++c) R[(size_t)r * nc + c] = Dm[(size_t)r * nd + cols[c]];
        { OpPause<T> pause_;
        if (sparse_solver) { if (splu.n != n) build_sparse_order(); splu.factor(solmat); splu.solve(R.data(), nc); }
        else { DenseLU<T>& lu = w_lu; lu.factor(solmat, n); lu.solve_refined(R.data(), nc, refine_steps); } }   // data_jacobian = solmat \ datamat
        std::fill(jac_state, jac_state + (size_t)nx * nx, T(0));
        std::fill(jac_control, jac_control + (size_t)nx * nu_, T(0));
        auto out = [&](int row, int c) -> T& { return c < nx ? jac_state[(size_t)row * nx + c] : jac_control[(size_t)row * nu_ + (c - nx)]; };
        for (int i = 0; i < Nb; ++i) {
            const State<T>& s = bodies[i].st;
            Q q3_ = next_orientation(s.q2, s.wsol[1], dt);
            M rw = LVTmat(q3_).t() * rotational_integrator_jacobian_velocity(s.q2, s.wsol[1], dt);      // 3x3
            M rq = LVTmat(q3_).t() * rotational_integrator_jacobian_orientation(s.q2, s.wsol[1], dt, true); // 3x3
            for (int c = 0; c < nc; ++c) {
                const T* dv = nullptr; (void)dv;
                T d[6]; for (int k = 0; k < 6; ++k) d[k] = R[(size_t)(boff[i] + k) * nc + c];
                for (int k = 0; k < 3; ++k) { out(12 * i + 3 + k, c) += d[k]; out(12 * i + 9 + k, c) += d[3 + k]; }
                for (int k = 0; k < 3; ++k) out(12 * i + k, c) += dt * d[k];
                for (int k = 0; k < 3; ++k) { T acc = 0; for (int l = 0; l < 3; ++l) acc += rw(k, l) * d[3 + l]; out(12 * i + 6 + k, c) += acc; }
            }
            for (int k = 0; k < 3; ++k) jac_state[(size_t)(12 * i + k) * nx + 12 * i + k] += T(1);
            for (int k = 0; k < 3; ++k) for (int l = 0; l < 3; ++l) jac_state[(size_t)(12 * i + 6 + k) * nx + 12 * i + 6 + l] += rq(k, l);
        }
    }

    // get_contact_gradients   src/gradients/contact.jl:1-55: d(x3, v25, φ3, ω25)/d(contact data θ), θ per contact =
    // [friction_coefficient, contact_radius, contact_origin(3)]: the contact-data columns of solmat \ datamat through the
    // same integrator chain as get_maximal_gradients (no identity terms).  jac_contact: 12Nb x 5Nc, row-major.
    void get_contact_gradients(const std::vector<T>& solmat, T* jac_contact) const {
        int Nb = (int)bodies.size(), nx = 12 * Nb, nd;
        std::vector<T> Dm; jacobian_data(Dm, nd);
        int o = 0;
        for (auto& J : joints) o += J.nu() + 2;
        o += 19 * Nb;
        int nc = 5 * (int)contacts.size();
        std::vector<T> R((size_t)n * std::max(nc, 1));
        for (int r = 0; r < n; ++r) for (int c = 0; c < nc; ++c) R[(size_t)r * nc + c] = Dm[(size_t)r * nd + o + c];
        { OpPause<T> pause_; DenseLU<T> lu; lu.factor(solmat, n); if (nc > 0) lu.solve_refined(R.data(), nc, refine_steps); }
        std::fill(jac_contact, jac_contact + (size_t)nx * nc, T(0));
        for (int i = 0; i < Nb; ++i) {
            const State<T>& s = bodies[i].st;
            Q q3_ = next_orientation(s.q2, s.wsol[1], dt);
            M rw = LVTmat(q3_).t() * rotational_integrator_jacobian_velocity(s.q2, s.wsol[1], dt);
            for (int c = 0; c < nc; ++c) {
                T d[6]; for (int k = 0; k < 6; ++k) d[k] = R[(size_t)(boff[i] + k) * nc + c];
                for (int k = 0; k < 3; ++k) { jac_contact[(size_t)(12 * i + 3 + k) * nc + c] += d[k]; jac_contact[(size_t)(12 * i + 9 + k) * nc + c] += d[3 + k]; }
                for (int k = 0; k < 3; ++k) jac_contact[(size_t)(12 * i + k) * nc + c] += dt * d[k];
                for (int k = 0; k < 3; ++k) { T acc = 0; for (int l = 0; l < 3; ++l) acc += rw(k, l) * d[3 + l]; jac_contact[(size_t)(12 * i + 6 + k) * nc + c] += acc; }
            }
        }
    }
};

} // namespace orc
