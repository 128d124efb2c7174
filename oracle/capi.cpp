// capi.cpp -- C entry points of the CPU oracle (liboracle.so), loaded with ctypes by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg ONLY.
// All arrays at this boundary are fp64 row-major; `dtype` selects the arithmetic the
// oracle computes in (0 = fp64, 1 = fp32).
#include "dojo_oracle.hpp"
#include <thread>
#include <atomic>
#include <chrono>
#include <memory>
#include <algorithm>
#include <cstdio>
#include <sched.h>
#include <pthread.h>
#include <malloc.h>

using namespace orc;

namespace {

struct IOracle {
    virtual ~IOracle() {}
    virtual void set_options(const DojoSolverOptions& o) = 0;
    virtual void dims(int* out) = 0;
    virtual int step(const double* z, const double* u, double* z_state, double* z_return, int* iters) = 0;
    virtual void get_solution(double* sol) = 0;
    virtual void set_solution(const double* sol) = 0;
    virtual void gradients(int mode, double* dz, double* du) = 0;
    virtual void contact_gradients(int mode, double* dc) = 0;
    virtual void get_data(double* d) = 0;
    virtual void set_data(const double* d) = 0;
    virtual void evaluate_residual(const double* data, const double* sol, double* out) = 0;
    virtual void full_matrix(double* out) = 0;
    virtual void data_matrix(double* out) = 0;
    virtual void data_attjac(double* out) = 0;
    virtual void set_state(const double* z) = 0;
    virtual void get_state(double* z) = 0;
    virtual void set_external_force(int body, const double* force, const double* torque, const double* vertex) = 0;
    virtual int simulate_step(const double* u, int last) = 0;
    virtual int simulate_step_record(const double* u, int last, double* row) = 0;
    virtual void body_velocity_solution(double* v) = 0;
    virtual void save_to_storage(double* out) = 0;
    virtual void energy_of_storage_row(const double* row, double* ke_pe) = 0;
    virtual void debug_assemble(const double* z, const double* u, double* A, double* b) = 0;
    virtual void check_solution(const double* z, const double* u, const double* sol, double* viol) = 0;
    virtual IOracle* clone() = 0;
    virtual void set_refine_steps(int n) = 0;
    virtual void set_sparse_solver(int on) = 0;
    virtual long long sparse_flops() = 0;
    virtual long long sparse_solve_flops() = 0;
    virtual void ls_stats(long long* out) = 0;
    virtual int joint_unit(int joint, int half, int what, const double* in, double* out) = 0;
    virtual int contact_unit(int contact, int what, const double* in, double* out) = 0;
    virtual void input_impulses(const double* z, const double* u, double* jf) = 0;
    virtual void maximal_to_minimal(const double* z, double* x) = 0;
    virtual void minimal_to_maximal(const double* x, double* z) = 0;
};

template <class T>
struct OracleT : IOracle {
    Mechanism<T> m;
    std::vector<State<T>> pre;   // body states before update_state! of the last step()
    explicit OracleT(const DojoTopology& tp) : m(tp) {}
    std::vector<T> cast(const double* p, int n) { std::vector<T> v(n); for (int i = 0; i < n; ++i) v[i] = T(p[i]); return v; }
    void set_options(const DojoSolverOptions& o) override {
        m.opts.rtol = o.rtol; m.opts.btol = o.btol; m.opts.undercut = o.undercut; m.opts.no_progress_undercut = o.no_progress_undercut;
        m.opts.max_iter = o.max_iter; m.opts.max_ls = o.max_ls; m.opts.no_progress_max = o.no_progress_max;
    }
    void set_refine_steps(int n) override { m.refine_steps = n; }
    void set_sparse_solver(int on) override { m.sparse_solver = on != 0; }
    long long sparse_flops() override { return m.splu.flops_factor; }
    long long sparse_solve_flops() override { return m.splu.flops_solve; }
    void ls_stats(long long* out) override { out[0] = m.stat_ls_calls; out[1] = m.stat_ls_trials; }
    // unit functions of one joint half at given configurations in = [xa(3) qa(4) xb(3) qb(4)]:
    //   what 0: displacement (translational/minimal.jl:4-12; rotational/minimal.jl:4-11, vmat = true) -> 3
    //   what 1 / 2: displacement_jacobian_configuration(:parent / :child, ...; attjac = true) -> [X Q] 3x6   (joints/joint.jl:141-153)
    int joint_unit(int joint, int half, int what, const double* in, double* out) override {
        using M = orc::SM<T>; using Q = orc::Quat<T>;
        const orc::Joint<T>& J = m.joints[joint]; const orc::Half<T>& h = half ? J.rot : J.tra;
        const M xa = M::vec({(T)in[0], (T)in[1], (T)in[2]}), xb = M::vec({(T)in[7], (T)in[8], (T)in[9]});
        const Q qa = Q((T)in[3], (T)in[4], (T)in[5], (T)in[6]), qb = Q((T)in[10], (T)in[11], (T)in[12], (T)in[13]);
        if (what == 0) {
            M d = half ? orc::Vmat(m.rot_displacement_q(J, qa, qb)) : m.tra_displacement(J, xa, qa, xb, qb);
            for (int i = 0; i < 3; ++i) out[i] = (double)d[i];
            return 3;
        }
        if (what >= 19) {
            // what 19 / 20: impulse_map(:parent / :child, joint half, xa, qa, xb, qb, η) * λ -> 6   (joints/joint.jl:67-86), λ = in[14 : 14 + N]
            // what 21..24: impulse_map_jacobian(relative, jacobian, joint half, pbody, cbody, λ) -> 6x6, (relative, jacobian) = pp, pc, cp, cc
            //              (joints/impulses.jl:13-21) -- what test/impulse_map.jl:171-292 ("Impulse map") differentiates
            const int N = h.N();                          // impulses_length: Nλ + 2 Nb (the projector's columns; <= 15: the p + velocity slots of `in`)
            M lam(N, 1); for (int i = 0; i < N; ++i) lam[i] = (T)in[14 + i];
            if (what <= 20) { M r = m.half_impulse_map(what == 19, J, h, xa, qa, xb, qb) * lam; for (int i = 0; i < 6; ++i) out[i] = (double)r[i]; return 6; }
            if (what > 24) return -1;
            orc::State<T> pa, ch;
            pa.x2 = xa; pa.q2 = qa; ch.x2 = xb; ch.q2 = qb;
            const int k = what - 21;
            M Jm = m.half_impulse_map_jacobian(k < 2, (k % 2) == 0, J, h, pa, ch, lam);
            for (int i = 0; i < 36; ++i) out[i] = (double)Jm.a[i];
            return 36;
        }
        if (what >= 9) {
            // dampers of the joint half at the states in = [xa qa xb qb | p(3, unused) | va ωa vb ωb] (candidate velocities: vsol[2], ωsol[2]):
            //   what 9 / 10: damper_impulses(:parent / :child, ...) = timestep * damper_force(...; rotate = true, unitary = false) -> 6
            //                (rotational/dampers.jl:4-30, translational/dampers.jl:5-37)
            //   what 11..14: damper_jacobian_configuration(relative, jacobian, ...) -> 6x6, (relative, jacobian) = pp, pc, cp, cc
            //   what 15..18: damper_jacobian_velocity(relative, jacobian, ...)      -> 6x6     (dampers.jl:36-84 / :70-123)
            orc::State<T> pa, ch;
            pa.x2 = xa; pa.q2 = qa; ch.x2 = xb; ch.q2 = qb;
            pa.vsol[1] = M::vec({(T)in[17], (T)in[18], (T)in[19]}); pa.wsol[1] = M::vec({(T)in[20], (T)in[21], (T)in[22]});
            ch.vsol[1] = M::vec({(T)in[23], (T)in[24], (T)in[25]}); ch.wsol[1] = M::vec({(T)in[26], (T)in[27], (T)in[28]});
            if (what <= 10) { M r = m.half_damper_impulses(what == 9, J, h, pa, ch, false); for (int i = 0; i < 6; ++i) out[i] = (double)r[i]; return 6; }
            const int k = (what - 11) % 4; const bool rel_parent = k < 2, jac_parent = (k % 2) == 0;
            M Jm = what <= 14 ? m.half_damper_jacobian_configuration(rel_parent, jac_parent, J, h, pa, ch) : m.half_damper_jacobian_velocity(rel_parent, jac_parent, J, h, pa, ch);
            for (int i = 0; i < 36; ++i) out[i] = (double)Jm.a[i];
            return 36;
        }
        if (what >= 3) {
            // what 3 / 4: impulse_transform(:parent / :child, ...) * p -> 6 (joints/impulses.jl:4-8), p = in[14:17]
            // what 5..8: impulse_transform_jacobian(relative, jacobian, ..., p) -> 6x6, (relative, jacobian) = (parent,parent), (parent,child),
            //            (child,parent), (child,child)   (translational/impulses.jl:9-46, rotational/impulses.jl:9-39)
            const M p = M::vec({(T)in[14], (T)in[15], (T)in[16]});
            if (what <= 4) { M r = m.impulse_transform(what == 3, J, h, xa, qa, xb, qb) * p; for (int i = 0; i < 6; ++i) out[i] = (double)r[i]; return 6; }
            const int k = what - 5;
            M Jm = m.impulse_transform_jacobian(k < 2, (k % 2) == 0, J, h, xa, qa, xb, qb, p);
            for (int i = 0; i < 36; ++i) out[i] = (double)Jm.a[i];
            return 36;
        }
        M X, Qm; m.disp_jac(what == 1, J, h, xa, qa, xb, qb, true, X, Qm);
        M XQ = orc::hcat(X, Qm);
        for (int i = 0; i < 18; ++i) out[i] = (double)XQ.a[i];
        return 18;
    }
    // unit functions of one body-body contact's collision at given configurations in = [xp(3) qp(4) xc(3) qc(4)]
    // (src/contacts/collisions/{collision,sphere_sphere}.jl; what test/collisions.jl:59-170 differentiates):
    //   0 distance -> 1    1 / 2 contact_point(:parent / :child) -> 3    3 contact_normal -> 3    4 contact_tangent -> 2x3
    //   10 / 11 ∂distance∂x(:parent / :child) -> 1x3        12 / 13 ∂distance∂q -> 1x4
    //   14..17 ∂contact_point∂x(relative, jacobian) -> 3x3, (relative, jacobian) = pp, pc, cp, cc      18..21 ∂contact_point∂q -> 3x4
    //   22 / 23 ∂contact_normal_transpose∂x -> 3x3          24 / 25 ∂contact_normal_transpose∂q -> 3x4
    //   26 / 27, 28 / 29 ∂contact_tangent_one / two_transpose∂x -> 3x3      30 / 31, 32 / 33 ... ∂q -> 3x4
    int contact_unit(int contact, int what, const double* in, double* out) override {
        using M = orc::SM<T>; using Q = orc::Quat<T>;
        const orc::Contact<T>& c = m.contacts[contact];
        if (c.kind != 1) return -1;
        typename orc::Mechanism<T>::SS k{M::vec({(T)in[0], (T)in[1], (T)in[2]}), M::vec({(T)in[7], (T)in[8], (T)in[9]}),
                                         Q((T)in[3], (T)in[4], (T)in[5], (T)in[6]), Q((T)in[10], (T)in[11], (T)in[12], (T)in[13])};
        auto put = [&](const M& a) { for (int i = 0; i < a.r * a.c; ++i) out[i] = (double)a.a[i]; return a.r * a.c; };
        const bool jp = (what % 2) == 0;
        switch (what) {
            case 0: out[0] = (double)m.ss_distance(c, k); return 1;
            case 1: case 2: return put(m.ss_contact_point(what == 1, c, k));
            case 3: return put(m.ss_normal(c, k));
            case 4: return put(m.ss_tangent(c, k));
            case 10: case 11: return put(m.ss_dd_dx(jp, c, k));
            case 12: case 13: return put(m.ss_dd_dq(jp, c, k));
            case 14: case 15: case 16: case 17: return put(m.ss_dcp_dx(what < 16, jp, c, k));
            case 18: case 19: case 20: case 21: return put(m.ss_dcp_dq(what < 20, jp, c, k));
            case 22: case 23: return put(m.ss_dnT_dx(jp, c, k));
            case 24: case 25: return put(m.ss_dnT_dq(jp, c, k));
            case 26: case 27: return put(m.ss_dt1T_dx(jp, c, k));
            case 28: case 29: return put(m.ss_dt2T_dx(jp, c, k));
            case 30: case 31: return put(m.ss_dt1T_dq(jp, c, k));
            case 32: case 33: return put(m.ss_dt2T_dq(jp, c, k));
        }
        return -1;
    }
    void maximal_to_minimal(const double* z, double* x) override {
        int nz = 13 * (int)m.bodies.size(), nm = 2 * m.nu();
        std::vector<T> zz = cast(z, nz), xx(nm); m.maximal_to_minimal(zz.data(), xx.data()); for (int i = 0; i < nm; ++i) x[i] = (double)xx[i];
    }
    void minimal_to_maximal(const double* x, double* z) override {
        int nz = 13 * (int)m.bodies.size(), nm = 2 * m.nu();
        std::vector<T> xx = cast(x, nm), zz(nz); m.minimal_to_maximal(xx.data(), zz.data()); for (int i = 0; i < nz; ++i) z[i] = (double)zz[i];
    }
    // set_maximal_state! + set_input! (src/mechanism/set.jl:10-53): what the bodies hold when mehrotra! starts, [JF2; Jτ2] per body
    void input_impulses(const double* z, const double* u, double* jf) override {
        int nz = 13 * (int)m.bodies.size(), nu = m.nu();
        std::vector<T> zz = cast(z, nz), uu = cast(u, nu);
        m.set_maximal_state(zz.data());
        m.set_input_all(uu.data());
        for (size_t i = 0; i < m.bodies.size(); ++i) for (int k = 0; k < 3; ++k) { jf[6 * i + k] = (double)m.bodies[i].st.JF2[k]; jf[6 * i + 3 + k] = (double)m.bodies[i].st.Jt2[k]; }
    }
    void dims(int* out) override {
        out[0] = m.n; out[1] = m.nu(); out[2] = m.data_dim(false); out[3] = m.data_dim(true);
        out[4] = (int)m.bodies.size(); out[5] = (int)m.joints.size(); out[6] = (int)m.contacts.size();
    }
    int step(const double* z, const double* u, double* z_state, double* z_return, int* iters) override {
        int nz = 13 * (int)m.bodies.size(), nu = m.nu();
        std::vector<T> zz = cast(z, nz), uu(nu, T(0));
        if (u) uu = cast(u, nu);
        m.set_maximal_state(zz.data());
        m.set_input_all(uu.data());
        int status = m.mehrotra();
        pre.clear(); for (auto& B : m.bodies) pre.push_back(B.st);
        m.update_state();
        std::vector<T> o(nz);
        if (z_state) { m.get_maximal_state(o.data()); for (int i = 0; i < nz; ++i) z_state[i] = (double)o[i]; }
        if (z_return) { m.get_next_state(o.data()); for (int i = 0; i < nz; ++i) z_return[i] = (double)o[i]; }
        if (iters) *iters = m.last_iters;
        return status;
    }
    void get_solution(double* sol) override { std::vector<T> s(m.n); m.get_solution(s.data()); for (int i = 0; i < m.n; ++i) sol[i] = (double)s[i]; }
    void set_solution(const double* sol) override { auto s = cast(sol, m.n); m.set_solution(s.data()); }
    void gradients(int mode, double* dz, double* du) override {
        int nx = 12 * (int)m.bodies.size(), nu = m.nu();
        for (auto& c : m.contacts) if (c.kind == 1) {          // body-body contacts are forward only (no data Jacobians restated): NaN, not numbers
            for (size_t i = 0; i < (size_t)nx * nx; ++i) dz[i] = std::numeric_limits<double>::quiet_NaN();
            for (size_t i = 0; i < (size_t)nx * nu; ++i) du[i] = std::numeric_limits<double>::quiet_NaN();
            return;
        }
        std::vector<T> a((size_t)nx * nx), b((size_t)nx * std::max(nu, 1));
        std::vector<T> solmat = m.A;   // un-factored matrix of the last set_entries! (mehrotra.jl:69)
        if (mode == DOJO_GRAD_CONSISTENT && pre.size() == m.bodies.size()) {
            std::vector<State<T>> post; for (auto& B : m.bodies) post.push_back(B.st);
            for (size_t i = 0; i < pre.size(); ++i) m.bodies[i].st = pre[i];
            m.get_maximal_gradients(solmat, a.data(), b.data());
            for (size_t i = 0; i < post.size(); ++i) m.bodies[i].st = post[i];
        } else {
            m.get_maximal_gradients(solmat, a.data(), b.data());
        }
        for (size_t i = 0; i < (size_t)nx * nx; ++i) dz[i] = (double)a[i];
        for (size_t i = 0; i < (size_t)nx * nu; ++i) du[i] = (double)b[i];
    }
    void contact_gradients(int mode, double* dc) override {
        int nx = 12 * (int)m.bodies.size(), nc = 5 * (int)m.contacts.size();
        std::vector<T> a((size_t)nx * std::max(nc, 1));
        std::vector<T> solmat = m.A;
        if (mode == DOJO_GRAD_CONSISTENT && pre.size() == m.bodies.size()) {
            std::vector<State<T>> post; for (auto& B : m.bodies) post.push_back(B.st);
            for (size_t i = 0; i < pre.size(); ++i) m.bodies[i].st = pre[i];
            m.get_contact_gradients(solmat, a.data());
            for (size_t i = 0; i < post.size(); ++i) m.bodies[i].st = post[i];
        } else m.get_contact_gradients(solmat, a.data());
        for (size_t i = 0; i < (size_t)nx * nc; ++i) dc[i] = (double)a[i];
    }
    void get_data(double* d) override { int nd = m.data_dim(false); std::vector<T> v(nd); m.get_data(v.data()); for (int i = 0; i < nd; ++i) d[i] = (double)v[i]; }
    void set_data(const double* d) override { auto v = cast(d, m.data_dim(false)); m.set_data(v.data()); }
    void evaluate_residual(const double* data, const double* sol, double* out) override {
        auto d = cast(data, m.data_dim(false)); auto s = cast(sol, m.n); std::vector<T> o(m.n);
        m.evaluate_residual(d.data(), s.data(), o.data());
        for (int i = 0; i < m.n; ++i) out[i] = (double)o[i];
    }
    void full_matrix(double* out) override { for (size_t i = 0; i < (size_t)m.n * m.n; ++i) out[i] = (double)m.A[i]; }
    void data_matrix(double* out) override { std::vector<T> D; int nd; m.jacobian_data(D, nd); for (size_t i = 0; i < D.size(); ++i) out[i] = (double)D[i]; }
    void data_attjac(double* out) override { std::vector<T> G; int nr, nc; m.data_attitude_jacobian(G, nr, nc); for (size_t i = 0; i < G.size(); ++i) out[i] = (double)G[i]; }
    void set_state(const double* z) override {   // state without touching inputs (for simulate!)
        int Nb = (int)m.bodies.size();
        for (int i = 0; i < Nb; ++i) {
            State<T>& s = m.bodies[i].st; const double* p = z + 13 * i;
            for (int k = 0; k < 3; ++k) { s.x2[k] = T(p[k]); s.v15[k] = T(p[3 + k]); s.w15[k] = T(p[10 + k]); }
            s.q2 = Quat<T>(T(p[6]), T(p[7]), T(p[8]), T(p[9]));
        }
        m.initialize_simulation();
    }
    void get_state(double* z) override { int nz = 13 * (int)m.bodies.size(); std::vector<T> o(nz); m.get_maximal_state(o.data()); for (int i = 0; i < nz; ++i) z[i] = (double)o[i]; }
    void set_external_force(int body, const double* force, const double* torque, const double* vertex) override {
        // set_external_force!(body; force, torque, vertex)  bodies/set.jl:96-101
        State<T>& s = m.bodies[body].st;
        SM<T> f = SM<T>::vec({T(force[0]), T(force[1]), T(force[2])}), t = SM<T>::vec({T(torque[0]), T(torque[1]), T(torque[2])}),
              v = SM<T>::vec({T(vertex[0]), T(vertex[1]), T(vertex[2])});
        s.Fext = vector_rotate(f, s.q2);
        s.text = t + skew(v) * f;
    }
    int simulate_step(const double* u, int last) override {
        if (u) { auto uu = cast(u, m.nu()); return m.simulate_step(uu.data(), last != 0); }
        return m.simulate_step(nullptr, last != 0);
    }
    int simulate_step_record(const double* u, int last, double* row) override {
        std::vector<T> o(25 * m.bodies.size()), uu(m.nu(), T(0)); if (u) uu = cast(u, m.nu());
        int st = m.simulate_step(uu.data(), last != 0, o.data());
        for (size_t i = 0; i < o.size(); ++i) row[i] = (double)o[i];
        return st;
    }
    void body_velocity_solution(double* v) override {
        for (size_t i = 0; i < m.bodies.size(); ++i) for (int k = 0; k < 3; ++k) { v[6 * i + k] = (double)m.bodies[i].st.vsol[1][k]; v[6 * i + 3 + k] = (double)m.bodies[i].st.wsol[1][k]; }
    }
    void save_to_storage(double* out) override {
        std::vector<T> o(25 * m.bodies.size()); m.save_to_storage(o.data());
        for (size_t i = 0; i < o.size(); ++i) out[i] = (double)o[i];
    }
    void energy_of_storage_row(const double* row, double* ke_pe) override {
        std::vector<T> r(25 * m.bodies.size()); for (size_t i = 0; i < r.size(); ++i) r[i] = T(row[i]);
        T ke, pe; m.energy_of_storage_row(r.data(), ke, pe); ke_pe[0] = (double)ke; ke_pe[1] = (double)pe;
    }
    void debug_assemble(const double* z, const double* u, double* A, double* b) override {
        // state of mehrotra! right after its first set_entries! (mehrotra.jl:10-21)
        int nz = 13 * (int)m.bodies.size(), nu = m.nu();
        std::vector<T> zz = cast(z, nz), uu(nu, T(0)); if (u) uu = cast(u, nu);
        m.set_maximal_state(zz.data()); m.set_input_all(uu.data());
        for (auto& c : m.contacts) m.reset_contact(c);
        for (auto& J : m.joints) m.reset_joint(J);
        m.mu = 0;
        for (auto& c : m.contacts) m.initialize_contact(c);
        m.set_entries();
        for (size_t i = 0; i < (size_t)m.n * m.n; ++i) A[i] = (double)m.A[i];
        for (int i = 0; i < m.n; ++i) b[i] = (double)m.b[i];
    }
    void check_solution(const double* z, const double* u, const double* sol, double* viol) override {
        // residual_violation / bilinear_violation (src/solver/violations.jl) of a candidate solution of step!(z, u)
        int nz = 13 * (int)m.bodies.size(), nu = m.nu();
        std::vector<T> zz = cast(z, nz), uu(nu, T(0)); if (u) uu = cast(u, nu);
        m.set_maximal_state(zz.data()); m.set_input_all(uu.data());
        auto s = cast(sol, m.n); m.set_solution(s.data());
        m.mu = 0;
        viol[0] = (double)m.residual_violation(); viol[1] = (double)m.bilinear_violation();
    }
    IOracle* clone() override { return new OracleT<T>(*this); }
};

} // namespace

extern "C" {

// dtype 99: the counting scalar (oracle/counted.hpp): orc_op_count(reset) = floating-point operations of this THREAD's counted instances outside the
// linear solves since the last reset
long long orc_op_count(int reset) { const long long n = Counted::ops(); if (reset) Counted::ops() = 0; return n; }
void* orc_create(const DojoTopology* tp, int dtype) {
    if (dtype == 99) return new OracleT<Counted>(*tp);
    if (dtype == DOJO_DTYPE_F32) return new OracleT<float>(*tp);
    return new OracleT<double>(*tp);
}
void orc_destroy(void* h) { delete (IOracle*)h; }
void orc_set_options(void* h, const DojoSolverOptions* o) { ((IOracle*)h)->set_options(*o); }
void orc_dims(void* h, int* out) { ((IOracle*)h)->dims(out); }
int  orc_step(void* h, const double* z, const double* u, double* z_state, double* z_return, int* iters) { return ((IOracle*)h)->step(z, u, z_state, z_return, iters); }
void orc_get_solution(void* h, double* sol) { ((IOracle*)h)->get_solution(sol); }
void orc_set_solution(void* h, const double* sol) { ((IOracle*)h)->set_solution(sol); }
void orc_gradients(void* h, int mode, double* dz, double* du) { ((IOracle*)h)->gradients(mode, dz, du); }
void orc_contact_gradients(void* h, int mode, double* dc) { ((IOracle*)h)->contact_gradients(mode, dc); }
void orc_get_data(void* h, double* d) { ((IOracle*)h)->get_data(d); }
void orc_set_data(void* h, const double* d) { ((IOracle*)h)->set_data(d); }
void orc_evaluate_residual(void* h, const double* data, const double* sol, double* out) { ((IOracle*)h)->evaluate_residual(data, sol, out); }
void orc_full_matrix(void* h, double* out) { ((IOracle*)h)->full_matrix(out); }
void orc_data_matrix(void* h, double* out) { ((IOracle*)h)->data_matrix(out); }
void orc_data_attjac(void* h, double* out) { ((IOracle*)h)->data_attjac(out); }
void orc_set_state(void* h, const double* z) { ((IOracle*)h)->set_state(z); }
void orc_get_state(void* h, double* z) { ((IOracle*)h)->get_state(z); }
void orc_set_external_force(void* h, int body, const double* f, const double* t, const double* v) { ((IOracle*)h)->set_external_force(body, f, t, v); }
int  orc_simulate_step(void* h, const double* u, int last) { return ((IOracle*)h)->simulate_step(u, last); }
int  orc_simulate_step_record(void* h, const double* u, int last, double* row) { return ((IOracle*)h)->simulate_step_record(u, last, row); }
void orc_body_velocity_solution(void* h, double* v) { ((IOracle*)h)->body_velocity_solution(v); }
void orc_save_to_storage(void* h, double* out) { ((IOracle*)h)->save_to_storage(out); }
// kinetic_energy / potential_energy (src/mechanics/energy.jl) of one Storage row [Nb][25] -> out = {ke, pe}
void orc_energy_of_storage_row(void* h, const double* row, double* out) { ((IOracle*)h)->energy_of_storage_row(row, out); }

void orc_debug_assemble(void* h, const double* z, const double* u, double* A, double* b) { ((IOracle*)h)->debug_assemble(z, u, A, b); }

void orc_check_solution(void* h, const double* z, const double* u, const double* sol, double* viol) { ((IOracle*)h)->check_solution(z, u, sol, viol); }

// Batched step for the CPU baseline: one environment per thread-task over `nthreads`
// host threads.  z [B,13Nb], u [B,nu] or NULL, outputs may be NULL.
void orc_step_batch(void* h, int B, const double* z, const double* u, double* z_next, int* status, int* iters,
                    int with_grad, int grad_mode, double* dz, double* du, int nthreads) {
    IOracle* base = (IOracle*)h; int d[7]; base->dims(d);
    int nz = 13 * d[4], nu = d[1], nx = 12 * d[4];
    nthreads = std::max(1, std::min(nthreads, B));
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) {
        th.emplace_back([=]() {
            std::unique_ptr<IOracle> o(base->clone());
            std::vector<double> zs(nz);
            for (int e = t; e < B; e += nthreads) {
                int it = 0;
                int st = o->step(z + (size_t)e * nz, u ? u + (size_t)e * nu : nullptr, z_next ? z_next + (size_t)e * nz : zs.data(), nullptr, &it);
                if (status) status[e] = st;
                if (iters) iters[e] = it;
                if (with_grad) o->gradients(grad_mode, dz + (size_t)e * nx * nx, du + (size_t)e * nx * nu);
            }
        });
    }
    for (auto& x : th) x.join();
}

void orc_maximal_to_minimal(void* h, const double* z, double* x) { ((IOracle*)h)->maximal_to_minimal(z, x); }
void orc_minimal_to_maximal(void* h, const double* x, double* z) { ((IOracle*)h)->minimal_to_maximal(x, z); }
void orc_input_impulses(void* h, const double* z, const double* u, double* jf) { ((IOracle*)h)->input_impulses(z, u, jf); }
// timing variant of the linear solves: 1 = sparse no-pivot LU in the elimination order of the mechanism graph (SparseLU)
void orc_set_sparse_solver(void* h, int on) { ((IOracle*)h)->set_sparse_solver(on); }
long long orc_sparse_flops(void* h) { return ((IOracle*)h)->sparse_flops(); }
long long orc_sparse_solve_flops(void* h) { return ((IOracle*)h)->sparse_solve_flops(); }   // per right-hand side
// line-search statistics of this instance since creation: out[0] = line searches, out[1] = residual evaluations (trials)
void orc_ls_stats(void* h, long long* out) { ((IOracle*)h)->ls_stats(out); }
// rounds of iterative refinement of every linear solve (default 2: the checker; 0: a plain LU solve like the reference's)
void orc_set_refine_steps(void* h, int n) { ((IOracle*)h)->set_refine_steps(n); }

// bench.py's cpu_baseline leg: every thread owns a clone of the mechanism and walks its share of the B environments `rounds`
// times (results discarded); the clock starts once all threads stand at the barrier.  Returns the wall-clock seconds.
// physical cores this process may run on (one hardware thread per core): what a CPU baseline should be timed on
static std::vector<int> physical_cpus() {
    cpu_set_t set; CPU_ZERO(&set);
    std::vector<int> cpus;
    if (sched_getaffinity(0, sizeof(set), &set) != 0) return cpus;
    std::vector<int> seen_core;
    for (int c = 0; c < CPU_SETSIZE; ++c) {
        if (!CPU_ISSET(c, &set)) continue;
        char path[128]; std::snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
        int first = c;
        if (FILE* f = std::fopen(path, "r")) { if (std::fscanf(f, "%d", &first) != 1) first = c; std::fclose(f); }
        if (first == c || !CPU_ISSET(first, &set)) cpus.push_back(c);      // the first sibling of every core (or a core whose first sibling is not ours)
    }
    return cpus;
}
int orc_physical_cores(void) { return (int)physical_cpus().size(); }

// Timing runs only (orc_time_batch): blocks up to glibc's limit for this knob (HEAP_MAX_SIZE / 2 = 32 MiB on 64-bit; larger requests are
// rejected) stay in the per-thread malloc arenas, and freed memory is not trimmed back to the kernel between passes -- mmap / munmap per
// call serialize a multi-threaded batch in the kernel.  Applied when a timing run starts, not when the library is loaded: the setting is
// process-wide, and pytest / torch processes that only load the checker are left alone.
static void malloc_tuning_for_timing() {
    static bool done = false;
    if (done) return;
    done = true;
    if (!mallopt(M_MMAP_THRESHOLD, 32 << 20)) std::fprintf(stderr, "oracle: mallopt(M_MMAP_THRESHOLD) rejected\n");
    if (!mallopt(M_TRIM_THRESHOLD, 1 << 30)) std::fprintf(stderr, "oracle: mallopt(M_TRIM_THRESHOLD) rejected\n");
}

double orc_time_batch(void* h, int B, const double* z, const double* u, int with_grad, int grad_mode, int nthreads, int rounds) {
    malloc_tuning_for_timing();
    const std::vector<int> cpus = physical_cpus();
    IOracle* base = (IOracle*)h; int d[7]; base->dims(d);
    const int nz = 13 * d[4], nu = d[1], nx = 12 * d[4];
    nthreads = std::max(1, std::min(nthreads, B));
    std::atomic<int> ready{0}; std::atomic<bool> go{false};
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) {
        th.emplace_back([&, t]() {
            if (!cpus.empty()) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpus[t % cpus.size()], &one); pthread_setaffinity_np(pthread_self(), sizeof(one), &one); }   // one thread per physical core
            std::unique_ptr<IOracle> o(base->clone());
            { int it0 = 0; std::vector<double> z0(nz);           // untimed first step: the clone's workspaces and symbolic factorization
              o->step(z + (size_t)t * nz, u ? u + (size_t)t * nu : nullptr, z0.data(), nullptr, &it0);
              if (with_grad) { std::vector<double> g0((size_t)nx * nx), g1((size_t)nx * std::max(nu, 1)); o->gradients(grad_mode, g0.data(), g1.data()); } }
            std::vector<double> zs(nz), dz(with_grad ? (size_t)nx * nx : 0), du(with_grad ? (size_t)nx * std::max(nu, 1) : 0);
            ready.fetch_add(1);
            while (!go.load()) std::this_thread::yield();
            for (int r = 0; r < rounds; ++r)
                for (int e = t; e < B; e += nthreads) {
                    int it = 0;
                    o->step(z + (size_t)e * nz, u ? u + (size_t)e * nu : nullptr, zs.data(), nullptr, &it);
                    if (with_grad) o->gradients(grad_mode, dz.data(), du.data());
                }
        });
    }
    while (ready.load() < nthreads) std::this_thread::yield();
    auto t0 = std::chrono::steady_clock::now();
    go.store(true);
    for (auto& x : th) x.join();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Unit functions of the restatement, for the reference's own unit tests restated in tests/test_oracle_units.py
// (test/integrator.jl, test/mrp.jl): `what` selects the function, `in` / `out` are flat fp64 arrays (matrices row-major).
//   0 next_orientation(q[4], w[3], dt) -> q'[4]                       src/integrators/integrator.jl:22-28
//   1 rotational_integrator_jacobian_velocity(q, w, dt) -> 4x3        integrator.jl:50-52
//   2 rotational_integrator_jacobian_orientation(q, w, dt; attjac=false) -> 4x4   integrator.jl:54-62
//   3 rotational_integrator_jacobian_orientation(...; attjac=true) -> 4x3
//   4 mrp(q[4]) -> 3     5 dmrpdq(q) -> 3x4     6 axis(q) -> 3     7 daxisdq(q) -> 3x4      src/orientation/mrp.jl
//   8 rotation_vector(q) -> 3     9 drotation_vectordq(q) -> 3x4                              mrp.jl:61-78
int orc_joint_unit(void* h, int joint, int half, int what, const double* in, double* out) { return ((IOracle*)h)->joint_unit(joint, half, what, in, out); }
int orc_contact_unit(void* h, int contact, int what, const double* in, double* out) { return ((IOracle*)h)->contact_unit(contact, what, in, out); }
int orc_unit(int what, const double* in, double* out) {
    using M = orc::SM<double>; using Q = orc::Quat<double>;
    auto put = [&](const M& m) { for (int i = 0; i < m.r * m.c; ++i) out[i] = m.a[i]; };
    const Q q(in[0], in[1], in[2], in[3]);
    const M qv = M::vec({in[0], in[1], in[2], in[3]});
    const M w = M::vec({in[4], in[5], in[6]}); const double dt = in[7];
    switch (what) {
        case 0: put(orc::vector(orc::next_orientation(q, w, dt))); return 4;
        case 1: put(orc::rotational_integrator_jacobian_velocity(q, w, dt)); return 12;
        case 2: put(orc::rotational_integrator_jacobian_orientation(q, w, dt, false)); return 16;
        case 3: put(orc::rotational_integrator_jacobian_orientation(q, w, dt, true)); return 12;
        case 4: put(orc::mrp(qv)); return 3;
        case 5: put(orc::dmrpdq(qv)); return 12;
        case 6: put(orc::axis_of(qv)); return 3;
        case 7: put(orc::daxisdq(qv)); return 12;
        case 8: put(orc::rotation_vector(q)); return 3;
        case 9: put(orc::drotation_vectordq(q)); return 12;
    }
    return -1;
}

} // extern "C"
