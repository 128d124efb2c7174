// dojo_hip.hip -- libdojo_hip.so: host side of the C ABI of include/dojo_hip.h (kernels: dojo_kernels.hip).
//
// One wavefront = 64/S environments x S supernodes (dojo_device.hpp).  One workgroup = one
// wavefront (64 threads), so a batch of B environments launches ceil(B*S/64) workgroups --
// 1024 for Ant at B = 4096, i.e. one wave per SIMD on the 256 CUs (>> 256 workgroups, and the
// kernel needs no LDS, so placement across the 8 XCDs is irrelevant: nothing is shared).
// There is deliberately no CPU fallback in this library: without a gfx950 device dojo_create
// fails with DOJO_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include "dojo_host.hpp"
#include "dojo_coords.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <map>
#include <vector>
#include <algorithm>
#include <mutex>
#include <dlfcn.h>

struct DojoSim;
namespace {

// Error text: one process-wide string behind a mutex (dojo_last_error: what the last failing call on ANY thread said -- a
// Julia task may migrate between the failing @ccall and the query, so it must not be thread-local) and a copy on the handle
// the failing entry point was called with (dojo_handle_error: per handle).
void handle_set_error(::DojoSim* s, const std::string& m);
thread_local ::DojoSim* t_handle = nullptr;            // handle of the entry point running on this thread
struct ErrSink {
    std::mutex m; std::string last; std::string ret;
    ErrSink& operator=(const std::string& v) { { std::lock_guard<std::mutex> g(m); last = v; } if (t_handle) handle_set_error(t_handle, v); return *this; }
    ErrSink& operator=(const char* v) { return *this = std::string(v); }
    const char* c_str() { std::lock_guard<std::mutex> g(m); ret = last; return ret.c_str(); }
} g_err;
struct Enter { ::DojoSim* prev; explicit Enter(::DojoSim* s) : prev(t_handle) { t_handle = s; } ~Enter() { t_handle = prev; } };

// kernel launchers, one per object file of dojo_kernels.hip: dojo_launch_<abi type>_<max contacts per body>_<quad>
extern "C" {
#define DJ_DECL(n) int n(const void*, int, void*, int, void*);
DJ_DECL(dojo_launch_float_1_1) DJ_DECL(dojo_launch_float_4_1) DJ_DECL(dojo_launch_float_8_1)
DJ_DECL(dojo_launch_double_1_1) DJ_DECL(dojo_launch_double_4_1) DJ_DECL(dojo_launch_double_8_1)
DJ_DECL(dojo_launch_float_4_0) DJ_DECL(dojo_launch_float_8_0) DJ_DECL(dojo_launch_double_4_0) DJ_DECL(dojo_launch_double_8_0)
DJ_DECL(dojo_launch_float_4_2) DJ_DECL(dojo_launch_double_4_2) DJ_DECL(dojo_launch_float_1_2) DJ_DECL(dojo_launch_double_1_2)
// the builds with translational springs / dampers (-DDJ_TSD=1): single-wavefront quad mapping, <= 4 contacts per body
DJ_DECL(dojo_launch_tsd_float_1_1) DJ_DECL(dojo_launch_tsd_float_4_1) DJ_DECL(dojo_launch_tsd_double_1_1) DJ_DECL(dojo_launch_tsd_double_4_1)
DJ_DECL(dojo_launch_tsd_float_4_0) DJ_DECL(dojo_launch_tsd_float_8_0) DJ_DECL(dojo_launch_tsd_double_4_0) DJ_DECL(dojo_launch_tsd_double_8_0)
DJ_DECL(dojo_launch_gen_float_4_0) DJ_DECL(dojo_launch_gen_float_8_0) DJ_DECL(dojo_launch_gen_double_4_0) DJ_DECL(dojo_launch_gen_double_8_0)
DJ_DECL(dojo_launch_lin_float_1_1) DJ_DECL(dojo_launch_lin_float_4_1) DJ_DECL(dojo_launch_lin_double_1_1) DJ_DECL(dojo_launch_lin_double_4_1)     // LinearContact builds
DJ_DECL(dojo_launch_lin_float_4_0) DJ_DECL(dojo_launch_lin_float_8_0) DJ_DECL(dojo_launch_lin_double_4_0) DJ_DECL(dojo_launch_lin_double_8_0)
DJ_DECL(dojo_launch_ss_float_1_1) DJ_DECL(dojo_launch_ss_double_1_1)     // body-body contacts (-DDJ_SS=1): single-wavefront quad mapping, <= 1 contact per body, forward only
#define DJ_CDECL(n) int n(const void*, int, void*);
DJ_CDECL(dojo_launch_cgrad_float_1_1) DJ_CDECL(dojo_launch_cgrad_float_4_1) DJ_CDECL(dojo_launch_cgrad_float_8_1)
DJ_CDECL(dojo_launch_cgrad_double_1_1) DJ_CDECL(dojo_launch_cgrad_double_4_1) DJ_CDECL(dojo_launch_cgrad_double_8_1)
DJ_CDECL(dojo_launch_cgrad_float_4_2) DJ_CDECL(dojo_launch_cgrad_double_4_2) DJ_CDECL(dojo_launch_cgrad_float_1_2) DJ_CDECL(dojo_launch_cgrad_double_1_2)
DJ_CDECL(dojo_launch_cgrad_tsd_float_1_1) DJ_CDECL(dojo_launch_cgrad_tsd_float_4_1) DJ_CDECL(dojo_launch_cgrad_tsd_double_1_1) DJ_CDECL(dojo_launch_cgrad_tsd_double_4_1)
#undef DJ_DECL
}

// temporary device buffer of a host-pointer entry point: freed on every return path
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    DevBuf() = default; DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
};
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); return DOJO_ERR_DEVICE; } } while (0)

} // namespace

struct DojoSim {
    dj::HostModel M;
    DojoSolverOptions opts;
    int B = 0, dtype = 0, device = 0, grad_mode = DOJO_GRAD_REFERENCE;
    size_t w = 8;                       // bytes per scalar
    void* d_tsd = nullptr;       // translational springs / dampers per supernode (mechanisms that have them)
    void* d_mlim = nullptr;      // joint limits on several coordinates per supernode (mechanisms that have them)
    void* d_cuts = nullptr;      // loop-closing joints (mechanisms with kinematic loops)
    void* d_cutws = nullptr;     // [B][dj::CUTWS] doubles: per-environment workspace of the cut elements (KernelArgs::cutws)
    void* d_nodes = nullptr; void* d_contacts = nullptr; int* d_order = nullptr;   // tables; bodies in root -> leaves order
    void *d_x = nullptr, *d_xn = nullptr;   // minimal-coordinate buffers of the host-pointer entry points
    void *d_cz = nullptr;                   // maximal-state scratch of dojo_minimal_to_maximal / dojo_maximal_to_minimal (d_z stays the state of the last step)
    void *d_jf = nullptr;                   // dojo_step_impulses: body impulses folded into the external-force slot
    double *d_mu = nullptr;                 // [B] mechanism.μ at the end of the last step (dojo_get_mu)
    double *d_diag = nullptr;               // [B][2] diagnostics of the last step's final linearization (dojo_get_diagnostics)
    void *d_jm = nullptr, *d_jt = nullptr, *d_jb = nullptr;   // get_minimal_gradients!: min->max Jacobian, dz * that, max->min blocks (fp64)
    // internal device buffers used by the host-pointer entry points
    const void* fext = nullptr;                     // device [B,6Nb] external forces applied by every step, or null (dojo_set_external_force)
    void *d_fext = nullptr, *d_res = nullptr, *d_z = nullptr, *d_u = nullptr, *d_zn = nullptr, *d_vel = nullptr, *d_jimp = nullptr, *d_csg = nullptr, *d_dz = nullptr, *d_du = nullptr;
    std::vector<hipStream_t> gstreams; std::vector<hipEvent_t> gevents; hipEvent_t fork_event = nullptr;   // environment groups (rollouts, dojo_step_dev)
    int groups = -1;                    // environment groups of dojo_step_dev: -1 = chosen from the batch size, 1 = one launch on the caller's stream
    int async = 0; bool pending = false;   // dojo_set_async: 1 = dojo_step_dev returns without joining the groups into the caller's stream; 2 = ... and the IFT of a
                                        // group's step runs on a second stream of the group, next to its NEXT step kernel (two hand-off records in turn)
    std::vector<hipStream_t> gstreams2; std::vector<hipEvent_t> gevents2, grad_done[2];   // pipelined groups: the IFT streams, their join events, "IFT that read record p is through"
    void* d_sol2 = nullptr; int sol_cur = 0; bool plain_phases = false;                   // the second hand-off record; the one in use; phased launches without the iteration cap's lists
    size_t last_NG = 0, last_per = 0;      // environment-group partition of the last grouped dojo_step_dev (per-group chaining assumes it repeats)
    void* d_sol = nullptr;              // step kernel -> IFT kernel hand-off (converged solution, fp64)
    void* d_fac = nullptr;              // ... and the final supernode factors (quad mapping; explicit-inverse consumers only)
    void* d_lu = nullptr;               // the IFT kernel's LU-form factors between its phases (quad mapping)
    void* d_blk = nullptr;              // un-factored supernode rows of the environments whose solves are refined (quad mapping, DJ_REFINE)
    void* d_msg = nullptr;              // quad mapping: messages of the IFT up-sweep to the roots (KernelArgs::msg)
    void* d_ypark = nullptr;            // fp32 ABI, quad mapping: the IFT's forward-substituted right-hand sides between its two sweeps, in fp64
    int* d_flag = nullptr;              // [B] environments the plain step kernel deferred to the refining kernels
    double refine_w = -1.0;             // refine once max γ/s of an environment exceeds this (dojo_set_refinement); < 0: chosen from the tolerances
    int *d_status = nullptr, *d_iters = nullptr;
    // iteration cap + continuation kernel (dojo_set_iteration_cap; dojo_device.hpp Globals::iter_cap)
    int iter_cap = -1;                  // < 0: automatic (DOJO_DEFAULT_ITERATION_CAP where the continuation kernel exists), 0: off, > 0: as given
    void* d_resume = nullptr;           // [B][CARRY_PER_ENV] loop scalars of the solves the step kernel left unfinished
    int *d_cont_list = nullptr, *d_cont_count = nullptr;   // [workgroups of the batch] the continuation list (batch-level workgroup indices), [1] its length
    int *d_cstat = nullptr;             // [B] status buffer of launches whose caller passed none (the continuation kernel reads it)
    hipStream_t cstream = nullptr; hipEvent_t cont_event = nullptr, allmain_event = nullptr; std::vector<hipEvent_t> main_events;   // the continuation's stream; "step kernel done" per group
    // dispatch order of the step kernel (dojo_set_dispatch_order): 0 off, 1 where a launch has more workgroups than the GPU holds at once and the step is
    // joined into the caller's stream (the step then waits for its last wavefront), 2 always
    int dispatch_mode = 1, sm_count = 0; size_t step_groups = 1; bool chained_now = false;      // step_groups: groups of the dojo_step_dev in progress;      // chained_now: launches of rollout_core (one group's steps back to back)
    int *d_dispatch = nullptr, *d_liters = nullptr;        // [workgroups of the batch] the permutations, one segment per launch; [B] iteration counts when the caller passes no buffer
    std::map<size_t, int> dispatch_have;                   // first workgroup of a segment -> environments of the launch whose permutation it holds
    int phase_slot = -1;                // timing slot of the phased launch in progress (PH_MAIN -> PH_GRAD of the same group)
    std::vector<int> group_slot;        // ... per environment group
    bool have_grad = false, have_solution = false, have_u = false;
    std::string err; std::mutex err_m;  // text of the last failure of a call on this handle (dojo_handle_error)
    void* comm = nullptr; int comm_rank = 0, comm_world = 1;   // RCCL communicator of this handle's process group (dojo_comm_init)
    // kernel timing: a ring of event triples (launch begin / between the step and the IFT kernel / end), so that
    // timed launches never make the host wait; totals are accumulated when a slot is reused or queried
    struct Ev3 { hipEvent_t a = nullptr, m = nullptr, b = nullptr; bool has_mid = false, used = false; int n = 1; };
    std::vector<Ev3> ring; size_t ring_next = 0; int last_slot = -1;
    double acc_step_ms = 0, acc_ift_ms = 0; long long acc_n = 0;
};

namespace {
void handle_set_error(::DojoSim* s, const std::string& m) { std::lock_guard<std::mutex> g(s->err_m); s->err = m; }

// ---------------------------------------------------------------------------------------------------------------
// Minimal <-> maximal coordinates: HBM-bound helper kernels around the step; every DojoEnvironments
// step! goes through them (src/simulation/step.jl:42-60).  The per-joint maps live in dojo_coords.hpp, written for any
// scalar: double gives the values, forward-mode dual numbers give the chain-rule Jacobians of get_minimal_gradients!
// (src/gradients/state.jl:9-56, 136-217).  x per joint (mechanism.joints order): [dx(nu_t); dtheta(nu_r); dv(nu_t); domega(nu_r)]
// ---------------------------------------------------------------------------------------------------------------
namespace ckern {
using namespace dj;
using namespace dj::coords;
template <class S, class TIO> __device__ __forceinline__ PoseVel<S> load_body(const TIO* z, int b) {
    PoseVel<S> p;
    for (int i = 0; i < 3; ++i) { p.x[i] = S((double)z[13 * b + i]); p.v[i] = S((double)z[13 * b + 3 + i]); p.w[i] = S((double)z[13 * b + 10 + i]); }
    double q_[4];
    for (int i = 0; i < 4; ++i) q_[i] = (double)z[13 * b + 6 + i];
    if (sizeof(TIO) < sizeof(double)) {    // a narrower ABI type cannot hold a unit quaternion: the state it stands for is (x, v, q/|q|, ω), as in the step / IFT kernels (DJ_LANE_SETUP)
        const double iq_ = 1.0 / sqrt(q_[0] * q_[0] + q_[1] * q_[1] + q_[2] * q_[2] + q_[3] * q_[3]);
        for (int i = 0; i < 4; ++i) q_[i] *= iq_;
    }
    for (int i = 0; i < 4; ++i) p.q[i] = S(q_[i]);
    return p;
}
template <class S> __device__ __forceinline__ PoseVel<S> origin_body() {
    PoseVel<S> p; for (int i = 0; i < 3; ++i) { p.x[i] = S(0.0); p.v[i] = S(0.0); p.w[i] = S(0.0); } p.q[0] = S(1.0); p.q[1] = S(0.0); p.q[2] = S(0.0); p.q[3] = S(0.0); return p;
}
// get_next_state (src/mechanism/get.jl:126-134) of a body whose velocities are its solution: x + dt v, q (x) xi(w)
template <class S> __device__ __forceinline__ void advance_body(PoseVel<S>& p, double dt) {
    for (int i = 0; i < 3; ++i) p.x[i] = p.x[i] + p.v[i] * dt;
    S qn[4]; next_qS(qn, p.q, p.w, dt); for (int i = 0; i < 4; ++i) p.q[i] = qn[i];
}

// one thread per environment: bodies in root -> leaves order (the parent's maximal state must exist first)
template <class TIO>
__global__ void min2max_kernel(const NodeP<double>* nodes, const int* order, int Nb, int nu, double dt, int B, const TIO* x, TIO* z) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= B) return;
    const TIO* xe = x + (size_t)env * 2 * nu; TIO* ze = z + (size_t)env * 13 * Nb;
    for (int oi = 0; oi < Nb; ++oi) {
        const int k = order[oi];
        const NodeP<double>& P = nodes[k];
        const int nt = P.nu_t, nr = P.nu_r, n = nt + nr;
        const TIO* xm = xe + 2 * P.u_off;
        double dx[3] = {0, 0, 0}, dth[3] = {0, 0, 0}, dv[3] = {0, 0, 0}, dw[3] = {0, 0, 0};
        for (int i = 0; i < 3; ++i) { if (i < nt) { dx[i] = (double)xm[i]; dv[i] = (double)xm[n + i]; } if (i < nr) { dth[i] = (double)xm[nt + i]; dw[i] = (double)xm[n + nt + i]; } }
        const PoseVel<double> a = P.parent >= 0 ? load_body<double>(ze, P.parent) : origin_body<double>();
        PoseVel<double> b;
        joint_min2max(b, P, dt, a, dx, dth, dv, dw);
        for (int i = 0; i < 3; ++i) { ze[13 * k + i] = (TIO)b.x[i]; ze[13 * k + 3 + i] = (TIO)b.v[i]; ze[13 * k + 10 + i] = (TIO)b.w[i]; }
        for (int i = 0; i < 4; ++i) ze[13 * k + 6 + i] = (TIO)b.q[i];
    }
}
// one thread per (environment, joint): the joints are independent
template <class TIO>
__global__ void max2min_kernel(const NodeP<double>* nodes, int Nb, int nu, double dt, int B, const TIO* z, TIO* x, int ldx) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int env = tid / Nb, k = tid % Nb;
    if (env >= B) return;
    const NodeP<double>& P = nodes[k];
    const TIO* ze = z + (size_t)env * 13 * Nb; TIO* xm = x + (size_t)env * ldx + 2 * P.u_off;
    const int nt = P.nu_t, nr = P.nu_r, n = nt + nr;
    const PoseVel<double> b = load_body<double>(ze, k), a = P.parent >= 0 ? load_body<double>(ze, P.parent) : origin_body<double>();
    double ct[3], cr[3], vt[3], vr[3];
    joint_max2min(ct, cr, vt, vr, P, dt, a, b);
    for (int i = 0; i < 3; ++i) { if (i < nt) { xm[i] = (TIO)ct[i]; xm[n + i] = (TIO)vt[i]; } if (i < nr) { xm[nt + i] = (TIO)cr[i]; xm[n + nt + i] = (TIO)vr[i]; } }
}

// the contact part of get_state(::AntARS) (DojoEnvironments/src/environments/ant_ars.jl:72-80): the normal impulse of
// every contact, clamped to [-1, 1], behind the minimal state.  csg = [s(4); gamma(4)] per contact of the last step.
template <class TIO>
__global__ void contact_obs_kernel(int Nc, int B, const TIO* csg, TIO* obs, int ldx, int off) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int env = tid / Nc, c = tid % Nc;
    if (env >= B) return;
    const double g = (double)csg[(size_t)env * 8 * Nc + 8 * c + 4];
    obs[(size_t)env * ldx + off + c] = (TIO)(g < -1.0 ? -1.0 : g > 1.0 ? 1.0 : g);
}

// save_to_storage! (src/simulation/storage.jl:50-67): one thread per (environment, body); inputs are the state the step
// was solved at and what the step kernel left behind (solution velocities, cone variables, body residual rows)
template <class TIO>
__global__ void storage_kernel(const NodeP<double>* nodes, const ContactP<double>* contacts, int Nb, int Nc, int model, int cper, double dt, int env0, int nenv,
                               const TIO* z, const TIO* vel, const TIO* csg, const TIO* res, const TIO* fext, TIO* storage) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = tid / Nb, k = tid % Nb;
    if (e >= nenv) return;
    const size_t env = (size_t)env0 + e;
    double zb[13], v[3], w[3], rb[6], row[25], fe[6];
    if (fext) for (int i = 0; i < 6; ++i) fe[i] = (double)fext[env * 6 * Nb + 6 * k + i];
    for (int i = 0; i < 13; ++i) zb[i] = (double)z[env * 13 * Nb + 13 * k + i];
    for (int i = 0; i < 3; ++i) { v[i] = (double)vel[env * 6 * Nb + 6 * k + i]; w[i] = (double)vel[env * 6 * Nb + 6 * k + 3 + i]; }
    for (int i = 0; i < 6; ++i) rb[i] = (double)res[env * 6 * Nb + 6 * k + i];
    auto other = [=](int b, double* zo, double* vo, double* wo) {            // (body-body contacts: the contact partner's state and solution)
        for (int i = 0; i < 13; ++i) zo[i] = (double)z[env * 13 * Nb + 13 * b + i];
        for (int i = 0; i < 3; ++i) { vo[i] = (double)vel[env * 6 * Nb + 6 * b + i]; wo[i] = (double)vel[env * 6 * Nb + 6 * b + 3 + i]; }
    };
    dj::storage_row(row, nodes[k], contacts, dt, zb, v, w, csg + env * (size_t)cper * Nc, rb, fext ? fe : (const double*)nullptr, nodes, other, model, cper, k, Nc);
    TIO* o = storage + (env * Nb + k) * 25;
    for (int i = 0; i < 25; ++i) o[i] = (TIO)row[i];
}

// ---- Jacobians by forward-mode differentiation of the same maps ----
typedef Dual<24> D24;
// a body state seeded with its 12 attitude-reduced directions [x, v, phi, omega] starting at direction d0
template <class TIO> __device__ __forceinline__ PoseVel<D24> seed_body(const PoseVel<double>& p, int d0) {
    PoseVel<D24> s;
    for (int i = 0; i < 3; ++i) { s.x[i] = D24::seed(p.x[i], d0 + i); s.v[i] = D24::seed(p.v[i], d0 + 3 + i); s.w[i] = D24::seed(p.w[i], d0 + 9 + i); }
    D24 e[4] = {D24(1.0), D24::seed(0.0, d0 + 6), D24::seed(0.0, d0 + 7), D24::seed(0.0, d0 + 8)}, q0[4] = {D24(p.q[0]), D24(p.q[1]), D24(p.q[2]), D24(p.q[3])};
    qmulS(s.q, q0, e);                                                        // q (x) (1, phi)
    return s;
}
// minimal_to_maximal_jacobian (src/gradients/state.jl:136-181): one thread per environment, root -> leaves;
// Jm[env][12 Nb rows][2 nu columns] row-major.  z must already hold minimal_to_maximal(x).
template <class TIO>
__global__ void min2max_jac_kernel(const NodeP<double>* nodes, const int* order, int Nb, int nu, double dt, int B, const TIO* x, const TIO* z, double* Jm) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= B) return;
    const int nm = 2 * nu;
    const TIO* xe = x + (size_t)env * nm; const TIO* ze = z + (size_t)env * 13 * Nb;
    double* J = Jm + (size_t)env * 12 * Nb * nm;
    for (int oi = 0; oi < Nb; ++oi) {
        const int k = order[oi];
        const NodeP<double>& P = nodes[k];
        const int nt = P.nu_t, nr = P.nu_r, n = nt + nr;
        const TIO* xm = xe + 2 * P.u_off;
        // directions 0..11: the parent's reduced state; 12..12+2n-1: the joint's own minimal coordinates in layout order
        D24 dx[3], dth[3], dv[3], dw[3];
        for (int i = 0; i < 3; ++i) {
            dx[i] = i < nt ? D24::seed((double)xm[i], 12 + i) : D24(0.0);             dv[i] = i < nt ? D24::seed((double)xm[n + i], 12 + n + i) : D24(0.0);
            dth[i] = i < nr ? D24::seed((double)xm[nt + i], 12 + nt + i) : D24(0.0);    dw[i] = i < nr ? D24::seed((double)xm[n + nt + i], 12 + n + nt + i) : D24(0.0);
        }
        const PoseVel<double> a0 = P.parent >= 0 ? load_body<double>(ze, P.parent) : origin_body<double>();
        PoseVel<D24> a = seed_body<TIO>(a0, 0), b;
        joint_min2max(b, P, dt, a, dx, dth, dv, dw);
        // rows of the child: x, v, phi = V(q_b^-1 (x) dq_b), omega
        double Pm[12][24];
        for (int d = 0; d < 24; ++d) {
            for (int i = 0; i < 3; ++i) { Pm[i][d] = b.x[i].d[d]; Pm[3 + i][d] = b.v[i].d[d]; Pm[9 + i][d] = b.w[i].d[d]; }
            const double q0 = b.q[0].v, q1 = b.q[1].v, q2 = b.q[2].v, q3 = b.q[3].v, e0 = b.q[0].d[d], e1 = b.q[1].d[d], e2 = b.q[2].d[d], e3 = b.q[3].d[d];
            Pm[6][d] = q0 * e1 - q1 * e0 - q2 * e3 + q3 * e2;                 // vector part of conj(q) (x) dq
            Pm[7][d] = q0 * e2 + q1 * e3 - q2 * e0 - q3 * e1;
            Pm[8][d] = q0 * e3 - q1 * e2 + q2 * e1 - q3 * e0;
        }
        for (int r = 0; r < 12; ++r) {
            double* Jr = J + (size_t)(12 * k + r) * nm;
            for (int c = 0; c < nm; ++c) {
                double acc = 0.0;
                if (P.parent >= 0) { const double* Jp = J + (size_t)(12 * P.parent) * nm + c; for (int m = 0; m < 12; ++m) acc += Pm[r][m] * Jp[(size_t)m * nm]; }
                const int lc = c - 2 * P.u_off;
                if (lc >= 0 && lc < 2 * n) acc += Pm[r][12 + lc];
                Jr[c] = acc;
            }
        }
    }
}
// maximal_to_minimal_jacobian (src/gradients/state.jl:9-56) at z (advanced by one integrator step when `advance`):
// one thread per (environment, joint); Jb[env][joint k][2n rows in layout order][24]: d/d(parent reduced state), d/d(child)
template <class TIO>
__global__ void max2min_jac_kernel(const NodeP<double>* nodes, int Nb, double dt, int B, const TIO* z, int advance, double* Jb) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int env = tid / Nb, k = tid % Nb;
    if (env >= B) return;
    const NodeP<double>& P = nodes[k];
    const TIO* ze = z + (size_t)env * 13 * Nb;
    const int nt = P.nu_t, nr = P.nu_r, n = nt + nr;
    PoseVel<double> b0 = load_body<double>(ze, k), a0 = P.parent >= 0 ? load_body<double>(ze, P.parent) : origin_body<double>();
    if (advance) { advance_body(b0, dt); if (P.parent >= 0) advance_body(a0, dt); }
    PoseVel<D24> a = seed_body<TIO>(a0, 0), b = seed_body<TIO>(b0, 12);
    D24 ct[3], cr[3], vt[3], vr[3];
    joint_max2min(ct, cr, vt, vr, P, dt, a, b);
    double* o = Jb + ((size_t)env * Nb + k) * 12 * 24;
    for (int i = 0; i < 3; ++i) for (int d = 0; d < 24; ++d) {
        if (i < nt) { o[(size_t)i * 24 + d] = ct[i].d[d]; o[(size_t)(n + i) * 24 + d] = vt[i].d[d]; }
        if (i < nr) { o[(size_t)(nt + i) * 24 + d] = cr[i].d[d]; o[(size_t)(n + nt + i) * 24 + d] = vr[i].d[d]; }
    }
}
// T = dz * Jm  (12Nb x 2nu per environment); dz is column-major per environment: dz(r, c) = dz[c * nx + r]
template <class TIO>
__global__ void chain_mid_kernel(int nx, int nm, int B, const TIO* dz, const double* Jm, double* T) {
    const int env = blockIdx.x;
    const TIO* D = dz + (size_t)env * nx * nx; const double* J = Jm + (size_t)env * nx * nm; double* Te = T + (size_t)env * nx * nm;
    for (int e = threadIdx.x; e < nx * nm; e += blockDim.x) {
        const int r = e % nx, j = e / nx;                                       // consecutive threads -> consecutive rows (coalesced dz reads)
        double acc = 0.0;
        for (int c = 0; c < nx; ++c) acc += (double)D[(size_t)c * nx + r] * J[(size_t)c * nm + j];
        Te[(size_t)r * nm + j] = acc;
    }
}
// jx = M2m * T, ju = M2m * du with the block-sparse M2m (each joint's rows touch its parent's and its child's 12 columns);
// outputs row-major per environment: jx[env][2nu][2nu], ju[env][2nu][nu]; du(r, c) = du[c * nx + r]
template <class TIO>
__global__ void chain_out_kernel(const NodeP<double>* nodes, int Nb, int nu, int B, const double* Jb, const double* T, const TIO* du, TIO* jx, TIO* ju) {
    const int env = blockIdx.x, nx = 12 * Nb, nm = 2 * nu;
    const double* Te = T + (size_t)env * nx * nm; const TIO* Du = du ? du + (size_t)env * nx * nu : nullptr;
    for (int e = threadIdx.x; e < nm * (nm + nu); e += blockDim.x) {
        const int i = e / (nm + nu), j = e % (nm + nu);
        // joint owning minimal row i
        int k = 0, lr = 0;
        for (int b = 0; b < Nb; ++b) { const int o = 2 * nodes[b].u_off, nn = 2 * (nodes[b].nu_t + nodes[b].nu_r); if (i >= o && i < o + nn) { k = b; lr = i - o; } }
        const double* Jr = Jb + (((size_t)env * Nb + k) * 12 + lr) * 24;
        const int par = nodes[k].parent;
        double acc = 0.0;
        for (int m = 0; m < 12; ++m) {
            if (j < nm) { if (par >= 0) acc += Jr[m] * Te[(size_t)(12 * par + m) * nm + j]; acc += Jr[12 + m] * Te[(size_t)(12 * k + m) * nm + j]; }
            else if (Du) { const int c = j - nm; if (par >= 0) acc += Jr[m] * (double)Du[(size_t)c * nx + 12 * par + m]; acc += Jr[12 + m] * (double)Du[(size_t)c * nx + 12 * k + m]; }
        }
        if (j < nm) jx[((size_t)env * nm + i) * nm + j] = (TIO)acc; else if (ju) ju[((size_t)env * nm + i) * nu + (j - nm)] = (TIO)acc;
    }
}
// get_next_state(mechanism) (src/mechanism/get.jl:126-134) of bodies whose stored velocities are their solution: one thread per (env, body)
template <class TIO>
__global__ void next_state_kernel(int Nb, double dt, int B, const TIO* z, TIO* zo) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= B * Nb) return;
    PoseVel<double> p = load_body<double>(z + (size_t)(tid / Nb) * 13 * Nb, tid % Nb);
    advance_body(p, dt);
    TIO* o = zo + (size_t)tid * 13;
    for (int i = 0; i < 3; ++i) { o[i] = (TIO)p.x[i]; o[3 + i] = (TIO)p.v[i]; o[10 + i] = (TIO)p.w[i]; }
    for (int i = 0; i < 4; ++i) o[6 + i] = (TIO)p.q[i];
}
// dojo_step_impulses: external force + body impulses / dt
template <class TIO> __global__ void fold_impulses_kernel(long long n, const TIO* fext, const TIO* jf, double inv_dt, TIO* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (TIO)((fext ? (double)fext[i] : 0.0) + (double)jf[i] * inv_dt);
}
// dojo_set_dispatch_order: the permutation the NEXT step kernel of these workgroups hands its hardware workgroups out by -- a counting sort of
// the launch's workgroups by the Newton iterations their environments took in this step, longest first (a solve that was long stays long for
// a while: the same contacts are closing).  One workgroup; the order inside a bucket is whatever the atomics give (results do not depend on it:
// every environment is computed by the same program wherever it runs).  E = environments per workgroup.
__global__ void __launch_bounds__(1024) dispatch_order_kernel(const int* iters, int nenv, int E, int nwg, int* order) {
    __shared__ int cnt[64];
    const int t = (int)threadIdx.x;
    if (t < 64) cnt[t] = 0;
    __syncthreads();
    auto bucket = [&](int wg) { int k = 0; for (int e = 0; e < E; ++e) { const int env = wg * E + e; if (env < nenv) k = max(k, iters[env]); } return 63 - min(max(k, 0), 63); };
    for (int wg = t; wg < nwg; wg += 1024) atomicAdd(&cnt[bucket(wg)], 1);
    __syncthreads();
    if (t == 0) { int a = 0; for (int b = 0; b < 64; ++b) { const int c = cnt[b]; cnt[b] = a; a += c; } }
    __syncthreads();
    for (int wg = t; wg < nwg; wg += 1024) order[atomicAdd(&cnt[bucket(wg)], 1)] = wg;
}
} // namespace ckern

template <class T>
int upload_tables(DojoSim* s) {   // tables are stored in the state precision (fp64)
    std::vector<dj::NodeP<T>> nodes; for (auto& n : s->M.nodes) nodes.push_back(dj::cast_node<T>(n));
    // one extra entry for the idle supernode slots of a workgroup: node 0 without contacts (a slot that kept node 0's
    // contacts would write the same contact-pool rows as the real node 0 in the two-wavefront mapping)
    { dj::NodeP<T> idle = nodes[0]; idle.ncontact = 0; for (int i = 0; i < 8; ++i) idle.contact[i] = 0; nodes.push_back(idle); }
    std::vector<dj::ContactP<T>> contacts; for (auto& c : s->M.contacts) contacts.push_back(dj::cast_contact<T>(c));
    if (contacts.empty()) contacts.push_back(dj::ContactP<T>());
    HIPCHK(hipMalloc(&s->d_nodes, nodes.size() * sizeof(dj::NodeP<T>)));
    HIPCHK(hipMalloc(&s->d_contacts, contacts.size() * sizeof(dj::ContactP<T>)));
    HIPCHK(hipMemcpy(s->d_nodes, nodes.data(), nodes.size() * sizeof(dj::NodeP<T>), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_contacts, contacts.data(), contacts.size() * sizeof(dj::ContactP<T>), hipMemcpyHostToDevice));
    if (s->M.has_tsd) {
        std::vector<dj::TraSD<T>> tsd;
        for (auto& a : s->M.tsd) { dj::TraSD<T> b; b.spring = T(a.spring); b.damper = T(a.damper); for (int i = 0; i < 3; ++i) b.off[i] = T(a.off[i]); b.lim_lo = T(a.lim_lo); b.lim_hi = T(a.lim_hi); b.nlim = a.nlim; tsd.push_back(b); }
        HIPCHK(hipMalloc(&s->d_tsd, tsd.size() * sizeof(dj::TraSD<T>)));
        HIPCHK(hipMemcpy(s->d_tsd, tsd.data(), tsd.size() * sizeof(dj::TraSD<T>), hipMemcpyHostToDevice));
    }
    if (s->M.has_cut) {
        std::vector<dj::NodeP<T>> cn; for (auto& n_ : s->M.cuts) cn.push_back(dj::cast_node<T>(n_));
        HIPCHK(hipMalloc(&s->d_cuts, cn.size() * sizeof(dj::NodeP<T>)));
        HIPCHK(hipMemcpy(s->d_cuts, cn.data(), cn.size() * sizeof(dj::NodeP<T>), hipMemcpyHostToDevice));
        HIPCHK(hipMalloc(&s->d_cutws, (size_t)s->B * dj::CUTWS * sizeof(T)));
    }
    if (s->M.has_mlim) {
        std::vector<dj::MLimP<T>> ml;
        for (auto& a : s->M.mlim) { dj::MLimP<T> b; b.nt = a.nt; b.nr = a.nr; for (int i = 0; i < 6; ++i) { b.lo[i] = T(a.lo[i]); b.hi[i] = T(a.hi[i]); } ml.push_back(b); }
        HIPCHK(hipMalloc(&s->d_mlim, ml.size() * sizeof(dj::MLimP<T>)));
        HIPCHK(hipMemcpy(s->d_mlim, ml.data(), ml.size() * sizeof(dj::MLimP<T>), hipMemcpyHostToDevice));
    }
    std::vector<int> order;
    for (int lev = 0; lev <= s->M.maxlevel; ++lev) for (int b = 0; b < s->M.Nb; ++b) if (s->M.nodes[b].level == lev) order.push_back(b);
    HIPCHK(hipMalloc((void**)&s->d_order, order.size() * sizeof(int)));
    HIPCHK(hipMemcpy(s->d_order, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
    return DOJO_OK;
}

int mapping_waves(const dj::HostModel& M);
bool quad_mapping_of(const DojoSim* s);
// A step with the iteration cap in force (dojo_set_iteration_cap) is launched in phases: the step kernel of every environment group
// (PH_MAIN), each followed on its stream by the IFT kernel of the workgroups it finished (PH_GRAD); behind the step kernels of ALL groups,
// on a stream of its own, the continuation of the listed workgroups and their IFT (PH_CONT, once over the whole batch).  PH_ALL = no cap:
// step kernel + IFT kernel in one go.
enum { PH_ALL = 0, PH_MAIN = 1, PH_GRAD = 2, PH_CONT = 3 };
// wavefronts per workgroup of the quad mapping for this mechanism; 0 = lane mapping
int mapping_waves(const dj::HostModel& M) {
    // translational springs / dampers / limits: the quad builds that carry them are the single-wavefront ones with <= 4 contacts per body;
    // larger mechanisms take the lane mapping (its DJ_TSD builds: k_*_{4,8}_0_tsd)
    if (M.has_mlim || M.has_cut) return 0;                                // joint limits on several coordinates / both halves, kinematic loops: k_*_{4,8}_0_gen
    if (M.has_tsd && (M.S > 16 || M.maxc > 4)) return 0;
    if (M.contact_model == 2 && (M.S > 16 || M.maxc > 4)) return 0;       // LinearContact likewise (k_*_{4,8}_0_lin)
    if (M.S <= 16) return 1;
    if (M.S <= 32 && M.maxc <= 4 && M.Nc <= 16) return 2;
    return 0;
}

bool quad_mapping_of(const DojoSim* s) { return mapping_waves(s->M) > 0; }

int drain_slot(DojoSim* s, DojoSim::Ev3& e) {
    if (!e.used) return DOJO_OK;
    HIPCHK(hipEventSynchronize(e.b));
    float t = 0, t1 = 0;
    HIPCHK(hipEventElapsedTime(&t, e.a, e.b));
    if (e.has_mid) { HIPCHK(hipEventElapsedTime(&t1, e.a, e.m)); s->acc_step_ms += t1; s->acc_ift_ms += (double)t - t1; }
    else s->acc_step_ms += t;
    s->acc_n += e.n;
    e.used = false;
    return DOJO_OK;
}
int acquire_slot(DojoSim* s, int* idx) {
    if (s->ring.empty()) {
        s->ring.resize(64);
        for (auto& e : s->ring) { HIPCHK(hipEventCreate(&e.a)); HIPCHK(hipEventCreate(&e.m)); HIPCHK(hipEventCreate(&e.b)); }
    }
    *idx = (int)(s->ring_next++ % s->ring.size());
    return drain_slot(s, s->ring[*idx]);
}

// Environments are independent, so a batch is stepped as NG groups on internal streams: a group that contains an
// environment running into max_iter (one wavefront, ~5x the mean step time) delays only itself while the other groups'
// launches keep the GPU busy.  ROCm multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and
// streams that share a queue serialize, so the count stays below that.
// Refinement threshold: explicit (dojo_set_refinement) or tied to the requested tolerances -- the reference's defaults
// (rtol 1e-6, btol 1e-4) are met by the plain solves; tighter ones enable the refining kernels.
double refine_threshold(const DojoSim* s) {
    if (s->M.contact_model == 2) return (double)INFINITY;       // LinearContact builds carry no refining kernels
    if (s->M.has_ss) return (double)INFINITY;                   // ... nor do the body-body contact builds
    return s->refine_w >= 0.0 ? s->refine_w : ((s->opts.rtol <= 1e-7 || s->opts.btol <= 1e-6) ? DOJO_DEFAULT_REFINE_STIFFNESS : (double)INFINITY);
}
// scalars per contact in the exported [s; gamma] block: 8 (NonlinearContact; ImpactContact uses the first of each four), 12 (LinearContact)
size_t csg_per(const DojoSim* s) { return s->M.contact_model == 2 ? 12 : 8; }
// does the batch have more workgroups than the GPU holds at once (these kernels: one wavefront per SIMD, four SIMDs per compute unit)?
bool several_rounds(const DojoSim* s) {
    const int NW = mapping_waves(s->M);
    const size_t E = (size_t)std::max(1, 64 * (NW > 0 ? NW : 1) / (s->M.S * (NW > 0 ? 4 : 1)));
    return ((size_t)s->B + E - 1) / E * (size_t)std::max(NW, 1) > (size_t)4 * (size_t)(s->sm_count > 0 ? s->sm_count : 256);
}
size_t group_count(const DojoSim* s, bool want, bool joined_steps = false) {
    const size_t B = (size_t)s->B;
    size_t NG = (want && B >= 512) ? std::min<size_t>(16, B / 256) : 1;
    // An asynchronous handle (dojo_set_async) chains its groups' steps: a group that holds a long solve delays only itself, so below 4096 environments
    // smaller groups hide more of that tail than 256-environment ones -- down to 64 WORKGROUPS per group, for the single-wavefront quad mapping
    // (Ant, Quadruped: one environment per wavefront; the Block's sixteen per wavefront leave its B = 1024 at four groups: sixteen launches of
    // four wavefronts halved its rate).  profiles/r06_f_small_batches.txt: Ant B = 1024 0.47 -> 0.60 M, B = 2048 0.94 -> 1.02-1.09 M, B = 512
    // 0.30 -> 0.37 M; Quadruped B = 1024 0.48 -> 0.66 M.  The two-wavefront mapping keeps the 256-environment groups (Atlas B = 2048 with 16 groups:
    // +4 % on BASELINE's perturbation, -1 % standing, -19 % over the landing steps), and so does a handle that joins after every step: the groups'
    // launches are on its critical path.
    if (s->async && want && mapping_waves(s->M) == 1) {
        const size_t per_wg = std::max<size_t>(1, 16 / (size_t)std::max(1, s->M.S));      // environments per 64-lane workgroup (16 supernode slots)
        const size_t min_envs = 64 * per_wg;
        if (B >= 2 * min_envs) NG = std::max<size_t>(NG, std::min<size_t>(16, B / min_envs));
    }
    // A handle that joins after every step and whose batch takes several rounds of workgroups: two groups.  Each launch then has the whole GPU for
    // its rounds (with the dispatch order: its long solves first), and the IFT kernel of the first group runs under the tail of the second's step
    // kernel.  profiles/r06_i_dispatch.txt, joined ms per step at 1 / 2 / 4 / 8 / 16 groups: Ant B = 4096 5.93 / 5.84 / 6.19 / 6.14 / 6.28
    // (batch order: 6.45 / 6.22 / 6.47 / 6.48 / 6.49), Atlas B = 2048 6.57 / 6.35 / 6.70 / 6.89 / 6.85, Quadruped B = 8192 12.5 / 12.1 / 12.1 / 12.0 / 11.9.
    if (joined_steps && !s->async && want && several_rounds(s)) NG = std::min<size_t>(NG, 2);
    if (s->groups > 0) NG = std::min<size_t>(std::min<size_t>((size_t)s->groups, 16), std::max<size_t>(1, B / 64));   // (more than 16 queues in flight collapse: 0.68 M against 1.00 M at 24, same session)
    const char* hq = getenv("GPU_MAX_HW_QUEUES");
    const int nq = hq ? atoi(hq) : 4;
    NG = std::min<size_t>(NG, (size_t)std::max(1, nq - 1));
    // The refining kernels keep 2-6 KB of scratch per lane, and ROCr sizes a hardware queue's scratch for a full GPU of such
    // wavefronts (~3 GB): sixteen queues asking for it at once end in HSA_STATUS_ERROR_OUT_OF_RESOURCES (the queue aborts the
    // process).  With refinement in force the batch is stepped as at most three groups (seen green; the default queue count).
    if (std::isfinite(refine_threshold(s))) NG = std::min<size_t>(NG, 3);
    // The general lane-mapping builds (several limits per joint, cut elements) keep ~40 KB of scratch per lane (the small dense system of the cut
    // elements, the per-coordinate limit rows): one queue's worth of it at a time
    if (s->M.has_mlim || s->M.has_cut) NG = 1;
    return NG;
}
int ensure_groups(DojoSim* s, size_t NG) {
    while (s->gstreams.size() < NG) {
        hipStream_t g_; HIPCHK(hipStreamCreateWithFlags(&g_, hipStreamNonBlocking)); s->gstreams.push_back(g_);
        hipEvent_t gev_; HIPCHK(hipEventCreateWithFlags(&gev_, hipEventDisableTiming)); s->gevents.push_back(gev_);
    }
    if (!s->fork_event) HIPCHK(hipEventCreateWithFlags(&s->fork_event, hipEventDisableTiming));
    return DOJO_OK;
}
// pipelined groups (dojo_set_async(h, 2)): a second stream per group for its IFT kernels, events, the second hand-off record
// (the IFT streams are shared: ROCm runs ~20 streams side by side before the hardware queues are time-sliced -- 24 collapse --, so the 16 groups get
//  pipe_streams() of them, group g on stream g % n; the IFT kernels of the groups that share one run one after the other, off the critical path)
size_t pipe_streams() { static const char* e_ = getenv("DOJO_PIPE_STREAMS"); const int n = e_ ? atoi(e_) : 4; return (size_t)std::max(1, std::min(n, 16)); }
int ensure_pipe(DojoSim* s, size_t NG) {
    while (s->gstreams2.size() < std::min(NG, pipe_streams())) {
        hipStream_t g_; HIPCHK(hipStreamCreateWithFlags(&g_, hipStreamNonBlocking)); s->gstreams2.push_back(g_);
        hipEvent_t jev_; HIPCHK(hipEventCreateWithFlags(&jev_, hipEventDisableTiming)); s->gevents2.push_back(jev_);
    }
    while (s->grad_done[0].size() < NG)
        for (int p = 0; p < 2; ++p) { hipEvent_t dev_; HIPCHK(hipEventCreateWithFlags(&dev_, hipEventDisableTiming)); s->grad_done[p].push_back(dev_); }
    while (s->main_events.size() < NG) { hipEvent_t ev_; HIPCHK(hipEventCreateWithFlags(&ev_, hipEventDisableTiming)); s->main_events.push_back(ev_); }
    if (!s->d_sol2) HIPCHK(hipMalloc(&s->d_sol2, (size_t)s->B * s->M.S * dj::sol_record<8, true>() * sizeof(double)));
    return DOJO_OK;
}
// the caller's stream waits for everything the environment groups have in flight
int join_groups(DojoSim* s, hipStream_t st) {
    if (!s->pending) return DOJO_OK;
    for (size_t gi = 0; gi < s->gstreams.size(); ++gi) { HIPCHK(hipEventRecord(s->gevents[gi], s->gstreams[gi])); HIPCHK(hipStreamWaitEvent(st, s->gevents[gi], 0)); }
    for (size_t gi = 0; gi < s->gstreams2.size(); ++gi) { HIPCHK(hipEventRecord(s->gevents2[gi], s->gstreams2[gi])); HIPCHK(hipStreamWaitEvent(st, s->gevents2[gi], 0)); }
    s->pending = false;
    return DOJO_OK;
}

// the iteration cap in force for this handle's next step, 0 = none: the continuation kernel exists for the single-wavefront quad mapping
// (<= 16 bodies) with plain solves (no refinement), NonlinearContact / no contacts (the variants that carry an IFT)
int effective_cap(const DojoSim* s) {
    static const char* ecap_ = getenv("DOJO_ITER_CAP");
    const int cap = s->iter_cap >= 0 ? s->iter_cap : (ecap_ ? atoi(ecap_) : DOJO_DEFAULT_ITERATION_CAP);
    if (cap <= 0 || cap >= s->opts.max_iter) return 0;
    if (mapping_waves(s->M) != 1 || std::isfinite(refine_threshold(s))) return 0;
    if (s->M.contact_model != 0 || s->M.has_ss) return 0;
    return cap;
}
// what a capped step needs besides the groups' streams: the continuation's stream, its end event, a "step kernel done" event per group,
// the list and its count (zeroed here, on `st`, before the step kernels are forked off it)
int begin_capped_step(DojoSim* s, size_t NG, hipStream_t st) {
    if (!s->cstream) {
        // (DOJO_CONT_PRIORITY=1: a high-priority stream.  Measured: with it the step kernels of the FOLLOWING step take 20..300 ms on some
        //  queues -- the scheduler's handling of queue priorities, presumably wave save / restore -- so the default is a plain stream, and
        //  the order of submission below is what puts the continuation in front of the IFT kernels)
        static const bool prio_ = getenv("DOJO_CONT_PRIORITY") != nullptr && atoi(getenv("DOJO_CONT_PRIORITY")) != 0;
        int lo_ = 0, hi_ = 0; HIPCHK(hipDeviceGetStreamPriorityRange(&lo_, &hi_));
        if (prio_) HIPCHK(hipStreamCreateWithPriority(&s->cstream, hipStreamNonBlocking, hi_));
        else HIPCHK(hipStreamCreateWithFlags(&s->cstream, hipStreamNonBlocking));
    }
    if (!s->allmain_event) HIPCHK(hipEventCreateWithFlags(&s->allmain_event, hipEventDisableTiming));
    if (!s->cont_event) HIPCHK(hipEventCreateWithFlags(&s->cont_event, hipEventDisableTiming));
    while (s->main_events.size() < NG) { hipEvent_t ev_; HIPCHK(hipEventCreateWithFlags(&ev_, hipEventDisableTiming)); s->main_events.push_back(ev_); }
    if (!s->d_cont_list) {
        const int NW = mapping_waves(s->M), E = 64 * NW / (s->M.S * 4);
        HIPCHK(hipMalloc((void**)&s->d_cont_list, (((size_t)s->B + E - 1) / E + 1) * sizeof(int))); HIPCHK(hipMalloc((void**)&s->d_cont_count, sizeof(int)));
    }
    HIPCHK(hipMemsetAsync(s->d_cont_count, 0, sizeof(int), st));
    return DOJO_OK;
}

// Launches the step (and IFT) kernels for the environments [env0, env0 + nenv) of the batch; all pointers are the
// batch-level buffers.  env0 must be a multiple of the environments per wavefront.
template <class TIO, class T, class TL>
int launch(DojoSim* s, const void* z, const void* u, void* zn, int* status, int* iters, void* vel, void* jimp, void* csg,
           void* dz, void* du, hipStream_t st, bool timed, size_t env0 = 0, int nenv = -1, void* dc = nullptr, void* storage = nullptr, int phase = PH_ALL) {
    if (nenv < 0) nenv = s->B;
    const size_t Nb = s->M.Nb, nu = s->M.nu, nx = 12 * Nb;
    auto off = [&](const void* p, size_t per_env) -> TIO* { return p ? (TIO*)p + env0 * per_env : (TIO*)nullptr; };
    dj::KernelArgs<TIO, T> A;
    const double rw_ = refine_threshold(s);
    A.G = dj::make_globals<T>(s->M, s->opts, s->grad_mode, rw_);
    A.nodes = (const dj::NodeP<T>*)s->d_nodes; A.contacts = (const dj::ContactP<T>*)s->d_contacts; A.B = nenv;
    A.z = off(z, 13 * Nb); A.u = off(u, nu); A.z_next = off(zn, 13 * Nb); A.fext = off(s->fext, 6 * Nb);
    A.status = status ? status + env0 : nullptr; A.iters = iters ? iters + env0 : nullptr;
    A.vel = off(vel, 6 * Nb); A.joint_imp = off(jimp, s->M.n_joint_imp); A.contact_sg = off(csg, csg_per(s) * s->M.Nc);
    A.dz = off(dz, nx * nx); A.du = off(du, nx * nu); A.dc = off(dc, nx * 5 * s->M.Nc);
    A.res = storage ? off(s->d_res, 6 * Nb) : (TIO*)nullptr;
    A.tsd = s->M.has_tsd ? (const dj::TraSD<T>*)s->d_tsd : nullptr;
    A.mlim = s->M.has_mlim ? (const dj::MLimP<T>*)s->d_mlim : nullptr;
    A.cuts = s->M.has_cut ? (const dj::NodeP<T>*)s->d_cuts : nullptr; A.ncut = (int)s->M.cuts.size();
    A.cutws = s->M.has_cut ? (T*)s->d_cutws + env0 * (size_t)dj::CUTWS : nullptr;
    A.mu_out = s->d_mu ? (T*)s->d_mu + env0 : nullptr;
    A.diag_out = (s->d_diag && quad_mapping_of(s)) ? (T*)s->d_diag + 2 * env0 : nullptr;
    // mapping: four lanes per supernode when the mechanism has <= 16 bodies (one Ant per wavefront) or <= 32 bodies
    // (one Atlas per two-wavefront workgroup; contact rows pooled per contact: <= 16 contacts, <= 4 per body);
    // else one lane per supernode
    const int NW = mapping_waves(s->M);
    const bool quad = NW > 0;
    if (NW == 1 || NW == 2) dj::set_row_passes(A.G, s->M);             // the factorization's level passes in the row layout (dojo_device.hpp, factorize_rows), where they pay
    int E = 64 * (quad ? NW : 1) / (s->M.S * (quad ? 4 : 1));
    dim3 grid((nenv + E - 1) / E);
    const size_t waves_total = (s->B + E - 1) / E, wave0 = env0 / E;        // workgroups, each 64 * NW lanes
    const int g = (dz != nullptr) || (dc != nullptr);
    // (refusals first: a timing slot taken before them would never be marked used)
    if (g && (s->M.has_ss || s->M.has_cc)) { g_err = "gradients are not available for mechanisms with a body-body contact"; return DOJO_ERR_UNSUPPORTED; }
    if (g && s->M.contact_model != 0) {   // the reference has no data Jacobians for ImpactContact / LinearContact either (src/gradients/data.jl:152-192 are NonlinearContact methods)
        g_err = "gradients are not available for ImpactContact / LinearContact mechanisms"; return DOJO_ERR_UNSUPPORTED;
    }
    if (g && quad && std::max<size_t>(2 * Nb + (nu + 5) / 6, dc != nullptr ? (size_t)s->M.Nc : 0) > 128) {
        // (dojo_device.hpp, gradient_columns_quad: the branch schedule keeps batch sets in two 64-bit words; unreachable with <= 32 bodies -- 2 x 32 + 32 --
        //  but a mapping change must not turn it into silently aliased batches)
        g_err = "the IFT sweeps support at most 128 column batches per environment"; return DOJO_ERR_UNSUPPORTED;
    }
    if (dc != nullptr && s->M.Nc > 64) {  // the sweeps' batch masks hold one bit per contact of the environment (dojo_device.hpp, sweep_masks)
        g_err = "contact-data gradients support at most 64 contacts per environment"; return DOJO_ERR_UNSUPPORTED;
    }
    int slot = -1;
    if (phase == PH_CONT) timed = false;
    if (timed && phase == PH_GRAD) slot = s->phase_slot;
    else if (timed) { int rc_ = acquire_slot(s, &slot); if (rc_ != DOJO_OK) return rc_; HIPCHK(hipEventRecord(s->ring[slot].a, st)); s->phase_slot = slot; }
    A.sol = nullptr;
    // doubles per supernode of the step -> IFT hand-off record in the kernels that serve this mechanism (the launcher selection below: MAXC by mapping)
    const int vmaxc = !quad ? (s->M.maxc <= 4 ? 4 : 8) : NW == 2 ? (s->M.maxc <= 1 ? 1 : 4) : (s->M.maxc <= 1 ? 1 : s->M.maxc <= 4 ? 4 : 8);
    const size_t sol_rec = (s->M.has_mlim || s->M.has_cut) ? (vmaxc == 4 ? dj::sol_record<4, true>() : dj::sol_record<8, true>())
                                         : vmaxc == 1 ? dj::sol_record<1>() : vmaxc == 4 ? dj::sol_record<4>() : dj::sol_record<8>();
    if (g) {
        if (!s->d_sol) HIPCHK(hipMalloc(&s->d_sol, (size_t)s->B * s->M.S * dj::sol_record<8, true>() * sizeof(T)));   // sized for the largest record
        A.sol = (T*)((s->sol_cur && s->d_sol2) ? s->d_sol2 : s->d_sol) + env0 * s->M.S * sol_rec;   // (the record size of the kernels of this mechanism: a launch over the whole batch
                                                                            //  -- the continuation -- must find the records where the groups' launches put them)
        if (quad && !s->d_fac) HIPCHK(hipMalloc(&s->d_fac, waves_total * dj::FAC_PER_LANE * 64 * NW * sizeof(T)));
        if (quad && !s->d_lu) HIPCHK(hipMalloc(&s->d_lu, waves_total * dj::LU_PER_LANE * 64 * NW * sizeof(T)));
    }
    // the explicit inverses of the Newton loop travel only when somebody reads them: the refining IFT kernel
    const bool want_fac = A.G.refine_w < INFINITY;
    A.fac = (g && quad && want_fac) ? (T*)s->d_fac + wave0 * dj::FAC_PER_LANE * 64 * NW : nullptr;
    A.lu = (g && quad) ? (T*)s->d_lu + wave0 * dj::LU_PER_LANE * 64 * NW : nullptr;
    A.ypark = nullptr; A.ypark_stride = 0;
    if (g && quad && sizeof(TIO) < sizeof(T)) {          // (dojo_device.hpp, gradient_columns_quad)
        const size_t batches = std::max<size_t>(2 * Nb + (nu + 5) / 6, (size_t)s->M.Nc);     // state + control batches | contact batches (dojo_cgrad_kernel)
        A.ypark_stride = (long long)(batches * 18 * (64 * NW));      // all four roles park (LU-form sweeps)
        if (!s->d_ypark) HIPCHK(hipMalloc(&s->d_ypark, waves_total * (size_t)A.ypark_stride * sizeof(T)));
        A.ypark = (T*)s->d_ypark + wave0 * (size_t)A.ypark_stride;
    }
    A.msg = nullptr; A.msg_stride = 0;
    if (g && quad) {                                       // (dojo_device.hpp, gradient_columns_quad: what the children of a root post to its body rows)
        size_t ntops = 0; for (auto& n_ : s->M.nodes) if (n_.level <= 1) ++ntops;      // one block per level-1 supernode (messages) and per root (its Δv, Δω)
        const size_t batches = std::max<size_t>(2 * Nb + (nu + 5) / 6, (size_t)s->M.Nc);
        A.msg_stride = (long long)(ntops * batches * 36 + 8);      // (+ a trash slot for the stores of columns that do not exist)
        if (A.msg_stride > 0) {
            if (!s->d_msg) HIPCHK(hipMalloc(&s->d_msg, (size_t)s->B * (size_t)A.msg_stride * sizeof(T)));
            A.msg = (T*)s->d_msg + env0 * (size_t)A.msg_stride;
        }
    }
    // Iteration cap (phased launches only; dojo_step_dev has checked that it applies and has zeroed the list's count): the solves that are
    // unfinished after `cap` Newton iterations leave the step kernel and go on in the continuation kernel (PH_CONT)
    A.G.iter_cap = 0;
    if (phase != PH_ALL) {
        if (!s->d_sol) HIPCHK(hipMalloc(&s->d_sol, (size_t)s->B * s->M.S * dj::sol_record<8, true>() * sizeof(T)));
        if (!s->d_resume) HIPCHK(hipMalloc(&s->d_resume, (size_t)s->B * dj::CARRY_PER_ENV * sizeof(T)));
        if (!status && !s->d_cstat) HIPCHK(hipMalloc((void**)&s->d_cstat, (size_t)s->B * sizeof(int)));
        A.sol = (T*)((s->sol_cur && s->d_sol2) ? s->d_sol2 : s->d_sol) + env0 * s->M.S * sol_rec;
        if (!s->plain_phases) {                               // (pipelined groups launch the two kernels apart without the cap's lists)
            A.G.iter_cap = effective_cap(s);
            A.resume = (T*)s->d_resume + env0 * dj::CARRY_PER_ENV;
            A.cont_list = s->d_cont_list; A.cont_count = s->d_cont_count; A.wave_base = (int)wave0;
        }
        if (!status) A.status = s->d_cstat + env0;
    }
    // Dispatch order (dojo_set_dispatch_order): this launch's step kernel hands its workgroups out by the permutation the previous launch over the same
    // environments left behind; the permutation for the next one is made behind this launch's kernels, from this step's iteration counts.
    bool reorder = false;
    if (phase == PH_ALL && dc == nullptr && s->dispatch_mode != 0) {
        // (mode 1: the order only matters when the GPU cannot hold the batch's workgroups at once)
        // (... and its launches are not chained next to other groups' -- an asynchronous handle of several groups, a rollout: a single launch per step
        //  waits for its last wavefront on an asynchronous handle too)
        reorder = s->dispatch_mode == 2 || ((!s->async || s->step_groups <= 1) && !s->chained_now && several_rounds(s));
    }
    if (reorder) {
        if (!s->d_dispatch) HIPCHK(hipMalloc((void**)&s->d_dispatch, (waves_total + 1) * sizeof(int)));
        if (!A.iters) {
            if (!s->d_liters) HIPCHK(hipMalloc((void**)&s->d_liters, (size_t)s->B * sizeof(int)));
            A.iters = s->d_liters + env0;
        }
        auto it_ = s->dispatch_have.find(wave0);
        if (it_ != s->dispatch_have.end() && it_->second == nenv) A.dispatch = s->d_dispatch + wave0;
    }
    A.blk = nullptr; A.flag = nullptr;
    if (quad && A.G.refine_w < INFINITY) {                  // the refining kernels follow the plain ones (dojo_kernels.hip)
        if (!s->d_blk) HIPCHK(hipMalloc(&s->d_blk, waves_total * 90 * 64 * NW * sizeof(T)));
        if (!s->d_flag) HIPCHK(hipMalloc((void**)&s->d_flag, (size_t)s->B * sizeof(int)));
        A.blk = (T*)s->d_blk + wave0 * 90 * 64 * NW;
        A.flag = s->d_flag + env0;
    }
    // Test hook (tests/conftest.py sets it for the GPU tier): the Jacobian buffers are filled with NaN bit patterns before the IFT kernels run, so that
    // an entry the device fails to write comes back as NaN instead of whatever the allocation held (the kernels must write every entry).
    static const bool poison_ = getenv("DOJO_POISON_OUTPUTS") != nullptr;
    if (poison_ && g && (phase == PH_ALL || phase == (s->plain_phases ? PH_GRAD : PH_MAIN))) {    // (pipelined: on the IFT's stream, behind the previous step's IFT)
        if (dc != nullptr) HIPCHK(hipMemsetAsync(A.dc, 0xFF, (size_t)nenv * nx * 5 * s->M.Nc * sizeof(TIO), st));
        else {
            if (A.dz) HIPCHK(hipMemsetAsync(A.dz, 0xFF, (size_t)nenv * nx * nx * sizeof(TIO), st));
            if (A.du && nu > 0) HIPCHK(hipMemsetAsync(A.du, 0xFF, (size_t)nenv * nx * nu * sizeof(TIO), st));
        }
    }
    typedef int (*launcher_t)(const void*, int, void*, int, void*);
    const bool f32 = sizeof(TIO) == 4;
    if (dc != nullptr) {                   // contact-data columns only: the hand-off of the last differentiable step is re-used
        if (!quad) { g_err = "contact-data gradients need the quad mapping (<= 32 bodies)"; return DOJO_ERR_UNSUPPORTED; }
        typedef int (*claunch_t)(const void*, int, void*);
        claunch_t cf = s->M.has_tsd ? (s->M.maxc <= 1 ? (f32 ? dojo_launch_cgrad_tsd_float_1_1 : dojo_launch_cgrad_tsd_double_1_1)
                                                      : (f32 ? dojo_launch_cgrad_tsd_float_4_1 : dojo_launch_cgrad_tsd_double_4_1))
                     : NW == 2 ? (s->M.maxc <= 1 ? (f32 ? dojo_launch_cgrad_float_1_2 : dojo_launch_cgrad_double_1_2) : (f32 ? dojo_launch_cgrad_float_4_2 : dojo_launch_cgrad_double_4_2))
                     : s->M.maxc <= 1 ? (f32 ? dojo_launch_cgrad_float_1_1 : dojo_launch_cgrad_double_1_1)
                     : s->M.maxc <= 4 ? (f32 ? dojo_launch_cgrad_float_4_1 : dojo_launch_cgrad_double_4_1)
                                      : (f32 ? dojo_launch_cgrad_float_8_1 : dojo_launch_cgrad_double_8_1);
        int lrc_ = cf(&A, (int)grid.x, (void*)st);
        if (lrc_ != 0) { g_err = std::string("kernel launch: ") + hipGetErrorString((hipError_t)lrc_); return DOJO_ERR_DEVICE; }
        HIPCHK(hipGetLastError());
        if (timed) { DojoSim::Ev3& e = s->ring[slot]; HIPCHK(hipEventRecord(e.b, st)); e.has_mid = false; e.n = 1; e.used = true; s->last_slot = slot; }
        return DOJO_OK;
    }
    launcher_t fn;
    if (s->M.has_ss && s->M.contact_model != 2) fn = f32 ? dojo_launch_ss_float_1_1 : dojo_launch_ss_double_1_1;
    else if (s->M.contact_model == 2 && !quad) fn = s->M.maxc <= 4 ? (f32 ? dojo_launch_lin_float_4_0 : dojo_launch_lin_double_4_0) : (f32 ? dojo_launch_lin_float_8_0 : dojo_launch_lin_double_8_0);
    else if (s->M.contact_model == 2) fn = s->M.maxc <= 1 ? (f32 ? dojo_launch_lin_float_1_1 : dojo_launch_lin_double_1_1) : (f32 ? dojo_launch_lin_float_4_1 : dojo_launch_lin_double_4_1);
    else if (s->M.has_mlim || s->M.has_cut) fn = s->M.maxc <= 4 ? (f32 ? dojo_launch_gen_float_4_0 : dojo_launch_gen_double_4_0) : (f32 ? dojo_launch_gen_float_8_0 : dojo_launch_gen_double_8_0);
    else if (s->M.has_tsd && !quad) fn = s->M.maxc <= 4 ? (f32 ? dojo_launch_tsd_float_4_0 : dojo_launch_tsd_double_4_0)
                                                       : (f32 ? dojo_launch_tsd_float_8_0 : dojo_launch_tsd_double_8_0);
    else if (s->M.has_tsd) fn = s->M.maxc <= 1 ? (f32 ? dojo_launch_tsd_float_1_1 : dojo_launch_tsd_double_1_1)
                                          : (f32 ? dojo_launch_tsd_float_4_1 : dojo_launch_tsd_double_4_1);
    else if (NW == 2) fn = s->M.maxc <= 1 ? (f32 ? dojo_launch_float_1_2 : dojo_launch_double_1_2) : (f32 ? dojo_launch_float_4_2 : dojo_launch_double_4_2);
    else if (quad) fn = s->M.maxc <= 1 ? (f32 ? dojo_launch_float_1_1 : dojo_launch_double_1_1)
                 : s->M.maxc <= 4 ? (f32 ? dojo_launch_float_4_1 : dojo_launch_double_4_1)
                                  : (f32 ? dojo_launch_float_8_1 : dojo_launch_double_8_1);
    else      fn = s->M.maxc <= 4 ? (f32 ? dojo_launch_float_4_0 : dojo_launch_double_4_0)
                                  : (f32 ? dojo_launch_float_8_0 : dojo_launch_double_8_0);
    const int phases = phase == PH_ALL ? (1 | (g ? 2 : 0)) : phase == PH_MAIN ? 1 : phase == PH_GRAD ? 2 : (4 | (g ? 8 : 0));
    const int lgrid = phase == PH_CONT ? (int)std::min<size_t>(waves_total, 128) : (int)grid.x;     // (a continuation workgroup takes a whole CU's LDS and loops over the list)
    int lrc = fn(&A, lgrid, (void*)st, phases, (timed && g && (phases & 1)) ? (void*)s->ring[slot].m : nullptr);
    if (lrc != 0) { g_err = std::string("kernel launch: ") + hipGetErrorString((hipError_t)lrc); return DOJO_ERR_DEVICE; }
    HIPCHK(hipGetLastError());
    if (timed && (phase == PH_ALL || phase == PH_GRAD || (phase == PH_MAIN && !g))) {
        DojoSim::Ev3& e = s->ring[slot]; HIPCHK(hipEventRecord(e.b, st)); e.has_mid = g != 0; e.n = 1; e.used = true; s->last_slot = slot;
    }
    if (reorder) {
        // (segments of another partition of the batch that overlap this one hold permutations of other ranges: forgotten before this one is written)
        for (auto it_ = s->dispatch_have.begin(); it_ != s->dispatch_have.end();) {
            const size_t a_ = it_->first, b_ = a_ + ((size_t)it_->second + E - 1) / E;
            if (a_ < wave0 + grid.x && wave0 < b_) it_ = s->dispatch_have.erase(it_); else ++it_;
        }
        hipLaunchKernelGGL(ckern::dispatch_order_kernel, dim3(1), dim3(1024), 0, st, (const int*)A.iters, nenv, E, (int)grid.x, s->d_dispatch + wave0);
        HIPCHK(hipGetLastError());
        s->dispatch_have[wave0] = nenv;
    }
    if (storage) {                         // record: the Storage rows of the environments of this launch
        const long long n = (long long)nenv * Nb; const int T_ = 128;
        hipLaunchKernelGGL((ckern::storage_kernel<TIO>), dim3((unsigned)((n + T_ - 1) / T_)), dim3(T_), 0, st, (const dj::NodeP<double>*)s->d_nodes,
                           (const dj::ContactP<double>*)s->d_contacts, (int)Nb, s->M.Nc, s->M.contact_model, (int)csg_per(s), s->M.dt, (int)env0, nenv,
                           (const TIO*)z, (const TIO*)vel, (const TIO*)csg, (const TIO*)s->d_res, (const TIO*)s->fext, (TIO*)storage);
        HIPCHK(hipGetLastError());
    }
    return DOJO_OK;
}

int launch_any(DojoSim* s, const void* z, const void* u, void* zn, int* status, int* iters, void* vel, void* jimp, void* csg,
               void* dz, void* du, hipStream_t st, bool timed, size_t env0 = 0, int nenv = -1, void* dc = nullptr, void* storage = nullptr, int phase = PH_ALL) {
    if (s->dtype == DOJO_DTYPE_F32) return launch<float, double, double>(s, z, u, zn, status, iters, vel, jimp, csg, dz, du, st, timed, env0, nenv, dc, storage, phase);
    return launch<double, double, double>(s, z, u, zn, status, iters, vel, jimp, csg, dz, du, st, timed, env0, nenv, dc, storage, phase);
}

// RCCL, opened on first use (dojo_comm_*)
namespace rccl {
typedef struct { char internal[128]; } UniqueId;
typedef int (*get_unique_id_t)(UniqueId*);
typedef int (*comm_init_rank_t)(void**, int, UniqueId, int);
typedef int (*all_gather_t)(const void*, void*, size_t, int /*ncclDataType_t*/, void*, hipStream_t);
typedef int (*comm_destroy_t)(void*);
typedef const char* (*get_error_string_t)(int);
get_unique_id_t get_unique_id = nullptr; comm_init_rank_t comm_init_rank = nullptr; all_gather_t all_gather = nullptr;
comm_destroy_t comm_destroy = nullptr; get_error_string_t get_error_string = nullptr;
bool load() {
    static std::mutex m; std::lock_guard<std::mutex> g(m);
    if (all_gather) return true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { g_err = std::string("cannot load librccl: ") + dlerror(); return false; }
    get_unique_id = (get_unique_id_t)dlsym(h, "ncclGetUniqueId"); comm_init_rank = (comm_init_rank_t)dlsym(h, "ncclCommInitRank");
    comm_destroy = (comm_destroy_t)dlsym(h, "ncclCommDestroy"); get_error_string = (get_error_string_t)dlsym(h, "ncclGetErrorString");
    all_gather_t ag = (all_gather_t)dlsym(h, "ncclAllGather");
    if (!get_unique_id || !comm_init_rank || !comm_destroy || !ag) { g_err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather"; return false; }
    all_gather = ag;
    return true;
}
const char* why(int rc) { return get_error_string ? get_error_string(rc) : "RCCL error"; }
}

int ensure(void** p, size_t bytes) {
    if (*p) return DOJO_OK;
    HIPCHK(hipMalloc(p, bytes ? bytes : 8));
    return DOJO_OK;
}

} // namespace

extern "C" {

const char* dojo_last_error(void) { return g_err.c_str(); }
// text of the last failure of a call on THIS handle ("" if none); the pointer stays valid until the next failing call on it
const char* dojo_handle_error(DojoHandle s) { if (!s) return ""; std::lock_guard<std::mutex> g(s->err_m); return s->err.c_str(); }

int dojo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void dojo_destroy(DojoHandle s);

int dojo_create(const DojoTopology* topo, int32_t batch, int32_t dtype, int32_t device, DojoHandle* out) {
    if (!topo || !out || batch < 1 || (dtype != DOJO_DTYPE_F64 && dtype != DOJO_DTYPE_F32)) { g_err = "dojo_create: bad argument"; return DOJO_ERR_INVALID; }
    int ndev = dojo_device_count();
    if (ndev <= 0) { g_err = "no HIP device visible: libdojo_hip has no CPU fallback"; return DOJO_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) { g_err = "dojo_create: device index out of range"; return DOJO_ERR_INVALID; }
    DojoSim* s = new DojoSim();
    int rc = dj::build_host_model(*topo, s->M);
    if (rc != DOJO_OK) { g_err = s->M.error; delete s; return rc; }
    if (s->M.has_ss && s->M.contact_model != 2 && (mapping_waves(s->M) != 1 || s->M.maxc > 1 || s->M.has_tsd || s->M.has_mlim || s->M.has_cut)) {
        // the tree-edge builds (k_*_1_1_ss) do not serve this mechanism: the contact travels as a cut element of the general lane-mapping builds
        rc = dj::promote_tree_edge_contacts(s->M);
        if (rc != DOJO_OK) { g_err = s->M.error; delete s; return rc; }
    }
    if (s->M.has_ss && (mapping_waves(s->M) != 1 || s->M.maxc > 1 || s->M.has_tsd)) {
        g_err = "a LinearContact body-body contact needs the single-wavefront quad mapping (<= 16 bodies), at most one contact per body and no translational springs / dampers / limits"; delete s; return DOJO_ERR_UNSUPPORTED;
    }
    if ((s->M.has_mlim || s->M.has_cut) && (s->M.contact_model == 2 || s->M.has_ss)) {   // (has_ss: a body-body contact along a tree edge -- the quad builds; next to cut elements: not built)
        g_err = "joint limits on several coordinates / both halves, or a kinematic loop, together with LinearContact or a body-body contact are not supported (no kernel build carries both)"; delete s; return DOJO_ERR_UNSUPPORTED;
    }
    if (s->M.contact_model == 2 && s->M.has_tsd) {
        g_err = "LinearContact together with translational springs / dampers / limits is not supported (no kernel build carries both)"; delete s; return DOJO_ERR_UNSUPPORTED;
    }
    s->B = batch; s->dtype = dtype; s->device = device; s->w = dtype == DOJO_DTYPE_F32 ? 4 : 8;
    s->opts = dj::default_options();
    if (hipSetDevice(device) != hipSuccess) { g_err = "dojo_create: hipSetDevice failed"; delete s; return DOJO_ERR_DEVICE; }
    if (hipDeviceGetAttribute(&s->sm_count, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) s->sm_count = 0;      // (several_rounds: 256 then)
    rc = upload_tables<double>(s);
    if (rc != DOJO_OK) { dojo_destroy(s); return rc; }          // frees whatever part of the tables was uploaded
    *out = s;
    return DOJO_OK;
}

void dojo_destroy(DojoHandle s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    void* ps[] = {s->d_tsd, s->d_mlim, s->d_cuts, s->d_cutws, s->d_sol2, s->d_fext, s->d_res, s->d_nodes, s->d_contacts, s->d_z, s->d_u, s->d_zn, s->d_vel, s->d_jimp, s->d_csg, s->d_dz, s->d_du, s->d_status, s->d_iters, s->d_sol, s->d_fac, s->d_lu, s->d_blk, s->d_ypark, s->d_msg, (void*)s->d_flag, s->d_cz, s->d_jf, (void*)s->d_mu, (void*)s->d_diag, s->d_order, s->d_x, s->d_xn, s->d_jm, s->d_jt, s->d_jb};
    for (void* p : ps) if (p) (void)hipFree(p);
    void* pc[] = {s->d_resume, (void*)s->d_cont_list, (void*)s->d_cont_count, (void*)s->d_cstat, (void*)s->d_dispatch, (void*)s->d_liters};
    for (void* p : pc) if (p) (void)hipFree(p);
    if (s->cstream) (void)hipStreamDestroy(s->cstream);
    if (s->cont_event) (void)hipEventDestroy(s->cont_event);
    if (s->allmain_event) (void)hipEventDestroy(s->allmain_event);
    for (auto e_ : s->main_events) (void)hipEventDestroy(e_);
    for (auto g_ : s->gstreams) (void)hipStreamDestroy(g_);
    for (auto g_ : s->gstreams2) (void)hipStreamDestroy(g_);
    for (auto e_ : s->gevents2) (void)hipEventDestroy(e_);
    for (int p_ = 0; p_ < 2; ++p_) for (auto e_ : s->grad_done[p_]) (void)hipEventDestroy(e_);
    for (auto gev_ : s->gevents) (void)hipEventDestroy(gev_);
    if (s->fork_event) (void)hipEventDestroy(s->fork_event);
    if (s->comm && rccl::comm_destroy) (void)rccl::comm_destroy(s->comm);
    for (auto& e : s->ring) { if (e.a) (void)hipEventDestroy(e.a); if (e.m) (void)hipEventDestroy(e.m); if (e.b) (void)hipEventDestroy(e.b); }
    delete s;
}

int dojo_get_dims(DojoHandle s, DojoDims* d) {
    Enter enter_(s);
    if (!s || !d) { g_err = "dojo_get_dims: bad argument"; return DOJO_ERR_INVALID; }
    d->n_bodies = s->M.Nb; d->n_joints = (int)s->M.nodes.size(); d->n_contacts = s->M.Nc;
    d->nz = 13 * s->M.Nb; d->nx = 12 * s->M.Nb; d->nu = s->M.nu; d->n_joint_impulses = s->M.n_joint_imp;
    d->n_solution = s->M.n_joint_imp + 6 * s->M.Nb + csg_per(s) * s->M.Nc; d->lanes_per_env = s->M.S;
    return DOJO_OK;
}

int dojo_set_options(DojoHandle s, const DojoSolverOptions* o) {
    Enter enter_(s);
    if (!s || !o || o->max_iter < 1 || o->max_ls < 1) { g_err = "dojo_set_options: bad argument"; return DOJO_ERR_INVALID; }
    s->opts = *o; return DOJO_OK;
}

// set_external_force!(body; force, torque, vertex) (src/bodies/set.jl:110-115) for every body of every environment: state.Fext
// (world frame) and state.τext (body frame) as they enter the body residual (integrators/constraint.jl:15-18)
int dojo_set_external_force_dev(DojoHandle s, const void* fext) {
    Enter enter_(s);
    if (!s) { g_err = "dojo_set_external_force_dev: bad argument"; return DOJO_ERR_INVALID; }
    s->fext = fext;
    return DOJO_OK;
}
int dojo_set_external_force(DojoHandle s, const void* fext) {
    Enter enter_(s);
    if (!s) { g_err = "dojo_set_external_force: bad argument"; return DOJO_ERR_INVALID; }
    if (!fext) { s->fext = nullptr; return DOJO_OK; }
    HIPCHK(hipSetDevice(s->device));
    const size_t bytes = (size_t)s->B * 6 * s->M.Nb * s->w;
    int rc;
    if ((rc = ensure(&s->d_fext, bytes))) return rc;
    HIPCHK(hipDeviceSynchronize());                          // steps still in flight read the previous forces
    HIPCHK(hipMemcpy(s->d_fext, fext, bytes, hipMemcpyHostToDevice));
    s->fext = s->d_fext;
    return DOJO_OK;
}

// Accuracy of the linear solves (no counterpart in the reference, whose LDU has no such knob): environments whose cone
// variables reach max γ/s > stiffness get their Newton and IFT solves refined against the uncondensed KKT system.
// INFINITY switches the refinement off, 0 refines every solve.
int dojo_set_refinement(DojoHandle s, double stiffness) {
    Enter enter_(s);
    if (!s || stiffness != stiffness) { g_err = "dojo_set_refinement: bad argument"; return DOJO_ERR_INVALID; }   // negative: back to the tolerance-based default
    s->refine_w = stiffness; return DOJO_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-GPU: one process per GPU, the batch sharded in contiguous slices, no exchange inside the solver; the
// one collective is an all-gather of per-rank outputs (final states of a rollout chunk, status, gradients when wanted) over
// RCCL / xGMI.  RCCL is opened lazily (dlopen), so a single-GPU user never loads it.  The 128-byte unique id travels over
// whatever the host already has between its processes (Julia Distributed / MPI; torch.distributed in bench.py).
// ---------------------------------------------------------------------------------------------------------------
// rank 0 creates the id (128 bytes) and the host hands it to every rank
int dojo_comm_unique_id(void* id128) {
    if (!id128) { g_err = "dojo_comm_unique_id: bad argument"; return DOJO_ERR_INVALID; }
    if (!rccl::load()) return DOJO_ERR_DEVICE;
    rccl::UniqueId id;
    int rc = rccl::get_unique_id(&id);
    if (rc != 0) { g_err = std::string("ncclGetUniqueId: ") + rccl::why(rc); return DOJO_ERR_DEVICE; }
    std::memcpy(id128, &id, 128);
    return DOJO_OK;
}
// joins the handle's device into a communicator of `world` ranks (one handle, one GPU, one process each)
int dojo_comm_init(DojoHandle s, int32_t rank, int32_t world, const void* id128) {
    Enter enter_(s);
    if (!s || !id128 || world < 1 || rank < 0 || rank >= world) { g_err = "dojo_comm_init: bad argument"; return DOJO_ERR_INVALID; }
    if (!rccl::load()) return DOJO_ERR_DEVICE;
    HIPCHK(hipSetDevice(s->device));
    if (s->comm) { HIPCHK(hipDeviceSynchronize()); (void)rccl::comm_destroy(s->comm); s->comm = nullptr; }     // (collectives of the old communicator may still be in flight)
    rccl::UniqueId id; std::memcpy(&id, id128, 128);
    int rc = rccl::comm_init_rank(&s->comm, world, id, rank);
    if (rc != 0) { s->comm = nullptr; g_err = std::string("ncclCommInitRank: ") + rccl::why(rc); return DOJO_ERR_DEVICE; }
    s->comm_rank = rank; s->comm_world = world;
    return DOJO_OK;
}
// recv[world][count] <- every rank's send[count] (scalars of the handle's dtype; int32 with as_int32), in rank order = the
// order of the contiguous batch shards.  Runs on `stream` behind everything the handle has in flight.
int dojo_allgather_dev(DojoHandle s, const void* send, void* recv, int64_t count, int32_t as_int32, void* stream) {
    Enter enter_(s);
    if (!s || !send || !recv || count < 0) { g_err = "dojo_allgather_dev: bad argument"; return DOJO_ERR_INVALID; }
    if (!s->comm) { g_err = "dojo_allgather_dev: no communicator (dojo_comm_init)"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    int rcj = join_groups(s, (hipStream_t)stream); if (rcj != DOJO_OK) return rcj;
    const int dt = as_int32 ? 2 /*ncclInt32*/ : (s->dtype == DOJO_DTYPE_F32 ? 7 /*ncclFloat32*/ : 8 /*ncclFloat64*/);
    int rc = rccl::all_gather(send, recv, (size_t)count, dt, s->comm, (hipStream_t)stream);
    if (rc != 0) { g_err = std::string("ncclAllGather: ") + rccl::why(rc); return DOJO_ERR_DEVICE; }
    return DOJO_OK;
}
int dojo_comm_info(DojoHandle s, int32_t* rank, int32_t* world) {
    Enter enter_(s);
    if (!s) { g_err = "dojo_comm_info: bad argument"; return DOJO_ERR_INVALID; }
    if (rank) *rank = s->comm_rank; if (world) *world = s->comm ? s->comm_world : 1;
    return DOJO_OK;
}

int dojo_set_gradient_mode(DojoHandle s, int32_t mode) {
    Enter enter_(s);
    if (!s || (mode != DOJO_GRAD_REFERENCE && mode != DOJO_GRAD_CONSISTENT)) { g_err = "dojo_set_gradient_mode: bad argument"; return DOJO_ERR_INVALID; }
    s->grad_mode = mode; return DOJO_OK;
}

int dojo_step_dev(DojoHandle s, const void* z, const void* u, void* z_next, int32_t* status, int32_t* iters, void* dz, void* du, void* stream) {
    Enter enter_(s);
    if (!s || !z || !z_next) { g_err = "dojo_step_dev: bad argument"; return DOJO_ERR_INVALID; }
    if ((dz == nullptr) != (du == nullptr) && s->M.nu > 0) { g_err = "dojo_step_dev: dz and du must both be given or both be NULL"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    size_t B = s->B, w = s->w;
    int rc;
    if ((rc = ensure(&s->d_vel, B * 6 * s->M.Nb * w))) return rc;
    if ((rc = ensure(&s->d_jimp, B * (s->M.n_joint_imp + 1) * w))) return rc;
    if ((rc = ensure(&s->d_csg, B * (csg_per(s) * s->M.Nc + 1) * w))) return rc;
    if ((rc = ensure((void**)&s->d_mu, B * sizeof(double)))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t NG = group_count(s, true, true);
    s->step_groups = NG;
    // Iteration cap (dojo_set_iteration_cap): in force for steps that are joined into the caller's stream -- there the step waits for its longest
    // solve.  An asynchronous handle chains its groups' steps without a barrier and hides that tail behind the other groups' kernels.
    const bool capped = !s->async && effective_cap(s) > 0;
    if (capped) {
        if ((rc = join_groups(s, st))) return rc;                  // (asynchronous steps still in flight)
        if ((rc = begin_capped_step(s, std::max<size_t>(NG, 1), st))) return rc;
    }
    if (NG <= 1) {
        if ((rc = join_groups(s, st))) return rc;
        if (capped) {
            // (the continuation is enqueued before the IFT of the finished workgroups and on a stream of higher priority: its workgroups need a
            //  whole CU's LDS each, which they only find before the IFT's wavefronts have spread over the GPU)
            if ((rc = launch_any(s, z, u, z_next, status, iters, s->d_vel, s->d_jimp, s->d_csg, dz, du, st, true, 0, -1, nullptr, nullptr, PH_MAIN))) return rc;
            HIPCHK(hipEventRecord(s->main_events[0], st));
            HIPCHK(hipStreamWaitEvent(s->cstream, s->main_events[0], 0));
            if ((rc = launch_any(s, z, u, z_next, status, iters, s->d_vel, s->d_jimp, s->d_csg, dz, du, s->cstream, false, 0, -1, nullptr, nullptr, PH_CONT))) return rc;
            HIPCHK(hipEventRecord(s->cont_event, s->cstream));
            if (dz && (rc = launch_any(s, z, u, z_next, status, iters, s->d_vel, s->d_jimp, s->d_csg, dz, du, st, true, 0, -1, nullptr, nullptr, PH_GRAD))) return rc;
            HIPCHK(hipStreamWaitEvent(st, s->cont_event, 0));
        } else
        rc = launch_any(s, z, u, z_next, status, iters, s->d_vel, s->d_jimp, s->d_csg, dz, du, st, true);
    } else if (capped) {
        // fork: the step kernel of every group, each followed by the IFT of what it finished; behind ALL step kernels the continuation, once
        // over the batch, on its own stream; join: the groups and the continuation into the caller's stream
        if ((rc = ensure_groups(s, NG))) return rc;
        const size_t per = ((B + NG - 1) / NG + 63) / 64 * 64;
        s->last_NG = NG; s->last_per = per;
        bool cont_recorded = false;
        // a failure after the fork: what is already running on the internal streams must be ordered before the caller's stream all the same
        auto unwind = [&](int rc_) { s->pending = true; (void)join_groups(s, st); if (cont_recorded) (void)hipStreamWaitEvent(st, s->cont_event, 0); return rc_; };
        HIPCHK(hipEventRecord(s->fork_event, st));
        for (size_t gi = 0; gi < NG; ++gi) {
            const size_t env0 = gi * per;
            if (env0 >= B) break;
            const int ne = (int)std::min(per, B - env0);
            HIPCHK(hipStreamWaitEvent(s->gstreams[gi], s->fork_event, 0));
            if ((rc = launch_any(s, z, u, z_next, status, iters, s->d_vel, s->d_jimp, s->d_csg, dz, du, s->gstreams[gi], true, env0, ne, nullptr, nullptr, PH_MAIN))) return unwind(rc);
            HIPCHK(hipEventRecord(s->main_events[gi], s->gstreams[gi]));
            HIPCHK(hipStreamWaitEvent(s->cstream, s->main_events[gi], 0));
            if (s->group_slot.size() <= gi) s->group_slot.resize(gi + 1, -1);
            s->group_slot[gi] = s->phase_slot;
        }
        // The continuation first: its workgroups take a whole CU's LDS each, which they only find while the GPU is empty -- so the groups' IFT
        // kernels wait until every step kernel is done as well (they would otherwise start behind their own group's step kernel, fill the CUs as
        // these drain, and keep the continuation out until they are through: measured, +1.2 ms per step), and the continuation's stream has
        // the higher priority.  The IFT kernels (1.2 ms of work) then run next to the continuation (2-3 ms on a few CUs).
        HIPCHK(hipEventRecord(s->allmain_event, s->cstream));
        if ((rc = launch_any(s, z, u, z_next, status, iters, s->d_vel, s->d_jimp, s->d_csg, dz, du, s->cstream, false, 0, -1, nullptr, nullptr, PH_CONT))) return unwind(rc);
        HIPCHK(hipEventRecord(s->cont_event, s->cstream)); cont_recorded = true;
        for (size_t gi = 0; gi < NG && dz; ++gi) {
            const size_t env0 = gi * per;
            if (env0 >= B) break;
            HIPCHK(hipStreamWaitEvent(s->gstreams[gi], s->allmain_event, 0));
            s->phase_slot = s->group_slot[gi];
            if ((rc = launch_any(s, z, u, z_next, status, iters, s->d_vel, s->d_jimp, s->d_csg, dz, du, s->gstreams[gi], true, env0, (int)std::min(per, B - env0), nullptr, nullptr, PH_GRAD))) return unwind(rc);
        }
        s->pending = true;
        if ((rc = join_groups(s, st))) return rc;
        HIPCHK(hipStreamWaitEvent(st, s->cont_event, 0));
    } else {
        // fork: every group waits for what the caller's stream holds (the inputs); group g of this call runs behind group g of
        // the previous call on the same internal stream.  join: the caller's stream waits for all groups -- unless the handle
        // is asynchronous (dojo_set_async), where consecutive calls chain per group and dojo_join() does it once.
        if ((rc = ensure_groups(s, NG))) return rc;
        const size_t per = ((B + NG - 1) / NG + 63) / 64 * 64;       // multiple of 64: whole wavefronts for every mapping
        // per-group chaining needs the same partition as the call still in flight: group g must cover the environments group g
        // covered.  Options, refinement or the group count may have changed it -- then everything in flight is joined first.
        if (s->pending && (s->last_NG != NG || s->last_per != per) && (rc = join_groups(s, st))) return rc;
        s->last_NG = NG; s->last_per = per;
        HIPCHK(hipEventRecord(s->fork_event, st));
        // Pipelined groups (dojo_set_async(h, 2), plain solves): the IFT kernel of a group's step k goes onto the group's SECOND stream, behind its
        // step kernel and next to the step kernel of step k + 1 -- both depend on step k alone.  Step k + 1 writes the other hand-off record; step
        // k + 2 re-uses record k's and waits for IFT k.  What this buys: while a group's step kernel drains (its launch lasts as long as its slowest
        // wavefront) the group has another kernel ready for the SIMDs that fall idle -- with 16 groups x 64 wavefronts = 1024 SIMDs there is no other
        // work to fill them.  The caller's inputs of un-joined calls stay untouched (the asynchronous contract): the IFT of step k reads z, u of step k.
        const bool piped = s->async == 2 && dz != nullptr && !std::isfinite(refine_threshold(s));
        if (piped) {
            if ((rc = ensure_pipe(s, NG))) return rc;
            s->sol_cur ^= 1; s->plain_phases = true;
            if (s->group_slot.size() < NG) s->group_slot.resize(NG, -1);
        }
        for (size_t gi = 0; gi < NG; ++gi) {
            const size_t env0 = gi * per;
            if (env0 >= B) break;
            const int ne = (int)std::min(per, B - env0);
            HIPCHK(hipStreamWaitEvent(s->gstreams[gi], s->fork_event, 0));
            if (piped) {
                HIPCHK(hipStreamWaitEvent(s->gstreams[gi], s->grad_done[s->sol_cur][gi], 0));      // (the IFT that read this record two calls ago)
                rc = launch_any(s, z, u, z_next, status, iters, s->d_vel, s->d_jimp, s->d_csg, dz, du, s->gstreams[gi], true, env0, ne, nullptr, nullptr, PH_MAIN);
                if (rc == DOJO_OK) {
                    s->group_slot[gi] = s->phase_slot;
                    HIPCHK(hipEventRecord(s->main_events[gi], s->gstreams[gi]));
                    hipStream_t st2 = s->gstreams2[gi % s->gstreams2.size()];
                    HIPCHK(hipStreamWaitEvent(st2, s->main_events[gi], 0));
                    s->phase_slot = s->group_slot[gi];
                    rc = launch_any(s, z, u, z_next, status, iters, s->d_vel, s->d_jimp, s->d_csg, dz, du, st2, true, env0, ne, nullptr, nullptr, PH_GRAD);
                    if (rc == DOJO_OK) HIPCHK(hipEventRecord(s->grad_done[s->sol_cur][gi], st2));
                }
            } else
            rc = launch_any(s, z, u, z_next, status, iters, s->d_vel, s->d_jimp, s->d_csg, dz, du, s->gstreams[gi], true, env0, ne);
            if (rc != DOJO_OK) { s->plain_phases = false; s->pending = true; (void)join_groups(s, st); return rc; }      // (the groups already launched stay ordered before the caller's stream)
        }
        s->plain_phases = false;
        s->pending = true;
        if (!s->async && (rc = join_groups(s, st))) return rc;
    }
    if (rc == DOJO_OK) s->have_solution = true;
    return rc;
}

// dojo_set_async(h, 1): dojo_step_dev no longer makes the caller's stream wait for the environment groups, so that
// consecutive steps of one group never wait for another group's straggler; dojo_join(h, stream) orders `stream` behind
// everything in flight (every host-pointer entry point does that implicitly).  dojo_set_groups: number of environment
// groups of dojo_step_dev (0 / negative: chosen from the batch size; 1: a single launch on the caller's stream).
int dojo_set_async(DojoHandle s, int32_t on) {
    Enter enter_(s);
    if (!s) { g_err = "dojo_set_async: bad argument"; return DOJO_ERR_INVALID; }
    s->async = on == 2 ? 2 : (on != 0 ? 1 : 0); return DOJO_OK;
}
// Order in which the step kernel's workgroups are handed to the GPU (no counterpart in the reference): 0 = batch order; 1 (default) = the longest
// solves of the previous step first, where that can matter -- steps joined into the caller's stream whose launch has more workgroups than the GPU
// holds at once: the launch ends with its last wavefront, and a 50-iteration solve that starts in the last round is that wavefront; 2 = always.
// Results do not depend on it.
int dojo_set_dispatch_order(DojoHandle s, int32_t mode) {
    Enter en_(s);
    if (!s || mode < 0 || mode > 2) { g_err = "dojo_set_dispatch_order: bad argument"; return DOJO_ERR_INVALID; }
    s->dispatch_mode = mode; return DOJO_OK;
}
int dojo_set_iteration_cap(DojoHandle s, int32_t cap) {
    Enter en_(s);
    if (!s) { g_err = "dojo_set_iteration_cap: bad argument"; return DOJO_ERR_INVALID; }
    s->iter_cap = cap <= 0 ? 0 : cap; return DOJO_OK;      // (0 or < 0: an EXPLICIT off -- DOJO_ITER_CAP in the environment only decides for handles that never called this)
}
int dojo_set_groups(DojoHandle s, int32_t n) {
    Enter enter_(s);
    if (!s) { g_err = "dojo_set_groups: bad argument"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());
    s->pending = false; s->groups = n > 0 ? n : -1; return DOJO_OK;
}
int dojo_join(DojoHandle s, void* stream) {
    Enter enter_(s);
    if (!s) { g_err = "dojo_join: bad argument"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    return join_groups(s, (hipStream_t)stream);
}

int dojo_step(DojoHandle s, const void* z, const void* u, void* z_next, int32_t* status, int32_t* iters, int32_t with_gradient) {
    Enter enter_(s);
    if (!s || !z || !z_next) { g_err = "dojo_step: bad argument"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    size_t B = s->B, w = s->w, nz = 13 * s->M.Nb, nx = 12 * s->M.Nb, nu = s->M.nu;
    int rc;
    if ((rc = ensure(&s->d_z, B * nz * w))) return rc;
    if ((rc = ensure(&s->d_zn, B * nz * w))) return rc;
    if ((rc = ensure(&s->d_u, B * (nu + 1) * w))) return rc;
    if ((rc = ensure((void**)&s->d_status, B * sizeof(int)))) return rc;
    if ((rc = ensure((void**)&s->d_iters, B * sizeof(int)))) return rc;
    if (with_gradient) {
        if ((rc = ensure(&s->d_dz, B * nx * nx * w))) return rc;
        if ((rc = ensure(&s->d_du, B * nx * (nu + 1) * w))) return rc;
    }
    HIPCHK(hipMemcpy(s->d_z, z, B * nz * w, hipMemcpyHostToDevice));
    if (u && nu) HIPCHK(hipMemcpy(s->d_u, u, B * nu * w, hipMemcpyHostToDevice));
    rc = dojo_step_dev(s, s->d_z, (u && nu) ? s->d_u : nullptr, s->d_zn, s->d_status, s->d_iters,
                       with_gradient ? s->d_dz : nullptr, with_gradient ? s->d_du : nullptr, nullptr);
    if (rc != DOJO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(z_next, s->d_zn, B * nz * w, hipMemcpyDeviceToHost));
    if (status) HIPCHK(hipMemcpy(status, s->d_status, B * sizeof(int), hipMemcpyDeviceToHost));
    if (iters) HIPCHK(hipMemcpy(iters, s->d_iters, B * sizeof(int), hipMemcpyDeviceToHost));
    s->have_grad = with_gradient != 0; s->have_u = (u && nu);
    return DOJO_OK;
}

// The mehrotra!(mechanism) seam (src/solver/mehrotra.jl:9): at that point set_input!/input_impulse! (src/mechanism/set.jl:40-53,
// src/joints/*/input.jl) have already folded the controls into every body's state.JF2 / state.Jτ2 and cleared the joints'
// inputs, so the drop-in hands over body impulses, not u.  jf [B, 6Nb] = [JF2 (world frame); Jτ2 (body frame)] per body, as
// they enter the body residual (src/integrators/constraint.jl:20-21: d -= [JF2; Jτ2]).  They share the external-force slot:
// -Δt·[Fext; τext] and -[JF2; Jτ2] are the same term (constraint.jl:15-18), so the kernels read fext + jf/Δt.
int dojo_step_impulses(DojoHandle s, const void* z, const void* jf, void* z_next, int32_t* status, int32_t* iters) {
    Enter enter_(s);
    if (!s || !z || !jf || !z_next) { g_err = "dojo_step_impulses: bad argument"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    const size_t B = s->B, w = s->w, nz = 13 * s->M.Nb, nf = 6 * s->M.Nb;
    int rc;
    if ((rc = ensure(&s->d_z, B * nz * w))) return rc;
    if ((rc = ensure(&s->d_zn, B * nz * w))) return rc;
    if ((rc = ensure(&s->d_jf, 2 * B * nf * w))) return rc;               // [raw impulses | folded forces]
    if ((rc = ensure((void**)&s->d_status, B * sizeof(int)))) return rc;
    if ((rc = ensure((void**)&s->d_iters, B * sizeof(int)))) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(s->d_z, z, B * nz * w, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_jf, jf, B * nf * w, hipMemcpyHostToDevice));
    void* folded = (char*)s->d_jf + B * nf * w;
    const long long n = (long long)(B * nf); const int T_ = 256;
    if (s->dtype == DOJO_DTYPE_F32) hipLaunchKernelGGL((ckern::fold_impulses_kernel<float>), dim3((unsigned)((n + T_ - 1) / T_)), dim3(T_), 0, nullptr, n, (const float*)s->fext, (const float*)s->d_jf, 1.0 / s->M.dt, (float*)folded);
    else hipLaunchKernelGGL((ckern::fold_impulses_kernel<double>), dim3((unsigned)((n + T_ - 1) / T_)), dim3(T_), 0, nullptr, n, (const double*)s->fext, (const double*)s->d_jf, 1.0 / s->M.dt, (double*)folded);
    HIPCHK(hipGetLastError());
    const void* user_fext = s->fext;
    s->fext = folded;
    rc = dojo_step_dev(s, s->d_z, nullptr, s->d_zn, s->d_status, s->d_iters, nullptr, nullptr, nullptr);
    s->fext = user_fext;
    if (rc != DOJO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(z_next, s->d_zn, B * nz * w, hipMemcpyDeviceToHost));
    if (status) HIPCHK(hipMemcpy(status, s->d_status, B * sizeof(int), hipMemcpyDeviceToHost));
    if (iters) HIPCHK(hipMemcpy(iters, s->d_iters, B * sizeof(int), hipMemcpyDeviceToHost));
    s->have_grad = false; s->have_u = false;
    return DOJO_OK;
}

int dojo_get_solution(DojoHandle s, void* vel, void* joint_imp, void* contact_sg) {
    Enter enter_(s);
    if (!s || !s->have_solution) { g_err = "dojo_get_solution: no step has been taken"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());
    size_t B = s->B, w = s->w;
    if (vel) HIPCHK(hipMemcpy(vel, s->d_vel, B * 6 * s->M.Nb * w, hipMemcpyDeviceToHost));
    if (joint_imp && s->M.n_joint_imp) HIPCHK(hipMemcpy(joint_imp, s->d_jimp, B * s->M.n_joint_imp * w, hipMemcpyDeviceToHost));
    if (contact_sg && s->M.Nc) HIPCHK(hipMemcpy(contact_sg, s->d_csg, B * csg_per(s) * s->M.Nc * w, hipMemcpyDeviceToHost));
    return DOJO_OK;
}

// mechanism.μ (the central-path parameter, src/solver/mehrotra.jl:45) of every environment when the last step's solve
// returned: what a mehrotra! drop-in writes back before it calls set_entries! to leave mechanism.system as the reference does
int dojo_get_mu(DojoHandle s, double* mu) {
    Enter enter_(s);
    if (!s || !mu || !s->have_solution || !s->d_mu) { g_err = "dojo_get_mu: no step has been taken"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(mu, s->d_mu, (size_t)s->B * sizeof(double), hipMemcpyDeviceToHost));
    return DOJO_OK;
}

// Diagnostics of the final linearization of every environment's last step (quad mappings), fp64 [B, 2]:
// [0] max γ/s over its cones (what dojo_set_refinement's threshold is compared with), [1] the largest multiplier of the
// un-pivoted Gauss-Jordan eliminations (growth: ~1 for a well-scaled step, >> 1e3 when a joint row repeats a stiff contact row).
// Switched on by the first call (the following steps record them).
int dojo_get_diagnostics(DojoHandle s, double* diag) {
    Enter enter_(s);
    if (!s) { g_err = "dojo_get_diagnostics: bad argument"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    if (!s->d_diag) {
        HIPCHK(hipMalloc((void**)&s->d_diag, (size_t)s->B * 2 * sizeof(double)));
        HIPCHK(hipMemset(s->d_diag, 0, (size_t)s->B * 2 * sizeof(double)));
    }
    HIPCHK(hipDeviceSynchronize());
    if (diag) HIPCHK(hipMemcpy(diag, s->d_diag, (size_t)s->B * 2 * sizeof(double), hipMemcpyDeviceToHost));
    return DOJO_OK;
}

int dojo_gradients(DojoHandle s, void* dz, void* du) {
    Enter enter_(s);
    if (!s || !s->have_grad) { g_err = "dojo_gradients: the last dojo_step was not run with with_gradient=1"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    size_t B = s->B, w = s->w, nx = 12 * s->M.Nb, nu = s->M.nu;
    // device layout is column-major per environment (Julia-native); the host ABI is row-major [B,nx,nx] / [B,nx,nu]
    std::vector<char> tmp(B * nx * std::max(nx, nu) * w);
    auto transpose_out = [&](void* dst, const void* src, size_t ncols) -> int {
        HIPCHK(hipMemcpy(tmp.data(), src, B * nx * ncols * w, hipMemcpyDeviceToHost));
        for (size_t b = 0; b < B; ++b)
            for (size_t c = 0; c < ncols; ++c)
                for (size_t r = 0; r < nx; ++r)
                    std::memcpy((char*)dst + ((b * nx + r) * ncols + c) * w, tmp.data() + ((b * ncols + c) * nx + r) * w, w);
        return DOJO_OK;
    };
    int rc;
    if (dz && (rc = transpose_out(dz, s->d_dz, nx))) return rc;
    if (du && nu && (rc = transpose_out(du, s->d_du, nu))) return rc;
    return DOJO_OK;
}

// simulate! with pre-sampled controls (src/simulation/simulate.jl:16-37): H steps, each fed with the previous step's
// internal next state; storage != null records save_to_storage! rows [H][B][Nb][25] of every solved step
static int rollout_core(DojoHandle s, const void* z0, const void* U, int32_t H, void* Z, int32_t* status, void* storage, void* stream) {
    if (!s || !z0 || H < 1) { g_err = "dojo_rollout_dev: bad argument"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    size_t B = s->B, w = s->w, nz = 13 * s->M.Nb, nu = s->M.nu;
    int rc;
    if ((rc = ensure(&s->d_z, B * nz * w))) return rc;
    if ((rc = ensure(&s->d_zn, B * nz * w))) return rc;
    if ((rc = ensure(&s->d_vel, B * 6 * s->M.Nb * w))) return rc;
    if ((rc = ensure(&s->d_jimp, B * (s->M.n_joint_imp + 1) * w))) return rc;
    if ((rc = ensure(&s->d_csg, B * (csg_per(s) * s->M.Nc + 1) * w))) return rc;
    if (storage && (rc = ensure(&s->d_res, B * 6 * s->M.Nb * w))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const char* cur = (const char*)z0;
    int slot = -1;
    { int rc_ = acquire_slot(s, &slot); if (rc_ != DOJO_OK) return rc_; }
    HIPCHK(hipEventRecord(s->ring[slot].a, st));
    // Environments are independent, so the batch is rolled out as NG groups on internal streams: a group whose step
    // contains an environment that runs into max_iter (one wavefront, ~5x the mean step time) delays only itself while
    // the other groups' launches keep the GPU busy.  (ROCm runs at most GPU_MAX_HW_QUEUES streams concurrently.)
    if ((rc = join_groups(s, st))) return rc;                  // (asynchronous steps still in flight)
    const size_t NG = group_count(s, H >= 2);
    size_t per = ((B + NG - 1) / NG + 63) / 64 * 64;            // group size: multiple of 64 (and so of the environments per wave)
    if (NG > 1) {
        if ((rc = ensure_groups(s, NG))) return rc;
        HIPCHK(hipEventRecord(s->fork_event, st));
    }
    const char* last = cur;
    for (size_t gi = 0; gi < NG; ++gi) {
        const size_t env0 = gi * per;
        if (env0 >= B) break;
        const int nenv = (int)std::min(per, B - env0);
        hipStream_t gs = NG > 1 ? s->gstreams[gi] : st;
        if (NG > 1) HIPCHK(hipStreamWaitEvent(gs, s->fork_event, 0));
        const char* c = (const char*)z0;
        for (int k = 0; k < H; ++k) {
            char* nxt = Z ? (char*)Z + (size_t)k * B * nz * w : (char*)((k & 1) ? s->d_z : s->d_zn);
            const char* uk = (U && nu) ? (const char*)U + (size_t)k * B * nu * w : nullptr;
            void* sk = storage ? (char*)storage + (size_t)k * B * 25 * s->M.Nb * w : nullptr;
            s->chained_now = true;
            rc = launch_any(s, c, uk, nxt, status ? status + (size_t)k * B : nullptr, nullptr, s->d_vel, s->d_jimp, s->d_csg, nullptr, nullptr, gs, false, env0, nenv, nullptr, sk);
            s->chained_now = false;
            if (rc != DOJO_OK) return rc;
            c = nxt;
        }
        last = c;
        if (NG > 1) { HIPCHK(hipEventRecord(s->gevents[gi], gs)); HIPCHK(hipStreamWaitEvent(st, s->gevents[gi], 0)); }
    }
    cur = last;
    { DojoSim::Ev3& e = s->ring[slot]; HIPCHK(hipEventRecord(e.b, st)); e.has_mid = false; e.n = H; e.used = true; s->last_slot = slot; }
    if (!Z && cur != (const char*)s->d_zn) HIPCHK(hipMemcpyAsync(s->d_zn, cur, B * nz * w, hipMemcpyDeviceToDevice, st));
    s->have_solution = true; s->have_grad = false;
    return DOJO_OK;
}

int dojo_rollout_dev(DojoHandle s, const void* z0, const void* U, int32_t H, void* Z, int32_t* status, void* stream) {
    Enter enter_(s);
    return rollout_core(s, z0, U, H, Z, status, nullptr, stream);
}

int dojo_simulate_dev(DojoHandle s, const void* z0, const void* U, int32_t H, void* Z, void* storage, int32_t* status, void* stream) {
    Enter enter_(s);
    if (!storage) { g_err = "dojo_simulate_dev: storage must not be NULL (use dojo_rollout_dev for record = false)"; return DOJO_ERR_INVALID; }
    return rollout_core(s, z0, U, H, Z, status, storage, stream);
}

static int rollout_host(DojoHandle s, const void* z0, const void* U, int32_t H, void* Z, void* storage, int32_t* status);

int dojo_rollout(DojoHandle s, const void* z0, const void* U, int32_t H, void* Z, int32_t* status) {
    Enter enter_(s);
    return rollout_host(s, z0, U, H, Z, nullptr, status);
}

int dojo_simulate(DojoHandle s, const void* z0, const void* U, int32_t H, void* Z, void* storage, int32_t* status) {
    Enter enter_(s);
    if (!storage) { g_err = "dojo_simulate: storage must not be NULL (use dojo_rollout for record = false)"; return DOJO_ERR_INVALID; }
    return rollout_host(s, z0, U, H, Z, storage, status);
}

static int rollout_host(DojoHandle s, const void* z0, const void* U, int32_t H, void* Z, void* storage, int32_t* status) {
    if (!s || !z0 || H < 1) { g_err = "dojo_rollout: bad argument"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    size_t B = s->B, w = s->w, nz = 13 * s->M.Nb, nu = s->M.nu;
    DevBuf dz0, dU, dZ, dSt, dS;
    const size_t nst = (size_t)H * B * 25 * s->M.Nb * w;
    if (storage) HIPCHK(dSt.alloc(nst));
    HIPCHK(dz0.alloc(B * nz * w));
    HIPCHK(hipMemcpy(dz0.p, z0, B * nz * w, hipMemcpyHostToDevice));
    if (U && nu) { HIPCHK(dU.alloc((size_t)H * B * nu * w)); HIPCHK(hipMemcpy(dU.p, U, (size_t)H * B * nu * w, hipMemcpyHostToDevice)); }
    if (Z) HIPCHK(dZ.alloc((size_t)H * B * nz * w));
    if (status) HIPCHK(dS.alloc((size_t)H * B * sizeof(int)));
    int rc = rollout_core(s, dz0.p, dU.p, H, dZ.p, (int32_t*)dS.p, dSt.p, nullptr);
    if (rc != DOJO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    if (storage) HIPCHK(hipMemcpy(storage, dSt.p, nst, hipMemcpyDeviceToHost));
    if (Z) HIPCHK(hipMemcpy(Z, dZ.p, (size_t)H * B * nz * w, hipMemcpyDeviceToHost));
    if (status) HIPCHK(hipMemcpy(status, dS.p, (size_t)H * B * sizeof(int), hipMemcpyDeviceToHost));
    if (Z) HIPCHK(hipMemcpy(s->d_zn, (char*)dZ.p + (size_t)(H - 1) * B * nz * w, B * nz * w, hipMemcpyDeviceToDevice));
    return DOJO_OK;
}

int dojo_get_state(DojoHandle s, void* z) {
    Enter enter_(s);
    if (!s || !z || !s->d_zn) { g_err = "dojo_get_state: no state"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(z, s->d_zn, (size_t)s->B * 13 * s->M.Nb * s->w, hipMemcpyDeviceToHost));
    return DOJO_OK;
}

// ---- contact-data gradients: get_contact_gradients, src/gradients/contact.jl:1-55 ----
int dojo_contact_gradients_dev(DojoHandle s, const void* z, const void* u, void* dc, void* stream) {
    Enter enter_(s);
    if (!s || !z || !dc) { g_err = "dojo_contact_gradients_dev: bad argument"; return DOJO_ERR_INVALID; }
    if (!s->d_sol) { g_err = "dojo_contact_gradients_dev: no differentiable step (dz/du requested) has been run on this handle"; return DOJO_ERR_INVALID; }
    if (s->M.Nc == 0) return DOJO_OK;
    HIPCHK(hipSetDevice(s->device));
    { int rcj = join_groups(s, (hipStream_t)stream); if (rcj != DOJO_OK) return rcj; }
    return launch_any(s, z, u, s->d_zn ? s->d_zn : (void*)z, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream, true, 0, -1, dc);
}
int dojo_contact_gradients(DojoHandle s, void* dc) {
    Enter enter_(s);
    if (!s || !dc) { g_err = "dojo_contact_gradients: bad argument"; return DOJO_ERR_INVALID; }
    if (!s->have_grad || !s->d_z) { g_err = "dojo_contact_gradients: call dojo_step(..., with_gradient = 1) first"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    size_t B = s->B, w = s->w, nx = 12 * s->M.Nb, ncc = 5 * (size_t)s->M.Nc;
    if (ncc == 0) return DOJO_OK;
    DevBuf d_dc;
    HIPCHK(d_dc.alloc(B * nx * ncc * w));
    int rc = dojo_contact_gradients_dev(s, s->d_z, s->have_u ? s->d_u : nullptr, d_dc.p, nullptr);
    if (rc != DOJO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    // device layout: [B][5Nc columns][12Nb rows] -> host layout [B, 12Nb, 5Nc] row-major
    std::vector<char> tmp(B * nx * ncc * w);
    HIPCHK(hipMemcpy(tmp.data(), d_dc.p, tmp.size(), hipMemcpyDeviceToHost));
    for (size_t b = 0; b < B; ++b) for (size_t c = 0; c < ncc; ++c) for (size_t r = 0; r < nx; ++r)
        std::memcpy((char*)dc + ((b * nx + r) * ncc + c) * w, tmp.data() + ((b * ncc + c) * nx + r) * w, w);
    return DOJO_OK;
}

// ---- minimal <-> maximal coordinates ----
int dojo_minimal_to_maximal_dev(DojoHandle s, const void* x, void* z, void* stream) {
    Enter enter_(s);
    if (!s || !x || !z) { g_err = "dojo_minimal_to_maximal_dev: bad argument"; return DOJO_ERR_INVALID; }
    if (s->M.has_loop) { g_err = "dojo_minimal_to_maximal_dev: the minimal coordinates of a mechanism with a kinematic loop are not those of a tree traversal (the reference sets them with exclude_ids): not supported"; return DOJO_ERR_UNSUPPORTED; }
    HIPCHK(hipSetDevice(s->device));
    { int rcj = join_groups(s, (hipStream_t)stream); if (rcj != DOJO_OK) return rcj; }   // (asynchronous steps still in flight may write / read these buffers)
    const int B = s->B, T_ = 64;
    if (s->dtype == DOJO_DTYPE_F32) hipLaunchKernelGGL((ckern::min2max_kernel<float>), dim3((B + T_ - 1) / T_), dim3(T_), 0, (hipStream_t)stream, (const dj::NodeP<double>*)s->d_nodes, s->d_order, s->M.Nb, s->M.nu, s->M.dt, B, (const float*)x, (float*)z);
    else hipLaunchKernelGGL((ckern::min2max_kernel<double>), dim3((B + T_ - 1) / T_), dim3(T_), 0, (hipStream_t)stream, (const dj::NodeP<double>*)s->d_nodes, s->d_order, s->M.Nb, s->M.nu, s->M.dt, B, (const double*)x, (double*)z);
    HIPCHK(hipGetLastError());
    return DOJO_OK;
}
int dojo_maximal_to_minimal_dev(DojoHandle s, const void* z, void* x, void* stream) {
    Enter enter_(s);
    if (!s || !x || !z) { g_err = "dojo_maximal_to_minimal_dev: bad argument"; return DOJO_ERR_INVALID; }
    if (s->M.has_loop) { g_err = "dojo_maximal_to_minimal_dev: the minimal coordinates of a mechanism with a kinematic loop are not those of a tree traversal (the reference sets them with exclude_ids): not supported"; return DOJO_ERR_UNSUPPORTED; }
    HIPCHK(hipSetDevice(s->device));
    { int rcj = join_groups(s, (hipStream_t)stream); if (rcj != DOJO_OK) return rcj; }   // (asynchronous steps still in flight may write / read these buffers)
    const long long n = (long long)s->B * s->M.Nb; const int T_ = 256;
    if (s->dtype == DOJO_DTYPE_F32) hipLaunchKernelGGL((ckern::max2min_kernel<float>), dim3((unsigned)((n + T_ - 1) / T_)), dim3(T_), 0, (hipStream_t)stream, (const dj::NodeP<double>*)s->d_nodes, s->M.Nb, s->M.nu, s->M.dt, s->B, (const float*)z, (float*)x, 2 * s->M.nu);
    else hipLaunchKernelGGL((ckern::max2min_kernel<double>), dim3((unsigned)((n + T_ - 1) / T_)), dim3(T_), 0, (hipStream_t)stream, (const dj::NodeP<double>*)s->d_nodes, s->M.Nb, s->M.nu, s->M.dt, s->B, (const double*)z, (double*)x, 2 * s->M.nu);
    HIPCHK(hipGetLastError());
    return DOJO_OK;
}
// get_state(environment) of DojoEnvironments: the minimal state of z (environments.jl:100-102, quadruped_sampling.jl:67-72)
// and, with contact_forces != 0, the clamped normal impulses of the last step behind it (ant_ars.jl:72-80).
// obs [B, 2nu (+ Nc)]
int dojo_observe_dev(DojoHandle s, const void* z, void* obs, int32_t contact_forces, void* stream) {
    Enter enter_(s);
    if (!s || !obs) { g_err = "dojo_observe_dev: bad argument"; return DOJO_ERR_INVALID; }
    if (s->M.has_loop) { g_err = "dojo_observe_dev: the minimal coordinates of a mechanism with a kinematic loop are not those of a tree traversal (the reference sets them with exclude_ids): not supported"; return DOJO_ERR_UNSUPPORTED; }
    if (!z) z = s->d_zn;                   // the state the last host-buffer / minimal-coordinate step left on the handle
    if (!z) { g_err = "dojo_observe_dev: z is NULL and the handle holds no state yet"; return DOJO_ERR_INVALID; }
    if (contact_forces && !s->have_solution) { g_err = "dojo_observe_dev: contact forces need a step on this handle"; return DOJO_ERR_INVALID; }
    if (contact_forces && s->M.contact_model == 2) { g_err = "dojo_observe: contact forces are not available for LinearContact mechanisms"; return DOJO_ERR_UNSUPPORTED; }
    HIPCHK(hipSetDevice(s->device));
    { int rcj = join_groups(s, (hipStream_t)stream); if (rcj != DOJO_OK) return rcj; }
    const int Nc = contact_forces ? s->M.Nc : 0, ld = 2 * s->M.nu + Nc;
    const long long n = (long long)s->B * s->M.Nb, nc = (long long)s->B * Nc; const int T_ = 256;
    if (s->dtype == DOJO_DTYPE_F32) {
        hipLaunchKernelGGL((ckern::max2min_kernel<float>), dim3((unsigned)((n + T_ - 1) / T_)), dim3(T_), 0, (hipStream_t)stream, (const dj::NodeP<double>*)s->d_nodes, s->M.Nb, s->M.nu, s->M.dt, s->B, (const float*)z, (float*)obs, ld);
        if (Nc) hipLaunchKernelGGL((ckern::contact_obs_kernel<float>), dim3((unsigned)((nc + T_ - 1) / T_)), dim3(T_), 0, (hipStream_t)stream, Nc, s->B, (const float*)s->d_csg, (float*)obs, ld, 2 * s->M.nu);
    } else {
        hipLaunchKernelGGL((ckern::max2min_kernel<double>), dim3((unsigned)((n + T_ - 1) / T_)), dim3(T_), 0, (hipStream_t)stream, (const dj::NodeP<double>*)s->d_nodes, s->M.Nb, s->M.nu, s->M.dt, s->B, (const double*)z, (double*)obs, ld);
        if (Nc) hipLaunchKernelGGL((ckern::contact_obs_kernel<double>), dim3((unsigned)((nc + T_ - 1) / T_)), dim3(T_), 0, (hipStream_t)stream, Nc, s->B, (const double*)s->d_csg, (double*)obs, ld, 2 * s->M.nu);
    }
    HIPCHK(hipGetLastError());
    return DOJO_OK;
}
int dojo_observe(DojoHandle s, void* obs, int32_t contact_forces) {
    Enter enter_(s);
    if (!s || !obs || !s->d_zn || !s->have_solution) { g_err = "dojo_observe: no step has been taken on this handle"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    const size_t ld = 2 * s->M.nu + (contact_forces ? s->M.Nc : 0), bytes = (size_t)s->B * ld * s->w;
    DevBuf d;
    HIPCHK(d.alloc(bytes));
    HIPCHK(hipDeviceSynchronize());                          // (not the cached stream of the last step: the caller may have destroyed it)
    int rc = dojo_observe_dev(s, s->d_zn, d.p, contact_forces, nullptr);
    if (rc != DOJO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(obs, d.p, bytes, hipMemcpyDeviceToHost));
    return DOJO_OK;
}
// step_minimal_coordinates!  src/simulation/step.jl:42-60: x -> z -> step! -> z' -> x'
int dojo_step_minimal_dev(DojoHandle s, const void* x, const void* u, void* x_next, int32_t* status, int32_t* iters, void* stream) {
    Enter enter_(s);
    if (!s || !x || !x_next) { g_err = "dojo_step_minimal_dev: bad argument"; return DOJO_ERR_INVALID; }
    if (s->M.has_loop) { g_err = "dojo_step_minimal_dev: the minimal coordinates of a mechanism with a kinematic loop are not those of a tree traversal (the reference sets them with exclude_ids): not supported"; return DOJO_ERR_UNSUPPORTED; }
    HIPCHK(hipSetDevice(s->device));
    size_t B = s->B, w = s->w, nz = 13 * s->M.Nb;
    int rc;
    if ((rc = ensure(&s->d_z, B * nz * w))) return rc;
    if ((rc = ensure(&s->d_zn, B * nz * w))) return rc;
    // (asynchronous handle: environment groups of the previous call may still read d_z / write d_zn, and the kernels that follow
    //  the step on `stream` read what its groups write -- both sides of the step are joined into the caller's stream here)
    if ((rc = join_groups(s, (hipStream_t)stream))) return rc;
    if ((rc = dojo_minimal_to_maximal_dev(s, x, s->d_z, stream))) return rc;
    if ((rc = dojo_step_dev(s, s->d_z, u, s->d_zn, status, iters, nullptr, nullptr, stream))) return rc;
    if ((rc = join_groups(s, (hipStream_t)stream))) return rc;
    s->have_grad = false;
    return dojo_maximal_to_minimal_dev(s, s->d_zn, x_next, stream);
}
// get_minimal_gradients!(mechanism, y, u; opts)  src/gradients/state.jl:183-217: steps in minimal coordinates and chains
//   max_to_min_jacobian(z+) * maximal Jacobians * min_to_max_jacobian(x).
// DOJO_GRAD_REFERENCE reproduces the reference literally: after its update_state! the "current" state is already the new
// one, so it evaluates min_to_max at maximal_to_minimal(z_new) and max_to_min at get_next_state = z_new advanced by one
// more integrator step.  DOJO_GRAD_CONSISTENT: min_to_max at x, max_to_min at z_new.
// jx [B][2nu][2nu], ju [B][2nu][nu] row-major per environment.
int dojo_minimal_gradients_dev(DojoHandle s, const void* x, const void* u, void* x_next, int32_t* status, int32_t* iters,
                               void* jx, void* ju, void* stream) {
    Enter enter_(s);
    if (!s || !x || !x_next || !jx) { g_err = "dojo_minimal_gradients_dev: bad argument"; return DOJO_ERR_INVALID; }
    if (s->M.has_loop) { g_err = "dojo_minimal_gradients_dev: the minimal coordinates of a mechanism with a kinematic loop are not those of a tree traversal (the reference sets them with exclude_ids): not supported"; return DOJO_ERR_UNSUPPORTED; }
    HIPCHK(hipSetDevice(s->device));
    const size_t B = s->B, w = s->w, Nb = s->M.Nb, nz = 13 * Nb, nx = 12 * Nb, nu = s->M.nu, nm = 2 * nu;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = ensure(&s->d_z, B * nz * w))) return rc;
    if ((rc = ensure(&s->d_zn, B * nz * w))) return rc;
    if ((rc = ensure(&s->d_dz, B * nx * nx * w))) return rc;
    if ((rc = ensure(&s->d_du, B * nx * (nu + 1) * w))) return rc;
    if ((rc = ensure(&s->d_jm, B * nx * (nm + 1) * sizeof(double)))) return rc;
    if ((rc = ensure(&s->d_jt, B * nx * (nm + 1) * sizeof(double)))) return rc;
    if ((rc = ensure(&s->d_jb, B * Nb * 12 * 24 * sizeof(double)))) return rc;
    if ((rc = join_groups(s, st))) return rc;              // (asynchronous handle: see dojo_step_minimal_dev)
    if ((rc = dojo_minimal_to_maximal_dev(s, x, s->d_z, stream))) return rc;
    if ((rc = dojo_step_dev(s, s->d_z, u, s->d_zn, status, iters, s->d_dz, s->d_du, stream))) return rc;
    if ((rc = join_groups(s, st))) return rc;
    s->have_grad = false;          // the hand-off belongs to (d_z, the CALLER's u): the host variant below re-arms the flag with its own copy of u
    if ((rc = dojo_maximal_to_minimal_dev(s, s->d_zn, x_next, stream))) return rc;
    const bool literal = s->grad_mode == DOJO_GRAD_REFERENCE;
    const void* xj = literal ? (const void*)x_next : x;                // where the min -> max Jacobian is evaluated ...
    const void* zj = literal ? (const void*)s->d_zn : (const void*)s->d_z;   // ... and the maximal state that belongs to it
    const int T1 = 64, T2 = 256;
    const dj::NodeP<double>* nodes = (const dj::NodeP<double>*)s->d_nodes;
    if (s->dtype == DOJO_DTYPE_F32) {
        hipLaunchKernelGGL((ckern::min2max_jac_kernel<float>), dim3((B + T1 - 1) / T1), dim3(T1), 0, st, nodes, s->d_order, (int)Nb, (int)nu, s->M.dt, (int)B, (const float*)xj, (const float*)zj, (double*)s->d_jm);
        hipLaunchKernelGGL((ckern::max2min_jac_kernel<float>), dim3((unsigned)((B * Nb + T2 - 1) / T2)), dim3(T2), 0, st, nodes, (int)Nb, s->M.dt, (int)B, (const float*)s->d_zn, literal ? 1 : 0, (double*)s->d_jb);
        hipLaunchKernelGGL((ckern::chain_mid_kernel<float>), dim3((unsigned)B), dim3(T2), 0, st, (int)nx, (int)nm, (int)B, (const float*)s->d_dz, (const double*)s->d_jm, (double*)s->d_jt);
        hipLaunchKernelGGL((ckern::chain_out_kernel<float>), dim3((unsigned)B), dim3(T2), 0, st, nodes, (int)Nb, (int)nu, (int)B, (const double*)s->d_jb, (const double*)s->d_jt, (const float*)s->d_du, (float*)jx, (float*)ju);
    } else {
        hipLaunchKernelGGL((ckern::min2max_jac_kernel<double>), dim3((B + T1 - 1) / T1), dim3(T1), 0, st, nodes, s->d_order, (int)Nb, (int)nu, s->M.dt, (int)B, (const double*)xj, (const double*)zj, (double*)s->d_jm);
        hipLaunchKernelGGL((ckern::max2min_jac_kernel<double>), dim3((unsigned)((B * Nb + T2 - 1) / T2)), dim3(T2), 0, st, nodes, (int)Nb, s->M.dt, (int)B, (const double*)s->d_zn, literal ? 1 : 0, (double*)s->d_jb);
        hipLaunchKernelGGL((ckern::chain_mid_kernel<double>), dim3((unsigned)B), dim3(T2), 0, st, (int)nx, (int)nm, (int)B, (const double*)s->d_dz, (const double*)s->d_jm, (double*)s->d_jt);
        hipLaunchKernelGGL((ckern::chain_out_kernel<double>), dim3((unsigned)B), dim3(T2), 0, st, nodes, (int)Nb, (int)nu, (int)B, (const double*)s->d_jb, (const double*)s->d_jt, (const double*)s->d_du, (double*)jx, (double*)ju);
    }
    HIPCHK(hipGetLastError());
    return DOJO_OK;
}
int dojo_minimal_gradients(DojoHandle s, const void* x, const void* u, void* x_next, int32_t* status, int32_t* iters, void* jx, void* ju) {
    Enter enter_(s);
    if (!s || !x || !x_next || !jx) { g_err = "dojo_minimal_gradients: bad argument"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    size_t B = s->B, w = s->w, nu = s->M.nu, nm = 2 * nu;
    int rc;
    if ((rc = ensure(&s->d_x, B * (nm + 1) * w))) return rc;
    if ((rc = ensure(&s->d_xn, B * (nm + 1) * w))) return rc;
    if ((rc = ensure(&s->d_u, B * (nu + 1) * w))) return rc;
    if ((rc = ensure((void**)&s->d_status, B * sizeof(int)))) return rc;
    if ((rc = ensure((void**)&s->d_iters, B * sizeof(int)))) return rc;
    DevBuf bjx, bju;
    HIPCHK(bjx.alloc(B * nm * nm * w)); HIPCHK(bju.alloc(B * nm * (nu + 1) * w));
    void *d_jx = bjx.p, *d_ju = bju.p;
    HIPCHK(hipMemcpy(s->d_x, x, B * nm * w, hipMemcpyHostToDevice));
    if (u && nu) HIPCHK(hipMemcpy(s->d_u, u, B * nu * w, hipMemcpyHostToDevice));
    rc = dojo_minimal_gradients_dev(s, s->d_x, (u && nu) ? s->d_u : nullptr, s->d_xn, s->d_status, s->d_iters, d_jx, d_ju, nullptr);
    if (rc == DOJO_OK) { s->have_grad = true; s->have_u = (u && nu); }   // d_z, d_u, d_dz, d_du and the hand-off all belong to this step
    if (rc == DOJO_OK) {
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(x_next, s->d_xn, B * nm * w, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(jx, d_jx, B * nm * nm * w, hipMemcpyDeviceToHost));
        if (ju && nu) HIPCHK(hipMemcpy(ju, d_ju, B * nm * nu * w, hipMemcpyDeviceToHost));
        if (status) HIPCHK(hipMemcpy(status, s->d_status, B * sizeof(int), hipMemcpyDeviceToHost));
        if (iters) HIPCHK(hipMemcpy(iters, s->d_iters, B * sizeof(int), hipMemcpyDeviceToHost));
    }
    return rc;
}

static int coords_host(DojoHandle s, const void* in, void* out, size_t n_in, size_t n_out, bool to_max) {
    HIPCHK(hipSetDevice(s->device));
    size_t B = s->B, w = s->w;
    int rc;
    if ((rc = ensure(&s->d_x, B * (2 * s->M.nu + 1) * w))) return rc;
    if ((rc = ensure(&s->d_cz, B * 13 * s->M.Nb * w))) return rc;       // not d_z: that is the state dojo_contact_gradients re-linearizes at
    void* din = to_max ? s->d_x : s->d_cz; void* dout = to_max ? s->d_cz : s->d_x;
    HIPCHK(hipMemcpy(din, in, B * n_in * w, hipMemcpyHostToDevice));
    rc = to_max ? dojo_minimal_to_maximal_dev(s, din, dout, nullptr) : dojo_maximal_to_minimal_dev(s, din, dout, nullptr);
    if (rc != DOJO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dout, B * n_out * w, hipMemcpyDeviceToHost));
    return DOJO_OK;
}
int dojo_minimal_to_maximal(DojoHandle s, const void* x, void* z) {
    Enter enter_(s);
    if (!s || !x || !z) { g_err = "dojo_minimal_to_maximal: bad argument"; return DOJO_ERR_INVALID; }
    return coords_host(s, x, z, 2 * s->M.nu, 13 * s->M.Nb, true);
}
int dojo_maximal_to_minimal(DojoHandle s, const void* z, void* x) {
    Enter enter_(s);
    if (!s || !x || !z) { g_err = "dojo_maximal_to_minimal: bad argument"; return DOJO_ERR_INVALID; }
    return coords_host(s, z, x, 13 * s->M.Nb, 2 * s->M.nu, false);
}
// The vector the reference's step! literally RETURNS (src/simulation/step.jl:28: get_next_state after update_state!, SURVEY.md
// §8a Q1): the internal state z_next = (x3, v25, q3, ω25) advanced once more with its own velocities, (x3 + Δt v25, v25,
// q3 ⊗ ξ(ω25), ω25).  dojo_step returns the internal state; a caller that wants the literal return applies this to it.
int dojo_next_state_dev(DojoHandle s, const void* z, void* z_out, void* stream) {
    Enter enter_(s);
    if (!s || !z || !z_out) { g_err = "dojo_next_state_dev: bad argument"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    { int rcj = join_groups(s, (hipStream_t)stream); if (rcj != DOJO_OK) return rcj; }   // (asynchronous steps still in flight may write / read these buffers)
    const long long n = (long long)s->B * s->M.Nb; const int T_ = 256;
    if (s->dtype == DOJO_DTYPE_F32) hipLaunchKernelGGL((ckern::next_state_kernel<float>), dim3((unsigned)((n + T_ - 1) / T_)), dim3(T_), 0, (hipStream_t)stream, s->M.Nb, s->M.dt, s->B, (const float*)z, (float*)z_out);
    else hipLaunchKernelGGL((ckern::next_state_kernel<double>), dim3((unsigned)((n + T_ - 1) / T_)), dim3(T_), 0, (hipStream_t)stream, s->M.Nb, s->M.dt, s->B, (const double*)z, (double*)z_out);
    HIPCHK(hipGetLastError());
    return DOJO_OK;
}
int dojo_next_state(DojoHandle s, const void* z, void* z_out) {
    Enter enter_(s);
    if (!s || !z || !z_out) { g_err = "dojo_next_state: bad argument"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    const size_t bytes = (size_t)s->B * 13 * s->M.Nb * s->w;
    DevBuf a, b;
    HIPCHK(a.alloc(bytes)); HIPCHK(b.alloc(bytes));
    HIPCHK(hipMemcpy(a.p, z, bytes, hipMemcpyHostToDevice));
    int rc = dojo_next_state_dev(s, a.p, b.p, nullptr);
    if (rc != DOJO_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(z_out, b.p, bytes, hipMemcpyDeviceToHost));
    return DOJO_OK;
}

int dojo_step_minimal(DojoHandle s, const void* x, const void* u, void* x_next, int32_t* status, int32_t* iters) {
    Enter enter_(s);
    if (!s || !x || !x_next) { g_err = "dojo_step_minimal: bad argument"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    size_t B = s->B, w = s->w, nm = 2 * s->M.nu, nu = s->M.nu;
    int rc;
    if ((rc = ensure(&s->d_x, B * (nm + 1) * w))) return rc;
    if ((rc = ensure(&s->d_xn, B * (nm + 1) * w))) return rc;
    if ((rc = ensure(&s->d_u, B * (nu + 1) * w))) return rc;
    if ((rc = ensure((void**)&s->d_status, B * sizeof(int)))) return rc;
    if ((rc = ensure((void**)&s->d_iters, B * sizeof(int)))) return rc;
    HIPCHK(hipMemcpy(s->d_x, x, B * nm * w, hipMemcpyHostToDevice));
    if (u && nu) HIPCHK(hipMemcpy(s->d_u, u, B * nu * w, hipMemcpyHostToDevice));
    rc = dojo_step_minimal_dev(s, s->d_x, (u && nu) ? s->d_u : nullptr, s->d_xn, s->d_status, s->d_iters, nullptr);
    if (rc != DOJO_OK) return rc;
    s->have_grad = false;                                   // d_z / d_u now hold this (forward-only) step's inputs
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(x_next, s->d_xn, B * nm * w, hipMemcpyDeviceToHost));
    if (status) HIPCHK(hipMemcpy(status, s->d_status, B * sizeof(int), hipMemcpyDeviceToHost));
    if (iters) HIPCHK(hipMemcpy(iters, s->d_iters, B * sizeof(int), hipMemcpyDeviceToHost));
    return DOJO_OK;
}

int dojo_last_kernel_ms(DojoHandle s, double* ms) {
    Enter enter_(s);
    double a = 0, b = 0;
    int rc = dojo_last_kernel_times(s, &a, &b);
    if (rc == DOJO_OK && ms) *ms = a + b;
    return rc;
}

int dojo_last_kernel_times(DojoHandle s, double* step_ms, double* ift_ms) {
    Enter enter_(s);
    if (!s || !step_ms || !ift_ms || s->last_slot < 0) { g_err = "dojo_last_kernel_times: nothing was launched"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    DojoSim::Ev3& e = s->ring[s->last_slot];
    HIPCHK(hipEventSynchronize(e.b));
    float t = 0, t1 = 0;
    HIPCHK(hipEventElapsedTime(&t, e.a, e.b));
    if (e.has_mid) { HIPCHK(hipEventElapsedTime(&t1, e.a, e.m)); *step_ms = (double)t1; *ift_ms = (double)t - (double)t1; }
    else { *step_ms = (double)t / e.n; *ift_ms = 0.0; }
    return DOJO_OK;
}

int dojo_kernel_time_totals(DojoHandle s, double* step_ms, double* ift_ms, int64_t* launches, int32_t reset) {
    Enter enter_(s);
    if (!s) { g_err = "dojo_kernel_time_totals: bad argument"; return DOJO_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    for (auto& e : s->ring) { int rc = drain_slot(s, e); if (rc != DOJO_OK) return rc; }
    if (step_ms) *step_ms = s->acc_step_ms;
    if (ift_ms) *ift_ms = s->acc_ift_ms;
    if (launches) *launches = (int64_t)s->acc_n;
    if (reset) { s->acc_step_ms = s->acc_ift_ms = 0; s->acc_n = 0; }
    return DOJO_OK;
}

} // extern "C"
