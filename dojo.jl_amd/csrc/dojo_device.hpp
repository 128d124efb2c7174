// dojo_device.hpp -- the per-lane algorithm of the batched contact-implicit step.
//
// Mapping.  A supernode is a body together with its parent joint and the contacts attached to it.
//   quad mapping (<= 32 bodies): four lanes per supernode, lane role q owns rows 3q..3q+2 of its 12x12 system; one
//     environment per wavefront (or per two wavefronts, 17..32 bodies; 16 tiny environments per wavefront).  Factors and
//     working vectors live in VGPRs; what the four lanes would hold identically (constants, solver state, cold
//     linearization data) lives once per supernode in LDS; quads exchange along tree edges through an LDS mailbox,
//     inside a quad through DPP.
//   lane mapping (33..64 bodies): one lane per supernode, exchanges through wave shuffles.
//   children -> parent : impulses on the parent body, Schur complements (6x6 + 6)
//   parent -> children : configuration / velocity of the parent body, Newton step of the parent
// HBM is touched for the algorithmic bytes (state in, state/solution/gradient out) and one hand-off between the
// step kernel and the IFT kernels.
//
// The same source runs under tests/emu (threads + barriers implement the Wave interface) so the
// device algorithm is checked against the CPU oracle without a GPU.
//
// Reference functions restated here (all fused into the lane program):
//   src/simulation/step.jl:11-30, src/mechanism/set.jl:10-53, src/bodies/set.jl:1-36
//   src/integrators/constraint.jl:1-66          body residual d and Jacobian D
//   src/joints/joint.jl, limits.jl, constraints.jl:114-299, translational/ & rotational/*.jl
//   src/contacts/nonlinear.jl:50-97, contact.jl:37-138, velocity.jl, collisions/sphere_halfspace.jl
//   src/solver/*.jl (mehrotra!, line searches, centering, correction, violations, initialization)
//   src/gradients/state.jl:78-126, src/gradients/data.jl (data Jacobian blocks)
#pragma once
#include "dojo_math.hpp"
#include <type_traits>
#include <cstddef>
#ifdef DJ_DEBUG
#include <cstdio>
#include <cstdlib>
#endif

#ifndef DJ_TSD
#define DJ_TSD 0           // 1: the kernels evaluate translational springs / dampers (KernelArgs::tsd); builds of their own
#endif
#ifndef DJ_REFINE
#define DJ_REFINE 1        // quad mapping: iterative refinement of the Newton / IFT solves against the UNCONDENSED system for environments whose
                           // cone variables have become stiff (max γ/s > Globals::refine_w): the condensed body blocks D + (γ/s) g cᵀ lose the
                           // low bits of D there, which the dense pivoted LU of the reference-level oracle does not
#endif
#ifndef DJ_REFINE_STEPS
#define DJ_REFINE_STEPS 2  // rounds per solve (the contraction per round is ~ε·γ/s·cond of the un-stiff part; nearly redundant joint / contact rows need two)
#endif
#ifndef DJ_LINEAR
#define DJ_LINEAR 0         // 1: builds for LinearContact mechanisms (src/contacts/linear.jl: six cone pairs [γ ψ β1..β4] per contact, all on the positive
                            // orthant; forward only, like the reference) -- the step kernel alone, no refinement, no IFT
#endif
#if DJ_LINEAR
#undef DJ_REFINE
#define DJ_REFINE 0
#endif

#ifndef DJ_SS
#define DJ_SS 0             // 1: builds that carry body-body contacts (SphereSphereCollision between a body and its tree child, src/contacts/collisions/
                            // sphere_sphere.jl; forward only): ContactCold holds the parent-side rows as well
#endif

#ifndef DJ_MLIM
#define DJ_MLIM 0           // 1: builds for mechanisms whose joints carry limits on SEVERAL coordinates (all free coordinates of a half: three
                            // rotation-vector limits on a Spherical joint, two on a Planar joint's translation ...) or on BOTH halves
                            // (src/joints/limits.jl:1-61 in general; the other builds know one limited coordinate per joint).  Lane mapping
                            // only (one lane per supernode); every limit of such a mechanism runs through this path (KernelArgs::mlim).
#endif

#ifndef DJ_ROWS
#define DJ_ROWS 2           // the factorization's level passes: 0 = quad layout only (factorize_quad), 1 = row layout wherever the kernel has the staging
                            // areas (experiments: run such a build with DOJO_ROWS=1, it ignores Globals::rows = 0), 2 = both, Globals::rows decides at run time
#endif

#ifndef DJ_CUT
#define DJ_CUT 0            // 1: builds for mechanisms with KINEMATIC LOOPS (a body with more than one parent joint: src/solver/linear_system.jl:4-5
                            // `cyclic_children`, the four-bar of test/behaviors.jl:57-81).  The loop-closing ("cut") joints stay out of the tree
                            // elimination and come back through a low-rank correction of every solve (LaneProgram::cut_*).  Lane mapping only.
#endif

namespace dj {

constexpr int NCUT = 2;                  // loop-closing joints per mechanism in the DJ_CUT builds
// per-environment workspace of the cut elements in global memory (KernelArgs::cutws): M_c [NCUT][18][18], H [12 NCUT][12 NCUT], the LU of the small
// system [18 NCUT][18 NCUT] and its row exchanges [18 NCUT] -- written by one lane of the environment, read by all of them
constexpr int CUTWS_H = NCUT * 324, CUTWS_LU = CUTWS_H + 144 * NCUT * NCUT, CUTWS_PIV = CUTWS_LU + 324 * NCUT * NCUT, CUTWS = CUTWS_PIV + 18 * NCUT;
constexpr int NLM = DJ_MLIM ? 6 : 1;     // limited coordinates per joint in the DJ_MLIM builds: up to 3 translational + 3 rotational
// per supernode (DJ_MLIM builds): how many coordinates of the parent joint's translational / rotational half are limited (0 or all of the half's
// free ones) and the bounds, translational coordinates first.  Coordinate m keeps its Δκ row in the padded multiplier slot of its own free
// coordinate: 6 + nl_t + m for a translational one, 9 + nl_r + (m − nt) for a rotational one.
template <class T> struct MLimP { int nt, nr; T lo[6], hi[6]; };

constexpr int MAXCH = 4;          // children per body supported by the lane program
constexpr int LU_PER_LANE = 112;  // values per lane of the IFT kernel's LU-form factors between its phases (KernelArgs::lu; layout: LaneProgram::LU_LM .. LU_DI)
constexpr bool kLinear = DJ_LINEAR != 0;
constexpr int NCV = kLinear ? 6 : 4;      // cone variable pairs (s, γ) per contact: NonlinearContact 4 (ImpactContact uses the first), LinearContact 6
constexpr double REG = 1e-10;     // src/Dojo.jl:4

#define DJ_STATUS_SUCCESS 0
#define DJ_STATUS_FAILED 1
#define DJ_STATUS_EXCESSIVE_W 2
#ifndef DJ_TRACK_GROWTH
#define DJ_TRACK_GROWTH 0          // 1: the Gauss-Jordan passes record their largest |multiplier| (dojo_get_diagnostics, tools/hunt_parity.py); ~2 % of the step kernel
#endif
#define DJ_STATUS_DEFERRED 3      // internal: the environment's cones became stiff; the refining kernels re-solve it (never reaches the caller)
#define DJ_STATUS_CONTINUE 4      // internal: the solve reached Globals::iter_cap unfinished; the continuation kernel takes it from there (never reaches the caller)

template <class T>
struct Globals {
    T dt, idt2 /* 1/dt² */, input_scaling, g[3];
    T rtol, btol, undercut, no_progress_undercut;
    T refine_w;                  // refine the linear solves of an environment once max γ/s over its cones exceeds this (inf: never, 0: always)
    int max_iter, max_ls, no_progress_max;
    int iter_cap;                // > 0: the step kernel hands a solve that is unfinished after this many Newton iterations to the continuation kernel
                                 // (DJ_STATUS_CONTINUE; KernelArgs::resume / cont_list), which runs the rest with its line-search trials side by side
    int Nb, Nc, S, nu, n_joint_imp, maxch, maxlevel, grad_mode;
    int contact_model;           // 0: NonlinearContact; 1: ImpactContact = the same rows without the friction block (γ2:4, s2:4 pinned)
    unsigned char maxch_lev[64]; // largest number of children among the supernodes of each level (bounds the level sweeps' gathers)
    // ... the same, four bits per level: the device reads THIS (a uniform 64-bit scalar load and a shift).  A byte table indexed by the level
    // becomes a per-lane global load with a full wait in front of every level pass of every sweep (round 6: 12 memory round trips per Newton iteration)
    unsigned long long maxch_pack[4];
    DJ_HD int maxch_at(int lev) const { return (int)((maxch_pack[lev >> 4] >> (4 * (lev & 15))) & 15ull); }
    // Row layout of the factorization's level passes (LaneProgram::factorize_rows; single-wavefront quad mapping, Wave::kRows): pass t
    // factorizes up to four supernodes of one tree level, each on a 16-lane row of the wavefront.  rp_slot[t][g] = the supernode slot
    // (lane >> 2) group g serves in pass t, -1 = none; passes run leaves -> root.  rows = 0: the quad-layout passes (factorize_quad).
    // (one 32-bit word per entry, a signed byte per group: a uniform index then is a scalar load -- byte tables become per-lane global loads)
    int rows;
    int rp_slot4[16];                     // byte g: rp_slot[t][g]
    int rp_lev[16];                       // the level of pass t, and in bits 8.. the largest child count of that level
    int rp_child4[16][MAXCH];             // byte g of [t][ci]: the slot of child ci of group g's supernode (-1: none), in NodeP::child order
    // Two-wavefront workgroups (17..32 bodies): every wavefront runs the passes of ITS OWN sixteen supernode slots, four per pass, both in step (pass t
    // = the same tree level on both; a wavefront without a supernode in a pass idles through it).  The tables above are wavefront 0's, these wavefront 1's.
    int rp_slot4_w1[16];
    int rp_child4_w1[16][MAXCH];
    DJ_HD static int rp_byte(int w, int g) { return (int)(signed char)((unsigned)w >> (8 * g)); }
};

// per-supernode constants (body k, its parent joint, its contacts)
template <class T>
struct NodeP {
    int parent, level, nchild, child[MAXCH];
    int ncontact, contact[8];
    int nl_t, nl_r, nlim_r, spring_on, damper_on;
    int u_off, nu_t, nu_r, imp_off, n_imp;
    T m, J[9];
    T Ct[9], Cr[9], At[9], Ar[9];          // constraint / nullspace masks, zero-padded rows
    T pa[3], pb[3], qoff[4];
    T spring_r, damper_r, spring_off_r[3], lim_lo, lim_hi;
};
template <class T>
struct ContactP { T n[3], t[6], o[3], off[3], r, mu; T r2, o2[3]; int kind, pbody, cbody; };   // kind 1: SphereSphereCollision: (o, r) = (origin_parent, radius_parent), (o2, r2) = the child's; pbody / cbody: its parent body and its child body (the owner; half-space: -1 / the body)

// the node constants of a workgroup's supernode slots in LDS (lock-step quad mappings: StepLds, DJ_LANE_SETUP)
template <class T>
struct NodeSlot { NodeP<T> P; char pad_[(sizeof(NodeP<T>) / 8) % 2 == 0 ? 8 : 16]; };                    // odd stride in 8-byte words

// ------------------------------------------------------------------------------------------------
// Lane-local dynamic state
// ------------------------------------------------------------------------------------------------
template <class T, int MAXC>
struct Lane {
    // configuration (constant during the solve)
    T x2[3], q2[4], xa2[3], qa2[4];
    T dconst[6];                 // velocity-independent part of the body residual
    T v15[3], w15[3];            // midpoint velocities of the previous interval (data of the IFT)
    // candidate solution (vsol[2]/ωsol[2], impulses[2], impulses_dual[2] in the reference); the accepted
    // iterate lives in a SolSnap during the line search
    T v[3], w[3];
    T lam[6];                    // equality multipliers: 3 translational slots, 3 rotational slots
    T ls[2], lg[2];              // rotational joint limit: s = (s_up, s_lo), γ = (γ_up, γ_lo)
    T cs[MAXC][NCV], cg[MAXC][NCV];
#if DJ_MLIM
    T mls[NLM][2], mlg[NLM][2];  // limits on several coordinates: (s_up, s_lo), (γ_up, γ_lo) per limited coordinate
#endif
};

// factor data of one supernode, kept between the two solves of a Mehrotra iteration and
// re-used by the IFT back-solves
template <class T, class TL, int MAXC, bool QUAD>
struct Factors {
    // linear-algebra part, in the factorization precision TL (fp32 in the mixed "f32" mode)
    // lane = supernode mapping: the whole factor lives on one lane
    TL Sinv[QUAD ? 1 : 144];     // inverse of the 12x12 supernode matrix [[D_b, P_b],[G_b, REG]]
    TL W[QUAD ? 1 : 72];         // L S⁻¹   (6x12): rhs of the parent  -= W r_k
    TL Z[QUAD ? 1 : 72];         // S⁻¹ U   (12x6): Δ_k = S⁻¹ r_k − Z Δv_parent
    // quad mapping: lane q of a supernode keeps rows 3q..3q+2 of S⁻¹ and of U, columns 3q..3q+2 of L
    TL Sq[QUAD ? 3 : 1][12], Uq[QUAD ? 3 : 1][6], Lq[6][QUAD ? 3 : 1];
    // condensation pieces, in the state precision T
    T th_a[3], th_b[3], t_a[6], t_b[6];
#if DJ_TSD
    T thv_a[3], thv_b[3];        // translational limit: the linear-velocity part of ∂θ/∂(velocities)
#endif
};

// what a solve stopped at the iteration cap carries over to its continuation besides the iterate itself (which travels in the
// step -> IFT hand-off record, KernelArgs::sol): the scalars of mehrotra!'s loop (src/solver/mehrotra.jl:17-21, 51-60)
template <class T>
struct SolveCarry { T undercut, rvio, bvio; int n, no_progress, excessive; };
constexpr int CARRY_PER_ENV = 8;   // doubles per environment of KernelArgs::resume
constexpr int CARRY_MARK = 7;      // ... of which this one says whether the environment is on the continuation list

template <class T, int MAXC>
struct SolSnap { T v[3], w[3], lam[6], ls[2], lg[2], cs[MAXC][NCV], cg[MAXC][NCV];     // solution variables at the start of a line search
#if DJ_MLIM
    T mls[NLM][2], mlg[NLM][2];
#endif
#if DJ_CUT
    T clam[NCUT][6];
#endif
};

template <class T, int MAXC>
struct Step {                    // Newton step of this lane's unknowns
    T dv[3], dw[3], dlam[6], dls[2], dlg[2], dcs[MAXC][NCV], dcg[MAXC][NCV];
#if DJ_MLIM
    T dmls[NLM][2], dmlg[NLM][2];
#endif
#if DJ_CUT
    T dclam[NCUT][6];
#endif
};


// ------------------------------------------------------------------------------------------------
// Containers of the un-factored supernode blocks.  The assembly code only *adds* entries through
// these; FullBlocks keeps everything on one lane (lane = supernode mapping), QuadBlocks keeps the
// three rows (columns for L) that belong to the lane's role q in the 4-lanes-per-supernode mapping:
//   q = 0: body rows of v, 1: body rows of ω, 2: translational multiplier rows, 3: rotational ones.
// All (r, c) are compile-time constants after unrolling, so `r / 3 == q` is a predicate on a
// register and every array element has a static index (stays in VGPRs).
//   S[12x12] rows 0:6 = body rows (D_b | P_b), rows 6:12 = joint rows (G_b | REG)
//   U[12x6]  = [M_ba; G_a]   rows of this supernode wrt the parent's velocity
//   L[6x12]  = [M_ab | P_a]  rows of the parent body wrt this supernode's unknowns
//   D[6x6]   = contribution of this joint to the parent's diagonal block
// ------------------------------------------------------------------------------------------------
template <class T>
struct FullBlocks {
    T S[144], U[72], L[72], D[36];
    DJ_HD void zero() { for (int i = 0; i < 144; ++i) S[i] = T(0); for (int i = 0; i < 72; ++i) { U[i] = T(0); L[i] = T(0); } for (int i = 0; i < 36; ++i) D[i] = T(0); }
    DJ_HD void addS(int r, int c, T v) { S[12 * r + c] += v; }
    DJ_HD void addU(int r, int c, T v) { U[6 * r + c] += v; }
    DJ_HD void addL(int r, int c, T v) { L[12 * r + c] += v; }
    DJ_HD void addD(int r, int c, T v) { D[6 * r + c] += v; }
};
struct NullBlocks {
    DJ_HD void zero() {}
    template <class V> DJ_HD void addS(int, int, V) {}
    template <class V> DJ_HD void addU(int, int, V) {}
    template <class V> DJ_HD void addL(int, int, V) {}
    template <class V> DJ_HD void addD(int, int, V) {}
};
// Quad mapping: the lane's rows are assembled straight into the factor storage (Factors::Sq/Uq/Lq),
// so the assembled blocks and the factors never coexist in registers.
template <class T>
struct QuadBlocks {
    T (&S)[3][12]; T (&U)[3][6]; T (&L)[6][3];
    T D[3][6];
    int q;
    DJ_HD QuadBlocks(T (&s)[3][12], T (&u)[3][6], T (&l)[6][3], int q_) : S(s), U(u), L(l), q(q_) {}
    DJ_HD void zero() {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 12; ++j) S[i][j] = T(0);
#pragma unroll
            for (int j = 0; j < 6; ++j) { U[i][j] = T(0); D[i][j] = T(0); L[j][i] = T(0); }
        }
    }
    template <class V> DJ_HD void addS(int r, int c, V v) { if (r / 3 == q) S[r % 3][c] += T(v); }
    template <class V> DJ_HD void addU(int r, int c, V v) { if (r / 3 == q) U[r % 3][c] += T(v); }
    template <class V> DJ_HD void addL(int r, int c, V v) { if (c / 3 == q) L[r][c % 3] += T(v); }
    template <class V> DJ_HD void addD(int r, int c, V v) { if (r / 3 == q) D[r % 3][c] += T(v); }
};

// kinematic quantities of a body at the candidate velocity
template <class T>
struct Kin { T x3[3], q3[4], R3[9], Phi[9], Xi[9], c; };   // Xi = ∂φ3/∂φ2 = R(ξ)ᵀ

template <class T> DJ_HD void kin_of(Kin<T>& k, const T* x2, const T* q2, const T* v, const T* w, T dt) {
    T xi[4];
    qstep(xi, w, dt, &k.c);
    qmul(k.q3, q2, xi);
    for (int i = 0; i < 3; ++i) k.x3[i] = x2[i] + dt * v[i];
    qrot(k.R3, k.q3);
    phi_of(k.Phi, w, k.c, dt);
    T Rx[9];
    qrot(Rx, xi);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) k.Xi[3 * i + j] = Rx[3 * j + i];
}

// rᵀ L(q) and rᵀ R(q) for 4-vectors (rows of quaternion product Jacobians)
template <class T> DJ_HD void rowL(T* o, const T* r, const T* q) {
    T s = q[0], x = q[1], y = q[2], z = q[3];
    o[0] = r[0] * s + r[1] * x + r[2] * y + r[3] * z;
    o[1] = -r[0] * x + r[1] * s + r[2] * z - r[3] * y;
    o[2] = -r[0] * y - r[1] * z + r[2] * s + r[3] * x;
    o[3] = -r[0] * z + r[1] * y - r[2] * x + r[3] * s;
}
template <class T> DJ_HD void rowR(T* o, const T* r, const T* q) {
    T s = q[0], x = q[1], y = q[2], z = q[3];
    o[0] = r[0] * s + r[1] * x + r[2] * y + r[3] * z;
    o[1] = -r[0] * x + r[1] * s - r[2] * z + r[3] * y;
    o[2] = -r[0] * y + r[1] * z + r[2] * s - r[3] * x;
    o[3] = -r[0] * z - r[1] * y + r[2] * x + r[3] * s;
}

// ------------------------------------------------------------------------------------------------
// joint impulse maps at the CURRENT configuration (x2,q2): impulse_map / impulse_transform,
// src/joints/joint.jl:67-93, src/joints/impulses.jl:4-8.  `JointCfg` caches what they need.
// ------------------------------------------------------------------------------------------------
template <class T>
struct JointCfg {
    T Ra[9], Rba[9];             // R(qa2), R(qb2)ᵀ R(qa2)
    T u2[3];                     // e2 + pa   (translational displacement at config 2, parent frame)
    T qr[4];                     // qoff⁻¹ ⊗ qa2⁻¹ ⊗ qb2
    T Roff[9];
};
template <class T> DJ_HD void joint_cfg(JointCfg<T>& c, const NodeP<T>& P, const T* xa, const T* qa, const T* xb, const T* qb) {
    T Rb[9];
    qrot(c.Ra, qa); qrot(Rb, qb); qrot(c.Roff, P.qoff);
    m3tmul(c.Rba, Rb, c.Ra);
    T t[3], wv[3];
    m3vec(t, Rb, P.pb);
    for (int i = 0; i < 3; ++i) wv[i] = xb[i] + t[i] - xa[i];
    m3tvec(c.u2, c.Ra, wv);                         // = e2 + pa
    T qab[4];
    qcmul(qab, qa, qb);
    qcmul(c.qr, P.qoff, qab);
}
// Data that is read rarely: the joint frames at the current configuration (Cold) and the contact Jacobian rows of the
// last linearization (ContactCold).  Quad mapping on the GPU: once per supernode / per contact in LDS.
template <class T, int MAXC>
struct Cold {
    JointCfg<T> cfg;
    T pad_[((sizeof(JointCfg<T>) / sizeof(T)) % 2 == 0) ? 1 : 2];       // odd stride in 8-byte words
};
template <class T>
struct ContactCold { T C134[18], G134[18];
#if DJ_SS
                     T Cp134[18], Gp134[18];      // body-body contact: the rows of the contact's PARENT body (the owner's tree parent); zero otherwise
#endif
};

// IFT data-Jacobian blocks of one supernode (datamat = −∂residual/∂θ, src/gradients/data.jl), stored
// once per supernode (per quad) in the precision of the ABI buffers
template <class TB, int MAXC>
struct GradBlocks {
    TB OwnB[6][12], OwnJ[6][6], ParB[6][6], ParJ[6][6], UpOwn[6][6], UpPar[6][6], sl_own[6], sl_par[6], UB[6][6], UA[6][6];
    TB Cc[MAXC][4][6];
    TB pad_[1];
};
// Quad mapping: the IFT right-hand sides of one supernode, laid out per role (lane q owns rows
// 3q..3q+2) with the cone condensation already folded in, so that the column sweeps load their
// rows with one address computation and no divergence.  Three flat arrays:
//   a       the body-row blocks, in the ABI type (an fp32 rounding of these moves the gradient by < 1e-7 relative)
//   jd      the joint rows and the joint-limit slack rows, in double: they are constraint rows -- their right-hand sides reach
//           the velocities amplified by 1/Δt and more, and rounded to fp32 they cost up to 1.4e-5 of the gradient (measured on
//           the fp32-ABI Ant batch, tools/hunt_f32.py; the body-row blocks: no effect)
//   own_cfg the owner's body rows of the configuration columns with the cone condensation folded in, in double (the folded
//           terms ~γ/s cancel against the stiff rows of the factors)
template <class TB>
struct alignas(8) QuadRhs {
    enum { ROWNV = 0,     // [2 roles][3 rows][6]  owner body rows, velocity columns (v15 ω15)
           RPARB = 36,    // [2][3][6]  body rows of this supernode for the PARENT's configuration columns
           UOWN = 72,     // [2][3][6]  what the owner's configuration columns put on the parent's body rows
           UPAR = 108,    // [2][3][6]  the same for the parent's own configuration columns
           UB = 144,      // [2][3][6]  control columns: child body rows
           UA = 180,      // [2][3][6]  control columns: parent body rows
           SIZE = 216 };
    enum { ROWNJ = 0,     // [2 roles][3][6]  owner joint rows (roles 2, 3), configuration columns (x2 φ2)
           RPARJ = 36,    // [2][3][6]  joint rows of this supernode for the PARENT's configuration columns
           SLO = 72,      // [6]  joint-limit slack rows, owner's configuration columns   (enter the Δκ row as σ·slack in the sweep)
           SLP = 78,      // [6]  ... parent's configuration columns
           JSIZE = 84 };
    TB a[SIZE];
    double jd[JSIZE];
    double own_cfg[36];                                            // [2 roles][3][6]
    double pad_[(SIZE * sizeof(TB) / 8 + JSIZE + 36) % 2 == 0 ? 1 : 2];     // odd stride in 8-byte words
};
// Quad mapping: the right-hand sides of the contact-data columns (get_contact_gradients, src/gradients/contact.jl):
// per contact of the supernode, per body-row role, 3 rows x 6 columns (5 used: friction, radius, origin(3)); cone
// condensation folded in, hence double.
template <int MAXC>
struct ConRhs { double a[MAXC * 36]; double pad_[(MAXC * 36) % 2 == 0 ? 1 : 2]; };
// T_a p and T_b p for the translational half: 6-vectors (force in world frame, torque in body frame)
template <class T> DJ_HD void tra_impulse(T* ia, T* ib, const JointCfg<T>& c, const NodeP<T>& P, const T* p) {
    T F[3], t[3];
    m3vec(F, c.Ra, p);
    v3cross(t, c.u2, p);
    ia[0] = -F[0]; ia[1] = -F[1]; ia[2] = -F[2]; ia[3] = -t[0]; ia[4] = -t[1]; ia[5] = -t[2];
    T Fb[3];
    m3vec(Fb, c.Rba, p);
    v3cross(t, P.pb, Fb);
    ib[0] = F[0]; ib[1] = F[1]; ib[2] = F[2]; ib[3] = t[0]; ib[4] = t[1]; ib[5] = t[2];
}
template <class T> DJ_HD void rot_impulse(T* ia, T* ib, const JointCfg<T>& c, const T* p) {
    T vp[3], a[3], b[3];
    v3cross(vp, &c.qr[1], p);
    for (int i = 0; i < 3; ++i) { a[i] = c.qr[0] * p[i] + vp[i]; b[i] = T(0.5) * (c.qr[0] * p[i] - vp[i]); }
    T ra[3];
    m3vec(ra, c.Roff, a);
    ia[0] = ia[1] = ia[2] = T(0); ia[3] = T(-0.5) * ra[0]; ia[4] = T(-0.5) * ra[1]; ia[5] = T(-0.5) * ra[2];
    ib[0] = ib[1] = ib[2] = T(0); ib[3] = b[0]; ib[4] = b[1]; ib[5] = b[2];
}

// ------------------------------------------------------------------------------------------------
// everything a lane computes about its parent joint at the candidate point
// ------------------------------------------------------------------------------------------------
template <class T>
struct JointEval {
    T g[6];                      // equality residual (3 tra slots, 3 rot slots; padded slots = 0)
    T theta;                     // limited rotational coordinate
    T imp_a[6], imp_b[6];        // joint impulse + damper on parent / child body (to be subtracted from d)
    // Jacobian pieces (only when JAC).  The 6x6 blocks go straight into the supernode arrays:
    //   S[12x12] rows 0:6 = body rows (D_b | P_b), rows 6:12 = joint rows (G_b | REG)
    //   U[12x6]  = [M_ba; G_a]   (rows of this supernode wrt the parent's velocity)
    //   L[6x12]  = [M_ab | P_a]  (rows of the parent body wrt this supernode's unknowns)
    //   Dup[6x6] = contribution of this joint to the parent's diagonal block
    T th_a[3], th_b[3];          // ∂θ/∂ω_a, ∂θ/∂ω_b
    T GaX[18], GaP[18], GbX[18], GbP[18];   // raw rows ∂g/∂x3, ∂g/∂φ3 (6 slots x 3) of parent / child
    T thp_a[3], thp_b[3];        // raw ∂θ/∂φ3
#if DJ_TSD
    T thv_a[3], thv_b[3];        // translational limit: ∂θ/∂v_a, ∂θ/∂v_b
    T thx_a[3], thx_b[3];        // raw ∂θ/∂x3
#endif
    T t_a[6], t_b[6];            // impulse_transform · Arᵀ (limit impulse direction)
};

// MODE 0: residual pieces only; 1: + Jacobian blocks into S/U/L/Dup; 2: + raw rows for the data Jacobian
template <int MODE, class T, class BK>
DJ_HD void joint_eval(JointEval<T>& E, const NodeP<T>& P, const JointCfg<T>& cfg, bool has_parent,
                      const Kin<T>& ka, const Kin<T>& kb, const T* wa, const T* wb,
                      const T* lam, const T* lg, T dt, BK& K) {
    constexpr bool JAC = MODE >= 1;
    constexpr bool MAT = MODE == 1;
    // ---------------- translational displacement at (x3,q3): translational/minimal.jl:4-12 ----------------
    T t3[3], wv[3], e[3], u[3];
    m3vec(t3, kb.R3, P.pb);
    for (int i = 0; i < 3; ++i) wv[i] = kb.x3[i] + t3[i] - ka.x3[i];
    m3tvec(u, ka.R3, wv);                                  // u = e + pa
    for (int i = 0; i < 3; ++i) e[i] = u[i] - P.pa[i];
    // ---------------- rotational displacement: rotational/minimal.jl:4-11 ----------------
    T qab[4], qr[4];
    qcmul(qab, ka.q3, kb.q3);
    qcmul(qr, P.qoff, qab);
    for (int i = 0; i < 3; ++i) {
        E.g[i] = (i < P.nl_t) ? v3dot(&P.Ct[3 * i], e) : T(0);
        E.g[3 + i] = (i < P.nl_r) ? v3dot(&P.Cr[3 * i], &qr[1]) : T(0);
    }
    T rho[4] = {0, 0, 0, 0};
    E.theta = T(0);
    if (P.nlim_r > 0) { T rv[3]; rotvec(rv, qr); E.theta = v3dot(P.Ar, rv); }
    // ---------------- impulses at the current configuration ----------------
    T pt[3] = {0, 0, 0}, pr[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) {
        if (i < P.nl_t) for (int k = 0; k < 3; ++k) pt[k] += P.Ct[3 * i + k] * lam[i];
        if (i < P.nl_r) for (int k = 0; k < 3; ++k) pr[k] += P.Cr[3 * i + k] * lam[3 + i];
    }
    if (P.nlim_r > 0) { T kk = lg[1] - lg[0]; for (int k = 0; k < 3; ++k) pr[k] += P.Ar[k] * kk; }   // projector [0; −A; A; C]ᵀ, joint.jl:92
    T ia[6], ib[6], ja[6], jb[6];
    tra_impulse(ia, ib, cfg, P, pt);
    rot_impulse(ja, jb, cfg, pr);
    for (int i = 0; i < 6; ++i) { E.imp_a[i] = ia[i] + ja[i]; E.imp_b[i] = ib[i] + jb[i]; }
    if (JAC) {
        for (int i = 0; i < 18; ++i) { E.GaX[i] = E.GaP[i] = E.GbX[i] = E.GbP[i] = T(0); }
        // translational rows: E_xa = −Raᵀ, E_φa = 2[u]x, E_xb = Raᵀ, E_φb = −2 RaᵀRb[pb]x
        T RaT_Rb[9], Spb[9], Eb[9], Su[9], EaPhi[9], EbPhi[9];
        m3tmul(RaT_Rb, ka.R3, kb.R3);
        m3skew(Spb, P.pb);
        m3mul(Eb, RaT_Rb, Spb);
        for (int i = 0; i < 9; ++i) Eb[i] *= T(-2);
        m3skew(Su, u);
        for (int i = 0; i < 9; ++i) Su[i] *= T(2);
        m3mul(EaPhi, Su, ka.Phi);
        m3mul(EbPhi, Eb, kb.Phi);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i < P.nl_t) {
                const T* c = &P.Ct[3 * i];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    T cRaT = c[0] * ka.R3[3 * j] + c[1] * ka.R3[3 * j + 1] + c[2] * ka.R3[3 * j + 2];   // (c Raᵀ)_j
                    E.GaX[3 * i + j] = -cRaT; E.GbX[3 * i + j] = cRaT;
                    E.GaP[3 * i + j] = c[0] * Su[j] + c[1] * Su[3 + j] + c[2] * Su[6 + j];
                    E.GbP[3 * i + j] = c[0] * Eb[j] + c[1] * Eb[3 + j] + c[2] * Eb[6 + j];
                    if (MAT) {
                        K.addU(6 + i, j, -dt * cRaT);
                        K.addS(6 + i, j, dt * cRaT);
                        K.addU(6 + i, 3 + j, c[0] * EaPhi[j] + c[1] * EaPhi[3 + j] + c[2] * EaPhi[6 + j]);
                        K.addS(6 + i, 3 + j, c[0] * EbPhi[j] + c[1] * EbPhi[3 + j] + c[2] * EbPhi[6 + j]);
                    }
                }
                if (MAT) {
                    T a6[6], b6[6];
                    tra_impulse(a6, b6, cfg, P, c);
#pragma unroll
                    for (int r = 0; r < 6; ++r) { K.addL(r, 6 + i, -a6[r]); K.addS(r, 6 + i, -b6[r]); }
                }
            }
        }
        // rotational rows: E_φb = s I + [v]x ; E_φa = −(s I − [v]x) Roffᵀ
        T Erb[9], Em[9], Era[9], EraPhi[9], ErbPhi[9];
        m3sIpskew(Erb, qr[0], &qr[1]);
        T nv[3] = {-qr[1], -qr[2], -qr[3]};
        m3sIpskew(Em, qr[0], nv);
        m3mult(Era, Em, cfg.Roff);
        for (int i = 0; i < 9; ++i) Era[i] = -Era[i];
        m3mul(EraPhi, Era, ka.Phi);
        m3mul(ErbPhi, Erb, kb.Phi);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i < P.nl_r) {
                const T* c = &P.Cr[3 * i];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (MAT) {
                        K.addU(9 + i, 3 + j, c[0] * EraPhi[j] + c[1] * EraPhi[3 + j] + c[2] * EraPhi[6 + j]);
                        K.addS(9 + i, 3 + j, c[0] * ErbPhi[j] + c[1] * ErbPhi[3 + j] + c[2] * ErbPhi[6 + j]);
                    }
                    E.GaP[3 * (3 + i) + j] = c[0] * Era[j] + c[1] * Era[3 + j] + c[2] * Era[6 + j];
                    E.GbP[3 * (3 + i) + j] = c[0] * Erb[j] + c[1] * Erb[3 + j] + c[2] * Erb[6 + j];
                }
                if (MAT) {
                    T a6[6], b6[6];
                    rot_impulse(a6, b6, cfg, c);
#pragma unroll
                    for (int r = 0; r < 6; ++r) { K.addL(r, 9 + i, -a6[r]); K.addS(r, 9 + i, -b6[r]); }
                }
            }
        }
        if (P.nlim_r > 0) {
            // θ = Ar·rotation_vector(qr): ∂θ/∂φb = ρ L(qr)Vᵀ, ∂θ/∂φa = −ρ R(qr)Vᵀ Roffᵀ   (rotational/minimal.jl:69-80)
            rotvec_jac_row(rho, P.Ar, qr);
            T rl[4], rr[4], tb[3], ta0[3], ta[3];
            rowL(rl, rho, qr); rowR(rr, rho, qr);
            for (int j = 0; j < 3; ++j) { tb[j] = rl[1 + j]; ta0[j] = -rr[1 + j]; }
            m3vec(ta, cfg.Roff, ta0);               // (ta0ᵀ Roffᵀ)ᵀ = Roff ta0
            m3tvec(E.th_b, kb.Phi, tb);             // (tbᵀ Φ)ᵀ = Φᵀ tb
            for (int j = 0; j < 3; ++j) { E.thp_a[j] = ta[j]; E.thp_b[j] = tb[j]; }
            m3tvec(E.th_a, ka.Phi, ta);
            rot_impulse(E.t_a, E.t_b, cfg, P.Ar);
        } else {
            for (int j = 0; j < 3; ++j) E.th_a[j] = E.th_b[j] = E.thp_a[j] = E.thp_b[j] = T(0);
            for (int j = 0; j < 6; ++j) E.t_a[j] = E.t_b[j] = T(0);
        }
    }
    // ---------------- rotational damper (implicit in the candidate velocities): rotational/dampers.jl:4-30 ----------------
    if (P.damper_on && P.nl_r < 3 && P.damper_r != T(0)) {
        const T idt = trcp(dt), four_dt2 = T(4) * idt * idt;
        T ca = tsqrt(four_dt2 - v3dot(wa, wa)), cb = tsqrt(four_dt2 - v3dot(wb, wb));
        const T ica = trcp(ca), icb = trcp(cb);
        T h = dt * T(0.5);
        T wab[3];
        m3vec(wab, cfg.Rba, wa);                     // ωa expressed in the child frame
        T xbi[4] = {h * cb, h * wb[0], h * wb[1], h * wb[2]};
        T roa[4] = {h * ca, -h * wab[0], -h * wab[1], -h * wab[2]};
        T qd[4];
        qmul(qd, xbi, roa);                          // = inv(q1) ⊗ q of rotational/minimal.jl:103-118
        T rv[3];
        rotvec(rv, qd);
        T force[3] = {0, 0, 0};                      // damper · Aᵀ A rotvec / Δt   (offset frame)
        const int nur = 3 - P.nl_r;
        for (int i = 0; i < 3; ++i) if (i < nur) { T vel = v3dot(&P.Ar[3 * i], rv) * idt; for (int k = 0; k < 3; ++k) force[k] += P.damper_r * P.Ar[3 * i + k] * vel; }
        T ta[3], tb[3];
        m3vec(ta, cfg.Roff, force);                  // parent: vector_rotate(force, qoff)
        m3vec(tb, cfg.Rba, ta);                      // child: −R(qb⁻¹ qa qoff) force
        for (int k = 0; k < 3; ++k) { E.imp_a[3 + k] += dt * ta[k]; E.imp_b[3 + k] -= dt * tb[k]; }
        if (MAT) {
            // ∂vel_i/∂ωb, ∂vel_i/∂ωa  (rotational/minimal.jl:151-174 in closed form)
            T dFa[9], dFb[9];                        // ∂force/∂ωa, ∂force/∂ωb (3x3, offset frame)
            for (int i = 0; i < 9; ++i) dFa[i] = dFb[i] = T(0);
            for (int i = 0; i < 3; ++i) if (i < nur) {
                T row[4], r1[4], r2[4], gb[3], ga0[3], ga[3];
                rotvec_jac_row(row, &P.Ar[3 * i], qd);
                rowR(r1, row, roa);                  // row · R(ρa) : δqd = R(ρa) δξb⁻¹
                rowL(r2, row, xbi);                  // row · L(ξb⁻¹): δqd = L(ξb⁻¹) δρa
                for (int j = 0; j < 3; ++j) {
                    gb[j] = (T(0.5)) * (-r1[0] * wb[j] * icb + r1[1 + j]);                 // (1/Δt)(Δt/2)[−ωbᵀ/cb; I]
                    ga0[j] = r2[1 + j];
                }
                // [−ωaᵀ/ca; −Rba]: contribution −r2[0] ωa/ca − Rbaᵀ r2[1:3]
                m3tvec(ga, cfg.Rba, ga0);
                for (int j = 0; j < 3; ++j) ga[j] = T(0.5) * (-r2[0] * wa[j] * ica - ga[j]);
                for (int k = 0; k < 3; ++k) for (int j = 0; j < 3; ++j) {
                    dFa[3 * k + j] += P.damper_r * P.Ar[3 * i + k] * ga[j];
                    dFb[3 * k + j] += P.damper_r * P.Ar[3 * i + k] * gb[j];
                }
            }
            T A1[9], A2[9], B1[9], B2[9];
            m3mul(A1, cfg.Roff, dFa); m3mul(A2, cfg.Roff, dFb);       // ∂τa/∂ωa, ∂τa/∂ωb (without Δt)
            m3mul(B1, cfg.Rba, A1);   m3mul(B2, cfg.Rba, A2);         // −∂τb/∂ωa, −∂τb/∂ωb
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
              for (int j = 0; j < 3; ++j) {
                // d_a −= Δt τa, d_b −= −Δt τb  =>  ∂d_a/∂ω = −Δt ∂τa/∂ω, ∂d_b/∂ω = +Δt Rba ∂τa/∂ω
                K.addD(3 + k, 3 + j, -dt * A1[3 * k + j]);
                K.addL(3 + k, 3 + j, -dt * A2[3 * k + j]);
                K.addU(3 + k, 3 + j, dt * B1[3 * k + j]);
                K.addS(3 + k, 3 + j, dt * B2[3 * k + j]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ∂(joint impulse + spring + damper on parent / child body)/∂(x2, φ2 of parent / child) at the
// configuration `c`: impulse_map_jacobian (translational/impulses.jl:9-46, rotational/impulses.jl:9-39),
// spring_jacobian_configuration (rotational/springs.jl:46-94) and damper_jacobian_configuration
// (rotational/dampers.jl:36-64) in closed form.  J_xy = ∂(what is subtracted from d_x)/∂z2_y, 6x6,
// columns (x2(3), φ2(3)).
// ------------------------------------------------------------------------------------------------
template <class T>
DJ_HD void joint_impulse_cfg_jac(T* Jaa, T* Jab, T* Jba, T* Jbb, const NodeP<T>& P, const JointCfg<T>& c,
                                 const T* pt, const T* pr, const T* wa, const T* wb, T dt) {
    for (int i = 0; i < 36; ++i) Jaa[i] = Jab[i] = Jba[i] = Jbb[i] = T(0);
    // ---------------- translational ----------------
    {
        T Sp[9], RaSp[9], Su[9], SpRaT[9], SpSu[9], Spb[9], RbaT_Spb[9], SpE[9], RbaSp[9], Rp[3], SRp[9], M1[9], M2[9];
        m3skew(Sp, pt);
        m3mul(RaSp, c.Ra, Sp);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Jaa[6 * i + 3 + j] += T(2) * RaSp[3 * i + j]; Jba[6 * i + 3 + j] += T(-2) * RaSp[3 * i + j]; }
        m3mult(SpRaT, Sp, c.Ra);                               // Sp Raᵀ
        m3skew(Su, c.u2); m3mul(SpSu, Sp, Su);
        m3skew(Spb, P.pb); m3tmul(RbaT_Spb, c.Rba, Spb); m3mul(SpE, Sp, RbaT_Spb);   // Sp Rbaᵀ[pb]x
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            Jaa[6 * (3 + i) + j] += -SpRaT[3 * i + j];
            Jaa[6 * (3 + i) + 3 + j] += T(2) * SpSu[3 * i + j];
            Jab[6 * (3 + i) + j] += SpRaT[3 * i + j];
            Jab[6 * (3 + i) + 3 + j] += T(-2) * SpE[3 * i + j];
        }
        m3mul(RbaSp, c.Rba, Sp); m3mul(M1, Spb, RbaSp);         // [pb]x Rba [p]x
        m3vec(Rp, c.Rba, pt); m3skew(SRp, Rp); m3mul(M2, Spb, SRp);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Jba[6 * (3 + i) + 3 + j] += T(-2) * M1[3 * i + j]; Jbb[6 * (3 + i) + 3 + j] += T(2) * M2[3 * i + j]; }
    }
    // ---------------- rotational ----------------
    const T s = c.qr[0]; const T* v = &c.qr[1];
    T dsa[3], dva[9], dvb[9];                                   // δs = ds·φ, δv = dv φ
    m3vec(dsa, c.Roff, v);                                      // δs_a = (Roff v)·φa ; δs_b = −v·φb
    {
        T Em[9], nv[3] = {-v[0], -v[1], -v[2]};
        m3sIpskew(Em, s, nv);                                   // s I − [v]x
        m3mult(dva, Em, c.Roff);
        for (int i = 0; i < 9; ++i) dva[i] = -dva[i];
        m3sIpskew(dvb, s, v);
    }
    {
        T Sp[9], SpA[9], SpB[9], Ma[9], Mb[9], RMa[9], RMb[9];
        m3skew(Sp, pr);
        m3mul(SpA, Sp, dva); m3mul(SpB, Sp, dvb);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            Ma[3 * i + j] = pr[i] * dsa[j] - SpA[3 * i + j];        // ∂(s p + v×p)/∂φa
            Mb[3 * i + j] = -pr[i] * v[j] - SpB[3 * i + j];         // ∂(s p + v×p)/∂φb
        }
        m3mul(RMa, c.Roff, Ma); m3mul(RMb, c.Roff, Mb);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            Jaa[6 * (3 + i) + 3 + j] += T(-0.5) * RMa[3 * i + j];
            Jab[6 * (3 + i) + 3 + j] += T(-0.5) * RMb[3 * i + j];
            Jba[6 * (3 + i) + 3 + j] += T(0.5) * (pr[i] * dsa[j] + SpA[3 * i + j]);   // ∂(s p − v×p)/∂φa
            Jbb[6 * (3 + i) + 3 + j] += T(0.5) * (-pr[i] * v[j] + SpB[3 * i + j]);
        }
    }
    // ---------------- rotational spring + damper: torque in the offset frame f, with ∂f/∂φa, ∂f/∂φb ----------------
    const int nur = 3 - P.nl_r;
    T f[3] = {0, 0, 0}, dFa[9], dFb[9];
    for (int i = 0; i < 9; ++i) dFa[i] = dFb[i] = T(0);
    bool any = false;
    if (P.spring_on && P.nl_r < 3 && P.spring_r != T(0)) {
        any = true;
        T rv[3];
        rotvec(rv, c.qr);
        for (int i = 0; i < 3; ++i) if (i < nur) {
            T dist = P.spring_off_r[i] - v3dot(&P.Ar[3 * i], rv);
            T row[4], rl[4], rr[4], ta0[3], ta[3];
            rotvec_jac_row(row, &P.Ar[3 * i], c.qr);
            rowL(rl, row, c.qr); rowR(rr, row, c.qr);
            for (int j = 0; j < 3; ++j) ta0[j] = -rr[1 + j];
            m3vec(ta, c.Roff, ta0);
            for (int k_ = 0; k_ < 3; ++k_) {
                f[k_] += -P.spring_r * P.Ar[3 * i + k_] * dist;
                for (int j = 0; j < 3; ++j) { dFb[3 * k_ + j] += P.spring_r * P.Ar[3 * i + k_] * rl[1 + j]; dFa[3 * k_ + j] += P.spring_r * P.Ar[3 * i + k_] * ta[j]; }
            }
        }
    }
    if (P.damper_on && P.nl_r < 3 && P.damper_r != T(0)) {
        any = true;
        T ca = tsqrt(T(4) / (dt * dt) - v3dot(wa, wa)), cb = tsqrt(T(4) / (dt * dt) - v3dot(wb, wb));
        T h = dt * T(0.5);
        T wab[3];
        m3vec(wab, c.Rba, wa);
        T xbi[4] = {h * cb, h * wb[0], h * wb[1], h * wb[2]};
        T roa[4] = {h * ca, -h * wab[0], -h * wab[1], -h * wab[2]};
        T qd[4], rv[3];
        qmul(qd, xbi, roa);
        rotvec(rv, qd);
        T Swab[9], Swa[9], RSwa[9];
        m3skew(Swab, wab); m3skew(Swa, wa); m3mul(RSwa, c.Rba, Swa);
        for (int i = 0; i < 3; ++i) if (i < nur) {
            T vel = v3dot(&P.Ar[3 * i], rv) / dt;
            T row[4], r2[4], gb[3], ga[3];
            rotvec_jac_row(row, &P.Ar[3 * i], qd);
            rowL(r2, row, xbi);
            // δvel = −(1/2) r2[1:3]·δ(wab),  δ(wab) = 2[wab]x φb − 2 Rba[ωa]x φa
            for (int j = 0; j < 3; ++j) {
                gb[j] = -(r2[1] * Swab[j] + r2[2] * Swab[3 + j] + r2[3] * Swab[6 + j]);
                ga[j] = (r2[1] * RSwa[j] + r2[2] * RSwa[3 + j] + r2[3] * RSwa[6 + j]);
            }
            for (int k_ = 0; k_ < 3; ++k_) {
                f[k_] += P.damper_r * P.Ar[3 * i + k_] * vel;
                for (int j = 0; j < 3; ++j) { dFb[3 * k_ + j] += P.damper_r * P.Ar[3 * i + k_] * gb[j]; dFa[3 * k_ + j] += P.damper_r * P.Ar[3 * i + k_] * ga[j]; }
            }
        }
    }
    if (any) {
        // τa = Δt Roff f (subtracted from d_a), τb = −Δt Rba Roff f (subtracted from d_b)
        T w_[3], RdFa[9], RdFb[9], Sw[9], RbaSw[9], Rw[3], SRw[9], RbaRdFa[9], RbaRdFb[9];
        m3vec(w_, c.Roff, f);
        m3mul(RdFa, c.Roff, dFa); m3mul(RdFb, c.Roff, dFb);
        m3skew(Sw, w_); m3mul(RbaSw, c.Rba, Sw);
        m3vec(Rw, c.Rba, w_); m3skew(SRw, Rw);
        m3mul(RbaRdFa, c.Rba, RdFa); m3mul(RbaRdFb, c.Rba, RdFb);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            Jaa[6 * (3 + i) + 3 + j] += dt * RdFa[3 * i + j];
            Jab[6 * (3 + i) + 3 + j] += dt * RdFb[3 * i + j];
            Jba[6 * (3 + i) + 3 + j] += -dt * (T(-2) * RbaSw[3 * i + j] + RbaRdFa[3 * i + j]);
            Jbb[6 * (3 + i) + 3 + j] += -dt * (T(2) * SRw[3 * i + j] + RbaRdFb[3 * i + j]);
        }
    }
}

// rotational spring impulse (velocity independent): rotational/springs.jl:5-40; returns −(what d subtracts) pieces
template <class T> DJ_HD void spring_impulses(T* sa, T* sb, const NodeP<T>& P, const JointCfg<T>& cfg, T dt) {
    for (int i = 0; i < 6; ++i) sa[i] = sb[i] = T(0);
    if (!P.spring_on || P.nl_r == 3 || P.spring_r == T(0)) return;
    T rv[3];
    rotvec(rv, cfg.qr);
    T force[3] = {0, 0, 0};
    const int nur = 3 - P.nl_r;
    for (int i = 0; i < 3; ++i) if (i < nur) {
        T dist = P.spring_off_r[i] - v3dot(&P.Ar[3 * i], rv);
        for (int k = 0; k < 3; ++k) force[k] += -P.spring_r * P.Ar[3 * i + k] * dist;
    }
    T ta[3], tb[3];
    m3vec(ta, cfg.Roff, force);
    m3vec(tb, cfg.Rba, ta);
    for (int k = 0; k < 3; ++k) { sa[3 + k] = dt * ta[k]; sb[3 + k] = -dt * tb[k]; }
}

// ------------------------------------------------------------------------------------------------
// Translational springs and dampers of joints with free translations (Prismatic, Planar, Cylindrical ...):
// src/joints/translational/springs.jl:5-76, dampers.jl:5-123, minimal.jl:91-177 in closed form.
// Their parameters live outside NodeP (KernelArgs::tsd, one entry per supernode) and only the DJ_TSD builds of the
// kernels evaluate them: the kernels of mechanisms without them keep their code and their register allocation.
//   e(x, q) = Raᵀ(xb + Rb pb − xa) − pa,  u = e + pa            displacement in the parent frame (minimal.jl:4-12)
//   spring  f = k Aᵀ(offset − A e2)                              at the current configuration (velocity independent)
//   damper  f = −(c/Δt) AᵀA (e2 − e1),  e1 = e one step back along the CANDIDATE velocities (minimal.jl:91-110)
//   both enter the body residuals as Δt·T_a f (parent), Δt·T_b f (child), T = impulse_transform (tra_impulse)
// ------------------------------------------------------------------------------------------------
template <class T> struct TraSD { T spring, damper, off[3], lim_lo, lim_hi; int nlim; };   // nlim = 1: limits on the free translational coordinate (nl_t = 2)

template <class T> DJ_HD void tra_AtA(T* M, const NodeP<T>& P) {
    for (int i = 0; i < 9; ++i) M[i] = T(0);
    for (int i = 0; i < 3; ++i) if (i < P.nu_t) for (int k = 0; k < 3; ++k) for (int j = 0; j < 3; ++j) M[3 * k + j] += P.At[3 * i + k] * P.At[3 * i + j];
}
template <class T> DJ_HD void tra_spring_force(T* f, const NodeP<T>& P, const TraSD<T>& sd, const JointCfg<T>& c) {
    f[0] = f[1] = f[2] = T(0);
    if (!P.spring_on || sd.spring == T(0)) return;
    T e[3] = {c.u2[0] - P.pa[0], c.u2[1] - P.pa[1], c.u2[2] - P.pa[2]};
    for (int i = 0; i < 3; ++i) if (i < P.nu_t) {
        T dist = sd.off[i] - v3dot(&P.At[3 * i], e);
        for (int k = 0; k < 3; ++k) f[k] += sd.spring * P.At[3 * i + k] * dist;
    }
}
template <class T>
struct TraDamperEval {
    T f[3];                      // damper_force (parent frame)
    T u1[3];                     // e1 + pa
    T R1T[9];                    // R(qa1)ᵀ = Xaᵀ Raᵀ
    T XaT[9], XbT[9];            // R(ξ(−ωa))ᵀ, R(ξ(−ωb))ᵀ  = ∂φ1/∂φ2 of parent / child
    T Eb1[9];                    // ∂u1/∂φb1 = −2 Xaᵀ Rbaᵀ Xb [pb]x
    T Pa[9], Pb[9];              // Φ(−ωa), Φ(−ωb):  δφ1 = −Φ(−ω) δω
};
template <bool JAC, class T>
DJ_HD void tra_damper_force(TraDamperEval<T>& D, const NodeP<T>& P, const TraSD<T>& sd, const JointCfg<T>& c,
                            const T* va, const T* wa, const T* vb, const T* wb, T dt) {
    T nwa[3] = {-wa[0], -wa[1], -wa[2]}, nwb[3] = {-wb[0], -wb[1], -wb[2]};
    T xia[4], xib[4], ca, cb, Xa[9], Xb[9];
    qstep(xia, nwa, dt, &ca); qstep(xib, nwb, dt, &cb);
    qrot(Xa, xia); qrot(Xb, xib);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { D.XaT[3 * i + j] = Xa[3 * j + i]; D.XbT[3 * i + j] = Xb[3 * j + i]; }
    T dv[3] = {vb[0] - va[0], vb[1] - va[1], vb[2] - va[2]}, dvw[3], Xbpb[3], t[3], Rt[3], r[3];
    m3tvec(dvw, c.Ra, dv);
    m3vec(Xbpb, Xb, P.pb);
    for (int i = 0; i < 3; ++i) t[i] = Xbpb[i] - P.pb[i];
    m3tvec(Rt, c.Rba, t);                                   // Raᵀ Rb (Xb pb − pb)
    for (int i = 0; i < 3; ++i) r[i] = c.u2[i] + Rt[i] - dt * dvw[i];
    m3vec(D.u1, D.XaT, r);
    T M[9], du[3], Mdu[3];
    tra_AtA(M, P);
    for (int i = 0; i < 3; ++i) du[i] = c.u2[i] - D.u1[i];
    m3vec(Mdu, M, du);
    const T g = -sd.damper / dt;
    for (int i = 0; i < 3; ++i) D.f[i] = g * Mdu[i];
    if (JAC) {
        m3mult(D.R1T, D.XaT, c.Ra);
        phi_of(D.Pa, nwa, ca, dt); phi_of(D.Pb, nwb, cb, dt);
        T Spb[9], XbS[9], RX[9];
        m3skew(Spb, P.pb); m3mul(XbS, Xb, Spb); m3tmul(RX, c.Rba, XbS);
        m3mul(D.Eb1, D.XaT, RX);
        for (int i = 0; i < 9; ++i) D.Eb1[i] *= T(-2);
    }
}
// residual pieces (+ velocity Jacobian blocks when MAT) of the translational damper, added to what joint_eval left in E / K
// (damper_jacobian_velocity, translational/dampers.jl:99-123)
template <bool MAT, class T, class BK>
DJ_HD void tra_damper_eval(JointEval<T>& E, const NodeP<T>& P, const TraSD<T>& sd, const JointCfg<T>& c,
                           const T* va, const T* wa, const T* vb, const T* wb, T dt, BK& K) {
    if (!P.damper_on || sd.damper == T(0) || P.nu_t == 0) return;
    TraDamperEval<T> D;
    tra_damper_force<MAT>(D, P, sd, c, va, wa, vb, wb, dt);
    T p[3] = {dt * D.f[0], dt * D.f[1], dt * D.f[2]}, ia[6], ib[6];
    tra_impulse(ia, ib, c, P, p);
    for (int i = 0; i < 6; ++i) { E.imp_a[i] += ia[i]; E.imp_b[i] += ib[i]; }
    if (MAT) {
        // ∂f/∂(va, ωa) = (c/Δt) AᵀA [Δt R1ᵀ | −2[u1]x Φa],  ∂f/∂(vb, ωb) = (c/Δt) AᵀA [−Δt R1ᵀ | −Eb1 Φb]
        T M[9], Su1[9], SP[9], EP[9], FA[18], FB[18];
        tra_AtA(M, P);
        m3skew(Su1, D.u1); m3mul(SP, Su1, D.Pa); m3mul(EP, D.Eb1, D.Pb);
        const T g = sd.damper / dt;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            T a0 = T(0), a1 = T(0), b1 = T(0);
            for (int m = 0; m < 3; ++m) { a0 += M[3 * i + m] * D.R1T[3 * m + j]; a1 += M[3 * i + m] * SP[3 * m + j]; b1 += M[3 * i + m] * EP[3 * m + j]; }
            FA[6 * i + j] = g * dt * a0; FA[6 * i + 3 + j] = T(-2) * g * a1;
            FB[6 * i + j] = -g * dt * a0; FB[6 * i + 3 + j] = -g * b1;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            T ca_[3] = {dt * FA[j], dt * FA[6 + j], dt * FA[12 + j]}, cb_[3] = {dt * FB[j], dt * FB[6 + j], dt * FB[12 + j]};
            T aa[6], ba[6], ab[6], bb[6];
            tra_impulse(aa, ba, c, P, ca_);                 // ∂imp_a/∂(vel_a)_j, ∂imp_b/∂(vel_a)_j
            tra_impulse(ab, bb, c, P, cb_);                 // ∂imp_a/∂(vel_b)_j, ∂imp_b/∂(vel_b)_j
#pragma unroll
            for (int r = 0; r < 6; ++r) { K.addD(r, j, -aa[r]); K.addU(r, j, -ba[r]); K.addL(r, j, -ab[r]); K.addS(r, j, -bb[r]); }
        }
    }
}
// ∂(Δt T f)/∂z2 of the translational spring and damper for the data Jacobian (spring_jacobian_configuration,
// translational/springs.jl:37-76; damper_jacobian_configuration, dampers.jl:70-97): adds Δt T_x ∂f/∂z_y to the four blocks
// and returns Δt f, which the caller adds to the joint's translational impulse so that joint_impulse_cfg_jac supplies the
// impulse_transform_jacobian terms (they are linear in the transported vector).
template <class T>
DJ_HD void tra_sd_cfg_jac(T* Jaa, T* Jab, T* Jba, T* Jbb, T* pf, const NodeP<T>& P, const TraSD<T>& sd, const JointCfg<T>& c,
                          const T* va, const T* wa, const T* vb, const T* wb, T dt) {
    pf[0] = pf[1] = pf[2] = T(0);
    const bool sp = P.spring_on && sd.spring != T(0) && P.nu_t > 0, da = P.damper_on && sd.damper != T(0) && P.nu_t > 0;
    if (!sp && !da) return;
    T M[9], Su2[9], Spb[9], E2b[9], RaT[9];
    tra_AtA(M, P);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) RaT[3 * i + j] = c.Ra[3 * j + i];
    m3skew(Su2, c.u2); m3skew(Spb, P.pb); m3tmul(E2b, c.Rba, Spb);       // ∂u2/∂φa = 2[u2]x, ∂u2/∂φb = −2 Rbaᵀ[pb]x
    T GA[18], GB[18];                                                       // Σ coefficient · ∂u/∂za, ∂u/∂zb  (3x6 each)
    T ks = T(0), kd = T(0);
    if (sp) { T f[3]; tra_spring_force(f, P, sd, c); for (int i = 0; i < 3; ++i) pf[i] += dt * f[i]; ks = -sd.spring; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        GA[6 * i + j] = -RaT[3 * i + j]; GA[6 * i + 3 + j] = T(2) * Su2[3 * i + j];
        GB[6 * i + j] = RaT[3 * i + j]; GB[6 * i + 3 + j] = T(-2) * E2b[3 * i + j];
    }
    T HA[18], HB[18];
    for (int i = 0; i < 18; ++i) HA[i] = HB[i] = T(0);
    if (da) {
        TraDamperEval<T> D;
        tra_damper_force<true>(D, P, sd, c, va, wa, vb, wb, dt);
        for (int i = 0; i < 3; ++i) pf[i] += dt * D.f[i];
        kd = -sd.damper / dt;
        T Su1[9], SX[9], EX[9];
        m3skew(Su1, D.u1); m3mul(SX, Su1, D.XaT); m3mul(EX, D.Eb1, D.XbT);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            HA[6 * i + j] = -D.R1T[3 * i + j]; HA[6 * i + 3 + j] = T(2) * SX[3 * i + j];
            HB[6 * i + j] = D.R1T[3 * i + j]; HB[6 * i + 3 + j] = EX[3 * i + j];
        }
    }
    // ∂f/∂z = ks M ∂u2/∂z + kd M (∂u2/∂z − ∂u1/∂z)
    T FA[18], FB[18];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 6; ++j) {
        T a = T(0), b = T(0);
        for (int m = 0; m < 3; ++m) {
            a += M[3 * i + m] * ((ks + kd) * GA[6 * m + j] - kd * HA[6 * m + j]);
            b += M[3 * i + m] * ((ks + kd) * GB[6 * m + j] - kd * HB[6 * m + j]);
        }
        FA[6 * i + j] = a; FB[6 * i + j] = b;
    }
    for (int j = 0; j < 6; ++j) {
        T ca_[3] = {dt * FA[j], dt * FA[6 + j], dt * FA[12 + j]}, cb_[3] = {dt * FB[j], dt * FB[6 + j], dt * FB[12 + j]};
        T aa[6], ba[6], ab[6], bb[6];
        tra_impulse(aa, ba, c, P, ca_);
        tra_impulse(ab, bb, c, P, cb_);
        for (int r = 0; r < 6; ++r) { Jaa[6 * r + j] += aa[r]; Jba[6 * r + j] += ba[r]; Jab[6 * r + j] += ab[r]; Jbb[6 * r + j] += bb[r]; }
    }
}

#if DJ_TSD
// Limit on the free translational coordinate of a joint with nl_t = 2 (Prismatic, Cylindrical, PrismaticOrbital,
// CylindricalFree): src/joints/limits.jl:4-61 with the translational coordinate θ = A·e(x3, q3) (translational/minimal.jl:56-61).
// Same elimination as the rotational limit (one Δκ row), kept in the free third TRANSLATIONAL slot (row 8).
template <bool JAC, class T>
DJ_HD void tra_limit_eval(JointEval<T>& E, const NodeP<T>& P, const JointCfg<T>& cfg, const Kin<T>& ka, const Kin<T>& kb, const T* lg, T dt) {
    T t3[3], wv[3], u[3];
    m3vec(t3, kb.R3, P.pb);
    for (int i = 0; i < 3; ++i) wv[i] = kb.x3[i] + t3[i] - ka.x3[i];
    m3tvec(u, ka.R3, wv);
    const T* a = P.At;
    E.theta = a[0] * (u[0] - P.pa[0]) + a[1] * (u[1] - P.pa[1]) + a[2] * (u[2] - P.pa[2]);
    const T kk = lg[1] - lg[0];                              // projector [0; −A; A; C]ᵀ, joint.jl:92
    T p[3] = {a[0] * kk, a[1] * kk, a[2] * kk}, ia[6], ib[6];
    tra_impulse(ia, ib, cfg, P, p);
    for (int i = 0; i < 6; ++i) { E.imp_a[i] += ia[i]; E.imp_b[i] += ib[i]; }
    if (JAC) {
        T RaT_Rb[9], Spb[9], Eb[9], Su[9], SuP[9], EbP[9];
        m3tmul(RaT_Rb, ka.R3, kb.R3); m3skew(Spb, P.pb); m3mul(Eb, RaT_Rb, Spb);
        for (int i = 0; i < 9; ++i) Eb[i] *= T(-2);
        m3skew(Su, u);
        for (int i = 0; i < 9; ++i) Su[i] *= T(2);
        m3mul(SuP, Su, ka.Phi); m3mul(EbP, Eb, kb.Phi);
        for (int j = 0; j < 3; ++j) {
            const T aRaT = a[0] * ka.R3[3 * j] + a[1] * ka.R3[3 * j + 1] + a[2] * ka.R3[3 * j + 2];
            E.thx_a[j] = -aRaT; E.thx_b[j] = aRaT; E.thv_a[j] = -dt * aRaT; E.thv_b[j] = dt * aRaT;
            E.thp_a[j] = a[0] * Su[j] + a[1] * Su[3 + j] + a[2] * Su[6 + j];
            E.thp_b[j] = a[0] * Eb[j] + a[1] * Eb[3 + j] + a[2] * Eb[6 + j];
            E.th_a[j] = a[0] * SuP[j] + a[1] * SuP[3 + j] + a[2] * SuP[6 + j];
            E.th_b[j] = a[0] * EbP[j] + a[1] * EbP[3 + j] + a[2] * EbP[6 + j];
        }
        tra_impulse(E.t_a, E.t_b, cfg, P, a);
    }
}
#endif

// ------------------------------------------------------------------------------------------------
// contact evaluation at (x3, q3) of the body: residual rows, impulse, and (JAC) C/G blocks
// ------------------------------------------------------------------------------------------------
template <class T>
struct ContactEval { T c[NCV]; T imp[6]; T C134[18]; T G134[18]; T Dww[9]; T c1p[3], c34p[6], Qraw[9]; };   // raw = before Φ

template <bool JAC, class T>
DJ_HD void contact_eval(ContactEval<T>& E, const ContactP<T>& K, const Kin<T>& kb, const T* v, const T* w, const T* s, const T* gam, T dt) {
    T Ro[3], l[3], Rw[3], t[3], vp[3];
    m3vec(Ro, kb.R3, K.o);
    for (int i = 0; i < 3; ++i) l[i] = Ro[i] - K.off[i] - K.n[i] * K.r;           // contact_point − x  (sphere_halfspace.jl:55-58)
    T dist = T(0);
    for (int i = 0; i < 3; ++i) dist += K.n[i] * (kb.x3[i] + Ro[i] - K.off[i]);
    dist -= K.r;                                                                // distance (sphere_halfspace.jl:34-36)
    m3vec(Rw, kb.R3, w);
    v3cross(t, Rw, l);
    for (int i = 0; i < 3; ++i) vp[i] = v[i] + t[i];                              // contact_point_velocity (velocity.jl:2-4)
    E.c[0] = dist - s[0];
    T F[3], tau[3], lxF[3];
    if constexpr (kLinear) {
        // LinearContact (linear.jl:71-102), variables [γ ψ β1..β4]: μγ − Σβ − sψ;  P vt + ψ 1 − sβ with the friction_parameterization
        // P = [0 1; 0 −1; 1 0; −1 0] (linear.jl:33-38);  force = n γ + Tᵀ Pᵀ β = n γ + t1 (β3 − β4) + t2 (β1 − β2)  (contact.jl:141-154)
        const T vt1 = v3dot(&K.t[0], vp), vt2 = v3dot(&K.t[3], vp), psi = gam[1];
        E.c[1] = K.mu * gam[0] - (gam[2] + gam[3] + gam[4] + gam[5]) - s[1];
        E.c[2] = vt2 + psi - s[2]; E.c[3] = -vt2 + psi - s[3]; E.c[4] = vt1 + psi - s[4]; E.c[5] = -vt1 + psi - s[5];
        for (int i = 0; i < 3; ++i) F[i] = K.n[i] * gam[0] + K.t[i] * (gam[4] - gam[5]) + K.t[3 + i] * (gam[2] - gam[3]);
    } else {
        E.c[1] = K.mu * gam[0] - gam[1];
        E.c[2] = v3dot(&K.t[0], vp) - s[2];
        E.c[3] = v3dot(&K.t[3], vp) - s[3];
        for (int i = 0; i < 3; ++i) F[i] = K.n[i] * gam[0] + K.t[i] * gam[2] + K.t[3 + i] * gam[3];
    }
    v3cross(lxF, l, F);
    m3tvec(tau, kb.R3, lxF);
    for (int i = 0; i < 3; ++i) { E.imp[i] = F[i]; E.imp[3 + i] = tau[i]; }
    if (JAC) {
        const T* dirs[3] = {K.n, &K.t[0], &K.t[3]};
        for (int d = 0; d < 3; ++d) {
            T lx[3], g3[3];
            v3cross(lx, l, dirs[d]);
            m3tvec(g3, kb.R3, lx);
            for (int i = 0; i < 3; ++i) { E.G134[6 * d + i] = dirs[d][i]; E.G134[6 * d + 3 + i] = g3[i]; }   // stored as 3 rows of 6 (= G134ᵀ)
        }
        // row 1: [Δt n | n·(−2 R[o]x) Φ]
        T So[9], RSo[9], M1[9], A[9];
        m3skew(So, K.o);
        m3mul(RSo, kb.R3, So);                               // R[o]x
        for (int i = 0; i < 9; ++i) M1[i] = T(-2) * RSo[i];
        m3mul(A, M1, kb.Phi);                                // (−2R[o]x)Φ
        for (int j = 0; j < 3; ++j) { E.C134[j] = dt * K.n[j]; E.C134[3 + j] = K.n[0] * A[j] + K.n[1] * A[3 + j] + K.n[2] * A[6 + j]; E.c1p[j] = K.n[0] * M1[j] + K.n[1] * M1[3 + j] + K.n[2] * M1[6 + j]; }
        // rows 3,4: [T_i | T_i(−[l]x R + (2[l]x R [ω]x − 2[Rω]x R [o]x)Φ)]
        T Sl[9], SlR[9], Sw[9], SRw[9], B1[9], B2[9], B[9], BPhi[9];
        m3skew(Sl, l); m3mul(SlR, Sl, kb.R3);
        m3skew(Sw, w); m3mul(B1, SlR, Sw);
        m3skew(SRw, Rw); m3mul(B2, SRw, RSo);
        for (int i = 0; i < 9; ++i) B[i] = T(2) * (B1[i] - B2[i]);
        m3mul(BPhi, B, kb.Phi);
        for (int r = 0; r < 2; ++r) {
            const T* tt = &K.t[3 * r];
            for (int j = 0; j < 3; ++j) {
                E.C134[6 * (1 + r) + j] = tt[j];
                T a = -(tt[0] * SlR[j] + tt[1] * SlR[3 + j] + tt[2] * SlR[6 + j]);
                T b = tt[0] * BPhi[j] + tt[1] * BPhi[3 + j] + tt[2] * BPhi[6 + j];
                E.C134[6 * (1 + r) + 3 + j] = a + b;
                E.c34p[3 * r + j] = tt[0] * B[j] + tt[1] * B[3 + j] + tt[2] * B[6 + j];
            }
        }
        // ∂(G γ)/∂ω (rows 3:6, cols 3:6) = 2([τ]x + [F_b]x[o]x)Φ      (contact.jl:102-138)
        T Fb[3], St[9], SF[9], SFSo[9], Q[9];
        m3tvec(Fb, kb.R3, F);
        m3skew(St, tau); m3skew(SF, Fb); m3mul(SFSo, SF, So);
        for (int i = 0; i < 9; ++i) Q[i] = T(2) * (St[i] + SFSo[i]);
        m3mul(E.Dww, Q, kb.Phi);
        for (int i = 0; i < 9; ++i) E.Qraw[i] = Q[i];
    }
}

// Body-body contact: SphereSphereCollision between the owner's tree parent (the contact's parent_id: sphere of radius K.r about the point
// K.o of that body) and the owner (child_id: K.r2 about K.o2).  src/contacts/collisions/{collision,sphere_sphere}.jl,
// src/contacts/{contact,velocity}.jl in closed form (the analytic expressions; the reference returns FiniteDiff values of the same).  With
// c_p = x_p + R_p o_p, c_c = x_c + R_c o_c:
//   n = (c_p − c_c)/|c_p − c_c| (child -> parent), N = ∂n/∂c_p = (I − n nᵀ)/|c_p − c_c| = −∂n/∂c_c,  t1 = w × n (w = e_x, or e_y when
//   |e_x × n| <= 1e-6), t2 = t1 × n (neither normalized: collision.jl:102-129),  levers l_p = R_p o_p − r n, l_c = R_c o_c + r2 n,
//   force on the parent F = n γ1 + t1 g1 + t2 g2, on the child −F  (force_mapping, contact.jl:141-154),
//   E_s = ∂(R_s o_s)/∂q_s ∂q_s/∂ω_s = −2 R_s [o_s]x Φ_s   (the sphere centres move with the bodies' rotations; zero for centred spheres).
// `kb` / `ka`: the owner's and the parent's kinematics at the candidate velocities.
template <class T>
struct ContactEvalSS { T imp_p[6]; T Cp134[18], Gp134[18]; T Sxx[9], Sxw[9], Swx[9], Pxw[9], Pwx[9], Pww[9]; };   // parent-side rows; −∂(impulse)/∂(v, ω) blocks: own (v,v) [= parent's], (v,ω), (ω,v); parent (v,ω), (ω,v), (ω,ω)
template <bool JAC, class T>
DJ_HD void contact_eval_ss(ContactEval<T>& E, ContactEvalSS<T>& P2, const ContactP<T>& K, const Kin<T>& kb, const Kin<T>& ka,
                           const T* v, const T* w, const T* va, const T* wa, const T* s, const T* gam, T dt, bool impact) {
    T dx[3], n[3], t1[3], t2[3], wax[3] = {T(1), T(0), T(0)}, Rop[3], Roc[3];
    m3vec(Rop, ka.R3, K.o); m3vec(Roc, kb.R3, K.o2);
    for (int i = 0; i < 3; ++i) dx[i] = (ka.x3[i] + Rop[i]) - (kb.x3[i] + Roc[i]);
    const T dist = tsqrt(v3dot(dx, dx)), id = trcp(dist);
    for (int i = 0; i < 3; ++i) n[i] = dx[i] * id;
    v3cross(t1, wax, n);
    if (!(tsqrt(v3dot(t1, t1)) > T(1e-6))) { wax[0] = T(0); wax[1] = T(1); v3cross(t1, wax, n); }
    v3cross(t2, t1, n);
    if (impact) for (int i = 0; i < 3; ++i) t1[i] = t2[i] = T(0);      // ImpactContact (impact.jl:106-118): the normal alone; the friction block stays at the neutral vector
    T lp[3], lc[3], Rwp[3], Rwc[3], cp_[3], cc_[3], dv[3];
    for (int i = 0; i < 3; ++i) { lp[i] = Rop[i] - K.r * n[i]; lc[i] = Roc[i] + K.r2 * n[i]; }
    m3vec(Rwp, ka.R3, wa); m3vec(Rwc, kb.R3, w);
    v3cross(cp_, Rwp, lp); v3cross(cc_, Rwc, lc);
    for (int i = 0; i < 3; ++i) dv[i] = (va[i] + cp_[i]) - (v[i] + cc_[i]);           // contact point velocities, velocity.jl:2-38
    E.c[0] = (dist - (K.r + K.r2)) - s[0];                                               // distance, sphere_sphere.jl:28-38
    T gt1, gt2;                                                                          // the tangential impulse along t1, t2 (Pᵀβ for the LinearContact pyramid)
    if constexpr (kLinear) {                                                             // linear.jl:71-102, as contact_eval
        const T vt1 = v3dot(t1, dv), vt2 = v3dot(t2, dv), psi = gam[1];
        E.c[1] = K.mu * gam[0] - (gam[2] + gam[3] + gam[4] + gam[5]) - s[1];
        E.c[2] = vt2 + psi - s[2]; E.c[3] = -vt2 + psi - s[3]; E.c[4] = vt1 + psi - s[4]; E.c[5] = -vt1 + psi - s[5];
        gt1 = gam[4] - gam[5]; gt2 = gam[2] - gam[3];
    } else {
        E.c[1] = K.mu * gam[0] - gam[1];
        E.c[2] = v3dot(t1, dv) - s[2];
        E.c[3] = v3dot(t2, dv) - s[3];
        gt1 = gam[2]; gt2 = gam[3];
    }
    T F[3], lxF[3], tau[3];
    for (int i = 0; i < 3; ++i) F[i] = n[i] * gam[0] + t1[i] * gt1 + t2[i] * gt2;
    v3cross(lxF, lp, F); m3tvec(tau, ka.R3, lxF);
    for (int i = 0; i < 3; ++i) { P2.imp_p[i] = F[i]; P2.imp_p[3 + i] = tau[i]; }                                // impulse_map(:parent) γ
    T Fc[3] = {-F[0], -F[1], -F[2]}, tauc[3];
    v3cross(lxF, lc, Fc); m3tvec(tauc, kb.R3, lxF);
    for (int i = 0; i < 3; ++i) { E.imp[i] = Fc[i]; E.imp[3 + i] = tauc[i]; }                                    // impulse_map(:child) γ
    if (JAC) {
        const T* dirs[3] = {n, t1, t2};
        for (int d = 0; d < 3; ++d) {
            T lx[3], g3[3];
            v3cross(lx, lp, dirs[d]); m3tvec(g3, ka.R3, lx);
            for (int i = 0; i < 3; ++i) { P2.Gp134[6 * d + i] = dirs[d][i]; P2.Gp134[6 * d + 3 + i] = g3[i]; }
            v3cross(lx, lc, dirs[d]); m3tvec(g3, kb.R3, lx);
            for (int i = 0; i < 3; ++i) { E.G134[6 * d + i] = -dirs[d][i]; E.G134[6 * d + 3 + i] = -g3[i]; }
        }
        T N[9], Sn[9], Swx_[9], St1[9], SnSw[9], T2m[9], NE[2][9], Es[2][9], Swp[9], Swc[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) N[3 * i + j] = ((i == j ? T(1) : T(0)) - n[i] * n[j]) * id;
        m3skew(Sn, n); m3skew(Swx_, wax); m3skew(St1, t1); m3mul(SnSw, Sn, Swx_);
        for (int i = 0; i < 9; ++i) T2m[i] = St1[i] - SnSw[i];                          // ∂t2/∂n = [t1]x − [n]x [w]x
        m3skew(Swp, Rwp); m3skew(Swc, Rwc);
        for (int side = 0; side < 2; ++side) {                                          // E_s and N E_s
            const Kin<T>& kk = side == 0 ? ka : kb;
            T So[9], RSo[9];
            m3skew(So, side == 0 ? K.o : K.o2); m3mul(RSo, kk.R3, So);
            for (int i = 0; i < 9; ++i) RSo[i] *= T(-2);
            m3mul(Es[side], RSo, kk.Phi);
            m3mul(NE[side], N, Es[side]);
        }
        // constraint_jacobian_velocity(:parent | :child)  contact.jl:37-77.  Row 1 = ±[Δt n | nᵀ E_s].  Rows 3, 4: the v columns ±t_i (the position
        // dependence of vt is not in V, as in the reference); the ω columns ∂vt∂ω + ∂vt∂q ∂q∂ω (velocity.jl:71-143):
        //   ±t_i(−[l_s]x R_s + 2 [l_s]x R_s [ω_s]x Φ_s)  +  t_i([R_p ω_p]x ∂c_p'/∂ω_s − [R_c ω_c]x ∂c_c'/∂ω_s)  +  Δvᵀ ∂t_iᵀ/∂q_s ∂q_s/∂ω_s
        // with the contact points c_p' = c_p − r n, c_c' = c_c + r2 n, and ∂t1ᵀ/∂q = [t1]x ∂nᵀ/∂q -- the reference's literal form (collision.jl:207;
        // the ∂x version has [w]x there) -- , ∂t2ᵀ/∂q = ([t1]x − [n]x [w]x) ∂nᵀ/∂q.
        for (int side = 0; side < 2; ++side) {
            const Kin<T>& kk = side == 0 ? ka : kb; const T* l = side == 0 ? lp : lc; const T* ww = side == 0 ? wa : w;
            const T sg = side == 0 ? T(1) : T(-1);
            T* Cx = side == 0 ? P2.Cp134 : E.C134;
            T Sl[9], SlR[9], Sw[9], B1[9], BPhi[9];
            m3skew(Sl, l); m3mul(SlR, Sl, kk.R3); m3skew(Sw, ww); m3mul(B1, SlR, Sw);
            for (int i = 0; i < 9; ++i) B1[i] *= T(2);
            m3mul(BPhi, B1, kk.Phi);
            // M = [R_p ω_p]x ∂c_p'/∂ω_s − [R_c ω_c]x ∂c_c'/∂ω_s;  ∂c_s'/∂ω_s = E_s − r_s N E_s, the other point's: r_other N E_s
            T Dp[9], Dc[9], M1[9], M2[9], Q1[9], Q2[9];
            for (int i = 0; i < 9; ++i) { Dp[i] = side == 0 ? Es[0][i] - K.r * NE[0][i] : K.r * NE[1][i]; Dc[i] = side == 0 ? K.r2 * NE[0][i] : Es[1][i] - K.r2 * NE[1][i]; }
            m3mul(M1, Swp, Dp); m3mul(M2, Swc, Dc);
            m3mul(Q1, St1, NE[side]); m3mul(Q2, T2m, NE[side]);                        // (∂n/∂ω_s = ±N E_s: the sign is sg)
            for (int j = 0; j < 3; ++j) { Cx[j] = sg * dt * n[j]; Cx[3 + j] = sg * (n[0] * Es[side][j] + n[1] * Es[side][3 + j] + n[2] * Es[side][6 + j]); }
            for (int r = 0; r < 2; ++r) {
                const T* tt = r == 0 ? t1 : t2; const T* Qr = r == 0 ? Q1 : Q2;
                for (int j = 0; j < 3; ++j) {
                    Cx[6 * (1 + r) + j] = sg * tt[j];
                    const T a = -(tt[0] * SlR[j] + tt[1] * SlR[3 + j] + tt[2] * SlR[6 + j]);
                    const T b = tt[0] * BPhi[j] + tt[1] * BPhi[3 + j] + tt[2] * BPhi[6 + j];
                    const T m_ = tt[0] * (M1[j] - M2[j]) + tt[1] * (M1[3 + j] - M2[3 + j]) + tt[2] * (M1[6 + j] - M2[6 + j]);
                    const T q_ = impact ? T(0) : sg * (dv[0] * Qr[j] + dv[1] * Qr[3 + j] + dv[2] * Qr[6 + j]);
                    Cx[6 * (1 + r) + 3 + j] = sg * (a + b) + m_ + q_;
                }
            }
        }
        // impulse_map_jacobian(relative, relative, ..., γ) · integrator_jacobian_velocity  contact.jl:102-138, for both bodies:
        //   Xx = K N,  K = γ1 I + g1 [w]x + g2 ([t1]x − [n]x [w]x);   Xq ∂q∂ω = Kq N E_s,  Kq = γ1 I + g1 [t1]x + g2 ([t1]x − [n]x [w]x)  (literal ∂t1ᵀ/∂q)
        //   Qx = R_sᵀ([l_s]x K N + r_s [F_s]x N)      (∂c_s'/∂x_s − I = −r_s N)
        //   Qq ∂q∂ω = R_sᵀ([l_s]x Kq N E_s − [F_s]x (E_s − r_s N E_s)) + 2 [τ_s]x Φ_s
        T Kw[9], Kq[9], KN[9];
        for (int i = 0; i < 9; ++i) { Kw[i] = gt1 * Swx_[i] + gt2 * T2m[i]; Kq[i] = gt1 * St1[i] + gt2 * T2m[i]; }
        for (int i = 0; i < 3; ++i) { Kw[4 * i] += gam[0]; Kq[4 * i] += gam[0]; }
        m3mul(KN, Kw, N);
        for (int i = 0; i < 9; ++i) P2.Sxx[i] = dt * KN[i];
        for (int side = 0; side < 2; ++side) {
            const Kin<T>& kk = side == 0 ? ka : kb; const T* l = side == 0 ? lp : lc; const T* Fs = side == 0 ? F : Fc;
            const T rr = side == 0 ? K.r : K.r2; const T* tb = side == 0 ? tau : tauc;
            T Sl[9], SF[9], A1[9], A2[9], W[9], Xw[9], B3[9], B4[9], Dd[9];
            m3skew(Sl, l); m3skew(SF, Fs); m3mul(A1, Sl, KN); m3mul(A2, SF, N);
            for (int i = 0; i < 9; ++i) W[i] = A1[i] + rr * A2[i];
            T* Qx = side == 0 ? P2.Pwx : P2.Swx;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Qx[3 * i + j] = dt * (kk.R3[i] * W[j] + kk.R3[3 + i] * W[3 + j] + kk.R3[6 + i] * W[6 + j]);   // Rᵀ W
            m3mul(Xw, Kq, NE[side]);                                                   // Xq ∂q∂ω
            T* Xo = side == 0 ? P2.Pxw : P2.Sxw;
            for (int i = 0; i < 9; ++i) { Xo[i] = Xw[i]; Dd[i] = Es[side][i] - rr * NE[side][i]; }
            m3mul(B3, Sl, Xw); m3mul(B4, SF, Dd);
            for (int i = 0; i < 9; ++i) B3[i] -= B4[i];
            T St[9], Q2[9], QP[9];
            m3skew(St, tb);
            for (int i = 0; i < 9; ++i) Q2[i] = T(2) * St[i];
            m3mul(QP, Q2, kk.Phi);
            T* Dw = side == 0 ? P2.Pww : E.Dww;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Dw[3 * i + j] = QP[3 * i + j] + (kk.R3[i] * B3[j] + kk.R3[3 + i] * B3[3 + j] + kk.R3[6 + i] * B3[6 + j]);
            if (side == 1) for (int i = 0; i < 9; ++i) E.Qraw[i] = Q2[i];
        }
        for (int j = 0; j < 3; ++j) E.c1p[j] = T(0);
        for (int j = 0; j < 6; ++j) E.c34p[j] = T(0);
    }
}

// second-order-cone helpers: src/contacts/cone.jl, src/solver/line_search.jl:98-139
template <class T> DJ_HD T ort_step(T lam, T dl, T tau) { return dl < T(0) ? tmin(T(1), -tau * lam * trcp(dl)) : T(1); }
template <class T> DJ_HD T soc_step(const T* l, const T* d, T tau) {
    const T eps = T(1e-14);
    T l0 = l[0];
    T ll = tmax(l0 * l0 - (l[1] * l[1] + l[2] * l[2]), T(1e-25));
    ll += eps;
    T ld = l0 * d[0] - (l[1] * d[1] + l[2] * d[2]) + eps;
    const T ill = trcp(ll), sq = tsqrt(ll), isq = trcp(sq);
    T rs = ld * ill;
    T f = (ld * isq + d[0]) * trcp(l0 * isq + T(1));
    T r1 = d[1] * isq - f * l[1] * ill, r2 = d[2] * isq - f * l[2] * ill;
    T nr = tsqrt(r1 * r1 + r2 * r2);
    T a = T(1);
    if (nr - rs > T(0)) a = tmin(a, tau * trcp(nr - rs));
    return a;
}

// ------------------------------------------------------------------------------------------------
// Wave-level helpers.  `Wave` provides: lane(), shfl(v, src_lane), any(pred).
// ------------------------------------------------------------------------------------------------
template <class Wave, class T> DJ_HD T lane_xor(Wave& w, T v, int o) { return o == 1 ? w.quad_xor(v, 1) : o == 2 ? w.quad_xor(v, 2) : w.shfl(v, w.lane() ^ o); }
// reductions over the lanes of one environment.  Several wavefronts per environment (Wave::kWaves > 1, one environment
// per workgroup): through the workgroup reductions of the Wave.
template <class Wave, class T> DJ_HD T env_max(Wave& w, T v, int S) { if constexpr (Wave::kWaves > 1) return T(w.wg_max((double)v)); else { for (int o = S >> 1; o > 0; o >>= 1) v = tmax(v, lane_xor(w, v, o)); return v; } }
template <class Wave, class T> DJ_HD T env_min(Wave& w, T v, int S) { if constexpr (Wave::kWaves > 1) return T(w.wg_min((double)v)); else { for (int o = S >> 1; o > 0; o >>= 1) v = tmin(v, lane_xor(w, v, o)); return v; } }
template <class Wave, class T> DJ_HD T env_sum(Wave& w, T v, int S) { if constexpr (Wave::kWaves > 1) return T(w.wg_sum((double)v)); else { for (int o = S >> 1; o > 0; o >>= 1) v = v + lane_xor(w, v, o); return v; } }
template <class Wave> DJ_HD int env_or(Wave& w, int v, int S) { if constexpr (Wave::kWaves > 1) return w.wg_or(v); else { for (int o = S >> 1; o > 0; o >>= 1) v = v | lane_xor(w, v, o); return v; } }

template <int N, class Wave, class T> DJ_HD void shfl_vec(Wave& w, T* out, const T* in, int src) {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = w.shfl(in[i], src);
}
// acc[0:N] += Σ_children in[child]
template <int N, class Wave, class T, class NP> DJ_HD void gather_children(Wave& w, T* acc, const T* in, const NP& P, int base, int maxch, bool active, int stride = 1, int q = 0) {
#pragma unroll
    for (int ci = 0; ci < MAXCH; ++ci) {
        if (ci < maxch) {
            int src = (active && ci < P.nchild) ? base + stride * P.child[ci] + q : w.lane();
            bool use = active && ci < P.nchild;
            const T usef = use ? T(1) : T(0);                  // unused slots read the lane's own (finite) value and add 0·t
#pragma unroll
            for (int i = 0; i < N; ++i) { T t = w.shfl(in[i], src); acc[i] += usef * t; }
        }
    }
}

// the NodeP fields the IFT column sweeps need, cached in registers (the sweeps' right-hand sides overlay NodeP in LDS)
struct SweepP { int level, parent, pb, u_off, myu, nlim_r, nchild, child_lane0[MAXCH], ncontact, contact[8];
                unsigned long long sub_mask, ub_mask,      // bodies in this supernode's subtree (itself included) | control / contact batches owned inside it
                                   sub_t, ub_t; };         // ... the same for every slot with k < Nb, whether its environment exists or not (topology)

// 8 bytes at p read as a double, or their first 4 as a float (isd false) -- without a branch: two 32-bit reads and a select
template <bool ALWAYS_DOUBLE> DJ_HD double lds_read_as_double(const char* p, bool isd) {
    if constexpr (ALWAYS_DOUBLE) { double v; __builtin_memcpy(&v, p, 8); return v; }
    else {
        unsigned lo, hi; __builtin_memcpy(&lo, p, 4); __builtin_memcpy(&hi, p + 4, 4);
        float f; __builtin_memcpy(&f, &lo, 4);
        const unsigned long long u = ((unsigned long long)hi << 32) | lo;
        double d; __builtin_memcpy(&d, &u, 8);
        return isd ? d : (double)f;
    }
}

// What the IFT column sweeps publish per supernode in LDS (quad mapping): the up-sweep's root phase reads other supernodes' entries
// (any quad of the environment forward-substitutes a batch on a root's behalf, gradient_columns_quad).  bm = the batches whose
// forward-substituted right-hand side is non-zero on this supernode (bit b; topology only); topid = rank of a level-1 supernode
// among the level-1 supernodes of the environment.
struct SweepInfo { int level, parent, u_off, myu, nlim_r, ncontact, nchild, topid; int child[MAXCH]; unsigned char contact[8]; double wk; unsigned long long bm[2]; };

// ================================================================================================
// The lane program
// ================================================================================================
// QUAD = false: one lane per supernode.  QUAD = true: four lanes per supernode (roles q = 0..3 own
// rows 3q..3q+2 of the supernode system); all other per-lane state is replicated inside the quad.
template <class T, class TL, int MAXC, bool QUAD, class Wave>
struct LaneProgram {
    Wave& wv;
    const Globals<T>& G;
    const NodeP<T>& P;
    const ContactP<T>* CP;       // contact table
    int base;                    // first lane of this environment inside the wave
    int k;                       // node index
    bool active;                 // lane maps to a real (env, body)
    bool has_parent;
    int plane;                   // wave lane of the parent (or own lane)
    Lane<T, MAXC>& L;            // per lane, or one copy per supernode in LDS (lock-step quad mapping)
    Factors<T, TL, MAXC, QUAD> F;
    void* gb_lds = nullptr;      // LDS home of this supernode's IFT right-hand sides (QuadRhs / ConRhs, quad mapping)
    // LDS mailbox (quad mapping): one slot of MAIL_N doubles per (supernode, body-row role).  Tree neighbours exchange
    // their 3-row pieces through it with wide ds_read/ds_write instead of one ds_bpermute per dword.
    enum { MAIL_N = 18, MAIL_STRIDE = 20 };
    double* mail = nullptr;
    double* qred = nullptr;            // [3][supernode slots]: reduction scratch (quad mapping)
    SweepInfo* sinfo = nullptr;        // [supernode slots of the workgroup] (IFT kernels, quad mapping)
    T* msg = nullptr;                  // this ENVIRONMENT's block of KernelArgs::msg (IFT kernels, quad mapping)
    DJ_HD double* mail_slot(int supernode_lane0, int role) const { return mail + (size_t)((supernode_lane0 >> 2) * 2 + role) * MAIL_STRIDE; }
    // Reduction over the supernodes of this lane's environment of NV values that the four lanes of a supernode hold
    // identically (quad mapping): one LDS write per supernode, then every lane folds the S values of its environment --
    // instead of log2(4S) butterfly steps of two ds_bpermute each.
    template <int NV, class OP> DJ_HD void env_reduce_quad(T (&v)[NV], OP op) {
        if constexpr (Wave::kWaveReduce) {
            if (G.S == 16 * Wave::kWaves) {                    // (uniform) the environment is the workgroup: Wave::reduce_quads16 per wavefront, no LDS loop
                static_assert(NV * Wave::kWaves <= 8, "Wave::red_ holds eight doubles");
                wv.template reduce_slots<NV>(v, op);
                return;
            }
        }
        const int nsn = wv.width() >> 2;
        wv.sync();
        if (q == 0) {
#pragma unroll
            for (int n = 0; n < NV; ++n) qred[n * nsn + (wv.lane() >> 2)] = (double)v[n];
        }
        wv.sync();
        const int s0 = base >> 2;
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            T r = T(qred[n * nsn + s0]);
            for (int i = 1; i < G.S; ++i) r = op(r, T(qred[n * nsn + s0 + i]));
            v[n] = r;
        }
    }
    // the same reduction (max) for values that differ between the four lanes of a quad: folded inside the quad first
    template <int NV> DJ_HD void env_reduce_quad_all(T (&v)[NV]) {
#pragma unroll
        for (int n = 0; n < NV; ++n) { v[n] = tmax(v[n], T(wv.quad_xor(v[n], 1))); v[n] = tmax(v[n], T(wv.quad_xor(v[n], 2))); }
        env_reduce_quad<NV>(v, [](T a_, T b_) { return a_ > b_ ? a_ : b_; });
    }
    // the two body-row roles of every supernode post N values ...
    template <int N, class TV> DJ_HD void mail_post_roles(const TV* v) {
        wv.sync();
        if (q < 2) { double* ms_ = mail_slot(qb, q);
#pragma unroll
            for (int i = 0; i < N; ++i) ms_[i] = (double)v[i]; }
        wv.sync();
    }
    // ... and a parent adds up what the same role of each of its children posted.
    // Two LDS round trips for all children (round 6): the children's mailbox slots come from ONE word (a byte per child: the supernode slot
    // inside the workgroup; 0xFF: none) kept behind the supernode's NodeP in LDS (NodeSlot::pad_), and the posts of all MAXCH children are
    // read before the first is used -- a missing child reads the quad's own post and is not added.  The plain loop (`ci < maxch` and
    // `ci < P.nchild` in front of every child, the child's index read from LDS, then its post) is three dependent round trips per child
    // with the one wave of the SIMD waiting through each.  The order of the additions is the order of the children, as before.  Only the
    // three-value exchange of the solves' forward sweeps takes this path: with six or eighteen values per child in flight, or with the word
    // in a register, the step kernel's register allocation tips over (6 -> 77 spilled registers) and the gain is gone (profiles/r06_b_ab.txt).
    DJ_HD int& chpack_ref() const { return *(int*)((char*)&P + sizeof(NodeP<T>)); }
    DJ_HD void chpack_init() {
        if constexpr (QUAD && Wave::kLockstep) {
            unsigned pk = 0xFFFFFFFFu;
            if (active) for (int ci = 0; ci < MAXCH; ++ci) if (ci < P.nchild) pk = (pk & ~(0xFFu << (8 * ci))) | ((unsigned)(((base + stride * P.child[ci]) >> 2) & 0xFF) << (8 * ci));
            wv.sync();
            if (q == 0) chpack_ref() = (int)pk;
            wv.sync();
        }
    }
    template <int N, class TV> DJ_HD void mail_add_children(TV* acc, bool act, int maxch) {
        if constexpr (N == 3 && Wave::kLockstep) {
            (void)maxch;
            double got[MAXCH][N]; bool use[MAXCH];
            const int chpack = chpack_ref();
#pragma unroll
            for (int ci = 0; ci < MAXCH; ++ci) {
                const int cs = (chpack >> (8 * ci)) & 0xFF;
                use[ci] = act && q < 2 && cs != 0xFF;
                const double* cs_ = mail + (size_t)((use[ci] ? cs : (qb >> 2)) * 2 + (q & 1)) * MAIL_STRIDE;
#pragma unroll
                for (int i = 0; i < N; ++i) got[ci][i] = cs_[i];
            }
#pragma unroll
            for (int ci = 0; ci < MAXCH; ++ci)
#pragma unroll
                for (int i = 0; i < N; ++i) acc[i] = use[ci] ? acc[i] + TV(got[ci][i]) : acc[i];
        } else {
        for (int ci = 0; ci < maxch; ++ci) {
            if (act && q < 2 && ci < P.nchild) {
                const double* cs_ = mail_slot(base + stride * P.child[ci], q);
#pragma unroll
                for (int i = 0; i < N; ++i) acc[i] += TV(cs_[i]);
            }
        }
        }
    }
    template <int N, class TV> DJ_HD void mail_add_children_node(TV* acc, bool act, int maxch) {   // all four lanes read the children's node posts
        for (int ci = 0; ci < maxch; ++ci) {
            if (act && ci < P.nchild) {
                const double* cs_ = mail_slot(base + stride * P.child[ci], 0);
#pragma unroll
                for (int i = 0; i < N; ++i) acc[i] += TV(cs_[i]);
            }
        }
    }
    // role 0 of every supernode posts N values that all four lanes hold identically; any lane can read a neighbour's post
    template <int N, class TV> DJ_HD void mail_post_node(const TV* v) {
        wv.sync();
        if (q == 0) { double* ms_ = mail_slot(qb, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) ms_[i] = (double)v[i]; }
        wv.sync();
    }
    int stride, q, envl, qb;     // lanes per supernode, role in the quad, lanes per environment, first lane of the quad
    Cold<T, MAXC>& cold;
    ContactCold<T>* cpool = nullptr;   // contact rows: slot (supernode, c) [pool_by_id = false] or slot = contact index of the environment
    bool pool_by_id = false; int pool_base = 0;
    // Newton step + line-search base iterate once per supernode in LDS (single-wave quad mapping with one contact per body: there is room)
    static constexpr bool kLsInLds = QUAD && Wave::kLockstep && Wave::kWaves == 1 && MAXC == 1;
    char* ls_lds = nullptr;
    char* lane_slots = nullptr; int lane_slot_stride = 0;   // lock-step quad mapping: the Lane blocks of all supernodes of the workgroup (parents are read in place)
    DJ_HD const Lane<T, MAXC>& parent_state() const { return *(const Lane<T, MAXC>*)(lane_slots + (size_t)((has_parent ? base + stride * P.parent : qb) >> 2) * lane_slot_stride); }
    DJ_HD ContactCold<T>& ccold(int c) const { return cpool[pool_by_id ? P.contact[c] : pool_base + c]; }
    JointCfg<T>& cfg;
    T mu;                        // mechanism.μ
#if DJ_TSD
    const TraSD<T>* tsd = nullptr;   // translational spring / damper / limit of the parent joint (null: the mechanism has none)
    bool tlim = false;               // the joint limit of this supernode sits on its translational coordinate (Δκ row = slot 8)
    DJ_HD bool lim_on() const { return P.nlim_r > 0 || tlim; }
    DJ_HD T lim_lo() const { return tlim ? tsd->lim_lo : P.lim_lo; }
    DJ_HD T lim_hi() const { return tlim ? tsd->lim_hi : P.lim_hi; }
#else
    static constexpr bool tlim = false;
    DJ_HD bool lim_on() const { return P.nlim_r > 0; }
    DJ_HD T lim_lo() const { return P.lim_lo; }
    DJ_HD T lim_hi() const { return P.lim_hi; }
#endif
    // ---- kinematic loops: loop-closing ("cut") joints (DJ_CUT builds, lane mapping) ----
    // The tree elimination handles the spanning tree K_t.  A cut joint c between bodies a and b (both in the tree) adds, in the unknowns
    // z_c = [Δw_a(6); Δw_b(6); Δλ_c(6)], the 18 x 18 block M_c = [N_c (rows of the two bodies); Q_c (the joint's rows)]:
    //     K_t x + Σ_c S_cᵀ N_c z_c = r,     Q_c z_c = r_c,     z_c[0:12] = S_c x   (S_c picks the velocities of a and b)
    // With W = K_t⁻¹ Sᵀ (12 columns per cut: tree solves with unit right-hand sides, once per linearization) and H = S W:
    //     x = K_t⁻¹ r − W (N z),      (I + H N) z|bodies + ... = S K_t⁻¹ r,      Q z = r_c        (18·ncut unknowns, dense LU with pivoting)
    // -- the exact solution of the cyclic system, which the reference reaches through the fill-in of its LDU (src/solver/linear_system.jl:4-5).
    // Every lane of the environment holds the small system (the values arrive through wave shuffles) and solves it redundantly.
    static constexpr bool kCut = DJ_CUT != 0 && !QUAD;
    const NodeP<T>* cutp = nullptr; int ncut = 0;
#if DJ_CUT
    T clam[NCUT][6];                  // multipliers of the cut joints (3 translational, 3 rotational slots; identical on all lanes of the environment)
    T crj[NCUT][6];                   // their residual rows at the last evaluation
    T cue[NCUT][6];                   // their control inputs of this step
    T crc[NCUT][6];                   // right-hand sides of their rows in the solve in progress
    JointCfg<T> ccfg[NCUT];           // joint_cfg at (x2, q2) of the two bodies (the owner lane = body b's lane uses it)
    T cW[NCUT * 12][12];              // this lane's rows of W
    // M_c of the last linearization, H and the factored small system live in global memory, once per environment (KernelArgs::cutws; as private
    // arrays they were 23 KB of scratch per lane): the owner lanes write M_c, lane 0 of the environment writes H and factorizes, every lane reads
    T* cws = nullptr;
    DJ_HD T& gM(int c, int r, int j) const { return cws[(c * 18 + r) * 18 + j]; }
    DJ_HD T& gH(int i, int j) const { return cws[CUTWS_H + i * (12 * NCUT) + j]; }
    DJ_HD T& gLU(int i, int j) const { return cws[CUTWS_LU + i * (18 * NCUT) + j]; }
    DJ_HD T& gpiv(int i) const { return cws[CUTWS_PIV + i]; }
    DJ_HD int cut_a(int c) const { return cutp[c].parent; }
    DJ_HD int cut_b(int c) const { return cutp[c].child[0]; }
    DJ_HD bool cut_owner(int c) const { return active && k == cut_b(c); }
    // A body-body contact (SphereSphereCollision) between two bodies that are no tree neighbours is a cut element as well: entry c of
    // KernelArgs::cuts with ncontact = 1, contact[0] = its index, parent = the contact's parent body a, child[0] = its child body b (the owner:
    // the contact sits in b's contact list with its cone variables, line searches and centering like any other contact of b).  It has no
    // multipliers of its own -- the cone rows are condensed as for every contact -- so its M_c carries the identity in the λ block.
    DJ_HD bool cut_is_contact(int c) const { return cutp[c].ncontact == 1; }
    DJ_HD int cut_of_contact(int cid) const { for (int c = 0; c < NCUT; ++c) if (c < ncut && cut_is_contact(c) && cutp[c].contact[0] == cid) return c; return -1; }
    T cPc[NCUT][36];                  // −∂(contact impulse on body a)/∂(v_a, ω_a) of a cut contact at the last Jacobian evaluation (owner lane)
    // body a's (x2, q2, v, w) on every lane
    DJ_HD void cut_fetch_a(int c, T* xa, T* qa, T* va, T* wa_) {
        const int la = base + cut_a(c);
        T own13[13] = {L.x2[0], L.x2[1], L.x2[2], L.q2[0], L.q2[1], L.q2[2], L.q2[3], L.v[0], L.v[1], L.v[2], L.w[0], L.w[1], L.w[2]}, o13[13];
        shfl_vec<13>(wv, o13, own13, la);
        for (int i = 0; i < 3; ++i) { xa[i] = o13[i]; va[i] = o13[7 + i]; wa_[i] = o13[10 + i]; }
        for (int i = 0; i < 4; ++i) qa[i] = o13[3 + i];
    }
    // set_input! / springs of the cut joints (begin_step's part for the joints that are not a supernode's own)
    DJ_HD void cut_begin() {
        for (int c = 0; c < NCUT; ++c) { for (int i = 0; i < 6; ++i) clam[c][i] = crj[c][i] = T(0); }
        for (int c = 0; c < NCUT; ++c) if (c < ncut && !cut_is_contact(c)) {
            const NodeP<T>& Pc = cutp[c];
            T xa[3], qa[4], va[3], wa_[3];
            cut_fetch_a(c, xa, qa, va, wa_);
            T ina[6] = {0, 0, 0, 0, 0, 0};
            if (cut_owner(c)) {
                joint_cfg(ccfg[c], Pc, xa, qa, L.x2, L.q2);
                T it[3] = {0, 0, 0}, ir[3] = {0, 0, 0};
                for (int i = 0; i < 3; ++i) {
                    if (i < Pc.nu_t) for (int kx = 0; kx < 3; ++kx) it[kx] += Pc.At[3 * i + kx] * cue[c][i];
                    if (i < Pc.nu_r) for (int kx = 0; kx < 3; ++kx) ir[kx] += Pc.Ar[3 * i + kx] * cue[c][Pc.nu_t + i];
                }
                for (int kx = 0; kx < 3; ++kx) { it[kx] *= G.input_scaling; ir[kx] *= G.input_scaling; }
                T ia[6], ib[6];
                tra_impulse(ia, ib, ccfg[c], Pc, it);
                for (int i = 0; i < 3; ++i) { ina[i] = ia[i]; ina[3 + i] = T(0.5) * ia[3 + i]; L.dconst[i] -= ib[i]; L.dconst[3 + i] -= T(0.5) * ib[3 + i]; }
                T ta[3], tb[3];
                m3vec(ta, ccfg[c].Roff, ir); m3vec(tb, ccfg[c].Rba, ta);
                for (int i = 0; i < 3; ++i) { ina[3 + i] += -ta[i]; L.dconst[3 + i] -= tb[i]; }
                T sa[6], sb[6];
                spring_impulses(sa, sb, Pc, ccfg[c], G.dt);
                for (int i = 0; i < 6; ++i) { L.dconst[i] -= sb[i]; ina[i] += sa[i]; }
            }
            T got[6];
            shfl_vec<6>(wv, got, ina, base + cut_b(c));
            if (active && k == cut_a(c)) for (int i = 0; i < 6; ++i) L.dconst[i] -= got[i];
        }
    }
    // residual (and, JAC, M_c) of the cut joints; d = this lane's body residual under construction
    template <bool JAC>
    DJ_HD void cut_eval(T* d, const Kin<T>& kb, const T (*cimp_p)[6]) {
        for (int c = 0; c < NCUT; ++c) if (c < ncut) {
            const NodeP<T>& Pc = cutp[c];
            if (cut_is_contact(c)) {             // a cut contact: what it applies to body a (evaluated by the owner in the contact loop of evaluate())
                T got[6];
                shfl_vec<6>(wv, got, cimp_p[c], base + cut_b(c));
                if (active && k == cut_a(c)) for (int i = 0; i < 6; ++i) d[i] -= got[i];
                for (int i = 0; i < 6; ++i) crj[c][i] = T(0);
                continue;
            }
            T xa[3], qa[4], va[3], wa_[3];
            cut_fetch_a(c, xa, qa, va, wa_);
            T ia6[6] = {0, 0, 0, 0, 0, 0}, g6[6] = {0, 0, 0, 0, 0, 0};
            if (cut_owner(c)) {
                Kin<T> ka;
                kin_of(ka, xa, qa, va, wa_, G.dt);
                JointEval<T> E;
                const T lg0[2] = {0, 0};
                if (JAC) {
                    FullBlocks<T> Kc;
                    Kc.zero();
                    joint_eval<1>(E, Pc, ccfg[c], true, ka, kb, wa_, L.w, clam[c], lg0, G.dt, Kc);
                    // M_c: rows [body a; body b; joint] x columns [w_a; w_b; λ] out of the blocks joint_eval fills for a (parent, child) pair
                    for (int r = 0; r < 6; ++r) for (int j = 0; j < 6; ++j) {
                        gM(c, r, j) = Kc.D[6 * r + j]; gM(c, r, 6 + j) = Kc.L[12 * r + j]; gM(c, r, 12 + j) = Kc.L[12 * r + 6 + j];
                        gM(c, 6 + r, j) = Kc.U[6 * r + j]; gM(c, 6 + r, 6 + j) = Kc.S[12 * r + j]; gM(c, 6 + r, 12 + j) = Kc.S[12 * r + 6 + j];
                        gM(c, 12 + r, j) = Kc.U[6 * (6 + r) + j]; gM(c, 12 + r, 6 + j) = Kc.S[12 * (6 + r) + j];
                        gM(c, 12 + r, 12 + j) = Kc.S[12 * (6 + r) + 6 + j] + ((r == j) ? ((r < 3 ? r < Pc.nl_t : r - 3 < Pc.nl_r) ? T(REG) : T(1)) : T(0));
                    }
                } else { NullBlocks nk; joint_eval<0>(E, Pc, ccfg[c], true, ka, kb, wa_, L.w, clam[c], lg0, G.dt, nk); }
                for (int i = 0; i < 6; ++i) { d[i] -= E.imp_b[i]; ia6[i] = E.imp_a[i]; g6[i] = E.g[i]; }
            }
            T got[6];
            shfl_vec<6>(wv, got, ia6, base + cut_b(c));
            if (active && k == cut_a(c)) for (int i = 0; i < 6; ++i) d[i] -= got[i];
            shfl_vec<6>(wv, crj[c], g6, base + cut_b(c));
        }
    }
#if DJ_SS
    // M_c of the cut contacts, after evaluate<true> and the condensation (the coefficients of Δγ = k0 + coef (C Δw_b + Cp Δw_a) belong to the
    // current cone variables): rows of body a: −Gpᵀ coef Cp − ∂(impulse on a)/∂(v_a, ω_a) | −Gpᵀ coef C;  rows of body b: −Gᵀ coef Cp | (own block: in the tree)
    DJ_HD void cut_contacts_M() {
        for (int c = 0; c < NCUT; ++c) if (c < ncut && cut_is_contact(c) && cut_owner(c)) {
            int lc = 0;
            for (int i = 0; i < MAXC; ++i) if (i < P.ncontact && P.contact[i] == cutp[c].contact[0]) lc = i;
            CCoef Q; T rc0[NCV] = {}, r580[NCV] = {};
            contact_coef(Q, lc, rc0, r580);
            const ContactCold<T>& cc_ = ccold(lc);
            for (int r = 0; r < 6; ++r) for (int j = 0; j < 6; ++j) {
                T aa = T(0), ab = T(0), ba = T(0);
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
                    const T cf = Q.coef[3 * a + b];
                    aa += cc_.Gp134[6 * a + r] * cf * cc_.Cp134[6 * b + j];
                    ab += cc_.Gp134[6 * a + r] * cf * cc_.C134[6 * b + j];
                    ba += cc_.G134[6 * a + r] * cf * cc_.Cp134[6 * b + j];
                }
                gM(c, r, j) = -aa - cPc[c][6 * r + j]; gM(c, r, 6 + j) = -ab; gM(c, 6 + r, j) = -ba; gM(c, 6 + r, 6 + j) = T(0);
                gM(c, r, 12 + j) = gM(c, 6 + r, 12 + j) = gM(c, 12 + r, j) = gM(c, 12 + r, 6 + j) = T(0);
                gM(c, 12 + r, 12 + j) = (r == j) ? T(1) : T(0);
            }
        }
    }
#endif
    DJ_HD int cut_body(int c, int row) const { return row < 6 ? cut_a(c) : cut_b(c); }     // the body behind row / column 0..11 of cut c
    // after the tree factorization: W, H and the LU of the small system
    DJ_HD void cut_factor() {
        const int n = 18 * ncut;
        for (int c = 0; c < NCUT; ++c) if (c < ncut) for (int col = 0; col < 12; ++col) {
            T rk[12], up[6] = {0, 0, 0, 0, 0, 0}, dk[12], dva[6];
            for (int i = 0; i < 12; ++i) rk[i] = T(0);
            if (active && k == cut_body(c, col)) rk[col % 6] = T(1);
            core_solve(rk, up, dk, dva);
            for (int i = 0; i < 12; ++i) cW[12 * c + col][i] = dk[i];
            for (int e = 0; e < NCUT; ++e) if (e < ncut) for (int row = 0; row < 12; ++row) {
                const T h_ = wv.shfl(dk[row % 6], base + cut_body(e, row));
                if (active && k == 0) gH(12 * e + row, 12 * c + col) = h_;
            }
        }
        wv.sync_mem();                                            // (M_c from the owner lanes, H from lane 0)
        if (active && k == 0) {                                   // one lane per environment: I + H M with the elements' own rows below, LU with partial pivoting in place
            for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) gLU(i, j) = T(0);
            for (int c = 0; c < NCUT; ++c) if (c < ncut) {
                for (int i = 0; i < 12; ++i) {
                    for (int e = 0; e < NCUT; ++e) if (e < ncut) for (int j = 0; j < 18; ++j) {
                        T a_ = (e == c && j == i) ? T(1) : T(0);
                        for (int m = 0; m < 12; ++m) a_ += gH(12 * c + i, 12 * e + m) * gM(e, m, j);
                        gLU(18 * c + i, 18 * e + j) = a_;
                    }
                }
                for (int i = 0; i < 6; ++i) for (int j = 0; j < 18; ++j) gLU(18 * c + 12 + i, 18 * c + j) = gM(c, 12 + i, j);
            }
            for (int kk = 0; kk < n; ++kk) {
                int p = kk; T best = tabs(gLU(kk, kk));
                for (int i = kk + 1; i < n; ++i) { const T v_ = tabs(gLU(i, kk)); if (v_ > best) { best = v_; p = i; } }
                gpiv(kk) = T(p);
                // (columns kk.. only: cut_solve applies row exchange kk right before elimination step kk, so the multipliers of the earlier steps stay where they were stored)
                if (p != kk) for (int j = kk; j < n; ++j) { const T t_ = gLU(kk, j); gLU(kk, j) = gLU(p, j); gLU(p, j) = t_; }
                const T ip = T(1) / gLU(kk, kk);
                for (int i = kk + 1; i < n; ++i) {
                    const T f_ = gLU(i, kk) * ip; gLU(i, kk) = f_;
                    for (int j = kk + 1; j < n; ++j) gLU(i, j) -= f_ * gLU(kk, j);
                }
            }
        }
        wv.sync_mem();
    }
    // the tree solve followed by the correction for the cut joints; rc = right-hand sides of their rows, dlc = their multipliers' part of the solution
    DJ_HD void cut_solve(const T* rk, const T* up, const T (*rc)[6], T* dk, T* dva, T (*dlc)[6]) {
        core_solve(rk, up, dk, dva);
        if (ncut == 0) return;
        const int n = 18 * ncut;
        T z[18 * NCUT];
        for (int c = 0; c < NCUT; ++c) if (c < ncut) {
            for (int row = 0; row < 12; ++row) z[18 * c + row] = wv.shfl(dk[row % 6], base + cut_body(c, row));
            for (int i = 0; i < 6; ++i) z[18 * c + 12 + i] = rc[c][i];
        }
        if (active) {                                             // (every lane of the environment reads the same factors: broadcast loads)
            for (int kk = 0; kk < n; ++kk) { const int p = (int)gpiv(kk); if (p != kk) { const T t_ = z[kk]; z[kk] = z[p]; z[p] = t_; } for (int i = kk + 1; i < n; ++i) z[i] -= gLU(i, kk) * z[kk]; }
            for (int i = n - 1; i >= 0; --i) { T a_ = z[i]; for (int j = i + 1; j < n; ++j) a_ -= gLU(i, j) * z[j]; z[i] = a_ / gLU(i, i); }
            for (int c = 0; c < NCUT; ++c) if (c < ncut) {
                for (int m = 0; m < 12; ++m) {
                    T f_ = T(0);
                    for (int j = 0; j < 18; ++j) f_ += gM(c, m, j) * z[18 * c + j];
                    for (int i = 0; i < 12; ++i) dk[i] -= cW[12 * c + m][i] * f_;
                }
                for (int i = 0; i < 6; ++i) dlc[c][i] = z[18 * c + 12 + i];
            }
        } else { for (int c = 0; c < NCUT; ++c) for (int i = 0; i < 6; ++i) dlc[c][i] = T(0); }
        T own6[6] = {dk[0], dk[1], dk[2], dk[3], dk[4], dk[5]}, par6[6];
        shfl_vec<6>(wv, par6, own6, plane);
        for (int i = 0; i < 6; ++i) dva[i] = has_parent ? par6[i] : T(0);
    }
#endif
    // ---- joint limits on several coordinates / both halves (DJ_MLIM builds, lane mapping) ----
    static constexpr bool kMLim = DJ_MLIM != 0 && !QUAD;
    const MLimP<T>* mlp = nullptr;     // this supernode's entry of KernelArgs::mlim
    DJ_HD int nlm() const { if constexpr (kMLim) return mlp != nullptr ? mlp->nt + mlp->nr : 0; else return 0; }
    DJ_HD int mslot(int m) const { return m < mlp->nt ? 6 + P.nl_t + m : 9 + P.nl_r + (m - mlp->nt); }     // the Δκ row of limited coordinate m
#if DJ_MLIM
    T mtheta[NLM];                     // the limited minimal coordinates at the last evaluation
    T mrs[NLM][2];                     // right-hand sides of the slack rows (up, lo) of the solve in progress
    T mth_a[NLM][6], mth_b[NLM][6];    // ∂θ/∂(v, ω) of parent / child   (Jacobian evaluations)
    T mt_a[NLM][6], mt_b[NLM][6];      // what a unit of net limit impulse κ = γ_lo − γ_up applies to parent / child
    // MODE 0: θ and the limit impulses (added to imp_a / imp_b); 1: + ∂θ/∂(velocities), impulse directions; 2: raw ∂θ/∂(x3, φ3) into rx_a / rx_b as well
    template <int MODE>
    DJ_HD void mlim_eval(T* imp_a, T* imp_b, const JointCfg<T>& c_, const Kin<T>& ka, const Kin<T>& kb, T dt, T (*rx_a)[6] = nullptr, T (*rx_b)[6] = nullptr) {
        const int nt = mlp->nt, nr = mlp->nr;
        // translational coordinates: θ = A_l·e(x3, q3), translational/minimal.jl:56-61 (as tra_limit_eval, one row of the nullspace mask each)
        if (nt > 0) {
            T t3[3], wv_[3], u[3];
            m3vec(t3, kb.R3, P.pb);
            for (int i = 0; i < 3; ++i) wv_[i] = kb.x3[i] + t3[i] - ka.x3[i];
            m3tvec(u, ka.R3, wv_);
            T RaT_Rb[9], Spb[9], Eb[9], Su[9], SuP[9], EbP[9];
            if (MODE >= 1) {
                m3tmul(RaT_Rb, ka.R3, kb.R3); m3skew(Spb, P.pb); m3mul(Eb, RaT_Rb, Spb);
                for (int i = 0; i < 9; ++i) Eb[i] *= T(-2);
                m3skew(Su, u);
                for (int i = 0; i < 9; ++i) Su[i] *= T(2);
                m3mul(SuP, Su, ka.Phi); m3mul(EbP, Eb, kb.Phi);
            }
            for (int l = 0; l < 3; ++l) if (l < nt) {
                const T* a = &P.At[3 * l];
                mtheta[l] = a[0] * (u[0] - P.pa[0]) + a[1] * (u[1] - P.pa[1]) + a[2] * (u[2] - P.pa[2]);
                const T kk = L.mlg[l][1] - L.mlg[l][0];              // projector [0; −A; A; C]ᵀ, joint.jl:92
                T p[3] = {a[0] * kk, a[1] * kk, a[2] * kk}, ia[6], ib[6];
                tra_impulse(ia, ib, c_, P, p);
                for (int i = 0; i < 6; ++i) { imp_a[i] += ia[i]; imp_b[i] += ib[i]; }
                if (MODE >= 1) {
                    for (int j = 0; j < 3; ++j) {
                        const T aRaT = a[0] * ka.R3[3 * j] + a[1] * ka.R3[3 * j + 1] + a[2] * ka.R3[3 * j + 2];
                        mth_a[l][j] = -dt * aRaT; mth_b[l][j] = dt * aRaT;
                        mth_a[l][3 + j] = a[0] * SuP[j] + a[1] * SuP[3 + j] + a[2] * SuP[6 + j];
                        mth_b[l][3 + j] = a[0] * EbP[j] + a[1] * EbP[3 + j] + a[2] * EbP[6 + j];
                        if (MODE == 2) {
                            rx_a[l][j] = -aRaT; rx_b[l][j] = aRaT;
                            rx_a[l][3 + j] = a[0] * Su[j] + a[1] * Su[3 + j] + a[2] * Su[6 + j];
                            rx_b[l][3 + j] = a[0] * Eb[j] + a[1] * Eb[3 + j] + a[2] * Eb[6 + j];
                        }
                    }
                    tra_impulse(mt_a[l], mt_b[l], c_, P, a);
                }
            }
        }
        // rotational coordinates: θ = A_l·rotation_vector(qoff⁻¹ qa⁻¹ qb), rotational/minimal.jl:4-11, 69-80
        if (nr > 0) {
            T qab[4], qr[4], rv[3];
            qcmul(qab, ka.q3, kb.q3);
            qcmul(qr, P.qoff, qab);
            rotvec(rv, qr);
            for (int l = 0; l < 3; ++l) if (l < nr) {
                const int m = nt + l;
                const T* a = &P.Ar[3 * l];
                mtheta[m] = v3dot(a, rv);
                const T kk = L.mlg[m][1] - L.mlg[m][0];
                T p[3] = {a[0] * kk, a[1] * kk, a[2] * kk}, ja[6], jb[6];
                rot_impulse(ja, jb, c_, p);
                for (int i = 0; i < 6; ++i) { imp_a[i] += ja[i]; imp_b[i] += jb[i]; }
                if (MODE >= 1) {
                    T rho[4], rl[4], rr[4], tb[3], ta0[3], ta[3];
                    rotvec_jac_row(rho, a, qr);
                    rowL(rl, rho, qr); rowR(rr, rho, qr);
                    for (int j = 0; j < 3; ++j) { tb[j] = rl[1 + j]; ta0[j] = -rr[1 + j]; }
                    m3vec(ta, c_.Roff, ta0);
                    T thb[3], tha[3];
                    m3tvec(thb, kb.Phi, tb); m3tvec(tha, ka.Phi, ta);
                    for (int j = 0; j < 3; ++j) {
                        mth_a[m][j] = T(0); mth_b[m][j] = T(0); mth_a[m][3 + j] = tha[j]; mth_b[m][3 + j] = thb[j];
                        if (MODE == 2) { rx_a[m][j] = T(0); rx_b[m][j] = T(0); rx_a[m][3 + j] = ta[j]; rx_b[m][3 + j] = tb[j]; }
                    }
                    rot_impulse(mt_a[m], mt_b[m], c_, a);
                }
            }
        }
    }
#endif
#ifdef DJ_DEBUG
    T* dbg = nullptr; bool dbg_on = false; bool trace = false;
#endif
#ifdef DJ_PROF
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0 = 0;    // cycle counters per phase (profiling builds only)
#define DJ_PB() (pt0 = wv.clock())
#define DJ_PE(i) (pc[i] += wv.clock() - pt0)
#else
#define DJ_PB() ((void)0)
#define DJ_PE(i) ((void)0)
#endif
    // DJ_PROF2: finer cycle counters of the Newton loop (tools/gpu_probe.py phases2).  The counters live in LDS (StepLds has no room: a block of
    // the kernel's own), lane 0 adds the uniform differences: no register stays live for them.  Every probe is a full LDS wait (s_memtime
    // returns through lgkmcnt), so the sum of the parts is somewhat longer than the uninstrumented loop.
#ifdef DJ_PROF2
    unsigned long long* p2c = nullptr; unsigned long long pt1 = 0;
#define DJ_P2B() (pt1 = wv.clock())
#define DJ_P2E(i) do { const unsigned long long t_ = wv.clock(); if (wv.lane() == 0) p2c[i] += t_ - pt1; pt1 = t_; } while (0)
#else
#define DJ_P2B() ((void)0)
#define DJ_P2E(i) ((void)0)
#endif
    // Bodies with four or more contacts (Block, Atlas' feet; MAXC >= 4 builds of the quad mapping): the contacts are split over the four
    // lanes of the quad -- lane q handles contacts q, q + 4, ... -- instead of every lane doing all of them.  What a contact contributes
    // to the whole supernode (impulse on the body, its curvature term, the condensed right-hand side, its share of the centering sums) is
    // added up inside the quad with two DPP steps; maxima / minima over contacts fold the same way; the per-contact cone variables
    // live once per supernode in LDS (Lane::cs, cg), every lane updating its own contacts' entries.  With one contact per body nothing
    // changes (the lane-local arrays have MAXC entries, cidx(i) = i).
    static constexpr bool kSS = DJ_SS != 0 && QUAD && MAXC == 1;   // body-body contacts: the single-contact quad builds (the host refuses them elsewhere)
    static constexpr bool kSplitC = QUAD && MAXC >= 4;
    static constexpr int CPL = kSplitC ? MAXC / 4 : MAXC;      // contact slots per lane
    typedef Step<T, CPL> StepT;
    typedef SolSnap<T, CPL> SnapT;
    DJ_HD int cidx(int li) const { return kSplitC ? q + 4 * li : li; }      // contact (slot of the supernode) behind this lane's local slot li
    template <int N> DJ_HD void quad_sum(T (&v)[N]) {
        if constexpr (kSplitC) {
#pragma unroll
            for (int n = 0; n < N; ++n) v[n] += T(wv.quad_xor(v[n], 1));
#pragma unroll
            for (int n = 0; n < N; ++n) v[n] += T(wv.quad_xor(v[n], 2));
        }
    }
    DJ_HD T quad_maxv(T v) { if constexpr (kSplitC) { v = tmax(v, T(wv.quad_xor(v, 1))); v = tmax(v, T(wv.quad_xor(v, 2))); } return v; }
    DJ_HD T quad_minv(T v) { if constexpr (kSplitC) { v = tmin(v, T(wv.quad_xor(v, 1))); v = tmin(v, T(wv.quad_xor(v, 2))); } return v; }
    // SIMT emulator (its lanes are threads with private copies of what the GPU keeps once per supernode in LDS): after a lane has
    // changed its own contacts' entries the other three lanes of the quad get them
    DJ_HD void share_cone_state() {
        if constexpr (kSplitC && !Wave::kLockstep) {
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
#pragma unroll
                for (int i = 0; i < NCV; ++i) { L.cs[c][i] = T(wv.quad_bcast(L.cs[c][i], c & 3)); L.cg[c][i] = T(wv.quad_bcast(L.cg[c][i], c & 3)); }
        }
    }
    DJ_HD void share_contact_rows() {
        if constexpr (kSplitC && !Wave::kLockstep) {
#pragma unroll
            for (int c = 0; c < MAXC; ++c) { ContactCold<T>& cc_ = ccold(c);
                for (int i = 0; i < 18; ++i) { cc_.C134[i] = T(wv.quad_bcast(cc_.C134[i], c & 3)); cc_.G134[i] = T(wv.quad_bcast(cc_.G134[i], c & 3)); } }
        }
    }
    // residual pieces of the last evaluation
    T rb[6], rj[6], theta, cres[CPL][NCV];
    // Iterative refinement of the linear solves (DJ_REFINE, quad mapping).  `refine` is a per-environment flag (sticky once the
    // cones are stiff); the un-factored, contact-UNcondensed supernode rows of the last linearization live in global memory
    // (KernelArgs::blk, [workgroup][BLK_PER_LANE][lanes]: lane index fastest), written only while some environment of the
    // workgroup refines.
    // Two builds of the kernels share this program: the plain one only TRACKS the stiffness and hands stiff environments over
    // (DJ_STATUS_DEFERRED); the refining one (Wave::kRefine) re-solves exactly those -- so the plain kernels' register
    // allocation never sees the refinement code.
    static constexpr bool kTrack = DJ_REFINE && QUAD;
    static constexpr bool kRefine = kTrack && Wave::kRefine;
    enum { BLK_PER_LANE = 90 };        // S rows 3x12, U rows 3x6, L columns 6x3, Dup rows 3x6
    T* blk = nullptr; int blk_stride = 0;
    T* ypark = nullptr;                // this lane's first slot of the workgroup's ỹ park (KernelArgs::ypark: one slot per lane, all four roles park)
    bool refine = false;
    T wstiff = T(0);                   // max γ/s over the cones of the environment at the last evaluated iterate
    T growth = T(0);                   // largest |multiplier| of the last factorization's Gauss-Jordan passes on this lane (no pivoting: the growth indicator)
    T dk_lim = T(0);                   // Δκ of the last solve (the multiplier slot that carries the net limit impulse)

    DJ_HD LaneProgram(Wave& w, const Globals<T>& g, const NodeP<T>& p, const ContactP<T>* cp, int base_, int k_, int q_, bool act, Lane<T, MAXC>& lane_, Cold<T, MAXC>& cold_)
        : wv(w), G(g), P(p), CP(cp), base(base_), k(k_), active(act), L(lane_), cold(cold_), cfg(cold_.cfg) {
        stride = QUAD ? 4 : 1; q = QUAD ? q_ : 0; envl = stride * g.S; qb = w.lane() - q;
        has_parent = act && p.parent >= 0;
        plane = has_parent ? base + stride * p.parent + q : w.lane();
        mu = T(0);
    }

    // ---------------------------------------------------------------- residual (+ Jacobian blocks)
    // Evaluates at candidate index 1.  After the call: rb = full body residual d (including what the
    // children's joints apply to this body), rj, theta, cres.  When JAC, fills S/U/Lm/Daa_up etc.
    template <bool JAC, class BK>
    DJ_HD void evaluate(BK& K) {
        const T dt = G.dt;
        // parent's candidate velocity
        T va[3], wa[3], own6[6] = {L.v[0], L.v[1], L.v[2], L.w[0], L.w[1], L.w[2]}, par6[6];
        if constexpr (QUAD) {
            if (lane_slots) {                                    // the parent's candidate velocity, read in place
                wv.sync();
                const Lane<T, MAXC>& Lp = parent_state();
                for (int i = 0; i < 3; ++i) { par6[i] = Lp.v[i]; par6[3 + i] = Lp.w[i]; }
            } else {
                mail_post_node<6>(own6);
                const double* pp_ = mail_slot(has_parent ? base + stride * P.parent : qb, 0);
#pragma unroll
                for (int i = 0; i < 6; ++i) par6[i] = T(pp_[i]);
            }
        } else shfl_vec<6>(wv, par6, own6, plane);
        if (has_parent) { v3cpy(va, par6); v3cpy(wa, par6 + 3); } else { va[0] = va[1] = va[2] = wa[0] = wa[1] = wa[2] = T(0); }
        Kin<T> kb, ka;
        kin_of(kb, L.x2, L.q2, L.v, L.w, dt);
        kin_of(ka, L.xa2, L.qa2, va, wa, dt);
        DJ_P2E(JAC ? 15 : 9);
        JointEval<T> E;
        if (JAC) K.zero();
        joint_eval<JAC ? 1 : 0>(E, P, cfg, has_parent, ka, kb, wa, L.w, L.lam, L.lg, dt, K);
        DJ_P2E(JAC ? 16 : 10);
#if DJ_TSD
        if (tsd) tra_damper_eval<JAC>(E, P, *tsd, cfg, va, wa, L.v, L.w, dt, K);
        if (tlim) tra_limit_eval<JAC>(E, P, cfg, ka, kb, L.lg, dt);
#endif
#if DJ_MLIM
        if constexpr (kMLim) { if (nlm() > 0) mlim_eval<JAC ? 1 : 0>(E.imp_a, E.imp_b, cfg, ka, kb, dt); }
#endif
        for (int i = 0; i < 6; ++i) rj[i] = E.g[i];
        theta = E.theta;
        // body residual: src/integrators/constraint.jl:1-34 in closed form
        T Jw[3], wxJw[3];
        m3vec(Jw, P.J, L.w);
        v3cross(wxJw, L.w, Jw);
        T d[6];
        for (int i = 0; i < 3; ++i) { d[i] = P.m * L.v[i] + L.dconst[i]; d[3 + i] = T(0.5) * dt * (kb.c * Jw[i] + wxJw[i]) + L.dconst[3 + i]; }
        for (int i = 0; i < 6; ++i) d[i] -= E.imp_b[i];
        // contacts.  With several contacts per body (MAXC > 1) everything one contact contributes (impulse on the body, its
        // cone-row residuals, its curvature term of the ω block and the rows the condensation reads from LDS) is consumed
        // inside its own iteration, so that one ContactEval is live at a time (Atlas step kernel: 714 -> 202 spilled VGPRs,
        // +9 % env-steps/s; Block +12 %); with one contact the blocks are added after the body terms (the MAXC = 1 register
        // allocation is 1 % faster that way, same session).  joint_eval has zeroed / filled K by now.
        constexpr bool kEarly = MAXC > 1;
        ContactEval<T> CE[kEarly ? 1 : MAXC];
#if DJ_SS
        ContactEvalSS<T> CS[MAXC];                               // body-body contacts: the rows of the contact's parent body
        T upc[6] = {0, 0, 0, 0, 0, 0};                           // ... and what they apply to it
#endif
        T dcon[6] = {0, 0, 0, 0, 0, 0}, dww[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // (split contacts: this lane's share of the contact impulses / curvature terms)
#if DJ_CUT && DJ_SS
        // cut contacts (a body-body contact between two bodies that are no tree neighbours): the other body's (x2, q2, v, w) on every lane,
        // what the contact applies to it from the owner's evaluation below
        T cob[NCUT][13], cimp_p[NCUT][6];
        for (int e = 0; e < NCUT; ++e) { for (int i = 0; i < 6; ++i) cimp_p[e][i] = T(0); for (int i = 0; i < 13; ++i) cob[e][i] = T(0); }
        if constexpr (kCut) { for (int e = 0; e < NCUT; ++e) if (e < ncut && cut_is_contact(e)) {
            T xa_[3], qa_[4], va_[3], wa__[3];
            cut_fetch_a(e, xa_, qa_, va_, wa__);
            for (int i = 0; i < 3; ++i) { cob[e][i] = xa_[i]; cob[e][7 + i] = va_[i]; cob[e][10 + i] = wa__[i]; }
            for (int i = 0; i < 4; ++i) cob[e][3 + i] = qa_[i];
        } }
#elif DJ_CUT
        T cimp_p[NCUT][6];
        for (int e = 0; e < NCUT; ++e) for (int i = 0; i < 6; ++i) cimp_p[e][i] = T(0);
#endif
#pragma unroll
        for (int li = 0; li < CPL; ++li) {
            const int c = cidx(li);
            if (c < P.ncontact) {
                ContactEval<T>& CEc = CE[kEarly ? 0 : li];
                bool done_ss = false;                            // a cut contact: evaluated here against the other body's state
#if DJ_CUT && DJ_SS
                if constexpr (kCut) { if (CP[P.contact[c]].kind == 1) {
                    const int e = cut_of_contact(P.contact[c]);
                    if (e >= 0) {
                        Kin<T> kae; ContactEvalSS<T> CSe;
                        kin_of(kae, &cob[e][0], &cob[e][3], &cob[e][7], &cob[e][10], dt);
                        contact_eval_ss<JAC>(CEc, CSe, CP[P.contact[c]], kb, kae, L.v, L.w, &cob[e][7], &cob[e][10], L.cs[c], L.cg[c], dt, G.contact_model == 1);
                        for (int i = 0; i < 6; ++i) cimp_p[e][i] = CSe.imp_p[i];
                        if (JAC) {       // the own body's −∂(impulse)/∂(v, ω) beside Dww (contacts/constraints.jl:54-57); the other body's rows and curvature for its cut element
                            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
                                K.addS(i, j, -CSe.Sxx[3 * i + j]); K.addS(i, 3 + j, -CSe.Sxw[3 * i + j]); K.addS(3 + i, j, -CSe.Swx[3 * i + j]);
                                cPc[e][6 * i + j] = CSe.Sxx[3 * i + j]; cPc[e][6 * i + 3 + j] = CSe.Pxw[3 * i + j]; cPc[e][6 * (3 + i) + j] = CSe.Pwx[3 * i + j]; cPc[e][6 * (3 + i) + 3 + j] = CSe.Pww[3 * i + j];
                            }
                            ContactCold<T>& cc_ = ccold(c);
                            for (int i = 0; i < 18; ++i) { cc_.Cp134[i] = CSe.Cp134[i]; cc_.Gp134[i] = CSe.Gp134[i]; }
                        }
                        done_ss = true;
                    }
                } }
#endif
                if (!done_ss) {
#if DJ_SS
                if (kSS && CP[P.contact[c]].kind == 1) {
                    contact_eval_ss<JAC>(CEc, CS[li], CP[P.contact[c]], kb, ka, L.v, L.w, va, wa, L.cs[c], L.cg[c], dt, G.contact_model == 1);
                    for (int i = 0; i < 6; ++i) upc[i] += CS[li].imp_p[i];
                } else
#endif
                contact_eval<JAC>(CEc, CP[P.contact[c]], kb, L.v, L.w, L.cs[c], L.cg[c], dt);
                }
                if constexpr (kSplitC) { for (int i = 0; i < 6; ++i) dcon[i] += CEc.imp[i]; } else { for (int i = 0; i < 6; ++i) d[i] -= CEc.imp[i]; }
                for (int i = 0; i < NCV; ++i) cres[li][i] = CEc.c[i];
                if (!kLinear && G.contact_model == 1) cres[li][1] = T(0);       // ImpactContact: no friction rows (rows 3, 4 are 0 − 0 already)
                if (JAC && kEarly) {
                    if constexpr (kSplitC) { for (int i = 0; i < 9; ++i) dww[i] += CEc.Dww[i]; }
                    else {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) K.addS(3 + i, 3 + j, -CEc.Dww[3 * i + j]);
                    }
                    ContactCold<T>& cc_ = ccold(c);
                    for (int i = 0; i < 18; ++i) { cc_.C134[i] = CEc.C134[i]; cc_.G134[i] = CEc.G134[i]; }
                }
            } else { for (int i = 0; i < NCV; ++i) cres[li][i] = T(0); }
        }
        if constexpr (kSplitC) {
            quad_sum(dcon);
            for (int i = 0; i < 6; ++i) d[i] -= dcon[i];
            if (JAC) {
                quad_sum(dww);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) K.addS(3 + i, 3 + j, -dww[3 * i + j]);
                share_contact_rows();
            }
        }
        DJ_P2E(JAC ? 17 : 11);
        // what this lane's joint applies to the parent body travels up the tree
        T up[6];
        for (int i = 0; i < 6; ++i) up[i] = has_parent ? -E.imp_a[i] : T(0);
#if DJ_SS
        for (int i = 0; i < 6; ++i) up[i] -= has_parent ? upc[i] : T(0);
#endif
        if constexpr (QUAD) { mail_post_node<6>(up); mail_add_children_node<6>(d, active, G.maxch); }
        else gather_children<6>(wv, d, up, P, base, G.maxch, active, stride, q);
#if DJ_CUT
        if constexpr (kCut) { if (ncut > 0) cut_eval<JAC>(d, kb, cimp_p); }
#endif
        for (int i = 0; i < 6; ++i) rb[i] = d[i];
        if (JAC) {
            // ---- supernode matrix S = [[D_b, P_b],[G_b, REG]]  (rows/cols: v(3) ω(3) λt(3) λr(3)); joint blocks are already in ----
            T Dw[9];                                          // (Δt/2)(cJ − Jω ωᵀ/c + [ω]xJ − [Jω]x)
            {
                T Sw[9], SJw[9], SwJ[9];
                m3skew(Sw, L.w); m3skew(SJw, Jw); m3mul(SwJ, Sw, P.J);
                const T ic = trcp(kb.c);
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
                    Dw[3 * i + j] = T(0.5) * dt * (kb.c * P.J[3 * i + j] - Jw[i] * L.w[j] * ic + SwJ[3 * i + j] - SJw[3 * i + j]);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                K.addS(i, i, P.m + T(REG));
#pragma unroll
                for (int j = 0; j < 3; ++j) K.addS(3 + i, 3 + j, Dw[3 * i + j]);
                K.addS(3 + i, 3 + i, T(REG));
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                K.addS(6 + i, 6 + i, (i < P.nl_t) ? T(REG) : T(1));
                K.addS(9 + i, 9 + i, (i < P.nl_r) ? T(REG) : T(1));
            }
            if constexpr (!kEarly) {
#pragma unroll
                for (int c = 0; c < MAXC; ++c) {
                    if (c < P.ncontact) {
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < 3; ++j) K.addS(3 + i, 3 + j, -CE[c].Dww[3 * i + j]);
                        { ContactCold<T>& cc_ = ccold(c); for (int i = 0; i < 18; ++i) { cc_.C134[i] = CE[c].C134[i]; cc_.G134[i] = CE[c].G134[i]; } }
#if DJ_SS
                        if constexpr (kSS) {   // body-body contact: −∂(impulses)/∂(v, ω) of both bodies (contacts/constraints.jl:54-57) and the parent-side rows
                            const bool ss_ = CP[P.contact[c]].kind == 1;
                            ContactCold<T>& cc_ = ccold(c);
                            for (int i = 0; i < 18; ++i) { cc_.Cp134[i] = ss_ ? CS[c].Cp134[i] : T(0); cc_.Gp134[i] = ss_ ? CS[c].Gp134[i] : T(0); }
                            if (ss_) {
#pragma unroll
                                for (int i = 0; i < 3; ++i)
#pragma unroll
                                    for (int j = 0; j < 3; ++j) {
                                        K.addS(i, j, -CS[c].Sxx[3 * i + j]); K.addS(i, 3 + j, -CS[c].Sxw[3 * i + j]); K.addS(3 + i, j, -CS[c].Swx[3 * i + j]);
                                        K.addD(i, j, -CS[c].Sxx[3 * i + j]); K.addD(i, 3 + j, -CS[c].Pxw[3 * i + j]); K.addD(3 + i, j, -CS[c].Pwx[3 * i + j]); K.addD(3 + i, 3 + j, -CS[c].Pww[3 * i + j]);
                                    }
                            }
                        }
#endif
                    }
                }
            }
            for (int j = 0; j < 3; ++j) { F.th_a[j] = E.th_a[j]; F.th_b[j] = E.th_b[j]; }
#if DJ_TSD
            if (tlim) for (int j = 0; j < 3; ++j) { F.thv_a[j] = E.thv_a[j]; F.thv_b[j] = E.thv_b[j]; }
#endif
            for (int j = 0; j < 6; ++j) { F.t_a[j] = E.t_a[j]; F.t_b[j] = E.t_b[j]; }
        }
        DJ_P2E(JAC ? 18 : 12);
    }

    // ---------------------------------------------------------------- violations (src/solver/violations.jl)
    // Also reduces the stiffness of the evaluated iterate, wstiff = max γ/s over the cones of the environment (DJ_REFINE).
    DJ_HD void violations(T& rvio, T& bvio) {
        T r = T(0), b = T(0), wq = T(0);
        const bool track_stiffness = G.refine_w < T(1e300);      // wave-uniform: no threshold (reference-default options), no stiffness arithmetic
        if (active) {
            for (int i = 0; i < 6; ++i) r = tmax(r, tabs(rb[i]));
            for (int i = 0; i < 6; ++i) r = tmax(r, tabs(rj[i]));          // only the Nλ equality rows (padded slots are 0)
#pragma unroll
            for (int li = 0; li < CPL; ++li) if (cidx(li) < P.ncontact) {
                const int c = cidx(li);
                for (int i = 0; i < NCV; ++i) r = tmax(r, tabs(cres[li][i]));
                const T* g = L.cg[c]; const T* s = L.cs[c];
                b = tmax(b, tabs(g[0] * s[0]));
                if constexpr (kLinear) { for (int i = 1; i < NCV; ++i) b = tmax(b, tabs(g[i] * s[i])); }      // complementarity.jl:16 (γ .* s)
                else if (G.contact_model == 0) {
                    b = tmax(b, tabs(g[1] * s[1] + g[2] * s[2] + g[3] * s[3]));
                    b = tmax(b, tabs(g[1] * s[2] + s[1] * g[2]));
                    b = tmax(b, tabs(g[1] * s[3] + s[1] * g[3]));
                }
                if (kTrack && track_stiffness) {
                    wq = tmax(wq, (g[0] + T(REG)) * trcp(s[0] + T(REG)));
                    if (G.contact_model == 0) wq = tmax(wq, (g[1] + T(REG)) * trcp(s[1] + T(REG)));
                }
            }
#if DJ_MLIM
            if constexpr (kMLim) { for (int m = 0; m < NLM; ++m) if (m < nlm()) { b = tmax(b, tabs(L.mls[m][0] * L.mlg[m][0])); b = tmax(b, tabs(L.mls[m][1] * L.mlg[m][1])); } }
#endif
#if DJ_CUT
            if constexpr (kCut) { for (int c = 0; c < NCUT; ++c) if (c < ncut) for (int i = 0; i < 6; ++i) r = tmax(r, tabs(crj[c][i])); }
#endif
            if (lim_on()) {
                b = tmax(b, tabs(L.ls[0] * L.lg[0])); b = tmax(b, tabs(L.ls[1] * L.lg[1]));
                if (kTrack && track_stiffness) { wq = tmax(wq, (L.lg[0] + T(REG)) * trcp(L.ls[0] + T(REG))); wq = tmax(wq, (L.lg[1] + T(REG)) * trcp(L.ls[1] + T(REG))); }
            }
        }
        if constexpr (kSplitC) { r = quad_maxv(r); b = quad_maxv(b); if constexpr (kTrack) wq = quad_maxv(wq); }      // (the lanes of a quad looked at different contacts)
        if constexpr (QUAD) {
            if constexpr (kTrack) { T v3[3] = {r, b, wq}; env_reduce_quad<3>(v3, [](T a_, T b_) { return a_ > b_ ? a_ : b_; }); rvio = v3[0]; bvio = v3[1]; wstiff = v3[2]; }
            else { T v2[2] = {r, b}; env_reduce_quad<2>(v2, [](T a_, T b_) { return a_ > b_ ? a_ : b_; }); rvio = v2[0]; bvio = v2[1]; }
        }
        else { rvio = env_max(wv, r, envl); bvio = env_max(wv, b, envl); if constexpr (kTrack) wstiff = env_max(wv, wq, envl); }
        DJ_P2E(13);
    }

    // ---------------------------------------------------------------- condensation of cone rows
    // comp-row right-hand sides (r1..r4 per contact, r_cu, r_cl for the limit) -> condensed rhs
    // additions for the parent (ra) and own (rbody) body rows; coefficients for the recovery.
    struct ConeRhs { T cc[CPL][NCV]; T lim[2];         // (cc: this lane's contact slots)
#if DJ_MLIM
        T mlim[NLM][2];
#endif
    };

    DJ_HD void cone_rhs_from_state(ConeRhs& R, T mu_asm) const {
#pragma unroll
        for (int li = 0; li < CPL; ++li) {
            const int c = kSplitC ? (cidx(li) < MAXC ? cidx(li) : 0) : li;
            const T* g = L.cg[c]; const T* s = L.cs[c];
            if constexpr (kLinear) { for (int i = 0; i < NCV; ++i) R.cc[li][i] = -(g[i] * s[i] - mu_asm); continue; }   // complementarityμ: − μ·ones(6)
            R.cc[li][0] = -(g[0] * s[0] - mu_asm);
            R.cc[li][1] = -(g[1] * s[1] + g[2] * s[2] + g[3] * s[3] - mu_asm);
            R.cc[li][2] = -(g[1] * s[2] + s[1] * g[2]);
            R.cc[li][3] = -(g[1] * s[3] + s[1] * g[3]);
        }
        R.lim[0] = -(L.ls[0] * L.lg[0] - mu_asm);
        R.lim[1] = -(L.ls[1] * L.lg[1] - mu_asm);
#if DJ_MLIM
        for (int m = 0; m < NLM; ++m) for (int i = 0; i < 2; ++i) R.mlim[m][i] = -(L.mls[m][i] * L.mlg[m][i] - mu_asm);
#endif
    }

    // contact condensation coefficients: Δγ_{1,3,4} = k0 + coef·(C134 Δw);  also Δs2 etc. for recovery
    struct CCoef { T k0[3], coef[9]; T a1, b1, den, al2, al3, al4, g0, g1, g2, h0, h1, h2;
                   T le[kLinear ? 4 : 1], lw[kLinear ? 4 : 1], kpsi, pn, p1, p2; };     // (LinearContact: Δβ_i = le_i − lw_i (±t + Δψ), Δψ = kpsi + pn n + p1 t1 + p2 t2)
    DJ_HD void contact_coef(CCoef& Q, int c, const T* rc /*r1..r4*/, const T* r58 /*−constraint rows*/) const {
        const ContactP<T>& K = CP[P.contact[c]];
        const T* gam = L.cg[c]; const T* s = L.cs[c];
        if constexpr (kLinear) {
            // LinearContact: six orthant pairs, γ̃ = γ + REG, s̃ = s + REG on all of them (linear.jl:49-69).  With n = C1Δw, t1 = C3Δw,
            // t2 = C4Δw (rows of C134):   Δsγ = n − r5;   Δsψ = μΔγ − ΣΔβ − r6;   Δsβ = (t2, −t2, t1, −t1) + Δψ − r7..10;
            //   γ̃_i Δs_i + s̃_i Δγ_i = rc_i  =>  Δγ = a1 + b1 n;  Δβ_i = e_i − w_i(±t + Δψ), w_i = β̃_i/s̃β_i, e_i = (rc_i + β̃_i r_i)/s̃β_i;
            //   Δψ (s̃ψ + ψ̃ Σw) = rc_ψ + ψ̃(r6 + Σe) − ψ̃ μ Δγ − ψ̃[(w3 − w4) t1 + (w1 − w2) t2].
            // The body rows see γ and the tangential impulse b = Pᵀβ = (β3 − β4, β1 − β2) through G134, as for the nonlinear cone.
            const T g0t = gam[0] + T(REG), s0t = s[0] + T(REG), is0 = trcp(s0t);
            Q.a1 = (rc[0] + g0t * r58[0]) * is0; Q.b1 = -g0t * is0;
            T E_ = T(0), W_ = T(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const T bt = gam[2 + i] + T(REG), isb = trcp(s[2 + i] + T(REG));
                Q.lw[i] = bt * isb; Q.le[i] = (rc[2 + i] + bt * r58[2 + i]) * isb;
                E_ += Q.le[i]; W_ += Q.lw[i];
            }
            const T pt = gam[1] + T(REG), spt = s[1] + T(REG), iden = trcp(spt + pt * W_);
            Q.kpsi = (rc[1] + pt * (r58[1] + E_) - pt * K.mu * Q.a1) * iden;
            Q.pn = -pt * K.mu * Q.b1 * iden; Q.p1 = -pt * (Q.lw[2] - Q.lw[3]) * iden; Q.p2 = -pt * (Q.lw[0] - Q.lw[1]) * iden;
            const T d1 = Q.lw[2] - Q.lw[3], d2 = Q.lw[0] - Q.lw[1];
            Q.k0[0] = Q.a1; Q.k0[1] = (Q.le[2] - Q.le[3]) - d1 * Q.kpsi; Q.k0[2] = (Q.le[0] - Q.le[1]) - d2 * Q.kpsi;
            Q.coef[0] = Q.b1; Q.coef[1] = T(0); Q.coef[2] = T(0);
            Q.coef[3] = -d1 * Q.pn; Q.coef[4] = -(Q.lw[2] + Q.lw[3]) - d1 * Q.p1; Q.coef[5] = -d1 * Q.p2;
            Q.coef[6] = -d2 * Q.pn; Q.coef[7] = -d2 * Q.p1; Q.coef[8] = -(Q.lw[0] + Q.lw[1]) - d2 * Q.p2;
            return;
        }
        T g1t = gam[0] + T(REG), s1t = s[0] + T(REG);
        Q.g0 = gam[1] + T(REG); Q.g1 = gam[2]; Q.g2 = gam[3];
        Q.h0 = s[1] + T(REG); Q.h1 = s[2]; Q.h2 = s[3];
        // Δγ1 = a1 + b1 (C1Δw):  γ̃1 Δs1 + s̃1 Δγ1 = r1, Δs1 = C1Δw − r5
        const T is1 = trcp(s1t);
        Q.a1 = (rc[0] + g1t * r58[0]) * is1; Q.b1 = -g1t * is1;
        T ih = trcp(Q.h0);
        Q.al3 = Q.g1 - Q.h1 * Q.g0 * ih; Q.al4 = Q.g2 - Q.h2 * Q.g0 * ih; Q.al2 = Q.h0 - (Q.h1 * Q.h1 + Q.h2 * Q.h2) * ih;
        Q.den = Q.g0 - (Q.h1 * Q.g1 + Q.h2 * Q.g2) * ih;
        // Δγ2 = μf Δγ1 − r6 = (μf a1 − r6) + μf b1 (C1Δw); Δs3 = C3Δw − r7; Δs4 = C4Δw − r8
        T r2p = rc[1] - (Q.h1 * rc[2] + Q.h2 * rc[3]) * ih;
        T dg2_0 = K.mu * Q.a1 - r58[1], dg2_1 = K.mu * Q.b1;
        // Δs2 = [r2p − al3 Δs3 − al4 Δs4 − al2 Δγ2] / den
        T id = trcp(Q.den);
        T s2_0 = (r2p + Q.al3 * r58[2] + Q.al4 * r58[3] - Q.al2 * dg2_0) * id;
        T s2_c1 = -Q.al2 * dg2_1 * id, s2_c3 = -Q.al3 * id, s2_c4 = -Q.al4 * id;
        // Δγ3 = (r3 − g1Δs2 − g0Δs3 − h1Δγ2)/h0 ; Δγ4 = (r4 − g2Δs2 − g0Δs4 − h2Δγ2)/h0
        Q.k0[0] = Q.a1;
        Q.coef[0] = Q.b1; Q.coef[1] = T(0); Q.coef[2] = T(0);
        Q.k0[1] = (rc[2] - Q.g1 * s2_0 + Q.g0 * r58[2] - Q.h1 * dg2_0) * ih;
        Q.coef[3] = (-Q.g1 * s2_c1 - Q.h1 * dg2_1) * ih; Q.coef[4] = (-Q.g1 * s2_c3 - Q.g0) * ih; Q.coef[5] = (-Q.g1 * s2_c4) * ih;
        Q.k0[2] = (rc[3] - Q.g2 * s2_0 + Q.g0 * r58[3] - Q.h2 * dg2_0) * ih;
        Q.coef[6] = (-Q.g2 * s2_c1 - Q.h2 * dg2_1) * ih; Q.coef[7] = (-Q.g2 * s2_c3) * ih; Q.coef[8] = (-Q.g2 * s2_c4 - Q.g0) * ih;
    }

    // ---------------------------------------------------------------- condensation of the cone rows
    // Contacts and joint limits are eliminated analytically onto the body rows.
    template <class BK>
    DJ_HD void condense(BK& K) { condense_contacts(K); condense_limits(K); }
    template <class BK>
    DJ_HD void condense_contacts(BK& K) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (c < P.ncontact) {
                CCoef Q; T rc[NCV] = {}, r58[NCV] = {};
                contact_coef(Q, c, rc, r58);
                const ContactCold<T>& cc_ = ccold(c);
                // body rows: −G134 Δγ134 = −G134 (k0 + coef C134 Δw)  ->  S[0:6,0:6] −= G134 coef C134
                if constexpr (QUAD) {
                    // quad mapping: only the two body-row roles hold such rows, three each -- M = coef C134 once (3 x 6), then
                    // the lane's rows G134[:, 3q + i]ᵀ M (the rows are picked by address in LDS, not by selects over six)
                    if (q < 2) {
                        T Mc[3][6];
#pragma unroll
                        for (int a = 0; a < 3; ++a)
#pragma unroll
                            for (int j = 0; j < 6; ++j) Mc[a][j] = Q.coef[3 * a] * cc_.C134[j] + Q.coef[3 * a + 1] * cc_.C134[6 + j] + Q.coef[3 * a + 2] * cc_.C134[12 + j];
                        const T* gq = cc_.G134 + 3 * q;
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            const T g0 = gq[i], g1 = gq[6 + i], g2 = gq[12 + i];
#pragma unroll
                            for (int j = 0; j < 6; ++j) K.S[i][j] -= TL(g0 * Mc[0][j] + g1 * Mc[1][j] + g2 * Mc[2][j]);
                        }
#if DJ_SS
                        // body-body contact: Δγ134 = k0 + coef (C134 Δw + Cp134 Δw_parent); the parent's body rows carry −Gp134 Δγ134:
                        //   U −= G134ᵀ coef Cp134,   L −= Gp134ᵀ coef C134,   Dup −= Gp134ᵀ coef Cp134     (all zero for a half-space contact)
                        if constexpr (kSS) {
                        T Mp[3][6];
#pragma unroll
                        for (int a = 0; a < 3; ++a)
#pragma unroll
                            for (int j = 0; j < 6; ++j) Mp[a][j] = Q.coef[3 * a] * cc_.Cp134[j] + Q.coef[3 * a + 1] * cc_.Cp134[6 + j] + Q.coef[3 * a + 2] * cc_.Cp134[12 + j];
                        const T* pq = cc_.Gp134 + 3 * q;
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            const T g0 = gq[i], g1 = gq[6 + i], g2 = gq[12 + i], p0 = pq[i], p1 = pq[6 + i], p2 = pq[12 + i];
#pragma unroll
                            for (int j = 0; j < 6; ++j) {
                                K.U[i][j] -= TL(g0 * Mp[0][j] + g1 * Mp[1][j] + g2 * Mp[2][j]);
                                K.D[i][j] -= TL(p0 * Mp[0][j] + p1 * Mp[1][j] + p2 * Mp[2][j]);
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 6; ++r) {
                            const T p0 = cc_.Gp134[r], p1 = cc_.Gp134[6 + r], p2 = cc_.Gp134[12 + r];
#pragma unroll
                            for (int i = 0; i < 3; ++i) K.L[r][i] -= TL(p0 * Mc[0][3 * q + i] + p1 * Mc[1][3 * q + i] + p2 * Mc[2][3 * q + i]);
                        }
                        }
#endif
                    }
                    continue;
                }
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        T acc = T(0);
#pragma unroll
                        for (int a = 0; a < 3; ++a)
#pragma unroll
                            for (int b = 0; b < 3; ++b) acc += cc_.G134[6 * a + i] * Q.coef[3 * a + b] * cc_.C134[6 * b + j];
                        K.addS(i, j, -acc);
                    }
            }
        }
    }
    template <class BK>
    DJ_HD void condense_limits(BK& K) {
#if DJ_MLIM
        // limits on several coordinates: the same elimination per limited coordinate (below), each Δκ row in the padded multiplier slot of its own
        // free coordinate; θ of a translational coordinate depends on v and ω
        if constexpr (kMLim) {
            for (int m = 0; m < NLM; ++m) if (m < nlm()) {
                const T wk = (L.mlg[m][1] + T(REG)) * trcp(L.mls[m][1] + T(REG)) + (L.mlg[m][0] + T(REG)) * trcp(L.mls[m][0] + T(REG));
                const T c_ = trcp(T(1) + wk), sg_ = wk * c_;
                const int sl_ = mslot(m);
                K.addS(sl_, sl_, c_ - T(1));
                for (int j = 0; j < 6; ++j) { K.addS(sl_, j, sg_ * mth_b[m][j]); K.addU(sl_, j, sg_ * mth_a[m][j]); }
                for (int i = 0; i < 6; ++i) { K.addS(i, sl_, -mt_b[m][i]); K.addL(i, sl_, -mt_a[m][i]); }
            }
        }
#endif
        // joint limit: the pair (s, γ) of both sides is eliminated down to ONE unknown, the net limit impulse Δκ = Δγ_lo − Δγ_up
        // = κ0 − wκ (θ_a Δω_a + θ_b Δω_b), wκ = γ_up/s_up + γ_lo/s_lo, which keeps its own row -- the third rotational multiplier
        // slot, free for the one-dimensional rotational joints that may carry limits -- scaled by 1/(1 + wκ):
        //     Δκ/(1 + wκ) + σ (θ_a Δω_a + θ_b Δω_b) = κ0/(1 + wκ),   σ = wκ/(1 + wκ);   body rows: −t_x Δκ.
        // Entries stay in [0, 1] whether the limit is inactive (wκ ~ 1e-10: Δκ ≈ 0) or strongly active (wκ ~ 1e11: an equality
        // row with a tiny regularisation); folding wκ t θᵀ into the body blocks instead costs the IFT its digits there.
        if (lim_on()) {
            const T wk = (L.lg[1] + T(REG)) * trcp(L.ls[1] + T(REG)) + (L.lg[0] + T(REG)) * trcp(L.ls[0] + T(REG));
            const T c_ = trcp(T(1) + wk), sg_ = wk * c_;
#if DJ_TSD
            if (tlim) {                                             // the same row in the third translational slot, θ depends on v and ω
                K.addS(8, 8, c_ - T(1));
#pragma unroll
                for (int j = 0; j < 3; ++j) { K.addS(8, 3 + j, sg_ * F.th_b[j]); K.addU(8, 3 + j, sg_ * F.th_a[j]); K.addS(8, j, sg_ * F.thv_b[j]); K.addU(8, j, sg_ * F.thv_a[j]); }
#pragma unroll
                for (int i = 0; i < 6; ++i) { K.addS(i, 8, -F.t_b[i]); K.addL(i, 8, -F.t_a[i]); }
                return;
            }
#endif
            K.addS(11, 11, c_ - T(1));                              // (the padding put 1 there)
#pragma unroll
            for (int j = 0; j < 3; ++j) { K.addS(11, 3 + j, sg_ * F.th_b[j]); K.addU(11, 3 + j, sg_ * F.th_a[j]); }
#pragma unroll
            for (int i = 0; i < 6; ++i) { K.addS(i, 11, -F.t_b[i]); K.addL(i, 11, -F.t_a[i]); }
        }
    }

    // ---------------------------------------------------------------- factorization sweep (lane = supernode)
    // Eliminates supernodes leaves -> root.
    DJ_HD void factorize(FullBlocks<T>& K) {
        T* Smat = K.S; T* Umat = K.U; T* Lmat = K.L; T* Dup = K.D;
        // leaves -> root, in the factorization precision
        TL Sl[144], Ul[72], Ll[72], up[36];
        for (int i = 0; i < 144; ++i) Sl[i] = TL(Smat[i]);
        for (int i = 0; i < 72; ++i) { Ul[i] = TL(Umat[i]); Ll[i] = TL(Lmat[i]); }
        for (int i = 0; i < 36; ++i) up[i] = TL(0);
        for (int lev = G.maxlevel; lev >= 0; --lev) {
            // receive the children's contributions (they were produced at lev+1)
            TL acc[36];
            for (int i = 0; i < 36; ++i) acc[i] = TL(0);
            gather_children<36>(wv, acc, up, P, base, G.maxch_at(lev), active && P.level == lev, stride, q);
            if (active && P.level == lev) {
                for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) Sl[12 * i + j] += acc[6 * i + j];
                // plain LU of [S U; L Dup] down to the parent's body rows, no pivoting (the host's elimination order makes every pivot
                // the Schur complement of a well-conditioned block): Sinv <- unit-lower L11 \ U11 with the pivots' reciprocals on the
                // diagonal, Z <- L11⁻¹ U, W <- L U11⁻¹, up <- Dup − W Z.  Triangular solves instead of products with an explicit S⁻¹:
                // the explicit form loses cond(S) digits in y − Z Δv_parent.
                if (has_parent) for (int i = 0; i < 36; ++i) up[i] = TL(Dup[i]);
                for (int k = 0; k < 12; ++k) {
                    const TL ip = TL(1) / Sl[13 * k];
                    Sl[13 * k] = ip;
                    for (int i = k + 1; i < 12; ++i) {
                        const TL f = Sl[12 * i + k] * ip;
                        Sl[12 * i + k] = f;
                        for (int j = k + 1; j < 12; ++j) Sl[12 * i + j] -= f * Sl[12 * k + j];
                        if (has_parent) for (int j = 0; j < 6; ++j) Ul[6 * i + j] -= f * Ul[6 * k + j];
                    }
                    if (has_parent) for (int i = 0; i < 6; ++i) {
                        const TL f = Ll[12 * i + k] * ip;
                        Ll[12 * i + k] = f;
                        for (int j = k + 1; j < 12; ++j) Ll[12 * i + j] -= f * Sl[12 * k + j];
                        for (int j = 0; j < 6; ++j) up[6 * i + j] -= f * Ul[6 * k + j];
                    }
                }
                for (int i = 0; i < 144; ++i) F.Sinv[i] = Sl[i];
                if (has_parent) for (int i = 0; i < 72; ++i) { F.W[i] = Ll[i]; F.Z[i] = Ul[i]; }
            }
        }
    }

    // ---------------------------------------------------------------- quad mapping: distributed factorization
    // In-place Gauss-Jordan inverse of the 12x12 supernode matrix whose rows are spread over the four
    // lanes of the quad: for every pivot the owner's (normalised) pivot row is broadcast with 12
    // shuffles and every lane updates its three rows.  No W / Z are stored: the solves use the row
    // block of S⁻¹ together with the lane's rows of U and columns of L.
    DJ_HD void factorize_quad(QuadBlocks<TL>& K) {
        if constexpr (kTrack && DJ_TRACK_GROWTH) growth = T(0);
        // F.Sq holds the raw rows until the lane's level is reached and the inverse rows afterwards:
        // the elimination runs in place and is a no-op (fe = 0) on lanes that are not at the level.
        TL up[3][6];
        TL (&A)[3][12] = F.Sq;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) up[i][j] = TL(0);
        for (int lev = G.maxlevel; lev >= 0; --lev) {
            const bool at = active && P.level == lev;
            // children's Schur complements: rows 0:3 go to role 0, rows 3:6 to role 1 (same role on the child side)
            TL acc[18], upf[18];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) { acc[6 * i + j] = TL(0); upf[6 * i + j] = up[i][j]; }
            mail_post_roles<18>(upf);
            mail_add_children<18>(acc, at, G.maxch_at(lev));
            if (at && q < 2) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) A[i][j] += acc[6 * i + j];
            }
            // distributed Gauss-Jordan (all lanes take part in the shuffles; only lanes at this level change).
            // The owner keeps its pivot row UNSCALED until all twelve pivots are done (a row that has been a
            // pivot only ever receives row operations afterwards, so the factor 1/pivot commutes to the end):
            // every row then takes the same update A −= fe·prow with fe = 0 on the pivot row itself.
            TL ipown[3] = {TL(1), TL(1), TL(1)};
            // Pivot order v, λ_t, ω, λ_r instead of v, ω, λ_t, λ_r: with the translational constraint rows eliminated
            // right after the linear velocity, the ω pivots are the inertia about the JOINT POINT (J + m r², ~1e-3) rather than about the
            // centre of mass (~1e-5 for an Ant foot), and the elimination's growth drops by that ratio -- without pivoting, the order
            // is all there is (numpy model of the tree elimination on the worst default-tolerance cases: 3e-5 -> 1e-8 relative).
#pragma unroll
            for (int pp = 0; pp < 12; ++pp) {
                const int p = pp < 3 ? pp : pp < 6 ? pp + 3 : pp < 9 ? pp - 3 : pp;
                const int o = p / 3, ro = p % 3;
                const bool own = (q == o);
                TL prow[12];
#pragma unroll
                for (int c = 0; c < 12; ++c) prow[c] = wv.quad_bcast(A[ro][c], o);
                const TL ip = Wave::rcp(at ? prow[p] : TL(1));      // (GPU: v_rcp_f64 + two Newton steps = the IEEE quotient, tools/ubench/rcp_test.hip)
                if (own) ipown[ro] = ip;
                // the multiplier f/pivot is formed once per row (three products) instead of scaling the eleven
                // broadcast pivot-row entries; lanes elsewhere get 0 and keep column p
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const TL f = A[r][p];
                    const TL g = f * ip;
                    const TL ge = at ? g : TL(0);
                    if constexpr (kTrack && DJ_TRACK_GROWTH) { const TL ag = ge < TL(0) ? -ge : ge; growth = (r == ro && own) ? growth : tmax(growth, T(ag)); }
                    const TL fe = (r == ro) ? (own ? TL(0) : ge) : ge;
#pragma unroll
                    for (int c = 0; c < 12; ++c) if (c != p) A[r][c] -= fe * prow[c];
                    const TL colp = at ? -g : f;
                    A[r][p] = (r == ro) ? (own ? (at ? TL(1) : f) : colp) : colp;
                }
            }
            if (at) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int c = 0; c < 12; ++c) A[i][c] *= ipown[i];
            }
            if (lev > 0) {   // level 0 = the roots of the trees: nothing to pass up (wave-uniform skip)
            // Schur complement onto the parent: Dup − L S⁻¹ U  (rows 0:3 of U are structurally zero -- except with a
            // translational damper, whose force on the child body depends on the parent's velocity: DJ_TSD builds)
            // Tq = S⁻¹(own rows) U gathered one source role at a time (18 values in flight instead of 54), then the 6x6 product
            // L Tq in two row halves (rows 0:3 end on role 0, rows 3:6 on role 1): 18 partial sums in flight instead of 36
            constexpr int U0 = (DJ_TSD || DJ_SS) ? 0 : 1;     // (a body-body contact fills them too: the force on the child depends on the velocity of the parent)
            TL Tq[3][6];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) Tq[i][j] = TL(0);
#pragma unroll
            for (int o = U0; o < 4; ++o) {
#pragma unroll
                for (int m_ = 0; m_ < 3; ++m_)
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const TL u_ = wv.quad_bcast(F.Uq[m_][j], o);
#pragma unroll
                        for (int i = 0; i < 3; ++i) Tq[i][j] += A[i][3 * o + m_] * u_;
                    }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                TL part[18];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) part[6 * i + j] = F.Lq[3 * h + i][0] * Tq[0][j] + F.Lq[3 * h + i][1] * Tq[1][j] + F.Lq[3 * h + i][2] * Tq[2][j];
#pragma unroll
                for (int i = 0; i < 18; ++i) { part[i] += wv.quad_xor(part[i], 1); }
#pragma unroll
                for (int i = 0; i < 18; ++i) { part[i] += wv.quad_xor(part[i], 2); }
                if (at && has_parent) {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            if (h == 0) up[i][j] = (q == 0) ? TL(K.D[i][j]) - part[6 * i + j] : TL(0);
                            else up[i][j] = (q == 1) ? TL(K.D[i][j]) - part[6 * i + j] : up[i][j];
                        }
                }
            }
            }
        }
    }

    // ---------------------------------------------------------------- quad mapping: the level passes in the ROW layout
    // factorize_quad runs every level pass on all 16 supernode slots of the wavefront although only the supernodes AT the level change
    // (an Ant level holds 1 / 4 / 4 / 4 of its 13), and each of its pivot rows costs 24 v_mov_b32_dpp next to 33 multiply-adds.  Here
    // the supernodes of a level cross to a 16-lane row each (Globals::rp_slot: four per pass; lane r of the row holds matrix row r,
    // lanes 12..15 idle) through LDS, and the pass runs there: the Gauss-Jordan step  A[r][c] -= f_r A[p][c]  is ONE instruction per
    // column, v_fmac_f64_dpp row_newbcast:p (Wave::row_fmac) -- the pivot row arrives inside the multiply-add -- and so are the terms of
    // S^-1 U and L (S^-1 U).  The inverse rows go back to the quad lanes, which keep them for the solves exactly as factorize_quad leaves
    // them; the Schur complement is posted to the mailbox by the row lanes in the slots the parent's gather reads.  Every number is
    // produced by the same operations in the same order as in factorize_quad: the two are interchangeable bit for bit
    // (tests/test_device_program_emu.py::test_row_layout_factorization_is_the_quad_one).
    static constexpr bool kRowsOk = QUAD && Wave::kRows && (Wave::kWaves == 1 ? MAXC == 1 : Wave::kWaves == 2) && !(kTrack && DJ_TRACK_GROWTH);
    static constexpr bool kRowsLuOk = kRowsOk && Wave::kWaves == 1;      // (the IFT's LU-form passes: single-wavefront layouts only)
    DJ_HD int rp_slots_of(int t, int w) const { if constexpr (Wave::kWaves > 1) return w ? G.rp_slot4_w1[t] : G.rp_slot4[t]; else return G.rp_slot4[t]; }
    DJ_HD int rp_children_of(int t, int ci, int w) const { if constexpr (Wave::kWaves > 1) return w ? G.rp_child4_w1[t][ci] : G.rp_child4[t][ci]; else return G.rp_child4[t][ci]; }
    double* stA = nullptr; double* stB = nullptr;      // StepLds::stA_off / stB_off (luA_off / luB_off in the IFT kernel)
    bool rows_here = false;                            // this kernel's LDS layout has the staging areas (a constant of the kernel: StepLds::rows / rows_lu)
    int rp_pack = -256;                                // (pass << 8) | group: the pass and the group that serve this lane's supernode (pass -1: none)
    DJ_HD void rows_init() {
        if constexpr (kRowsOk) {
            const int myslot = wv.lane() >> 2, myw = Wave::kWaves > 1 ? (wv.lane() >> 6) : 0;
            for (int t = 0; t < G.rows; ++t)
                for (int g = 0; g < 4; ++g) if (Globals<T>::rp_byte(rp_slots_of(t, myw), g) == myslot) rp_pack = (t << 8) | g;
        }
    }
    DJ_HD void factorize_rows(QuadBlocks<TL>& K) {
        constexpr int RS = 13, US = 7;                 // (StepLds::ROW_RS / ROW_US)
        TL (&A)[3][12] = F.Sq;
#ifdef DJ_DEBUG
        if (wv.lane() == 0 && std::getenv("DJ_TRACE_ROWS")) std::fprintf(stderr, "factorize_rows: %d passes\n", G.rows);
#endif
        for (int t = 0; t < G.rows; ++t) {
            // (the lane's staging addresses are a handful of integer operations: derived here, from a value the compiler cannot trace,
            //  instead of living in ten registers through the whole Newton loop -- the loop-invariant form cost the rest of the kernel
            //  its registers: 43 -> 149 spilled, profiles/r06_*)
            int ln = wv.lane(), pk = rp_pack;
            DJ_OPAQUE(ln); DJ_OPAQUE(pk);
            // (two-wavefront workgroups: wavefront wq serves its own sixteen slots from staging areas of its own; g = the row group inside the wavefront)
            const int wq = Wave::kWaves > 1 ? (ln >> 6) : 0;
            const int g = (Wave::kWaves > 1 ? (ln & 63) : ln) >> 4, r = ln & 15, rp_pass = pk >> 8, rp_grp = pk & 3;
            double* const stAw = stA + (size_t)wq * (4 * 12 * RS); double* const stBw = stB + (size_t)wq * (4 * 12 * US);
            double* const stL = stAw; double* const stD = stAw + 4 * 6 * RS;
            const double* const rowS = stAw + (size_t)(g * 12 + (r < 12 ? r : 0)) * RS;      // this lane's row as a row lane ...
            const double* const rowU = stBw + (size_t)(g * 12 + (r < 12 ? r : 0)) * US;
            const double* const rowL = stL + (size_t)(g * 6 + (r < 6 ? r : 0)) * RS;
            const double* const rowD = stD + (size_t)(g * 6 + (r < 6 ? r : 0)) * US;
            double* const qS = stAw + (size_t)(rp_grp * 12 + 3 * q) * RS;                    // ... and its three rows as a quad lane
            double* const qU = stBw + (size_t)(rp_grp * 12 + 3 * q) * US;
            const int lev = G.rp_lev[t] & 255, maxch = G.rp_lev[t] >> 8;
            const bool mine = rp_pass == t, at = active && mine;
            wv.sync();
            // ---- quad lanes -> LDS: the rows of S and of U (a supernode slot beyond the batch stages whatever its registers hold: its
            // pass runs on that and nobody reads the results -- no read-back below, and its parent is beyond the batch as well)
            if (mine) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int c = 0; c < 12; ++c) qS[i * RS + c] = (double)A[i][c];
                    if (lev > 0) {
#pragma unroll
                        for (int j = 0; j < 6; ++j) qU[i * US + j] = (double)F.Uq[i][j];
                    }
                }
            }
            wv.sync();
            // ---- row lanes <- LDS.  Unconditional loads: lanes 12..15 of a row and rows without a supernode (slot < 0) compute on whatever
            // the staging area holds -- nobody reads their results (no write-back, no post below), and the loads stay branch-free.
            const int sl = Globals<T>::rp_byte(rp_slots_of(t, wq), g);      // this row's supernode slot (< 0: none): a uniform load and a per-lane shift
            const bool rowon = sl >= 0 && r < 12;
            TL R[12], Ur[6], Lr[12], Dr[6];
#pragma unroll
            for (int c = 0; c < 12; ++c) R[c] = TL(rowS[c]);
            // children's Schur complements onto rows 0:6 x columns 0:6, summed in NodeP::child order first as in factorize_quad (its row
            // lanes r < 6 posted them in the children's passes)
            if (maxch > 0) {
                TL acc[6] = {TL(0), TL(0), TL(0), TL(0), TL(0), TL(0)};
                for (int ci = 0; ci < maxch; ++ci) {
                    const int cs = Globals<T>::rp_byte(rp_children_of(t, ci, wq), g);
                    if (cs >= 0 && r < 6) {
                        const double* m_ = mail_slot(4 * cs, r / 3) + 6 * (r % 3);
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[j] += TL(m_[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) R[j] += acc[j];
            }
            wv.sync();
            if (lev > 0) {                                 // (uniform) the parent-side rows: L by rows, Dup -- over the staged S rows, which are in registers now
                if (mine) {
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int c = 0; c < 3; ++c) stL[(size_t)(rp_grp * 6 + i) * RS + 3 * q + c] = (double)F.Lq[i][c];
                    if (q < 2) {
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < 6; ++j) stD[(size_t)(rp_grp * 6 + 3 * q + i) * US + j] = (double)K.D[i][j];
                    }
                }
                wv.sync();
            }
            // ---- Gauss-Jordan in the pivot order of factorize_quad; the pivot row stays unscaled until the end
            TL ipown = TL(1);
            // (compile-time pivot index: the lane of the DPP control is part of the instruction; static_for, not `#pragma unroll` -- past its size
            //  threshold the unroller leaves a run-time loop with a 12-way switch in front of every multiply-add)
            static_for<0, 12>([&](auto pp_) {
                constexpr int pp = decltype(pp_)::value, p = lu_piv(pp), pn = pp + 1 < 12 ? lu_piv(pp + 1) : -1;
                const TL ip = Wave::rcp(wv.template row_bcast_c<p>(R[p]));
                const bool own = (r == p);
                const TL g_ = R[p] * ip;
                const TL nfe = own ? TL(0) : -g_;
                ipown = own ? ip : ipown;
                // (the column of the NEXT pivot first: its broadcast must not follow the write within two wait states, and the compiler
                //  does not see the write inside Wave::row_fmac)
                if constexpr (pn >= 0) wv.template row_fmac_c<p>(R[pn], R[pn], nfe);
                static_for<0, 12>([&](auto c_) { constexpr int c = decltype(c_)::value; if constexpr (c != p && c != pn) wv.template row_fmac_c<p>(R[c], R[c], nfe); });
                R[p] = own ? TL(1) : -g_;
            });
#pragma unroll
            for (int c = 0; c < 12; ++c) R[c] *= ipown;
            if (lev > 0) {                                 // (their rows of L and Dup: read here, not before the elimination, which needs the registers)
#pragma unroll
                for (int c = 0; c < 12; ++c) Lr[c] = TL(rowL[c]);
#pragma unroll
                for (int j = 0; j < 6; ++j) Dr[j] = TL(rowD[j]);
#pragma unroll
                for (int j = 0; j < 6; ++j) Ur[j] = TL(rowU[j]);
                wv.sync();
            }
            // ---- inverse rows -> LDS -> quad lanes
            if (rowon) {
                double* w_ = stAw + (size_t)(g * 12 + r) * RS;
#pragma unroll
                for (int c = 0; c < 12; ++c) w_[c] = (double)R[c];
            }
            wv.sync();
            if (at) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int c = 0; c < 12; ++c) A[i][c] = TL(qS[i * RS + c]);
            }
            if (lev > 0) {                                 // (uniform) Schur complement onto the parent: Dup - L (S^-1 U), summed as in factorize_quad
                constexpr int U0 = (DJ_TSD || DJ_SS) ? 0 : 1;
                TL Tq[6] = {TL(0), TL(0), TL(0), TL(0), TL(0), TL(0)};
                static_for<3 * U0, 12>([&](auto m_) { constexpr int m = decltype(m_)::value;
#pragma unroll
                    for (int j = 0; j < 6; ++j) wv.template row_fmac_c<m>(Tq[j], Ur[j], R[m]); });
                Wave::dpp_settle();
                TL upr[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    TL pq[4] = {TL(0), TL(0), TL(0), TL(0)};
                    static_for<0, 12>([&](auto m_) { constexpr int m = decltype(m_)::value; wv.template row_fmac_c<m>(pq[m / 3], Tq[j], Lr[m]); });
                    upr[j] = Dr[j] - ((pq[0] + pq[1]) + (pq[2] + pq[3]));
                }
                if (sl >= 0 && r < 6) {                    // rows 0:3 -> the slot of role 0, rows 3:6 -> role 1 (mail_add_children reads them there)
                    double* ms_ = mail_slot(4 * sl, r / 3) + 6 * (r % 3);
#pragma unroll
                    for (int j = 0; j < 6; ++j) ms_[j] = (double)upr[j];
                }
            }
        }
        wv.sync();
    }


    // ---------------------------------------------------------------- quad mapping: LU form of the tree elimination (IFT)
    // The Newton loop inverts every 12x12 supernode matrix explicitly (factorize_quad) and its solves are products with those
    // inverses:  y = S⁻¹ r  on the way up,  x = y − S⁻¹(U x_parent)  on the way down.  Where a joint row repeats a stiff contact
    // row (a foot in sticking contact behind a Fixed joint: entries ~γ/s ~ 1e6 .. 1e8 next to inertias of 1e-5) the blocks of
    // S⁻¹ are differences of terms many orders larger than themselves, and y and S⁻¹(U x_parent) are both large against their
    // difference -- harmless for the Newton iteration (it self-corrects), but the IFT columns lose five digits on about one Ant
    // environment in a thousand (gradient errors up to 2e-5 relative; tools/model/dev_rhs_model.py reproduces it block by block).
    // The IFT therefore factors the same final linearization once more, as the plain LU of the whole tree system in the
    // elimination order (supernodes leaves -> root, inside a supernode v, λt, ω, λr), which is backward stable with the growth
    // this order has (~30): every pivot of a supernode also eliminates the six body rows of its PARENT, so that
    //   Lm / Um   the lane's rows of L11 − I and of D⁻¹U11 − I (S = L11 U11), zero-filled: select-free substitutions (staged, see below)
    //   F.Uq      becomes T = L11⁻¹ U            (rows of this supernode x parent's velocity)
    //   F.Lq      becomes m = L U11⁻¹            (the parent's body rows' multipliers, columns of this supernode)
    //   K.D       becomes Dup − m T              (Schur complement onto the parent's diagonal block)
    // and a solve is  ỹ = L11⁻¹ r;  r_parent −= m ỹ  on the way up,  x = U11⁻¹(ỹ − T x_parent)  on the way down: per column 33 + 18
    // multiply-adds and 11 DPP broadcasts each way (the products with the explicit inverse: 36 + 18 and 12).  What waits between
    // the sweeps is ỹ, all twelve rows of it.
    struct QuadLU { TL Lm[3][12], Um[3][12], di[3]; };
    static constexpr DJ_HD int lu_piv(int pp) { return pp < 3 ? pp : pp < 6 ? pp + 3 : pp < 9 ? pp - 3 : pp; }   // pp-th pivot (an involution: also the position of column c)
    static constexpr DJ_HD int lu_rolepos(int o) { return o == 1 ? 2 : o == 2 ? 1 : o; }                         // position of role o's rows in the pivot order

    // forward substitution with the unit lower factor for NCOLS right-hand sides at once (r[n] = the lane's three rows of column n):
    // the columns are independent chains, which is what hides the DPP and FMA latencies
    template <int NCOLS> DJ_HD void lu_forward_quad(const TL (&Lm)[3][12], TL (&r)[NCOLS][3]) {
#pragma unroll
        for (int pp = 0; pp < 11; ++pp) {                      // (nothing comes after the last pivot)
            const int p = lu_piv(pp), o = p / 3, ro = p % 3;
            TL yp[NCOLS];
#pragma unroll
            for (int n = 0; n < NCOLS; ++n) yp[n] = wv.quad_bcast(r[n][ro], o);
#pragma unroll
            for (int n = 0; n < NCOLS; ++n)
#pragma unroll
                for (int i = 0; i < 3; ++i) r[n][i] -= Lm[i][p] * yp[n];
        }
    }
    // x = U11⁻¹ r: scaling by the reciprocal pivots, then backward substitution with the unit upper factor
    template <int NCOLS> DJ_HD void lu_backward_quad(const TL (&Um)[3][12], const TL (&di)[3], TL (&r)[NCOLS][3]) {
#pragma unroll
        for (int n = 0; n < NCOLS; ++n)
#pragma unroll
            for (int i = 0; i < 3; ++i) r[n][i] *= di[i];
#pragma unroll
        for (int pp = 11; pp > 0; --pp) {                      // (nothing comes before the first pivot)
            const int p = lu_piv(pp), o = p / 3, ro = p % 3;
            TL xp[NCOLS];
#pragma unroll
            for (int n = 0; n < NCOLS; ++n) xp[n] = wv.quad_bcast(r[n][ro], o);
#pragma unroll
            for (int n = 0; n < NCOLS; ++n)
#pragma unroll
                for (int i = 0; i < 3; ++i) r[n][i] -= Um[i][p] * xp[n];
        }
    }

    // The factors wait in global memory (KernelArgs::lu, [workgroup][LU_PER_LANE][lanes]: lane index fastest) while the data blocks
    // are assembled, and each sweep loads only its half: L11, m for the up-sweep, U11, T, D⁻¹ for the down-sweep -- 54 / 57 values
    // per lane in registers instead of 111 (all of them resident cost the sweeps ~30 scratch accesses per pipeline step, which one
    // wave per SIMD cannot hide: the sweeps ran at half speed).
    enum { LU_LM = 0, LU_M = 36, LU_UM = 54, LU_T = 90, LU_DI = 108, LU_END = 111 };            // (dj::LU_PER_LANE values per lane, static_assert in store_lu)
    T* lu = nullptr; int lu_stride = 0;
    // (the stride is the workgroup's width: a compile-time constant on the GPU, so that the 54 / 57 elements a sweep loads sit at immediate offsets
    //  from a few bases.  As a run-time member the compiler formed all their 64-bit addresses early and spilled them: every factor load of the
    //  sweeps' prologues was `scratch_load (the address); s_waitcnt vmcnt(0); global_load` -- 160 serialized round trips per environment-step)
    DJ_HD size_t lu_S() const { if constexpr (Wave::kWidth > 0) return (size_t)Wave::kWidth; else return (size_t)lu_stride; }
    // A supernode's rows of the in-place factors (F.Sq) stay as its own level left them -- on every other level its updates have a zero
    // multiplier -- so the split into the zero-filled triangles the substitutions read happens here, once, after the last level (kept
    // inside the level loop, the 75 values were loop-carried state next to the 90 of the factorization itself: past the 256 architectural
    // registers, every pivot step paid for it in v_accvgpr moves).  Idle supernode slots store an identity system: their lanes run the sweeps too.
    DJ_HD void store_lu() {
        static_assert(LU_END < LU_PER_LANE, "the staged factors must fit dj::LU_PER_LANE");
        QuadLU W;
        const int rq = lu_rolepos(q);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int pos = 3 * rq + r;                          // this row's place in the pivot order
            TL dg = TL(1);
#pragma unroll
            for (int c = 0; c < 12; ++c) { const TL a_ = F.Sq[r][c]; dg = (active && lu_piv(c) == pos) ? a_ : dg; }
            const TL di_ = Wave::rcp(dg);
            W.di[r] = di_;
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                const TL a_ = F.Sq[r][c];
                W.Lm[r][c] = (active && lu_piv(c) < pos) ? a_ : TL(0);
                W.Um[r][c] = (active && lu_piv(c) > pos) ? a_ * di_ : TL(0);
            }
        }
        T* o = lu; const size_t S_ = lu_S();
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 12; ++j) { o[(size_t)(LU_LM + 12 * i + j) * S_] = T(W.Lm[i][j]); o[(size_t)(LU_UM + 12 * i + j) * S_] = T(W.Um[i][j]); }
#pragma unroll
            for (int j = 0; j < 6; ++j) { o[(size_t)(LU_M + 3 * j + i) * S_] = T(F.Lq[j][i]); o[(size_t)(LU_T + 6 * i + j) * S_] = has_parent ? T(F.Uq[i][j]) : T(0); }
            o[(size_t)(LU_DI + i) * S_] = T(W.di[i]);
        }
    }
    // (the loads start from a base the optimizer cannot connect with store_lu's addresses: it kept those 111 addresses alive from the stores to
    //  the loads, i.e. spilled them, and every factor load of a sweep's prologue became `scratch_load (its address); s_waitcnt vmcnt(0); global_load`)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const T __attribute__((address_space(1)))* LuPtr;     // (behind the opaque statement the pointer is generic again: the loads would be FLAT ones)
    DJ_HD LuPtr lu_base(int dl = 0) const { const T* o = lu + dl; DJ_OPAQUE(o); return (LuPtr)o; }
#else
    typedef const T* LuPtr;
    DJ_HD LuPtr lu_base(int dl = 0) const { return lu + dl; }
#endif
    DJ_HD void load_lu_up(TL (&Lm)[3][12], TL (&m)[6][3]) const {
        LuPtr o = lu_base(); const size_t S_ = lu_S();
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 12; ++j) Lm[i][j] = TL(o[(size_t)(LU_LM + 12 * i + j) * S_]);
#pragma unroll
            for (int j = 0; j < 6; ++j) m[j][i] = TL(o[(size_t)(LU_M + 3 * j + i) * S_]);
        }
    }
    DJ_HD void load_lu_down(TL (&Um)[3][12], TL (&Tq)[3][6], TL (&di)[3]) const {
        LuPtr o = lu_base(); const size_t S_ = lu_S();
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 12; ++j) Um[i][j] = TL(o[(size_t)(LU_UM + 12 * i + j) * S_]);
#pragma unroll
            for (int j = 0; j < 6; ++j) Tq[i][j] = TL(o[(size_t)(LU_T + 6 * i + j) * S_]);
            di[i] = TL(o[(size_t)(LU_DI + i) * S_]);
        }
    }

    // Rows stay distributed as in factorize_quad: role q owns rows 3q .. 3q+2 of S and U, columns 3q .. 3q+2 of L, roles 0 / 1
    // rows 0:3 / 3:6 of Dup.  On lanes that are not at the current level every update has a zero multiplier.
    DJ_HD void factorize_quad_lu(QuadBlocks<TL>& K) {
        TL up[3][6];
        TL (&A)[3][12] = F.Sq;
        const int rq = lu_rolepos(q);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) up[i][j] = TL(0);
        for (int lev = G.maxlevel; lev >= 0; --lev) {
            const bool at = active && P.level == lev;
            TL acc[18], upf[18];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) { acc[6 * i + j] = TL(0); upf[6 * i + j] = up[i][j]; }
            mail_post_roles<18>(upf);
            mail_add_children<18>(acc, at, G.maxch_at(lev));
            if (at && q < 2) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) A[i][j] += acc[6 * i + j];
            }
#pragma unroll
            for (int pp = 0; pp < 12; ++pp) {
                const int p = lu_piv(pp), o = p / 3, ro = p % 3, opos = lu_rolepos(o);
                // the pivot row over the columns still to come: [S (later columns) | U]
                TL prow[12], pU[6];
#pragma unroll
                for (int c = 0; c < 12; ++c) prow[c] = (lu_piv(c) >= pp) ? wv.quad_bcast(A[ro][c], o) : TL(0);
#pragma unroll
                for (int j = 0; j < 6; ++j) pU[j] = wv.quad_bcast(F.Uq[ro][j], o);
                const TL ip = Wave::rcp(at ? prow[p] : TL(1));
                // rows of this supernode that come later in the order
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const bool later = at && (rq > opos || (rq == opos && r > ro));
                    const TL f = A[r][p] * ip;
                    const TL fe = later ? f : TL(0);
#pragma unroll
                    for (int c = 0; c < 12; ++c) if (lu_piv(c) > pp) A[r][c] -= fe * prow[c];
#pragma unroll
                    for (int j = 0; j < 6; ++j) F.Uq[r][j] -= fe * pU[j];
                    A[r][p] = later ? f : A[r][p];
                }
                // the six body rows of the parent: multipliers m_i = (parent row i, column p) / pivot, from the lane that owns column p
                if (lev > 0) {                                   // (uniform; the roots have no parent rows)
                    TL m[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) { const TL t_ = F.Lq[i][ro] * ip; m[i] = wv.quad_bcast(t_, o); }
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        const bool laterc = at && (rq > opos || (rq == opos && cc > ro));
                        const TL c0_ = prow[cc], c1_ = prow[3 + cc], c2_ = prow[6 + cc], c3_ = prow[9 + cc];
                        const TL pm = q == 0 ? c0_ : q == 1 ? c1_ : q == 2 ? c2_ : c3_;   // the pivot row's entry in this lane's column cc
                        const TL pme = laterc ? pm : TL(0);
#pragma unroll
                        for (int i = 0; i < 6; ++i) F.Lq[i][cc] -= m[i] * pme;
                    }
                    const bool own = at && q == o;
#pragma unroll
                    for (int i = 0; i < 6; ++i) F.Lq[i][ro] = own ? m[i] : F.Lq[i][ro];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const TL m0_ = m[i], m1_ = m[3 + i];
                        const TL mm = (at && q < 2) ? (q == 0 ? m0_ : m1_) : TL(0);
#pragma unroll
                        for (int j = 0; j < 6; ++j) K.D[i][j] -= mm * pU[j];
                    }
                }
            }
            if (lev > 0 && at && has_parent) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) up[i][j] = q < 2 ? TL(K.D[i][j]) : TL(0);
            }
        }
    }

    // The same LU-form pass in the ROW layout (see factorize_rows): the supernodes of a level on a 16-lane row each, lane r with row r of
    // [S | U], lanes 0..5 also with row i of the parent's [L | Dup]; a pivot step is one v_fmac_f64_dpp per entry that changes.  Factor rows,
    // T and m go back to the quad lanes in the places factorize_quad_lu leaves them (store_lu reads them there); the Schur complement is
    // posted by the row lanes.  Same operations in the same order: bit for bit the quad passes.
    DJ_HD void factorize_rows_lu(QuadBlocks<TL>& K) {
        constexpr int RS = 12, US = 7, DS = 6;         // (StepLds::LU_RS / ROW_US / LU_DS)
        TL (&A)[3][12] = F.Sq;
#ifdef DJ_DEBUG
        if (wv.lane() == 0 && std::getenv("DJ_TRACE_ROWS")) std::fprintf(stderr, "factorize_rows_lu: %d passes\n", G.rows);
#endif
        for (int t = 0; t < G.rows; ++t) {
            int ln = wv.lane(), pk = rp_pack;
            DJ_OPAQUE(ln); DJ_OPAQUE(pk);
            const int g = ln >> 4, r = ln & 15, rp_pass = pk >> 8, rp_grp = pk & 3;
            double* const stU = stB; double* const stD = stB + 4 * 12 * US; double* const stL = stA;
            double* const rowS = stA + (size_t)(g * 12 + (r < 12 ? r : 0)) * RS;            // this lane's rows as a row lane ...
            double* const rowU = stU + (size_t)(g * 12 + (r < 12 ? r : 0)) * US;
            double* const rowL = stL + (size_t)(g * 6 + (r < 6 ? r : 0)) * RS;
            const double* const rowD = stD + (size_t)(g * 6 + (r < 6 ? r : 0)) * DS;
            double* const qS = stA + (size_t)(rp_grp * 12 + 3 * q) * RS;                    // ... and its three rows as a quad lane
            double* const qU = stU + (size_t)(rp_grp * 12 + 3 * q) * US;
            const int lev = G.rp_lev[t] & 255, maxch = G.rp_lev[t] >> 8;
            const bool mine = rp_pass == t, at = active && mine;
            wv.sync();
            if (mine) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int c = 0; c < 12; ++c) qS[i * RS + c] = (double)A[i][c];
                    if (lev > 0) {
#pragma unroll
                        for (int j = 0; j < 6; ++j) qU[i * US + j] = (double)F.Uq[i][j];
                    }
                }
                if (lev > 0 && q < 2) {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 6; ++j) stD[(size_t)(rp_grp * 6 + 3 * q + i) * DS + j] = (double)K.D[i][j];
                }
            }
            wv.sync();
            const int sl = Globals<T>::rp_byte(G.rp_slot4[t], g);
            const bool rowon = sl >= 0 && r < 12;
            TL R[12], Ur[6], Lr[12], Dr[6];
#pragma unroll
            for (int c = 0; c < 12; ++c) R[c] = TL(rowS[c]);
            if (lev > 0) {
#pragma unroll
                for (int j = 0; j < 6; ++j) { Ur[j] = TL(rowU[j]); Dr[j] = TL(rowD[j]); }
            }
            if (maxch > 0) {                               // children's Schur complements, summed first in NodeP::child order (factorize_quad_lu)
                TL acc[6] = {TL(0), TL(0), TL(0), TL(0), TL(0), TL(0)};
                for (int ci = 0; ci < maxch; ++ci) {
                    const int cs = Globals<T>::rp_byte(G.rp_child4[t][ci], g);
                    if (cs >= 0 && r < 6) {
                        const double* m_ = mail_slot(4 * cs, r / 3) + 6 * (r % 3);
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[j] += TL(m_[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) R[j] += acc[j];
            }
            wv.sync();
            if (lev > 0) {                                 // (uniform) the parent's rows of L, over the staged S rows
                if (mine) {
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int c = 0; c < 3; ++c) stL[(size_t)(rp_grp * 6 + i) * RS + 3 * q + c] = (double)F.Lq[i][c];
                }
                wv.sync();
#pragma unroll
                for (int c = 0; c < 12; ++c) Lr[c] = TL(rowL[c]);
                wv.sync();
            }
            // ---- elimination in the pivot order: rows (and the parent's rows) that come later take the update
            static_for<0, 12>([&](auto pp_) {
                constexpr int pp = decltype(pp_)::value, p = lu_piv(pp), pn = pp + 1 < 12 ? lu_piv(pp + 1) : -1;
                const TL ip = Wave::rcp(wv.template row_bcast_c<p>(R[p]));
                const int pos = r < 3 ? r : r < 6 ? r + 3 : r < 9 ? r - 3 : r;            // lu_piv(r): this row's place in the order
                const bool later = pos > pp;
                const TL f = R[p] * ip;
                const TL nfe = later ? -f : TL(0);
                if constexpr (pn >= 0) wv.template row_fmac_c<p>(R[pn], R[pn], nfe);
                static_for<0, 12>([&](auto c_) { constexpr int c = decltype(c_)::value; if constexpr (lu_piv(c) > pp && c != pn) wv.template row_fmac_c<p>(R[c], R[c], nfe); });
                if (lev > 0) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) wv.template row_fmac_c<p>(Ur[j], Ur[j], nfe);
                }
                R[p] = later ? f : R[p];
                if (lev > 0) {                             // the six body rows of the parent: m = (row i, column p) / pivot
                    const TL m = Lr[p] * ip, nm = -m;
                    static_for<0, 12>([&](auto c_) { constexpr int c = decltype(c_)::value; if constexpr (lu_piv(c) > pp) wv.template row_fmac_c<p>(Lr[c], R[c], nm); });
                    Lr[p] = m;
#pragma unroll
                    for (int j = 0; j < 6; ++j) wv.template row_fmac_c<p>(Dr[j], Ur[j], nm);
                }
            });
            // ---- factor rows and T -> LDS -> quad lanes; the Schur complement to the mailbox
            if (rowon) {
#pragma unroll
                for (int c = 0; c < 12; ++c) rowS[c] = (double)R[c];
                if (lev > 0) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) rowU[j] = (double)Ur[j];
                }
            }
            if (lev > 0 && sl >= 0 && r < 6) {
                double* ms_ = mail_slot(4 * sl, r / 3) + 6 * (r % 3);
#pragma unroll
                for (int j = 0; j < 6; ++j) ms_[j] = (double)Dr[j];
            }
            wv.sync();
            if (at) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int c = 0; c < 12; ++c) A[i][c] = TL(qS[i * RS + c]);
                    if (lev > 0) {
#pragma unroll
                        for (int j = 0; j < 6; ++j) F.Uq[i][j] = TL(qU[i * US + j]);
                    }
                }
            }
            if (lev > 0) {                                 // ... and the rows of m, over the factor rows the quad lanes have read
                wv.sync();
                if (sl >= 0 && r < 6) {
#pragma unroll
                    for (int c = 0; c < 12; ++c) rowL[c] = (double)Lr[c];
                }
                wv.sync();
                if (at) {
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int c = 0; c < 3; ++c) F.Lq[i][c] = TL(stL[(size_t)(rp_grp * 6 + i) * RS + 3 * q + c]);
                }
            }
        }
        wv.sync();
    }

    // The right-hand sides of one batch (NC columns, this lane's role: three rows each) of the supernode `idk` -- the executing lane's own
    // supernode, or a root on whose behalf it works (up-sweep, root phase) -- from that supernode's block Rn, plus what its children sent:
    //   body roles (ABI-type block a):   own batch: own_cfg (configuration columns, double) | ROWNV (velocity columns);
    //                                    parent's configuration batch: RPARB;  control batch: UB on the owner (child body of the joint)
    //   joint roles (double block jd):   own configuration batch: ROWNJ;  parent's configuration batch: RPARJ
    // MODE 1: contact batch b − nbs = contact index of the environment, five columns, on the body rows of the contact's supernode.
    template <int MODE, int NC, class RH, class TG>
    DJ_HD void sweep_rhs(TG (&r3a)[NC][3], const RH& Rn, int idk, int idparent, bool idhasp, int idu_off, int idmyu, int idnlim, TG idwk,
                         int idncontact, const int* idcontact, int b, int nbs, bool valid, const TG* acc) const {
        const int qh = q & 1;
        const bool isS = b < nbs;
        const int kk = b >> 1, typ = b & 1;
        const bool mine = valid && isS && (idk == kk), par = valid && isS && typ == 0 && idhasp && (idparent == kk);
        const int cu0 = NC * (b - nbs) - idu_off;             // control batch: local input index of column 0
        int r_off = 0, j_off = 0, sl_off = 0, cl = -1;
        bool od = false; TG rm_s = TG(0), wkm = TG(0);
        if constexpr (MODE == 0) {
            r_off = isS ? (mine ? RH::ROWNV : RH::RPARB) + qh * 18 : RH::UB + qh * 18 + cu0;
            j_off = (mine ? RH::ROWNJ : RH::RPARJ) + qh * 18;
            od = mine && typ == 0 && q < 2;                     // the folded owner rows come from the double block
            rm_s = (mine ? ((q < 2) == (typ == 1)) : par) ? TG(1) : TG(0);
            // joint-limit condensation: the slack rows (rs, −rs) enter the Δκ row (row 2 of role 3; role 2 for a translational limit) as σ rs
            wkm = (idnlim > 0 && q == ((DJ_TSD && idnlim == 2) ? 2 : 3) && (mine || par) && typ == 0) ? idwk : TG(0);
            sl_off = mine ? RH::SLO : RH::SLP;
        } else {
#pragma unroll
            for (int c_ = 0; c_ < MAXC; ++c_) if (c_ < idncontact && idcontact[c_] == b - nbs) cl = c_;
            r_off = (cl < 0 ? 0 : cl) * 36 + qh * 18;
        }
#pragma unroll
        for (int cI = 0; cI < NC; ++cI) {
            TG r_[3];
            if constexpr (MODE == 0) {
                const bool uok = !isS && valid && q < 2 && (cu0 + cI) >= 0 && (cu0 + cI) < idmyu;
                const TG rm = isS ? rm_s : (uok ? TG(1) : TG(0));
                const int ir = (isS || uok) ? r_off + cI : 0, ij = isS ? j_off + cI : 0;
                // ONE read per value, at a lane-dependent byte offset into the block: the body roles' ABI-type block a, or the double blocks
                // own_cfg / jd.  (Reading all three and selecting made the compiler put every read behind a branch of its own: 18 x
                // [branch, three reads, two waits] per step, each exposing the LDS latency to the one wave of the SIMD.)
                typedef typename std::remove_reference<decltype(Rn.a[0])>::type TBv;
                const char* const Rb = (const char*)&Rn;
                const bool isd = sizeof(TBv) == 8 || q >= 2 || od;
                const TG sc = od ? TG(1) : rm;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int ob = q < 2 ? (od ? (int)offsetof(RH, own_cfg) + 8 * (qh * 18 + i * 6 + cI) : (int)sizeof(TBv) * (ir + i * 6))
                                         : (int)offsetof(RH, jd) + 8 * (ij + i * 6);
                    r_[i] = sc * TG(lds_read_as_double<sizeof(TBv) == 8>(Rb + ob, isd));
                }
                r_[2] += wkm * TG(Rn.jd[sl_off + cI]);
            } else {
                const TG rm = (valid && q < 2 && cl >= 0 && cI < 5) ? TG(1) : TG(0);
#pragma unroll
                for (int i = 0; i < 3; ++i) r_[i] = rm * TG(Rn.a[r_off + i * 6 + cI]);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) r3a[cI][i] = valid ? TG(r_[i]) + acc[3 * cI + i] : TG(0);
        }
    }

    // quad mapping: IFT column solves  X = solmat⁻¹ · datamat  (DESIGN.md §5).
    // Columns travel through the tree six at a time ("batches") and the two sweeps are software-pipelined over the batches: supernodes
    // at different levels work on different batches in the same step, so that every lane does a block substitution per step instead of
    // idling at the other levels' turns.  Since round 4 the pipelines run per BRANCH (a child of a root with its subtree) over the batches
    // that are non-zero in that branch, all branches at once, and the roots are handled apart (see the two sweeps below): the Ant's 32
    // steps per sweep became 10 + 2 root rounds (up) and 10 + a chain-free product phase (down).
    // The sweeps are the forward / backward substitutions of the tree's LU form (factorize_quad_lu):
    //   up    ỹ = L11⁻¹ (r + Σ_children messages);  message to the parent's body rows: u − m ỹ
    //   down  x = U11⁻¹ (ỹ − T x_parent)
    // Between the sweeps ỹ -- all twelve rows of every branch supernode; a root's never leaves its registers -- waits in the output column itself (ABI type = arithmetic
    // type: role 0 -> the v rows of the lane's body, role 1 -> the ω rows, roles 2 / 3 -> the x3 / φ3 rows; nobody writes those
    // before the down-sweep has fetched them: the fetch of a batch is issued one step before its output stores, by the same
    // wavefront) or, with a narrower ABI type, in a buffer of its own in the arithmetic type (KernelArgs::ypark,
    // [batch][column][row][lane]: every store / load of a wavefront is one 512-byte run): a rounded ỹ costs stiff
    // environments their gradient.
    // MODE 0: state + control columns (get_maximal_gradients);  MODE 1: contact-data columns (get_contact_gradients):
    // one batch per contact of the environment, five columns (friction, radius, origin), owner = the contact's body.
    template <int MODE = 0, class KA, class RH, class KN>
    DJ_HD void gradient_columns_quad(const KA& A, int env, const RH& R, T wk, const KN& kb0, const SweepP& sp) {
        typedef typename KA::io_type TIO;
        typedef TL TG;
        constexpr bool ypk = sizeof(TIO) < sizeof(TG);          // (the host allocates KernelArgs::ypark whenever this holds)
        const int yW = wv.width();
        T* const yp0 = ypark;                                  // this lane's slot of batch 0, column 0, row 0
        const int prk_off = (q == 0 ? 3 : q == 1 ? 9 : q == 2 ? 0 : 6) - 6 * q;     // ỹ's rows in the output column, relative to row0 = 12 k + 6 q
        constexpr int NC = 6;
        const T dt = G.dt;
        const int nx = 12 * G.Nb;
        const int nbs = MODE == 0 ? 2 * G.Nb : 0;              // state batches: (body kk, configuration | velocity columns)
        const int nbu = MODE == 0 ? ((A.du != nullptr && G.nu > 0) ? (G.nu + NC - 1) / NC : 0) : G.Nc;   // control batches: six input columns each | contact batches
        const int NB = nbs + nbu;
        const int ncol_u = MODE == 0 ? G.nu : 5 * G.Nc;        // columns of the second output buffer
        const int myu = sp.myu;
        const int lvl = sp.level;
        const int pb = sp.pb;
        const int qh = q & 1;
        // this lane's six rows of the first column of batch b in the output buffers, element stride between columns = nx;
        // state batches: column cI sits at index cI (+3 for cI >= 3: the batch holds x2|φ2 or v15|ω15)
        TIO* const dz_p = DJ_GLOBAL_PTR(TIO, MODE == 0 ? A.dz : A.dc);
        TIO* const du_p = DJ_GLOBAL_PTR(TIO, MODE == 0 ? (nbu > 0 ? A.du : A.dz) : A.dc);
        const size_t env_dz = (size_t)env * nx, env_du = (size_t)env * ncol_u, nxs = (size_t)nx;
        const int ucols = MODE == 0 ? NC : 5;                  // columns a control / contact batch owns in the output buffer
        auto colbase = [=](int b, int kx, int role) -> TIO* {     // kx: the supernode whose rows are addressed (this lane's own; a root's in the root phase), role: whose six rows
            const size_t rowx = (size_t)(12 * kx + 6 * role);
            TIO* pz = dz_p + (env_dz + (size_t)(12 * (b >> 1) + 3 * (b & 1))) * nxs + rowx;
            TIO* pu = du_p + (env_du + (size_t)(ucols * (b - nbs))) * nxs + rowx;
            return b < nbs ? pz : pu;
        };
        // does column cI of batch b exist?  (padding columns of the last control batch / the sixth column of a contact batch)
        auto col_ok = [=](int b, int cI) -> bool { return b < nbs || (MODE == 0 ? NC * (b - nbs) + cI < ncol_u : cI < 5); };
        // is this supernode's ỹ of batch b non-zero (sweep_masks)?  (b must be a valid batch index)
        const unsigned long long sub_mask = sp.sub_mask, ub_mask = sp.ub_mask; const int par_k = has_parent ? sp.parent : -1;
        auto y_nz = [=](int b) -> bool { return b < nbs ? (((sub_mask >> (b >> 1)) & 1ull) != 0 || ((b & 1) == 0 && par_k == (b >> 1))) : ((ub_mask >> (b - nbs)) & 1ull) != 0; };
        // ---------------- up-sweep (leaves -> root): branches in parallel, then the roots ----------------
        // ỹ of a batch is non-zero only on the supernodes that carry one of its right-hand sides and on their ancestors (sweep_masks):
        // for a tree like the Ant's a quarter of the (supernode, batch) pairs.  Marching every batch through every supernode (rounds
        // 1-3: NB + depth pipeline steps) spends the rest on zeros, and cannot skip them -- the quads of a wavefront run in lock step,
        // each on a different batch.  So the sweep is scheduled by BRANCH (a child of a root and its subtree):
        //   phase 1  every branch marches only ITS batches (those that are non-zero somewhere in it, in batch order) through its own
        //            supernodes, all branches at once: a supernode at level l >= 1 works on the i-th batch of its branch in step
        //            i + (maxlevel − l); the level-1 supernodes leave their messages to the root's body rows in KernelArgs::msg;
        //   phase 2  the roots' substitutions -- every batch reaches a root -- are dealt out over ALL quads of the environment (idle
        //            supernode slots included): quad s takes batches s, s + S, ... on the root's behalf, with the root's factors (staged in
        //            KernelArgs::lu), right-hand sides (its QuadRhs block in LDS) and the messages of phase 1; the backward substitution
        //            follows at once in registers (a root has no parent term) and Δv, Δω of the root go to xr (behind msg) for the whole tree.
        // Steps: max over branches of their batch count + maxlevel − 1, plus ceil(NB / S) root rounds (Ant: 8 + 2 + 2 instead of 32).
        // The arithmetic of every (supernode, batch) pair is what it was: the same operations in the same order.
        // -- what every supernode publishes (topology only: identical in every environment of the workgroup)
        SweepInfo* const inf = sinfo + (base >> 2);              // this environment's entries, indexed by supernode
        unsigned long long bm0 = 0ull, bm1 = 0ull;
        {
            const unsigned long long sub_t = sp.sub_t, ub_t = sp.ub_t; const int par_t = sp.parent;
            if (k < G.Nb) for (int b_ = 0; b_ < NB; ++b_) {
                const bool nz_ = b_ < nbs ? (((sub_t >> (b_ >> 1)) & 1ull) != 0 || ((b_ & 1) == 0 && par_t == (b_ >> 1))) : ((ub_t >> (b_ - nbs)) & 1ull) != 0;
                if (nz_) { if (b_ < 64) bm0 |= 1ull << b_; else bm1 |= 1ull << (b_ - 64); }
            }
        }
        wv.sync();
        if (q == 0) {
            SweepInfo& me = inf[k];
            me.level = k < G.Nb ? lvl : -1; me.parent = sp.parent; me.u_off = sp.u_off; me.myu = sp.myu; me.nlim_r = sp.nlim_r; me.ncontact = sp.ncontact; me.nchild = k < G.Nb ? sp.nchild : 0;
#pragma unroll
            for (int ci = 0; ci < MAXCH; ++ci) me.child[ci] = (sp.child_lane0[ci] - base) >> 2;
#pragma unroll
            for (int c_ = 0; c_ < 8; ++c_) me.contact[c_] = (unsigned char)sp.contact[c_];
            me.wk = (double)wk; me.bm[0] = bm0; me.bm[1] = bm1; me.topid = 0;
        }
        wv.sync();
        // -- this supernode's branch: its level-1 ancestor, the batches of that branch, its rank among the level-1 supernodes
        int top = k, topid = 0, maxn = 0;
        if (k < G.Nb && lvl >= 1) { int guard = 0; while (inf[top].level > 1 && guard++ < 64) top = inf[top].parent; }
        unsigned long long tbm0 = 0ull, tbm1 = 0ull;              // the batches of this supernode's branch (none for a root)
        if (k < G.Nb && lvl >= 1) { tbm0 = inf[top].bm[0]; tbm1 = inf[top].bm[1]; }
        unsigned long long rem0 = tbm0, rem1 = tbm1;
        int rootk = k < G.Nb ? k : 0, rootrank = 0, ntops = 0;  // this supernode's root, its rank among the roots; level-1 supernodes of the environment
        if (k < G.Nb && lvl >= 1) rootk = inf[top].parent;
        int nroots = 0;
        for (int a = 0; a < G.Nb; ++a) { if (inf[a].level == 0) { ++nroots; if (a < rootk) ++rootrank; } if (inf[a].level == 1) ++ntops; }
        // x of the roots' body rows (Δv, Δω of every batch), posted by the root phase for everybody: behind the messages in KernelArgs::msg
        T* const xr_mine = msg + (size_t)(ntops + rootrank) * (size_t)NB * 36;
        TIO* const trash = (TIO*)(msg + (size_t)(ntops + nroots) * (size_t)NB * 36);   // where the stores of columns that do not exist go (and their loads come from)
        for (int a = 0; a < G.Nb; ++a) if (inf[a].level == 1) {
            if (a < top) ++topid;
            const int n_ = __builtin_popcountll(inf[a].bm[0]) + __builtin_popcountll(inf[a].bm[1]);
            maxn = n_ > maxn ? n_ : maxn;
        }
        if (q == 0 && k < G.Nb && lvl == 1) inf[k].topid = topid;
        wv.sync();
        T* const msg_top = msg + (size_t)topid * (size_t)NB * 36 + (size_t)(q & 1) * 18;      // (+ 36 per branch-local batch position)
        // ---- phase 1 ----
#ifdef DJ_PROF
        unsigned long long tp1 = wv.clock();
#endif
        {
        TG Lm[3][12], mq[6][3];                                // L11 − I and the parent rows' multipliers m (load_lu_up)
        load_lu_up(Lm, mq);
        // a supernode's messages to its parent are posted at the END of the step that produced them and read at the top of the next one
        // (no message registers live across the substitution); before the first step: zeros
        wv.sync();
        if (q < 2) { double* ms_ = mail_slot(qb, q);
#pragma unroll
            for (int i = 0; i < 3 * NC; ++i) ms_[i] = 0.0; }
        const int nsteps1 = G.maxlevel >= 1 ? maxn + G.maxlevel - 1 : 0;
        DJ_P2B();
        for (int t = 0; t < nsteps1; ++t) {
            const int ib = t - (G.maxlevel - lvl);                // position of this step's batch in the branch's list
            const bool have = k < G.Nb && lvl >= 1 && ib >= 0 && (rem0 | rem1) != 0ull;
            int b = 0;
            if (have) { if (rem0 != 0ull) { b = __builtin_ctzll(rem0); rem0 &= rem0 - 1ull; } else { b = 64 + __builtin_ctzll(rem1); rem1 &= rem1 - 1ull; } }
            const bool valid = active && have;
            // the children finished this batch in the previous step
            TG acc[3 * NC];
#pragma unroll
            for (int i = 0; i < 3 * NC; ++i) acc[i] = TG(0);
            {   // children -> parent through the mailbox (only the body-row roles carry anything)
                wv.sync();
                // (no branches: a slot without a child reads the quad's own, finite post and adds 0 times it)
#pragma unroll
                for (int ci = 0; ci < MAXCH; ++ci) if (ci < G.maxch) {   // (unrolled: a run-time index would put sp.child_lane0 into scratch; the test is uniform)
                    const bool use = valid && q < 2 && ci < sp.nchild;
                    const double* cs_ = mail_slot(use ? sp.child_lane0[ci] : qb, q & 1);
                    const TG uf = use ? TG(1) : TG(0);
#pragma unroll
                    for (int i = 0; i < 3 * NC; ++i) acc[i] += uf * TG(cs_[i]);
                }
            }
            DJ_P2E(0);
            const bool isS = b < nbs;
            const int kk = b >> 1, typ = b & 1;
            const bool mine = valid && isS && (k == kk), par = valid && isS && typ == 0 && has_parent && (sp.parent == kk);
            const int cu0 = NC * (b - nbs) - sp.u_off;            // control batch: local input index of column 0
            int u_off_ = 0; TG um_s = TG(0);
            if constexpr (MODE == 0) {
                u_off_ = isS ? (mine ? RH::UOWN : RH::UPAR) + qh * 18 : RH::UA + qh * 18 + cu0;
                um_s = ((mine || par) && typ == 0 && q < 2) ? TG(1) : TG(0);
            }
            TIO* const cb = colbase(valid ? b : 0, k, q);            // never a null select: the pointer must stay a GLOBAL pointer (flat stores tie up lgkmcnt)
            // the six columns of the batch together: right-hand sides, one forward substitution of all six, then the messages
            TG r3a[NC][3];
            sweep_rhs<MODE, NC>(r3a, R, k, sp.parent, has_parent, sp.u_off, myu, sp.nlim_r, TG(wk), sp.ncontact, sp.contact, b, nbs, valid, acc);
            DJ_P2E(1);
            lu_forward_quad<NC>(Lm, r3a);                         // ỹ = L11⁻¹ r
            DJ_P2E(2);
            T* const mg = msg_top + (size_t)(ib > 0 ? ib : 0) * 36;
            const bool to_root = valid && lvl == 1 && q < 2;      // the parent is a root: it reads the message in phase 2
#pragma unroll
            for (int cI = 0; cI < NC; ++cI) {
                const int cx = (isS && cI >= 3) ? cI + 3 : cI;      // column index inside the batch's block of the output buffer
                const TG (&yy)[3] = r3a[cI];
                TG ua[3] = {TG(0), TG(0), TG(0)};                 // the direct right-hand side of the parent's body rows (read here: not live across the substitution)
                if constexpr (MODE == 0) {
                    const bool uok = !isS && valid && q < 2 && (cu0 + cI) >= 0 && (cu0 + cI) < myu;
                    const TG um = isS ? um_s : (uok ? TG(1) : TG(0));
                    const int iu = (isS || uok) ? u_off_ + cI : 0;
#pragma unroll
                    for (int i = 0; i < 3; ++i) ua[i] = um * TG(R.a[iu + i * 6]);
                }
                TG part[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) part[i] = mq[i][0] * yy[0] + mq[i][1] * yy[1] + mq[i][2] * yy[2];
#pragma unroll
                for (int i = 0; i < 6; ++i) part[i] += wv.quad_xor(part[i], 1);
#pragma unroll
                for (int i = 0; i < 6; ++i) part[i] += wv.quad_xor(part[i], 2);
                // the message to the parent's body rows, posted at once (an invalid step posts zeros: its batch is out of range for the
                // parent's next step as well)
                if (cI == 0) wv.sync();                             // (every supernode has read its children's messages of the previous step)
                TG ms3[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) { const TG p0_ = part[i], p1_ = part[3 + i]; ms3[i] = (valid && has_parent) ? ua[i] - (q == 0 ? p0_ : p1_) : TG(0); }
                if (q < 2) { double* ms_ = mail_slot(qb, q);
#pragma unroll
                    for (int i = 0; i < 3; ++i) ms_[3 * cI + i] = (double)ms3[i]; }
                if (to_root) { mg[3 * cI] = T(ms3[0]); mg[3 * cI + 1] = T(ms3[1]); mg[3 * cI + 2] = T(ms3[2]); }
                if (valid) {
                    if (col_ok(b, cI) && y_nz(b)) {
                        if constexpr (ypk) { T* yo = yp0 + (size_t)((b * NC + cI) * 3) * yW; yo[0] = T(yy[0]); yo[yW] = T(yy[1]); yo[2 * yW] = T(yy[2]); }
                        else { TIO* o = cb + (size_t)cx * nx + prk_off; o[0] = TIO(yy[0]); o[1] = TIO(yy[1]); o[2] = TIO(yy[2]); }
                    }
                }
            }
            DJ_P2E(3);
        }
        }
#ifdef DJ_PROF
        pc[3] += wv.clock() - tp1; tp1 = wv.clock();
#endif
        // ---- phase 2: the roots ----
        wv.sync_mem();                                            // (the messages of phase 1 and the staged factors are read by other lanes)
        {
        const int nrounds = (NB + G.S - 1) / G.S;
        const bool env_ok = (env < A.B);
        int rrank = 0;                                            // rank of the root at hand among the roots
        for (int a = 0; a < G.Nb; ++a) {
            if (inf[a].level != 0) continue;                      // (uniform: the tables are the same in every environment)
            const int dl = base + 4 * a + q - wv.lane();          // from this lane to the root's lane of the same role
            TG Lm[3][12];
            { LuPtr o = lu_base(dl); const size_t S_ = lu_S();
#pragma unroll
              for (int i = 0; i < 3; ++i)
#pragma unroll
                  for (int j = 0; j < 12; ++j) Lm[i][j] = TL(o[(size_t)(LU_LM + 12 * i + j) * S_]); }
            TG Um[3][12], di[3];
            { LuPtr o = lu_base(dl); const size_t S_ = lu_S();
#pragma unroll
              for (int i = 0; i < 3; ++i) {
#pragma unroll
                  for (int j = 0; j < 12; ++j) Um[i][j] = TL(o[(size_t)(LU_UM + 12 * i + j) * S_]);
                  di[i] = TL(o[(size_t)(LU_DI + i) * S_]); } }
            T* const xr_root = msg + (size_t)(ntops + rrank) * (size_t)NB * 36;
            const RH& Rr = *((const RH*)&R + (a - k));             // the root's right-hand-side block
            const SweepInfo& ri = inf[a];
            int rcontact[8];
#pragma unroll
            for (int c_ = 0; c_ < 8; ++c_) rcontact[c_] = ri.contact[c_];
            for (int rd = 0; rd < nrounds; ++rd) {
                const int b = rd * G.S + k;
                const bool inb = b < NB;
                const bool valid = env_ok && inb && (((b < 64 ? ri.bm[0] >> (b < 64 ? b : 0) : ri.bm[1] >> (b >= 64 ? b - 64 : 0)) & 1ull) != 0);
                TG acc[3 * NC];
#pragma unroll
                for (int i = 0; i < 3 * NC; ++i) acc[i] = TG(0);
                for (int ci = 0; ci < ri.nchild; ++ci) {
                    const SweepInfo& cinf = inf[ri.child[ci]];
                    const unsigned long long c0 = cinf.bm[0], c1 = cinf.bm[1];
                    const bool cnz = valid && q < 2 && (((b < 64 ? c0 >> (b < 64 ? b : 0) : c1 >> (b >= 64 ? b - 64 : 0)) & 1ull) != 0);
                    // position of b among the child's batches = the slot its phase-1 step wrote
                    const int pos = b < 64 ? __builtin_popcountll(c0 & ((1ull << (b < 64 ? b : 0)) - 1ull))
                                           : __builtin_popcountll(c0) + __builtin_popcountll(c1 & ((1ull << (b >= 64 ? b - 64 : 0)) - 1ull));
                    const T* mg = msg + ((size_t)cinf.topid * (size_t)NB + (size_t)(cnz ? pos : 0)) * 36 + (size_t)(q & 1) * 18;
                    if (cnz) {
#pragma unroll
                        for (int i = 0; i < 3 * NC; ++i) acc[i] += TG(mg[i]);
                    }
                }
                TG r3a[NC][3];
                sweep_rhs<MODE, NC>(r3a, Rr, a, -1, false, ri.u_off, ri.myu, ri.nlim_r, TG(ri.wk), ri.ncontact, rcontact, inb ? b : 0, nbs, valid, acc);
                lu_forward_quad<NC>(Lm, r3a);                     // ỹ of the root ...
                // ... and at once x = U11⁻¹ ỹ (a root has no parent term): ỹ never leaves the registers; Δv, Δω of the root go to xr for the whole tree
#pragma unroll
                for (int cI = 0; cI < NC; ++cI) { const bool ok_ = valid && col_ok(b, cI);
#pragma unroll
                    for (int i = 0; i < 3; ++i) r3a[cI][i] = ok_ ? r3a[cI][i] : TG(0); }
                lu_backward_quad<NC>(Um, di, r3a);
                if (valid && q < 2) {
                    T* xo = xr_root + (size_t)b * 36 + 18 * q;
#pragma unroll
                    for (int cI = 0; cI < NC; ++cI)
#pragma unroll
                        for (int i = 0; i < 3; ++i) xo[3 * cI + i] = T(r3a[cI][i]);
                }
            }
            ++rrank;
        }
        }
        wv.sync_mem();                                            // (what the root phase left in global memory is read by other lanes)
#ifdef DJ_PROF
        pc[7] += wv.clock() - tp1;
        unsigned long long td0 = wv.clock();
#endif
        // ---------------- down-sweep (roots -> leaves) ----------------
        // x = U11⁻¹ (ỹ − T x_parent).  The roots are done (root phase above); x is dense, but where ỹ vanishes in a whole branch -- every batch
        // that is not one of the branch's own -- the body rows (Δv, Δω) of a supernode are a fixed linear map of its root's:
        //   x_body = P x_body(parent),  P = body rows of −U11⁻¹ T;   Q = P Q(parent) down the tree,   x_body = Q x_body(root).
        // So:  phase A  every branch marches its OWN batches through its supernodes with the full backward substitution, all branches at
        //               once (the mirror image of the up-sweep's phase 1; the level-1 supernodes take x of their root from xr);
        //      phase B  every supernode (the roots too: Q = I) writes its rows of all other batches as Q times xr -- 36 multiply-adds per
        //               column and body instead of a substitution step, no exchange, no dependency chain.
        static_assert(NC == 6, "the Q maps are built as one batch of six columns");
        TG Um[3][12], Tq[3][6], di[3];                          // D⁻¹U11 − I, T = L11⁻¹U and the reciprocal pivots (load_lu_down)
        load_lu_down(Um, Tq, di);
        TG Qr[3][6];                                            // roles 0 / 1: this lane's rows (Δv / Δω) of Q
        {
            TG w3[NC][3];
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int i = 0; i < 3; ++i) w3[n][i] = -Tq[i][n];
            lu_backward_quad<NC>(Um, di, w3);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int n = 0; n < NC; ++n) Qr[i][n] = lvl == 0 ? ((n == 3 * q + i) ? TG(1) : TG(0)) : w3[n][i];
        }
        // Q = U11⁻¹(−T Q(parent)), level by level: the substitution is applied to T Q(parent) -- the product (U11⁻¹ T) Q(parent) of explicitly
        // formed factors loses the digits of a stiff supernode (measured on tests/golden/hard_cases_ant.npz: 5e-7 instead of 1e-8 relative)
        for (int lev = 2; lev <= G.maxlevel; ++lev) {
            TG qf[18];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) qf[6 * i + j] = Qr[i][j];
            mail_post_roles<18>(qf);
            const double* p0 = mail_slot(pb, 0); const double* p1 = mail_slot(pb, 1);
            TG w3[NC][3];
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    TG a_ = TG(0);
#pragma unroll
                    for (int m_ = 0; m_ < 3; ++m_) a_ -= Tq[i][m_] * TG(p0[6 * m_ + n]) + Tq[i][3 + m_] * TG(p1[6 * m_ + n]);
                    w3[n][i] = a_;
                }
            lu_backward_quad<NC>(Um, di, w3);
            const bool at = lvl == lev;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) Qr[i][j] = at ? w3[j][i] : Qr[i][j];
        }
        // ---- phase A: the branches' own batches ----
        // (as in the up-sweep: Δv, Δω of a batch are posted at the end of the step that solved them)
        wv.sync();
        if (q < 2) { double* ms_ = mail_slot(qb, q);
#pragma unroll
            for (int i = 0; i < 3 * NC; ++i) ms_[i] = 0.0; }
        T Mq[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Mq[i] = (q & 1) == 0 ? ((i % 4 == 0) ? dt : T(0)) : kb0.Phi[i];
        {
        // the parked ỹ of the NEXT step's batch, and (level 1) this lane's quarter of its root's x, are fetched while this step computes (the
        // loads are one HBM / L2 round trip away and nothing else hides it with one wave per SIMD)
        typedef typename std::conditional<ypk, TG, TIO>::type TY;
        TY ynext[NC][3];
        T xnext[9];
        double* const stage = (double*)gb_lds;                  // [2 roles][18]: the root's Δv, Δω of the current batch (level 1; the right-hand sides are dead)
        unsigned long long rem0 = tbm0, rem1 = tbm1;
        int bnxt = -1;
        auto advance = [&](int tn) {                              // pops the batch of step tn into bnxt and fetches what it needs
            const int in_ = tn - (lvl - 1);
            bnxt = -1;
            if (k < G.Nb && lvl >= 1 && in_ >= 0 && (rem0 | rem1) != 0ull) {
                if (rem0 != 0ull) { bnxt = __builtin_ctzll(rem0); rem0 &= rem0 - 1ull; } else { bnxt = 64 + __builtin_ctzll(rem1); rem1 &= rem1 - 1ull; }
            }
            const int b_ = bnxt;
            const bool v_ = active && b_ >= 0 && y_nz(b_ >= 0 ? b_ : 0);
            TIO* const cbn = colbase(v_ ? b_ : 0, k, q) + prk_off;
            const bool isS_ = b_ < nbs;
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                const bool ok_ = v_ && col_ok(b_, n);
                if constexpr (ypk) {
                    const T* yi = yp0 + (size_t)(((v_ ? b_ : 0) * NC + n) * 3) * yW;
#pragma unroll
                    for (int i = 0; i < 3; ++i) ynext[n][i] = ok_ ? TG(yi[i * yW]) : TG(0);
                } else {
                    const TIO* o_ = cbn + (size_t)((isS_ && n >= 3) ? n + 3 : n) * nx;
#pragma unroll
                    for (int i = 0; i < 3; ++i) ynext[n][i] = ok_ ? o_[i] : TIO(0);
                }
            }
            const bool x_ = active && lvl == 1 && b_ >= 0;
            const T* xs = xr_mine + (size_t)(x_ ? b_ : 0) * 36 + 9 * q;
#pragma unroll
            for (int j = 0; j < 9; ++j) xnext[j] = x_ ? xs[j] : T(0);
        };
        advance(0);
        const int nstepsA = G.maxlevel >= 1 ? maxn + G.maxlevel - 1 : 0;
        DJ_P2B();
        for (int t = 0; t < nstepsA; ++t) {
            const int b = bnxt < 0 ? 0 : bnxt;
            const bool valid = active && bnxt >= 0;
            TY ycur[NC][3];
#pragma unroll
            for (int n = 0; n < NC; ++n) for (int i = 0; i < 3; ++i) ycur[n][i] = ynext[n][i];
            T xcur[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) xcur[j] = xnext[j];
            advance(t + 1);
            DJ_P2E(4);
            const bool isS = b < nbs;
            const bool mine = valid && isS && (k == (b >> 1)) && ((b & 1) == 0);
            TIO* const cb = colbase(b, k, q);                     // never a null select: the pointer must stay a GLOBAL pointer (flat stores tie up lgkmcnt)
            // the parent finished this batch in the previous step: its two body-row roles posted Δv, Δω of the six columns
            wv.sync();
            if (lvl == 1) {
#pragma unroll
                for (int j = 0; j < 9; ++j) stage[9 * q + j] = (double)xcur[j];
            }
            wv.sync();
            // x = U11⁻¹ (ỹ − T x_parent) of the six columns
            TG x3[NC][3];
            {
                const double* p0 = lvl == 1 ? stage : mail_slot(pb, 0); const double* p1 = lvl == 1 ? stage + 18 : mail_slot(pb, 1);
#pragma unroll
                for (int n = 0; n < NC; ++n) {
                    const bool ok_ = valid && col_ok(b, n);
                    TG pall[6];
#pragma unroll
                    for (int i = 0; i < 3; ++i) { pall[i] = TG(p0[3 * n + i]); pall[3 + i] = TG(p1[3 * n + i]); }
#pragma unroll
                    for (int i = 0; i < 3; ++i) { TG a_ = ok_ ? TG(ycur[n][i]) : TG(0);
#pragma unroll
                        for (int j = 0; j < 6; ++j) a_ -= Tq[i][j] * pall[j];
                        x3[n][i] = a_; }
                }
            }
            DJ_P2E(5);
            lu_backward_quad<NC>(Um, di, x3);
            DJ_P2E(6);
            wv.sync();                                            // (every supernode has read its parent's Δv, Δω of the previous step)
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                // posted at once (an invalid step posts zeros: its batch is out of range for the children's next step as well)
                TG d3[1][3];
#pragma unroll
                for (int i = 0; i < 3; ++i) d3[0][i] = valid ? x3[n][i] : TG(0);
                if (q < 2) { double* ms_ = mail_slot(qb, q);
#pragma unroll
                    for (int i = 0; i < 3; ++i) ms_[3 * n + i] = (double)d3[0][i]; }
                {
                    // one code path for both body-row roles: rows [Mq d (+ identity term); d] with Mq = Δt I (role 0: x3 rows) or Φ (role 1: φ3 rows);
                    // the joint roles, idle steps and columns that do not exist store to the trash slot (no branch around the stores)
                    TIO* const o = (valid && q < 2 && col_ok(b, n)) ? cb + (size_t)((isS && n >= 3) ? n + 3 : n) * nx : trash;
                    T d_[3] = {T(d3[0][0]), T(d3[0][1]), T(d3[0][2])};
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        T x = Mq[3 * i] * d_[0] + Mq[3 * i + 1] * d_[1] + Mq[3 * i + 2] * d_[2];
                        const T idt = q == 0 ? (n == i ? T(1) : T(0)) : (n >= 3 ? kb0.Xi[3 * i + (n >= 3 ? n - 3 : 0)] : T(0));
                        if (mine) x += idt;
                        o[i] = TIO(x); o[3 + i] = TIO(d_[i]);
                    }
                }
            }
            DJ_P2E(7);
        }
        }
        // ---- phase B: everybody's rows of the batches of the OTHER branches (a root: of all its batches) ----
#ifdef DJ_PROF
        unsigned long long tpb = wv.clock();
#endif
        {
            const int rq = q & 1, half = q >> 1;                  // lanes 2, 3 work as roles 0, 1 on every other batch
            TG Qb[3][6];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) { const TG o_ = wv.quad_xor(Qr[i][j], 2); Qb[i][j] = q < 2 ? Qr[i][j] : o_; }
            unsigned long long f0 = 0ull, f1 = 0ull;            // the root's batches that are not this branch's own
            if (active) { f0 = inf[rootk].bm[0] & ~tbm0; f1 = inf[rootk].bm[1] & ~tbm1; }
            int cnt = 0;
            auto pop = [&]() -> int {                             // this lane's next batch: every other one of the set
                while ((f0 | f1) != 0ull) {
                    int b_;
                    if (f0 != 0ull) { b_ = __builtin_ctzll(f0); f0 &= f0 - 1ull; } else { b_ = 64 + __builtin_ctzll(f1); f1 &= f1 - 1ull; }
                    if (((cnt++) & 1) == half) return b_;
                }
                return -1;
            };
            // Software pipeline: the root's x of the NEXT batch is loaded while this one is computed, and every load and store of an iteration
            // is issued unconditionally (columns that do not exist go to a trash slot behind xr): the waits then count instructions instead of
            // draining the memory pipeline (vmcnt(0)), which with one wave per SIMD cost a full round trip per batch
            T xc[36];
            int bc = pop();
            { const T* xb = xr_mine + (size_t)(bc >= 0 ? bc : 0) * 36;
#pragma unroll
              for (int j = 0; j < 36; ++j) xc[j] = xb[j]; }
            while (bc >= 0) {
                const int b = bc;
                const int bn = pop();
                T xn[36];
                { const T* xb = xr_mine + (size_t)(bn >= 0 ? bn : 0) * 36;
#pragma unroll
                  for (int j = 0; j < 36; ++j) xn[j] = xb[j]; }
                const bool isS = b < nbs;
                const bool own = isS && (k == (b >> 1)) && ((b & 1) == 0);     // (a root's own configuration batch: the identity terms)
                TIO* const cb = colbase(b, k, rq);
#pragma unroll
                for (int n = 0; n < NC; ++n) {
                    T d_[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) { TG a_ = TG(0);
#pragma unroll
                        for (int j = 0; j < 3; ++j) a_ += Qb[i][j] * TG(xc[3 * n + j]) + Qb[i][3 + j] * TG(xc[18 + 3 * n + j]);
                        d_[i] = T(a_); }
                    TIO* const o = col_ok(b, n) ? cb + (size_t)((isS && n >= 3) ? n + 3 : n) * nx : trash;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        T x = Mq[3 * i] * d_[0] + Mq[3 * i + 1] * d_[1] + Mq[3 * i + 2] * d_[2];
                        const T idt = rq == 0 ? (n == i ? T(1) : T(0)) : (n >= 3 ? kb0.Xi[3 * i + (n >= 3 ? n - 3 : 0)] : T(0));
                        if (own) x += idt;
                        o[i] = TIO(x); o[3 + i] = TIO(d_[i]);
                    }
                }
                bc = bn;
#pragma unroll
                for (int j = 0; j < 36; ++j) xc[j] = xn[j];
            }
            // A mechanism with several trees (bodies hanging on the origin independently): the batches of the OTHER trees never reach this one -- their
            // blocks of the Jacobians are zero, and every entry of the output has to be written (uniform test: one tree, nothing to do)
            if (nroots > 1) {
                unsigned long long z0 = 0ull, z1 = 0ull;
                if (active) {
                    const unsigned long long all0 = NB >= 64 ? ~0ull : ((1ull << NB) - 1ull), all1 = NB > 64 ? ((NB >= 128 ? ~0ull : (1ull << (NB - 64)) - 1ull)) : 0ull;
                    z0 = all0 & ~inf[rootk].bm[0]; z1 = all1 & ~inf[rootk].bm[1];
                }
                int cz = 0;
                while ((z0 | z1) != 0ull) {
                    int b;
                    if (z0 != 0ull) { b = __builtin_ctzll(z0); z0 &= z0 - 1ull; } else { b = 64 + __builtin_ctzll(z1); z1 &= z1 - 1ull; }
                    if (((cz++) & 1) != half) continue;
                    const bool isS = b < nbs;
                    TIO* const cb = colbase(b, k, rq);
#pragma unroll
                    for (int n = 0; n < NC; ++n) if (col_ok(b, n)) {
                        TIO* const o = cb + (size_t)((isS && n >= 3) ? n + 3 : n) * nx;
#pragma unroll
                        for (int i = 0; i < 6; ++i) o[i] = TIO(0);
                    }
                }
            }
        }
#ifdef DJ_PROF
        pc[4] += wv.clock() - td0; pc[2] = wv.clock() - tpb;
#endif
    }

    // quad mapping: solve with the distributed factors
    DJ_HD void solve_quad(const T* rk, const T* upv, T* dk_out, T* dva_out) {
        TL r3[3], y3[3] = {0, 0, 0}, send3[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < 3; ++i) {   // values first, then the select: `q == 0 ? rk[i] : ...` on lvalues becomes a select of POINTERS and keeps rk[] in scratch
            const T a_ = rk[i], b_ = rk[3 + i], c_ = rk[6 + i], d_ = rk[9 + i];
            r3[i] = TL(q == 0 ? a_ : q == 1 ? b_ : q == 2 ? c_ : d_);
        }
        // forward: leaves -> root
        for (int lev = G.maxlevel; lev >= 0; --lev) {
            const bool at = active && P.level == lev;
            TL acc[3] = {0, 0, 0};
            mail_post_roles<3>(send3);
            mail_add_children<3>(acc, at, G.maxch_at(lev));
            if (at) { r3[0] += acc[0]; r3[1] += acc[1]; r3[2] += acc[2]; }
            TL rf[12];
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int i = 0; i < 3; ++i) rf[3 * o + i] = wv.quad_bcast(r3[i], o);
            TL yy[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { TL a_ = TL(0);
#pragma unroll
                for (int m_ = 0; m_ < 12; ++m_) a_ += F.Sq[i][m_] * rf[m_];
                yy[i] = a_; }
            if (at) { y3[0] = yy[0]; y3[1] = yy[1]; y3[2] = yy[2]; }
            if (lev > 0) {   // the roots have nothing to pass up (wave-uniform skip)
                TL part[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) part[i] = F.Lq[i][0] * yy[0] + F.Lq[i][1] * yy[1] + F.Lq[i][2] * yy[2];
#pragma unroll
                for (int i = 0; i < 6; ++i) part[i] += wv.quad_xor(part[i], 1);
#pragma unroll
                for (int i = 0; i < 6; ++i) part[i] += wv.quad_xor(part[i], 2);
                if (at && has_parent) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const TL s0_ = TL(upv[i]) - part[i], s1_ = TL(upv[3 + i]) - part[3 + i];
                        send3[i] = (q == 0) ? s0_ : (q == 1) ? s1_ : TL(0);
                    }
                }
            }
        }
        DJ_P2E(2);
        // backward: root -> leaves
        TL d3[3] = {y3[0], y3[1], y3[2]};
        TL dva[6] = {0, 0, 0, 0, 0, 0};
        const int pb = has_parent ? base + stride * P.parent : qb;
        for (int lev = 1; lev <= G.maxlevel; ++lev) {
            const bool at = active && P.level == lev && has_parent;
            TL pa_[6];
            mail_post_roles<3>(d3);
            { const double* p0 = mail_slot(pb, 0); const double* p1 = mail_slot(pb, 1);
#pragma unroll
              for (int i = 0; i < 3; ++i) { pa_[i] = TL(p0[i]); pa_[3 + i] = TL(p1[i]); } }
            TL t3[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { TL a_ = TL(0);
#pragma unroll
                for (int j = 0; j < 6; ++j) a_ += F.Uq[i][j] * pa_[j];
                t3[i] = a_; }
            TL tf[12];
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int i = 0; i < 3; ++i) tf[3 * o + i] = wv.quad_bcast(t3[i], o);
            if (at) {
#pragma unroll
                for (int i = 0; i < 6; ++i) dva[i] = pa_[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) { TL a_ = TL(0);
#pragma unroll
                    for (int m_ = 0; m_ < 12; ++m_) a_ += F.Sq[i][m_] * tf[m_];
                    d3[i] = y3[i] - a_; }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int i = 0; i < 3; ++i) dk_out[3 * o + i] = T(wv.quad_bcast(d3[i], o));
#pragma unroll
        for (int i = 0; i < 6; ++i) dva_out[i] = T(dva[i]);
        DJ_P2E(3);
    }

    // ---------------------------------------------------------------- solve with the current factors
    // Generic right-hand side: rk0 = rhs of the body (6) and joint-equality (6) rows, R = rhs of the
    // cone (complementarity) rows, rs = rhs of the two limit slack rows, r58 = rhs of the contact
    // constraint rows, upx = direct rhs contribution to the parent's body rows.
    DJ_HD void solve_rhs(const T* rk0, const ConeRhs& R, const T* rs, const T (*r58)[NCV], const T* upx, StepT& D, T* dva_out = nullptr) {
        T rk[12], up[6];
        for (int i = 0; i < 12; ++i) rk[i] = rk0[i];
        for (int i = 0; i < 6; ++i) up[i] = upx[i];
        CCoef Q[CPL];
        T rkc[6] = {0, 0, 0, 0, 0, 0};                            // (split contacts: this lane's share of the condensed right-hand side)
#pragma unroll
        for (int li = 0; li < CPL; ++li) {
            const int c = cidx(li);
            if (c < P.ncontact) {
                contact_coef(Q[li], c, R.cc[li], r58[li]);
                if constexpr (kSplitC) { for (int i = 0; i < 6; ++i) for (int a = 0; a < 3; ++a) rkc[i] += ccold(c).G134[6 * a + i] * Q[li].k0[a]; }
                else { for (int i = 0; i < 6; ++i) for (int a = 0; a < 3; ++a) rk[i] += ccold(c).G134[6 * a + i] * Q[li].k0[a]; }
#if DJ_SS
                if constexpr (kSS) { for (int i = 0; i < 6; ++i) for (int a = 0; a < 3; ++a) up[i] += ccold(c).Gp134[6 * a + i] * Q[li].k0[a]; }      // (zero rows for a half-space contact)
#endif
            }
        }
        if constexpr (kSplitC) { quad_sum(rkc); for (int i = 0; i < 6; ++i) rk[i] += rkc[i]; }
#if DJ_CUT && DJ_SS
        if constexpr (kCut) { for (int e = 0; e < NCUT; ++e) if (e < ncut && cut_is_contact(e)) {     // a cut contact's condensed right-hand side on the rows of its other body
            T v6[6] = {0, 0, 0, 0, 0, 0}, got[6];
            if (cut_owner(e)) for (int li = 0; li < CPL; ++li) if (cidx(li) < P.ncontact && P.contact[cidx(li)] == cutp[e].contact[0])
                for (int i = 0; i < 6; ++i) for (int a = 0; a < 3; ++a) v6[i] += ccold(cidx(li)).Gp134[6 * a + i] * Q[li].k0[a];
            shfl_vec<6>(wv, got, v6, base + cut_b(e));
            if (active && k == cut_a(e)) for (int i = 0; i < 6; ++i) rk[i] += got[i];
        } }
#endif
        T kap0 = T(0), rsu = rs[0], rsl = rs[1], su = T(0), sl = T(0), gu = T(0), gl = T(0), isu = T(0), isl = T(0);
        if (lim_on()) {
            su = L.ls[0] + T(REG); sl = L.ls[1] + T(REG); gu = L.lg[0] + T(REG); gl = L.lg[1] + T(REG);
            isu = trcp(su); isl = trcp(sl);
            kap0 = (R.lim[1] - gl * rsl) * isl - (R.lim[0] - gu * rsu) * isu;
            const T rkap = kap0 * trcp(T(1) + gl * isl + gu * isu);   // the Δκ row (see evaluate)
            if (tlim) rk[8] = rkap; else rk[11] = rkap;
        }
#if DJ_MLIM
        if constexpr (kMLim) { for (int m = 0; m < NLM; ++m) if (m < nlm()) {     // the Δκ rows of the limited coordinates (condense_limits)
            const T su_ = L.mls[m][0] + T(REG), sl_ = L.mls[m][1] + T(REG), gu_ = L.mlg[m][0] + T(REG), gl_ = L.mlg[m][1] + T(REG);
            const T isu_ = trcp(su_), isl_ = trcp(sl_);
            const T k0_ = (R.mlim[m][1] - gl_ * mrs[m][1]) * isl_ - (R.mlim[m][0] - gu_ * mrs[m][0]) * isu_;
            rk[mslot(m)] = k0_ * trcp(T(1) + gl_ * isl_ + gu_ * isu_);
        } }
#endif
        T dk[12], dva[6];
        DJ_P2E(1);
#if DJ_CUT
        if constexpr (kCut) {
            for (int c = 0; c < NCUT; ++c) for (int i = 0; i < 6; ++i) D.dclam[c][i] = T(0);
            if (ncut > 0) cut_solve(rk, up, crc, dk, dva, D.dclam); else core_solve(rk, up, dk, dva);
        } else
#endif
        core_solve(rk, up, dk, dva);
        for (int i = 0; i < 3; ++i) { D.dv[i] = dk[i]; D.dw[i] = dk[3 + i]; }
        for (int i = 0; i < 6; ++i) D.dlam[i] = dk[6 + i];
        if (dva_out != nullptr) for (int i = 0; i < 6; ++i) dva_out[i] = dva[i];
#if DJ_MLIM
        if constexpr (kMLim) { for (int m = 0; m < NLM; ++m) {
            if (m < nlm()) {
                D.dlam[mslot(m) - 6] = T(0);                         // the slot carried Δκ, not a joint multiplier
                T thd = T(0);
                for (int j = 0; j < 3; ++j) thd += mth_a[m][j] * dva[j] + mth_a[m][3 + j] * dva[3 + j] + mth_b[m][j] * D.dv[j] + mth_b[m][3 + j] * D.dw[j];
                const T su_ = L.mls[m][0] + T(REG), sl_ = L.mls[m][1] + T(REG), gu_ = L.mlg[m][0] + T(REG), gl_ = L.mlg[m][1] + T(REG);
                D.dmls[m][0] = mrs[m][0] - thd; D.dmls[m][1] = mrs[m][1] + thd;
                D.dmlg[m][0] = (R.mlim[m][0] - gu_ * D.dmls[m][0]) * trcp(su_);
                D.dmlg[m][1] = (R.mlim[m][1] - gl_ * D.dmls[m][1]) * trcp(sl_);
            } else { D.dmls[m][0] = D.dmls[m][1] = D.dmlg[m][0] = D.dmlg[m][1] = T(0); }
        } }
#endif
        if (lim_on()) { if (tlim) D.dlam[2] = T(0); else D.dlam[5] = T(0); }   // slot 11 (8) carried Δκ, not a joint multiplier
        // recovery of the condensed variables
        if (lim_on()) {
            T thd = v3dot(F.th_a, dva + 3) + v3dot(F.th_b, D.dw);
#if DJ_TSD
            if (tlim) thd += v3dot(F.thv_a, dva) + v3dot(F.thv_b, D.dv);
#endif
            D.dls[0] = rsu - thd; D.dls[1] = rsl + thd;
            D.dlg[0] = (R.lim[0] - gu * D.dls[0]) * isu;
            D.dlg[1] = (R.lim[1] - gl * D.dls[1]) * isl;
        } else { D.dls[0] = D.dls[1] = D.dlg[0] = D.dlg[1] = T(0); }
#if DJ_CUT && DJ_SS
        T cdwa[NCUT][6];                                       // Δ(v, ω) of a cut contact's other body, for the recovery of its cone variables
        for (int e = 0; e < NCUT; ++e) for (int i = 0; i < 6; ++i) cdwa[e][i] = T(0);
        if constexpr (kCut) { for (int e = 0; e < NCUT; ++e) if (e < ncut && cut_is_contact(e)) {
            T own6[6] = {D.dv[0], D.dv[1], D.dv[2], D.dw[0], D.dw[1], D.dw[2]};
            shfl_vec<6>(wv, cdwa[e], own6, base + cut_a(e));
        } }
#endif
#pragma unroll
        for (int li = 0; li < CPL; ++li) {
            const int c = cidx(li);
            if (c < P.ncontact) {
                const ContactP<T>& K = CP[P.contact[c]];
                T cw[3];
                for (int a = 0; a < 3; ++a) { cw[a] = T(0); for (int j = 0; j < 3; ++j) cw[a] += ccold(c).C134[6 * a + j] * D.dv[j] + ccold(c).C134[6 * a + 3 + j] * D.dw[j]; }
#if DJ_CUT && DJ_SS
                if constexpr (kCut) { if (K.kind == 1) { const int e = cut_of_contact(P.contact[c]);
                    if (e >= 0) for (int a = 0; a < 3; ++a) for (int j = 0; j < 6; ++j) cw[a] += ccold(c).Cp134[6 * a + j] * cdwa[e][j]; } }
#endif
#if DJ_SS
                if constexpr (kSS) { for (int a = 0; a < 3; ++a) for (int j = 0; j < 6; ++j) cw[a] += ccold(c).Cp134[6 * a + j] * dva[j]; }
#endif
                const CCoef& q = Q[li];
                if constexpr (kLinear) {                               // recovery of the twelve LinearContact variables (see contact_coef)
                    const T n_ = cw[0], t1_ = cw[1], t2_ = cw[2];
                    const T dgam = q.a1 + q.b1 * n_, dpsi = q.kpsi + q.pn * n_ + q.p1 * t1_ + q.p2 * t2_;
                    const T ct[4] = {t2_, -t2_, t1_, -t1_};
                    T sb = T(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const T db = q.le[i] - q.lw[i] * (ct[i] + dpsi);
                        D.dcg[li][2 + i] = db; D.dcs[li][2 + i] = ct[i] + dpsi - r58[li][2 + i]; sb += db;
                    }
                    D.dcg[li][0] = dgam; D.dcs[li][0] = n_ - r58[li][0];
                    D.dcg[li][1] = dpsi; D.dcs[li][1] = K.mu * dgam - sb - r58[li][1];
                    continue;
                }
                T ds1 = cw[0] - r58[li][0], ds3 = cw[1] - r58[li][2], ds4 = cw[2] - r58[li][3];
                T dg1 = q.a1 + q.b1 * cw[0];
                T dg2 = K.mu * dg1 - r58[li][1];
                T ih = trcp(q.h0);
                T r2p = R.cc[li][1] - (q.h1 * R.cc[li][2] + q.h2 * R.cc[li][3]) * ih;
                T ds2 = (r2p - q.al3 * ds3 - q.al4 * ds4 - q.al2 * dg2) * trcp(q.den);
                T dg3 = (R.cc[li][2] - q.g1 * ds2 - q.g0 * ds3 - q.h1 * dg2) * ih;
                T dg4 = (R.cc[li][3] - q.g2 * ds2 - q.g0 * ds4 - q.h2 * dg2) * ih;
                const T fz = G.contact_model == 1 ? T(0) : T(1);      // ImpactContact: the friction block stays at the neutral vector
                D.dcs[li][0] = ds1; D.dcs[li][1] = fz * ds2; D.dcs[li][2] = fz * ds3; D.dcs[li][3] = fz * ds4;
                D.dcg[li][0] = dg1; D.dcg[li][1] = fz * dg2; D.dcg[li][2] = fz * dg3; D.dcg[li][3] = fz * dg4;
            } else { for (int i = 0; i < NCV; ++i) D.dcs[li][i] = D.dcg[li][i] = T(0); }
        }
        DJ_P2E(4);
    }

    // forward / backward substitution through the tree with the current factors:
    // rk = rhs of this supernode's 12 rows (cone rows already condensed), up = direct rhs for the parent's body rows
    DJ_HD void core_solve(const T* rk, const T* up, T* dk, T* dva) {
        for (int i = 0; i < 6; ++i) dva[i] = T(0);
        if constexpr (QUAD) {
            solve_quad(rk, up, dk, dva);
        } else {
        // forward: leaves -> root (factorization precision)
        TL rl[12], send[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 12; ++i) rl[i] = TL(rk[i]);
        for (int lev = G.maxlevel; lev >= 0; --lev) {
            TL acc[6] = {0, 0, 0, 0, 0, 0};
            gather_children<6>(wv, acc, send, P, base, G.maxch_at(lev), active && P.level == lev, stride, q);
            if (active && P.level == lev) {
                for (int i = 0; i < 6; ++i) rl[i] += acc[i];
                for (int k = 0; k < 12; ++k) for (int i = k + 1; i < 12; ++i) rl[i] -= F.Sinv[12 * i + k] * rl[k];      // y = L11⁻¹ r
                if (has_parent) { TL t[6]; mv<6, 12>(t, F.W, rl); for (int i = 0; i < 6; ++i) send[i] = TL(up[i]) - t[i]; }
            }
        }
        // backward: root -> leaves, x = U11⁻¹ (y − Z Δv_parent)
        auto back = [&](TL* x) {
            for (int i = 11; i >= 0; --i) {
                TL a = x[i];
                for (int j = i + 1; j < 12; ++j) a -= F.Sinv[12 * i + j] * x[j];
                x[i] = a * F.Sinv[13 * i];
            }
        };
        TL dkl[12];
        for (int i = 0; i < 12; ++i) dkl[i] = rl[i];
        if (active && P.level == 0) back(dkl);
        for (int lev = 1; lev <= G.maxlevel; ++lev) {
            TL par[6];
            shfl_vec<6>(wv, par, dkl, plane);
            if (active && P.level == lev && has_parent) {
                for (int i = 0; i < 6; ++i) dva[i] = T(par[i]);
                TL t[12]; mv<12, 6>(t, F.Z, par);
                for (int i = 0; i < 12; ++i) dkl[i] = rl[i] - t[i];
                back(dkl);
            }
        }
        for (int i = 0; i < 12; ++i) dk[i] = T(dkl[i]);
        }
    }

    // Newton right-hand side: −residual with the given cone-row right-hand sides
    DJ_HD void solve(const ConeRhs& R, StepT& D) {
        T rk[12], rs[2] = {0, 0}, r58[CPL][NCV], upx[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 6; ++i) { rk[i] = -rb[i]; rk[6 + i] = -rj[i]; }
#pragma unroll
        for (int c = 0; c < CPL; ++c) for (int i = 0; i < NCV; ++i) r58[c][i] = -cres[c][i];
        if (lim_on()) {
            rs[0] = -(L.ls[0] - (lim_hi() - theta));       // limits.jl:13-14
            rs[1] = -(L.ls[1] - (theta - lim_lo()));
        }
#if DJ_CUT
        if constexpr (kCut) { for (int c = 0; c < NCUT; ++c) for (int i = 0; i < 6; ++i) crc[c][i] = c < ncut ? -crj[c][i] : T(0); }
#endif
#if DJ_MLIM
        if constexpr (kMLim) { for (int m = 0; m < NLM; ++m) {
            mrs[m][0] = m < nlm() ? -(L.mls[m][0] - (mlp->hi[m] - mtheta[m])) : T(0);
            mrs[m][1] = m < nlm() ? -(L.mls[m][1] - (mtheta[m] - mlp->lo[m])) : T(0); } }
#endif
        if constexpr (kRefine) {
            T dva[6];
            solve_rhs(rk, R, rs, r58, upx, D, dva);
            if (blk != nullptr && wv.any(refine)) {
#pragma unroll 1
                for (int rstep = 0; rstep < DJ_REFINE_STEPS; ++rstep) refine_solution(rk, R, rs, r58, upx, D, dva);
            }
        } else solve_rhs(rk, R, rs, r58, upx, D);
    }

    // ---------------------------------------------------------------- iterative refinement (DJ_REFINE)
    // The condensed body block D + Σ (γ/s) g cᵀ is formed in fp64, so with γ/s ~ 1e8 .. 1e12 (tight tolerances, strongly active
    // contacts and limits) the solve loses that many digits of D and the recovered Δγ = (γ/s)(k − cᵀΔw) inherits ε·γ/s·|cᵀΔw|.
    // One step of refinement against the UNCONDENSED equations restores them: every term of
    //     body rows      S⁰ [Δw; Δλ] + U⁰ Δw_a + Σ_children (L⁰ [Δw_c; Δλ_c] + Dup⁰ Δw) − Σ_contacts G Δγ₁₃₄  = r_body (+ Σ upx_c)
    //     joint rows     S⁰ [Δw; Δλ] + U⁰ Δw_a = r_joint
    //     cone rows      γ̃ Δs + s̃ Δγ = r_c   (orthant and arrow products, contacts and joint limits)
    // is O(1), so the residual is exact to ε, and the correction is one more condensed solve with that residual as its
    // right-hand side (solve_rhs is linear in rk0, R, rs, r58, upx).  The rows that define Δs₁, Δs₃, Δs₄, Δγ₂ and the limit
    // slacks are satisfied by construction.  S⁰ U⁰ L⁰ Dup⁰ = the rows store_blocks() parked in global memory.
    template <class BK>
    DJ_HD void store_blocks(const BK& K) {
        T* o = blk;
        const size_t W = Wave::kWidth > 0 ? (size_t)Wave::kWidth : (size_t)blk_stride;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 12; ++j) o[(size_t)(12 * i + j) * W] = T(K.S[i][j]);
#pragma unroll
            for (int j = 0; j < 6; ++j) { o[(size_t)(36 + 6 * i + j) * W] = T(K.U[i][j]); o[(size_t)(54 + 6 * i + j) * W] = T(K.L[j][i]); o[(size_t)(72 + 6 * i + j) * W] = T(K.D[i][j]); }
        }
    }
    DJ_HD void refine_solution(const T* rk0, const ConeRhs& R, const T* rs, const T (*r58)[NCV], const T* upx, StepT& D, T* dva) {
        // (an opaque base and a compile-time stride, as for the IFT's staged factors -- lu_base: otherwise the 90 addresses live from store_blocks to here, spilled)
#if defined(__HIP_DEVICE_COMPILE__)
        const T* blo = blk; DJ_OPAQUE(blo);
        const T __attribute__((address_space(1)))* const bl = (const T __attribute__((address_space(1)))*)blo;
#else
        const T* bl = blk;
#endif
        const size_t W = Wave::kWidth > 0 ? (size_t)Wave::kWidth : (size_t)blk_stride;
        T xk[12];
        for (int i = 0; i < 3; ++i) { xk[i] = D.dv[i]; xk[3 + i] = D.dw[i]; }
        for (int i = 0; i < 6; ++i) xk[6 + i] = D.dlam[i];
        if (lim_on()) { const T dkap = D.dlg[1] - D.dlg[0]; if (tlim) xk[8] = dkap; else xk[11] = dkap; }   // net limit impulse Δκ = Δγ_lo − Δγ_up
        // own rows 3q .. 3q+2
        T rho[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            T a_ = T(0);
#pragma unroll
            for (int j = 0; j < 12; ++j) a_ += bl[(size_t)(12 * i + j) * W] * xk[j];
#pragma unroll
            for (int j = 0; j < 6; ++j) a_ += bl[(size_t)(36 + 6 * i + j) * W] * dva[j];
            rho[i] = a_;
        }
        // what this supernode puts on its parent's body rows: L⁰ x (this lane: columns 3q .. 3q+2) + Dup⁰ Δw_a (roles 0, 1: rows 3q ..)
        T xq[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { const T a_ = xk[j], b_ = xk[3 + j], c_ = xk[6 + j], d_ = xk[9 + j]; xq[j] = q == 0 ? a_ : q == 1 ? b_ : q == 2 ? c_ : d_; }
        T m6[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) m6[r] = bl[(size_t)(54 + r) * W] * xq[0] + bl[(size_t)(54 + 6 + r) * W] * xq[1] + bl[(size_t)(54 + 12 + r) * W] * xq[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            T a_ = T(0);
#pragma unroll
            for (int j = 0; j < 6; ++j) a_ += bl[(size_t)(72 + 6 * i + j) * W] * dva[j];
            m6[i] += q == 0 ? a_ : T(0); m6[3 + i] += q == 1 ? a_ : T(0);
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) m6[r] += wv.quad_xor(m6[r], 1);
#pragma unroll
        for (int r = 0; r < 6; ++r) m6[r] += wv.quad_xor(m6[r], 2);
        T msg[6], acc[6] = {0, 0, 0, 0, 0, 0};
        for (int r = 0; r < 6; ++r) msg[r] = has_parent ? upx[r] - m6[r] : T(0);
        mail_post_node<6>(msg); mail_add_children_node<6>(acc, active, G.maxch);
        // contacts: −G Δγ₁₃₄ on the body rows; cone rows
        ConeRhs R2;
        T accc[6] = {0, 0, 0, 0, 0, 0};                           // (this lane's contacts' share)
#pragma unroll
        for (int li = 0; li < CPL; ++li) {
            const int c = cidx(li);
            if (c < P.ncontact) {
                const T* gam = L.cg[c]; const T* s = L.cs[c]; const T* ds = D.dcs[li]; const T* dg = D.dcg[li];
                const ContactCold<T>& cc_ = ccold(c);
                for (int i = 0; i < 6; ++i) accc[i] += cc_.G134[i] * dg[0] + cc_.G134[6 + i] * dg[2] + cc_.G134[12 + i] * dg[3];
                const T g1t = gam[0] + T(REG), s1t = s[0] + T(REG), g0 = gam[1] + T(REG), h0 = s[1] + T(REG);
                R2.cc[li][0] = R.cc[li][0] - (g1t * ds[0] + s1t * dg[0]);
                if (G.contact_model == 0) {
                    R2.cc[li][1] = R.cc[li][1] - (g0 * ds[1] + gam[2] * ds[2] + gam[3] * ds[3] + h0 * dg[1] + s[2] * dg[2] + s[3] * dg[3]);
                    R2.cc[li][2] = R.cc[li][2] - (gam[2] * ds[1] + g0 * ds[2] + s[2] * dg[1] + h0 * dg[2]);
                    R2.cc[li][3] = R.cc[li][3] - (gam[3] * ds[1] + g0 * ds[3] + s[3] * dg[1] + h0 * dg[3]);
                } else { R2.cc[li][1] = R2.cc[li][2] = R2.cc[li][3] = T(0); }
            } else { for (int i = 0; i < 4; ++i) R2.cc[li][i] = T(0); }
        }
        quad_sum(accc);
        for (int i = 0; i < 6; ++i) acc[i] += accc[i];
        R2.lim[0] = R2.lim[1] = T(0);
        if (lim_on()) {
            R2.lim[0] = R.lim[0] - ((L.lg[0] + T(REG)) * D.dls[0] + (L.ls[0] + T(REG)) * D.dlg[0]);
            R2.lim[1] = R.lim[1] - ((L.lg[1] + T(REG)) * D.dls[1] + (L.ls[1] + T(REG)) * D.dlg[1]);
        }
        // residual rows of this lane's role (the solve reads rows 3q .. 3q+2 of rk only)
        T rk2[12];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const T b0_ = rk0[i] + acc[i] - rho[i], b1_ = rk0[3 + i] + acc[3 + i] - rho[i], j0_ = rk0[6 + i] - rho[i], j1_ = rk0[9 + i] - rho[i];
            rk2[i] = q == 0 ? b0_ : T(0); rk2[3 + i] = q == 1 ? b1_ : T(0); rk2[6 + i] = q == 2 ? j0_ : T(0); rk2[9 + i] = q == 3 ? j1_ : T(0);
        }
        const T rs2[2] = {T(0), T(0)}, up2[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
        T r582[CPL][4];
#pragma unroll
        for (int c = 0; c < CPL; ++c) for (int i = 0; i < 4; ++i) r582[c][i] = T(0);
        (void)rs; (void)r58;
        StepT D2;
        T dva2[6];
        solve_rhs(rk2, R2, rs2, r582, up2, D2, dva2);
        for (int i = 0; i < 6; ++i) dva[i] += dva2[i];
#ifdef DJ_DEBUG
        if (trace && active && q == 0) {
            T m1 = 0, m2 = 0, mr = 0, mc = 0;
            for (int i = 0; i < 3; ++i) { m1 = tmax(m1, tmax(tabs(D.dv[i]), tabs(D.dw[i]))); m2 = tmax(m2, tmax(tabs(D2.dv[i]), tabs(D2.dw[i]))); }
            for (int i = 0; i < 12; ++i) mr = tmax(mr, tabs(rk2[i]));
            for (int c = 0; c < CPL; ++c) for (int i = 0; i < 4; ++i) mc = tmax(mc, tabs(R2.cc[c][i]));
            if (P.level == 1) std::printf("   node %d: m6 %.3e %.3e %.3e xk9-11 %.3e %.3e %.3e dls %.3e %.3e dlg %.3e %.3e ls %.3e %.3e lg %.3e %.3e Rlim %.3e %.3e\n", k, (double)m6[0], (double)m6[1], (double)m6[2], (double)xk[9], (double)xk[10], (double)xk[11],
                                         (double)D.dls[0], (double)D.dls[1], (double)D.dlg[0], (double)D.dlg[1], (double)L.ls[0], (double)L.ls[1], (double)L.lg[0], (double)L.lg[1], (double)R.lim[0], (double)R.lim[1]);
            if (k == 0) std::printf("   root: rk0 %.3e %.3e %.3e acc %.3e %.3e %.3e rho %.3e %.3e %.3e  dv %.3e %.3e %.3e\n", (double)rk0[0], (double)rk0[1], (double)rk0[2], (double)acc[0], (double)acc[1], (double)acc[2], (double)rho[0], (double)rho[1], (double)rho[2], (double)D.dv[0], (double)D.dv[1], (double)D.dv[2]);
            std::printf("   refine k=%d |dw| %.2e |corr| %.2e  |res own rows| %.2e |res cone| %.2e lim %.2e %.2e\n", k, (double)m1, (double)m2, (double)mr, (double)mc, (double)R2.lim[0], (double)R2.lim[1]);
        }
#endif
        for (int i = 0; i < 3; ++i) { D.dv[i] += D2.dv[i]; D.dw[i] += D2.dw[i]; }
        for (int i = 0; i < 6; ++i) D.dlam[i] += D2.dlam[i];
        for (int i = 0; i < 2; ++i) { D.dls[i] += D2.dls[i]; D.dlg[i] += D2.dlg[i]; }
#pragma unroll
        for (int c = 0; c < CPL; ++c) for (int i = 0; i < 4; ++i) { D.dcs[c][i] += D2.dcs[c][i]; D.dcg[c][i] += D2.dcg[c][i]; }
    }

    // cone_line_search!  src/solver/line_search.jl:36-96
    DJ_HD T cone_line_search(const StepT& D, T tort, T tsoc) {
        T a = T(1);
        if (active) {
#pragma unroll
            for (int li = 0; li < CPL; ++li) if (cidx(li) < P.ncontact) {
                const int c = cidx(li);
                a = tmin(a, ort_step(L.cs[c][0], D.dcs[li][0], tort));
                a = tmin(a, ort_step(L.cg[c][0], D.dcg[li][0], tort));
                if constexpr (kLinear) {                                   // line_search.jl:68-83: every pair on the positive orthant
                    for (int i = 1; i < NCV; ++i) { a = tmin(a, ort_step(L.cs[c][i], D.dcs[li][i], tort)); a = tmin(a, ort_step(L.cg[c][i], D.dcg[li][i], tort)); }
                    continue;
                }
                a = tmin(a, soc_step(&L.cs[c][1], &D.dcs[li][1], tsoc));
                a = tmin(a, soc_step(&L.cg[c][1], &D.dcg[li][1], tsoc));
            }
            if (lim_on()) for (int i = 0; i < 2; ++i) {
                a = tmin(a, ort_step(L.ls[i], D.dls[i], tort));
                a = tmin(a, ort_step(L.lg[i], D.dlg[i], tort));
            }
#if DJ_MLIM
            if constexpr (kMLim) { for (int m = 0; m < NLM; ++m) if (m < nlm()) for (int i = 0; i < 2; ++i) {
                a = tmin(a, ort_step(L.mls[m][i], D.dmls[m][i], tort));
                a = tmin(a, ort_step(L.mlg[m][i], D.dmlg[m][i], tort));
            } }
#endif
        }
        a = quad_minv(a);
        if constexpr (QUAD) { T v1[1] = {a}; env_reduce_quad<1>(v1, [](T a_, T b_) { return a_ < b_ ? a_ : b_; }); DJ_P2E(5); return v1[0]; }
        else return env_min(wv, a, envl);
    }

    // candidate_step!  src/solver/line_search.jl:141-163: candidate = base + f Δ (base = the current iterate,
    // live only during the line search).  Returns 1 if ω stays beyond the error threshold after clipping.
    DJ_HD void snapshot(SnapT& B) const {
        for (int i = 0; i < 3; ++i) { B.v[i] = L.v[i]; B.w[i] = L.w[i]; }
        for (int i = 0; i < 6; ++i) B.lam[i] = L.lam[i];
        for (int i = 0; i < 2; ++i) { B.ls[i] = L.ls[i]; B.lg[i] = L.lg[i]; }
#if DJ_MLIM
        for (int m = 0; m < NLM; ++m) for (int i = 0; i < 2; ++i) { B.mls[m][i] = L.mls[m][i]; B.mlg[m][i] = L.mlg[m][i]; }
#endif
#if DJ_CUT
        for (int c = 0; c < NCUT; ++c) for (int i = 0; i < 6; ++i) B.clam[c][i] = clam[c][i];
#endif
#pragma unroll
        for (int li = 0; li < CPL; ++li) { const int c = kSplitC ? (cidx(li) < MAXC ? cidx(li) : 0) : li; for (int i = 0; i < NCV; ++i) { B.cs[li][i] = L.cs[c][i]; B.cg[li][i] = L.cg[c][i]; } }
    }
    DJ_HD int candidate_step(const SnapT& B, const StepT& D, T f) {
        int bad = 0;
        for (int i = 0; i < 3; ++i) { L.v[i] = B.v[i] + f * D.dv[i]; L.w[i] = B.w[i] + f * D.dw[i]; }
        T wmax = T(3.9) * G.idt2;
        T wd = v3dot(L.w, L.w);
        if (wd > wmax) { T sc = wmax / wd; for (int i = 0; i < 3; ++i) L.w[i] *= sc; }
        if (v3dot(L.w, L.w) > T(3.91) * G.idt2) bad = 1;
        for (int i = 0; i < 6; ++i) L.lam[i] = B.lam[i] + f * D.dlam[i];
        for (int i = 0; i < 2; ++i) { L.ls[i] = B.ls[i] + f * D.dls[i]; L.lg[i] = B.lg[i] + f * D.dlg[i]; }
#if DJ_MLIM
        for (int m = 0; m < NLM; ++m) for (int i = 0; i < 2; ++i) { L.mls[m][i] = B.mls[m][i] + f * D.dmls[m][i]; L.mlg[m][i] = B.mlg[m][i] + f * D.dmlg[m][i]; }
#endif
#if DJ_CUT
        for (int c = 0; c < NCUT; ++c) for (int i = 0; i < 6; ++i) clam[c][i] = B.clam[c][i] + f * D.dclam[c][i];
#endif
#pragma unroll
        for (int li = 0; li < CPL; ++li) { const int c = cidx(li); if (!kSplitC || c < MAXC) for (int i = 0; i < NCV; ++i) { L.cs[c][i] = B.cs[li][i] + f * D.dcs[li][i]; L.cg[c][i] = B.cg[li][i] + f * D.dcg[li][i]; } }
        share_cone_state();
        return bad;
    }

    // ---------------------------------------------------------------- set-up of one step
    // set_maximal_state! + set_input! (src/mechanism/set.jl:10-53): loads z, applies u, builds dconst.
    DJ_HD void begin_step(const T* zb /*13 values of this body*/, const T* u /*this joint's inputs (<= 6) or null*/, const T* fe = nullptr /*[Fext(3) world; τext(3) body] or null*/) {
        const T dt = G.dt;
        T v15[3] = {0, 0, 0}, w15[3] = {0, 0, 0};
        if (active) {
            for (int i = 0; i < 3; ++i) { L.x2[i] = zb[i]; v15[i] = zb[3 + i]; w15[i] = zb[10 + i]; }
            for (int i = 0; i < 4; ++i) L.q2[i] = zb[6 + i];
        } else {
            for (int i = 0; i < 3; ++i) L.x2[i] = T(0);
            L.q2[0] = T(1); L.q2[1] = L.q2[2] = L.q2[3] = T(0);
        }
        T own7[7] = {L.x2[0], L.x2[1], L.x2[2], L.q2[0], L.q2[1], L.q2[2], L.q2[3]}, par7[7];
        if (lane_slots) {
            wv.sync();
            const Lane<T, MAXC>& Lp = parent_state();
            for (int i = 0; i < 3; ++i) par7[i] = Lp.x2[i];
            for (int i = 0; i < 4; ++i) par7[3 + i] = Lp.q2[i];
        } else shfl_vec<7>(wv, par7, own7, plane);
        if (has_parent) { for (int i = 0; i < 3; ++i) L.xa2[i] = par7[i]; for (int i = 0; i < 4; ++i) L.qa2[i] = par7[3 + i]; }
        else { L.xa2[0] = L.xa2[1] = L.xa2[2] = T(0); L.qa2[0] = T(1); L.qa2[1] = L.qa2[2] = L.qa2[3] = T(0); }
        joint_cfg(cfg, P, L.xa2, L.qa2, L.x2, L.q2);
        // warm start (set_velocity_solution!, bodies/set.jl:1-7), reset!/initialize! of the cone variables
        for (int i = 0; i < 3; ++i) { L.v[i] = v15[i]; L.w[i] = w15[i]; L.v15[i] = v15[i]; L.w15[i] = w15[i]; }
        {
            for (int i = 0; i < 6; ++i) L.lam[i] = T(0);
            for (int i = 0; i < 2; ++i) { L.ls[i] = T(1); L.lg[i] = T(1); }            // joints/constraints.jl:440-448
#if DJ_MLIM
            for (int m = 0; m < NLM; ++m) for (int i = 0; i < 2; ++i) { L.mls[m][i] = T(1); L.mlg[m][i] = T(1); }
#endif
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {                                                  // reset! to [1,1,0,0] then initialize! -> 1.5·[1,1,0,0]
                // (the generic initialize! of an ImpactContact discards its result, initialization.jl:1-5: it starts from reset!'s 1)
                if constexpr (kLinear) { for (int i = 0; i < NCV; ++i) L.cs[c][i] = L.cg[c][i] = T(1); continue; }   // LinearContact: reset!'s ones(6), generic initialize! (no effect)
                const T c0_ = G.contact_model == 1 ? T(1) : T(1.5);
                L.cs[c][0] = L.cs[c][1] = c0_; L.cs[c][2] = L.cs[c][3] = T(0);
                L.cg[c][0] = L.cg[c][1] = c0_; L.cg[c][2] = L.cg[c][3] = T(0);
            }
        }
        // velocity-independent part of d: D1x + D1q (constraint.jl:15-18 in closed form) − gravity − inputs − springs
        T J15[3], wxJ[3];
        m3vec(J15, P.J, w15);
        v3cross(wxJ, w15, J15);
        T c15 = tsqrt(T(4) / (dt * dt) - v3dot(w15, w15));
        for (int i = 0; i < 3; ++i) {
            L.dconst[i] = -P.m * v15[i] - dt * P.m * G.g[i];
            L.dconst[3 + i] = T(-0.5) * dt * (c15 * J15[i] - wxJ[i]);
        }
        // external force / torque of the body (state.Fext, state.τext): −½Δt in D1 and again in D2 (constraint.jl:15-18)
        if (fe != nullptr) for (int i = 0; i < 6; ++i) L.dconst[i] -= dt * fe[i];
        // control input: set_input! + input_impulse!  (joints/joint.jl:96-99, translational/input.jl:5-27, rotational/input.jl:5-17)
        T ina[6] = {0, 0, 0, 0, 0, 0};                          // what this joint's input applies to the parent body
        if (active && u != nullptr) {
            T it[3] = {0, 0, 0}, ir[3] = {0, 0, 0};
            for (int i = 0; i < 3; ++i) {
                if (i < P.nu_t) for (int kx = 0; kx < 3; ++kx) it[kx] += P.At[3 * i + kx] * u[i];
                if (i < P.nu_r) for (int kx = 0; kx < 3; ++kx) ir[kx] += P.Ar[3 * i + kx] * u[P.nu_t + i];
            }
            for (int kx = 0; kx < 3; ++kx) { it[kx] *= G.input_scaling; ir[kx] *= G.input_scaling; }
            T ia[6], ib[6];
            tra_impulse(ia, ib, cfg, P, it);                    // Ta·input, Tb·input; torques get an extra 1/2 (input.jl:21-23)
            for (int i = 0; i < 3; ++i) { ina[i] = ia[i]; ina[3 + i] = T(0.5) * ia[3 + i]; L.dconst[i] -= ib[i]; L.dconst[3 + i] -= T(0.5) * ib[3 + i]; }
            T ta[3], tb[3];
            m3vec(ta, cfg.Roff, ir);                            // parent: vector_rotate(−τ, qoff); child: vector_rotate(τ, qb⁻¹ qa qoff)
            m3vec(tb, cfg.Rba, ta);
            for (int i = 0; i < 3; ++i) { ina[3 + i] += -ta[i]; L.dconst[3 + i] -= tb[i]; }
        }
        // springs (velocity independent)
        T sa[6], sb[6];
        spring_impulses(sa, sb, P, cfg, dt);
#if DJ_TSD
        if (tsd) {                                             // translational spring: Δt·T f  (translational/springs.jl:21-31)
            T f[3], ia[6], ib[6];
            tra_spring_force(f, P, *tsd, cfg);
            for (int i = 0; i < 3; ++i) f[i] *= dt;
            tra_impulse(ia, ib, cfg, P, f);
            for (int i = 0; i < 6; ++i) { sa[i] += ia[i]; sb[i] += ib[i]; }
        }
#endif
        for (int i = 0; i < 6; ++i) L.dconst[i] -= sb[i];
        T up[6], acc[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 6; ++i) up[i] = has_parent ? -(ina[i] + sa[i]) : T(0);
        if constexpr (QUAD) { mail_post_node<6>(up); mail_add_children_node<6>(acc, active, G.maxch); }
        else gather_children<6>(wv, acc, up, P, base, G.maxch, active, stride, q);
        wv.sync();                                             // (the four lanes of a quad update the shared dconst with the same values)
        for (int i = 0; i < 6; ++i) L.dconst[i] += acc[i];
        mu = T(0);
    }

    // ---------------------------------------------------------------- mehrotra!  src/solver/mehrotra.jl:9-73
    // returns status; iters_out = number of Newton iterations.  On return F holds the factors of the
    // final linearization (mehrotra.jl:69 runs set_entries! last) and rb/rj/theta/cres the residual
    // pieces at the solution.
    // set_entries! followed at once by the supernode factorization, so that the un-factored blocks
    // (S, U, L, Dup: 324 scalars) are transient and only the factors persist through the solves.
    DJ_HD void linearize() {
        if constexpr (QUAD) {
            QuadBlocks<TL> K(F.Sq, F.Uq, F.Lq, q);
            DJ_PB();
            evaluate<true>(K);
            condense_limits(K);
            if constexpr (kRefine) { if (blk != nullptr && wv.any(refine)) store_blocks(K); }   // the rows before the contacts are folded in
            condense_contacts(K);
            DJ_P2E(19);
            DJ_PE(0); DJ_PB();
#if DJ_ROWS == 1
            if constexpr (kRowsOk) { if (rows_here) factorize_rows(K); else factorize_quad(K); } else factorize_quad(K);
#elif DJ_ROWS == 2
            if constexpr (kRowsOk) { if (rows_here && G.rows > 0) factorize_rows(K); else factorize_quad(K); }
            else factorize_quad(K);
#else
            factorize_quad(K);
#endif
            DJ_P2E(20);
            DJ_PE(1);
            return;
        } else {
        FullBlocks<T> K;
        evaluate<true>(K);
        condense(K);
#ifdef DJ_DEBUG
        if (dbg_on && dbg) {   // test hook: dump the first assembly of this lane (after condensation)
            T* o = dbg; int q = 0;
            for (int i = 0; i < 6; ++i) o[q++] = rb[i];
            for (int i = 0; i < 6; ++i) o[q++] = rj[i];
            o[q++] = theta;
            for (int i = 0; i < 144; ++i) o[q++] = K.S[i];
            for (int i = 0; i < 72; ++i) o[q++] = K.U[i];
            for (int i = 0; i < 72; ++i) o[q++] = K.L[i];
            for (int i = 0; i < 36; ++i) o[q++] = K.D[i];
        }
#endif
        factorize(K);
#if DJ_CUT
#if DJ_SS
        if constexpr (kCut) { if (ncut > 0) cut_contacts_M(); }
#endif
        if constexpr (kCut) { if (ncut > 0) cut_factor(); }
#endif
        }
    }

    // One Mehrotra direction (src/solver/mehrotra.jl:36-49 with the current factors): affine solve, centering!, correction!, corrected
    // solve; returns the cone step length alpha of the corrected direction D and the centering target mutarget.
    DJ_HD T newton_direction(StepT& D, T rvio, T bvio, T undercut, T& mutarget) {
        ConeRhs R;
        cone_rhs_from_state(R, mu);                             // pull_residual!: cone rows carry μ of the last set_entries!
        DJ_P2E(0);
        DJ_PB();
        solve(R, D);                                            // affine direction
        DJ_PE(4);
        T aaff = cone_line_search(D, T(0.95), T(0.95));
        // centering!  src/solver/centering.jl
        T p0 = T(0), p1 = T(0), p2 = T(0);
        if (active) {
#pragma unroll
            for (int li = 0; li < CPL; ++li) if (cidx(li) < P.ncontact) {
                const int c = cidx(li);
                // cone_degree: 2 for NonlinearContact (nonlinear.jl:101), N½ = 1 for ImpactContact (contact.jl:203), whose
                // pinned friction variables do not take part
                const int nv_ = kLinear ? NCV : (G.contact_model == 1 ? 1 : 4);
                for (int i = 0; i < NCV; ++i) if (i < nv_) { p0 += L.cs[c][i] * L.cg[c][i]; p1 += (L.cs[c][i] + aaff * D.dcs[li][i]) * (L.cg[c][i] + aaff * D.dcg[li][i]); }
                p2 += kLinear ? T(NCV) : (G.contact_model == 1 ? T(1) : T(2));      // (LinearContact: cone_degree = N½ = 6, contact.jl:203)
            }
        }
        if constexpr (kSplitC) { T pc_[3] = {p0, p1, p2}; quad_sum(pc_); p0 = pc_[0]; p1 = pc_[1]; p2 = pc_[2]; }   // (the limit terms below are the same on all four lanes)
        if (active) {
            if (lim_on()) for (int i = 0; i < 2; ++i) { p0 += L.ls[i] * L.lg[i]; p1 += (L.ls[i] + aaff * D.dls[i]) * (L.lg[i] + aaff * D.dlg[i]); p2 += T(1); }
#if DJ_MLIM
            if constexpr (kMLim) { for (int m = 0; m < NLM; ++m) if (m < nlm()) for (int i = 0; i < 2; ++i) {
                p0 += L.mls[m][i] * L.mlg[m][i]; p1 += (L.mls[m][i] + aaff * D.dmls[m][i]) * (L.mlg[m][i] + aaff * D.dmlg[m][i]); p2 += T(1); } }
#endif
        }
        if constexpr (QUAD) { T v3[3] = {p0, p1, p2}; env_reduce_quad<3>(v3, [](T a_, T b_) { return a_ + b_; }); p0 = v3[0]; p1 = v3[1]; p2 = v3[2]; }
        else { p0 = env_sum(wv, p0, envl); p1 = env_sum(wv, p1, envl); p2 = env_sum(wv, p2, envl); }
        T munew = G.btol / undercut;
        if (p2 > T(0)) {
            T nu = p0 / p2, nuaff = p1 / p2;
            T sc = nuaff / (nu + T(1e-20));
            sc = tmin(tmax(sc, T(0)), T(1));
            munew = tmax(sc * sc * sc * nu, G.btol / undercut);
        }
        mutarget = munew;
        // correction!: cached residual += [−Δs∘Δγ + μ]   src/solver/correction.jl
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            if constexpr (kLinear) { for (int i = 0; i < NCV; ++i) R.cc[c][i] += -D.dcs[c][i] * D.dcg[c][i] + mutarget; continue; }   // correction.jl:13-19
            R.cc[c][0] += -D.dcs[c][0] * D.dcg[c][0] + mutarget;
            R.cc[c][1] += -(D.dcs[c][1] * D.dcg[c][1] + D.dcs[c][2] * D.dcg[c][2] + D.dcs[c][3] * D.dcg[c][3]) + mutarget;
            R.cc[c][2] += -(D.dcs[c][1] * D.dcg[c][2] + D.dcg[c][1] * D.dcs[c][2]);
            R.cc[c][3] += -(D.dcs[c][1] * D.dcg[c][3] + D.dcg[c][1] * D.dcs[c][3]);
        }
        R.lim[0] += -D.dls[0] * D.dlg[0] + mutarget;
        R.lim[1] += -D.dls[1] * D.dlg[1] + mutarget;
#if DJ_MLIM
        for (int m = 0; m < NLM; ++m) for (int i = 0; i < 2; ++i) R.mlim[m][i] += -D.dmls[m][i] * D.dmlg[m][i] + mutarget;
#endif
        DJ_P2E(6);
        DJ_PB();
        solve(R, D);                                            // corrected direction
        DJ_PE(2);
        T mx = tmax(rvio, bvio);
        T tau = tmax(T(0.95), T(1) - mx * mx);
        return cone_line_search(D, tau, tmin(tau, T(0.95)));
    }

    // need_factors: somebody reads the factors of the final linearization (the refining IFT kernel, through KernelArgs::fac); otherwise
    // a step may skip the set_entries! + factorization after the iteration that converged.
    // Iteration cap (Globals::iter_cap > 0, only with need_factors = false): a solve that is unfinished after iter_cap iterations leaves
    // with DJ_STATUS_CONTINUE and its loop scalars in carry_out[] (KernelArgs::resume); RESUME = true is the other end -- the continuation kernel has restored the
    // iterate (and μ) and *cy, and the loop goes on at iteration cy->n + 1 with the linearization the capped kernel skipped.  The iterates
    // of the two halves are those of the uncapped loop bit for bit (same functions of the same values in the same order).
    template <bool RESUME = false>
    DJ_HD int mehrotra(int& iters_out, bool need_factors = true, const SolveCarry<T>* cy = nullptr, T* carry_out = nullptr) {
        int status = DJ_STATUS_FAILED, excessive = 0;
        T mutarget = T(0), undercut = G.undercut;
        int no_progress = 0;
        T rvio, bvio;
        bool done = false;               // per-environment flag (identical on all lanes of an environment)
        int iters = 0, n0 = 1;
        if constexpr (RESUME) {
            undercut = cy->undercut; no_progress = cy->no_progress; excessive = cy->excessive; rvio = cy->rvio; bvio = cy->bvio;
            iters = cy->n; n0 = cy->n + 1;
            done = !active;              // (lanes of environments that had finished are inactive here and never change anything)
            linearize();
        } else {
        mu = T(0);
        if constexpr (kTrack) refine = T(1) > G.refine_w;        // reset! / initialize! leave every cone at γ/s = 1
        linearize();
#ifdef DJ_DEBUG
        if (dbg_on) { iters_out = 0; return 0; }   // wave-uniform early exit of the test hook
#endif
        violations(rvio, bvio);
        }
        DJ_P2B();
        for (int n = n0; n <= G.max_iter; ++n) {
#ifdef DJ_DEBUG
            if (trace && wv.lane() == 0) std::printf("%3d  bvio %.3e  rvio %.3e  mu %.3e\n", n, (double)bvio, (double)rvio, (double)mu);
#endif
            if (!done && rvio < G.rtol && bvio < G.btol) { status = DJ_STATUS_SUCCESS; done = true; }
            if constexpr (kTrack && !kRefine) { if (!done && refine) { status = DJ_STATUS_DEFERRED; done = true; } }   // left to the refining kernel
            if (!wv.any(active && !done)) break;
            // Lanes of finished environments keep executing (wave-uniform control flow, all lanes must
            // take part in the shuffles) but never change their state: their step factor is 0.
            if (!done) iters = n;
            DJ_P2E(21);
            StepT D_local;
            StepT& D = kLsInLds ? *(StepT*)ls_lds : D_local;
            T alpha = newton_direction(D, rvio, bvio, undercut, mutarget);
            // line_search!  src/solver/line_search.jl:1-34 (halving; the last trial is taken if all are rejected)
            T rc = rvio, bc = bvio;
            {
                bool searching = !done;
                T f = done ? T(0) : alpha;
                SnapT base_local;
                SnapT& base_sol = kLsInLds ? *(SnapT*)(ls_lds + sizeof(StepT)) : base_local;
                snapshot(base_sol);
                DJ_P2E(7);
                DJ_PB();
                // one trial: candidate, residual, violations, accept / halve
                auto trial = [&](int ls) {
                    int bad = candidate_step(base_sol, D, f);       // finished searches recompute the same candidate
                    DJ_P2E(8);
                    { NullBlocks nk; evaluate<false>(nk); }
                    T r2, b2;
                    violations(r2, b2);
#ifdef DJ_DEBUG
                    if (trace && active && q == 0) {
                        T mb = 0, mj = 0, mc = 0;
                        for (int i = 0; i < 6; ++i) { mb = tmax(mb, tabs(rb[i])); mj = tmax(mj, tabs(rj[i])); }
                        for (int c = 0; c < CPL; ++c) for (int i = 0; i < 4; ++i) mc = tmax(mc, tabs(cres[c][i]));
                        if (tmax(mb, tmax(mj, mc)) > T(0.3) * r2) std::printf("   trial %d f %.3e node %d: body %.2e joint %.2e contact %.2e (rvio %.2e)\n", ls, (double)f, k, (double)mb, (double)mj, (double)mc, (double)r2);
                    }
#endif
                    int anybad;
                    if constexpr (QUAD) { T vb[1] = {((active && searching) ? bad : 0) ? T(1) : T(0)}; env_reduce_quad<1>(vb, [](T a_, T b_) { return a_ > b_ ? a_ : b_; }); anybad = vb[0] > T(0.5) ? 1 : 0; }
                    else anybad = env_or(wv, (active && searching) ? bad : 0, envl);
                    if (searching) {
                        excessive |= anybad;
                        rc = r2; bc = b2;
                        if (r2 > rvio && b2 > bvio) { if (ls + 1 < G.max_ls) f *= T(0.5); } else searching = false;
                    }
                };
                if constexpr (Wave::kReplicas > 1) {
                    // Continuation kernel: kReplicas wavefronts hold the same environment(s) with identical state, each in its own LDS block.
                    // Replica r evaluates trial ls0 + r (step factor alpha / 2^(ls0 + r): halving is exact, so this is the factor the
                    // sequential search would have reached), the verdicts are exchanged through LDS, and every replica replays the
                    // sequential accept / halve decisions over them -- the accepted trial is the one line_search! would have stopped
                    // at (line_search.jl:1-34), the later ones were evaluated for nothing.
                    constexpr int R = Wave::kReplicas;
                    const int rep = wv.replica();
                    const T f0 = f;
                    T fch = f0, wst = wstiff;
                    double* const xq = wv.replica_xchg();                       // [R][supernode slots][4]
                    const int nsn_ = wv.width() >> 2;
                    for (int ls0 = 0; ls0 < G.max_ls; ls0 += R) {
                        if (!wv.any(active && searching)) break;
                        T ft = f0;
                        for (int i = 0; i < ls0 + rep; ++i) ft *= T(0.5);
                        int bad = candidate_step(base_sol, D, ft);
                        { NullBlocks nk; evaluate<false>(nk); }
                        T r2, b2;
                        violations(r2, b2);
                        int anybad;
                        { T vb[1] = {((active && searching) ? bad : 0) ? T(1) : T(0)}; env_reduce_quad<1>(vb, [](T a_, T b_) { return a_ > b_ ? a_ : b_; }); anybad = vb[0] > T(0.5) ? 1 : 0; }
                        if (q == 0) { double* x_ = xq + (size_t)(rep * nsn_ + (wv.lane() >> 2)) * 4; x_[0] = (double)r2; x_[1] = (double)b2; x_[2] = (double)anybad; x_[3] = (double)wstiff; }
                        wv.replica_sync();
                        T fj = f0;
                        for (int i = 0; i < ls0; ++i) fj *= T(0.5);
                        for (int j = 0; j < R; ++j) {
                            if (ls0 + j < G.max_ls && searching) {
                                const double* y_ = xq + (size_t)(j * nsn_ + (wv.lane() >> 2)) * 4;
                                excessive |= (int)y_[2];
                                rc = T(y_[0]); bc = T(y_[1]); wst = T(y_[3]); fch = fj;
                                if (!(rc > rvio && bc > bvio)) searching = false;
                            }
                            fj *= T(0.5);
                        }
                        wv.replica_sync();                                      // (the slots are rewritten by the next pass)
                    }
                    candidate_step(base_sol, D, fch);                           // every replica moves to the accepted trial's candidate
                    wstiff = wst;
                } else {
                for (int ls = 0; ls < G.max_ls; ++ls) {
                    if (!wv.any(active && searching)) break;
                    trial(ls);
                }
                }
                DJ_PE(3);
                if (!done) {
                    bool made = (!(rc < G.rtol) && rc < T(0.8) * rvio) || (!(bc < G.btol) && bc < T(0.8) * bvio);
                    if (made) no_progress = no_progress > 0 ? no_progress - 1 : 0; else no_progress += 1;
                    rvio = rc; bvio = bc;
                    if (no_progress >= G.no_progress_max) undercut *= G.no_progress_undercut;
                    mu = mutarget;
                    if constexpr (kTrack) refine = refine || (wstiff > G.refine_w);    // (wstiff: from the accepted trial's violations)
                }
                // Forward-only launch, every environment of the workgroup converged with this step: the next iteration would only
                // flag success (mehrotra.jl:23-27) -- do that here and skip the linearization nobody will use.
                if (!need_factors && n < G.max_iter) {
                    const bool conv = done || (rvio < G.rtol && bvio < G.btol);
                    if (!wv.any(active && !conv)) {
                        if (!done) { status = DJ_STATUS_SUCCESS; done = true; }
                        // (replicas: rb / rj / cres are those of this replica's own trial, not of the accepted one -- evaluate there once more)
                        if constexpr (Wave::kReplicas > 1) { NullBlocks nk; evaluate<false>(nk); }
                        break;
                    }
                    if constexpr (!RESUME) {
                        // iteration cap: the unfinished environments of this workgroup go on in the continuation kernel (wave-uniform exit)
                        if (G.iter_cap > 0 && n >= G.iter_cap && carry_out != nullptr) {
                            if (!done) {
                                status = DJ_STATUS_CONTINUE; done = true;
                                if (active && q == 0 && k == 0) {
                                    carry_out[0] = undercut; carry_out[1] = rvio; carry_out[2] = bvio; carry_out[3] = T(n); carry_out[4] = T(no_progress); carry_out[5] = T(excessive);
                                }
                            }
                            break;
                        }
                    }
                }
                // set_entries! + factorization (cone rows now carry the new μ)
                DJ_P2E(14);
                linearize();
            }
        }
        if (excessive && status != DJ_STATUS_DEFERRED && status != DJ_STATUS_CONTINUE) status = DJ_STATUS_EXCESSIVE_W;
        iters_out = iters;
        return status;
    }

    // Which batches of the column sweeps reach this supernode with a non-zero right-hand side: the forward-substituted ỹ of a batch is
    // non-zero only on the supernodes that carry a right-hand side (the owner body of the columns; for configuration columns its children
    // too) and on their ancestors.  Everywhere else it is exactly zero and is neither parked nor fetched (more than half of the park's
    // traffic for a tree like the Ant's).  MODE 0: second-kind batches = control batches (six input columns each); 1: contacts.
    template <int MODE, class KA>
    DJ_HD void sweep_masks(const KA& A, SweepP& sp) const {
        // (the walk reads the environment's node constants where the workgroup keeps them -- its LDS node slots: from KernelArgs::nodes every step
        //  up the tree was a dependent GLOBAL load, ~40 of them per lane and launch: 25 k cycles of the IFT kernel's prologue)
        auto node_at = [&](int a) -> const NodeP<T>& {
            if constexpr (QUAD && Wave::kLockstep) return ((const NodeSlot<T>*)&P)[a - k].P; else return A.nodes[a];
        };
        unsigned long long sub = 0, ub = 0;
        for (int j = 0; j < G.Nb; ++j) {
            int a = j, guard = 0; bool in = false;
            while (a >= 0 && guard++ < 64) { if (a == k) { in = true; break; } a = node_at(a).parent; }
            if (!in) continue;
            sub |= 1ull << j;
            const NodeP<T>& Pj = node_at(j);
            if (MODE == 0) { const int n = Pj.nu_t + Pj.nu_r; for (int c = 0; c < n; ++c) ub |= 1ull << ((Pj.u_off + c) / 6); }
            else for (int c = 0; c < Pj.ncontact; ++c) ub |= 1ull << Pj.contact[c];
        }
        sp.sub_mask = active ? sub : 0ull; sp.ub_mask = active ? ub : 0ull;
        sp.sub_t = k < G.Nb ? sub : 0ull; sp.ub_t = k < G.Nb ? ub : 0ull;
    }

    // the final linearization once more at the restored solution, factored in LU form and staged (quad mapping, IFT kernels)
    DJ_HD void lu_prepare() {
        if constexpr (QUAD) {
        QuadBlocks<TL> K(F.Sq, F.Uq, F.Lq, q);
        evaluate<true>(K);
        condense_limits(K);
        condense_contacts(K);
        DJ_PE(0); DJ_PB();
#ifdef DJ_DEBUG
        if (std::getenv("DJ_DUMP_BLOCKS") && active && base == 0) {   // model input (tools/model): the condensed supernode rows of this lane
            static char buf_[64][4096]; char* b_ = buf_[wv.lane() & 63]; int n_ = std::snprintf(b_, 4096, "BLK %d %d %d", k, q, P.parent);
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 12; ++j) n_ += std::snprintf(b_ + n_, 4096 - n_, " %.17g", (double)K.S[i][j]);
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 6; ++j) n_ += std::snprintf(b_ + n_, 4096 - n_, " %.17g", (double)K.U[i][j]);
            for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) n_ += std::snprintf(b_ + n_, 4096 - n_, " %.17g", (double)K.L[i][j]);
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 6; ++j) n_ += std::snprintf(b_ + n_, 4096 - n_, " %.17g", (double)K.D[i][j]);
            std::snprintf(b_ + n_, 4096 - n_, "\n"); std::fputs(b_, stderr);
        }
#endif
        // (a root's joint hangs on the origin, whose "velocity" is no unknown: its U block was assembled like any other, but the
        //  down-sweep must not apply it -- store_lu writes T = 0 for the roots)
#if DJ_ROWS == 1
        if constexpr (kRowsLuOk) { if (rows_here) factorize_rows_lu(K); else factorize_quad_lu(K); } else factorize_quad_lu(K);
#elif DJ_ROWS == 2
        if constexpr (kRowsLuOk) { if (rows_here && G.rows > 0) factorize_rows_lu(K); else factorize_quad_lu(K); }
        else factorize_quad_lu(K);
#else
        factorize_quad_lu(K);
#endif
        store_lu();
        DJ_PE(1); DJ_PB();
        }
    }

    // ---------------------------------------------------------------- IFT gradients
    // get_maximal_gradients (src/gradients/state.jl:78-126): data_jacobian = solmat \ datamat for the
    // state columns [x2 v15 φ2 ω15] of every body and the control columns of every joint, then the
    // integrator chain to (x3, v25, φ3, ω25).  solmat is the final linearization (re-using the
    // supernode factors instead of the reference's dense `\`); datamat = −∂residual/∂θ is assembled
    // block-wise (src/gradients/data.jl) in closed form at the evaluation state selected by
    // G.grad_mode: DOJO_GRAD_REFERENCE = post-update_state! states (literal
    // reference behaviour), DOJO_GRAD_CONSISTENT = pre-update states.
    // Output layout: column-major per environment (Julia-native): dz[env][col][row], du[env][ucol][row].
    // PRECISE (quad mapping, DJ_REFINE): the column-by-column path below with every column solved through solve_rhs and
    // refined against the uncondensed system -- for the environments whose cones were stiff at the solution (`write_out`
    // selects them inside a workgroup); the pipelined sweeps serve everything else.
    template <bool PRECISE = false, class KA>
    DJ_HD void gradients(const KA& A, int env, bool write_out = true) {
        const T dt = G.dt;
        const int nx = 12 * G.Nb;
        DJ_PB();
        // ---- quad mapping: the final linearization once more at the restored solution, factored in LU form (factorize_quad_lu) ----
        if constexpr (QUAD && !PRECISE) lu_prepare();
        DJ_P2B();
        // ---- kinematics of the solution (chain) ----
        T own6[6] = {L.v[0], L.v[1], L.v[2], L.w[0], L.w[1], L.w[2]}, par6[6], va[3], wa[3];
        if (lane_slots) {
            wv.sync();
            const Lane<T, MAXC>& Lp = parent_state();
            for (int i = 0; i < 3; ++i) { par6[i] = Lp.v[i]; par6[3 + i] = Lp.w[i]; }
        } else shfl_vec<6>(wv, par6, own6, plane);
        for (int i = 0; i < 3; ++i) { va[i] = has_parent ? par6[i] : T(0); wa[i] = has_parent ? par6[3 + i] : T(0); }
        Kin<T> kb0, ka0;
        kin_of(kb0, L.x2, L.q2, L.v, L.w, dt);
        kin_of(ka0, L.xa2, L.qa2, va, wa, dt);
        // ---- evaluation state of the data blocks ----
        T x2e[3], q2e[4], xa2e[3], qa2e[4], w15e[3];
        if (G.grad_mode == 0) {                                   // reference: x2 <- x3, q2 <- q3, ω15 <- ω25
            for (int i = 0; i < 3; ++i) { x2e[i] = kb0.x3[i]; xa2e[i] = ka0.x3[i]; w15e[i] = L.w[i]; }
            for (int i = 0; i < 4; ++i) { q2e[i] = kb0.q3[i]; qa2e[i] = ka0.q3[i]; }
        } else {
            for (int i = 0; i < 3; ++i) { x2e[i] = L.x2[i]; xa2e[i] = L.xa2[i]; w15e[i] = L.w15[i]; }
            for (int i = 0; i < 4; ++i) { q2e[i] = L.q2[i]; qa2e[i] = L.qa2[i]; }
        }
        JointCfg<T> ce;
        joint_cfg(ce, P, xa2e, qa2e, x2e, q2e);
        Kin<T> kb, ka;
        kin_of(kb, x2e, q2e, L.v, L.w, dt);
        kin_of(ka, xa2e, qa2e, va, wa, dt);
        JointEval<T> E;
        { NullBlocks nk; joint_eval<2>(E, P, ce, has_parent, ka, kb, wa, L.w, L.lam, L.lg, dt, nk); }
        DJ_P2E(8);
#if DJ_TSD
        if (tlim) tra_limit_eval<true>(E, P, ce, ka, kb, L.lg, dt);
#endif
#if DJ_CUT
        // cut joints: their data blocks at the evaluation state, computed by the owner (body b's lane) like a supernode's own joint (below) with
        // body a in the parent's place, then handed to every lane: joint rows wrt the configuration of b / a (CJb, CJa), rows of body b wrt b / a
        // (CBbb, CBba), rows of body a wrt b / a (CBab, CBaa), control columns on b / a (CUB, CUA)
        T CJb[NCUT][6][6], CJa[NCUT][6][6], CBbb[NCUT][6][6], CBba[NCUT][6][6], CBab[NCUT][6][6], CBaa[NCUT][6][6], CUB[NCUT][6][6], CUA[NCUT][6][6], crc0[NCUT][6];
        for (int c = 0; c < NCUT; ++c) for (int i = 0; i < 6; ++i) { crc0[c][i] = T(0); for (int j = 0; j < 6; ++j) CJb[c][i][j] = CJa[c][i][j] = CBbb[c][i][j] = CBba[c][i][j] = CBab[c][i][j] = CBaa[c][i][j] = CUB[c][i][j] = CUA[c][i][j] = T(0); }
        if constexpr (kCut) { for (int c = 0; c < NCUT; ++c) if (c < ncut) {
            const NodeP<T>& Pc = cutp[c];
            const int lb = base + cut_b(c), la = base + cut_a(c);
            T own10[13] = {x2e[0], x2e[1], x2e[2], q2e[0], q2e[1], q2e[2], q2e[3], L.v[0], L.v[1], L.v[2], L.w[0], L.w[1], L.w[2]}, a13[13];
            shfl_vec<13>(wv, a13, own10, la);
            T blk[8][36];
            for (int b_ = 0; b_ < 8; ++b_) for (int i = 0; i < 36; ++i) blk[b_][i] = T(0);
            if (cut_owner(c)) {
                const T* xae = a13; const T* qae = a13 + 3; const T* vae = a13 + 7; const T* wae = a13 + 10;
                JointCfg<T> cc_; joint_cfg(cc_, Pc, xae, qae, x2e, q2e);
                Kin<T> kac; kin_of(kac, xae, qae, vae, wae, dt);
                JointEval<T> Ec; const T lg0[2] = {0, 0};
                { NullBlocks nk; joint_eval<2>(Ec, Pc, cc_, true, kac, kb, wae, L.w, clam[c], lg0, dt, nk); }
                for (int sl = 0; sl < 6; ++sl) for (int j = 0; j < 3; ++j) {
                    blk[0][6 * sl + j] = -Ec.GbX[3 * sl + j]; blk[1][6 * sl + j] = -Ec.GaX[3 * sl + j];
                    T pb_ = T(0), pa_ = T(0);
                    for (int m_ = 0; m_ < 3; ++m_) { pb_ += Ec.GbP[3 * sl + m_] * kb.Xi[3 * m_ + j]; pa_ += Ec.GaP[3 * sl + m_] * kac.Xi[3 * m_ + j]; }
                    blk[0][6 * sl + 3 + j] = -pb_; blk[1][6 * sl + 3 + j] = -pa_;
                }
                T pt[3] = {0, 0, 0}, pr[3] = {0, 0, 0};
                for (int i = 0; i < 3; ++i) {
                    if (i < Pc.nl_t) for (int q_ = 0; q_ < 3; ++q_) pt[q_] += Pc.Ct[3 * i + q_] * clam[c][i];
                    if (i < Pc.nl_r) for (int q_ = 0; q_ < 3; ++q_) pr[q_] += Pc.Cr[3 * i + q_] * clam[c][3 + i];
                }
                T Jaa[36], Jab[36], Jba[36], Jbb[36];
                joint_impulse_cfg_jac(Jaa, Jab, Jba, Jbb, Pc, cc_, pt, pr, wae, L.w, dt);
                for (int i = 0; i < 36; ++i) { blk[2][i] = Jbb[i]; blk[3][i] = Jba[i]; blk[4][i] = Jab[i]; blk[5][i] = Jaa[i]; }
                for (int i = 0; i < 3; ++i) {
                    if (i < Pc.nu_t) {
                        T ia[6], ib[6];
                        tra_impulse(ia, ib, cc_, Pc, &Pc.At[3 * i]);
                        for (int r = 0; r < 3; ++r) { blk[6][6 * r + i] = G.input_scaling * ib[r]; blk[6][6 * (3 + r) + i] = G.input_scaling * T(0.5) * ib[3 + r]; blk[7][6 * r + i] = G.input_scaling * ia[r]; blk[7][6 * (3 + r) + i] = G.input_scaling * T(0.5) * ia[3 + r]; }
                    }
                    if (i < Pc.nu_r) {
                        T ta[3], tb[3];
                        m3vec(ta, cc_.Roff, &Pc.Ar[3 * i]); m3vec(tb, cc_.Rba, ta);
                        const int col = Pc.nu_t + i;
                        for (int r = 0; r < 3; ++r) { blk[6][6 * (3 + r) + col] = G.input_scaling * tb[r]; blk[7][6 * (3 + r) + col] = -G.input_scaling * ta[r]; }
                    }
                }
            }
            for (int b_ = 0; b_ < 8; ++b_) {
                T got[36];
                shfl_vec<36>(wv, got, blk[b_], lb);
                T (*dst)[6] = b_ == 0 ? CJb[c] : b_ == 1 ? CJa[c] : b_ == 2 ? CBbb[c] : b_ == 3 ? CBba[c] : b_ == 4 ? CBab[c] : b_ == 5 ? CBaa[c] : b_ == 6 ? CUB[c] : CUA[c];
                for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) dst[i][j] = got[6 * i + j];
            }
        } }
#endif
#if DJ_MLIM
        // limits on several coordinates: raw ∂θ_m/∂(x3, φ3) at the evaluation state -> the slack rows' data columns (up = −∂θ/∂z2, lo = +∂θ/∂z2),
        // and σ_m = wκ/(1 + wκ) of every limited coordinate
        T msl_own[NLM][6], msl_par[NLM][6], mwk[NLM], mrs0[NLM];
        for (int m = 0; m < NLM; ++m) { mwk[m] = mrs0[m] = T(0); for (int j = 0; j < 6; ++j) msl_own[m][j] = msl_par[m][j] = T(0); }
        if constexpr (kMLim) { if (nlm() > 0) {
            T rx_a[NLM][6], rx_b[NLM][6], ia_[6] = {0, 0, 0, 0, 0, 0}, ib_[6] = {0, 0, 0, 0, 0, 0};
            mlim_eval<2>(ia_, ib_, ce, ka, kb, dt, rx_a, rx_b);
            for (int m = 0; m < NLM; ++m) if (m < nlm()) {
                for (int j = 0; j < 3; ++j) {
                    T pb_ = T(0), pa_ = T(0);
                    for (int m_ = 0; m_ < 3; ++m_) { pb_ += rx_b[m][3 + m_] * kb.Xi[3 * m_ + j]; pa_ += rx_a[m][3 + m_] * ka.Xi[3 * m_ + j]; }
                    msl_own[m][j] = -rx_b[m][j]; msl_par[m][j] = -rx_a[m][j];            // ∂x3/∂x2 = I
                    msl_own[m][3 + j] = -pb_; msl_par[m][3 + j] = -pa_;
                }
                const T w_ = (L.mlg[m][1] + T(REG)) / (L.mls[m][1] + T(REG)) + (L.mlg[m][0] + T(REG)) / (L.mls[m][0] + T(REG));
                mwk[m] = w_ / (T(1) + w_);
            }
        } }
#endif
        DJ_P2E(9);
        // ---- data blocks (datamat = −∂residual/∂θ) ----
        T OwnB[6][12], OwnJ[6][6], ParB[6][6], ParJ[6][6], UpOwn[6][6], UpPar[6][6], sl_own[6], sl_par[6], Cc[MAXC][4][6], UB[6][6], UA[6][6];
        for (int i = 0; i < 6; ++i) { for (int j = 0; j < 12; ++j) OwnB[i][j] = T(0); for (int j = 0; j < 6; ++j) { OwnJ[i][j] = ParB[i][j] = ParJ[i][j] = UpOwn[i][j] = UpPar[i][j] = UB[i][j] = UA[i][j] = T(0); } sl_own[i] = sl_par[i] = T(0); }
        // body rows <- own (v15, ω15): data.jl:16-55 in closed form
        {
            T J15[3], SwJ[9], SJw[9], Sw[9];
            m3vec(J15, P.J, w15e);
            T c15 = tsqrt(T(4) / (dt * dt) - v3dot(w15e, w15e));
            m3skew(Sw, w15e); m3skew(SJw, J15); m3mul(SwJ, Sw, P.J);
            for (int i = 0; i < 3; ++i) {
                OwnB[i][3 + i] = P.m;
                for (int j = 0; j < 3; ++j) OwnB[3 + i][9 + j] = T(0.5) * dt * (c15 * P.J[3 * i + j] - J15[i] * w15e[j] / c15 - SwJ[3 * i + j] + SJw[3 * i + j]);
            }
        }
        // joint rows <- configuration of child / parent: −∂g/∂z3 · ∂z3/∂z2   (data.jl:4-14)
        for (int sl = 0; sl < 6; ++sl) for (int j = 0; j < 3; ++j) {
            OwnJ[sl][j] = -E.GbX[3 * sl + j];
            ParJ[sl][j] = -E.GaX[3 * sl + j];
            T pb_ = T(0), pa_ = T(0);
            for (int m_ = 0; m_ < 3; ++m_) { pb_ += E.GbP[3 * sl + m_] * kb.Xi[3 * m_ + j]; pa_ += E.GaP[3 * sl + m_] * ka.Xi[3 * m_ + j]; }
            OwnJ[sl][3 + j] = -pb_; ParJ[sl][3 + j] = -pa_;
        }
        // limit slack rows: up = −∂θ/∂φ2, lo = +∂θ/∂φ2
        if (lim_on()) for (int j = 0; j < 3; ++j) {
            T pb_ = T(0), pa_ = T(0);
            for (int m_ = 0; m_ < 3; ++m_) { pb_ += E.thp_b[m_] * kb.Xi[3 * m_ + j]; pa_ += E.thp_a[m_] * ka.Xi[3 * m_ + j]; }
            sl_own[3 + j] = -pb_; sl_par[3 + j] = -pa_;
#if DJ_TSD
            if (tlim) { sl_own[j] = -E.thx_b[j]; sl_par[j] = -E.thx_a[j]; }      // translational θ also depends on x2 (∂x3/∂x2 = I)
#endif
        }
        // body rows <- configurations through the joint impulse map, springs and dampers (data.jl:57-124)
        {
            T pt[3] = {0, 0, 0}, pr[3] = {0, 0, 0};
            for (int i = 0; i < 3; ++i) {
                if (i < P.nl_t) for (int q_ = 0; q_ < 3; ++q_) pt[q_] += P.Ct[3 * i + q_] * L.lam[i];
                if (i < P.nl_r) for (int q_ = 0; q_ < 3; ++q_) pr[q_] += P.Cr[3 * i + q_] * L.lam[3 + i];
            }
            if (P.nlim_r > 0) { T kk = L.lg[1] - L.lg[0]; for (int q_ = 0; q_ < 3; ++q_) pr[q_] += P.Ar[q_] * kk; }
            if (tlim) { T kk = L.lg[1] - L.lg[0]; for (int q_ = 0; q_ < 3; ++q_) pt[q_] += P.At[q_] * kk; }
#if DJ_MLIM
            if constexpr (kMLim) { for (int m = 0; m < NLM; ++m) if (m < nlm()) {
                const T kk = L.mlg[m][1] - L.mlg[m][0];
                const bool tra_ = m < mlp->nt;
                const T* a_ = tra_ ? &P.At[3 * m] : &P.Ar[3 * (m - mlp->nt)];
                for (int q_ = 0; q_ < 3; ++q_) { if (tra_) pt[q_] += a_[q_] * kk; else pr[q_] += a_[q_] * kk; }
            } }
#endif
            T Jaa[36], Jab[36], Jba[36], Jbb[36];
#if DJ_TSD
            T Saa[36], Sab[36], Sba[36], Sbb[36], pf[3] = {0, 0, 0};
            if (tsd) {
                for (int i = 0; i < 36; ++i) Saa[i] = Sab[i] = Sba[i] = Sbb[i] = T(0);
                tra_sd_cfg_jac(Saa, Sab, Sba, Sbb, pf, P, *tsd, ce, va, wa, L.v, L.w, dt);
                for (int i = 0; i < 3; ++i) pt[i] += pf[i];
            }
#endif
            joint_impulse_cfg_jac(Jaa, Jab, Jba, Jbb, P, ce, pt, pr, wa, L.w, dt);
#if DJ_TSD
            if (tsd) for (int i = 0; i < 36; ++i) { Jaa[i] += Saa[i]; Jab[i] += Sab[i]; Jba[i] += Sba[i]; Jbb[i] += Sbb[i]; }
#endif
            for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) {
                int cj = j < 3 ? j : 3 + j;                       // x2 -> cols 0:3, φ2 -> cols 6:9 of the 12 own-data columns
                OwnB[i][cj] += Jbb[6 * i + j];
                ParB[i][j] = Jba[6 * i + j];
                UpOwn[i][j] = Jab[6 * i + j];
                UpPar[i][j] = Jaa[6 * i + j];
            }
        }
        DJ_P2E(10);
        // contacts: body rows <- own φ2 (data.jl:126-135), contact rows <- own configuration (data.jl:194-205)
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) Cc[c][i][j] = T(0);
            if (c < P.ncontact) {
                ContactEval<T> CE;
                contact_eval<true>(CE, CP[P.contact[c]], kb, L.v, L.w, L.cs[c], L.cg[c], dt);
                const ContactP<T>& K = CP[P.contact[c]];
                for (int j = 0; j < 3; ++j) {
                    Cc[c][0][j] = -K.n[j];
                    T a0 = T(0), a2 = T(0), a3 = T(0);
                    for (int m_ = 0; m_ < 3; ++m_) { a0 += CE.c1p[m_] * kb.Xi[3 * m_ + j]; a2 += CE.c34p[m_] * kb.Xi[3 * m_ + j]; a3 += CE.c34p[3 + m_] * kb.Xi[3 * m_ + j]; }
                    Cc[c][0][3 + j] = -a0; Cc[c][2][3 + j] = -a2; Cc[c][3][3 + j] = -a3;
                    for (int i = 0; i < 3; ++i) { T q_ = T(0); for (int m_ = 0; m_ < 3; ++m_) q_ += CE.Qraw[3 * i + m_] * kb.Xi[3 * m_ + j]; OwnB[3 + i][6 + j] += q_; }
                }
            }
        }
        DJ_P2E(11);
        // control columns: input_jacobian_control (translational/input.jl:33-44, rotational/input.jl:23-40)
        for (int i = 0; i < 3; ++i) {
            if (i < P.nu_t) {
                T ia[6], ib[6];
                tra_impulse(ia, ib, ce, P, &P.At[3 * i]);
                for (int r = 0; r < 3; ++r) { UB[r][i] = G.input_scaling * ib[r]; UB[3 + r][i] = G.input_scaling * T(0.5) * ib[3 + r]; UA[r][i] = G.input_scaling * ia[r]; UA[3 + r][i] = G.input_scaling * T(0.5) * ia[3 + r]; }
            }
            if (i < P.nu_r) {
                T ta[3], tb[3];
                m3vec(ta, ce.Roff, &P.Ar[3 * i]);
                m3vec(tb, ce.Rba, ta);
                // column P.nu_t + i, written with selects (a dynamic index would put UB / UA into scratch)
#pragma unroll
                for (int c_ = 0; c_ < 6; ++c_) if (c_ >= i) {
                    const bool hit = (c_ == P.nu_t + i);
                    for (int r = 0; r < 3; ++r) { UB[3 + r][c_] = hit ? G.input_scaling * tb[r] : UB[3 + r][c_]; UA[3 + r][c_] = hit ? -G.input_scaling * ta[r] : UA[3 + r][c_]; }
                }
            }
        }
        DJ_P2E(12);
        // ---- condensation maps for a right-hand side without cone terms (computed once) ----
        // contact rows r58 -> body rhs:  rk[0:6] += GK r58 ;  limit slack rows (rs, −rs) -> rk += t_b wk rs, up += t_a wk rs
        T GK[MAXC][6][4], wk = T(0);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                T rc0[4] = {0, 0, 0, 0}, e[4] = {0, 0, 0, 0};
                e[j] = T(1);
                CCoef Q;
                if (c < P.ncontact) contact_coef(Q, c, rc0, e); else { Q.k0[0] = Q.k0[1] = Q.k0[2] = T(0); }
#pragma unroll
                for (int i = 0; i < 6; ++i) GK[c][i][j] = (c < P.ncontact) ? ccold(c).G134[i] * Q.k0[0] + ccold(c).G134[6 + i] * Q.k0[1] + ccold(c).G134[12 + i] * Q.k0[2] : T(0);
            }
        }
        // σ = wκ/(1 + wκ): the slack rows of a joint limit enter the Δκ row (see evaluate) as σ·(∂ slack row / ∂ data)
        if (lim_on()) { const T w_ = (L.lg[1] + T(REG)) / (L.ls[1] + T(REG)) + (L.lg[0] + T(REG)) / (L.ls[0] + T(REG)); wk = w_ / (T(1) + w_); }
        // the body-row blocks of the IFT right-hand sides are kept in the ABI type (QuadRhs)
        typedef typename KA::io_type TIO; typedef TIO TB;
        if constexpr (QUAD && !PRECISE) {
            DJ_P2E(13);
            // ---- stash the right-hand sides once per supernode in LDS, cone condensation folded in ----
            // everything read from NodeP / Lane / Cold below this point is cached first: the right-hand sides overlay them
            SweepP sp;
            sp.level = P.level; sp.parent = P.parent; sp.pb = has_parent ? base + stride * P.parent : qb; sp.u_off = P.u_off;
            sp.myu = P.nu_t + P.nu_r; sp.nlim_r = tlim ? 2 : P.nlim_r /* 2: translational limit, Δκ row of role 2 */; sp.nchild = P.nchild; sp.ncontact = P.ncontact;
#pragma unroll
            for (int c_ = 0; c_ < 8; ++c_) sp.contact[c_] = P.contact[c_];
#pragma unroll
            for (int ci = 0; ci < MAXCH; ++ci) sp.child_lane0[ci] = base + stride * P.child[ci];
            sweep_masks<0>(A, sp);
            const bool lim = lim_on();
            const int ncon = P.ncontact;
            wv.sync();
            QuadRhs<TB>& R = *(QuadRhs<TB>*)gb_lds;
            if (q == 0) {
#pragma unroll
                for (int cI = 0; cI < 6; ++cI) { R.jd[QuadRhs<TB>::SLO + cI] = (double)(lim ? sl_own[cI] : T(0)); R.jd[QuadRhs<TB>::SLP + cI] = (double)(lim ? sl_par[cI] : T(0)); }
#pragma unroll
                for (int row = 0; row < 12; ++row) {
                    const int qh_ = (row / 3) & 1, i = row % 3, o_ = qh_ * 18 + i * 6;
#pragma unroll
                    for (int cI = 0; cI < 6; ++cI) {
                        const int cc = cI < 3 ? cI : cI + 3;        // configuration column (x2 | φ2) of the 12 own-data columns
                        if (row < 6) {
                            T v = OwnB[row][cc];
#pragma unroll
                            for (int cn = 0; cn < MAXC; ++cn) if (cn < ncon) for (int j = 0; j < 4; ++j) v += GK[cn][row][j] * Cc[cn][j][cI];
                            R.own_cfg[o_ + cI] = (double)v;
                            R.a[QuadRhs<TB>::ROWNV + o_ + cI] = TB(OwnB[row][cc + 3]);
                            R.a[QuadRhs<TB>::UOWN + o_ + cI] = TB(UpOwn[row][cI]);
                            R.a[QuadRhs<TB>::UPAR + o_ + cI] = TB(UpPar[row][cI]);
                            R.a[QuadRhs<TB>::UB + o_ + cI] = TB(UB[row][cI]);
                            R.a[QuadRhs<TB>::UA + o_ + cI] = TB(UA[row][cI]);
                            R.a[QuadRhs<TB>::RPARB + o_ + cI] = TB(ParB[row][cI]);
                        } else {
                            R.jd[QuadRhs<TB>::ROWNJ + o_ + cI] = (double)OwnJ[row - 6][cI];
                            R.jd[QuadRhs<TB>::RPARJ + o_ + cI] = (double)ParJ[row - 6][cI];
                        }
                    }
                }
            }
            wv.sync();
            DJ_PE(5); DJ_PB();
            gradient_columns_quad(A, env, R, wk, kb0, sp);
            DJ_PE(6);
            return;
        }
        // ---- lane = supernode mapping: data blocks in per-lane local memory ----
        GradBlocks<TB, MAXC> gb_local;
        {
            GradBlocks<TB, MAXC>& g_ = gb_local;
            for (int i = 0; i < 6; ++i) {
                for (int j = 0; j < 12; ++j) g_.OwnB[i][j] = TB(OwnB[i][j]);
                for (int j = 0; j < 6; ++j) { g_.OwnJ[i][j] = TB(OwnJ[i][j]); g_.ParB[i][j] = TB(ParB[i][j]); g_.ParJ[i][j] = TB(ParJ[i][j]); g_.UpOwn[i][j] = TB(UpOwn[i][j]); g_.UpPar[i][j] = TB(UpPar[i][j]); g_.UB[i][j] = TB(UB[i][j]); g_.UA[i][j] = TB(UA[i][j]); }
                g_.sl_own[i] = TB(sl_own[i]); g_.sl_par[i] = TB(sl_par[i]);
            }
#pragma unroll
            for (int c = 0; c < MAXC; ++c) for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) g_.Cc[c][i][j] = TB(Cc[c][i][j]);
        }
        const GradBlocks<TB, MAXC>& gb = gb_local;
        // ---- column loop (lane = supernode mapping) ----
        struct { T dv[3], dw[3]; } D;
        auto grad_solve = [&](T* rk, T rs0, const T (*r58)[4], T* upx) {
            if constexpr (PRECISE) {
                // the same column through the general solve (cone right-hand sides zero, slack rows (rs, −rs)) and refined
                ConeRhs R0;
#pragma unroll
                for (int c = 0; c < CPL; ++c) for (int i = 0; i < 4; ++i) R0.cc[c][i] = T(0);
                R0.lim[0] = R0.lim[1] = T(0);
                const T rs2[2] = {rs0, -rs0};
                StepT Dp; T dva[6];
                T r58l[CPL][4];                                   // this lane's contact slots of the contact-row right-hand sides (values first, then the select)
#pragma unroll
                for (int li = 0; li < CPL; ++li)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (kSplitC) { const T a0_ = r58[4 * li][i], a1_ = r58[4 * li + 1][i], a2_ = r58[4 * li + 2][i], a3_ = r58[4 * li + 3][i]; r58l[li][i] = q == 0 ? a0_ : q == 1 ? a1_ : q == 2 ? a2_ : a3_; }
                        else r58l[li][i] = r58[li][i];
                    }
                solve_rhs(rk, R0, rs2, r58l, upx, Dp, dva);
#pragma unroll 1
                for (int rstep = 0; rstep < DJ_REFINE_STEPS; ++rstep) refine_solution(rk, R0, rs2, r58l, upx, Dp, dva);
                for (int i = 0; i < 3; ++i) { D.dv[i] = Dp.dv[i]; D.dw[i] = Dp.dw[i]; }
                return;
            }
#pragma unroll
            for (int c = 0; c < MAXC; ++c) if (c < P.ncontact) {
#pragma unroll
                for (int i = 0; i < 6; ++i) rk[i] += GK[c][i][0] * r58[c][0] + GK[c][i][1] * r58[c][1] + GK[c][i][2] * r58[c][2] + GK[c][i][3] * r58[c][3];
            }
            if (lim_on()) { if (tlim) rk[8] += wk * rs0; else rk[11] += wk * rs0; }   // the Δκ row
#if DJ_MLIM
            if constexpr (kMLim) { for (int m = 0; m < NLM; ++m) if (m < nlm()) rk[mslot(m)] += mwk[m] * mrs0[m]; }
#endif
            T dk[12], dva[6];
#if DJ_CUT
            if constexpr (kCut) { T dlc_[NCUT][6]; if (ncut > 0) cut_solve(rk, upx, crc0, dk, dva, dlc_); else core_solve(rk, upx, dk, dva); } else
#endif
            core_solve(rk, upx, dk, dva);
            for (int i = 0; i < 3; ++i) { D.dv[i] = dk[i]; D.dw[i] = dk[3 + i]; }
        };
        typedef decltype(A.dz) OutPtr;
        for (int kk = 0; kk < G.Nb; ++kk) {
            const bool mine = active && (k == kk), child_of = active && has_parent && (P.parent == kk);
            for (int c = 0; c < 12; ++c) {
                const bool is_cfg = (c < 3) || (c >= 6 && c < 9);
                const int cc = c < 3 ? c : c - 3;                  // configuration column 0..5 (valid when is_cfg)
                T rk[12], rs[2] = {0, 0}, r58[MAXC][4], upx[6] = {0, 0, 0, 0, 0, 0};
                for (int i = 0; i < 12; ++i) rk[i] = T(0);
#pragma unroll
                for (int q_ = 0; q_ < MAXC; ++q_) for (int i = 0; i < 4; ++i) r58[q_][i] = T(0);
                if (mine) {
                    for (int i = 0; i < 6; ++i) rk[i] = T(gb.OwnB[i][c]);
                    if (is_cfg) {
                        for (int i = 0; i < 6; ++i) { rk[6 + i] = T(gb.OwnJ[i][cc]); upx[i] = T(gb.UpOwn[i][cc]); }
                        rs[0] = T(gb.sl_own[cc]); rs[1] = -rs[0];
#pragma unroll
                        for (int q_ = 0; q_ < MAXC; ++q_) for (int i = 0; i < 4; ++i) r58[q_][i] = T(gb.Cc[q_][i][cc]);
                    }
                } else if (child_of && is_cfg) {
                    for (int i = 0; i < 6; ++i) { rk[i] = T(gb.ParB[i][cc]); rk[6 + i] = T(gb.ParJ[i][cc]); upx[i] = T(gb.UpPar[i][cc]); }
                    rs[0] = T(gb.sl_par[cc]); rs[1] = -rs[0];
                }
#if DJ_MLIM
                for (int m = 0; m < NLM; ++m) mrs0[m] = (is_cfg && mine) ? msl_own[m][cc] : (is_cfg && child_of) ? msl_par[m][cc] : T(0);
#endif
#if DJ_CUT
                if constexpr (kCut) { for (int c_ = 0; c_ < NCUT; ++c_) {
                    const bool isb = c_ < ncut && is_cfg && kk == cut_b(c_), isa = c_ < ncut && is_cfg && kk == cut_a(c_);
                    for (int i = 0; i < 6; ++i) crc0[c_][i] = isb ? CJb[c_][i][cc] : isa ? CJa[c_][i][cc] : T(0);
                    if (c_ < ncut && active && k == cut_b(c_)) for (int i = 0; i < 6; ++i) rk[i] += isb ? CBbb[c_][i][cc] : isa ? CBba[c_][i][cc] : T(0);
                    if (c_ < ncut && active && k == cut_a(c_)) for (int i = 0; i < 6; ++i) rk[i] += isb ? CBab[c_][i][cc] : isa ? CBaa[c_][i][cc] : T(0);
                } }
#endif
                grad_solve(rk, rs[0], r58, upx);
                if (active && q == 0 && A.dz && write_out) {
                    OutPtr o = A.dz + ((size_t)env * nx + (size_t)(12 * kk + c)) * nx + 12 * k;
                    T pw[3];
                    m3vec(pw, kb0.Phi, D.dw);
                    for (int i = 0; i < 3; ++i) {
                        T x = dt * D.dv[i], ph = pw[i];
                        if (mine && c < 3 && c == i) x += T(1);
                        if (mine && c >= 6 && c < 9) ph += kb0.Xi[3 * i + (c - 6)];
                        o[i] = x; o[3 + i] = D.dv[i]; o[6 + i] = ph; o[9 + i] = D.dw[i];
                    }
                }
            }
        }
        // control columns, joint by joint (owner lane = child body of the joint)
        for (int kk = 0; kk < G.Nb; ++kk) {
            const NodeP<T>& Pk = A.nodes[kk];
            const int nuk = Pk.nu_t + Pk.nu_r;
            const bool mine = active && (k == kk);
            for (int c = 0; c < nuk; ++c) {
                T rk[12], rs[2] = {0, 0}, r58[MAXC][4], upx[6] = {0, 0, 0, 0, 0, 0};
                for (int i = 0; i < 12; ++i) rk[i] = T(0);
#pragma unroll
                for (int q_ = 0; q_ < MAXC; ++q_) for (int i = 0; i < 4; ++i) r58[q_][i] = T(0);
                if (mine) for (int i = 0; i < 6; ++i) { rk[i] = T(gb.UB[i][c]); upx[i] = T(gb.UA[i][c]); }
#if DJ_MLIM
                for (int m = 0; m < NLM; ++m) mrs0[m] = T(0);
#endif
#if DJ_CUT
                for (int c_ = 0; c_ < NCUT; ++c_) for (int i = 0; i < 6; ++i) crc0[c_][i] = T(0);
#endif
                grad_solve(rk, rs[0], r58, upx);
                if (active && q == 0 && A.du && write_out) {
                    OutPtr o = A.du + ((size_t)env * G.nu + (size_t)(Pk.u_off + c)) * nx + 12 * k;
                    T pw[3];
                    m3vec(pw, kb0.Phi, D.dw);
                    for (int i = 0; i < 3; ++i) { o[i] = dt * D.dv[i]; o[3 + i] = D.dv[i]; o[6 + i] = pw[i]; o[9 + i] = D.dw[i]; }
                }
            }
        }
#if DJ_CUT
        // ... and the control columns of the cut joints
        if constexpr (kCut) { for (int c_ = 0; c_ < NCUT; ++c_) if (c_ < ncut) {
            const NodeP<T>& Pc = cutp[c_];
            for (int c = 0; c < Pc.nu_t + Pc.nu_r; ++c) {
                T rk[12], rs[2] = {0, 0}, r58[MAXC][4], upx[6] = {0, 0, 0, 0, 0, 0};
                for (int i = 0; i < 12; ++i) rk[i] = T(0);
                for (int q_ = 0; q_ < MAXC; ++q_) for (int i = 0; i < 4; ++i) r58[q_][i] = T(0);
                if (active && k == cut_b(c_)) for (int i = 0; i < 6; ++i) rk[i] += CUB[c_][i][c];
                if (active && k == cut_a(c_)) for (int i = 0; i < 6; ++i) rk[i] += CUA[c_][i][c];
#if DJ_MLIM
                for (int m = 0; m < NLM; ++m) mrs0[m] = T(0);
#endif
                for (int e_ = 0; e_ < NCUT; ++e_) for (int i = 0; i < 6; ++i) crc0[e_][i] = T(0);
                grad_solve(rk, rs[0], r58, upx);
                if (active && q == 0 && A.du && write_out) {
                    OutPtr o = A.du + ((size_t)env * G.nu + (size_t)(Pc.u_off + c)) * nx + 12 * k;
                    T pw[3];
                    m3vec(pw, kb0.Phi, D.dw);
                    for (int i = 0; i < 3; ++i) { o[i] = dt * D.dv[i]; o[3 + i] = D.dv[i]; o[6 + i] = pw[i]; o[9 + i] = D.dw[i]; }
                }
            }
        } }
#endif
    }

    // ---------------------------------------------------------------- contact-data gradients (quad mapping)
    // get_contact_gradients (src/gradients/contact.jl:1-55): the columns of solmat \ datamat that belong to the contact data
    // θ = [friction_coefficient, contact_radius, contact_origin(3)], through the same factors, sweeps and integrator chain
    // as get_maximal_gradients.  Data blocks in closed form (sphere-halfspace, NonlinearContact):
    //   body rows       body_constraint_jacobian_contact_data     src/gradients/data.jl:152-171
    //   contact rows    contact_constraint_jacobian_contact_data  src/gradients/data.jl:173-192
    // Output: dc[env][5 Nc columns][12 Nb rows] (column-major per environment, like dz).
    template <class KA>
    DJ_HD void gradients_contact(const KA& A, int env) {
        static_assert(QUAD, "contact-data gradients are implemented for the quad mapping");
        const T dt = G.dt;
        // this kernel factors the final linearization itself: the state-column kernel's staged factors may belong to another launch,
        // or its workgroup may have left the environment to the refining kernel
        lu_prepare();
        Kin<T> kb0;
        kin_of(kb0, L.x2, L.q2, L.v, L.w, dt);
        T x2e[3], q2e[4];
        if (G.grad_mode == 0) { for (int i = 0; i < 3; ++i) x2e[i] = kb0.x3[i]; for (int i = 0; i < 4; ++i) q2e[i] = kb0.q3[i]; }
        else { for (int i = 0; i < 3; ++i) x2e[i] = L.x2[i]; for (int i = 0; i < 4; ++i) q2e[i] = L.q2[i]; }
        Kin<T> kb;
        kin_of(kb, x2e, q2e, L.v, L.w, dt);
        T rhs[MAXC][6][6];                                            // [contact][body row][column], cone condensation folded in
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
#pragma unroll
            for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) rhs[c][i][j] = T(0);
            if (c < P.ncontact) {
                const ContactP<T>& K = CP[P.contact[c]];
                const T* gam = L.cg[c];
                T F[3], Fb[3], nb[3], Rw[3], cr[3];
                for (int i = 0; i < 3; ++i) F[i] = K.n[i] * gam[0] + K.t[i] * gam[2] + K.t[3 + i] * gam[3];
                m3tvec(Fb, kb.R3, F);                                // contact force in the body frame
                m3tvec(nb, kb.R3, K.n);                              // normal in the body frame
                m3vec(Rw, kb.R3, L.w);                               // angular velocity in the world frame
                // body rows (torque rows 3:6): col radius = F_b x (R' n), cols origin = −[F_b]x
                T Bq[3][5];
                v3cross(cr, Fb, nb);
                for (int i = 0; i < 3; ++i) { Bq[i][0] = T(0); Bq[i][1] = cr[i]; }
                Bq[0][2] = T(0);   Bq[0][3] = Fb[2];  Bq[0][4] = -Fb[1];
                Bq[1][2] = -Fb[2]; Bq[1][3] = T(0);   Bq[1][4] = Fb[0];
                Bq[2][2] = Fb[1];  Bq[2][3] = -Fb[0]; Bq[2][4] = T(0);
                // contact constraint rows (4): col friction = (0, −γ1, 0, 0); col radius = (1·(n·n), 0, T (Rω x n));
                // cols origin = −(n R; 0; T [Rω]x R)
                T Cq[4][5], Swn[3], SR[9], Sw_[9];
                v3cross(Swn, Rw, K.n);
                m3skew(Sw_, Rw); m3mul(SR, Sw_, kb.R3);
                for (int j = 0; j < 5; ++j) { Cq[0][j] = Cq[1][j] = Cq[2][j] = Cq[3][j] = T(0); }
                Cq[1][0] = -gam[0];
                Cq[0][1] = v3dot(K.n, K.n); Cq[2][1] = v3dot(&K.t[0], Swn); Cq[3][1] = v3dot(&K.t[3], Swn);
                for (int j = 0; j < 3; ++j) {
                    Cq[0][2 + j] = -(K.n[0] * kb.R3[j] + K.n[1] * kb.R3[3 + j] + K.n[2] * kb.R3[6 + j]);
                    Cq[2][2 + j] = -(K.t[0] * SR[j] + K.t[1] * SR[3 + j] + K.t[2] * SR[6 + j]);
                    Cq[3][2 + j] = -(K.t[3] * SR[j] + K.t[4] * SR[3 + j] + K.t[5] * SR[6 + j]);
                }
                // contact rows -> body rhs through the cone condensation map GK (as for the state columns)
                T GKc[6][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    T rc0[4] = {0, 0, 0, 0}, e[4] = {0, 0, 0, 0};
                    e[j] = T(1);
                    CCoef Q;
                    contact_coef(Q, c, rc0, e);
#pragma unroll
                    for (int i = 0; i < 6; ++i) GKc[i][j] = ccold(c).G134[i] * Q.k0[0] + ccold(c).G134[6 + i] * Q.k0[1] + ccold(c).G134[12 + i] * Q.k0[2];
                }
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        T v = i >= 3 ? Bq[i - 3][j] : T(0);
                        for (int a = 0; a < 4; ++a) v += GKc[i][a] * Cq[a][j];
                        rhs[c][i][j] = v;
                    }
            }
        }
        // cache what the sweeps need from NodeP, then overlay the right-hand sides (one block per supernode) on the dead data
        SweepP sp;
        sp.level = P.level; sp.parent = P.parent; sp.pb = has_parent ? base + stride * P.parent : qb; sp.u_off = P.u_off;
        sp.myu = P.nu_t + P.nu_r; sp.nlim_r = tlim ? 2 : P.nlim_r /* 2: translational limit, Δκ row of role 2 */; sp.nchild = P.nchild; sp.ncontact = P.ncontact;
#pragma unroll
        for (int c_ = 0; c_ < 8; ++c_) sp.contact[c_] = P.contact[c_];
#pragma unroll
        for (int ci = 0; ci < MAXCH; ++ci) sp.child_lane0[ci] = base + stride * P.child[ci];
        sweep_masks<1>(A, sp);
        wv.sync();
        ConRhs<MAXC>& R = *(ConRhs<MAXC>*)gb_lds;
        if (q == 0) {
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) R.a[c * 36 + (i / 3) * 18 + (i % 3) * 6 + j] = (double)rhs[c][i][j];
        }
        wv.sync();
        gradient_columns_quad<1>(A, env, R, T(0), kb0, sp);
    }

    // update_state!  src/bodies/set.jl:22-36: -> (x3, v25, q3, ω25) = the next maximal state of this body
    DJ_HD void next_state(T* zb) const {
        Kin<T> kb;
        kin_of(kb, L.x2, L.q2, L.v, L.w, G.dt);
        for (int i = 0; i < 3; ++i) { zb[i] = kb.x3[i]; zb[3 + i] = L.v[i]; zb[10 + i] = L.w[i]; }
        for (int i = 0; i < 4; ++i) zb[6 + i] = kb.q3[i];
    }
};

// save_to_storage! (src/simulation/storage.jl:50-67) for one body at the solved step, before update_state!:
// row = [x2(3) q2(4) v15(3) ω15(3) px(3) pq(3) vl(3) ωl(3)].  momentum(mechanism, body) (src/mechanics/momentum.jl:17-41)
// is D2 − ½(inputs + joint impulses); with the body residual d = D1 + D2 − inputs − joint impulses − contact impulses of
// the solved step at hand (the step kernel's `res` output),  p = ½(D2 − D1 + contact impulses + d)  needs no joint at
// all: gravity and the external force cancel in
//   D2 − D1 = [m(v25 + v15); ½Δt(c25 Jω25 + ω25×Jω25) + ½Δt(c15 Jω15 − ω15×Jω15)].
// zb = the body's 13 state values the step was solved at, v/w = its solution (v25, ω25), csg = [s(4); γ(4)] per contact
// of the environment, rb = the body's six residual rows.  Shared by the HIP storage kernel and the SIMT emulator.
// The impulse one contact applies to its body (and, for a body-body contact, to the contact's parent body): impulse_map γ of
// src/contacts/contact.jl:79-100 for either collision and any of the three contact models, from the exported cone variables
// (gam = γ: [γ ψ β1 β2] NonlinearContact / ImpactContact, [γ ψ β1..β4] LinearContact, linear.jl:33-38).  Runtime switches: the Storage
// kernel is compiled once for all mechanisms.  ka = the parent body's kinematics (body-body contacts only).
template <class T>
DJ_HD void contact_impulses(T* imp, T* imp_par, const ContactP<T>& K, int model, const Kin<T>& kb, const Kin<T>* ka, const T* gam) {
    T n[3], t1[3], t2[3], l[3], lp[3] = {0, 0, 0};
    if (K.kind == 1) {
        T dx[3], wax[3] = {T(1), T(0), T(0)}, Rop[3], Roc[3];
        m3vec(Rop, ka->R3, K.o); m3vec(Roc, kb.R3, K.o2);
        for (int i = 0; i < 3; ++i) dx[i] = (ka->x3[i] + Rop[i]) - (kb.x3[i] + Roc[i]);
        const T id = trcp(tsqrt(v3dot(dx, dx)));
        for (int i = 0; i < 3; ++i) n[i] = dx[i] * id;
        v3cross(t1, wax, n);
        if (!(tsqrt(v3dot(t1, t1)) > T(1e-6))) { wax[0] = T(0); wax[1] = T(1); v3cross(t1, wax, n); }
        v3cross(t2, t1, n);
        for (int i = 0; i < 3; ++i) { l[i] = Roc[i] + K.r2 * n[i]; lp[i] = Rop[i] - K.r * n[i]; }
    } else {
        T Ro[3];
        m3vec(Ro, kb.R3, K.o);
        for (int i = 0; i < 3; ++i) { n[i] = K.n[i]; t1[i] = K.t[i]; t2[i] = K.t[3 + i]; l[i] = Ro[i] - K.off[i] - K.n[i] * K.r; }
    }
    const T g1 = model == 1 ? T(0) : model == 2 ? gam[4] - gam[5] : gam[2], g2 = model == 1 ? T(0) : model == 2 ? gam[2] - gam[3] : gam[3];
    T F[3], lxF[3];
    const T sg = K.kind == 1 ? T(-1) : T(1);                      // (the owner of a body-body contact is its CHILD: −X γ)
    for (int i = 0; i < 3; ++i) F[i] = sg * (n[i] * gam[0] + t1[i] * g1 + t2[i] * g2);
    v3cross(lxF, l, F);
    m3tvec(imp + 3, kb.R3, lxF);
    for (int i = 0; i < 3; ++i) imp[i] = F[i];
    if (K.kind == 1 && imp_par != nullptr) {
        T Fp[3] = {-F[0], -F[1], -F[2]};
        v3cross(lxF, lp, Fp);
        m3tvec(imp_par + 3, ka->R3, lxF);
        for (int i = 0; i < 3; ++i) imp_par[i] = Fp[i];
    }
}

// Body-body contacts (ContactP::kind 1) need the other body: `other(body index, zb[13], v[3], w[3])` loads its state and solution; `nodes`
// non-null says the mechanism may have such contacts, `self` is this body's index and `Nc` the number of contacts of the mechanism (the
// contacts this body is the PARENT of are found by their ContactP::pbody -- a tree neighbour or any other body).  model / cper:
// Globals::contact_model and the scalars per contact in csg (8, LinearContact 12).
struct NoOtherBody { template <class T> DJ_HD void operator()(int, T*, T*, T*) const {} };
template <class T, class TC, class OTHER = NoOtherBody>
DJ_HD void storage_row(T* row, const NodeP<T>& P, const ContactP<T>* CP, T dt, const T* zb, const T* v, const T* w, const TC* csg, const T* rb, const T* fe = nullptr,
                       const NodeP<T>* nodes = nullptr, OTHER other = OTHER(), int model = 0, int cper = 8, int self = -1, int Nc = 0) {
    const T x2[3] = {zb[0], zb[1], zb[2]}, v15[3] = {zb[3], zb[4], zb[5]}, q2[4] = {zb[6], zb[7], zb[8], zb[9]}, w15[3] = {zb[10], zb[11], zb[12]};
    Kin<T> kb;
    kin_of(kb, x2, q2, v, w, dt);
    T p[6], J25[3], J15[3], x25[3], x15[3];
    m3vec(J25, P.J, w); v3cross(x25, w, J25);
    m3vec(J15, P.J, w15); v3cross(x15, w15, J15);
    const T c15 = tsqrt(T(4) / (dt * dt) - v3dot(w15, w15));
    for (int i = 0; i < 3; ++i) {
        p[i] = P.m * (v[i] + v15[i]) + rb[i];
        p[3 + i] = T(0.5) * dt * (kb.c * J25[i] + x25[i] + c15 * J15[i] - x15[i]) + rb[3 + i];
    }
    const int nh = cper / 2;
    for (int c = 0; c < P.ncontact; ++c) {
        const int id = P.contact[c];
        T cg[6] = {0, 0, 0, 0, 0, 0}, imp[6];
        for (int i = 0; i < nh; ++i) cg[i] = T(csg[cper * id + nh + i]);
        if (CP[id].kind == 1) {                                  // this body is the child of a body-body contact: the contact's parent body at the same step
            if (nodes == nullptr) continue;
            T zo[13], vo[3], wo[3]; other(CP[id].pbody, zo, vo, wo);
            Kin<T> ka; kin_of(ka, zo, zo + 6, vo, wo, dt);
            contact_impulses<T>(imp, nullptr, CP[id], model, kb, &ka, cg);
        } else contact_impulses<T>(imp, nullptr, CP[id], model, kb, nullptr, cg);
        for (int i = 0; i < 6; ++i) p[i] += imp[i];
    }
    if (nodes != nullptr) for (int id = 0; id < Nc; ++id) {            // ... and the parent of others' body-body contacts
        if (CP[id].kind != 1 || CP[id].pbody != self) continue;
        T cg[6] = {0, 0, 0, 0, 0, 0}, zo[13], vo[3], wo[3], imp[6], impp[6];
        for (int i = 0; i < nh; ++i) cg[i] = T(csg[cper * id + nh + i]);
        other(CP[id].cbody, zo, vo, wo);
        Kin<T> kc; kin_of(kc, zo, zo + 6, vo, wo, dt);
        contact_impulses<T>(imp, impp, CP[id], model, kc, &kb, cg);
        for (int i = 0; i < 6; ++i) p[i] += impp[i];
    }
    for (int i = 0; i < 6; ++i) p[i] *= T(0.5);
    // simulate! clears the external force before it records (simulate.jl:29-31): momentum's D2 is evaluated without it
    if (fe != nullptr) for (int i = 0; i < 6; ++i) p[i] += T(0.5) * dt * fe[i];
    T R2[9], pw[3];
    qrot(R2, q2);
    m3vec(pw, R2, p + 3);                                      // vector_rotate(p_angular_body, q2)
    // ω2 = inertia \ (angular momentum in the body frame): adjugate of the 3x3
    const T* Jm = P.J;
    const T a0 = Jm[4] * Jm[8] - Jm[5] * Jm[7], a1 = Jm[2] * Jm[7] - Jm[1] * Jm[8], a2 = Jm[1] * Jm[5] - Jm[2] * Jm[4];
    const T b0 = Jm[5] * Jm[6] - Jm[3] * Jm[8], b1 = Jm[0] * Jm[8] - Jm[2] * Jm[6], b2 = Jm[2] * Jm[3] - Jm[0] * Jm[5];
    const T c0 = Jm[3] * Jm[7] - Jm[4] * Jm[6], c1 = Jm[1] * Jm[6] - Jm[0] * Jm[7], c2 = Jm[0] * Jm[4] - Jm[1] * Jm[3];
    const T idet = T(1) / (Jm[0] * a0 + Jm[1] * b0 + Jm[2] * c0);
    const T wl[3] = {(a0 * p[3] + a1 * p[4] + a2 * p[5]) * idet, (b0 * p[3] + b1 * p[4] + b2 * p[5]) * idet, (c0 * p[3] + c1 * p[4] + c2 * p[5]) * idet};
    for (int i = 0; i < 3; ++i) {
        row[i] = x2[i]; row[7 + i] = v15[i]; row[10 + i] = w15[i];
        row[13 + i] = p[i]; row[16 + i] = pw[i]; row[19 + i] = p[i] / P.m; row[22 + i] = wl[i];
    }
    for (int i = 0; i < 4; ++i) row[3 + i] = q2[i];
}

// ================================================================================================
// Kernel-level entry: one call per lane.  Shared by the HIP kernel (dojo_hip.hip) and the
// thread-based SIMT emulator (tests/emu).
// ================================================================================================
// TIO = scalar type of the ABI buffers, T = state/residual precision (tables are stored in T)
template <class TIO, class T>
struct KernelArgs {
    typedef TIO io_type;
    Globals<T> G;
    const NodeP<T>* nodes;
    const ContactP<T>* contacts;
    int B;                       // number of environments
    const TIO* z;                  // [B,13Nb]
    const TIO* u;                  // [B,nu] or null
    const TIO* fext;               // [B,6Nb] or null          external force (world) and torque (body frame) per body
    TIO* z_next;                   // [B,13Nb]
    int* status;                 // [B] or null
    int* iters;                  // [B] or null
    TIO* vel;                      // [B,6Nb] or null          (v25, ω25)
    TIO* joint_imp;                // [B,n_joint_imp] or null  (get_solution order)
    TIO* contact_sg;               // [B,8Nc] or null          ([s; γ] per contact)
    TIO* dz;                       // [B][12Nb cols][12Nb rows] column-major per env, or null
    TIO* du;                       // [B][nu cols][12Nb rows] column-major per env, or null
    TIO* dc;                       // [B][5Nc cols][12Nb rows] contact-data Jacobian (get_contact_gradients), or null
    TIO* res;                      // [B,6Nb] or null          body residual rows at the solution (for the Storage kernel)
    T* sol;                        // [B][S][sol_record<MAXC>] converged solution in state precision: step kernel -> IFT kernel (or null)
    T* fac;                        // [waves][72][64] quad mapping: the final supernode factors of every lane (explicit inverses; or null: nobody reads them)
    T* lu = nullptr;               // [workgroups][dj::LU_PER_LANE][lanes] quad mapping: the IFT kernel's own LU-form factors between its phases (or null)
    T* blk = nullptr;              // [workgroups][90][lanes] quad mapping: un-factored supernode rows of refining environments (DJ_REFINE; or null)
    long long ypark_stride = 0;    // elements per workgroup of ypark
    T* ypark = nullptr;            // [workgroups][batches][6][3][lanes / 2] quad mapping, ABI type narrower than the arithmetic: the IFT's forward-substituted
                                   // right-hand sides in full precision between the two sweeps (or null: they wait in the output buffer, in the ABI type)
    T* msg = nullptr;              // [B][level-1 supernodes][batches][2 roles][18] quad mapping: what the children of a root post to its body rows during the
    long long msg_stride = 0;      // up-sweep (read back by the root phase, gradient_columns_quad); msg_stride = elements per environment
    int* flag = nullptr;           // [B] 1: the plain step kernel deferred this environment to the refining kernels (DJ_REFINE; or null)
    T* mu_out = nullptr;           // [B] or null: mechanism.μ when mehrotra! returned (src/solver/mehrotra.jl:45), fp64
    T* diag_out = nullptr;         // [B][2] or null: diagnostics of the final linearization: max γ/s of the cones, largest Gauss-Jordan multiplier
    const TraSD<T>* tsd = nullptr; // [Nb + 1] translational springs / dampers per supernode, or null (read by the DJ_TSD builds only)
    const MLimP<T>* mlim = nullptr;// [Nb + 1] limits on several coordinates per supernode, or null (read by the DJ_MLIM builds only)
    const NodeP<T>* cuts = nullptr;// [ncut] loop-closing joints (read by the DJ_CUT builds only): the joint fields of NodeP, parent = body a, child[0] = body b
    int ncut = 0;
    T* cutws = nullptr;            // [B][CUTWS] workspace of the cut elements (M_c, H, the factored small system), or null
    // iteration cap + continuation (Globals::iter_cap > 0; all three set or all null):
    T* resume = nullptr;           // [B][CARRY_PER_ENV] solver scalars of the environments the step kernel left unfinished (DJ_STATUS_CONTINUE);
                                   // entry CARRY_MARK: 1 for the environments of a workgroup on the continuation list, 0 for the others (written by
                                   // the step kernel for every environment, never by the continuation: the IFT kernel of the others runs next to it)
    int* cont_list = nullptr;      // [workgroups of this launch] workgroup indices with an unfinished environment, in the order they finished
    int* cont_count = nullptr;     // [1] entries of cont_list (zeroed before the step kernel, atomically advanced by it)
    int wave_base = 0;             // index of this launch's first workgroup in the batch: cont_list holds batch-level workgroup indices (the
                                   // continuation kernels run once over the whole batch, behind the step kernels of all environment groups)
    const int* dispatch = nullptr; // [workgroups of this launch] the step kernel's hardware workgroup -> workgroup of this launch it works for (a permutation:
                                   // the longest solves of the previous step first, dojo_set_dispatch_order), or null: in order
#ifdef DJ_DEBUG
    T* dbg = nullptr;            // [B][Nb][512] test hook
#endif
};

// LDS layout of one workgroup (= one wavefront, or NW wavefronts for mechanisms of 17..32 bodies) in the quad mapping.
// LOCKSTEP (the GPU): everything the four lanes of a supernode hold identically exists once per supernode in LDS; the
// four lanes read it with broadcast ds_reads and write identical values in the same instruction.  The SIMT emulator's
// threads are not in lock step, so there NodeP / Lane stay per-lane and Cold is per-lane in "LDS".
//   phase A (Newton loop; data blocks of the IFT):
//     [NodeP x NSN][Lane x NSN][Cold x NSN (or per lane)][ContactCold pool][mailbox (step kernel)]
//   IFT column sweeps (overlay from offset 0: NodeP / Lane / Cold are dead by then, the few NodeP fields the sweeps
//   need are cached in registers):
//     [QuadRhs x NSN][mailbox]
//   [reduction scratch, 64 B] at the end
template <class T, int MAXC>
struct LaneSlot { Lane<T, MAXC> L; T pad_[(sizeof(Lane<T, MAXC>) / sizeof(T)) % 2 == 0 ? 1 : 2]; };   // odd stride in 8-byte words
constexpr int lds_imax(int a, int b) { return a > b ? a : b; }
// GRAD: 0 = step kernel, 1 = IFT kernel (state + control columns), 2 = IFT kernel for the contact-data columns
template <class TIO, class T, int MAXC, int GRAD, bool QUAD, bool LOCKSTEP, int NW = 1>
struct StepLds {
    static constexpr int NSN = 16 * NW;                                                     // supernode slots of the workgroup
    static constexpr bool share = QUAD && LOCKSTEP;
    static constexpr int node_bytes = share ? (int)sizeof(NodeSlot<T>) * NSN : 0;
    static constexpr int lane_off = node_bytes;
    static constexpr int lane_bytes = node_bytes + (share ? (int)sizeof(LaneSlot<T, MAXC>) * NSN : 0);   // end of the Lane block
    static constexpr int cold_n = LOCKSTEP ? NSN : 64 * NW;
    static constexpr bool cold_in_lds = QUAD;
    static constexpr int cold_off = lane_bytes;
    static constexpr int cold_bytes = QUAD ? (int)sizeof(Cold<T, MAXC>) * cold_n : 0;
    // contact rows: one slot per (supernode, contact) for a single-wave workgroup, one slot per contact of the
    // environment (at most 16) for NW > 1 -- there the per-supernode array would not fit
    static constexpr bool pool_by_id = NW > 1;
    static constexpr int pool_n = !QUAD ? 0 : (pool_by_id ? 16 : cold_n * MAXC);
    static constexpr int pool_off = cold_off + cold_bytes;
    static constexpr int pool_bytes = (int)sizeof(ContactCold<T>) * pool_n;
    // Newton step and the iterate at the start of the line search, once per supernode (step kernel, when there is room:
    // frees ~100 registers during the residual evaluations of the line search)
    static constexpr int ls_slot = (int)((sizeof(Step<T, MAXC>) + sizeof(SolSnap<T, MAXC>)) / 8 + (((sizeof(Step<T, MAXC>) + sizeof(SolSnap<T, MAXC>)) / 8) % 2 == 0 ? 1 : 0)) * 8;
    static constexpr bool ls_in_lds = share && !GRAD && NW == 1 && MAXC == 1;                 // (must match LaneProgram::kLsInLds)
    static constexpr int ls_off = pool_off + pool_bytes;
    static constexpr int a_end = ls_off + (ls_in_lds ? ls_slot * NSN : 0);
    static constexpr int rhs_bytes = !QUAD ? 0 : GRAD == 1 ? (int)sizeof(QuadRhs<TIO>) * NSN : GRAD == 2 ? (int)sizeof(ConRhs<MAXC>) * NSN : 0;
    static constexpr int mail_need = QUAD ? 2 * NSN * 20 * 8 : 0;
    static constexpr int rhs_off = 0;
    // (contact-data kernel: the mailbox keeps its own room behind the phase-A data; packed right behind the ConRhs blocks
    //  the last supernode's block read back zeros on the GPU -- not understood, the kernel is not LDS-critical)
    // (IFT kernel: behind both the phase-A data and the right-hand sides -- a workgroup that factors its own LU form
    //  (gradients) evaluates the linearization, which uses the mailbox while NodeP / Lane / Cold / the contact rows are alive)
    static constexpr int mail_off = (QUAD && GRAD == 2) ? a_end : (QUAD && GRAD) ? lds_imax(a_end, rhs_off + rhs_bytes) : a_end;
    static constexpr int red_off = lds_imax(a_end, mail_off + mail_need);
    static constexpr int qred_off = red_off + 64;                    // per-supernode values of the environment reductions (3 at a time)
    static constexpr int info_off = qred_off + (QUAD ? 3 * NSN * 8 : 0);   // SweepInfo per supernode (IFT kernels)
    static constexpr int info_end = info_off + ((QUAD && GRAD) ? NSN * (int)sizeof(SweepInfo) : 0);
    // Staging areas of the row-layout level passes (LaneProgram::factorize_rows; step kernels of the single-wavefront quad mapping with one
    // contact per body): A = 4 supernodes x 12 rows of S (stride ROW_RS doubles: odd, so that the 32 lanes of a ds_read_b64 group hit 32
    // bank pairs), later their 6 rows of L and of Dup; B = their 12 rows of U (stride ROW_US).  On the GPU A lies over the Newton step /
    // line-search base, which are dead between the line search and the next solve; B takes what is left of the 40 KB a workgroup may
    // use with four workgroups per CU.
    static constexpr int ROW_RS = 13, ROW_US = 7;
    static constexpr bool rows = QUAD && GRAD == 0 && ((NW == 1 && MAXC == 1) || NW == 2);    // (two wavefronts: staging areas per wavefront, behind the layout)
    static constexpr int stA_bytes = NW * 4 * 12 * ROW_RS * 8, stB_bytes = NW * 4 * 12 * ROW_US * 8;
    static_assert(4 * 6 * (ROW_RS + ROW_US) * 8 <= stA_bytes / NW, "L and Dup rows reuse area A");
    static constexpr int stA_off = !rows ? 0 : ls_in_lds ? ls_off : info_end;
    static constexpr int stB_off = !rows ? 0 : ls_in_lds ? info_end : info_end + stA_bytes;
    static_assert(!rows || !ls_in_lds || ls_slot * NSN >= stA_bytes, "area A must fit the line-search block it overlays");
    // ... and of the IFT kernel's LU-form passes (LaneProgram::factorize_rows_lu; GRAD == 1): A = the S rows in and the factor rows out (stride
    // LU_RS), the rows of L in and of m out over its first half; B = the rows of U in / T out (stride ROW_US) and of Dup in (stride LU_DS).  B
    // lies where the right-hand-side blocks reach beyond the phase-A data -- nothing of them is alive while the linearization is factored --
    // when that stretch is long enough (fp32 ABI: exactly), A behind everything else: 40 768 of the 40 960 bytes of four workgroups per CU.
    static constexpr int LU_RS = 12, LU_DS = 6;
    static constexpr bool rows_lu = QUAD && NW == 1 && MAXC == 1 && GRAD == 1;
    static constexpr int luA_bytes = 4 * 12 * LU_RS * 8, luB_bytes = 4 * 12 * ROW_US * 8 + 4 * 6 * LU_DS * 8;
    static constexpr bool luB_in_rhs = rows_lu && (rhs_off + rhs_bytes - a_end >= luB_bytes) && (mail_off >= a_end + luB_bytes);
    static constexpr int luA_off = info_end;
    static constexpr int luB_off = luB_in_rhs ? a_end : info_end + luA_bytes;
    static constexpr int bytes = rows ? stB_off + stB_bytes : rows_lu ? (luB_in_rhs ? info_end + luA_bytes : info_end + luA_bytes + luB_bytes) : info_end;
};
template <class TIO, class T, int MAXC, int GRAD, bool QUAD, bool LOCKSTEP = true, int NW = 1>
constexpr int step_lds_bytes() { return StepLds<TIO, T, MAXC, GRAD, QUAD, LOCKSTEP, NW>::bytes; }

// doubles per supernode in the step -> IFT hand-off record: v ω λ(6), s,γ of the joint limit, s,γ of the contacts, μ,
// and the pieces of the final linearization the IFT needs besides the factors: t_a, t_b (limit condensation), G134.
// NOTE: a step that skips the linearization after its converging iteration (need_factors = false: every launch without the refining
// kernels) leaves t_a / t_b / G134 -- and KernelArgs::diag_out -- at the LAST linearization it did perform, i.e. one iterate back; the plain
// IFT kernel never reads them (lu_prepare() evaluates the linearization at the restored solution before any use), the refining one
// re-evaluates them too (grad_entry MODE 2).  They travel for the explicit-inverse consumers only.
// (GEN: the general lane-mapping builds -- DJ_MLIM and DJ_CUT together -- append (s_up, s_lo, γ_up, γ_lo) of up to six limited coordinates and the multipliers
//  of the cut joints; the host sizes the buffer with GEN = true for mechanisms that take those builds)
static_assert((DJ_MLIM != 0) == (DJ_CUT != 0), "the general lane-mapping builds carry DJ_MLIM and DJ_CUT together");
template <int MAXC, bool GEN = (DJ_MLIM != 0)> constexpr int sol_record() { return 6 + 6 + 4 + 8 * MAXC + 1 + 12 + 18 * MAXC + (GEN ? 24 + 6 * NCUT : 0) + 1; }   // last: 1.0 if the environment's solves were being refined (DJ_REFINE)
template <int MAXC> constexpr int sol_flag_off() { return sol_record<MAXC>() - 1; }
template <int MAXC> constexpr int sol_mlim_off() { return 6 + 6 + 4 + 8 * MAXC + 1 + 12 + 18 * MAXC; }
template <int MAXC> constexpr int sol_cut_off() { return sol_mlim_off<MAXC>() + 24; }
// quad mapping: the factors themselves travel too (72 values per lane, stored [wave][72][64 lanes]: coalesced)
constexpr int FAC_PER_LANE = 72;

// The forward step and the IFT gradients are separate kernels (separate register allocations: the
// gradient sweeps must not cost the Newton loop its registers).  The IFT kernel rebuilds the lane
// program from (z, u), restores the converged solution from the hand-off record and re-linearizes
// there -- the same final linearization mehrotra! leaves behind (src/gradients/state.jl:78-84).
#if DJ_CUT
#define DJ_CUT_SETUP if constexpr (!QUAD) { prog.cutp = A.cuts; prog.ncut = (A.cuts && A.cutws) ? A.ncut : 0;                                              \
        prog.cws = A.cutws ? A.cutws + (size_t)(env < A.B ? env : 0) * CUTWS : nullptr;                                                             \
        for (int c_ = 0; c_ < NCUT; ++c_) for (int i = 0; i < 6; ++i)                                                                                \
            prog.cue[c_][i] = (has_u && c_ < prog.ncut && env < A.B && i < A.cuts[c_].nu_t + A.cuts[c_].nu_r) ? T(A.u[(size_t)env * G.nu + A.cuts[c_].u_off + i]) : T(0); \
        prog.cut_begin(); }
#else
#define DJ_CUT_SETUP
#endif
#if DJ_TSD && DJ_MLIM
#define DJ_TSD_SETUP prog.tsd = A.tsd ? A.tsd + (k < G.Nb ? k : G.Nb) : nullptr; prog.tlim = prog.tsd != nullptr && prog.tsd->nlim > 0; prog.mlp = A.mlim ? A.mlim + (k < G.Nb ? k : G.Nb) : nullptr;
#elif DJ_TSD
#define DJ_TSD_SETUP prog.tsd = A.tsd ? A.tsd + (k < G.Nb ? k : G.Nb) : nullptr; prog.tlim = prog.tsd != nullptr && prog.tsd->nlim > 0;
#elif DJ_MLIM
#define DJ_TSD_SETUP prog.mlp = A.mlim ? A.mlim + (k < G.Nb ? k : G.Nb) : nullptr;
#else
#define DJ_TSD_SETUP
#endif
#ifdef DJ_PROF2
#define DJ_P2_SETUP { prog.p2c = (unsigned long long*)wv.p2_; wv.sync(); if (lane < 24) prog.p2c[lane] = 0ull; wv.sync(); }
#else
#define DJ_P2_SETUP
#endif
#define DJ_LANE_SETUP(GRAD_LAYOUT, ENV_OK)                                                                                \
    const Globals<T>& G = A.G;                                                                                            \
    const int stride = QUAD ? 4 : 1;                                                                                      \
    const int envl = stride * G.S, E = wv.width() / envl;        /* lanes per environment, environments per workgroup */  \
    const int lane = wv.lane();                                                                                           \
    const int slot = lane / envl, k = (lane % envl) / stride, q = lane % stride;                                          \
    const int env = wave_index * E + slot;                                                                                \
    const bool active = (env < A.B) && (k < G.Nb) && (ENV_OK);                                                            \
    const int base = slot * envl;                                                                                         \
    typedef StepLds<TIO, T, MAXC, (GRAD_LAYOUT), QUAD, Wave::kLockstep, Wave::kWaves> LY;                                   \
    constexpr bool SHARE = QUAD && Wave::kLockstep;                                                                       \
    char* lds = (char*)wv.lds();                                                                                          \
    const NodeP<T>& Pg = A.nodes[k < G.Nb ? k : G.Nb];         /* idle supernode slots: the table's extra entry (node 0 without contacts) */ \
    if (SHARE) ((NodeSlot<T>*)lds)[lane / 4].P = Pg;           /* the four lanes store identical values */                \
    const NodeP<T>& P = SHARE ? ((NodeSlot<T>*)lds)[lane / 4].P : Pg;                                                     \
    Lane<T, MAXC> lane_local;                                                                                             \
    Cold<T, MAXC> cold_local;                                                                                             \
    ContactCold<T> pool_local[QUAD ? 1 : MAXC];                                                                           \
    Lane<T, MAXC>& lane_state = SHARE ? ((LaneSlot<T, MAXC>*)(lds + LY::lane_off))[lane / 4].L : lane_local;             \
    Cold<T, MAXC>& cold = LY::cold_in_lds ? ((Cold<T, MAXC>*)(lds + LY::cold_off))[SHARE ? lane / 4 : lane] : cold_local; \
    LaneProgram<T, TL, MAXC, QUAD, Wave> prog(wv, G, P, A.contacts, base, k, q, active, lane_state, cold);                \
    if (QUAD) {                                                                                                           \
        prog.cpool = (ContactCold<T>*)(lds + LY::pool_off); prog.pool_by_id = LY::pool_by_id;                             \
        prog.pool_base = LY::pool_by_id ? 0 : (SHARE ? lane / 4 : lane) * MAXC;                                           \
        prog.gb_lds = ((GRAD_LAYOUT) == 2) ? (void*)(((ConRhs<MAXC>*)(lds + LY::rhs_off)) + lane / 4) : (void*)(((QuadRhs<TIO>*)lds) + lane / 4); \
        prog.mail = (double*)(lds + LY::mail_off); prog.chpack_init();                                                    \
        DJ_P2_SETUP                                                                                                       \
        prog.qred = (double*)(lds + LY::qred_off);                                                                        \
        if (GRAD_LAYOUT) prog.sinfo = (SweepInfo*)(lds + LY::info_off);                                                   \
        if (A.msg) prog.msg = DJ_GLOBAL_PTR(T, A.msg) + (size_t)(env < A.B ? env : 0) * (size_t)A.msg_stride;                 \
        if (LY::ls_in_lds) prog.ls_lds = lds + LY::ls_off + (size_t)(lane / 4) * LY::ls_slot;                              \
        prog.rows_here = LY::rows || LY::rows_lu;                                                                         \
        if (LY::rows) { prog.stA = (double*)(lds + LY::stA_off); prog.stB = (double*)(lds + LY::stB_off); prog.rows_init(); }  \
        if (LY::rows_lu) { prog.stA = (double*)(lds + LY::luA_off); prog.stB = (double*)(lds + LY::luB_off); prog.rows_init(); } \
        if (SHARE) { prog.lane_slots = lds + LY::lane_off; prog.lane_slot_stride = (int)sizeof(LaneSlot<T, MAXC>); }              \
        if (A.lu) { prog.lu = DJ_GLOBAL_PTR(T, A.lu) + (size_t)wave_index * LU_PER_LANE * wv.width() + lane; prog.lu_stride = wv.width(); }                \
        if (A.blk) { prog.blk = DJ_GLOBAL_PTR(T, A.blk) + (size_t)wave_index * 90 * wv.width() + lane; prog.blk_stride = wv.width(); }     \
        if (A.ypark) prog.ypark = DJ_GLOBAL_PTR(T, A.ypark) + (size_t)wave_index * (size_t)A.ypark_stride + lane;                                \
    } else { prog.cpool = pool_local; prog.pool_by_id = false; prog.pool_base = 0; }                                      \
    DJ_TSD_SETUP                                                                                                          \
    T zb[13], ue[6] = {0, 0, 0, 0, 0, 0};                                                                                 \
    for (int i = 0; i < 13; ++i) zb[i] = active ? T(A.z[(size_t)env * 13 * G.Nb + 13 * k + i]) : T(0);                   \
    if (sizeof(TIO) < sizeof(T) && active) {   /* a narrower ABI type cannot hold a unit quaternion: the state it stands for is (x, v, q/|q|, ω) */ \
        const T iq_ = trcp(tsqrt(zb[6] * zb[6] + zb[7] * zb[7] + zb[8] * zb[8] + zb[9] * zb[9]));                         \
        for (int i = 6; i < 10; ++i) zb[i] *= iq_;                                                                        \
    }                                                                                                                     \
    const bool has_u = A.u != nullptr;                                                                                    \
    if (active && has_u) for (int i = 0; i < 6; ++i) if (i < P.nu_t + P.nu_r) ue[i] = T(A.u[(size_t)env * G.nu + P.u_off + i]); \
    T fe[6] = {0, 0, 0, 0, 0, 0};                                                                                         \
    const bool has_f = A.fext != nullptr;                                                                                 \
    if (active && has_f) for (int i = 0; i < 6; ++i) fe[i] = T(A.fext[(size_t)env * 6 * G.Nb + 6 * k + i]);             \
    prog.begin_step(zb, has_u ? ue : nullptr, has_f ? fe : nullptr);                                                        \
    DJ_CUT_SETUP

// IFT kernel entry: one call per lane
// MODE 0: state + control columns, pipelined sweeps; 1: contact-data columns; 2: state + control columns of the environments
// whose solves were being refined, column by column through the refined general solve (LDS layout of the step kernel:
// NodeP / Lane / Cold stay alive)
// CONT (MODE 0, iteration cap): false = the environments the step kernel finished itself (all of them without a cap), true = `wave_index`
// comes from the continuation list and the environments the continuation kernel finished are served
template <class TIO, class T, class TL, int MAXC, bool QUAD, class Wave, int MODE = 0, bool CONT = false>
DJ_HD void grad_entry(Wave& wv, const KernelArgs<TIO, T>& A, int wave_index) {
    static_assert(!CONT || MODE == 0, "continuation lists exist for the plain IFT kernel only");
    if constexpr (QUAD && MODE == 0) {
        if (A.resume != nullptr) {                              // (uniform) a workgroup without an environment of this launch's kind leaves at once
            const int stride_ = 4, envl_ = stride_ * A.G.S, E_ = wv.width() / envl_, lane_ = wv.lane();
            const int env_ = wave_index * E_ + lane_ / envl_, k_ = (lane_ % envl_) / stride_;
            const bool act_ = (env_ < A.B) && (k_ < A.G.Nb);
            if (!wv.any(act_ && (A.resume[(size_t)env_ * CARRY_PER_ENV + CARRY_MARK] != T(0)) == CONT)) return;
        }
    }
    if constexpr (QUAD && DJ_REFINE && (MODE == 0 || MODE == 2)) {
        // refined environments belong to the MODE 2 kernel; a workgroup without work for this kernel leaves at once (uniform)
        const int stride_ = 4, envl_ = stride_ * A.G.S, E_ = wv.width() / envl_, lane_ = wv.lane();
        const int env_ = wave_index * E_ + lane_ / envl_, k_ = (lane_ % envl_) / stride_;
        const bool act_ = (env_ < A.B) && (k_ < A.G.Nb);
        const bool fl_ = act_ && A.sol[((size_t)env_ * A.G.S + (size_t)k_) * sol_record<MAXC>() + sol_flag_off<MAXC>()] != T(0);
        if (MODE == 0 ? !wv.any(act_ && !fl_) : !wv.any(fl_)) return;
    }
    static_assert(MODE != 2 || Wave::kRefine || !(QUAD && DJ_REFINE), "the MODE 2 IFT kernel needs a refining Wave");
    DJ_LANE_SETUP(MODE == 0 ? 1 : MODE == 1 ? 2 : 0,
                  MODE == 2 ? A.sol[(size_t)env * G.S * sol_record<MAXC>() + sol_flag_off<MAXC>()] != T(0)
                            : (MODE != 0 || !QUAD || A.resume == nullptr || (A.resume[(size_t)env * CARRY_PER_ENV + CARRY_MARK] != T(0)) == CONT))
    bool flagged = false;
    {   // restore the converged solution (identical on the four lanes of a quad)
        const T* r = A.sol + ((size_t)env * G.S + (size_t)k) * sol_record<MAXC>();
        if (active) {
            flagged = r[sol_flag_off<MAXC>()] != T(0);
            for (int i = 0; i < 3; ++i) { prog.L.v[i] = r[i]; prog.L.w[i] = r[3 + i]; }
            for (int i = 0; i < 6; ++i) prog.L.lam[i] = r[6 + i];
            prog.L.ls[0] = r[12]; prog.L.ls[1] = r[13]; prog.L.lg[0] = r[14]; prog.L.lg[1] = r[15];
#pragma unroll
            for (int c = 0; c < MAXC; ++c) for (int i = 0; i < 4; ++i) { prog.L.cs[c][i] = r[16 + 8 * c + i]; prog.L.cg[c][i] = r[20 + 8 * c + i]; }
            prog.mu = r[16 + 8 * MAXC];
#if DJ_MLIM
            for (int m = 0; m < NLM; ++m) for (int i = 0; i < 2; ++i) { prog.L.mls[m][i] = r[sol_mlim_off<MAXC>() + 4 * m + i]; prog.L.mlg[m][i] = r[sol_mlim_off<MAXC>() + 4 * m + 2 + i]; }
#endif
#if DJ_CUT
            for (int c = 0; c < NCUT; ++c) for (int i = 0; i < 6; ++i) prog.clam[c][i] = r[sol_cut_off<MAXC>() + 6 * c + i];
#endif
            if (QUAD) {
                const T* r2 = r + 17 + 8 * MAXC;
                for (int i = 0; i < 6; ++i) { prog.F.t_a[i] = r2[i]; prog.F.t_b[i] = r2[6 + i]; }
#pragma unroll
                for (int c = 0; c < MAXC; ++c) if (c < P.ncontact) for (int i = 0; i < 18; ++i) prog.ccold(c).G134[i] = r2[12 + 18 * c + i];
            }
        }
    }
#ifdef DJ_PROF
    unsigned long long t_all = wv.clock();
#endif
    if constexpr (QUAD) { if (A.fac != nullptr) {              // the final factors of the Newton loop, as the step kernel left them (uniform)
        const T* f = A.fac + (size_t)wave_index * FAC_PER_LANE * wv.width() + lane;
        const int W = wv.width();
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 12; ++j) prog.F.Sq[i][j] = TL(f[(size_t)(12 * i + j) * W]);
#pragma unroll
            for (int j = 0; j < 6; ++j) { prog.F.Uq[i][j] = TL(f[(size_t)(36 + 6 * i + j) * W]); prog.F.Lq[j][i] = TL(f[(size_t)(54 + 6 * i + j) * W]); }
        }
    } } else {
        prog.linearize();                                     // lane mapping: rebuild the final linearization instead
    }
    if constexpr (MODE == 0) prog.gradients(A, env);
    else if constexpr (MODE == 2) {
        if constexpr (QUAD && DJ_REFINE) {
            // the un-factored rows of the final linearization (the step kernel's copy may be older than its last iterate's):
            // set_entries! once more into scratch rows; this also restores C134 / G134 and the limit rows of F
            TL S2[3][12], U2[3][6], L2[6][3];
            QuadBlocks<TL> K2(S2, U2, L2, q);
            prog.refine = flagged;
            prog.template evaluate<true>(K2);
            prog.condense_limits(K2);
            prog.store_blocks(K2);
            prog.template gradients<true>(A, env, flagged);
        }
    }
    else if constexpr (QUAD) prog.gradients_contact(A, env);
#ifdef DJ_PROF2
    if (active && q == 0 && A.vel && k == 8) { TIO* vo = A.vel + (size_t)env * 6 * G.Nb + 48; for (int i = 0; i < 16; ++i) vo[i] = TIO((double)prog.p2c[i]); }
#endif
#ifdef DJ_PROF
    if (active && q == 0 && A.vel && k == 2) { TIO* vo = A.vel + (size_t)env * 6 * G.Nb + 12; vo[0] = TIO((double)prog.pc[0]); vo[1] = TIO((double)prog.pc[1]); vo[2] = TIO((double)prog.pc[5]); vo[3] = TIO((double)prog.pc[6]); vo[4] = TIO((double)(wv.clock() - t_all)); vo[5] = TIO((double)prog.pc[4]); vo[6] = TIO((double)prog.pc[2]); vo[7] = TIO((double)prog.pc[3]); vo[8] = TIO((double)prog.pc[7]); }
#endif
}

// CONT = true: the continuation kernel's entry (Globals::iter_cap): `wave_index` is a workgroup of the step kernel that left
// environments unfinished (DJ_STATUS_CONTINUE in KernelArgs::status); their iterate comes back from the hand-off record, the
// loop scalars from KernelArgs::resume, and the Newton loop goes on.  The other environments of the workgroup stay as they are.
template <class TIO, class T, class TL, int MAXC, bool QUAD, class Wave, bool CONT = false>
DJ_HD void step_entry(Wave& wv, const KernelArgs<TIO, T>& A, int wave_index) {
    constexpr bool RF = QUAD && DJ_REFINE && Wave::kRefine;      // the refining build: re-solves the environments the plain kernel deferred
    static_assert(!(RF && CONT), "the continuation kernel is a plain build");
    if constexpr (RF) {
        const int envl_ = 4 * A.G.S, E_ = wv.width() / envl_, env_ = wave_index * E_ + wv.lane() / envl_;
        if (A.flag == nullptr || !wv.any(env_ < A.B && A.flag[env_] != 0)) return;          // nothing deferred in this workgroup (uniform)
    }
    DJ_LANE_SETUP(0, CONT ? A.status[env] == DJ_STATUS_CONTINUE : (!RF || A.flag[env] != 0))
#ifdef DJ_DEBUG
    prog.dbg_on = A.dbg != nullptr; prog.trace = getenv("DJ_TRACE") != nullptr;
    if (A.dbg && active && q == 0) prog.dbg = A.dbg + ((size_t)env * G.Nb + k) * 512;
#endif
    int iters = 0;
#ifdef DJ_PROF
    unsigned long long t_all = wv.clock();
#endif
    int status;
    if constexpr (CONT) {
        SolveCarry<T> carry;
        carry.undercut = G.undercut; carry.rvio = carry.bvio = T(0); carry.n = 0; carry.no_progress = 0; carry.excessive = 0;
        if (active) {                                          // the iterate the capped solve stopped at (identical on the four lanes of a quad)
            const T* r = A.sol + ((size_t)env * G.S + (size_t)k) * sol_record<MAXC>();
            for (int i = 0; i < 3; ++i) { prog.L.v[i] = r[i]; prog.L.w[i] = r[3 + i]; }
            for (int i = 0; i < 6; ++i) prog.L.lam[i] = r[6 + i];
            prog.L.ls[0] = r[12]; prog.L.ls[1] = r[13]; prog.L.lg[0] = r[14]; prog.L.lg[1] = r[15];
#pragma unroll
            for (int c = 0; c < MAXC; ++c) for (int i = 0; i < 4; ++i) { prog.L.cs[c][i] = r[16 + 8 * c + i]; prog.L.cg[c][i] = r[20 + 8 * c + i]; }
            prog.mu = r[16 + 8 * MAXC];
            const T* cr = A.resume + (size_t)env * CARRY_PER_ENV;
            carry.undercut = cr[0]; carry.rvio = cr[1]; carry.bvio = cr[2]; carry.n = (int)cr[3]; carry.no_progress = (int)cr[4]; carry.excessive = (int)cr[5];
        }
        carry.n = G.iter_cap;                                  // (wave-uniform: the capped loop leaves at exactly this iteration)
        status = prog.template mehrotra<true>(iters, /*need_factors=*/false, &carry);
    } else {
    // The factors of the final linearization leave this kernel only through A.fac (the refining IFT kernel reads them); the plain IFT
    // kernels linearize once more themselves (lu_prepare / linearize in grad_entry), so a differentiable step without refinement may
    // skip the set_entries! + factorization after the converging iteration exactly like a forward-only one.
    status = prog.mehrotra(iters, /*need_factors=*/QUAD && A.fac != nullptr, nullptr,
                           (!RF && G.iter_cap > 0 && A.resume != nullptr) ? DJ_GLOBAL_PTR(T, A.resume) + (size_t)(env < A.B ? env : 0) * CARRY_PER_ENV : nullptr);
    }
#ifdef DJ_PROF
    prog.pc[7] = wv.clock() - t_all;
#endif
    if constexpr (!RF && !CONT) {                             // iteration cap: the loop scalars of the unfinished solves, and this workgroup on the continuation list
        if (A.resume != nullptr) {
            // (the mark is per WORKGROUP: the IFT of a listed workgroup's finished environments waits for the continuation as well, so that
            //  no IFT launch ever works on part of a workgroup -- the per-lane staging areas KernelArgs::lu / ypark are indexed by workgroup)
            const bool listed = wv.any(active && status == DJ_STATUS_CONTINUE);
            if (active && q == 0 && k == 0) A.resume[(size_t)env * CARRY_PER_ENV + CARRY_MARK] = listed ? T(1) : T(0);
            if (listed && lane == 0) A.cont_list[wv.atomic_inc(A.cont_count)] = A.wave_base + wave_index;
        }
    }
    if constexpr (CONT) { if (Wave::kReplicas > 1 && wv.replica() != 0) return; }   // every replica holds the same result; the first one writes it
    T gr_env = T(0);
    if constexpr (QUAD && DJ_REFINE) { if (A.diag_out) { T v1[1] = {prog.growth}; prog.template env_reduce_quad_all<1>(v1); gr_env = v1[0]; } }
    if (active && q == 0 && A.sol) {                          // hand-off to the IFT kernel, in the state precision
        T* r = A.sol + ((size_t)env * G.S + (size_t)k) * sol_record<MAXC>();
        for (int i = 0; i < 3; ++i) { r[i] = prog.L.v[i]; r[3 + i] = prog.L.w[i]; }
        for (int i = 0; i < 6; ++i) r[6 + i] = prog.L.lam[i];
        r[12] = prog.L.ls[0]; r[13] = prog.L.ls[1]; r[14] = prog.L.lg[0]; r[15] = prog.L.lg[1];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) for (int i = 0; i < 4; ++i) { r[16 + 8 * c + i] = prog.L.cs[c][i]; r[20 + 8 * c + i] = prog.L.cg[c][i]; }
        r[16 + 8 * MAXC] = prog.mu;
#if DJ_MLIM
        for (int m = 0; m < NLM; ++m) for (int i = 0; i < 2; ++i) { r[sol_mlim_off<MAXC>() + 4 * m + i] = prog.L.mls[m][i]; r[sol_mlim_off<MAXC>() + 4 * m + 2 + i] = prog.L.mlg[m][i]; }
#endif
#if DJ_CUT
        for (int c = 0; c < NCUT; ++c) for (int i = 0; i < 6; ++i) r[sol_cut_off<MAXC>() + 6 * c + i] = prog.clam[c][i];
#endif
        if (QUAD) {
            T* r2 = r + 17 + 8 * MAXC;
            for (int i = 0; i < 6; ++i) { r2[i] = prog.F.t_a[i]; r2[6 + i] = prog.F.t_b[i]; }
#pragma unroll
            for (int c = 0; c < MAXC; ++c) if (c < P.ncontact) for (int i = 0; i < 18; ++i) r2[12 + 18 * c + i] = prog.ccold(c).G134[i];
        }
        r[sol_flag_off<MAXC>()] = (RF || status == DJ_STATUS_DEFERRED) ? T(1) : T(0);
    }
    if (!RF && A.flag && active && q == 0 && k == 0) A.flag[env] = status == DJ_STATUS_DEFERRED ? 1 : 0;
    if (QUAD && A.fac && (!RF || active)) {
        T* f = A.fac + (size_t)wave_index * FAC_PER_LANE * wv.width() + lane;
        const int W = wv.width();
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 12; ++j) f[(size_t)(12 * i + j) * W] = T(prog.F.Sq[i][j]);
#pragma unroll
            for (int j = 0; j < 6; ++j) { f[(size_t)(36 + 6 * i + j) * W] = T(prog.F.Uq[i][j]); f[(size_t)(54 + 6 * i + j) * W] = T(prog.F.Lq[j][i]); }
        }
    }
    if (active && q == 0) {
        T zn[13];
        prog.next_state(zn);
        TIO* o = A.z_next + (size_t)env * 13 * G.Nb + 13 * k;
        for (int i = 0; i < 13; ++i) o[i] = TIO(zn[i]);
        if (k == 0) { if (A.status) A.status[env] = status; if (A.iters) A.iters[env] = iters; if (A.mu_out) A.mu_out[env] = prog.mu; }
        if constexpr (QUAD && DJ_REFINE) { if (k == 0 && A.diag_out) { A.diag_out[2 * env] = prog.wstiff; A.diag_out[2 * env + 1] = gr_env; } }
        if (A.vel) { TIO* vo = A.vel + (size_t)env * 6 * G.Nb + 6 * k; for (int i = 0; i < 3; ++i) { vo[i] = TIO(prog.L.v[i]); vo[3 + i] = TIO(prog.L.w[i]); } }
        if (A.res) { TIO* ro = A.res + (size_t)env * 6 * G.Nb + 6 * k; for (int i = 0; i < 6; ++i) ro[i] = TIO(prog.rb[i]); }
#ifdef DJ_PROF2
        if (A.vel && k == 4 && G.Nb >= 9) { TIO* vo = A.vel + (size_t)env * 6 * G.Nb + 24; for (int i = 0; i < 24; ++i) vo[i] = TIO((double)prog.p2c[i]); }
#endif
#ifdef DJ_PROF
        if (A.vel && k == 0) { TIO* vo = A.vel + (size_t)env * 6 * G.Nb; for (int i = 0; i < 6; ++i) vo[i] = TIO((double)prog.pc[i]); }   // phases 0-5 (cycles)
        if (A.vel && k == 1) { TIO* vo = A.vel + (size_t)env * 6 * G.Nb + 6; vo[0] = TIO((double)prog.pc[7]); vo[1] = TIO((double)iters); vo[2] = TIO((double)prog.pc[6]); }
#endif
        if (A.joint_imp && P.n_imp > 0) {
            // get_solution order per joint: [tra: λ_t] [rot: s_up s_lo γ_up γ_lo λ_r]   (translational limits unsupported)
            TIO* jo = A.joint_imp + (size_t)env * G.n_joint_imp + P.imp_off;
            int o2 = 0;
            if (prog.tlim) { jo[o2++] = TIO(prog.L.ls[0]); jo[o2++] = TIO(prog.L.ls[1]); jo[o2++] = TIO(prog.L.lg[0]); jo[o2++] = TIO(prog.L.lg[1]); }   // [tra: s_up s_lo γ_up γ_lo λ_t]
#if DJ_MLIM
            // limits on several coordinates: η = [s_up(n); s_lo(n); γ_up(n); γ_lo(n); λ] per half (split_impulses, src/joints/joint.jl)
            if (prog.kMLim && prog.nlm() > 0) { const int nt_ = prog.mlp->nt;
                for (int i = 0; i < 2; ++i) for (int m = 0; m < nt_; ++m) jo[o2++] = TIO(prog.L.mls[m][i]);
                for (int i = 0; i < 2; ++i) for (int m = 0; m < nt_; ++m) jo[o2++] = TIO(prog.L.mlg[m][i]); }
#endif
            for (int i = 0; i < 3; ++i) if (i < P.nl_t) jo[o2++] = TIO(prog.L.lam[i]);
            if (P.nlim_r > 0) { jo[o2++] = TIO(prog.L.ls[0]); jo[o2++] = TIO(prog.L.ls[1]); jo[o2++] = TIO(prog.L.lg[0]); jo[o2++] = TIO(prog.L.lg[1]); }
#if DJ_MLIM
            if (prog.kMLim && prog.nlm() > 0) { const int nt_ = prog.mlp->nt, nr_ = prog.mlp->nr;
                for (int i = 0; i < 2; ++i) for (int m = 0; m < nr_; ++m) jo[o2++] = TIO(prog.L.mls[nt_ + m][i]);
                for (int i = 0; i < 2; ++i) for (int m = 0; m < nr_; ++m) jo[o2++] = TIO(prog.L.mlg[nt_ + m][i]); }
#endif
            for (int i = 0; i < 3; ++i) if (i < P.nl_r) jo[o2++] = TIO(prog.L.lam[3 + i]);
        }
#if DJ_CUT
        if constexpr (!QUAD) { if (A.joint_imp && k == 0) for (int c = 0; c < NCUT; ++c) if (c < prog.ncut) {     // the cut joints' multipliers, get_solution order [λ_t; λ_r]
            TIO* jo = A.joint_imp + (size_t)env * G.n_joint_imp + A.cuts[c].imp_off; int o2 = 0;
            for (int i = 0; i < 3; ++i) if (i < A.cuts[c].nl_t) jo[o2++] = TIO(prog.clam[c][i]);
            for (int i = 0; i < 3; ++i) if (i < A.cuts[c].nl_r) jo[o2++] = TIO(prog.clam[c][3 + i]);
        } }
#endif
        if (A.contact_sg) {
#pragma unroll
            for (int c = 0; c < MAXC; ++c) if (c < P.ncontact) {
                TIO* co = A.contact_sg + (size_t)env * (2 * NCV) * G.Nc + (2 * NCV) * P.contact[c];       // [s; γ] per contact (8; LinearContact builds: 12)
                for (int i = 0; i < NCV; ++i) { co[i] = TIO(prog.L.cs[c][i]); co[NCV + i] = TIO(prog.L.cg[c][i]); }
            }
        }
    }
}

} // namespace dj
