// dojo_host.hpp -- host-side "symbolic phase": turns the C-POD topology of include/dojo_hip.h
// into the per-supernode parameter tables the lane program consumes (elimination order =
// levels of the kinematic tree, leaves first).  Plain C++ (no HIP), shared
// by the library and by the CPU SIMT emulator used in the tests.
#pragma once
#include "dojo_device.hpp"
#include "../../include/dojo_hip.h"
#include <vector>
#include <string>
#include <cmath>
#include <cstdlib>

namespace dj {

struct HostModel {
    int Nb = 0, Nc = 0, S = 1, nu = 0, n_joint_imp = 0, maxch = 0, maxlevel = 0, maxc = 0;
    int contact_model = 0;       // 0: NonlinearContact, 1: ImpactContact, 2: LinearContact (one model per mechanism)
    std::vector<NodeP<double>> nodes;
    std::vector<ContactP<double>> contacts;
    std::vector<TraSD<double>> tsd;   // [Nb + 1] translational springs / dampers per supernode (+ the idle slot's zero entry)
    bool has_tsd = false;
    std::vector<MLimP<double>> mlim;  // [Nb + 1] joint limits on several coordinates / both halves (has_mlim: every limit of the mechanism lives here)
    bool has_mlim = false;            // ... the DJ_MLIM kernels (lane mapping)
    std::vector<NodeP<double>> cuts;  // loop-closing joints (a body's second, third ... parent joint): the joint fields of NodeP, parent = body a, child[0] = body b
    bool has_cut = false;             // ... the DJ_CUT kernels (lane mapping)
    bool has_loop = false;            // ... of which at least one is a loop-closing JOINT (has_cut alone may be a body-body contact carried as a cut element: still a tree)
    bool has_cc = false;              // a body-body contact between bodies that are no tree neighbours (a cut element as well; forward only)
    bool has_ss = false;          // a body-body contact (SphereSphereCollision): the DJ_SS kernels, forward only
    double dt = 0.01, input_scaling = 0.01, g[3] = {0, 0, -9.81};
    std::string error;
};

// the cut element of a body-body contact (LaneProgram::cut_*): parent = the contact's parent body a, child[0] = its child body b (the owner), contact[0] = its index
inline NodeP<double> contact_cut_node(int a, int b, int c) {
    NodeP<double> cn = NodeP<double>();
    cn.parent = a; cn.nchild = 1; for (int i = 0; i < MAXCH; ++i) cn.child[i] = b;
    cn.level = 0; cn.ncontact = 1; for (int i = 0; i < 8; ++i) cn.contact[i] = c;
    cn.nl_t = cn.nl_r = 0; cn.nlim_r = 0; cn.spring_on = cn.damper_on = 0; cn.u_off = 0; cn.nu_t = cn.nu_r = 0; cn.imp_off = 0; cn.n_imp = 0;
    cn.m = 0; for (int i = 0; i < 9; ++i) { cn.J[i] = cn.Ct[i] = cn.Cr[i] = cn.At[i] = cn.Ar[i] = 0; }
    for (int i = 0; i < 3; ++i) { cn.pa[i] = cn.pb[i] = cn.spring_off_r[i] = 0; }
    cn.qoff[0] = 1; cn.qoff[1] = cn.qoff[2] = cn.qoff[3] = 0; cn.spring_r = cn.damper_r = cn.lim_lo = cn.lim_hi = 0;
    return cn;
}
// A body-body contact along a TREE EDGE has builds of its own in the single-wavefront quad mapping (has_ss).  Where those do not serve the mechanism
// (more than 16 bodies, several contacts per body, translational springs / dampers / limits, several limits per joint, cut elements next to it) the
// contact is carried as a cut element of the general lane-mapping builds instead -- the cut machinery does not care whether a and b are neighbours.
inline int promote_tree_edge_contacts(HostModel& M) {
    for (int c = 0; c < (int)M.contacts.size(); ++c) {
        const ContactP<double>& Q = M.contacts[c];
        if (Q.kind != 1 || M.nodes[Q.cbody].parent != Q.pbody) continue;
        bool have = false;
        for (auto& cn : M.cuts) if (cn.ncontact == 1 && cn.contact[0] == c) have = true;
        if (have) continue;
        if ((int)M.cuts.size() >= NCUT) { M.error = "more than two cut elements (loop-closing joints + body-body contacts outside the single-wavefront quad builds) are not supported"; return DOJO_ERR_UNSUPPORTED; }
        M.cuts.push_back(contact_cut_node(Q.pbody, Q.cbody, c)); M.has_cut = true; M.has_cc = true;
    }
    M.has_ss = false;
    return DOJO_OK;
}

inline int build_host_model(const DojoTopology& tp, HostModel& M) {
    M.Nb = tp.n_bodies; M.Nc = tp.n_contacts; M.dt = tp.timestep; M.input_scaling = tp.input_scaling;
    for (int i = 0; i < 3; ++i) M.g[i] = tp.gravity[i];
    if (M.Nb < 1) { M.error = "mechanism has no bodies"; return DOJO_ERR_INVALID; }
    if (M.Nb > 64) { M.error = "more than 64 bodies per environment is not supported by the lane=supernode mapping"; return DOJO_ERR_UNSUPPORTED; }
    int S = 1; while (S < M.Nb) S <<= 1; M.S = S;
    M.nodes.assign(M.Nb, NodeP<double>());
    { TraSD<double> z0; z0.spring = z0.damper = 0; z0.off[0] = z0.off[1] = z0.off[2] = 0; z0.lim_lo = z0.lim_hi = 0; z0.nlim = 0; M.tsd.assign(M.Nb + 1, z0); M.has_tsd = false; }
    std::vector<int> pj(M.Nb, -1);
    int uoff = 0, ioff = 0;
    // limits on several coordinates of a joint half (all of its free ones, src/joints/limits.jl:1-17) or on both halves of a joint: the DJ_MLIM
    // kernels; then EVERY limit of the mechanism goes through their table instead of NodeP::nlim_r / TraSD::nlim
    { MLimP<double> z0; z0.nt = z0.nr = 0; for (int i = 0; i < 6; ++i) z0.lo[i] = z0.hi[i] = 0; M.mlim.assign(M.Nb + 1, z0); M.has_mlim = false; }
    for (int j = 0; j < tp.n_joints; ++j) {
        const DojoJoint& J = tp.joints[j];
        for (int h = 0; h < 2; ++h) { const DojoJointHalf& H = h ? J.rot : J.tra;
            if (H.nlim != 0 && H.nlim != 3 - H.nl) { M.error = "joint limits cover all free coordinates of a joint half or none (nlim = 0 or 3 - nl)"; return DOJO_ERR_INVALID; } }
        if (J.tra.nlim > 1 || J.rot.nlim > 1 || (J.tra.nlim > 0 && J.rot.nlim > 0)) M.has_mlim = true;
    }
    for (int j = 0; j < tp.n_joints; ++j) {
        const DojoJoint& J = tp.joints[j];
        if (J.child < 0 || J.child >= M.Nb || J.parent >= M.Nb) { M.error = "joint with invalid body index"; return DOJO_ERR_INVALID; }
        const bool is_cut = pj[J.child] >= 0;       // the body already hangs on an earlier joint: this one closes a loop (src/solver/linear_system.jl:4-5)
        NodeP<double> cutnode;
        if (is_cut) {
            if (J.parent < 0) { M.error = "a loop-closing joint to the origin is not supported (both of its bodies must be bodies of the tree)"; return DOJO_ERR_UNSUPPORTED; }
            if ((int)M.cuts.size() >= NCUT) { M.error = "more than two loop-closing joints are not supported"; return DOJO_ERR_UNSUPPORTED; }
            if (J.tra.nlim != 0 || J.rot.nlim != 0) { M.error = "limits on a loop-closing joint are not supported"; return DOJO_ERR_UNSUPPORTED; }
            if (J.tra.nl < 3 && ((J.spring_on && J.tra.spring != 0) || (J.damper_on && J.tra.damper != 0))) { M.error = "translational springs / dampers on a loop-closing joint are not supported"; return DOJO_ERR_UNSUPPORTED; }
            cutnode = NodeP<double>(); cutnode.nchild = 1; for (int i = 0; i < MAXCH; ++i) cutnode.child[i] = J.child;
            cutnode.level = 0; cutnode.ncontact = 0; for (int i = 0; i < 8; ++i) cutnode.contact[i] = 0;
            cutnode.m = 0; for (int i = 0; i < 9; ++i) cutnode.J[i] = 0;
            M.has_cut = true; M.has_loop = true;
        } else pj[J.child] = j;
        NodeP<double>& P = is_cut ? cutnode : M.nodes[J.child];
        P.parent = J.parent;
        P.nl_t = J.tra.nl; P.nl_r = J.rot.nl; P.nlim_r = M.has_mlim ? 0 : J.rot.nlim;
        if (M.has_mlim && !is_cut) {
            MLimP<double>& ml = M.mlim[J.child];
            ml.nt = J.tra.nlim; ml.nr = J.rot.nlim;
            for (int i = 0; i < J.tra.nlim; ++i) { ml.lo[i] = J.tra.limit_lo[i]; ml.hi[i] = J.tra.limit_hi[i]; }
            for (int i = 0; i < J.rot.nlim; ++i) { ml.lo[J.tra.nlim + i] = J.rot.limit_lo[i]; ml.hi[J.tra.nlim + i] = J.rot.limit_hi[i]; }
        } else
        if (J.tra.nlim != 0) {            // one limited translational coordinate: joints with nl_t = 2, and no rotational limit on the same joint
            if (J.tra.nlim > 1 || J.tra.nl != 2) { M.error = "translational limits need a one-dimensional translational joint (Prismatic type)"; return DOJO_ERR_UNSUPPORTED; }
            if (J.rot.nlim != 0) { M.error = "limits on both halves of one joint are not supported"; return DOJO_ERR_UNSUPPORTED; }
        }
        if (!M.has_mlim && J.rot.nlim > 1) { M.error = "joint limits on more than one rotational coordinate are not supported"; return DOJO_ERR_UNSUPPORTED; }
        if (!M.has_mlim && J.rot.nlim == 1 && J.rot.nl != 2) { M.error = "rotational limits need a one-dimensional rotational joint"; return DOJO_ERR_UNSUPPORTED; }
        if (J.tra.nl < 3 && ((J.spring_on && J.tra.spring != 0) || (J.damper_on && J.tra.damper != 0))) {
            // translational spring / damper (joints with free translations): a table of its own next to the nodes, DJ_TSD kernels
            M.has_tsd = true;
            TraSD<double>& sd = M.tsd[J.child];
            sd.spring = J.spring_on ? J.tra.spring : 0.0; sd.damper = J.damper_on ? J.tra.damper : 0.0;
            for (int i = 0; i < 3 - J.tra.nl; ++i) sd.off[i] = J.tra.spring_offset[i];
        }
        if (J.tra.nlim == 1 && !M.has_mlim) {
            M.has_tsd = true;
            TraSD<double>& sd = M.tsd[J.child];
            sd.nlim = 1; sd.lim_lo = J.tra.limit_lo[0]; sd.lim_hi = J.tra.limit_hi[0];
        }
        P.spring_on = J.spring_on; P.damper_on = J.damper_on;
        P.nu_t = 3 - J.tra.nl; P.nu_r = 3 - J.rot.nl;
        P.u_off = uoff; uoff += P.nu_t + P.nu_r;
        P.n_imp = J.tra.nl + 4 * J.tra.nlim + J.rot.nl + 4 * J.rot.nlim;
        P.imp_off = ioff; ioff += P.n_imp;
        for (int i = 0; i < 9; ++i) { P.Ct[i] = P.Cr[i] = P.At[i] = P.Ar[i] = 0; }
        for (int i = 0; i < 3 * J.tra.nl; ++i) P.Ct[i] = J.tra.cmask[i];
        for (int i = 0; i < 3 * J.rot.nl; ++i) P.Cr[i] = J.rot.cmask[i];
        for (int i = 0; i < 3 * (3 - J.tra.nl); ++i) P.At[i] = J.tra.amask[i];
        for (int i = 0; i < 3 * (3 - J.rot.nl); ++i) P.Ar[i] = J.rot.amask[i];
        for (int i = 0; i < 3; ++i) { P.pa[i] = J.vertex_parent[i]; P.pb[i] = J.vertex_child[i]; P.spring_off_r[i] = J.rot.spring_offset[i]; }
        for (int i = 0; i < 4; ++i) P.qoff[i] = J.orientation_offset[i];
        P.spring_r = J.rot.spring; P.damper_r = J.rot.damper;
        P.lim_lo = J.rot.limit_lo[0]; P.lim_hi = J.rot.limit_hi[0];
        if (is_cut) M.cuts.push_back(cutnode);
    }
    M.nu = uoff; M.n_joint_imp = ioff;
    // NB: the u / joint-impulse offsets above follow mechanism.joints order because joints are visited in that order
    for (int b = 0; b < M.Nb; ++b) {
        if (pj[b] < 0) { M.error = "body without a parent joint (every body needs one; use a Floating joint to the origin)"; return DOJO_ERR_UNSUPPORTED; }
        NodeP<double>& P = M.nodes[b];
        P.m = tp.bodies[b].mass;
        for (int i = 0; i < 9; ++i) P.J[i] = tp.bodies[b].inertia[i];
        P.nchild = 0; P.ncontact = 0;
        for (int i = 0; i < MAXCH; ++i) P.child[i] = 0;
        for (int i = 0; i < 8; ++i) P.contact[i] = 0;
    }
    for (int b = 0; b < M.Nb; ++b) {
        int p = M.nodes[b].parent;
        if (p >= 0) {
            if (M.nodes[p].nchild >= MAXCH) { M.error = "more than 4 child joints on one body is not supported"; return DOJO_ERR_UNSUPPORTED; }
            M.nodes[p].child[M.nodes[p].nchild++] = b;
        }
    }
    // levels (distance from the origin-connected root of each tree)
    M.maxlevel = 0; M.maxch = 0;
    for (int b = 0; b < M.Nb; ++b) {
        int lev = 0, p = M.nodes[b].parent, guard = 0;
        while (p >= 0) { ++lev; p = M.nodes[p].parent; if (++guard > M.Nb) { M.error = "cycle in the parent relation"; return DOJO_ERR_INVALID; } }
        M.nodes[b].level = lev;
        if (lev > M.maxlevel) M.maxlevel = lev;
        if (M.nodes[b].nchild > M.maxch) M.maxch = M.nodes[b].nchild;
    }
    M.contacts.assign(M.Nc, ContactP<double>());
    M.maxc = 0;
    for (int c = 0; c < M.Nc; ++c) {
        const DojoContact& K = tp.contacts[c];
        if (K.body < 0 || K.body >= M.Nb) { M.error = "contact with invalid body index"; return DOJO_ERR_INVALID; }
        // a body-body contact (SphereSphereCollision) belongs to the supernode of its CHILD body, next to the joint that ties it to the parent
        int owner = K.body;
        if (K.collision == 1) {
            if (K.child_body < 0 || K.child_body >= M.Nb || K.child_body == K.body) { M.error = "body-body contact: invalid child_body"; return DOJO_ERR_INVALID; }
            owner = K.child_body;
            if (M.nodes[K.child_body].parent == K.body) M.has_ss = true;              // an edge of the tree: the DJ_SS builds of the quad mapping
            else {
                // SphereSphereCollision between ANY two bodies (src/contacts/collisions/sphere_sphere.jl:11-16): not an edge of the tree -- a cut element of
                // the general lane-mapping builds, like a loop-closing joint (LaneProgram::cut_*)
                if ((int)M.cuts.size() >= NCUT) { M.error = "more than two cut elements (loop-closing joints + body-body contacts between bodies that are no tree neighbours) are not supported"; return DOJO_ERR_UNSUPPORTED; }
                M.cuts.push_back(contact_cut_node(K.body, K.child_body, c)); M.has_cut = true; M.has_cc = true;
            }
        } else if (K.collision != 0) { M.error = "unknown collision (0 = SphereHalfSpaceCollision, 1 = SphereSphereCollision)"; return DOJO_ERR_UNSUPPORTED; }
        NodeP<double>& P = M.nodes[owner];
        if (P.ncontact >= 8) { M.error = "more than 8 contacts on one body is not supported"; return DOJO_ERR_UNSUPPORTED; }
        P.contact[P.ncontact++] = c;
        if (P.ncontact > M.maxc) M.maxc = P.ncontact;
        ContactP<double>& Q = M.contacts[c];
        for (int i = 0; i < 3; ++i) { Q.n[i] = K.normal[i]; Q.o[i] = K.origin[i]; Q.off[i] = K.offset[i]; }
        if (K.model != 0 && K.model != 1 && K.model != 2) { M.error = "unknown contact model (0 = NonlinearContact, 1 = ImpactContact, 2 = LinearContact)"; return DOJO_ERR_UNSUPPORTED; }
        if (c > 0 && K.model != M.contact_model) { M.error = "mixing contact models in one mechanism is not supported"; return DOJO_ERR_UNSUPPORTED; }
        M.contact_model = K.model;
        // ImpactContact (src/contacts/impact.jl) runs as the nonlinear model without its friction block: no tangents,
        // no friction coefficient, and the device pins the friction variables at the neutral vector
        for (int i = 0; i < 6; ++i) Q.t[i] = K.model == 1 ? 0.0 : K.tangent[i];
        Q.r = K.radius; Q.mu = K.model == 1 ? 0.0 : K.friction_coefficient;
        Q.kind = K.collision; Q.r2 = K.collision == 1 ? K.child_radius : 0.0;
        Q.pbody = K.collision == 1 ? K.body : -1; Q.cbody = owner;
        for (int i = 0; i < 3; ++i) Q.o2[i] = K.collision == 1 ? K.child_origin[i] : 0.0;
    }
    return DOJO_OK;
}

template <class T> inline NodeP<T> cast_node(const NodeP<double>& a) {
    NodeP<T> b;
    b.parent = a.parent; b.level = a.level; b.nchild = a.nchild; for (int i = 0; i < MAXCH; ++i) b.child[i] = a.child[i];
    b.ncontact = a.ncontact; for (int i = 0; i < 8; ++i) b.contact[i] = a.contact[i];
    b.nl_t = a.nl_t; b.nl_r = a.nl_r; b.nlim_r = a.nlim_r; b.spring_on = a.spring_on; b.damper_on = a.damper_on;
    b.u_off = a.u_off; b.nu_t = a.nu_t; b.nu_r = a.nu_r; b.imp_off = a.imp_off; b.n_imp = a.n_imp;
    b.m = T(a.m);
    for (int i = 0; i < 9; ++i) { b.J[i] = T(a.J[i]); b.Ct[i] = T(a.Ct[i]); b.Cr[i] = T(a.Cr[i]); b.At[i] = T(a.At[i]); b.Ar[i] = T(a.Ar[i]); }
    for (int i = 0; i < 3; ++i) { b.pa[i] = T(a.pa[i]); b.pb[i] = T(a.pb[i]); b.spring_off_r[i] = T(a.spring_off_r[i]); }
    for (int i = 0; i < 4; ++i) b.qoff[i] = T(a.qoff[i]);
    b.spring_r = T(a.spring_r); b.damper_r = T(a.damper_r); b.lim_lo = T(a.lim_lo); b.lim_hi = T(a.lim_hi);
    return b;
}
template <class T> inline ContactP<T> cast_contact(const ContactP<double>& a) {
    ContactP<T> b;
    for (int i = 0; i < 3; ++i) { b.n[i] = T(a.n[i]); b.o[i] = T(a.o[i]); b.off[i] = T(a.off[i]); }
    for (int i = 0; i < 6; ++i) b.t[i] = T(a.t[i]);
    b.r = T(a.r); b.mu = T(a.mu); b.r2 = T(a.r2); b.kind = a.kind; b.pbody = a.pbody; b.cbody = a.cbody;
    for (int i = 0; i < 3; ++i) b.o2[i] = T(a.o2[i]);
    return b;
}
// refine_w: stiffness (max γ/s over the cones of an environment) beyond which the device refines its linear solves against
// the uncondensed system (DJ_REFINE, dojo_device.hpp); INFINITY = never
template <class T> inline Globals<T> make_globals(const HostModel& M, const DojoSolverOptions& o, int grad_mode, double refine_w = INFINITY) {
    Globals<T> G;
    G.refine_w = T(refine_w);
    G.dt = T(M.dt); G.idt2 = T(1.0 / (M.dt * M.dt)); G.input_scaling = T(M.input_scaling);
    for (int i = 0; i < 3; ++i) G.g[i] = T(M.g[i]);
    G.rtol = T(o.rtol); G.btol = T(o.btol); G.undercut = T(o.undercut); G.no_progress_undercut = T(o.no_progress_undercut);
    G.max_iter = o.max_iter; G.max_ls = o.max_ls; G.no_progress_max = o.no_progress_max;
    G.iter_cap = 0;              // (the launch decides: dojo_hip.hip launch(), tests/emu)
    G.Nb = M.Nb; G.Nc = M.Nc; G.S = M.S; G.nu = M.nu; G.n_joint_imp = M.n_joint_imp; G.maxch = M.maxch; G.maxlevel = M.maxlevel; G.grad_mode = grad_mode; G.contact_model = M.contact_model;
    for (int l = 0; l < 64; ++l) G.maxch_lev[l] = 0;
    for (int b = 0; b < M.Nb; ++b) { int l = M.nodes[b].level; if (l < 64 && M.nodes[b].nchild > G.maxch_lev[l]) G.maxch_lev[l] = (unsigned char)M.nodes[b].nchild; }
    for (int w = 0; w < 4; ++w) G.maxch_pack[w] = 0ull;
    for (int l = 0; l < 64; ++l) G.maxch_pack[l >> 4] |= (unsigned long long)(G.maxch_lev[l] & 15) << (4 * (l & 15));      // (MAXCH = 4 children per body)
    G.rows = 0;
    for (int t = 0; t < 16; ++t) { G.rp_lev[t] = 0; G.rp_slot4[t] = -1; G.rp_slot4_w1[t] = -1; for (int c = 0; c < MAXCH; ++c) { G.rp_child4[t][c] = -1; G.rp_child4_w1[t][c] = -1; } }
    return G;
}
// The row-layout level passes of the factorization (Globals::rows / rp_slot / rp_lev) for a 64-lane wavefront that holds 16 / S
// environments of this mechanism in the quad mapping: the supernodes of a level, over all environments of the wavefront, four per pass.
// mode < 0: where the pass count says it pays (a row pass costs about a third of a quad pass, which serves every supernode of a level at
// once: DESIGN.md section 6, tools/ubench/gj_rows.hip); 0: never; > 0: always.  DOJO_ROWS in the environment overrides `mode`.
template <class T> inline void set_row_passes(Globals<T>& G, const HostModel& M, int mode = -1) {
    G.rows = 0;
    for (int t = 0; t < 16; ++t) { G.rp_slot4_w1[t] = -1; for (int c = 0; c < MAXCH; ++c) G.rp_child4_w1[t][c] = -1; }
    if (const char* e = std::getenv("DOJO_ROWS")) mode = std::atoi(e);
    auto put = [](int& w, int g_, int v) { w = (int)(((unsigned)w & ~(0xFFu << (8 * g_))) | ((unsigned)(v & 0xFF) << (8 * g_))); };
    if (M.S == 32) {
        // two-wavefront workgroups (one environment of 17..32 bodies): slot = body; wavefront w runs the passes of the slots 16 w .. 16 w + 15, both
        // in step -- a level takes as many passes as the fuller of its two halves needs (Atlas: one per level, 11 passes of at most 3 + 3 supernodes)
        if (mode == 0) return;
        int t = 0;
        for (int lev = M.maxlevel; lev >= 0; --lev) {
            int n[2] = {0, 0}, t0 = t;
            for (int b = 0; b < M.Nb; ++b) if (M.nodes[b].level == lev) {
                const int w = b >> 4, pass = t0 + n[w] / 4, g = n[w] % 4;
                if (pass >= 16) return;
                int* slots = w ? G.rp_slot4_w1 : G.rp_slot4;
                for (int c = 0; c < MAXCH; ++c) put(w ? G.rp_child4_w1[pass][c] : G.rp_child4[pass][c], g, c < M.nodes[b].nchild ? M.nodes[b].child[c] : -1);
                put(slots[pass], g, b); ++n[w];
            }
            const int np = std::max((n[0] + 3) / 4, (n[1] + 3) / 4);
            for (int p_ = 0; p_ < np; ++p_) { if (t0 + p_ >= 16) return; G.rp_lev[t0 + p_] = lev | ((int)G.maxch_lev[lev] << 8); }
            t = t0 + np;
        }
        if (mode < 0 && 10 * t >= 16 * (M.maxlevel + 1)) return;      // (a quad pass of two wavefronts costs what it costs one: rows pay while passes < 1.6 x levels)
        G.rows = t;
        return;
    }
    if (mode == 0 || M.S > 16 || M.S < 1 || 16 % M.S != 0) return;
    const int E = 16 / M.S;
    int t = 0;
    for (int lev = M.maxlevel; lev >= 0; --lev) {
        int g = 0;
        for (int e = 0; e < E; ++e) for (int b = 0; b < M.Nb; ++b) if (M.nodes[b].level == lev) {
            if (g == 4) { ++t; g = 0; }
            if (t >= 16) return;                 // (cannot happen: 16 slots, each in one pass)
            for (int c = 0; c < MAXCH; ++c) put(G.rp_child4[t][c], g, c < M.nodes[b].nchild ? e * M.S + M.nodes[b].child[c] : -1);
            put(G.rp_slot4[t], g++, e * M.S + b); G.rp_lev[t] = lev | ((int)G.maxch_lev[lev] << 8);
        }
        if (g > 0) ++t;
    }
    if (mode < 0 && 10 * t >= 32 * (M.maxlevel + 1)) return;
    G.rows = t;
}
inline DojoSolverOptions default_options() {
    DojoSolverOptions o; o.rtol = 1e-6; o.btol = 1e-4; o.undercut = INFINITY; o.no_progress_undercut = 10.0;
    o.max_iter = 50; o.max_ls = 10; o.no_progress_max = 3; o.reserved = 0; return o;
}

} // namespace dj
