// emu.cpp -- thread-based SIMT emulator for the lane program (TEST INFRASTRUCTURE).
//
// Runs the *same* device source (dojo.jl_amd/csrc/dojo_device.hpp) on the CPU: every lane of a
// "wave" is a std::thread and wave shuffles / votes are implemented with a barrier.  This lets
// the CPU-only test tier (-m "not gpu") check the shipped device algorithm against the oracle.
// It is not a product path: libdojo_hip.so has no CPU fallback.
#define DJ_DEBUG 1
#ifndef DJ_TSD
#define DJ_TSD 1        // the emulator carries the translational spring / damper code (KernelArgs::tsd decides) ...
#endif
#define DJ_SS 1         // ... and the body-body contact code (ContactP::kind decides)
// (-DDJ_TSD=0: the macro set of the GPU's body-body contact builds -- code paths that differ by build flags, like the rows of U the
//  Schur complement may skip, are then the GPU's)
#include "../../dojo.jl_amd/csrc/dojo_host.hpp"
#include <thread>
#include <limits>
#include <mutex>
#include <condition_variable>
#include <cstring>
#include <atomic>
#include <memory>
#include <cstdio>

namespace {

// Sense-reversing barrier: spin briefly, then yield.  (A mutex + condition variable costs two context switches per
// thread per barrier; the lane program crosses tens of thousands of barriers per step.)
struct Barrier {
    std::atomic<int> count{0}; std::atomic<int> gen{0}; int n;
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
        int g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            count.store(0, std::memory_order_relaxed);
            gen.store(g + 1, std::memory_order_release);
        } else {
            int spins = 0;
            while (gen.load(std::memory_order_acquire) == g) { if (++spins > 64) std::this_thread::yield(); }
        }
    }
};

struct Shared {
    int W; Barrier bar; std::vector<double> slot; std::vector<int> islot; std::vector<double> lds;
    // "LDS" of the emulated workgroup.  The emulator's lanes are not in lock step, so Cold / the contact pool are per LANE here
    // (StepLds, LOCKSTEP = false): the MAXC = 8 layout needs 167 KB (twice that with the body-body contact rows of DJ_SS) -- more than the GPU's 160 KB, which the lock-step layouts fit.
    static constexpr size_t kLdsBytes = 512 * 1024;
    explicit Shared(int w) : W(w), bar(w), slot(w), islot(w), lds(kLdsBytes / 8) {}
};

template <int NW, bool RF = false>
struct EmuWaveT {
    static constexpr bool kRefine = RF;         // the refining build of the kernels (DJ_REFINE)
    static constexpr bool kLockstep = false;    // lanes are free-running threads between barriers
    static constexpr int kWaves = NW;           // selects the LDS layout and the workgroup-reduction code paths of the multi-wave kernels
    static constexpr int kReplicas = 1;
    int atomic_inc(int* p) { static std::mutex m; std::lock_guard<std::mutex> g(m); return (*p)++; }
    Shared* sh; int l;
    static double rcp(double a) { return 1.0 / a; }
    static float rcp(float a) { return 1.0f / a; }
    int lane() const { return l; }
    int width() const { return sh->W; }
    void* lds() const { return (void*)sh->lds.data(); }
    void sync() { sh->bar.wait(); }
    void sync_mem() { sh->bar.wait(); }
    template <class T> T shfl(T v, int src) {
        sh->slot[l] = (double)v;
        sh->bar.wait();
        T r = (T)sh->slot[src];
        sh->bar.wait();
        return r;
    }
    int shfl(int v, int src) {
        sh->islot[l] = v;
        sh->bar.wait();
        int r = sh->islot[src];
        sh->bar.wait();
        return r;
    }
    template <class V> V quad_bcast(V v, int o) { return shfl(v, (l & ~3) + o); }
    template <class V> V quad_xor(V v, int m) { return shfl(v, l ^ m); }
    // the row layout of the level passes (LaneProgram::factorize_rows): lane p of my 16-lane row, alone and inside a multiply-add
    static constexpr bool kRows = NW <= 2;
    static constexpr int kWidth = 0;            // (the emulator's workgroup width is a run-time value: width())
    static constexpr bool kWaveReduce = false;  // (GPU: the 16-slot reductions on the DPP path, Wave::reduce_quads16)
    template <class V> V row_bcast(V v, int p) { return shfl(v, (l & ~15) + p); }
    template <class V> void row_fmac(V& acc, V src, V f, int p) { acc += shfl(src, (l & ~15) + p) * f; }
    template <int P, class V> V row_bcast_c(V v) { return row_bcast(v, P); }
    template <int P, class V> void row_fmac_c(V& acc, V src, V f) { row_fmac(acc, src, f, P); }
    static void dpp_settle() {}
    bool any(bool p) {
        sh->islot[l] = p ? 1 : 0;
        sh->bar.wait();
        int r = 0; for (int i = 0; i < sh->W; ++i) r |= sh->islot[i];
        sh->bar.wait();
        return r != 0;
    }
    // workgroup reductions (used when kWaves > 1)
    template <class OP> double wg_reduce(double v, OP op) {
        sh->slot[l] = v; sh->bar.wait();
        double r = sh->slot[0]; for (int i = 1; i < sh->W; ++i) r = op(r, sh->slot[i]);
        sh->bar.wait();
        return r;
    }
    double wg_max(double v) { return wg_reduce(v, [](double a, double b) { return a > b ? a : b; }); }
    double wg_min(double v) { return wg_reduce(v, [](double a, double b) { return a < b ? a : b; }); }
    double wg_sum(double v) { return wg_reduce(v, [](double a, double b) { return a + b; }); }
    int wg_or(int v) { return any(v != 0) ? 1 : 0; }
};

// the continuation kernel's replicas (Globals::iter_cap): R copies of the workgroup's threads, each copy with "LDS" and barrier of its
// own, plus one barrier and one exchange block across all of them
struct RepShared { Barrier bar; std::vector<double> xchg; RepShared(int threads, int n) : bar(threads), xchg(n) {} };
template <int R>
struct EmuWaveRep : EmuWaveT<1> {
    static constexpr int kReplicas = R;
    RepShared* rs; int rep;
    int replica() const { return rep; }
    double* replica_xchg() const { return rs->xchg.data(); }
    void replica_sync() { rs->bar.wait(); }
};

template <class TIO, class T, class TL, int MAXC, bool QUAD, int NW = 1>
void run(const dj::HostModel& M, const DojoSolverOptions& opts, int grad_mode, int B, int W,
         const double* z, const double* u, double* z_next, int* status, int* iters,
         double* vel, double* jimp, double* csg, double* dz, double* du, double* dbg, double* dc = nullptr, double* storage = nullptr, const double* fext = nullptr) {
    std::vector<dj::NodeP<T>> nodes; for (auto& n : M.nodes) nodes.push_back(dj::cast_node<T>(n));
    // one extra entry for the idle supernode slots of a workgroup: node 0 without contacts (a slot that kept node 0's
    // contacts would write the same contact-pool rows as the real node 0 in the two-wavefront mapping)
    { dj::NodeP<T> idle = nodes[0]; idle.ncontact = 0; for (int i = 0; i < 8; ++i) idle.contact[i] = 0; nodes.push_back(idle); }
    std::vector<dj::ContactP<T>> contacts; for (auto& c : M.contacts) contacts.push_back(dj::cast_contact<T>(c));
    if (contacts.empty()) contacts.push_back(dj::ContactP<T>());
    int nz = 13 * M.Nb, nx = 12 * M.Nb;
    auto castv = [](const double* p, size_t n) { std::vector<TIO> v(p ? n : 0); for (size_t i = 0; i < v.size(); ++i) v[i] = TIO(p[i]); return v; };
    std::vector<TIO> zt = castv(z, (size_t)B * nz), ut = castv(u, (size_t)B * M.nu);
    std::vector<TIO> zn((size_t)B * nz), velt(vel ? (size_t)B * 6 * M.Nb : 0), jt(jimp ? (size_t)B * std::max(M.n_joint_imp, 1) : 0),
        ct(csg ? (size_t)B * (2 * dj::NCV) * std::max(M.Nc, 1) : 0), dzt(dz ? (size_t)B * nx * nx : 0), dut(du ? (size_t)B * nx * std::max(M.nu, 1) : 0);
    // (the device must write EVERY entry of the Jacobians: the product's buffers are not zeroed either -- unwritten entries come back as NaN)
    for (auto& v : dzt) v = TIO(std::numeric_limits<double>::quiet_NaN());
    for (auto& v : dut) v = TIO(std::numeric_limits<double>::quiet_NaN());
    dj::KernelArgs<TIO, T> A;
    { const char* rw = std::getenv("EMU_REFINE_W"); A.G = dj::make_globals<T>(M, opts, grad_mode, rw ? std::atof(rw) : INFINITY); }
    if constexpr (QUAD && (NW == 1 || NW == 2)) { if (W == 64 * NW) dj::set_row_passes(A.G, M); }     // (as the product's launch(): the factorization's level passes in the row layout)
    A.nodes = nodes.data(); A.contacts = contacts.data(); A.B = B;
    std::vector<dj::TraSD<T>> tsd;
    for (auto& a : M.tsd) { dj::TraSD<T> b; b.spring = T(a.spring); b.damper = T(a.damper); for (int i = 0; i < 3; ++i) b.off[i] = T(a.off[i]); b.lim_lo = T(a.lim_lo); b.lim_hi = T(a.lim_hi); b.nlim = a.nlim; tsd.push_back(b); }
    A.tsd = M.has_tsd ? tsd.data() : nullptr;
    std::vector<dj::MLimP<T>> mlim;          // joint limits on several coordinates (the -DDJ_MLIM=1 build of the emulator reads it)
    for (auto& a : M.mlim) { dj::MLimP<T> b; b.nt = a.nt; b.nr = a.nr; for (int i = 0; i < 6; ++i) { b.lo[i] = T(a.lo[i]); b.hi[i] = T(a.hi[i]); } mlim.push_back(b); }
    A.mlim = M.has_mlim ? mlim.data() : nullptr;
    std::vector<dj::NodeP<T>> cuts; for (auto& n : M.cuts) cuts.push_back(dj::cast_node<T>(n));      // loop-closing joints (-DDJ_CUT=1 builds)
    A.cuts = M.has_cut ? cuts.data() : nullptr; A.ncut = (int)cuts.size();
    std::vector<T> cutws(M.has_cut ? (size_t)B * dj::CUTWS : 0); A.cutws = M.has_cut ? cutws.data() : nullptr;
    std::vector<TIO> fet = castv(fext, (size_t)B * 6 * M.Nb); A.fext = fext ? fet.data() : nullptr;
    A.z = zt.data(); A.u = u ? ut.data() : nullptr; A.z_next = zn.data(); A.status = status; A.iters = iters;
    A.vel = vel ? velt.data() : nullptr; A.joint_imp = jimp ? jt.data() : nullptr; A.contact_sg = csg ? ct.data() : nullptr;
    A.dz = dz ? dzt.data() : nullptr; A.du = du ? dut.data() : nullptr;
    std::vector<TIO> rest(storage ? (size_t)B * 6 * M.Nb : 0); A.res = storage ? rest.data() : nullptr;
    std::vector<TIO> dct((dc && QUAD) ? (size_t)B * nx * 5 * std::max(M.Nc, 1) : 0, TIO(std::numeric_limits<double>::quiet_NaN())); A.dc = nullptr;
    std::vector<T> dbgt(dbg ? (size_t)B * M.Nb * 512 : 0); A.dbg = dbg ? dbgt.data() : nullptr;
    std::vector<T> solbuf(dz ? (size_t)B * M.S * dj::sol_record<MAXC>() : 0); A.sol = dz ? solbuf.data() : nullptr;
    int E = W / (M.S * (QUAD ? 4 : 1)), nwaves = (B + E - 1) / E;
    // (as the product's launch(): the explicit inverses of the Newton loop travel only when the refining IFT kernel will read them)
    std::vector<T> facbuf((dz && QUAD && A.G.refine_w < INFINITY) ? (size_t)nwaves * dj::FAC_PER_LANE * W : 0); A.fac = facbuf.empty() ? nullptr : facbuf.data();
    std::vector<T> lubuf((dz && QUAD) ? (size_t)nwaves * dj::LU_PER_LANE * W : 0); A.lu = lubuf.empty() ? nullptr : lubuf.data();
    std::vector<T> blkbuf(QUAD ? (size_t)nwaves * 90 * W : 0); A.blk = blkbuf.empty() ? nullptr : blkbuf.data();
    // the up-sweep's messages to the roots (as the product's launch())
    int ntops = 0; for (auto& n : M.nodes) if (n.level <= 1) ++ntops;       // one block per level-1 supernode (messages) and per root (its body rows' x)
    std::vector<T> msgbuf; A.msg = nullptr; A.msg_stride = (long long)ntops * std::max(2 * M.Nb + (M.nu + 5) / 6, M.Nc) * 36 + 8;
    if (dz && QUAD && A.msg_stride > 0) { msgbuf.resize((size_t)B * A.msg_stride); A.msg = msgbuf.data(); }
    std::vector<T> yparkbuf; A.ypark = nullptr;
    if (dz && QUAD && sizeof(TIO) < sizeof(T)) { A.ypark_stride = (long long)(std::max(2 * M.Nb + (M.nu + 5) / 6, M.Nc) * 18 * W); yparkbuf.resize((size_t)nwaves * A.ypark_stride); A.ypark = yparkbuf.data(); }
    // the same two launches as the product: step kernel, then (when gradients are wanted) the IFT kernel
    // the product's launches: step kernel, refining step kernel (quad mapping; re-solves what the first deferred), IFT kernel,
    // refining IFT kernel
    std::vector<int> flagbuf(B, 0); A.flag = (QUAD && A.G.refine_w < INFINITY) ? flagbuf.data() : nullptr;
    if (!A.flag) A.blk = nullptr;
    static_assert((size_t)dj::step_lds_bytes<TIO, T, MAXC, 0, QUAD, false, NW>() <= Shared::kLdsBytes && (size_t)dj::step_lds_bytes<TIO, T, MAXC, 1, QUAD, false, NW>() <= Shared::kLdsBytes
                  && (size_t)dj::step_lds_bytes<TIO, T, MAXC, 2, QUAD, false, NW>() <= Shared::kLdsBytes, "the emulated LDS block is too small for this layout");
    // iteration cap + continuation kernel (EMU_ITER_CAP=<cap>[:<replicas>]; as the product's launch(): single-wavefront quad mapping, no refinement)
    std::vector<T> resumebuf; std::vector<int> contlist, contcount(1, 0), cstat; int cont_replicas = 4;
    if constexpr (QUAD && NW == 1) {
        const char* ec = std::getenv("EMU_ITER_CAP");
        const int cap = ec ? std::atoi(ec) : 0;
        if (ec && std::strchr(ec, ':')) cont_replicas = std::atoi(std::strchr(ec, ':') + 1);
        if (cap > 0 && cap < opts.max_iter && !A.flag && !dbg) {
            A.G.iter_cap = cap;
            if (!A.sol) { solbuf.resize((size_t)B * M.S * dj::sol_record<MAXC>()); A.sol = solbuf.data(); }
            resumebuf.resize((size_t)B * dj::CARRY_PER_ENV); A.resume = resumebuf.data();
            contlist.resize(nwaves); A.cont_list = contlist.data(); A.cont_count = contcount.data();
            if (!A.status) { cstat.resize(B); A.status = cstat.data(); }
        }
    }
    for (int pass = 0; pass < 4; ++pass) {
        if ((pass == 1 || pass == 3) && !A.flag) continue;
        if (pass >= 2 && !(dz && !dbg)) continue;
        for (int wi = 0; wi < nwaves; ++wi) {
            Shared sh(W);
            std::vector<std::thread> th;
            for (int l = 0; l < W; ++l) th.emplace_back([&, l, pass]() {
                EmuWaveT<NW> w{&sh, l}; EmuWaveT<NW, true> wr{&sh, l};
                if (pass == 0) dj::step_entry<TIO, T, TL, MAXC, QUAD>(w, A, wi);
                else if (pass == 1) { if constexpr (QUAD) dj::step_entry<TIO, T, TL, MAXC, QUAD>(wr, A, wi); }
                else if (pass == 2) dj::grad_entry<TIO, T, TL, MAXC, QUAD>(w, A, wi);
                else { if constexpr (QUAD) dj::grad_entry<TIO, T, TL, MAXC, QUAD, EmuWaveT<NW, true>, 2>(wr, A, wi); }
            });
            for (auto& t : th) t.join();
        }
        if constexpr (QUAD && NW == 1) if (pass == 0 && A.G.iter_cap > 0) {
            // the continuation kernel: every listed workgroup once more, on `cont_replicas` copies of its threads
            auto cont = [&](auto rtag) {
                constexpr int R = decltype(rtag)::value;
                for (int ci = 0; ci < contcount[0]; ++ci) {
                    const int wi = contlist[ci];
                    std::vector<std::unique_ptr<Shared>> shs; for (int r = 0; r < R; ++r) shs.emplace_back(new Shared(W));
                    RepShared rs(W * R, R * (W / 4) * 4);
                    std::vector<std::thread> th;
                    for (int r = 0; r < R; ++r) for (int l = 0; l < W; ++l) th.emplace_back([&, r, l]() {
                        EmuWaveRep<R> w; w.sh = shs[r].get(); w.l = l; w.rs = &rs; w.rep = r;
                        dj::step_entry<TIO, T, TL, MAXC, QUAD, EmuWaveRep<R>, true>(w, A, wi);
                    });
                    for (auto& t : th) t.join();
                }
            };
            if (cont_replicas == 2) cont(std::integral_constant<int, 2>()); else if (cont_replicas == 3) cont(std::integral_constant<int, 3>()); else cont(std::integral_constant<int, 4>());
            for (int e = 0; e < B; ++e) if (A.status[e] == DJ_STATUS_CONTINUE) { std::fprintf(stderr, "emu: environment %d left unfinished by the continuation kernel\n", e); std::abort(); }
        }
        if constexpr (QUAD && NW == 1) if (pass == 2 && A.G.iter_cap > 0) {
            // ... and the IFT of the listed workgroups (dojo_gradc_kernel; pass 2 above has skipped them)
            for (int ci = 0; ci < contcount[0]; ++ci) {
                const int wi = contlist[ci];
                Shared sh(W);
                std::vector<std::thread> th;
                for (int l = 0; l < W; ++l) th.emplace_back([&, l]() { EmuWaveT<NW> w{&sh, l}; dj::grad_entry<TIO, T, TL, MAXC, QUAD, EmuWaveT<NW>, 0, true>(w, A, wi); });
                for (auto& t : th) t.join();
            }
        }
    }
    if constexpr (QUAD) if (dz && dc && !dbg && M.Nc > 0) {          // third launch: the contact-data columns (re-uses the hand-off)
        A.dc = dct.data(); A.dz = nullptr; A.du = nullptr;
        for (int wi = 0; wi < nwaves; ++wi) {
            Shared sh(W);
            std::vector<std::thread> th;
            for (int l = 0; l < W; ++l) th.emplace_back([&, l]() { EmuWaveT<NW> w{&sh, l}; dj::grad_entry<TIO, T, TL, MAXC, QUAD, EmuWaveT<NW>, 1>(w, A, wi); });
            for (auto& t : th) t.join();
        }
        for (size_t i = 0; i < (size_t)B * nx * 5 * M.Nc; ++i) dc[i] = dct[i];
    }
    for (size_t i = 0; i < zn.size(); ++i) z_next[i] = zn[i];
    if (storage && vel && csg)                                      // the Storage kernel's body, per (environment, body)
        for (int e = 0; e < B; ++e) for (int k = 0; k < M.Nb; ++k) {
            T zb[13], v[3], w[3], rb[6], row[25], fe[6];
            if (fext) for (int i = 0; i < 6; ++i) fe[i] = T(fet[(size_t)e * 6 * M.Nb + 6 * k + i]);
            for (int i = 0; i < 13; ++i) zb[i] = T(zt[(size_t)e * nz + 13 * k + i]);
            for (int i = 0; i < 3; ++i) { v[i] = T(velt[(size_t)e * 6 * M.Nb + 6 * k + i]); w[i] = T(velt[(size_t)e * 6 * M.Nb + 6 * k + 3 + i]); }
            for (int i = 0; i < 6; ++i) rb[i] = T(rest[(size_t)e * 6 * M.Nb + 6 * k + i]);
            auto other = [&](int b, T* zo, T* vo, T* wo) {
                for (int i = 0; i < 13; ++i) zo[i] = T(zt[(size_t)e * nz + 13 * b + i]);
                for (int i = 0; i < 3; ++i) { vo[i] = T(velt[(size_t)e * 6 * M.Nb + 6 * b + i]); wo[i] = T(velt[(size_t)e * 6 * M.Nb + 6 * b + 3 + i]); }
            };
            dj::storage_row(row, nodes[k], contacts.data(), T(M.dt), zb, v, w, ct.data() + (size_t)e * (2 * dj::NCV) * M.Nc, rb, fext ? fe : (const T*)nullptr, nodes.data(), other, M.contact_model, 2 * dj::NCV, k, M.Nc);
            for (int i = 0; i < 25; ++i) storage[((size_t)e * M.Nb + k) * 25 + i] = (double)TIO(row[i]);
        }
    for (size_t i = 0; i < velt.size(); ++i) vel[i] = velt[i];
    for (size_t i = 0; i < (jimp ? (size_t)B * M.n_joint_imp : 0); ++i) jimp[i] = jt[i];
    for (size_t i = 0; i < (csg ? (size_t)B * (2 * dj::NCV) * M.Nc : 0); ++i) csg[i] = ct[i];
    for (size_t i = 0; i < dzt.size(); ++i) dz[i] = dzt[i];
    for (size_t i = 0; i < dbgt.size(); ++i) dbg[i] = dbgt[i];
    for (size_t i = 0; i < (du ? (size_t)B * nx * M.nu : 0); ++i) du[i] = dut[i];
}

} // namespace

extern "C" int emu_step(const DojoTopology* tp, const DojoSolverOptions* opts, int grad_mode, int dtype, int quad, int B, int envs_per_wave,
                        const double* z, const double* u, double* z_next, int* status, int* iters,
                        double* vel, double* jimp, double* csg, double* dz, double* du, double* dbg, char* err, int errlen, double* dc, double* storage, const double* fext) {
    dj::HostModel M;
    int rc = dj::build_host_model(*tp, M);
    if (rc != DOJO_OK) { if (err) std::strncpy(err, M.error.c_str(), errlen - 1); return rc; }
    if (quad && M.S > 32) { if (err) std::strncpy(err, "quad mapping needs <= 32 bodies", errlen - 1); return DOJO_ERR_UNSUPPORTED; }
    if (M.has_ss && M.contact_model != 2 && DJ_MLIM && (!quad || M.S > 16 || M.maxc > 1 || M.has_tsd || M.has_mlim || M.has_cut)) {      // (as dojo_create: the contact travels as a cut element)
        int prc = dj::promote_tree_edge_contacts(M);
        if (prc != DOJO_OK) { if (err) std::strncpy(err, M.error.c_str(), errlen - 1); return prc; }
    }
    if (M.has_ss && (!quad || M.S > 16 || M.maxc > 1 || M.has_tsd || dz != nullptr)) {       // (as dojo_create / launch)
        if (err) std::strncpy(err, "a body-body contact needs the single-wavefront quad mapping, <= 1 contact per body, and has no gradients", errlen - 1); return DOJO_ERR_UNSUPPORTED; }
    if (M.has_cc && dz != nullptr) { if (err) std::strncpy(err, "a body-body contact has no gradients", errlen - 1); return DOJO_ERR_UNSUPPORTED; }
    if ((M.has_mlim || M.has_cut) && (M.has_ss || M.contact_model == 2)) { if (err) std::strncpy(err, "cut elements / several limits next to a tree-edge body-body contact or LinearContact: not built", errlen - 1); return DOJO_ERR_UNSUPPORTED; }
    if ((M.has_mlim || M.has_cut) && (quad || !DJ_MLIM)) {      // (as the product: the general builds of the lane mapping)
        if (err) std::strncpy(err, "joint limits on several coordinates / both halves and kinematic loops need the lane mapping of a -DDJ_MLIM=1 -DDJ_CUT=1 build", errlen - 1); return DOJO_ERR_UNSUPPORTED; }
    const bool two = quad && M.S > 16;          // one environment over two wavefronts: the NW = 2 layout / reduction paths
    if (two && (M.maxc > 4 || M.Nc > 16 || M.has_tsd || M.contact_model == 2)) {   // (as the product's mapping_waves(): such mechanisms take the lane mapping)
        if (err) std::strncpy(err, "the two-wavefront quad mapping serves <= 4 contacts per body, <= 16 contacts, no translational springs / dampers, no LinearContact: use the lane mapping", errlen - 1); return DOJO_ERR_UNSUPPORTED; }
    int W = M.S * (quad ? 4 : 1) * (envs_per_wave > 0 ? envs_per_wave : 1);
    DojoSolverOptions o = opts ? *opts : dj::default_options();
#define RUN(TIO, TS, TL, MC) do { if (two) run<TIO, TS, TL, (MC < 4 ? 4 : MC), true, 2>(M, o, grad_mode, B, W, z, u, z_next, status, iters, vel, jimp, csg, dz, du, dbg, dc, storage, fext); \
                                  else if (quad) run<TIO, TS, TL, MC, true>(M, o, grad_mode, B, W, z, u, z_next, status, iters, vel, jimp, csg, dz, du, dbg, dc, storage, fext); \
                                  else      run<TIO, TS, TL, MC, false>(M, o, grad_mode, B, W, z, u, z_next, status, iters, vel, jimp, csg, dz, du, dbg, dc, storage, fext); } while (0)
    // dtype 0: fp64 everywhere; dtype 1: fp32 I/O with fp64 internals (the product's "f32" mode); dtype 3: fp32 factorization (experiments)
    if (dtype == DOJO_DTYPE_F64) { if (M.maxc <= 1) RUN(double, double, double, 1); else if (M.maxc <= 4) RUN(double, double, double, 4); else RUN(double, double, double, 8); }
    else if (dtype == DOJO_DTYPE_F32) { if (M.maxc <= 1) RUN(float, double, double, 1); else if (M.maxc <= 4) RUN(float, double, double, 4); else RUN(float, double, double, 8); }
    else { if (M.maxc <= 1) RUN(float, double, float, 1); else RUN(float, double, float, 4); }
    return DOJO_OK;
}
