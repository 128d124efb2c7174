// Micro-benchmark, round 6 (review item 1): ONE level pass of the supernodal tree factorization -- Gauss-Jordan inverse of the 12 x 12
// supernode block, Tq = S^-1 U, up = Dup - L Tq -- for the 4 supernodes of a wavefront that are at the level, in two layouts:
//   A  the shipped quad layout (dojo_device.hpp factorize_quad): 4 lanes per supernode, 3 rows per lane, pivot rows broadcast with
//      v_mov_b32_dpp quad_perm pairs; all 16 supernode slots of the wavefront execute the pass, 4 of them usefully
//   R  the row layout: the 4 supernodes' rows cross to 16 lanes each through LDS (1 row per lane, 12 of 16 lanes busy), the pivot row
//      arrives INSIDE the multiply-add (v_fmac_f64_dpp row_newbcast:p -- the one DPP control gfx950's DP ALU has), inverse rows go back
//      through LDS to the quad lanes, `up` is left where the parent's pass reads it
// Both produce the same numbers in the same order of operations; the host compares them bit for bit.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/gj_rows.hip -o /tmp/gj_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define N_IT 100
__device__ __forceinline__ double rcpd(double a) { double r = __builtin_amdgcn_rcp(a); double e = fma(-a, r, 1.0); r = fma(r, e, r); e = fma(-a, r, 1.0); return fma(r, e, r); }
template <int CTRL> __device__ __forceinline__ double dppd(double v) {
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ double qb(double v, int o) { switch (o) { case 0: return dppd<0x00>(v); case 1: return dppd<0x55>(v); case 2: return dppd<0xAA>(v); default: return dppd<0xFF>(v); } }
__device__ __forceinline__ double qx(double v, int m) { return m == 1 ? dppd<0xB1>(v) : dppd<0x4E>(v); }

// row layout primitives: value of lane P of my 16-lane row, alone (v_mov_b64_dpp) and fused into the multiply-add
template <int P> __device__ __forceinline__ double rb_(double v) {
    long long x = __builtin_bit_cast(long long, v);
    const long long y = __builtin_amdgcn_update_dpp(x, x, 0x150 + P, 0xF, 0xF, false);
    return __builtin_bit_cast(double, y);
}
__device__ __forceinline__ double rb(double v, int p) {
    switch (p) { case 0: return rb_<0>(v); case 1: return rb_<1>(v); case 2: return rb_<2>(v); case 3: return rb_<3>(v); case 4: return rb_<4>(v); case 5: return rb_<5>(v);
                 case 6: return rb_<6>(v); case 7: return rb_<7>(v); case 8: return rb_<8>(v); case 9: return rb_<9>(v); case 10: return rb_<10>(v); default: return rb_<11>(v); }
}
// acc += (src of lane P) * f
template <int P> __device__ __forceinline__ void fm_(double& acc, double src, double f) {
    __asm__("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(f), "n"(P));
}
__device__ __forceinline__ void fm(double& acc, double src, double f, int p) {
    switch (p) { case 0: fm_<0>(acc, src, f); break; case 1: fm_<1>(acc, src, f); break; case 2: fm_<2>(acc, src, f); break; case 3: fm_<3>(acc, src, f); break;
                 case 4: fm_<4>(acc, src, f); break; case 5: fm_<5>(acc, src, f); break; case 6: fm_<6>(acc, src, f); break; case 7: fm_<7>(acc, src, f); break;
                 case 8: fm_<8>(acc, src, f); break; case 9: fm_<9>(acc, src, f); break; case 10: fm_<10>(acc, src, f); break; default: fm_<11>(acc, src, f); break; }
}

__device__ __forceinline__ double elemS(int sn, int r, int c, double seed) { return seed * (0.013 * ((r * 7 + c * 3 + sn) % 11) - 0.05) + (r == c ? 4.0 + 0.1 * sn : 0.0); }
__device__ __forceinline__ double elemU(int sn, int r, int j, double seed) { return r < 3 ? 0.0 : seed * (0.02 * ((r * 5 + j + sn) % 7) - 0.06); }
__device__ __forceinline__ double elemL(int sn, int i, int c, double seed) { return seed * (0.017 * ((i * 3 + c * 5 + sn) % 9) - 0.07); }
__device__ __forceinline__ double elemD(int sn, int i, int j, double seed) { return seed * (0.011 * ((i + j * 2 + sn) % 5)) + (i == j ? 2.0 : 0.0); }

constexpr int RS = 13;                    // row stride of the staged S rows (doubles): odd -> conflict-free ds_read_b64 over 32 lanes
constexpr int US = 7;                     // ... of the staged U / Tq rows
// MODE 0: quad layout; 1: row layout; 2: row layout, Gauss-Jordan alone on resident rows (the review's kill criterion: <= 2.6 k cycles)
template <int MODE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) k(double* out, unsigned long long* cyc, double seed, int lev_slot0, unsigned long long* ph) {
    __shared__ double lds[4 * 16 * RS + 4 * 16 * US + 4 * 6 * RS + 4 * 6 * US + 16 * 40];
    const int lane = threadIdx.x, wave = blockIdx.x;
    const int s = lane >> 2, q = lane & 3;                  // quad layout: supernode slot, role
    const int g = lane >> 4, r = lane & 15;                 // row layout: group (supernode of the level), row
    // the four supernodes at the level: slots lev_slot0 + 3 g'  (any four; their quads hold the data)
    const int gq = (s - lev_slot0) / 3;
    const bool at = (s >= lev_slot0) && ((s - lev_slot0) % 3 == 0) && gq < 4;
    const int sn = wave * 4 + gq;                           // global supernode id (data seed)
    double A[3][12], Uq[3][6], Lq[6][3], D[3][6], up[3][6];
    for (int i = 0; i < 3; ++i) {
        for (int c = 0; c < 12; ++c) A[i][c] = at ? elemS(sn, 3 * q + i, c, seed) : (3 * q + i == c ? 1.0 : 0.0);
        for (int j = 0; j < 6; ++j) { Uq[i][j] = at ? elemU(sn, 3 * q + i, j, seed) : 0.0; D[i][j] = (at && q < 2) ? elemD(sn, 3 * q + i, j, seed) : 0.0; up[i][j] = 0.0; }
    }
    for (int i = 0; i < 6; ++i) for (int c = 0; c < 3; ++c) Lq[i][c] = at ? elemL(sn, i, 3 * q + c, seed) : 0.0;
    double* stS = lds; double* stU = stS + 4 * 16 * RS; double* stL = stU + 4 * 16 * US; double* stD = stL + 4 * 6 * RS; double* mail = stD + 4 * 6 * US;
    double Rres[12];
    for (int c = 0; c < 12; ++c) Rres[c] = r < 12 ? elemS(wave * 4 + g, r, c, seed) : (r == c ? 1.0 : 0.0);
    unsigned long long pa[5] = {0, 0, 0, 0, 0}, tp = 0;
#define PH_B() do { if (ph) { __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tp = __builtin_readcyclecounter(); } } while (0)
#define PH_E(i) do { if (ph) { __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); unsigned long long t_ = __builtin_readcyclecounter(); pa[i] += t_ - tp; tp = t_; } } while (0)
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N_IT; ++it) {
        double A0[3][12];
        double eps = 0.0;
        __asm__ volatile("" : "+v"(eps));                  // opaque zero: every repetition recomputes the same pass
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int c = 0; c < 12; ++c) A0[i][c] = A[i][c] + eps;
        if (MODE == 0) {
            double ipown[3] = {1.0, 1.0, 1.0};
#pragma unroll
            for (int pp = 0; pp < 12; ++pp) {
                const int p = pp < 3 ? pp : pp < 6 ? pp + 3 : pp < 9 ? pp - 3 : pp;
                const int o = p / 3, ro = p % 3;
                const bool own = (q == o);
                double prow[12];
#pragma unroll
                for (int c = 0; c < 12; ++c) prow[c] = qb(A0[ro][c], o);
                const double ip = rcpd(at ? prow[p] : 1.0);
                if (own) ipown[ro] = ip;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    const double f = A0[rr][p];
                    const double g_ = f * ip;
                    const double ge = at ? g_ : 0.0;
                    const double fe = (rr == ro) ? (own ? 0.0 : ge) : ge;
#pragma unroll
                    for (int c = 0; c < 12; ++c) if (c != p) A0[rr][c] -= fe * prow[c];
                    const double colp = at ? -g_ : f;
                    A0[rr][p] = (rr == ro) ? (own ? (at ? 1.0 : f) : colp) : colp;
                }
            }
            if (at) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int c = 0; c < 12; ++c) A0[i][c] *= ipown[i];
            }
            double Tq[3][6];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) Tq[i][j] = 0.0;
#pragma unroll
            for (int o = 1; o < 4; ++o)
#pragma unroll
                for (int m_ = 0; m_ < 3; ++m_)
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const double u_ = qb(Uq[m_][j], o);
#pragma unroll
                        for (int i = 0; i < 3; ++i) Tq[i][j] += A0[i][3 * o + m_] * u_;
                    }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double part[18];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) part[6 * i + j] = Lq[3 * h + i][0] * Tq[0][j] + Lq[3 * h + i][1] * Tq[1][j] + Lq[3 * h + i][2] * Tq[2][j];
#pragma unroll
                for (int i = 0; i < 18; ++i) part[i] += qx(part[i], 1);
#pragma unroll
                for (int i = 0; i < 18; ++i) part[i] += qx(part[i], 2);
                if (at) {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            if (h == 0) up[i][j] = (q == 0) ? D[i][j] - part[6 * i + j] : 0.0;
                            else up[i][j] = (q == 1) ? D[i][j] - part[6 * i + j] : up[i][j];
                        }
                }
            }
            // (the shipped pass posts `up` to the mailbox for the parent: 18 values on roles 0 / 1)
            if (q < 2) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) mail[(2 * s + q) * 20 + 6 * i + j] = up[i][j];
            }
        } else {
            double R[12], Ur[6], Lr[12], Dr[6];
            PH_B();
            if (MODE == 1) {
                // ---- quad lanes at the level -> LDS
                if (at) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
#pragma unroll
                        for (int c = 0; c < 12; ++c) stS[(gq * 16 + 3 * q + i) * RS + c] = A0[i][c];
#pragma unroll
                        for (int j = 0; j < 6; ++j) stU[(gq * 16 + 3 * q + i) * US + j] = Uq[i][j];
                    }
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int c = 0; c < 3; ++c) stL[(gq * 6 + i) * RS + 3 * q + c] = Lq[i][c];
                    if (q < 2) {
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < 6; ++j) stD[(gq * 6 + 3 * q + i) * US + j] = D[i][j];
                    }
                }
                __asm__ volatile("" ::: "memory");
                // ---- row lanes <- LDS
#pragma unroll
                for (int c = 0; c < 12; ++c) { R[c] = (r == c ? 1.0 : 0.0); Lr[c] = 0.0; }
#pragma unroll
                for (int j = 0; j < 6; ++j) { Ur[j] = 0.0; Dr[j] = 0.0; }
                if (r < 12) {
#pragma unroll
                    for (int c = 0; c < 12; ++c) R[c] = stS[(g * 16 + r) * RS + c];
#pragma unroll
                    for (int j = 0; j < 6; ++j) Ur[j] = stU[(g * 16 + r) * US + j];
                }
                if (r < 6) {
#pragma unroll
                    for (int c = 0; c < 12; ++c) Lr[c] = stL[(g * 6 + r) * RS + c];
#pragma unroll
                    for (int j = 0; j < 6; ++j) Dr[j] = stD[(g * 6 + r) * US + j];
                }
                __asm__ volatile("" ::: "memory");
            } else {
#pragma unroll
                for (int c = 0; c < 12; ++c) { R[c] = Rres[c] + eps; Lr[c] = 0.0; }
#pragma unroll
                for (int j = 0; j < 6; ++j) { Ur[j] = 0.0; Dr[j] = 0.0; }
            }
            PH_E(0);
            // ---- Gauss-Jordan, the pivot row inside the multiply-add
            double ipown = 1.0;
#pragma unroll
            for (int pp = 0; pp < 12; ++pp) {
                const int p = pp < 3 ? pp : pp < 6 ? pp + 3 : pp < 9 ? pp - 3 : pp;
                const double pe = rb(R[p], p);
                const double ip = rcpd(pe);
                const bool own = (r == p);
                const double g_ = R[p] * ip;
                const double nfe = own ? 0.0 : -g_;
                ipown = own ? ip : ipown;
                // (the column the NEXT pivot broadcasts first: its v_mov_b64_dpp must not follow the write within 2 wait states -- the
                //  compiler's hazard recognizer does not see the VALU write inside the asm statement)
                const int pn = pp + 1 < 12 ? (pp + 1 < 3 ? pp + 1 : pp + 1 < 6 ? pp + 4 : pp + 1 < 9 ? pp - 2 : pp + 1) : -1;
                if (pn >= 0) fm(R[pn], R[pn], nfe, p);
#pragma unroll
                for (int c = 0; c < 12; ++c) if (c != p && c != pn) fm(R[c], R[c], nfe, p);
                R[p] = own ? 1.0 : -g_;
            }
#pragma unroll
            for (int c = 0; c < 12; ++c) R[c] *= ipown;
            PH_E(1);
            if (MODE == 1) {
                // ---- inverse rows back to the quad lanes
                if (r < 12) {
#pragma unroll
                    for (int c = 0; c < 12; ++c) stS[(g * 16 + r) * RS + c] = R[c];
                }
                PH_E(2);
                // ---- Tq = S^-1 U (rows 0:3 of U are structurally zero), up = Dup - L Tq in the quad layout's summation order
                double Tq[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int m = 3; m < 12; ++m)
#pragma unroll
                    for (int j = 0; j < 6; ++j) fm(Tq[j], Ur[j], R[m], m);
                __asm__ volatile("s_nop 1" ::: "memory");
                double upr[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    double pq[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int o = 0; o < 4; ++o)
#pragma unroll
                        for (int m_ = 0; m_ < 3; ++m_) fm(pq[o], Tq[j], Lr[3 * o + m_], 3 * o + m_);
                    upr[j] = Dr[j] - ((pq[0] + pq[1]) + (pq[2] + pq[3]));
                }
                // parent-side mailbox: rows 0:3 -> role 0 slot, rows 3:6 -> role 1 slot of this supernode
                const int sl = lev_slot0 + 3 * g;
                if (r < 6) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) mail[(2 * sl + r / 3) * 20 + 6 * (r % 3) + j] = upr[j];
                }
                PH_E(3);
                if (at) {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int c = 0; c < 12; ++c) A0[i][c] = stS[(gq * 16 + 3 * q + i) * RS + c];
                }
                PH_E(4);
            } else {
#pragma unroll
                for (int c = 0; c < 12; ++c) A0[0][c] = R[c];
            }
        }
        if (it == N_IT - 1) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int c = 0; c < 12; ++c) A[i][c] = A0[i][c];
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // results: inverse rows (quad layout: modes 0 / 1) and the Schur complement from the mailbox
    double* o = out + (size_t)wave * (4 * (144 + 36));
    if (MODE != 2) {
        if (at) {
            for (int i = 0; i < 3; ++i) for (int c = 0; c < 12; ++c) o[gq * 180 + (3 * q + i) * 12 + c] = A[i][c];
            if (q < 2) for (int i = 0; i < 3; ++i) for (int j = 0; j < 6; ++j) o[gq * 180 + 144 + (3 * q + i) * 6 + j] = mail[(2 * s + q) * 20 + 6 * i + j];
        }
    } else if (r < 12) {
        for (int c = 0; c < 12; ++c) o[g * 180 + r * 12 + c] = A[0][c];
    }
    if (lane == 0) cyc[wave] = t1 - t0;
    if (ph && lane == 0) for (int i = 0; i < 5; ++i) ph[wave * 5 + i] = pa[i];
}
template <int MODE> std::vector<double> run(const char* name, int waves, bool phases = false) {
    double* out; unsigned long long* cyc;
    const size_t n = (size_t)waves * 4 * 180;
    hipMalloc(&out, n * 8); hipMalloc(&cyc, waves * 8); hipMemset(out, 0, n * 8);
    unsigned long long* ph = nullptr; if (phases) hipMalloc(&ph, waves * 5 * 8);
    k<MODE><<<waves, 64>>>(out, cyc, 1.0, 2, ph); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<MODE><<<waves, 64>>>(out, cyc, 1.0, 2, ph); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(waves); hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= waves;
    printf("%-78s waves %5d  %.3f ms   cycles per level pass %.0f\n", name, waves, ms, avg / N_IT);
    if (phases) { std::vector<unsigned long long> hp(waves * 5); hipMemcpy(hp.data(), ph, waves * 5 * 8, hipMemcpyDeviceToHost); double a5[5] = {0, 0, 0, 0, 0}; for (int w = 0; w < waves; ++w) for (int i = 0; i < 5; ++i) a5[i] += hp[w * 5 + i]; printf("    phases (cycles per pass, a full LDS wait at every boundary): stage in + row loads %.0f | Gauss-Jordan %.0f | inverse rows out %.0f | S^-1 U, Schur, post %.0f | read-back %.0f\n", a5[0] / waves / N_IT, a5[1] / waves / N_IT, a5[2] / waves / N_IT, a5[3] / waves / N_IT, a5[4] / waves / N_IT); }
    std::vector<double> r(n); hipMemcpy(r.data(), out, n * 8, hipMemcpyDeviceToHost);
    hipFree(out); hipFree(cyc);
    return r;
}
int main() {
    auto a = run<0>("A quad layout: GJ 12x12 + S^-1 U + Schur, quad_perm broadcasts (16 slots, 4 at the level)", 1024);
    auto b = run<1>("R row layout: LDS transposition + fused row_newbcast GJ + S^-1 U + Schur + back", 1024);
    run<1>("R row layout once more, with a timer and a full LDS wait at every phase boundary", 1024, true);
    auto c = run<2>("R' row layout: the 12-pivot Gauss-Jordan alone on resident rows (kill criterion 2.6 k)", 1024);
    size_t nd = 0, ndc = 0; double worst = 0, sum = 0;
    for (size_t i = 0; i < a.size(); ++i) { if (std::memcmp(&a[i], &b[i], 8) != 0) { ++nd; double d = a[i] - b[i]; if (d < 0) d = -d; if (d > worst) worst = d; } sum += a[i]; }
    for (size_t w = 0; w < a.size() / 180; ++w) for (int i = 0; i < 144; ++i) if (std::memcmp(&a[w * 180 + i], &c[w * 180 + i], 8) != 0) ++ndc;
    printf("A vs R: %zu of %zu values differ bitwise (max |diff| %.3e); A vs R' inverse rows: %zu differ; checksum %.17g\n", nd, a.size(), worst, ndc, sum);
    // sanity: S * S^-1 = I for supernode 0 (host recomputation of the input)
    double err = 0;
    for (int r = 0; r < 12; ++r) for (int c = 0; c < 12; ++c) {
        double s = 0;
        for (int m = 0; m < 12; ++m) { double e = 1.0 * (0.013 * ((r * 7 + m * 3 + 0) % 11) - 0.05) + (r == m ? 4.0 : 0.0); s += e * a[m * 12 + c]; }
        double d = s - (r == c ? 1.0 : 0.0); if (d < 0) d = -d; if (d > err) err = d;
    }
    printf("|S S^-1 - I|_max (supernode 0, layout A) = %.3e\n", err);
    return 0;
}
