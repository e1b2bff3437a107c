// Slow path behind the wavefront kernel: problems whose working set outgrows the 64 lanes of a wavefront.
//
// The wavefront kernel keeps working-set position i on lane i; a parameter point that wants a 65th row ends there
// with EXIT_WSCAP and is appended to an overflow list.  This kernel re-solves the listed points from scratch, ONE
// PROBLEM PER THREAD, every array of the solver in a per-thread slice of global memory: no lane mapping, no
// capacity other than the scratch the host sized (kBigCap rows).  It is the dual active-set iteration of
// wave_kernel / lane_kernel written as plain serial loops -- the same fma chains in the same order (Gram-matrix
// lookups for the products of rows, as in the other kernels), so its results are the ones the CPU oracle gives,
// bit for bit.  It is slow (scalar loops, uncoalesced scratch) and meant to be: on the reference's benchmark
// class no sampled point needs it (largest working set 42 rows at N = 100); it exists so that a point with soft
// output bounds over a long horizon gets its answer instead of a flag.
//
// Follows the same reference items as the wavefront kernel: daqp_ldp [EXT] as called at src/utils.jl:282 /
// codegen/mpc_update_qp.c:48, bounds update codegen/mpc_update_qp.c:1-10, solution recovery :14-22, warm start
// :44-47, soft rows (rho_soft, src/setup.jl:26).  No branch and bound here.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lmpc_pack.hpp"
#include "lmpc_wave_layout.hpp"

namespace lmpc {

constexpr int kBigCap = 256;          // working-set rows the slow path has scratch for
constexpr int kBigThreads = 512;      // problems in flight (8 wavefronts of one thread per problem)

__host__ __device__ constexpr long long big_tri(long long i) { return i * (i + 1) / 2; }
// reals / ints of scratch per thread
__host__ __device__ inline long long big_scratch_reals(int n, int m, int cap) {
    return big_tri(cap + 1) + 7ll * (cap + 1) + n + 2ll * m;
}
__host__ __device__ inline long long big_scratch_ints(int m, int cap) { return (cap + 1) + (long long)m; }

template <typename R> __device__ __forceinline__ R big_fma(R a, R b, R c);
template <> __device__ __forceinline__ double big_fma<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float big_fma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename R>
__global__ __launch_bounds__(64) void big_kernel(
    const WaveLayout P, const R *__restrict__ C, const int32_t *__restrict__ S, const R *__restrict__ theta,
    R *__restrict__ X, int32_t *__restrict__ exitflag, int32_t *__restrict__ iters, uint64_t *__restrict__ active,
    const uint64_t *__restrict__ warm, const int32_t *__restrict__ ovf_list, const int32_t *__restrict__ ovf_count,
    R *__restrict__ scratch_r, int32_t *__restrict__ scratch_i, int cap, const WaveSim sim) {
    const int n = P.n, m = P.m, nth = P.nth;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (long long)gridDim.x * blockDim.x;
    const long long total = *ovf_count;
    if (tid >= total) return;
    const R primal_tol = (R)P.primal_tol, dual_tol = (R)P.dual_tol, zero_tol = (R)P.zero_tol,
            progress_tol = (R)P.progress_tol, fval_bound = (R)P.fval_bound, rho_soft = (R)P.rho_soft;
    // per-thread slices
    R *sr = scratch_r + tid * big_scratch_reals(n, m, cap);
    int32_t *si = scratch_i + tid * big_scratch_ints(m, cap);
    R *L = sr; sr += big_tri(cap + 1);
    R *D = sr; sr += cap + 1;
    R *Dinv = sr; sr += cap + 1;
    R *lam = sr; sr += cap + 1;
    R *ls = sr; sr += cap + 1;          // lambda*
    R *xl = sr; sr += cap + 1;
    R *zl = sr; sr += cap + 1;
    R *wv = sr; sr += cap + 1;
    R *u = sr; sr += n;
    R *dup = sr; sr += m;
    R *dlo = sr;
    int32_t *WS = si;
    int32_t *sense = si + (cap + 1);
    const R *Mr = C + P.oM, *G = C + P.oG;
    auto Gat = [&](int a, int c) -> R {
        return a >= c ? G[(size_t)a * (a + 1) / 2 + c] : G[(size_t)c * (c + 1) / 2 + a];
    };

    for (long long q = tid; q < total; q += nthreads) {
        const long long pid = ovf_list[q];
        const R *th = theta + pid * nth;
        // bounds of this parameter point   (mpc_update_qp.c:1-10)
        for (int j = 0; j < m; j++) {
            R sh = (R)0;
            for (int t = 0; t < nth; t++) sh = big_fma<R>(C[P.oDth + (size_t)j * nth + t], th[t], sh);
            dup[j] = C[P.odu + j] + sh;
            dlo[j] = C[P.odl + j] + sh;
        }
        int na = 0, sing = -1, reuse = 0, nsoft_act = 0;
        int flag = EXIT_ITERLIMIT, iter = 1, cycle = 0;
        R fval = (R)0, soft_slack = (R)0, best_fval = (R)-1;
        bool done = false;
        for (int k = 0; k < n; k++) u[k] = (R)0;

        // append row j: new row of L, new pivot
        auto ldl_add = [&](int j) {
            R *row = L + big_tri(na);
            for (int i = 0; i < na; i++) row[i] = Gat(WS[i], j);
            R dnew = Gat(j, j);
            if (sense[j] & SENSE_SOFT) dnew += rho_soft;
            for (int i = 0; i < na; i++) {
                R acc = row[i];
                const R *li = L + big_tri(i);
                for (int t = 0; t < i; t++) acc = big_fma<R>(-li[t], row[t], acc);
                row[i] = acc;
            }
            for (int i = 0; i < na; i++) {
                const R qv = row[i];
                const R l = qv * Dinv[i];
                row[i] = l;
                dnew = big_fma<R>(-l, qv, dnew);
            }
            row[na] = (R)1;
            const bool is_soft = (sense[j] & SENSE_SOFT) != 0;
            if (dnew < zero_tol || (!is_soft && na - nsoft_act >= n)) {
                D[na] = (R)0; Dinv[na] = (R)0; sing = na;
            } else {
                D[na] = dnew; Dinv[na] = (R)1 / dnew;
            }
            WS[na] = j; lam[na] = (R)0; ls[na] = (R)0;
            sense[j] |= SENSE_ACTIVE;
            nsoft_act += is_soft ? 1 : 0;
            na++;
        };
        // drop position r: compact L, rank-one update of the trailing block
        auto ldl_remove = [&](int r) {
            const int nup = na - r - 1;
            R alpha = D[r];
            for (int t = 0; t < nup; t++) wv[t] = L[big_tri(r + 1 + t) + r];
            for (int i = r; i < na - 1; i++) {
                R *dst = L + big_tri(i);
                const R *src = L + big_tri(i + 1);
                for (int c = 0; c < r; c++) dst[c] = src[c];
                for (int c = r; c < i; c++) dst[c] = src[c + 1];
                dst[i] = (R)1;
            }
            sing = -1;
            for (int t = 0; t < nup; t++) {
                const int i = r + t;
                const R pt = wv[t];
                const R dold = D[i + 1];
                const R dbar = big_fma<R>(alpha * pt, pt, dold);
                if (dbar < zero_tol) {
                    D[i] = (R)0; Dinv[i] = (R)0; sing = i;
                    for (int qq = i + 1; qq < na - 1; qq++) { D[qq] = D[qq + 1]; Dinv[qq] = Dinv[qq + 1]; }
                    break;
                }
                const R rinv = (R)1 / dbar;
                const R beta = (pt * alpha) * rinv;
                alpha = (dold * alpha) * rinv;
                D[i] = dbar; Dinv[i] = rinv;
                for (int qq = t + 1; qq < nup; qq++) {
                    R *lqi = L + big_tri(r + qq) + i;
                    wv[qq] = big_fma<R>(-pt, *lqi, wv[qq]);
                    *lqi = big_fma<R>(beta, wv[qq], *lqi);
                }
            }
            if (sense[WS[r]] & SENSE_SOFT) nsoft_act--;
            sense[WS[r]] &= ~(SENSE_ACTIVE | SENSE_LOWER);
            for (int i = r; i < na - 1; i++) { WS[i] = WS[i + 1]; lam[i] = lam[i + 1]; }
            na--;
            if (r < reuse) reuse = r;
        };

        // ---- initial working set: rows flagged ACTIVE and the caller's warm-start mask, in row order
        for (int j = 0; j < m; j++) sense[j] = S[j] & ~SENSE_LOWER;
        const uint64_t *wp = warm ? warm + pid * P.words : nullptr;
        for (int j = 0; j < m && !done; j++) {
            const int s0 = S[j];
            bool want = (s0 & SENSE_ACTIVE) != 0, lower = want && (s0 & SENSE_LOWER);
            if (wp && !(s0 & SENSE_IMMUTABLE)) {
                if ((wp[j >> 6] >> (j & 63)) & 1ull) want = true;
                else if ((wp[(m + j) >> 6] >> ((m + j) & 63)) & 1ull) { want = true; lower = true; }
            }
            if (!want) { sense[j] &= ~SENSE_ACTIVE; continue; }
            if (lower) sense[j] |= SENSE_LOWER;
            if (na >= cap) { flag = EXIT_WSCAP; done = true; break; }
            ldl_add(j);
            if (sing >= 0) {
                if (s0 & SENSE_IMMUTABLE) { flag = EXIT_OVERDETERMINED; done = true; break; }
                na--; sing = -1;                                   // dependent warm-start row: drop it again
                if (sense[j] & SENSE_SOFT) nsoft_act--;
                sense[j] &= ~(SENSE_ACTIVE | SENSE_LOWER);
            }
        }

        // ---- dual active-set iterations
        for (; !done && iter < P.iter_limit; iter++) {
            if (sing < 0) {
                int nblock = 0, rm = -1, add = -1;
                bool isupper = false;
                R alpha = (R)0;
                // constrained stationary point (L D L') lam* = -d_W
                for (int i = reuse; i < na; i++) {
                    const int j = WS[i];
                    R acc = (sense[j] & SENSE_LOWER) ? -dlo[j] : -dup[j];
                    const R *li = L + big_tri(i);
                    for (int t = 0; t < i; t++) acc = big_fma<R>(-li[t], xl[t], acc);
                    xl[i] = acc;
                }
                for (int i = reuse; i < na; i++) zl[i] = xl[i] * Dinv[i];
                for (int i = na - 1; i >= 0; i--) {
                    R acc = zl[i];
                    for (int t = na - 1; t > i; t--) acc = big_fma<R>(-L[big_tri(t) + i], ls[t], acc);
                    ls[i] = acc;
                }
                reuse = na;
                for (int i = 0; i < na; i++) {
                    const int j = WS[i];
                    if (sense[j] & SENSE_IMMUTABLE) continue;
                    if (sense[j] & SENSE_LOWER) { if (ls[i] < dual_tol) continue; }
                    else if (ls[i] > -dual_tol) continue;
                    const R cand = -lam[i] / (ls[i] - lam[i]);
                    if (nblock == 0 || cand < alpha) { alpha = cand; rm = i; }
                    nblock++;
                }
                if (nblock == 0) {
                    // primal iterate and objective
                    R soft = (R)0;
                    for (int k = 0; k < n; k++) u[k] = (R)0;
                    for (int i = 0; i < na; i++) {
                        const int j = WS[i];
                        const R *mi = Mr + (size_t)j * n;
                        const R l = ls[i];
                        for (int k = 0; k < n; k++) u[k] = big_fma<R>(-mi[k], l, u[k]);
                        if (sense[j] & SENSE_SOFT) soft = big_fma<R>(l * l, rho_soft, soft);
                    }
                    R fv = (R)0;
                    for (int k = 0; k < n; k++) fv = big_fma<R>(u[k], u[k], fv);
                    soft_slack = soft;
                    fval = fv + soft;
                    if (fval > fval_bound) { flag = EXIT_INFEASIBLE; break; }
                    R min_val = -primal_tol;
                    bool broken = false;
                    for (int j = 0; j < m; j++) {
                        if (sense[j] & SENSE_IMMUTABLE) continue;
                        const R *mj = Mr + (size_t)j * n;
                        R Mu = (R)0;
                        for (int k = 0; k < n; k++) Mu = big_fma<R>(mj[k], u[k], Mu);
                        const R vu = dup[j] - Mu;
                        const R vl = -(dlo[j] - Mu);
                        if (sense[j] & SENSE_ACTIVE) {
                            if (!(sense[j] & SENSE_SOFT) && (vu < -primal_tol || vl < -primal_tol)) broken = true;
                            continue;
                        }
                        if (vu < min_val) { add = j; isupper = true; min_val = vu; }
                        else if (vl < min_val) { add = j; isupper = false; min_val = vl; }
                    }
                    if (add < 0) {
                        if (broken) flag = EXIT_CYCLE;
                        else flag = (soft_slack > primal_tol) ? EXIT_SOFT_OPTIMAL : EXIT_OPTIMAL;
                        break;
                    }
                }
                if (add >= 0) {
                    if (na >= cap) { flag = EXIT_WSCAP; break; }
                    for (int i = 0; i < na; i++) lam[i] = ls[i];
                    if (!isupper) sense[add] |= SENSE_LOWER;
                    ldl_add(add);
                    if (fval - best_fval < progress_tol) {
                        if (++cycle > P.cycle_tol) { flag = EXIT_CYCLE; break; }
                    } else { best_fval = fval; cycle = 0; }
                } else {
                    for (int i = 0; i < na; i++) lam[i] = big_fma<R>(alpha, ls[i] - lam[i], lam[i]);
                    ldl_remove(rm);
                }
            } else {
                // singular working set: direction p with M_W' p = 0, p_sing = +-1
                const int sg = sing;
                const R *lsg = L + big_tri(sg);
                for (int i = sg - 1; i >= 0; i--) {
                    R acc = -lsg[i];
                    for (int t = sg - 1; t > i; t--) acc = big_fma<R>(-L[big_tri(t) + i], ls[t], acc);
                    ls[i] = acc;
                }
                ls[sg] = (R)1;
                if (sense[WS[sg]] & SENSE_LOWER)
                    for (int i = 0; i <= sg; i++) ls[i] = -ls[i];
                int nblock = 0, rm = -1;
                R alpha = (R)0;
                for (int i = 0; i < na; i++) {
                    const int j = WS[i];
                    if (sense[j] & SENSE_IMMUTABLE) continue;
                    if (sense[j] & SENSE_LOWER) { if (ls[i] < dual_tol) continue; }
                    else if (ls[i] > -dual_tol) continue;
                    const R cand = -lam[i] / ls[i];
                    if (nblock == 0 || cand < alpha) { alpha = cand; rm = i; }
                    nblock++;
                }
                if (nblock == 0) { flag = EXIT_INFEASIBLE; break; }
                for (int i = 0; i < na; i++) lam[i] = big_fma<R>(alpha, ls[i], lam[i]);
                ldl_remove(rm);
            }
        }

        // ---- x = R^-1 u + x0 + Xth theta   (mpc_update_qp.c:14-22), flags, working set
        double xsim[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < P.nout; k++) {
            R xs = (R)0, sh = C[P.ox0 + k];
            for (int c = 0; c < n; c++) xs = big_fma<R>(C[P.oRout + (size_t)k * n + c], u[c], xs);
            for (int t = 0; t < nth; t++) sh = big_fma<R>(C[P.oXth + (size_t)k * nth + t], th[t], sh);
            const R xk = xs + sh;
            if (X != nullptr) X[pid * P.nout + k] = xk;
            if (k < 8) xsim[k] = (double)xk;
        }
        if constexpr (sizeof(R) == 8) {
            if (sim.FG != nullptr) {             // closed loop with the plant step fused in: advance in place (WaveSim)
                const int snx = sim.nx, snu = sim.nu, ks = sim.kfix >= 0 ? sim.kfix : sim.kstep[pid];
                double xn[8];
                for (int a = 0; a < snx; a++) {
                    double acc = 0.0;
                    for (int c = 0; c < snx; c++) acc = __builtin_fma(sim.FG[a * snx + c], (double)th[c], acc);
                    for (int l = 0; l < snu; l++) acc = __builtin_fma(sim.FG[snx * snx + a * snu + l], xsim[l], acc);
                    xn[a] = acc;
                }
                double *to = const_cast<double *>(reinterpret_cast<const double *>(theta)) + pid * nth;
                for (int a = 0; a < snx; a++) {
                    to[a] = xn[a];
                    if (sim.xtraj) sim.xtraj[((long long)(ks + 1) * sim.nscen + pid) * snx + a] = xn[a];
                }
                for (int l = 0; l < snu; l++) {
                    if (l < sim.nup) to[snx + sim.nr + l] = xsim[l];
                    if (sim.utraj) sim.utraj[((long long)ks * sim.nscen + pid) * snu + l] = xsim[l];
                }
                if (sim.flag_min) sim.flag_min[pid] = ks == 0 ? flag : (flag < sim.flag_min[pid] ? flag : sim.flag_min[pid]);
                if (sim.kfix < 0) sim.kstep[pid] = ks + 1;
            }
        }
        if (active) {
            uint64_t *ap = active + pid * P.words;
            for (int w = 0; w < P.words; w++) ap[w] = 0ull;
            for (int i = 0; i < na; i++) {
                const int j = WS[i];
                const int bit = (sense[j] & SENSE_LOWER) ? m + j : j;
                ap[bit >> 6] |= 1ull << (bit & 63);
            }
        }
        if (exitflag != nullptr) exitflag[pid] = flag;
        if (iters) iters[pid] = iter;
    }
}

}  // namespace lmpc
