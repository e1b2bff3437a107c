// Affine-variational-inequality mode of the batched path: several objectives (players) on one controller give a
// NON-symmetric H, which the reference hands to DAQP with is_avi = !mpQP.is_symmetric
// (/root/reference/src/setup.jl:11-13; objective: /root/reference/src/mpc2mpqp.jl:900-950; online call unchanged:
// /root/reference/src/utils.jl:268-283).  Find x with the rows within their bounds and
// (H x + f + f_theta th)'(y - x) >= 0 for all feasible y -- the Nash equilibrium of the players' QPs.
//
// Host transform (lmpc_setup.cpp, qp_to_avi): u = x - x_unc(th), ML = [I;A] scaled, MR_j = (H^-1 ML_j')', the full
// non-symmetric Gram matrix G = ML MR' (G_jj = 1), bounds d = du/dl + Dth th.  Working set W: G_WW lam* = -d_W,
// u = -MR_W' lam.
//
// Kernel: ONE PROBLEM PER LANE, 64 problems per wavefront walking the batch with a fixed stride.  Every solver array of
// a lane -- the recursively updated L D U factorisation of G_WW (unit lower L, U stored transposed, shared pivots),
// multipliers, iterate, shifted bounds, working set -- lives in a per-wavefront slab of global scratch laid out ELEMENT
// MAJOR (element e of lane l at slab[e * 64 + l]): the 64 lanes of a wavefront touch 512 contiguous bytes whenever they
// are at the same index, which they are except inside data-dependent Gram lookups.  The constant pack (ML, MR, G,
// bounds, maps) is staged in LDS when it fits (LDSC) and read from global memory through the scalar / vector caches
// otherwise.  No MFMA: the factor is at most (n + 1 + #soft)^2 / 2 reals per problem and every step is a short
// dependent chain; the kernel is latency bound on its own scratch, which residency (16 wavefronts per CU) hides.
//
// Algorithm = the CPU checker's AVI solver statement for statement (same fma chains, same order), i.e. the dual active-set
// loop of the QP kernels with two changes an AVI needs: the factorisation is L D U (appending a row adds a row of L
// and a column of U; removing one is Bennett's rank-one update), and a step lam -> lam* also stops where a row that is
// satisfied at the current iterate would become violated, taking that row into the working set (Cottle-Dantzig
// principal pivoting: without a dual objective the QP loop can cycle on a non-symmetric G, with the satisfied rows
// protected the set of violated rows only shrinks).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lmpc_pack.hpp"
#include "lmpc_wave_layout.hpp"

namespace lmpc {

__host__ __device__ constexpr long long avi_tri(long long i) { return i * (i + 1) / 2; }
__host__ __device__ inline long long avi_scratch_reals(int n, int m, int cap) {
    return 2 * avi_tri(cap + 1) + 8ll * (cap + 1) + 4ll * n + 2ll * m;      // (2 n of them: the proximal iterates)
}
__host__ __device__ inline long long avi_scratch_ints(int m, int cap) { return (cap + 1) + (long long)m; }

// PROX: proximal-point iterations around the solve (DAQP's eps_prox; a symmetric positive SEMIdefinite H, which the
// reference's setup! otherwise answers with "-5 nonconvex", /root/reference/src/setup.jl:18-19): subproblem k has the
// Hessian H + eps I (that is the pack) and the linear term moved by -eps x_k, i.e. bounds d0 - eps MR x_k and the
// unconstrained optimum x_unc0 + eps (H + eps I)^-1 x_k; it starts from the previous subproblem's final working set;
// the loop ends when |x_{k+1} - x_k|_inf < eta_prox.  The CPU checker's oracle_avi_prox_solve_batch statement for
// statement.
template <bool LDSC, bool PROX = false>
__global__ __launch_bounds__(64) void avi_kernel(
    const AviLayout P, const double *__restrict__ C, const int32_t *__restrict__ S, const double *__restrict__ theta,
    double *__restrict__ X, int32_t *__restrict__ exitflag, int32_t *__restrict__ iters, uint64_t *__restrict__ active,
    const uint64_t *__restrict__ warm, double *__restrict__ scratch_r, int32_t *__restrict__ scratch_i, long long nprob,
    const int32_t *__restrict__ list = nullptr, const int32_t *__restrict__ count = nullptr, long long seg_cap = 0,
    int32_t *__restrict__ count_clear = nullptr) {
    extern __shared__ double lds_pack[];
    const int n = P.n, m = P.m, nth = P.nth, cap = P.cap;
    const int lane = threadIdx.x;
    // work-list mode (behind avi_tiers_kernel, lmpc_avi_tiers_kernel.hpp): the problems to solve are the entries of
    // kShards segments of `list` (segment s: count[s * kCountStride] entries from list[s * seg_cap]); workgroup b walks
    // segment b % kShards (the grid is a multiple of kShards).  The counters of the chain's FIRST list are cleared here
    // for the next call (the kernel that read them has finished: stream order).
    if (count_clear && blockIdx.x == 0 && lane < kShards) count_clear[lane * kCountStride] = 0;
    long long first = (long long)blockIdx.x * 64, stride = (long long)gridDim.x * 64, cnt = nprob;
    if (list) {
        const int shard = (int)(blockIdx.x % kShards);
        first = (long long)(blockIdx.x / kShards) * 64;
        stride = (long long)(gridDim.x / kShards) * 64;
        cnt = (long long)count[shard * kCountStride];
        list += (long long)shard * seg_cap;
        if (first >= cnt) return;                 // (nothing listed for this wavefront: the usual case behind the lane kernel)
    }
    if constexpr (LDSC) {
        for (int e = lane; e < P.nC; e += 64) lds_pack[e] = C[e];
        __syncthreads();
    }
    auto cst = [&](int off) -> double {
        if constexpr (LDSC) return lds_pack[off];
        else return C[off];
    };
    // this wavefront's slab, element major
    double *sr = scratch_r + (long long)blockIdx.x * avi_scratch_reals(n, m, cap) * 64 + lane;
    int32_t *si = scratch_i + (long long)blockIdx.x * avi_scratch_ints(m, cap) * 64 + lane;
    long long o = 0;
    const long long oL = o; o += avi_tri(cap + 1);
    const long long oU = o; o += avi_tri(cap + 1);
    const long long oD = o; o += cap + 1;
    const long long oDi = o; o += cap + 1;
    const long long oLam = o; o += cap + 1;
    const long long oLs = o; o += cap + 1;
    const long long oXl = o; o += cap + 1;
    const long long oZl = o; o += cap + 1;
    const long long oPv = o; o += cap + 1;
    const long long oQv = o; o += cap + 1;
    const long long oUc = o; o += n;
    const long long oUt = o; o += n;
    const long long oDup = o; o += m;
    const long long oDlo = o; o += m;
    const long long oXk = o; o += n;
    const long long oXn = o;
#define RS(off, i) sr[((off) + (i)) * 64]
#define WSI(i) si[(long long)(i) * 64]
#define SEN(j) si[((long long)(cap + 1) + (j)) * 64]
    const double primal_tol = P.primal_tol, dual_tol = P.dual_tol, zero_tol = P.zero_tol, rho_soft = P.rho_soft;

    for (long long base = first; base < cnt; base += stride) {
        if (base + lane >= cnt) continue;         // (no barrier below: lanes of the last tile may leave)
        const long long pid = list ? (long long)list[base + lane] : base + lane;
        const double *th = theta + pid * nth;
        int na = 0, sing = -1, reuse = 0, nsoft_act = 0;
        int flag = EXIT_ITERLIMIT, iter = 1;
        double soft_slack = 0.0;
        bool done = false;
        int total = 0, outer = 0;                 // PROX: iterations of all subproblems so far, subproblems solved
        if constexpr (PROX) { for (int k = 0; k < n; k++) RS(oXk, k) = 0.0; }

        auto ldu_add = [&](int j) {
            const long long rl = oL + avi_tri(na), ru = oU + avi_tri(na);
            for (int i = 0; i < na; i++) {
                const int wi = WSI(i);
                RS(rl, i) = cst(P.oG + j * m + wi);
                RS(ru, i) = cst(P.oG + wi * m + j);
            }
            double dnew = cst(P.oG + j * m + j);
            const int sj = SEN(j);
            if (sj & SENSE_SOFT) dnew += rho_soft;
            for (int i = 0; i < na; i++) {
                double al = RS(rl, i), au = RS(ru, i);
                const long long ui = oU + avi_tri(i), li = oL + avi_tri(i);
                for (int t = 0; t < i; t++) {
                    al = __builtin_fma(-RS(ui, t), RS(rl, t), al);
                    au = __builtin_fma(-RS(li, t), RS(ru, t), au);
                }
                RS(rl, i) = al; RS(ru, i) = au;
            }
            for (int i = 0; i < na; i++) {
                const double ql = RS(rl, i), qu = RS(ru, i), di = RS(oDi, i);
                const double l = ql * di;
                RS(rl, i) = l;
                RS(ru, i) = qu * di;
                dnew = __builtin_fma(-l, qu, dnew);
            }
            RS(rl, na) = 1.0; RS(ru, na) = 1.0;
            const bool is_soft = (sj & SENSE_SOFT) != 0;
            if (dnew < zero_tol || (!is_soft && na - nsoft_act >= n)) {
                RS(oD, na) = 0.0; RS(oDi, na) = 0.0; sing = na;
            } else {
                RS(oD, na) = dnew; RS(oDi, na) = 1.0 / dnew;
            }
            WSI(na) = j; RS(oLam, na) = 0.0; RS(oLs, na) = 0.0;
            SEN(j) = sj | SENSE_ACTIVE;
            nsoft_act += is_soft ? 1 : 0;
            na++;
        };
        auto ldu_remove = [&](int r) {
            const int nup = na - r - 1;
            double alpha = RS(oD, r);
            for (int t = 0; t < nup; t++) {
                RS(oPv, t) = RS(oL + avi_tri(r + 1 + t), r);
                RS(oQv, t) = RS(oU + avi_tri(r + 1 + t), r);
            }
            for (int i = r; i < na - 1; i++) {
                const long long dl_ = oL + avi_tri(i), du_ = oU + avi_tri(i), sl = oL + avi_tri(i + 1), su = oU + avi_tri(i + 1);
                for (int c = 0; c < r; c++) { RS(dl_, c) = RS(sl, c); RS(du_, c) = RS(su, c); }
                for (int c = r; c < i; c++) { RS(dl_, c) = RS(sl, c + 1); RS(du_, c) = RS(su, c + 1); }
                RS(dl_, i) = 1.0; RS(du_, i) = 1.0;
            }
            sing = -1;
            for (int t = 0; t < nup; t++) {
                const int i = r + t;
                const double pt = RS(oPv, t), qt = RS(oQv, t);
                const double dold = RS(oD, i + 1);
                const double dbar = __builtin_fma(alpha * pt, qt, dold);
                if (dbar < zero_tol) {
                    RS(oD, i) = 0.0; RS(oDi, i) = 0.0; sing = i;
                    for (int q = i + 1; q < na - 1; q++) { RS(oD, q) = RS(oD, q + 1); RS(oDi, q) = RS(oDi, q + 1); }
                    break;
                }
                const double rinv = 1.0 / dbar;
                const double betaL = (qt * alpha) * rinv;
                const double betaU = (pt * alpha) * rinv;
                alpha = (dold * alpha) * rinv;
                RS(oD, i) = dbar; RS(oDi, i) = rinv;
                for (int q = t + 1; q < nup; q++) {
                    const long long lqi = oL + avi_tri(r + q) + i, uqi = oU + avi_tri(r + q) + i;
                    const double pq = __builtin_fma(-pt, RS(lqi, 0), RS(oPv, q));
                    RS(oPv, q) = pq;
                    RS(lqi, 0) = __builtin_fma(betaL, pq, RS(lqi, 0));
                    const double qq = __builtin_fma(-qt, RS(uqi, 0), RS(oQv, q));
                    RS(oQv, q) = qq;
                    RS(uqi, 0) = __builtin_fma(betaU, qq, RS(uqi, 0));
                }
            }
            const int jr = WSI(r);
            if (SEN(jr) & SENSE_SOFT) nsoft_act--;
            SEN(jr) &= ~(SENSE_ACTIVE | SENSE_LOWER);
            for (int i = r; i < na - 1; i++) { WSI(i) = WSI(i + 1); RS(oLam, i) = RS(oLam, i + 1); }
            na--;
            if (r < reuse) reuse = r;
        };

        for (;;) {                                // (one trip unless PROX: the proximal-point iterations)
        for (int j = 0; j < m; j++) {             // bounds of this parameter point   (mpc_update_qp.c:1-10)
            double sh = 0.0;
            for (int t = 0; t < nth; t++) sh = __builtin_fma(cst(P.oDth + j * nth + t), th[t], sh);
            double du_ = cst(P.odu + j) + sh, dl_ = cst(P.odl + j) + sh;
            if constexpr (PROX) {
                double acc = 0.0;
                for (int c = 0; c < n; c++) acc = __builtin_fma(cst(P.oMR + j * n + c), RS(oXk, c), acc);
                du_ = du_ - P.eps_prox * acc;
                dl_ = dl_ - P.eps_prox * acc;
            }
            RS(oDup, j) = du_;
            RS(oDlo, j) = dl_;
        }
        na = 0; sing = -1; reuse = 0; nsoft_act = 0; flag = EXIT_ITERLIMIT; iter = 1; soft_slack = 0.0; done = false;
        for (int k = 0; k < n; k++) RS(oUc, k) = 0.0;
        const int ilimit = PROX ? P.iter_limit - total : P.iter_limit;
        // ---- initial working set: rows flagged ACTIVE, the caller's warm-start mask and (PROX, later subproblems) the
        // previous subproblem's final working set, in row order
        const uint64_t *wp = (warm && outer == 0) ? warm + pid * P.words : nullptr;
        for (int j = 0; j < m && !done; j++) {
            const int s0 = S[j];
            const int prev = (PROX && outer > 0) ? SEN(j) : 0;
            SEN(j) = s0 & ~SENSE_LOWER;
            bool want = (s0 & SENSE_ACTIVE) != 0, lower = want && (s0 & SENSE_LOWER);
            if (wp && !(s0 & SENSE_IMMUTABLE)) {
                if ((wp[j >> 6] >> (j & 63)) & 1ull) want = true;
                else if ((wp[(m + j) >> 6] >> ((m + j) & 63)) & 1ull) { want = true; lower = true; }
            }
            if ((prev & SENSE_ACTIVE) && !(s0 & SENSE_IMMUTABLE)) { want = true; lower = (prev & SENSE_LOWER) != 0; }
            if (!want) { SEN(j) &= ~SENSE_ACTIVE; continue; }
            if (lower) SEN(j) |= SENSE_LOWER;
            if (na >= cap) { flag = EXIT_WSCAP; done = true; break; }
            ldu_add(j);
            if (sing >= 0) {
                if (s0 & SENSE_IMMUTABLE) { flag = EXIT_OVERDETERMINED; done = true; break; }
                na--; sing = -1;
                if (SEN(j) & SENSE_SOFT) nsoft_act--;
                SEN(j) &= ~(SENSE_ACTIVE | SENSE_LOWER);
            }
        }
        // (a start that ends early leaves the rows behind it with the flags of this lane's previous problem: nothing
        // reads them -- the outputs go through the working set, and no further subproblem follows a failed one)

        for (; !done && iter < ilimit; iter++) {
            if (sing < 0) {
                int nblock = 0, rm = -1, add = -1;
                bool isupper = false;
                double alpha = 0.0;
                for (int i = reuse; i < na; i++) {                      // (L D U) lam* = -d_W
                    const int j = WSI(i);
                    double acc = (SEN(j) & SENSE_LOWER) ? -RS(oDlo, j) : -RS(oDup, j);
                    const long long li = oL + avi_tri(i);
                    for (int t = 0; t < i; t++) acc = __builtin_fma(-RS(li, t), RS(oXl, t), acc);
                    RS(oXl, i) = acc;
                }
                for (int i = reuse; i < na; i++) RS(oZl, i) = RS(oXl, i) * RS(oDi, i);
                for (int i = na - 1; i >= 0; i--) {
                    double acc = RS(oZl, i);
                    for (int t = na - 1; t > i; t--) acc = __builtin_fma(-RS(oU + avi_tri(t), i), RS(oLs, t), acc);
                    RS(oLs, i) = acc;
                }
                reuse = na;
                for (int i = 0; i < na; i++) {
                    const int sj = SEN(WSI(i));
                    if (sj & SENSE_IMMUTABLE) continue;
                    const double lsi = RS(oLs, i), lai = RS(oLam, i);
                    if (sj & SENSE_LOWER) { if (lsi < dual_tol) continue; }
                    else if (lsi > -dual_tol) continue;
                    const double cand = -lai / (lsi - lai);
                    if (nblock == 0 || cand < alpha) { alpha = cand; rm = i; }
                    nblock++;
                }
                // the iterate now (from lam) and the target of this step (from lam*)
                double soft = 0.0;
                for (int k = 0; k < n; k++) { RS(oUc, k) = 0.0; RS(oUt, k) = 0.0; }
                for (int i = 0; i < na; i++) {
                    const int j = WSI(i);
                    const double lc = RS(oLam, i), lt = RS(oLs, i);
                    for (int k = 0; k < n; k++) {
                        const double mk = cst(P.oMR + j * n + k);
                        RS(oUc, k) = __builtin_fma(-mk, lc, RS(oUc, k));
                        RS(oUt, k) = __builtin_fma(-mk, lt, RS(oUt, k));
                    }
                    if (SEN(j) & SENSE_SOFT) soft = __builtin_fma(lt * lt, rho_soft, soft);
                }
                double min_val = -primal_tol, tblk = nblock ? alpha : 1.0;
                bool broken = false, pup = false;
                int pblk = -1;
                for (int j = 0; j < m; j++) {
                    const int sj = SEN(j);
                    if (sj & SENSE_IMMUTABLE) continue;
                    double Mc = 0.0, Mt = 0.0;
                    for (int k = 0; k < n; k++) {
                        const double mk = cst(P.oML + j * n + k);
                        Mc = __builtin_fma(mk, RS(oUc, k), Mc);
                        Mt = __builtin_fma(mk, RS(oUt, k), Mt);
                    }
                    const double dj = RS(oDup, j), ej = RS(oDlo, j);
                    const double vu = dj - Mt, vl = -(ej - Mt);
                    if (sj & SENSE_ACTIVE) {
                        if (!(sj & SENSE_SOFT) && (vu < -primal_tol || vl < -primal_tol)) broken = true;
                        continue;
                    }
                    if (vu < min_val) { add = j; isupper = true; min_val = vu; }
                    else if (vl < min_val) { add = j; isupper = false; min_val = vl; }
                    const double cu = dj - Mc, cl = -(ej - Mc);
                    if (vu < -primal_tol && cu >= -primal_tol) {
                        const double t = cu > 0.0 ? cu / (cu - vu) : 0.0;
                        if (t < tblk) { tblk = t; pblk = j; pup = true; }
                    } else if (vl < -primal_tol && cl >= -primal_tol) {
                        const double t = cl > 0.0 ? cl / (cl - vl) : 0.0;
                        if (t < tblk) { tblk = t; pblk = j; pup = false; }
                    }
                }
                if (pblk >= 0) {
                    if (na >= cap) { flag = EXIT_WSCAP; break; }
                    for (int i = 0; i < na; i++) RS(oLam, i) = __builtin_fma(tblk, RS(oLs, i) - RS(oLam, i), RS(oLam, i));
                    if (!pup) SEN(pblk) |= SENSE_LOWER;
                    ldu_add(pblk);
                } else if (nblock) {
                    for (int i = 0; i < na; i++) RS(oLam, i) = __builtin_fma(alpha, RS(oLs, i) - RS(oLam, i), RS(oLam, i));
                    ldu_remove(rm);
                } else {
                    for (int k = 0; k < n; k++) RS(oUc, k) = RS(oUt, k);
                    soft_slack = soft;
                    if (add < 0) {
                        if (broken) flag = EXIT_CYCLE;
                        else flag = (soft_slack > primal_tol) ? EXIT_SOFT_OPTIMAL : EXIT_OPTIMAL;
                        break;
                    }
                    if (na >= cap) { flag = EXIT_WSCAP; break; }
                    for (int i = 0; i < na; i++) RS(oLam, i) = RS(oLs, i);
                    if (!isupper) SEN(add) |= SENSE_LOWER;
                    ldu_add(add);
                }
            } else {
                // singular working set: G_WW p = 0 <=> U p = e_sg, p_sg = +-1
                const int sg = sing;
                const long long us = oU + avi_tri(sg);
                for (int i = sg - 1; i >= 0; i--) {
                    double acc = -RS(us, i);
                    for (int t = sg - 1; t > i; t--) acc = __builtin_fma(-RS(oU + avi_tri(t), i), RS(oLs, t), acc);
                    RS(oLs, i) = acc;
                }
                RS(oLs, sg) = 1.0;
                if (SEN(WSI(sg)) & SENSE_LOWER)
                    for (int i = 0; i <= sg; i++) RS(oLs, i) = -RS(oLs, i);
                for (int i = sg + 1; i < na; i++) RS(oLs, i) = 0.0;
                int nblock = 0, rm = -1;
                double alpha = 0.0;
                for (int i = 0; i < na; i++) {
                    const int sj = SEN(WSI(i));
                    if (sj & SENSE_IMMUTABLE) continue;
                    const double lsi = RS(oLs, i);
                    if (sj & SENSE_LOWER) { if (lsi < dual_tol) continue; }
                    else if (lsi > -dual_tol) continue;
                    const double cand = -RS(oLam, i) / lsi;
                    if (nblock == 0 || cand < alpha) { alpha = cand; rm = i; }
                    nblock++;
                }
                if (nblock == 0) { flag = EXIT_INFEASIBLE; break; }
                for (int i = 0; i < na; i++) RS(oLam, i) = __builtin_fma(alpha, RS(oLs, i), RS(oLam, i));
                ldu_remove(rm);
            }
        }

        if constexpr (!PROX) break;
        else {
            total += iter;
            if (flag < 1) break;
            double diff = 0.0;
            for (int k = 0; k < n; k++) {
                double a = cst(P.ox0f + k), b = 0.0;
                for (int t = 0; t < nth; t++) a = __builtin_fma(cst(P.oXthf + k * nth + t), th[t], a);
                for (int c = 0; c < n; c++) b = __builtin_fma(cst(P.oHinv + k * n + c), RS(oXk, c), b);
                const double xv = (RS(oUc, k) + a) + P.eps_prox * b;
                RS(oXn, k) = xv;
                const double d = __builtin_fabs(xv - RS(oXk, k));
                if (d > diff) diff = d;
            }
            for (int k = 0; k < n; k++) RS(oXk, k) = RS(oXn, k);
            outer++;
            if (diff < P.eta_prox) break;
            if (total >= P.iter_limit) { flag = EXIT_ITERLIMIT; break; }
        }
        }   // proximal-point iterations

        // ---- x = Rout u + x0 + Xth theta   (mpc_update_qp.c:14-22), flags, working set
        for (int k = 0; k < P.nout; k++) {
            if constexpr (PROX) {
                double sh = 0.0;
                for (int t = 0; t < nth; t++) sh = __builtin_fma(cst(P.oKth + k * nth + t), th[t], sh);
                X[pid * P.nout + k] = RS(oXk, k) + sh;
            } else {
                double xs = 0.0, sh = cst(P.ox0 + k);
                for (int c = 0; c < n; c++) xs = __builtin_fma(cst(P.oRout + k * n + c), RS(oUc, c), xs);
                for (int t = 0; t < nth; t++) sh = __builtin_fma(cst(P.oXth + k * nth + t), th[t], sh);
                X[pid * P.nout + k] = xs + sh;
            }
        }
        if (active) {
            uint64_t *ap = active + pid * P.words;
            for (int w = 0; w < P.words; w++) ap[w] = 0ull;
            for (int i = 0; i < ((PROX && flag < 1) ? 0 : na); i++) {
                const int j = WSI(i);
                const int bit = (SEN(j) & SENSE_LOWER) ? m + j : j;
                ap[bit >> 6] |= 1ull << (bit & 63);
            }
        }
        exitflag[pid] = flag;
        if (iters) iters[pid] = PROX ? total : iter;
    }
#undef RS
#undef WSI
#undef SEN
}

}  // namespace lmpc
