// One-QP-per-wavefront dual active-set kernel: the general path (n <= 63 variables, working sets
// up to 64 rows, m <= 256 constraints, hard and SOFT rows).
//
// Mapping (gfx950): a 64-lane wavefront owns one parameter point.  Working-set position i lives
// on lane i (its multiplier, right-hand side, pivot, constraint id are that lane's registers),
// variable k of the primal iterate u lives on lane k, constraint j on lane j % 64 (register slot
// j / 64).  Only the LDL' factor sits in LDS, column-major with an odd leading dimension so that
// both the column sweeps of the triangular solves and the row append are (near) conflict-free.
// Triangular solves and the rank-one update are column sweeps: one v_readlane broadcast of the
// finished entry, one fma on every lane behind it; arg-min searches are butterfly reductions over
// (value, index) pairs.  All control flow is wave-uniform: no divergence inside a problem.
// The shared problem data (M row-major for M_W rows, M transposed for the constraint scan, the
// packed Gram matrix, Dth, bounds) are read from the one constant buffer through L1/L2.
//
// Every fma chain runs in the same order as the CPU oracle's loops (per lane sequentially, or as
// a wave-uniform serial chain where the oracle reduces over positions), so the results are
// bit-comparable with it, exactly as for the one-QP-per-lane kernel.
//
// Replaces, per problem: mpc_update_qp (reference codegen/mpc_update_qp.c:1-10), daqp_ldp incl.
// soft constraints ([EXT] libdaqp, called at mpc_update_qp.c:48 / utils.jl:282; rho_soft from
// setup.jl:26) and mpc_get_solution (mpc_update_qp.c:14-22).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lmpc_pack.hpp"

namespace lmpc {

struct WaveLayout {
    int n, m, ms, nth, nout, words;
    int cap, ldc;                                   // working-set capacity, leading dim of L
    int oM, oMt, oG, odu, odl, oDth, oRout, ox0, oXth;
    double primal_tol, dual_tol, zero_tol, progress_tol, fval_bound, rho_soft;
    int cycle_tol, iter_limit;
};

__device__ __forceinline__ double wv_bcast(double v, int src) {
    // `src` is wave-uniform: two v_readlane_b32
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// value held identically by all lanes -> scalar registers (tells the compiler it is wave-uniform)
__device__ __forceinline__ double wv_first(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// lexicographic (value, index) minimum over the wave; idx < 0 marks "no candidate"
__device__ __forceinline__ void wv_argmin(double &val, int &idx) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(val, off);
        const int oi = __shfl_xor(idx, off);
        const bool take = (oi >= 0) && (idx < 0 || ov < val || (ov == val && oi < idx));
        if (take) { val = ov; idx = oi; }
    }
}

// MR: register slots per lane for constraints (m <= 64*MR).  LDSC: the shared problem data
// (M, M transposed, packed Gram) are staged once per workgroup in LDS behind the per-wave factors
// (when they fit) -- their reads sit on every iteration's critical path, and an LDS read returns in
// ~1/8 of the time of an L2 hit.
// BNB: rows flagged BINARY must end up active at one of their bounds -- depth-first branch and
// bound over them around the same node solver (what the reference gets from daqp_bnb, [EXT]).
template <int MR, bool LDSC, bool BNB>
__global__ __launch_bounds__(256) void wave_kernel(
    const WaveLayout P, const double *__restrict__ C, const int32_t *__restrict__ S,
    const double *__restrict__ theta, double *__restrict__ X, int32_t *__restrict__ exitflag,
    int32_t *__restrict__ iters, uint64_t *__restrict__ active, const uint64_t *__restrict__ warm,
    long long nprob) {
    extern __shared__ __align__(16) double lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const int n = P.n, m = P.m, nth = P.nth, ldc = P.ldc;
    double *L = lds + (size_t)wv * P.cap * ldc;      // L(i,t) = L[t*ldc + i], i > t
    const double *Mr = C + P.oM, *Mt = C + P.oMt, *G = C + P.oG;
    if (LDSC) {
        double *sc = lds + (size_t)nwv * P.cap * ldc;
        const int nM = m * n, nG = m * (m + 1) / 2;
        for (int i = threadIdx.x; i < nM; i += blockDim.x) { sc[i] = C[P.oM + i]; sc[nM + i] = C[P.oMt + i]; }
        for (int i = threadIdx.x; i < nG; i += blockDim.x) sc[2 * nM + i] = C[P.oG + i];
        __syncthreads();
        Mr = sc; Mt = sc + nM; G = sc + 2 * nM;
    }

    int sense0[MR], sense[MR];                       // constraint slots of this lane: as given / of the
#pragma unroll                                       // current solve (a B&B node adds its fixed binaries)
    for (int r = 0; r < MR; r++) {
        const int j = lane + 64 * r;
        sense0[r] = (j < m) ? S[j] : SENSE_IMMUTABLE;
        sense[r] = sense0[r];
    }
    // flags of row j (wave-uniform) from the slot that owns it
    auto sense_of = [&](int j) -> int {
        int v = 0;
#pragma unroll
        for (int r = 0; r < MR; r++) if (r == (j >> 6)) v = sense[r];
        return __builtin_amdgcn_readlane(v, j & 63);
    };

    for (long long pid = (long long)blockIdx.x * nwv + wv; pid < nprob; pid += (long long)gridDim.x * nwv) {
        const double *th = theta + pid * nth;
        double b[MR];                            // b_j = Dth_j . theta   (mpc_update_qp.c:5-6)
        unsigned actb = 0u, lowb = 0u;           // bit r: slot r active / active at its lower bound
#pragma unroll
        for (int r = 0; r < MR; r++) {
            const int j = lane + 64 * r;
            double acc = 0.0;
            if (j < m)
                for (int t = 0; t < nth; t++) acc = __builtin_fma(C[P.oDth + j * nth + t], th[t], acc);
            b[r] = acc;
        }
        // registers of working-set position `lane`
        int WSi = 0, possoft = 0, posimm = 0, poslow = 0;
        double lam = 0.0, ls = 0.0, rhs = 0.0, D = 0.0, Dinv = 0.0;
        double u = 0.0;                          // lane k < n holds u_k
        int na = 0, sing = -1, iter = 1, cyc = 0, flag = EXIT_ITERLIMIT, nsoft_act = 0;
        double best = -1.0, fval = 0.0, soft_slack = 0.0;
        bool done = false;
        double fbound = P.fval_bound;            // a B&B node stops as soon as it is dominated

        auto Gat = [&](int a, int c) -> double {
            return a >= c ? G[(size_t)a * (a + 1) / 2 + c] : G[(size_t)c * (c + 1) / 2 + a];
        };

        // Column sweeps: one v_readlane broadcast of the finished entry, one fma on every lane behind it.
        // (A four-columns-per-trip variant with the LDS reads hoisted was measured: +3 % on working
        // sets of ~10 rows, -15 % on small ones; not kept.)
        // forward: v_i -= L(i,t) v_t for t = 0 .. na-2 in order (lane i holds v_i)
        auto sweep_fwd = [&](double v) -> double {
            for (int t = 0; t + 1 < na; t++) {
                const double vt = wv_bcast(v, t);
                if (lane > t && lane < na) v = __builtin_fma(-L[t * ldc + lane], vt, v);
            }
            return v;
        };
        // backward: v_i -= L(t,i) v_t for t = top .. 1 in descending order
        auto sweep_bwd = [&](double v, int top) -> double {
            for (int t = top; t >= 1; t--) {
                const double vt = wv_bcast(v, t);
                if (lane < t) v = __builtin_fma(-L[lane * ldc + t], vt, v);
            }
            return v;
        };

        // ---- append constraint j (wave-uniform) to the working set
        auto ldl_add = [&](int j, bool lower) {
            const int sj = sense_of(j);
            const bool is_soft = (sj & SENSE_SOFT) != 0;
            double q = (lane < na) ? Gat(WSi, j) : 0.0;
            q = sweep_fwd(q);
            const double l = q * Dinv;               // lanes >= na: 0 * 0
            double dnew = Gat(j, j);
            if (is_soft) dnew += P.rho_soft;
            for (int i = 0; i < na; i++) dnew = __builtin_fma(-wv_bcast(l, i), wv_bcast(q, i), dnew);
            const bool singular = (dnew < P.zero_tol) || (!is_soft && (na - nsoft_act) >= n);
            if (lane < na) L[lane * ldc + na] = l;   // new row: L(na, t) written by lane t
            // bound of row j from the slot that owns it
            double bj = 0.0;
#pragma unroll
            for (int r = 0; r < MR; r++) if (r == (j >> 6)) bj = b[r];
            bj = wv_bcast(bj, j & 63);
            const double rj = lower ? -(C[P.odl + j] + bj) : -(C[P.odu + j] + bj);
            if (lane == na) {
                WSi = j; possoft = is_soft ? 1 : 0; posimm = (sj & SENSE_IMMUTABLE) ? 1 : 0; poslow = lower ? 1 : 0;
                rhs = rj; lam = 0.0; ls = 0.0;
                D = singular ? 0.0 : dnew;
                Dinv = singular ? 0.0 : 1.0 / dnew;
            }
            if (lane == (j & 63)) {
                actb |= 1u << (j >> 6);
                if (lower) lowb |= 1u << (j >> 6);
            }
            if (singular) sing = na;
            nsoft_act += is_soft ? 1 : 0;
            na++;
        };

        // ---- drop working-set position r (wave-uniform): compact L, rank-one update of the tail
        auto ldl_remove = [&](int r) {
            const int nao = na;
            double w = (lane > r && lane < nao) ? L[r * ldc + lane] : 0.0;   // old row index = lane
            double alpha = wv_bcast(D, r);
            const int jrem = __builtin_amdgcn_readlane(WSi, r);
            const int softrem = __builtin_amdgcn_readlane(possoft, r);
            // new L(i,c): old L(i+1,c) for c < r, old L(i+1,c+1) for c >= r   (i >= r)
            for (int c = 0; c + 1 < nao - 1; c++) {
                const int srcc = c < r ? c : c + 1;
                const int lo = (c + 1 > r) ? c + 1 : r;
                double v = 0.0;
                const bool mine = lane >= lo && lane < nao - 1;
                if (mine) v = L[srcc * ldc + lane + 1];
                if (mine) L[c * ldc + lane] = v;
            }
            // shift the per-position registers down by one from position r on
            {
                const int wn = __shfl_down(WSi, 1), sn = __shfl_down(possoft, 1), in = __shfl_down(posimm, 1),
                          ln = __shfl_down(poslow, 1);
                const double lamn = __shfl_down(lam, 1), rhsn = __shfl_down(rhs, 1), Dn = __shfl_down(D, 1),
                             Din = __shfl_down(Dinv, 1), wnn = __shfl_down(w, 1);
                if (lane >= r && lane < nao - 1) {
                    WSi = wn; possoft = sn; posimm = in; poslow = ln; lam = lamn; rhs = rhsn; D = Dn; Dinv = Din;
                    w = wnn;
                } else if (lane == nao - 1) {
                    WSi = 0; possoft = 0; posimm = 0; poslow = 0; lam = 0.0; rhs = 0.0; D = 0.0; Dinv = 0.0; w = 0.0;
                } else {
                    w = 0.0;
                }
            }
            na = nao - 1;
            sing = -1;
            for (int t = r; t < na; t++) {
                const double pt = wv_bcast(w, t);
                const double dold = wv_bcast(D, t);
                const double dbar = __builtin_fma(alpha * pt, pt, dold);
                if (dbar < P.zero_tol) {
                    if (lane == t) { D = 0.0; Dinv = 0.0; }
                    sing = t;
                    break;
                }
                const double rinv = 1.0 / dbar;
                const double beta = (pt * alpha) * rinv;
                alpha = (dold * alpha) * rinv;
                if (lane == t) { D = dbar; Dinv = rinv; }
                if (lane > t && lane < na) {
                    const double lq = L[t * ldc + lane];
                    w = __builtin_fma(-pt, lq, w);
                    L[t * ldc + lane] = __builtin_fma(beta, w, lq);
                }
            }
            if (lane == (jrem & 63)) {
                actb &= ~(1u << (jrem >> 6));
                lowb &= ~(1u << (jrem >> 6));
            }
            nsoft_act -= softrem;
        };

        // ---- blocking search over the working set: (alpha, rm) = first minimum of the ratios
        auto blocking = [&](bool singular_dir, double &alpha, int &rm) {
            const bool okd = poslow ? (ls < P.dual_tol) : (ls > -P.dual_tol);
            const bool blk = (lane < na) && !posimm && !okd;
            double cand = 0.0;
            if (blk) cand = singular_dir ? (-lam / ls) : (-lam / (ls - lam));
            int idx = blk ? lane : -1;
            wv_argmin(cand, idx);
            alpha = wv_first(cand);
            rm = __builtin_amdgcn_readfirstlane(idx);
        };

        // ---- one LDP solve with the flags in sense[] (cold, or warm from the caller's mask)
        auto solve_node = [&]() {
        WSi = 0; possoft = 0; posimm = 0; poslow = 0;
        lam = 0.0; ls = 0.0; rhs = 0.0; D = 0.0; Dinv = 0.0; u = 0.0;
        actb = 0u; lowb = 0u;
        na = 0; sing = -1; iter = 1; cyc = 0; flag = EXIT_ITERLIMIT; nsoft_act = 0;
        best = -1.0; fval = 0.0; soft_slack = 0.0; done = false;
        // initial working set: rows flagged ACTIVE (equalities, fixed binaries), then the warm-start mask
        for (int j = 0; j < m && !done; j++) {
            const int sj = sense_of(j);
            bool want = (sj & SENSE_ACTIVE) != 0, lower = want && (sj & SENSE_LOWER) != 0;
            if (warm != nullptr && !(sj & SENSE_IMMUTABLE)) {
                const uint64_t *wp = warm + pid * P.words;
                if ((wp[j >> 6] >> (j & 63)) & 1ull) want = true;
                else if ((wp[(m + j) >> 6] >> ((m + j) & 63)) & 1ull) { want = true; lower = true; }
            }
            if (!want) continue;
            ldl_add(j, lower);
            if (sing >= 0) {
                if (sj & SENSE_IMMUTABLE) { flag = EXIT_OVERDETERMINED; done = true; }
                else {                              // dependent warm-start row: take it out again
                    na--;
                    sing = -1;
                    if (lane == na) { WSi = 0; possoft = 0; posimm = 0; poslow = 0; rhs = 0.0; D = 0.0; Dinv = 0.0; }
                    if (sj & SENSE_SOFT) nsoft_act--;
                    if (lane == (j & 63)) { actb &= ~(1u << (j >> 6)); lowb &= ~(1u << (j >> 6)); }
                }
            }
        }

        // ---- dual active-set iterations
        while (!done) {
            if (iter >= P.iter_limit) { flag = EXIT_ITERLIMIT; break; }
            int rm = -1;
            double alpha = 0.0;
            if (sing < 0) {
                // constrained stationary point (L D L') lam* = rhs by two column sweeps
                double x = (lane < na) ? rhs : 0.0;
                x = sweep_fwd(x);
                double acc = sweep_bwd(x * Dinv, na - 1);
                ls = (lane < na) ? acc : 0.0;
                blocking(false, alpha, rm);
                if (rm < 0) {
                    // primal iterate u = -M_W' lam* (lane k owns u_k), objective, then the scan
                    double uk = 0.0;
                    for (int i = 0; i < na; i++) {
                        const int w = __builtin_amdgcn_readlane(WSi, i);
                        const double l = wv_bcast(ls, i);
                        if (lane < n) uk = __builtin_fma(-Mr[(size_t)w * n + lane], l, uk);
                    }
                    u = uk;
                    double fv = 0.0, soft = 0.0;
                    for (int k = 0; k < n; k++) { const double v = wv_bcast(u, k); fv = __builtin_fma(v, v, fv); }
                    if (nsoft_act > 0)
                        for (int i = 0; i < na; i++)
                            if (__builtin_amdgcn_readlane(possoft, i)) {
                                const double l = wv_bcast(ls, i);
                                soft = __builtin_fma(l * l, P.rho_soft, soft);
                            }
                    soft_slack = soft;
                    fval = fv + soft;
                    if (fval > fbound) { flag = EXIT_INFEASIBLE; break; }
                    double Mu[MR];
#pragma unroll
                    for (int r = 0; r < MR; r++) Mu[r] = 0.0;
                    for (int k = 0; k < n; k++) {
                        const double v = wv_bcast(u, k);
#pragma unroll
                        for (int r = 0; r < MR; r++) {
                            const int j = lane + 64 * r;
                            if (j < m) Mu[r] = __builtin_fma(Mt[(size_t)k * m + j], v, Mu[r]);
                        }
                    }
                    double mval = -P.primal_tol;
                    int midx = -1;
                    bool broken = false;
#pragma unroll
                    for (int r = 0; r < MR; r++) {
                        const int j = lane + 64 * r;
                        if (j < m && !(sense[r] & SENSE_IMMUTABLE)) {
                            const double vu = (C[P.odu + j] + b[r]) - Mu[r];
                            const double vl = -((C[P.odl + j] + b[r]) - Mu[r]);
                            if (!((actb >> r) & 1u)) {
                                if (vu < mval) { mval = vu; midx = 2 * j; }
                                else if (vl < mval) { mval = vl; midx = 2 * j + 1; }
                            } else if (!(sense[r] & SENSE_SOFT) && (vu < -P.primal_tol || vl < -P.primal_tol)) {
                                broken = true;  // the iterate violates a hard row of its own working set
                            }
                        }
                    }
                    wv_argmin(mval, midx);
                    midx = __builtin_amdgcn_readfirstlane(midx);
                    if (midx < 0) {
                        if (__ballot(broken) != 0ull) flag = EXIT_CYCLE;
                        else flag = (soft_slack > P.primal_tol) ? EXIT_SOFT_OPTIMAL : EXIT_OPTIMAL;
                        break;
                    }
                    lam = ls;
                    ldl_add(midx >> 1, (midx & 1) != 0);
                    if (fval - best < P.progress_tol) {
                        if (++cyc > P.cycle_tol) { flag = EXIT_CYCLE; break; }
                    } else { best = fval; cyc = 0; }
                } else {
                    lam = __builtin_fma(alpha, ls - lam, lam);
                    ldl_remove(rm);
                }
            } else {
                // singular working set: direction p with M_W' p = 0, p_sing = +-1
                const int sg = sing;
                double acc = (lane < sg) ? -L[lane * ldc + sg] : 0.0;
                acc = sweep_bwd(acc, sg - 1);
                if (lane == sg) acc = 1.0;
                if (lane > sg) acc = 0.0;
                if (__builtin_amdgcn_readlane(poslow, sg)) acc = -acc;
                ls = acc;
                blocking(true, alpha, rm);
                if (rm < 0) { flag = EXIT_INFEASIBLE; break; }
                lam = __builtin_fma(alpha, ls, lam);
                ldl_remove(rm);
            }
            iter++;
        }
        };   // solve_node

        if (!BNB) {
            solve_node();
        } else {
            // depth-first branch and bound; stack entry d lives on lane d
            int stk_j = 0, stk_side = 0, stk_tried = 0;
            int depth = 0, nodes = 0, total_it = 0, have = 0, bflag = EXIT_INFEASIBLE;
            double ubest = 0.0, bestval = P.fval_bound;
            unsigned bestact = 0u, bestlow = 0u;
            for (;;) {
                if (nodes >= 100000) { bflag = EXIT_ITERLIMIT; break; }
#pragma unroll
                for (int r = 0; r < MR; r++) sense[r] = sense0[r];
                for (int d = 0; d < depth; d++) {
                    const int jf = __builtin_amdgcn_readlane(stk_j, d);
                    const int sd = __builtin_amdgcn_readlane(stk_side, d);
                    if (lane == (jf & 63)) {
#pragma unroll
                        for (int r = 0; r < MR; r++)
                            if (r == (jf >> 6)) sense[r] |= SENSE_ACTIVE | SENSE_IMMUTABLE | (sd ? SENSE_LOWER : 0);
                    }
                }
                fbound = bestval;
                solve_node();
                nodes++;
                total_it += iter;
                bool descend = false;
                if (flag >= 1) {
                    // lowest-index binary row that is not in the final working set
                    int cand = 0x7fffffff;
#pragma unroll
                    for (int r = MR - 1; r >= 0; r--) {
                        const int j = lane + 64 * r;
                        if (j < m && (sense0[r] & SENSE_BINARY) && !((actb >> r) & 1u)) cand = j;
                    }
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) {
                        const int o = __shfl_xor(cand, off);
                        cand = o < cand ? o : cand;
                    }
                    const int jb = __builtin_amdgcn_readfirstlane(cand);
                    if (jb == 0x7fffffff) {              // leaf: every binary sits on a bound
                        if (!have || fval < bestval) {
                            have = 1; bestval = fval; ubest = u; bestact = actb; bestlow = lowb;
                        }
                    } else {
                        double Mu = 0.0;
                        for (int k = 0; k < n; k++) Mu = __builtin_fma(Mr[(size_t)jb * n + k], wv_bcast(u, k), Mu);
                        double bj = 0.0;
#pragma unroll
                        for (int r = 0; r < MR; r++) if (r == (jb >> 6)) bj = b[r];
                        bj = wv_bcast(bj, jb & 63);
                        const double dlo = C[P.odl + jb] + bj, dup = C[P.odu + jb] + bj;
                        const int lower_first = (Mu - dlo) < (dup - Mu) ? 1 : 0;
                        if (lane == depth) { stk_j = jb; stk_side = lower_first; stk_tried = 1; }
                        depth++;
                        descend = true;
                    }
                }
                if (!descend) {                          // backtrack to the next untried side
                    while (depth > 0 && __builtin_amdgcn_readlane(stk_tried, depth - 1) == 2) depth--;
                    if (depth == 0) break;
                    if (lane == depth - 1) { stk_side ^= 1; stk_tried = 2; }
                }
            }
            u = have ? ubest : 0.0;
            actb = have ? bestact : 0u;
            lowb = have ? bestlow : 0u;
            flag = have ? (bflag == EXIT_ITERLIMIT ? EXIT_ITERLIMIT : EXIT_OPTIMAL) : bflag;
            iter = total_it;
        }

        // ---- x = R^-1 u + x0 + Xth theta   (mpc_update_qp.c:14-22); lane k writes output k
        {
            double xs = 0.0;
            for (int c = 0; c < n; c++) {
                const double v = wv_bcast(u, c);
                if (lane < P.nout) xs = __builtin_fma(C[P.oRout + lane * n + c], v, xs);
            }
            if (lane < P.nout) {
                double sh = C[P.ox0 + lane];
                for (int t = 0; t < nth; t++) sh = __builtin_fma(C[P.oXth + lane * nth + t], th[t], sh);
                X[pid * P.nout + lane] = xs + sh;
            }
        }
        if (active) {
            unsigned long long wd[2 * MR + 1];
#pragma unroll
            for (int q = 0; q < 2 * MR + 1; q++) wd[q] = 0ull;
#pragma unroll
            for (int r = 0; r < MR; r++) {
                const bool a = (actb >> r) & 1u, lo = (lowb >> r) & 1u;
                const unsigned long long up = __ballot(a && !lo), dn = __ballot(a && lo);
                wd[r] |= up;
                const int pos = m + 64 * r, q0 = pos >> 6, sft = pos & 63;
#pragma unroll
                for (int q = 0; q < 2 * MR + 1; q++) {
                    if (q == q0) wd[q] |= dn << sft;
                    if (q == q0 + 1 && sft) wd[q] |= dn >> (64 - sft);
                }
            }
            if (lane == 0)
#pragma unroll
                for (int q = 0; q < 2 * MR + 1; q++)
                    if (q < P.words) active[pid * P.words + q] = wd[q];
        }
        if (lane == 0) {
            exitflag[pid] = flag;
            if (iters) iters[pid] = iter;
        }
    }
}

}  // namespace lmpc
