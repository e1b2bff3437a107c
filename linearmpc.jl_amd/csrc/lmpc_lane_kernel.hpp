// One-QP-per-lane dual active-set kernel for small condensed-MPC problems (n <= 12, m <= 64, hard rows).
//
// Mapping (gfx950): every lane of a 64-wide wavefront owns one parameter point theta and runs the
// whole dual active-set iteration on it.  The LDL' factor, multipliers and working set of that
// lane live in VGPRs: every loop over working-set positions is fully unrolled against the
// compile-time capacity MA = N+1, so all register-array indices are static and positions beyond
// the lane's current |W| are kept at exact zeros (they then drop out of every fma chain).  What is
// shared by all problems is read uniformly: rows of M / Dth / bounds through the scalar cache
// (s_load, one SGPR operand per v_fma_f64), and the rows a lane needs by *its own* constraint
// index (M_W rows for the primal step, Gram entries for the new L row) from an LDS copy.
// The per-lane constraint shift b_j = Dth_j . theta is kept in LDS as [j][lane] (conflict-free for
// ds_read_b64: consecutive lanes hit consecutive bank pairs).  There is no cross-lane arithmetic,
// so the summation order is exactly the sequential order of the CPU oracle: results are
// bit-comparable, not just close.
//
// Replaces, per problem: mpc_update_qp (reference codegen/mpc_update_qp.c:1-10), daqp_ldp
// ([EXT] libdaqp, called at mpc_update_qp.c:48 / utils.jl:282) and mpc_get_solution
// (mpc_update_qp.c:14-22).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lmpc_pack.hpp"
#include "lmpc_wave_layout.hpp"
#include "lmpc_tiers.hpp"

#ifndef LMPC_LANE_WAVES
#define LMPC_LANE_WAVES 3   // wavefronts per SIMD the small lane kernels are register-budgeted for
#endif

namespace lmpc {



// Closed-loop tail of a finished problem (SimFuse): x+ = F x + G u with the sums in plant_kernel's /
// the oracle's order (F's terms, then G's), the next record [x+; r; u[0:nup]], bookkeeping of the run.
// th: this problem's current record; u: its nu outputs.  kMaxSimU bounds the outputs kept in registers.
constexpr int kMaxSimU = 4;
__device__ __forceinline__ void sim_advance(const SimFuse &S, const double *__restrict__ FGc,
                                            const double *__restrict__ th, const double *u,
                                            const long long pid, const int flag) {
    const int nx = S.nx, nu = S.nu, nr = S.nr, nup = S.nup;
    const int nth = nx + nr + nup;
    const double *F = FGc, *G = FGc + nx * nx;            // inside the constant pack: scalar loads
    double *to = S.theta_out + pid * nth;
    if (S.kstep != nullptr) {
        // scenario-asynchronous mode: in place (to == th), so the old state is taken into registers first
        // (nx <= 8, checked by the host); trajectories at this scenario's own step
        const int k = S.kstep[pid];
        double xo[8];
#pragma unroll
        for (int c = 0; c < 8; c++) xo[c] = c < nx ? th[c] : 0.0;
        double xn[8];
#pragma unroll
        for (int a = 0; a < 8; a++) {
            double acc = 0.0;
            if (a < nx) {
#pragma unroll
                for (int c = 0; c < 8; c++)
                    if (c < nx) acc = __builtin_fma(F[a * nx + c], xo[c], acc);
#pragma unroll
                for (int l = 0; l < kMaxSimU; l++)
                    if (l < nu) acc = __builtin_fma(G[a * nu + l], u[l], acc);
            }
            xn[a] = acc;
        }
#pragma unroll
        for (int a = 0; a < 8; a++)
            if (a < nx) {
                to[a] = xn[a];
                if (S.xtraj_base) S.xtraj_base[((long long)(k + 1) * S.nscen + pid) * nx + a] = xn[a];
            }
#pragma unroll
        for (int l = 0; l < kMaxSimU; l++) {
            if (l < nup) to[nx + nr + l] = u[l];
            if (l < nu && S.utraj) S.utraj[((long long)k * S.nscen + pid) * nu + l] = u[l];
        }
        if (S.flag_min) S.flag_min[pid] = k == 0 ? flag : (flag < S.flag_min[pid] ? flag : S.flag_min[pid]);
        S.kstep[pid] = k + 1;
        return;
    }
    for (int a = 0; a < nx; a++) {
        double acc = 0.0;
        for (int c = 0; c < nx; c++) acc = __builtin_fma(F[a * nx + c], th[c], acc);
#pragma unroll
        for (int l = 0; l < kMaxSimU; l++)
            if (l < nu) acc = __builtin_fma(G[a * nu + l], u[l], acc);
        to[a] = acc;
        if (S.xtraj) S.xtraj[pid * nx + a] = acc;
    }
    for (int k = 0; k < nr; k++) to[nx + k] = th[nx + k];
#pragma unroll
    for (int l = 0; l < kMaxSimU; l++)
        if (l < nup) to[nx + nr + l] = u[l];
    if (S.flag_min) S.flag_min[pid] = S.first ? flag : (flag < S.flag_min[pid] ? flag : S.flag_min[pid]);
}

// Solver state of one lane for a working-set capacity of MA rows.
template <int N, int MA> struct LaneState {
    static constexpr int NSL = MA * (MA - 1) / 2;
    double SL[NSL > 0 ? NSL : 1];
    double D[MA], Dinv[MA], lam[MA], ls[MA], rhs[MA], u[N];
    int WS[MA];
    unsigned long long act, low;
    int na, sing, iter, cyc, flag;
    double best, fval;
    bool done;

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < NSL; i++) SL[i] = 0.0;
#pragma unroll
        for (int i = 0; i < MA; i++) { D[i] = 0.0; Dinv[i] = 0.0; lam[i] = 0.0; ls[i] = 0.0; rhs[i] = 0.0; WS[i] = 0; }
#pragma unroll
        for (int k = 0; k < N; k++) u[k] = 0.0;
        act = 0ull; low = 0ull;
        na = 0; sing = -1; iter = 1; cyc = 0; flag = EXIT_ITERLIMIT;
        best = -1.0; fval = 0.0;
        done = false;
    }
    // continue a solve that outgrew the capacity MB < MA: same values, the extra positions at exact zeros
    template <int MB> __device__ __forceinline__ void extend_from(const LaneState<N, MB> &o) {
        static_assert(MB <= MA, "capacity");
        init();
#pragma unroll
        for (int i = 0; i < MB; i++) {
            D[i] = o.D[i]; Dinv[i] = o.Dinv[i]; lam[i] = o.lam[i]; ls[i] = o.ls[i]; rhs[i] = o.rhs[i]; WS[i] = o.WS[i];
#pragma unroll
            for (int t = 0; t < i; t++) SL[lmpc_sl(i, t)] = o.SL[lmpc_sl(i, t)];
        }
#pragma unroll
        for (int k = 0; k < N; k++) u[k] = o.u[k];
        act = o.act; low = o.low;
        na = o.na; sing = o.sing; iter = o.iter; cyc = o.cyc; flag = o.flag;
        best = o.best; fval = o.fval;
        done = o.done;
    }
};

// The dual active-set iterations of problem `pid` on this lane (no barriers inside), on the state `s`.
// sM/sG/sdu/sdl: LDS copies of the constant pack; sB: this block's b[j][lane] columns (filled by the caller).
// MS > 0: the number of constraints is the compile-time constant MS (the common box-constrained
// MPC, m == n): the scans over rows are fully unrolled, so all their scalar loads are issued in one
// batch instead of one round trip per row.  MS == 0: m is a run-time value.
// MA: working-set capacity.  N+1 in general (n independent rows plus the one dependent row of a
// singular working set).  A problem whose constraints are all simple bounds (ms == m == n) can
// never hold more than n rows -- one per variable -- and those rows (rows of R^-1) are independent,
// so MA = N is enough there: one position less in every unrolled loop and ~22 VGPRs less.
// TIER: MA is a FIRST-TIER capacity below the full one (cold starts only): every unrolled loop over
// working-set positions is that much shorter; when the working set wants to outgrow it the function
// returns true BEFORE touching the state, and the caller continues the same solve in place on a
// full-capacity state (extend_from) -- the interrupted iteration is recomputed there from identical
// values, so the result does not change by a bit.  Most problems of an MPC batch end with a handful
// of active rows, far below n.
template <int N, int MS, int MA, bool TIER>
__device__ __forceinline__ bool lane_loop(
    const PackLayout &P, const double *__restrict__ C, const double *sM, const double *sG,
    const double *sdu, const double *sdl, double *sB, const int B, const int tid, const long long pid,
    const uint64_t *__restrict__ warm, LaneState<N, MA> &s) {
    static_assert(TIER || MA == N + 1 || MA == N, "capacity");
    const int m = MS > 0 ? MS : P.m;
    constexpr int NSLA = (MA * (MA - 1) / 2 > 0) ? MA * (MA - 1) / 2 : 1;
    double (&SL)[NSLA] = s.SL;
    double (&D)[MA] = s.D, (&Dinv)[MA] = s.Dinv, (&lam)[MA] = s.lam, (&ls)[MA] = s.ls, (&rhs)[MA] = s.rhs;
    double (&u)[N] = s.u;
    int (&WS)[MA] = s.WS;
    unsigned long long &act = s.act, &low = s.low;
    int &na = s.na, &sing = s.sing, &iter = s.iter, &cyc = s.cyc, &flag = s.flag;
    double &best = s.best, &fval = s.fval;
    bool &done = s.done;
    // ---- append constraint j (at its lower bound if `lower`) to the working set
    auto ldl_add = [&](int j, bool lower) {
        double row[MA > 1 ? MA - 1 : 1];
#pragma unroll
        for (int t = 0; t < MA - 1; t++) {
            row[t] = 0.0;
            if (MA <= 6 || t < na) {                             // small capacities: every position, no branch
                const int a = WS[t];                             // 0 beyond |W|: a valid entry, dropped by the select
                const double g = sG[a >= j ? lmpc_tri(a) + j : lmpc_tri(j) + a];
                row[t] = (t < na) ? g : 0.0;
            }
        }
        double dnew = sG[lmpc_tri(j) + j];
#pragma unroll
        for (int i = 1; i < MA - 1; i++) {
            double acc = row[i];
#pragma unroll
            for (int t = 0; t < i; t++) acc = __builtin_fma(-SL[lmpc_sl(i, t)], row[t], acc);
            row[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < MA - 1; i++) {
            const double q = row[i];
            const double l = q * Dinv[i];
            row[i] = l;
            dnew = __builtin_fma(-l, q, dnew);
        }
        const bool singular = (dnew < P.zero_tol) || (na >= P.n);
        const double b = sB[j * B + tid];
        const double r = lower ? -(sdl[j] + b) : -(sdu[j] + b);
        const double dval = singular ? 0.0 : dnew;
        const double dinv = singular ? 0.0 : 1.0 / dnew;
#pragma unroll
        for (int i = 0; i < MA; i++) {
            if (i == na) {
                WS[i] = j; rhs[i] = r; lam[i] = 0.0; ls[i] = 0.0; D[i] = dval; Dinv[i] = dinv;
#pragma unroll
                for (int t = 0; t < i; t++) SL[lmpc_sl(i, t)] = row[t];
            }
        }
        if (singular) sing = na;
        act |= 1ull << j;
        if (lower) low |= 1ull << j;
        na++;
    };

    // ---- drop working-set position r (rank-one update of the trailing block)
    auto ldl_remove = [&](int r) {
        double wv[MA];
        double alpha = 0.0;
        int jrem = 0;
#pragma unroll
        for (int i = 0; i < MA; i++) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < i; c++) s = (c == r) ? SL[lmpc_sl(i, c)] : s;
            wv[i] = s;
            if (i == r) { alpha = D[i]; jrem = WS[i]; }
        }
#pragma unroll
        for (int i = 1; i < MA - 1; i++) {
#pragma unroll
            for (int c = 0; c < i; c++) {
                // read all three unconditionally: a select of VALUES keeps SL in registers, a
                // select of addresses (what phi-of-loads folds to) would push it to scratch
                const double va = SL[lmpc_sl(i + 1, c)], vb = SL[lmpc_sl(i + 1, c + 1)];
                const double keep = SL[lmpc_sl(i, c)];
                const double sh = (c < r) ? va : vb;
                SL[lmpc_sl(i, c)] = (i >= r) ? sh : keep;
            }
        }
#pragma unroll
        for (int c = 0; c < MA - 1; c++) SL[lmpc_sl(MA - 1, c)] = 0.0;
#pragma unroll
        for (int i = 0; i < MA - 1; i++) {
            if (i >= r) {
                WS[i] = WS[i + 1]; lam[i] = lam[i + 1]; rhs[i] = rhs[i + 1];
                D[i] = D[i + 1]; Dinv[i] = Dinv[i + 1];
            }
        }
        WS[MA - 1] = 0; lam[MA - 1] = 0.0; rhs[MA - 1] = 0.0; D[MA - 1] = 0.0; Dinv[MA - 1] = 0.0;
        na--;
        sing = -1;
        bool stop = false;
#pragma unroll
        for (int i = 0; i < MA - 1; i++) {
            if (i >= r && i < na && !stop) {
                const double pt = wv[i + 1];
                const double dold = D[i];
                const double dbar = __builtin_fma(alpha * pt, pt, dold);
                if (dbar < P.zero_tol) {
                    D[i] = 0.0; Dinv[i] = 0.0; sing = i; stop = true;
                } else {
                    const double rinv = 1.0 / dbar;
                    const double beta = (pt * alpha) * rinv;
                    alpha = (dold * alpha) * rinv;
                    D[i] = dbar; Dinv[i] = rinv;
#pragma unroll
                    for (int q = i + 1; q < MA - 1; q++) {
                        wv[q + 1] = __builtin_fma(-pt, SL[lmpc_sl(q, i)], wv[q + 1]);
                        SL[lmpc_sl(q, i)] = __builtin_fma(beta, wv[q + 1], SL[lmpc_sl(q, i)]);
                    }
                }
            }
        }
        act &= ~(1ull << jrem);
        low &= ~(1ull << jrem);
    };

    // ---- initial working set: equality rows, then the caller's warm-start mask
    if (!TIER && (P.eq_mask != 0ull || warm != nullptr)) {
        unsigned long long wup = 0ull, wlo = 0ull;
        if (warm != nullptr) {
            const uint64_t *wp = warm + pid * P.words;
            const unsigned long long w0 = wp[0], w1 = P.words > 1 ? wp[1] : 0ull;
            const unsigned long long mm = m >= 64 ? ~0ull : ((1ull << m) - 1ull);
            wup = w0 & mm;
            wlo = (m >= 64 ? w1 : ((w0 >> m) | (m > 0 ? (w1 << (64 - m)) : 0ull))) & mm;
            wup &= ~P.imm_mask; wlo &= ~P.imm_mask;
        }
        for (int j = 0; j < m && !done; j++) {
            const bool iseq = (P.eq_mask >> j) & 1ull;
            const bool up = (wup >> j) & 1ull, lo = ((wlo >> j) & 1ull) && !up;
            if ((iseq || up || lo) && !(MA < N + 1 && na >= MA)) {
                ldl_add(j, lo);
                if (sing >= 0) {
                    if ((P.imm_mask >> j) & 1ull) { flag = EXIT_OVERDETERMINED; done = true; }
                    else {      // dependent warm-start row: take it out again
                        na--;
                        sing = -1;
#pragma unroll
                        for (int i = 0; i < MA; i++) {
                            if (i == na) {
                                WS[i] = 0; rhs[i] = 0.0; D[i] = 0.0; Dinv[i] = 0.0;
#pragma unroll
                                for (int t = 0; t < i; t++) SL[lmpc_sl(i, t)] = 0.0;
                            }
                        }
                        act &= ~(1ull << j);
                        low &= ~(1ull << j);
                    }
                }
            }
        }
    }

    // ---- dual active-set iterations
    while (!done) {
        if (iter >= P.iter_limit) { flag = EXIT_ITERLIMIT; break; }
        int rm = -1;
        double alpha = 0.0;
        if (sing < 0) {
            // constrained stationary point: (L D L') lam* = rhs
            double xl[MA];
#pragma unroll
            for (int i = 0; i < MA; i++) {
                double acc = rhs[i];
#pragma unroll
                for (int t = 0; t < i; t++) acc = __builtin_fma(-SL[lmpc_sl(i, t)], xl[t], acc);
                xl[i] = acc;
            }
#pragma unroll
            for (int i = 0; i < MA; i++) xl[i] = xl[i] * Dinv[i];
#pragma unroll
            for (int i = MA - 1; i >= 0; i--) {
                double acc = xl[i];
#pragma unroll
                for (int t = MA - 1; t > i; t--) acc = __builtin_fma(-SL[lmpc_sl(t, i)], ls[t], acc);
                ls[i] = acc;
            }
#pragma unroll
            for (int i = 0; i < MA; i++) {
                if (i < na) {
                    const int j = WS[i];
                    const bool imm = (P.imm_mask >> j) & 1ull;
                    const bool isl = (low >> j) & 1ull;
                    const bool ok = isl ? (ls[i] < P.dual_tol) : (ls[i] > -P.dual_tol);
                    if (!imm && !ok) {
                        const double cand = -lam[i] / (ls[i] - lam[i]);
                        if (rm < 0 || cand < alpha) { alpha = cand; rm = i; }
                    }
                }
            }
            if (rm < 0) {
                // primal iterate u = -M_W' lam*, then the most violated inactive constraint
#pragma unroll
                for (int k = 0; k < N; k++) u[k] = 0.0;
                // every position, no branch: beyond |W| the row index is 0 (a valid row) and the
                // multiplier an exact zero, so the term adds nothing -- and the LDS reads of all
                // positions are in flight together
                // (small capacities only: with many positions skipping the empty ones wins)
#pragma unroll
                for (int i = 0; i < MA; i++) {
                    if (MA <= 6 || i < na) {
                        const double *mi = sM + WS[i] * N;
                        const double l = (i < na) ? ls[i] : 0.0;
#pragma unroll
                        for (int k = 0; k < N; k++) u[k] = __builtin_fma(-mi[k], l, u[k]);
                    }
                }
                fval = 0.0;
#pragma unroll
                for (int k = 0; k < N; k++) fval = __builtin_fma(u[k], u[k], fval);
                if (fval > P.fval_bound) { flag = EXIT_INFEASIBLE; break; }
                double min_val = -P.primal_tol;
                int add = -1;
                bool addlow = false, broken = false;
                auto scan_row = [&](int j) {
                    // an IMMUTABLE row is never a candidate: unrolled scan -- folded into the selects, so the
                    // rows' LDS reads stay in one block; run-time row loop -- a uniform branch around the row
                    const bool imm = (P.imm_mask >> j) & 1ull;
                    if ((MS > 0 && MA <= 6) || !imm) {
                        double Mu = 0.0;
                        // unrolled scan (MS > 0): rows and bounds as LDS broadcast reads (uniform address,
                        // in-order return, +2.5 % on the headline batch over scalar loads, whose SGPR
                        // operands cannot stay resident between iterations); run-time row count: scalar loads
                        const double *mrow = MS > 0 ? sM + j * N : C + P.oM + j * N;
#pragma unroll
                        for (int k = 0; k < N; k++) Mu = __builtin_fma(mrow[k], u[k], Mu);
                        const double b = sB[j * B + tid];
                        const double duj = MS > 0 ? sdu[j] : C[P.odu + j], dlj = MS > 0 ? sdl[j] : C[P.odl + j];
                        const double vu = (duj + b) - Mu;
                        const double vl = -((dlj + b) - Mu);
                        if (imm) {
                        } else if (!((act >> j) & 1ull)) {
                            if (vu < min_val) { add = j; addlow = false; min_val = vu; }
                            else if (vl < min_val) { add = j; addlow = true; min_val = vl; }
                        } else if (vu < -P.primal_tol || vl < -P.primal_tol) {
                            broken = true;      // the iterate violates a row of its own working set
                        }
                    }
                };
                if constexpr (MS > 0) {
#pragma unroll
                    for (int j = 0; j < MS; j++) scan_row(j);
                } else {
                    for (int j = 0; j < m; j++) scan_row(j);
                }
                if (add < 0) { flag = broken ? EXIT_CYCLE : EXIT_OPTIMAL; break; }
                if (TIER && na >= MA) return true;      // outgrown the first tier: nothing has been touched yet
                if (!TIER && MA < N + 1 && na >= MA) { flag = EXIT_CYCLE; break; }   // cannot happen for pure bounds
#pragma unroll
                for (int i = 0; i < MA; i++) lam[i] = ls[i];
                ldl_add(add, addlow);
                if (fval - best < P.progress_tol) {
                    if (++cyc > P.cycle_tol) { flag = EXIT_CYCLE; break; }
                } else { best = fval; cyc = 0; }
            } else {
#pragma unroll
                for (int i = 0; i < MA; i++) lam[i] = __builtin_fma(alpha, ls[i] - lam[i], lam[i]);
                ldl_remove(rm);
            }
        } else {
            // singular working set: direction p with M_W' p = 0, p_sing = +-1
            double lrow[MA], p[MA];
            int js = 0;
#pragma unroll
            for (int c = 0; c < MA; c++) { lrow[c] = 0.0; p[c] = 0.0; }
#pragma unroll
            for (int i = 0; i < MA; i++) {
                if (i == sing) {
                    js = WS[i];
#pragma unroll
                    for (int c = 0; c < i; c++) lrow[c] = SL[lmpc_sl(i, c)];
                }
            }
#pragma unroll
            for (int i = MA - 2; i >= 0; i--) {
                if (i < sing) {
                    double acc = -lrow[i];
#pragma unroll
                    for (int t = MA - 2; t > i; t--) acc = __builtin_fma(-SL[lmpc_sl(t, i)], p[t], acc);
                    p[i] = acc;
                }
            }
            const bool slow = (low >> js) & 1ull;
#pragma unroll
            for (int i = 0; i < MA; i++) {
                if (i == sing) p[i] = 1.0;
                if (slow) p[i] = -p[i];
                ls[i] = p[i];
            }
#pragma unroll
            for (int i = 0; i < MA; i++) {
                if (i < na) {
                    const int j = WS[i];
                    const bool imm = (P.imm_mask >> j) & 1ull;
                    const bool isl = (low >> j) & 1ull;
                    const bool ok = isl ? (ls[i] < P.dual_tol) : (ls[i] > -P.dual_tol);
                    if (!imm && !ok) {
                        const double cand = -lam[i] / ls[i];
                        if (rm < 0 || cand < alpha) { alpha = cand; rm = i; }
                    }
                }
            }
            if (rm < 0) { flag = EXIT_INFEASIBLE; break; }
#pragma unroll
            for (int i = 0; i < MA; i++) lam[i] = __builtin_fma(alpha, ls[i], lam[i]);
            ldl_remove(rm);
        }
        iter++;
    }

    return false;
}

// first-tier capacity of an instantiation (0 = none): the boxed variants from n = 4 on
#ifndef LMPC_LANE_TIER_CAP
#define LMPC_LANE_TIER_CAP 3
#endif
template <int N, int MS, int MA> struct lane_tier { static constexpr int value = (MS > 0 && N >= 4) ? LMPC_LANE_TIER_CAP : 0; };

// Whole solve of problem `pid` on this lane: b = Dth theta into LDS, the iterations, the outputs.
// MULTI: instantiation for several outputs per problem (compute_control_trajectory): the epilogue reads the
// record back ONCE into registers and runs one chain per output (small instantiations, plain solve only; in
// the shared kernel the same code cost the single-output path 7 %).
template <int N, int MS, int MA, bool SIM, bool MULTI>
__device__ __forceinline__ void lane_solve(
    const PackLayout &P, const double *__restrict__ C, const double *sM, const double *sG,
    const double *sdu, const double *sdl, double *sB, const int B, const int tid, const long long pid,
    const double *__restrict__ theta, double *__restrict__ X, int32_t *__restrict__ exitflag,
    int32_t *__restrict__ iters, uint64_t *__restrict__ active, const uint64_t *__restrict__ warm, const int tier) {
    static_assert(MA == N + 1 || MA == N, "capacity");
    const int m = MS > 0 ? MS : P.m, nth = P.nth;
    const bool one_out = P.nout == 1;
    double sh0 = one_out ? C[P.ox0] : 0.0;      // x0 + Xth theta of the first output, built on the fly
    const double *th = theta + pid * nth;

    // b_j = Dth_j . theta   (mpc_update_qp.c:5-6): theta is pulled in four values at a time
    // (loads issued back to back), the running sums live in this lane's LDS column; per row the
    // products are still added in ascending t, as the oracle does.
    // nth <= 32: the rows come from the zero-padded copies the screening kernel uses (DthP, XthP: rows of
    // nthp columns, a multiple of four) -- no guard on any term, so the scalar loads of a chunk are
    // issued together instead of one exposed round trip per fma; theta's index is clamped into the
    // record and meets a zero coefficient there (the padded terms add +0 exactly)
    if (MA <= 6 && nth <= 32 && nth > 0) {     // (the register-bound large instantiations keep the guarded form)
        const int nthp = P.nthp;
        for (int t0 = 0; t0 < nthp; t0 += 4) {
            double tv[4];
#pragma unroll
            for (int q = 0; q < 4; q++) tv[q] = th[t0 + q < nth ? t0 + q : nth - 1];
            auto brow = [&](int j) {
                double acc = t0 ? sB[j * B + tid] : 0.0;
                const double *dj = C + P.oDthP + j * nthp + t0;
#pragma unroll
                for (int q = 0; q < 4; q++) acc = __builtin_fma(dj[q], tv[q], acc);
                sB[j * B + tid] = acc;
            };
            if constexpr (MS > 0) {
#pragma unroll
                for (int j = 0; j < MS; j++) brow(j);
            } else {
                for (int j = 0; j < m; j++) brow(j);
            }
            if (one_out) {
                const double *xk = C + P.oXthP + t0;
#pragma unroll
                for (int q = 0; q < 4; q++) sh0 = __builtin_fma(xk[q], tv[q], sh0);
            }
        }
    } else {
    for (int t0 = 0; t0 < nth; t0 += 4) {
        double tv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) tv[q] = (t0 + q < nth) ? th[t0 + q] : 0.0;
        auto brow = [&](int j) {
            double acc = t0 ? sB[j * B + tid] : 0.0;
            const double *dj = C + P.oDth + j * nth + t0;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (t0 + q < nth) acc = __builtin_fma(dj[q], tv[q], acc);
            sB[j * B + tid] = acc;
        };
        if constexpr (MS > 0) {
#pragma unroll
            for (int j = 0; j < MS; j++) brow(j);
        } else {
            for (int j = 0; j < m; j++) brow(j);
        }
        if (one_out) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (t0 + q < nth) sh0 = __builtin_fma(C[P.oXth + t0 + q], tv[q], sh0);
        }
    }
    }
    if (nth == 0)
        for (int j = 0; j < m; j++) sB[j * B + tid] = 0.0;

    double u[N];
    int flag = EXIT_ITERLIMIT, iter = 1;
    unsigned long long act = 0ull, low = 0ull;
    bool solved = false;
    // Small boxed problems, cold start (tier == 2: the host has checked the settings and the row flags):
    // the straight-line tiers of lmpc_tiers.hpp first.  They finish all but a handful of an MPC batch's
    // problems; only a wavefront that holds one of the others runs the generic loop below (on those lanes).
    if constexpr (MS > 0 && MS == N && N <= 6) {
        if (tier >= 2 && warm == nullptr) {
            constexpr int KMAX = LMPC_FAST_KMAX < N ? LMPC_FAST_KMAX : N;
            double bb[N];
#pragma unroll
            for (int j = 0; j < N; j++) bb[j] = sB[j * B + tid];
            int wrow[KMAX], nact = 0, it1 = 1;
            bool wlow[KMAX];
            const int res = fast_tiers<N, KMAX>(P, sM, sG, sdu, sdl, true, bb, u, it1, wrow, wlow, nact);
            if (res == EXIT_OPTIMAL) {
                solved = true; flag = EXIT_OPTIMAL; iter = it1;
#pragma unroll
                for (int i = 0; i < KMAX; i++)
                    if (i < nact) { act |= 1ull << wrow[i]; if (wlow[i]) low |= 1ull << wrow[i]; }
            }
        }
    }
    if (!solved) {
        LaneState<N, MA> s;
        constexpr int MT = lane_tier<N, MS, MA>::value;
        bool full = true;
        if constexpr (MT > 0) {
            if (tier && warm == nullptr && P.eq_mask == 0ull) {
                LaneState<N, MT> s1;
                s1.init();
                full = lane_loop<N, MS, MT, true>(P, C, sM, sG, sdu, sdl, sB, B, tid, pid, nullptr, s1);
                s.extend_from(s1);
            } else {
                s.init();
            }
        } else {
            s.init();
        }
        if (full) lane_loop<N, MS, MA, false>(P, C, sM, sG, sdu, sdl, sB, B, tid, pid, warm, s);
#pragma unroll
        for (int c = 0; c < N; c++) u[c] = s.u[c];
        flag = s.flag; iter = s.iter; act = s.act; low = s.low;
    }

    // ---- x = R^-1 u + x0 + Xth theta   (mpc_update_qp.c:14-22)
    double uo[kMaxSimU];
#pragma unroll
    for (int l = 0; l < kMaxSimU; l++) uo[l] = 0.0;
    if (one_out) {
        double xs = 0.0;
#pragma unroll
        for (int c = 0; c < N; c++) xs = __builtin_fma(C[P.oRout + c], u[c], xs);
        uo[0] = xs + sh0;
        if (!SIM || X != nullptr) X[pid] = uo[0];
    } else if (MULTI && nth <= 16 && nth > 0) {
        double tv[16];
#pragma unroll
        for (int t = 0; t < 16; t++) tv[t] = th[t < nth ? t : nth - 1];     // clamped entries meet no coefficient
        for (int k = 0; k < P.nout; k++) {
            double xs = 0.0, sh = C[P.ox0 + k];
#pragma unroll
            for (int c = 0; c < N; c++) xs = __builtin_fma(C[P.oRout + k * N + c], u[c], xs);
#pragma unroll
            for (int t = 0; t < 16; t++)
                if (t < nth) sh = __builtin_fma(C[P.oXth + k * nth + t], tv[t], sh);
            X[pid * P.nout + k] = xs + sh;
        }
    } else {
        for (int k = 0; k < P.nout; k++) {
            double xs = 0.0, sh = C[P.ox0 + k];
#pragma unroll
            for (int c = 0; c < N; c++) xs = __builtin_fma(C[P.oRout + k * N + c], u[c], xs);
            for (int t = 0; t < nth; t++) sh = __builtin_fma(C[P.oXth + k * nth + t], th[t], sh);
            const double xo = xs + sh;
#pragma unroll
            for (int l = 0; l < kMaxSimU; l++) if (SIM && l == k) uo[l] = xo;
            if (!SIM || X != nullptr) X[pid * P.nout + k] = xo;
        }
    }
    if constexpr (SIM) sim_advance(P.sim, C + P.oFG, th, uo, pid, flag);
    if (!SIM || exitflag != nullptr) exitflag[pid] = flag;
    if (iters) iters[pid] = iter;
    if (active) {
        const unsigned long long up = act & ~low, lo = act & low;
        unsigned long long w0 = up, w1 = 0ull;
        if (m < 64) { w0 |= lo << m; if (m > 0) w1 = lo >> (64 - m); }
        else w1 = lo;
        active[pid * P.words] = w0;
        if (P.words > 1) active[pid * P.words + 1] = w1;
    }
}

template <int N, int MS, int MA, bool SIM, bool MULTI>
__global__ __launch_bounds__(256, (N <= 5 ? LMPC_LANE_WAVES : 1)) void lane_kernel(
    const PackLayout P, const double *__restrict__ C, const double *__restrict__ theta,
    double *__restrict__ X, int32_t *__restrict__ exitflag, int32_t *__restrict__ iters,
    uint64_t *__restrict__ active, const uint64_t *__restrict__ warm,
    const int32_t *__restrict__ list, const int32_t *__restrict__ count, int32_t *__restrict__ count_next,
    long long seg_cap, int nshards, long long nprob, int tier) {
    extern __shared__ __align__(16) double lds[];

    const int m = P.m, B = blockDim.x, tid = threadIdx.x;
    // `list` (from screen_kernel) holds the problems that need iterations, in `nshards` segments of
    // capacity seg_cap with one counter each; block b works on segment b % nshards.  Without a list
    // the kernel walks the whole batch.  Blocks stride over the work so a fixed grid covers any count.
    // the counter set the next call's screening pass will use is cleared here (saves a memset
    // node per call; this call's screening pass finished before this kernel started)
    if (count_next && blockIdx.x == 0 && (int)threadIdx.x < nshards) count_next[threadIdx.x * kCountStride] = 0;
    const int shard = list ? (int)(blockIdx.x % nshards) : 0;
    const long long first = (list ? (long long)(blockIdx.x / nshards) : (long long)blockIdx.x) * B;
    const long long stride = (list ? (long long)(gridDim.x / nshards) : (long long)gridDim.x) * B;
    // Small instantiations: everything the first problem needs that does not depend on anything else
    // is requested in ONE round trip -- the segment's count, this lane's first list entry
    // (speculatively: the segment is allocated up to seg_cap, an entry at or beyond the count is simply
    // not used) and the constants for the LDS copy (M, G, du0, dl0 follow each other in the pack in the
    // order of the LDS copy: one loop).  The register-bound instantiations (one wavefront per SIMD)
    // keep the plain order: one more live register costs them more than the two round trips.
    constexpr bool kEarly = MA <= 6;
    if (list) list += (long long)shard * seg_cap;
    int32_t pid0 = 0;
    if (kEarly && list) {
        const long long i0 = first + tid;
        pid0 = list[i0 < seg_cap ? i0 : seg_cap - 1];
    }
    const long long cnt = list ? (long long)count[shard * kCountStride] : nprob;
    if (!kEarly && first >= cnt) return;
    double *sM = lds;                    // m x N   rows by per-lane constraint index
    double *sG = sM + m * N;             // packed lower triangle of M M'
    double *sdu = sG + lmpc_tri(m);      // du0
    double *sdl = sdu + m;               // dl0
    double *sB = sdl + m;                // b[j][lane]
    const int nconst = m * N + lmpc_tri(m) + 2 * m;
    for (int i = tid; i < nconst; i += B) lds[i] = C[P.oM + i];
    if (kEarly && first >= cnt) return;
    __syncthreads();

  for (long long base = first; base < cnt; base += stride) {
    const long long idx = base + tid;
    if (idx >= cnt) continue;
    const long long pid = list ? ((kEarly && base == first) ? (long long)pid0 : (long long)list[idx]) : idx;
    lane_solve<N, MS, MA, SIM, MULTI>(P, C, sM, sG, sdu, sdl, sB, B, tid, pid, theta, X, exitflag, iters, active, warm, tier);
}   // chunk loop
}

}  // namespace lmpc
