// Straight-line dual active-set iterations for the first rows of a small box-constrained problem's working set
// (shared by lane_kernel's first tier and fast_kernel).
//
// 99.9999 % of the problems of an MPC batch that need iterations at all follow the same short path: rows are
// appended one after the other (no removal, no singular pivot) and the solve ends with a handful of active rows.
// On that path every working-set position is a compile-time constant, so it is written as straight-line code:
//
//   tier k (k + 1 rows in the working set, k = 0 .. KMAX-1):  append the most violated row (new L row, pivot),
//   constrained stationary point on the factor, dual feasibility test, u = -M_W' lam*, the scan of all rows.
//
// Each chain is the fma chain of lane_loop / the CPU oracle in the same order (positions beyond |W| hold exact
// zeros there and drop out of every chain), so a problem finished here has the bits the generic loop would give
// it -- at about a third of its instructions (no register arrays with data-dependent positions emulated by
// selects).  Anything else -- a blocking multiplier (removal), a singular pivot, a row violated inside its own
// working set, a working set that wants more than KMAX rows, an objective above fval_bound -- is NOT finished
// here: the caller runs the generic loop on it from scratch.  The caller guarantees: m == n == N rows (simple
// bounds), no IMMUTABLE / ACTIVE-flagged rows, cold start, iter_limit > KMAX + 1, cycle_tol >= KMAX + 1 (then
// none of the generic loop's guards can fire inside these iterations).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lmpc_pack.hpp"
#include "lmpc_wave_layout.hpp"

// rows the tiers go up to (capped at n).  5: on the headline batch (n = 5) a problem that wants a fourth or
// fifth row is rare (1.2 % / 1e-6), but ONE of them in the generic loop is a 18 us latency chain at the end of
// a 25 us kernel; with all five tiers only removals and singular pivots leave the straight-line path.
#ifndef LMPC_FAST_KMAX
#define LMPC_FAST_KMAX 5
#endif

namespace lmpc {

__host__ __device__ constexpr int lmpc_tri(int i) { return i * (i + 1) / 2; }
// strict lower triangle, row i > col t
__host__ __device__ constexpr int lmpc_sl(int i, int t) { return i * (i - 1) / 2 + t; }

// straight-line tiers on one problem per lane (`mine`: this lane holds a problem).  Returns EXIT_OPTIMAL or
// 0 = not finished here (the generic loop takes it).  On success u, iter and the working set (wrow / wlow,
// nact rows) are set.
// Written WITHOUT per-lane branches: every lane computes every tier (until no lane of the wavefront is running
// any more -- a uniform branch), a lane that has finished or dropped out computes values nobody reads, and
// results are committed by selects.  (With `if (running) { ... }` blocks the compiler built nested exec-mask
// regions: 150 branches, saved masks spilling out of the scalar registers through v_writelane / v_readlane,
// 1000 moves -- a tier pass of 1500 instructions took 5 us on its own.)
// LEN doubles from base[START ..] with the widest LDS reads the alignment allows (base 16-byte aligned: an odd START
// takes one 8-byte read first, then 16-byte pairs).  ds_read_b128 costs 4 LDS cycles per 16 bytes a lane, the
// ds_read2_b64 the compiler picks for pairs of plain doubles 8 -- and one LDS pipe serves the four SIMDs of a CU.
template <int LEN>
__device__ __forceinline__ void lmpc_lds_run(const double *base, const int start, double (&dst)[LEN]) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const int head = start & 1;                       // (`start` is a constant wherever this is inlined)
    if (head) dst[0] = base[start];
#pragma unroll
    for (int q = 0; q < (LEN + 1) / 2; q++) {
        const int i = head + 2 * q;
        if (i + 1 < LEN) {
            const v2d v = *reinterpret_cast<const v2d *>(base + start + i);
            dst[i] = v.x; dst[i + 1] = v.y;
        } else if (i < LEN) {
            dst[i] = base[start + i];
        }
    }
}

template <int N, int KMAX>
__device__ __forceinline__ int fast_tiers(const PackLayout &P, const double *sM, const double *sG, const double *sdu,
                                          const double *sdl, const bool mine, const double (&b)[N], double (&u)[N],
                                          int &iter, int (&wrow)[KMAX], bool (&wlow)[KMAX], int &nact) {
    const double ptol = P.primal_tol, dtol = P.dual_tol, ztol = P.zero_tol, fbound = P.fval_bound;
    // ---- iteration 1 (empty working set, u = 0): the most violated row, as lane_loop's scan finds it
    double min_val = -ptol;
    int add = -1;
    bool addlow = false;
#pragma unroll
    for (int j = 0; j < N; j++) {
        const double vu = (sdu[j] + b[j]) - 0.0;
        const double vl = -((sdl[j] + b[j]) - 0.0);
        const bool tu = vu < min_val;
        const bool tl = !tu && (vl < min_val);
        add = (tu || tl) ? j : add;
        addlow = tu ? false : (tl ? true : addlow);
        min_val = tu ? vu : (tl ? vl : min_val);
    }
    double SL[KMAX * (KMAX - 1) / 2 > 0 ? KMAX * (KMAX - 1) / 2 : 1], D[KMAX], Dinv[KMAX], rhs[KMAX], ls[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; i++) { D[i] = 0.0; Dinv[i] = 0.0; rhs[i] = 0.0; ls[i] = 0.0; wrow[i] = 0; wlow[i] = false; }
#pragma unroll
    for (int c = 0; c < N; c++) u[c] = 0.0;
    int result = 0;
    bool running = mine && add >= 0;       // (a queued problem always has a violated row)
    iter = 1; nact = 0;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        // ---- append row `add` at position k (lane_loop's ldl_add with na = k)
        const int j = add < 0 ? 0 : add;
        double row[KMAX > 1 ? KMAX - 1 : 1];
#pragma unroll
        for (int t = 0; t < k; t++) {
            const int a = wrow[t];
            row[t] = sG[a >= j ? lmpc_tri(a) + j : lmpc_tri(j) + a];
        }
        double dnew = sG[lmpc_tri(j) + j];
#pragma unroll
        for (int i = 1; i < k; i++) {
            double acc = row[i];
#pragma unroll
            for (int t = 0; t < i; t++) acc = __builtin_fma(-SL[lmpc_sl(i, t)], row[t], acc);
            row[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < k; i++) {
            const double q = row[i];
            const double l = q * Dinv[i];
            row[i] = l;
            dnew = __builtin_fma(-l, q, dnew);
        }
        running = running && !(dnew < ztol);                   // singular working set: generic loop
        double bsel = 0.0;
#pragma unroll
        for (int q = 0; q < N; q++) bsel = (q == j) ? b[q] : bsel;
        wrow[k] = j; wlow[k] = addlow;
        rhs[k] = addlow ? -(sdl[j] + bsel) : -(sdu[j] + bsel);
        D[k] = dnew; Dinv[k] = 1.0 / dnew;
#pragma unroll
        for (int t = 0; t < k; t++) SL[lmpc_sl(k, t)] = row[t];
        if (!__any(running)) break;
        // ---- iteration k + 2 on the working set of k + 1 rows
        // (the rows of M and the bounds are re-read from LDS in every tier: hoisted out of the caller's loop they
        // would sit in ~70 registers for the whole kernel; the empty asm hides that the addresses repeat)
        int ofs = 0;
        asm volatile("" : "+s"(ofs));
        const double *sMk = sM + ofs, *sduk = sdu + ofs, *sdlk = sdl + ofs;
        (void)sduk; (void)sdlk;
        const int na = k + 1;
        double xl[KMAX];
#pragma unroll
        for (int i = 0; i < na; i++) {
            double acc = rhs[i];
#pragma unroll
            for (int t = 0; t < i; t++) acc = __builtin_fma(-SL[lmpc_sl(i, t)], xl[t], acc);
            xl[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < na; i++) xl[i] = xl[i] * Dinv[i];
#pragma unroll
        for (int i = na - 1; i >= 0; i--) {
            double acc = xl[i];
#pragma unroll
            for (int t = na - 1; t > i; t--) acc = __builtin_fma(-SL[lmpc_sl(t, i)], ls[t], acc);
            ls[i] = acc;
        }
        bool blocked = false;
#pragma unroll
        for (int i = 0; i < na; i++) {
            const bool ok = wlow[i] ? (ls[i] < dtol) : (ls[i] > -dtol);
            blocked = blocked || !ok;
        }
        running = running && !blocked;                         // a removal: generic loop
        // primal iterate and objective
        double uu[N];
#pragma unroll
        for (int c = 0; c < N; c++) uu[c] = 0.0;
#pragma unroll
        for (int i = 0; i < na; i++) {
            const double *mi = sMk + wrow[i] * N;
#pragma unroll
            for (int c = 0; c < N; c++) uu[c] = __builtin_fma(-mi[c], ls[i], uu[c]);
        }
        double fval = 0.0;
#pragma unroll
        for (int c = 0; c < N; c++) fval = __builtin_fma(uu[c], uu[c], fval);
        running = running && !(fval > fbound);                 // EXIT_INFEASIBLE: the generic loop reports it
        unsigned actmask = 0u;
#pragma unroll
        for (int i = 0; i < na; i++) actmask |= 1u << wrow[i];
        min_val = -ptol;
        add = -1;
        addlow = false;
        bool broken = false;
#ifndef LMPC_TIERS_NARROW_LDS
        // M (N x N, at the 16-byte aligned start of the block), then the Gram triangle, then du0 and dl0
        constexpr int kDu = N * N + N * (N + 1) / 2;
        double duv[N], dlv[N];
        lmpc_lds_run<N>(sMk, kDu, duv);
        lmpc_lds_run<N>(sMk, kDu + N, dlv);
#endif
#pragma unroll
        for (int jj = 0; jj < N; jj++) {
            double Mu = 0.0;
#ifndef LMPC_TIERS_NARROW_LDS
            double mrow[N];
            lmpc_lds_run<N>(sMk, jj * N, mrow);
            const double duj = duv[jj], dlj = dlv[jj];
#else
            const double *mrow = sMk + jj * N;
            const double duj = sduk[jj], dlj = sdlk[jj];
#endif
#pragma unroll
            for (int c = 0; c < N; c++) Mu = __builtin_fma(mrow[c], uu[c], Mu);
            const double vu = (duj + b[jj]) - Mu;
            const double vl = -((dlj + b[jj]) - Mu);
            const bool inact = !((actmask >> jj) & 1u);
            const bool tu = inact && (vu < min_val);
            const bool tl = inact && !tu && (vl < min_val);
            add = (tu || tl) ? jj : add;
            addlow = tu ? false : (tl ? true : addlow);
            min_val = tu ? vu : (tl ? vl : min_val);
            broken = broken || (!inact && (vu < -ptol || vl < -ptol));
        }
        running = running && !broken;                          // EXIT_CYCLE: the generic loop reports it
        const bool fin = running && add < 0;
        result = fin ? (int)EXIT_OPTIMAL : result;
        iter = fin ? k + 2 : iter;
        nact = fin ? na : nact;
#pragma unroll
        for (int c = 0; c < N; c++) u[c] = fin ? uu[c] : u[c];
        running = running && !fin && (k + 1 < KMAX);           // (wants a further row than KMAX: generic loop)
    }
    return result;
}

}  // namespace lmpc
