// Small box-constrained variational problems (the reference's game-theoretic MPC, /root/reference/test/runtests.jl:1337-1358:
// two players, n = 6 moves, 6 two-sided input bounds, is_avi, /root/reference/src/setup.jl:11-13) in REGISTERS.
//
// avi_kernel (lmpc_avi_kernel.hpp) keeps a lane's L D U factorisation in a slab of global scratch and is bound by the
// latency of that scratch (~1 800 accesses per problem).  For n <= 8 the whole state of a problem fits the registers
// of its lane -- if every array index is a compile-time constant.  Two kernels do that, one problem per lane:
//
// avi_tiers_kernel<N, KMAX>: the path 98 % of the problems of such a batch take -- rows are appended one after the
//   other (by the most-violated-row rule or by the principal-pivoting rule that protects satisfied rows), nothing is
//   ever removed -- as straight-line code, as lmpc_tiers.hpp does for the symmetric kernels:
//     iteration 1 (empty working set): the most violated bound, or done;
//     tier k (k + 1 rows, k = 0 .. KMAX-1): the new row's forward solve, lam* by the backward solve on U, the test
//     for blocking multipliers, both iterates u(lam) and u(lam*), the scan of all rows (most violated row at the
//     target, first satisfied row the step would break), then the append of the next row (new row of L, new column
//     of U, pivot).
//   Anything else is queued for the next kernel (work list: kShards segments, one counter each).
//
// avi_lane_kernel<N>: the complete iteration, removals included, with the working set's SIZE as the compile-time
//   quantity: the code of an iteration exists once per size 0 .. N (`level`), a wavefront sweeps the levels in
//   ascending order and a lane takes part in the level it is at.  A lane that appends a row moves on to the next level
//   of the same sweep (so a tile without removals costs what the tiers cost); a lane that removes one (Bennett's
//   rank-one update of L D U at a run-time position: selects over compile-time positions) waits for the next sweep.
//   What it does not finish -- a singular pivot, a row violated inside its own working set at the end, the iteration
//   limit -- is queued for avi_kernel (never seen on the reference's problem; the list is there for completeness).
//
// The chain of a batch: avi_tiers_kernel<N, 3> over the whole batch (three tiers: what 93 % of the reference's game
// problem needs, at 3 wavefronts per SIMD) -> avi_lane_kernel<N> on its list -> avi_kernel on that one's list.
//
// Each chain is the fma chain of avi_kernel / the CPU checker in the same order, so a problem finished here has the
// bits the generic kernel gives it.
//
// Preconditions (checked on the host, lmpc_avi.hip): m == n == N simple bounds whose rows of ML are s_j e_j' (then the
// row value ML_j u is the single product s_j u_j: the checker's chain adds exact zeros around it), no SOFT / IMMUTABLE /
// ACTIVE-flagged rows, cold start, iter_limit > N + 1, no proximal-point iterations.
//
// No MFMA (6 x 6 problems, dependent chains); HBM traffic = the algorithmic bytes (theta in, x and flag out) plus the
// lists; the kernels are VALU-issue bound.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lmpc_pack.hpp"
#include "lmpc_tiers.hpp"
#include "lmpc_wave_layout.hpp"

namespace lmpc {

// Constants: what every lane reads at the same index (Dth, bounds, the diagonal of ML, output maps) comes straight from
// the pack by SCALAR loads (uniform addresses: s_load into SGPRs, operands of the fma -- as LDS broadcasts the same
// reads kept the LDS pipe of a CU busy for half of the kernel's time); what a lane reads at its OWN index (rows of MR,
// entries of G, by working-set row) is copied to LDS at compile-time offsets (rows of MR start on 16-byte boundaries:
// stride NP = N rounded up to even).
template <int N> struct AviSmallLds {
    static constexpr int NP = N + (N & 1);
    static constexpr int oMR = 0, oG = N * NP, oSd = oG + ((N * N + 1) & ~1);
    static constexpr int reals = oSd + NP;
};

template <int N>
__device__ __forceinline__ void avi_small_stage(const AviLayout &P, const double *__restrict__ C, double *lds, int tid, int nthreads) {
    typedef AviSmallLds<N> Ly;
    for (int e = tid; e < N * N; e += nthreads) {
        const int j = e / N, c = e - j * N;
        lds[Ly::oMR + j * Ly::NP + c] = C[P.oMR + e];
        lds[Ly::oG + e] = C[P.oG + e];
    }
    for (int e = tid; e < N; e += nthreads) lds[Ly::oSd + e] = C[P.oSd + e];
}

// bounds of a parameter point (mpc_update_qp.c:1-10): d = du0 / dl0 + Dth theta, sums in index order; in the same pass
// over theta the parameter's part of the outputs, x0 + Xth theta (mpc_update_qp.c:14-22), parked in LDS (element kk of
// lane l at shx[kk * 64]) until the problem is finished.  Per column of theta ONE batch of scalar loads (AviLayout::oTh2).
template <int N>
__device__ __forceinline__ void avi_small_bounds(const AviLayout &P, const double *__restrict__ C, const double *th, int nth, int nout,
                                                 double (&dup)[N], double (&dlo)[N], double *shx) {
    double sh[N], so[N];
    {
        const double *b3 = C + P.oBnd3;
        double x0v[N];
#pragma unroll
        for (int j = 0; j < N; j++) x0v[j] = b3[2 * N + j];
#pragma unroll
        for (int j = 0; j < N; j++) { sh[j] = 0.0; so[j] = x0v[j]; }
    }
    const double *col = C + P.oTh2;
    for (int t = 0; t < nth; t++, col += 2 * N) {
        const double tv = th[t];
        double cv[2 * N];
#pragma unroll
        for (int j = 0; j < 2 * N; j++) cv[j] = col[j];
#pragma unroll
        for (int j = 0; j < N; j++) sh[j] = __builtin_fma(cv[j], tv, sh[j]);
#pragma unroll
        for (int kk = 0; kk < N; kk++) so[kk] = __builtin_fma(cv[N + kk], tv, so[kk]);      // (rows >= nout: zeros, not stored)
    }
    {
        const double *b3 = C + P.oBnd3;
        double bv[2 * N];
#pragma unroll
        for (int j = 0; j < 2 * N; j++) bv[j] = b3[j];
#pragma unroll
        for (int j = 0; j < N; j++) { dup[j] = bv[j] + sh[j]; dlo[j] = bv[N + j] + sh[j]; }
    }
#pragma unroll
    for (int kk = 0; kk < N; kk++)
        if (kk < nout) shx[kk * 64] = so[kk];
}

// x = Rout u + (x0 + Xth theta), flag, iteration count, working set of a finished problem
template <int N>
__device__ __forceinline__ void avi_small_output(const AviLayout &P, const double *__restrict__ C, const double *shx, int nout,
                                                 long long pid, const double (&u)[N], int iter, unsigned act, unsigned low,
                                                 double *__restrict__ X, int32_t *__restrict__ exitflag,
                                                 int32_t *__restrict__ iters, uint64_t *__restrict__ active) {
    const int words = P.words;
    for (int kk = 0; kk < nout; kk++) {
        const double *ro = C + P.oRout + kk * N;
        double xs = 0.0;
#pragma unroll
        for (int c = 0; c < N; c++) xs = __builtin_fma(ro[c], u[c], xs);
        X[pid * nout + kk] = xs + shx[kk * 64];
    }
    exitflag[pid] = EXIT_OPTIMAL;
    if (iters) iters[pid] = iter;
    if (active) {
        // bit j: row j at its upper bound, bit m + j: at its lower bound (2 m <= 16 bits: one word)
        const unsigned up = act & ~low, lo = act & low;
        active[pid * words] = (uint64_t)up | ((uint64_t)lo << N);
    }
}

// the lanes flagged `queue` append their problem index to segment `shard` of a work list -- in two halves, so that the
// round trip of the counter's atomic is not waited for: avi_push_reserve at the end of a tile, avi_push_write at the end
// of the NEXT one (and after the last)
struct AviPush { unsigned long long mask = 0ull; int base = 0; int pid = 0; };
__device__ __forceinline__ void avi_push_reserve(AviPush &q, bool queue, long long pid, int lane, int shard,
                                                 int32_t *__restrict__ count_out) {
    q.mask = __ballot(queue);
    q.pid = (int)pid;
    q.base = 0;
    if (q.mask != 0ull && lane == 0) q.base = atomicAdd(&count_out[shard * kCountStride], __popcll(q.mask));
}
__device__ __forceinline__ void avi_push_write(AviPush &q, int lane, int shard, long long seg_cap, int32_t *__restrict__ list_out) {
    if (q.mask != 0ull) {
        const int qbase = __shfl(q.base, 0);
        if ((q.mask >> lane) & 1ull)
            list_out[(long long)shard * seg_cap + qbase + __popcll(q.mask & ((1ull << lane) - 1ull))] = q.pid;
    }
    q.mask = 0ull;
}
__device__ __forceinline__ void avi_small_push(bool queue, long long pid, int lane, int shard, long long seg_cap,
                                               int32_t *__restrict__ list_out, int32_t *__restrict__ count_out) {
    AviPush q;
    avi_push_reserve(q, queue, pid, lane, shard, count_out);
    avi_push_write(q, lane, shard, seg_cap, list_out);
}

// one scanned row: most violated inactive row at the target (ties: the first), first satisfied row the step breaks
#define LMPC_AVI_SCAN_ROW(jj, ACTIVE_LANES)                                                                    \
    {                                                                                                          \
        const double s_ = sd[jj];                                                                              \
        const double Mc = __builtin_fma(s_, uc[jj], 0.0), Mt = __builtin_fma(s_, ut[jj], 0.0);                 \
        const double dj = dup[jj], ej = dlo[jj];                                                               \
        const double vu = dj - Mt, vl = -(ej - Mt);                                                            \
        const bool inact = !((actrow >> (jj)) & 1u);                                                           \
        broken = broken || (!inact && (vu < -ptol || vl < -ptol));                                             \
        const bool tu = inact && (vu < min_val);                                                               \
        const bool tl = inact && !tu && (vl < min_val);                                                        \
        add = (tu || tl) ? (jj) : add;                                                                         \
        addlow = tu ? false : (tl ? true : addlow);                                                            \
        min_val = tu ? vu : (tl ? vl : min_val);                                                               \
        const double cu = dj - Mc, cl = -(ej - Mc);                                                            \
        const bool bu_ = inact && (vu < -ptol) && (cu >= -ptol);                                               \
        const bool bl_ = inact && !bu_ && (vl < -ptol) && (cl >= -ptol);                                       \
        if (__any((ACTIVE_LANES) && (bu_ || bl_))) {                                                           \
            const double cv = bu_ ? cu : cl, vv = bu_ ? vu : vl;                                               \
            const double t = cv > 0.0 ? cv / (cv - vv) : 0.0;                                                  \
            const bool take = (bu_ || bl_) && (t < tblk);                                                      \
            tblk = take ? t : tblk;                                                                            \
            pblk = take ? (jj) : pblk;                                                                         \
            pup = take ? bu_ : pup;                                                                            \
        }                                                                                                      \
    }

template <int N, int KMAX, bool LIST>
__global__ __launch_bounds__(256) void avi_tiers_kernel(
    const AviLayout P, const double *__restrict__ C, const double *__restrict__ theta, double *__restrict__ X,
    int32_t *__restrict__ exitflag, int32_t *__restrict__ iters, uint64_t *__restrict__ active,
    const int32_t *__restrict__ list_in, const int32_t *__restrict__ count_in, int32_t *__restrict__ list_out,
    int32_t *__restrict__ count_out, int32_t *__restrict__ count_clear, const long long seg_cap, const long long nprob) {
    static_assert(KMAX >= 1 && KMAX <= N && N <= 8, "tiers up to the problem's size");
    typedef AviSmallLds<N> Ly;
    extern __shared__ __align__(16) double lds[];
    const int nth = P.nth, nout = P.nout;
    const int tid = threadIdx.x, lane = tid & 63;
    // the counters of the list AFTER the next one are cleared here (the kernel that read them has finished: stream order)
    if (count_clear && blockIdx.x == 0 && tid < kShards) count_clear[tid * kCountStride] = 0;
    avi_small_stage<N>(P, C, lds, tid, 256);
    __syncthreads();
    const double ptol = P.primal_tol, dtol = P.dual_tol, ztol = P.zero_tol;

    const long long gw = (long long)blockIdx.x * 4 + (tid >> 6), nw = (long long)gridDim.x * 4;   // (nw % kShards == 0)
    const int shard = (int)(gw % kShards);
    long long first, stride, cnt;
    if constexpr (LIST) {
        first = (gw / kShards) * 64; stride = (nw / kShards) * 64; cnt = (long long)count_in[shard * kCountStride];
        list_in += (long long)shard * seg_cap;
    } else {
        first = gw * 64; stride = nw * 64; cnt = nprob;
    }

    double *shx = lds + Ly::reals + (tid >> 6) * (nout * 64) + lane;   // this lane's parked outputs
    // the diagonal of ML in vector registers for the whole kernel (through LDS: the compiler keeps what it read from
    // there in VGPRs; as scalar loads inside the scan each row waited for its own)
    double sd[N];
    lmpc_lds_run<N>(lds + Ly::oSd, 0, sd);
    AviPush pend;
    for (long long base = first; base < cnt; base += stride) {
        const long long idx = base + lane;
        const bool mine = idx < cnt;
        long long pid = 0;
        if constexpr (LIST) pid = mine ? (long long)list_in[idx] : 0;
        else pid = mine ? idx : 0;
        const double *th = theta + pid * nth;
        double dup[N], dlo[N];
        // (the scalar loads of the constants stay where they are used: hoisted out of this loop they would sit in more
        // scalar registers than there are -- the empty asm hides that the addresses repeat)
        int zofs = 0;
        asm volatile("" : "+s"(zofs));
        const double *Cz = C + zofs;
        avi_small_bounds<N>(P, Cz, th, nth, nout, dup, dlo, shx);
        // ---- iteration 1: empty working set, both iterates 0
        double min_val = -ptol;
        int add = -1;
        bool addlow = false;
#pragma unroll
        for (int j = 0; j < N; j++) {
            const double vu = dup[j] - 0.0, vl = -(dlo[j] - 0.0);
            const bool tu = vu < min_val;
            const bool tl = !tu && (vl < min_val);
            add = (tu || tl) ? j : add;
            addlow = tu ? false : (tl ? true : addlow);
            min_val = tu ? vu : (tl ? vl : min_val);
        }
        constexpr int NS = KMAX * (KMAX - 1) / 2 > 0 ? KMAX * (KMAX - 1) / 2 : 1;
        double Ls[NS], Us[NS], Dinv[KMAX], lam[KMAX], ls[KMAX], xl[KMAX], zl[KMAX], ufin[N];
        int wrow[KMAX];
        unsigned lowpos = 0u;        // bit i: position i holds its row at the LOWER bound
        unsigned actrow = 0u, lowrow = 0u;   // by row
#pragma unroll
        for (int i = 0; i < KMAX; i++) { Dinv[i] = 0.0; lam[i] = 0.0; ls[i] = 0.0; xl[i] = 0.0; zl[i] = 0.0; wrow[i] = 0; }
#pragma unroll
        for (int i = 0; i < NS; i++) { Ls[i] = 0.0; Us[i] = 0.0; }
#pragma unroll
        for (int c = 0; c < N; c++) ufin[c] = 0.0;
        bool finished = mine && add < 0;     // the unconstrained equilibrium is feasible
        bool running = mine && add >= 0;
        int iter_fin = 1;
        unsigned act_fin = 0u, low_fin = 0u;
        bool lower_next = addlow;            // bound the next appended row goes to

#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            // ---- append row `add` at position k (avi_kernel's ldu_add with na = k)
            const int j = add < 0 ? 0 : add;
            {
                double rl[KMAX > 1 ? KMAX - 1 : 1], ru[KMAX > 1 ? KMAX - 1 : 1];
#pragma unroll
                for (int i = 0; i < k; i++) { rl[i] = lds[Ly::oG + j * N + wrow[i]]; ru[i] = lds[Ly::oG + wrow[i] * N + j]; }
                double dnew = lds[Ly::oG + j * N + j];
#pragma unroll
                for (int i = 1; i < k; i++) {
                    double al = rl[i], au = ru[i];
#pragma unroll
                    for (int t = 0; t < i; t++) {
                        al = __builtin_fma(-Us[lmpc_sl(i, t)], rl[t], al);
                        au = __builtin_fma(-Ls[lmpc_sl(i, t)], ru[t], au);
                    }
                    rl[i] = al; ru[i] = au;
                }
#pragma unroll
                for (int i = 0; i < k; i++) {
                    const double ql = rl[i], qu = ru[i], di = Dinv[i];
                    const double l = ql * di;
                    rl[i] = l; ru[i] = qu * di;
                    dnew = __builtin_fma(-l, qu, dnew);
                }
                running = running && !(dnew < ztol);           // singular working set: the generic kernel
#pragma unroll
                for (int t = 0; t < k; t++) { Ls[lmpc_sl(k, t)] = rl[t]; Us[lmpc_sl(k, t)] = ru[t]; }
                Dinv[k] = 1.0 / dnew;
            }
            wrow[k] = j; lam[k] = 0.0;
            lowpos = lower_next ? (lowpos | (1u << k)) : lowpos;
            actrow |= 1u << j;
            lowrow = lower_next ? (lowrow | (1u << j)) : lowrow;
            if (!__any(running)) break;
            // ---- iteration k + 2 on k + 1 rows: (L D U) lam* = -d_W; rows 0 .. k-1 of the forward solve are kept
            const int na = k + 1;
            {
                double dsel = 0.0;
#pragma unroll
                for (int q = 0; q < N; q++) dsel = (q == j) ? (lower_next ? dlo[q] : dup[q]) : dsel;
                double acc = -dsel;
#pragma unroll
                for (int t = 0; t < k; t++) acc = __builtin_fma(-Ls[lmpc_sl(k, t)], xl[t], acc);
                xl[k] = acc;
                zl[k] = acc * Dinv[k];
            }
#pragma unroll
            for (int i = na - 1; i >= 0; i--) {
                double acc = zl[i];
#pragma unroll
                for (int t = na - 1; t > i; t--) acc = __builtin_fma(-Us[lmpc_sl(t, i)], ls[t], acc);
                ls[i] = acc;
            }
            bool blocked = false;
#pragma unroll
            for (int i = 0; i < na; i++) {
                const bool ok = ((lowpos >> i) & 1u) ? (ls[i] < dtol) : (ls[i] > -dtol);
                blocked = blocked || !ok;
            }
            running = running && !blocked;                     // a removal: the next kernel
            // ---- the iterate now (from lam) and the target of this step (from lam*)
            double uc[N], ut[N];
#pragma unroll
            for (int c = 0; c < N; c++) { uc[c] = 0.0; ut[c] = 0.0; }
#pragma unroll
            for (int i = 0; i < na; i++) {
                double mi[N];
                lmpc_lds_run<N>(lds + Ly::oMR + wrow[i] * Ly::NP, 0, mi);
                const double lc = lam[i], lt = ls[i];
#pragma unroll
                for (int c = 0; c < N; c++) {
                    uc[c] = __builtin_fma(-mi[c], lc, uc[c]);
                    ut[c] = __builtin_fma(-mi[c], lt, ut[c]);
                }
            }
            // ---- scan: most violated row at the target; first satisfied row the step would break
            min_val = -ptol;
            add = -1;
            addlow = false;
            double tblk = 1.0;
            int pblk = -1;
            bool pup = false, broken = false;
#pragma unroll
            for (int jj = 0; jj < N; jj++) LMPC_AVI_SCAN_ROW(jj, running)
            const bool stepblk = pblk >= 0;
            const bool fin = running && !stepblk && add < 0;
            running = running && !(fin && broken);             // EXIT_CYCLE: the generic kernel reports it
            const bool done = fin && !broken;
            finished = finished || done;
            iter_fin = done ? k + 2 : iter_fin;
            act_fin = done ? actrow : act_fin;
            low_fin = done ? lowrow : low_fin;
#pragma unroll
            for (int c = 0; c < N; c++) ufin[c] = done ? ut[c] : ufin[c];
            running = running && !fin;
            // ---- the step: to the blocking row (partial) or to the target (full); the row appended next
#pragma unroll
            for (int i = 0; i < na; i++) lam[i] = stepblk ? __builtin_fma(tblk, ls[i] - lam[i], lam[i]) : ls[i];
            add = stepblk ? pblk : add;
            lower_next = stepblk ? !pup : addlow;
            running = running && (k + 1 < KMAX);               // wants a further row than KMAX: the next kernel
        }

        if (finished)
            avi_small_output<N>(P, C + zofs, shx, nout, pid, ufin, iter_fin, act_fin, low_fin, X, exitflag, iters, active);
        avi_push_write(pend, lane, shard, seg_cap, list_out);          // (the tile before's)
        avi_push_reserve(pend, mine && !finished, pid, lane, shard, count_out);
    }
    avi_push_write(pend, lane, shard, seg_cap, list_out);
}

// ---------------------------------------------------------------------------------------------------------------------
// avi_lane_kernel: the complete iteration in registers, one code block per working-set size.
// Per lane in registers: the two triangles, reciprocal pivots, multipliers, working set.  What an iteration reads once or
// rarely -- the bounds of all rows (the scan), the pivots themselves (a removal), the bound each position sits at, the
// finished iterate -- is PARKED in LDS, element major (element e of lane l at park[e * 64 + l]: conflict-free, immediate
// offsets): 6 N reals a lane, so that the kernel fits 2 wavefronts per SIMD up to n = 6.
template <int N> struct AviLaneState {
    static constexpr int NS = N * (N - 1) / 2 > 0 ? N * (N - 1) / 2 : 1;
    static constexpr int kPark = 6 * N;                 // dup, dlo, D, rhs, ufin, the outputs' parameter part (nout <= n)
    double Ls[NS], Us[NS];           // strict lower triangles of L and of U' (row i > column t)
    double Di[N], lam[N];            // reciprocal pivots, multipliers
    double *park;                    // this lane's column of the wavefront's parking area
    int wrow[N];
    unsigned lowpos, actrow, lowrow;
    int level, iter, iter_fin;
    unsigned act_fin, low_fin;
    bool running, finished;
    __device__ __forceinline__ double &dup(int j) { return park[j * 64]; }
    __device__ __forceinline__ double &dlo(int j) { return park[(N + j) * 64]; }
    __device__ __forceinline__ double &D(int i) { return park[(2 * N + i) * 64]; }
    __device__ __forceinline__ double &rhs(int i) { return park[(3 * N + i) * 64]; }
    __device__ __forceinline__ double &ufin(int c) { return park[(4 * N + c) * 64]; }
};

// one scanned row of the lane kernel (bounds from the parking area)
#define LMPC_AVI_LANE_SCAN_ROW(jj)                                                                             \
    {                                                                                                          \
        const double s_ = sd[jj];                                                                              \
        const double Mc = __builtin_fma(s_, uc[jj], 0.0), Mt = __builtin_fma(s_, ut[jj], 0.0);                 \
        const double dj = s.dup(jj), ej = s.dlo(jj);                                                           \
        const double vu = dj - Mt, vl = -(ej - Mt);                                                            \
        const bool inact = !((actrow >> (jj)) & 1u);                                                           \
        broken = broken || (!inact && (vu < -ptol || vl < -ptol));                                             \
        const bool tu = inact && (vu < min_val);                                                               \
        const bool tl = inact && !tu && (vl < min_val);                                                        \
        add = (tu || tl) ? (jj) : add;                                                                         \
        addlow = tu ? false : (tl ? true : addlow);                                                            \
        min_val = tu ? vu : (tl ? vl : min_val);                                                               \
        const double cu = dj - Mc, cl = -(ej - Mc);                                                            \
        const bool bu_ = inact && (vu < -ptol) && (cu >= -ptol);                                               \
        const bool bl_ = inact && !bu_ && (vl < -ptol) && (cl >= -ptol);                                       \
        if (__any(bu_ || bl_)) {                                                                               \
            const double cv = bu_ ? cu : cl, vv = bu_ ? vu : vl;                                               \
            const double t = cv > 0.0 ? cv / (cv - vv) : 0.0;                                                  \
            const bool take = (bu_ || bl_) && (t < tblk);                                                      \
            tblk = take ? t : tblk;                                                                            \
            pblk = take ? (jj) : pblk;                                                                         \
            pup = take ? bu_ : pup;                                                                            \
        }                                                                                                      \
    }

// one iteration of the lanes at level NA (`part`: this lane takes part).  Statement for statement avi_kernel's loop body
// for sing < 0 with na == NA a constant.
template <int N, int NA>
__device__ __forceinline__ void avi_lane_level(AviLaneState<N> &s, const bool part, const double *lds, const double *__restrict__ sdp,
                                               const double ptol, const double dtol, const double ztol, const int ilimit) {
    typedef AviSmallLds<N> Ly;
    if (!part) return;
    if (s.iter >= ilimit) { s.running = false; return; }      // iteration limit: the generic kernel reports it
    double sd[N];                                              // the diagonal of ML: one batch of scalar loads per level
    {
        int zofs = 0;
        asm volatile("" : "+s"(zofs));
#pragma unroll
        for (int j = 0; j < N; j++) sd[j] = sdp[zofs + j];
    }
    const unsigned actrow = s.actrow;
    // ---- (L D U) lam* = -d_W
    double ls[NA > 0 ? NA : 1];
    {
        double xl[NA > 0 ? NA : 1];
#pragma unroll
        for (int i = 0; i < NA; i++) {
            double acc = -s.rhs(i);
#pragma unroll
            for (int t = 0; t < i; t++) acc = __builtin_fma(-s.Ls[lmpc_sl(i, t)], xl[t], acc);
            xl[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < NA; i++) xl[i] = xl[i] * s.Di[i];
#pragma unroll
        for (int i = NA - 1; i >= 0; i--) {
            double acc = xl[i];
#pragma unroll
            for (int t = NA - 1; t > i; t--) acc = __builtin_fma(-s.Us[lmpc_sl(t, i)], ls[t], acc);
            ls[i] = acc;
        }
    }
    // ---- blocking multipliers: the first to reach zero on the way lam -> lam*
    int nblock = 0, rm = -1;
    double alpha = 0.0;
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const bool blk = ((s.lowpos >> i) & 1u) ? !(ls[i] < dtol) : !(ls[i] > -dtol);
        if (__any(blk)) {
            const double cand = -s.lam[i] / (ls[i] - s.lam[i]);
            const bool take = blk && (nblock == 0 || cand < alpha);
            alpha = take ? cand : alpha;
            rm = take ? i : rm;
            nblock += blk ? 1 : 0;
        }
    }
    // ---- the iterate now (from lam) and the target of this step (from lam*)
    double uc[N], ut[N];
#pragma unroll
    for (int c = 0; c < N; c++) { uc[c] = 0.0; ut[c] = 0.0; }
#pragma unroll
    for (int i = 0; i < NA; i++) {
        double mi[N];
        lmpc_lds_run<N>(lds + Ly::oMR + s.wrow[i] * Ly::NP, 0, mi);
        const double lc = s.lam[i], lt = ls[i];
#pragma unroll
        for (int c = 0; c < N; c++) {
            uc[c] = __builtin_fma(-mi[c], lc, uc[c]);
            ut[c] = __builtin_fma(-mi[c], lt, ut[c]);
        }
    }
    // ---- scan
    double min_val = -ptol, tblk = nblock ? alpha : 1.0;
    int add = -1, pblk = -1;
    bool addlow = false, pup = false, broken = false;
#pragma unroll
    for (int jj = 0; jj < N; jj++) LMPC_AVI_LANE_SCAN_ROW(jj)
    const bool stepblk = pblk >= 0;
    const bool do_remove = !stepblk && nblock > 0;
    if (!stepblk && !do_remove && add < 0) {
        // the target is feasible and no multiplier blocks: done (a row violated inside its own working set: the
        // generic kernel reports the cycle)
        s.running = false;
        if (!broken) {
            s.finished = true;
            s.iter_fin = s.iter;
            s.act_fin = s.actrow; s.low_fin = s.lowrow;
#pragma unroll
            for (int c = 0; c < N; c++) s.ufin(c) = ut[c];
        }
        return;
    }
    s.iter++;
    if (!do_remove) {
        // ---- step (partial up to the blocking row, or full), then append row j at position NA
        if constexpr (NA >= N) {
            s.running = false;                                 // a further row than n: singular, the generic kernel
        } else {
#pragma unroll
            for (int i = 0; i < NA; i++) s.lam[i] = stepblk ? __builtin_fma(tblk, ls[i] - s.lam[i], s.lam[i]) : ls[i];
            const int j = stepblk ? pblk : add;
            const bool lower = stepblk ? !pup : addlow;
            double rl[NA > 0 ? NA : 1], ru[NA > 0 ? NA : 1];
#pragma unroll
            for (int i = 0; i < NA; i++) { rl[i] = lds[Ly::oG + j * N + s.wrow[i]]; ru[i] = lds[Ly::oG + s.wrow[i] * N + j]; }
            double dnew = lds[Ly::oG + j * N + j];
#pragma unroll
            for (int i = 1; i < NA; i++) {
                double al = rl[i], au = ru[i];
#pragma unroll
                for (int t = 0; t < i; t++) {
                    al = __builtin_fma(-s.Us[lmpc_sl(i, t)], rl[t], al);
                    au = __builtin_fma(-s.Ls[lmpc_sl(i, t)], ru[t], au);
                }
                rl[i] = al; ru[i] = au;
            }
#pragma unroll
            for (int i = 0; i < NA; i++) {
                const double ql = rl[i], qu = ru[i], di = s.Di[i];
                const double l = ql * di;
                rl[i] = l; ru[i] = qu * di;
                dnew = __builtin_fma(-l, qu, dnew);
            }
            if (dnew < ztol) { s.running = false; return; }    // singular working set: the generic kernel
#pragma unroll
            for (int t = 0; t < NA; t++) { s.Ls[lmpc_sl(NA, t)] = rl[t]; s.Us[lmpc_sl(NA, t)] = ru[t]; }
            s.D(NA) = dnew; s.Di[NA] = 1.0 / dnew;
            s.wrow[NA] = j; s.lam[NA] = 0.0;
            s.rhs(NA) = s.park[((lower ? N : 0) + j) * 64];    // (the bound it sits at: dlo / dup of row j)
            s.lowpos = lower ? (s.lowpos | (1u << NA)) : (s.lowpos & ~(1u << NA));
            s.lowrow = lower ? (s.lowrow | (1u << j)) : (s.lowrow & ~(1u << j));
            s.actrow |= 1u << j;
            s.level = NA + 1;
        }
        return;
    }
    // ---- step up to the blocking multiplier, then remove position rm (avi_kernel's ldu_remove with na == NA)
    if constexpr (NA >= 1) {
#pragma unroll
        for (int i = 0; i < NA; i++) s.lam[i] = __builtin_fma(alpha, ls[i] - s.lam[i], s.lam[i]);
        const int r = rm;
        double Dv[NA], Rv[NA];                                 // pivots and bounds by position, back from the parking area
#pragma unroll
        for (int i = 0; i < NA; i++) { Dv[i] = s.D(i); Rv[i] = s.rhs(i); }
        // old column r below the diagonal, by NEW position i = r .. NA-2 (old row i + 1); the pivot removed
        double pv[NA > 1 ? NA - 1 : 1], qv[NA > 1 ? NA - 1 : 1];
        double al = s.park[(2 * N + r) * 64];
#pragma unroll
        for (int i = 0; i < NA - 1; i++) {
            double p = 0.0, qq = 0.0;
#pragma unroll
            for (int c = 0; c <= i; c++) {
                p = (c == r) ? s.Ls[lmpc_sl(i + 1, c)] : p;
                qq = (c == r) ? s.Us[lmpc_sl(i + 1, c)] : qq;
            }
            pv[i] = p; qv[i] = qq;
        }
        // rows r .. NA-2 <- rows r+1 .. NA-1 without column r; pivots, working set, multipliers, bounds likewise
#pragma unroll
        for (int i = 0; i < NA - 1; i++) {
            const bool mv = i >= r;
#pragma unroll
            for (int c = 0; c < i; c++) {
                const double l0 = s.Ls[lmpc_sl(i, c)], l1 = s.Ls[lmpc_sl(i + 1, c)], l2 = s.Ls[lmpc_sl(i + 1, c + 1)];
                const double u0 = s.Us[lmpc_sl(i, c)], u1 = s.Us[lmpc_sl(i + 1, c)], u2 = s.Us[lmpc_sl(i + 1, c + 1)];
                s.Ls[lmpc_sl(i, c)] = mv ? (c < r ? l1 : l2) : l0;
                s.Us[lmpc_sl(i, c)] = mv ? (c < r ? u1 : u2) : u0;
            }
            Dv[i] = mv ? Dv[i + 1] : Dv[i];
            Rv[i] = mv ? Rv[i + 1] : Rv[i];
            s.Di[i] = mv ? s.Di[i + 1] : s.Di[i];
            s.lam[i] = mv ? s.lam[i + 1] : s.lam[i];
        }
        int jr = 0;
#pragma unroll
        for (int q = 0; q < NA; q++) jr = (q == r) ? s.wrow[q] : jr;
#pragma unroll
        for (int i = 0; i < NA - 1; i++) s.wrow[i] = (i >= r) ? s.wrow[i + 1] : s.wrow[i];
        {
            const unsigned keep = (1u << r) - 1u;
            s.lowpos = (s.lowpos & keep) | ((s.lowpos >> 1) & ~keep);
        }
        s.actrow &= ~(1u << jr);
        s.lowrow &= ~(1u << jr);
        // Bennett's update along the new positions r .. NA-2
        bool stop = false;
#pragma unroll
        for (int i = 0; i < NA - 1; i++) {
            const bool on = i >= r && !stop;
            const double pt = pv[i], qt = qv[i];
            const double dold = Dv[i];                         // (already shifted: old D[i + 1])
            const double dbar = __builtin_fma(al * pt, qt, dold);
            const bool sg = on && (dbar < ztol);
            stop = stop || sg;
            const bool upd = on && !sg;
            const double rinv = 1.0 / dbar;
            const double betaL = (qt * al) * rinv;
            const double betaU = (pt * al) * rinv;
            al = upd ? (dold * al) * rinv : al;
            Dv[i] = upd ? dbar : Dv[i];
            s.Di[i] = upd ? rinv : s.Di[i];
#pragma unroll
            for (int p = i + 1; p < NA - 1; p++) {
                const double lqi = s.Ls[lmpc_sl(p, i)], uqi = s.Us[lmpc_sl(p, i)];
                const double pq = __builtin_fma(-pt, lqi, pv[p]);
                const double qq = __builtin_fma(-qt, uqi, qv[p]);
                pv[p] = upd ? pq : pv[p];
                qv[p] = upd ? qq : qv[p];
                s.Ls[lmpc_sl(p, i)] = upd ? __builtin_fma(betaL, pq, lqi) : lqi;
                s.Us[lmpc_sl(p, i)] = upd ? __builtin_fma(betaU, qq, uqi) : uqi;
            }
        }
        if (stop) { s.running = false; return; }               // a singular pivot: the generic kernel
#pragma unroll
        for (int i = 0; i < NA - 1; i++) { s.D(i) = Dv[i]; s.rhs(i) = Rv[i]; }
        s.level = NA - 1;
    }
}

template <int N, int L>
__device__ __forceinline__ void avi_lane_sweep(AviLaneState<N> &s, const double *lds, const double *__restrict__ sd, const double ptol,
                                               const double dtol, const double ztol, const int ilimit) {
    if constexpr (L <= N) {
        avi_lane_level<N, L>(s, s.running && s.level == L, lds, sd, ptol, dtol, ztol, ilimit);
        avi_lane_sweep<N, L + 1>(s, lds, sd, ptol, dtol, ztol, ilimit);
    }
}

// LDS of a workgroup of avi_lane_kernel: the constants, then the four wavefronts' parking areas
template <int N> __host__ __device__ constexpr int avi_lane_lds_reals() {
    return AviSmallLds<N>::reals + 4 * AviLaneState<N>::kPark * 64;
}

template <int N, bool LIST>
__global__ __launch_bounds__(256, (N <= 6 ? 2 : 1)) void avi_lane_kernel(
    const AviLayout P, const double *__restrict__ C, const double *__restrict__ theta, double *__restrict__ X,
    int32_t *__restrict__ exitflag, int32_t *__restrict__ iters, uint64_t *__restrict__ active,
    const int32_t *__restrict__ list_in, const int32_t *__restrict__ count_in, int32_t *__restrict__ list_out,
    int32_t *__restrict__ count_out, int32_t *__restrict__ count_clear, const long long seg_cap, const long long nprob) {
    static_assert(N >= 2 && N <= 8, "register-resident L D U");
    extern __shared__ __align__(16) double lds[];
    const int nth = P.nth, nout = P.nout;
    const int tid = threadIdx.x, lane = tid & 63;
    if (count_clear && blockIdx.x == 0 && tid < kShards) count_clear[tid * kCountStride] = 0;
    avi_small_stage<N>(P, C, lds, tid, 256);
    __syncthreads();
    const double ptol = P.primal_tol, dtol = P.dual_tol, ztol = P.zero_tol;
    const long long gw = (long long)blockIdx.x * 4 + (tid >> 6), nw = (long long)gridDim.x * 4;   // (nw % kShards == 0)
    const int shard = (int)(gw % kShards);
    long long first, stride, cnt;
    if constexpr (LIST) {
        first = (gw / kShards) * 64; stride = (nw / kShards) * 64; cnt = (long long)count_in[shard * kCountStride];
        list_in += (long long)shard * seg_cap;
    } else {
        first = gw * 64; stride = nw * 64; cnt = nprob;
    }
    AviLaneState<N> s;
    s.park = lds + AviSmallLds<N>::reals + (tid >> 6) * (AviLaneState<N>::kPark * 64) + lane;
    const double *sd = C + P.oSd;
    for (long long base = first; base < cnt; base += stride) {
        const long long idx = base + lane;
        const bool mine = idx < cnt;
        long long pid = 0;
        if constexpr (LIST) pid = mine ? (long long)list_in[idx] : 0;
        else pid = mine ? idx : 0;
        const double *th = theta + pid * nth;
        {
            double dup[N], dlo[N];
            int zofs = 0;
            asm volatile("" : "+s"(zofs));
            avi_small_bounds<N>(P, C + zofs, th, nth, nout, dup, dlo, s.park + 5 * N * 64);
#pragma unroll
            for (int j = 0; j < N; j++) { s.dup(j) = dup[j]; s.dlo(j) = dlo[j]; }
        }
#pragma unroll
        for (int i = 0; i < AviLaneState<N>::NS; i++) { s.Ls[i] = 0.0; s.Us[i] = 0.0; }
#pragma unroll
        for (int i = 0; i < N; i++) { s.Di[i] = 0.0; s.lam[i] = 0.0; s.wrow[i] = 0; }
        s.lowpos = s.actrow = s.lowrow = 0u;
        s.level = 0; s.iter = 1; s.iter_fin = 1; s.act_fin = s.low_fin = 0u;
        s.running = mine; s.finished = false;
        // sweeps over the levels until no lane of the wavefront is running (each pass of the loop advances every
        // running lane by at least one iteration, and a lane stops at the iteration limit at the latest)
        while (__any(s.running)) avi_lane_sweep<N, 0>(s, lds, sd, ptol, dtol, ztol, P.iter_limit);
        if (s.finished) {
            double uf[N];
#pragma unroll
            for (int c = 0; c < N; c++) uf[c] = s.ufin(c);
            avi_small_output<N>(P, C, s.park + 5 * N * 64, nout, pid, uf, s.iter_fin, s.act_fin, s.low_fin, X, exitflag, iters, active);
        }
        avi_small_push(mine && !s.finished, pid, lane, shard, seg_cap, list_out, count_out);
    }
}

#undef LMPC_AVI_LANE_SCAN_ROW
#undef LMPC_AVI_SCAN_ROW

}  // namespace lmpc
