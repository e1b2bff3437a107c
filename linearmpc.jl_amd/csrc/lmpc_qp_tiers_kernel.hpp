// Straight-line tiers, one problem per LANE, in front of the wavefront kernel -- for small problems with MANY rows
// (n <= 12 variables, up to 64 hard or SOFT rows: the reference's own mass_spring example, n = 10, m = 63,
// /root/reference/src/mpc_examples.jl:241-286).
//
// The wavefront kernel gives such a problem a whole wavefront: ~800 instructions per iteration whatever the size of
// the working set, with 10 of its 64 lanes holding variables.  But 86 % of that example's sample (feasible and
// infeasible points alike) follow a path on which rows are only ever APPENDED -- until the optimum, or until the
// working set turns singular with nothing to drop: "infeasible".  On that path every working-set position is a
// compile-time constant, so it is straight-line code on register arrays, 64 problems per wavefront:
//
//   iteration 1 (empty working set): the most violated row, or done;
//   tier k (k + 1 rows, k = 0 .. N): append the row (new row of L, pivot); if the pivot vanished -- the singular
//   direction, no blocking multiplier => INFEASIBLE; else the new entry of the forward solve, lam* by the backward
//   solve, dual feasibility, u = -M_W' lam*, |u|^2 against fval_bound, the scan of all m rows.
//
// Each chain is the wavefront kernel's / the CPU checker's (oracle mode 0) in the same order, so a problem finished
// here -- flag 1, 2 (a SOFT row violated at the optimum) or -1 -- has their bits (x, flag, iteration count, active set).  Anything else -- a blocking multiplier
// (a removal), a row violated inside its own working set at the end, no progress of the dual objective -- is queued
// (work list: kShards segments, one counter each, the screening pass's format) and the wavefront kernel solves it from
// scratch.  The pass also does what the screening pass does (iteration 1), so it replaces it.
//
// Data: the scan pack (rows M_j, du0_j, dl0_j) and the packed Gram matrix in LDS for what a lane reads at its OWN row
// index; the per-problem bound shifts b_j = Dth_j theta in LDS, [j][lane] per wavefront (32 KB at m = 63: one workgroup
// of four wavefronts per CU, one wavefront per SIMD).  The scan -- m (n + ~25) instructions per iteration for 64
// problems, 9 of 10 instructions of the kernel -- takes its rows by scalar loads, one batch per row issued a row ahead
// through inline assembly (QpRow): with a single wavefront per SIMD nothing else hides the scalar cache's latency, and
// the compiler waits for a load it knows about right in front of its first use.  (Rows as LDS broadcast reads instead:
// 2.36 against 1.98 ms per 10^6 problems; as broadcast vector loads two rows ahead, three register sets: 2.50 -- a
// broadcast costs the vector cache its full 64 lanes x 16 bytes.)  No MFMA (n <= 12).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "lmpc_pack.hpp"
#include "lmpc_tiers.hpp"
#include "lmpc_wave_layout.hpp"

namespace lmpc {

// reals a row of the scan pack: M_j (n), du0_j, dl0_j, padded to 4, 8, 12 or 16
__host__ __device__ constexpr int qp_scan_row_reals(int n) { return n <= 2 ? 4 : (n <= 6 ? 8 : (n <= 10 ? 12 : 16)); }

// LDS of a workgroup: the scan pack (m rows of NR reals: M_j, du0_j, dl0_j), the packed Gram matrix, then the four
// wavefronts' bound shifts b[m][64]
template <int N> struct QpTiersLds {
    static constexpr int NR = qp_scan_row_reals(N);
    __host__ __device__ static constexpr int oSP() { return 0; }
    __host__ __device__ static constexpr int oG(int m) { return m * NR; }
    __host__ __device__ static constexpr int oB(int m) { return ((m * NR + m * (m + 1) / 2 + 1) & ~1); }
    __host__ __device__ static constexpr int reals(int m) { return oB(m) + 4 * m * 64; }
};

constexpr int kQpTiersMaxNth = 16;

// One row of the scan pack in scalar registers plus this lane's bound shift of that row, loaded by INLINE assembly:
// the batch of row j + 1 is issued before row j is worked on and waited for behind it (the compiler would wait for a
// load it knows about right in front of its first use -- one exposed scalar-cache round trip per row and a single
// wavefront per SIMD to hide it behind -- and folds a software prefetch written in C++ back into that).  NRD dwords
// a row: 8, 16, 24 or 32.
typedef int qp_v8i __attribute__((ext_vector_type(8)));
typedef int qp_v16i __attribute__((ext_vector_type(16)));
template <int NRD> struct QpRow;
template <> struct QpRow<8> {
    qp_v8i a;
    __device__ __forceinline__ void load(const double *p, unsigned lds_addr, double &b, double &dep) {
        asm volatile("s_load_dwordx8 %0, %3, 0x0\n\tds_read_b64 %1, %4" : "=&s"(a), "=&v"(b), "+v"(dep) : "s"(p), "v"(lds_addr) : "memory");
    }
    __device__ __forceinline__ void wait(double &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+v"(b)); }
    __device__ __forceinline__ int dw(int i) const { return a[i]; }
};
template <> struct QpRow<16> {
    qp_v16i a;
    __device__ __forceinline__ void load(const double *p, unsigned lds_addr, double &b, double &dep) {
        asm volatile("s_load_dwordx16 %0, %3, 0x0\n\tds_read_b64 %1, %4" : "=&s"(a), "=&v"(b), "+v"(dep) : "s"(p), "v"(lds_addr) : "memory");
    }
    __device__ __forceinline__ void wait(double &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+v"(b)); }
    __device__ __forceinline__ int dw(int i) const { return a[i]; }
};
template <> struct QpRow<24> {
    qp_v16i a; qp_v8i c;
    __device__ __forceinline__ void load(const double *p, unsigned lds_addr, double &b, double &dep) {
        asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, 0x40\n\tds_read_b64 %2, %5"
                     : "=&s"(a), "=&s"(c), "=&v"(b), "+v"(dep) : "s"(p), "v"(lds_addr) : "memory");
    }
    __device__ __forceinline__ void wait(double &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(c), "+v"(b)); }
    __device__ __forceinline__ int dw(int i) const { return i < 16 ? a[i] : c[i - 16]; }
};
template <> struct QpRow<32> {
    qp_v16i a, c;
    __device__ __forceinline__ void load(const double *p, unsigned lds_addr, double &b, double &dep) {
        asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\tds_read_b64 %2, %5"
                     : "=&s"(a), "=&s"(c), "=&v"(b), "+v"(dep) : "s"(p), "v"(lds_addr) : "memory");
    }
    __device__ __forceinline__ void wait(double &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(c), "+v"(b)); }
    __device__ __forceinline__ int dw(int i) const { return i < 16 ? a[i] : c[i - 16]; }
};

template <class F, int... K>
__device__ __forceinline__ void qp_tiers_each(F &f, std::integer_sequence<int, K...>) {
    (f(std::integral_constant<int, K>{}), ...);
}

template <int N>
__global__ __launch_bounds__(256) void qp_tiers_kernel(
    const WaveLayout P, const double *__restrict__ C, const double *__restrict__ theta, double *__restrict__ X,
    int32_t *__restrict__ exitflag, int32_t *__restrict__ iters, uint64_t *__restrict__ active,
    int32_t *__restrict__ list, int32_t *__restrict__ count, const long long seg_cap, const long long nprob,
    const double *__restrict__ SP, const unsigned long long soft_mask) {
    typedef QpTiersLds<N> Ly;
    constexpr int KMAX = N + 1;
    constexpr int NR = Ly::NR;                                    // reals a row of the scan pack
    constexpr int NS = KMAX * (KMAX - 1) / 2;
    extern __shared__ __align__(16) double lds[];
    const int m = P.m, nth = P.nth, nout = P.nout;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < m * NR; e += 256) lds[Ly::oSP() + e] = SP[e];
    for (int e = tid; e < m * (m + 1) / 2; e += 256) lds[Ly::oG(m) + e] = C[P.oG + e];
    __syncthreads();
    const double *sSP = lds + Ly::oSP(), *sG = lds + Ly::oG(m);
    double *sB = lds + Ly::oB(m) + wv * (m * 64) + lane;          // b_j of this lane at sB[j * 64]
    const double ptol = P.primal_tol, dtol = P.dual_tol, ztol = P.zero_tol, fbound = P.fval_bound, prog = P.progress_tol;
    const int cyc_tol = P.cycle_tol;
    const double rho = P.rho_soft;                 // SOFT rows (bit j of soft_mask): slack weighted 1 / rho_soft (utils.jl:329-364)

    const long long gw = (long long)blockIdx.x * 4 + wv, nw = (long long)gridDim.x * 4;     // (nw % kShards == 0)
    const int shard = (int)(gw % kShards);
    for (long long base = gw * 64; base < nprob; base += nw * 64) {
        const long long idx = base + lane;
        const bool mine = idx < nprob;
        const long long pid = mine ? idx : 0;
        const double *th = theta + pid * nth;
        // ---- bound shifts of this parameter point (mpc_update_qp.c:1-10), sums in index order
        {
            double thv[kQpTiersMaxNth];
#pragma unroll
            for (int t = 0; t < kQpTiersMaxNth; t++) thv[t] = t < nth ? th[t] : 0.0;
            for (int j = 0; j < m; j++) {
                const double *dr = C + P.oDth + j * nth;
                double sh = 0.0;
#pragma unroll
                for (int t = 0; t < kQpTiersMaxNth; t++)
                    if (t < nth) sh = __builtin_fma(dr[t], thv[t], sh);
                sB[j * 64] = sh;
            }
        }
        double u[N];
#pragma unroll
        for (int c = 0; c < N; c++) u[c] = 0.0;
        unsigned long long act = 0ull, low = 0ull;
        // one scan of all rows at the iterate u: the most violated inactive row (ties: the first), and whether a row of
        // the working set is violated
        double min_val;
        int add;
        bool addlow, broken;
        // (rows of the scan pack -- M_j, du0_j, dl0_j, NR reals a row -- come by scalar loads, one batch per row, the batch
        // of row j + 1 in flight while row j is worked on: QpRow; no short-circuit logic: it would turn into branches)
        typedef const double __attribute__((address_space(3))) *lds_cdp;
        const unsigned sB_addr = (unsigned)(size_t)(lds_cdp)sB;
        auto scan = [&](const bool at_zero) {
            min_val = -ptol; add = -1; addlow = false; broken = false;
            double worst = 0.0;
            typedef QpRow<2 * NR> Row;
            Row ra, rb;                            // two register sets used alternately: no copies between rows
            double ba, bb, dep = 0.0;             // (`dep`: a value the row's work depends on, so that the work stays
            auto fetch = [&](Row &r, const int jr, double &b, double &d) {       //  behind the issue of the next row's loads)
                r.load(SP + jr * NR, sB_addr + (unsigned)jr * 512u, b, d);
            };
            fetch(ra, 0, ba, dep);
            ra.wait(ba);
            auto row_work = [&](const Row &r, const double b, const int j, const double u0) {
                auto rd = [&](int q) -> double { return __hiloint2double(r.dw(2 * q + 1), r.dw(2 * q)); };
                double Mu = 0.0;
                if (!at_zero) {
#pragma unroll
                    for (int c = 0; c < N; c++) Mu = __builtin_fma(rd(c), c == 0 ? u0 : u[c], Mu);
                }
                const double vu = (rd(N) + b) - Mu;
                const double vl = -((rd(N + 1) + b) - Mu);
                // most violated INACTIVE row so far, upper bound before lower, ties to the first; and the worst value over
                // all rows (two v_min): if no inactive row is violated at the end, anything below -primal_tol is a row of
                // the working set -- the factorisation has broken down
                const bool inact = !((act >> j) & 1ull);
                const bool tu = inact & (vu < min_val);
                const bool hit = tu | (inact & (vl < min_val));
                const double cand = tu ? vu : vl;
                min_val = hit ? cand : min_val;
                add = hit ? j : add;
                addlow = hit ? !tu : addlow;
                // (an ACTIVE SOFT row may be violated: that is its slack)
                const bool slack_row = ((soft_mask >> j) & 1ull) != 0ull && !inact;
                worst = __builtin_fmin(worst, slack_row ? 0.0 : __builtin_fmin(vu, vl));
            };
            int j = 0;
            for (; j + 1 < m; j += 2) {
                double u0 = u[0];
                fetch(rb, j + 1, bb, u0);
                row_work(ra, ba, j, u0);
                rb.wait(bb);
                const int jn = j + 2 < m ? j + 2 : m - 1;
                double u1 = u[0];
                fetch(ra, jn, ba, u1);
                row_work(rb, bb, j + 1, u1);
                ra.wait(ba);
            }
            if (m & 1) row_work(ra, ba, m - 1, u[0]);
            broken = add < 0 && worst < -ptol;
        };
        // ---- iteration 1: empty working set, u = 0
        scan(true);
        double SL[NS > 0 ? NS : 1], Dinv[KMAX], xl[KMAX];
        int wrow[KMAX];
#pragma unroll
        for (int i = 0; i < NS; i++) SL[i] = 0.0;
#pragma unroll
        for (int i = 0; i < KMAX; i++) { Dinv[i] = 0.0; xl[i] = 0.0; wrow[i] = 0; }
        bool running = mine && add >= 0;
        bool finished = mine && add < 0;            // the unconstrained optimum is feasible
        int flag_fin = EXIT_OPTIMAL, iter_fin = 1, cyc = 0, nsoft_act = 0;
        unsigned softpos = 0u;                      // bit i: working-set position i holds a SOFT row
        double best = -1.0, fval = 0.0;

        // (one generic lambda instantiated per tier: a `for` over the tiers is too large for the unroller from n = 8 on,
        // and a rolled loop would index the register arrays at run time)
        auto tier = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if (!__any(running)) return;
            // ---- append row `add` at position k (ldl_add with na = k), then the progress guard of the iteration
            // that chose it
            const int j = add < 0 ? 0 : add;
            double row[KMAX > 1 ? KMAX - 1 : 1];
#pragma unroll
            for (int t = 0; t < k; t++) {
                const int a = wrow[t];
                row[t] = sG[a >= j ? lmpc_tri(a) + j : lmpc_tri(j) + a];
            }
            const bool jsoft = ((soft_mask >> j) & 1ull) != 0ull;
            double dnew = sG[lmpc_tri(j) + j] + (jsoft ? rho : 0.0);
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
            // (n + 1 HARD rows in R^n are always dependent, whatever rounding says)
            const bool singular = (dnew < ztol) || (!jsoft && k - nsoft_act >= P.n);
            const double bj = sB[j * 64];
            const double rk = addlow ? -(sSP[j * NR + N + 1] + bj) : -(sSP[j * NR + N] + bj);
#pragma unroll
            for (int t = 0; t < k; t++) SL[lmpc_sl(k, t)] = row[t];
            Dinv[k] = singular ? 0.0 : 1.0 / dnew;
            wrow[k] = j;
            act |= running ? (1ull << j) : 0ull;
            low |= (running && addlow) ? (1ull << j) : 0ull;
            nsoft_act += (running && jsoft) ? 1 : 0;
            softpos |= (running && jsoft) ? (1u << k) : 0u;
            {
                const bool stall = fval - best < prog;
                cyc = stall ? cyc + 1 : 0;
                best = stall ? best : fval;
                running = running && !(stall && cyc > cyc_tol);          // EXIT_CYCLE: the wavefront kernel reports it
            }
            if (!__any(running)) return;
            // ---- iteration k + 2 on k + 1 rows
            // singular working set: direction p (M_W' p = 0, p_k = +-1); no blocking multiplier => infeasible
            bool sblocked = false;
            if (__any(running && singular)) {
                double p[KMAX];
#pragma unroll
                for (int i = k - 1; i >= 0; i--) {
                    double acc = -row[i];
#pragma unroll
                    for (int t = k - 1; t > i; t--) acc = __builtin_fma(-SL[lmpc_sl(t, i)], p[t], acc);
                    p[i] = acc;
                }
                p[k] = 1.0;
#pragma unroll
                for (int i = 0; i <= k; i++) {
                    const double pi = addlow ? -p[i] : p[i];
                    const bool isl = (low >> wrow[i]) & 1ull;
                    const bool ok = isl ? (pi < dtol) : (pi > -dtol);
                    sblocked = sblocked || !ok;
                }
            }
            const bool fin_inf = running && singular && !sblocked;
            running = running && !singular;                              // (singular and blocked: a removal, next kernel)
            // (L D L') lam* = rhs: the new entry of the forward solve, the backward solve
            double ls[KMAX];
            {
                double acc = rk;
#pragma unroll
                for (int t = 0; t < k; t++) acc = __builtin_fma(-SL[lmpc_sl(k, t)], xl[t], acc);
                xl[k] = acc;
            }
#pragma unroll
            for (int i = k; i >= 0; i--) {
                double acc = xl[i] * Dinv[i];
#pragma unroll
                for (int t = k; t > i; t--) acc = __builtin_fma(-SL[lmpc_sl(t, i)], ls[t], acc);
                ls[i] = acc;
            }
            bool blocked = false;
#pragma unroll
            for (int i = 0; i <= k; i++) {
                const bool isl = (low >> wrow[i]) & 1ull;
                const bool ok = isl ? (ls[i] < dtol) : (ls[i] > -dtol);
                blocked = blocked || !ok;
            }
            running = running && !blocked;                               // a removal: the next kernel
            // primal iterate and dual objective
            double un[N];
#pragma unroll
            for (int c = 0; c < N; c++) un[c] = 0.0;
#pragma unroll
            for (int i = 0; i <= k; i++) {
                double mi[N];
                lmpc_lds_run<N>(sSP + wrow[i] * NR, 0, mi);
                const double l = ls[i];
#pragma unroll
                for (int c = 0; c < N; c++) un[c] = __builtin_fma(-mi[c], l, un[c]);
            }
            double fv = 0.0, soft = 0.0;
#pragma unroll
            for (int i = 0; i <= k; i++) soft = ((softpos >> i) & 1u) ? __builtin_fma(ls[i] * ls[i], rho, soft) : soft;
#pragma unroll
            for (int c = 0; c < N; c++) fv = __builtin_fma(un[c], un[c], fv);
            fv = fv + soft;
#pragma unroll
            for (int c = 0; c < N; c++) u[c] = running ? un[c] : u[c];
            fval = running ? fv : fval;
            const bool fin_bound = running && (fv > fbound);
            running = running && !fin_bound;
            scan(false);
            const bool fin_opt = running && add < 0 && !broken;
            running = running && add >= 0;                               // (add < 0 and broken: EXIT_CYCLE, next kernel)
            const bool fin = fin_inf || fin_bound || fin_opt;
            finished = finished || fin;
            flag_fin = fin ? (fin_opt ? (soft > ptol ? (int)EXIT_SOFT_OPTIMAL : (int)EXIT_OPTIMAL) : (int)EXIT_INFEASIBLE) : flag_fin;
            iter_fin = fin ? k + 2 : iter_fin;
            running = running && (k + 1 < KMAX);
        };
        qp_tiers_each(tier, std::make_integer_sequence<int, KMAX>{});

        // ---- outputs of the finished problems: x = Rout u + x0 + Xth theta (mpc_update_qp.c:14-22); the others are queued
        if (finished) {
            for (int kk = 0; kk < nout; kk++) {
                const double *ro = C + P.oRout + kk * N, *xt = C + P.oXth + kk * nth;
                double xs = 0.0, sh = C[P.ox0 + kk];
#pragma unroll
                for (int c = 0; c < N; c++) xs = __builtin_fma(ro[c], u[c], xs);
                for (int t = 0; t < nth; t++) sh = __builtin_fma(xt[t], th[t], sh);
                X[pid * nout + kk] = xs + sh;
            }
            exitflag[pid] = flag_fin;
            if (iters) iters[pid] = iter_fin;
            if (active) {
                const unsigned long long up = act & ~low, lo = act & low;
                unsigned long long w0 = up, w1 = 0ull;
                if (m < 64) { w0 |= lo << m; if (m > 0) w1 = lo >> (64 - m); }
                else w1 = lo;
                active[pid * P.words] = w0;
                if (P.words > 1) active[pid * P.words + 1] = w1;
            }
        }
        const bool queue = mine && !finished;
        const unsigned long long qmask = __ballot(queue);
        if (qmask != 0ull) {
            int qbase = 0;
            if (lane == 0) qbase = atomicAdd(&count[shard * kCountStride], __popcll(qmask));
            qbase = __shfl(qbase, 0);
            if (queue) list[(long long)shard * seg_cap + qbase + __popcll(qmask & ((1ull << lane) - 1ull))] = (int32_t)pid;
        }
    }
}

}  // namespace lmpc
