// FOUR QPs per wavefront: the dual active-set kernel on the DPP rows of a wavefront (round 5).
//
// The one-QP-per-wavefront kernel (lmpc_wave_kernel.hpp) is issue bound on mostly empty wavefronts: a step of one of
// its serial chains is 2 v_readlane + 1 fma with ONE useful lane, a working set of <= 31 rows leaves half the lanes
// idle by construction (config 3: 25 000 vector + 20 000 scalar instructions per problem, 8 % of the lane-flops they
// offer are the algorithm's).  Here a problem owns a ROW of 16 lanes -- the unit the DPP cross-lane network works on --
// and a wavefront carries four problems through the iteration in lock step:
//
//  * working-set position p of a problem lives on lane p % 16 of its row, register slot p / 16 (S slots: working sets
//    of up to 16 S rows); variable k on lane k % 16, slot k / 16 (NS slots); constraint j on lane j % 16, slot j / 16
//    (MS slots);
//  * the operand a chain step broadcasts -- v_t of a triangular sweep, lam*_i of the primal step, u_k of the constraint
//    scan -- is delivered PER ROW by the DPP control row_newbcast:t (lane t of each row to all lanes of that row; the
//    only DPP control the 64-bit ALU takes, gfx90a and later): one v_mov_b64_dpp (or the modifier of a v_fmac_f64_dpp)
//    serves four problems, where the wavefront kernel spends two v_readlane and an SGPR round trip on one.  t must be a
//    literal, so every chain is unrolled over its positions and skipped block-wise by wave-uniform tests;
//  * what is one number per problem (working-set size, iteration count, step length, ...) is a ROW-UNIFORM vector
//    register; the four problems of a wavefront take different decisions, so the phases of an iteration (stationary
//    point, blocking test, then row append OR row removal) run under per-row predicates -- selects, never EXEC masks
//    around DPP instructions;
//  * the factor L of each problem sits in LDS, square and zero padded exactly as in the wavefront kernel (a sweep step
//    is one fma on all lanes: fma(-0, v_t, v) = v), column major with an odd leading dimension LDC; the four problems of
//    a wavefront are 16 (mod 32) reals apart, so that the two rows a 32-lane LDS phase serves never share a bank;
//  * a row that finishes a problem writes its outputs and takes the next problem from the batch by itself (tickets of
//    a few problems each from one global counter), the other three rows keep iterating.
//
// Arithmetic: statement for statement the fma chains of oracle/daqp_ldp_oracle.c (mode 0, the n-chain form) -- the
// results are bit-identical to the oracle's and to the wavefront kernel's (x, exit flag, iteration count, active set).
//
// Covers: cold plain solves, binary64 / binary32, n <= 16 NS, m <= 16 MS, hard / SOFT / IMMUTABLE rows, no rows
// flagged ACTIVE or BINARY, working sets up to min(16 S, LDC) rows; a point that outgrows that is listed for the
// wavefront kernel (exit flag -7 inside this pass), exactly like the first of that kernel's two passes.
//
// Replaces, per problem: mpc_update_qp (reference codegen/mpc_update_qp.c:1-10), daqp_ldp incl. soft constraints
// ([EXT] libdaqp, called at mpc_update_qp.c:48 / utils.jl:282) and mpc_get_solution (mpc_update_qp.c:14-22).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "lmpc_pack.hpp"
#include "lmpc_wave_layout.hpp"
#include "lmpc_wave_kernel.hpp"      // (the wave-level helpers wv_*)

namespace lmpc {

// ---- row-level helpers ---------------------------------------------------------------------------------------------
// lane T of each 16-lane row to all lanes of that row (DPP row_newbcast:T)
template <int T, typename V> __device__ __forceinline__ V rw_bc(V v) {
    return __builtin_amdgcn_mov_dpp(v, 0x150 + (T & 15), 0xF, 0xF, false);
}
template <int I, int N, class F> __device__ __forceinline__ void rw_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rw_static_for<I + 1, N>(f);
    }
}
// descending: N-1 ... I
template <int I, int N, class F> __device__ __forceinline__ void rw_static_rfor(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, N - 1>{});
        rw_static_rfor<I, N - 1>(f);
    }
}
// minimum over the 16 lanes of each row, returned in every lane of the row
template <typename V> __device__ __forceinline__ V rw_min(V v) {
    v = wv_min2(v, wv_dpp<0xB1>(v));
    v = wv_min2(v, wv_dpp<0x4E>(v));
    v = wv_min2(v, wv_dpp<0x141>(v));
    v = wv_min2(v, wv_dpp<0x140>(v));
    return v;
}
__device__ __forceinline__ int rw_or(int v) {
    v |= wv_dpp<0xB1>(v);
    v |= wv_dpp<0x4E>(v);
    v |= wv_dpp<0x141>(v);
    v |= wv_dpp<0x140>(v);
    return v;
}
// lane `src` (0..15, row-uniform, run time) of each row to all lanes of the row: through the LDS crossbar
__device__ __forceinline__ int rw_pick(int v, int src, int rowbase) {
    return __builtin_amdgcn_ds_bpermute((rowbase + (src & 15)) << 2, v);
}
__device__ __forceinline__ float rw_pick(float v, int src, int rowbase) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((rowbase + (src & 15)) << 2, __float_as_int(v)));
}
__device__ __forceinline__ double rw_pick(double v, int src, int rowbase) {
    const int a = (rowbase + (src & 15)) << 2;
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(a, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(a, __double2loint(v)));
}
// lane i <- lane i + 1 inside a row (lane 15: unspecified; the callers overwrite it)
__device__ __forceinline__ int rw_shl1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x101, 0xF, 0xF, false); }
__device__ __forceinline__ float rw_shl1(float v) { return __int_as_float(rw_shl1(__float_as_int(v))); }
__device__ __forceinline__ double rw_shl1(double v) {
    return __hiloint2double(rw_shl1(__double2hiint(v)), rw_shl1(__double2loint(v)));
}
// maximum over the four rows of a row-uniform int, wave-uniform (scalar)
__device__ __forceinline__ int rw_max4(int v) {
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32),
              d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
__device__ __forceinline__ int rw_min4(int v) {
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32),
              d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
__device__ __forceinline__ bool rw_any(bool p) { return __ballot(p) != 0ull; }

constexpr int kRowPosFlagSoft = 1 << 16, kRowPosFlagImm = 1 << 17, kRowPosFlagLow = 1 << 18;
constexpr int kRowBig = 0x7fffffff;
#ifndef LMPC_ROW_WPE
#define LMPC_ROW_WPE 2      // resident wavefronts per SIMD the instantiations are register-budgeted for
#endif

// reals of LDS one problem's factor takes: (cap - 1) columns of LDC rows, rounded up to 16 (mod 32)
// reals per row of the staged M' (odd)
__host__ __device__ constexpr int row_mpad(int ms) { return 16 * ms + 1; }
__host__ __device__ constexpr int row_problem_stride(int cap, int ldc) {
    const int need = (cap - 1) * ldc;
    int ps = (need + 31) / 32 * 32 + 16;
    if (ps - 32 >= need) ps -= 32;
    return ps;
}

// R: arithmetic type.  S / NS / MS: register slots of 16 working-set positions / variables / constraints.  LDC: leading
// dimension of the factor (odd; the launch's capacity P.cap <= min(LDC, 16 S)).
template <typename R, int S, int NS, int MS, int LDC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LMPC_ROW_WPE))) void row_kernel(
    const WaveLayout P, const R *__restrict__ C, const int32_t *__restrict__ Sg, const R *__restrict__ theta,
    R *__restrict__ X, int32_t *__restrict__ exitflag, int32_t *__restrict__ iters, uint64_t *__restrict__ active,
    int32_t *__restrict__ queue, int qchunk_arg, long long nprob, const int32_t *__restrict__ list,
    const int32_t *__restrict__ count, int32_t *__restrict__ count_next, long long seg_cap,
    int32_t *__restrict__ ovf_list, int32_t *__restrict__ ovf_count, int32_t *__restrict__ queue_next,
    int32_t *__restrict__ ovf_next, int32_t *__restrict__ ovf_next1, unsigned long long *__restrict__ stat,
    volatile unsigned long long *__restrict__ stat_host) {
    static_assert((LDC & 1) == 1, "odd leading dimension");
    static_assert(LDC <= 16 * S + 1, "rows of the factor live on S slots");
    // the counters of the next launch / call on this handle, and the working-set statistics' host copy: the duties of
    // every wavefront-kernel launch (lmpc_wave_kernel.hpp, lmpc_wave_launch.hpp)
    if (blockIdx.x == 0 && threadIdx.x == 0) { *queue_next = 0; *ovf_next = 0; *ovf_next1 = 0; }
    if (stat != nullptr && stat_host != nullptr && blockIdx.x == 0 && threadIdx.x < 4) {
        unsigned long long sum = 0ull;
        for (int sidx = 0; sidx < 64; sidx++) sum += stat[sidx * 16 + threadIdx.x];
        stat_host[threadIdx.x] = sum;
    }

    extern __shared__ __align__(16) unsigned char lds_raw[];
    R *lds = reinterpret_cast<R *>(lds_raw);
    const int lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, rowbase = lane & 48, g = lane >> 4;
    const int n = P.n, m = P.m, nth = P.nth, cap = P.cap, nout = P.nout;
    constexpr int CAPP = 16 * S < LDC ? 16 * S : LDC;            // positions the instantiation has code for
    constexpr int PS = row_problem_stride(CAPP, LDC);          // (sized for the instantiation: a sweep block may read columns up to CAPP - 2)
    // LDS: [32 zeros][M': ceil4(n) rows of MPAD reals, zero padded][factors: nwv * 4 problems, PS reals each][sense flags: m ints]
    // ONE copy of the problem matrix serves both passes over it: the constraint scan reads ROW k of M' (lane = constraint,
    // consecutive addresses), the primal step reads COLUMN w of it (lane = variable, stride MPAD -- odd, so the 16 lanes
    // of a row hit 16 different banks).  MPAD is a compile-time constant: every address of the scan is one per-lane
    // base register plus an immediate (with a run-time stride the compiler keeps one address register per (k, slot)
    // alive across the whole kernel: 64 to 320 registers).
    constexpr int MPAD = row_mpad(MS);
    const int nP = (n + 3) & ~3;
    const int oZ = 0, oMt = 32, oL = oMt + nP * MPAD;
    int32_t *sens = reinterpret_cast<int32_t *>(lds + oL + nwv * 4 * PS);
    for (int i = threadIdx.x; i < 32; i += blockDim.x) lds[oZ + i] = (R)0;
    for (int i = threadIdx.x; i < nP * MPAD; i += blockDim.x) {
        const int k = i / MPAD, j = i - k * MPAD;
        lds[oMt + i] = (k < n && j < m) ? C[P.oMt + k * m + j] : (R)0;
    }
    for (int i = threadIdx.x; i < nwv * 4 * PS; i += blockDim.x) lds[oL + i] = (R)0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) sens[i] = Sg[i];
    __syncthreads();

    const R primal_tol = (R)P.primal_tol, dual_tol = (R)P.dual_tol, zero_tol = (R)P.zero_tol,
            progress_tol = (R)P.progress_tol, rho_soft = (R)P.rho_soft, fbound = (R)P.fval_bound;
    const R kInf = wv_lim<R>::inf();
    // constant pack and parameter records through buffer resources: scalar base + this lane's 32-bit index, no 64-bit
    // per-lane pointers (lmpc_wave_kernel.hpp)
    const __amdgpu_buffer_rsrc_t crs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<R *>(C), 0, P.nC * (int)sizeof(R), 0x00020000);
    auto ldc = [&](int soff, int voff) -> R { return wv_bufld(crs, (unsigned)voff, (unsigned)soff, R()); };   // C[soff + voff]
    const __amdgpu_buffer_rsrc_t trs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<R *>(theta), 0, (int)(nprob * nth * (long long)sizeof(R)), 0x00020000);
    auto ldth = [&](int voff) -> R { return wv_bufld(trs, (unsigned)voff, 0u, R()); };                       // theta[voff]
    // The unrolled chains below run in BLOCKS of a few steps behind one wave-uniform test each; inside a block every step
    // runs (steps beyond a row's sizes multiply by the zeros the padding guarantees).  The empty statement keeps the
    // compiler from folding a block's test into selects -- which turns the whole unrolled chain into one basic block
    // whose loads are all hoisted to its head (first build: 390 registers).
#define RW_BLOCK() asm volatile("" ::: "memory")

    // element (row p, column t) of this row's factor: lds[Lg + t * LDC + p]
    const int Lg = oL + (wv * 4 + g) * PS;
    int pos[S], fo[S], bo[S];          // this lane's positions; offsets of its ROW (forward sweeps) and of its COLUMN
#pragma unroll
    for (int s = 0; s < S; s++) {
        pos[s] = li + 16 * s;
        fo[s] = Lg + (pos[s] < cap ? pos[s] : 0);                       // (row 0 has no entries: zeros in every column)
        bo[s] = pos[s] < cap - 1 ? Lg + pos[s] * LDC : oZ;              // (beyond the last column: the block of zeros)
    }
    int jc[MS], mcol[NS];
#pragma unroll
    for (int r = 0; r < MS; r++) jc[r] = li + 16 * r < m ? li + 16 * r : m - 1;
#pragma unroll
    for (int s = 0; s < NS; s++) mcol[s] = oMt + (li + 16 * s < n ? li + 16 * s : n - 1) * MPAD;   // this lane's variables: their rows of M'
    const int mrow = oMt + li;                                  // this lane's constraints: column li + 16 r of M'

    // ---- the state of this row's problem
    int live = 0, dead = 0;                        // row-uniform: a problem is running / the batch is exhausted
    int pid = 0, na = 0, sing = -1, iter = 1, cyc = 0, nsoft = 0, napk = 0, ydirty = 0;
    R best = (R)-1, fval = (R)0, soft_slack = (R)0;
    int ws[S];                                     // per position: row index | kRowPosFlag*
    R D[S], Dinv[S], lam[S], ls[S], rhs[S], y[S];
    R u[NS], dub[MS], dlb[MS];
    unsigned actb = 0u, lowb = 0u;                 // bit r: this lane's row of slot r is active / active at its lower bound
#pragma unroll
    for (int s = 0; s < S; s++) { ws[s] = 0; D[s] = Dinv[s] = lam[s] = ls[s] = rhs[s] = y[s] = (R)0; }
#pragma unroll
    for (int s = 0; s < NS; s++) u[s] = (R)0;
#pragma unroll
    for (int r = 0; r < MS; r++) { dub[r] = (R)0; dlb[r] = (R)0; }

    // ---- problems: tickets of qchunk consecutive problems per row; the first from the row's index in the grid, further
    // ones from the shared counter (drawn one ticket ahead)
    // Work-list mode (list != nullptr): a pass in front (screening, lmpc_screen_kernel.hpp) has finished what needs no
    // iterations and left the others in the kShards segments of `list` (segment s: count[s * kCountStride] entries from
    // list[s * seg_cap]); lane s of every wavefront holds segment s's range of positions in the concatenated list.
    const long long nrows = (long long)gridDim.x * nwv * 4;
    const long long myrow = ((long long)blockIdx.x * nwv + wv) * 4 + g;
    long long ntotal = nprob;
    int seg_end = 0, seg_beg = 0;
    int qchunk = qchunk_arg;
    if (list != nullptr) {
        static_assert(kShards == 64, "one work-list segment per lane");
        if (count_next != nullptr && blockIdx.x == 0 && threadIdx.x < 64) count_next[threadIdx.x * kCountStride] = 0;
        const int c = count[lane * kCountStride];
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        seg_end = incl; seg_beg = incl - c;
        ntotal = (long long)__builtin_amdgcn_readlane(incl, 63);
        const long long q = ntotal / (32 * nrows);               // (the host cannot know the list's length)
        qchunk = q < 1 ? 1 : (q > 16 ? 16 : (int)q);
    }
    long long cur = myrow * qchunk, endc = cur + qchunk < ntotal ? cur + qchunk : ntotal;
    long long chunk_static = myrow;
    int ticket = 0;
    if (queue != nullptr && li == 0) ticket = atomicAdd(queue, 1);

    // ---- sweeps over the factor.  forward: v_p -= L(p,t) v_t for t = 0 .. top-1 in order; backward: v_i -= L(t,i) v_t
    // for t = top .. 1 descending.  `top` is wave-uniform (the largest of the four rows); a row whose working set is
    // smaller reads zeros (its factor's rows beyond na are zeros).
    auto sweep_fwd = [&](R (&v)[S], int nmax) {
        constexpr int CH = 8;
        rw_static_for<0, (CAPP + CH - 1) / CH>([&](auto B) {
            constexpr int t0 = decltype(B)::value * CH;
            if (t0 + 1 < nmax) {
                RW_BLOCK();
                R Lr[CH][S];
#pragma unroll
                for (int q = 0; q < CH; q++)
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const int t = t0 + q < CAPP - 1 ? t0 + q : CAPP - 2;
                        Lr[q][s] = lds[fo[s] + t * LDC];
                    }
                rw_static_for<0, CH>([&](auto Q) {
                    constexpr int t = t0 + decltype(Q)::value;
                    if constexpr (t + 1 < CAPP) {
                        const R vt = rw_bc<t>(v[t >> 4]);
#pragma unroll
                        for (int s = S - 1; s >= 0; s--)
                            if (16 * s + 15 > t) v[s] = wv_fma(-Lr[decltype(Q)::value][s], vt, v[s]);
                    }
                });
            }
        });
    };
    auto sweep_bwd = [&](R (&v)[S], int top) {
        constexpr int CH = 8;
        constexpr int NB = (CAPP - 1 + CH - 1) / CH;              // steps t = CAPP-1 .. 1 in blocks from the top
        rw_static_for<0, NB>([&](auto B) {
            constexpr int thi = CAPP - 1 - decltype(B)::value * CH;   // this block: t = thi .. thi-CH+1
            constexpr int tlo = thi - CH + 1 > 1 ? thi - CH + 1 : 1;
            if (top >= tlo) {
                RW_BLOCK();
                R Lc[CH][S];
#pragma unroll
                for (int q = 0; q < CH; q++)
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const int t = thi - q > 1 ? thi - q : 1;
                        Lc[q][s] = lds[bo[s] + t];
                    }
                rw_static_for<0, CH>([&](auto Q) {
                    constexpr int t = thi - decltype(Q)::value;
                    if constexpr (t >= 1) {
                        const R vt = rw_bc<t>(v[t >> 4]);
#pragma unroll
                        for (int s = 0; s < S; s++)
                            if (16 * s < t) v[s] = wv_fma(-Lc[decltype(Q)::value][s], vt, v[s]);
                    }
                });
            }
        });
    };

    for (;;) {
        // =============================================================== a row without a problem takes the next one
        {
            const bool want = !live && !dead;
            if (rw_any(want)) {
                const bool newchunk = want && cur >= endc;
                if (rw_any(newchunk)) {
                    long long ch;
                    if (queue != nullptr) {
                        const int tk = rw_bc<0>(ticket);
                        ch = nrows + (long long)tk;
                        if (newchunk && li == 0) ticket = atomicAdd(queue, 1);
                    } else {
                        ch = chunk_static + nrows;
                    }
                    if (newchunk) {
                        chunk_static = ch;
                        cur = ch * qchunk;
                        endc = cur + qchunk < ntotal ? cur + qchunk : ntotal;
                    }
                }
                const bool got = want && cur < ntotal && cur < endc;
                dead = (want && !got) ? 1 : dead;
                if (rw_any(got)) {
                    int npid = got ? (int)cur : 0;
                    if (list != nullptr) {
                        // position in the concatenated list -> (segment, offset) -> problem, row by row (scalar)
#pragma unroll
                        for (int gg = 0; gg < 4; gg++) {
                            if (__builtin_amdgcn_readlane(got ? 1 : 0, 16 * gg)) {
                                const int ix = __builtin_amdgcn_readlane((int)cur, 16 * gg);
                                int sgm = (int)__popcll(__ballot(seg_end <= ix));
                                sgm = sgm < 63 ? sgm : 63;
                                const int off = ix - __builtin_amdgcn_readlane(seg_beg, sgm);
                                const int p = list[(long long)sgm * seg_cap + off];
                                npid = g == gg ? p : npid;
                            }
                        }
                    }
                    cur = got ? cur + 1 : cur;
                    const int tho = npid * nth;
                    R b[MS];
#pragma unroll
                    for (int r = 0; r < MS; r++) b[r] = (R)0;
                    for (int t = 0; t < nth; t++) {
                        const R tv = ldth(tho + t);
#pragma unroll
                        for (int r = 0; r < MS; r++) b[r] = wv_fma(ldc(P.oDth + t, jc[r] * nth), tv, b[r]);
                    }
#pragma unroll
                    for (int r = 0; r < MS; r++) {
                        const R du = ldc(P.odu, jc[r]) + b[r], dl = ldc(P.odl, jc[r]) + b[r];
                        dub[r] = got ? du : dub[r];
                        dlb[r] = got ? dl : dlb[r];
                    }
                    if (got) {
                        pid = npid; live = 1; na = 0; sing = -1; iter = 1; cyc = 0; nsoft = 0; napk = 0; ydirty = 0;
                        best = (R)-1; fval = (R)0; soft_slack = (R)0; actb = 0u; lowb = 0u;
#pragma unroll
                        for (int s = 0; s < S; s++) { ws[s] = 0; D[s] = Dinv[s] = lam[s] = ls[s] = rhs[s] = y[s] = (R)0; }
#pragma unroll
                        for (int s = 0; s < NS; s++) u[s] = (R)0;
                    }
                }
            }
            if (!rw_any(live != 0)) break;                       // every row of the wavefront has run out of problems
        }

        int flag = 0;                                            // != 0: this row's problem ends in this trip
        bool fin = false;
        const bool act = live != 0;
        if (act && iter >= P.iter_limit) { flag = EXIT_ITERLIMIT; fin = true; }
        const bool run = act && !fin;
        const bool sgl = sing >= 0;
        const int namax = rw_max4(run ? na : 0);

        // =============================================================== stationary point / singular direction
        {
            const bool dirty = run && !sgl && ydirty != 0;
            if (rw_any(dirty)) {
                R v[S];
#pragma unroll
                for (int s = 0; s < S; s++) v[s] = (dirty && pos[s] < na) ? rhs[s] : (R)0;
                sweep_fwd(v, rw_max4(dirty ? na : 0));
#pragma unroll
                for (int s = 0; s < S; s++) y[s] = dirty ? v[s] : y[s];
                ydirty = dirty ? 0 : ydirty;
            }
            R v[S];
#pragma unroll
            for (int s = 0; s < S; s++) v[s] = run ? y[s] * Dinv[s] : (R)0;
            int lowsg = 0;
            if (rw_any(run && sgl)) {
                const int sgc = sgl ? sing : 1;
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const R lv = lds[bo[s] + (sgc > 1 ? sgc : 1)];       // L(sing, pos)
                    if (sgl) v[s] = (run && pos[s] < sing) ? -lv : (R)0;
                }
                int wsel = ws[0];
#pragma unroll
                for (int s = 1; s < S; s++) wsel = (sgc >> 4) == s ? ws[s] : wsel;
                lowsg = (rw_pick(wsel, sgc, rowbase) & kRowPosFlagLow) ? 1 : 0;
            }
            sweep_bwd(v, namax - 1);
#pragma unroll
            for (int s = 0; s < S; s++) {
                R acc = v[s];
                if (sgl) {
                    acc = pos[s] == sing ? (R)1 : (pos[s] > sing ? (R)0 : acc);
                    acc = lowsg ? -acc : acc;
                } else {
                    acc = pos[s] < na ? acc : (R)0;
                }
                ls[s] = run ? acc : ls[s];
            }
        }

        // =============================================================== blocking multipliers: (alpha, rm) = first minimum
        int rm = -1;
        R alpha = (R)0;
        {
            bool blk[S];
            bool anyb = false;
#pragma unroll
            for (int s = 0; s < S; s++) {
                const bool okd = (ws[s] & kRowPosFlagLow) ? (ls[s] < dual_tol) : (ls[s] > -dual_tol);
                blk[s] = run && pos[s] < na && !(ws[s] & kRowPosFlagImm) && !okd;
                anyb = anyb || blk[s];
            }
            if (rw_any(anyb)) {
                R cand[S], cm = kInf;
#pragma unroll
                for (int s = 0; s < S; s++) {
                    cand[s] = sgl ? (-lam[s] / ls[s]) : (-lam[s] / (ls[s] - lam[s]));
                    cm = wv_min2(cm, blk[s] ? cand[s] : kInf);
                }
                const R gmin = rw_min(cm);
                int tp = kRowBig, bp = kRowBig;
#pragma unroll
                for (int s = S - 1; s >= 0; s--) {
                    if (blk[s] && cand[s] == gmin) tp = pos[s];
                    if (blk[s]) bp = pos[s];
                }
                tp = rw_min(tp);
                bp = rw_min(bp);
                const int r0 = tp != kRowBig ? tp : bp;              // (no position equals the minimum: NaNs -- the first blocked one)
                if (r0 != kRowBig) {
                    rm = r0;
                }
                R csel = cand[0];
#pragma unroll
                for (int s = 1; s < S; s++) csel = ((r0 & 0xffff) >> 4) == s ? cand[s] : csel;
                const R a0 = rw_pick(csel, r0 & 15, rowbase);
                alpha = rm >= 0 ? a0 : (R)0;
            }
        }
        if (run && sgl && rm < 0) { flag = EXIT_INFEASIBLE; fin = true; }
        const bool doRem = run && rm >= 0;
        const bool doAdd = run && !sgl && rm < 0;

        // =============================================================== no blocking multiplier: primal iterate, scan, append
        if (rw_any(doAdd)) {
            const int namaxA = rw_max4(doAdd ? na : 0);
            R un[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) un[s] = (R)0;
            // u = -M_W' lam*  (rows beyond a working set: lam* = 0 there, row 0 of M)
            constexpr int CHP = 4;
            rw_static_for<0, (CAPP + CHP - 1) / CHP>([&](auto B) {
                constexpr int i0 = decltype(B)::value * CHP;
                if (i0 < namaxA) {
                    RW_BLOCK();
                    R mv[CHP][NS];
                    rw_static_for<0, CHP>([&](auto Q) {
                        constexpr int i = i0 + decltype(Q)::value < CAPP ? i0 + decltype(Q)::value : CAPP - 1;
                        const int w = rw_bc<i>(ws[i >> 4]) & 0xffff;
#pragma unroll
                        for (int s = 0; s < NS; s++) mv[decltype(Q)::value][s] = lds[mcol[s] + w];
                    });
                    rw_static_for<0, CHP>([&](auto Q) {
                        constexpr int i = i0 + decltype(Q)::value;
                        if constexpr (i < CAPP) {
                            const R l = rw_bc<i>(ls[i >> 4]);
#pragma unroll
                            for (int s = 0; s < NS; s++) un[s] = wv_fma(-mv[decltype(Q)::value][s], l, un[s]);
                        }
                    });
                }
            });
#pragma unroll
            for (int s = 0; s < NS; s++) un[s] = li + 16 * s < n ? un[s] : (R)0;
            R soft = (R)0;
            if (rw_any(doAdd && nsoft > 0)) {
                rw_static_for<0, (CAPP + 3) / 4>([&](auto B) {
                    constexpr int i0 = decltype(B)::value * 4;
                    if (i0 < namaxA) {
                        RW_BLOCK();
                        rw_static_for<0, 4>([&](auto Q) {
                            constexpr int i = i0 + decltype(Q)::value;
                            if constexpr (i < CAPP) {
                                const int w = rw_bc<i>(ws[i >> 4]);          // (beyond a working set: no flags)
                                const R l = rw_bc<i>(ls[i >> 4]);
                                const R sn = wv_fma(l * l, rho_soft, soft);
                                soft = (w & kRowPosFlagSoft) ? sn : soft;
                            }
                        });
                    }
                });
            }
            // objective u'u and the row values M u in one pass over the variables
            R fv = (R)0, Mu[MS];
#pragma unroll
            for (int r = 0; r < MS; r++) Mu[r] = (R)0;
            constexpr int CHK = MS <= 4 ? 4 : 2;                   // (divides 4: the staged M' has ceil4(n) rows)
            rw_static_for<0, 16 * NS / CHK>([&](auto B) {
                constexpr int k0 = decltype(B)::value * CHK;
                if (k0 < n) {
                    RW_BLOCK();
                    R mt[CHK][MS];
#pragma unroll
                    for (int q = 0; q < CHK; q++)                              // (rows beyond n, columns beyond m: zeros)
#pragma unroll
                        for (int r = 0; r < MS; r++) mt[q][r] = lds[mrow + (k0 + q) * MPAD + 16 * r];
                    rw_static_for<0, CHK>([&](auto Q) {
                        constexpr int k = k0 + decltype(Q)::value;
                        const R v = rw_bc<k>(un[k >> 4]);
                        fv = wv_fma(v, v, fv);
#pragma unroll
                        for (int r = 0; r < MS; r++) Mu[r] = wv_fma(mt[decltype(Q)::value][r], v, Mu[r]);
                    });
                }
            });
            const R fvalN = fv + soft;
            bool addp = doAdd;
            if (doAdd) {                                         // (this trip's iterate is the row's iterate from here on)
#pragma unroll
                for (int s = 0; s < NS; s++) u[s] = un[s];
                fval = fvalN; soft_slack = soft;
            }
            if (addp && fvalN > fbound) { flag = EXIT_INFEASIBLE; fin = true; addp = false; }
            // most violated row: smallest value, ties to the lowest (row, side) index
            R mval = -primal_tol;
            int midx = -1;
            bool broken = false;
#pragma unroll
            for (int r = 0; r < MS; r++) {
                const int j = li + 16 * r;
                if (j < m && !(sens[jc[r]] & SENSE_IMMUTABLE)) {
                    const R vu = dub[r] - Mu[r];
                    const R vl = -(dlb[r] - Mu[r]);
                    if (!((actb >> r) & 1u)) {
                        if (vu < mval) { mval = vu; midx = 2 * j; }
                        else if (vl < mval) { mval = vl; midx = 2 * j + 1; }
                    } else if (!(sens[jc[r]] & SENSE_SOFT) && (vu < -primal_tol || vl < -primal_tol)) {
                        broken = true;
                    }
                }
            }
            const int anyv = rw_or(midx >= 0 ? 1 : 0), anybr = rw_or(broken ? 1 : 0);
            const R gsel = rw_min(midx >= 0 ? mval : kInf);
            int mt = rw_min((midx >= 0 && mval == gsel) ? midx : kRowBig);
            if (mt == kRowBig) mt = rw_min(midx >= 0 ? midx : kRowBig);
            if (addp && !anyv) {
                flag = anybr ? EXIT_CYCLE : (soft > primal_tol ? EXIT_SOFT_OPTIMAL : EXIT_OPTIMAL);
                fin = true; addp = false;
            }
            if (addp && na >= cap) { flag = EXIT_WSCAP; fin = true; addp = false; }

            // ---- append row jadd to the working sets of the rows with addp
            if (rw_any(addp)) {
                const int jadd = addp ? (mt >> 1) : 0;
                const bool lower = addp && (mt & 1);
                const int sj = sens[jadd];
                const bool is_soft = (sj & SENSE_SOFT) != 0;
                R q[S];
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const int a = (addp && pos[s] < na) ? (ws[s] & 0xffff) : jadd;
                    const int gi = a >= jadd ? a * (a + 1) / 2 + jadd : jadd * (jadd + 1) / 2 + a;
                    const R gv = ldc(P.oG, gi);
                    q[s] = (addp && pos[s] < na) ? gv : (R)0;
                }
                const R gjj = ldc(P.oG, jadd * (jadd + 1) / 2 + jadd);
#pragma unroll
                for (int s = 0; s < S; s++) lam[s] = addp ? ls[s] : lam[s];
                const int namaxQ = rw_max4(addp ? na : 0);
                sweep_fwd(q, namaxQ);
                R l[S];
#pragma unroll
                for (int s = 0; s < S; s++) l[s] = q[s] * Dinv[s];
                R dnew = gjj;
                if (is_soft) dnew += rho_soft;
                // the bound of row jadd that enters: from the lane and slot that own the row
                R bsel = lower ? dlb[0] : dub[0];
#pragma unroll
                for (int r = 1; r < MS; r++) bsel = (jadd >> 4) == r ? (lower ? dlb[r] : dub[r]) : bsel;
                const R rj = -rw_pick(bsel, jadd & 15, rowbase);
                R ynew = rj;
                rw_static_for<0, (CAPP + 3) / 4>([&](auto B) {
                    constexpr int i0 = decltype(B)::value * 4;
                    if (i0 < namaxQ) {
                        RW_BLOCK();
                        rw_static_for<0, 4>([&](auto Q) {
                            constexpr int i = i0 + decltype(Q)::value;
                            if constexpr (i < CAPP) {                         // (beyond a working set: l_i = 0)
                                const R lq = rw_bc<i>(l[i >> 4]);
                                dnew = wv_fma(-lq, rw_bc<i>(q[i >> 4]), dnew);
                                ynew = wv_fma(-lq, rw_bc<i>(y[i >> 4]), ynew);
                            }
                        });
                    }
                });
                const bool singular = (dnew < zero_tol) || (!is_soft && (na - nsoft) >= n);
                const R dinv = (R)1 / dnew;
#pragma unroll
                for (int s = 0; s < S; s++) {
                    if (addp && pos[s] < na) lds[bo[s] + na] = l[s];         // new row: L(na, t) written by lane t
                    if (addp && pos[s] == na) {
                        ws[s] = jadd | (is_soft ? kRowPosFlagSoft : 0) | ((sj & SENSE_IMMUTABLE) ? kRowPosFlagImm : 0) |
                                (lower ? kRowPosFlagLow : 0);
                        rhs[s] = rj; lam[s] = (R)0; ls[s] = (R)0; y[s] = ynew;
                        D[s] = singular ? (R)0 : dnew;
                        Dinv[s] = singular ? (R)0 : dinv;
                    }
                }
                if (addp && li == (jadd & 15)) {
                    actb |= 1u << (jadd >> 4);
                    if (lower) lowb |= 1u << (jadd >> 4);
                }
                if (addp) {
                    if (singular) sing = na;
                    nsoft += is_soft ? 1 : 0;
                    na++;
                    napk = na > napk ? na : napk;
                    if (fvalN - best < progress_tol) {
                        if (++cyc > P.cycle_tol) { flag = EXIT_CYCLE; fin = true; }
                    } else { best = fvalN; cyc = 0; }
                }
            }
        }

        // =============================================================== a blocking multiplier: step, drop its row
        if (rw_any(doRem)) {
#pragma unroll
            for (int s = 0; s < S; s++) {
                const R ln = sgl ? wv_fma(alpha, ls[s], lam[s]) : wv_fma(alpha, ls[s] - lam[s], lam[s]);
                lam[s] = doRem ? ln : lam[s];
            }
            const int r = doRem ? rm : 0;
            const int nao = na;
            R w[S];
#pragma unroll
            for (int s = 0; s < S; s++) {
                const R lv = lds[fo[s] + (r < cap - 1 ? r : 0) * LDC];          // L(pos, r), old row index = pos
                w[s] = (doRem && pos[s] > r && pos[s] < nao) ? lv : (R)0;
            }
            R dsel = D[0];
            int wsel = ws[0];
#pragma unroll
            for (int s = 1; s < S; s++) { dsel = (r >> 4) == s ? D[s] : dsel; wsel = (r >> 4) == s ? ws[s] : wsel; }
            R al = rw_pick(dsel, r & 15, rowbase);
            const int wsr = rw_pick(wsel, r & 15, rowbase);
            const int jrem = wsr & 0xffff, softrem = (wsr & kRowPosFlagSoft) ? 1 : 0;
            // new row i = old row i + 1 without column r (i >= r): lane c moves entry (i + 1, c') to (i, c)
            const int rlo = rw_min4(doRem ? r : kRowBig), nhi = rw_max4(doRem ? nao : 0);
            {
                int so[S];
#pragma unroll
                for (int s = 0; s < S; s++) {
                    int cp = pos[s] + (pos[s] >= r ? 1 : 0);
                    cp = cp < cap - 1 ? cp : 0;
                    so[s] = Lg + cp * LDC;
                }
                for (int i = rlo; i < nhi - 1; i++) {
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const R v = lds[so[s] + i + 1];
                        if (doRem && i >= r && i < nao - 1 && pos[s] < i) lds[bo[s] + i] = v;
                    }
                }
#pragma unroll
                for (int s = 0; s < S; s++)
                    if (doRem && pos[s] < nao - 1) lds[bo[s] + nao - 1] = (R)0;       // the row that left: back to zeros
            }
            // the per-position registers move down by one from position r on
            {
                auto shift = [&](auto *a, bool clear_outside) {
                    using V = std::remove_pointer_t<decltype(a)>;
                    V nx[S];
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const V sh = rw_shl1(a[s]);
                        V nb = (V)0;
                        if (s + 1 < S) nb = rw_bc<0>(a[s + 1 < S ? s + 1 : s]);
                        nx[s] = li == 15 ? nb : sh;
                    }
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        if (doRem) {
                            if (pos[s] >= r && pos[s] < nao - 1) a[s] = nx[s];
                            else if (pos[s] == nao - 1 || clear_outside) a[s] = (V)0;
                        }
                    }
                };
                shift(ws, false); shift(lam, false); shift(rhs, false); shift(D, false); shift(Dinv, false);
                shift(w, true);
            }
            if (doRem) { na = nao - 1; sing = -1; ydirty = 1; }
            // rank-one update of the trailing block, column by column
            {
                bool stop = false;
                const int nhi2 = rw_max4(doRem ? na : 0);
                rw_static_for<0, CAPP - 1>([&](auto T) {
                    constexpr int t = decltype(T)::value;
                    if (t >= rlo && t < nhi2) {
                        const bool actv = doRem && t >= r && t < na && !stop;
                        if (rw_any(actv)) {
                            const R pt = rw_bc<t>(w[t >> 4]);
                            const R dold = rw_bc<t>(D[t >> 4]);
                            const R dbar = wv_fma(al * pt, pt, dold);
                            const bool sng = actv && dbar < zero_tol;
                            const bool upd = actv && !sng;
                            const R rinv = (R)1 / dbar;
                            const R beta = (pt * al) * rinv;
                            const R aln = (dold * al) * rinv;
                            if (pos[t >> 4] == t) {
                                if (sng) { D[t >> 4] = (R)0; Dinv[t >> 4] = (R)0; }
                                else if (upd) { D[t >> 4] = dbar; Dinv[t >> 4] = rinv; }
                            }
                            if (sng) { sing = t; stop = true; }
                            al = upd ? aln : al;
#pragma unroll
                            for (int s = 0; s < S; s++) {
                                if (16 * s + 15 > t) {
                                    const R lq = lds[fo[s] + t * LDC];
                                    const bool m2 = upd && pos[s] > t && pos[s] < na;
                                    const R wn = wv_fma(-pt, lq, w[s]);
                                    w[s] = m2 ? wn : w[s];
                                    if (m2) lds[fo[s] + t * LDC] = wv_fma(beta, wn, lq);
                                }
                            }
                        }
                    }
                });
            }
            if (doRem && li == (jrem & 15)) {
                actb &= ~(1u << (jrem >> 4));
                lowb &= ~(1u << (jrem >> 4));
            }
            if (doRem) nsoft -= softrem;
        }
        if (run && !fin) iter++;

        // =============================================================== rows whose problem has ended: outputs, clean-up
        if (rw_any(fin)) {
            const bool listed = fin && flag == EXIT_WSCAP && ovf_list != nullptr;
            const int tho = (fin ? pid : 0) * nth;
            // x = R^-1 u + x0 + Xth theta   (mpc_update_qp.c:14-22); lane k of slot s writes output k + 16 s
#pragma unroll
            for (int s = 0; s < NS; s++) {
                if (16 * s < nout) {
                    const int ko = li + 16 * s;
                    const int lo = ko < nout ? ko : nout - 1;
                    R xs = (R)0;
                    rw_static_for<0, 16 * NS / 4>([&](auto B) {
                        constexpr int c0 = decltype(B)::value * 4;
                        if (c0 < n) {
                            RW_BLOCK();
                            R rv[4];
#pragma unroll
                            for (int q = 0; q < 4; q++) rv[q] = ldc(P.oRout, lo * n + (c0 + q < n ? c0 + q : n - 1));   // (beyond n: u_c = 0)
                            rw_static_for<0, 4>([&](auto Q) {
                                constexpr int c = c0 + decltype(Q)::value;
                                xs = wv_fma(rv[decltype(Q)::value], rw_bc<c>(u[c >> 4]), xs);
                            });
                        }
                    });
                    R sh = ldc(P.ox0, lo);
                    for (int t = 0; t < nth; t++) sh = wv_fma(ldc(P.oXth + t, lo * nth), ldth(tho + t), sh);
                    const R xo = xs + sh;
                    if (fin && ko < nout && X != nullptr) X[(long long)pid * nout + ko] = xo;
                }
            }
            if (active != nullptr) {
                unsigned long long acc = 0ull;
#pragma unroll
                for (int r = 0; r < MS; r++) {
                    const bool a = fin && ((actb >> r) & 1u), lo = (lowb >> r) & 1u;
                    const unsigned long long bu = __ballot(a && !lo), bl = __ballot(a && lo);
                    const unsigned long long mu = (bu >> (16 * g)) & 0xffffull, ml = (bl >> (16 * g)) & 0xffffull;
                    const int pu = 16 * r, pl = m + 16 * r;
                    if ((pu >> 6) == li) acc |= mu << (pu & 63);
                    if ((pu >> 6) + 1 == li && (pu & 63) > 48) acc |= mu >> (64 - (pu & 63));
                    if ((pl >> 6) == li) acc |= ml << (pl & 63);
                    if ((pl >> 6) + 1 == li && (pl & 63) > 48) acc |= ml >> (64 - (pl & 63));
                }
                if (fin && !listed && li < P.words) active[(long long)pid * P.words + li] = acc;
            }
            if (fin && li == 0) {
                if (exitflag != nullptr) exitflag[pid] = flag;
                if (iters != nullptr) iters[pid] = iter;
                if (listed) ovf_list[atomicAdd(ovf_count, 1)] = (int32_t)pid;
                if (stat != nullptr && !listed)
                    atomicAdd(&stat[(((int)myrow) & 63) * 16 + (napk <= 24 ? 0 : (napk <= 32 ? 1 : (napk <= 48 ? 2 : 3)))], 1ull);
            }
            // the factor's rows back to zeros: the next problem of this row starts on a factor of zeros
            const int nclr = rw_max4(fin ? na : 0);
            for (int i = 1; i < nclr; i++) {
#pragma unroll
                for (int s = 0; s < S; s++)
                    if (fin && i < na && pos[s] < i) lds[bo[s] + i] = (R)0;
            }
            if (fin) live = 0;
        }
    }
}

}  // namespace lmpc
