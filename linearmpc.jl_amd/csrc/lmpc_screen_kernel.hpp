// Streaming first pass of the batched solve: finishes every problem whose unconstrained optimum
// is already feasible and queues the rest for the iterating lane kernel.
//
// A cold-started dual active-set solve begins with an empty working set: u = 0, and the first
// thing it does is look for a violated row of  dl + b <= 0 <= du + b  (b = Dth theta).  If there is
// none the problem is optimal after that one iteration and x = x0 + Xth theta.  In MPC batches
// this is the common case (a controller near its set-point), so it gets its own kernel: a few
// dozen VGPRs, high occupancy, every constant read through the scalar cache -- it streams theta
// once and leaves the register-heavy iterating kernel a dense list of the problems that actually
// need iterations.
//
// The arithmetic is the same as the first iteration of lane_kernel / the CPU oracle (same fma
// chains, same comparisons), so which kernel finishes a problem does not change a single bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lmpc_lane_kernel.hpp"

namespace lmpc {

// Each lane walks kScreenTPB problems (strided by the workgroup size; the record of the next one is
// loaded into registers before the current one is screened).  Records are read straight from HBM,
// one 8-byte load per parameter: lane i reads record i, so the 64 lanes of a load touch 64*8*nth
// contiguous bytes over the nth loads -- every fetched line is used completely.  Measured at 10^6
// pendulum points: kScreenTPB = 1 / 2 / 4 / 8 -> 18.1 / 19.3 / 20.5 / 24.4 us, and a variant that
// staged 16-byte coalesced tiles through LDS (two barriers per tile) 22.9 us: at this batch size
// the pass is two rounds of resident wavefronts, more wavefronts in flight beat fewer, longer ones.
#ifndef LMPC_SCREEN_TPB
#define LMPC_SCREEN_TPB 1
#endif
constexpr int kScreenTPB = LMPC_SCREEN_TPB;

// Cache policy of the streaming accesses, decided by same-box A/B (tools/hot_ab.py) on 10^6 pendulum points:
// record LOADS plain -- nontemporal loads cost the pass 6 us of 16 (22.3 vs 16.4 us; round 1 believed it had
// measured them "neutral": its run-time switch between the two forms had been folded into a plain load);
// output STORES plain as well (nontemporal stores: same pass time, but the iterating kernel behind it, which
// overwrites the queued problems' outputs, ran 0.3-0.7 us longer on lines that had been pushed out).
#ifndef LMPC_SCREEN_NT_LOAD
#define LMPC_SCREEN_NT_LOAD 0
#endif
#ifndef LMPC_SCREEN_NT_STORE
#define LMPC_SCREEN_NT_STORE 0
#endif
template <typename T> __device__ __forceinline__ T screen_ld(const T *p) {
#if LMPC_SCREEN_NT_LOAD
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <typename T> __device__ __forceinline__ void screen_st(T v, T *p) {
#if LMPC_SCREEN_NT_STORE
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// NTHMAX: column stride of the padded rows (8, 16 or 32).  NT: columns the unrolled chains run over --
// the exact nth for nth <= 16 (one instantiation per value: with three batches in flight the pass is
// bound by vector issue, and the padded column of the 7-parameter pendulum cost 6 %), else NTHMAX.
// MODE: 0 the plain solve, 1 closed loop (SimFuse), 2 generated controller (GatherArgs), 3 the plain solve
// with several outputs per problem (their stores go through a wave-private LDS transpose).
template <int NTHMAX, int NT, int MODE>
__global__ __launch_bounds__(256) void screen_kernel(
    const PackLayout P, const double *__restrict__ C, const double *__restrict__ theta,
    double *__restrict__ X, int32_t *__restrict__ exitflag, int32_t *__restrict__ iters,
    uint64_t *__restrict__ active, const uint64_t *__restrict__ warm, int32_t *__restrict__ list,
    int32_t *__restrict__ count, long long seg_cap, int nshards, long long nprob) {
    const int m = P.m, nth = P.nth, B = blockDim.x, tid = threadIdx.x;
    const long long first = (long long)blockIdx.x * kScreenTPB * B + tid;
    const double ntol = -P.primal_tol;
    const int lane = tid & 63;
    const int shard = blockIdx.x & (nshards - 1);      // nshards is a power of two (checked by the host)
    constexpr bool SIM = MODE == 1, GATHER = MODE == 2, WIDE = MODE == 3;

    double nx[NT];                                     // record of the problem after the current one
    // No guard on any load: a problem index past the end is clamped to the last problem (its results are
    // dropped by `valid`), a parameter index past the record (NT > nth, padded instantiations only) to the
    // record's last entry -- it meets a zero coefficient in the padded rows, so the term adds +0 exactly.
    // (Guarded, every one of the loads sat in its own exec-masked block: six scalar instructions and a
    // branch per parameter.)
    auto load_record = [&](long long pid, double *dst) {
        const long long pc = pid < nprob ? pid : nprob - 1;
        if constexpr (GATHER) {
            // theta = [state; reference; disturbance; control[0:nup]; parameter] straight from the caller's
            // arrays (codegen/mpc_update_parameter.c without the detour through a theta buffer); a NULL
            // array stands for zeros
            const GatherArgs &Ga = P.gat;
            const int o1 = Ga.nx, o2 = o1 + Ga.nr, o3 = o2 + Ga.nd, o4 = o3 + Ga.nup;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                double v = 0.0;
                if (t < o1) v = Ga.state[pc * Ga.nx + t];
                else if (t < o2) { if (Ga.reference) v = Ga.reference[pc * Ga.nr + (t - o1)]; }
                else if (t < o3) { if (Ga.disturbance) v = Ga.disturbance[pc * Ga.nd + (t - o2)]; }
                else if (t < o4) { if (Ga.control) v = Ga.control[pc * Ga.ncontrol + (t - o3)]; }
                else if (t < nth) { if (Ga.parameter) v = Ga.parameter[pc * Ga.np + (t - o4)]; }
                dst[t] = v;
            }
        } else {
        const double *src = theta + pc * nth;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int tc = (NT <= 16 || t < nth) ? t : nth - 1;      // NT <= 16: NT == nth exactly
            dst[t] = screen_ld(src + tc);
        }
        }
    };
    load_record(first, nx);

    // list write of the previous problem, held back so that its atomic's round trip runs underneath
    // the next problem's screening
    unsigned long long pmask = 0ull;
    int pbase = 0;
    long long ppid = 0;
    bool phard = false;

    for (int it = 0; it < kScreenTPB; it++) {
        const long long pid = first + (long long)it * B;
        if (pid - tid >= nprob) break;                 // uniform over the workgroup
        const bool valid = pid < nprob;
        double th[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) th[t] = nx[t];
        if (it + 1 < kScreenTPB) load_record(pid + B, nx);

        bool hard = false;
        // rows of Dth zero-padded to NTHMAX columns (theta is zero-padded in registers): a fixed,
        // guard-free fma chain per row and one wide scalar load; the padded terms add +0 exactly.
        // Four rows per trip (the row count is padded to a multiple of four with rows that can never
        // be violated), so four independent fma chains are in flight instead of one.
        const double *dj = C + P.oDthP;
        const double *bj = C + P.oBnd;
        unsigned long long imm = P.imm_mask;
        const int mp = (m + 3) & ~3;
        for (int j = 0; j < mp; j += 4, dj += 4 * NTHMAX, bj += 8, imm >>= 4) {
            double b[4];
#pragma unroll
            for (int q = 0; q < 4; q++) b[q] = 0.0;
            if (j + 4 <= m) {
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int q = 0; q < 4; q++) b[q] = __builtin_fma(dj[q * NTHMAX + t], th[t], b[q]);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const double vu = (bj[2 * q] + b[q]) - 0.0;
                    const double vl = -((bj[2 * q + 1] + b[q]) - 0.0);
                    if (!((imm >> q) & 1ull)) hard = hard || (vu < ntol) || (vl < ntol);
                }
            } else {
                // last, partial group of rows: only the real ones (the padded rows can never be violated,
                // skipping them changes nothing but the instruction count)
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    if (j + q < m) {
#pragma unroll
                        for (int t = 0; t < NT; t++) b[q] = __builtin_fma(dj[q * NTHMAX + t], th[t], b[q]);
                        const double vu = (bj[2 * q] + b[q]) - 0.0;
                        const double vl = -((bj[2 * q + 1] + b[q]) - 0.0);
                        if (!((imm >> q) & 1ull)) hard = hard || (vu < ntol) || (vl < ntol);
                    }
                }
            }
        }
        // a warm start with an EMPTY initial working set is a cold start (the usual case once a
        // closed loop has settled); any other mask goes to the iterating kernel, which starts from it
        if (warm != nullptr && valid) {
            unsigned long long wany = 0ull;
            for (int w = 0; w < P.words; w++) wany |= warm[pid * P.words + w];
            hard = hard || wany != 0ull;
        }
        hard = hard && valid;

        // Append the problems that need iterations to the work list: one atomic per wavefront, issued
        // before the outputs are formed so that its round trip overlaps them.  The list is cut into
        // `nshards` segments with their own counters (128 B apart): a single counter word saturates
        // near 90 atomics/us, which 15k wavefronts would turn into the bottleneck of the whole pass.
        const unsigned long long mask = __ballot(hard);
        int basei = 0;
        if (mask != 0ull && lane == 0) basei = atomicAdd(&count[shard * kCountStride], __popcll(mask));
        if (pmask != 0ull) {
            const int pb = __shfl(pbase, 0);
            if (phard)
                list[(long long)shard * seg_cap + pb + __popcll(pmask & ((1ull << lane) - 1ull))] = (int32_t)ppid;
        }
        pmask = mask; pbase = basei; ppid = pid; phard = hard;

        // Outputs are written for EVERY valid problem, queued ones included (the iterating kernel runs
        // behind this one on the same stream and overwrites theirs): the stores of a wavefront then
        // cover whole lines instead of lines with holes; they bypass the caches (written once, read by
        // nobody on this GPU soon)
        // (generated-controller mode: X is the caller's control array, whose previous-control entries the
        // iterating kernel's problems have not consumed yet -- there only finished problems are written, and
        // the queued ones get their assembled record handed over instead)
        // several outputs per problem (compute_control_trajectory): the wavefront's 64 * nout outputs are one
        // contiguous run of X -- they go through a wave-private LDS transpose and leave as nout coalesced
        // stores (8-byte stores at a stride of 8 * nout bytes cost this pass 12 us at nout = 5).  Plain mode
        // only: there every valid problem of the wavefront writes its outputs.
        extern __shared__ double sxo[];                // 256 * nout doubles, given by the launch (MODE 3)
        constexpr bool wide_out = WIDE;
        double rec[NT];                                // closed loop: the next record of a finished problem
        bool recok = false;
        const bool fill = GATHER ? (valid && !hard) : valid;
        if constexpr (GATHER) {
            if (hard) {
                double *to = P.gat.theta_out + pid * nth;
#pragma unroll
                for (int t = 0; t < NT; t++)
                    if (NT <= 16 || t < nth) to[t] = th[t];
            }
        }
        if (fill) {
            const double *xk = C + P.oXthP;
            double uo[kMaxSimU];
#pragma unroll
            for (int l = 0; l < kMaxSimU; l++) uo[l] = 0.0;
            for (int k = 0; k < P.nout; k++, xk += NTHMAX) {
                double sh = C[P.ox0 + k];
#pragma unroll
                for (int t = 0; t < NT; t++) sh = __builtin_fma(xk[t], th[t], sh);
#pragma unroll
                for (int l = 0; l < kMaxSimU; l++) if (SIM && l == k) uo[l] = 0.0 + sh;
                if (SIM && X == nullptr) {                          // closed loop without an input trajectory
                } else if (wide_out) sxo[(tid & ~63) * P.nout + lane * P.nout + k] = 0.0 + sh;   // see below
                else screen_st(0.0 + sh, X + pid * P.nout + k);
            }
            // closed loop: a problem finished here also advances its scenario (queued ones: lane kernel)
            if (SIM && !hard) {
                // The record sits in registers here, and the next one is assembled in registers too --
                // every index a compile-time constant -- and stored in one run of wide stores (element by
                // element from a run-time loop the 8-byte stores at a 56-byte stride made this pass 3.6x
                // slower than the plain one).
                const SimFuse &S = P.sim;
                const int nx = S.nx, nu = S.nu, nr = S.nr, nup = S.nup;
                const double *F = C + P.oFG, *G = F + nx * nx;         // in the constant pack: scalar loads
#pragma unroll
                for (int a = 0; a < NT; a++) {
                    rec[a] = th[a];                                    // the reference block is carried over
                    if (a < nx) {
                        double acc = 0.0;
#pragma unroll
                        for (int c = 0; c < NT; c++)
                            if (c < nx) acc = __builtin_fma(F[a * nx + c], th[c], acc);
#pragma unroll
                        for (int l = 0; l < kMaxSimU; l++)
                            if (l < nu) acc = __builtin_fma(G[a * nu + l], uo[l], acc);
                        rec[a] = acc;
                    }
#pragma unroll
                    for (int l = 0; l < kMaxSimU; l++)
                        if (l < nup && a == nx + nr + l) rec[a] = uo[l];
                }
                recok = true;
                if constexpr (NT > 16) {                               // padded instantiation: record stride nth != NT
                    double *to = S.theta_out + pid * nth;
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        if (t < nth) to[t] = rec[t];
                }
                if (S.xtraj) {
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        if (t < nx) S.xtraj[pid * nx + t] = rec[t];
                }
                if (S.flag_min) S.flag_min[pid] = S.first ? (int)EXIT_OPTIMAL
                                                           : (EXIT_OPTIMAL < S.flag_min[pid] ? (int)EXIT_OPTIMAL : S.flag_min[pid]);
            }
            if (SIM && exitflag == nullptr) {                       // closed loop: flag_min carries the flags
            } else screen_st((int32_t)EXIT_OPTIMAL, exitflag + pid);
            // (iteration count and active set only for the problems finished here: `active` may be the
            // very buffer the iterating kernel still has to read its warm-start masks from)
            if (iters && !hard) iters[pid] = 1;
            if (active && !hard)
                for (int w = 0; w < P.words; w++) active[pid * P.words + w] = 0ull;
        }
        if (WIDE && wide_out) {
            __builtin_amdgcn_wave_barrier();
            const unsigned long long vmask = __ballot(valid);
            const double *sw = sxo + (tid & ~63) * P.nout;
            double *xb = X + (pid - lane) * P.nout;
            for (int j = 0; j < P.nout; j++) {
                const int idx = j * 64 + lane;
                if ((vmask >> (idx / P.nout)) & 1ull) xb[idx] = sw[idx];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (SIM && NT <= 16) {
            // The wavefront's 64 records are one contiguous run of 64*NT doubles in theta_out: they go through
            // a wave-private LDS transpose and leave as NT fully coalesced stores (lane l writes doubles
            // j*64 + l), each double under the "finished here" bit of the record it belongs to.  (Record by
            // record every store instruction touched 64 different 56-byte records.)
            __shared__ double srec[256 * NT];
            double *sw = srec + (tid & ~63) * NT;
            const unsigned long long okmask = __ballot(recok);
#pragma unroll
            for (int t = 0; t < NT; t++) sw[lane * NT + t] = rec[t];
            __builtin_amdgcn_wave_barrier();
            double *tob = P.sim.theta_out + (pid - lane) * NT;      // NT == nth for these instantiations
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int idx = j * 64 + lane;
                if ((okmask >> (idx / NT)) & 1ull) tob[idx] = sw[idx];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (pmask != 0ull) {
        const int pb = __shfl(pbase, 0);
        if (phard)
            list[(long long)shard * seg_cap + pb + __popcll(pmask & ((1ull << lane) - 1ull))] = (int32_t)ppid;
    }
}

}  // namespace lmpc
