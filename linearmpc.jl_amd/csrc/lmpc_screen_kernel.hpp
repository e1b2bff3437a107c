// Streaming first pass of the batched solve: finishes every problem whose unconstrained optimum
// is already feasible and queues the rest for the iterating lane kernel.
//
// A cold-started dual active-set solve begins with an empty working set: u = 0, and the first
// thing it does is look for a violated row of  dl + b <= 0 <= du + b  (b = Dth theta).  If there is
// none the problem is optimal after that one iteration and x = x0 + Xth theta.  In MPC batches
// this is the common case (a controller near its set-point), so it gets its own kernel: a few
// dozen VGPRs, full occupancy, theta tiles staged through LDS with 16-byte coalesced loads, every
// constant read through the scalar cache -- it runs at HBM speed and leaves the register-heavy
// iterating kernel a dense list of the problems that actually need iterations.
//
// The arithmetic is the same as the first iteration of lane_kernel / the CPU oracle (same fma
// chains, same comparisons), so which kernel finishes a problem does not change a single bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lmpc_lane_kernel.hpp"

namespace lmpc {

// TPB consecutive tiles per workgroup: the loads of tile t+1 are issued (into registers) before tile
// t is screened, so a workgroup's HBM round trips overlap its own arithmetic instead of every
// resident workgroup alternating between "all loading" and "all computing" in lock-step.
constexpr int kScreenTPB = 4;

template <int NTHMAX>
__global__ __launch_bounds__(256) void screen_kernel(
    const PackLayout P, const double *__restrict__ C, const double *__restrict__ theta,
    double *__restrict__ X, int32_t *__restrict__ exitflag, int32_t *__restrict__ iters,
    uint64_t *__restrict__ active, int32_t *__restrict__ list, int32_t *__restrict__ count,
    long long seg_cap, int nshards, long long nprob, int vec16, int ablate) {
    constexpr int R2 = NTHMAX / 2;                     // 16-byte pieces of a tile per thread
    extern __shared__ __align__(16) double tile[];     // B records of nth doubles, as in HBM
    const int m = P.m, nth = P.nth, B = blockDim.x, tid = threadIdx.x;
    const long long ntiles = (nprob + B - 1) / B;
    const long long tile0 = (long long)blockIdx.x * kScreenTPB;
    const double ntol = -P.primal_tol;
    const int lane = tid & 63;
    const int shard = blockIdx.x % nshards;

    double2 pf[R2];
#define LMPC_PREFETCH_TILE(TIX)                                                                  \
    {                                                                                             \
        const long long bp_ = (TIX) * B;                                                          \
        const int nv_ = (nprob - bp_) < (long long)B ? (int)(nprob - bp_) : B;                    \
        const int n2_ = (nv_ * nth) >> 1;                                                         \
        const double2 *s2_ = reinterpret_cast<const double2 *>(theta + bp_ * nth);                \
        _Pragma("unroll") for (int r = 0; r < R2; r++) {                                          \
            const int i_ = tid + r * B;                                                           \
            pf[r] = (i_ < n2_) ? s2_[i_] : make_double2(0.0, 0.0);                                \
        }                                                                                         \
    }
    if (vec16 && tile0 < ntiles && !(ablate & 8)) LMPC_PREFETCH_TILE(tile0)
    else {
#pragma unroll
        for (int r = 0; r < R2; r++) pf[r] = make_double2(0.0, 0.0);
    }

    // list write of the previous tile, held back so that its atomic's round trip (the longest single
    // latency in a wave's life) runs underneath the next tile's screening
    unsigned long long pmask = 0ull;
    int pbase = 0;
    long long ppid = 0;
    bool phard = false;

    for (int it = 0; it < kScreenTPB; it++) {
        const long long tix = tile0 + it;
        if (tix >= ntiles) break;                      // uniform over the workgroup
        const long long bp = tix * B;
        const int nvalid = (nprob - bp) < (long long)B ? (int)(nprob - bp) : B;
        const int elems = nvalid * nth;
        const double *src = theta + bp * nth;
        if (vec16) {
            double2 *t2 = reinterpret_cast<double2 *>(tile);
#pragma unroll
            for (int r = 0; r < R2; r++) {
                const int i = tid + r * B;
                if (i < (elems >> 1)) t2[i] = pf[r];
            }
            if ((elems & 1) && tid == 0) tile[elems - 1] = src[elems - 1];
        } else {
            for (int i = tid; i < elems; i += B) tile[i] = src[i];
        }
        __syncthreads();
        if (vec16 && it + 1 < kScreenTPB && tix + 1 < ntiles && !(ablate & 8)) LMPC_PREFETCH_TILE(tix + 1)

        const bool valid = tid < nvalid;
        const long long pid = bp + tid;
        double th[NTHMAX];
#pragma unroll
        for (int t = 0; t < NTHMAX; t++) th[t] = (valid && t < nth) ? tile[tid * nth + t] : 0.0;

        bool hard = false;
        // rows of Dth zero-padded to NTHMAX columns (theta is zero-padded in registers): a fixed,
        // guard-free fma chain per row and one wide scalar load; the padded terms add +0 exactly
        // Four rows per trip (the row count is padded to a multiple of four with rows that can never
        // be violated), so four independent fma chains are in flight instead of one.
        const double *dj = C + P.oDthP;
        const double *bj = C + P.oBnd;
        unsigned long long imm = P.imm_mask;
        const int mp = (ablate & 2) ? 0 : ((m + 3) & ~3);
        for (int j = 0; j < mp; j += 4, dj += 4 * NTHMAX, bj += 8, imm >>= 4) {
            double b[4];
#pragma unroll
            for (int q = 0; q < 4; q++) b[q] = 0.0;
#pragma unroll
            for (int t = 0; t < NTHMAX; t++)
#pragma unroll
                for (int q = 0; q < 4; q++) b[q] = __builtin_fma(dj[q * NTHMAX + t], th[t], b[q]);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const double vu = (bj[2 * q] + b[q]) - 0.0;
                const double vl = -((bj[2 * q + 1] + b[q]) - 0.0);
                if (!((imm >> q) & 1ull)) hard = hard || (vu < ntol) || (vl < ntol);
            }
        }
        hard = hard && valid;

        // Append the problems that need iterations to the work list: one atomic per wavefront, issued
        // before the outputs are formed so that its round trip overlaps them.  The list is cut into
        // `nshards` segments with their own counters (128 B apart): a single counter word saturates
        // near 90 atomics/us, which 15k wavefronts would turn into the bottleneck of the whole pass.
        const unsigned long long mask = __ballot(hard);
        int basei = 0;
        if (mask != 0ull && lane == 0 && !(ablate & 1)) basei = atomicAdd(&count[shard * kCountStride], __popcll(mask));
        if (pmask != 0ull) {
            const int pb = __shfl(pbase, 0);
            if (phard && !(ablate & 1))
                list[(long long)shard * seg_cap + pb + __popcll(pmask & ((1ull << lane) - 1ull))] = (int32_t)ppid;
        }
        pmask = mask; pbase = basei; ppid = pid; phard = hard;

        if (valid && !hard && !(ablate & 4)) {
            const double *xk = C + P.oXthP;
            for (int k = 0; k < P.nout; k++, xk += NTHMAX) {
                double sh = C[P.ox0 + k];
#pragma unroll
                for (int t = 0; t < NTHMAX; t++) sh = __builtin_fma(xk[t], th[t], sh);
                X[pid * P.nout + k] = 0.0 + sh;
            }
            exitflag[pid] = EXIT_OPTIMAL;
            if (iters) iters[pid] = 1;
            if (active)
                for (int w = 0; w < P.words; w++) active[pid * P.words + w] = 0ull;
        }

        __syncthreads();                               // tile is rewritten by the next round
    }
    if (pmask != 0ull) {
        const int pb = __shfl(pbase, 0);
        if (phard && !(ablate & 1))
            list[(long long)shard * seg_cap + pb + __popcll(pmask & ((1ull << lane) - 1ull))] = (int32_t)ppid;
    }
}

#undef LMPC_PREFETCH_TILE

}  // namespace lmpc
