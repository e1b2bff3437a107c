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

template <int NTHMAX>
__global__ __launch_bounds__(256) void screen_kernel(
    const PackLayout P, const double *__restrict__ C, const double *__restrict__ theta,
    double *__restrict__ X, int32_t *__restrict__ exitflag, int32_t *__restrict__ iters,
    uint64_t *__restrict__ active, int32_t *__restrict__ list, int32_t *__restrict__ count,
    long long seg_cap, int nshards, long long nprob, int vec16) {
    extern __shared__ __align__(16) double tile[];     // B records of nth doubles, as in HBM
    const int m = P.m, nth = P.nth, B = blockDim.x, tid = threadIdx.x;
    const long long bp = (long long)blockIdx.x * B;
    const int nvalid = (nprob - bp) < (long long)B ? (int)(nprob - bp) : B;
    const int elems = nvalid * nth;
    const double *src = theta + bp * nth;
    if (vec16) {
        const double2 *s2 = reinterpret_cast<const double2 *>(src);
        double2 *t2 = reinterpret_cast<double2 *>(tile);
        for (int i = tid; i < (elems >> 1); i += B) t2[i] = s2[i];
        if ((elems & 1) && tid == 0) tile[elems - 1] = src[elems - 1];
    } else {
        for (int i = tid; i < elems; i += B) tile[i] = src[i];
    }
    __syncthreads();

    const bool valid = tid < nvalid;
    const long long pid = bp + tid;
    double th[NTHMAX];
#pragma unroll
    for (int t = 0; t < NTHMAX; t++) th[t] = (valid && t < nth) ? tile[tid * nth + t] : 0.0;

    const double ntol = -P.primal_tol;
    bool hard = false;
    for (int j = 0; j < m; j++) {
        if ((P.imm_mask >> j) & 1ull) continue;
        double b = 0.0;
        const double *dj = C + P.oDth + j * nth;
#pragma unroll
        for (int t = 0; t < NTHMAX; t++)
            if (t < nth) b = __builtin_fma(dj[t], th[t], b);
        const double vu = (C[P.odu + j] + b) - 0.0;
        const double vl = -((C[P.odl + j] + b) - 0.0);
        hard = hard || (vu < ntol) || (vl < ntol);
    }
    hard = hard && valid;

    // Append the problems that need iterations to the work list: one atomic per wavefront, issued
    // before the outputs are formed so that its round trip overlaps them.  The list is cut into
    // `nshards` segments with their own counters (128 B apart): a single counter word saturates
    // near 90 atomics/us, which 15k wavefronts would turn into the bottleneck of the whole pass.
    const unsigned long long mask = __ballot(hard);
    const int lane = tid & 63;
    const int shard = blockIdx.x % nshards;
    int basei = 0;
    if (mask != 0ull && lane == 0) basei = atomicAdd(&count[shard * kCountStride], __popcll(mask));

    if (valid && !hard) {
        for (int k = 0; k < P.nout; k++) {
            double sh = C[P.ox0 + k];
            const double *xk = C + P.oXth + k * nth;
#pragma unroll
            for (int t = 0; t < NTHMAX; t++)
                if (t < nth) sh = __builtin_fma(xk[t], th[t], sh);
            X[pid * P.nout + k] = 0.0 + sh;
        }
        exitflag[pid] = EXIT_OPTIMAL;
        if (iters) iters[pid] = 1;
        if (active)
            for (int w = 0; w < P.words; w++) active[pid * P.words + w] = 0ull;
    }

    if (mask != 0ull) {
        basei = __shfl(basei, 0);
        if (hard)
            list[(long long)shard * seg_cap + basei + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)pid;
    }
}

}  // namespace lmpc
