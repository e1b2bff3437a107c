// Scenario-asynchronous closed loop, streaming half: every lane owns one scenario and advances it, IN
// REGISTERS, through as many time steps as its problem stays "finished by screening" (no violated row at
// u = 0, empty warm mask) -- u = x0 + Xth theta, x+ = F x + G u, theta <- [x+; r; u] -- and stops at the first
// step that needs iterations (or at T).  It writes the record and the scenario's step counter back once and
// queues the scenario for the iterating kernel, which solves that one step, advances the scenario
// (sim_advance, in place) and hands it back for the next round.
//
// Scenarios are independent of each other, so the lock-step of the reference's Simulation loop
// (src/simulation.jl:93-113) is not needed for the result: per scenario and step the arithmetic is the one
// of screen_kernel / lane_solve / plant_kernel, hence the trajectories equal the lock-step ones bit for
// bit.  What changes is the traffic: a settled loop costs no memory round trip per step at all.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "lmpc_lane_kernel.hpp"

namespace lmpc {

// SMALL: the instantiation for single-input problems with nx <= 4 and m <= 8 rows, all constants in registers
// (244 VGPRs, two wavefronts per SIMD, no LDS read in the step); the general one keeps 4 wavefronts per SIMD.
template <int NTHMAX, int NT, bool SMALL>
__global__ __launch_bounds__(256) void sim_run_kernel(
    const PackLayout P, const double *__restrict__ C, double *theta, int32_t *kstep, const int T,
    uint64_t *active, const int use_warm, double *U_traj, double *X_traj, int32_t *flag_min,
    int32_t *__restrict__ list, int32_t *__restrict__ count, const long long seg_cap, const int nshards,
    const long long nprob, const int32_t *__restrict__ list_in, const int32_t *count_in,
    const int step_cap, int32_t *__restrict__ park_list, int32_t *park_count, const double *__restrict__ x_in,
    const double *__restrict__ r_in, const double *__restrict__ up_in) {
    static_assert(NT <= 16, "exact parameter count");
    const int m = P.m, tid = threadIdx.x, lane = tid & 63;
    const int shard = blockIdx.x & (nshards - 1);
    // round 0 walks all scenarios; later rounds only those the iterating kernel just advanced -- the list of
    // the round before, shard by shard, so that wavefronts stay full however few scenarios are still running
    long long pid = (long long)blockIdx.x * blockDim.x + tid;
    bool valid = pid < nprob;
    if (list_in != nullptr) {
        const long long idx = (long long)(blockIdx.x / nshards) * blockDim.x + tid;
        valid = idx < (long long)count_in[shard * kCountStride];
        pid = valid ? (long long)list_in[(long long)shard * seg_cap + idx] : 0;
    }
    const long long pc = valid ? pid : nprob - 1;
    const double ntol = -P.primal_tol;
    const int nx = P.sim.nx, nu = P.sim.nu, nr = P.sim.nr, nup = P.sim.nup;
    // The constants of a step are used again at every one of up to T steps: they are staged in LDS once per
    // workgroup and read from there by uniform (broadcast) addresses.  They are staged PADDED, so that the
    // step below is branch-free and its chains are independent (a wavefront that runs alone through ~100
    // steps is bound by the latency of one step, not by its instruction count):
    //   rows of Dth to a multiple of 8; padding rows and rows that are ignored at start (imm_mask) get the
    //     bounds (+1e300, -1e300), which no finite b violates -- only the truth value `hard` leaves the test;
    //   Xth / x0 to kMaxSimU rows of zeros (u_l = 0 + 0 = +0 for l >= nu, used nowhere);
    //   F to NXP x NXP, G to NXP x kMaxSimU with zeros: fma(0, finite, acc) == acc bit for bit (acc starts
    //     at +0 and a sum of finite terms is never -0), so the padded chains equal plant_kernel's.
    extern __shared__ double sc[];
    const int mp = (m + 7) & ~7, nxp = nx <= 4 ? 4 : 8;
    double *sD = sc, *sBnd = sD + mp * NTHMAX, *sXth = sBnd + 2 * mp, *sx0 = sXth + kMaxSimU * NTHMAX,
           *sF = sx0 + kMaxSimU, *sG = sF + 64;
    for (int i = tid; i < mp * NTHMAX; i += blockDim.x) sD[i] = i < m * NTHMAX ? C[P.oDthP + i] : 0.0;
    for (int i = tid; i < 2 * mp; i += blockDim.x) {
        const int j = i >> 1;
        // (imm_mask covers the first 64 rows; a wavefront-kernel handle's screening pack has the never-violated
        // bounds of its IMMUTABLE rows written in already, whatever their index)
        const bool live = j < m && !(j < 64 && ((P.imm_mask >> j) & 1ull));
        sBnd[i] = live ? C[P.oBnd + i] : ((i & 1) ? -1e300 : 1e300);
    }
    for (int i = tid; i < kMaxSimU * NTHMAX; i += blockDim.x) sXth[i] = i < nu * NTHMAX ? C[P.oXthP + i] : 0.0;
    for (int i = tid; i < kMaxSimU; i += blockDim.x) sx0[i] = i < nu ? C[P.ox0 + i] : 0.0;
    for (int i = tid; i < 64; i += blockDim.x) {
        const int a_ = i / nxp, c_ = i % nxp;
        sF[i] = (a_ < nx && c_ < nx) ? C[P.oFG + a_ * nx + c_] : 0.0;
    }
    for (int i = tid; i < 8 * kMaxSimU; i += blockDim.x) {
        const int a_ = i / kMaxSimU, l_ = i % kMaxSimU;
        sG[i] = (a_ < nx && l_ < nu) ? C[P.oFG + nx * nx + a_ * nu + l_] : 0.0;
    }
    __syncthreads();

    int k = valid ? kstep[pc] : T;
    const int k0 = k;
    double th[NT];
    const bool formed = x_in != nullptr;     // first pass: theta = [x; r; uprev] is formed here (form_theta_kernel's rule)
    if (formed) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
            double v = 0.0;
            if (t < nx) v = x_in[pc * nx + t];
            else if (t < nx + nr) v = r_in ? r_in[pc * nr + (t - nx)] : 0.0;
            else v = up_in ? up_in[pc * nup + (t - nx - nr)] : 0.0;
            th[t] = v;
        }
    } else {
        const double *src = theta + pc * NT;
#pragma unroll
        for (int t = 0; t < NT; t++) th[t] = src[t];
    }
    bool wany = false;
    if (use_warm && active != nullptr && k < T)
        for (int w = 0; w < P.words; w++) wany = wany || active[pc * P.words + w] != 0ull;

    const bool xvec = (nx & 1) == 0 && (reinterpret_cast<uintptr_t>(X_traj) & 15) == 0;
    bool hard = false;
    auto run = [&](auto nxp_c, auto nu_c, auto m_c) {
        constexpr int NXP = decltype(nxp_c)::value;
        constexpr int NUP = decltype(nu_c)::value;
        constexpr int MC = decltype(m_c)::value;          // > 0: exactly MC rows, held in registers (small problems)        // controls computed: 1 (a single input) or all kMaxSimU
        constexpr int NXS = NXP < NT ? NXP : NT;          // state rows that exist in the record
        // single input, nx <= 4: the plant and the output map (28 values at NT = 7) are kept as wave-uniform
        // values -- scalar registers, a free operand of the fma -- instead of being read from LDS every step
        constexpr bool UNI = (NXP == 4 && NUP == 1);
        constexpr int NUF = UNI ? NXS * NXS : 1, NUG = UNI ? NXS : 1, NUX = UNI ? NT : 1;
        double cF[NUF], cG[NUG], cX[NUX], cx0 = 0.0;
        if constexpr (UNI) {
            auto uni = [](double v) {
                const long long b = __builtin_bit_cast(long long, v);
                const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
                const unsigned hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
                return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
            };
#pragma unroll
            for (int a = 0; a < NXS; a++) {
#pragma unroll
                for (int c = 0; c < NXS; c++) cF[a * NXS + c] = uni(sF[a * NXP + c]);
                cG[a] = uni(sG[a * kMaxSimU]);
            }
#pragma unroll
            for (int t = 0; t < NT; t++) cX[t] = uni(sXth[t]);
            cx0 = uni(sx0[0]);
        }
        // small problems (m <= 8, single input, nx <= 4): the rows of Dth and the bounds live in vector registers
        // for the whole run -- the step then has no LDS read left, which is what bounded it (64 lanes x 16 B
        // per broadcast read, one LDS pipe for four SIMDs)
        constexpr int NCD = MC > 0 ? MC * NT : 1, NCB = MC > 0 ? MC : 1;
        double cD[NCD], cBu[NCB], cBl[NCB];
        if constexpr (MC > 0) {
#pragma unroll
            for (int j = 0; j < MC; j++) {
#pragma unroll
                for (int t = 0; t < NT; t++) cD[j * NT + t] = sD[j * NTHMAX + t];
                cBu[j] = sBnd[2 * j];
                cBl[j] = sBnd[2 * j + 1];
            }
        }
        while (k < T && k - k0 < step_cap) {
            // the screening test of this step: rows of  dl + b <= 0 <= du + b,  b = Dth theta  (screen_kernel's chains)
            bool h_ = (k == k0) && wany;
            auto rows = [&](auto ch_c, const int j0) {
                constexpr int CH = decltype(ch_c)::value;
                double b[CH];
#pragma unroll
                for (int jj = 0; jj < CH; jj++) {
                    const double *dj = sD + (j0 + jj) * NTHMAX;
                    b[jj] = 0.0;
#pragma unroll
                    for (int t = 0; t < NT; t++) b[jj] = __builtin_fma(dj[t], th[t], b[jj]);
                }
#pragma unroll
                for (int jj = 0; jj < CH; jj++) {
                    const double vu = (sBnd[2 * (j0 + jj)] + b[jj]) - 0.0;
                    const double vl = -((sBnd[2 * (j0 + jj) + 1] + b[jj]) - 0.0);
                    h_ = h_ | (vu < ntol) | (vl < ntol);
                }
            };
            if constexpr (MC > 0) {
                double b[NCB];
#pragma unroll
                for (int j = 0; j < MC; j++) {
                    b[j] = 0.0;
#pragma unroll
                    for (int t = 0; t < NT; t++) b[j] = __builtin_fma(cD[j * NT + t], th[t], b[j]);
                }
#pragma unroll
                for (int j = 0; j < MC; j++) {
                    const double vu = (cBu[j] + b[j]) - 0.0;
                    const double vl = -((cBl[j] + b[j]) - 0.0);
                    h_ = h_ | (vu < ntol) | (vl < ntol);
                }
            } else {
                // rows in groups of 8 independent chains, then 4, then one at a time (m = 5: 4 + 1, nothing padded)
                int j0 = 0;
                for (; j0 + 8 <= m; j0 += 8) rows(std::integral_constant<int, 8>{}, j0);
                if (j0 + 4 <= m) { rows(std::integral_constant<int, 4>{}, j0); j0 += 4; }
                for (; j0 < m; j0++) rows(std::integral_constant<int, 1>{}, j0);
            }
            hard = h_;
            if (hard) break;
            // finished by screening: u = x0 + Xth theta, then the plant step, all in registers
            double uo[kMaxSimU];
#pragma unroll
            for (int l = 0; l < kMaxSimU; l++) {
                uo[l] = 0.0;
                if (l < NUP) {
                    double sh = UNI ? cx0 : sx0[l];
                    const double *xk = sXth + l * NTHMAX;
#pragma unroll
                    for (int t = 0; t < NT; t++) sh = __builtin_fma(UNI ? cX[UNI ? t : 0] : xk[t], th[t], sh);
                    uo[l] = 0.0 + sh;
                }
            }
            double rec[NT];
#pragma unroll
            for (int a = 0; a < NT; a++) {
                rec[a] = th[a];
                if (a < NXS) {
                    double acc = 0.0;
#pragma unroll
                    for (int c = 0; c < NXS; c++) acc = __builtin_fma(UNI ? cF[UNI ? a * NXS + c : 0] : sF[a * NXP + c], th[c], acc);
#pragma unroll
                    for (int l = 0; l < NUP; l++) acc = __builtin_fma(UNI ? cG[UNI ? a : 0] : sG[a * kMaxSimU + l], uo[l], acc);
                    rec[a] = a < nx ? acc : th[a];
                }
            }
            // the previous-control block of the record (uniform branches: nup is 0 or 1 for most controllers)
#pragma unroll
            for (int l = 0; l < NUP; l++)
                if (l < nup) {
#pragma unroll
                    for (int a = 1; a < NT; a++) rec[a] = (a == nx + nr + l) ? uo[l] : rec[a];
                }
            if (U_traj) {
#pragma unroll
                for (int l = 0; l < kMaxSimU; l++)
                    if (l < nu) U_traj[((long long)k * nprob + pid) * nu + l] = uo[l];
            }
            if (X_traj) {
                double *xd = X_traj + ((long long)(k + 1) * nprob + pid) * nx;
                if (xvec) {          // even nx, 16-byte aligned rows: half as many store instructions
#pragma unroll
                    for (int a = 0; a + 1 < NXS; a += 2)
                        if (a < nx) *reinterpret_cast<double2 *>(xd + a) = make_double2(rec[a], rec[a + 1]);
                } else {
#pragma unroll
                    for (int a = 0; a < NXS; a++)
                        if (a < nx) xd[a] = rec[a];
                }
            }
#pragma unroll
            for (int t = 0; t < NT; t++) th[t] = rec[t];
            k++;
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>;
    using IU = std::integral_constant<int, kMaxSimU>;
    if constexpr (SMALL) {       // host guarantees nxp == 4, nu == 1, 1 <= m <= 8
        switch (m) {
            case 1: run(I4{}, I1{}, std::integral_constant<int, 1>{}); break;
            case 2: run(I4{}, I1{}, std::integral_constant<int, 2>{}); break;
            case 3: run(I4{}, I1{}, std::integral_constant<int, 3>{}); break;
            case 4: run(I4{}, I1{}, std::integral_constant<int, 4>{}); break;
            case 5: run(I4{}, I1{}, std::integral_constant<int, 5>{}); break;
            case 6: run(I4{}, I1{}, std::integral_constant<int, 6>{}); break;
            case 7: run(I4{}, I1{}, std::integral_constant<int, 7>{}); break;
            default: run(I4{}, I1{}, std::integral_constant<int, 8>{}); break;
        }
    } else if (nxp == 4 && nu == 1) run(I4{}, I1{}, I0{});
    else if (nxp == 4) run(I4{}, IU{}, I0{});
    else if (nu == 1) run(I8{}, I1{}, I0{});
    else run(I8{}, IU{}, I0{});
    if (valid && (k > k0 || formed)) {       // (a record formed here is stored even if its first step needs iterations)
        double *dst = theta + pid * NT;
#pragma unroll
        for (int t = 0; t < NT; t++) dst[t] = th[t];
    }
    if (valid && k > k0) {
        kstep[pid] = k;
        if (active)
            for (int w = 0; w < P.words; w++) active[pid * P.words + w] = 0ull;   // the last step left no active row
        if (flag_min) flag_min[pid] = (k0 == 0) ? (int)EXIT_OPTIMAL
                                                : (EXIT_OPTIMAL < flag_min[pid] ? (int)EXIT_OPTIMAL : flag_min[pid]);
    }
    // scenarios that stopped at a step which needs iterations go on the work list (one atomic per wavefront);
    // scenarios that used up their step allowance without meeting such a step are parked: a wavefront of a
    // short list would otherwise spend ~2 us per step on a handful of lanes.  The host runs the parked
    // scenarios on, compacted, once the work list has drained.
    const bool queue = valid && k < T && hard;
    const unsigned long long mask = __ballot(queue);
    int basei = 0;
    if (mask != 0ull && lane == 0) basei = atomicAdd(&count[shard * kCountStride], __popcll(mask));
    basei = __shfl(basei, 0);
    if (queue) list[(long long)shard * seg_cap + basei + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)pid;
    const bool park = valid && k < T && !hard;
    const unsigned long long pmask = __ballot(park);
    int pbase = 0;
    if (pmask != 0ull && lane == 0) pbase = atomicAdd(&park_count[shard * kCountStride], __popcll(pmask));
    pbase = __shfl(pbase, 0);
    if (park) park_list[(long long)shard * seg_cap + pbase + __popcll(pmask & ((1ull << lane) - 1ull))] = (int32_t)pid;
}

}  // namespace lmpc
