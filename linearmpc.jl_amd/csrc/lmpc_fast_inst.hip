// Instantiations and launch of the one-launch solver for small box-constrained problems
// (lmpc_fast_kernel.hpp).  Its own translation unit: the kernel set builds next to the others.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lmpc_internal.hpp"
#include "lmpc_fast_kernel.hpp"

namespace lmpc {

// Which handles take the fast kernel for a cold, plain (no closed loop, no generated-controller gather) batch:
// the lane path's boxed problems up to n = 5 with up to 16 parameters and no IMMUTABLE / ACTIVE-flagged rows,
// and settings under which the tiers cannot meet a guard of the generic loop (iteration limit, cycle counter).
bool fast_covers(const lmpc_handle *h) {
    const HostPack &P = h->P;
    constexpr int kmax = LMPC_FAST_KMAX;
    // (n = 6, and n = 5 with more than 8 parameters, would spill at three wavefronts per SIMD: they keep the
    // two-kernel form)
    return h->fastPath && !h->useWave && h->laneN >= 2 && h->laneN <= 5 && P.n == h->laneN && P.m == P.n && P.ms == P.m &&
           P.nth >= 1 && P.nth <= (h->laneN == 5 ? 8 : 16) && h->L.eq_mask == 0ull && h->L.imm_mask == 0ull && P.nsoft == 0 && !h->bnb &&
           h->S.iter_limit > kmax + 1 && h->S.cycle_tol >= kmax + 1;
}

template <int NTHMAX, int NT, int N, bool GATHER>
static int launch_fast_t(lmpc_handle *h, int64_t nprob, const double *theta, double *x, int32_t *flag, int32_t *iters,
                         uint64_t *active, hipStream_t st) {
    // R tiles of 64 problems per workgroup: enough that a workgroup's queue fills a 64-problem claim several
    // times over, few enough that the grid is a couple of workgroups per CU (one resident round)
    const int nstr = h->fastNstr >= 1 && h->fastNstr <= 4 ? h->fastNstr : 3;
    const long long ntiles = (nprob + 63) / 64;
    // ONE resident round with the same number of workgroups on every CU (LMPC_FAST_WAVES per CU: four
    // wavefronts each): with 652 workgroups on 768 slots the CUs that got three were still streaming 5 us
    // after those that got two had finished (tools/fast_trace.py)
    long long slots = (long long)h->numCU * LMPC_FAST_WAVES;
    long long Rl = (ntiles + slots - 1) / slots;
    if (Rl < 8) Rl = 8;
    if (h->fastTiles > 0) Rl = h->fastTiles;
    // The workgroup's queue holds one index per problem of its R tiles (256 R bytes of LDS next to ~23 KB of fixed
    // data): R is capped so that three workgroups keep fitting one CU (kFastMaxTiles = 96: 47 KB per workgroup);
    // beyond ~6e6 points per call the grid grows past one resident round instead (tests: N = 3e7)
    if (Rl > kFastMaxTiles) Rl = kFastMaxTiles;
    if (Rl > ntiles) Rl = ntiles > 0 ? ntiles : 1;
    const int R = (int)Rl;
    const unsigned grid = (unsigned)((ntiles + R - 1) / R);
    // static share + dynamic tail ("fast_dyn": tiles per workgroup handed out through global tickets instead of owned;
    // the queue then needs room for Rs + Dcap tiles).  OFF by default: it levels the streams' ends across the XCDs as
    // intended (tools/fast_trace_cu.py: all eight XCDs within 0.5 us of each other instead of up to 4 us apart) but the
    // levelled end is 1.3 us LATER than the static split's median, and a call takes 23.25 us instead of 22.2
    // (tools/hot_ab.py, same box) -- the tickets' device-scope atomics cost what the levelling saves.
    int D = GATHER ? 0 : (h->fastDyn >= 0 ? h->fastDyn : 0);
    if (R < 12 || nprob >= (int64_t)0x7fffff00 || D >= R) D = 0;
    const int Rs = R - D, Dcap = 3 * D, Rq = D ? Rs + Dcap + 1 : R;      // (+ 1: the batch's partial tile, first workgroup)
    // streaming wavefronts take their records by LDS-DMA into a ring of dk tile slots each ("fast_dma": 0 = through
    // registers, 2 / 3 = ring depth; default LMPC_FAST_DMA_DEPTH); the generated controller's gather stays on registers
    // (default: a ring of two tiles with three streaming wavefronts; with four of them -- the shape for several batches
    // in flight -- the rings would cost the third workgroup per CU its LDS: registers there)
    int dk = GATHER ? 0 : (h->fastDma >= 0 ? h->fastDma : (nstr <= 3 ? LMPC_FAST_DMA_DEPTH : 0));
    if (dk == 1 || dk > 3) dk = dk == 1 ? 0 : 3;
    if (nprob * (int64_t)NT * 8 < 16) dk = 0;
    // 16-byte pieces: a batch that does not start on a 16-byte boundary (a view into a larger array) or whose tiles
    // are not multiples of 16 bytes takes its records through registers
    if ((reinterpret_cast<uintptr_t>(theta) & 15u) != 0 || (fast_tile_bytes(NT) & 15u) != 0) dk = 0;
    const size_t lds = dk ? fast_lds_bytes_dma(N, Rq, NTHMAX, NT, nstr, dk) : fast_lds_bytes(N, Rq, NTHMAX);
    int32_t *ctrNow = nullptr, *ctrNext = nullptr;
    if (D) {
        if (!h->dFastCtr) {
            HIP_TRY(h, hipMalloc(&h->dFastCtr, sizeof(int32_t) * 2 * kFastCtrs * 32));
            HIP_TRY(h, hipMemsetAsync(h->dFastCtr, 0, sizeof(int32_t) * 2 * kFastCtrs * 32, st));
        }
        ctrNow = h->dFastCtr + (size_t)h->fastCtrSet * kFastCtrs * 32;
        ctrNext = h->dFastCtr + (size_t)(h->fastCtrSet ^ 1) * kFastCtrs * 32;
        h->fastCtrSet ^= 1;
    }
    auto kern = fast_kernel<NTHMAX, NT, N, GATHER>;
    if (lds > 48 * 1024)
        HIP_TRY(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (!h->dFastErr) {
#ifdef LMPC_FAST_TRACE
        const size_t eb = 64 + sizeof(long long) * 8 * 4 * 65536;
        HIP_TRY(h, hipMalloc(&h->dFastErr, eb));
        HIP_TRY(h, hipMemsetAsync(h->dFastErr, 0, eb, st));
#else
        // the kernel's error word lives in pinned host memory mapped into the device: the kernel writes it in the
        // (never expected) case that one of its bounded waits runs out, the host reads it without a copy or a
        // synchronisation at the handle's next call and in lmpc_check / lmpc_profile_read / lmpc_release_scratch
        int32_t *hp = nullptr;
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void **>(&hp), 64, hipHostMallocMapped));
        *hp = 0;
        h->hFastErr = hp;
        HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void **>(&h->dFastErr), hp, 0));
#endif
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, h->L, h->dC, theta, x, flag, iters,
                       active, (long long)nprob, Rq, nstr, h->dFastErr, h->fastSpinLimit > 0 ? h->fastSpinLimit - 1 : kFastSpinLimit,
                       dk, Rs, Dcap, ctrNow, ctrNext);
    HIP_TRY(h, hipGetLastError());
#ifdef LMPC_FAST_TRACE
    if (const char *f = std::getenv("LMPC_FAST_TRACE_FILE")) {
        std::vector<long long> tr((size_t)grid * 4 * 8);
        if (hipStreamSynchronize(st) == hipSuccess &&
            hipMemcpy(tr.data(), reinterpret_cast<char *>(h->dFastErr) + 64, sizeof(long long) * tr.size(), hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *fp = std::fopen(f, "wb")) { std::fwrite(tr.data(), sizeof(long long), tr.size(), fp); std::fclose(fp); }
        }
    }
#endif
    return LMPC_OK;
}

// Several equally sized batches in one launch (see fast_kernel_multi): the launch shape of a single batch, gridDim.y batches.
template <int NTHMAX, int NT, int N>
static int launch_fast_multi_t(lmpc_handle *h, int nb, int64_t nprob, const double *const *theta, double *const *x,
                               int32_t *const *flag, hipStream_t st) {
    const int nstr = h->fastNstr >= 1 && h->fastNstr <= 4 ? h->fastNstr : 3;
    const long long ntiles = (nprob + 63) / 64;
    long long slots = (long long)h->numCU * LMPC_FAST_WAVES;
    long long Rl = (ntiles + slots - 1) / slots;
    if (Rl < 8) Rl = 8;
    if (h->fastTiles > 0) Rl = h->fastTiles;
    if (Rl > kFastMaxTiles) Rl = kFastMaxTiles;
    if (Rl > ntiles) Rl = ntiles > 0 ? ntiles : 1;
    const int R = (int)Rl;
    const unsigned grid = (unsigned)((ntiles + R - 1) / R);
    int dk = h->fastDma >= 0 ? h->fastDma : (nstr <= 3 ? LMPC_FAST_DMA_DEPTH : 0);
    if (dk == 1 || dk > 3) dk = dk == 1 ? 0 : 3;
    if (nprob * (int64_t)NT * 8 < 16 || (fast_tile_bytes(NT) & 15u) != 0) dk = 0;
    FastBatches B{};
    for (int b = 0; b < nb; b++) {
        B.theta[b] = theta[b]; B.x[b] = x[b]; B.flag[b] = flag[b];
        if ((reinterpret_cast<uintptr_t>(theta[b]) & 15u) != 0) dk = 0;
    }
    const size_t lds = dk ? fast_lds_bytes_dma(N, R, NTHMAX, NT, nstr, dk) : fast_lds_bytes(N, R, NTHMAX);
    auto kern = fast_kernel_multi<NTHMAX, NT, N>;
    if (lds > 48 * 1024)
        HIP_TRY(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (!h->dFastErr) {
        int32_t *hp = nullptr;
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void **>(&hp), 64, hipHostMallocMapped));
        *hp = 0;
        h->hFastErr = hp;
        HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void **>(&h->dFastErr), hp, 0));
    }
    hipLaunchKernelGGL(kern, dim3(grid, (unsigned)nb), dim3(256), lds, st, h->L, h->dC, B, (long long)nprob, R, nstr, h->dFastErr,
                       h->fastSpinLimit > 0 ? h->fastSpinLimit - 1 : kFastSpinLimit, dk);
    HIP_TRY(h, hipGetLastError());
    return LMPC_OK;
}

// nb <= kFastMaxBatches batches of nprob points each on a handle the one-launch kernel covers; LMPC_ERR_UNSUPPORTED where
// no instantiation exists (the caller then falls back to one call per batch)
int launch_fast_multi(lmpc_handle *h, int nb, int64_t nprob, const double *const *theta, double *const *x, int32_t *const *flag,
                      hipStream_t st) {
    if (nb < 1 || nb > kFastMaxBatches) return LMPC_ERR_UNSUPPORTED;
#ifdef LMPC_FAST_TRACE
    return LMPC_ERR_UNSUPPORTED;
#else
#define LMPC_FM(NM, NT)                                                                                     \
    switch (h->laneN) {                                                                                     \
        case 2: return launch_fast_multi_t<NM, NT, 2>(h, nb, nprob, theta, x, flag, st);                    \
        case 3: return launch_fast_multi_t<NM, NT, 3>(h, nb, nprob, theta, x, flag, st);                    \
        case 4: return launch_fast_multi_t<NM, NT, 4>(h, nb, nprob, theta, x, flag, st);                    \
        case 5: if constexpr (NT <= 8) return launch_fast_multi_t<NM, NT, 5>(h, nb, nprob, theta, x, flag, st); break; \
        default: break;                                                                                     \
    }                                                                                                       \
    break;
    switch (h->P.nth) {
#ifdef LMPC_FAST_ONLY_PENDULUM
        case 7: if (h->laneN == 5) return launch_fast_multi_t<8, 7, 5>(h, nb, nprob, theta, x, flag, st); break;
#else
        case 1: LMPC_FM(8, 1)   case 2: LMPC_FM(8, 2)   case 3: LMPC_FM(8, 3)   case 4: LMPC_FM(8, 4)
        case 5: LMPC_FM(8, 5)   case 6: LMPC_FM(8, 6)   case 7: LMPC_FM(8, 7)   case 8: LMPC_FM(8, 8)
        case 9: LMPC_FM(16, 9)  case 10: LMPC_FM(16, 10) case 11: LMPC_FM(16, 11) case 12: LMPC_FM(16, 12)
        case 13: LMPC_FM(16, 13) case 14: LMPC_FM(16, 14) case 15: LMPC_FM(16, 15) case 16: LMPC_FM(16, 16)
#endif
        default: break;
    }
#undef LMPC_FM
    return LMPC_ERR_UNSUPPORTED;
#endif
}

int check_fast_err(lmpc_handle *h) {
    if (!h->hFastErr) return LMPC_OK;
    const int32_t e = *h->hFastErr;
    if (e == 0) return LMPC_OK;
    *h->hFastErr = 0;                                      // reported once
    return fail(h, LMPC_ERR_HIP, "lmpc: the one-launch kernel gave up a bounded wait (code " + std::to_string(e) +
                                 ") in an earlier call on this handle: problems it had queued keep exit flag -8 "
                                 "(LMPC_EXIT_UNFINISHED), their x is not a solution");
}

// this unit's code object onto the device without a launch (setup; see preload_code in lmpc_api.hip)
void fast_preload(lmpc_handle *h) {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void *)fast_kernel<8, 7, 5, false>);
    // ... and what the kernel's first launch on this handle would allocate: its ticket counters, its error word
    if (h && !h->dFastCtr && hipMalloc(&h->dFastCtr, sizeof(int32_t) * 2 * kFastCtrs * 32) == hipSuccess)
        (void)hipMemset(h->dFastCtr, 0, sizeof(int32_t) * 2 * kFastCtrs * 32);
#ifndef LMPC_FAST_TRACE
    if (h && !h->dFastErr) {
        int32_t *hp = nullptr;
        if (hipHostMalloc(reinterpret_cast<void **>(&hp), 64, hipHostMallocMapped) == hipSuccess) {
            *hp = 0;
            if (hipHostGetDevicePointer(reinterpret_cast<void **>(&h->dFastErr), hp, 0) == hipSuccess) h->hFastErr = hp;
            else { (void)hipHostFree(hp); h->dFastErr = nullptr; }
        }
    }
#endif
    (void)hipGetLastError();
}

int launch_fast(lmpc_handle *h, int64_t nprob, const double *theta, double *x, int32_t *flag, int32_t *iters,
                uint64_t *active, hipStream_t st) {
    const bool gather = h->L.gat.state != nullptr;        // generated-controller call: theta from the five arrays
#define LMPC_FN(NM, NT)                                                                                               \
    switch (h->laneN) {                                                                                               \
        case 2: return gather ? launch_fast_t<NM, NT, 2, true>(h, nprob, theta, x, flag, iters, active, st) : launch_fast_t<NM, NT, 2, false>(h, nprob, theta, x, flag, iters, active, st);    \
        case 3: return gather ? launch_fast_t<NM, NT, 3, true>(h, nprob, theta, x, flag, iters, active, st) : launch_fast_t<NM, NT, 3, false>(h, nprob, theta, x, flag, iters, active, st);    \
        case 4: return gather ? launch_fast_t<NM, NT, 4, true>(h, nprob, theta, x, flag, iters, active, st) : launch_fast_t<NM, NT, 4, false>(h, nprob, theta, x, flag, iters, active, st);    \
        case 5: if constexpr (NT <= 8) return gather ? launch_fast_t<NM, NT, 5, true>(h, nprob, theta, x, flag, iters, active, st) : launch_fast_t<NM, NT, 5, false>(h, nprob, theta, x, flag, iters, active, st); break; \
        default: break;                                                                                               \
    }                                                                                                                 \
    break;
    switch (h->P.nth) {
#ifdef LMPC_FAST_ONLY_PENDULUM
        case 7: if (h->laneN == 5) return gather ? launch_fast_t<8, 7, 5, true>(h, nprob, theta, x, flag, iters, active, st) : launch_fast_t<8, 7, 5, false>(h, nprob, theta, x, flag, iters, active, st); break;
#else
        case 1: LMPC_FN(8, 1)   case 2: LMPC_FN(8, 2)   case 3: LMPC_FN(8, 3)   case 4: LMPC_FN(8, 4)
        case 5: LMPC_FN(8, 5)   case 6: LMPC_FN(8, 6)   case 7: LMPC_FN(8, 7)   case 8: LMPC_FN(8, 8)
        case 9: LMPC_FN(16, 9)  case 10: LMPC_FN(16, 10) case 11: LMPC_FN(16, 11) case 12: LMPC_FN(16, 12)
        case 13: LMPC_FN(16, 13) case 14: LMPC_FN(16, 14) case 15: LMPC_FN(16, 15) case 16: LMPC_FN(16, 16)
#endif
        default: break;
    }
#undef LMPC_FN
    return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: no fast-kernel instantiation");
}

}  // namespace lmpc
