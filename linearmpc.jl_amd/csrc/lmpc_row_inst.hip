// Launch of the four-problems-per-wavefront kernel (lmpc_row_kernel.hpp): which instantiation covers a problem, its
// launch shape, and the counter protocol it shares with the wavefront kernel's launches (lmpc_wave_launch.hpp) -- it
// runs as the only pass of a batch (capacity = the problem's) or as the FIRST of two (a smaller capacity; what outgrows
// it is listed for the wavefront kernel's second pass).
#include "lmpc_internal.hpp"
#include "lmpc_row_kernel.hpp"

namespace lmpc {

namespace {

struct RowShape { int S, NS, MS, CAPP; };

// the instantiations built (LMPC_ROW_REAL per translation unit): position / variable / constraint slots, leading dimension
constexpr RowShape kRowShapes[] = {
    {1, 1, 4, 16},      // n <= 16, m <= 64, working sets <= 16 rows (the reference's mass_spring example: n = 10, m = 63)
    {2, 2, 6, 31},      // n <= 32, m <= 96, <= 31 rows (BASELINE config 3: n = 30, m = 84)
    {2, 2, 6, 32},      // ... <= 32 rows
    {1, 4, 10, 16},     // n <= 64, m <= 160, <= 16 rows: two wavefronts per SIMD (256 registers), eight per CU next to 67 KB of M'
                        // (pendulum_N50: 99 % of the points that iterate stay within 16 rows)
    {2, 4, 10, 31},     // n <= 64, m <= 160, <= 31 rows (the reference's benchmark class at N = 50: first of two passes -- at 31
                        // rows, the capacity whose column order is free of bank conflicts)
    {2, 4, 10, 32},     // ... <= 32 rows
};

// ... with branch and bound (rows flagged BINARY)
constexpr RowShape kRowShapesBnb[] = {
    {1, 1, 2, 16},      // n <= 16, m <= 32, <= 16 rows (the satellite example at Np = 4: n = m = 12)
    {3, 4, 4, 44},      // n <= 64, m <= 64, <= 44 rows (BASELINE config 5, satellite Np = 20: n = m = 60, 40 binaries -- first of two
                        // passes; eight wavefronts per CU in binary32)
    {3, 4, 4, 48},      // ... <= 48 rows (seven)
};

bool shape_covers(const RowShape &sh, int n, int m, int cap) {
    return n <= 16 * sh.NS && m <= 16 * sh.MS && cap <= sh.CAPP && m <= 256 /* mask words on 16 lanes */;
}

struct RowLaunch { int shape, nwv, blocks, ps; size_t lds; int aux; };

bool row_launch_for(const lmpc_handle *h, int cap, size_t rs, RowLaunch *out, bool bnb = false) {
    const int n = h->P.n, m = h->P.m;
    const RowShape *shapes = bnb ? kRowShapesBnb : kRowShapes;
    const int nshapes = bnb ? (int)(sizeof(kRowShapesBnb) / sizeof(kRowShapesBnb[0])) : (int)(sizeof(kRowShapes) / sizeof(kRowShapes[0]));
    for (int q = 0; q < nshapes; q++) {
        const RowShape &sh = shapes[q];
        if (!shape_covers(sh, n, m, cap)) continue;
        if (bnb && h->nBinary > row_bnb_depth_max(sh.MS)) continue;
        // most wavefronts per CU (each carries four problems; two per SIMD is what the registers allow); the staged M' is
        // shared by a workgroup's wavefronts
        RowLaunch best{-1, 0, 0, 0, 0, 0};
        int bestWaves = 0;
        const size_t mt = (size_t)((n + 3) & ~3) * row_mpad(sh.MS, (int)rs);
        for (int nwv : {8, 7, 6, 5, 4, 3, 2, 1}) {
            if (64 * nwv > row_launch_bound(sh.MS, sh.S, (int)rs)) continue;
            const int ps = (bnb && row_copy16(sh.CAPP, (int)rs)) ? row_ps4(sh.CAPP, nwv) : row_ps(sh.CAPP, nwv);
            if (ps < 0) continue;
            size_t lds = rs * (32 + mt + (size_t)nwv * 4 * ps + 2) + sizeof(int32_t) * ((size_t)m + sh.CAPP) + 16;
            if (lds > kLdsMax) continue;
            // ten-slot shapes: the per-problem constants in LDS too (row_kernel: AUXS) -- or a smaller workgroup
            int aux = 0;
            if (sh.MS >= 10) {
                const size_t front = rs * (32 + mt + (size_t)nwv * 4 * ps + 2) + sizeof(int32_t) * ((size_t)m + sh.CAPP);
                lds = row_aux_offset(front) + rs * (size_t)row_aux_reals(h->W) + 16;
                if (lds > kLdsMax) continue;
                aux = 1;
            }
            int blocks = (int)(kLdsMax / lds);
            const int maxw = 4 * row_waves_per_simd(sh.S, sh.MS, (int)rs);
            if (blocks * nwv > maxw) blocks = maxw / nwv;
            if (blocks < 1) continue;
            const int waves = blocks * nwv;
            if (waves > bestWaves) { bestWaves = waves; best = RowLaunch{q, nwv, blocks, ps, lds, aux}; }
        }
        if (best.shape < 0) return false;
        *out = best;
        return true;
    }
    return false;
}

template <typename R, int S, int NS, int MS, int CAPP, bool BNB = false>
int launch_row_shape(lmpc_handle *h, const RowLaunch &rl, const R *dC, int64_t nprob, const R *theta, R *x, int32_t *flag,
                     int32_t *iters, uint64_t *active, hipStream_t st, int cap, int pass) {
    auto kern = row_kernel<R, S, NS, MS, CAPP, BNB>;
    if (rl.lds > 48 * 1024)
        HIP_TRY(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl.lds));
    if (h->preloadOnly) {
        hipFuncAttributes fa;
        HIP_TRY(h, hipFuncGetAttributes(&fa, (const void *)kern));
        return LMPC_OK;
    }
    WaveLayout Wl = h->W;
    Wl.cap = cap; Wl.ldc = 0;
    int blocks = rl.blocks;
    if (h->rowBlocks > 0) blocks = h->rowBlocks;
    long long grid = (long long)h->numCU * blocks;
    const long long need = (nprob + 4 * rl.nwv - 1) / (4 * rl.nwv);
    if (grid > need) grid = need;
    R *bnbR = nullptr;
    int32_t *bnbI = nullptr;
    const int bdepth = h->nBinary > 0 ? h->nBinary : 1;
    if constexpr (BNB) {
        // one snapshot slot per search depth and problem row of the grid, at most half of what the device has free (a
        // smaller grid before an allocation that crowds out the caller: launch_wave_cfg's rule)
        const size_t perRow = (size_t)bdepth * (sizeof(R) * (size_t)row_snap_reals(S, CAPP) + sizeof(int32_t) * (size_t)row_snap_ints(S));
        auto bytes = [&](long long gr) { return perRow * (size_t)gr * (size_t)rl.nwv * 4; };
        if (bytes(grid) > h->rowBnbBytes) {
            size_t freeB = 0, totalB = 0;
            if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { freeB = 0; (void)hipGetLastError(); }
            const size_t budget = (freeB + h->rowBnbBytes) / 2;
            while (grid > 1 && bytes(grid) > budget) grid = (grid + 1) / 2;
            if (bytes(grid) > budget)
                return fail(h, LMPC_ERR_HIP, "lmpc: not enough free device memory for the branch-and-bound snapshots");
            if (bytes(grid) > h->rowBnbBytes) {
                hipFree(h->dRowBnb); h->dRowBnb = nullptr; h->rowBnbBytes = 0;
                HIP_TRY(h, hipMalloc(&h->dRowBnb, bytes(grid)));
                h->rowBnbBytes = bytes(grid);
            }
        }
        const size_t nrowsB = (size_t)grid * rl.nwv * 4;
        bnbR = static_cast<R *>(h->dRowBnb);
        bnbI = reinterpret_cast<int32_t *>(static_cast<char *>(h->dRowBnb) + sizeof(R) * nrowsB * bdepth * (size_t)row_snap_reals(S, CAPP));
    }
    // counters: the protocol of launch_wave_cfg (two alternating sets, each launch clears the next one's)
    constexpr int kP1 = kShards * kCountStride;
    if (!h->dQueue) {
        if (!h->dOvfCount) HIP_TRY(h, hipMalloc(&h->dOvfCount, sizeof(int32_t) * (2 * kP1 + 64)));
        HIP_TRY(h, hipMemsetAsync(h->dOvfCount, 0, sizeof(int32_t) * (2 * kP1 + 64), st));
        HIP_TRY(h, hipMalloc(&h->dQueue, 64));
        HIP_TRY(h, hipMemsetAsync(h->dQueue, 0, 64, st));
        h->waveCtrSet = 0; h->waveOvfSet = 0;
    }
    const int os = h->waveOvfSet;
    int32_t *const p1Count = h->dOvfCount + os * kP1, *const p1Next = h->dOvfCount + (os ^ 1) * kP1;
    int32_t *const p2Next = h->dOvfCount + 2 * kP1 + 8 * (os ^ 1);
    int32_t *const queueNext = h->dQueue + 8 * (h->waveCtrSet ^ 1);
    int32_t *queue = nullptr;
    int qchunk = 1;
    const long long nrows = grid * rl.nwv * 4;
    if (h->waveList.list != nullptr) {
        queue = h->dQueue + 8 * h->waveCtrSet;
    } else if (nprob > 2 * nrows && h->waveQueue) {
        const long long q = nprob / (32 * nrows);
        qchunk = q < 1 ? 1 : (q > 16 ? 16 : (int)q);
        queue = h->dQueue + 8 * h->waveCtrSet;
    }
    if (pass == 1 && nprob > h->ovfCap1) {
        hipFree(h->dOvfList1); h->dOvfList1 = nullptr; h->ovfCap1 = 0;
        HIP_TRY(h, hipMalloc(&h->dOvfList1, sizeof(int32_t) * (size_t)nprob));
        h->ovfCap1 = nprob;
    }
    if (!BNB && !h->hStat) {
        unsigned long long *hp = nullptr;
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void **>(&hp), 64, hipHostMallocMapped));
        for (int q = 0; q < 8; q++) hp[q] = 0ull;
        h->hStat = hp;
        HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void **>(&h->dStatHost), hp, 0));
        HIP_TRY(h, hipMalloc(&h->dStat, sizeof(unsigned long long) * 64 * 16));
        HIP_TRY(h, hipMemsetAsync(h->dStat, 0, sizeof(unsigned long long) * 64 * 16, st));
    }
    const WaveList &wl = h->waveList;
    RowParams<R> prm{};
    prm.P = Wl; prm.C = dC; prm.Sg = h->dSw; prm.theta = theta; prm.X = x; prm.exitflag = flag; prm.iters = iters;
    prm.active = active; prm.queue = queue; prm.qchunk = qchunk; prm.ps = rl.ps; prm.nprob = (long long)nprob;
    prm.list = wl.list; prm.count = wl.count; prm.count_next = wl.count_next; prm.seg_cap = wl.seg_cap;
    prm.ovf_list = pass == 1 ? h->dOvfList1 : nullptr; prm.ovf_count = pass == 1 ? p1Count : nullptr;
    prm.queue_next = queueNext; prm.ovf_next = p2Next; prm.ovf_next1 = p1Next;
    prm.stat = BNB ? nullptr : h->dStat; prm.stat_host = BNB ? nullptr : h->dStatHost;
    prm.bnb_r = bnbR; prm.bnb_i = bnbI; prm.bnb_depth = bdepth;
    prm.aux = rl.aux;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * rl.nwv), rl.lds, st, prm);
    h->waveCtrSet ^= 1;
    HIP_TRY(h, hipGetLastError());
    return LMPC_OK;
}

}  // namespace

#ifdef LMPC_ROW_HELPERS
// Working-set capacity at which a batch of this handle would first run on the row kernel: the problem's own capacity
// (one pass), a smaller one (first of two passes; the statistics of the handle's earlier launches must say that nearly
// all working sets stay within it), or 0 = the row kernel does not take this batch.
int row_pass_cap(lmpc_handle *h, int64_t nprob, size_t rs, bool warm, bool gram, bool bnb) {
    if (h->rowKernel == 0 || bnb || gram || warm || h->avi) return 0;
    if (h->waveSim.FG != nullptr || h->keepOn || nprob >= (int64_t)0x3fffffff) return 0;
    // (small batches: the two-slot shape is ahead down to a single problem -- config 3's class 0.14 against 0.17 ms for one
    // solve, 0.53 against 0.63 ms for 512 --, the one-slot and ten-slot shapes only from a few thousand problems on)
    const bool twoSlot = h->P.n > 16 && h->P.n <= 32 && h->P.m <= 96;
    if (h->rowKernel < 0 && nprob < 8192 && !twoSlot) return 0;
    for (int j = 0; j < h->P.m; j++)
        if (h->P.sense[j] & (SENSE_ACTIVE | SENSE_BINARY)) return 0;
    if (h->S.iter_limit < 2) return 0;
    const int full = h->W.cap;
    RowLaunch rl;
    int cap = 0;
    // binary32 by default only in the two-slot shape (config 3's class: 15.4 -> 9.4 ms per 4e5; soft_doc level, the one-slot
    // and ten-slot shapes behind the wavefront kernel there: no tiers / screening pass in front of them in binary32)
    if (rs == 4 && h->rowKernel < 0 && !(h->P.n > 16 && h->P.n <= 32 && h->P.m <= 96)) return 0;
    // where it is the default (measured, tools/row_check.py): every shape built -- config 3 runs 1.9x the wavefront kernel,
    // soft_doc (n = 10, SOFT rows, 12 iterations) 1.7x, mass_spring behind the tiers pass level, the ten-slot shape (one
    // wavefront per SIMD; pendulum_N50 behind the screening pass) 1.2x
    if (h->waveTwoPass > 0 && h->waveCap1 > 0 && h->waveCap1 < full && h->bigPath) {      // (a first pass at the caller's capacity)
        if (row_launch_for(h, h->waveCap1, rs, &rl)) cap = h->waveCap1;
    } else
    if (full <= 32 && row_launch_for(h, full, rs, &rl)) cap = full;
    else if (full > 32 && h->bigPath && h->waveTwoPass != 0) {
        // first of two passes at 31 rows: when at most 1 in 20 of the working sets seen lately went beyond 24 (the
        // statistics' bucket below; 31 is the capacity whose column order is free of bank conflicts)
        unsigned long long sum[4] = {0, 0, 0, 0};
        wave_stat_read(h, sum);
        const bool known = sum[0] >= 1000ull;
        const bool fits = known && (sum[0] - sum[2]) * 20ull <= sum[0];
        // ... at 16 rows for the ten-slot class when 98 of 100 problems of the handle's life stayed within them (the one-slot
        // shape: eight wavefronts per CU instead of four -- pendulum_N50 1.33 -> 1.16 ms per 2e5)
        unsigned long long tot = 0ull, le16 = 0ull;
        if (h->hStat) { for (int q = 0; q < 4; q++) tot += h->hStat[q]; le16 = h->hStat[4]; }
        const bool fits16 = tot >= 1000ull && le16 <= tot && (tot - le16) * 50ull <= tot && (h->P.n > 32 || h->P.m > 96);
        if (fits16 && row_launch_for(h, 16, rs, &rl)) cap = 16;
        else if ((h->rowKernel > 0 || fits) && row_launch_for(h, 31, rs, &rl)) cap = 31;
    }
    return cap;
}

// ... for a handle with binary rows: the problem's own capacity if 16 rows hold it, else 48 rows as the first
// of two passes when that leaves eight rows beyond the binaries (the wavefront kernel's rule, bnb_first_pass_cap)
int row_bnb_pass_cap(lmpc_handle *h, int64_t nprob, size_t rs) {
    if (h->rowKernel == 0 || !h->bnb || h->avi || h->waveGram != 0) return 0;
    if (h->waveSim.FG != nullptr || h->keepOn || nprob >= (int64_t)0x3fffffff) return 0;
    // (whatever the batch size: a search takes 110 trips here against 185 iterations + 41 node starts on the wavefront
    // kernel -- ONE search 0.98 against 1.50 ms in binary32, 1.18 against 1.84 ms in binary64)
    for (int j = 0; j < h->P.m; j++)
        if (h->P.sense[j] & SENSE_ACTIVE) return 0;
    if (h->S.iter_limit < 2 || h->nBinary > 47) return 0;
    const int full = h->W.cap;
    RowLaunch rl;
    if (full <= 16) return row_launch_for(h, full, rs, &rl, true) ? full : 0;
    // (on request also as the only pass of a problem whose capacity the 48-row shape holds: the randomized parity tests)
    if (full <= 48 && h->rowKernel > 0) return row_launch_for(h, full, rs, &rl, true) ? full : 0;
    // ("wave_two_pass" 1 with "wave_cap1" c: first pass at min(c, 48) rows whatever the binaries leave -- tests of the listing)
    if (h->waveTwoPass > 0 && h->waveCap1 > 0)
        return row_launch_for(h, h->waveCap1 < 48 ? h->waveCap1 : 48, rs, &rl, true) ? (h->waveCap1 < 48 ? h->waveCap1 : 48) : 0;
    if (h->waveTwoPass == 0 || full <= 52) return 0;
    // 44 rows where that leaves four beyond the binaries (config 5: no search of 10^5 goes beyond 42), else 48
    const int c1 = h->nBinary + 4 <= 44 ? 44 : 48;
    return row_launch_for(h, c1, rs, &rl, true) ? c1 : 0;
}
#endif

template <typename R>
int launch_row(lmpc_handle *h, const R *dC, int64_t nprob, const R *theta, R *x, int32_t *flag, int32_t *iters,
               uint64_t *active, hipStream_t st, int cap, int pass) {
    RowLaunch rl;
    if (!row_launch_for(h, cap, sizeof(R), &rl)) return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: no row-kernel instantiation");
#define LMPC_ROW(S_, NS_, MS_, CAPP_) launch_row_shape<R, S_, NS_, MS_, CAPP_>(h, rl, dC, nprob, theta, x, flag, iters, active, st, cap, pass)
    switch (rl.shape) {
        case 0: return LMPC_ROW(1, 1, 4, 16);
        case 1: return LMPC_ROW(2, 2, 6, 31);
        case 2: return LMPC_ROW(2, 2, 6, 32);
        case 3: return LMPC_ROW(1, 4, 10, 16);
        case 4: return LMPC_ROW(2, 4, 10, 31);
        default: return LMPC_ROW(2, 4, 10, 32);
    }
#undef LMPC_ROW
}

#ifdef LMPC_ROW_BNB
template <typename R>
int launch_row_bnb(lmpc_handle *h, const R *dC, int64_t nprob, const R *theta, R *x, int32_t *flag, int32_t *iters,
                   uint64_t *active, hipStream_t st, int cap, int pass) {
    RowLaunch rl;
    if (!row_launch_for(h, cap, sizeof(R), &rl, true)) return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: no row-kernel instantiation");
    if (rl.shape == 0) return launch_row_shape<R, 1, 1, 2, 16, true>(h, rl, dC, nprob, theta, x, flag, iters, active, st, cap, pass);
    if (rl.shape == 1) return launch_row_shape<R, 3, 4, 4, 44, true>(h, rl, dC, nprob, theta, x, flag, iters, active, st, cap, pass);
    return launch_row_shape<R, 3, 4, 4, 48, true>(h, rl, dC, nprob, theta, x, flag, iters, active, st, cap, pass);
}
template int launch_row_bnb<LMPC_ROW_REAL>(lmpc_handle *, const LMPC_ROW_REAL *, int64_t, const LMPC_ROW_REAL *, LMPC_ROW_REAL *, int32_t *,
                                           int32_t *, uint64_t *, hipStream_t, int, int);
#else
template int launch_row<LMPC_ROW_REAL>(lmpc_handle *, const LMPC_ROW_REAL *, int64_t, const LMPC_ROW_REAL *, LMPC_ROW_REAL *, int32_t *,
                                       int32_t *, uint64_t *, hipStream_t, int, int);
#endif

}  // namespace lmpc

#if defined(LMPC_ROW_TRACE) && (defined(LMPC_ROW_HELPERS) || defined(LMPC_ROW_BNB))
// diagnostic build only: per-phase shader-clock sums of the row kernel (see lmpc_row_kernel.hpp)
extern "C" int lmpc_debug_row_trace(unsigned long long *out32, int reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(lmpc::g_row_trace), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(lmpc::g_row_trace), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
