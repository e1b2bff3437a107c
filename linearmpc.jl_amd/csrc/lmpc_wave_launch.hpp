// Launch configuration and dispatch of the wavefront kernel (included by lmpc_wave_inst.hip only).
#pragma once

#include "lmpc_internal.hpp"
#include "lmpc_wave_kernel.hpp"
#include "lmpc_big_kernel.hpp"

namespace lmpc {

// bytes of shared problem data a wave-kernel workgroup keeps in LDS at staging level `level`
// (1: M transposed, 2: + M, 3: + packed Gram), rs = sizeof(real)
// (Gram-scan form: level 1 = the full m x m Gram matrix, nothing else is read inside an iteration)
inline size_t wave_shared_bytes(const HostPack &P, int level, size_t rs, bool gram = false) {
    if (gram) return level >= 1 ? rs * (size_t)P.m * P.m : 0;
    const size_t nM = (size_t)P.m * P.n, nG = (size_t)P.m * (P.m + 1) / 2;
    return rs * (level >= 3 ? 2 * nM + nG : (level == 2 ? 2 * nM : (level == 1 ? nM : 0)));
}

struct WaveConfig { int nwv, level, blocksPerCU; size_t lds; bool packed; };

// constraint slots per lane of the instantiation that takes m rows: 1, 2, 3, 4, 5, 6, 8 or 16 (branch and bound:
// 1, 2, 4, 8, 16).  Every slot costs registers and a pass over it in each row loop, so 3 / 5 / 6 exist next to the
// powers of two: the reference's benchmark class has m = 148 / 223 / 298 / 373 rows (Np = 50 / 75 / 100 / 125).
inline int wave_slots(int m, bool bnb) {
    const int need = (m + 63) / 64;
    if (need <= 2) return need < 1 ? 1 : need;
    if (need == 3) return bnb ? 4 : 3;
    if (need == 4) return 4;
    if (need <= 6) return bnb ? 8 : need;
    return need <= 8 ? 8 : 16;
}
// resident wavefronts per CU the instantiation's registers allow (`make asm`, kernel-resource-usage): 128 VGPRs for
// 1-2 slots (4 per SIMD), 168 for 3 (3 per SIMD), up to 256 for 4-6 (2 per SIMD); 8 slots fit 256 only
// with M' staged in LDS (level >= 1) and one variable slot, otherwise -- and with 16 slots -- one wavefront per SIMD
inline int wave_max_resident(int slots, bool bnb, size_t rs, int level, int nu, bool gram = false) {
    // Gram-scan form without branch and bound (resource_usage_wave_*_gram.txt): <= 128 VGPRs up to 2 slots, <= 168 at
    // 3 and 4 slots, <= 256 up to 8 slots
    if (gram && !bnb) return slots <= 2 ? 16 : (slots <= 5 ? (LMPC_WAVE_LB4G >= 768 || slots == 3 ? 12 : 8) : (slots <= 8 ? 8 : 4));
    if (bnb) return slots <= 2 ? ((rs == 4 && slots == 1) ? 16 : 12) : (slots <= 4 ? 8 : 4);
    if (slots <= 2) return LMPC_WAVE_LB >= 1024 ? ((rs == 8 && LMPC_WAVE_WPE > 4) ? 4 * LMPC_WAVE_WPE : 16) : 12;
    if (slots == 3) return LMPC_WAVE_LB3 >= 768 ? 12 : 8;
    if (slots == 4) return LMPC_WAVE_LB4 >= 768 ? 12 : 8;
    if (slots <= 6) return 8;
    return (slots == 8 && level >= 1 && nu == 1) ? 8 : 4;
}

// Workgroup shape of the wave kernel: nwv wavefronts (= problems in flight) share one LDS copy of
// the problem data.  Registers allow 4 wavefronts per SIMD (16 per CU; 12 for the 512-thread
// instantiations); the per-wave factors and the shared copy compete for the 160 KiB of LDS.
// Measured (tools/wave_sweep.sh): resident wavefronts matter more than the staging level (3-input
// mass-spring: 16 waves with only M' staged 1.23e7/s, 8 waves with everything staged 0.85e7/s), and
// at equal residency small workgroups win.  So: most wavefronts per CU first, then the highest
// staging level that reaches it, then the smallest workgroup.
inline WaveConfig wave_config_for(const lmpc_handle *h, size_t rs, bool packed) {
    const WaveLayout &Wl = h->W;
    // per-wave factor L: square with an odd leading dimension, or packed strict lower triangle
    // (+ 64 reals: the multipliers, read back as LDS broadcasts)
    const size_t perWave = rs * ((packed ? ((size_t)Wl.cap * (Wl.cap - 1) / 2) : ((size_t)Wl.cap * Wl.ldc)) + 64);
    // instantiations with many constraint slots or the B&B state are built for 512-thread workgroups
    // (more registers per lane, fewer resident wavefronts)
    const int slots = wave_slots(h->P.m, h->bnb);
    const int maxNwv = wave_launch_bound(slots, h->bnb, h->waveGram != 0) / 64;   // workgroup size the kernel is built for
    WaveConfig best{1, 0, 1, perWave, packed};
    int bestWaves = -1;
    // packed, and everything beyond 256 rows (256-thread instantiations), is instantiated for levels 0, 1
    // (packed, the 3 / 5 / 6-slot instantiations and everything beyond 256 rows are built for levels 0 and 1 only)
    const bool twoLevels = packed || h->P.m > 256 || slots == 3 || h->waveGram;
    for (int level = h->P.n > 64 ? 0 : (twoLevels ? 1 : 3); level >= 0; level--) {
        if (h->waveLevel >= 0 && level != (twoLevels && h->waveLevel > 1 ? 1 : h->waveLevel)) continue;
        const int maxWaves = wave_max_resident(slots, h->bnb, rs, level, h->P.n > 64 ? 2 : 1, h->waveGram != 0);
        for (int nwv : {4, 8, 16, 12, 2, 1}) {
            if (nwv > maxNwv) continue;
            if (h->waveNwv > 0 && nwv != h->waveNwv && !(h->waveNwv > maxNwv && nwv == maxNwv)) continue;
            const size_t lds = perWave * nwv + wave_shared_bytes(h->P, level, rs, h->waveGram != 0);
            if (lds > kLdsMax) continue;
            int blocks = (int)(kLdsMax / lds);
            if (blocks * nwv > maxWaves) blocks = maxWaves / nwv;
            if (blocks < 1) continue;
            const int waves = blocks * nwv;
            if (waves > bestWaves) { bestWaves = waves; best = WaveConfig{nwv, level, blocks, lds, packed}; }
        }
    }
    return best;
}

// square L unless the packed layout keeps at least 1.75x the wavefronts resident (round 3: the square layout's sweeps
// run on a zero-padded factor, three vector instructions per step instead of five, so it takes more than a quarter
// more wavefronts to beat it -- hybrid f32, n = 60: 10 wavefronts per CU square 1.54e6/s, 16 packed 1.40e6/s; f64:
// 5 square 6.9e5/s, 10 packed 8.4e5/s; the benchmark class at cap = 64: 4 square against 9 packed, packed 1.5x faster)
inline WaveConfig wave_config(const lmpc_handle *h, size_t rs) {
    const WaveConfig sq = wave_config_for(h, rs, false), pk = wave_config_for(h, rs, true);
    if (h->wavePacked == 0) return sq;
    if (h->wavePacked == 1) return pk;
    // (round 4, branch and bound with lazy snapshots: the truncating restore and the zero-padded sweeps favour the square
    // layout further -- hybrid n = 60 binary64: 5 square 1.06e6/s against 10 packed 7.5e5/s, Gram-scan form 1.89e6
    // against 1.24e6; binary32: 10 square 2.24e6 against 16 packed 9.6e5 -- so there packed needs 3x the wavefronts)
    if (h->bnb) return pk.blocksPerCU * pk.nwv >= 3 * sq.blocksPerCU * sq.nwv ? pk : sq;
    return 4 * pk.blocksPerCU * pk.nwv >= 7 * sq.blocksPerCU * sq.nwv ? pk : sq;
}

// the handle's working-set statistics (four buckets in mapped host memory, see the kernel) as cumulative counts
inline void wave_stat_sums(const lmpc_handle *h, unsigned long long out[4]) {
    for (int q = 0; q < 4; q++) out[q] = 0ull;
    if (!h->hStat) return;
    unsigned long long b[4];                                     // buckets: <= 24, <= 32, <= 48, more
    for (int q = 0; q < 4; q++) b[q] = h->hStat[q];
    out[0] = b[0] + b[1] + b[2] + b[3]; out[1] = b[0]; out[2] = b[0] + b[1]; out[3] = b[0] + b[1] + b[2];
}

// ... over a WINDOW of the most recent one to two million problems (the counters of a handle's whole life
// kept a small first pass long after the parameter batches had turned hard).  statB follows the cumulative counters in
// steps of 2^20 problems, statA is the statB before it; what a decision sees is (now - statA).
inline void wave_stat_window(lmpc_handle *h, unsigned long long out[4]) {
    unsigned long long cur[4];
    wave_stat_sums(h, cur);
    if (cur[0] < h->statB[0]) {                                  // (counters restarted: lmpc_release_scratch does not, but be safe)
        for (int q = 0; q < 4; q++) h->statA[q] = h->statB[q] = 0ull;
    }
    if (cur[0] - h->statB[0] >= (1ull << 20)) {
        for (int q = 0; q < 4; q++) { h->statA[q] = h->statB[q]; h->statB[q] = cur[q]; }
    }
    for (int q = 0; q < 4; q++) out[q] = cur[q] - h->statA[q];
}

// capacity of the first of two passes for a batch of nprob problems on this handle, 0 = one pass (see launch_wave_inst)
inline int wave_first_pass_cap_impl(lmpc_handle *h, int64_t nprob, size_t rs) {
    if (h->bnb || h->waveTwoPass == 0 || !h->bigPath || h->W.cap < 40 || nprob >= (int64_t)0x7fffffff) return 0;
    // run-ahead (closed loop, consecutive steps of a scenario on the factor in LDS), warm: a step that outgrows a first
    // pass restarts from the factor of the step before it -- which the first pass then has to write out after every
    // step (the kept-state buffers of the step-synchronous loop); without them, one pass
    if (h->waveSim.FG != nullptr && h->waveSim.T > 0 && h->raWarm && !h->keepOn) return 0;
    int c1 = 0;
    if (h->waveTwoPass > 0) c1 = h->waveCap1;
    else if (h->hStat && nprob >= 4096) {
        unsigned long long sum[4];
        wave_stat_window(h, sum);
        const int caps[3] = {24, 32, 48};
        for (int q = 0; q < 3 && c1 == 0 && sum[0] >= 1000ull; q++)
            if ((sum[0] - sum[1 + q]) * 100ull <= 3ull * sum[0]) c1 = caps[q];
    }
    if (c1 >= h->W.cap) c1 = 0;
    if (c1 > 0 && h->waveTwoPass < 0) {                   // only where the smaller factor buys residency, layout or staging
        const WaveConfig full = wave_config(h, rs);
        const int capW = h->W.cap, ldcW = h->W.ldc;
        h->W.cap = c1; h->W.ldc = c1 | 1;
        const WaveConfig t = wave_config(h, rs);
        h->W.cap = capW; h->W.ldc = ldcW;
        // (a wavefront on the square factor counts 1.75 times one on the packed triangle: the weight wave_config itself
        // chooses the layout by -- N = 125 at 48 rows: 8 square against 8 packed, 10 % faster)
        const int st = t.nwv * t.blocksPerCU * (t.packed ? 4 : 7), sf = full.nwv * full.blocksPerCU * (full.packed ? 4 : 7);
        // (n-chain form: only for a higher staging level -- its iterations read M' from L2 when it is not staged, and more
        // resident wavefronts then cost as much as they bring: N = 75 at 32 rows was 8 % slower, N = 50 with M' staged
        // again 1.8x faster)
        if (!((h->waveGram && st > sf) || t.level > full.level)) c1 = 0;
    }
    return c1;
}

// lmpc_reserve: everything a wavefront-kernel launch on a batch of nprob problems would allocate lazily (ticket and
// overflow counters, working-set statistics, the overflow lists of both passes, the slow path's scratch), allocated
// now -- the launch paths then find it in place.  Branch-and-bound snapshots depend on the launch shape and stay lazy.
inline int wave_reserve_impl(lmpc_handle *h, int64_t nprob, hipStream_t st) {
    constexpr int kP1 = kShards * kCountStride;
    if (!h->dQueue) {
        if (!h->dOvfCount) HIP_TRY(h, hipMalloc(&h->dOvfCount, sizeof(int32_t) * (2 * kP1 + 64)));
        HIP_TRY(h, hipMemsetAsync(h->dOvfCount, 0, sizeof(int32_t) * (2 * kP1 + 64), st));
        HIP_TRY(h, hipMalloc(&h->dQueue, 64));
        HIP_TRY(h, hipMemsetAsync(h->dQueue, 0, 64, st));
        h->waveCtrSet = 0; h->waveOvfSet = 0;
    }
    // (the first-pass list: a first pass may run on any handle since the row kernel takes batches of every size -- plain
    // solves as the first of two passes whenever the slow path lies behind its capacity, searches at 48 rows)
    if (h->waveTwoPass != 0 && nprob < (int64_t)0x7fffffff && nprob > h->ovfCap1) {
        hipFree(h->dOvfList1); h->dOvfList1 = nullptr; h->ovfCap1 = 0;
        HIP_TRY(h, hipMalloc(&h->dOvfList1, sizeof(int32_t) * (size_t)nprob));
        h->ovfCap1 = nprob;
    }
    if (h->bnb || nprob >= (int64_t)0x7fffffff) return LMPC_OK;
    if (!h->hStat) {
        unsigned long long *hp = nullptr;
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void **>(&hp), 64, hipHostMallocMapped));
        for (int q = 0; q < 8; q++) hp[q] = 0ull;
        h->hStat = hp;
        HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void **>(&h->dStatHost), hp, 0));
        HIP_TRY(h, hipMalloc(&h->dStat, sizeof(unsigned long long) * 64 * 16));
        HIP_TRY(h, hipMemsetAsync(h->dStat, 0, sizeof(unsigned long long) * 64 * 16, st));
    }
    if (h->bigPath && h->capFull > h->W.cap) {                                            // the slow path may run
        if (nprob > h->ovfCap) {
            hipFree(h->dOvfList); h->dOvfList = nullptr; h->ovfCap = 0;
            HIP_TRY(h, hipMalloc(&h->dOvfList, sizeof(int32_t) * (size_t)nprob));
            h->ovfCap = nprob;
        }
        if (!h->dBigR) {
            const int bigCap = h->capFull < kBigCap ? h->capFull : kBigCap;
            HIP_TRY(h, hipMalloc(&h->dBigR, sizeof(double) * (size_t)kBigThreads * (size_t)big_scratch_reals(h->W.n, h->W.m, bigCap)));
            HIP_TRY(h, hipMalloc(&h->dBigI, sizeof(int32_t) * (size_t)kBigThreads * (size_t)big_scratch_ints(h->W.m, bigCap)));
        }
    }
    return LMPC_OK;
}

// Branch and bound in two passes (round 4): the search's working sets hold the fixed binaries plus a few more rows -- far
// fewer than the n + 1 rows the factor is sized for (satellite: 40 binaries, capacity 61) -- and what bounds the search
// is issue latency at the residency the factor's LDS allows.  So the batch first runs at the smallest of 24 / 32 / 48
// rows that leaves eight rows beyond the binaries, IF that keeps more wavefronts resident; a point that outgrows it
// (exit flag -7 inside the pass) is listed and searched again from scratch at the full capacity.  The search is
// deterministic, so the result does not depend on the split.  "wave_two_pass" 0 switches it off.
inline int bnb_first_pass_cap(lmpc_handle *h, int64_t nprob, size_t rs) {
    if (!h->bnb || h->waveTwoPass == 0 || nprob >= (int64_t)0x7fffffff || nprob < 4096) return 0;
    int c1 = 0;
    for (int c : {24, 32, 48})
        if (c1 == 0 && c >= h->nBinary + 8) c1 = c;
    if (h->waveTwoPass > 0 && h->waveCap1 > 0) c1 = h->waveCap1;
    if (c1 == 0 || c1 + 4 > h->W.cap) return 0;
    const WaveConfig full = wave_config(h, rs);
    const int capW = h->W.cap, ldcW = h->W.ldc;
    h->W.cap = c1; h->W.ldc = c1 | 1;
    const WaveConfig t = wave_config(h, rs);
    h->W.cap = capW; h->W.ldc = ldcW;
    return t.nwv * t.blocksPerCU > full.nwv * full.blocksPerCU ? c1 : 0;
}

template <typename R, int MR, int LDSC, bool BNB, bool PACKED, int NU = 1, bool GRAM = false, bool SIM = true>
int launch_wave_cfg(lmpc_handle *h, const WaveConfig &cfg, const R *dC, int64_t nprob, const R *theta, R *x,
                    int32_t *flag, int32_t *iters, uint64_t *active, const uint64_t *warm, hipStream_t st) {
    const WaveLayout &Wl = h->W;                         // (work-list mode, screening pass in front: h->waveList, see launch())
    // (SIM false: the instantiation without the closed-loop machinery -- the caller chose it for a plain batched solve)
    auto kern = wave_kernel<R, MR, LDSC, BNB, PACKED, NU, GRAM, SIM>;
    if (cfg.lds > 48 * 1024)
        HIP_TRY(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfg.lds));
    if (h->preloadOnly) {                              // setup: the translation unit's code object on the device, no launch
        hipFuncAttributes fa;
        HIP_TRY(h, hipFuncGetAttributes(&fa, (const void *)kern));
        return LMPC_OK;
    }
    int blocksPerCU = cfg.blocksPerCU;
    if (h->waveCap > 0) {                              // tuning: wavefronts per CU of the persistent grid
        blocksPerCU = h->waveCap / cfg.nwv;
        if (blocksPerCU < 1) blocksPerCU = 1;
    }
    long long grid = (long long)h->numCU * blocksPerCU;
    const long long need = (nprob + cfg.nwv - 1) / cfg.nwv;
    if (grid > need) grid = need;
    if constexpr (BNB) {
        // branch and bound keeps one snapshot slot per search depth and resident wavefront (see below): at most half of
        // what the device has free -- a smaller grid (fewer wavefronts in flight) before an allocation that crowds out
        // the caller or fails (the slots already held count as free: they are replaced)
        const size_t perWave = (size_t)(h->nBinary > 0 ? h->nBinary : 1) *
                               (sizeof(double) * (6 * 64 + (size_t)Wl.cap * (Wl.cap - 1) / 2) + sizeof(int32_t) * 5 * 64);
        size_t freeB = 0, totalB = 0;
        if (perWave * (size_t)grid * cfg.nwv > h->bnbBytesR + h->bnbBytesI) {
            if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { freeB = 0; (void)hipGetLastError(); }
            const size_t budget = (freeB + h->bnbBytesR + h->bnbBytesI) / 2;
            while (grid > 1 && perWave * (size_t)grid * cfg.nwv > budget) grid = (grid + 1) / 2;
            if (perWave * (size_t)grid * cfg.nwv > budget)
                return fail(h, LMPC_ERR_HIP, "lmpc: not enough free device memory for the branch-and-bound snapshots");
        }
    }
    // The ticket counter and the overflow counters exist twice and are used alternately: a kernel clears the ones of the
    // launch / call after it (nobody touches those while it runs; calls on a handle are stream-ordered), so no launch
    // needs a memset in front of it -- two 5 us fill kernels per launch, 4 % of a closed-loop step.  Tickets alternate per
    // LAUNCH; the overflow counters per CALL (h->waveOvfSet, toggled by launch_wave_inst), because the second of two
    // passes reads the first one's as the length of its work list.  Layout of dOvfCount (ints): two first-pass sets
    // shaped like a work list's counters (kShards words kCountStride apart, only word 0 ever non-zero), then the two
    // words of the last pass's overflow (the slow path's list).
    constexpr int kP1 = kShards * kCountStride;
    if (!h->dQueue) {
        if (!h->dOvfCount) HIP_TRY(h, hipMalloc(&h->dOvfCount, sizeof(int32_t) * (2 * kP1 + 64)));
        HIP_TRY(h, hipMemsetAsync(h->dOvfCount, 0, sizeof(int32_t) * (2 * kP1 + 64), st));
        HIP_TRY(h, hipMalloc(&h->dQueue, 64));
        HIP_TRY(h, hipMemsetAsync(h->dQueue, 0, 64, st));
        h->waveCtrSet = 0; h->waveOvfSet = 0;
    }
    const int pass = h->wavePass;                        // 0: the only pass; 1: first of two (smaller capacity); 2: second
    const int os = h->waveOvfSet;
    int32_t *const p1Count = h->dOvfCount + os * kP1, *const p1Next = h->dOvfCount + (os ^ 1) * kP1;
    int32_t *const ovfCount = h->dOvfCount + 2 * kP1 + 8 * os, *const p2Next = h->dOvfCount + 2 * kP1 + 8 * (os ^ 1);
    int32_t *const queueNext = h->dQueue + 8 * (h->waveCtrSet ^ 1);

    WaveList wl2{};
    if (pass == 2) {                                     // the first pass's overflow list IS this pass's work list
        wl2.list = h->dOvfList1; wl2.count = p1Count; wl2.count_next = nullptr; wl2.seg_cap = (long long)nprob;
    }
    const WaveList &wl = pass == 2 ? wl2 : h->waveList;
    // more than two problems per resident wavefront: hand them out through the shared counter
    int32_t *queue = nullptr;
    int qchunk = 1;
    if (wl.list != nullptr && h->waveQueue) {
        // work-list mode: the length of the list is only known on the device -- one problem per ticket
        queue = h->dQueue + 8 * h->waveCtrSet;
    } else if (nprob > 2 * grid * cfg.nwv && nprob < (int64_t)0x7fffffff && h->waveQueue) {
        // ~16 tickets per resident wavefront over the whole batch, at most 64 problems per ticket
        qchunk = (int)(nprob / (16 * grid * cfg.nwv));
        qchunk = qchunk < 1 ? 1 : (qchunk > 64 ? 64 : qchunk);
        queue = h->dQueue + 8 * h->waveCtrSet;
    }
    // working sets that can outgrow the 64 lanes (n + 1 + #soft > 64): such points are listed by the kernel and
    // re-solved behind it, one problem per thread (no branch and bound there)
    const bool big = pass != 1 && !BNB && h->bigPath && h->capFull > Wl.cap && nprob < (int64_t)0x7fffffff;
    const int bigCap = h->capFull < kBigCap ? h->capFull : kBigCap;
    if (pass == 1 && nprob > h->ovfCap1) {
        hipFree(h->dOvfList1); h->dOvfList1 = nullptr; h->ovfCap1 = 0;
        HIP_TRY(h, hipMalloc(&h->dOvfList1, sizeof(int32_t) * (size_t)nprob));
        h->ovfCap1 = nprob;
    }
    if (!BNB && !h->hStat) {                             // working-set statistics: device counters + their mapped host copy
        unsigned long long *hp = nullptr;
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void **>(&hp), 64, hipHostMallocMapped));
        for (int q = 0; q < 8; q++) hp[q] = 0ull;
        h->hStat = hp;
        HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void **>(&h->dStatHost), hp, 0));
        HIP_TRY(h, hipMalloc(&h->dStat, sizeof(unsigned long long) * 64 * 16));
        HIP_TRY(h, hipMemsetAsync(h->dStat, 0, sizeof(unsigned long long) * 64 * 16, st));
    }
    if (big) {
        if (nprob > h->ovfCap) {
            hipFree(h->dOvfList); h->dOvfList = nullptr; h->ovfCap = 0;
            HIP_TRY(h, hipMalloc(&h->dOvfList, sizeof(int32_t) * (size_t)nprob));
            h->ovfCap = nprob;
        }
        if (!h->dBigR) {     // sized for binary64: the binary32 calls of the handle use the same slices
            HIP_TRY(h, hipMalloc(&h->dBigR, sizeof(double) * (size_t)kBigThreads * (size_t)big_scratch_reals(Wl.n, Wl.m, bigCap)));
            HIP_TRY(h, hipMalloc(&h->dBigI, sizeof(int32_t) * (size_t)kBigThreads * (size_t)big_scratch_ints(Wl.m, bigCap)));
        }
    }
    // branch and bound: room for one snapshot of a node's state per search depth and resident wavefront (sized for
    // binary64; the binary32 calls of the handle use the same slices)
    R *bnbR = nullptr;
    int32_t *bnbI = nullptr;
    if constexpr (BNB) {
        const size_t snapR = 6 * 64 + (size_t)Wl.cap * (Wl.cap - 1) / 2, snapI = 5 * 64;
        const size_t slots = (size_t)grid * cfg.nwv, depth = (size_t)(h->nBinary > 0 ? h->nBinary : 1);
        const size_t needR = sizeof(double) * slots * depth * snapR, needI = sizeof(int32_t) * slots * depth * snapI;
        if (needR > h->bnbBytesR) {
            hipFree(h->dBnbR); h->dBnbR = nullptr; h->bnbBytesR = 0;
            HIP_TRY(h, hipMalloc(&h->dBnbR, needR));
            h->bnbBytesR = needR;
        }
        if (needI > h->bnbBytesI) {
            hipFree(h->dBnbI); h->dBnbI = nullptr; h->bnbBytesI = 0;
            HIP_TRY(h, hipMalloc(&h->dBnbI, needI));
            h->bnbBytesI = needI;
        }
        bnbR = static_cast<R *>(h->dBnbR);
        bnbI = h->dBnbI;
    } else if (h->keepOn && sizeof(R) == 8) {
        // closed loop (lmpc_simulate_device, step-synchronous): every scenario's final working set and factor stay on
        // the device between two steps, indexed by scenario
        bnbR = static_cast<R *>(h->dKeepR);
        bnbI = h->dKeepI;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * cfg.nwv), cfg.lds, st, Wl, dC, h->dSw, theta, x, flag,
                       iters, active, warm, queue, qchunk, (long long)nprob, wl.list, wl.count, wl.count_next, wl.seg_cap,
                       pass == 1 ? h->dOvfList1 : (big ? h->dOvfList : nullptr), pass == 1 ? p1Count : (big ? ovfCount : nullptr),
                       h->waveSim, bnbR, bnbI, BNB ? h->nBinary : (pass == 1 ? 1 : 0), queueNext, p2Next, p1Next,
                       BNB ? nullptr : h->dStat, BNB ? nullptr : h->dStatHost);
    h->waveCtrSet ^= 1;
    HIP_TRY(h, hipGetLastError());
    if (big) {
        hipLaunchKernelGGL(big_kernel<R>, dim3(kBigThreads / 64), dim3(64), 0, st, Wl, dC, h->dSw, theta, x, flag, iters,
                           active, warm, h->dOvfList, ovfCount, static_cast<R *>(h->dBigR), h->dBigI, bigCap, h->waveSim);
        HIP_TRY(h, hipGetLastError());
    }
    return LMPC_OK;
}

template <typename R, bool BNB, bool GRAM, bool SIM>
int launch_wave_inst(lmpc_handle *h, const R *dC, int64_t nprob, const R *theta, R *x, int32_t *flag,
                     int32_t *iters, uint64_t *active, const uint64_t *warm, hipStream_t st) {
    EventTriple ev{};
    const bool prof = h->prof && h->waveList.list == nullptr && !h->preloadOnly;     // (behind the screening pass: launch_wave_screened's events)
    if (prof) {
        HIP_TRY(h, pool_event(h, &ev.a));
        HIP_TRY(h, pool_event(h, &ev.mid));
        HIP_TRY(h, pool_event(h, &ev.b));
        HIP_TRY(h, hipEventRecord(ev.a, st));
        HIP_TRY(h, hipEventRecord(ev.mid, st));
    }
    int rc;
    const int mr = wave_slots(h->P.m, BNB);
    if (BNB) warm = nullptr;                              // a B&B node takes its start from the search, not the caller
    // Two passes (no binaries): the factor of a 64-row working set takes 16-33 KB of LDS per wavefront, which is what
    // limits residency (and the staging level of the n-chain form) for every problem with n + 1 + #soft > 40 -- while
    // the working sets of an MPC problem in normal operation stay far below that.  So the batch first runs at a SMALLER
    // capacity (24, 32 or 48 rows: square factor, 12 wavefronts per CU instead of 9, M' staged again where it had not
    // fit); a point that outgrows it is listed and started again -- same warm start, same kept state -- by a second
    // launch at the full capacity, whose own overflow goes to the slow path as before.  Results do not depend on the
    // split (a solve's arithmetic does not depend on the capacity).  Whether, and at which capacity, is decided from
    // what the handle has seen: every wavefront-kernel launch counts how large the working sets of the problems it
    // finished became (cumulative device counters, copied by each launch into mapped host memory and read here without
    // a copy or a synchronisation) -- the smallest of 24 / 32 / 48 rows that would have held 97 % of them, if the
    // launch configuration at that capacity is the better one (more wavefronts per CU or a higher staging level); one
    // pass until 1000 problems have been seen, for small batches, and if no capacity qualifies.
    // "wave_two_pass" 0 = never, 1 = always at "wave_cap1" rows (default 24), -1 (default) = as described.
    WaveConfig cfg = wave_config(h, sizeof(R));
    const int c1 = BNB ? bnb_first_pass_cap(h, nprob, sizeof(R)) : wave_first_pass_cap_impl(h, nprob, sizeof(R));
    const int capW = h->W.cap, ldcW = h->W.ldc;
    auto dispatch = [&]() -> int {
    int rc;
    // long horizons (64 <= n <= 127): two variable slots per lane; built without LDS staging of the problem data
    // (M alone is up to 1 MB there) and without branch and bound
    if (h->P.n > 64) {
        if constexpr (BNB) {
            rc = fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: branch and bound covers n <= 64");
        } else {
#define LMPC_WVU(MRR) (cfg.packed ? launch_wave_cfg<R, MRR, 0, false, true, 2, GRAM, SIM>(h, cfg, dC, nprob, theta, x, flag, iters, active, warm, st) \
                                  : launch_wave_cfg<R, MRR, 0, false, false, 2, GRAM, SIM>(h, cfg, dC, nprob, theta, x, flag, iters, active, warm, st))
            if (mr <= 2) rc = LMPC_WVU(2);
            else if (mr == 3) rc = LMPC_WVU(3);
            else if (mr == 4) rc = LMPC_WVU(4);
            else if (mr == 5) rc = LMPC_WVU(5);
            else if (mr == 6) rc = LMPC_WVU(6);
            else if (mr <= 8) rc = LMPC_WVU(8);
            else rc = LMPC_WVU(16);
#undef LMPC_WVU
        }
        return rc;
    }
#define LMPC_WV4(MRR, LV, PK) launch_wave_cfg<R, MRR, LV, BNB, PK, 1, GRAM, SIM>(h, cfg, dC, nprob, theta, x, flag, iters, active, warm, st)
#define LMPC_WV3(MRR, LV) LMPC_WV4(MRR, LV, false)
#define LMPC_WV(MRR) (cfg.packed ? (cfg.level >= 1 ? LMPC_WV4(MRR, 1, true) : LMPC_WV4(MRR, 0, true)) \
                                 : (cfg.level >= 3 ? LMPC_WV3(MRR, 3) : (cfg.level == 2 ? LMPC_WV3(MRR, 2) : (cfg.level == 1 ? LMPC_WV3(MRR, 1) : LMPC_WV3(MRR, 0)))))
#define LMPC_WVB(MRR) (cfg.packed ? (cfg.level >= 1 ? LMPC_WV4(MRR, 1, true) : LMPC_WV4(MRR, 0, true)) \
                                  : (cfg.level >= 1 ? LMPC_WV3(MRR, 1) : LMPC_WV3(MRR, 0)))
    if constexpr (GRAM) {                                 // the Gram-scan form is built for staging levels 0 and 1
        if (mr <= 1) rc = LMPC_WVB(1);
        else if (mr == 2) rc = LMPC_WVB(2);
        else if (mr == 4) rc = LMPC_WVB(4);
        else if (mr == 8) rc = LMPC_WVB(8);
        else if (mr == 16) rc = LMPC_WVB(16);
        else if constexpr (!BNB) {
            if (mr == 3) rc = LMPC_WVB(3);
            else if (mr == 5) rc = LMPC_WVB(5);
            else rc = LMPC_WVB(6);
        } else rc = fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: no wavefront-kernel instantiation");
    } else
    if (mr <= 1) rc = LMPC_WV(1);
    else if (mr == 2) rc = LMPC_WV(2);
    else if (mr == 4) rc = LMPC_WV(4);
    else if (mr == 8) rc = LMPC_WVB(8);
    else if (mr == 16) rc = LMPC_WVB(16);
    else if constexpr (!BNB) {                            // 3, 5, 6 slots (never chosen for branch and bound)
        if (mr == 3) rc = LMPC_WVB(3);
        else if (mr == 5) rc = LMPC_WVB(5);
        else rc = LMPC_WVB(6);
    } else rc = fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: no wavefront-kernel instantiation");
#undef LMPC_WVB
#undef LMPC_WV
#undef LMPC_WV3
#undef LMPC_WV4
    return rc;
    };   // dispatch
    if (h->preloadOnly) {
        if constexpr (!BNB && !GRAM) {                        // (the row kernel's code object too, where a large batch would take it)
            const int rcap = row_pass_cap(h, (int64_t)1 << 20, sizeof(R), false, GRAM, BNB);
            if (rcap > 0) (void)launch_row<R>(h, dC, nprob, theta, x, flag, iters, active, st, rcap, 0);
        }
        if constexpr (BNB && !GRAM) {
            const int rcap = row_bnb_pass_cap(h, (int64_t)1 << 20, sizeof(R));
            if (rcap > 0) (void)launch_row_bnb<R>(h, dC, nprob, theta, x, flag, iters, active, st, rcap, 0);
        }
        return dispatch();
    }
    // Four problems per wavefront (lmpc_row_kernel.hpp) where it applies: as the only pass when its capacity holds every
    // working set of the problem, else as the first of two passes in place of this kernel's own first pass.
    if constexpr (!BNB && !GRAM) {
        const int rcap = row_pass_cap(h, nprob, sizeof(R), warm != nullptr, GRAM, BNB);
        if (rcap > 0) {
            // (as the only pass where nothing lies behind its capacity; else what outgrows it is listed: for this kernel's
            // second pass and, behind that, the slow path)
            if (rcap >= capW && !(h->bigPath && h->capFull > capW)) {
                rc = launch_row<R>(h, dC, nprob, theta, x, flag, iters, active, st, rcap, 0);
            } else {
                h->wavePass = 1;
                rc = launch_row<R>(h, dC, nprob, theta, x, flag, iters, active, st, rcap, 1);
                h->wavePass = 2;
                // (what the row kernel lists is a handful of long solves: the second pass is a latency chain, so it takes the
                // highest staging level that fits with small workgroups instead of the most wavefronts per CU --
                // pendulum_N50: 0.22 -> 0.12 ms for the 327 of 2e5 points beyond 16 rows)
                cfg = wave_config(h, sizeof(R));
                if (h->waveLevel < 0 && h->waveNwv <= 0) {
                    for (int lv = 3; lv >= 1; lv--) {
                        h->waveLevel = lv; h->waveNwv = 2;
                        const WaveConfig c2 = wave_config(h, sizeof(R));
                        if (c2.level >= 1) { cfg = c2; break; }
                    }
                    h->waveLevel = -1; h->waveNwv = 0;
                }
                if (rc == LMPC_OK) rc = dispatch();
                h->wavePass = 0;
            }
            h->waveOvfSet ^= 1;
            if (prof) {
                if (rc == LMPC_OK) { HIP_TRY(h, hipEventRecord(ev.b, st)); h->events.push_back(ev); }
                else { hipEventDestroy(ev.a); hipEventDestroy(ev.mid); hipEventDestroy(ev.b); }
            }
            return rc;
        }
    }
    if constexpr (BNB && !GRAM) {           // ... four searches per wavefront
        const int rcap = row_bnb_pass_cap(h, nprob, sizeof(R));
        if (rcap > 0) {
            if (rcap >= capW) {
                rc = launch_row_bnb<R>(h, dC, nprob, theta, x, flag, iters, active, st, rcap, 0);
            } else {
                h->wavePass = 1;
                rc = launch_row_bnb<R>(h, dC, nprob, theta, x, flag, iters, active, st, rcap, 1);
                h->wavePass = 2;
                cfg = wave_config(h, sizeof(R));
                if (rc == LMPC_OK) rc = dispatch();
                h->wavePass = 0;
            }
            h->waveOvfSet ^= 1;
            if (prof) {
                if (rc == LMPC_OK) { HIP_TRY(h, hipEventRecord(ev.b, st)); h->events.push_back(ev); }
                else { hipEventDestroy(ev.a); hipEventDestroy(ev.mid); hipEventDestroy(ev.b); }
            }
            return rc;
        }
    }
    if (c1 > 0) {
        h->W.cap = c1; h->W.ldc = c1 | 1; h->wavePass = 1;
        cfg = wave_config(h, sizeof(R));
        rc = dispatch();
        h->W.cap = capW; h->W.ldc = ldcW; h->wavePass = 2;
        cfg = wave_config(h, sizeof(R));
        if (rc == LMPC_OK) rc = dispatch();
        h->wavePass = 0;
    } else {
        rc = dispatch();
    }
    h->waveOvfSet ^= 1;
    if (prof) {
        if (rc == LMPC_OK) { HIP_TRY(h, hipEventRecord(ev.b, st)); h->events.push_back(ev); }
        else { hipEventDestroy(ev.a); hipEventDestroy(ev.mid); hipEventDestroy(ev.b); }
    }
    return rc;
}

}  // namespace lmpc
