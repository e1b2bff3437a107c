// C ABI of liblmpc_hip.so (see include/lmpc_hip.h) and kernel dispatch.
// There is deliberately no CPU fallback: without a HIP device every solve entry point fails
// with LMPC_ERR_NOGPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lmpc_internal.hpp"
#include "lmpc_lane_kernel.hpp"
#include "lmpc_screen_kernel.hpp"
#include "lmpc_simrun_kernel.hpp"

using namespace lmpc;

thread_local std::string g_setup_err;      // text of a failed setup call (no handle to carry it)

namespace lmpc {

hipError_t pool_event(lmpc_handle *h, hipEvent_t *e) {
    if (!h->eventPool.empty()) { *e = h->eventPool.back(); h->eventPool.pop_back(); return hipSuccess; }
    // timing events only: without the system-scope fence a default event performs when it completes (the fence is
    // for host visibility of the kernel's writes, which these events are never used to order)
    return hipEventCreateWithFlags(e, hipEventDisableSystemFence);
}

int fail(lmpc_handle *h, int code, const std::string &msg) {
    if (h) { std::lock_guard<std::mutex> lk(h->errMu); h->err = msg; }
    else g_setup_err = msg;
    return code;
}

}  // namespace lmpc

namespace {

void fill_layout(lmpc_handle *h) {
    const HostPack &P = h->P;
    PackLayout &L = h->L;
    const int N = h->laneN;
    L.n = P.n; L.m = P.m; L.ms = P.ms; L.nth = P.nth; L.nout = P.nout; L.words = P.words();
    int o = 0;
    L.oM = o; o += P.m * N;
    L.oG = o; o += lmpc_tri(P.m);
    L.odu = o; o += P.m;
    L.odl = o; o += P.m;
    L.oDth = o; o += P.m * P.nth;
    L.oRout = o; o += P.nout * N;
    L.ox0 = o; o += P.nout;
    L.oXth = o; o += P.nout * P.nth;
    L.nthp = P.nth <= 8 ? 8 : (P.nth <= 16 ? 16 : 32);
    o = (o + 7) & ~7;                                  // 64-byte aligned rows for wide scalar loads
    const int mp = (P.m + 3) & ~3;                     // rows padded to a multiple of four
    L.oDthP = o; o += mp * L.nthp;
    L.oBnd = o; o += 2 * mp;
    o = (o + 7) & ~7;
    L.oXthP = o; o += P.nout * L.nthp;
    o = (o + 7) & ~7;
    L.oFG = o; o += 32 * 32 + 32 * kMaxSimU;           // closed loop: F (nx <= 32) and G (nu <= kMaxSimU)
    h->nC = (size_t)o;
    L.imm_mask = 0; L.eq_mask = 0;
    for (int j = 0; j < P.m && j < 64; j++) {
        if (P.sense[j] & SENSE_IMMUTABLE) L.imm_mask |= 1ull << j;
        if (P.sense[j] & SENSE_ACTIVE) L.eq_mask |= 1ull << j;
    }
    const lmpc_settings &S = h->S;
    L.primal_tol = S.primal_tol; L.dual_tol = S.dual_tol; L.zero_tol = S.zero_tol;
    L.progress_tol = S.progress_tol; L.fval_bound = S.fval_bound; L.rho_soft = S.rho_soft;
    L.cycle_tol = S.cycle_tol; L.iter_limit = S.iter_limit;
    WaveLayout &Wl = h->W;
    Wl.primal_tol = S.primal_tol; Wl.dual_tol = S.dual_tol; Wl.zero_tol = S.zero_tol;
    Wl.progress_tol = S.progress_tol; Wl.fval_bound = S.fval_bound; Wl.rho_soft = S.rho_soft;
    Wl.cycle_tol = S.cycle_tol; Wl.iter_limit = S.iter_limit;
    if (h->avi) avi_fill_settings(h);
}

// choose the kernel variant and upload the constant pack
int finalize_handle(lmpc_handle *h) {
    const HostPack &P = h->P;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(h, LMPC_ERR_NOGPU, "lmpc: no HIP device available (this library has no CPU path)");
    if (h->device < 0 || h->device >= ndev) return fail(h, LMPC_ERR_BADARG, "lmpc: bad device ordinal");
    int nBinary = 0;
    for (int j = 0; j < P.m; j++) nBinary += (P.sense[j] & SENSE_BINARY) ? 1 : 0;
    const bool anyBinary = nBinary > 0;
    h->bnb = anyBinary;
    h->nBinary = nBinary;
    if (nBinary > 64)    // the B&B stack of a problem lives on the 64 lanes of its wavefront
        return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: more than 64 binary rows");
    if (anyBinary && P.n > 64)
        return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: branch and bound covers n <= 64 variables");
    const bool laneOk = P.n <= kLaneMaxN && P.m <= kLaneMaxM && P.nsoft == 0 && !anyBinary;
    // working-set capacity: n hard rows + 1 (the row that makes it singular) + the soft rows, but never
    // more than the 64 lanes; a problem that wants more rows at once ends with exit flag -7
    // (branch and bound: one row more -- the row a node has just fixed enters on top of its parent's final working set,
    // whatever that holds; the oracle's arrays have that row too)
    const int cap = std::min(P.n + 1 + P.nsoft + (anyBinary ? 1 : 0), kWaveMaxCap);
    // ... and what a working set can reach in floating point: one row more (a singular working set that the pivot test
    // misses takes another row before it fails; the oracle's arrays have that spare row).  A plain solve that wants it
    // is listed for the slow path, like one that outgrows the 64 rows; a search has it in the kernels (no slow path there).
    h->capFull = P.n + 2 + P.nsoft;
    const bool waveOk = P.n <= kWaveMaxN && P.m <= kWaveMaxM && P.m >= 1;
    if (!laneOk && !waveOk)
        return fail(h, LMPC_ERR_UNSUPPORTED,
                    "lmpc: problem outside what the kernels cover (lane: n<=12, m<=64, hard rows; "
                    "wave: n<=127, 1<=m<=1024)");
    // Which kernel by default when both cover the problem: the lane kernels (one QP per lane, screening
    // pass in front) are 5-10x faster on box-constrained problems of every size they are built for, but a
    // lane scans all m rows of general constraints by itself every iteration -- from about m*n = 600 on
    // the wavefront kernel, which spreads the rows over its lanes, is ahead (tools/lane_vs_wave.py:
    // mass-spring n=10: m=46 lane 1.40e8 vs 1.26e8, m=63 lane 1.06e8 vs wave 1.45e8; n=8, m=49 lane
    // 1.69e8 vs 1.60e8).  lmpc_set_option("wave", 0 | 1) overrides.
    const bool manyGeneralRows = P.ms < P.m && (long long)P.m * P.n >= 600;
    h->useWave = !laneOk || (waveOk && manyGeneralRows);
    // ... and in front of the wavefront kernel, for exactly that class (small n, many rows, nothing but plain hard rows):
    // straight-line tiers with one problem per lane (lmpc_qp_tiers_kernel.hpp)
    h->qpTiersOk = false;
    if (waveOk && !anyBinary && P.n >= 2 && P.n <= 12 && P.m <= 64 && P.nth >= 1 && P.nth <= 16 &&
        qp_tiers_lds_bytes(P.n, P.m) <= (size_t)160 * 1024) {
        bool plain = true;                                 // (hard or SOFT rows, no other flag)
        for (int j = 0; j < P.m; j++) plain = plain && (P.sense[j] & ~SENSE_SOFT) == 0;
        h->qpTiersOk = plain;
    }
    LMPC_ENTER_DEVICE(h);                              // the caller's current device comes back when setup returns
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0)
            h->numCU = prop.multiProcessorCount;
    }
    if (waveOk) {
        // constant pack of the wavefront kernel: M, M transposed, Gram, bounds, maps (stride n)
        WaveLayout &Wl = h->W;
        Wl.n = P.n; Wl.m = P.m; Wl.ms = P.ms; Wl.nth = P.nth; Wl.nout = P.nout; Wl.words = P.words();
        Wl.cap = cap; Wl.ldc = cap | 1;
        Wl.keepStride = 2 * 64 + cap * (cap - 1) / 2;
        int o = 0;
        Wl.oM = o; o += P.m * P.n;
        Wl.oMt = o; o += P.m * P.n;
        Wl.oG = o; o += lmpc_tri(P.m);
        Wl.odu = o; o += P.m;
        Wl.odl = o; o += P.m;
        Wl.oDth = o; o += P.m * P.nth;
        Wl.oRout = o; o += P.nout * P.n;
        Wl.ox0 = o; o += P.nout;
        Wl.oXth = o; o += P.nout * P.nth;
        Wl.oGf = o; o += P.m * P.m;                    // (last: the binary32 copy is laid out the same way)
        Wl.nC = o;
        std::vector<double> wb((size_t)o, 0.0);
        std::memcpy(&wb[Wl.oM], P.M.data(), sizeof(double) * P.M.size());
        for (int j = 0; j < P.m; j++)
            for (int k = 0; k < P.n; k++) wb[Wl.oMt + (size_t)k * P.m + j] = P.M[(size_t)j * P.n + k];
        std::memcpy(&wb[Wl.oG], P.G.data(), sizeof(double) * P.G.size());
        for (int a = 0; a < P.m; a++)
            for (int b = 0; b <= a; b++)
                wb[Wl.oGf + (size_t)a * P.m + b] = wb[Wl.oGf + (size_t)b * P.m + a] = P.G[(size_t)lmpc_tri(a) + b];
        std::memcpy(&wb[Wl.odu], P.du0.data(), sizeof(double) * P.m);
        std::memcpy(&wb[Wl.odl], P.dl0.data(), sizeof(double) * P.m);
        if (P.m * P.nth) std::memcpy(&wb[Wl.oDth], P.Dth.data(), sizeof(double) * P.Dth.size());
        std::memcpy(&wb[Wl.oRout], P.Rout.data(), sizeof(double) * P.Rout.size());
        std::memcpy(&wb[Wl.ox0], P.x0.data(), sizeof(double) * P.x0.size());
        if (P.nout * P.nth) std::memcpy(&wb[Wl.oXth], P.Xth.data(), sizeof(double) * P.Xth.size());
        HIP_TRY(h, hipMalloc(&h->dCw, sizeof(double) * wb.size()));
        HIP_TRY(h, hipMemcpy(h->dCw, wb.data(), sizeof(double) * wb.size(), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMalloc(&h->dSw, sizeof(int32_t) * P.m));
        HIP_TRY(h, hipMemcpy(h->dSw, P.sense.data(), sizeof(int32_t) * P.m, hipMemcpyHostToDevice));
    }
    fill_layout(h);
    if (!laneOk) {
        h->kname = "wave";
        // The screening pass (lmpc_screen_kernel.hpp) also runs in front of the wavefront kernel: it needs the
        // padded rows of Dth, the bounds and the output map, nothing else of the lane pack.  Rows that can never
        // enter the working set get bounds that are never violated (the pass's own mask covers 64 rows).
        bool initActive = false;
        for (int j = 0; j < P.m; j++) initActive = initActive || (P.sense[j] & SENSE_ACTIVE);
        if (anyBinary || initActive || P.nth < 1 || P.nth > 32) return LMPC_OK;
        h->laneN = 0;
        fill_layout(h);
        const PackLayout &L = h->L;
        std::vector<double> buf(h->nC, 0.0);
        const int mp = (P.m + 3) & ~3;
        for (int j = 0; j < mp; j++) {
            const bool real = j < P.m && !(P.sense[j] & SENSE_IMMUTABLE);
            if (real)
                for (int t = 0; t < P.nth; t++) buf[L.oDthP + (size_t)j * L.nthp + t] = P.Dth[(size_t)j * P.nth + t];
            buf[L.oBnd + 2 * j] = real ? P.du0[j] : 1e300;
            buf[L.oBnd + 2 * j + 1] = real ? P.dl0[j] : -1e300;
        }
        for (int k = 0; k < P.nout; k++) {
            buf[L.ox0 + k] = P.x0[k];
            for (int t = 0; t < P.nth; t++) buf[L.oXthP + (size_t)k * L.nthp + t] = P.Xth[(size_t)k * P.nth + t];
        }
        HIP_TRY(h, hipMalloc(&h->dC, sizeof(double) * h->nC));
        HIP_TRY(h, hipMemcpy(h->dC, buf.data(), sizeof(double) * h->nC, hipMemcpyHostToDevice));
        h->screenPackOnly = true;
        return LMPC_OK;
    }
    h->laneN = 0;
    for (int s : kLaneSizes) if (s >= P.n) { h->laneN = s; break; }
    h->kname = "screen+lane<" + std::to_string(h->laneN) + ">";       // lmpc_kernel_name says "wave" while useWave is set
    fill_layout(h);
    const int N = h->laneN;
    std::vector<double> buf(h->nC, 0.0);
    const PackLayout &L = h->L;
    for (int j = 0; j < P.m; j++)
        for (int k = 0; k < P.n; k++) buf[L.oM + j * N + k] = P.M[(size_t)j * P.n + k];
    std::memcpy(&buf[L.oG], P.G.data(), sizeof(double) * P.G.size());
    for (int j = 0; j < P.m; j++) { buf[L.odu + j] = P.du0[j]; buf[L.odl + j] = P.dl0[j]; }
    if (P.m * P.nth) std::memcpy(&buf[L.oDth], P.Dth.data(), sizeof(double) * P.Dth.size());
    for (int k = 0; k < P.nout; k++)
        for (int c = 0; c < P.n; c++) buf[L.oRout + k * N + c] = P.Rout[(size_t)k * P.n + c];
    for (int k = 0; k < P.nout; k++) buf[L.ox0 + k] = P.x0[k];
    if (P.nout * P.nth) std::memcpy(&buf[L.oXth], P.Xth.data(), sizeof(double) * P.Xth.size());
    if (P.nth <= 32) {
        for (int j = 0; j < P.m; j++) {
            for (int t = 0; t < P.nth; t++) buf[L.oDthP + j * L.nthp + t] = P.Dth[(size_t)j * P.nth + t];
            buf[L.oBnd + 2 * j] = P.du0[j];
            buf[L.oBnd + 2 * j + 1] = P.dl0[j];
        }
        for (int j = P.m; j < ((P.m + 3) & ~3); j++) {  // padding rows: never violated
            buf[L.oBnd + 2 * j] = 1e300;
            buf[L.oBnd + 2 * j + 1] = -1e300;
        }
        for (int k = 0; k < P.nout; k++)
            for (int t = 0; t < P.nth; t++) buf[L.oXthP + k * L.nthp + t] = P.Xth[(size_t)k * P.nth + t];
    }
    HIP_TRY(h, hipMalloc(&h->dC, sizeof(double) * (h->nC ? h->nC : 1)));
    HIP_TRY(h, hipMemcpy(h->dC, buf.data(), sizeof(double) * h->nC, hipMemcpyHostToDevice));
    return LMPC_OK;
}

size_t lane_lds_bytes(const HostPack &P, int N, int B) {
    return sizeof(double) * ((size_t)P.m * N + lmpc_tri(P.m) + 2 * (size_t)P.m + (size_t)P.m * B);
}

// capacity of one work-list segment: the problems of all screening workgroups with the same
// (blockIdx % kShards), each covering kScreenTPB tiles of 256
long long lane_seg_cap(long long nprob) {
    // screening kernel: workgroup b (kScreenTPB tiles of 256 problems) -> shard b % kShards.  A shard's segment holds
    // an even share of the batch plus one workgroup's worth, rounded up (the one-launch kernel writes no list).
    static_assert(kScreenTPB * 256 <= 64 * 32 + 256, "a screening workgroup's problems must fit the segment's slack");
    return (((nprob + kShards - 1) / kShards + 64 * 32 + 256) + 255) & ~255ll;
}

template <int N, int MS, int MA, bool SIM, bool MULTI>
int launch_lane(lmpc_handle *h, int B, size_t lds, int64_t nprob, const double *theta, double *x,
                int32_t *flag, int32_t *iters, uint64_t *active, const uint64_t *warm,
                const int32_t *list, const int32_t *count, int32_t *count_next, hipStream_t st) {
    auto kern = lane_kernel<N, MS, MA, SIM, MULTI>;
    const long long segCap = lane_seg_cap(nprob);
    if (lds > 48 * 1024)
        HIP_TRY(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned grid = (unsigned)((nprob + B - 1) / B);
    // with a work list the counts are only known on the device: a fixed grid (a multiple of the
    // shard count) strides over each segment
    if (list) {
        // ... sized to ONE resident round (workgroups with nothing to do still cost a count load
        // before they can exit): what the register budget keeps on the chip, spread over the shards
        const unsigned wavesPerSimd = (N <= 5) ? (unsigned)LMPC_LANE_WAVES : 1u;
        const unsigned resident = (unsigned)h->numCU * (wavesPerSimd * 4u * 64u / (unsigned)B);
        unsigned per = (unsigned)((segCap + B - 1) / B);
        unsigned cap = resident / (unsigned)kShards;
        if (h->lanePer > 0) cap = (unsigned)h->lanePer;
        if (cap < 1u) cap = 1u;
        if (per > cap) per = cap;
        if (per < 1u) per = 1u;
        grid = per * (unsigned)kShards;
    }
    // first tier of the boxed instantiations: 0 none, 1 the generic loop at capacity 3, 2 the straight-line tiers
    // of lmpc_tiers.hpp (their preconditions on the row flags and the settings checked here)
    int tierArg = h->laneTier;
    if (tierArg && h->laneStraight && h->L.imm_mask == 0ull && h->L.eq_mask == 0ull &&
        h->S.iter_limit > LMPC_FAST_KMAX + 1 && h->S.cycle_tol >= LMPC_FAST_KMAX + 1)
        tierArg = 2;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(B), lds, st, h->L, h->dC, theta, x, flag, iters, active,
                       warm, list, count, count_next, segCap, kShards, (long long)nprob, tierArg);
    HIP_TRY(h, hipGetLastError());
    return LMPC_OK;
}

template <int NTHMAX, int NT, int MODE>
int launch_screen(lmpc_handle *h, int64_t nprob, const double *theta, double *x, int32_t *flag,
                  int32_t *iters, uint64_t *active, const uint64_t *warm, int32_t *count, hipStream_t st) {
    const int B = 256;
    // several outputs per problem: room for the wave-private transpose of the outputs (plain mode)
    const size_t lds = MODE == 3 ? sizeof(double) * B * (size_t)h->P.nout : 0;
    const long long ntiles = (nprob + B - 1) / B;
    const unsigned grid = (unsigned)((ntiles + kScreenTPB - 1) / kScreenTPB);
    const long long segCap = lane_seg_cap(nprob);
    hipLaunchKernelGGL((screen_kernel<NTHMAX, NT, MODE>), dim3(grid), dim3(B), lds, st, h->L, h->dC, theta, x, flag,
                       iters, active, warm, h->dList, count, segCap, kShards, (long long)nprob);
    HIP_TRY(h, hipGetLastError());
    return LMPC_OK;
}

template <typename R>
int launch_wave_t(lmpc_handle *h, const R *dC, int64_t nprob, const R *theta, R *x, int32_t *flag,
                  int32_t *iters, uint64_t *active, const uint64_t *warm, hipStream_t st) {
    if constexpr (sizeof(R) == 8) {
        // plain batched solve (no plant step to fuse, no kept factorisation, no run-ahead): the instantiations built
        // without the closed-loop machinery
        if (!h->bnb && h->waveSim.FG == nullptr && !h->keepOn)
            return h->waveGram ? launch_wave_inst<R, false, true, false>(h, dC, nprob, theta, x, flag, iters, active, warm, st)
                               : launch_wave_inst<R, false, false, false>(h, dC, nprob, theta, x, flag, iters, active, warm, st);
    }
    if (h->waveGram)
        return h->bnb ? launch_wave_inst<R, true, true>(h, dC, nprob, theta, x, flag, iters, active, warm, st)
                      : launch_wave_inst<R, false, true>(h, dC, nprob, theta, x, flag, iters, active, warm, st);
    return h->bnb ? launch_wave_inst<R, true, false>(h, dC, nprob, theta, x, flag, iters, active, warm, st)
                  : launch_wave_inst<R, false, false>(h, dC, nprob, theta, x, flag, iters, active, warm, st);
}

int launch_wave(lmpc_handle *h, int64_t nprob, const double *theta, double *x, int32_t *flag,
                int32_t *iters, uint64_t *active, const uint64_t *warm, hipStream_t st) {
    return launch_wave_t<double>(h, h->dCw, nprob, theta, x, flag, iters, active, warm, st);
}

// binary32 constant pack of the wave kernel (same layout as the binary64 one): every array of the
// f64 pack rounded to nearest, the Gram matrix re-accumulated IN binary32 from the rounded rows with
// the same fma chain, so that a lookup stays bit-identical to recomputing the product in binary32.
int ensure_f32(lmpc_handle *h) {
    if (h->dCwf) return LMPC_OK;
    if (!h->dCw)
        return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: the binary32 path runs on the wavefront kernel, which does not "
                                             "cover this problem (n <= 127, 1 <= m <= 1024)");
    const HostPack &P = h->P;
    const WaveLayout &Wl = h->W;
    const size_t total = (size_t)Wl.oGf + (size_t)P.m * P.m;
    std::vector<float> wb(total ? total : 1, 0.f);
    std::vector<float> Mf((size_t)P.m * P.n);
    for (size_t i = 0; i < Mf.size(); i++) Mf[i] = (float)P.M[i];
    std::memcpy(&wb[Wl.oM], Mf.data(), sizeof(float) * Mf.size());
    for (int j = 0; j < P.m; j++)
        for (int k = 0; k < P.n; k++) wb[Wl.oMt + (size_t)k * P.m + j] = Mf[(size_t)j * P.n + k];
    for (int a = 0; a < P.m; a++)
        for (int b = 0; b <= a; b++) {
            float acc = 0.f;
            for (int k = 0; k < P.n; k++) acc = std::fmaf(Mf[(size_t)a * P.n + k], Mf[(size_t)b * P.n + k], acc);
            wb[Wl.oG + (size_t)lmpc_tri(a) + b] = acc;
            wb[Wl.oGf + (size_t)a * P.m + b] = wb[Wl.oGf + (size_t)b * P.m + a] = acc;
        }
    for (int j = 0; j < P.m; j++) { wb[Wl.odu + j] = (float)P.du0[j]; wb[Wl.odl + j] = (float)P.dl0[j]; }
    for (size_t i = 0; i < P.Dth.size(); i++) wb[Wl.oDth + i] = (float)P.Dth[i];
    for (size_t i = 0; i < P.Rout.size(); i++) wb[Wl.oRout + i] = (float)P.Rout[i];
    for (size_t i = 0; i < P.x0.size(); i++) wb[Wl.ox0 + i] = (float)P.x0[i];
    for (size_t i = 0; i < P.Xth.size(); i++) wb[Wl.oXth + i] = (float)P.Xth[i];
    HIP_TRY(h, hipMalloc(&h->dCwf, sizeof(float) * wb.size()));
    HIP_TRY(h, hipMemcpy(h->dCwf, wb.data(), sizeof(float) * wb.size(), hipMemcpyHostToDevice));
    return LMPC_OK;
}

// does a batch of this handle go through the screening pass first?
bool will_screen(const lmpc_handle *h, int64_t nprob) {
    return !h->useWave && h->screen && h->L.eq_mask == 0ull && h->S.iter_limit > 1 && h->P.nth >= 1 &&
           h->P.nth <= 32 && nprob < (int64_t)0x7fffffff;
}

// ... and in front of the wavefront kernel?  (binary64 solves without binaries and without initially active rows:
// what the pass finishes is exactly what the wavefront kernel's first iteration would finish, bit for bit)
bool wave_screens(const lmpc_handle *h, int64_t nprob) {
    if (!h->useWave || !h->screen || !h->screenWave || h->bnb || h->dC == nullptr) return false;
    if (h->S.iter_limit <= 1 || h->P.nth < 1 || h->P.nth > 32 || nprob >= (int64_t)0x7fffffff) return false;
    if (h->screenPackOnly) return true;                  // (checked at setup)
    return h->L.eq_mask == 0ull;
}

int ensure_lists(lmpc_handle *h, int64_t nprob, hipStream_t st) {
    if (nprob <= h->listCap) return LMPC_OK;
    hipFree(h->dList); hipFree(h->dCount); hipFree(h->dList2); hipFree(h->dList3);
    h->dList = h->dCount = h->dList2 = h->dList3 = nullptr; h->listCap = 0;
    const size_t segCap = (size_t)lane_seg_cap(nprob);
    HIP_TRY(h, hipMalloc(&h->dList, sizeof(int32_t) * segCap * kShards));
    HIP_TRY(h, hipMalloc(&h->dCount, sizeof(int32_t) * 3 * kShards * kCountStride));   // two alternating sets + the parked list's
    HIP_TRY(h, hipMemsetAsync(h->dCount, 0, sizeof(int32_t) * 3 * kShards * kCountStride, st));
    h->listCap = nprob;
    h->countSet = 0;
    return LMPC_OK;
}

#define LMPC_SCREEN_SWITCH(CALL)                                                                       \
    switch (h->P.nth <= 16 ? h->P.nth : 32) { /* exact column count up to 16, padded beyond */        \
        case 1: rc = CALL(8, 1); break;    case 2: rc = CALL(8, 2); break;                             \
        case 3: rc = CALL(8, 3); break;    case 4: rc = CALL(8, 4); break;                             \
        case 5: rc = CALL(8, 5); break;    case 6: rc = CALL(8, 6); break;                             \
        case 7: rc = CALL(8, 7); break;    case 8: rc = CALL(8, 8); break;                             \
        case 9: rc = CALL(16, 9); break;   case 10: rc = CALL(16, 10); break;                          \
        case 11: rc = CALL(16, 11); break; case 12: rc = CALL(16, 12); break;                          \
        case 13: rc = CALL(16, 13); break; case 14: rc = CALL(16, 14); break;                          \
        case 15: rc = CALL(16, 15); break; case 16: rc = CALL(16, 16); break;                          \
        default: rc = CALL(32, 32); break;                                                             \
    }

// wavefront-kernel handles: the streaming pass finishes what needs no iterations, the wavefront kernel walks the rest
int launch_wave_screened(lmpc_handle *h, int64_t nprob, const double *theta, double *x, int32_t *flag,
                         int32_t *iters, uint64_t *active, const uint64_t *warm, hipStream_t st) {
    const int rc0 = ensure_lists(h, nprob, st);
    if (rc0 != LMPC_OK) return rc0;
    EventTriple ev{};
    if (h->prof) {
        HIP_TRY(h, pool_event(h, &ev.a));
        HIP_TRY(h, pool_event(h, &ev.mid));
        HIP_TRY(h, pool_event(h, &ev.b));
        HIP_TRY(h, hipEventRecord(ev.a, st));
    }
    int32_t *cnt_now = h->dCount + (size_t)h->countSet * kShards * kCountStride;
    int32_t *cnt_next = h->dCount + (size_t)(h->countSet ^ 1) * kShards * kCountStride;
    h->countSet ^= 1;
    const bool wide = h->P.nout > 1 && h->P.nout <= 16;
    int rc = LMPC_OK;
    const bool simf = h->L.sim.FG != nullptr;     // closed loop, plant step fused in (lmpc_simulate_device)
#define LMPC_SCRW(NM, NT) (simf ? launch_screen<NM, NT, 1>(h, nprob, theta, x, flag, iters, active, warm, cnt_now, st) \
                           : wide ? launch_screen<NM, NT, 3>(h, nprob, theta, x, flag, iters, active, warm, cnt_now, st) \
                                  : launch_screen<NM, NT, 0>(h, nprob, theta, x, flag, iters, active, warm, cnt_now, st))
    LMPC_SCREEN_SWITCH(LMPC_SCRW)
#undef LMPC_SCRW
    if (h->prof) HIP_TRY(h, hipEventRecord(ev.mid, st));
    if (rc == LMPC_OK) {
        h->waveList.list = h->dList; h->waveList.count = cnt_now; h->waveList.count_next = cnt_next;
        h->waveList.seg_cap = lane_seg_cap(nprob);
        rc = launch_wave(h, nprob, theta, x, flag, iters, active, warm, st);
        h->waveList = WaveList{};
    }
    if (h->prof) {
        if (rc == LMPC_OK) { HIP_TRY(h, hipEventRecord(ev.b, st)); h->events.push_back(ev); }
        else { hipEventDestroy(ev.a); hipEventDestroy(ev.mid); hipEventDestroy(ev.b); }
    }
    return rc;
}

// With or without the tiers pass?  Returns the variant for this call (0 = with, 1 = without) and, in *measure, the
// variant this call is to be TIMED as (-1: none).  Small batches and "qp_tiers" 2: with.  Large batches: the handle times
// three calls each way (events, read back without waiting by later calls), takes the best of each and goes with the
// faster; again every 512 calls.  (The pass costs ~2 ms per 10^6 problems whatever it finishes: a sample of mostly hard points -- every second
// one with removals, soft_doc -- is 3 % faster without it, the reference's mass_spring 1.9x with it.)
int qp_ab_choose(lmpc_handle *h, int64_t nprob, hipStream_t st, int *measure) {
    *measure = -1;
    if (h->qpTiers != 1 || nprob < 65536) return 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return 0; }
    for (int v = 0; v < 2; v++)
        if (h->qpAbPending[v] && hipEventQuery(h->qpAbEv[v][1]) == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, h->qpAbEv[v][0], h->qpAbEv[v][1]) == hipSuccess && h->qpAbN[v] > 0) {
                // the best of three samples per variant counts (a sample can hold unrelated queueing on the caller's stream)
                const double ns = 1e6 * (double)ms / (double)h->qpAbN[v];
                h->qpAbAcc[v] = (h->qpAbCnt[v] == 0 || ns < h->qpAbAcc[v]) ? ns : h->qpAbAcc[v];
                h->qpAbCnt[v]++;
            }
            h->qpAbPending[v] = false;
        }
    if (h->qpAbCnt[0] >= 3 && h->qpAbCnt[1] >= 3) {
        h->qpAbNsPer[0] = h->qpAbAcc[0]; h->qpAbNsPer[1] = h->qpAbAcc[1];
        h->qpAbCnt[0] = h->qpAbCnt[1] = 0;
    }
    (void)hipGetLastError();
    const long long phase = h->qpAbCalls++ % 512;
    int variant;
    if (phase < 6 && !h->qpAbPending[phase & 1]) {
        variant = (int)(phase & 1);
        *measure = variant;
        for (int e = 0; e < 2; e++)
            if (!h->qpAbEv[variant][e] && hipEventCreate(&h->qpAbEv[variant][e]) != hipSuccess) { *measure = -1; (void)hipGetLastError(); break; }
    } else {
        variant = (h->qpAbNsPer[0] >= 0.0 && h->qpAbNsPer[1] >= 0.0 && h->qpAbNsPer[1] < h->qpAbNsPer[0]) ? 1 : 0;
    }
    return variant;
}

// Small problems with many rows (lmpc_qp_tiers_kernel.hpp): the tiers pass over the whole batch -- it finishes what the
// screening pass would and every problem on an append-only path, optimal or infeasible -- then the wavefront kernel on
// its work list.  Cold plain binary64 solves in the n-chain form only.
bool qp_tiers_applies(const lmpc_handle *h, int64_t nprob, const double *x, const int32_t *flag, const uint64_t *warm) {
    return h->qpTiersOk && h->qpTiers && h->useWave && !h->waveGram && !h->bnb && warm == nullptr && x != nullptr &&
           flag != nullptr && h->asyncPhase == 0 && h->L.sim.FG == nullptr && h->waveSim.FG == nullptr && !h->keepOn &&
           h->L.gat.state == nullptr && h->S.iter_limit > h->P.n + 2 && nprob < (int64_t)0x7fffffff && h->screen;
}

int launch_wave_tiered(lmpc_handle *h, int64_t nprob, const double *theta, double *x, int32_t *flag, int32_t *iters,
                       uint64_t *active, hipStream_t st) {
    const int rc0 = ensure_lists(h, nprob, st);
    if (rc0 != LMPC_OK) return rc0;
    EventTriple ev{};
    if (h->prof) {
        HIP_TRY(h, pool_event(h, &ev.a));
        HIP_TRY(h, pool_event(h, &ev.mid));
        HIP_TRY(h, pool_event(h, &ev.b));
        HIP_TRY(h, hipEventRecord(ev.a, st));
    }
    int32_t *cnt_now = h->dCount + (size_t)h->countSet * kShards * kCountStride;
    int32_t *cnt_next = h->dCount + (size_t)(h->countSet ^ 1) * kShards * kCountStride;
    h->countSet ^= 1;
    int rc = launch_qp_tiers(h, nprob, theta, x, flag, iters, active, h->dList, cnt_now, lane_seg_cap(nprob), st, false);
    if (h->prof) HIP_TRY(h, hipEventRecord(ev.mid, st));
    if (rc == LMPC_OK) {
        h->waveList.list = h->dList; h->waveList.count = cnt_now; h->waveList.count_next = cnt_next;
        h->waveList.seg_cap = lane_seg_cap(nprob);
        rc = launch_wave(h, nprob, theta, x, flag, iters, active, nullptr, st);
        h->waveList = WaveList{};
    }
    if (h->prof) {
        if (rc == LMPC_OK) { HIP_TRY(h, hipEventRecord(ev.b, st)); h->events.push_back(ev); }
        else { hipEventDestroy(ev.a); hipEventDestroy(ev.mid); hipEventDestroy(ev.b); }
    }
    return rc;
}

// First large batch of a FRESH wavefront-kernel handle: whether a batch first runs at a smaller working-set capacity (two
// passes, lmpc_wave_launch.hpp) is decided from the working-set sizes the handle has seen -- and a fresh handle has
// seen none, so its first calls ran in one pass (pendulum N = 50: 7.2e7 against 1.3e8 solves/s) until a launch or two
// had reported.  Instead the first call solves the leading kProbe points of ITS OWN batch once more in front, into
// scratch outputs (cold, no closed-loop side effects), waits for that one small launch and reads the counters it left:
// the batch itself then runs in the configuration a warmed-up handle would choose.  Once per handle, only where two
// passes are possible at all; results are not affected (they never depend on the split).  Not inside a hipGraph
// capture: make one call before capturing, or lmpc_set_option("wave_probe", 0).
int wave_probe(lmpc_handle *h, const double *theta, int64_t nprob, hipStream_t st) {
    if (h->waveProbed || !h->waveProbe || !h->useWave || h->avi || h->bnb || h->waveTwoPass >= 0 || !h->bigPath ||
        h->W.cap < 40 || nprob < 4 * kProbe || theta == nullptr)
        return LMPC_OK;
    // (the probe allocates, copies and waits -- none of which a stream capture allows.  Inside a capture
    // nothing is probed and nothing is marked: the first call outside one probes.)
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return LMPC_OK; }
    }
    h->waveProbed = true;
    double *px = nullptr;
    int32_t *pf = nullptr;
    HIP_TRY(h, hipMalloc(&px, sizeof(double) * (size_t)kProbe * h->P.nout));
    if (hipMalloc(&pf, sizeof(int32_t) * (size_t)kProbe) != hipSuccess) { hipFree(px); return fail(h, LMPC_ERR_HIP, "lmpc: probe scratch"); }
    // the probe is a plain cold solve: closed-loop fusion, work lists, kept factors and profiling stay out of it
    const SimFuse sim = h->L.sim; const WaveSim wsim = h->waveSim; const WaveList wl = h->waveList;
    const int phase = h->asyncPhase; const bool keep = h->keepOn, prof = h->prof;
    h->L.sim = SimFuse{}; h->waveSim = WaveSim{}; h->waveList = WaveList{}; h->asyncPhase = 0; h->keepOn = false; h->prof = false;
    int rc = LMPC_OK;
    if (wave_screens(h, nprob)) rc = ensure_lists(h, nprob, st);          // (sized for the batch that follows, once)
    if (rc == LMPC_OK)
        rc = wave_screens(h, kProbe) ? launch_wave_screened(h, kProbe, theta, px, pf, nullptr, nullptr, nullptr, st)
                                     : launch_wave(h, kProbe, theta, px, pf, nullptr, nullptr, nullptr, st);
    h->L.sim = sim; h->waveSim = wsim; h->waveList = wl; h->asyncPhase = phase; h->keepOn = keep; h->prof = prof;
    if (rc == LMPC_OK && h->dStat && h->hStat) {
        unsigned long long raw[64 * 16];
        if (hipMemcpyAsync(raw, h->dStat, sizeof(raw), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) {
            for (int q = 0; q < 5; q++) {
                unsigned long long sum = 0ull;
                for (int sh = 0; sh < 64; sh++) sum += raw[sh * 16 + q];
                h->hStat[q] = sum;                    // what the next launch would have published by itself
            }
        } else { (void)hipGetLastError(); }
    } else if (rc == LMPC_OK) {
        (void)hipStreamSynchronize(st);
    }
    (void)hipFree(px); (void)hipFree(pf);
    return rc;
}

// Scenario-asynchronous closed loop, streaming half (lmpc_simrun_kernel.hpp): every scenario -- first round -- or the
// scenarios of the round before's list run ahead in registers to their next step that needs iterations; those go on
// the work list the iterating half (lane kernel or wavefront kernel) consumes.
int launch_sim_run(lmpc_handle *h, int64_t nprob, const double *theta, uint64_t *active, const uint64_t *warm, hipStream_t st) {
    { const int rc0 = ensure_lists(h, nprob, st); if (rc0 != LMPC_OK) return rc0; }
    int32_t *cnt_now = h->dCount + (size_t)h->countSet * kShards * kCountStride;
    int32_t *cnt_next = h->dCount + (size_t)(h->countSet ^ 1) * kShards * kCountStride;
    h->countSet ^= 1;
    h->asyncCntNow = cnt_now; h->asyncCntNext = cnt_next;
    const long long segCap = lane_seg_cap(nprob);
    if (!h->dList2) HIP_TRY(h, hipMalloc(&h->dList2, sizeof(int32_t) * (size_t)lane_seg_cap(h->listCap) * kShards));
    if (!h->dList3) HIP_TRY(h, hipMalloc(&h->dList3, sizeof(int32_t) * (size_t)lane_seg_cap(h->listCap) * kShards));
    int32_t *parkCnt = h->dCount + 2 * (size_t)kShards * kCountStride;
    // first round: every scenario; later rounds: the scenarios of the round before's list, compacted
    const int32_t *lin = h->asyncListIn, *cin = h->asyncCntIn;
    int32_t *lout = (lin == h->dList) ? h->dList2 : h->dList;   // (the parked list as input: the work list is empty)
    h->asyncListOut = lout;
    const unsigned grid = lin ? (unsigned)(((h->asyncMaxIn + 255) / 256) * kShards) : (unsigned)((nprob + 255) / 256);
    const SimFuse &Sf = h->L.sim;
    const int nthp_ = h->L.nthp;
    const size_t mp_ = ((size_t)h->P.m + 7) & ~(size_t)7;
    const size_t ldsr = sizeof(double) * (mp_ * nthp_ + 2 * mp_ + (size_t)kMaxSimU * nthp_ + kMaxSimU + 64 + 8 * kMaxSimU);
    const bool small_ = h->simSmall && Sf.nx <= 4 && Sf.nu == 1 && h->P.m >= 1 && h->P.m <= 8 && h->P.nth <= 8;
#define LMPC_SRUN_(NM, NT, SM) hipLaunchKernelGGL((sim_run_kernel<NM, NT, SM>), dim3(grid), dim3(256), ldsr, st, h->L, h->dC, \
        const_cast<double *>(theta), Sf.kstep, h->asyncT, active, warm != nullptr ? 1 : 0, Sf.utraj, Sf.xtraj_base, \
        Sf.flag_min, lout, cnt_now, segCap, kShards, (long long)nprob, lin, cin, h->asyncCap, h->dList3, parkCnt, \
        h->asyncX, h->asyncR, h->asyncUp)
#define LMPC_SRUN(NM, NT) do { if constexpr (NT <= 8) { if (small_) LMPC_SRUN_(NM, NT, true); else LMPC_SRUN_(NM, NT, false); } \
                            else LMPC_SRUN_(NM, NT, false); } while (0)
    switch (h->P.nth) {
        case 1: LMPC_SRUN(8, 1); break;    case 2: LMPC_SRUN(8, 2); break;    case 3: LMPC_SRUN(8, 3); break;
        case 4: LMPC_SRUN(8, 4); break;    case 5: LMPC_SRUN(8, 5); break;    case 6: LMPC_SRUN(8, 6); break;
        case 7: LMPC_SRUN(8, 7); break;    case 8: LMPC_SRUN(8, 8); break;    case 9: LMPC_SRUN(16, 9); break;
        case 10: LMPC_SRUN(16, 10); break; case 11: LMPC_SRUN(16, 11); break; case 12: LMPC_SRUN(16, 12); break;
        case 13: LMPC_SRUN(16, 13); break; case 14: LMPC_SRUN(16, 14); break; case 15: LMPC_SRUN(16, 15); break;
        case 16: LMPC_SRUN(16, 16); break;
        default: return fail(h, LMPC_ERR_BADARG, "lmpc: sim_run needs nth <= 16");
    }
#undef LMPC_SRUN
#undef LMPC_SRUN_
    HIP_TRY(h, hipGetLastError());
    h->asyncX = h->asyncR = h->asyncUp = nullptr;       // only the first pass forms theta
    if (h->asyncResetPark) HIP_TRY(h, hipMemsetAsync(parkCnt, 0, sizeof(int32_t) * kShards * kCountStride, st));
    return LMPC_OK;
}

int launch(lmpc_handle *h, int64_t nprob, const double *theta, double *x, int32_t *flag,
           int32_t *iters, uint64_t *active, const uint64_t *warm, hipStream_t st) {
    if (h->avi) {
        // non-symmetric H (variational objective): its own kernel; timing events around the one launch
        EventTriple ev{};
        if (h->prof) {
            HIP_TRY(h, pool_event(h, &ev.a));
            HIP_TRY(h, pool_event(h, &ev.b));
            HIP_TRY(h, hipEventRecord(ev.a, st));
        }
        const int rca = launch_avi(h, nprob, theta, x, flag, iters, active, warm, st);
        if (h->prof) {
            if (rca == LMPC_OK) { HIP_TRY(h, hipEventRecord(ev.b, st)); h->events.push_back(ev); }
            else { hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
        }
        return rca;
    }
    if (h->useWave) {
        // scenario-asynchronous closed loop on the wavefront path: streaming half = sim_run_kernel on the handle's
        // screening pack, iterating half = the wavefront kernel on that pass's work list (it advances the scenarios
        // it solves, WaveSim)
        if (h->asyncPhase == 1) return launch_sim_run(h, nprob, theta, active, warm, st);
        if (h->asyncPhase == 2) {
            h->waveList.list = h->asyncListOut; h->waveList.count = h->asyncCntNow; h->waveList.count_next = h->asyncCntNext;
            h->waveList.seg_cap = lane_seg_cap(nprob);
            const int rcw = launch_wave(h, nprob, theta, x, flag, iters, active, warm, st);
            h->waveList = WaveList{};
            return rcw;
        }
        if (!h->waveProbed && h->L.sim.FG == nullptr && h->waveSim.FG == nullptr) {
            const int rcp = wave_probe(h, theta, nprob, st);
            if (rcp != LMPC_OK) return rcp;
        }
        if (qp_tiers_applies(h, nprob, x, flag, warm)) {
            int measure = -1;
            const int variant = qp_ab_choose(h, nprob, st, &measure);
            if (measure >= 0) HIP_TRY(h, hipEventRecord(h->qpAbEv[measure][0], st));
            const int rct = variant == 0 ? launch_wave_tiered(h, nprob, theta, x, flag, iters, active, st)
                          : (wave_screens(h, nprob) ? launch_wave_screened(h, nprob, theta, x, flag, iters, active, warm, st)
                                                    : launch_wave(h, nprob, theta, x, flag, iters, active, warm, st));
            if (measure >= 0 && rct == LMPC_OK) {
                HIP_TRY(h, hipEventRecord(h->qpAbEv[measure][1], st));
                h->qpAbN[measure] = nprob; h->qpAbPending[measure] = true;
            }
            return rct;
        }
        return wave_screens(h, nprob) ? launch_wave_screened(h, nprob, theta, x, flag, iters, active, warm, st)
                                      : launch_wave(h, nprob, theta, x, flag, iters, active, warm, st);
    }
    // block size: the one that keeps most wavefronts resident per CU under the 160 KiB LDS cap
    int bestB = 0, bestWaves = -1;
    size_t bestLds = 0;
    for (int B : {256, 128, 64}) {
        const size_t lds = lane_lds_bytes(h->P, h->laneN, B);
        if (lds > kLdsMax) continue;
        int blocks = (int)(kLdsMax / (lds ? lds : 1));
        int waves = blocks * (B / 64);
        if (waves > 32) waves = 32;
        if (waves > bestWaves) { bestWaves = waves; bestB = B; bestLds = lds; }
    }
    if (h->laneBlock > 0 && lane_lds_bytes(h->P, h->laneN, h->laneBlock) <= kLdsMax) {
        bestB = h->laneBlock;
        bestLds = lane_lds_bytes(h->P, h->laneN, h->laneBlock);
    }
    if (!bestB) return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: constant pack does not fit in LDS");
    // Cold starts without initially-active rows go through the screening pass first.
    // (warm starts too: the screening pass finishes the problems whose warm mask is empty and whose
    // unconstrained optimum is feasible, everything else is queued with its mask)
    const bool screened = will_screen(h, nprob);
    static_assert((kShards & (kShards - 1)) == 0, "the screening kernel masks the shard index");
    if (screened) {
        const int rc0 = ensure_lists(h, nprob, st);
        if (rc0 != LMPC_OK) return rc0;
    }
    EventTriple ev{};
    if (h->prof) {
        HIP_TRY(h, pool_event(h, &ev.a));
        HIP_TRY(h, pool_event(h, &ev.mid));
        HIP_TRY(h, pool_event(h, &ev.b));
        HIP_TRY(h, hipEventRecord(ev.a, st));
    }
    int rc = LMPC_OK;
    int ab_variant = 0, ab_measure = -1;          // (tiers pass on this path: measured like on the wavefront path)
    const bool sim = h->L.sim.FG != nullptr;      // closed-loop instantiations (SimFuse), lmpc_simulate* only
    const bool gather = !sim && h->L.gat.state != nullptr;   // generated-controller screening (GatherArgs)
    const bool wide = !sim && !gather && h->P.nout > 1 && h->P.nout <= 16;   // several outputs: transposed stores
    const bool multi = !sim && h->P.nout > 1;                                  // ... and the iterating kernel's epilogue
    if (gather && !screened) return fail(h, LMPC_ERR_BADARG, "lmpc: gather mode needs the screening pass");
    // two counter sets used alternately: the iterating kernel of call k clears the set of call k+1
    int32_t *cnt_now = nullptr, *cnt_next = nullptr;
    if (screened && h->asyncPhase == 2) {          // scenario-asynchronous closed loop, iterating half: the work
        cnt_now = h->asyncCntNow;                  // list and its counters come from the sim_run pass before
        cnt_next = h->asyncCntNext;
    } else if (screened && h->asyncPhase == 1) {   // ... streaming half: sim_run_kernel instead of the screening pass
        return launch_sim_run(h, nprob, theta, active, warm, st);
    } else if (screened && !sim && warm == nullptr && (!gather || active == nullptr) && fast_covers(h)) {
        // small boxed problems, cold plain solve: ONE kernel streams the batch and solves what needs iterations
        // (lmpc_fast_kernel.hpp); no work list, no second launch
        // (an error word raised by an EARLIER call on this handle is reported here, before anything new is enqueued)
        rc = check_fast_err(h);
        if (rc == LMPC_OK) rc = launch_fast(h, nprob, theta, x, flag, iters, active, st);
        if (h->prof) {       // one kernel: no event between "the two kernels" (each record is a packet the queue retires)
            h->eventPool.push_back(ev.mid); ev.mid = nullptr;
            if (rc == LMPC_OK) { HIP_TRY(h, hipEventRecord(ev.b, st)); h->events.push_back(ev); }
            else { hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
        }
        return rc;
    } else if (screened && !sim && !gather && warm == nullptr && x != nullptr && flag != nullptr && h->qpTiersOk &&
               h->qpTiers && h->P.ms < h->P.m && h->dCw != nullptr && h->S.iter_limit > h->P.n + 2 && h->asyncPhase == 0 &&
               (ab_variant = qp_ab_choose(h, nprob, st, &ab_measure)) == 0) {
        // general rows on the lane path: the tiers pass (lmpc_qp_tiers_kernel.hpp) instead of the screening pass -- it
        // finishes every append-only path, the lane kernel walks the rest
        if (ab_measure >= 0) HIP_TRY(h, hipEventRecord(h->qpAbEv[ab_measure][0], st));
        cnt_now = h->dCount + (size_t)h->countSet * kShards * kCountStride;
        cnt_next = h->dCount + (size_t)(h->countSet ^ 1) * kShards * kCountStride;
        h->countSet ^= 1;
        rc = launch_qp_tiers(h, nprob, theta, x, flag, iters, active, h->dList, cnt_now, lane_seg_cap(nprob), st, false);
    } else if (screened) {
        if (ab_measure >= 0) HIP_TRY(h, hipEventRecord(h->qpAbEv[ab_measure][0], st));
        cnt_now = h->dCount + (size_t)h->countSet * kShards * kCountStride;
        cnt_next = h->dCount + (size_t)(h->countSet ^ 1) * kShards * kCountStride;
        h->countSet ^= 1;
#define LMPC_SCR(NM, NT) (sim ? launch_screen<NM, NT, 1>(h, nprob, theta, x, flag, iters, active, warm, cnt_now, st) \
                         : gather ? launch_screen<NM, NT, 2>(h, nprob, theta, x, flag, iters, active, warm, cnt_now, st) \
                         : wide ? launch_screen<NM, NT, 3>(h, nprob, theta, x, flag, iters, active, warm, cnt_now, st) \
                                : launch_screen<NM, NT, 0>(h, nprob, theta, x, flag, iters, active, warm, cnt_now, st))
        LMPC_SCREEN_SWITCH(LMPC_SCR)
#undef LMPC_SCR
    }
    if (h->prof) HIP_TRY(h, hipEventRecord(ev.mid, st));
    const int32_t *list = screened ? (h->asyncPhase == 2 ? h->asyncListOut : h->dList) : nullptr;
    const int32_t *count = cnt_now;
    // problems whose constraints are exactly the n simple bounds (ms == m == n == N) get the
    // instantiation with the row scans unrolled and the working-set capacity cut to N
    const bool boxed = h->P.m == h->laneN && h->P.n == h->laneN && h->P.ms == h->P.m;
    if (rc == LMPC_OK) switch (h->laneN) {
#define LMPC_LN(NN, MSS, MAA) (sim ? launch_lane<NN, MSS, MAA, true, false>(h, bestB, bestLds, nprob, theta, x, flag, iters, active, warm, list, count, cnt_next, st) \
                               : (multi && NN <= 6) ? launch_lane<NN, MSS, MAA, false, (NN <= 6)>(h, bestB, bestLds, nprob, theta, x, flag, iters, active, warm, list, count, cnt_next, st) \
                                   : launch_lane<NN, MSS, MAA, false, false>(h, bestB, bestLds, nprob, theta, x, flag, iters, active, warm, list, count, cnt_next, st))
#define LMPC_CASE(NN) case NN: rc = boxed ? LMPC_LN(NN, NN, NN) : LMPC_LN(NN, 0, NN + 1); break;
        LMPC_CASE(2) LMPC_CASE(3) LMPC_CASE(4) LMPC_CASE(5) LMPC_CASE(6) LMPC_CASE(8) LMPC_CASE(10) LMPC_CASE(12)
#undef LMPC_CASE
#undef LMPC_LN
        default: rc = fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: no kernel instantiation"); break;
    }
    if (ab_measure >= 0 && rc == LMPC_OK) {
        HIP_TRY(h, hipEventRecord(h->qpAbEv[ab_measure][1], st));
        h->qpAbN[ab_measure] = nprob; h->qpAbPending[ab_measure] = true;
    }
    if (h->prof) {
        if (rc == LMPC_OK) { HIP_TRY(h, hipEventRecord(ev.b, st)); h->events.push_back(ev); }
        else { hipEventDestroy(ev.a); hipEventDestroy(ev.mid); hipEventDestroy(ev.b); }
    }
    return rc;
}

// scratch of the closed-loop entry points (binary64-sized; the binary32 loop uses the same buffers)
int ensure_sim(lmpc_handle *h, int64_t N) {
    if (N <= h->simCap) return LMPC_OK;
    const int nu = h->P.nout;
    const size_t w = (size_t)h->P.words();
    hipFree(h->simTheta); hipFree(h->simTheta2); hipFree(h->simU); hipFree(h->simFlag); hipFree(h->simAct); hipFree(h->simFG); hipFree(h->simK);
    h->simTheta = h->simTheta2 = h->simU = h->simFG = nullptr; h->simFlag = nullptr; h->simAct = nullptr; h->simK = nullptr; h->simCap = 0;
    HIP_TRY(h, hipMalloc(&h->simTheta, sizeof(double) * (size_t)N * (h->P.nth ? h->P.nth : 1)));
    HIP_TRY(h, hipMalloc(&h->simU, sizeof(double) * (size_t)N * (nu ? nu : 1)));
    HIP_TRY(h, hipMalloc(&h->simFlag, sizeof(int32_t) * (size_t)N));
    HIP_TRY(h, hipMalloc(&h->simAct, sizeof(uint64_t) * (size_t)N * w));
    HIP_TRY(h, hipMalloc(&h->simFG, sizeof(double) * (32 * 32 + 32 * 64)));
    h->simCap = N;
    return LMPC_OK;
}

}  // namespace

// what the second API unit (lmpc_api_loop.hip: closed loop, generated controller, observer) needs of this one
namespace lmpc {
int api_launch(lmpc_handle *h, int64_t nprob, const double *theta, double *x, int32_t *flag, int32_t *iters, uint64_t *active,
               const uint64_t *warm, hipStream_t st) { return launch(h, nprob, theta, x, flag, iters, active, warm, st); }
int api_ensure_sim(lmpc_handle *h, int64_t N) { return ensure_sim(h, N); }
int api_ensure_f32(lmpc_handle *h) { return ensure_f32(h); }
bool api_will_screen(const lmpc_handle *h, int64_t nprob) { return will_screen(h, nprob); }
bool api_wave_screens(const lmpc_handle *h, int64_t nprob) { return wave_screens(h, nprob); }
int api_wave_probe(lmpc_handle *h, const double *theta, int64_t nprob, hipStream_t st) { return wave_probe(h, theta, nprob, st); }
int api_launch_wave_f32(lmpc_handle *h, const float *dC, int64_t nprob, const float *theta, float *x, int32_t *flag, int32_t *iters,
                        uint64_t *active, const uint64_t *warm, hipStream_t st) {
    return launch_wave_t<float>(h, dC, nprob, theta, x, flag, iters, active, warm, st);
}
}  // namespace lmpc

extern "C" {

int lmpc_abi_version(void) { return 2; }     // 2: lmpc_settings grew eps_prox / eta_prox

void lmpc_default_settings(lmpc_settings *s) {
    if (!s) return;
    s->primal_tol = 1e-6; s->dual_tol = 1e-12; s->zero_tol = 1e-11; s->progress_tol = 1e-6;
    s->fval_bound = 1e30; s->rho_soft = 1e-6; s->cycle_tol = 10; s->iter_limit = 10000;
    s->eps_prox = 0.0; s->eta_prox = 1e-6;
}

void lmpc_default_settings_f32(lmpc_settings *s) {
    if (!s) return;
    s->primal_tol = 1e-4; s->dual_tol = 1e-6; s->zero_tol = 1e-6; s->progress_tol = 1e-4;
    s->fval_bound = 1e30; s->rho_soft = 1e-3; s->cycle_tol = 10; s->iter_limit = 10000;
    s->eps_prox = 0.0; s->eta_prox = 1e-6;
}

// The code objects of the kernels this handle's plain solves will launch are brought onto the device at SETUP: HIP
// loads a translation unit's code lazily with the first launch of one of its kernels -- 5 to 15 ms for the wavefront
// kernel's units -- which otherwise lands in the caller's first solve (once per process and unit).
static void preload_code(lmpc_handle *h) {
    lmpc::DeviceScope scope;
    if (scope.enter(h->device) != hipSuccess) { (void)hipGetLastError(); return; }
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void *)screen_kernel<8, 7, 0>);         // this unit: screening and lane kernels
    loop_preload();                                                                // ... the closed-loop / controller unit
    if (h->avi) avi_preload(h);
    else {
        if (fast_covers(h)) fast_preload(h);
        if (h->qpTiersOk && (h->useWave || h->P.ms < h->P.m))
            (void)launch_qp_tiers(h, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, true);
        if (h->dCw && h->useWave) {
            h->preloadOnly = true;
            (void)launch_wave(h, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
            // (hybrid problems are the ones solved in binary32 too -- lmpc_solve_batch_f32*: that unit and its pack as well)
            if (h->bnb && ensure_f32(h) == LMPC_OK)
                (void)launch_wave_t<float>(h, h->dCwf, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
            h->preloadOnly = false;
        }
    }
    (void)hipGetLastError();
    h->err.clear();
}

static int setup_common(lmpc_handle **out, int n, int m, int ms, int nth, int nout, const double *H,
                        const double *f, const double *f_theta, const double *A, const double *bu,
                        const double *bl, const double *W, const int32_t *sense, const double *Kfb, int nx,
                        const lmpc_settings *s, const int32_t *break_points, int n_break_points, int is_avi, int device) {
    if (!out) return fail(nullptr, LMPC_ERR_BADARG, "lmpc_setup: out is NULL");
    *out = nullptr;
    if (n_break_points < 0 || (n_break_points > 0 && !break_points))
        return fail(nullptr, LMPC_ERR_BADARG, "lmpc_setup_ex: break_points is NULL");
    if (n_break_points > 0)
        // priority levels of the constraints (reference setup.jl:12, mpc2mpqp.jl:890-892): DAQP then solves a
        // hierarchy of problems, one per level; that mode is not built -- refuse rather than ignore the levels
        return fail(nullptr, LMPC_ERR_UNSUPPORTED, "lmpc_setup_ex: prioritised constraints (mpQP.break_points non-empty) "
                                                   "are not supported by the batched backend");
    if (!H || n <= 0) return fail(nullptr, LMPC_ERR_BADARG, "lmpc_setup: bad dimensions or NULL array");
    const bool sym = h_is_symmetric(H, n);
    if (is_avi < 0) is_avi = sym ? 0 : 1;         // lmpc_setup: decided as the reference decides it (mpc2mpqp.jl:897)
    if (!is_avi && !sym)
        return fail(nullptr, LMPC_ERR_BADARG, "lmpc_setup_ex: H is not symmetric but is_avi is 0 (the reference passes "
                                              "is_avi = !mpQP.is_symmetric, setup.jl:13)");
    lmpc_handle *h = new lmpc_handle();
    if (s) h->S = *s; else lmpc_default_settings(&h->S);
    h->device = device;
    int rc;
    if (h->S.eps_prox > 0.0) {
        // DAQP's proximal-point iterations (a merely semidefinite H): the subproblems run on the L D U kernel
        if (is_avi) rc = fail(h, LMPC_ERR_UNSUPPORTED, "lmpc_setup: eps_prox > 0 together with a non-symmetric H (is_avi) is not supported");
        else rc = qp_to_prox(h->P, n, m, ms, nth, nout, H, f, f_theta, A, bu, bl, W, sense, Kfb, nx, h->S.eps_prox, h->err);
        if (rc == LMPC_OK) rc = finalize_avi(h);
        if (rc == LMPC_OK) fill_layout(h);
    } else if (is_avi) {
        rc = qp_to_avi(h->P, n, m, ms, nth, nout, H, f, f_theta, A, bu, bl, W, sense, Kfb, nx, h->err);
        if (rc == LMPC_OK) rc = finalize_avi(h);
        if (rc == LMPC_OK) fill_layout(h);
    } else {
        rc = qp_to_ldp(h->P, n, m, ms, nth, nout, H, f, f_theta, A, bu, bl, W, sense, Kfb, nx, h->err);
        if (rc == LMPC_OK) rc = finalize_handle(h);
    }
    if (rc != LMPC_OK) { g_setup_err = h->err; lmpc_free(h); return rc; }
    preload_code(h);
    *out = h;
    return LMPC_OK;
}

int lmpc_setup(lmpc_handle **out, int n, int m, int ms, int nth, int nout, const double *H,
               const double *f, const double *f_theta, const double *A, const double *bu,
               const double *bl, const double *W, const int32_t *sense, const double *Kfb, int nx,
               const lmpc_settings *s, int device) {
    return setup_common(out, n, m, ms, nth, nout, H, f, f_theta, A, bu, bl, W, sense, Kfb, nx, s, nullptr, 0, -1, device);
}

int lmpc_setup_ex(lmpc_handle **out, int n, int m, int ms, int nth, int nout, const double *H,
                  const double *f, const double *f_theta, const double *A, const double *bu,
                  const double *bl, const double *W, const int32_t *sense, const double *Kfb, int nx,
                  const lmpc_settings *s, const int32_t *break_points, int n_break_points, int is_avi, int device) {
    return setup_common(out, n, m, ms, nth, nout, H, f, f_theta, A, bu, bl, W, sense, Kfb, nx, s, break_points,
                        n_break_points, is_avi != 0 ? 1 : 0, device);
}

int lmpc_is_avi(const lmpc_handle *h) { return h ? (h->avi ? 1 : 0) : LMPC_ERR_BADARG; }

int lmpc_get_avi(const lmpc_handle *h, double *MR, double *G) {
    if (!h || !h->avi) return LMPC_ERR_BADARG;
    if (MR) std::memcpy(MR, h->P.MR.data(), sizeof(double) * h->P.MR.size());
    if (G) std::memcpy(G, h->P.Gf.data(), sizeof(double) * h->P.Gf.size());
    return LMPC_OK;
}

int lmpc_setup_ldp(lmpc_handle **out, int n, int m, int ms, int nth, int nout, const double *M,
                   const double *du, const double *dl, const double *Dth, const double *Rout,
                   const double *x0, const double *Xth, const int32_t *sense,
                   const lmpc_settings *s, int device) {
    if (!out) return fail(nullptr, LMPC_ERR_BADARG, "lmpc_setup_ldp: out is NULL");
    *out = nullptr;
    if (n <= 0 || m < 0 || nth < 0 || nout <= 0 || nout > n || ms < 0 || ms > m ||
        (m > 0 && (!M || !du || !dl)) || (m * nth > 0 && !Dth) || !Rout || !x0 || (nth > 0 && !Xth))
        return fail(nullptr, LMPC_ERR_BADARG, "lmpc_setup_ldp: bad dimensions or NULL array");
    lmpc_handle *h = new lmpc_handle();
    if (s) h->S = *s; else lmpc_default_settings(&h->S);
    h->device = device;
    HostPack &P = h->P;
    P.n = n; P.m = m; P.ms = ms; P.nth = nth; P.nout = nout;
    P.M.assign(M, M + (size_t)m * n);
    P.du0.assign(du, du + m);
    P.dl0.assign(dl, dl + m);
    P.Dth.assign(Dth, Dth + (size_t)m * nth);
    P.Rout.assign(Rout, Rout + (size_t)nout * n);
    P.x0.assign(x0, x0 + nout);
    P.Xth.assign(Xth, Xth + (size_t)nout * nth);
    P.sense.assign(m, 0);
    if (sense) P.sense.assign(sense, sense + m);
    int rc = finish_pack(P, h->err);
    if (rc == LMPC_OK) rc = finalize_handle(h);
    if (rc != LMPC_OK) { g_setup_err = h->err; lmpc_free(h); return rc; }
    preload_code(h);
    *out = h;
    return LMPC_OK;
}

int lmpc_transform(int n, int m, int ms, int nth, int nout, const double *H, const double *f,
                   const double *f_theta, const double *A, const double *bu, const double *bl,
                   const double *W, const int32_t *sense, const double *Kfb, int nx, double *M,
                   double *du, double *dl, double *Dth, double *Rout, double *x0, double *Xth) {
    HostPack P;
    std::string err;
    int rc = qp_to_ldp(P, n, m, ms, nth, nout, H, f, f_theta, A, bu, bl, W, sense, Kfb, nx, err);
    if (rc != LMPC_OK) return fail(nullptr, rc, err);
    auto cp = [](double *dst, const std::vector<double> &src) {
        if (dst && !src.empty()) std::memcpy(dst, src.data(), sizeof(double) * src.size());
    };
    cp(M, P.M); cp(du, P.du0); cp(dl, P.dl0); cp(Dth, P.Dth); cp(Rout, P.Rout); cp(x0, P.x0); cp(Xth, P.Xth);
    return LMPC_OK;
}

int lmpc_get_prox(const lmpc_handle *h, double *Hinv, double *x0f, double *Xthf, double *Kth) {
    if (!h || !h->P.prox) return LMPC_ERR_BADARG;
    auto cp = [](double *dst, const std::vector<double> &src) {
        if (dst && !src.empty()) std::memcpy(dst, src.data(), sizeof(double) * src.size());
    };
    cp(Hinv, h->P.Hinv); cp(x0f, h->P.x0f); cp(Xthf, h->P.Xthf); cp(Kth, h->P.Kth);
    return LMPC_OK;
}

int lmpc_transform_avi(int n, int m, int ms, int nth, int nout, const double *H, const double *f,
                       const double *f_theta, const double *A, const double *bu, const double *bl,
                       const double *W, const int32_t *sense, const double *Kfb, int nx, double *ML, double *MR,
                       double *G, double *du, double *dl, double *Dth, double *Rout, double *x0, double *Xth) {
    HostPack P;
    std::string err;
    int rc = qp_to_avi(P, n, m, ms, nth, nout, H, f, f_theta, A, bu, bl, W, sense, Kfb, nx, err);
    if (rc != LMPC_OK) return fail(nullptr, rc, err);
    auto cp = [](double *dst, const std::vector<double> &src) {
        if (dst && !src.empty()) std::memcpy(dst, src.data(), sizeof(double) * src.size());
    };
    cp(ML, P.M); cp(MR, P.MR); cp(G, P.Gf); cp(du, P.du0); cp(dl, P.dl0); cp(Dth, P.Dth); cp(Rout, P.Rout);
    cp(x0, P.x0); cp(Xth, P.Xth);
    return LMPC_OK;
}

int lmpc_get_ldp(const lmpc_handle *h, double *M, double *du, double *dl, double *Dth, double *Rout,
                 double *x0, double *Xth, int32_t *sense) {
    if (!h) return LMPC_ERR_BADARG;
    const HostPack &P = h->P;
    auto cp = [](double *dst, const std::vector<double> &src) {
        if (dst && !src.empty()) std::memcpy(dst, src.data(), sizeof(double) * src.size());
    };
    cp(M, P.M); cp(du, P.du0); cp(dl, P.dl0); cp(Dth, P.Dth); cp(Rout, P.Rout); cp(x0, P.x0); cp(Xth, P.Xth);
    if (sense && !P.sense.empty()) std::memcpy(sense, P.sense.data(), sizeof(int32_t) * P.sense.size());
    return LMPC_OK;
}

int lmpc_get_dims(const lmpc_handle *h, int32_t dims[6]) {
    if (!h || !dims) return LMPC_ERR_BADARG;
    dims[0] = h->P.n; dims[1] = h->P.m; dims[2] = h->P.ms; dims[3] = h->P.nth; dims[4] = h->P.nout;
    dims[5] = h->P.words();
    return LMPC_OK;
}

int lmpc_active_words(const lmpc_handle *h) { return h ? h->P.words() : LMPC_ERR_BADARG; }

int lmpc_set_settings(lmpc_handle *h, const lmpc_settings *s) {
    if (!h || !s) return LMPC_ERR_BADARG;
    if (s->eps_prox != h->S.eps_prox)
        return fail(h, LMPC_ERR_BADARG, "lmpc_set_settings: eps_prox is fixed at setup (the factorisation depends on it): set up again");
    h->S = *s;
    fill_layout(h);
    return LMPC_OK;
}

int lmpc_solve_batch_device(lmpc_handle *h, int64_t N, const double *theta, double *x,
                            int32_t *exitflag, int32_t *iters, uint64_t *active,
                            const uint64_t *warm, void *stream) {
    if (!h) return LMPC_ERR_BADARG;
    if (N < 0 || (N > 0 && (!x || !exitflag || (h->P.nth > 0 && !theta))))
        return fail(h, LMPC_ERR_BADARG, "lmpc_solve_batch_device: NULL array or negative N");
    if (N == 0) return LMPC_OK;
    if (N > (int64_t)0x7fffffff * 64) return fail(h, LMPC_ERR_BADARG, "lmpc: batch too large for one launch");
    LMPC_ENTER_DEVICE(h);
    return launch(h, N, theta, x, exitflag, iters, active, warm, (hipStream_t)stream);
}

// SEVERAL batches of N points each in one call (round 5): on the handles the one-launch kernel covers -- the headline's
// class -- ONE kernel launch takes all of them (fast_kernel_multi: the solving tail of a batch runs under the stream of
// the next); everywhere else the batches are enqueued one after the other on `stream`, exactly as n_batches calls of
// lmpc_solve_batch_device would.  Results are those of the single-batch call bit for bit.
int lmpc_solve_batches_device(lmpc_handle *h, int32_t n_batches, int64_t N, const double *const *theta, double *const *x,
                              int32_t *const *exitflag, void *stream) {
    if (!h) return LMPC_ERR_BADARG;
    if (n_batches < 0 || N < 0 || (n_batches > 0 && (!theta || !x || !exitflag)))
        return fail(h, LMPC_ERR_BADARG, "lmpc_solve_batches_device: NULL table or negative count");
    if (n_batches == 0 || N == 0) return LMPC_OK;
    for (int b = 0; b < n_batches; b++)
        if (!x[b] || !exitflag[b] || (h->P.nth > 0 && !theta[b]))
            return fail(h, LMPC_ERR_BADARG, "lmpc_solve_batches_device: NULL array in the table");
    if (N > (int64_t)0x7fffffff * 64) return fail(h, LMPC_ERR_BADARG, "lmpc: batch too large for one launch");
    LMPC_ENTER_DEVICE(h);
    hipStream_t st = (hipStream_t)stream;
    const bool one_launch = !h->avi && !h->useWave && h->asyncPhase == 0 && h->L.sim.FG == nullptr && h->L.gat.state == nullptr &&
                            will_screen(h, N) && fast_covers(h) && !h->prof;
    int done = 0;
    while (done < n_batches) {
        const int nb = n_batches - done < 8 ? n_batches - done : 8;
        int rc = LMPC_ERR_UNSUPPORTED;
        if (one_launch && nb > 1) {
            rc = check_fast_err(h);
            if (rc == LMPC_OK) rc = launch_fast_multi(h, nb, N, theta + done, x + done, exitflag + done, st);
        }
        if (rc == LMPC_ERR_UNSUPPORTED) {
            for (int b = 0; b < nb; b++) {
                rc = launch(h, N, theta[done + b], x[done + b], exitflag[done + b], nullptr, nullptr, nullptr, st);
                if (rc != LMPC_OK) return rc;
            }
        } else if (rc != LMPC_OK) {
            return rc;
        }
        done += nb;
    }
    return LMPC_OK;
}

int lmpc_solve_batch_f32_device(lmpc_handle *h, int64_t N, const float *theta, float *x, int32_t *exitflag,
                                int32_t *iters, uint64_t *active, const uint64_t *warm, void *stream) {
    if (!h) return LMPC_ERR_BADARG;
    if (N < 0 || (N > 0 && (!x || !exitflag || (h->P.nth > 0 && !theta))))
        return fail(h, LMPC_ERR_BADARG, "lmpc_solve_batch_f32_device: NULL array or negative N");
    if (N == 0) return LMPC_OK;
    if (N > (int64_t)0x7fffffff * 64) return fail(h, LMPC_ERR_BADARG, "lmpc: batch too large for one launch");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(h, LMPC_ERR_NOGPU, "lmpc: no HIP device available (this library has no CPU path)");
    LMPC_ENTER_DEVICE(h);
    int rc = ensure_f32(h);
    if (rc != LMPC_OK) return rc;
    return launch_wave_t<float>(h, h->dCwf, N, theta, x, exitflag, iters, active, warm, (hipStream_t)stream);
}

// ONE theta, as the reference's online call has it (utils.jl:268-283; compute_control in a Simulation loop): no staging
// copies and no pipeline -- theta is written into a record of MAPPED host memory the kernels read directly, x and the
// exit flag are written back into it by the kernels, the host waits for the handle's own stream once.  (Through
// lmpc_solve_batch a single problem paid three pageable copies on three streams: 70 us against ~20 here.)
int lmpc_solve_one(lmpc_handle *h, const double *theta, double *x) {
    if (!h) return LMPC_ERR_BADARG;
    const size_t nth = (size_t)h->P.nth, nout = (size_t)h->P.nout;
    if (!x || (nth > 0 && !theta)) return fail(h, LMPC_ERR_BADARG, "lmpc_solve_one: NULL array");
    const size_t oX = (sizeof(double) * nth + 63) & ~(size_t)63, oF = oX + ((sizeof(double) * nout + 63) & ~(size_t)63);
    if (!h->oneHost) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            return fail(h, LMPC_ERR_NOGPU, "lmpc: no HIP device available (this library has no CPU path)");
    }
    LMPC_ENTER_DEVICE(h);
    if (!h->oneHost) {
        char *hp = nullptr;
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void **>(&hp), oF + 4096, hipHostMallocMapped));
        std::memset(hp, 0, oF + 4096);
        if (hipHostGetDevicePointer(reinterpret_cast<void **>(&h->oneDev), hp, 0) != hipSuccess ||
            hipStreamCreateWithFlags(&h->oneStream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipHostFree(hp); h->oneDev = nullptr; h->oneStream = nullptr;
            return fail(h, LMPC_ERR_HIP, "lmpc_solve_one: mapped record / stream");
        }
        h->oneHost = hp;
    }
    if (nth) std::memcpy(h->oneHost, theta, sizeof(double) * nth);
    int32_t *hflag = reinterpret_cast<int32_t *>(h->oneHost + oF);
    *hflag = LMPC_EXIT_UNFINISHED;
    const int rc = launch(h, 1, reinterpret_cast<const double *>(h->oneDev), reinterpret_cast<double *>(h->oneDev + oX),
                          reinterpret_cast<int32_t *>(h->oneDev + oF), nullptr, nullptr, nullptr, h->oneStream);
    const hipError_t es = hipStreamSynchronize(h->oneStream);      // (also after a failed launch: nothing may still run)
    if (rc != LMPC_OK) return rc;
    if (es != hipSuccess) return fail(h, LMPC_ERR_HIP, std::string("lmpc_solve_one: ") + hipGetErrorString(es));
    const int rcf = check_fast_err(h);
    if (rcf != LMPC_OK) return rcf;
    std::memcpy(x, h->oneHost + oX, sizeof(double) * nout);
    return *hflag;
}

int lmpc_wave_stats(lmpc_handle *h, unsigned long long out[5]) {
    if (!h || !out) return LMPC_ERR_BADARG;
    wave_stat_read(h, out);
    out[4] = (unsigned long long)wave_first_pass_cap(h, (int64_t)1 << 20);
    return LMPC_OK;
}

const char *lmpc_kernel_name(const lmpc_handle *h) {
    if (!h) return "";
    if (h->avi) return h->kname.c_str();          // "avi" / "avi+prox"
    if (h->useWave) {
        // (small problems with many rows: cold plain binary64 batches pass through the tiers kernel first)
        if (h->qpTiersOk && h->qpTiers && !h->waveGram && !h->bnb) {
            thread_local std::string wname;
            wname = "qp_tiers<" + std::to_string(h->P.n) + ">|wave";
            return wname.c_str();
        }
        // (large cold plain binary64 batches: four problems per wavefront where that kernel is the default)
        if (row_pass_cap(const_cast<lmpc_handle *>(h), (int64_t)1 << 20, sizeof(double), false, h->waveGram != 0, h->bnb) > 0) return "row|wave";
        if (h->bnb && row_bnb_pass_cap(const_cast<lmpc_handle *>(h), (int64_t)1 << 20, sizeof(double)) > 0) return "row|wave";
        return "wave";
    }
    // small boxed problems: cold plain batches take the one-launch kernel, everything else on the handle (warm
    // starts, closed loop, generated-controller call) the two-kernel form
    if (fast_covers(h)) {
        thread_local std::string name;
        name = "fast<" + std::to_string(h->laneN) + ">|" + h->kname;
        return name.c_str();
    }
    return h->kname.c_str();
}

int lmpc_profile(lmpc_handle *h, int enable) {
    if (!h) return LMPC_ERR_BADARG;
    h->prof = enable != 0;
    return LMPC_OK;
}

int lmpc_profile_read(lmpc_handle *h, double avg_ms[3]) {
    if (!h || !avg_ms) return LMPC_ERR_BADARG;
    double tot = 0.0, scr = 0.0, itr = 0.0;
    int cnt = 0;
    for (auto &ev : h->events) {
        float ms = 0.f, ms1 = 0.f, ms2 = 0.f;
        if (hipEventSynchronize(ev.b) == hipSuccess && hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess &&
            (ev.mid == nullptr || (hipEventElapsedTime(&ms1, ev.a, ev.mid) == hipSuccess &&
                                   hipEventElapsedTime(&ms2, ev.mid, ev.b) == hipSuccess))) {
            if (ev.mid == nullptr) ms1 = ms;             // single-kernel call (lmpc_fast_kernel.hpp)
            tot += ms; scr += ms1; itr += ms2;
            cnt++;
        }
        h->eventPool.push_back(ev.a);
        if (ev.mid) h->eventPool.push_back(ev.mid);
        h->eventPool.push_back(ev.b);
    }
    h->events.clear();
    avg_ms[0] = cnt ? tot / cnt : 0.0;
    avg_ms[1] = cnt ? scr / cnt : 0.0;
    avg_ms[2] = cnt ? itr / cnt : 0.0;
    if (check_fast_err(h) != LMPC_OK) return LMPC_ERR_HIP;     // (the timed calls have completed: their error word is final)
    return cnt;
}

int lmpc_set_option(lmpc_handle *h, const char *name, int value) {
    if (!h || !name) return LMPC_ERR_BADARG;
    if (std::strcmp(name, "screen") == 0) { h->screen = value != 0; return LMPC_OK; }
    if (std::strcmp(name, "screen_wave") == 0) { h->screenWave = value != 0; return LMPC_OK; }
    if (std::strcmp(name, "big_path") == 0) { h->bigPath = value != 0; return LMPC_OK; }
    if (std::strcmp(name, "lane_per") == 0) { h->lanePer = value; return LMPC_OK; }
    if (std::strcmp(name, "host_chunk") == 0) { h->hostChunk = value < 1024 ? 1024 : value; return LMPC_OK; }
    if (std::strcmp(name, "host_register") == 0) {
        // (removed in round 5: pinning ordinary heap memory call after call ended in a GPU memory fault on this runtime
        // after a few hundred calls; lmpc_pin_host on long-lived arrays is the supported way)
        if (value == 0) return LMPC_OK;
        return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc_set_option: \"host_register\" 1 was removed (it faulted after a few hundred calls); "
                                             "pin long-lived arrays with lmpc_pin_host instead");
    }
    if (std::strcmp(name, "host_threads") == 0) { h->hostThreads = value ? 1 : 0; return LMPC_OK; }
    if (std::strcmp(name, "gram_scan") == 0) { h->waveGram = value != 0; return LMPC_OK; }
    if (std::strcmp(name, "wave_two_pass") == 0) { h->waveTwoPass = value < 0 ? -1 : (value != 0); return LMPC_OK; }
    if (std::strcmp(name, "wave_cap1") == 0) { h->waveCap1 = value <= 0 ? 0 : (value < 8 ? 8 : (value > 64 ? 64 : value)); return LMPC_OK; }
    if (std::strcmp(name, "wave_cap") == 0) {
        // working-set rows the wavefront kernel holds per problem (8 .. 64, at most n + 1 + #soft): a smaller factor in
        // LDS keeps more wavefronts resident; a point that wants more goes to the slow path
        const int full = std::min(h->P.n + 1 + h->P.nsoft + (h->bnb ? 1 : 0), kWaveMaxCap);      // (the capacity of setup)
        int c = value <= 0 ? full : (value < 8 ? 8 : value);
        if (c > full) c = full;
        h->W.cap = c; h->W.ldc = c | 1;
        h->W.keepStride = 2 * 64 + c * (c - 1) / 2;
        hipFree(h->dKeepR); hipFree(h->dKeepI); h->dKeepR = nullptr; h->dKeepI = nullptr; h->keepCap = 0;
        return LMPC_OK;
    }
    if (std::strcmp(name, "wave_packed") == 0) { h->wavePacked = value < 0 ? -1 : (value ? 1 : 0); return LMPC_OK; }
    if (std::strcmp(name, "wave_queue") == 0) { h->waveQueue = value != 0; return LMPC_OK; }
    if (std::strcmp(name, "wave_level") == 0) { h->waveLevel = value > 3 ? 3 : value; return LMPC_OK; }
    if (std::strcmp(name, "wave_nwv") == 0) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16)
            return fail(h, LMPC_ERR_BADARG, "lmpc_set_option: wave_nwv must be 0, 1, 2, 4, 8 or 16");
        h->waveNwv = value;
        return LMPC_OK;
    }
    // (wavefronts per CU of the wavefront kernel's grid; "wave_cap" above is the working-set CAPACITY -- until round 4
    // both answered to that name and this one was unreachable)
    if (std::strcmp(name, "wave_waves") == 0) { h->waveCap = value < 0 ? 0 : value; return LMPC_OK; }
    if (std::strcmp(name, "lane_tier") == 0) { h->laneTier = value ? 1 : 0; return LMPC_OK; }
    if (std::strcmp(name, "fast") == 0) { h->fastPath = value ? 1 : 0; return LMPC_OK; }
    if (std::strcmp(name, "lane_straight") == 0) { h->laneStraight = value ? 1 : 0; return LMPC_OK; }
    if (std::strcmp(name, "fast_tiles") == 0) { h->fastTiles = value < 0 ? 0 : (value > 256 ? 256 : value); return LMPC_OK; }
    if (std::strcmp(name, "fast_nstr") == 0) { h->fastNstr = value; return LMPC_OK; }
    if (std::strcmp(name, "fast_dma") == 0) { h->fastDma = value; return LMPC_OK; }
    if (std::strcmp(name, "fast_dyn") == 0) { h->fastDyn = value; return LMPC_OK; }
    if (std::strcmp(name, "in_flight") == 0) {
        // hint: how many independent batches the caller keeps in flight on this GPU (one handle and stream each).
        // From two on a call no longer owns the chip: the one-launch kernel then runs with all four wavefronts of
        // a workgroup streaming and 28 tiles per workgroup, the iterating kernel with 64-lane workgroups
        // (same-box sweeps, three 1e6-point batches in flight: 17.1 us/step at the stand-alone shape, 15.0 with this)
        h->fastNstr = value >= 2 ? 4 : 0;
        h->fastTiles = value >= 2 ? 28 : 0;
        h->laneBlock = value >= 2 ? 64 : 0;
        return LMPC_OK;
    }
    if (std::strcmp(name, "fast_spin_limit") == 0) { h->fastSpinLimit = value < 0 ? 0 : value; return LMPC_OK; }
    if (std::strcmp(name, "sim_fused") == 0) { h->simFused = value ? 1 : 0; return LMPC_OK; }
    if (std::strcmp(name, "cc_fused") == 0) { h->ccFused = value ? 1 : 0; return LMPC_OK; }
    if (std::strcmp(name, "sim_small") == 0) { h->simSmall = value ? 1 : 0; return LMPC_OK; }
    if (std::strcmp(name, "sim_blind") == 0) { h->simBlind = value < 0 ? 0 : value; return LMPC_OK; }
    if (std::strcmp(name, "sim_run_ahead") == 0) { h->simRunAhead = value != 0; return LMPC_OK; }
    if (std::strcmp(name, "sim_keep_factor") == 0) { h->simKeep = value != 0; return LMPC_OK; }
    if (std::strcmp(name, "sim_async") == 0) { h->simAsync = value < 0 ? 0 : (value > 2 ? 2 : value); return LMPC_OK; }
    if (std::strcmp(name, "lane_block") == 0) {
        if (value != 0 && value != 64 && value != 128 && value != 256)
            return fail(h, LMPC_ERR_BADARG, "lmpc_set_option: lane_block must be 0, 64, 128 or 256");
        h->laneBlock = value;
        return LMPC_OK;
    }
    if (std::strcmp(name, "wave_probe") == 0) { h->waveProbe = value != 0; return LMPC_OK; }
    if (std::strcmp(name, "row_kernel") == 0) { h->rowKernel = value < 0 ? -1 : (value != 0); return LMPC_OK; }
    if (std::strcmp(name, "row_blocks") == 0) { h->rowBlocks = value < 0 ? 0 : (value > 8 ? 8 : value); return LMPC_OK; }
    if (std::strcmp(name, "region_lockfree") == 0) { h->regW1 = value != 0; return LMPC_OK; }
    if (std::strcmp(name, "region_blocks") == 0) { h->regBlocks = value < 0 ? 0 : (value > 16 ? 16 : value); return LMPC_OK; }
    if (std::strcmp(name, "avi_waves") == 0) { h->aviWaves = value < 0 ? 0 : (value > 32 ? 32 : value); return LMPC_OK; }
    if (std::strcmp(name, "qp_tiers") == 0) {
        h->qpTiers = value < 0 ? 0 : (value > 2 ? 2 : value);
        h->qpAbCalls = 0; h->qpAbNsPer[0] = h->qpAbNsPer[1] = -1.0; h->qpAbPending[0] = h->qpAbPending[1] = false; h->qpAbCnt[0] = h->qpAbCnt[1] = 0;   // (measure again)
        return LMPC_OK;
    }
    if (std::strcmp(name, "avi_tiers") == 0) { h->aviTiers = value != 0; return LMPC_OK; }
    if (std::strcmp(name, "avi_tiers_first") == 0) {
        h->aviTiersFirst = value < 0 ? -1 : (value > 3 ? 3 : value); h->aviTiersOcc[0] = 0; return LMPC_OK;
    }
    if (std::strcmp(name, "wave") == 0) {
        if (h->avi) return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: a variational-inequality handle has one kernel");
        if (value && !h->dCw) return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: wavefront kernel does not cover this problem");
        if (!value && (h->laneN == 0 || h->bnb))
            return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: lane kernel does not cover this problem");
        h->useWave = value != 0;
        h->qpAbCalls = 0; h->qpAbNsPer[0] = h->qpAbNsPer[1] = -1.0; h->qpAbPending[0] = h->qpAbPending[1] = false; h->qpAbCnt[0] = h->qpAbCnt[1] = 0;   // (another path: measured again)
        return LMPC_OK;
    }
    return fail(h, LMPC_ERR_BADARG, std::string("lmpc_set_option: unknown option ") + name);
}

int lmpc_reserve(lmpc_handle *h, int64_t N, void *stream) {
    if (!h || N < 0) return LMPC_ERR_BADARG;
    if (N == 0) return LMPC_OK;
    LMPC_ENTER_DEVICE(h);
    hipStream_t st = (hipStream_t)stream;
    if (h->avi) return avi_reserve(h, N, st);
    if (h->useWave) {
        if (wave_screens(h, N)) { const int rc = ensure_lists(h, N, st); if (rc != LMPC_OK) return rc; }
        const int rcw = wave_reserve(h, N, st);
        if (rcw != LMPC_OK || h->bnb || h->waveWarmed) return rcw;
        // ONE dummy problem (theta = 0) through the wavefront kernel on the caller's stream, into scratch outputs: what a
        // kernel's first dispatch on a queue still costs after the code is loaded (4 ms measured on pendulum N = 50:
        // the queue's scratch memory for the kernel's spills) is paid here and not in the first solve
        h->waveWarmed = true;
        double *pt = nullptr;
        int32_t *pf = nullptr;
        const size_t nr = (size_t)h->P.nth + (size_t)h->P.nout + 2;
        if (hipMalloc(&pt, sizeof(double) * nr) != hipSuccess) { (void)hipGetLastError(); return LMPC_OK; }
        if (hipMalloc(&pf, sizeof(int32_t) * 4) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(pt); return LMPC_OK; }
        int rc = LMPC_OK;
        if (hipMemsetAsync(pt, 0, sizeof(double) * nr, st) == hipSuccess) {
            const bool prof = h->prof, probe = h->waveProbe;
            h->prof = false; h->waveProbe = false;             // (not a batch: no events, no probe)
            rc = launch_wave(h, 1, pt, pt + h->P.nth, pf, nullptr, nullptr, nullptr, st);
            h->prof = prof; h->waveProbe = probe;
            (void)hipStreamSynchronize(st);
        }
        (void)hipGetLastError();
        (void)hipFree(pt); (void)hipFree(pf);
        return rc;
    }
    if (will_screen(h, N)) return ensure_lists(h, N, st);
    return LMPC_OK;
}

int lmpc_release_scratch(lmpc_handle *h) {
    if (!h) return LMPC_ERR_BADARG;
    lmpc::DeviceScope scope;
    if (scope.enter(h->device) != hipSuccess) return fail(h, LMPC_ERR_HIP, "lmpc_release_scratch: hipSetDevice");
    (void)hipDeviceSynchronize();
    auto rel = [](auto *&p) { if (p) { (void)hipFree(p); p = nullptr; } };
    rel(h->sTheta); rel(h->sX); rel(h->sFlag); rel(h->sIter); rel(h->sAct); rel(h->sWarm); h->sCap = 0;
    rel(h->dList); rel(h->dList2); rel(h->dList3); rel(h->dCount); h->listCap = 0; h->countSet = 0;
    rel(h->simTheta); rel(h->simTheta2); rel(h->simU); rel(h->simFG); rel(h->simFlag); rel(h->simAct); rel(h->simK); h->simCap = 0;
    rel(h->ccTheta); rel(h->ccAct); rel(h->ccFlag); h->ccCap = 0; h->ccWarmN = -1;
    rel(h->ccStage); rel(h->ccStageFlag); h->ccStageCap = 0; h->ccStagePer = 0;
    rel(h->ccObsScratch); h->ccObsCap = 0;
    rel(h->dOvfList); h->ovfCap = 0; rel(h->dOvfList1); h->ovfCap1 = 0; rel(h->dBigR); rel(h->dBigI);
    rel(h->dBnbR); rel(h->dBnbI); h->bnbBytesR = h->bnbBytesI = 0;
    rel(h->dRowBnb); h->rowBnbBytes = 0;
    rel(h->dKeepR); rel(h->dKeepI); h->keepCap = 0;
    avi_release(h, false);
    return check_fast_err(h);
}

int lmpc_check(lmpc_handle *h) {
    if (!h) return LMPC_ERR_BADARG;
    lmpc::DeviceScope scope;
    if (scope.enter(h->device) != hipSuccess) return fail(h, LMPC_ERR_HIP, "lmpc_check: hipSetDevice");
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(h, LMPC_ERR_HIP, std::string("lmpc_check: ") + hipGetErrorString(e));
    return check_fast_err(h);
}

void lmpc_free(lmpc_handle *h) {
    if (!h) return;
    lmpc::DeviceScope scope;
    if (h->dC || h->dCw || h->dCa || h->sTheta) scope.enter(h->device);
    avi_release(h, true);
    for (auto &ev : h->events) { hipEventDestroy(ev.a); if (ev.mid) hipEventDestroy(ev.mid); hipEventDestroy(ev.b); }
    for (auto &e : h->eventPool) hipEventDestroy(e);
    hipFree(h->dC); hipFree(h->sTheta); hipFree(h->sX); hipFree(h->sFlag); hipFree(h->sIter);
    hipFree(h->sAct); hipFree(h->sWarm); hipFree(h->dList); hipFree(h->dList2); hipFree(h->dList3); hipFree(h->dCount); hipFree(h->dCw); hipFree(h->dCwf); hipFree(h->dSw); hipFree(h->dQueue);
    hipFree(h->dOvfList); hipFree(h->dOvfCount); hipFree(h->dBigR); hipFree(h->dBigI); hipFree(h->dRegTable); hipFree(h->dRegW1); hipFree(h->dQpScan); hipFree(h->dFastCtr);
    hipFree(h->dRowBnb);
    hipFree(h->dBnbR); hipFree(h->dBnbI); hipFree(h->dKeepR); hipFree(h->dKeepI); hipFree(h->dOvfList1);
    if (h->hStat) hipHostFree(const_cast<unsigned long long *>(h->hStat));
    if (h->hRegOut) hipHostFree(h->hRegOut);
    for (int v = 0; v < 2; v++) for (int e = 0; e < 2; e++) if (h->qpAbEv[v][e]) hipEventDestroy(h->qpAbEv[v][e]);
    if (h->oneHost) hipHostFree(h->oneHost);
    if (h->ccMapHost) hipHostFree(h->ccMapHost);
    if (h->oneStream) hipStreamDestroy(h->oneStream);
    hipFree(h->dStat);
    hipFree(h->simTheta); hipFree(h->simTheta2); hipFree(h->simU); hipFree(h->simFG); hipFree(h->simFlag); hipFree(h->simAct); hipFree(h->simK);
    hipFree(h->ccT2S); hipFree(h->ccTheta); hipFree(h->ccAct); hipFree(h->ccFlag); hipFree(h->obsC);
    hipFree(h->ccStage); hipFree(h->ccStageFlag); hipFree(h->ccObsScratch);
    if (h->hFastErr) {
        if (*h->hFastErr != 0) std::fprintf(stderr, "lmpc_free: unreported error word %d of the one-launch kernel\n", (int)*h->hFastErr);
        hipHostFree(const_cast<int32_t *>(h->hFastErr));
    } else hipFree(h->dFastErr);
    for (auto &e : h->pipeEv) hipEventDestroy(e);
    if (h->sUp) hipStreamDestroy(h->sUp);
    if (h->sRun) hipStreamDestroy(h->sRun);
    if (h->sDown) hipStreamDestroy(h->sDown);
    delete h;
}

const char *lmpc_last_error(const lmpc_handle *h) { return h ? h->err.c_str() : g_setup_err.c_str(); }

}  // extern "C"
