// Affine-variational-inequality mode of a handle (non-symmetric H; reference setup.jl:11-13 is_avi): constant pack,
// launch of avi_kernel (lmpc_avi_kernel.hpp), scratch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "lmpc_avi_kernel.hpp"
#include "lmpc_internal.hpp"

namespace lmpc {

constexpr int kAviMaxN = 128, kAviMaxM = 1024, kAviMaxCap = 160;
constexpr size_t kAviLdsPack = 64 * 1024;      // the constant pack is staged in LDS up to this size

void avi_fill_settings(lmpc_handle *h) {
    AviLayout &A = h->A;
    A.primal_tol = h->S.primal_tol; A.dual_tol = h->S.dual_tol; A.zero_tol = h->S.zero_tol; A.rho_soft = h->S.rho_soft;
    A.iter_limit = h->S.iter_limit;
    A.eps_prox = h->P.prox ? h->P.eps_prox : 0.0;      // (fixed at setup: the pack was factorised with it)
    A.eta_prox = h->S.eta_prox > 0.0 ? h->S.eta_prox : 1e-6;
}

int finalize_avi(lmpc_handle *h) {
    const HostPack &P = h->P;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(h, LMPC_ERR_NOGPU, "lmpc: no HIP device available (this library has no CPU path)");
    if (h->device < 0 || h->device >= ndev) return fail(h, LMPC_ERR_BADARG, "lmpc: bad device ordinal");
    const int cap = P.n + 1 + P.nsoft;
    if (P.n > kAviMaxN || P.m > kAviMaxM || P.m < 1 || cap > kAviMaxCap)
        return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: variational objective: problem outside what the kernel covers "
                                             "(n <= 128, 1 <= m <= 1024, n + 1 + #soft <= 160)");
    h->avi = true;
    h->useWave = true;                 // keeps every lane / screening shortcut of the QP kernels off this handle
    h->capFull = cap;
    h->kname = P.prox ? "avi+prox" : "avi";
    LMPC_ENTER_DEVICE(h);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0)
            h->numCU = prop.multiProcessorCount;
    }
    AviLayout &A = h->A;
    A.n = P.n; A.m = P.m; A.nth = P.nth; A.nout = P.nout; A.words = P.words(); A.cap = cap;
    int o = 0;
    A.oML = o; o += P.m * P.n;
    A.oMR = o; o += P.m * P.n;
    A.oG = o; o += P.m * P.m;
    A.odu = o; o += P.m;
    A.odl = o; o += P.m;
    A.oDth = o; o += P.m * P.nth;
    A.oRout = o; o += P.nout * P.n;
    A.ox0 = o; o += P.nout;
    A.oXth = o; o += P.nout * P.nth;
    A.oHinv = A.ox0f = A.oXthf = A.oKth = 0;
    if (P.prox) {
        A.oHinv = o; o += P.n * P.n;
        A.ox0f = o; o += P.n;
        A.oXthf = o; o += P.n * P.nth;
        A.oKth = o; o += P.nout * P.nth;
    }
    const bool small = P.m == P.n && P.n <= 8 && P.nout <= P.n;
    A.oTh2 = A.oBnd3 = A.oSd = 0;
    if (small) {
        o = (o + 3) & ~3;                       // (batched scalar loads: 32-byte aligned starts)
        A.oTh2 = o; o += P.nth * 2 * P.n;
        o = (o + 3) & ~3;
        A.oBnd3 = o; o += 3 * P.n;
        o = (o + 3) & ~3;
        A.oSd = o; o += P.n;
    }
    A.nC = o;
    avi_fill_settings(h);
    std::vector<double> buf((size_t)o, 0.0);
    std::memcpy(&buf[A.oML], P.M.data(), sizeof(double) * P.M.size());
    std::memcpy(&buf[A.oMR], P.MR.data(), sizeof(double) * P.MR.size());
    std::memcpy(&buf[A.oG], P.Gf.data(), sizeof(double) * P.Gf.size());
    std::memcpy(&buf[A.odu], P.du0.data(), sizeof(double) * P.m);
    std::memcpy(&buf[A.odl], P.dl0.data(), sizeof(double) * P.m);
    if (P.m * P.nth) std::memcpy(&buf[A.oDth], P.Dth.data(), sizeof(double) * P.Dth.size());
    std::memcpy(&buf[A.oRout], P.Rout.data(), sizeof(double) * P.Rout.size());
    std::memcpy(&buf[A.ox0], P.x0.data(), sizeof(double) * P.x0.size());
    if (P.nout * P.nth) std::memcpy(&buf[A.oXth], P.Xth.data(), sizeof(double) * P.Xth.size());
    if (small) {
        const int n = P.n;
        for (int t = 0; t < P.nth; t++) {
            for (int j = 0; j < n; j++) buf[A.oTh2 + (size_t)t * 2 * n + j] = P.Dth[(size_t)j * P.nth + t];
            for (int k = 0; k < P.nout; k++) buf[A.oTh2 + (size_t)t * 2 * n + n + k] = P.Xth[(size_t)k * P.nth + t];
        }
        for (int j = 0; j < n; j++) {
            buf[A.oBnd3 + j] = P.du0[j];
            buf[A.oBnd3 + n + j] = P.dl0[j];
            buf[A.oBnd3 + 2 * n + j] = j < P.nout ? P.x0[j] : 0.0;
            buf[A.oSd + j] = P.M[(size_t)j * n + j];
        }
    }
    if (P.prox) {
        std::memcpy(&buf[A.oHinv], P.Hinv.data(), sizeof(double) * P.Hinv.size());
        std::memcpy(&buf[A.ox0f], P.x0f.data(), sizeof(double) * P.x0f.size());
        if (P.n * P.nth) std::memcpy(&buf[A.oXthf], P.Xthf.data(), sizeof(double) * P.Xthf.size());
        if (P.nout * P.nth) std::memcpy(&buf[A.oKth], P.Kth.data(), sizeof(double) * P.Kth.size());
    }
    HIP_TRY(h, hipMalloc(&h->dCa, sizeof(double) * buf.size()));
    HIP_TRY(h, hipMemcpy(h->dCa, buf.data(), sizeof(double) * buf.size(), hipMemcpyHostToDevice));
    // small box-constrained problem: the register-resident kernels in front of the generic one
    // (lmpc_avi_tiers_kernel.hpp: bounds only, rows of ML = s_j e_j', no row flags, no proximal-point iterations)
    h->aviTiersN = 0;
    if (!P.prox && P.m == P.n && P.ms == P.m && P.n >= 2 && P.n <= 8 && P.nth >= 1 && P.nout <= P.n && P.words() == 1) {
        bool ok = true;
        for (int j = 0; j < P.m && ok; j++) {
            if (P.sense[j] != 0) ok = false;
            for (int c = 0; c < P.n && ok; c++)
                if (c != j && P.M[(size_t)j * P.n + c] != 0.0) ok = false;
        }
        if (ok) { h->aviTiersN = P.n; h->kname = "avi_tiers<" + std::to_string(P.n) + ">|avi"; }
    }
    HIP_TRY(h, hipMalloc(&h->dSa, sizeof(int32_t) * P.m));
    HIP_TRY(h, hipMemcpy(h->dSa, P.sense.data(), sizeof(int32_t) * P.m, hipMemcpyHostToDevice));
    return LMPC_OK;
}

// one segment of a work list of the tiers chain: wavefront w of a pass queues into segment w % kShards and takes the
// tiles w, w + #wavefronts, ... (#wavefronts a multiple of kShards) -- an even share of the tiles, rounded up
static long long avi_seg_cap(long long nprob) {
    const long long tiles = (nprob + 63) / 64;
    return ((tiles + kShards - 1) / kShards) * 64 + 64;
}

static bool avi_use_tiers(const lmpc_handle *h, int64_t nprob, const uint64_t *warm) {
    return h->aviTiers && h->aviTiersN > 0 && warm == nullptr && h->S.iter_limit > h->aviTiersN + 1 &&
           nprob < (int64_t)0x7fffffff;
}

static int avi_ensure_lists(lmpc_handle *h, int64_t nprob, hipStream_t st) {
    if (nprob <= h->aviListCap) return LMPC_OK;
    if (h->dAviCnt) (void)hipStreamSynchronize(st);
    hipFree(h->dAviList[0]); hipFree(h->dAviList[1]); hipFree(h->dAviCnt);
    h->dAviList[0] = h->dAviList[1] = h->dAviCnt = nullptr; h->aviListCap = 0;
    const size_t seg = (size_t)avi_seg_cap(nprob);
    HIP_TRY(h, hipMalloc(&h->dAviList[0], sizeof(int32_t) * seg * kShards));
    HIP_TRY(h, hipMalloc(&h->dAviList[1], sizeof(int32_t) * seg * kShards));
    HIP_TRY(h, hipMalloc(&h->dAviCnt, sizeof(int32_t) * 2 * kShards * kCountStride));
    HIP_TRY(h, hipMemsetAsync(h->dAviCnt, 0, sizeof(int32_t) * 2 * kShards * kCountStride, st));
    h->aviListCap = nprob;
    return LMPC_OK;
}

// the generic kernel's grid for `tiles` tiles of 64 problems -- one wavefront per workgroup, resident wavefronts bounded
// by the scratch a slab takes (at most 1 GiB in all), a multiple of kShards behind a work list -- and its slabs
static int avi_ensure_slabs(lmpc_handle *h, long long tiles, bool listed, hipStream_t st, long long *grid_out) {
    const AviLayout &A = h->A;
    const size_t slabR = (size_t)avi_scratch_reals(A.n, A.m, A.cap) * 64, slabI = (size_t)avi_scratch_ints(A.m, A.cap) * 64;
    long long grid = (long long)h->numCU * (h->aviWaves > 0 ? h->aviWaves : 16);
    const long long fit = (long long)(((size_t)1 << 30) / (sizeof(double) * slabR + sizeof(int32_t) * slabI));
    grid = std::min(grid, std::max(1ll, fit));
    grid = std::min(grid, tiles);
    if (listed) grid = std::max<long long>(kShards, (grid / kShards) * kShards);      // (fit >= kShards: slabs of small problems)
    if (grid > h->aviSlabs) {
        // (grows with the largest batch seen; stream-ordered work of earlier calls on this handle finishes first)
        if (h->dAviR || h->dAviI) { (void)hipStreamSynchronize(st); hipFree(h->dAviR); hipFree(h->dAviI); }
        h->dAviR = nullptr; h->dAviI = nullptr; h->aviSlabs = 0;
        HIP_TRY(h, hipMalloc(&h->dAviR, sizeof(double) * slabR * (size_t)grid));
        HIP_TRY(h, hipMalloc(&h->dAviI, sizeof(int32_t) * slabI * (size_t)grid));
        h->aviSlabs = (int)grid;
    }
    *grid_out = grid;
    return LMPC_OK;
}

// lmpc_reserve on a variational handle: what the first call on a batch of nprob problems would allocate
int avi_reserve(lmpc_handle *h, int64_t nprob, hipStream_t st) {
    const long long tiles = (nprob + 63) / 64;
    long long grid = 0;
    if (avi_use_tiers(h, nprob, nullptr)) {
        const int rc = avi_ensure_lists(h, nprob, st);
        if (rc != LMPC_OK) return rc;
        return avi_ensure_slabs(h, std::min<long long>(tiles, (long long)h->numCU * 4), true, st, &grid);
    }
    return avi_ensure_slabs(h, tiles, false, st, &grid);
}

static int avi_generic(lmpc_handle *h, int64_t nprob, long long max_tiles, const double *theta, double *x, int32_t *flag,
                       int32_t *iters, uint64_t *active, const uint64_t *warm, const int32_t *list, const int32_t *count,
                       long long seg_cap, int32_t *count_clear, hipStream_t st);

int launch_avi(lmpc_handle *h, int64_t nprob, const double *theta, double *x, int32_t *flag, int32_t *iters,
               uint64_t *active, const uint64_t *warm, hipStream_t st) {
    if (!x || !flag) return fail(h, LMPC_ERR_BADARG, "lmpc: the variational-inequality kernel needs x and exitflag arrays");
    const long long tiles = (nprob + 63) / 64;
    if (!avi_use_tiers(h, nprob, warm))
        return avi_generic(h, nprob, tiles, theta, x, flag, iters, active, warm, nullptr, nullptr, 0, nullptr, st);
    // chain: tiers over the whole batch -> all n tiers on its list -> the generic kernel on that one's list
    const int rc0 = avi_ensure_lists(h, nprob, st);
    if (rc0 != LMPC_OK) return rc0;
    // tiers of the first pass: three finish 93 % of the reference's game problem at 3 wavefronts per SIMD (n <= 6)
    const int kfirst = h->aviTiersFirst >= 0 ? h->aviTiersFirst : (h->aviTiersN <= 6 ? 3 : 2);
    for (int s = 0; s < 2; s++)
        if (h->aviTiersOcc[s] == 0) {
            const int rco = launch_avi_tiers(h, s == 0, kfirst, 0, st, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                             nullptr, nullptr, nullptr, nullptr, 0, 0, &h->aviTiersOcc[s]);
            if (rco != LMPC_OK) return rco;
        }
    const long long seg = avi_seg_cap(h->aviListCap);
    int32_t *cnt0 = h->dAviCnt, *cnt1 = h->dAviCnt + (size_t)kShards * kCountStride;
    // workgroups of four wavefronts, a multiple of 16 (wavefronts a multiple of kShards), one resident round at most
    auto grid_for = [&](int occ) {
        long long g = (long long)h->numCU * occ;
        const long long need = (tiles + 3) / 4;
        if (g > need) g = need;
        g = ((g + 15) / 16) * 16;
        return (unsigned)g;
    };
    int rc = launch_avi_tiers(h, true, kfirst, grid_for(h->aviTiersOcc[0]), st, theta, x, flag, iters, active, nullptr,
                              nullptr, h->dAviList[0], cnt0, cnt1, seg, (long long)nprob, nullptr);
    if (rc != LMPC_OK) return rc;
    rc = launch_avi_tiers(h, false, 0, grid_for(h->aviTiersOcc[1]), st, theta, x, flag, iters, active, h->dAviList[0], cnt0,
                          h->dAviList[1], cnt1, nullptr, seg, (long long)nprob, nullptr);
    if (rc != LMPC_OK) return rc;
    return avi_generic(h, nprob, std::min<long long>(tiles, (long long)h->numCU * 4), theta, x, flag, iters, active, nullptr,
                       h->dAviList[1], cnt1, seg, cnt0, st);
}

// the generic kernel over the whole batch (list == nullptr) or over a work list of the tiers chain
static int avi_generic(lmpc_handle *h, int64_t nprob, long long tiles, const double *theta, double *x, int32_t *flag,
                       int32_t *iters, uint64_t *active, const uint64_t *warm, const int32_t *list, const int32_t *count,
                       long long seg_cap, int32_t *count_clear, hipStream_t st) {
    const AviLayout &A = h->A;
    long long grid = 0;
    { const int rcs = avi_ensure_slabs(h, tiles, list != nullptr, st, &grid); if (rcs != LMPC_OK) return rcs; }
    const size_t packBytes = sizeof(double) * (size_t)A.nC;
#define LMPC_AVI_GO(PK, PX, LDSB)                                                                                          \
    do {                                                                                                                  \
        if ((LDSB) > 48 * 1024)                                                                                           \
            HIP_TRY(h, hipFuncSetAttribute((const void *)avi_kernel<PK, PX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDSB))); \
        hipLaunchKernelGGL((avi_kernel<PK, PX>), dim3((unsigned)grid), dim3(64), (LDSB), st, A, h->dCa, h->dSa, theta, x, flag, \
                           iters, active, warm, h->dAviR, h->dAviI, (long long)nprob, list, count, seg_cap, count_clear);    \
    } while (0)
    if (packBytes <= kAviLdsPack) {
        if (h->P.prox) LMPC_AVI_GO(true, true, packBytes); else LMPC_AVI_GO(true, false, packBytes);
    } else {
        if (h->P.prox) LMPC_AVI_GO(false, true, (size_t)0); else LMPC_AVI_GO(false, false, (size_t)0);
    }
#undef LMPC_AVI_GO
    HIP_TRY(h, hipGetLastError());
    return LMPC_OK;
}

// the code objects of this handle's kernels onto the device without a launch (setup; preload_code in lmpc_api.hip):
// the generic kernel's unit, and through the occupancy queries the register-resident kernels'
void avi_preload(lmpc_handle *h) {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void *)avi_kernel<true, false>);
    if (h->aviTiers && h->aviTiersN > 0) {
        const int kfirst = h->aviTiersFirst >= 0 ? h->aviTiersFirst : (h->aviTiersN <= 6 ? 3 : 2);
        for (int s = 0; s < 2; s++)
            if (h->aviTiersOcc[s] == 0)
                (void)launch_avi_tiers(h, s == 0, kfirst, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                       nullptr, nullptr, 0, 0, &h->aviTiersOcc[s]);
    }
    (void)hipGetLastError();
}

void avi_release(lmpc_handle *h, bool pack_too) {
    if (h->dAviR) { (void)hipFree(h->dAviR); h->dAviR = nullptr; }
    if (h->dAviI) { (void)hipFree(h->dAviI); h->dAviI = nullptr; }
    h->aviSlabs = 0;
    if (h->dAviList[0]) { (void)hipFree(h->dAviList[0]); h->dAviList[0] = nullptr; }
    if (h->dAviList[1]) { (void)hipFree(h->dAviList[1]); h->dAviList[1] = nullptr; }
    if (h->dAviCnt) { (void)hipFree(h->dAviCnt); h->dAviCnt = nullptr; }
    h->aviListCap = 0;
    if (pack_too) {
        if (h->dCa) { (void)hipFree(h->dCa); h->dCa = nullptr; }
        if (h->dSa) { (void)hipFree(h->dSa); h->dSa = nullptr; }
    }
}

}  // namespace lmpc
