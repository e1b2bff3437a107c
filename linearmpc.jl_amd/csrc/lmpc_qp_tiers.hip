// Instantiations and launch of qp_tiers_kernel (lmpc_qp_tiers_kernel.hpp): straight-line tiers, one problem per lane,
// in front of the wavefront kernel for n = 2 .. 12 variables and up to 64 hard rows.
#include <hip/hip_runtime.h>

#include <vector>

#include "lmpc_internal.hpp"
#include "lmpc_qp_tiers_kernel.hpp"

namespace lmpc {

namespace {
template <int N>
int go(lmpc_handle *h, int64_t nprob, const double *theta, double *x, int32_t *flag, int32_t *iters, uint64_t *active,
       int32_t *list, int32_t *count, long long seg_cap, hipStream_t st, bool preload) {
    auto kern = qp_tiers_kernel<N>;
    const size_t lds = sizeof(double) * (size_t)QpTiersLds<N>::reals(h->P.m);
    if (lds > 48 * 1024)
        HIP_TRY(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (!h->dQpScan) {
        // the scan pack: per row M_j, du0_j, dl0_j (padded to an even count), laid out for one batch of scalar loads
        constexpr int NR = qp_scan_row_reals(N);
        std::vector<double> sp((size_t)h->P.m * NR + 16, 0.0);        // (+ slack: a batch never reads past the buffer)
        for (int j = 0; j < h->P.m; j++) {
            for (int c = 0; c < N; c++) sp[(size_t)j * NR + c] = h->P.M[(size_t)j * N + c];
            sp[(size_t)j * NR + N] = h->P.du0[j];
            sp[(size_t)j * NR + N + 1] = h->P.dl0[j];
        }
        // (the handle gets the pointer only once the pack is in place: a failed copy must not leave a buffer of garbage
        // that every later launch would take for the pack -- ADVICE round 4)
        double *scan = nullptr;
        HIP_TRY(h, hipMalloc(&scan, sizeof(double) * sp.size()));
        if (hipMemcpy(scan, sp.data(), sizeof(double) * sp.size(), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(scan);
            return fail(h, LMPC_ERR_HIP, "lmpc: upload of the tiers pass's scan pack failed");
        }
        h->dQpScan = scan;
    }
    if (preload) {
        hipFuncAttributes fa;
        HIP_TRY(h, hipFuncGetAttributes(&fa, (const void *)kern));
        return LMPC_OK;
    }
    unsigned long long soft = 0ull;
    for (int j = 0; j < h->P.m; j++) soft |= (h->P.sense[j] & SENSE_SOFT) ? (1ull << j) : 0ull;
    // one workgroup of four wavefronts per CU (its LDS), a multiple of 16 workgroups (wavefronts a multiple of kShards)
    const long long tiles = (nprob + 63) / 64;
    long long grid = std::min<long long>((long long)h->numCU, (tiles + 3) / 4);
    grid = ((grid + 15) / 16) * 16;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, h->W, h->dCw, theta, x, flag, iters, active, list, count,
                       seg_cap, (long long)nprob, h->dQpScan, soft);
    HIP_TRY(h, hipGetLastError());
    return LMPC_OK;
}
}  // namespace

size_t qp_tiers_lds_bytes(int n, int m) {
    switch (n) {
#define LMPC_QT(N) case N: return sizeof(double) * (size_t)QpTiersLds<N>::reals(m);
        LMPC_QT(2) LMPC_QT(3) LMPC_QT(4) LMPC_QT(5) LMPC_QT(6) LMPC_QT(7) LMPC_QT(8) LMPC_QT(9) LMPC_QT(10) LMPC_QT(11) LMPC_QT(12)
#undef LMPC_QT
        default: return (size_t)-1;
    }
}

int launch_qp_tiers(lmpc_handle *h, int64_t nprob, const double *theta, double *x, int32_t *flag, int32_t *iters,
                    uint64_t *active, int32_t *list, int32_t *count, long long seg_cap, hipStream_t st, bool preload) {
    switch (h->P.n) {
#define LMPC_QT(N) case N: return go<N>(h, nprob, theta, x, flag, iters, active, list, count, seg_cap, st, preload);
        LMPC_QT(2) LMPC_QT(3) LMPC_QT(4) LMPC_QT(5) LMPC_QT(6) LMPC_QT(7) LMPC_QT(8) LMPC_QT(9) LMPC_QT(10) LMPC_QT(11) LMPC_QT(12)
#undef LMPC_QT
        default: break;
    }
    return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: the tiers pass covers n = 2 .. 12");
}

}  // namespace lmpc
