// Instantiations and launch of avi_tiers_kernel (lmpc_avi_tiers_kernel.hpp): small box-constrained variational
// problems in registers, n = 2 .. 8.
#include <hip/hip_runtime.h>

#include "lmpc_avi_tiers_kernel.hpp"
#include "lmpc_internal.hpp"

namespace lmpc {

namespace {
template <int N, int KMAX, bool LIST>
int go(lmpc_handle *h, unsigned grid, hipStream_t st, const double *theta, double *x, int32_t *flag, int32_t *iters,
       uint64_t *active, const int32_t *list_in, const int32_t *count_in, int32_t *list_out, int32_t *count_out,
       int32_t *count_clear, long long seg_cap, long long nprob, int *occ) {
    auto kern = avi_tiers_kernel<N, KMAX, LIST>;
    // the constants, then the four wavefronts' parked outputs (nout reals a lane)
    const size_t lds = sizeof(double) * ((size_t)AviSmallLds<N>::reals + 4 * (size_t)h->A.nout * 64);
    if (occ) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, 256, lds) != hipSuccess || nb < 1) nb = 1;
        *occ = nb;
        return LMPC_OK;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, h->A, h->dCa, theta, x, flag, iters, active, list_in, count_in,
                       list_out, count_out, count_clear, seg_cap, nprob);
    HIP_TRY(h, hipGetLastError());
    return LMPC_OK;
}
template <int N>
int go_lane(lmpc_handle *h, unsigned grid, hipStream_t st, const double *theta, double *x, int32_t *flag, int32_t *iters,
            uint64_t *active, const int32_t *list_in, const int32_t *count_in, int32_t *list_out, int32_t *count_out,
            int32_t *count_clear, long long seg_cap, long long nprob, int *occ) {
    const size_t lds = sizeof(double) * (size_t)avi_lane_lds_reals<N>();
    if (lds > 48 * 1024) {
        HIP_TRY(h, hipFuncSetAttribute((const void *)avi_lane_kernel<N, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(h, hipFuncSetAttribute((const void *)avi_lane_kernel<N, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const bool listed = list_in != nullptr || occ != nullptr;
    const void *kern = listed ? (const void *)avi_lane_kernel<N, true> : (const void *)avi_lane_kernel<N, false>;
    if (occ) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds) != hipSuccess || nb < 1) nb = 1;
        *occ = nb;
        return LMPC_OK;
    }
    if (listed)
        hipLaunchKernelGGL((avi_lane_kernel<N, true>), dim3(grid), dim3(256), lds, st, h->A, h->dCa, theta, x, flag, iters, active,
                           list_in, count_in, list_out, count_out, count_clear, seg_cap, nprob);
    else
        hipLaunchKernelGGL((avi_lane_kernel<N, false>), dim3(grid), dim3(256), lds, st, h->A, h->dCa, theta, x, flag, iters, active,
                           list_in, count_in, list_out, count_out, count_clear, seg_cap, nprob);
    HIP_TRY(h, hipGetLastError());
    return LMPC_OK;
}
}  // namespace

// first == true: the pass over the whole batch with `kfirst` tiers (1 .. 3, capped at n; 0: the complete lane kernel
// over the whole batch); else the lane kernel on that pass's list.  occ != nullptr: only report the workgroups per CU the instantiation keeps resident.
int launch_avi_tiers(lmpc_handle *h, bool first, int kfirst, unsigned grid, hipStream_t st, const double *theta, double *x,
                     int32_t *flag, int32_t *iters, uint64_t *active, const int32_t *list_in, const int32_t *count_in,
                     int32_t *list_out, int32_t *count_out, int32_t *count_clear, long long seg_cap, long long nprob, int *occ) {
    const int n = h->A.n;
#define LMPC_AT(N, K, L) go<N, K, L>(h, grid, st, theta, x, flag, iters, active, list_in, count_in, list_out, count_out, \
                                     count_clear, seg_cap, nprob, occ)
#define LMPC_AT_N(N)                                                         \
    case N:                                                                  \
        if (!first || kfirst <= 0)                                           \
            return go_lane<N>(h, grid, st, theta, x, flag, iters, active, list_in, count_in, list_out, count_out, \
                              count_clear, seg_cap, nprob, occ);             \
        if (kfirst <= 1) return LMPC_AT(N, 1, false);                        \
        if (kfirst == 2 || N == 2) return LMPC_AT(N, 2, false);              \
        return LMPC_AT(N, (N < 3 ? N : 3), false);
    switch (n) {
        LMPC_AT_N(2) LMPC_AT_N(3) LMPC_AT_N(4) LMPC_AT_N(5) LMPC_AT_N(6) LMPC_AT_N(7) LMPC_AT_N(8)
        default: break;
    }
#undef LMPC_AT_N
#undef LMPC_AT
    return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc: the register-resident variational kernel covers n = 2 .. 8");
}

}  // namespace lmpc
