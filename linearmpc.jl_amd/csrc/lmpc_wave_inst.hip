// One instantiation set of the wavefront kernel: real type LMPC_WV_REAL (double | float) with or
// without branch and bound (LMPC_WV_BNB 0 | 1), n-chain or Gram-scan form (LMPC_WV_GRAM 0 | 1).  Compiled
// eight times (see Makefile) so that the library's ~200 kernel instantiations build in parallel.
#include "lmpc_wave_launch.hpp"

namespace lmpc {
#ifndef LMPC_WV_SIM
#define LMPC_WV_SIM 1
#endif
template int launch_wave_inst<LMPC_WV_REAL, (LMPC_WV_BNB != 0), (LMPC_WV_GRAM != 0), (LMPC_WV_SIM != 0)>(lmpc_handle *, const LMPC_WV_REAL *, int64_t,
                                                                const LMPC_WV_REAL *, LMPC_WV_REAL *, int32_t *,
                                                                int32_t *, uint64_t *, const uint64_t *, hipStream_t);
#ifdef LMPC_WV_HELPERS
// (one translation unit carries the launcher's non-template helpers the API file needs)
int wave_first_pass_cap(lmpc_handle *h, int64_t nprob) { return wave_first_pass_cap_impl(h, nprob, sizeof(double)); }
void wave_stat_read(const lmpc_handle *h, unsigned long long out[4]) { wave_stat_sums(h, out); }
int wave_reserve(lmpc_handle *h, int64_t nprob, hipStream_t st) { return wave_reserve_impl(h, nprob, st); }
#endif
}  // namespace lmpc

#ifdef LMPC_WAVE_TRACE
// diagnostic build only: per-phase shader-clock sums of THIS translation unit's kernels (see lmpc_wave_kernel.hpp)
extern "C" int lmpc_debug_wave_trace(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(lmpc::g_wave_trace), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(lmpc::g_wave_trace), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
