// Distinct optimal active sets of a solved batch, found on the device.
//
// Caller side of /root/reference/src/explicit.jl:23-48 (and certify.jl:18-30): a sampling-based region discovery
// solves the condensed QP on a large sample of the parameter range and needs the DISTINCT final active sets (one
// per critical region the sample hit), how many samples fell into each, and one representative sample.  The masks
// are already on the GPU (lmpc_solve_batch_device's `active` output); bringing 8 N bytes to the host and sorting
// them there cost hundreds of milliseconds around a 24-80 us solve.  Here the batch is reduced where it lies:
//
//   * a wavefront groups its 64 samples by mask (ballot loop: the lowest remaining lane leads, every lane compares
//     its words with the leader's), so one table operation is made per distinct mask per wavefront -- a sample of a
//     parameter range hits a few hundred regions, neighbouring samples mostly the same one;
//   * the leader inserts into a global open-addressing hash table (linear probing; slots hold an index into the
//     dense output arrays, -1 empty, -2 being filled): the first wavefront to see a mask claims the slot, draws the
//     next dense index, writes the mask, publishes; later ones add their count and take the minimum sample index.
//
// Only n_sets x (words + 2) x 8 bytes cross PCIe afterwards.  Order of the dense arrays depends on the race; the
// host wrapper sorts by (count, first index), which does not.
#include <hip/hip_runtime.h>

#include <cstring>

#include "lmpc_internal.hpp"

namespace lmpc {

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

__global__ __launch_bounds__(256) void distinct_masks_kernel(
    long long N, int words, const uint64_t *__restrict__ active, const int32_t *__restrict__ exitflag,
    int capacity, int tcap_mask, int32_t *__restrict__ table, uint64_t *__restrict__ set_masks,
    unsigned long long *__restrict__ set_count, long long *__restrict__ set_first, int32_t *__restrict__ n_sets,
    int32_t *__restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = idx < N && (exitflag == nullptr || exitflag[idx] >= 1);
    const uint64_t *mine = active + (valid ? idx : 0) * (long long)words;
    unsigned long long remaining = __ballot(valid);
    while (remaining != 0ull) {
        const int leader = (int)__builtin_ctzll(remaining);
        const long long lidx = ((long long)__builtin_amdgcn_readlane((int)(idx >> 32), leader) << 32) |
                               (unsigned)__builtin_amdgcn_readlane((int)(idx & 0xffffffffll), leader);
        const uint64_t *lead = active + lidx * (long long)words;          // wave-uniform address: scalar loads
        bool eq = valid && ((remaining >> lane) & 1ull);
        unsigned long long hsh = 0x9e3779b97f4a7c15ull;
        for (int q = 0; q < words; q++) {
            const uint64_t lw = lead[q];
            eq = eq && (mine[q] == lw);
            hsh = mix64(hsh ^ lw);
        }
        const unsigned long long same = __ballot(eq);
        if (lane == leader) {
            const unsigned long long cnt = (unsigned long long)__popcll(same);
            int slot = (int)(hsh & (unsigned long long)tcap_mask);
            for (int probes = 0; probes <= tcap_mask; probes++) {
                int s = -1;
                bool done = false;
                for (int spins = 0; spins < (1 << 20); spins++) {
                    s = __hip_atomic_load(&table[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (s == -1) {
                        int expect = -1;
                        if (__hip_atomic_compare_exchange_strong(&table[slot], &expect, -2, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT)) {
                            const int k = atomicAdd(n_sets, 1);
                            if (k >= capacity) {                           // more distinct sets than the caller made room for:
                                atomicExch(overflow, 1);                   // the host retries with more (slot given back)
                                __hip_atomic_store(&table[slot], -1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                            } else {
                                for (int q = 0; q < words; q++) set_masks[(long long)k * words + q] = lead[q];
                                set_count[k] = cnt;
                                set_first[k] = lidx;
                                __hip_atomic_store(&table[slot], k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            done = true;
                            break;
                        }
                        continue;                                          // somebody else took it: look again
                    }
                    if (s >= 0) break;                                     // published: compare below
                    // s == -2: being filled by another wavefront -- unless the call has overflowed anyway
                    if (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { done = true; break; }
                }
                if (done) break;
                if (s < 0) { atomicCAS(overflow, 0, 2); break; }          // (a slot never stays "being filled")
                bool match = true;
                for (int q = 0; q < words; q++)
                    match = match && (__hip_atomic_load(&set_masks[(long long)s * words + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == lead[q]);
                if (match) {
                    atomicAdd(&set_count[s], cnt);
                    atomicMin(&set_first[s], lidx);
                    break;
                }
                slot = (slot + 1) & tcap_mask;
            }
        }
        remaining &= ~same;
    }
}

}  // namespace lmpc

extern "C" {

int lmpc_distinct_active_sets_device(lmpc_handle *h, int64_t N, const uint64_t *active, const int32_t *exitflag,
                                     int32_t capacity, uint64_t *set_masks, int64_t *set_count, int64_t *set_first,
                                     int32_t *n_sets, void *stream) {
    using namespace lmpc;
    if (!h) return LMPC_ERR_BADARG;
    if (N < 0 || capacity < 1 || !set_masks || !set_count || !set_first || !n_sets || (N > 0 && !active))
        return fail(h, LMPC_ERR_BADARG, "lmpc_distinct_active_sets_device: NULL array, negative N or capacity < 1");
    LMPC_ENTER_DEVICE(h);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int tcap = 64;
    while (tcap < 4 * (long long)capacity && tcap < (1 << 28)) tcap <<= 1;
    if (tcap > h->regCap) {
        hipFree(h->dRegTable); h->dRegTable = nullptr; h->regCap = 0;
        HIP_TRY(h, hipMalloc(&h->dRegTable, sizeof(int32_t) * ((size_t)tcap + 16)));
        h->regCap = tcap;
    }
    int32_t *table = h->dRegTable + 16, *overflow = h->dRegTable;
    HIP_TRY(h, hipMemsetAsync(h->dRegTable, 0, sizeof(int32_t) * 16, st));
    HIP_TRY(h, hipMemsetAsync(table, 0xff, sizeof(int32_t) * (size_t)tcap, st));
    HIP_TRY(h, hipMemsetAsync(n_sets, 0, sizeof(int32_t), st));
    if (N > 0) {
        const unsigned grid = (unsigned)((N + 255) / 256);
        hipLaunchKernelGGL(distinct_masks_kernel, dim3(grid), dim3(256), 0, st, (long long)N, h->P.words(), active, exitflag,
                           (int)capacity, tcap - 1, table, set_masks, reinterpret_cast<unsigned long long *>(set_count),
                           reinterpret_cast<long long *>(set_first), n_sets, overflow);
        HIP_TRY(h, hipGetLastError());
    }
    return LMPC_OK;
}

int lmpc_distinct_active_sets_overflowed(lmpc_handle *h, void *stream) {
    using namespace lmpc;
    if (!h) return LMPC_ERR_BADARG;
    if (!h->dRegTable) return 0;
    LMPC_ENTER_DEVICE(h);
    int32_t o = 0;
    HIP_TRY(h, hipMemcpyAsync(&o, h->dRegTable, sizeof(o), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    HIP_TRY(h, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return o;
}

}  // extern "C"
