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

// one (mask, count, first index) contribution into the global table; called by ONE lane of a wavefront at a time (a lane
// that waits for a slot another lane of its own wavefront is filling would wait forever)
__device__ __forceinline__ void insert_global(const uint64_t *lead, int words, unsigned long long hsh, unsigned long long cnt,
                                              long long lidx, int capacity, int tcap_mask, int32_t *table, uint64_t *set_masks,
                                              unsigned long long *set_count, long long *set_first, int32_t *n_sets,
                                              int32_t *overflow) {
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

// The same reduction in two levels, for masks of up to kLocalWords words: a workgroup walks `tiles` tiles of 256 samples
// and collects their masks in a table in ITS LDS first (same scheme: claim, fill, publish; counts and first indices by
// LDS atomics), then one lane inserts the workgroup's distinct masks into the global table.  A sample of a parameter
// range falls mostly into a few dozen regions: with one global insert per wavefront and distinct mask (the one-level
// kernel) a million samples made some 40 000 atomics on a few dozen words -- 1.2 ms of a 1.3 ms step; a workgroup per
// resident slot makes a few thousand.
constexpr int kLocalWords = 4, kLocalSets = 192, kLocalTab = 512;
__global__ __launch_bounds__(256) void distinct_masks_local_kernel(
    long long N, int words, const uint64_t *__restrict__ active, const int32_t *__restrict__ exitflag,
    int capacity, int tcap_mask, int32_t *__restrict__ table, uint64_t *__restrict__ set_masks,
    unsigned long long *__restrict__ set_count, long long *__restrict__ set_first, int32_t *__restrict__ n_sets,
    int32_t *__restrict__ overflow, int tiles) {
    __shared__ int ltab[kLocalTab];
    __shared__ uint64_t lmask[kLocalSets * kLocalWords];
    __shared__ unsigned long long lcnt[kLocalSets], lhash[kLocalSets];
    __shared__ long long lfirst[kLocalSets];
    __shared__ int lused;
    for (int i = threadIdx.x; i < kLocalTab; i += blockDim.x) ltab[i] = -1;
    if (threadIdx.x == 0) lused = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int tl = 0; tl < tiles; tl++) {
        const long long idx = ((long long)blockIdx.x * tiles + tl) * blockDim.x + threadIdx.x;
        const bool valid = idx < N && (exitflag == nullptr || exitflag[idx] >= 1);
        const uint64_t *mine = active + (valid ? idx : 0) * (long long)words;
        unsigned long long remaining = __ballot(valid);
        while (remaining != 0ull) {
            const int leader = (int)__builtin_ctzll(remaining);
            const long long lidx = ((long long)__builtin_amdgcn_readlane((int)(idx >> 32), leader) << 32) |
                                   (unsigned)__builtin_amdgcn_readlane((int)(idx & 0xffffffffll), leader);
            const uint64_t *lead = active + lidx * (long long)words;
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
                int slot = (int)(hsh & (unsigned long long)(kLocalTab - 1));
                bool placed = false;
                for (int probes = 0; probes < kLocalTab && !placed; probes++) {
                    int sv = -1;
                    for (int spins = 0; spins < (1 << 20); spins++) {
                        sv = __hip_atomic_load(&ltab[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (sv != -2) break;
                    }
                    if (sv == -1) {
                        int expect = -1;
                        if (!__hip_atomic_compare_exchange_strong(&ltab[slot], &expect, -2, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_WORKGROUP)) { probes--; continue; }
                        const int k = atomicAdd(&lused, 1);
                        if (k >= kLocalSets) {                     // LDS table full: this one goes straight to the global table
                            __hip_atomic_store(&ltab[slot], -1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            break;
                        }
                        for (int q = 0; q < words; q++) lmask[k * kLocalWords + q] = lead[q];
                        lcnt[k] = cnt; lfirst[k] = lidx; lhash[k] = hsh;
                        __hip_atomic_store(&ltab[slot], k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        placed = true;
                    } else if (sv >= 0) {
                        bool match = true;
                        for (int q = 0; q < words; q++) match = match && (lmask[sv * kLocalWords + q] == lead[q]);
                        if (match) {
                            atomicAdd(&lcnt[sv], cnt);
                            atomicMin(&lfirst[sv], lidx);
                            placed = true;
                        } else slot = (slot + 1) & (kLocalTab - 1);
                    } else break;                                  // (a slot stayed "being filled": leave it to the global table)
                }
                if (!placed)
                    insert_global(lead, words, hsh, cnt, lidx, capacity, tcap_mask, table, set_masks, set_count, set_first,
                                  n_sets, overflow);
            }
            remaining &= ~same;
        }
    }
    __syncthreads();
    if (lane == 0) {                                   // one lane per wavefront, the sets dealt out to the wavefronts
        const int nu = lused < kLocalSets ? lused : kLocalSets;
        for (int kk = threadIdx.x >> 6; kk < nu; kk += (int)(blockDim.x >> 6)) {
            const int k = (kk + (int)blockIdx.x) % nu;     // (workgroups start at different sets: fewer of them on one word at a time)
            insert_global(&lmask[k * kLocalWords], words, lhash[k], lcnt[k], lfirst[k], capacity, tcap_mask, table, set_masks,
                          set_count, set_first, n_sets, overflow);
        }
    }
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
        if (lane == leader)
            insert_global(lead, words, hsh, (unsigned long long)__popcll(same), lidx, capacity, tcap_mask, table, set_masks,
                          set_count, set_first, n_sets, overflow);
        remaining &= ~same;
    }
}

// ONE-WORD masks (2 m <= 64 bits: every problem the lane kernels cover): the mask is its own key, so no slot is ever
// "being filled" and nobody waits for anybody -- EVERY lane inserts its own sample: one 64-bit compare-and-swap on the
// key table (empty -> mask; linear probing), then an atomic add on the slot's count and an atomic min on its first
// index.  First into a table in the workgroup's LDS, then the workgroup's distinct masks (all its lanes at once) into
// the global tables; the dense outputs are made by publish_w1_kernel.  The ballot loop of the kernels above spends
// ~400 cycles per distinct mask and wavefront on its leader's round trips: 150 us per 10^6 samples against 12 here.
constexpr unsigned long long kW1Empty = ~0ull;       // (no mask: a row is never active at both of its bounds)
constexpr int kW1Tab = 1024;
__global__ __launch_bounds__(256) void distinct_masks_w1_kernel(
    long long N, const uint64_t *__restrict__ active, const int32_t *__restrict__ exitflag, int tcap_mask,
    unsigned long long *__restrict__ gkey, unsigned long long *__restrict__ gcnt, long long *__restrict__ gfirst,
    int32_t *__restrict__ overflow, int tiles) {
    __shared__ unsigned long long lkey[kW1Tab];
    __shared__ unsigned int lcnt[kW1Tab];
    __shared__ long long lfirst[kW1Tab];
    for (int i = threadIdx.x; i < kW1Tab; i += blockDim.x) { lkey[i] = kW1Empty; lcnt[i] = 0u; lfirst[i] = 0x7fffffffffffffffll; }
    __syncthreads();
    auto to_global = [&](unsigned long long key, unsigned long long cnt, long long first) {
        int slot = (int)(mix64(key ^ 0x9e3779b97f4a7c15ull) & (unsigned long long)tcap_mask);
        for (int probes = 0; probes <= tcap_mask; probes++) {
            const unsigned long long old = atomicCAS(&gkey[slot], kW1Empty, key);
            if (old == kW1Empty || old == key) {
                atomicAdd(&gcnt[slot], cnt);
                atomicMin(&gfirst[slot], first);
                return;
            }
            slot = (slot + 1) & tcap_mask;
        }
        atomicExch(overflow, 1);                             // more distinct masks than slots (4 x capacity)
    };
    for (int tl = 0; tl < tiles; tl++) {
        const long long idx = ((long long)blockIdx.x * tiles + tl) * blockDim.x + threadIdx.x;
        if (idx >= N) break;
        if (exitflag != nullptr && exitflag[idx] < 1) continue;
        const unsigned long long key = active[idx];
        int slot = (int)(mix64(key ^ 0x9e3779b97f4a7c15ull) & (unsigned long long)(kW1Tab - 1));
        bool placed = false;
        for (int probes = 0; probes < kW1Tab; probes++) {
            const unsigned long long old = atomicCAS(&lkey[slot], kW1Empty, key);
            if (old == kW1Empty || old == key) {
                atomicAdd(&lcnt[slot], 1u);
                atomicMin(&lfirst[slot], idx);
                placed = true;
                break;
            }
            slot = (slot + 1) & (kW1Tab - 1);
        }
        if (!placed) to_global(key, 1ull, idx);              // (more than 1024 distinct masks in one workgroup's share)
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kW1Tab; i += blockDim.x)
        if (lkey[i] != kW1Empty) to_global(lkey[i], (unsigned long long)lcnt[i], lfirst[i]);
}

// The global tables of distinct_masks_w1_kernel made dense: set_masks / set_count / set_first / n_sets (device), the
// overflow word, and -- `out` non-null -- the block in mapped host memory publish_sets_kernel writes (same layout).
// Every slot read is reset: the tables are clean for the next call without a memset.  One workgroup.
__global__ __launch_bounds__(1024) void publish_w1_kernel(
    int tcap, int capacity, unsigned long long *__restrict__ gkey, unsigned long long *__restrict__ gcnt,
    long long *__restrict__ gfirst, uint64_t *__restrict__ set_masks, unsigned long long *__restrict__ set_count,
    long long *__restrict__ set_first, int32_t *__restrict__ n_sets, int32_t *__restrict__ overflow, long long *__restrict__ out) {
    __shared__ int found;
    __shared__ unsigned long long total;
    if (threadIdx.x == 0) { found = 0; total = 0ull; }
    __syncthreads();
    unsigned long long mine = 0ull;
    for (int s = threadIdx.x; s < tcap; s += blockDim.x) {
        const unsigned long long key = gkey[s];
        if (key == kW1Empty) continue;
        const unsigned long long cnt = gcnt[s];
        const long long first = gfirst[s];
        gkey[s] = kW1Empty; gcnt[s] = 0ull; gfirst[s] = 0x7fffffffffffffffll;
        const int k = atomicAdd(&found, 1);
        if (k < capacity) {
            set_masks[k] = key; set_count[k] = cnt; set_first[k] = first;
            if (out) { long long *row = out + 4 + (long long)k * 3; row[0] = (long long)key; row[1] = (long long)cnt; row[2] = first; }
            mine += cnt;
        }
    }
    atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int ov = (found > capacity || *overflow != 0) ? 1 : 0;
        *n_sets = found;
        overflow[8] = ov;                                    // (what lmpc_distinct_active_sets_overflowed reads)
        *overflow = 0;
        if (out) {
            out[1] = (long long)ov;
            out[2] = (long long)total;
            __threadfence_system();
            out[0] = (long long)found;
        }
    }
}

// The distinct sets of one call, written by the device straight into mapped host memory: word 0 = sets found (claims,
// if more than `capacity`), word 1 = overflow word, word 2 = problems counted, then per set its mask, count and first
// index.  One block; the host reads the block after ONE synchronisation of the stream -- no copy calls.
__global__ __launch_bounds__(256) void publish_sets_kernel(
    int words, int capacity, const uint64_t *__restrict__ set_masks, const unsigned long long *__restrict__ set_count,
    const long long *__restrict__ set_first, const int32_t *__restrict__ n_sets, const int32_t *__restrict__ overflow,
    long long *__restrict__ out) {
    const int ns = *n_sets < capacity ? *n_sets : capacity;
    __shared__ unsigned long long total;
    if (threadIdx.x == 0) total = 0ull;
    __syncthreads();
    unsigned long long mine = 0ull;
    const int per = words + 2;
    for (int k = threadIdx.x; k < ns; k += blockDim.x) {
        long long *row = out + 4 + (long long)k * per;
        for (int q = 0; q < words; q++) row[q] = (long long)set_masks[(long long)k * words + q];
        row[words] = (long long)set_count[k];
        row[words + 1] = set_first[k];
        mine += set_count[k];
    }
    atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        out[1] = (long long)*overflow;
        out[2] = (long long)total;
        __threadfence_system();
        out[0] = (long long)*n_sets;
    }
}

}  // namespace lmpc

extern "C" {

// words == 1 and a large batch: the lock-free per-lane reduction, then the dense outputs (and, `out` non-null, the block
// in mapped host memory) by one more launch; the tables are handle-owned and clean between calls
static int distinct_w1(lmpc_handle *h, int64_t N, const uint64_t *active, const int32_t *exitflag, int32_t capacity,
                       uint64_t *set_masks, int64_t *set_count, int64_t *set_first, int32_t *n_sets, long long *out, hipStream_t st) {
    using namespace lmpc;
    int tcap = 64;
    while (tcap < 4 * (long long)capacity && tcap < (1 << 24)) tcap <<= 1;
    if (!h->dRegTable) {
        HIP_TRY(h, hipMalloc(&h->dRegTable, sizeof(int32_t) * ((size_t)64 + 16)));
        h->regCap = 64;
        HIP_TRY(h, hipMemsetAsync(h->dRegTable, 0, sizeof(int32_t) * 16, st));
    }
    if (tcap != h->regW1Cap) {
        if (h->dRegW1) { (void)hipStreamSynchronize(st); hipFree(h->dRegW1); }
        h->dRegW1 = nullptr; h->regW1Cap = 0;
        HIP_TRY(h, hipMalloc(&h->dRegW1, sizeof(unsigned long long) * 3 * (size_t)tcap));
        HIP_TRY(h, hipMemsetAsync(h->dRegW1, 0xff, sizeof(unsigned long long) * (size_t)tcap, st));                 // keys: empty
        HIP_TRY(h, hipMemsetAsync(h->dRegW1 + tcap, 0, sizeof(unsigned long long) * (size_t)tcap, st));             // counts
        HIP_TRY(h, hipMemsetAsync(h->dRegW1 + 2 * (size_t)tcap, 0x7f, sizeof(unsigned long long) * (size_t)tcap, st));   // first indices: large
        HIP_TRY(h, hipMemsetAsync(h->dRegTable, 0, sizeof(int32_t) * 16, st));
        h->regW1Cap = tcap;
    }
    unsigned long long *gkey = h->dRegW1, *gcnt = h->dRegW1 + tcap;
    long long *gfirst = reinterpret_cast<long long *>(h->dRegW1 + 2 * (size_t)tcap);
    const long long tilesAll = (N + 255) / 256;
    const long long wantBlocks = (long long)h->numCU * (h->regBlocks > 0 ? h->regBlocks : 2);
    const int tiles = (int)((tilesAll + wantBlocks - 1) / wantBlocks);
    const unsigned grid = (unsigned)((tilesAll + tiles - 1) / tiles);
    hipLaunchKernelGGL(distinct_masks_w1_kernel, dim3(grid), dim3(256), 0, st, (long long)N, active, exitflag, tcap - 1, gkey, gcnt,
                       gfirst, h->dRegTable, tiles);
    HIP_TRY(h, hipGetLastError());
    hipLaunchKernelGGL(publish_w1_kernel, dim3(1), dim3(1024), 0, st, tcap, (int)capacity, gkey, gcnt, gfirst, set_masks,
                       reinterpret_cast<unsigned long long *>(set_count), reinterpret_cast<long long *>(set_first), n_sets,
                       h->dRegTable, out);
    HIP_TRY(h, hipGetLastError());
    return LMPC_OK;
}

static bool use_w1(const lmpc_handle *h, int64_t N) { return h->regW1 && h->P.words() == 1 && N >= 65536; }

int lmpc_distinct_active_sets_device(lmpc_handle *h, int64_t N, const uint64_t *active, const int32_t *exitflag,
                                     int32_t capacity, uint64_t *set_masks, int64_t *set_count, int64_t *set_first,
                                     int32_t *n_sets, void *stream) {
    using namespace lmpc;
    if (!h) return LMPC_ERR_BADARG;
    if (N < 0 || capacity < 1 || !set_masks || !set_count || !set_first || !n_sets || (N > 0 && !active))
        return fail(h, LMPC_ERR_BADARG, "lmpc_distinct_active_sets_device: NULL array, negative N or capacity < 1");
    LMPC_ENTER_DEVICE(h);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (use_w1(h, N)) return distinct_w1(h, N, active, exitflag, capacity, set_masks, set_count, set_first, n_sets, nullptr, st);
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
    if (N > 0 && h->P.words() <= kLocalWords && N >= 65536) {
        // two levels: one workgroup per CU ("region_blocks": per CU), each over its share of the tiles
        const long long tilesAll = (N + 255) / 256;
        const long long wantBlocks = (long long)h->numCU * (h->regBlocks > 0 ? h->regBlocks : 1);
        const int tiles = (int)((tilesAll + wantBlocks - 1) / wantBlocks);
        const unsigned grid = (unsigned)((tilesAll + tiles - 1) / tiles);
        hipLaunchKernelGGL(distinct_masks_local_kernel, dim3(grid), dim3(256), 0, st, (long long)N, h->P.words(), active,
                           exitflag, (int)capacity, tcap - 1, table, set_masks, reinterpret_cast<unsigned long long *>(set_count),
                           reinterpret_cast<long long *>(set_first), n_sets, overflow, tiles);
        HIP_TRY(h, hipGetLastError());
    } else if (N > 0) {
        const unsigned grid = (unsigned)((N + 255) / 256);
        hipLaunchKernelGGL(distinct_masks_kernel, dim3(grid), dim3(256), 0, st, (long long)N, h->P.words(), active, exitflag,
                           (int)capacity, tcap - 1, table, set_masks, reinterpret_cast<unsigned long long *>(set_count),
                           reinterpret_cast<long long *>(set_first), n_sets, overflow);
        HIP_TRY(h, hipGetLastError());
    }
    return LMPC_OK;
}

int lmpc_discover_regions_device(lmpc_handle *h, int64_t N, const double *theta, double *x, int32_t *exitflag,
                                 uint64_t *active, int32_t capacity, uint64_t *set_masks, int64_t *set_count,
                                 int64_t *set_first, int32_t *n_sets, const long long **result_host, void *stream) {
    using namespace lmpc;
    if (!h) return LMPC_ERR_BADARG;
    if (N < 0 || capacity < 1 || !set_masks || !set_count || !set_first || !n_sets || !result_host ||
        (N > 0 && (!active || !x || !exitflag)))
        return fail(h, LMPC_ERR_BADARG, "lmpc_discover_regions_device: NULL array, negative N or capacity < 1");
    LMPC_ENTER_DEVICE(h);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t need = 4 + (size_t)capacity * ((size_t)h->P.words() + 2);
    if (need > h->regOutWords) {
        if (h->hRegOut) { (void)hipStreamSynchronize(st); (void)hipHostFree(h->hRegOut); }
        h->hRegOut = nullptr; h->dRegOut = nullptr; h->regOutWords = 0;
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void **>(&h->hRegOut), sizeof(long long) * need, hipHostMallocMapped));
        HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void **>(&h->dRegOut), h->hRegOut, 0));
        h->regOutWords = need;
    }
    h->hRegOut[0] = -1;                                   // "not published yet" until the kernel's last store
    int rc = N > 0 ? lmpc_solve_batch_device(h, N, theta, x, exitflag, nullptr, active, nullptr, stream) : LMPC_OK;
    if (rc != LMPC_OK) return rc;
    if (use_w1(h, N)) {                                   // (one-word masks: the dense outputs and the host block by one kernel)
        rc = distinct_w1(h, N, active, exitflag, capacity, set_masks, set_count, set_first, n_sets, h->dRegOut, st);
        if (rc == LMPC_OK) *result_host = h->hRegOut;
        return rc;
    }
    rc = lmpc_distinct_active_sets_device(h, N, active, exitflag, capacity, set_masks, set_count, set_first, n_sets, stream);
    if (rc != LMPC_OK) return rc;
    hipLaunchKernelGGL(publish_sets_kernel, dim3(1), dim3(256), 0, st, h->P.words(), (int)capacity, set_masks,
                       reinterpret_cast<const unsigned long long *>(set_count), reinterpret_cast<const long long *>(set_first),
                       n_sets, h->dRegTable, h->dRegOut);
    HIP_TRY(h, hipGetLastError());
    *result_host = h->hRegOut;
    return LMPC_OK;
}

int lmpc_distinct_active_sets_overflowed(lmpc_handle *h, void *stream) {
    using namespace lmpc;
    if (!h) return LMPC_ERR_BADARG;
    if (!h->dRegTable) return 0;
    LMPC_ENTER_DEVICE(h);
    int32_t o[9] = {0};                                   // word 0: this call's flag; word 8: the last publish_w1_kernel's
    HIP_TRY(h, hipMemcpyAsync(o, h->dRegTable, sizeof(o), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    HIP_TRY(h, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return o[0] != 0 ? o[0] : o[8];
}

}  // extern "C"
