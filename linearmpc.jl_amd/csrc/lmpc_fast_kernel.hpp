// ONE launch for small box-constrained problems (m == ms == n <= 6, cold start): streaming pass and the
// iterations of the problems that need them in the same kernel, overlapped in time.
//
// Why: in the two-launch form (screen_kernel, then lane_kernel on the work list) the second kernel is a latency
// chain -- launch, count -> list -> record loads, then ~1500 dependent instructions per wavefront -- of 16 us
// behind 16 us of streaming, although its work is a tenth of the first kernel's.  Here the same work runs
// UNDER the stream:
//
//   * a workgroup of four wavefronts covers R consecutive tiles of 64 problems.  The first `nstr` wavefronts
//     stream them (the record of a wavefront's next tile is in flight while the current one is screened exactly
//     as screen_kernel does it), finish the problems whose unconstrained optimum is feasible and push the
//     indices of the others into the workgroup's queue in LDS (reserve with one LDS atomic per tile, write);
//   * the remaining wavefronts -- and every streaming wavefront once its tiles are done -- claim 64 queued
//     problems at a time (compare-and-swap on the read index) and solve them one per lane: the straight-line
//     tiers of lmpc_tiers.hpp first, the generic loop of lane_kernel (lane_loop, same LDS layout) for the lanes
//     the tiers do not finish, so the kernel is complete by itself and no second launch follows.
//
// Synchronisation is workgroup-local and one-directional: producers never wait (the queue holds every problem
// of the workgroup), consumers wait only for producers -- for the queue to fill up to a claim, and for a
// reserved slot to be written (slots start at -1; a producer writes its slots right after reserving them).
// Every wait is bounded by a spin limit that raises an error flag instead of hanging.
//
// Results are those of the two-launch form bit for bit: the screening test, the tiers and lane_loop are the
// same code or the same fma chains.
//
// Replaces, per problem: mpc_update_qp (reference codegen/mpc_update_qp.c:1-10), daqp_ldp ([EXT] libdaqp,
// called at mpc_update_qp.c:48 / utils.jl:282) and mpc_get_solution (mpc_update_qp.c:14-22).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lmpc_lane_kernel.hpp"
#include "lmpc_tiers.hpp"

namespace lmpc {

#ifndef LMPC_FAST_WAVES
#define LMPC_FAST_WAVES 3      // wavefronts per SIMD the kernel is register-budgeted for (lane_loop sets the need)
#endif
constexpr int kFastSpinLimit = 1 << 22;
constexpr int kFastCtrs = 16;              // ticket counters of the dynamic tail (a power of two)
constexpr int kFastMaxTiles = 96;          // tiles of 64 problems per workgroup at most (LDS queue: 256 bytes per tile)
#ifndef LMPC_FAST_AHEAD
#define LMPC_FAST_AHEAD 1
#endif
constexpr int kFastAhead = LMPC_FAST_AHEAD;      // tiles a streaming wavefront requests before it consumes the first

// One LDS-DMA piece: every active lane moves 16 bytes from its own global address straight into LDS at
// lds_dst + 16 * lane (lds_dst wave-uniform, in M0) -- no vector register on the way, 1 KiB per wave-instruction.
// Inline assembly on purpose: the compiler would drain a DMA it knows about with vmcnt(0) in front of the next LDS
// read, and the point of the ring below is to keep the NEXT tile's pieces in flight across that read.
#ifndef LMPC_FAST_DMA_POLICY
#define LMPC_FAST_DMA_POLICY " nt"         // "" default cache policy, " nt" nontemporal
#endif
__device__ __forceinline__ void fast_dma16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" LMPC_FAST_DMA_POLICY
                 "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
#ifndef LMPC_FAST_DMA_DEPTH
#define LMPC_FAST_DMA_DEPTH 2              // tiles per streaming wavefront in its LDS ring (0 = records through registers)
#endif
// bytes of one ring slot = one tile of 64 records
__host__ __device__ constexpr size_t fast_tile_bytes(int NT) { return (size_t)64 * NT * sizeof(double); }

__device__ __forceinline__ int lds_load(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// bytes of dynamic LDS a workgroup of fast_kernel<.., N> needs for R tiles
// doubles of the screening constants kept in LDS: rows of Dth padded to NTHMAX columns, (du, dl) pairs, the first
// output's row of Xth, x0 of the first output (+ padding to an even count)
__host__ __device__ constexpr int fast_scr_doubles(int N, int NTHMAX) { return (N * NTHMAX + 2 * N + NTHMAX + 1 + 1) & ~1; }
// Hand-over of a queued problem: the streaming wavefront that found it has its shifts b_j = Dth_j . theta and the
// output shift x0 + Xth theta in registers; the first kFastPay queue positions of a workgroup carry them in LDS next
// to the index, so the solving wavefront starts on an LDS read instead of re-reading the record from L2 and
// re-forming the products (1.2 us of a 6 us pass).  Positions beyond (a batch where most points iterate) take the
// record from memory as before.  The values are the ones the solving side would compute: same chains, same constants.
#ifndef LMPC_FAST_PAY
#define LMPC_FAST_PAY 256
#endif
constexpr int kFastPay = LMPC_FAST_PAY;
__host__ __device__ constexpr size_t fast_lds_bytes(int N, int R, int NTHMAX) {
    return sizeof(double) * (size_t)(((N * N + N * (N + 1) / 2 + 2 * N + 1) & ~1) + 4 * N * 64 + fast_scr_doubles(N, NTHMAX) +
                                     (size_t)kFastPay * (N + 1)) +
           sizeof(int32_t) * ((size_t)R * 64 + 4);
}
// ... plus the streaming wavefronts' LDS-DMA rings behind it (dk slots of one tile per streaming wavefront)
__host__ __device__ constexpr size_t fast_lds_bytes_dma(int N, int R, int NTHMAX, int NT, int nstr, int dk) {
    return fast_lds_bytes(N, R, NTHMAX) + (size_t)dk * nstr * fast_tile_bytes(NT);
}

// GATHER: the generated controller's call (lmpc_compute_control*): theta is assembled from the five argument arrays
// of mpc_compute_control (codegen/mpc_update_parameter.c) instead of read from a buffer, X is the caller's `control`
// array -- read (previous control, first n_control_prev entries) and overwritten (u*).  A problem's control entries
// are read by the wavefront that streams it and, if it needs iterations, once more by the lane that solves it; they
// are written once, by whichever of the two finishes the problem -- after its own reads, so the in/out array needs
// no copy.
// (the kernel's body: shared by the one-batch launch and the several-batches launch below)
template <int NTHMAX, int NT, int N, bool GATHER = false>
__device__ __forceinline__ void fast_body(
    const PackLayout &P, const double *__restrict__ C, const double *__restrict__ theta,
    double *__restrict__ X, int32_t *__restrict__ exitflag, int32_t *__restrict__ iters,
    uint64_t *__restrict__ active, long long nprob, int R, int nstr, int32_t *__restrict__ errflag, int spinLimit,
    int dk, int Rs, int Dcap, int32_t *__restrict__ tctr, int32_t *__restrict__ tctr_next) {
    constexpr int KMAX = LMPC_FAST_KMAX < N ? LMPC_FAST_KMAX : N;
    constexpr int nconst = N * N + N * (N + 1) / 2 + 2 * N;
    extern __shared__ __align__(16) double lds[];
    double *sconst = lds;                                      // M, G, du0, dl0 (as in the pack, as lane_kernel keeps them)
    double *sBall = sconst + ((nconst + 1) & ~1);              // b[j][lane] of the four wavefronts (generic loop)
    double *sScr = sBall + 4 * N * 64;                          // screening constants (see fast_scr_doubles)
    double *sPay = sScr + fast_scr_doubles(N, NTHMAX);         // (b_0 .. b_{N-1}, x shift) of the first kFastPay queue positions
    int32_t *ring = reinterpret_cast<int32_t *>(sPay + (size_t)kFastPay * (N + 1));
    int *ctrl = reinterpret_cast<int *>(ring + R * 64);        // [0] write index, [1] read index, [2] producers done
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nth = P.nth;
    for (int i = tid; i < nconst; i += 256) sconst[i] = C[P.oM + i];
    for (int i = tid; i < N * NTHMAX; i += 256) sScr[i] = C[P.oDthP + i];
    if (tid < 2 * N) sScr[N * NTHMAX + tid] = C[P.oBnd + tid];
    if (tid < NTHMAX) sScr[N * NTHMAX + 2 * N + tid] = C[P.oXthP + tid];
    if (tid == 0) sScr[N * NTHMAX + 2 * N + NTHMAX] = C[P.ox0];
    for (int i = tid; i < R * 64; i += 256) ring[i] = -1;
    if (tid < 4) ctrl[tid] = 0;
    __syncthreads();
    const double *sM = sconst, *sG = sM + N * N, *sdu = sG + N * (N + 1) / 2, *sdl = sdu + N;
    double *sB = sBall + wv * N * 64;
    const double ntol = -P.primal_tol;
    const long long ntiles = (nprob + 63) / 64;
#ifdef LMPC_FAST_TRACE      // diagnostic build: 100 MHz timestamps per wavefront into errflag[16 + 8 * wave ..]
    long long *trc = reinterpret_cast<long long *>(errflag + 16) + ((long long)blockIdx.x * 4 + wv) * 8;
    int npass = 0;
#define LMPC_TRC(i) do { if (lane == 0) trc[i] = (long long)wall_clock64(); } while (0)
#else
#define LMPC_TRC(i) do { } while (0)
#endif
    LMPC_TRC(0);
    // Static share + dynamic tail (round 3, tctr != nullptr).  Streams end 12 - 16 us after the launch depending on
    // the XCD a workgroup landed on (tools/fast_trace_cu.py: per-XCD means 11.7 ... 16.4 us in one call, different
    // XCDs slow from batch to batch), and the kernel ends one solving pass after the LAST of them.  A workgroup
    // therefore owns only Rs of its R tiles; the batch's remaining tiles [nstat, ntiles) are handed out one at a time
    // through kFastCtrs global counters (tile = nstat + ticket * kFastCtrs + counter; 128 bytes apart, ~190 tickets
    // each) to the streaming wavefronts that finish their share early.  Tickets are drawn while the previous tile is
    // processed; a workgroup takes at most Dcap of them (its LDS queue has room for Rs + Dcap tiles).  The counter
    // set of the NEXT launch on this handle is cleared here (two sets alternate: no memset in front of the kernel).
    const bool dyn = tctr != nullptr;
    const long long nstat = dyn ? ((long long)gridDim.x * Rs < ntiles ? (long long)gridDim.x * Rs : ntiles) : ntiles;
    if (dyn && blockIdx.x == 0 && tid < kFastCtrs) tctr_next[tid * 32] = 0;
    const long long t0 = (long long)blockIdx.x * (dyn ? Rs : R);
    const long long t1 = t0 + (dyn ? Rs : R) < nstat ? t0 + (dyn ? Rs : R) : nstat;
    const long long base0 = dyn ? 0 : t0 * 64;               // queue entries are problem indices relative to this

    auto load_record = [&](long long pid, double *dst) {
        const long long pc = pid < nprob ? pid : nprob - 1;
        if constexpr (GATHER) {
            const GatherArgs &Ga = P.gat;      // theta = [state; reference; disturbance; control[0:nup]; parameter]; NULL = zeros
            const int o1 = Ga.nx, o2 = o1 + Ga.nr, o3 = o2 + Ga.nd, o4 = o3 + Ga.nup;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                double v = 0.0;
                if (t < o1) v = Ga.state[pc * Ga.nx + t];
                else if (t < o2) { if (Ga.reference) v = Ga.reference[pc * Ga.nr + (t - o1)]; }
                else if (t < o3) { if (Ga.disturbance) v = Ga.disturbance[pc * Ga.nd + (t - o2)]; }
                else if (t < o4) { if (Ga.control) v = Ga.control[pc * Ga.ncontrol + (t - o3)]; }
                else { if (Ga.parameter) v = Ga.parameter[pc * Ga.np + (t - o4)]; }
                dst[t] = v;
            }
        } else {
            const double *src = theta + pc * nth;
#pragma unroll
            for (int t = 0; t < NT; t++) dst[t] = src[t];      // NT == nth (nth <= 16)
        }
    };
    auto write_x = [&](long long pid, const double (&th)[NT], const double (&u)[N], bool with_u) {
        const double *xk = C + P.oXthP;
        for (int k = 0; k < P.nout; k++, xk += NTHMAX) {
            double sh = C[P.ox0 + k];
#pragma unroll
            for (int t = 0; t < NT; t++) sh = __builtin_fma(xk[t], th[t], sh);
            double xs = 0.0;
            if (with_u) {
#pragma unroll
                for (int c = 0; c < N; c++) xs = __builtin_fma(C[P.oRout + k * N + c], u[c], xs);
            }
            X[pid * P.nout + k] = xs + sh;
        }
    };

    // Roles rotate with the workgroup index: a workgroup's wavefronts sit on the four SIMDs of its CU in order,
    // and with a fixed role all dedicated solving wavefronts of a CU -- the VALU-heavy ones -- would share ONE
    // SIMD while the other three only stream.
    const int role = (wv + (int)(blockIdx.x & 3)) & 3;
    // ------------------------------------------------------------------ producer: stream this wavefront's tiles
    if (role < nstr) {
        // A streaming wavefront asks for ONE tile, waits, screens it, asks for the next (kFastAhead = 1).  Measured on
        // the headline batch with no point needing iterations (tools/stream_floor.py; one call, cold HBM, event-timed):
        // this form 20.3 us; the next tile requested before the current one is screened (two register sets used
        // alternately, LMPC_FAST_PP) 21.2 us; two / four / seven tiles requested back to back before the first is
        // consumed (straight-line code, so that the compiler's waits are vmcnt(4 (T - 1 - d)) instead of the vmcnt(0)
        // it puts at the head of every loop with stores behind it: loads and stores share one in-order counter)
        // 22.6 / 22.6 / 26.3 us; each wavefront on a contiguous run of tiles (LMPC_FAST_CONTIG) the same.  More
        // requests in flight per wavefront make THIS access shape slower, not faster: a record is 56 bytes per lane,
        // the four load instructions of a tile touch the same 28 lines one after the other and rely on the 32 KB L1
        // to merge them -- nine streaming wavefronts with one tile each are 31.5 KB.  (Read in address order through a
        // wave-private LDS image instead -- no reuse of lines across instructions -- a call took 31.3 us: the round
        // trip through LDS costs more than it saves.  tools/stream_floor.hip has the shapes in isolation: one tile
        // per wavefront at full occupancy reads the same bytes in 15.7 us, event overhead of 6.3 us included.)
        double buf[kFastAhead][NT];
        auto process = [&](long long tile, const double (&th)[NT], bool live) {
            const long long pid = tile * 64 + lane;
            const bool valid = live && pid < nprob;
            // screening test of screen_kernel: any row of dl + b <= 0 <= du + b violated by more than primal_tol?
            // (the constants come from LDS, uniform addresses, for every tile anew: as scalar loads the compiler
            // kept all ~110 scalar registers' worth of them live, more than there are, parked the kernel's pointers
            // in vector-register lanes instead and paid 60-100 v_readlane per tile -- a third of the pass's vector
            // instructions (436 -> 330 instructions per tile); the empty asm hides that the addresses repeat.
            // Bitwise | instead of ||: no branch per comparison.)
            typedef const double __attribute__((address_space(3))) *lds_cdp;
            unsigned scr_off = (unsigned)(size_t)(lds_cdp)sScr;
            asm volatile("" : "+v"(scr_off));                 // (opaque, in a vector register: ONE base address, immediate offsets)
            const lds_cdp scr = (lds_cdp)(size_t)scr_off;
            bool hard = false;
            // (16-byte reads: ds_read_b128 moves a broadcast pair in half the LDS cycles of the ds_read2_b64 the
            // compiler picks for 8-byte-aligned doubles; one LDS pipe serves the four SIMDs of a CU)
            typedef double v2d __attribute__((ext_vector_type(2)));
            typedef const v2d __attribute__((address_space(3))) *lds_c2p;
            const lds_c2p c2 = (lds_c2p)(size_t)scr_off;
            static_assert(NTHMAX % 2 == 0, "rows of pairs");
            double bq[N];
#pragma unroll
            for (int j = 0; j < N; j++) {
                double row[NTHMAX];
#pragma unroll
                for (int q = 0; q < (NT + 1) / 2; q++) {
                    const v2d v = c2[j * (NTHMAX / 2) + q];
                    row[2 * q] = v.x; row[2 * q + 1] = v.y;
                }
                double acc = 0.0;
#pragma unroll
                for (int t = 0; t < NT; t++) acc = __builtin_fma(row[t], th[t], acc);
                bq[j] = acc;
                const v2d bb = c2[N * (NTHMAX / 2) + j];                   // (du, dl) of row j
                const double vu = (bb.x + acc) - 0.0;
                const double vl = -((bb.y + acc) - 0.0);
                hard = hard | (vu < ntol) | (vl < ntol);
            }
            hard = hard & valid;
            // x0 + Xth theta of the first output: stored by a finished lane, handed over by a queued one
            double sh = scr[N * NTHMAX + 2 * N + NTHMAX];
            {
                double row[NTHMAX];
#pragma unroll
                for (int q = 0; q < (NT + 1) / 2; q++) {
                    const v2d v = c2[(N * NTHMAX + 2 * N) / 2 + q];
                    row[2 * q] = v.x; row[2 * q + 1] = v.y;
                }
#pragma unroll
                for (int t = 0; t < NT; t++) sh = __builtin_fma(row[t], th[t], sh);
            }
            const unsigned long long mask = __ballot(hard);
            // every problem gets a flag HERE -- BEFORE a queued problem is published in the ring, so that the release below
            // orders this store in front of the solving lane's final one (issued after the publish, the
            // provisional flag could land last and leave a solved problem at -8) --, queued ones the provisional
            // EXIT_UNFINISHED that the solving lane overwrites: should a bounded wait below ever run out, no caller reads a stale flag as success (the
            // reference asserts exitflag >= 1, utils.jl:46); an unmasked store of whole lines, as screen_kernel's
#ifdef LMPC_FAST_NT_STORES
            if (valid) __builtin_nontemporal_store(hard ? EXIT_UNFINISHED : EXIT_OPTIMAL, &exitflag[pid]);
#else
            if (valid) exitflag[pid] = hard ? EXIT_UNFINISHED : EXIT_OPTIMAL;
#endif
            if (mask != 0ull) {
                int base = 0;
                if (lane == 0) base = __hip_atomic_fetch_add(&ctrl[0], __popcll(mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                base = __shfl(base, 0);
                if (hard) {
                    const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
                    if (pos < kFastPay) {
                        double *pay = sPay + pos * (N + 1);
#pragma unroll
                        for (int j = 0; j < N; j++) pay[j] = bq[j];
                        pay[N] = sh;
                    }
                    // (release: the payload is in LDS before the index that announces it)
                    __hip_atomic_store(&ring[pos], (int32_t)(pid - base0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            if (valid && !hard) {
                if (P.nout == 1) {
#ifdef LMPC_FAST_NT_STORES
                    __builtin_nontemporal_store(0.0 + sh, &X[pid]);
#else
                    X[pid] = 0.0 + sh;                         // x = x0 + Xth theta (screen_kernel's `0.0 + sh`)
#endif
                } else {
                    double u0[N];
#pragma unroll
                    for (int c = 0; c < N; c++) u0[c] = 0.0;
                    write_x(pid, th, u0, false);
                }
                if (iters) iters[pid] = 1;
                if (active) active[pid * P.words] = 0ull;
            }
        };
#ifdef LMPC_FAST_CONTIG     // each streaming wavefront takes a contiguous run of the workgroup's tiles
        const long long per = (R + nstr - 1) / nstr;
        const long long s0 = t0 + role * per, s1 = s0 + per < t1 ? s0 + per : t1;
        for (long long base = s0; base < s1; base += kFastAhead) {
#pragma unroll
            for (int d = 0; d < kFastAhead; d++) {
                const long long tl = base + d;
                load_record((tl < s1 ? tl : base) * 64 + lane, buf[d]);
            }
#pragma unroll
            for (int d = 0; d < kFastAhead; d++) {
                const long long tl = base + d;
                process(tl < s1 ? tl : base, buf[d], tl < s1);
            }
        }
#elif defined(LMPC_FAST_PP)    // two register sets used alternately, the next tile requested before the current one is consumed
        {
            double bufB[NT];
            long long tile = t0 + role;
            if (tile < t1) {
                load_record(tile * 64 + lane, buf[0]);
                for (;;) {
                    long long nxt = tile + nstr < t1 ? tile + nstr : tile;
                    load_record(nxt * 64 + lane, bufB);
                    process(tile, buf[0], true);
                    if (tile + nstr >= t1) break;
                    tile += nstr;
                    nxt = tile + nstr < t1 ? tile + nstr : tile;
                    load_record(nxt * 64 + lane, buf[0]);
                    process(tile, bufB, true);
                    if (tile + nstr >= t1) break;
                    tile += nstr;
                }
            }
        }
#else
        if (!GATHER && dk >= 2) {
            // Records by LDS-DMA (round 3).  The wavefront owns a ring of dk tile slots in LDS; a tile (64 records,
            // 64 NT 8 bytes, contiguous in theta) is moved by ceil(bytes / 1 KiB) wave-instructions of 16 bytes per
            // lane -- fully coalesced, no vector register involved -- and the lane then reads ITS record from the slot
            // (NT 8-byte LDS reads at a stride of NT doubles).  The pieces of the next dk - 1 tiles are in flight while
            // this one is screened: twice (dk = 2) the bytes in flight of the register path at no register cost, which
            // is what the stream was short of (4.7 TB/s with one 3.5 KB tile per streaming wavefront in flight).
            // Vector-memory operations complete in order, so "all but the youngest (dk - 1) tiles' pieces" = this
            // tile's pieces and every store issued before them: a counted s_waitcnt vmcnt.
            constexpr unsigned TB = (unsigned)fast_tile_bytes(NT);
            constexpr int PIECES = (int)((TB + 1023) / 1024);
            constexpr unsigned LASTB = TB - (PIECES - 1) * 1024u;            // bytes of the last piece (16 per lane)
            typedef const double __attribute__((address_space(3))) *lds_cdp;
            char *dring = reinterpret_cast<char *>(ctrl + 4) + (size_t)role * dk * TB;
            const unsigned dbase = (unsigned)(size_t)(lds_cdp) reinterpret_cast<double *>(dring);
            const char *tbytes = reinterpret_cast<const char *>(theta);
            // whole tiles only: the batch's last, partial tile (one per call) takes the register path below
            const long long tfull = nprob / 64;
            const long long te = t1 < tfull ? t1 : tfull;
            auto issue = [&](long long tile, int slot) {
                const char *src = tbytes + tile * (long long)TB + lane * 16;
#pragma unroll
                for (int p = 0; p < PIECES; p++) {
                    const unsigned dst = __builtin_amdgcn_readfirstlane(dbase + (unsigned)slot * TB + (unsigned)p * 1024u);
                    if (p + 1 < PIECES || LASTB == 1024u || lane * 16 < (int)LASTB) fast_dma16(src + p * 1024, dst);
                }
            };
            // ticket of the dynamic tail (see `dyn` above; the lambda is shared with the register path's loop below)
            int cc = (int)((blockIdx.x * (unsigned)nstr + (unsigned)role) & (kFastCtrs - 1));
            auto take = [&]() -> long long {
                int t = -1;
                if (lane == 0) {
                    if (__hip_atomic_fetch_add(&ctrl[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < Dcap)
                        t = atomicAdd(&tctr[cc * 32], 1);
                }
                t = __builtin_amdgcn_readfirstlane(t);
                const long long tile = nstat + (long long)t * kFastCtrs + cc;
                return (t >= 0 && tile < ntiles) ? tile : -1;
            };
            if (dk == 2) {
                // Two-slot pipeline over "this wavefront's static tiles, then tickets": while tile `cur` is screened the
                // pieces of `nxt` are in flight and -- in the dynamic tail -- the ticket for the tile after it is being
                // drawn (one returning atomic: a vector-memory operation like the pieces, counted with them below).
                // Whole tiles only (the ticket range ends at tfull): the batch's partial tile is handled behind the loop.
                long long sidx = t0 + role, tk = -1;
                bool drew = false;                                           // a ticket's atomic was issued in this iteration
                auto take_full = [&]() -> long long { const long long t = take(); return t < tfull ? t : -1; };
                if (dyn && sidx >= te) tk = take_full();
                auto peek = [&]() -> long long { return sidx < te ? sidx : (dyn ? tk : -1); };
                auto advance = [&]() {                                       // hand the peeked tile out; prefetch behind it
                    drew = false;
                    if (sidx < te) { sidx += nstr; if (dyn && sidx >= te) { tk = take_full(); drew = true; } }
                    else if (dyn && tk >= 0) { tk = take_full(); drew = true; }
                };
                long long cur = peek();
                int sc = 0;
                if (cur >= 0) { issue(cur, 0); advance(); }
                // (the prologue's atomic, if any, is older than everything the loop counts: harmless)
                while (cur >= 0) {
                    const long long nxt = peek();
                    const bool nd = nxt >= 0;
                    if (nd) { issue(nxt, sc ^ 1); advance(); } else drew = false;
                    // all but the youngest operations: the pieces of `nxt` (if requested) and this iteration's ticket
                    if (nd && drew) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES + 1) : "memory");
                    else if (nd) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const double *rec = reinterpret_cast<const double *>(dring + (size_t)sc * TB) + lane * NT;
#pragma unroll
                    for (int t = 0; t < NT; t++) buf[0][t] = rec[t];
                    process(cur, buf[0], true);
                    cur = nxt;
                    sc ^= 1;
                }
                // the batch's last, partial tile: its static owner, or (dynamic tail) the first workgroup's first streamer
                const long long part = dyn ? ((blockIdx.x == 0 && role == 0 && tfull < ntiles) ? tfull : -1)
                                           : ((sidx < t1 && sidx == tfull) ? tfull : -1);
                if (part >= 0) {
                    load_record(part * 64 + lane, buf[0]);
                    process(part, buf[0], true);
                }
            } else {
            const long long first = t0 + role;
            long long ahead = first;                                         // next tile to request
            int sa = 0;                                                      // ... and its slot
            for (int d = 0; d < dk - 1 && ahead < te; d++, ahead += nstr) { issue(ahead, sa); sa = sa + 1 == dk ? 0 : sa + 1; }
            int sc = 0;
            long long tile = first;
            for (; tile < te; tile += nstr) {
                if (ahead < te) { issue(ahead, sa); sa = sa + 1 == dk ? 0 : sa + 1; ahead += nstr; }
                // pieces still allowed in flight: those of the tiles requested after this one
                const long long younger = (ahead - tile) / nstr - 1;
                if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PIECES) : "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const double *rec = reinterpret_cast<const double *>(dring + (size_t)sc * TB) + lane * NT;
#pragma unroll
                for (int t = 0; t < NT; t++) buf[0][t] = rec[t];
                sc = sc + 1 == dk ? 0 : sc + 1;
                process(tile, buf[0], true);
            }
            if (tile < t1) {                                                 // (tile == tfull: the partial one)
                load_record(tile * 64 + lane, buf[0]);
                process(tile, buf[0], true);
            }
            if (dyn) {                                                       // deeper rings: the tail one tile at a time
                long long tl = take();
                while (tl >= 0) {
                    load_record(tl * 64 + lane, buf[0]);
                    const long long nx = take();
                    process(tl, buf[0], true);
                    tl = nx;
                }
            }
            }
        } else
        for (long long base = t0 + role; base < t1; base += (long long)kFastAhead * nstr) {
#pragma unroll
            for (int d = 0; d < kFastAhead; d++) {
                const long long tl = base + (long long)d * nstr;
                load_record((tl < t1 ? tl : base) * 64 + lane, buf[d]);
            }
#pragma unroll
            for (int d = 0; d < kFastAhead; d++) {
                const long long tl = base + (long long)d * nstr;
                process(tl < t1 ? tl : base, buf[d], tl < t1);
            }
        }
#endif
        if (dyn && !(!GATHER && dk >= 2)) {
            // dynamic tail (register path): one tile per ticket, the next ticket in flight while this tile is screened
            int cc = (int)((blockIdx.x * (unsigned)nstr + (unsigned)role) & (kFastCtrs - 1));
            auto take = [&]() -> long long {
                int t = -1;
                if (lane == 0) {
                    if (__hip_atomic_fetch_add(&ctrl[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < Dcap)
                        t = atomicAdd(&tctr[cc * 32], 1);
                }
                t = __builtin_amdgcn_readfirstlane(t);
                const long long tile = nstat + (long long)t * kFastCtrs + cc;
                return (t >= 0 && tile < ntiles) ? tile : -1;
            };
            long long tile = take();
            while (tile >= 0) {
                load_record(tile * 64 + lane, buf[0]);
                const long long nxt = take();
                process(tile, buf[0], true);
                tile = nxt;
            }
        }
        // all of this wavefront's reservations are in the LDS queue ahead of this add (LDS keeps a wave's order)
        if (lane == 0) __hip_atomic_fetch_add(&ctrl[2], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        LMPC_TRC(1);
    }

    // ------------------------------------------------------------------ consumer: claim and solve queued problems
    int spins = 0;
    for (;;) {
        int start = 0, n = 0;
        if (lane == 0) {
            for (;;) {
                // read BEFORE the write index (acquire keeps the order): d == nstr => the write index is final
                const int d = __hip_atomic_load(&ctrl[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int w = lds_load(&ctrl[0]);
                int r = lds_load(&ctrl[1]);
                const int avail = w - r;
                if (avail >= 64 || (d >= nstr && avail > 0)) {
                    const int take = avail < 64 ? avail : 64;
                    if (__hip_atomic_compare_exchange_strong(&ctrl[1], &r, r + take, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_WORKGROUP)) { start = r; n = take; break; }
                } else if (d >= nstr) {
                    n = -1;                                    // nothing left and nothing to come
                    break;
                } else {
                    __builtin_amdgcn_s_sleep(8);
                }
                if (++spins > spinLimit) { n = -2; break; }
            }
        }
        start = __builtin_amdgcn_readfirstlane(start);
        n = __builtin_amdgcn_readfirstlane(n);
        // (system scope: the word lives in host memory, the host looks at it without a copy -- lmpc_check)
        if (n == -2 && lane == 0 && errflag) __hip_atomic_store(errflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (n < 0) break;
#ifdef LMPC_FAST_TRACE
        if (npass < 2) LMPC_TRC(2 + npass);
        npass++;
#endif
#ifdef LMPC_FAST_PRIO
        __builtin_amdgcn_s_setprio(LMPC_FAST_PRIO);           // a solving pass is a latency chain: ahead of the streamers
#endif
        const bool mine = lane < n;
        int rel = -1;
        if (mine) {
            int sp = 0;
            while ((rel = __hip_atomic_load(&ring[start + lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < 0) {
                if (++sp > spinLimit) break;                   // (a reserved slot is written at once: never in practice)
            }
        }
        if (__any(mine && rel < 0)) {
            if (lane == 0 && errflag) __hip_atomic_store(errflag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        const long long pid = mine ? base0 + rel : base0;
        double b[N], u[N];
        double sh0 = C[P.ox0];                                 // x0 + Xth theta of the first output (kept: 2 registers)
        const bool handed = start + lane < kFastPay;           // shifts handed over in LDS by the streaming wavefront
#pragma unroll
        for (int j = 0; j < N; j++) b[j] = 0.0;
        if (handed) {
            const double *pay = sPay + (start + (mine ? lane : 0)) * (N + 1);
#pragma unroll
            for (int j = 0; j < N; j++) b[j] = pay[j];
            sh0 = pay[N];
        }
        if (__any(mine && !handed)) {
            double th0[NT];
            load_record(pid, th0);
            // (scalar loads here: the LDS copy the streaming pass uses made a call 1.7 us SLOWER on this side --
            // a solving pass already lives on LDS reads, 26.8 vs 28.5 us same-box)
            const double *dj = C + P.oDthP;
            double sh1 = C[P.ox0];
#pragma unroll
            for (int j = 0; j < N; j++) {
                double acc = 0.0;
#pragma unroll
                for (int t = 0; t < NT; t++) acc = __builtin_fma(dj[j * NTHMAX + t], th0[t], acc);
                b[j] = handed ? b[j] : acc;
            }
            const double *xk = C + P.oXthP;
#pragma unroll
            for (int t = 0; t < NT; t++) sh1 = __builtin_fma(xk[t], th0[t], sh1);
            sh0 = handed ? sh0 : sh1;
        }
#ifdef LMPC_FAST_TRACE
        if (npass == 1) { if (lane == 0) trc[4] = (long long)(b[0] != 12345.678 ? wall_clock64() : 0); }
#endif
        int iter = 1, nact = 0, wrow[KMAX], flag = EXIT_ITERLIMIT;
        bool wlow[KMAX];
        unsigned long long act = 0ull, low = 0ull;
        const int res = fast_tiers<N, KMAX>(P, sM, sG, sdu, sdl, mine, b, u, iter, wrow, wlow, nact);
        bool solved = res == EXIT_OPTIMAL;
#ifdef LMPC_FAST_TRACE
        if (npass == 1) { if (lane == 0) trc[5] = (long long)(res != 77 ? wall_clock64() : 0); }
#endif
        if (solved) {
            flag = EXIT_OPTIMAL;
#pragma unroll
            for (int i = 0; i < KMAX; i++)
                if (i < nact) { act |= 1ull << wrow[i]; if (wlow[i]) low |= 1ull << wrow[i]; }
        }
#ifndef LMPC_FAST_NO_FALLBACK      // (diagnostic build without it: what the tiers and the stream need on their own)
        if (mine && !solved) {
            // the generic loop of lane_kernel on the lanes the tiers did not finish (from scratch)
#pragma unroll
            for (int j = 0; j < N; j++) sB[j * 64 + lane] = b[j];
            LaneState<N, N> s;
            s.init();
            lane_loop<N, N, N, false>(P, C, sM, sG, sdu, sdl, sB, 64, lane, pid, nullptr, s);
#pragma unroll
            for (int c = 0; c < N; c++) u[c] = s.u[c];
            flag = s.flag; iter = s.iter; act = s.act; low = s.low;
        }
#endif
        if (mine) {
            if (P.nout == 1) {
                double xs = 0.0;
#pragma unroll
                for (int c = 0; c < N; c++) xs = __builtin_fma(C[P.oRout + c], u[c], xs);
                X[pid] = xs + sh0;
            } else {
                double th[NT];                                 // several outputs: the record is read again (an L2 hit)
                load_record(pid, th);
                write_x(pid, th, u, true);
            }
            exitflag[pid] = flag;
            if (iters) iters[pid] = iter;
            if (active) active[pid * P.words] = (act & ~low) | ((act & low) << N);     // m == N <= 6: one word
        }
#ifdef LMPC_FAST_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
    LMPC_TRC(6);
#ifdef LMPC_FAST_TRACE
    // (+ where the wavefront ran: HW_REG_HW_ID (4) and HW_REG_XCC_ID (20), for per-CU / per-XCD statistics)
    if (lane == 0) trc[7] = (long long)npass | ((long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 16) |
                            ((long long)((unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xffu) << 48);
#endif
#undef LMPC_TRC
}

template <int NTHMAX, int NT, int N, bool GATHER = false>
__global__ __launch_bounds__(256, LMPC_FAST_WAVES) void fast_kernel(
    const PackLayout P, const double *__restrict__ C, const double *__restrict__ theta,
    double *__restrict__ X, int32_t *__restrict__ exitflag, int32_t *__restrict__ iters,
    uint64_t *__restrict__ active, long long nprob, int R, int nstr, int32_t *__restrict__ errflag, int spinLimit,
    int dk, int Rs, int Dcap, int32_t *__restrict__ tctr, int32_t *__restrict__ tctr_next) {
    fast_body<NTHMAX, NT, N, GATHER>(P, C, theta, X, exitflag, iters, active, nprob, R, nstr, errflag, spinLimit, dk, Rs, Dcap, tctr,
                                     tctr_next);
}

// SEVERAL batches of the same size in ONE launch (lmpc_solve_batches_device): blockIdx.y picks the batch, its buffers
// come from a table in the kernel arguments.  A launch of the one-batch kernel lives 22 us for a 13 us stream: the
// solving tail, the spread of the streams' ends over the XCDs and the ramp are paid per launch.  Here the workgroups of
// batch b + 1 are dispatched as those of batch b retire, so the tail of one batch runs under the stream of the next
// -- what a caller gets from three streams, inside one launch on one stream.
constexpr int kFastMaxBatches = 8;
struct FastBatches {
    const double *theta[kFastMaxBatches];
    double *x[kFastMaxBatches];
    int32_t *flag[kFastMaxBatches];
};
template <int NTHMAX, int NT, int N>
__global__ __launch_bounds__(256, LMPC_FAST_WAVES) void fast_kernel_multi(
    const PackLayout P, const double *__restrict__ C, const FastBatches B, long long nprob, int R, int nstr,
    int32_t *__restrict__ errflag, int spinLimit, int dk) {
    const int b = blockIdx.y;
    fast_body<NTHMAX, NT, N, false>(P, C, B.theta[b], B.x[b], B.flag[b], nullptr, nullptr, nprob, R, nstr, errflag, spinLimit, dk, R, 0,
                                    nullptr, nullptr);
}

}  // namespace lmpc
