// One-QP-per-wavefront dual active-set kernel: the general path (n <= 127 variables, working sets
// up to 64 rows, m <= 1024 constraints, hard and SOFT rows, BINARY rows by branch and bound),
// in binary64 or binary32 (the reference's generated C has both: codegen.jl:19,31-37,82 `float_type`).
//
// Mapping (gfx950): a 64-lane wavefront owns one parameter point.  Working-set position i lives
// on lane i (its multiplier, right-hand side, pivot, forward-solved right-hand side, constraint id
// are that lane's registers), variable k of the primal iterate u lives on lane k, constraint j on
// lane j % 64 (register slot j / 64).  Only the LDL' factor sits in LDS, column-major with an odd
// leading dimension, so that a lane reads its ROW (forward sweep) at consecutive addresses across
// lanes and its COLUMN (backward sweep) at an odd stride -- both conflict-free.
//
// What the inner loops look like (and why):
//  * every serial chain of the algorithm (triangular solves, the pivot recurrence, u'u, M u) is a
//    chain of dependent fma's; everything that does NOT depend on the chain -- the L entries, the
//    rows of M, the broadcast operands -- is fetched eight steps ahead into registers, so a step
//    costs two v_readlane + one fma (+ the select that keeps finished lanes), not an LDS round trip;
//  * lane masks of the sweeps ("rows behind column t") are scalar shifts handed to the VALU
//    through inverse_ballot: no per-step vector compare;
//  * the forward solve y = L^-1 rhs is kept per position and extended by ONE serial chain when a row
//    is appended (it shares the loop with the new pivot's recurrence); only a removal re-runs the
//    full sweep.  Per entry this is the same fma chain as the full sweep -> identical bits;
//  * reductions: "is anything blocking / violated" is a ballot (the common answer is no); a minimum
//    is four DPP butterflies inside the 16-lane rows plus three scalar-operand mins across rows;
//    the winner is the lowest tied lane/index found with a second ballot;
//  * all control flow is wave-uniform and the compiler is told so (readfirstlane on everything
//    that steers a branch), so loop counters and branch conditions live in SGPRs.
// The shared problem data (M row-major for M_W rows, M transposed for the constraint scan, the
// packed Gram matrix) are staged in LDS per workgroup as far as they fit (LDSC level), the rest is
// read from the one constant buffer through L1/L2.
//
// Every fma chain runs in the same order as the CPU oracle's loops, so the results are
// bit-comparable with it, exactly as for the one-QP-per-lane kernel.
//
// Replaces, per problem: mpc_update_qp (reference codegen/mpc_update_qp.c:1-10), daqp_ldp incl.
// soft constraints ([EXT] libdaqp, called at mpc_update_qp.c:48 / utils.jl:282; rho_soft from
// setup.jl:26), daqp_bnb (mpc_update_qp.c:40-43) and mpc_get_solution (mpc_update_qp.c:14-22).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lmpc_pack.hpp"
#include "lmpc_wave_layout.hpp"

#ifndef LMPC_WAVE_LB
#define LMPC_WAVE_LB 1024   // threads per workgroup the small instantiations are register-budgeted for
#endif

namespace lmpc {

// Diagnostic build (-DLMPC_WAVE_TRACE, tools/wave_trace.sh): shader-clock stamps at the phase boundaries of an
// iteration, summed per phase over all wavefronts into g_wave_trace (read back by lmpc_debug_wave_trace).  The stamps
// are scalar memory instructions with a wait behind them, so they stretch what they measure; shares, not absolutes.
#ifdef LMPC_WAVE_TRACE
__device__ unsigned long long g_wave_trace[16];
#define WVT_DECL long long wvt_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long wvt_prev = (long long)clock64()
#define WVT(k) do { const long long t__ = (long long)clock64(); wvt_acc[k] += t__ - wvt_prev; wvt_prev = t__; } while (0)
#define WVT_COUNT(k) do { wvt_acc[k] += 1; } while (0)
#define WVT_FLUSH do { if (lane == 0) { for (int q__ = 0; q__ < 16; q__++) atomicAdd(&g_wave_trace[q__], (unsigned long long)wvt_acc[q__]); } \
                       } while (0)
#else
#define WVT_DECL do { } while (0)
#define WVT(k) do { } while (0)
#define WVT_COUNT(k) do { } while (0)
#define WVT_FLUSH do { } while (0)
#endif

// ---- wave-level helpers, for double and float ---------------------------------------------------
// a wave-uniform pointer the compiler can see as such: scalar base + per-lane 32-bit offset addressing instead of a
// 64-bit pointer per lane (which it hoists out of the problem loop and then spills)
template <typename T>
__device__ __forceinline__ T *wv_uniform_ptr(T *p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wv_bcast(double v, int src) {
    // `src` is wave-uniform: two v_readlane_b32
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float wv_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

__device__ __forceinline__ int wv_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

// value held identically by all lanes -> scalar registers (tells the compiler it is wave-uniform)
__device__ __forceinline__ double wv_first(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float wv_first(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// DPP lane permutations inside a row of 16 lanes (quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E,
// row_half_mirror = 0x141, row_mirror = 0x140), wave_shl:1 = 0x130 (lane i reads lane i+1)
template <int CTRL> __device__ __forceinline__ int wv_dpp(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL> __device__ __forceinline__ double wv_dpp(double v) {
    return __hiloint2double(wv_dpp<CTRL>(__double2hiint(v)), wv_dpp<CTRL>(__double2loint(v)));
}
template <int CTRL> __device__ __forceinline__ float wv_dpp(float v) {
    return __int_as_float(wv_dpp<CTRL>(__float_as_int(v)));
}
// lane i <- lane i+1 (lane 63 gets 0)
template <typename V> __device__ __forceinline__ V wv_down1(V v) { return wv_dpp<0x130>(v); }

__device__ __forceinline__ double wv_min2(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float wv_min2(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ int wv_min2(int a, int b) { return a < b ? a : b; }

// minimum over the 64 lanes, returned wave-uniform
template <typename V> __device__ __forceinline__ V wv_min(V v) {
    v = wv_min2(v, wv_dpp<0xB1>(v));
    v = wv_min2(v, wv_dpp<0x4E>(v));
    v = wv_min2(v, wv_dpp<0x141>(v));
    v = wv_min2(v, wv_dpp<0x140>(v));            // every lane of a row holds the row's minimum
    const V a = wv_bcast(v, 0), b = wv_bcast(v, 16), c = wv_bcast(v, 32), d = wv_bcast(v, 48);
    return wv_min2(wv_min2(a, b), wv_min2(c, d));
}

// pairwise sum over the 64 lanes, returned wave-uniform: neighbours, pairs of pairs, ... inside the 16-lane rows,
// then (row0 + row1) + (row2 + row3) -- the order the CPU checker restates (its tree64, mode 1)
template <typename V> __device__ __forceinline__ V wv_sum(V v) {
    v = v + wv_dpp<0xB1>(v);
    v = v + wv_dpp<0x4E>(v);
    v = v + wv_dpp<0x141>(v);
    v = v + wv_dpp<0x140>(v);
    const V a = wv_bcast(v, 0), b = wv_bcast(v, 16), c = wv_bcast(v, 32), d = wv_bcast(v, 48);
    return (a + b) + (c + d);
}

__device__ __forceinline__ double wv_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float wv_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Constant-pack loads with a per-lane index go through a BUFFER resource: C[soff + voff] with soff wave-uniform (a
// scalar register) and voff this lane's 32-bit index.  The address is formed by the load unit from the descriptor, so
// no 64-bit per-lane pointer exists for the compiler to hoist out of the loops and spill (plain pointer arithmetic
// left the hot loop with four scratch reloads in front of four dependent loads per iteration), and the many bases
// into the one pack are integer offsets.
typedef unsigned int wv_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double wv_bufld(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, double) {
    const wv_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(voff * 8u), (int)(soff * 8u), 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
__device__ __forceinline__ float wv_bufld(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, float) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(voff * 4u), (int)(soff * 4u), 0));
}

__device__ __forceinline__ bool wv_in(unsigned long long mask) {
    // per-lane predicate from a scalar lane mask: no vector compare
    return __builtin_amdgcn_inverse_ballot_w64(mask);
}
// lanes [0, k)
__device__ __forceinline__ unsigned long long wv_below(int k) { return k >= 64 ? ~0ull : ((1ull << k) - 1ull); }
// the same for 0 <= k <= 63 (a working-set position): one s_bfm_b64 instead of shift, not and a guarded select
__device__ __forceinline__ unsigned long long wv_below63(int k) { return (1ull << (k & 63)) - 1ull; }

template <typename R> struct wv_lim;
template <> struct wv_lim<double> { static __device__ __forceinline__ double inf() { return __builtin_huge_val(); } };
template <> struct wv_lim<float> { static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); } };

// R: arithmetic type.  MR: register slots per lane for constraints (m <= 64*MR).  LDSC: how much of
// the shared problem data is staged once per workgroup in LDS behind the per-wave factors:
// 0 nothing, 1 M transposed (constraint scan), 2 + M (primal step), 3 + packed Gram (row append).
// BNB: rows flagged BINARY must end up active at one of their bounds -- depth-first branch and
// bound over them around the same node solver (what the reference gets from daqp_bnb, [EXT]).
// PACKED: layout of the per-wave factor L, see below.
// NU: register slots per lane for the primal iterate: variable k lives on lane k % 64, slot k / 64 (n <= 64 NU;
// NU = 2 takes the condensed problems of long horizons, n up to 127 -- the reference's own benchmark sweeps
// Np = Nc = 50 .. 125 with one input, docs/src/manual/benchmark.md:4).  The working set still lives on the 64
// lanes: a point whose working set wants more rows ends with EXIT_WSCAP.
// threads per workgroup an instantiation is register-budgeted for (512 VGPRs per SIMD lane are shared by the
// resident wavefronts: 1024 threads -> 4 wavefronts per SIMD, 128 VGPRs; 768 -> 3, 168; 512 -> 2, 256; 256 -> 1)
#ifndef LMPC_WAVE_LB3
#define LMPC_WAVE_LB3 768
#endif
#ifndef LMPC_WAVE_LB4
#define LMPC_WAVE_LB4 512
#endif
#ifndef LMPC_WAVE_CHM5
#define LMPC_WAVE_CHM5 LMPC_WAVE_CHM_BIG
#endif
#ifndef LMPC_WAVE_LB4G
#define LMPC_WAVE_LB4G 768     // Gram-scan form at 4 / 5 slots: built for three wavefronts per SIMD (168 registers)
#endif
// resident wavefronts per SIMD the 1-2 slot binary64 instantiations (no branch and bound) are built for: 4 = what the
// 1024-thread launch bound gives (128 VGPRs); 5 / 6 tell the compiler to stay within 96 / 80 registers (A/B builds)
#ifndef LMPC_WAVE_WPE
#define LMPC_WAVE_WPE 4
#endif
__host__ __device__ constexpr int wave_launch_bound(int MR, bool BNB, bool GRAM = false) {
    if (GRAM && !BNB && (MR == 4 || MR == 5)) return LMPC_WAVE_LB4G;
    return MR >= 7 ? 256 : (MR >= 5 || BNB) ? 512 : (MR == 4 ? LMPC_WAVE_LB4 : (MR == 3 ? LMPC_WAVE_LB3 : LMPC_WAVE_LB));
}

// GRAM: the Gram-scan form ("gram_scan" option).  Same algorithm and decisions, no n-step chain left inside an
// iteration: row values from Gram columns, M_j u = -sum_{i in W} G(j, W_i) lam*_i (|W| terms instead of n; u itself is
// formed once, when the solve ends), the dual objective as sum_i y_i z_i from the factorisation, and the two dot
// products of a row append plus the soft slack as pairwise lane trees (wv_sum) instead of serial chains.  It reads
// the full symmetric Gram matrix (WaveLayout::oGf; LDSC 1 stages THAT) and neither M' nor the packed triangle.
// Bit-comparable with the CPU checker in its mode 1 ("Gram-scan form"), not with mode 0.
// SIM (binary64 without branch and bound only): the closed-loop machinery is compiled in -- plant step fused behind the
// solve, run-ahead of a scenario's consecutive steps, the kept factorisation between steps.  A plain batched solve
// runs the instantiation without it (round 4: the hot loop of the plain solve carried that state through its spills).
template <typename R, int MR, int LDSC, bool BNB, bool PACKED, int NU = 1, bool GRAM = false, bool SIM = true>
__global__ __launch_bounds__(wave_launch_bound(MR, BNB, GRAM))
__attribute__((amdgpu_waves_per_eu((BNB && MR <= 2) ? 3 : ((!BNB && MR <= 2 && sizeof(R) == 8 && LMPC_WAVE_WPE > 4) ? LMPC_WAVE_WPE : 1)))) void wave_kernel(
    const WaveLayout P, const R *__restrict__ C, const int32_t *__restrict__ S,
    const R *__restrict__ theta, R *__restrict__ X, int32_t *__restrict__ exitflag,
    int32_t *__restrict__ iters, uint64_t *__restrict__ active, const uint64_t *__restrict__ warm,
    int32_t *__restrict__ queue, int qchunk_arg, long long nprob,
    const int32_t *__restrict__ list, const int32_t *__restrict__ count, int32_t *__restrict__ count_next,
    long long seg_cap, int32_t *__restrict__ ovf_list, int32_t *__restrict__ ovf_count, const WaveSim sim,
    R *__restrict__ bnb_r, int32_t *__restrict__ bnb_i, int bnb_depth,
    int32_t *__restrict__ queue_next, int32_t *__restrict__ ovf_next, int32_t *__restrict__ ovf_next1,
    unsigned long long *__restrict__ stat, volatile unsigned long long *__restrict__ stat_host) {
    // the ticket counter of the NEXT launch and BOTH overflow counters of the next call on this handle (the other set:
    // see the launcher).  Both, whatever this call is: a call in one pass followed by one in two must not leave the
    // first pass's counter of the set after next uncleared (found by tools/fuzz_closed_loop.py: stale count, list read
    // past its entries)
    if (blockIdx.x == 0 && threadIdx.x == 0) { *queue_next = 0; *ovf_next = 0; *ovf_next1 = 0; }
    // How large did working sets get?  stat (device, cumulative over the handle's launches; sharded): problems finished by
    // this kernel by the largest size their working set reached: up to 24 / 32 / 48 rows / more.  Every launch first
    // copies the counters as they stand into mapped host memory, where the launcher reads them -- without a copy or a
    // synchronisation -- to decide whether a first pass at a smaller capacity pays (lmpc_wave_launch.hpp).
    // (64 shards of four counters, 128 bytes apart: a few thousand wavefronts adding to ONE word at their ends cost a
    // closed-loop step 45 us -- a word takes about 90 atomics per microsecond)
    // (the shards are summed here: four stores into host memory per launch -- 256 of them, one per shard counter, took a
    // launch 10 us to retire over PCIe)
    if (stat != nullptr && stat_host != nullptr && blockIdx.x == 0 && threadIdx.x < 5) {
        unsigned long long sum = 0ull;
        for (int sidx = 0; sidx < 64; sidx++) sum += stat[sidx * 16 + threadIdx.x];
        stat_host[threadIdx.x] = sum;
    }

    // Without binaries bnb_depth carries a flag instead: bit 0 = this is the FIRST of two passes (a smaller working-set
    // capacity, more wavefronts resident); a point that outgrows it is listed for the second pass, which starts it again
    // from the same warm start / kept state -- so this pass leaves both alone.
    extern __shared__ __align__(16) unsigned char lds_raw[];
    R *lds = reinterpret_cast<R *>(lds_raw);
    constexpr int CH = 8;                            // steps fetched ahead of a serial chain
    const int lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
    // wave-uniform by construction; telling the compiler keeps every loop counter of the solve in SGPRs
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = P.n, m = P.m, nth = P.nth, cap = P.cap;
    // L(i,t), i > t, lives at L[cbase(t) + i].  Square layout (cbase(t) = t*ldc, odd ldc): a forward
    // sweep reads consecutive addresses across lanes, a backward sweep (lane i reads its own column)
    // an odd stride -- both conflict-free.  PACKED layout (cbase(t) = t(2cap-1-t)/2 - t - 1: column
    // after column of the strict lower triangle): half the LDS, which is what bounds the resident
    // wavefronts once n reaches ~45 (hybrid MPC, n = 60: 5 -> 10 wavefronts per CU in binary64, +60 %);
    // it costs ~6 % on small problems (index arithmetic, bank conflicts of the column reads), so the
    // host picks it only where it buys residency.
    const int lsize = PACKED ? cap * (cap - 1) / 2 : cap * P.ldc;
    // ... followed by 64 reals per wavefront: the multipliers lam*, written once per iteration and read back as LDS
    // broadcasts by the loops that multiply by one lam*_i per step (Gram scan, primal step) -- one LDS read instead of
    // two v_readlane per step on the vector ALU, which is what bounds this kernel
    const int wsize = lsize + 64;
    R *L = lds + (size_t)wv * wsize;
    R *Lam = L + lsize;
    // ZP (square layout): the factor's storage is kept at exact zeros wherever no entry lives -- on and above the
    // diagonal, and in the rows at and beyond the working set's size.  A sweep step is then v = fma(-L, v_t, v) on
    // ALL lanes: fma(-0, v_t, v) = v leaves the lanes the step does not concern as they are (bit for bit, finite v_t),
    // so the per-step lane mask and the select behind it are gone: 3 vector instructions per step instead of 5.
    constexpr bool ZP = !PACKED;
    if constexpr (ZP) {
        for (int i = lane; i < lsize; i += 64) L[i] = (R)0;
    }
    auto cbase = [&](int t) -> int { return PACKED ? t * (2 * cap - 1 - t) / 2 - t - 1 : t * P.ldc; };
    // entry e of the strict lower triangle stored row after row: (row i, column t), e = i (i - 1) / 2 + t
    auto tri_row = [&](int e, int &i, int &t) {
        int ii = (int)((1.0f + __builtin_sqrtf(1.0f + 8.0f * (float)e)) * 0.5f);
        if (ii * (ii - 1) / 2 > e) ii--;
        else if ((ii + 1) * ii / 2 <= e) ii++;
        i = ii; t = e - ii * (ii - 1) / 2;
    };
    constexpr int kKeepI = 5 * 64;                   // ints of a kept closed-loop state (layout of a B&B snapshot's)
    const R *Mr = C + P.oM, *Mt = C + P.oMt, *G = C + P.oG, *Gf = C + P.oGf;
    if constexpr (GRAM) {
        if (LDSC > 0) {
            R *sc = lds + (size_t)nwv * wsize;
            const int nGf = m * m;
            for (int i = threadIdx.x; i < nGf; i += blockDim.x) sc[i] = C[P.oGf + i];
            Gf = sc;
            __syncthreads();
        }
    } else if (LDSC > 0) {
        R *sc = lds + (size_t)nwv * wsize;
        const int nM = m * n, nG = m * (m + 1) / 2;
        for (int i = threadIdx.x; i < nM; i += blockDim.x) sc[i] = C[P.oMt + i];
        Mt = sc;
        if (LDSC > 1) {
            for (int i = threadIdx.x; i < nM; i += blockDim.x) sc[nM + i] = C[P.oM + i];
            Mr = sc + nM;
        }
        if (LDSC > 2) {
            for (int i = threadIdx.x; i < nG; i += blockDim.x) sc[2 * nM + i] = C[P.oG + i];
            G = sc + 2 * nM;
        }
        __syncthreads();
    }
    const R primal_tol = (R)P.primal_tol, dual_tol = (R)P.dual_tol, zero_tol = (R)P.zero_tol,
            progress_tol = (R)P.progress_tol, rho_soft = (R)P.rho_soft;
    const R kInf = wv_lim<R>::inf();
    const int lrow = lane < cap ? lane : cap - 1;                // clamped row/column of L for prefetches
    const int lrow1 = lane + 1 < cap ? lane + 1 : cap - 1;
    const int lr1 = lrow > 0 ? lrow : 1;                         // row index that is valid in every column read
    const int lz = lane < cap ? lane : 0;                        // ZP: this lane's row; row 0 holds zeros in every column
    const int mycol = cbase(lrow);                               // this lane's own column: L(t, lane) = L[mycol + t]
    // Per-lane indices into the constant pack are UNSIGNED 32-bit offsets from wave-uniform bases: the loads then take
    // the scalar-base + vector-offset form (one offset register each) instead of 64-bit per-lane pointers, which the
    // compiler hoists out of the loops as register pairs and then spills (the n-chain form's scan reloaded four of
    // them from scratch in every iteration, each reload one more round trip in front of its load)
    unsigned lanen[NU];                                        // this lane's variables, clamped into [0, n)
#pragma unroll
    for (int s = 0; s < NU; s++) lanen[s] = (unsigned)(lane + 64 * s < n ? lane + 64 * s : n - 1);
    const __amdgpu_buffer_rsrc_t crs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<R *>(C), 0, P.nC * (int)sizeof(R), 0x00020000);
    auto ldc = [&](unsigned soff, unsigned voff) -> R { return wv_bufld(crs, voff, soff, R()); };   // C[soff + voff]

    int sense0[MR], sense[MR];                       // constraint slots of this lane: as given / of the
    unsigned jc[MR];                                 // current solve (a B&B node adds its fixed binaries)
#pragma unroll
    for (int r = 0; r < MR; r++) {
        const int j = lane + 64 * r;
        sense0[r] = (j < m) ? S[j] : SENSE_IMMUTABLE;
        sense[r] = sense0[r];
        jc[r] = (unsigned)(j < m ? j : m - 1);
    }
    // flags of row j (wave-uniform) from the slot that owns it
    auto sense_of = [&](int j) -> int {
        int v = 0;
#pragma unroll
        for (int r = 0; r < MR; r++) if (r == (j >> 6)) v = sense[r];
        return __builtin_amdgcn_readlane(v, j & 63);
    };

    // Every resident wavefront starts on its own chunk of qchunk consecutive problems; further chunks
    // come from a shared counter (iteration counts vary by 10x inside a batch, a static split leaves
    // a tail of idle SIMDs).  The ticket for the next chunk is drawn at the start of the current one so
    // that its latency is hidden; the host sizes the chunk so that one counter word is never the
    // bottleneck (it serves ~90 atomics per microsecond).
    //
    // Work-list mode (list != nullptr): the screening pass in front (lmpc_screen_kernel.hpp) has finished every
    // problem whose unconstrained optimum is feasible and left the others in the kShards segments of `list`
    // (segment s: count[s * kCountStride] entries from list[s * seg_cap]).  Lane s holds segment s's count;
    // a prefix sum over the lanes turns a position in the concatenated list into (segment, offset).  The
    // problems then go through the same loop, position by position.
    const long long gwaves = (long long)gridDim.x * nwv;
    long long ntotal = nprob;
    int seg_end = 0, seg_beg = 0;
    if (list != nullptr) {
        static_assert(kShards == 64, "one work-list segment per lane");
        // the counter set of the NEXT call is cleared here (the screening pass of that call adds to it)
        if (count_next != nullptr && blockIdx.x == 0 && threadIdx.x < 64) count_next[threadIdx.x * kCountStride] = 0;
        const int c = count[lane * kCountStride];
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        seg_end = incl; seg_beg = incl - c;
        ntotal = (long long)__builtin_amdgcn_readlane(incl, 63);
    }
    // problems per ticket.  Work-list mode: the host cannot know the list's length, the kernel does -- about 64 tickets per
    // resident wavefront over the whole list, at most 8 problems each (round 4: with ONE problem per ticket a list of
    // 10^6 entries drew 10^6 atomics on one word, 11 ms at ~90 per microsecond -- the reference's mass_spring example ran
    // 1.75x SLOWER behind the screening pass than without it; 16 tickets per wavefront, up to 32 problems each, cost the
    // 3-input masses 5 % in the tail: its problems run 32 iterations)
    int qchunk = qchunk_arg;
    if (list != nullptr && queue != nullptr) {
        const long long q = ntotal / (64 * gwaves);
        qchunk = q < 1 ? 1 : (q > 8 ? 8 : (int)q);
    }
    long long chunk = (long long)blockIdx.x * nwv + wv;
    long long idx = chunk * qchunk;
    int kin = 0, ticket = 0;
    WVT_DECL;
    while (idx < ntotal) {
        WVT(15);                                 // (trace: loop overhead)
        if (queue != nullptr && kin == 0 && lane == 0) ticket = atomicAdd(queue, 1);
        long long pid = idx;
        if (list != nullptr) {
            const int sg = __builtin_amdgcn_readfirstlane((int)__popcll(__ballot(seg_end <= (int)idx)));   // segments that end at or before idx
            const int off = (int)idx - __builtin_amdgcn_readlane(seg_beg, sg);
            pid = (long long)__builtin_amdgcn_readfirstlane(list[(long long)sg * seg_cap + off]);
        }
        const R *th = theta + pid * nth;
        // Run-ahead (scenario-asynchronous closed loop, WaveSim with step counters and a horizon): a scenario whose step
        // ended with a non-empty working set stays with this wavefront for its next step -- no work-list round trip,
        // no record, bounds, kept-state traffic in between, and (warm) the factor continues as it stands in LDS, which is
        // what the kept-state restart of the step-synchronous loop does through memory.  These live across its steps:
        int WSi = 0, possoft = 0, posimm = 0, poslow = 0;   // registers of working-set position `lane`
        R D = (R)0, Dinv = (R)0;
        unsigned actb = 0u, lowb = 0u;           // bit r: slot r active / active at its lower bound
        int na = 0, nsoft_act = 0;
        bool again = false, cont = false;        // again: a further step of the same scenario; cont: ... on the factor as it stands
        int kcur = -1;                           // the scenario's step counter once it has been read
        // (a later step reads the record this wavefront has just written: past its own L1)
        // (built for up to six constraint slots per lane: beyond that the registers are gone -- the eight-slot
        // instantiation dropped from two wavefronts per SIMD to one with it; lmpc_simulate_device knows)
        constexpr bool RUNAHEAD = SIM && sizeof(R) == 8 && !BNB && MR <= kWaveRunAheadSlots;
        auto thld = [&](int t) -> R {
            if constexpr (RUNAHEAD) return again ? __hip_atomic_load(th + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : th[t];
            else return th[t];
        };
        for (;;) {
        R b[MR];                                 // b_j = Dth_j . theta   (mpc_update_qp.c:5-6)
#pragma unroll
        for (int r = 0; r < MR; r++) b[r] = (R)0;
        for (int t0 = 0; t0 < nth; t0 += CH) {
            R tv[CH];
#pragma unroll
            for (int q = 0; q < CH; q++) tv[q] = thld(t0 + q < nth ? t0 + q : nth - 1);
#pragma unroll
            for (int r = 0; r < MR; r++) {
                R dv[CH];
                const unsigned drow = jc[r] * (unsigned)nth;
#pragma unroll
                for (int q = 0; q < CH; q++) dv[q] = ldc((unsigned)(P.oDth + (t0 + q < nth ? t0 + q : nth - 1)), drow);
#pragma unroll
                for (int q = 0; q < CH; q++)
                    if (t0 + q < nth) b[r] = wv_fma(dv[q], tv[q], b[r]);
            }
        }
        // the shifted bounds of this lane's rows, once per problem: dupper_j = du_j + b_j, dlower_j = dl_j + b_j
        // (mpc_update_qp.c:7-8).  They used to be re-formed from a load of du / dl in every iteration's violation test
        // and in every row append -- two to four global round trips on the critical path of each iteration.
        R dub[MR], dlb[MR];
#pragma unroll
        for (int r = 0; r < MR; r++) { dub[r] = ldc((unsigned)P.odu, jc[r]) + b[r]; dlb[r] = ldc((unsigned)P.odl, jc[r]) + b[r]; }
        WVT(12);                                 // (trace: record, bounds, b = Dth theta)
        R lam = (R)0, ls = (R)0, rhs = (R)0, y = (R)0;   // (further registers of working-set position `lane`)
        R u[NU];                                 // lane k % 64, slot k / 64 holds u_k
#pragma unroll
        for (int s = 0; s < NU; s++) u[s] = (R)0;
        // u_k for a wave-uniform k
        auto ubc = [&](int k) -> R {
            if constexpr (NU == 1) return wv_bcast(u[0], k);
            else {
                R v = u[0];
#pragma unroll
                for (int s = 1; s < NU; s++) v = (k >> 6) == s ? u[s] : v;
                return wv_bcast(v, k & 63);
            }
        };
        int sing = -1, iter = 1, cyc = 0, flag = EXIT_ITERLIMIT;
        int napk = na;                           // largest working set of this solve (statistics)
        R best = (R)-1, fval = (R)0, soft_slack = (R)0;
        bool done = false, ydirty = false;
        R fbound = (R)P.fval_bound;              // a B&B node stops as soon as it is dominated

        auto Gat = [&](int a, int c) -> R {
            if constexpr (GRAM) {                  // symmetric, same bits either way round
                if constexpr (LDSC > 0) return Gf[c * m + a];
                else return ldc((unsigned)(P.oGf + c * m), (unsigned)a);
            }
            else {
                const unsigned gi = (unsigned)(a >= c ? a * (a + 1) / 2 + c : c * (c + 1) / 2 + a);
                if constexpr (LDSC > 2) return G[gi];
                else return ldc((unsigned)P.oG, gi);
            }
        };
        R MuG[GRAM ? MR : 1];                    // GRAM: row values of the last constraint scan (B&B branches on them)

        // Column sweeps.  forward: v_i -= L(i,t) v_t for t = 0 .. na-2 in order (lane i holds v_i);
        // lane i reads its row of L eight columns ahead of the chain.
        auto sweep_fwd = [&](R v) -> R {
            const unsigned long long rows = wv_below(na);
            for (int t0 = 0; t0 + 1 < na; t0 += CH) {
                R Lr[CH];
#pragma unroll
                for (int q = 0; q < CH; q++) {
                    const int t = t0 + q < cap ? t0 + q : cap - 1;
                    Lr[q] = L[cbase(t) + (ZP ? lz : lr1)];
                }
#pragma unroll
                for (int q = 0; q < CH; q++) {
                    const int t = t0 + q;
                    if (t + 1 < na) {
                        const R vt = wv_bcast(v, t);
                        if constexpr (ZP) v = wv_fma(-Lr[q], vt, v);                 // L(i,t) = 0 outside t < i < na
                        else if (wv_in(rows & (~1ull << t))) v = wv_fma(-Lr[q], vt, v);   // rows t < i < na
                    }
                }
            }
            return v;
        };
        // backward: v_i -= L(t,i) v_t for t = top .. 1 in descending order; lane i reads its column
        auto sweep_bwd = [&](R v, int top) -> R {
            for (int t1 = top; t1 >= 1; t1 -= CH) {
                R Lc[CH];
#pragma unroll
                for (int q = 0; q < CH; q++) {
                    const int t = t1 - q > 1 ? t1 - q : 1;
                    Lc[q] = L[mycol + t];
                }
#pragma unroll
                for (int q = 0; q < CH; q++) {
                    const int t = t1 - q;
                    if (t >= 1) {
                        const R vt = wv_bcast(v, t);
                        if constexpr (ZP) v = wv_fma(-Lc[q], vt, v);                  // L(t,i) = 0 for i >= t
                        else if (wv_in(wv_below63(t))) v = wv_fma(-Lc[q], vt, v);     // columns i < t  (1 <= t <= 63)
                    }
                }
            }
            return v;
        };

        // ZP: rows [from, to) of the factor back to zeros (a row that left the working set, or all of them between
        // two solves): lane t clears its column's entries
        auto clear_rows = [&](int from, int to) {
            if constexpr (ZP) {
                for (int i = from; i < to; i++)
                    if (lane < i) L[mycol + i] = (R)0;
            }
        };

        // ---- branch and bound: snapshots of a node's optimal state, one slot per search depth in this wavefront's slice
        // of global scratch.  A node's factor is written out LAZILY: appends never touch the rows already there, so as
        // long as no row below the node's working-set size has been removed since it branched, its factor IS the leading
        // block of the factor in LDS -- its second child then starts by truncating (no triangle read back), and a subtree
        // that never removes such a row never writes the triangle at all.  The first removal that would break this
        // saves the triangles of the nodes concerned just before it (snap_clean / snap_saved, one bit per depth;
        // stk_na on lane d: that node's working-set size).
        constexpr int kSnapUn = 8;
        int stk_na = 0, bdepth = 0;
        unsigned long long snap_clean = 0ull, snap_saved = 0ull;
        const int snapR = 6 * 64 + cap * (cap - 1) / 2, snapI = 5 * 64;
        R *snr0 = nullptr;
        int32_t *sni0 = nullptr;
        if constexpr (BNB) {
            const long long slot = (long long)blockIdx.x * nwv + wv;
            snr0 = wv_uniform_ptr(bnb_r + slot * (long long)bnb_depth * snapR);
            sni0 = wv_uniform_ptr(bnb_i + slot * (long long)bnb_depth * snapI);
        }
        // rows [0, nd) of the factor as it stands -> slot d.  Whole-wavefront block copies: entry e = i (i - 1) / 2 + t of
        // the strict lower triangle (row after row) goes to word e, 64 consecutive words per instruction, kSnapUn
        // instructions in flight (round 3 copied column by column, 1 .. na lanes each, every load of a restore waited
        // for before the next was issued: the search spent most of its time waiting for its own scratch).
        auto save_tri = [&](int d, int nd) {
            R *sr = snr0 + (long long)d * snapR;
            int32_t *si = sni0 + (long long)d * snapI;
            // ... and the registers of the node's working-set positions, which are still the node's own for lanes < nd
            // (appends write position na only); behind them a node holds zeros
            const bool in = lane < nd;
            sr[64 + lane] = in ? rhs : (R)0; sr[128 + lane] = in ? D : (R)0; sr[192 + lane] = in ? Dinv : (R)0;
            if constexpr (!GRAM) sr[256 + lane] = y;
            si[lane] = in ? WSi : 0; si[64 + lane] = in ? (possoft | (posimm << 1) | (poslow << 2)) : 0;
            const int ne = nd * (nd - 1) / 2;
            for (int e0 = 0; e0 < ne; e0 += 64 * kSnapUn) {
                R v[kSnapUn];
#pragma unroll
                for (int q = 0; q < kSnapUn; q++) {
                    const int e = e0 + 64 * q + lane;
                    int i, t;
                    tri_row(e < ne ? e : 0, i, t);
                    v[q] = L[cbase(t) + i];
                }
#pragma unroll
                for (int q = 0; q < kSnapUn; q++) {
                    const int e = e0 + 64 * q + lane;
                    if (e < ne) sr[384 + e] = v[q];
                }
            }
        };

        // ---- append constraint j (wave-uniform) to the working set
        auto ldl_add = [&](int j, bool lower) {
            const int sj = sense_of(j);
            const bool is_soft = (sj & SENSE_SOFT) != 0;
            const int wsc = lane < na ? WSi : j;
            R q = Gat(wsc, j);
            const R gjj = wv_bcast(q, na < 64 ? na : 63);             // lanes >= na read G(j,j)
            q = lane < na ? q : (R)0;
            q = sweep_fwd(q);
            const R l = q * Dinv;                    // lanes >= na: 0 * 0
            R dnew = na < 64 ? gjj : Gat(j, j);
            if (is_soft) dnew += rho_soft;
            // bound of row j from the slot that owns it
            R bj = (R)0;                             // the bound of row j that enters: from the slot that owns the row
#pragma unroll
            for (int r = 0; r < MR; r++) if (r == (j >> 6)) bj = lower ? dlb[r] : dub[r];
            const R rj = -wv_bcast(bj, j & 63);
            // two serial chains over the old positions: the new pivot and the new entry of y = L^-1 rhs
            R ynew = rj;
            if constexpr (GRAM) {
                // lanes >= na hold l = 0 (q masked, Dinv = 0); y may be stale there
                dnew = dnew - wv_sum(l * q);
                ynew = rj - wv_sum(lane < na ? l * y : (R)0);
            } else {
            for (int i0 = 0; i0 < na; i0 += CH) {
#pragma unroll
                for (int qq = 0; qq < CH; qq++) {
                    const int i = i0 + qq;
                    if (i < na) {
                        const R li = wv_bcast(l, i);
                        dnew = wv_fma(-li, wv_bcast(q, i), dnew);
                        ynew = wv_fma(-li, wv_bcast(y, i), ynew);
                    }
                }
            }
            }
            dnew = wv_first(dnew);
            const bool singular = (dnew < zero_tol) || (!is_soft && (na - nsoft_act) >= n);
            if (lane < na) L[mycol + na] = l;        // new row: L(na, t) written by lane t
            if (lane == na) {
                WSi = j; possoft = is_soft ? 1 : 0; posimm = (sj & SENSE_IMMUTABLE) ? 1 : 0; poslow = lower ? 1 : 0;
                rhs = rj; lam = (R)0; ls = (R)0; y = ynew;
                D = singular ? (R)0 : dnew;
                Dinv = singular ? (R)0 : (R)1 / dnew;
            }
            if (lane == (j & 63)) {
                actb |= 1u << (j >> 6);
                if (lower) lowb |= 1u << (j >> 6);
            }
            if (singular) sing = na;
            nsoft_act += is_soft ? 1 : 0;
            na++;
            napk = na > napk ? na : napk;
        };

        // ---- drop working-set position r (wave-uniform): compact L, rank-one update of the tail
        auto ldl_remove = [&](int r) {
            if constexpr (BNB) {
                // nodes on the current path whose factor is still the leading block of this one and reaches beyond row r
                const unsigned long long hit = __ballot(stk_na > r) & wv_below(bdepth) & snap_clean;
                unsigned long long todo = hit & ~snap_saved;
                while (todo != 0ull) {
                    const int d = (int)__builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    save_tri(d, __builtin_amdgcn_readlane(stk_na, d));
                }
                snap_saved |= hit;
                snap_clean &= ~hit;
            }
            const int nao = na;
            R w = L[cbase(r) + lr1];                 // old row index = lane
            w = (lane > r && lane < nao) ? w : (R)0;
            R alpha = wv_bcast(D, r);
            const int jrem = __builtin_amdgcn_readlane(WSi, r);
            const int softrem = __builtin_amdgcn_readlane(possoft, r);
            // new L(i,c): old L(i+1,c) for c < r, old L(i+1,c+1) for c >= r   (i >= r); eight
            // columns are read before they are written (the reads of the next eight start behind them)
            for (int c0 = 0; c0 + 1 < nao - 1; c0 += CH) {
                R tmp[CH];
#pragma unroll
                for (int qq = 0; qq < CH; qq++) {
                    int c = c0 + qq;
                    c = c < cap - 2 ? c : (cap - 2 > 0 ? cap - 2 : 0);
                    const int srcc = c < r ? c : c + 1;
                    tmp[qq] = L[cbase(srcc) + lrow1];
                }
#pragma unroll
                for (int qq = 0; qq < CH; qq++) {
                    const int c = c0 + qq;
                    if (c + 1 < nao - 1) {
                        const int lo = (c + 1 > r) ? c + 1 : r;
                        if (wv_in(wv_below63(nao - 1) & ~wv_below63(lo))) L[cbase(c) + lane] = tmp[qq];   // lo <= nao - 1 <= 63
                    }
                }
            }
            // shift the per-position registers down by one from position r on
            {
                const int wn = wv_down1(WSi), sn = wv_down1(possoft), in = wv_down1(posimm), ln = wv_down1(poslow);
                const R lamn = wv_down1(lam), rhsn = wv_down1(rhs), Dn = wv_down1(D), Din = wv_down1(Dinv),
                        wnn = wv_down1(w);
                if (lane >= r && lane < nao - 1) {
                    WSi = wn; possoft = sn; posimm = in; poslow = ln; lam = lamn; rhs = rhsn; D = Dn; Dinv = Din;
                    w = wnn;
                } else if (lane == nao - 1) {
                    WSi = 0; possoft = 0; posimm = 0; poslow = 0; lam = (R)0; rhs = (R)0; D = (R)0; Dinv = (R)0; w = (R)0;
                } else {
                    w = (R)0;
                }
            }
            clear_rows(nao - 1, nao);
            na = nao - 1;
            sing = -1;
            ydirty = true;
            bool stop = false;
            for (int t0 = r; t0 < na && !stop; t0 += CH) {
                R lq[CH];
#pragma unroll
                for (int qq = 0; qq < CH; qq++) {
                    const int t = t0 + qq < cap ? t0 + qq : cap - 1;
                    lq[qq] = L[cbase(t) + lr1];
                }
#pragma unroll
                for (int qq = 0; qq < CH; qq++) {
                    const int t = t0 + qq;
                    if (t < na && !stop) {
                        const R pt = wv_bcast(w, t);
                        const R dold = wv_bcast(D, t);
                        const R dbar = wv_fma(alpha * pt, pt, dold);
                        if (dbar < zero_tol) {
                            if (lane == t) { D = (R)0; Dinv = (R)0; }
                            sing = t;
                            stop = true;
                        } else {
                            const R rinv = (R)1 / dbar;
                            const R beta = (pt * alpha) * rinv;
                            alpha = (dold * alpha) * rinv;
                            if (lane == t) { D = dbar; Dinv = rinv; }
                            if (wv_in(wv_below(na) & (~1ull << t))) {
                                w = wv_fma(-pt, lq[qq], w);
                                L[cbase(t) + lane] = wv_fma(beta, w, lq[qq]);
                            }
                        }
                    }
                }
            }
            if (lane == (jrem & 63)) {
                actb &= ~(1u << (jrem >> 6));
                lowb &= ~(1u << (jrem >> 6));
            }
            nsoft_act -= softrem;
        };

        // ---- primal iterate u = -M_W' lam* (lane k owns u_k): rows of M eight positions ahead.  Every iteration in
        // the n-chain form; once, when the solve (or a B&B leaf) ends, in the Gram-scan form
        auto primal_step = [&]() {
            R uk[NU];
#pragma unroll
            for (int s = 0; s < NU; s++) uk[s] = (R)0;
            constexpr int CHU = NU == 1 ? CH : CH / 2;      // same register budget for either NU
            Lam[lane] = ls;
            for (int i0 = 0; i0 < na; i0 += CHU) {
                R mv[CHU][NU], lq[CHU];
#pragma unroll
                for (int q = 0; q < CHU; q++) {
                    const int w = __builtin_amdgcn_readlane(WSi, i0 + q < na ? i0 + q : na - 1);
#pragma unroll
                    for (int s = 0; s < NU; s++) {
                        if constexpr (!GRAM && LDSC > 1) mv[q][s] = Mr[w * n + (int)lanen[s]];
                        else mv[q][s] = ldc((unsigned)(P.oM + w * n), lanen[s]);
                    }
                    lq[q] = Lam[i0 + q < 64 ? i0 + q : 63];
                }
#pragma unroll
                for (int q = 0; q < CHU; q++)
                    if (i0 + q < na) {
#pragma unroll
                        for (int s = 0; s < NU; s++) uk[s] = wv_fma(-mv[q][s], lq[q], uk[s]);
                    }
            }
#pragma unroll
            for (int s = 0; s < NU; s++) u[s] = lane + 64 * s < n ? uk[s] : (R)0;
        };

        // ---- blocking search over the working set: (alpha, rm) = first minimum of the ratios
        auto blocking = [&](bool singular_dir, R &alpha, int &rm) {
            const bool okd = poslow ? (ls < dual_tol) : (ls > -dual_tol);
            const bool blk = (lane < na) && !posimm && !okd;
            const unsigned long long bm = __ballot(blk);
            if (bm == 0ull) { rm = -1; alpha = (R)0; return; }
            R cand = (R)0;
            if (blk) cand = singular_dir ? (-lam / ls) : (-lam / (ls - lam));
            const R g = wv_min(blk ? cand : kInf);
            unsigned long long tie = __ballot(blk && cand == g);
            if (tie == 0ull) tie = bm;
            rm = (int)__builtin_ctzll(tie);
            alpha = wv_bcast(cand, rm);
        };

        // ---- one LDP solve with the flags in sense[].
        // forced < 0: fresh start; the initial working set is the rows flagged ACTIVE (equalities) plus the
        // caller's warm-start mask (closed loop).
        // forced >= 0 (B&B): continue IN PLACE from the parent node's optimal working set, factorisation and
        // multipliers -- as they stand (first child) or as restored from the node's snapshot (second child);
        // row forced>>1 (side forced&1) has just been fixed and enters like a violated row would.
        auto solve_node = [&](int forced) {
        iter = 1; cyc = 0; flag = EXIT_ITERLIMIT; best = (R)-1; done = false;
        if (!BNB || forced < 0) {
        if (!cont) {
        clear_rows(1, na);                       // (rows of the previous solve / node; row 0 has no entries)
        WSi = 0; possoft = 0; posimm = 0; poslow = 0;
        D = (R)0; Dinv = (R)0;
        actb = 0u; lowb = 0u;
        na = 0; nsoft_act = 0;
        }
        lam = (R)0; ls = (R)0; rhs = (R)0; y = (R)0;
#pragma unroll
        for (int s = 0; s < NU; s++) u[s] = (R)0;
        sing = -1;
        fval = (R)0; soft_slack = (R)0; ydirty = false;
        // Closed loop with a kept factorisation (bnb_r / bnb_i double as the per-scenario state of a solve without
        // binaries): the previous step's final working set comes back IN ITS ORDER with its L and D -- what libdaqp's
        // DAQP_WARMSTART does by simply not clearing its workspace (/root/reference/codegen/mpc_update_qp.c:44-54) --
        // instead of being re-appended row by row from the mask.  Only the bounds changed: rhs per position from the
        // shifted bounds, y = L^-1 rhs redone by the first stationary point, multipliers restart at zero.
        // (The state is read and written with nontemporal accesses: 2 KB per listed scenario and step streaming through
        // an L2 that should keep the Gram matrix -- with plain accesses every constraint scan got 1.5x slower.)
        bool kept = false;
        if (cont) {                                  // run-ahead: the factor is where the previous step left it
            kept = true;
            for (int i = 0; i < na; i++) {
                const int j = __builtin_amdgcn_readlane(WSi, i);
                const int lw = __builtin_amdgcn_readlane(poslow, i);
                R bj = (R)0;
#pragma unroll
                for (int r = 0; r < MR; r++) if (r == (j >> 6)) bj = lw ? dlb[r] : dub[r];
                const R rj = -wv_bcast(bj, j & 63);
                if (lane == i) rhs = rj;
            }
            ydirty = na > 0;
        }
        if constexpr (!BNB && SIM) {
            if (!cont && bnb_i != nullptr && warm != nullptr) {
                const int32_t *si = wv_uniform_ptr(bnb_i + pid * kKeepI);
                const R *sr = wv_uniform_ptr(bnb_r + pid * (long long)P.keepStride);
                const int pna = __builtin_amdgcn_readfirstlane(__builtin_nontemporal_load(si + 256));
                if (pna > cap) {                     // (kept by a pass with a larger capacity: that pass takes it again)
                    kept = true;
                    flag = EXIT_WSCAP; done = true;
                } else if (pna >= 0) {
                    kept = true;
                    WSi = __builtin_nontemporal_load(si + lane);
                    const int fl = __builtin_nontemporal_load(si + 64 + lane);
                    possoft = fl & 1; posimm = (fl >> 1) & 1; poslow = (fl >> 2) & 1;
                    actb = (unsigned)__builtin_nontemporal_load(si + 128 + lane);
                    lowb = (unsigned)__builtin_nontemporal_load(si + 192 + lane);
                    D = __builtin_nontemporal_load(sr + lane); Dinv = __builtin_nontemporal_load(sr + 64 + lane);
                    na = pna;
                    napk = pna;
                    nsoft_act = (int)__popcll(__ballot(lane < na && possoft));
                    const int ne = na * (na - 1) / 2;            // the triangle, row after row, is one contiguous run
                    for (int e0 = 0; e0 < ne; e0 += 256) {
                        R tv[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int e = e0 + 64 * q + lane;
                            tv[q] = e < ne ? __builtin_nontemporal_load(sr + 128 + e) : (R)0;
                        }
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int e = e0 + 64 * q + lane;
                            if (e < ne) {
                                int i, t;
                                tri_row(e, i, t);
                                L[cbase(t) + i] = tv[q];
                            }
                        }
                    }
                    for (int i = 0; i < na; i++) {
                        const int j = __builtin_amdgcn_readlane(WSi, i);
                        const int lw = __builtin_amdgcn_readlane(poslow, i);
                        R bj = (R)0;
#pragma unroll
                        for (int r = 0; r < MR; r++) if (r == (j >> 6)) bj = lw ? dlb[r] : dub[r];
                        const R rj = -wv_bcast(bj, j & 63);
                        if (lane == i) rhs = rj;
                    }
                    ydirty = na > 0;
                }
            }
        }
        if (!kept) {
            unsigned long long wm[MR], lm[MR];
#pragma unroll
            for (int r = 0; r < MR; r++) {
                const int j = lane + 64 * r;
                const int sj = sense[r];
                const bool flagged = (j < m) && (sj & SENSE_ACTIVE) != 0;
                bool want = flagged, lower = flagged && (sj & SENSE_LOWER) != 0;
                if (!BNB && warm != nullptr && j < m && !(sj & SENSE_IMMUTABLE)) {
                    const uint64_t *wp = warm + pid * P.words;
                    if ((wp[j >> 6] >> (j & 63)) & 1ull) want = true;
                    else if ((wp[(m + j) >> 6] >> ((m + j) & 63)) & 1ull) { want = true; lower = true; }
                }
                wm[r] = __ballot(want);
                lm[r] = __ballot(lower);
            }
#pragma unroll
            for (int r = 0; r < MR; r++) {
                unsigned long long todo = wm[r];
                while (todo != 0ull && !done) {
                    const int bit = (int)__builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    const int j = 64 * r + bit;
                    const bool lower = (lm[r] >> bit) & 1ull;
                    const int sj = sense_of(j);
                    if (na >= cap) { flag = EXIT_WSCAP; done = true; break; }
                    ldl_add(j, lower);
                    if (sing >= 0) {
                        if (sj & SENSE_IMMUTABLE) { flag = EXIT_OVERDETERMINED; done = true; }
                        else {                              // dependent warm-start row: take it out again
                            na--;
                            sing = -1;
                            clear_rows(na, na + 1);
                            if (lane == na) { WSi = 0; possoft = 0; posimm = 0; poslow = 0; rhs = (R)0; D = (R)0; Dinv = (R)0; y = (R)0; }
                            if (sj & SENSE_SOFT) nsoft_act--;
                            if (lane == (j & 63)) { actb &= ~(1u << (j >> 6)); lowb &= ~(1u << (j >> 6)); }
                        }
                    }
                }
            }
        }
        }

        // ---- first step of a node that continues in place: the freshly fixed row joins the working set
        if (BNB && forced >= 0 && iter < P.iter_limit && na >= cap) { flag = EXIT_WSCAP; done = true; }
        if (BNB && forced >= 0 && iter < P.iter_limit && !done) {
            lam = ls;
            ldl_add(forced >> 1, (forced & 1) != 0);
            if (fval - best < progress_tol) {
                if (++cyc > P.cycle_tol) { flag = EXIT_CYCLE; done = true; }
            } else { best = fval; cyc = 0; }
            iter++;
        }

        // ---- dual active-set iterations
        WVT(0);
        while (!done) {
            if (iter >= P.iter_limit) { flag = EXIT_ITERLIMIT; break; }
            WVT_COUNT(10);
            int rm = -1;
            R alpha = (R)0;
            if (sing < 0) {
                {
                // constrained stationary point (L D L') lam* = rhs: y = L^-1 rhs is kept up to date by
                // ldl_add, only a removal re-runs the forward sweep; then one backward sweep
                if (ydirty) {
                    y = sweep_fwd((lane < na) ? rhs : (R)0);
                    ydirty = false;
                    WVT(1);
                }
                const R acc = sweep_bwd(y * Dinv, na - 1);
                ls = (lane < na) ? acc : (R)0;
                WVT(2);
                blocking(false, alpha, rm);
                WVT(3);
                if (rm < 0) {
                    if constexpr (!GRAM) primal_step();
                    // objective u'u and the row values M u in one pass over the variables
                    R fv = (R)0, soft = (R)0;
                    R Mu[MR];
#pragma unroll
                    for (int r = 0; r < MR; r++) Mu[r] = (R)0;
                    if constexpr (GRAM) {
                        // dual objective lam*' K lam* = sum_i y_i z_i (z = D^-1 y is what the backward sweep started from)
                        fval = wv_sum(lane < na ? y * (y * Dinv) : (R)0);
                        // row values from Gram columns, working-set order: M_j u = -sum_i G(W_i, j) lam*_i
#ifndef LMPC_WAVE_CHG_SCALE
#define LMPC_WAVE_CHG_SCALE 1
#endif
                        constexpr int CHG = LMPC_WAVE_CHG_SCALE * (MR == 1 ? 8 : (MR <= 3 ? 4 : 2));
                        Lam[lane] = ls;                  // (lanes >= na hold 0)
                        for (int i0 = 0; i0 < na; i0 += CHG) {
                            R gv[CHG][MR], lq[CHG];
#pragma unroll
                            for (int q = 0; q < CHG; q++) {
                                const int w = __builtin_amdgcn_readlane(WSi, i0 + q < na ? i0 + q : na - 1);
#pragma unroll
                                for (int r = 0; r < MR; r++) {
                                    if constexpr (LDSC > 0) gv[q][r] = Gf[w * m + (int)jc[r]];
                                    else gv[q][r] = ldc((unsigned)(P.oGf + w * m), jc[r]);
                                }
                                lq[q] = Lam[i0 + q < 64 ? i0 + q : 63];
                            }
#pragma unroll
                            for (int q = 0; q < CHG; q++)
                                if (i0 + q < na) {
#pragma unroll
                                    for (int r = 0; r < MR; r++) Mu[r] = wv_fma(-gv[q][r], lq[q], Mu[r]);
                                }
                        }
#pragma unroll
                        for (int r = 0; r < MR; r++) MuG[r] = Mu[r];
                    } else {
#ifndef LMPC_WAVE_CHM_BIG
#define LMPC_WAVE_CHM_BIG 4
#endif
                    // rows of M' fetched this many variables ahead of the chain: 8 x MR values in flight for the small
                    // instantiations; the many-slot ones (MR >= 4: 256-thread workgroups, two wavefronts per SIMD at
                    // most, M' too large for LDS) read it from L2 and were waiting on those loads for 57 % of their
                    // cycles with two variables in flight (pendulum N = 125: rocprofv3 SQ_WAIT_ANY)
                    constexpr int CHM = MR == 1 ? 8 : (MR == 2 ? 4 : (MR == 5 ? LMPC_WAVE_CHM5 : (MR <= 8 ? LMPC_WAVE_CHM_BIG : 2)));   // (16 slots: 4 would spill)
                    // An empty working set gives u = +0 exactly, and every chain below then ends at +0 (fma(x, +0, +0)
                    // = +0): the n-step pass is skipped.  That is the whole first iteration of every cold solve --
                    // for a closed-loop batch, where most points need one to four iterations, a quarter of the kernel.
                    for (int k0 = 0; k0 < (na > 0 ? n : 0); k0 += CHM) {
                        R mt[CHM][MR];
#pragma unroll
                        for (int q = 0; q < CHM; q++) {
                            const int k = k0 + q < n ? k0 + q : n - 1;
#pragma unroll
                            for (int r = 0; r < MR; r++) {
                                if constexpr (LDSC > 0) mt[q][r] = Mt[k * m + (int)jc[r]];
                                else mt[q][r] = ldc((unsigned)(P.oMt + k * m), jc[r]);
                            }
                        }
#pragma unroll
                        for (int q = 0; q < CHM; q++) {
                            if (k0 + q < n) {
                                const R v = ubc(k0 + q);
                                fv = wv_fma(v, v, fv);
#pragma unroll
                                for (int r = 0; r < MR; r++) Mu[r] = wv_fma(mt[q][r], v, Mu[r]);
                            }
                        }
                    }
                    if (nsoft_act > 0)
                        for (int i = 0; i < na; i++)
                            if (__builtin_amdgcn_readlane(possoft, i)) {
                                const R l = wv_bcast(ls, i);
                                soft = wv_fma(l * l, rho_soft, soft);
                            }
                    soft_slack = wv_first(soft);
                    fval = wv_first(fv) + soft_slack;
                    }
                    WVT(4);
                    if (fval > fbound) { flag = EXIT_INFEASIBLE; break; }
                    R mval = -primal_tol;
                    int midx = -1;
                    bool broken = false;
#pragma unroll
                    for (int r = 0; r < MR; r++) {
                        const int j = lane + 64 * r;
                        if (j < m && !(sense[r] & SENSE_IMMUTABLE)) {
                            const R vu = dub[r] - Mu[r];
                            const R vl = -(dlb[r] - Mu[r]);
                            if (!((actb >> r) & 1u)) {
                                if (vu < mval) { mval = vu; midx = 2 * j; }
                                else if (vl < mval) { mval = vl; midx = 2 * j + 1; }
                            } else if (!(sense[r] & SENSE_SOFT) && (vu < -primal_tol || vl < -primal_tol)) {
                                broken = true;  // the iterate violates a hard row of its own working set
                            }
                        }
                    }
                    const unsigned long long viol = __ballot(midx >= 0);
                    if (viol == 0ull) {
                        if constexpr (GRAM)
                            soft_slack = nsoft_act > 0 ? wv_sum((lane < na && possoft) ? (ls * ls) * rho_soft : (R)0) : (R)0;
                        if (__ballot(broken) != 0ull) flag = EXIT_CYCLE;
                        else flag = (soft_slack > primal_tol) ? EXIT_SOFT_OPTIMAL : EXIT_OPTIMAL;
                        break;
                    }
                    // most violated row: smallest value, ties to the lowest (row, side) index
                    {
                        const R g = wv_min(midx >= 0 ? mval : kInf);
                        unsigned long long tie = __ballot(midx >= 0 && mval == g);
                        if (tie == 0ull) tie = viol;
                        if ((tie & (tie - 1ull)) == 0ull) {
                            midx = __builtin_amdgcn_readlane(midx, (int)__builtin_ctzll(tie));
                        } else {
                            midx = wv_min(wv_in(tie) ? midx : 0x7fffffff);
                        }
                    }
                    // the working set lives on the 64 lanes: a problem whose working set would outgrow
                    // them (n + 1 + #soft > 64 rows possible, more than 64 wanted at once) is given up
                    if (na >= cap) { flag = EXIT_WSCAP; break; }
                    lam = ls;
                    WVT(5);
                    ldl_add(midx >> 1, (midx & 1) != 0);
                    WVT(6);
                    if (fval - best < progress_tol) {
                        if (++cyc > P.cycle_tol) { flag = EXIT_CYCLE; break; }
                    } else { best = fval; cyc = 0; }
                } else {
                    lam = wv_fma(alpha, ls - lam, lam);
                    ldl_remove(rm);
                    WVT(7);
                }
                }
            } else {
                // singular working set: direction p with M_W' p = 0, p_sing = +-1
                const int sg = sing;
                R acc = L[mycol + (sg > 1 ? sg : 1)];
                acc = (lane < sg) ? -acc : (R)0;
                acc = sweep_bwd(acc, sg - 1);
                if (lane == sg) acc = (R)1;
                if (lane > sg) acc = (R)0;
                if (__builtin_amdgcn_readlane(poslow, sg)) acc = -acc;
                ls = acc;
                blocking(true, alpha, rm);
                if (rm < 0) { flag = EXIT_INFEASIBLE; break; }
                lam = wv_fma(alpha, ls, lam);
                WVT(8);
                ldl_remove(rm);
                WVT(7);
            }
            iter++;
        }
        };   // solve_node

        if (!BNB) {
            solve_node(-1);
            if constexpr (GRAM) primal_step();
        } else {
            // Depth-first branch and bound; stack entry d lives on lane d: the row branched on, the side tried
            // first.  BOTH children of a node continue in place from its optimal state: the first right away, the
            // second from a SNAPSHOT of that state taken when the node branched (round 3; until then the second
            // child rebuilt the parent's working set row by row -- 41 rebuilds of ~30 appends per parameter point
            // of the satellite problem, two thirds of the kernel's time).  A snapshot = the registers of the
            // working-set positions, the rows' activity bits and the strict lower triangle of L, in this
            // wavefront's slice of global scratch (one slot per depth; a few KB, written once and read once).
            int stk_j = 0, stk_side = 0, stk_tried = 0;
            // eagerly, a node writes only what an append changes: the multipliers lam*, the rows' activity bits, the scalars
            auto snapshot = [&](int d) {
                R *sr = snr0 + (long long)d * snapR;
                int32_t *si = sni0 + (long long)d * snapI;
                sr[lane] = ls;
                // (Gram-scan form: an entry of y = L^-1 rhs formed at its row's append -- a lane tree -- and the same entry
                // after a later removal re-ran the forward sweep -- a serial chain -- differ in the last bit, so there y
                // is part of what an append-only subtree can change)
                if constexpr (GRAM) sr[256 + lane] = y;
                si[128 + lane] = (int32_t)actb; si[192 + lane] = (int32_t)lowb;
                if (lane == 0) { sr[320] = fval; si[256] = na; si[257] = nsoft_act; }
                if (lane == d) stk_na = na;
                snap_clean |= 1ull << d;             // everything else stays where it is (see save_tri / ldl_remove)
                snap_saved &= ~(1ull << d);
            };
            auto restore = [&](int d) {
                const R *sr = snr0 + (long long)d * snapR;
                const int32_t *si = sni0 + (long long)d * snapI;
                const int naold = na;
                const bool clean = (snap_clean >> d) & 1ull;
                if (!clean) clear_rows(1, naold);
                ls = sr[lane];
                if constexpr (GRAM) y = sr[256 + lane];
                actb = (unsigned)si[128 + lane]; lowb = (unsigned)si[192 + lane];
                fval = wv_first(sr[320]);
                na = __builtin_amdgcn_readfirstlane(si[256]);
                nsoft_act = __builtin_amdgcn_readfirstlane(si[257]);
                if (clean) {
                    // the node's factor and positions are the leading part of what is here: cut what lies behind them
                    clear_rows(na > 1 ? na : 1, naold);
                    if (lane >= na) { WSi = 0; possoft = 0; posimm = 0; poslow = 0; rhs = (R)0; D = (R)0; Dinv = (R)0; }
                } else {
                    rhs = sr[64 + lane]; D = sr[128 + lane]; Dinv = sr[192 + lane];
                    if constexpr (!GRAM) y = sr[256 + lane];
                    WSi = si[lane];
                    const int fl = si[64 + lane];
                    possoft = fl & 1; posimm = (fl >> 1) & 1; poslow = (fl >> 2) & 1;
                    const int ne = na * (na - 1) / 2;
                    for (int e0 = 0; e0 < ne; e0 += 64 * kSnapUn) {
                        R v[kSnapUn];
#pragma unroll
                        for (int q = 0; q < kSnapUn; q++) {
                            const int e = e0 + 64 * q + lane;
                            v[q] = sr[384 + (e < ne ? e : 0)];
                        }
#pragma unroll
                        for (int q = 0; q < kSnapUn; q++) {
                            const int e = e0 + 64 * q + lane;
                            int i, t;
                            tri_row(e < ne ? e : 0, i, t);
                            if (e < ne) L[cbase(t) + i] = v[q];
                        }
                    }
                }
                // (a node is restored once, for its second child: from here on nobody needs slot d's triangle)
                snap_clean &= ~(1ull << d);
                lam = (R)0; sing = -1; ydirty = false;
            };
            int depth = 0, nodes = 0, total_it = 0, have = 0, bflag = EXIT_INFEASIBLE;
            bool inplace = false;
            R ubest[NU], bestval = (R)P.fval_bound;
#pragma unroll
            for (int s = 0; s < NU; s++) ubest[s] = (R)0;
            unsigned bestact = 0u, bestlow = 0u;
            for (;;) {
                if (nodes >= 100000) { bflag = EXIT_ITERLIMIT; break; }
#pragma unroll
                for (int r = 0; r < MR; r++) sense[r] = sense0[r];
                for (int d = 0; d < depth; d++) {
                    const int jf = __builtin_amdgcn_readlane(stk_j, d);
                    const int sd = __builtin_amdgcn_readlane(stk_side, d);
                    if (lane == (jf & 63)) {
#pragma unroll
                        for (int r = 0; r < MR; r++)
                            if (r == (jf >> 6)) sense[r] |= SENSE_ACTIVE | SENSE_IMMUTABLE | (sd ? SENSE_LOWER : 0);
                    }
                }
                fbound = bestval;
                const int dpar = depth > 0 ? depth - 1 : 0;
                int forced = -1;
                if (inplace)
                    forced = 2 * __builtin_amdgcn_readlane(stk_j, dpar) + __builtin_amdgcn_readlane(stk_side, dpar);
                solve_node(forced);
                nodes++;
                total_it += iter;
                if (flag == EXIT_WSCAP) { bflag = EXIT_WSCAP; have = 0; break; }
                bool descend = false;
                if (flag >= 1) {
                    // lowest-index binary row that is not in the final working set
                    int cand = 0x7fffffff;
#pragma unroll
                    for (int r = MR - 1; r >= 0; r--) {
                        const int j = lane + 64 * r;
                        if (j < m && (sense0[r] & SENSE_BINARY) && !((actb >> r) & 1u)) cand = j;
                    }
                    const int jb = wv_min(cand);
                    if (jb == 0x7fffffff) {              // leaf: every binary sits on a bound
                        if (!have || fval < bestval) {
                            if constexpr (GRAM) primal_step();
                            have = 1; bestval = fval; bestact = actb; bestlow = lowb;
#pragma unroll
                            for (int s = 0; s < NU; s++) ubest[s] = u[s];
                        }
                    } else {
                        R Mu = (R)0;
                        if constexpr (GRAM) {
#pragma unroll
                            for (int r = 0; r < MR; r++) if (r == (jb >> 6)) Mu = MuG[r];
                            Mu = wv_bcast(Mu, jb & 63);
                        } else {
                            for (int k = 0; k < n; k++) Mu = wv_fma(Mr[(size_t)jb * n + k], ubc(k), Mu);
                        }
                        R blo = (R)0, bup = (R)0;
#pragma unroll
                        for (int r = 0; r < MR; r++) if (r == (jb >> 6)) { blo = dlb[r]; bup = dub[r]; }
                        const R dlo = wv_bcast(blo, jb & 63), dup = wv_bcast(bup, jb & 63);
                        const int lower_first = (wv_first(Mu) - dlo) < (dup - wv_first(Mu)) ? 1 : 0;
                        if (lane == depth) { stk_j = jb; stk_side = lower_first; stk_tried = 1; }
                        snapshot(depth);                 // this node's optimal state, for its second child
                        depth++;
                        bdepth = depth;
                        descend = true;
                    }
                }
                inplace = descend;
                if (!descend) {                          // backtrack to the next untried side
                    while (depth > 0 && __builtin_amdgcn_readlane(stk_tried, depth - 1) == 2) depth--;
                    bdepth = depth;
                    if (depth == 0) break;
                    if (lane == depth - 1) { stk_side ^= 1; stk_tried = 2; }
                    restore(depth - 1);                  // back to that node's optimal state: its second child, in place
                    inplace = true;
                }
            }
#pragma unroll
            for (int s = 0; s < NU; s++) u[s] = have ? ubest[s] : (R)0;
            actb = have ? bestact : 0u;
            lowb = have ? bestlow : 0u;
            flag = have ? (bflag == EXIT_ITERLIMIT ? EXIT_ITERLIMIT : EXIT_OPTIMAL) : bflag;
            iter = total_it;
        }

        // ---- x = R^-1 u + x0 + Xth theta   (mpc_update_qp.c:14-22); lane k writes outputs k, k + 64, ...
        // (an empty final working set means u = +0 exactly: the R^-1 u chain then ends at +0 and is skipped)
        const int nchain = (!BNB && na == 0) ? 0 : n;
        for (int o0 = 0; o0 < P.nout; o0 += 64) {
            R xs = (R)0;
            const int ko = o0 + lane;
            const int lout = ko < P.nout ? ko : P.nout - 1;
            for (int c0 = 0; c0 < nchain; c0 += CH) {
                R rv[CH];
#pragma unroll
                for (int q = 0; q < CH; q++) rv[q] = ldc((unsigned)(P.oRout + (c0 + q < n ? c0 + q : n - 1)), (unsigned)lout * (unsigned)n);
#pragma unroll
                for (int q = 0; q < CH; q++)
                    if (c0 + q < n) xs = wv_fma(rv[q], ubc(c0 + q), xs);
            }
            R xo = (R)0;
            if (ko < P.nout) {
                R sh = ldc((unsigned)P.ox0, (unsigned)ko);
                for (int t = 0; t < nth; t++) sh = wv_fma(ldc((unsigned)(P.oXth + t), (unsigned)ko * (unsigned)nth), thld(t), sh);
                xo = xs + sh;
                if (X != nullptr) (X + pid * P.nout)[(unsigned)ko] = xo;
            }
            if constexpr (sizeof(R) == 8 && !BNB && SIM) {
                // closed loop with the plant step fused in: advance this scenario in place (WaveSim; lane a < nx forms x+_a,
                // lane l < nu holds u_l).  A point handed to the slow path is advanced there, after its re-solve.
                if (o0 == 0 && sim.FG != nullptr && !(flag == EXIT_WSCAP && ovf_list != nullptr)) {
                    const int snx = sim.nx, snu = sim.nu;
                    const int k = sim.kfix >= 0 ? sim.kfix : (kcur >= 0 ? kcur : sim.kstep[pid]);
                    kcur = k + 1;
                    const int ar = lane < snx ? lane : 0;
                    double acc = 0.0;
                    // (tried: F's row, G's row and the state requested in one batch in front of this chain, likewise the
                    // columns of Xth above -- the exit is a third of a run-ahead step -- but the 40 registers of the batch
                    // spill elsewhere: 1.28e9 -> 1.10e9 scenario-steps/s, n-chain form 1.10e9 -> 7.6e8)
                    for (int c = 0; c < snx; c++) acc = __builtin_fma(sim.FG[ar * snx + c], (double)thld(c), acc);
                    for (int l = 0; l < snu; l++) acc = __builtin_fma(sim.FG[snx * snx + ar * snu + l], (double)wv_bcast(xo, l), acc);
                    double *to = const_cast<double *>(reinterpret_cast<const double *>(theta)) + pid * nth;
                    if (lane < snx) {
                        to[lane] = acc;
                        if (sim.xtraj) sim.xtraj[((long long)(k + 1) * sim.nscen + pid) * snx + lane] = acc;
                    }
                    if (lane < sim.nup) to[snx + sim.nr + lane] = (double)xo;
                    if (lane < snu && sim.utraj) sim.utraj[((long long)k * sim.nscen + pid) * snu + lane] = (double)xo;
                    if (lane == 0) {
                        // (smallest flag so far: an atomic, so that nothing waits for the old value to come back)
                        if (sim.flag_min) { if (k == 0) sim.flag_min[pid] = flag; else atomicMin(&sim.flag_min[pid], flag); }
                        if (sim.kfix < 0) sim.kstep[pid] = k + 1;
                    }
                }
            }
        }
        WVT(13);                                 // (trace: primal step, outputs, plant step)
        // (a point handed to the slow path keeps its mask: in a closed loop `active` IS the warm-start buffer the slow
        // path is about to start from, and what this kernel holds is a working set that was cut off at 64 rows)
        if (active && !(flag == EXIT_WSCAP && ovf_list != nullptr)) {
            unsigned long long wd[2 * MR + 1];
#pragma unroll
            for (int q = 0; q < 2 * MR + 1; q++) wd[q] = 0ull;
#pragma unroll
            for (int r = 0; r < MR; r++) {
                const bool a = (actb >> r) & 1u, lo = (lowb >> r) & 1u;
                const unsigned long long up = __ballot(a && !lo), dn = __ballot(a && lo);
                wd[r] |= up;
                const int pos = m + 64 * r, q0 = pos >> 6, sft = pos & 63;
#pragma unroll
                for (int q = 0; q < 2 * MR + 1; q++) {
                    if (q == q0) wd[q] |= dn << sft;
                    if (q == q0 + 1 && sft) wd[q] |= dn >> (64 - sft);
                }
            }
            if (lane == 0)
#pragma unroll
                for (int q = 0; q < 2 * MR + 1; q++)
                    if (q < P.words) active[pid * P.words + q] = wd[q];
        }
        if (lane == 0) {
            if (exitflag != nullptr) exitflag[pid] = flag;
            if (iters) iters[pid] = iter;
            // a working set that outgrew the 64 lanes: queued for the one-problem-per-thread kernel behind this one
            // (lmpc_big_kernel.hpp), which overwrites the outputs of this problem
            if (flag == EXIT_WSCAP && ovf_list != nullptr) ovf_list[atomicAdd(ovf_count, 1)] = (int32_t)pid;
        }
        bool more = false;                       // run-ahead: this scenario's next step follows right here
        if constexpr (RUNAHEAD) {
            more = sim.FG != nullptr && sim.kfix < 0 && kcur >= 0 && kcur < sim.T && flag >= 1 && na > 0;
        }
        if constexpr (!BNB && SIM) {
            // closed loop: this step's final working set and factor, kept.  (Run-ahead keeps them in LDS; it writes them
            // out when the scenario leaves the kernel, and after every step of a FIRST pass -- a later step that
            // outgrows that pass's capacity restarts from here in the second pass)
            if (bnb_i != nullptr && (!more || (bnb_depth & 1))) {
                int32_t *si = wv_uniform_ptr(bnb_i + pid * kKeepI);
                R *sr = wv_uniform_ptr(bnb_r + pid * (long long)P.keepStride);
                if (flag >= 1) {
                    __builtin_nontemporal_store(WSi, si + lane);
                    __builtin_nontemporal_store(possoft | (posimm << 1) | (poslow << 2), si + 64 + lane);
                    __builtin_nontemporal_store((int32_t)actb, si + 128 + lane);
                    __builtin_nontemporal_store((int32_t)lowb, si + 192 + lane);
                    __builtin_nontemporal_store(D, sr + lane);
                    __builtin_nontemporal_store(Dinv, sr + 64 + lane);
                    if (lane == 0) { __builtin_nontemporal_store(na, si + 256); __builtin_nontemporal_store(nsoft_act, si + 257); }
                    const int ne = na * (na - 1) / 2;
                    for (int e = lane; e < ne; e += 64) {
                        int i, t;
                        tri_row(e, i, t);
                        __builtin_nontemporal_store(L[cbase(t) + i], sr + 128 + e);
                    }
                } else if (lane == 0 && !(flag == EXIT_WSCAP && ovf_list != nullptr && (bnb_depth & 1))) {
                    __builtin_nontemporal_store(-1, si + 256);      // failed or handed to the slow path: the next step starts from the mask
                }
            }
        }
        // (one fire-and-forget atomic per finished problem, on the wavefront's shard: counters kept in registers across
        // the problems of a wavefront cost the kernel more -- spills -- than these)
        if (stat != nullptr && lane == 0 && !(flag == EXIT_WSCAP && ovf_list != nullptr)) {
            atomicAdd(&stat[((blockIdx.x * nwv + wv) & 63) * 16 + (napk <= 24 ? 0 : (napk <= 32 ? 1 : (napk <= 48 ? 2 : 3)))], 1ull);
            if (napk <= 16) atomicAdd(&stat[((blockIdx.x * nwv + wv) & 63) * 16 + 4], 1ull);   // (... and how many within 16 rows: the row kernel's smallest first pass)
        }
        WVT(14);                                 // (trace: masks, flags, kept state)
        cont = more && warm != nullptr;
        if (!cont) clear_rows(1, na);            // ZP: the next solve starts on a factor of zeros
        WVT(9);
        WVT_COUNT(11);
        if (!more) break;
        again = true;
        }   // steps of one scenario
        if (++kin < qchunk) {
            idx++;
        } else {
            kin = 0;
            chunk = queue != nullptr ? gwaves + (long long)__builtin_amdgcn_readfirstlane(ticket) : chunk + gwaves;
            idx = chunk * qchunk;
        }
    }
    WVT_FLUSH;                                   // (trace: once per wavefront -- per problem the 16 atomics per flush, all
                                                 // wavefronts on the same 16 words, held up the next problem's loads)
}

}  // namespace lmpc
