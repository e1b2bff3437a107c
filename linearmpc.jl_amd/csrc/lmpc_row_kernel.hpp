// FOUR QPs per wavefront: the dual active-set kernel on the DPP rows of a wavefront (round 5).
//
// The one-QP-per-wavefront kernel (lmpc_wave_kernel.hpp) is issue bound on mostly empty wavefronts: a step of one of
// its serial chains is 2 v_readlane + 1 fma with ONE useful lane, a working set of <= 31 rows leaves half the lanes
// idle by construction (config 3: 25 000 vector + 20 000 scalar instructions per problem, 8 % of the lane-flops they
// offer are the algorithm's).  Here a problem owns a ROW of 16 lanes -- the unit the DPP cross-lane network works on --
// and a wavefront carries four problems through the iteration in lock step:
//
//  * working-set position p of a problem lives on lane p % 16 of its row, register slot p / 16 (S slots: working sets
//    of up to 16 S rows); variable k on lane k % 16, slot k / 16 (NS slots); constraint j on lane j % 16, slot j / 16
//    (MS slots);
//  * the operand a chain step broadcasts -- v_t of a triangular sweep, lam*_i of the primal step, u_k of the constraint
//    scan -- is delivered PER ROW by the DPP control row_newbcast:t (lane t of each row to all lanes of that row; the
//    only DPP control the 64-bit ALU takes, gfx90a and later): one v_mov_b64_dpp (or the modifier of a v_fmac_f64_dpp)
//    serves four problems, where the wavefront kernel spends two v_readlane and an SGPR round trip on one.  t must be a
//    literal, so every chain is unrolled over its positions and skipped block-wise by wave-uniform tests;
//  * what is one number per problem (working-set size, iteration count, step length, ...) is a ROW-UNIFORM vector
//    register; the four problems of a wavefront take different decisions, so the phases of an iteration (stationary
//    point, blocking test, then row append OR row removal) run under per-row predicates -- selects, never EXEC masks
//    around DPP instructions;
//  * the factor L of each problem sits in LDS as its strict lower triangle, column after column, every column padded
//    upwards to a multiple of four rows (rowp_p0 below): a sweep step is ONE v_fmac_f64_dpp per slot of positions --
//    the broadcast is the instruction's own DPP control, the lanes outside the column are its bank mask or read
//    padding zeros -- and 32 problems' factors fit a CU's LDS (two wavefronts per SIMD); the two rows a 32-lane LDS
//    phase serves sit 16 (mod 32) reals apart, so they never share a bank;
//  * a row that finishes a problem writes its outputs and takes the next problem from the batch by itself (tickets of
//    a few problems each from one global counter), the other three rows keep iterating.
//
// Arithmetic: statement for statement the fma chains of the CPU checker (its mode 0, the n-chain form; the file the wavefront kernel is checked against) -- the
// results are bit-identical to the oracle's and to the wavefront kernel's (x, exit flag, iteration count, active set).
//
// Covers: cold plain solves, binary64 / binary32, n <= 16 NS, m <= 16 MS, hard / SOFT / IMMUTABLE rows, no rows
// flagged ACTIVE, working sets up to CAPP <= 16 S rows; a point that outgrows that is listed for the wavefront kernel
// (exit flag -7 inside this pass), exactly like the first of that kernel's two passes.  Rows flagged BINARY: the BNB
// instantiations, one depth-first search per row of the wavefront (see the search state below).
//
// Replaces, per problem: mpc_update_qp (reference codegen/mpc_update_qp.c:1-10), daqp_ldp incl. soft constraints
// ([EXT] libdaqp, called at mpc_update_qp.c:48 / utils.jl:282) and mpc_get_solution (mpc_update_qp.c:14-22).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "lmpc_pack.hpp"
#include "lmpc_wave_layout.hpp"
#include "lmpc_wave_kernel.hpp"      // (the wave-level helpers wv_*)

namespace lmpc {

// Diagnostic build (-DLMPC_ROW_TRACE, tools/row_trace.py): shader-clock stamps at the phase boundaries of a trip, summed
// per phase over all wavefronts into g_row_trace (read back by lmpc_debug_row_trace).  Shares, not absolutes.
#ifdef LMPC_ROW_TRACE
__device__ unsigned long long g_row_trace[32];
#define RWT_DECL long long rwt_acc[32] = {0}; long long rwt_prev = (long long)clock64()
#define RWT(k) do { const long long t__ = (long long)clock64(); rwt_acc[k] += t__ - rwt_prev; rwt_prev = t__; } while (0)
#define RWT_COUNT(k, n) do { rwt_acc[k] += (n); } while (0)
#define RWS_BEGIN const long long rws_t0__ = (long long)clock64()
#define RWS_END(k) do { rwt_acc[k] += (long long)clock64() - rws_t0__; } while (0)
#define RWT_FLUSH do { if (lane == 0) { for (int q__ = 0; q__ < 32; q__++) atomicAdd(&g_row_trace[q__], (unsigned long long)rwt_acc[q__]); } } while (0)
#else
#define RWT_DECL do { } while (0)
#define RWT(k) do { } while (0)
#define RWT_COUNT(k, n) do { } while (0)
#define RWS_BEGIN do { } while (0)
#define RWS_END(k) do { } while (0)
#define RWT_FLUSH do { } while (0)
#endif

// ---- row-level helpers ---------------------------------------------------------------------------------------------
// lane T of each 16-lane row to all lanes of that row (DPP row_newbcast:T)
template <int T, typename V> __device__ __forceinline__ V rw_bc(V v) {
    return __builtin_amdgcn_mov_dpp(v, 0x150 + (T & 15), 0xF, 0xF, false);
}
template <int I, int N, class F> __device__ __forceinline__ void rw_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rw_static_for<I + 1, N>(f);
    }
}
// descending: N-1 ... I
template <int I, int N, class F> __device__ __forceinline__ void rw_static_rfor(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, N - 1>{});
        rw_static_rfor<I, N - 1>(f);
    }
}
// minimum over the 16 lanes of each row, returned in every lane of the row
template <typename V> __device__ __forceinline__ V rw_min(V v) {
    v = wv_min2(v, wv_dpp<0xB1>(v));
    v = wv_min2(v, wv_dpp<0x4E>(v));
    v = wv_min2(v, wv_dpp<0x141>(v));
    v = wv_min2(v, wv_dpp<0x140>(v));
    return v;
}
__device__ __forceinline__ int rw_or(int v) {
    v |= wv_dpp<0xB1>(v);
    v |= wv_dpp<0x4E>(v);
    v |= wv_dpp<0x141>(v);
    v |= wv_dpp<0x140>(v);
    return v;
}
// lane `src` (0..15, row-uniform, run time) of each row to all lanes of the row: through the LDS crossbar
__device__ __forceinline__ int rw_pick(int v, int src, int rowbase) {
    return __builtin_amdgcn_ds_bpermute((rowbase + (src & 15)) << 2, v);
}
__device__ __forceinline__ float rw_pick(float v, int src, int rowbase) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((rowbase + (src & 15)) << 2, __float_as_int(v)));
}
__device__ __forceinline__ double rw_pick(double v, int src, int rowbase) {
    const int a = (rowbase + (src & 15)) << 2;
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(a, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(a, __double2loint(v)));
}
// lane i <- lane i + 1 inside a row; lane 15 (whose source lies outside the row) keeps `edge`
__device__ __forceinline__ int rw_shl1(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x101, 0xF, 0xF, false); }
__device__ __forceinline__ float rw_shl1(float v, float edge) { return __int_as_float(rw_shl1(__float_as_int(v), __float_as_int(edge))); }
__device__ __forceinline__ double rw_shl1(double v, double edge) {
    return __hiloint2double(rw_shl1(__double2hiint(v), __double2hiint(edge)), rw_shl1(__double2loint(v), __double2loint(edge)));
}
// maximum over the four rows of a row-uniform int, wave-uniform (scalar)
__device__ __forceinline__ int rw_max4(int v) {
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32),
              d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
__device__ __forceinline__ int rw_min4(int v) {
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32),
              d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
__device__ __forceinline__ bool rw_any(bool p) { return __ballot(p) != 0ull; }

constexpr int kRowPosFlagSoft = 1 << 16, kRowPosFlagImm = 1 << 17, kRowPosFlagLow = 1 << 18;
constexpr int kRowBig = 0x7fffffff;
// The factor of a problem in LDS: the strict lower triangle column after column, each column padded UPWARDS to a row
// index that is a multiple of four: column t holds rows p0(t) = 4 floor(t / 4) .. capp-1, of which p0(t) .. t are zeros
// that are never written.  Entry (p, t) sits at cbm(t) + p.  A sweep step then needs no lane mask and no select: the
// lanes of the 4-lane DPP banks entirely outside a column are switched off by the instruction's bank_mask (a literal),
// the at most three lanes left in the bank of the diagonal read the padding.  538 reals for 31 rows (the square,
// zero padded layout of the wavefront kernel: 930) -- which is what lets a CU hold the factors of 32 problems instead
// of 16, i.e. two wavefronts per SIMD.
//
// The ORDER of the columns in memory is free, and for 31 rows it is chosen (by a backtracking search, tools/row_layout.py)
// so that the 16 columns of a slot of positions start at 16 different offsets modulo 16: a lane that reads its own
// COLUMN (backward sweep, row append, compaction) then never shares an LDS bank with another lane of its 32-lane phase.
// In column order the starts collide four ways (8 LDS cycles an access instead of 2: rocprofv3 counted more bank-conflict
// cycles than LDS instructions).  For 32, 16, 48 and 44 rows every column length is a multiple of four and no order helps; the
// 16-, 44- and 48-row layouts (binary32 branch and bound: 54 000 conflict cycles per search in column order) therefore pads every
// column by one entry and orders them likewise (tools/row_layout.py 48).
__host__ __device__ constexpr int rowp_p0(int t) { return t & ~3; }
__host__ __device__ constexpr int rowp_cb(int capp, int t) { return 4 * (t >> 2) * capp - 8 * (t >> 2) * ((t >> 2) - 1) + (t & 3) * (capp - 4 * (t >> 2)); }
// (48 rows: every column one entry longer than its rows need -- odd lengths, without which no order separates the starts)
__host__ __device__ constexpr int rowp_size(int capp) { return capp == 48 ? 1291 : (capp == 44 ? 1095 : (capp == 16 ? 171 : rowp_cb(capp, capp - 1))); }      // columns 0 .. capp-2
__host__ __device__ constexpr int rowp_cbm(int capp, int t) {
    constexpr int k31[30] = {379, 152, 183, 0, 237, 485, 458, 210, 318, 431, 102, 260, 121, 19, 348, 49,
                             507, 79, 275, 64, 329, 390, 401, 30, 282, 292, 408, 492, 295, 285};
    constexpr int k48[47] = {807, 0, 898, 687, 117, 264, 1001, 758, 605, 1164, 430, 1242, 356, 219, 467, 1201,
                             638, 956, 389, 1063, 314, 543, 1030, 34, 160, 849, 923, 1123, 72, 505, 285, 564,
                             177, 824, 51, 484, 713, 1085, 700, 1098, 135, 126, 514, 1072, 182, 319, 5};
    // (44 rows: the same, for searches whose working sets stay within the binaries + 4 rows -- small enough for EIGHT
    // wavefronts' factors per CU next to M' in binary32)
    constexpr int k44[43] = {322, 9, 776, 277, 525, 223, 1050, 363, 620, 867, 400, 983, 929, 182, 830, 484,
                             554, 583, 896, 136, 713, 107, 645, 451, 68, 30, 797, 1004, 662, 946, 417, 47,
                             675, 731, 688, 149, 232, 77, -36, 426, 82, 718, 1009};
    // (16 rows likewise: the one-slot shapes counted 0.8 conflict cycles per LDS instruction in column order)
    constexpr int k16[15] = {69, 0, 30, 52, 13, 82, 105, 118, 127, 136, 154, 145, 92, 35, 87};
    if (capp == 48) return k48[t < 46 ? t : 46];
    if (capp == 44) return k44[t < 42 ? t : 42];
    if (capp == 16) return k16[t < 14 ? t : 14];
    return capp == 31 ? k31[t < 30 ? t : 29] : rowp_cb(capp, t) - rowp_p0(t);
}
// reals between the factors of two problems, for nwv wavefronts per workgroup: the two DPP rows a 32-lane LDS phase
// serves (problems nwv slots apart) must sit 16 reals apart modulo 32
__host__ __device__ constexpr int row_ps(int capp, int nwv) {
    int ps = rowp_size(capp);
    while ((nwv * ps) % 32 != 16) ps++;
    return ps;
}
// factors copied as 16-byte vectors (branch and bound's snapshots)?  Not the binary32 44-row shape: see row_kernel
__host__ __device__ constexpr bool row_copy16(int capp, int rs) { return !(capp == 44 && rs == 4); }
// ... and a multiple of four on top (branch and bound copies factors as 16-byte vectors); -1: no such spacing for this nwv
__host__ __device__ constexpr int row_ps4(int capp, int nwv) {
    int ps = (rowp_size(capp) + 3) & ~3;
    for (int q = 0; q < 16; q++, ps += 4)
        if ((nwv * ps) % 32 == 16) return ps;
    return -1;
}
// reals per row of the staged M' (even, half of it odd): row k holds, for lane i = 0 .. 15, the MS entries M'(k, i + 16 r)
// side by side
// (binary32 with constraint slots in fours: a multiple of four, so that a lane's four entries are one 16-byte load)
__host__ __device__ constexpr int row_mpad(int ms, int rs = 8) { return 16 * ms + ((rs == 4 && ms % 4 == 0) ? 4 : 2); }
// Everything a launch passes, in ONE block: the kernel copies the few scalars its iterations need into registers and
// reads the rest -- the pointers of the outputs, the work list, the counters -- from the kernel-argument segment where
// it uses them (a row takes or ends a problem once in ~8 trips), through a pointer the compiler cannot see through.
// (As ~25 separate arguments they all stayed live in scalar registers across the whole loop: 230 scalar spills, reloaded
// by v_readlane in the middle of every phase.)
template <typename R> struct RowParams {
    WaveLayout P;
    const R *C; const int32_t *Sg; const R *theta;
    R *X; int32_t *exitflag, *iters; uint64_t *active;
    int32_t *queue; int qchunk; int ps; long long nprob;
    const int32_t *list, *count; int32_t *count_next; long long seg_cap;
    int32_t *ovf_list, *ovf_count, *queue_next, *ovf_next, *ovf_next1;
    unsigned long long *stat; volatile unsigned long long *stat_host;
    R *bnb_r; int32_t *bnb_i; int bnb_depth;      // branch and bound: snapshot slots (row_snap_reals / row_snap_ints each), depths per row
    int aux;                                      // (ten-slot shapes: the per-problem constants are staged in LDS; the host has sized for them)
};
// bytes of LDS in front of the staged constants (row_aux_offset) and their size in reals: everything between the bounds
// and the full Gram matrix of the constant pack (lmpc_api.hip: odu, odl, oDth, oRout, ox0, oXth are contiguous)
__host__ __device__ constexpr size_t row_aux_offset(size_t front) { return (front + 15) & ~(size_t)15; }
__host__ __device__ inline int row_aux_reals(const WaveLayout &P) { return P.oGf - P.odu; }
// what one (lazy) snapshot of a branch-and-bound node takes, per problem row of the grid and search depth: reals = right-hand
// side, D, 1/D, y per position and the padded triangle as it lies in LDS; ints = the positions' rows, one size
__host__ __device__ constexpr int row_snap_reals(int s, int capp) { return 64 * s + ((rowp_size(capp) + 3) & ~3); }
__host__ __device__ constexpr int row_snap_ints(int s) { return 16 * s + 16; }
// binary rows a search on the row kernel can hold (its depth; the per-depth state lives in registers)
__host__ __device__ constexpr int row_bnb_depth_max(int ms) { return (16 * ms < 48 ? 16 * ms : 48) - 1; }

// threads per workgroup an instantiation is built for: 512 (two wavefronts per SIMD, 256 registers) up to six constraint
// slots; the ten-slot one takes 256 (one per SIMD: its bounds and row values alone are 60 registers, and its M' leaves
// LDS for five wavefronts' factors anyway)
// ... and the three-slot shape in binary64 (branch and bound at 48 rows: 300 + registers)
// (the ONE-slot shape with ten constraint slots -- a first pass of 16 rows -- fits 256 registers: two per SIMD)
__host__ __device__ constexpr int row_launch_bound(int ms, int s = 2, int rs = 4) { return ((ms > 6 && s >= 2) || (s >= 3 && rs == 8)) ? 256 : 512; }
// resident wavefronts per SIMD an instantiation is register-budgeted for: three for one slot of positions (small factors:
// LDS allows them, 168 registers), two for two slots with up to six constraint slots, one beyond
#ifndef LMPC_ROW_WPS1
#define LMPC_ROW_WPS1 3
#endif
__host__ __device__ constexpr int row_waves_per_simd(int s, int ms, int rs = 4) { return ((ms > 6 && s >= 2) || (s >= 3 && rs == 8)) ? 1 : ((s == 1 && ms <= 6) ? LMPC_ROW_WPS1 : 2); }

// One sweep step as ONE statement of inline assembly (no builtin reaches v_fmac_f64_dpp, and the wait states have to sit
// right in front of the instruction): NOP + 1 wait states (2 behind a vector instruction that wrote `src`; the compiler
// knows no DPP hazard inside inline assembly), then, with b = lane T of src's row,
//   every "other" slot:  oth_k += b * (-loth_k)   in all lanes,
//   the source's slot:   src   += b * (-lsrc)     in the lanes of the banks BM (the last instruction: it rewrites the register
//                                                 the others read their broadcast from).
#define RW_STEP_DEFS(RT, SUF)                                                                                                      \
    template <int T, int BM, int NOP> __device__ __forceinline__ void rw_step1(RT &src, RT lsrc) {                                 \
        asm volatile("s_nop %2\n\tv_fmac_" SUF "_dpp %0, %0, -%1 row_newbcast:%3 row_mask:0xf bank_mask:%4"                        \
                     : "+v"(src) : "v"(lsrc), "n"(NOP), "n"(T), "n"(BM));                                                          \
    }                                                                                                                              \
    template <int T, int BM, int NOP> __device__ __forceinline__ void rw_step2(RT &src, RT &o1, RT lsrc, RT l1) {                  \
        asm volatile("s_nop %4\n\tv_fmac_" SUF "_dpp %1, %0, -%3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"                   \
                     "v_fmac_" SUF "_dpp %0, %0, -%2 row_newbcast:%5 row_mask:0xf bank_mask:%6"                                    \
                     : "+v"(src), "+v"(o1) : "v"(lsrc), "v"(l1), "n"(NOP), "n"(T), "n"(BM));                                       \
    }                                                                                                                              \
    template <int T, int BM, int NOP> __device__ __forceinline__ void rw_step3(RT &src, RT &o1, RT &o2, RT lsrc, RT l1, RT l2) {   \
        asm volatile("s_nop %6\n\tv_fmac_" SUF "_dpp %1, %0, -%4 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"                   \
                     "v_fmac_" SUF "_dpp %2, %0, -%5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"                               \
                     "v_fmac_" SUF "_dpp %0, %0, -%3 row_newbcast:%7 row_mask:0xf bank_mask:%8"                                    \
                     : "+v"(src), "+v"(o1), "+v"(o2) : "v"(lsrc), "v"(l1), "v"(l2), "n"(NOP), "n"(T), "n"(BM));                    \
    }                                                                                                                              \
    template <int T, int BM, int NOP>                                                                                              \
    __device__ __forceinline__ void rw_step4(RT &src, RT &o1, RT &o2, RT &o3, RT lsrc, RT l1, RT l2, RT l3) {                      \
        asm volatile("s_nop %8\n\tv_fmac_" SUF "_dpp %1, %0, -%5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"                   \
                     "v_fmac_" SUF "_dpp %2, %0, -%6 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"                               \
                     "v_fmac_" SUF "_dpp %3, %0, -%7 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"                               \
                     "v_fmac_" SUF "_dpp %0, %0, -%4 row_newbcast:%9 row_mask:0xf bank_mask:%10"                                   \
                     : "+v"(src), "+v"(o1), "+v"(o2), "+v"(o3) : "v"(lsrc), "v"(l1), "v"(l2), "v"(l3), "n"(NOP), "n"(T), "n"(BM)); \
    }
RW_STEP_DEFS(double, "f64")
RW_STEP_DEFS(float, "f32")
#undef RW_STEP_DEFS
// a step of a sweep over S slots: source position T0 (slot T0 / 16), the NO other slots OFF .. OFF + NO - 1
template <int T0, int BM, int NOP, int NO, int OFF, typename RT, int S>
__device__ __forceinline__ void rw_step(RT (&v)[S], const RT (&l)[S]) {
    constexpr int st = T0 >> 4, T = T0 & 15;
    if constexpr (NO == 0) rw_step1<T, BM, NOP>(v[st], l[st]);
    else if constexpr (NO == 1) rw_step2<T, BM, NOP>(v[st], v[OFF], l[st], l[OFF]);
    else if constexpr (NO == 2) rw_step3<T, BM, NOP>(v[st], v[OFF], v[OFF + 1], l[st], l[OFF], l[OFF + 1]);
    else rw_step4<T, BM, NOP>(v[st], v[OFF], v[OFF + 1], v[OFF + 2], l[st], l[OFF], l[OFF + 1], l[OFF + 2]);
}
// In front of and behind a sweep: five wait states with the sweep's vector as an operand.  In front: a scalar write of
// EXEC (the end of a divergent region) must be five states away from the first DPP instruction, a vector write of the
// vector two; behind: the compiler's own DPP reads of the vector must be two states behind the last step.  Inside a
// sweep a step's own two states suffice: between two blocks there are only a scalar compare, a uniform branch and loads.
template <typename V, int S> __device__ __forceinline__ void rw_sweep_fence(V (&v)[S]) {
    if constexpr (S == 1) asm volatile("s_nop 4" : "+v"(v[0]));
    else if constexpr (S == 2) asm volatile("s_nop 4" : "+v"(v[0]), "+v"(v[1]));
    else if constexpr (S == 3) asm volatile("s_nop 4" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
    else asm volatile("s_nop 4" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}
// banks (of four lanes) of slot s that hold rows >= p0(t) / columns <= t
__host__ __device__ constexpr int rw_bm_rows(int s, int t) {
    int bm = 0;
    for (int b = 0; b < 4; b++) bm |= (16 * s + 4 * b >= rowp_p0(t)) ? (1 << b) : 0;
    return bm;
}
__host__ __device__ constexpr int rw_bm_cols(int s, int t) {
    int bm = 0;
    for (int b = 0; b < 4; b++) bm |= (16 * s + 4 * b <= t) ? (1 << b) : 0;
    return bm;
}

// R: arithmetic type.  S / NS / MS: register slots of 16 working-set positions / variables / constraints.  CAPP: rows the
// factor is laid out for (the launch's capacity P.cap <= CAPP <= 16 S).
// BNB: rows flagged BINARY end up active at one of their bounds -- the wavefront kernel's depth-first search (its header),
// one search per row of the wavefront, every row at its own node.
template <typename R, int S, int NS, int MS, int CAPP, bool BNB = false>
__global__ __launch_bounds__(row_launch_bound(MS, S, (int)sizeof(R))) __attribute__((amdgpu_waves_per_eu(row_waves_per_simd(S, MS, (int)sizeof(R))))) void row_kernel(const RowParams<R> prm) {
    static_assert(S >= 1 && S <= 4, "one to four slots of working-set positions");
    // (lanes beyond the last row -- one with 31 / 47 rows, four with 44 -- read whatever lies behind a column: finite values
    // that stay in those lanes, see the masks where a sweep's result is kept)
    static_assert(CAPP <= 16 * S && CAPP > 16 * (S - 1), "rows of the factor live on S slots");
    // the launch's parameter block as the rare phases read it (the same bytes as `prm`)
#if defined(__HIP_DEVICE_COMPILE__)
    const RowParams<R> *kparg = (const RowParams<R> *)__builtin_amdgcn_kernarg_segment_ptr();
#else
    const RowParams<R> *kparg = &prm;
#endif
#define RW_ARGS() ([&]() { const RowParams<R> *a__ = kparg; asm volatile("" : "+s"(a__)); return a__; }())
    // the counters of the next launch / call on this handle, and the working-set statistics' host copy: the duties of
    // every wavefront-kernel launch (lmpc_wave_kernel.hpp, lmpc_wave_launch.hpp)
    {
        const RowParams<R> *a = RW_ARGS();
        if (blockIdx.x == 0 && threadIdx.x == 0) { *a->queue_next = 0; *a->ovf_next = 0; *a->ovf_next1 = 0; }
        if (a->stat != nullptr && a->stat_host != nullptr && blockIdx.x == 0 && threadIdx.x < 5) {
            unsigned long long sum = 0ull;
            for (int sidx = 0; sidx < 64; sidx++) sum += a->stat[sidx * 16 + threadIdx.x];
            a->stat_host[threadIdx.x] = sum;
        }
    }

    extern __shared__ __align__(16) unsigned char lds_raw[];
    R *lds = reinterpret_cast<R *>(lds_raw);
    const int lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int li = lane & 15;
    const int rowbase = lane & 48, g = lane >> 4;
    // (scalars of the iterations: each in a register of its own -- as fields of the by-value block they are sub-registers
    // of 16-wide loads that get spilled and reloaded whole)
    int n = prm.P.n, m = prm.P.m, nth = prm.P.nth, cap = prm.P.cap, nout = prm.P.nout, oG = prm.P.oG;
    int iter_limit = prm.P.iter_limit, cycle_tol = prm.P.cycle_tol;
    asm volatile("" : "+s"(n), "+s"(m), "+s"(nth), "+s"(cap), "+s"(nout), "+s"(oG), "+s"(iter_limit), "+s"(cycle_tol));
    const int PS = prm.ps;                                     // reals between two problems' factors (row_ps)
    // LDS: [32 zeros][M': ceil4(n) rows of MPAD reals, zero padded][factors: nwv * 4 problems, PS reals each][2 zeros][sense flags: m ints]
    // ONE copy of the problem matrix serves both passes over it: the constraint scan reads ROW k of M' -- a lane's MS
    // constraints side by side, 16-byte loads (ds_read_b128 moves 256 bytes a clock, the ds_read2_b64 the compiler makes
    // of strided 8-byte loads 128) -- the primal step reads COLUMN w of it (lane = variable, stride MPAD: even with an
    // odd half, so the 16 lanes of a row hit 16 different bank pairs).  MPAD is a compile-time constant: every address
    // of the scan is one per-lane base register plus an immediate (with a run-time stride the compiler keeps one
    // address register per (k, slot) alive across the whole kernel: 64 to 320 registers).
    static_assert(MS % 2 == 0, "constraint slots in pairs (16-byte loads)");
    constexpr int MPAD = row_mpad(MS, (int)sizeof(R));
    const int nP = (n + 3) & ~3;
    const int oZ = 0, oMt = 32, oL = oMt + nP * MPAD;
    int32_t *sens = reinterpret_cast<int32_t *>(lds + oL + nwv * 4 * PS + 2);
    int32_t *cbt = sens + m;                                   // cbm(t), t = 0 .. CAPP-2, for run-time column indices
    const R *__restrict__ C = prm.C;
    for (int i = threadIdx.x; i < 32; i += blockDim.x) lds[oZ + i] = (R)0;
    for (int i = threadIdx.x; i < nP * MPAD; i += blockDim.x) {
        const int k = i / MPAD, e = i - k * MPAD, j = (e / MS) + 16 * (e % MS);      // e = lane * MS + slot
        lds[oMt + i] = (k < n && e < 16 * MS && j < m) ? C[prm.P.oMt + k * m + j] : (R)0;
    }
    for (int i = threadIdx.x; i < nwv * 4 * PS + 2; i += blockDim.x) lds[oL + i] = (R)0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) sens[i] = prm.Sg[i];
    // Ten constraint slots (the reference's benchmark class): a row takes or ends a problem every fifth trip there, and what
    // that reads per problem -- 70 entries of Dth a lane, the bounds, a row of R^-1 -- came from L2 in four to six round
    // trips with the other three rows of the wavefront waiting (a third of the kernel's time at N = 50).  Where LDS has
    // The ten-slot shapes stage these constants as well (the host sizes the workgroup so that they fit, or leaves the
    // problem to the wavefront kernel) and read them by a 32-bit LDS index like everything else.
    constexpr bool AUXS = MS >= 10;
    const int oAux = (int)(row_aux_offset((size_t)((const unsigned char *)(cbt + CAPP) - lds_raw)) / sizeof(R)) - prm.P.odu;
    if constexpr (AUXS) {
        const int na_ = row_aux_reals(prm.P);
        for (int i = threadIdx.x; i < na_; i += blockDim.x) lds[oAux + prm.P.odu + i] = C[prm.P.odu + i];
    }
    rw_static_for<0, CAPP - 1>([&](auto T) { if (threadIdx.x == 0) cbt[decltype(T)::value] = rowp_cbm(CAPP, decltype(T)::value); });
    __syncthreads();

    const R primal_tol = (R)prm.P.primal_tol, dual_tol = (R)prm.P.dual_tol, zero_tol = (R)prm.P.zero_tol,
            progress_tol = (R)prm.P.progress_tol, rho_soft = (R)prm.P.rho_soft, fbound = (R)prm.P.fval_bound;
    const R kInf = wv_lim<R>::inf();
    // constant pack through a buffer resource: scalar base + this lane's 32-bit index, no 64-bit per-lane pointers
    // (lmpc_wave_kernel.hpp)
    const __amdgpu_buffer_rsrc_t crs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<R *>(C), 0, prm.P.nC * (int)sizeof(R), 0x00020000);
    auto ldc = [&](int soff, int voff) -> R { return wv_bufld(crs, (unsigned)voff, (unsigned)soff, R()); };   // C[soff + voff]
    // ... the per-problem constants (bounds, Dth, R^-1 rows, x0, Xth): from LDS where they are staged (ten-slot shapes)
    auto ldk = [&](int soff, int voff) -> R {
        if constexpr (AUXS) return lds[oAux + soff + voff];
        else return ldc(soff, voff);
    };
    // The unrolled chains below run in BLOCKS of a few steps behind one wave-uniform test each; inside a block every step
    // runs (steps beyond a row's sizes multiply by the zeros the padding guarantees).  The empty statement keeps the
    // compiler from folding a block's test into selects -- which turns the whole unrolled chain into one basic block
    // whose loads are all hoisted to its head (first build: 390 registers).
#define RW_BLOCK() asm volatile("" ::: "memory")

    // element (row p, column t) of this row's factor: lds[Lg + cbm(t) + p]
    auto cbm = [&](int t) -> int { return cbt[t]; };            // (run-time t in [0, CAPP-2])
    const int Lg = oL + (wv + nwv * g) * PS;
    int pos[S], fo[S], bo[S];          // this lane's positions; offsets of its ROW (forward sweeps) and of its COLUMN
#pragma unroll
    for (int s = 0; s < S; s++) {
        pos[s] = li + 16 * s;
        fo[s] = Lg + pos[s];                                            // (a lane beyond the last row reads the next column's padding)
        bo[s] = pos[s] < CAPP - 1 ? Lg + cbm(pos[s] < CAPP - 1 ? pos[s] : 0) : oZ;   // (beyond the last column: the block of zeros)
    }
    int jc[MS];
    unsigned binb = 0u;                // bit r: this lane's row of slot r is BINARY
    unsigned okb = 0u, hardb = 0u;     // bit r: this lane's row of slot r can enter a working set / ... and is a hard row
#pragma unroll
    for (int r = 0; r < MS; r++) {
        const int j = li + 16 * r;
        jc[r] = j < m ? j : m - 1;
        const int sj = sens[jc[r]];
        if (j < m && !(sj & SENSE_IMMUTABLE)) { okb |= 1u << r; if (!(sj & SENSE_SOFT)) hardb |= 1u << r; }
        if (BNB && j < m && (sj & SENSE_BINARY)) binb |= 1u << r;
    }
    // this lane's variables: rows li, li + 16, ... of M' (a lane beyond n reads on into the factors, or zeros beyond the
    // allocation: values that its masked result never shows) -- one address register, the slots are immediates
    const int mcol0 = oMt + li * MPAD;
    const int mrow = oMt + li * MS;                             // this lane's constraints: MS consecutive entries of a row of M'

    // ---- the state of this row's problem (what is one number per problem is a row-uniform vector register; predicates
    // are 0 / 1 integers -- as lane masks they would each hold a pair of scalar registers across the phases)
    int live = 0, dead = 0;                        // a problem is running / the batch is exhausted
    int pid = 0, na = 0, sing = -1, iter = 1, cyc = 0, nsoft = 0, napk = 0, ydirty = 0;
    R best = (R)-1, fval = (R)0, soft_slack = (R)0;
    int ws[S], wof[S];                             // per position: row index | kRowPosFlag*; the row's offset inside a row of M'
    R D[S], Dinv[S], lam[S], ls[S], rhs[S], y[S];
    R u[NS], dub[MS], dlb[MS];
    unsigned actb = 0u, lowb = 0u;                 // bit r: this lane's row of slot r is active / active at its lower bound
#pragma unroll
    for (int s = 0; s < S; s++) { ws[s] = 0; wof[s] = 0; D[s] = Dinv[s] = lam[s] = ls[s] = rhs[s] = y[s] = (R)0; }
#pragma unroll
    for (int s = 0; s < NS; s++) u[s] = (R)0;
#pragma unroll
    for (int r = 0; r < MS; r++) { dub[r] = (R)0; dlb[r] = (R)0; }

    // ---- problems: tickets of qchunk consecutive problems per row; the first from the row's index in the grid, further
    // ones from the shared counter (drawn one ticket ahead)
    // Work-list mode (list != nullptr): a pass in front (screening, lmpc_screen_kernel.hpp) has finished what needs no
    // iterations and left the others in the kShards segments of `list` (segment s: count[s * kCountStride] entries from
    // list[s * seg_cap]); lane s of every wavefront holds segment s's range of positions in the concatenated list.
    const int nrows = (int)gridDim.x * nwv * 4;
    const int myrow = ((int)blockIdx.x * nwv + wv) * 4 + g;
    int ntotal, qchunk, seg_end = 0, seg_beg = 0, ticket = 0;
    {
        const RowParams<R> *a = RW_ARGS();
        ntotal = (int)a->nprob;
        qchunk = a->qchunk;
        if (a->list != nullptr) {
            static_assert(kShards == 64, "one work-list segment per lane");
            if (a->count_next != nullptr && blockIdx.x == 0 && threadIdx.x < 64) a->count_next[threadIdx.x * kCountStride] = 0;
            const int c = a->count[lane * kCountStride];
            int incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            seg_end = incl; seg_beg = incl - c;
            ntotal = __builtin_amdgcn_readlane(incl, 63);
            const int q = ntotal / (32 * nrows);                     // (the host cannot know the list's length)
            qchunk = q < 1 ? 1 : (q > 16 ? 16 : q);
        }
        if (a->queue != nullptr && li == 0) ticket = atomicAdd(a->queue, 1);
    }
    int cur = myrow * qchunk, endc = cur + qchunk < ntotal ? cur + qchunk : ntotal;
    int chunk_static = myrow;
    // ---- sweeps over the factor.  forward: v_p -= L(p,t) v_t for t = 0 .. top-1 in order; backward: v_i -= L(t,i) v_t
    // for t = top .. 1 descending.  `top` is wave-uniform (the largest of the four rows); a row whose working set is
    // smaller reads zeros (its factor's rows beyond na are zeros).  A step is one v_fmac_f64_dpp per slot of target
    // positions: the multiplier v_t arrives as the instruction's row broadcast, the lanes outside the column are switched
    // off by its bank mask or read the column's padding (see rowp_p0).
    // The factor's entries of a block of CHS steps are fetched while the block before it runs (issued at its head, waited
    // for at their first use).
#ifndef LMPC_ROW_CHS
#define LMPC_ROW_CHS 4
#endif
    // (the binary32 three-slot shape -- branch and bound at 48 rows -- measured 1 % faster with blocks of eight)
    constexpr int CHS = (sizeof(R) == 4 && S >= 3) ? 8 : LMPC_ROW_CHS;
    auto sweep_fwd = [&](R (&v)[S], int nmax) {
        constexpr int NB = (CAPP - 1 + CHS - 1) / CHS;               // steps t = 0 .. CAPP-2
        R Ln[CHS][S];
        auto fetch = [&](auto B) {
            constexpr int t0 = decltype(B)::value * CHS;
#pragma unroll
            for (int q = 0; q < CHS; q++)
#pragma unroll
                for (int s = 0; s < S; s++) Ln[q][s] = lds[fo[s] + rowp_cbm(CAPP, t0 + q < CAPP - 1 ? t0 + q : CAPP - 2)];
        };
        fetch(std::integral_constant<int, 0>{});
        rw_sweep_fence(v);
        rw_static_for<0, NB>([&](auto B) {
            constexpr int b = decltype(B)::value, t0 = b * CHS;
            if (t0 + 1 < nmax) {
                RW_BLOCK();
                R Lr[CHS][S];
#pragma unroll
                for (int q = 0; q < CHS; q++)
#pragma unroll
                    for (int s = 0; s < S; s++) Lr[q][s] = Ln[q][s];
                if constexpr (b + 1 < NB) fetch(std::integral_constant<int, b + 1>{});
                rw_static_for<0, CHS>([&](auto Q) {
                    constexpr int q = decltype(Q)::value, t = t0 + q;
                    constexpr int NOP = 1;
                    if constexpr (t + 1 < CAPP) {                 // rows above the source's slot entirely, then its own slot
                        constexpr int st = t >> 4;
                        rw_step<t, rw_bm_rows(st, t), NOP, S - 1 - st, st + 1>(v, Lr[q]);
                    }
                });
                RW_BLOCK();
            }
        });
        rw_sweep_fence(v);
    };
    auto sweep_bwd = [&](R (&v)[S], int top) {
        constexpr int NB = (CAPP - 1 + CHS - 1) / CHS;              // steps t = CAPP-1 .. 1 in blocks from the top
        R Ln[CHS][S];
        auto fetch = [&](auto B) {
            constexpr int thi = CAPP - 1 - decltype(B)::value * CHS;
#pragma unroll
            for (int q = 0; q < CHS; q++)
#pragma unroll
                for (int s = 0; s < S; s++) Ln[q][s] = lds[bo[s] + (thi - q > 1 ? thi - q : 1)];
        };
        // (the first block that runs -- which one depends on `top` -- is fetched here, with a run-time row index)
        {
            const int b0 = top >= 1 ? (CAPP - 1 - top) / CHS : 0;
            const int thi0 = CAPP - 1 - b0 * CHS;
#pragma unroll
            for (int q = 0; q < CHS; q++)
#pragma unroll
                for (int s = 0; s < S; s++) Ln[q][s] = lds[bo[s] + (thi0 - q > 1 ? thi0 - q : 1)];
        }
        rw_sweep_fence(v);
        rw_static_for<0, NB>([&](auto B) {
            constexpr int b = decltype(B)::value;
            constexpr int thi = CAPP - 1 - b * CHS;                  // this block: t = thi .. thi-CHS+1
            constexpr int tlo = thi - CHS + 1 > 1 ? thi - CHS + 1 : 1;
            if (top >= tlo) {
                RW_BLOCK();
                R Lc[CHS][S];
#pragma unroll
                for (int q = 0; q < CHS; q++)
#pragma unroll
                    for (int s = 0; s < S; s++) Lc[q][s] = Ln[q][s];
                if constexpr (b + 1 < NB) fetch(std::integral_constant<int, b + 1>{});
                rw_static_for<0, CHS>([&](auto Q) {
                    constexpr int q = decltype(Q)::value, t = thi - q;
                    constexpr int NOP = 1;
                    if constexpr (t >= 1) {                       // columns below the source's slot entirely, then its own slot
                        constexpr int st = t >> 4;
                        rw_step<t, rw_bm_cols(st, t), NOP, st, 0>(v, Lc[q]);
                    }
                });
                RW_BLOCK();
            }
        });
        rw_sweep_fence(v);
    };

    RWT_DECL;
    // ---- branch and bound (BNB): the search state of this row's problem, all of it in registers.  The stack entry of
    // depth d sits on lane d % 16, slot d / 16: row branched on (bits 0-9), side being tried (10), the node's working-set
    // size (13-19) and its count of soft rows (20-26); next to it the node's objective (stkf) and, in every lane, one byte
    // per depth with the activity bits of the lane's rows at that node (abyte).  What an append changes about a node is
    // thereby kept without memory traffic: its second child starts by cutting the working set back (and takes the
    // node's multipliers from one backward sweep over the restored factor: in the trip that backtracks, see the add
    // phase; a backtrack decided behind the trip -- the rare moves -- from the next trip's sweep, `rls`).  The positions' registers and the
    // factor go to global scratch only before the first removal that would disturb them (snap_clean / snap_saved, one
    // bit per depth: the wavefront kernel's lazy snapshots); `tried2`: both sides of that depth's row have been tried.
    static_assert(!BNB || MS <= 4, "a byte per depth holds the activity bits of up to four rows per lane");
    constexpr int DS = BNB ? MS : 1;                             // (a search is at most m <= 16 MS levels deep ...
    constexpr int AW = BNB ? (4 * DS < 12 ? 4 * DS : 12) : 1;    //  ... and at most 4 AW <= 48: row_bnb_depth_max)
    int forced = -1, depth = 0, nodes = 0, total_it = 0, have = 0, bflag = EXIT_INFEASIBLE, jbX = kRowBig, sideX = 0, rls = 0;
    R bestval = fbound, ubest[NS], stkf[DS];
    unsigned bestact = 0u, bestlow = 0u, abyte[AW];
    int stk[DS], fixd[MS];                                       // fixd[r]: the depth that fixed this lane's row of slot r (255: none)
    unsigned long long snap_clean = 0ull, snap_saved = 0ull, tried2 = 0ull;
#pragma unroll
    for (int s = 0; s < NS; s++) ubest[s] = (R)0;
#pragma unroll
    for (int q = 0; q < DS; q++) { stk[q] = 0; stkf[q] = (R)0; }
#pragma unroll
    for (int q = 0; q < AW; q++) abyte[q] = 0u;
#pragma unroll
    for (int r = 0; r < MS; r++) fixd[r] = 255;
    constexpr int kSnapR = row_snap_reals(S, CAPP), kSnapI = row_snap_ints(S);
    // (the selects' operands pass through an empty statement: the compiler otherwise recognises an indexed array, keeps a
    // copy of it in scratch memory and answers every look-up with a load from there)
    auto stk_get = [&](int d) -> int {
        int sel = stk[0];
#pragma unroll
        for (int q = 1; q < DS; q++) {
            int t = stk[q];
            asm volatile("" : "+v"(t));
            sel = (d >> 4) == q ? t : sel;
        }
        return rw_pick(sel, d & 15, rowbase);
    };
    auto stkf_get = [&](int d) -> R {
        R sel = stkf[0];
#pragma unroll
        for (int q = 1; q < DS; q++) {
            R t = stkf[q];
            asm volatile("" : "+v"(t));
            sel = (d >> 4) == q ? t : sel;
        }
        return rw_pick(sel, d & 15, rowbase);
    };
    auto stk_set = [&](int d, int e, R f, bool pred) {
#pragma unroll
        for (int q = 0; q < DS; q++) {
            const bool h = pred && (d >> 4) == q && li == (d & 15);
            stk[q] = h ? e : stk[q]; stkf[q] = h ? f : stkf[q];
        }
    };
    auto abyte_get = [&](int d) -> unsigned {
        unsigned sel = abyte[0];
#pragma unroll
        for (int q = 1; q < AW; q++) {
            unsigned t = abyte[q];
            asm volatile("" : "+v"(t));
            sel = (d >> 2) == q ? t : sel;
        }
        return (sel >> (8 * (d & 3))) & 0xffu;
    };
    auto abyte_set = [&](int d, unsigned v, bool pred) {
        const int sh = 8 * (d & 3);
#pragma unroll
        for (int q = 0; q < AW; q++) abyte[q] = (pred && (d >> 2) == q) ? ((abyte[q] & ~(0xffu << sh)) | (v << sh)) : abyte[q];
    };
    auto snap_r = [&](int d) -> R * {
        const RowParams<R> *a = RW_ARGS();
        return a->bnb_r + ((long long)myrow * a->bnb_depth + d) * (long long)kSnapR;
    };
    auto snap_i = [&](int d) -> int32_t * {
        const RowParams<R> *a = RW_ARGS();
        return a->bnb_i + ((long long)myrow * a->bnb_depth + d) * (long long)kSnapI;
    };
    // the factor of a problem as 16-byte vectors, the whole padded triangle whatever the node's size, moved by ALL 64
    // lanes for one row of the wavefront after the other (seldom more than one row of a wavefront needs it at once): six
    // instructions a copy where a row-by-row one by the row's own 16 lanes takes 120.  Memory -> LDS without registers
    // (global_load_lds: wave-uniform LDS base + 16 bytes per lane), so that a restore waits for memory once.
    // (the binary32 44-row shape copies word by word: eight wavefronts per workgroup leave no spacing of the factors that is
    // both a multiple of four and 16 banks apart, row_copy16)
    constexpr int VB = row_copy16(CAPP, (int)sizeof(R)) ? 16 : 4;
    constexpr int VW = VB / (int)sizeof(R), FLV = (rowp_size(CAPP) + VW - 1) / VW, FLK = (FLV + 63) / 64;
    typedef R rw_vec __attribute__((ext_vector_type(VW)));
    auto uni64 = [&](const void *ptr, int gg) -> unsigned long long {
        const unsigned long long v = (unsigned long long)ptr;
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 16 * gg),
                       hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 16 * gg);
        return (unsigned long long)lo | ((unsigned long long)hi << 32);
    };
    // the registers of the positions below nd and the factor as it stands -> slot d, for the rows with `sv`
    auto save_tri = [&](bool sv, int d, int nd) {
        R *sr = snap_r(d);
        int32_t *si = snap_i(d);
        if (sv) {
#pragma unroll
            for (int s = 0; s < S; s++) {
                const bool in = pos[s] < nd;
                sr[16 * s + li] = in ? rhs[s] : (R)0; sr[16 * S + 16 * s + li] = in ? D[s] : (R)0;
                sr[32 * S + 16 * s + li] = in ? Dinv[s] : (R)0; sr[48 * S + 16 * s + li] = in ? y[s] : (R)0;
                si[16 * s + li] = in ? ws[s] : 0;
            }
            if (li == 0) si[16 * S] = na;                        // (rows nd .. na-1 of the copy belong to deeper nodes)
        }
#pragma unroll
        for (int gg = 0; gg < 4; gg++) {
            if (__builtin_amdgcn_readlane(sv ? 1 : 0, 16 * gg)) {
                const rw_vec *lsrc = reinterpret_cast<const rw_vec *>(lds + __builtin_amdgcn_readlane(Lg, 16 * gg));
                rw_vec *gdst = reinterpret_cast<rw_vec *>(uni64(sr + 64 * S, gg));
                rw_vec tv[FLK];
#pragma unroll
                for (int q = 0; q < FLK; q++) {
                    const int v = lane + 64 * q;
                    tv[q] = lsrc[v < FLV ? v : FLV - 1];
                }
#pragma unroll
                for (int q = 0; q < FLK; q++) {
                    const int v = lane + 64 * q;
                    if (v < FLV) gdst[v] = tv[q];
                }
            }
        }
    };
    // back to the optimal state of the node at depth d, entry e (rows with `rs`): its second child continues from there
    auto restore = [&](bool rs, int d, int e) {
        const R *sr = snap_r(d);
        const int32_t *si = snap_i(d);
        const int naold = na;
        const bool clean = rs && ((snap_clean >> d) & 1ull) != 0ull;
        const bool dirty = rs && !clean;
        RWT_COUNT(17, rw_any(clean) ? 1 : 0);
        RWT_COUNT(18, rw_any(dirty) ? 1 : 0);
        RWT_COUNT(26, rw_any(dirty) ? 0 : 1);
        RWS_BEGIN;
        {
            const unsigned ab = abyte_get(d);
            const R fv = stkf_get(d);
            actb = rs ? (ab & 15u) : actb; lowb = rs ? (ab >> 4) : lowb;
            fval = rs ? fv : fval; na = rs ? ((e >> 13) & 127) : na; nsoft = rs ? ((e >> 20) & 127) : nsoft;
            rls = rs ? 1 : rls;                                  // (the node's multipliers: the next trip's backward sweep)
        }
        int nahi = naold;                                        // rows of the factor in LDS that lie behind the node's
        if (rw_any(dirty)) {
            int nsv = 0;
            if (dirty) {
#pragma unroll
                for (int s = 0; s < S; s++) {
                    rhs[s] = sr[16 * s + li]; D[s] = sr[16 * S + 16 * s + li]; Dinv[s] = sr[32 * S + 16 * s + li];
                    y[s] = sr[48 * S + 16 * s + li]; ws[s] = si[16 * s + li];
                }
                nsv = si[16 * S];
            }
#pragma unroll
            for (int gg = 0; gg < 4; gg++) {
                if (__builtin_amdgcn_readlane(dirty ? 1 : 0, 16 * gg)) {
                    const int lgg = __builtin_amdgcn_readlane(Lg, 16 * gg);
                    const rw_vec *gsrc = reinterpret_cast<const rw_vec *>(uni64(sr + 64 * S, gg));
#pragma unroll
                    for (int q = 0; q < FLK; q++) {
                        const int v = lane + 64 * q;
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass knows neither the instruction nor the address space)
                        if (v < FLV)
                            __builtin_amdgcn_global_load_lds(gsrc + v, (__attribute__((address_space(3))) void *)(lds + lgg + 64 * q * VW), VB, 0, 0);
#else
                        (void)gsrc; (void)lgg; (void)v;
#endif
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the copies have landed; the registers' loads with them)
#pragma unroll
            for (int s = 0; s < S; s++) {
                const int j = ws[s] & 0xffff;
                wof[s] = dirty ? (j & 15) * MS + (j >> 4) : wof[s];
            }
            nahi = dirty ? nsv : nahi;
        }
        {
            // the node's factor and positions are the leading part of what is here: cut what lies behind them
            const int a0 = na > 1 ? na : 1;
            const int lo = rw_min4(rs ? a0 : kRowBig), hi = rw_max4(rs ? nahi : 0);
            for (int i = lo; i < hi; i++) {
#pragma unroll
                for (int s = 0; s < S; s++)
                    if (rs && i >= a0 && i < nahi && pos[s] < i) lds[bo[s] + i] = (R)0;
            }
        }
#pragma unroll
        for (int s = 0; s < S; s++) {
            const bool cut = clean && pos[s] >= na;
            ws[s] = cut ? 0 : ws[s]; wof[s] = cut ? 0 : wof[s]; rhs[s] = cut ? (R)0 : rhs[s]; D[s] = cut ? (R)0 : D[s]; Dinv[s] = cut ? (R)0 : Dinv[s];
            y[s] = cut ? (R)0 : y[s];
        }
        // (a node is restored once, for its second child: from here on nobody needs slot d's factor)
        snap_clean = rs ? (snap_clean & ~(1ull << d)) : snap_clean;
#pragma unroll
        for (int s = 0; s < S; s++) { lam[s] = rs ? (R)0 : lam[s]; ls[s] = rs ? (R)0 : ls[s]; }
        sing = rs ? -1 : sing; ydirty = rs ? 0 : ydirty;
#ifdef LMPC_ROW_TRACE
        if (rw_any(dirty)) RWS_END(24); else RWS_END(25);
#endif
    };

    for (;;) {
        RWT(15);
        // (what the loop does not change, made opaque once per trip: otherwise every comparison against it is formed in
        // front of the loop and kept -- as lane masks in scalar registers -- across all of it)
        asm volatile("" : "+s"(n), "+s"(cap), "+s"(iter_limit));
        // (likewise this lane's index: every lane-constant comparison -- "is this the lane of position t" for 31 values
        // of t -- would otherwise be formed once, kept as a 64-bit lane mask and, there being 100 scalar registers,
        // spilled and reloaded by two v_readlane where one v_cmp forms it again)
        asm volatile("" : "+v"(li));
#pragma unroll
        for (int s = 0; s < S; s++) pos[s] = li + 16 * s;
        // =============================================================== a row without a problem takes the next one
        {
            const bool want = !live && !dead;
            if (rw_any(want)) {
                const RowParams<R> *a = RW_ARGS();
                const bool newchunk = want && cur >= endc;
                if (rw_any(newchunk)) {
                    int ch;
                    if (a->queue != nullptr) {
                        ch = nrows + rw_bc<0>(ticket);
                        if (newchunk && li == 0) ticket = atomicAdd(a->queue, 1);
                    } else {
                        ch = chunk_static + nrows;
                    }
                    const long long c64 = (long long)ch * qchunk;
                    const int cn = c64 < (long long)ntotal ? (int)c64 : ntotal;
                    chunk_static = newchunk ? ch : chunk_static;
                    cur = newchunk ? cn : cur;
                    endc = newchunk ? (cn + qchunk < ntotal ? cn + qchunk : ntotal) : endc;
                }
                const bool got = want && cur < endc;
                dead = (want && !got) ? 1 : dead;
                if (rw_any(got)) {
                    int npid = got ? cur : 0;
                    if (a->list != nullptr) {
                        // position in the concatenated list -> (segment, offset) -> problem, row by row (scalar)
#pragma unroll
                        for (int gg = 0; gg < 4; gg++) {
                            if (__builtin_amdgcn_readlane(got ? 1 : 0, 16 * gg)) {
                                const int ix = __builtin_amdgcn_readlane(cur, 16 * gg);
                                int sgm = (int)__popcll(__ballot(seg_end <= ix));
                                sgm = sgm < 63 ? sgm : 63;
                                const int off = ix - __builtin_amdgcn_readlane(seg_beg, sgm);
                                const int p = a->list[(long long)sgm * a->seg_cap + off];
                                npid = g == gg ? p : npid;
                            }
                        }
                    }
                    cur = got ? cur + 1 : cur;
                    // b_j = Dth_j . theta (mpc_update_qp.c:5-6): the record first, in one batch (a miss all the way to HBM), then
                    // four columns of Dth per round trip (L2)
                    const R *th = a->theta + (long long)npid * nth;
                    R b[MS];
#pragma unroll
                    for (int r = 0; r < MS; r++) b[r] = (R)0;
                    constexpr int NTHB = (MS >= 10) ? 8 : 16;        // (ten-slot shapes: registers are short; longer records take the tail loop)
                    R tb[NTHB];
#pragma unroll
                    for (int t = 0; t < NTHB; t++) tb[t] = th[t < nth ? t : nth - 1];
                    constexpr int DTB = (MS >= 10) ? 1 : 4;               // (columns per batch: the ten-slot shapes read Dth from LDS, one at a time)
                    rw_static_for<0, NTHB / DTB>([&](auto B) {
                        constexpr int t0 = decltype(B)::value * DTB;
                        if (t0 < nth) {
                            RW_BLOCK();
                            R dv[DTB][MS];
#pragma unroll
                            for (int q = 0; q < DTB; q++)
#pragma unroll
                                for (int r = 0; r < MS; r++) dv[q][r] = ldk(prm.P.oDth + (t0 + q < nth ? t0 + q : nth - 1), jc[r] * nth);
#pragma unroll
                            for (int q = 0; q < DTB; q++)
                                if (t0 + q < nth) {
#pragma unroll
                                    for (int r = 0; r < MS; r++) b[r] = wv_fma(dv[q][r], tb[t0 + q], b[r]);
                                }
                        }
                    });
                    for (int t = NTHB; t < nth; t++) {              // (records longer than the batch: one column per round trip)
                        const R tv = th[t];
#pragma unroll
                        for (int r = 0; r < MS; r++) b[r] = wv_fma(ldk(prm.P.oDth + t, jc[r] * nth), tv, b[r]);
                    }
#pragma unroll
                    for (int r = 0; r < MS; r++) {
                        const R du = ldk(prm.P.odu, jc[r]) + b[r], dl = ldk(prm.P.odl, jc[r]) + b[r];
                        dub[r] = got ? du : dub[r];
                        dlb[r] = got ? dl : dlb[r];
                    }
                    pid = got ? npid : pid; live = got ? 1 : live; na = got ? 0 : na; sing = got ? -1 : sing;
                    iter = got ? 1 : iter; cyc = got ? 0 : cyc; nsoft = got ? 0 : nsoft; napk = got ? 0 : napk;
                    ydirty = got ? 0 : ydirty; best = got ? (R)-1 : best; fval = got ? (R)0 : fval;
                    soft_slack = got ? (R)0 : soft_slack; actb = got ? 0u : actb; lowb = got ? 0u : lowb;
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        ws[s] = got ? 0 : ws[s]; wof[s] = got ? 0 : wof[s]; D[s] = got ? (R)0 : D[s]; Dinv[s] = got ? (R)0 : Dinv[s];
                        lam[s] = got ? (R)0 : lam[s]; ls[s] = got ? (R)0 : ls[s]; rhs[s] = got ? (R)0 : rhs[s];
                        y[s] = got ? (R)0 : y[s];
                    }
#pragma unroll
                    for (int s = 0; s < NS; s++) u[s] = got ? (R)0 : u[s];
                    if constexpr (BNB) {                         // the search starts at the root: nothing fixed, nothing found
                        forced = got ? -1 : forced; depth = got ? 0 : depth; nodes = got ? 0 : nodes;
                        total_it = got ? 0 : total_it; have = got ? 0 : have; bflag = got ? EXIT_INFEASIBLE : bflag;
                        bestval = got ? fbound : bestval; bestact = got ? 0u : bestact; bestlow = got ? 0u : bestlow;
                        snap_clean = got ? 0ull : snap_clean; snap_saved = got ? 0ull : snap_saved; tried2 = got ? 0ull : tried2;
                        rls = got ? 0 : rls;
#pragma unroll
                        for (int s = 0; s < NS; s++) ubest[s] = got ? (R)0 : ubest[s];
#pragma unroll
                        for (int r = 0; r < MS; r++) fixd[r] = got ? 255 : fixd[r];
                    }
                }
            }
            if (!rw_any(live != 0)) break;                       // every row of the wavefront has run out of problems
        }

        RWT(0);
        RWT_COUNT(10, 1);
        int flag = 0, fin = 0;                                   // fin: this row's problem ends in this trip, with `flag`
        {
            const bool lim = live != 0 && iter >= iter_limit;
            flag = lim ? EXIT_ITERLIMIT : 0;
            fin = lim ? 1 : 0;
        }
        // (branch and bound) a node that continues in place starts by taking the row just fixed into the working set: no
        // stationary point, no scan in that trip
        int addF = 0, immX = 0;                                  // immX: the row appended in this trip enters as a fixed (immutable) row
        unsigned fixb = 0u;                                      // bit r: this lane's row of slot r is fixed on the path to this node
        if constexpr (BNB) {
            const bool frc = live != 0 && !fin && forced >= 0;
            const bool fullF = frc && na >= cap;
            flag = fullF ? EXIT_WSCAP : flag; fin = fullF ? 1 : fin;
            addF = (frc && !fullF) ? 1 : 0;
            immX = addF;
#pragma unroll
            for (int r = 0; r < MS; r++) fixb |= fixd[r] < depth ? (1u << r) : 0u;
        }
        const int run = (live != 0 && !fin && !addF) ? 1 : 0;
        const int sgl = sing >= 0 ? 1 : 0;
        const int stat = (run || (BNB && rls)) ? 1 : 0;          // rows that take part in the backward sweep
        const int namax = rw_max4(stat ? na : 0);

        // =============================================================== stationary point / singular direction
        {
            // (y = L^-1 rhs is up to date here: an append extends it, a removal re-runs the forward sweep in its own trip --
            // see the append's second half)
            R v[S];
#pragma unroll
            for (int s = 0; s < S; s++) v[s] = stat ? y[s] * Dinv[s] : (R)0;
            int lowsg = 0;
            if (rw_any(run && sgl)) {
                const int sgc = sgl ? sing : 1;
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const R lv = lds[bo[s] + (sgc > 1 ? sgc : 1)];       // L(sing, pos)
                    v[s] = sgl ? ((run && pos[s] < sing) ? -lv : (R)0) : v[s];
                }
                int wsel = ws[0];
#pragma unroll
                for (int s = 1; s < S; s++) wsel = (sgc >> 4) == s ? ws[s] : wsel;
                lowsg = (rw_pick(wsel, sgc, rowbase) & kRowPosFlagLow) ? 1 : 0;
            }
            sweep_bwd(v, namax - 1);
#pragma unroll
            for (int s = 0; s < S; s++) {
                const R a1 = pos[s] == sing ? (R)1 : (pos[s] > sing ? (R)0 : v[s]);
                const R asg = lowsg ? -a1 : a1;
                const R ans = pos[s] < na ? v[s] : (R)0;
                const R acc = sgl ? asg : ans;
                ls[s] = stat ? acc : ls[s];
            }
            rls = 0;
        }

        RWT(1);
        // =============================================================== blocking multipliers: (alpha, rm) = first minimum
        int rm = -1;
        R alpha = (R)0;
        {
            bool blk[S];
            bool anyb = false;
#pragma unroll
            for (int s = 0; s < S; s++) {
                const bool okd = (ws[s] & kRowPosFlagLow) ? (ls[s] < dual_tol) : (ls[s] > -dual_tol);
                blk[s] = run && pos[s] < na && !(ws[s] & kRowPosFlagImm) && !okd;
                anyb = anyb || blk[s];
            }
            if (rw_any(anyb)) {
                R cand[S], cm = kInf;
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const R den = sgl ? ls[s] : (ls[s] - lam[s]);
                    cand[s] = -lam[s] / den;
                    cm = wv_min2(cm, blk[s] ? cand[s] : kInf);
                }
                const R gmin = rw_min(cm);
                int tp = kRowBig, bp = kRowBig;
#pragma unroll
                for (int s = S - 1; s >= 0; s--) {
                    tp = (blk[s] && cand[s] == gmin) ? pos[s] : tp;
                    bp = blk[s] ? pos[s] : bp;
                }
                tp = rw_min(tp);
                bp = rw_min(bp);
                const int r0 = tp != kRowBig ? tp : bp;              // (no position equals the minimum: NaNs -- the first blocked one)
                rm = r0 != kRowBig ? r0 : -1;
                R csel = cand[0];
#pragma unroll
                for (int s = 1; s < S; s++) csel = ((r0 & 0xffff) >> 4) == s ? cand[s] : csel;
                const R a0 = rw_pick(csel, r0 & 15, rowbase);
                alpha = rm >= 0 ? a0 : (R)0;
            }
        }
        RWT(2);
        {
            const bool infx = run && sgl && rm < 0;
            flag = infx ? EXIT_INFEASIBLE : flag;
            fin = infx ? 1 : fin;
        }
        const int doRem = (run && rm >= 0) ? 1 : 0;
        const int doAdd = (run && !sgl && rm < 0) ? 1 : 0;

        // =============================================================== no blocking multiplier: primal iterate, scan, append
        int addpX = 0, jaddX = 0, lowerX = 0, mtX = 0;           // (the append's two halves: see below)
        R qX[S], gjjX = (R)0, rjX = (R)0, fvalX = (R)0;
#pragma unroll
        for (int s = 0; s < S; s++) qX[s] = (R)0;
        auto gather = [&]() {
            if (rw_any(addpX != 0)) {
                const bool ap = addpX != 0;
                jaddX = ap ? (mtX >> 1) : 0;
                lowerX = (ap && (mtX & 1)) ? 1 : 0;
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const bool in = ap && pos[s] < na;
                    const int a = in ? (ws[s] & 0xffff) : jaddX;
                    const int hi = a >= jaddX ? a : jaddX, lo = a >= jaddX ? jaddX : a;
                    qX[s] = ldc(oG, hi * (hi + 1) / 2 + lo);
                }
                gjjX = ldc(oG, jaddX * (jaddX + 1) / 2 + jaddX);
                // the bound of row jadd that enters: from the lane and slot that own the row
                R bsel = lowerX ? dlb[0] : dub[0];
#pragma unroll
                for (int r = 1; r < MS; r++) bsel = (jaddX >> 4) == r ? (lowerX ? dlb[r] : dub[r]) : bsel;
                rjX = -rw_pick(bsel, jaddX & 15, rowbase);
            }
        };
        if (rw_any(doAdd != 0)) {
            const int namaxA = rw_max4(doAdd ? na : 0);
            R un[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) un[s] = (R)0;
            // u = -M_W' lam*  (rows beyond a working set: lam* = 0 there, row 0 of M)
#ifndef LMPC_ROW_CHP
#define LMPC_ROW_CHP 4
#endif
            constexpr int CHP = LMPC_ROW_CHP, NBP = (CAPP + CHP - 1) / CHP;
            {
                R mn[CHP][NS], ln[CHP];
                // binary32: a position's offset and its multiplier travel as ONE 64-bit broadcast (the multiplier is used a
                // block later, with the rows of M' the offset fetched)
                typedef int rw_i2 __attribute__((ext_vector_type(2)));
                double wl[S];
                if constexpr (sizeof(R) == 4) {
#pragma unroll
                    for (int s = 0; s < S; s++) { const rw_i2 t = {wof[s], __float_as_int((float)ls[s])}; wl[s] = __builtin_bit_cast(double, t); }
                }
                auto fetch = [&](auto B) {
                    constexpr int i0 = decltype(B)::value * CHP;
                    rw_static_for<0, CHP>([&](auto Q) {
                        constexpr int i = i0 + decltype(Q)::value < CAPP ? i0 + decltype(Q)::value : CAPP - 1;
                        int w;
                        if constexpr (sizeof(R) == 4) {
                            const rw_i2 t = __builtin_bit_cast(rw_i2, rw_bc<i>(wl[i >> 4]));
                            w = mcol0 + t.x;
                            ln[decltype(Q)::value] = (R)__int_as_float(t.y);
                        } else {
                            w = mcol0 + rw_bc<i>(wof[i >> 4]);
                        }
#pragma unroll
                        for (int s = 0; s < NS; s++) mn[decltype(Q)::value][s] = lds[w + s * 16 * MPAD];
                    });
                };
                fetch(std::integral_constant<int, 0>{});
                rw_static_for<0, NBP>([&](auto B) {
                    constexpr int b = decltype(B)::value, i0 = b * CHP;
                    if (i0 < namaxA) {
                        RW_BLOCK();
                        R mv[CHP][NS], lv[CHP];
#pragma unroll
                        for (int q = 0; q < CHP; q++) {
                            lv[q] = sizeof(R) == 4 ? ln[q] : (R)0;
#pragma unroll
                            for (int s = 0; s < NS; s++) mv[q][s] = mn[q][s];
                        }
                        if constexpr (b + 1 < NBP) fetch(std::integral_constant<int, b + 1>{});
                        rw_static_for<0, CHP>([&](auto Q) {
                            constexpr int i = i0 + decltype(Q)::value;
                            if constexpr (i < CAPP) {
                                R l;
                                if constexpr (sizeof(R) == 4) l = lv[decltype(Q)::value];
                                else l = rw_bc<i>(ls[i >> 4]);
#pragma unroll
                                for (int s = 0; s < NS; s++) un[s] = wv_fma(-mv[decltype(Q)::value][s], l, un[s]);
                            }
                        });
                        RW_BLOCK();
                    }
                });
            }
#pragma unroll
            for (int s = 0; s < NS; s++) un[s] = li + 16 * s < n ? un[s] : (R)0;
            RWT(3);
            RWT_COUNT(11, 1);
            R soft = (R)0;
            if (rw_any(doAdd && nsoft > 0)) {
                rw_static_for<0, (CAPP + 3) / 4>([&](auto B) {
                    constexpr int i0 = decltype(B)::value * 4;
                    if (i0 < namaxA) {
                        RW_BLOCK();
                        rw_static_for<0, 4>([&](auto Q) {
                            constexpr int i = i0 + decltype(Q)::value;
                            if constexpr (i < CAPP) {
                                const int w = rw_bc<i>(ws[i >> 4]);          // (beyond a working set: no flags)
                                const R l = rw_bc<i>(ls[i >> 4]);
                                const R sn = wv_fma(l * l, rho_soft, soft);
                                soft = (w & kRowPosFlagSoft) ? sn : soft;
                            }
                        });
                    }
                });
            }
            // objective u'u and the row values M u in one pass over the variables
            R fv = (R)0, Mu[MS];
#pragma unroll
            for (int r = 0; r < MS; r++) Mu[r] = (R)0;
#ifndef LMPC_ROW_CHK6
#define LMPC_ROW_CHK6 2
#endif
            // (binary32 with four constraint slots: rows of 16 bytes per lane, four of them in flight measured 3 % faster)
            // (the one-slot shape with ten constraint slots -- two wavefronts per SIMD on 256 registers -- one row ahead: two
            // spill into the loop)
            constexpr int CHK = (sizeof(R) == 4 && MS == 4) ? 4 : ((S == 1 && MS >= 10) ? 1 : LMPC_ROW_CHK6), NBK = 16 * NS / CHK;   // (CHK divides 4: the staged M' has ceil4(n) rows)
            {
                R mn[CHK][MS];
                constexpr int PW = (sizeof(R) == 4 && MS % 4 == 0) ? 4 : 2;      // entries per load (16 bytes where they line up)
                typedef R rw_pair __attribute__((ext_vector_type(PW)));
                auto fetch = [&](auto B) {
                    constexpr int k0 = decltype(B)::value * CHK;
#pragma unroll
                    for (int q = 0; q < CHK; q++)                              // (rows beyond n, columns beyond m: zeros)
#pragma unroll
                        for (int r = 0; r < MS; r += PW) {
                            const rw_pair pr = *reinterpret_cast<const rw_pair *>(&lds[mrow + (k0 + q) * MPAD + r]);
#pragma unroll
                            for (int e = 0; e < PW; e++) mn[q][r + e] = pr[e];
                        }
                };
                fetch(std::integral_constant<int, 0>{});
                rw_static_for<0, NBK>([&](auto B) {
                    constexpr int b = decltype(B)::value, k0 = b * CHK;
                    if (k0 < n) {
                        RW_BLOCK();
                        R mt[CHK][MS];
#pragma unroll
                        for (int q = 0; q < CHK; q++)
#pragma unroll
                            for (int r = 0; r < MS; r++) mt[q][r] = mn[q][r];
                        // (unconditionally: a test here makes the compiler wait for ALL loads in flight at the join, the ones just
                        // issued included; rows beyond M' are the factors, finite, and meet u_k = 0)
                        if constexpr (b + 1 < NBK) fetch(std::integral_constant<int, b + 1>{});
                        rw_static_for<0, CHK>([&](auto Q) {
                            constexpr int k = k0 + decltype(Q)::value;
                            const R v = rw_bc<k>(un[k >> 4]);
                            fv = wv_fma(v, v, fv);
#pragma unroll
                            for (int r = 0; r < MS; r++) Mu[r] = wv_fma(mt[decltype(Q)::value][r], v, Mu[r]);
                        });
                        RW_BLOCK();                                  // (keeps the next block's loads HERE: the compiler sinks them into that block otherwise)
                    }
                });
            }
            RWT(4);
            const R fvalN = fv + soft;
            {                                                    // (this trip's iterate is the row's iterate from here on)
                const bool c = doAdd != 0;
#pragma unroll
                for (int s = 0; s < NS; s++) u[s] = c ? un[s] : u[s];
                fval = c ? fvalN : fval;
                soft_slack = c ? soft : soft_slack;
            }
            int addp = doAdd;
            {
                const bool dom = addp && fvalN > (BNB ? bestval : fbound);     // (the search prunes against its incumbent)
                flag = dom ? EXIT_INFEASIBLE : flag; fin = dom ? 1 : fin; addp = dom ? 0 : addp;
            }
            // most violated row: smallest value below -primal_tol, ties to the lowest (row, side) index.  Per row the side is
            // the upper one if that is violated, else the lower one (both at once would need dupper < dlower).
            R cval[MS], cm = kInf;
            const R ntol = -primal_tol;
#pragma unroll
            for (int r = 0; r < MS; r++) {
                const R vu = dub[r] - Mu[r];
                const R vl = Mu[r] - dlb[r];                     // = -(dlower_j - M_j u), exactly
                const bool fre = ((okb & ~actb & ~fixb) >> r) & 1u;
                const R c = vu < ntol ? vu : vl;
                cval[r] = fre ? c : kInf;
                cm = wv_min2(cm, cval[r]);
            }
            const R gsel = rw_min(cm);
            const bool viol = gsel < ntol;
            int mt = kRowBig;
#pragma unroll
            for (int r = MS - 1; r >= 0; r--) {
                const int side = (dub[r] - Mu[r]) < ntol ? 0 : 1;
                mt = cval[r] == gsel ? 2 * (li + 16 * r) + side : mt;
            }
            mt = rw_min(mt);
            {
                const bool opt = addp && !viol;
                int anybr = 0;
                if (rw_any(opt)) {                               // does the iterate violate a hard row of its own working set?
                    int broken = 0;
#pragma unroll
                    for (int r = 0; r < MS; r++) {
                        const bool hw = ((hardb & actb & ~fixb) >> r) & 1u;
                        const bool br = hw && ((dub[r] - Mu[r]) < ntol || (Mu[r] - dlb[r]) < ntol);
                        broken = br ? 1 : broken;
                    }
                    anybr = rw_or(broken);
                }
                const int fl = anybr ? EXIT_CYCLE : (soft > primal_tol ? EXIT_SOFT_OPTIMAL : EXIT_OPTIMAL);
                flag = opt ? fl : flag; fin = opt ? 1 : fin; addp = opt ? 0 : addp;
                if constexpr (BNB) {
                    if (rw_any(opt)) {
                        // the node's relaxation is solved: the lowest binary row outside its working set is branched on, the
                        // side first whose bound the row's value is closer to
                        int cand = kRowBig;
#pragma unroll
                        for (int r = MS - 1; r >= 0; r--) cand = (((binb & ~actb) >> r) & 1u) ? li + 16 * r : cand;
                        const int jb = rw_min(cand);
                        const int jq = jb != kRowBig ? jb : 0;
                        R msel = Mu[0], lsel = dlb[0], usel = dub[0];
#pragma unroll
                        for (int r = 1; r < MS; r++) {
                            const bool h = (jq >> 4) == r;
                            msel = h ? Mu[r] : msel; lsel = h ? dlb[r] : lsel; usel = h ? dub[r] : usel;
                        }
                        const R mj = rw_pick(msel, jq & 15, rowbase), dlo = rw_pick(lsel, jq & 15, rowbase),
                                dup = rw_pick(usel, jq & 15, rowbase);
                        jbX = opt ? jb : jbX;
                        sideX = opt ? (((mj - dlo) < (dup - mj)) ? 1 : 0) : sideX;
                        // A node that branches does so HERE: its first child continues in place, so the row just fixed joins this
                        // very trip's append phase (decided behind the trip, as the other moves of the search are, the fixed row's
                        // append would cost the row a trip of its own: one in ten).  Not at the node limit and not when
                        // the working set is full: those cases go the long way, with the wavefront kernel's accounting.
                        const bool early = opt && fl >= 1 && jb != kRowBig && na < cap && nodes + 1 < 100000;
                        if (rw_any(early)) {
                            nodes = early ? nodes + 1 : nodes;
                            total_it = early ? total_it + iter : total_it;
                            stk_set(depth, jb | (sideX << 10) | (na << 13) | (nsoft << 20), fvalN, early);
                            abyte_set(depth, actb | (lowb << 4), early);
                            const unsigned long long bit = 1ull << depth;
                            snap_clean = early ? (snap_clean | bit) : snap_clean;
                            snap_saved = early ? (snap_saved & ~bit) : snap_saved;
                            tried2 = early ? (tried2 & ~bit) : tried2;
#pragma unroll
                            for (int r = 0; r < MS; r++) {
                                const bool mine = early && li + 16 * r == jb;
                                fixd[r] = mine ? depth : ((early && fixd[r] == depth) ? 255 : fixd[r]);
                            }
                            depth = early ? depth + 1 : depth;
                            // the child's node starts: the fixed row is this trip's append
                            fin = early ? 0 : fin; addp = early ? 1 : addp; mt = early ? 2 * jb + sideX : mt;
                            immX = early ? 1 : immX;
                            iter = early ? 1 : iter; cyc = early ? 0 : cyc; best = early ? (R)-1 : best;
                        }
                    }
                }
                const bool full = addp && na >= cap;
                flag = full ? EXIT_WSCAP : flag; fin = full ? 1 : fin; addp = full ? 0 : addp;
            }
            // (branch and bound) ... and a node that ends here WITHOUT branching -- a leaf, a pruned node, a relaxation that
            // failed -- backtracks here too: back to the deepest node with an untried side (restore), that node's
            // multipliers by one more backward sweep, and its second child's fixed row joins this trip's append.  (Behind
            // the trip, the search's other moves' place, the fixed row's append costs the row a trip of its own: one in
            // seven.)  The search's last node, the node limit and a descent that the capacity blocks go the long way.
            bool flipE = false;
            if constexpr (BNB) {
                const bool endE = doAdd != 0 && fin != 0 && flag != EXIT_WSCAP;
                const bool leafE = endE && flag >= 1 && jbX == kRowBig;
                const bool blockedE = endE && flag >= 1 && jbX != kRowBig;
                const unsigned long long open = ~tried2 & ((1ull << depth) - 1ull);
                flipE = endE && !blockedE && open != 0ull && nodes + 1 < 100000;
                if (rw_any(flipE)) {
                    RWS_BEGIN;
                    nodes = flipE ? nodes + 1 : nodes;
                    total_it = flipE ? total_it + iter : total_it;
                    const bool better = flipE && leafE && (!have || fvalN < bestval);
                    have = better ? 1 : have; bestval = better ? fvalN : bestval;
                    bestact = better ? actb : bestact; bestlow = better ? lowb : bestlow;
#pragma unroll
                    for (int s = 0; s < NS; s++) ubest[s] = better ? un[s] : ubest[s];
                    depth = flipE ? 64 - (int)__builtin_clzll(open | 1ull) : depth;     // (open != 0 for these rows)
                    const int d = flipE ? depth - 1 : 0;
                    const int e = stk_get(d);
                    const int ne = e ^ (1 << 10);
#pragma unroll
                    for (int q = 0; q < DS; q++) stk[q] = (flipE && (d >> 4) == q && li == (d & 15)) ? ne : stk[q];
                    tried2 = flipE ? (tried2 | (1ull << d)) : tried2;
                    restore(flipE, d, e);
                    {
                        R v[S];
#pragma unroll
                        for (int s = 0; s < S; s++) v[s] = flipE ? y[s] * Dinv[s] : (R)0;
                        sweep_bwd(v, rw_max4(flipE ? na : 0) - 1);
#pragma unroll
                        for (int s = 0; s < S; s++) ls[s] = flipE ? (pos[s] < na ? v[s] : (R)0) : ls[s];
                        rls = flipE ? 0 : rls;
                    }
                    fin = flipE ? 0 : fin; addp = flipE ? 1 : addp; mt = flipE ? 2 * (ne & 1023) + ((ne >> 10) & 1) : mt;
                    immX = flipE ? 1 : immX;
                    iter = flipE ? 1 : iter; cyc = flipE ? 0 : cyc; best = flipE ? (R)-1 : best;
                    RWS_END(20);
                }
            }
            RWT(5);
            // ---- append row jadd to the working sets of the rows with addp, first half: the Gram entries G(W_i, jadd) are
            // requested here (L2: ~a microsecond), the factor is extended behind the removal phase of the other rows
            addpX = addp; fvalX = flipE ? fval : fvalN; mtX = mt;
            if constexpr (!BNB) gather();
        }
        if constexpr (BNB) {
            addpX = addF ? 1 : addpX; fvalX = addF ? fval : fvalX; mtX = addF ? forced : mtX;
            gather();
        }
        RWT(6);
        // =============================================================== a blocking multiplier: step, drop its row
        if (rw_any(doRem != 0)) {
            RWT_COUNT(12, 1);
            const bool dr = doRem != 0;
#pragma unroll
            for (int s = 0; s < S; s++) {
                const R ln = wv_fma(alpha, sgl ? ls[s] : (ls[s] - lam[s]), lam[s]);
                lam[s] = dr ? ln : lam[s];
            }
            const int r = dr ? rm : 0;
            const int nao = na;
            if constexpr (BNB) {
                // nodes on the path whose factor is still the leading block of this one and reaches beyond row r: their
                // positions and factors go to their slots before the row leaves
                unsigned long long hit = 0ull;
#pragma unroll
                for (int q = 0; q < DS; q++) {
                    const unsigned long long bal = __ballot(dr && li + 16 * q < depth && ((stk[q] >> 13) & 127) > r);
                    hit |= ((bal >> (16 * g)) & 0xffffull) << (16 * q);
                }
                hit &= snap_clean;
                unsigned long long todo = hit & ~snap_saved;
                while (rw_any(todo != 0ull)) {
                    RWS_BEGIN;
                    RWT_COUNT(19, 1);
                    const bool sv = todo != 0ull;
                    const int d = sv ? (int)__builtin_ctzll(todo) : 0;
                    todo = sv ? (todo & (todo - 1ull)) : todo;
                    save_tri(sv, d, (stk_get(d) >> 13) & 127);
                    RWS_END(21);
                }
                snap_saved |= hit;
                snap_clean &= ~hit;
            }
            R w[S];
#pragma unroll
            for (int s = 0; s < S; s++) {
                const R lv = lds[fo[s] + cbm(r < CAPP - 1 ? r : 0)];                // L(pos, r), old row index = pos
                w[s] = (dr && pos[s] > r && pos[s] < nao) ? lv : (R)0;
            }
            R dsel = D[0];
            int wsel = ws[0];
#pragma unroll
            for (int s = 1; s < S; s++) { dsel = (r >> 4) == s ? D[s] : dsel; wsel = (r >> 4) == s ? ws[s] : wsel; }
            R al = rw_pick(dsel, r & 15, rowbase);
            const int wsr = rw_pick(wsel, r & 15, rowbase);
            const int jrem = wsr & 0xffff, softrem = (wsr & kRowPosFlagSoft) ? 1 : 0;
            // new row i = old row i + 1 without column r (i >= r): lane c moves entry (i + 1, c') to (i, c)
            const int rlo = rw_min4(dr ? r : kRowBig), nhi = rw_max4(dr ? nao : 0);
            {
                int so[S];
#pragma unroll
                for (int s = 0; s < S; s++) {
                    int cp = pos[s] + (pos[s] >= r ? 1 : 0);
                    cp = cp < CAPP - 1 ? cp : 0;
                    so[s] = Lg + cbm(cp);
                }
                int clo[S];                                      // rows this lane's column takes part in: [clo, chi)
                const int chi = dr ? nao - 1 : 0;
#pragma unroll
                for (int s = 0; s < S; s++) clo[s] = pos[s] + 1 > r ? pos[s] + 1 : r;
                // (four rows read before they are written: the reads of a row do not wait for the row above it)
                for (int i0 = rlo; i0 < nhi - 1; i0 += 4) {
                    R tv[4][S];
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int s = 0; s < S; s++) tv[q][s] = lds[so[s] + (i0 + q + 1 < CAPP ? i0 + q + 1 : CAPP - 1)];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int i = i0 + q;
#pragma unroll
                        for (int s = 0; s < S; s++)
                            if (i >= clo[s] && i < chi) lds[bo[s] + i] = tv[q][s];
                    }
                }
#pragma unroll
                for (int s = 0; s < S; s++)
                    if (dr && pos[s] < nao - 1) lds[bo[s] + nao - 1] = (R)0;          // the row that left: back to zeros
            }
            // the per-position registers move down by one from position r on
            {
                // (positions beyond a working set hold zeros in every one of these arrays, so the position that falls free
                // takes its zero from its neighbour like any other: one select per value)
                auto shift = [&](auto *a) {
                    using V = std::remove_pointer_t<decltype(a)>;
                    V nx[S];
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        V nb = (V)0;
                        if (s + 1 < S) nb = rw_bc<0>(a[s + 1 < S ? s + 1 : s]);
                        nx[s] = rw_shl1(a[s], nb);
                    }
#pragma unroll
                    for (int s = 0; s < S; s++) a[s] = (dr && pos[s] >= r) ? nx[s] : a[s];
                };
                shift(ws); shift(wof); shift(lam); shift(rhs); shift(D); shift(Dinv); shift(w);
            }
            na = dr ? nao - 1 : na; sing = dr ? -1 : sing; ydirty = dr ? 1 : ydirty;
            RWT(7);
            // rank-one update of the trailing block, column by column
            {
                RWS_BEGIN;
                RWT_COUNT(28, nhi - rlo);
                int stop = 0;
                const int nhi2 = rw_max4(dr ? na : 0);
                // (the columns of a block of four steps are read at its head: a step changes its own column only)
                constexpr int CHR = 4;
                rw_static_for<0, (CAPP - 1 + CHR - 1) / CHR>([&](auto B) {
                    constexpr int t0 = decltype(B)::value * CHR;
                    if (t0 + CHR > rlo && t0 < nhi2) {
                        RW_BLOCK();
                        R lqb[CHR][S];
#pragma unroll
                        for (int q = 0; q < CHR; q++)
#pragma unroll
                            for (int s = 0; s < S; s++) lqb[q][s] = lds[fo[s] + rowp_cbm(CAPP, t0 + q < CAPP - 1 ? t0 + q : CAPP - 2)];
                        rw_static_for<0, CHR>([&](auto Q) {
                            constexpr int t = t0 + decltype(Q)::value;
                            if constexpr (t < CAPP - 1) {
                                const bool actv = dr && t >= r && t < na && !stop;
                                if (rw_any(actv)) {
                                    const R pt = rw_bc<t>(w[t >> 4]);
                                    const R dold = rw_bc<t>(D[t >> 4]);
                                    const R dbar = wv_fma(al * pt, pt, dold);
                                    const bool sng = actv && dbar < zero_tol;
                                    const bool upd = actv && !sng;
                                    const R rinv = (R)1 / dbar;
                                    const R beta = (pt * al) * rinv;
                                    const R aln = (dold * al) * rinv;
                                    {
                                        const bool hs = sng && pos[t >> 4] == t, hu = upd && pos[t >> 4] == t;
                                        D[t >> 4] = hs ? (R)0 : (hu ? dbar : D[t >> 4]);
                                        Dinv[t >> 4] = hs ? (R)0 : (hu ? rinv : Dinv[t >> 4]);
                                    }
                                    sing = sng ? t : sing;
                                    stop = sng ? 1 : stop;
                                    al = upd ? aln : al;
#pragma unroll
                                    for (int s = 0; s < S; s++) {
                                        if (16 * s + 15 > t) {
                                            const R lq = lqb[decltype(Q)::value][s];
                                            const bool m2 = upd && pos[s] > t && pos[s] < na;
                                            const R wn = wv_fma(-pt, lq, w[s]);
                                            w[s] = m2 ? wn : w[s];
                                            if (m2) lds[fo[s] + rowp_cbm(CAPP, t)] = wv_fma(beta, wn, lq);
                                        }
                                    }
                                }
                            }
                        });
                    }
                });
                RWS_END(27);
            }
            {
                const bool mine = dr && li == (jrem & 15);
                const unsigned bit = 1u << (jrem >> 4);
                actb = mine ? (actb & ~bit) : actb;
                lowb = mine ? (lowb & ~bit) : lowb;
            }
            nsoft = dr ? nsoft - softrem : nsoft;
        }
        // =============================================================== ... second half of the append: extend the factor
        // (the rows that dropped a row in this trip ride along in the forward sweep: their y = L^-1 rhs for the factor as the
        // removal left it -- what the next stationary point needs -- instead of a sweep of their own at the next trip's head)
        if (rw_any(addpX != 0 || doRem != 0)) {
                const bool ap = addpX != 0;
                const bool ry = doRem != 0;
                const int jadd = jaddX;
                const bool lower = lowerX != 0;
                const R fvalN = fvalX, gjj = gjjX, rj = rjX;
                const int sj = sens[jadd];
                const bool is_soft = (sj & SENSE_SOFT) != 0;
                R q[S];
#pragma unroll
                for (int s = 0; s < S; s++) q[s] = pos[s] < na ? (ap ? qX[s] : (ry ? rhs[s] : (R)0)) : (R)0;
#pragma unroll
                for (int s = 0; s < S; s++) lam[s] = ap ? ls[s] : lam[s];
                const int namaxQ = rw_max4((ap || ry) ? na : 0);
                sweep_fwd(q, namaxQ);
#pragma unroll
                for (int s = 0; s < S; s++) y[s] = ry ? (pos[s] < CAPP ? q[s] : (R)0) : y[s];
                ydirty = ry ? 0 : ydirty;
                R l[S];
#pragma unroll
                for (int s = 0; s < S; s++) l[s] = q[s] * Dinv[s];
                R dnew = is_soft ? gjj + rho_soft : gjj;
                R ynew = rj;
                if constexpr (sizeof(R) == 4) {
                    // binary32: (q_i, y_i) travel as ONE 64-bit broadcast and the two chains advance in one packed fma
                    typedef float rw_f2 __attribute__((ext_vector_type(2)));
                    double qy[S];
#pragma unroll
                    for (int s = 0; s < S; s++) { const rw_f2 t = {(float)q[s], (float)y[s]}; qy[s] = __builtin_bit_cast(double, t); }
                    rw_f2 acc = {(float)dnew, (float)ynew};
                    rw_static_for<0, (CAPP + 3) / 4>([&](auto B) {
                        constexpr int i0 = decltype(B)::value * 4;
                        if (i0 < namaxQ) {
                            RW_BLOCK();
                            rw_static_for<0, 4>([&](auto Q) {
                                constexpr int i = i0 + decltype(Q)::value;
                                if constexpr (i < CAPP) {                     // (beyond a working set: l_i = 0)
                                    const float lq = (float)rw_bc<i>(l[i >> 4]);
                                    const rw_f2 b = __builtin_bit_cast(rw_f2, rw_bc<i>(qy[i >> 4]));
                                    const rw_f2 nl = {-lq, -lq};
                                    acc = __builtin_elementwise_fma(nl, b, acc);
                                }
                            });
                        }
                    });
                    dnew = (R)acc.x; ynew = (R)acc.y;
                } else {
                rw_static_for<0, (CAPP + 3) / 4>([&](auto B) {
                    constexpr int i0 = decltype(B)::value * 4;
                    if (i0 < namaxQ) {
                        RW_BLOCK();
                        rw_static_for<0, 4>([&](auto Q) {
                            constexpr int i = i0 + decltype(Q)::value;
                            if constexpr (i < CAPP) {                         // (beyond a working set: l_i = 0)
                                const R lq = rw_bc<i>(l[i >> 4]);
                                dnew = wv_fma(-lq, rw_bc<i>(q[i >> 4]), dnew);
                                ynew = wv_fma(-lq, rw_bc<i>(y[i >> 4]), ynew);
                            }
                        });
                    }
                });
                }
                const bool singular = (dnew < zero_tol) || (!is_soft && (na - nsoft) >= n);
                const R dinv = (R)1 / dnew;
                const int wsn = jadd | (is_soft ? kRowPosFlagSoft : 0) | (((sj & SENSE_IMMUTABLE) || (BNB && immX)) ? kRowPosFlagImm : 0) |
                                (lower ? kRowPosFlagLow : 0);
                const int wofn = (jadd & 15) * MS + (jadd >> 4);      // where the row sits inside a row of the staged M'
#pragma unroll
                for (int s = 0; s < S; s++) {
                    if (ap && pos[s] < na) lds[bo[s] + na] = l[s];           // new row: L(na, t) written by lane t
                    const bool here = ap && pos[s] == na;
                    ws[s] = here ? wsn : ws[s]; wof[s] = here ? wofn : wof[s];
                    rhs[s] = here ? rj : rhs[s]; lam[s] = here ? (R)0 : lam[s]; ls[s] = here ? (R)0 : ls[s];
                    y[s] = here ? ynew : y[s];
                    D[s] = here ? (singular ? (R)0 : dnew) : D[s];
                    Dinv[s] = here ? (singular ? (R)0 : dinv) : Dinv[s];
                }
                {
                    const bool mine = ap && li == (jadd & 15);
                    const unsigned bit = 1u << (jadd >> 4);
                    actb = mine ? (actb | bit) : actb;
                    lowb = (mine && lower) ? (lowb | bit) : lowb;
                }
                sing = (ap && singular) ? na : sing;
                nsoft = (ap && is_soft) ? nsoft + 1 : nsoft;
                na = ap ? na + 1 : na;
                napk = na > napk ? na : napk;
                {
                    const bool slow = ap && (fvalN - best < progress_tol);
                    const int cy = slow ? cyc + 1 : 0;
                    const bool cyx = slow && cy > cycle_tol;
                    cyc = ap ? cy : cyc;
                    best = (ap && !slow) ? fvalN : best;
                    flag = cyx ? EXIT_CYCLE : flag; fin = cyx ? 1 : fin;
                }
        }

        RWT(8);
        iter = ((run && !fin) || addF) ? iter + 1 : iter;
        if constexpr (BNB) {
            forced = addF ? -1 : forced;
            // =========================================================== rows whose NODE has ended: the search's next move
            if (rw_any(fin != 0)) {
                const bool fn = fin != 0;
                nodes = fn ? nodes + 1 : nodes;
                total_it = fn ? total_it + iter : total_it;
                const bool wsc = fn && flag == EXIT_WSCAP;       // (the working set outgrew this pass: the whole search is listed)
                const bool okn = fn && !wsc && flag >= 1;
                const bool leaf = okn && jbX == kRowBig;         // every binary row sits on a bound
                const bool better = leaf && (!have || fval < bestval);
                have = better ? 1 : have; bestval = better ? fval : bestval;
                bestact = better ? actb : bestact; bestlow = better ? lowb : bestlow;
#pragma unroll
                for (int s = 0; s < NS; s++) ubest[s] = better ? u[s] : ubest[s];
                const bool desc = okn && jbX != kRowBig;
                if (rw_any(desc)) {
                    RWS_BEGIN;
                    RWT_COUNT(16, 1);
                    // branch: the entry, what an append changes (the rest lazily, see the removal phase), the first child
                    // continues in place
                    stk_set(depth, jbX | (sideX << 10) | (na << 13) | (nsoft << 20), fval, desc);
                    abyte_set(depth, actb | (lowb << 4), desc);
                    const unsigned long long bit = 1ull << depth;
                    snap_clean = desc ? (snap_clean | bit) : snap_clean;
                    snap_saved = desc ? (snap_saved & ~bit) : snap_saved;
                    tried2 = desc ? (tried2 & ~bit) : tried2;
#pragma unroll
                    for (int r = 0; r < MS; r++) {
                        const bool mine = desc && li + 16 * r == jbX;
                        fixd[r] = mine ? depth : ((desc && fixd[r] == depth) ? 255 : fixd[r]);
                    }
                    forced = desc ? 2 * jbX + sideX : forced;
                    depth = desc ? depth + 1 : depth;
                    RWS_END(22);
                }
                const bool back = fn && !wsc && !desc;           // backtrack to the next untried side
                bool flip = false;
                if (rw_any(back)) {
                    RWS_BEGIN;
                    {
                        // the deepest level whose row still has an untried side
                        const unsigned long long open = ~tried2 & ((1ull << depth) - 1ull);
                        const int dn = open != 0ull ? 64 - (int)__builtin_clzll(open) : 0;
                        depth = back ? dn : depth;
                    }
                    flip = back && depth > 0;
                    RWS_END(23);
                    if (rw_any(flip)) {
                        RWS_BEGIN;
                        const int d = flip ? depth - 1 : 0;
                        const int e = stk_get(d);
                        const int ne = e ^ (1 << 10);
#pragma unroll
                        for (int q = 0; q < DS; q++) stk[q] = (flip && (d >> 4) == q && li == (d & 15)) ? ne : stk[q];
                        tried2 = flip ? (tried2 | (1ull << d)) : tried2;
                        restore(flip, d, e);                     // that node's optimal state: its second child, in place
                        forced = flip ? 2 * (ne & 1023) + ((ne >> 10) & 1) : forced;
                        RWS_END(20);
                    }
                }
                const bool limn = (desc || flip) && nodes >= 100000;
                bflag = wsc ? EXIT_WSCAP : (limn ? EXIT_ITERLIMIT : bflag);
                have = wsc ? 0 : have;
                const bool pfin = wsc || (back && !flip) || limn;
                const bool cont = fn && !pfin;
                iter = cont ? 1 : iter; cyc = cont ? 0 : cyc; best = cont ? (R)-1 : best;
                fin = pfin ? 1 : 0;
                // the problem's result: the best leaf
#pragma unroll
                for (int s = 0; s < NS; s++) u[s] = pfin ? (have ? ubest[s] : (R)0) : u[s];
                actb = pfin ? (have ? bestact : 0u) : actb;
                lowb = pfin ? (have ? bestlow : 0u) : lowb;
                flag = pfin ? (have ? (bflag == EXIT_ITERLIMIT ? EXIT_ITERLIMIT : EXIT_OPTIMAL) : bflag) : flag;
                iter = pfin ? total_it : iter;
            }
            RWT(14);
        }

        // =============================================================== rows whose problem has ended: outputs, clean-up
        if (rw_any(fin != 0)) {
            const RowParams<R> *a = RW_ARGS();
            const bool fn = fin != 0;
            const bool listed = fn && flag == EXIT_WSCAP && a->ovf_list != nullptr;
            const R *th = a->theta + (long long)(fn ? pid : 0) * nth;
            // x = R^-1 u + x0 + Xth theta   (mpc_update_qp.c:14-22); lane k of slot s writes output k + 16 s
#pragma unroll
            for (int s = 0; s < NS; s++) {
                if (16 * s < nout) {
                    const int ko = li + 16 * s;
                    const int lo = ko < nout ? ko : nout - 1;
                    // the shift x0 + Xth theta first: the parameter record and this output's row of Xth in ONE round trip (a loop
                    // over the nth entries waits for memory once per entry: 12 round trips where one serves)
                    constexpr int NTHF = 16;
                    R sh = ldk(prm.P.ox0, lo);
                    {
                        R tf[NTHF], xf[NTHF];
#pragma unroll
                        for (int t = 0; t < NTHF; t++) {
                            tf[t] = th[t < nth ? t : nth - 1];
                            xf[t] = ldk(prm.P.oXth + (t < nth ? t : nth - 1), lo * nth);
                        }
                        rw_static_for<0, NTHF / 4>([&](auto B) {
                            constexpr int t0 = decltype(B)::value * 4;
                            if (t0 < nth) {
                                RW_BLOCK();
#pragma unroll
                                for (int q = 0; q < 4; q++)
                                    if (t0 + q < nth) sh = wv_fma(xf[t0 + q], tf[t0 + q], sh);
                            }
                        });
                        for (int t = NTHF; t < nth; t++) sh = wv_fma(ldk(prm.P.oXth + t, lo * nth), th[t], sh);
                        RW_BLOCK();
                    }
                    R xs = (R)0;
                    // (this output's row of R^-1 in batches of 32 entries: L2 round trips; beyond n: u_c = 0)
                    constexpr int RVB = 16 * NS < 32 ? 16 * NS : 32;
                    rw_static_for<0, 16 * NS / RVB>([&](auto H) {
                        constexpr int h0 = decltype(H)::value * RVB;
                        if (h0 < n) {
                            R rv[RVB];
#pragma unroll
                            for (int c = 0; c < RVB; c++) rv[c] = ldk(prm.P.oRout, lo * n + (h0 + c < n ? h0 + c : n - 1));
                            rw_static_for<0, RVB / 4>([&](auto B) {
                                constexpr int c0 = decltype(B)::value * 4;
                                if (h0 + c0 < n) {
                                    RW_BLOCK();
                                    rw_static_for<0, 4>([&](auto Q) {
                                        constexpr int c = h0 + c0 + decltype(Q)::value;
                                        xs = wv_fma(rv[c - h0], rw_bc<c>(u[c >> 4]), xs);
                                    });
                                }
                            });
                            RW_BLOCK();
                        }
                    });
                    const R xo = xs + sh;
                    if (fn && ko < nout && a->X != nullptr) a->X[(long long)pid * nout + ko] = xo;
                }
            }
            if (a->active != nullptr) {
                unsigned long long acc = 0ull;
                const int words = prm.P.words;
#pragma unroll
                for (int r = 0; r < MS; r++) {
                    const bool ab = fn && ((actb >> r) & 1u), lo = (lowb >> r) & 1u;
                    const unsigned long long bu = __ballot(ab && !lo), bl = __ballot(ab && lo);
                    const unsigned long long mu = (bu >> (16 * g)) & 0xffffull, ml = (bl >> (16 * g)) & 0xffffull;
                    const int pu = 16 * r, pl = m + 16 * r;
                    if ((pu >> 6) == li) acc |= mu << (pu & 63);
                    if ((pu >> 6) + 1 == li && (pu & 63) > 48) acc |= mu >> (64 - (pu & 63));
                    if ((pl >> 6) == li) acc |= ml << (pl & 63);
                    if ((pl >> 6) + 1 == li && (pl & 63) > 48) acc |= ml >> (64 - (pl & 63));
                }
                if (fn && !listed && li < words) a->active[(long long)pid * words + li] = acc;
            }
            if (fn && li == 0) {
                if (a->exitflag != nullptr) a->exitflag[pid] = flag;
                if (a->iters != nullptr) a->iters[pid] = iter;
                if (listed) a->ovf_list[atomicAdd(a->ovf_count, 1)] = (int32_t)pid;
                if (a->stat != nullptr && !listed) {
                    atomicAdd(&a->stat[(myrow & 63) * 16 + (napk <= 24 ? 0 : (napk <= 32 ? 1 : (napk <= 48 ? 2 : 3)))], 1ull);
                    if (napk <= 16) atomicAdd(&a->stat[(myrow & 63) * 16 + 4], 1ull);      // (... and how many stayed within 16 rows)
                }
            }
            // the factor's rows back to zeros: the next problem of this row starts on a factor of zeros
            const int nclr = rw_max4(fn ? na : 0);
            for (int i = 1; i < nclr; i++) {
#pragma unroll
                for (int s = 0; s < S; s++)
                    if (fn && i < na && pos[s] < i) lds[bo[s] + i] = (R)0;
            }
            live = fn ? 0 : live;
            RWT(9);
            RWT_COUNT(13, (int)__popcll(__ballot(fn && li == 0)));
        }
    }
    RWT_FLUSH;
}

}  // namespace lmpc
