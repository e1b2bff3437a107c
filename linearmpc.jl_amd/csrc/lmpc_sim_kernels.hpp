// Closed-loop glue kernels for batched Monte-Carlo simulation: what the reference's Simulation
// loop does between two solves (reference src/simulation.jl:93-113: measure -> form theta ->
// compute_control -> dynamics), for N independent scenarios in lock-step on the device.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lmpc {

__device__ __forceinline__ double sim_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float sim_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// theta_i = [x_i ; r_i ; uprev_i]   (reference src/explicit.jl:54-63 with nd = np = 0)
template <typename R>
__global__ __launch_bounds__(256) void form_theta_kernel(
    R *__restrict__ theta, const R *__restrict__ x, const R *__restrict__ r,
    const R *__restrict__ uprev, int nx, int nr, int nup, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int nth = nx + nr + nup;
    R *t = theta + i * nth;
    for (int k = 0; k < nx; k++) t[k] = x[i * nx + k];
    for (int k = 0; k < nr; k++) t[nx + k] = r ? r[i * nr + k] : (R)0;
    for (int k = 0; k < nup; k++) t[nx + nr + k] = uprev ? uprev[i * nup + k] : (R)0;
}

// One block of theta taken from a trajectory: `w` values per column, `T` columns stored column by
// column (a Julia w x T matrix), one matrix per scenario (`stride` doubles apart) or one shared by
// all (stride 0).  H == 0: the block is column k0; H > 0 (preview): columns k0 .. k0+H-1.  Columns
// past the end repeat the last one (reference utils.jl:101-106 pads a short trajectory with its
// last column, simulation.jl:128-134 get_preview clamps the same way); src == nullptr gives zeros
// (utils.jl:80-82 `isnothing(r)`).
struct ThetaBlock {
    const double *src;
    long long stride;
    int w, T, k0, H;
    __host__ __device__ int width() const { return w * (H > 0 ? H : 1); }
};

// theta_i = [x_i ; r-block ; d-block ; uprev_i ; p-block]   (reference explicit.jl:54-63 form_parameter
// after format_reference / format_disturbance / format_affine_parameters, utils.jl:78-261);
// one thread per entry of theta, consecutive threads write consecutive addresses.
__global__ __launch_bounds__(256) void form_parameter_kernel(
    double *__restrict__ theta, const double *__restrict__ x, int nx, ThetaBlock r, ThetaBlock d,
    const double *__restrict__ uprev, int nup, ThetaBlock p, long long n) {
    const int nr = r.width(), nd = d.width(), npp = p.width();
    const int nth = nx + nr + nd + nup + npp;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * nth) return;
    const long long i = idx / nth;
    int e = (int)(idx - i * nth);
    auto from_block = [&](const ThetaBlock &b, int q) -> double {
        if (b.src == nullptr) return 0.0;
        int col = b.k0 + q / b.w;
        col = col < b.T ? col : b.T - 1;
        col = col > 0 ? col : 0;
        return b.src[i * b.stride + (long long)col * b.w + (q % b.w)];
    };
    double v;
    if (e < nx) v = x[i * nx + e];
    else if ((e -= nx) < nr) v = from_block(r, e);
    else if ((e -= nr) < nd) v = from_block(d, e);
    else if ((e -= nd) < nup) v = uprev ? uprev[i * nup + e] : 0.0;
    else v = from_block(p, e - nup);
    theta[idx] = v;
}

// theta_i = [state_i ; reference_i ; disturbance_i ; control_i[0:nup] ; affine_parameter_i] from the
// five arrays the reference's generated controller takes (codegen/mpc_update_parameter.c:1-29), one
// record per problem in each array; a NULL array gives zeros.  nph > 0 (N_PREVIEW_HORIZON, i.e.
// settings.reference_condensation): reference_i holds nr*nph values (an nr x nph trajectory, column
// by column) and entry j of the block is sum_i reference_i[i] * t2s[i*nr + j], accumulated in index
// order with separate multiply and add exactly as the C loop does (mpc_update_parameter.c:10-16).
// One thread per entry of theta.
__global__ __launch_bounds__(256) void update_parameter_kernel(
    double *__restrict__ theta, const double *__restrict__ control, int ncontrol,
    const double *__restrict__ state, int nx, const double *__restrict__ reference, int nr, int nph,
    const double *__restrict__ t2s, const double *__restrict__ disturbance, int nd, int nup,
    const double *__restrict__ parameter, int npar, long long n) {
    const int nth = nx + nr + nd + nup + npar;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * nth) return;
    const long long i = idx / nth;
    int e = (int)(idx - i * nth);
    double v = 0.0;
    if (e < nx) v = state[i * nx + e];
    else if ((e -= nx) < nr) {
        if (reference == nullptr) v = 0.0;
        else if (nph > 0) {
            const double *ri = reference + i * (long long)nr * nph;
            for (int q = 0; q < nr * nph; q++) v = __dadd_rn(v, __dmul_rn(ri[q], t2s[(long long)q * nr + e]));
        } else v = reference[i * nr + e];
    }
    else if ((e -= nr) < nd) v = disturbance ? disturbance[i * nd + e] : 0.0;
    else if ((e -= nd) < nup) v = control ? control[i * ncontrol + e] : 0.0;
    else v = parameter ? parameter[i * npar + (e - nup)] : 0.0;
    theta[idx] = v;
}

// x_i <- F x_i + G u_i (sums in index order, F then G), uprev_i <- u_i, bookkeeping of the run.
// NXT > 0: compile-time state count (<= 8), the state record read and written with wide accesses
// (element by element from run-time loops a 32-byte record moved at a quarter of the rate); 0: run-time nx.
template <int NXT>
__global__ __launch_bounds__(256) void plant_kernel(
    double *__restrict__ x, double *__restrict__ uprev, const double *__restrict__ u,
    const int32_t *__restrict__ flag, const double *__restrict__ FG, int nx_rt, int nu, int nup,
    double *__restrict__ xtraj_next, double *__restrict__ utraj, int32_t *__restrict__ flag_min,
    int first, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int NXA = NXT > 0 ? NXT : 32;
    const int nx = NXT > 0 ? NXT : nx_rt;
    const double *F = FG, *G = FG + nx * nx;
    double xo[NXA], xn[NXA];
    if constexpr (NXT > 0) {
#pragma unroll
        for (int c = 0; c < NXT; c++) xo[c] = x[i * NXT + c];
#pragma unroll
        for (int a = 0; a < NXT; a++) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < NXT; c++) acc = __builtin_fma(F[a * NXT + c], xo[c], acc);
            for (int l = 0; l < nu; l++) acc = __builtin_fma(G[a * nu + l], u[i * nu + l], acc);
            xn[a] = acc;
        }
#pragma unroll
        for (int a = 0; a < NXT; a++) x[i * NXT + a] = xn[a];
        if (xtraj_next) {
#pragma unroll
            for (int a = 0; a < NXT; a++) xtraj_next[i * NXT + a] = xn[a];
        }
    } else {
        for (int c = 0; c < nx; c++) xo[c] = x[i * nx + c];
        for (int a = 0; a < nx; a++) {
            double acc = 0.0;
            for (int c = 0; c < nx; c++) acc = __builtin_fma(F[a * nx + c], xo[c], acc);
            for (int l = 0; l < nu; l++) acc = __builtin_fma(G[a * nu + l], u[i * nu + l], acc);
            xn[a] = acc;
        }
        for (int a = 0; a < nx; a++) {
            x[i * nx + a] = xn[a];
            if (xtraj_next) xtraj_next[i * nx + a] = xn[a];
        }
    }
    for (int l = 0; l < nu; l++) {
        if (l < nup) uprev[i * nup + l] = u[i * nu + l];
        if (utraj) utraj[i * nu + l] = u[i * nu + l];
    }
    if (flag_min) {
        const int32_t f = flag[i];
        flag_min[i] = first ? f : (f < flag_min[i] ? f : flag_min[i]);
    }
}

// The same plant step with the scenario's state kept INSIDE its theta record (theta = [x; r; uprev]):
// x <- F x + G u is written back into theta[0:nx], u into theta[nx+nr : nx+nr+nup], so the next
// step's solve reads the record as it stands and no theta has to be formed again (the closed loop
// with a constant reference then moves 2 x 8 nth + 8 nu bytes per scenario and step instead of
// re-reading x and re-writing the whole record in a separate kernel).  Same sums in the same order
// as plant_kernel.  x_out / uprev_out (last step): the caller's arrays.
template <typename R>
__global__ __launch_bounds__(256) void plant_theta_kernel(
    R *__restrict__ theta, int nth, int nr, const R *__restrict__ u,
    const int32_t *__restrict__ flag, const R *__restrict__ FG, int nx, int nu, int nup,
    R *__restrict__ xtraj_next, R *__restrict__ utraj, int32_t *__restrict__ flag_min,
    int first, R *__restrict__ x_out, R *__restrict__ uprev_out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const R *F = FG, *G = FG + nx * nx;
    R *t = theta + i * nth;
    R xo[32], xn[32];
    for (int c = 0; c < nx; c++) xo[c] = t[c];
    for (int a = 0; a < nx; a++) {
        R acc = (R)0;
        for (int c = 0; c < nx; c++) acc = sim_fma(F[a * nx + c], xo[c], acc);
        for (int l = 0; l < nu; l++) acc = sim_fma(G[a * nu + l], u[i * nu + l], acc);
        xn[a] = acc;
    }
    for (int a = 0; a < nx; a++) {
        t[a] = xn[a];
        if (xtraj_next) xtraj_next[i * nx + a] = xn[a];
        if (x_out) x_out[i * nx + a] = xn[a];
    }
    for (int l = 0; l < nu; l++) {
        const R ul = u[i * nu + l];
        if (l < nup) {
            t[nx + nr + l] = ul;
            if (uprev_out) uprev_out[i * nup + l] = ul;
        }
        if (utraj) utraj[i * nu + l] = ul;
    }
    if (flag_min) {
        const int32_t f = flag[i];
        flag_min[i] = first ? f : (f < flag_min[i] ? f : flag_min[i]);
    }
}

// The reference's generated state observer for N scenarios (codegen/mpc_observer.c:1-28; arrays as
// src/observer.jl:136-138 writes them): `dyn` = MPC_PLANT_DYNAMICS, one row [f_offset_i, F_i, G_i, Gd_i]
// per state; `meas` = MPC_MEASUREMENT_FUNCTION, one row [h_offset_j, C_j, Dd_j] per measurement;
// `kt` = K_TRANSPOSE_OBSERVER (ny x nx).  Sums in the C code's order with separate multiply and add
// (the reference compiles it with gcc -O3 -msse3: no fused multiply-add).  One thread per scenario.
// NXT > 0: the state count is the compile-time constant NXT (<= 8): the record is read and written with
// wide accesses (every index static); NXT == 0: run-time nx (<= 32).
template <int NXT>
__global__ __launch_bounds__(256) void predict_state_kernel(
    double *__restrict__ state, const double *__restrict__ control, const double *__restrict__ disturbance,
    const double *__restrict__ dyn, int nx_rt, int nu, int nd, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int NXA = NXT > 0 ? NXT : 32;
    const int nx = NXT > 0 ? NXT : nx_rt;
    double xo[NXA], xn[NXA];
    if constexpr (NXT > 0) {
#pragma unroll
        for (int c = 0; c < NXT; c++) xo[c] = state[i * NXT + c];
    } else {
        for (int c = 0; c < nx; c++) xo[c] = state[i * nx + c];
    }
    const int stride = 1 + nx + nu + nd;
    auto row = [&](int a) -> double {
        const double *d = dyn + a * stride;
        double acc = d[0];
        if constexpr (NXT > 0) {
#pragma unroll
            for (int c = 0; c < NXT; c++) acc = __dadd_rn(acc, __dmul_rn(d[1 + c], xo[c]));
        } else {
            for (int c = 0; c < nx; c++) acc = __dadd_rn(acc, __dmul_rn(d[1 + c], xo[c]));
        }
        for (int l = 0; l < nu; l++) acc = __dadd_rn(acc, __dmul_rn(d[1 + nx + l], control[i * nu + l]));
        for (int q = 0; q < nd; q++)
            acc = __dadd_rn(acc, __dmul_rn(d[1 + nx + nu + q], disturbance ? disturbance[i * nd + q] : 0.0));
        return acc;
    };
    if constexpr (NXT > 0) {
#pragma unroll
        for (int a = 0; a < NXT; a++) xn[a] = row(a);
#pragma unroll
        for (int a = 0; a < NXT; a++) state[i * NXT + a] = xn[a];
    } else {
        for (int a = 0; a < nx; a++) xn[a] = row(a);
        for (int a = 0; a < nx; a++) state[i * nx + a] = xn[a];
    }
}

template <int NXT>
__global__ __launch_bounds__(256) void correct_state_kernel(
    double *__restrict__ state, const double *__restrict__ measurement, const double *__restrict__ disturbance,
    const double *__restrict__ meas, const double *__restrict__ kt, int nx_rt, int ny, int nd, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int NXA = NXT > 0 ? NXT : 32;
    const int nx = NXT > 0 ? NXT : nx_rt;
    double xo[NXA], xn[NXA];
    if constexpr (NXT > 0) {
#pragma unroll
        for (int c = 0; c < NXT; c++) { xo[c] = state[i * NXT + c]; xn[c] = xo[c]; }
    } else {
        for (int c = 0; c < nx; c++) { xo[c] = state[i * nx + c]; xn[c] = xo[c]; }
    }
    const int stride = 1 + nx + nd;
    for (int j = 0; j < ny; j++) {
        const double *mr = meas + j * stride;
        double inno = __dsub_rn(measurement[i * ny + j], mr[0]);
        if constexpr (NXT > 0) {
#pragma unroll
            for (int c = 0; c < NXT; c++) inno = __dsub_rn(inno, __dmul_rn(mr[1 + c], xo[c]));
        } else {
            for (int c = 0; c < nx; c++) inno = __dsub_rn(inno, __dmul_rn(mr[1 + c], xo[c]));
        }
        for (int q = 0; q < nd; q++)
            inno = __dsub_rn(inno, __dmul_rn(mr[1 + nx + q], disturbance ? disturbance[i * nd + q] : 0.0));
        if constexpr (NXT > 0) {
#pragma unroll
            for (int c = 0; c < NXT; c++) xn[c] = __dadd_rn(xn[c], __dmul_rn(kt[j * NXT + c], inno));
        } else {
            for (int c = 0; c < nx; c++) xn[c] = __dadd_rn(xn[c], __dmul_rn(kt[j * nx + c], inno));
        }
    }
    if constexpr (NXT > 0) {
#pragma unroll
        for (int c = 0; c < NXT; c++) state[i * NXT + c] = xn[c];
    } else {
        for (int c = 0; c < nx; c++) state[i * nx + c] = xn[c];
    }
}

// theta = [x; r; uprev] records -> the caller's x and uprev arrays (end of a fused closed loop)
__global__ __launch_bounds__(256) void unpack_theta_kernel(
    const double *__restrict__ theta, double *__restrict__ x, double *__restrict__ uprev, int nx, int nr, int nup,
    long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *t = theta + i * (nx + nr + nup);
    for (int a = 0; a < nx; a++) x[i * nx + a] = t[a];
    if (uprev) for (int l = 0; l < nup; l++) uprev[i * nup + l] = t[nx + nr + l];
}

// mpc_get_estimated_state + mpc_get_estimated_disturbance of the generated offset-free observer code
// (reference src/observer.jl:163-175) for N scenarios: state = observer_state[0:nx], disturbance =
// [measured_disturbance (or zeros); observer_state[nx : nx+ndo]].  One thread per output entry.
__global__ __launch_bounds__(256) void split_observer_state_kernel(
    double *__restrict__ state, double *__restrict__ disturbance, const double *__restrict__ observer_state,
    const double *__restrict__ measured, int nx, int ndm, int ndo, long long n) {
    const int w = nx + ndm + ndo, nobs = nx + ndo;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * w) return;
    const long long i = idx / w;
    const int e = (int)(idx - i * w);
    if (e < nx) state[i * nx + e] = observer_state[i * nobs + e];
    else if (e < nx + ndm) disturbance[i * (ndm + ndo) + (e - nx)] = measured ? measured[i * ndm + (e - nx)] : 0.0;
    else disturbance[i * (ndm + ndo) + (e - nx)] = observer_state[i * nobs + nx + (e - nx - ndm)];
}

}  // namespace lmpc
