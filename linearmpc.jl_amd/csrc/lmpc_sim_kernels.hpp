// Closed-loop glue kernels for batched Monte-Carlo simulation: what the reference's Simulation
// loop does between two solves (reference src/simulation.jl:93-113: measure -> form theta ->
// compute_control -> dynamics), for N independent scenarios in lock-step on the device.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lmpc {

// theta_i = [x_i ; r_i ; uprev_i]   (reference src/explicit.jl:54-63 with nd = np = 0)
__global__ __launch_bounds__(256) void form_theta_kernel(
    double *__restrict__ theta, const double *__restrict__ x, const double *__restrict__ r,
    const double *__restrict__ uprev, int nx, int nr, int nup, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int nth = nx + nr + nup;
    double *t = theta + i * nth;
    for (int k = 0; k < nx; k++) t[k] = x[i * nx + k];
    for (int k = 0; k < nr; k++) t[nx + k] = r ? r[i * nr + k] : 0.0;
    for (int k = 0; k < nup; k++) t[nx + nr + k] = uprev ? uprev[i * nup + k] : 0.0;
}

// x_i <- F x_i + G u_i (sums in index order, F then G), uprev_i <- u_i, bookkeeping of the run
__global__ __launch_bounds__(256) void plant_kernel(
    double *__restrict__ x, double *__restrict__ uprev, const double *__restrict__ u,
    const int32_t *__restrict__ flag, const double *__restrict__ FG, int nx, int nu, int nup,
    double *__restrict__ xtraj_next, double *__restrict__ utraj, int32_t *__restrict__ flag_min,
    int first, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *F = FG, *G = FG + nx * nx;
    double xn[32];
    for (int a = 0; a < nx; a++) {
        double acc = 0.0;
        for (int c = 0; c < nx; c++) acc = __builtin_fma(F[a * nx + c], x[i * nx + c], acc);
        for (int l = 0; l < nu; l++) acc = __builtin_fma(G[a * nu + l], u[i * nu + l], acc);
        xn[a] = acc;
    }
    for (int a = 0; a < nx; a++) {
        x[i * nx + a] = xn[a];
        if (xtraj_next) xtraj_next[i * nx + a] = xn[a];
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

}  // namespace lmpc
