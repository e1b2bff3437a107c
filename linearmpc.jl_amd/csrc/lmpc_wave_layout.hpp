// Kernel argument blocks: problem sizes, offsets into the constant pack, settings -- of the lane /
// screening kernels (PackLayout) and of the wavefront kernel (WaveLayout).
#pragma once

namespace lmpc {

constexpr int kWaveRunAheadSlots = 6;  // wavefront-kernel instantiations (constraint slots per lane) built with run-ahead
constexpr int kShards = 64;            // work-list segments (one atomic counter each)
constexpr int kCountStride = 32;       // ... whose counters sit one per 128-byte line

// Closed-loop mode of the lane / screening kernels (lmpc_simulate*): the kernel that finishes a
// problem also advances its scenario -- x+ = F x + G u -- and writes the NEXT step's record
// [x+; r; u] to theta_out, so a closed-loop step is the solve and nothing else.  FG == nullptr: off.
struct SimFuse {
    const double *FG;            // non-null = closed-loop mode on; F (nx*nx) then G (nx*nu), row-major, are
                                 // read from the constant pack at PackLayout::oFG (scalar loads: the pack is
                                 // const __restrict__, a pointer inside this struct is not)
    double *theta_out;           // next step's records (a different buffer than the one being read)
    int *flag_min;               // smallest exit flag over the steps so far, or nullptr
    double *xtraj;               // this step's slot of the state trajectory, or nullptr
    int nx, nu, nr, nup, first;
    // scenario-asynchronous closed loop (kstep != nullptr): every scenario carries its own step counter, the
    // records are updated IN PLACE (theta_out is the buffer being read), trajectories are addressed by the
    // scenario's own step
    int *kstep;
    double *utraj, *xtraj_base;
    long long nscen;
};

// Generated-controller mode of the screening kernel (lmpc_compute_control*): theta is not read from a
// buffer but assembled from the five argument arrays of mpc_compute_control (state == nullptr: off);
// problems that need iterations get their record written to theta_out for the iterating kernel.
struct GatherArgs {
    const double *state, *reference, *disturbance, *control, *parameter;
    double *theta_out;
    int nx, nr, nd, nup, np, ncontrol;
};

// Offsets (in doubles) of the constant arrays inside the single device buffer.
struct PackLayout {
    int n, m, ms, nth, nout, words;
    int oM, oG, odu, odl, oDth, oRout, ox0, oXth;   // offsets into the double buffer
    int oFG;                                        // closed loop: room for F and G behind the pack (SimFuse)
    int oDthP, oBnd, oXthP, nthp;                   // screening copies: rows zero-padded to nthp columns,
                                                    // bounds interleaved (du0_j, dl0_j)
    unsigned long long imm_mask, eq_mask;           // m <= 64: IMMUTABLE rows / rows flagged ACTIVE
    double primal_tol, dual_tol, zero_tol, progress_tol, fval_bound, rho_soft;
    int cycle_tol, iter_limit;
    SimFuse sim;
    GatherArgs gat;
};

// Scenario-asynchronous closed loop on the wavefront-kernel path (lmpc_simulate*, kstep != nullptr): the kernel that
// solves a listed scenario's step also advances the scenario IN PLACE -- x+ = F x + G u (sums in the oracle's order,
// F then G), theta <- [x+; r; u], its step counter + 1, trajectories at the scenario's own step -- exactly what
// sim_advance does on the lane path (lmpc_lane_kernel.hpp).  Binary64 only.
struct WaveSim {
    const double *FG;            // F (nx*nx) then G (nx*nu), row-major
    int *kstep;                  // per-scenario step counters (scenario-asynchronous rounds); FG == nullptr = off
    double *utraj, *xtraj;       // (T, N, nu) / (T + 1, N, nx) or nullptr
    int *flag_min;               // smallest exit flag over the steps so far, or nullptr
    int nx, nu, nr, nup;
    long long nscen;
    int kfix = -1;               // step-synchronous loop: the step every scenario is at (no counters); -1: kstep[]
    int T = 0;                   // scenario-asynchronous loop: steps per scenario -- a scenario whose step ends with a
                                 // non-empty working set runs on inside the wavefront kernel (0: one step per visit)
};

struct WaveLayout {
    int n, m, ms, nth, nout, words;
    int cap, ldc;                                   // working-set capacity, leading dim of L
    int oM, oMt, oG, odu, odl, oDth, oRout, ox0, oXth;
    int oGf;                                        // full symmetric Gram matrix, m x m (Gram-scan form)
    int nC;                                         // reals in the pack (bounds of the kernels' buffer resource)
    int keepStride;                                 // reals of one scenario's kept closed-loop state: 2 x 64 + c (c - 1) / 2 at
                                                    // the handle's LARGEST capacity c (a first pass may run at a smaller cap)
    double primal_tol, dual_tol, zero_tol, progress_tol, fval_bound, rho_soft;
    int cycle_tol, iter_limit;
};

// Affine-variational-inequality kernel (lmpc_avi_kernel.hpp): offsets into its constant pack
struct AviLayout {
    int n, m, nth, nout, words, cap;
    int oML, oMR, oG, odu, odl, oDth, oRout, ox0, oXth, nC;
    double primal_tol, dual_tol, zero_tol, rho_soft;
    int iter_limit;
    // proximal-point mode (eps_prox > 0): (H + eps I)^-1, the full-length affine map, the outputs' feedback term
    int oHinv, ox0f, oXthf, oKth;
    double eps_prox, eta_prox;
    // register-resident kernels (lmpc_avi_tiers_kernel.hpp, m == n <= 8), laid out for batched scalar loads:
    // oTh2: per column t of theta 2 n reals -- column t of Dth, then column t of Xth (rows >= nout zero);
    // oBnd3: du0, dl0, x0 (padded to n); oSd: the diagonal of ML
    int oTh2, oBnd3, oSd;
};

}  // namespace lmpc
