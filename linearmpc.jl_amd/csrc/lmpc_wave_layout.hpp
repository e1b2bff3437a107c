// Kernel argument blocks: problem sizes, offsets into the constant pack, settings -- of the lane /
// screening kernels (PackLayout) and of the wavefront kernel (WaveLayout).
#pragma once

namespace lmpc {

// Offsets (in doubles) of the constant arrays inside the single device buffer.
struct PackLayout {
    int n, m, ms, nth, nout, words;
    int oM, oG, odu, odl, oDth, oRout, ox0, oXth;   // offsets into the double buffer
    int oDthP, oBnd, oXthP, nthp;                   // screening copies: rows zero-padded to nthp columns,
                                                    // bounds interleaved (du0_j, dl0_j)
    unsigned long long imm_mask, eq_mask;           // m <= 64: IMMUTABLE rows / rows flagged ACTIVE
    double primal_tol, dual_tol, zero_tol, progress_tol, fval_bound, rho_soft;
    int cycle_tol, iter_limit;
};

struct WaveLayout {
    int n, m, ms, nth, nout, words;
    int cap, ldc;                                   // working-set capacity, leading dim of L
    int oM, oMt, oG, odu, odl, oDth, oRout, ox0, oXth;
    double primal_tol, dual_tol, zero_tol, progress_tol, fval_bound, rho_soft;
    int cycle_tol, iter_limit;
};

}  // namespace lmpc
