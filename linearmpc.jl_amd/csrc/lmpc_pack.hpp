// Host-side constant pack of one condensed-MPC least-distance problem and the device view of it.
// Internal to liblmpc_hip.so (the public surface is include/lmpc_hip.h).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/lmpc_hip.h"

namespace lmpc {

// DAQP sense flags (reference mpc2mpqp.jl:868-885 uses DAQP.IMMUTABLE/EQUALITY/SOFT/BINARY)
enum : int32_t { SENSE_ACTIVE = 1, SENSE_LOWER = 2, SENSE_IMMUTABLE = 4, SENSE_SOFT = 8, SENSE_BINARY = 16 };

// DAQP exit flags
enum : int32_t {
    EXIT_SOFT_OPTIMAL = 2, EXIT_OPTIMAL = 1, EXIT_INFEASIBLE = -1, EXIT_CYCLE = -2,
    EXIT_UNBOUNDED = -3, EXIT_ITERLIMIT = -4, EXIT_NONCONVEX = -5, EXIT_OVERDETERMINED = -6,
    EXIT_WSCAP = -7,          // not a DAQP flag: the working set would have outgrown the 64 lanes of the wavefront
    EXIT_UNFINISHED = -8      // not a DAQP flag: queued inside the one-launch kernel and never solved (a bounded wait of
                              // that kernel ran out; the handle reports LMPC_ERR_HIP at its next check, lmpc_check)
};

// All matrices row-major.  G is the packed lower triangle of M M' (index a>=b: a(a+1)/2+b),
// accumulated with the same fma chain the solver would use at run time, so looking a product up
// is bit-identical to recomputing it.
struct HostPack {
    int n = 0, m = 0, ms = 0, nth = 0, nout = 0, nsoft = 0;
    std::vector<double> M, G, du0, dl0, Dth, Rout, x0, Xth;
    std::vector<int32_t> sense;
    // affine variational inequality (non-symmetric H, DAQP's is_avi; lmpc_avi_kernel.hpp): M holds ML = [I;A] scaled,
    // MR row j = (H^-1 ML_j')', Gf = ML MR' in full (m x m, not symmetric); G (the packed triangle) stays empty
    bool avi = false;
    std::vector<double> MR, Gf;
    // proximal-point mode (eps_prox > 0; a symmetric positive semidefinite H): the AVI pack of H + eps I plus
    // Hinv = (H + eps I)^-1 (n x n), x0f / Xthf = the full-length affine map of the unconstrained optimum, Kth = the
    // feedback term of the outputs (nout x nth)
    bool prox = false;
    double eps_prox = 0.0;
    std::vector<double> Hinv, x0f, Xthf, Kth;
    int words() const { return (2 * m + 63) / 64; }
};

// QP -> LDP transform of reference codegen.jl:239-280 (qp2ldp) on column-major Julia arrays.
// Returns LMPC_OK or LMPC_ERR_NONCONVEX / LMPC_ERR_INFEASIBLE / LMPC_ERR_BADARG.
int qp_to_ldp(HostPack &P, int n, int m, int ms, int nth, int nout,
              const double *H, const double *f, const double *f_theta, const double *A,
              const double *bu, const double *bl, const double *W, const int32_t *sense,
              const double *Kfb, int nx, std::string &err);

// The same boundary for a NON-symmetric H with H + H' positive definite (several objectives: reference
// mpc2mpqp.jl:900-950; setup.jl:13 is_avi): fills the AVI pack (see HostPack).  LMPC_ERR_NONCONVEX if the symmetric
// part of H is not positive definite.
int qp_to_avi(HostPack &P, int n, int m, int ms, int nth, int nout,
              const double *H, const double *f, const double *f_theta, const double *A,
              const double *bu, const double *bl, const double *W, const int32_t *sense,
              const double *Kfb, int nx, std::string &err);

// Proximal-point mode: qp_to_avi on H + eps I (H symmetric) + the arrays of the outer iteration.
int qp_to_prox(HostPack &P, int n, int m, int ms, int nth, int nout,
               const double *H, const double *f, const double *f_theta, const double *A,
               const double *bu, const double *bl, const double *W, const int32_t *sense,
               const double *Kfb, int nx, double eps, std::string &err);

// isapprox(H, H', rtol = 1e-9) as the reference decides mpQP.is_symmetric (mpc2mpqp.jl:897); H column-major n x n
bool h_is_symmetric(const double *H, int n);

// Fills P.G and P.nsoft from P.M / P.sense; validates shapes.
int finish_pack(HostPack &P, std::string &err);

}  // namespace lmpc
