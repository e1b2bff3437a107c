// Host half of the boundary: what DAQP.setup / DAQP.update precompute for LinearMPC.jl
// (reference setup.jl:11-13, utils.jl:272-281) and what its code generator bakes into C arrays
// (reference codegen.jl:239-280 `qp2ldp`), done once per controller on the CPU in f64.
#include <cmath>
#include <cstring>

#include "lmpc_pack.hpp"

namespace lmpc {

namespace {
inline size_t tri(size_t i) { return i * (i + 1) / 2; }
}

int finish_pack(HostPack &P, std::string &err) {
    const int n = P.n, m = P.m;
    if (n <= 0 || m < 0 || P.nth < 0 || P.nout <= 0 || P.nout > n || P.ms < 0 || P.ms > m) {
        err = "lmpc: inconsistent dimensions";
        return LMPC_ERR_BADARG;
    }
    P.G.assign(tri(m), 0.0);
    for (int a = 0; a < m; a++)
        for (int b = 0; b <= a; b++) {
            double acc = 0.0;
            for (int k = 0; k < n; k++) acc = std::fma(P.M[(size_t)a * n + k], P.M[(size_t)b * n + k], acc);
            P.G[tri(a) + b] = acc;
        }
    P.nsoft = 0;
    for (int j = 0; j < m; j++) {
        if (P.sense[j] & SENSE_SOFT) P.nsoft++;
        if (!(P.du0[j] >= P.dl0[j])) {
            err = "lmpc: lower bound above upper bound in row " + std::to_string(j);
            return LMPC_ERR_INFEASIBLE;
        }
    }
    return LMPC_OK;
}

int qp_to_ldp(HostPack &P, int n, int m, int ms, int nth, int nout,
              const double *H, const double *f, const double *f_theta, const double *A,
              const double *bu, const double *bl, const double *W, const int32_t *sense,
              const double *Kfb, int nx, std::string &err) {
    if (n <= 0 || m < 0 || ms < 0 || ms > m || ms > n || nth < 0 || nout <= 0 || nout > n || !H ||
        (m > 0 && (!bu || !bl)) || (m > ms && !A) || nx < 0 || nx > nth) {
        err = "lmpc_setup: bad dimensions or NULL array";
        return LMPC_ERR_BADARG;
    }
    const int mg = m - ms;
    P.n = n; P.m = m; P.ms = ms; P.nth = nth; P.nout = nout;

    // A variational objective (reference mpc2mpqp.jl:900-950, several players) gives a NON-symmetric H, which
    // the reference hands to DAQP as an affine variational inequality (setup.jl:13 is_avi = !is_symmetric,
    // mpc2mpqp.jl:897 isapprox(H, H', rtol = 1e-9)).  That mode is not built here: refuse instead of silently
    // solving the symmetrised problem.
    {
        double dif = 0.0, nrm = 0.0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                const double a = H[i + (size_t)n * j], b = H[j + (size_t)n * i];
                dif += (a - b) * (a - b);
                nrm += a * a;
            }
        if (!(std::sqrt(dif) <= 1e-9 * std::sqrt(nrm))) {
            err = "lmpc_setup: H is not symmetric (variational objective, DAQP's is_avi mode): not supported";
            return LMPC_ERR_UNSUPPORTED;
        }
    }
    // upper Cholesky factor of the symmetrised Hessian, H = R'R (codegen.jl:242)
    std::vector<double> R((size_t)n * n, 0.0);
    auto Hs = [&](int i, int j) { return 0.5 * (H[i + (size_t)n * j] + H[j + (size_t)n * i]); };
    for (int i = 0; i < n; i++) {
        double d = Hs(i, i);
        for (int k = 0; k < i; k++) d -= R[(size_t)k * n + i] * R[(size_t)k * n + i];
        if (!(d > 0.0) || !std::isfinite(d)) {
            err = "lmpc_setup: Hessian is not positive definite";
            return LMPC_ERR_NONCONVEX;
        }
        const double rii = std::sqrt(d);
        R[(size_t)i * n + i] = rii;
        for (int j = i + 1; j < n; j++) {
            double s = Hs(i, j);
            for (int k = 0; k < i; k++) s -= R[(size_t)k * n + i] * R[(size_t)k * n + j];
            R[(size_t)i * n + j] = s / rii;
        }
    }
    // Rinv = R^-1 (upper triangular), column by column
    std::vector<double> Rinv((size_t)n * n, 0.0);
    for (int c = 0; c < n; c++) {
        Rinv[(size_t)c * n + c] = 1.0 / R[(size_t)c * n + c];
        for (int i = c - 1; i >= 0; i--) {
            double s = 0.0;
            for (int k = i + 1; k <= c; k++) s -= R[(size_t)i * n + k] * Rinv[(size_t)k * n + c];
            Rinv[(size_t)i * n + c] = s / R[(size_t)i * n + i];
        }
    }
    // Mext = [I_ms; A] * Rinv  (codegen.jl:243); simple rows are rows of Rinv
    P.M.assign((size_t)m * n, 0.0);
    for (int i = 0; i < ms; i++)
        for (int c = 0; c < n; c++) P.M[(size_t)i * n + c] = Rinv[(size_t)i * n + c];
    for (int g = 0; g < mg; g++)
        for (int c = 0; c < n; c++) {
            double s = 0.0;
            for (int k = 0; k <= c; k++) s += A[g + (size_t)mg * k] * Rinv[(size_t)k * n + c];
            P.M[(size_t)(ms + g) * n + c] = s;
        }
    // Vth = R' \ f_theta, v = R' \ f  (forward substitution with the lower factor R')
    std::vector<double> Vth((size_t)n * nth, 0.0), v(n, 0.0);
    for (int t = 0; t <= nth; t++) {
        for (int i = 0; i < n; i++) {
            double s;
            if (t < nth) s = f_theta ? f_theta[i + (size_t)n * t] : 0.0;
            else s = f ? f[i] : 0.0;
            for (int k = 0; k < i; k++)
                s -= R[(size_t)k * n + i] * (t < nth ? Vth[(size_t)k * nth + t] : v[k]);
            s /= R[(size_t)i * n + i];
            if (t < nth) Vth[(size_t)i * nth + t] = s; else v[i] = s;
        }
    }
    // Dth = W + Mext*Vth ; du/dl = bu/bl + Mext*v  (codegen.jl:246-249), then row normalisation
    P.Dth.assign((size_t)m * nth, 0.0);
    P.du0.assign(m, 0.0);
    P.dl0.assign(m, 0.0);
    P.sense.assign(m, 0);
    for (int j = 0; j < m; j++) {
        const double *mj = &P.M[(size_t)j * n];
        double shift = 0.0, nrm2 = 0.0;
        for (int k = 0; k < n; k++) { shift += mj[k] * v[k]; nrm2 += mj[k] * mj[k]; }
        for (int t = 0; t < nth; t++) {
            double s = W ? W[j + (size_t)m * t] : 0.0;
            for (int k = 0; k < n; k++) s += mj[k] * Vth[(size_t)k * nth + t];
            P.Dth[(size_t)j * nth + t] = s;
        }
        P.du0[j] = bu[j] + shift;
        P.dl0[j] = bl[j] + shift;
        const double nrm = std::sqrt(nrm2);
        if (nrm > 0.0) {
            for (int k = 0; k < n; k++) P.M[(size_t)j * n + k] /= nrm;
            for (int t = 0; t < nth; t++) P.Dth[(size_t)j * nth + t] /= nrm;
            P.du0[j] /= nrm;
            P.dl0[j] /= nrm;
        }
        P.sense[j] = sense ? sense[j] : 0;
    }
    // output maps: x = Rinv u - H\f - (H\f_theta) theta - K theta[0:nx]   (codegen.jl:269-273,:157)
    P.Rout.assign((size_t)nout * n, 0.0);
    P.x0.assign(nout, 0.0);
    P.Xth.assign((size_t)nout * nth, 0.0);
    for (int k = 0; k < nout; k++) {
        double s0 = 0.0;
        for (int c = k; c < n; c++) {
            P.Rout[(size_t)k * n + c] = Rinv[(size_t)k * n + c];
            s0 -= Rinv[(size_t)k * n + c] * v[c];
        }
        P.x0[k] = s0;
        for (int t = 0; t < nth; t++) {
            double s = 0.0;
            for (int c = k; c < n; c++) s -= Rinv[(size_t)k * n + c] * Vth[(size_t)c * nth + t];
            if (Kfb && t < nx) s -= Kfb[k + (size_t)nout * t];
            P.Xth[(size_t)k * nth + t] = s;
        }
    }
    return finish_pack(P, err);
}

}  // namespace lmpc
