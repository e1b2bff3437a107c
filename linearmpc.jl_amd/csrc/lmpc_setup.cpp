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

bool h_is_symmetric(const double *H, int n) {
    double dif = 0.0, nrm = 0.0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            const double a = H[i + (size_t)n * j], b = H[j + (size_t)n * i];
            dif += (a - b) * (a - b);
            nrm += a * a;
        }
    return std::sqrt(dif) <= 1e-9 * std::sqrt(nrm);
}

int qp_to_avi(HostPack &P, int n, int m, int ms, int nth, int nout,
              const double *H, const double *f, const double *f_theta, const double *A,
              const double *bu, const double *bl, const double *W, const int32_t *sense,
              const double *Kfb, int nx, std::string &err) {
    if (n <= 0 || m < 0 || ms < 0 || ms > m || ms > n || nth < 0 || nout <= 0 || nout > n || !H ||
        (m > 0 && (!bu || !bl)) || (m > ms && !A) || nx < 0 || nx > nth) {
        err = "lmpc_setup: bad dimensions or NULL array";
        return LMPC_ERR_BADARG;
    }
    const int mg = m - ms;
    P.n = n; P.m = m; P.ms = ms; P.nth = nth; P.nout = nout; P.avi = true;
    // strong monotonicity: the symmetric part of H must be positive definite (Cholesky as the test)
    {
        std::vector<double> Rc((size_t)n * n, 0.0);
        auto Hs = [&](int i, int j) { return 0.5 * (H[i + (size_t)n * j] + H[j + (size_t)n * i]); };
        for (int i = 0; i < n; i++) {
            double d = Hs(i, i);
            for (int k = 0; k < i; k++) d -= Rc[(size_t)k * n + i] * Rc[(size_t)k * n + i];
            if (!(d > 0.0) || !std::isfinite(d)) {
                err = "lmpc_setup: the symmetric part of H is not positive definite";
                return LMPC_ERR_NONCONVEX;
            }
            const double rii = std::sqrt(d);
            Rc[(size_t)i * n + i] = rii;
            for (int j = i + 1; j < n; j++) {
                double s = Hs(i, j);
                for (int k = 0; k < i; k++) s -= Rc[(size_t)k * n + i] * Rc[(size_t)k * n + j];
                Rc[(size_t)i * n + j] = s / rii;
            }
        }
    }
    // Hinv = H^-1 by Gauss-Jordan elimination with partial pivoting (row-major work copies)
    std::vector<double> Aw((size_t)n * n), Hinv((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) Aw[(size_t)i * n + j] = H[i + (size_t)n * j];
        Hinv[(size_t)i * n + i] = 1.0;
    }
    for (int c = 0; c < n; c++) {
        int piv = c;
        for (int r = c + 1; r < n; r++)
            if (std::fabs(Aw[(size_t)r * n + c]) > std::fabs(Aw[(size_t)piv * n + c])) piv = r;
        if (!(std::fabs(Aw[(size_t)piv * n + c]) > 0.0)) { err = "lmpc_setup: H is singular"; return LMPC_ERR_NONCONVEX; }
        if (piv != c)
            for (int k = 0; k < n; k++) {
                std::swap(Aw[(size_t)piv * n + k], Aw[(size_t)c * n + k]);
                std::swap(Hinv[(size_t)piv * n + k], Hinv[(size_t)c * n + k]);
            }
        const double d = 1.0 / Aw[(size_t)c * n + c];
        for (int k = 0; k < n; k++) { Aw[(size_t)c * n + k] *= d; Hinv[(size_t)c * n + k] *= d; }
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            const double fct = Aw[(size_t)r * n + c];
            if (fct == 0.0) continue;
            for (int k = 0; k < n; k++) {
                Aw[(size_t)r * n + k] -= fct * Aw[(size_t)c * n + k];
                Hinv[(size_t)r * n + k] -= fct * Hinv[(size_t)c * n + k];
            }
        }
    }
    // ML = [I_ms; A], MR_j = (H^-1 ML_j')'
    P.M.assign((size_t)m * n, 0.0);
    P.MR.assign((size_t)m * n, 0.0);
    for (int i = 0; i < ms; i++) P.M[(size_t)i * n + i] = 1.0;
    for (int g = 0; g < mg; g++)
        for (int c = 0; c < n; c++) P.M[(size_t)(ms + g) * n + c] = A[g + (size_t)mg * c];
    for (int j = 0; j < m; j++)
        for (int k = 0; k < n; k++) {
            double s = 0.0;
            for (int c = 0; c < n; c++) s += Hinv[(size_t)k * n + c] * P.M[(size_t)j * n + c];
            P.MR[(size_t)j * n + k] = s;
        }
    // hf = H^-1 f, Hth = H^-1 f_theta
    std::vector<double> hf(n, 0.0), Hth((size_t)n * nth, 0.0);
    for (int k = 0; k < n; k++) {
        double s = 0.0;
        for (int c = 0; c < n; c++) s += Hinv[(size_t)k * n + c] * (f ? f[c] : 0.0);
        hf[k] = s;
        for (int t = 0; t < nth; t++) {
            double q = 0.0;
            for (int c = 0; c < n; c++) q += Hinv[(size_t)k * n + c] * (f_theta ? f_theta[c + (size_t)n * t] : 0.0);
            Hth[(size_t)k * nth + t] = q;
        }
    }
    P.Dth.assign((size_t)m * nth, 0.0);
    P.du0.assign(m, 0.0);
    P.dl0.assign(m, 0.0);
    P.sense.assign(m, 0);
    for (int j = 0; j < m; j++) {
        double *ml = &P.M[(size_t)j * n], *mr = &P.MR[(size_t)j * n];
        double shift = 0.0, gjj = 0.0;
        for (int k = 0; k < n; k++) { shift += ml[k] * hf[k]; gjj += ml[k] * mr[k]; }
        for (int t = 0; t < nth; t++) {
            double s = W ? W[j + (size_t)m * t] : 0.0;
            for (int k = 0; k < n; k++) s += ml[k] * Hth[(size_t)k * nth + t];
            P.Dth[(size_t)j * nth + t] = s;
        }
        P.du0[j] = bu[j] + shift;
        P.dl0[j] = bl[j] + shift;
        if (gjj > 0.0) {                                   // scaled so that G_jj = 1
            const double sc = std::sqrt(gjj);
            for (int k = 0; k < n; k++) { ml[k] /= sc; mr[k] /= sc; }
            for (int t = 0; t < nth; t++) P.Dth[(size_t)j * nth + t] /= sc;
            P.du0[j] /= sc;
            P.dl0[j] /= sc;
        }
        P.sense[j] = sense ? sense[j] : 0;
    }
    P.Gf.assign((size_t)m * m, 0.0);
    for (int a = 0; a < m; a++)
        for (int b = 0; b < m; b++) {
            double acc = 0.0;
            for (int k = 0; k < n; k++) acc = std::fma(P.M[(size_t)a * n + k], P.MR[(size_t)b * n + k], acc);
            P.Gf[(size_t)a * m + b] = acc;
        }
    // x = u + x_unc(theta): Rout = leading rows of I, x0 = -(H^-1 f), Xth = -(H^-1 f_theta) - K
    P.Rout.assign((size_t)nout * n, 0.0);
    P.x0.assign(nout, 0.0);
    P.Xth.assign((size_t)nout * nth, 0.0);
    for (int k = 0; k < nout; k++) {
        P.Rout[(size_t)k * n + k] = 1.0;
        P.x0[k] = -hf[k];
        for (int t = 0; t < nth; t++) {
            double s = -Hth[(size_t)k * nth + t];
            if (Kfb && t < nx) s -= Kfb[k + (size_t)nout * t];
            P.Xth[(size_t)k * nth + t] = s;
        }
    }
    P.G.clear();
    P.nsoft = 0;
    for (int j = 0; j < m; j++) {
        if (P.sense[j] & SENSE_SOFT) P.nsoft++;
        if (P.sense[j] & SENSE_BINARY) {
            err = "lmpc_setup: binary rows together with a variational objective are not supported";
            return LMPC_ERR_UNSUPPORTED;
        }
        if (!(P.du0[j] >= P.dl0[j])) {
            err = "lmpc: lower bound above upper bound in row " + std::to_string(j);
            return LMPC_ERR_INFEASIBLE;
        }
    }
    return LMPC_OK;
}

int qp_to_prox(HostPack &P, int n, int m, int ms, int nth, int nout,
               const double *H, const double *f, const double *f_theta, const double *A,
               const double *bu, const double *bl, const double *W, const int32_t *sense,
               const double *Kfb, int nx, double eps, std::string &err) {
    if (n <= 0 || !H || !(eps > 0.0)) { err = "lmpc_setup: bad dimensions or eps_prox <= 0"; return LMPC_ERR_BADARG; }
    if (!h_is_symmetric(H, n)) {
        err = "lmpc_setup: proximal iterations (eps_prox > 0) need a symmetric H";
        return LMPC_ERR_UNSUPPORTED;
    }
    std::vector<double> Ht((size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Ht[i + (size_t)n * j] = 0.5 * (H[i + (size_t)n * j] + H[j + (size_t)n * i]) + (i == j ? eps : 0.0);
    // the subproblems' pack: the AVI transform of H + eps I, WITHOUT the feedback in its output map (the outer iteration
    // works on x itself; the feedback is added to the outputs at the very end, Kth)
    int rc = qp_to_avi(P, n, m, ms, nth, nout, Ht.data(), f, f_theta, A, bu, bl, W, sense, nullptr, 0, err);
    if (rc != LMPC_OK) return rc;                       // (H + eps I not positive definite: H is indefinite -> -5)
    P.prox = true; P.eps_prox = eps;
    // Hinv again (qp_to_avi keeps it to itself), by the same elimination
    std::vector<double> Aw((size_t)n * n);
    P.Hinv.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) Aw[(size_t)i * n + j] = Ht[i + (size_t)n * j];
        P.Hinv[(size_t)i * n + i] = 1.0;
    }
    for (int c = 0; c < n; c++) {
        int piv = c;
        for (int r = c + 1; r < n; r++)
            if (std::fabs(Aw[(size_t)r * n + c]) > std::fabs(Aw[(size_t)piv * n + c])) piv = r;
        if (piv != c)
            for (int k = 0; k < n; k++) {
                std::swap(Aw[(size_t)piv * n + k], Aw[(size_t)c * n + k]);
                std::swap(P.Hinv[(size_t)piv * n + k], P.Hinv[(size_t)c * n + k]);
            }
        const double d = 1.0 / Aw[(size_t)c * n + c];
        for (int k = 0; k < n; k++) { Aw[(size_t)c * n + k] *= d; P.Hinv[(size_t)c * n + k] *= d; }
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            const double fct = Aw[(size_t)r * n + c];
            if (fct == 0.0) continue;
            for (int k = 0; k < n; k++) {
                Aw[(size_t)r * n + k] -= fct * Aw[(size_t)c * n + k];
                P.Hinv[(size_t)r * n + k] -= fct * P.Hinv[(size_t)c * n + k];
            }
        }
    }
    P.x0f.assign(n, 0.0);
    P.Xthf.assign((size_t)n * nth, 0.0);
    for (int k = 0; k < n; k++) {
        double s = 0.0;
        for (int c = 0; c < n; c++) s += P.Hinv[(size_t)k * n + c] * (f ? f[c] : 0.0);
        P.x0f[k] = -s;
        for (int t = 0; t < nth; t++) {
            double q = 0.0;
            for (int c = 0; c < n; c++) q += P.Hinv[(size_t)k * n + c] * (f_theta ? f_theta[c + (size_t)n * t] : 0.0);
            P.Xthf[(size_t)k * nth + t] = -q;
        }
    }
    P.Kth.assign((size_t)nout * nth, 0.0);
    for (int k = 0; k < nout; k++)
        for (int t = 0; t < nth && t < nx; t++)
            if (Kfb) P.Kth[(size_t)k * nth + t] = -Kfb[k + (size_t)nout * t];
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
    // the reference hands to DAQP as an affine variational inequality (setup.jl:13 is_avi = !is_symmetric): that is
    // qp_to_avi's problem, not this transform's -- refuse instead of silently solving the symmetrised problem.
    if (!h_is_symmetric(H, n)) {
        err = "lmpc: H is not symmetric (variational objective): set up with is_avi (lmpc_setup does so by itself, "
              "lmpc_setup_ex takes the flag DAQP.setup takes); the LDP transform does not apply";
        return LMPC_ERR_UNSUPPORTED;
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
