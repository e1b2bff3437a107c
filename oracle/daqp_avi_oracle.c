/*
 * TEST INFRASTRUCTURE ONLY -- CPU parity oracle for the batched path's affine-variational-inequality mode.
 *
 * Nothing in the shipped package may include, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it (as the checker / the reported CPU baseline, never as the product).
 *
 * What it restates
 * ----------------
 * A LinearMPC.jl controller with several objectives (set_objective!(mpc, uids; ...), /root/reference/src/setup.jl:137-151;
 * create_variational_objective, /root/reference/src/mpc2mpqp.jl:900-950) has a NON-symmetric H: block row i is the
 * gradient of player i's cost with respect to its own inputs.  The reference hands such a problem to DAQP with
 * is_avi = !mpQP.is_symmetric (/root/reference/src/setup.jl:11-13) and then calls the very same online path
 * (solve, /root/reference/src/utils.jl:268-283): find x with  bl + W th <= [I;A] x <= bu + W th  and
 *
 *        (H x + f + f_theta th)' (y - x) >= 0   for every feasible y            (affine variational inequality)
 *
 * i.e. the KKT system  H x + f(th) + A' mu = 0,  mu_j >= 0 on rows at their upper bound, <= 0 at their lower bound,
 * 0 elsewhere -- the Nash equilibrium of the players' QPs.  H + H' positive definite makes the solution unique, so
 * any correct solver reproduces what the reference's own test pins (test/runtests.jl:1337-1358: closed loop ends at
 * y = [10, 0], atol 1e-4).
 *
 * libdaqp's AVI code (DAQPBase ~0.4.4, /root/reference/Project.toml:9,30) is a third-party binary that is not under
 * /root/reference and not installed here; its iteration path is NOT restated ("parity vs DAQP internals: unpinned").
 * The algorithm below is this build's own statement of the dual active-set iteration for that problem class, the direct
 * generalisation of daqp_ldp_oracle.c:
 *
 *     x = x_unc(th) + u,   x_unc = -H^-1 (f + f_theta th),     u = - sum_{j in W} MR_j lam_j,
 *     ML = [I;A] (rows scaled),  MR_j = (H^-1 ML_j')',  G = ML MR'   (m x m, NOT symmetric, G + G' > 0),
 *     working set W:  G_WW lam* = -d_W   with   dl <= ML u <= du  the shifted bounds  (d = du / dl + Dth th)
 *
 * with a recursively updated L D U factorisation of G_WW (unit lower L, unit upper U, shared pivots D): appending a row
 * adds a row to L and a column to U, removing one is a rank-one update of the trailing block (Bennett's algorithm,
 * the non-symmetric form of the LDL' update in the cited paper's sec. IV-B).  Everything else is daqp_ldp_oracle.c's
 * loop: most-violated-row selection with primal_tol, ratio test with dual_tol along lam -> lam*, singular branch when a
 * pivot vanishes (rows of W linearly dependent: G_WW p = 0 <=> ML_W' p = 0), iteration limit.  Because a row stays
 * in W from the moment it is appended until the full step that satisfies it, this is Cottle & Dantzig's principal
 * pivoting method on the LCP with matrix G, which terminates for positive definite G.  There is no dual objective to
 * watch for an AVI, so the progress / cycle guard of the QP solver has no counterpart here (the iteration limit is the
 * guard) and fval_bound is not used.
 *
 * Pinned by: the reference's closed-loop end values above, an independent projection iteration on the box-constrained
 * test problem, and KKT residuals <= 1e-10 on random problems with general and soft rows (tests/test_oracle.py).
 *
 * Arithmetic contract (shared with lmpc_avi_kernel.hpp so that results are bit-comparable): IEEE binary64, every
 * multiply-add an explicit fma(), sums in index order, pivots applied through their stored reciprocal; build with
 * -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SENSE_ACTIVE 1
#define SENSE_LOWER 2
#define SENSE_IMMUTABLE 4
#define SENSE_SOFT 8

#define EXIT_SOFT_OPTIMAL 2
#define EXIT_OPTIMAL 1
#define EXIT_INFEASIBLE (-1)
#define EXIT_CYCLE (-2)
#define EXIT_ITERLIMIT (-4)
#define EXIT_OVERDETERMINED_INITIAL (-6)
#define EXIT_WSCAP (-7)

#define TRI(i) (((i) * ((i) + 1)) / 2)

typedef struct {
    int32_t n, m, ms, nth, nout;
    const double *ML;    /* m x n row-major: rows of [I;A], scaled so that G_jj = 1          */
    const double *MR;    /* m x n row-major: row j = (H^-1 ML_j')'                             */
    const double *G;     /* m x m row-major: G[i*m+j] = ML_i . MR_j                            */
    const double *du0, *dl0, *Dth, *Rout, *x0, *Xth;
    const int32_t *sense;
    /* proximal-point mode only (oracle_avi_prox_solve_batch), NULL otherwise */
    const double *Hinv;  /* n x n row-major: (H + eps I)^-1                                  */
    const double *x0f;   /* n: -(H + eps I)^-1 f                                              */
    const double *Xthf;  /* n x nth: -(H + eps I)^-1 f_theta                                  */
    const double *Kth;   /* nout x nth: what the outputs get on top of x (the -K x feedback)  */
} oracle_avi;

/* same layout as daqp_ldp_oracle.c's oracle_settings_abi */
typedef struct {
    double primal_tol, dual_tol, zero_tol, progress_tol, fval_bound, rho_soft;
    int32_t cycle_tol, iter_limit;
    int32_t mode, pad_;
} avi_settings;

typedef struct {
    int n, m, cap, na, sing, reuse, nsoft_act;
    double *L, *Ut, *D, *Dinv, *lam, *ls, *xl, *zl, *pv, *qv, *u, *ut, *dup, *dlo;
    int *WS;
    int32_t *sense;
    double soft_slack;
} awork;

static awork *awork_new(int n, int m, int cap) {
    awork *w = (awork *)calloc(1, sizeof(awork));
    w->n = n; w->m = m; w->cap = cap;
    w->L = (double *)calloc((size_t)TRI(cap + 1), sizeof(double));
    w->Ut = (double *)calloc((size_t)TRI(cap + 1), sizeof(double));
    double **v[] = {&w->D, &w->Dinv, &w->lam, &w->ls, &w->xl, &w->zl, &w->pv, &w->qv};
    for (unsigned k = 0; k < sizeof(v) / sizeof(v[0]); k++) *v[k] = (double *)calloc(cap + 1, sizeof(double));
    w->u = (double *)calloc(n > 0 ? n : 1, sizeof(double));
    w->ut = (double *)calloc(n > 0 ? n : 1, sizeof(double));
    w->dup = (double *)calloc(m > 0 ? m : 1, sizeof(double));
    w->dlo = (double *)calloc(m > 0 ? m : 1, sizeof(double));
    w->WS = (int *)calloc(cap + 1, sizeof(int));
    w->sense = (int32_t *)calloc(m > 0 ? m : 1, sizeof(int32_t));
    return w;
}

static void awork_free(awork *w) {
    free(w->L); free(w->Ut); free(w->D); free(w->Dinv); free(w->lam); free(w->ls); free(w->xl); free(w->zl);
    free(w->pv); free(w->qv); free(w->u); free(w->ut); free(w->dup); free(w->dlo); free(w->WS); free(w->sense); free(w);
}

/* Append row j:  [G_WW c; r' g_jj] = [L 0; l' 1] diag(D, d) [U u; 0 1]  with  L D u = c,  U' D l = r,
 * d = g_jj - l' D u.  Ut holds U transposed (unit lower), so both new rows are forward substitutions. */
static void ldu_add(awork *w, const oracle_avi *p, const avi_settings *s, int j) {
    const int na = w->na, m = w->m;
    double *rl = &w->L[TRI(na)], *ru = &w->Ut[TRI(na)];
    for (int i = 0; i < na; i++) {
        rl[i] = p->G[(size_t)j * m + w->WS[i]];          /* r: row j of G over the columns of W  */
        ru[i] = p->G[(size_t)w->WS[i] * m + j];          /* c: column j of G over the rows of W  */
    }
    double dnew = p->G[(size_t)j * m + j];
    if (w->sense[j] & SENSE_SOFT) dnew += s->rho_soft;
    for (int i = 0; i < na; i++) {
        double al = rl[i], au = ru[i];
        const double *ui = &w->Ut[TRI(i)], *li = &w->L[TRI(i)];
        for (int t = 0; t < i; t++) {
            al = fma(-ui[t], rl[t], al);                  /* U' qL = r */
            au = fma(-li[t], ru[t], au);                  /* L  qU = c */
        }
        rl[i] = al; ru[i] = au;
    }
    for (int i = 0; i < na; i++) {
        const double ql = rl[i], qu = ru[i];
        const double l = ql * w->Dinv[i];
        rl[i] = l;
        ru[i] = qu * w->Dinv[i];
        dnew = fma(-l, qu, dnew);
    }
    rl[na] = 1.0; ru[na] = 1.0;
    const int is_soft = (w->sense[j] & SENSE_SOFT) != 0;
    if (dnew < s->zero_tol || (!is_soft && na - w->nsoft_act >= w->n)) {
        w->D[na] = 0.0; w->Dinv[na] = 0.0; w->sing = na;
    } else {
        w->D[na] = dnew; w->Dinv[na] = 1.0 / dnew;
    }
    w->WS[na] = j; w->lam[na] = 0.0; w->ls[na] = 0.0;
    w->sense[j] |= SENSE_ACTIVE;
    w->nsoft_act += is_soft;
    w->na = na + 1;
}

/* Drop position r:  L33' D3' U33' = L33 D3 U33 + D_r l32 u23'  (Bennett's rank-one update of an LDU factorisation). */
static void ldu_remove(awork *w, const avi_settings *s, int r) {
    const int na = w->na, nup = na - r - 1;
    double alpha = w->D[r];
    for (int t = 0; t < nup; t++) {
        w->pv[t] = w->L[TRI(r + 1 + t) + r];
        w->qv[t] = w->Ut[TRI(r + 1 + t) + r];
    }
    for (int i = r; i < na - 1; i++) {
        double *dl_ = &w->L[TRI(i)], *du_ = &w->Ut[TRI(i)];
        const double *sl = &w->L[TRI(i + 1)], *su = &w->Ut[TRI(i + 1)];
        for (int c = 0; c < r; c++) { dl_[c] = sl[c]; du_[c] = su[c]; }
        for (int c = r; c < i; c++) { dl_[c] = sl[c + 1]; du_[c] = su[c + 1]; }
        dl_[i] = 1.0; du_[i] = 1.0;
    }
    w->sing = -1;
    for (int t = 0; t < nup; t++) {
        const int i = r + t;
        const double pt = w->pv[t], qt = w->qv[t];
        const double dold = w->D[i + 1];
        const double dbar = fma(alpha * pt, qt, dold);
        if (dbar < s->zero_tol) {
            w->D[i] = 0.0; w->Dinv[i] = 0.0; w->sing = i;
            for (int q = i + 1; q < na - 1; q++) { w->D[q] = w->D[q + 1]; w->Dinv[q] = w->Dinv[q + 1]; }
            break;
        }
        const double rinv = 1.0 / dbar;
        const double betaL = (qt * alpha) * rinv;
        const double betaU = (pt * alpha) * rinv;
        alpha = (dold * alpha) * rinv;
        w->D[i] = dbar; w->Dinv[i] = rinv;
        for (int q = t + 1; q < nup; q++) {
            double *lqi = &w->L[TRI(r + q) + i], *uqi = &w->Ut[TRI(r + q) + i];
            w->pv[q] = fma(-pt, *lqi, w->pv[q]);
            *lqi = fma(betaL, w->pv[q], *lqi);
            w->qv[q] = fma(-qt, *uqi, w->qv[q]);
            *uqi = fma(betaU, w->qv[q], *uqi);
        }
    }
    if (w->sense[w->WS[r]] & SENSE_SOFT) w->nsoft_act--;
    w->sense[w->WS[r]] &= ~(SENSE_ACTIVE | SENSE_LOWER);
    for (int i = r; i < na - 1; i++) { w->WS[i] = w->WS[i + 1]; w->lam[i] = w->lam[i + 1]; }
    w->na = na - 1;
    if (r < w->reuse) w->reuse = r;
}

static int avi_solve_shift(awork *w, const oracle_avi *p, const avi_settings *s, const double *theta,
                           const uint64_t *warm, int32_t *iters, const double *xk, double eps) {
    const int n = p->n, m = p->m, nth = p->nth;
    int flag = EXIT_ITERLIMIT, iter = 1;
    for (int j = 0; j < m; j++) {                               /* mpc_update_qp.c:1-10 */
        double sh = 0.0;
        for (int t = 0; t < nth; t++) sh = fma(p->Dth[(size_t)j * nth + t], theta[t], sh);
        w->dup[j] = p->du0[j] + sh;
        w->dlo[j] = p->dl0[j] + sh;
        if (xk) {                                               /* proximal term: the linear term moved by -eps x_k */
            double acc = 0.0;
            for (int c = 0; c < n; c++) acc = fma(p->MR[(size_t)j * n + c], xk[c], acc);
            w->dup[j] = w->dup[j] - eps * acc;
            w->dlo[j] = w->dlo[j] - eps * acc;
        }
    }
    w->na = 0; w->sing = -1; w->reuse = 0; w->nsoft_act = 0; w->soft_slack = 0.0;
    for (int k = 0; k < n; k++) w->u[k] = 0.0;
    for (int j = 0; j < m; j++) w->sense[j] = p->sense[j] & ~SENSE_LOWER;
    for (int j = 0; j < m; j++) {                               /* initial working set: equalities, warm mask */
        const int s0 = p->sense[j];
        int want = (s0 & SENSE_ACTIVE) != 0, lower = want && (s0 & SENSE_LOWER);
        if (warm && !(s0 & SENSE_IMMUTABLE)) {
            if ((warm[j >> 6] >> (j & 63)) & 1) want = 1;
            else if ((warm[(m + j) >> 6] >> ((m + j) & 63)) & 1) { want = 1; lower = 1; }
        }
        if (!want) { w->sense[j] &= ~SENSE_ACTIVE; continue; }
        if (lower) w->sense[j] |= SENSE_LOWER;
        if (w->na >= w->cap) { flag = EXIT_WSCAP; goto done; }
        ldu_add(w, p, s, j);
        if (w->sing >= 0) {
            if (s0 & SENSE_IMMUTABLE) { flag = EXIT_OVERDETERMINED_INITIAL; goto done; }
            w->na--; w->sing = -1;
            if (w->sense[j] & SENSE_SOFT) w->nsoft_act--;
            w->sense[j] &= ~(SENSE_ACTIVE | SENSE_LOWER);
        }
    }
    for (; iter < s->iter_limit; iter++) {
        const int na = w->na;
        if (w->sing < 0) {
            int nblock = 0, rm = -1, add = -1, isupper = 0;
            double alpha = 0.0;
            /* (L D U) lam* = -d_W */
            for (int i = w->reuse; i < na; i++) {
                const int j = w->WS[i];
                double acc = (w->sense[j] & SENSE_LOWER) ? -w->dlo[j] : -w->dup[j];
                const double *li = &w->L[TRI(i)];
                for (int t = 0; t < i; t++) acc = fma(-li[t], w->xl[t], acc);
                w->xl[i] = acc;
            }
            for (int i = w->reuse; i < na; i++) w->zl[i] = w->xl[i] * w->Dinv[i];
            for (int i = na - 1; i >= 0; i--) {
                double acc = w->zl[i];
                for (int t = na - 1; t > i; t--) acc = fma(-w->Ut[TRI(t) + i], w->ls[t], acc);
                w->ls[i] = acc;
            }
            w->reuse = na;
            for (int i = 0; i < na; i++) {
                const int j = w->WS[i];
                if (w->sense[j] & SENSE_IMMUTABLE) continue;
                if (w->sense[j] & SENSE_LOWER) { if (w->ls[i] < s->dual_tol) continue; }
                else if (w->ls[i] > -s->dual_tol) continue;
                const double cand = -w->lam[i] / (w->ls[i] - w->lam[i]);
                if (nblock == 0 || cand < alpha) { alpha = cand; rm = i; }
                nblock++;
            }
            /* the iterate now (from lam) and the target of this step (from lam*), soft slack at the target */
            double soft = 0.0;
            for (int k = 0; k < n; k++) { w->u[k] = 0.0; w->ut[k] = 0.0; }
            for (int i = 0; i < na; i++) {
                const int j = w->WS[i];
                const double *mi = &p->MR[(size_t)j * n];
                const double lc = w->lam[i], lt = w->ls[i];
                for (int k = 0; k < n; k++) {
                    w->u[k] = fma(-mi[k], lc, w->u[k]);
                    w->ut[k] = fma(-mi[k], lt, w->ut[k]);
                }
                if (w->sense[j] & SENSE_SOFT) soft = fma(lt * lt, s->rho_soft, soft);
            }
            /* one pass over the rows: most violated row AT THE TARGET, and the first row that is satisfied now but
             * violated at the target (it blocks the step where it reaches its bound) */
            double min_val = -s->primal_tol, tblk = nblock ? alpha : 1.0;
            int broken = 0, pblk = -1, pup = 0;
            for (int j = 0; j < m; j++) {
                if (w->sense[j] & SENSE_IMMUTABLE) continue;
                const double *mj = &p->ML[(size_t)j * n];
                double Mc = 0.0, Mt = 0.0;
                for (int k = 0; k < n; k++) { Mc = fma(mj[k], w->u[k], Mc); Mt = fma(mj[k], w->ut[k], Mt); }
                const double vu = w->dup[j] - Mt, vl = -(w->dlo[j] - Mt);
                if (w->sense[j] & SENSE_ACTIVE) {
                    if (!(w->sense[j] & SENSE_SOFT) && (vu < -s->primal_tol || vl < -s->primal_tol)) broken = 1;
                    continue;
                }
                if (vu < min_val) { add = j; isupper = 1; min_val = vu; }
                else if (vl < min_val) { add = j; isupper = 0; min_val = vl; }
                const double cu = w->dup[j] - Mc, cl = -(w->dlo[j] - Mc);
                if (vu < -s->primal_tol && cu >= -s->primal_tol) {
                    const double t = cu > 0.0 ? cu / (cu - vu) : 0.0;
                    if (t < tblk) { tblk = t; pblk = j; pup = 1; }
                } else if (vl < -s->primal_tol && cl >= -s->primal_tol) {
                    const double t = cl > 0.0 ? cl / (cl - vl) : 0.0;
                    if (t < tblk) { tblk = t; pblk = j; pup = 0; }
                }
            }
            if (pblk >= 0) {
                /* a satisfied row reaches its bound first: stop there and take it into the working set */
                if (na >= w->cap) { flag = EXIT_WSCAP; break; }
                for (int i = 0; i < na; i++) w->lam[i] = fma(tblk, w->ls[i] - w->lam[i], w->lam[i]);
                if (!pup) w->sense[pblk] |= SENSE_LOWER;
                ldu_add(w, p, s, pblk);
            } else if (nblock) {
                for (int i = 0; i < na; i++) w->lam[i] = fma(alpha, w->ls[i] - w->lam[i], w->lam[i]);
                ldu_remove(w, s, rm);
            } else {
                /* full step */
                for (int k = 0; k < n; k++) w->u[k] = w->ut[k];
                w->soft_slack = soft;
                if (add < 0) {
                    if (broken) flag = EXIT_CYCLE;
                    else flag = (w->soft_slack > s->primal_tol) ? EXIT_SOFT_OPTIMAL : EXIT_OPTIMAL;
                    break;
                }
                if (na >= w->cap) { flag = EXIT_WSCAP; break; }
                for (int i = 0; i < na; i++) w->lam[i] = w->ls[i];
                if (!isupper) w->sense[add] |= SENSE_LOWER;
                ldu_add(w, p, s, add);
            }
        } else {
            /* singular working set: G_WW p = 0 <=> U p = e_sg (D_sg = 0), p_sg = +-1 */
            const int sg = w->sing;
            const double *us = &w->Ut[TRI(sg)];
            for (int i = sg - 1; i >= 0; i--) {
                double acc = -us[i];
                for (int t = sg - 1; t > i; t--) acc = fma(-w->Ut[TRI(t) + i], w->ls[t], acc);
                w->ls[i] = acc;
            }
            w->ls[sg] = 1.0;
            if (w->sense[w->WS[sg]] & SENSE_LOWER)
                for (int i = 0; i <= sg; i++) w->ls[i] = -w->ls[i];
            for (int i = sg + 1; i < na; i++) w->ls[i] = 0.0;
            int nblock = 0, rm = -1;
            double alpha = 0.0;
            for (int i = 0; i < na; i++) {
                const int j = w->WS[i];
                if (w->sense[j] & SENSE_IMMUTABLE) continue;
                if (w->sense[j] & SENSE_LOWER) { if (w->ls[i] < s->dual_tol) continue; }
                else if (w->ls[i] > -s->dual_tol) continue;
                const double cand = -w->lam[i] / w->ls[i];
                if (nblock == 0 || cand < alpha) { alpha = cand; rm = i; }
                nblock++;
            }
            if (nblock == 0) { flag = EXIT_INFEASIBLE; break; }
            for (int i = 0; i < na; i++) w->lam[i] = fma(alpha, w->ls[i], w->lam[i]);
            ldu_remove(w, s, rm);
        }
    }
done:
    if (iters) *iters = iter;
    return flag;
}

static int avi_solve(awork *w, const oracle_avi *p, const avi_settings *s, const double *theta,
                     const uint64_t *warm, int32_t *iters) {
    return avi_solve_shift(w, p, s, theta, warm, iters, NULL, 0.0);
}

static void avi_outputs(const awork *w, const oracle_avi *p, const double *theta, double *xout, uint64_t *active, int nwords) {
    const int n = p->n, nth = p->nth;
    for (int k = 0; k < p->nout; k++) {                          /* mpc_update_qp.c:14-22 */
        double xs = 0.0, sh = p->x0[k];
        for (int c = 0; c < n; c++) xs = fma(p->Rout[(size_t)k * n + c], w->u[c], xs);
        for (int t = 0; t < nth; t++) sh = fma(p->Xth[(size_t)k * nth + t], theta[t], sh);
        xout[k] = xs + sh;
    }
    if (active) {
        for (int q = 0; q < nwords; q++) active[q] = 0;
        for (int i = 0; i < w->na; i++) {
            const int j = w->WS[i];
            const int bit = (w->sense[j] & SENSE_LOWER) ? p->m + j : j;
            active[bit >> 6] |= (uint64_t)1 << (bit & 63);
        }
    }
}

void oracle_avi_solve_batch(const oracle_avi *p, const avi_settings *s, int64_t N, const double *theta,
                            const uint64_t *warm, double *x, int32_t *exitflag, int32_t *iters, uint64_t *active) {
    int nsoft = 0;
    for (int j = 0; j < p->m; j++) nsoft += (p->sense[j] & SENSE_SOFT) ? 1 : 0;
    awork *w = awork_new(p->n, p->m, p->n + 1 + nsoft);
    const int nwords = (2 * p->m + 63) / 64;
    for (int64_t q = 0; q < N; q++) {
        const double *th = theta + q * p->nth;
        int32_t it = 0;
        const int ef = avi_solve(w, p, s, th, warm ? warm + q * nwords : NULL, &it);
        avi_outputs(w, p, th, x + q * p->nout, active ? active + q * nwords : NULL, nwords);
        exitflag[q] = ef;
        if (iters) iters[q] = it;
    }
    awork_free(w);
}

/* Closed loop, as oracle_simulate (daqp_ldp_oracle.c) with the AVI solve: theta = [x; r; uprev], u = first nu outputs,
 * x <- F x + G u (/root/reference/src/simulation.jl:93-113); warm != 0 starts each solve from the previous step's final
 * working set.  flag_min = smallest exit flag over the steps. */
void oracle_avi_simulate(const oracle_avi *p, const avi_settings *s, int64_t N, int32_t T, int32_t nx, int32_t nr,
                         int32_t nuprev, const double *F, const double *G, double *x, const double *r, double *uprev,
                         double *U_traj, double *X_traj, int32_t *flag_min, int32_t warm) {
    int nsoft = 0;
    for (int j = 0; j < p->m; j++) nsoft += (p->sense[j] & SENSE_SOFT) ? 1 : 0;
    awork *w = awork_new(p->n, p->m, p->n + 1 + nsoft);
    const int nwords = (2 * p->m + 63) / 64, nu = p->nout, nth = p->nth;
    double *th = (double *)calloc(nth > 0 ? nth : 1, sizeof(double));
    double *xn = (double *)calloc(nx > 0 ? nx : 1, sizeof(double));
    double *uo = (double *)calloc(nu, sizeof(double));
    uint64_t *act = (uint64_t *)calloc(nwords, sizeof(uint64_t));
    for (int64_t q = 0; q < N; q++) {
        double *xq = x + q * nx;
        int fmin = 0;
        for (int k = 0; k < T; k++) {
            for (int a = 0; a < nx; a++) th[a] = xq[a];
            for (int a = 0; a < nr; a++) th[nx + a] = r ? r[q * nr + a] : 0.0;
            for (int a = 0; a < nuprev; a++) th[nx + nr + a] = uprev[q * nuprev + a];
            if (X_traj) for (int a = 0; a < nx; a++) X_traj[((int64_t)k * N + q) * nx + a] = xq[a];
            const int ef = avi_solve(w, p, s, th, (warm && k > 0) ? act : NULL, NULL);
            avi_outputs(w, p, th, uo, act, nwords);
            fmin = (k == 0 || ef < fmin) ? ef : fmin;
            for (int a = 0; a < nx; a++) {
                double acc = 0.0;
                for (int c = 0; c < nx; c++) acc = fma(F[a * nx + c], xq[c], acc);
                for (int l = 0; l < nu; l++) acc = fma(G[a * nu + l], uo[l], acc);
                xn[a] = acc;
            }
            for (int a = 0; a < nx; a++) xq[a] = xn[a];
            for (int l = 0; l < nuprev && l < nu; l++) uprev[q * nuprev + l] = uo[l];
            if (U_traj) for (int l = 0; l < nu; l++) U_traj[((int64_t)k * N + q) * nu + l] = uo[l];
        }
        if (X_traj) for (int a = 0; a < nx; a++) X_traj[((int64_t)T * N + q) * nx + a] = xq[a];
        if (flag_min) flag_min[q] = fmin;
    }
    free(th); free(xn); free(uo); free(act);
    awork_free(w);
}

/* Proximal-point iterations for a merely positive SEMIDEFINITE H -- DAQP's eps_prox setting (settings named at
 * /root/reference/docs/src/manual/solver.md:46 "full documentation of all DAQP settings"; without it DAQP.setup answers
 * -5, /root/reference/src/setup.jl:18-19).  libdaqp's own prox loop is not available to restate; this is the textbook
 * method it is named after:  x_{k+1} = argmin 1/2 x'Hx + f'x + eps/2 |x - x_k|^2  over the constraint set, x_0 = 0,
 * every subproblem strictly convex with Hessian H + eps I (its pack: p->ML/MR/G/du0/...), solved by avi_solve from the
 * previous subproblem's final working set (mask start), until |x_{k+1} - x_k|_inf < eta_prox.  The iteration counts
 * of the subproblems add up against iter_limit.  Any limit point satisfies the KKT conditions of the ORIGINAL problem
 * (stationarity residual eps |x_{k+1} - x_k|), which is what the tests certify.
 *   subproblem k:  linear term f(theta) - eps x_k  =>  x_unc = x_unc0 + eps (H + eps I)^-1 x_k,
 *                  bounds d = d0 - eps MR x_k  (MR_j = s_j ((H + eps I)^-1 a_j')', symmetric Hessian)          */
void oracle_avi_prox_solve_batch(const oracle_avi *p, const avi_settings *s, double eps, double eta, int64_t N,
                                 const double *theta, double *x, int32_t *exitflag, int32_t *iters, uint64_t *active) {
    int nsoft = 0;
    for (int j = 0; j < p->m; j++) nsoft += (p->sense[j] & SENSE_SOFT) ? 1 : 0;
    awork *w = awork_new(p->n, p->m, p->n + 1 + nsoft);
    const int n = p->n, nth = p->nth, nwords = (2 * p->m + 63) / 64;
    double *xk = (double *)calloc(n, sizeof(double)), *xn = (double *)calloc(n, sizeof(double));
    uint64_t *act = (uint64_t *)calloc(nwords, sizeof(uint64_t));
    for (int64_t q = 0; q < N; q++) {
        const double *th = theta + q * nth;
        int total = 0, flag = EXIT_ITERLIMIT, outer = 0;
        for (int k = 0; k < n; k++) xk[k] = 0.0;
        for (;;) {
            int32_t it = 0;
            avi_settings si = *s;
            si.iter_limit = s->iter_limit - total;               /* what is left of the budget (the solve counts from 1) */
            const int ef = avi_solve_shift(w, p, &si, th, outer > 0 ? act : NULL, &it, xk, eps);
            total += it;
            if (ef < 1) { flag = ef; break; }
            for (int k2 = 0; k2 < nwords; k2++) act[k2] = 0;
            for (int i = 0; i < w->na; i++) {
                const int j = w->WS[i];
                const int bit = (w->sense[j] & SENSE_LOWER) ? p->m + j : j;
                act[bit >> 6] |= (uint64_t)1 << (bit & 63);
            }
            double diff = 0.0;
            for (int k = 0; k < n; k++) {
                double a = p->x0f[k], b = 0.0;
                for (int t = 0; t < nth; t++) a = fma(p->Xthf[(size_t)k * nth + t], th[t], a);
                for (int c = 0; c < n; c++) b = fma(p->Hinv[(size_t)k * n + c], xk[c], b);
                xn[k] = (w->u[k] + a) + eps * b;
                const double d = fabs(xn[k] - xk[k]);
                if (d > diff) diff = d;
            }
            for (int k = 0; k < n; k++) xk[k] = xn[k];
            outer++;
            if (diff < eta) { flag = ef; break; }
            if (total >= s->iter_limit) { flag = EXIT_ITERLIMIT; break; }
        }
        for (int k = 0; k < p->nout; k++) {
            double sh = 0.0;
            for (int t = 0; t < nth; t++) sh = fma(p->Kth[(size_t)k * nth + t], th[t], sh);
            x[q * p->nout + k] = xk[k] + sh;
        }
        if (active) for (int k2 = 0; k2 < nwords; k2++) active[q * nwords + k2] = (flag >= 1) ? act[k2] : 0;
        exitflag[q] = flag;
        if (iters) iters[q] = total;
    }
    free(xk); free(xn); free(act);
    awork_free(w);
}
