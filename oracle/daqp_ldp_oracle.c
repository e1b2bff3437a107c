/*
 * TEST INFRASTRUCTURE ONLY -- CPU parity oracle for the batched condensed-MPC QP path.
 *
 * Nothing in the shipped package may include, link or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it (as the checker /
 * the reported CPU baseline, never as the product).
 *
 * What it restates
 * ----------------
 * The reference's online path is compute_control -> solve -> DAQP.update + DAQP.solve
 * (/root/reference/src/utils.jl:43-51, :268-283) and, in its generated-C form,
 * mpc_compute_control (/root/reference/codegen/mpc_update_qp.c:29-54):
 *
 *     dupper/dlower = du/dl + Dth*theta            mpc_update_qp.c:1-10
 *     cold start (empty working set)               mpc_update_qp.c:44-47
 *     daqp_ldp                                     [EXT] libdaqp, called at mpc_update_qp.c:48
 *     control = R^-1 u* + u_offset + Uth_offset*theta   mpc_update_qp.c:14-22
 *
 * The solver itself (libdaqp, DAQPBase ~0.4.4, /root/reference/Project.toml:9,30) is a
 * third-party dependency that is NOT vendored under /root/reference and is not installed in
 * the build image.  daqp_ldp below is therefore a restatement of the PUBLISHED algorithm:
 * Arnstrom, Bemporad, Axehill, "A Dual Active-Set Solver for Embedded Quadratic Programming
 * Using Recursive LDL' Updates", IEEE TAC 2022 (cited at /root/reference/README.md:74-83):
 * dual active-set iterations on the least-distance problem
 *
 *     min 1/2 |u|^2   s.t.  dlower <= M u <= dupper
 *
 * with a recursively updated LDL' factorisation of M_W M_W', most-violated-constraint
 * selection with primal_tol, dual ratio test with dual_tol, a singular-direction branch when
 * the new row is linearly dependent (D < zero_tol), and progress/cycle/iteration guards.
 * Tolerances follow /root/reference/docs/src/manual/solver.md:49-56.  libdaqp's pivoting
 * heuristic (reordering of the last two working-set entries for conditioning) is not part of
 * the published algorithm and is not restated.
 *
 * Parity pinning: the reference's own known answers that any correct strictly-convex QP
 * solver must reproduce (K1 u = 1.7612519326 at
 * /root/reference/test/runtests.jl:62-66, K4 :1306-1318) are checked in
 * tests/test_oracle.py; iteration counts and working sets of libdaqp itself are not pinned
 * by any reference test ("parity vs DAQP internals: unpinned").
 *
 * Arithmetic contract (shared with the HIP kernels so that results are bit-comparable):
 * IEEE binary64 (binary32 in the _f32 build), every multiply-add written as an explicit fma(), sums accumulated in
 * index order, correctly rounded division, pivots applied through their stored reciprocal
 * 1/D_i (one division per new pivot instead of one per use); build with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* The file is type-generic: built as is it is the binary64 oracle; daqp_ldp_oracle_f32.c includes it
 * with ORACLE_F32 defined, which gives the binary32 twin (every exported name gets the suffix _f32)
 * that checks the kernels' single-precision path -- the reference's own single-precision build of
 * this path is /root/reference/src/codegen.jl:19,31-37,82 (float_type = "float": c_float = float,
 * DAQP_SINGLE_PRECISION). */
#ifdef ORACLE_F32
typedef float real;
#define RFMA fmaf
#define ORACLE_NAME(x) x##_f32
#else
typedef double real;
#define RFMA fma
#define ORACLE_NAME(x) x
#endif
#define oracle_active_words ORACLE_NAME(oracle_active_words)
#define oracle_default_settings ORACLE_NAME(oracle_default_settings)
#define oracle_solve_batch ORACLE_NAME(oracle_solve_batch)
#define oracle_simulate ORACLE_NAME(oracle_simulate)

#define SENSE_ACTIVE 1
#define SENSE_LOWER 2
#define SENSE_IMMUTABLE 4
#define SENSE_SOFT 8
#define SENSE_BINARY 16

#define EXIT_SOFT_OPTIMAL 2
#define EXIT_OPTIMAL 1
#define EXIT_INFEASIBLE (-1)
#define EXIT_CYCLE (-2)
#define EXIT_UNBOUNDED (-3)
#define EXIT_ITERLIMIT (-4)
#define EXIT_NONCONVEX (-5)
#define EXIT_OVERDETERMINED_INITIAL (-6)

#define TRI(i) (((i) * ((i) + 1)) / 2)

typedef struct {
    int32_t n, m, ms, nth, nout;
    const real *M;     /* m x n   row-major, rows normalised                 */
    const real *du0;   /* m                                                   */
    const real *dl0;   /* m                                                   */
    const real *Dth;   /* m x nth row-major                                   */
    const real *Rout;  /* nout x n row-major: first rows of R^-1              */
    const real *x0;    /* nout                                                */
    const real *Xth;   /* nout x nth row-major                                */
    const int32_t *sense;/* m                                                   */
} oracle_ldp;

/* as handed over by the caller: always binary64 (same struct for both builds) */
typedef struct {
    double primal_tol, dual_tol, zero_tol, progress_tol, fval_bound, rho_soft;
    int32_t cycle_tol, iter_limit;
    int32_t mode, pad_;    /* 0: the n-chain form above; 1: Gram-scan form (ORACLE_MODE_GRAM, see below) */
} oracle_settings_abi;

/* what the solver compares against: the caller's tolerances rounded to the working precision */
typedef struct {
    real primal_tol, dual_tol, zero_tol, progress_tol, fval_bound, rho_soft;
    int32_t cycle_tol, iter_limit;
    int32_t mode;
    int32_t cap_check;     /* internal: the n-chain form also stops (EXIT_WSCAP) where a working set would outgrow 64 rows */
} oracle_settings;

static oracle_settings settings_from_abi(const oracle_settings_abi *a) {
    oracle_settings s;
    s.primal_tol = (real)a->primal_tol; s.dual_tol = (real)a->dual_tol; s.zero_tol = (real)a->zero_tol;
    s.progress_tol = (real)a->progress_tol; s.fval_bound = (real)a->fval_bound; s.rho_soft = (real)a->rho_soft;
    s.cycle_tol = a->cycle_tol; s.iter_limit = a->iter_limit;
    s.mode = a->mode;
    s.cap_check = 0;
    return s;
}

typedef struct {
    int n, m, cap, na, sing, reuse, nsoft_act;
    real *L, *D, *Dinv, *lam, *lam_star, *xl, *zl, *u, *dupper, *dlower, *w;
    int *WS;
    int32_t *sense;
    real fval, soft_slack;
    /* Gram-scan form only */
    int cap64, ydirty;
    real *Gf, *Mu;
} work_t;

static work_t *work_new(int n, int m, int nsoft) {
    work_t *w = (work_t *)calloc(1, sizeof(work_t));
    int cap = n + 1 + nsoft;                /* hard rows <= n (+1 singular), soft rows extra */
    w->n = n; w->m = m; w->cap = cap;
    w->L = (real *)calloc((size_t)TRI(cap + 1), sizeof(real));
    w->D = (real *)calloc(cap + 1, sizeof(real));
    w->Dinv = (real *)calloc(cap + 1, sizeof(real));
    w->lam = (real *)calloc(cap + 1, sizeof(real));
    w->lam_star = (real *)calloc(cap + 1, sizeof(real));
    w->xl = (real *)calloc(cap + 1, sizeof(real));
    w->zl = (real *)calloc(cap + 1, sizeof(real));
    w->w = (real *)calloc(cap + 1, sizeof(real));
    w->u = (real *)calloc(n > 0 ? n : 1, sizeof(real));
    w->dupper = (real *)calloc(m > 0 ? m : 1, sizeof(real));
    w->dlower = (real *)calloc(m > 0 ? m : 1, sizeof(real));
    w->WS = (int *)calloc(cap + 1, sizeof(int));
    w->sense = (int32_t *)calloc(m > 0 ? m : 1, sizeof(int32_t));
    w->cap64 = cap < 64 ? cap : 64;
    return w;
}

static void work_free(work_t *w) {
    free(w->L); free(w->D); free(w->Dinv); free(w->lam); free(w->lam_star); free(w->xl); free(w->zl);
    free(w->w); free(w->u); free(w->dupper); free(w->dlower); free(w->WS); free(w->sense);
    free(w->Gf); free(w->Mu);
    free(w);
}

/* Append constraint j to the working set: new row of L, new pivot of D (paper sec. IV-A). */
static void ldl_add(work_t *w, const oracle_ldp *p, const oracle_settings *s, int j) {
    const int n = w->n, na = w->na;
    real *row = &w->L[TRI(na)];
    const real *mj = &p->M[(size_t)j * n];
    real dnew = 0.0;
    for (int i = 0; i < na; i++) {
        const real *mi = &p->M[(size_t)w->WS[i] * n];
        real acc = 0.0;
        for (int k = 0; k < n; k++) acc = RFMA(mi[k], mj[k], acc);
        row[i] = acc;
    }
    for (int k = 0; k < n; k++) dnew = RFMA(mj[k], mj[k], dnew);
    /* soft rows: slack measured in the normalised row's units, weight 1/rho_soft -- the
     * convention /root/reference/src/utils.jl:329-364 (make_singlesided) spells out */
    if (w->sense[j] & SENSE_SOFT) dnew += s->rho_soft;
    for (int i = 0; i < na; i++) {          /* q = L \ (M_W m_j) */
        real acc = row[i];
        const real *li = &w->L[TRI(i)];
        for (int t = 0; t < i; t++) acc = RFMA(-li[t], row[t], acc);
        row[i] = acc;
    }
    for (int i = 0; i < na; i++) {          /* l = D \ q ; d_new = m_j'm_j - sum l_i q_i */
        real q = row[i];
        real l = q * w->Dinv[i];
        row[i] = l;
        dnew = RFMA(-l, q, dnew);
    }
    row[na] = 1.0;
    const int is_soft = (w->sense[j] & SENSE_SOFT) != 0;
    /* n+1 hard rows in R^n are always dependent, whatever rounding says */
    if (dnew < s->zero_tol || (!is_soft && na - w->nsoft_act >= n)) {
        w->D[na] = 0.0;
        w->Dinv[na] = 0.0;
        w->sing = na;
    } else {
        w->D[na] = dnew;
        w->Dinv[na] = 1.0 / dnew;
    }
    w->WS[na] = j;
    w->lam[na] = 0.0;
    w->lam_star[na] = 0.0;
    w->sense[j] |= SENSE_ACTIVE;
    w->nsoft_act += is_soft;
    w->na = na + 1;
}

/* Drop working-set position r: compact L and apply the rank-one update
 * L~ D~ L~' = L D L' + D_r w w' to the trailing block (paper sec. IV-B). */
static void ldl_remove(work_t *w, const oracle_settings *s, int r) {
    const int na = w->na, nup = na - r - 1;
    real *wv = w->w;
    real alpha = w->D[r];
    for (int t = 0; t < nup; t++) wv[t] = w->L[TRI(r + 1 + t) + r];
    for (int i = r; i < na - 1; i++) {      /* new row i = old row i+1 without column r */
        real *dst = &w->L[TRI(i)];
        const real *src = &w->L[TRI(i + 1)];
        for (int c = 0; c < r; c++) dst[c] = src[c];
        for (int c = r; c < i; c++) dst[c] = src[c + 1];
        dst[i] = 1.0;
    }
    w->sing = -1;
    for (int t = 0; t < nup; t++) {
        const int i = r + t;
        const real pt = wv[t];
        const real dold = w->D[i + 1];
        const real dbar = RFMA(alpha * pt, pt, dold);
        if (dbar < s->zero_tol) {
            /* only the last pivot can vanish in exact arithmetic; keep the tail shifted */
            w->D[i] = 0.0;
            w->Dinv[i] = 0.0;
            w->sing = i;
            for (int q = i + 1; q < na - 1; q++) { w->D[q] = w->D[q + 1]; w->Dinv[q] = w->Dinv[q + 1]; }
            break;
        }
        const real rinv = 1.0 / dbar;
        const real beta = (pt * alpha) * rinv;
        alpha = (dold * alpha) * rinv;
        w->D[i] = dbar;
        w->Dinv[i] = rinv;
        for (int q = t + 1; q < nup; q++) {
            real *lqi = &w->L[TRI(r + q) + i];
            wv[q] = RFMA(-pt, *lqi, wv[q]);
            *lqi = RFMA(beta, wv[q], *lqi);
        }
    }
    if (w->sense[w->WS[r]] & SENSE_SOFT) w->nsoft_act--;
    w->sense[w->WS[r]] &= ~(SENSE_ACTIVE | SENSE_LOWER);
    for (int i = r; i < na - 1; i++) {
        w->WS[i] = w->WS[i + 1];
        w->lam[i] = w->lam[i + 1];
    }
    w->na = na - 1;
    if (r < w->reuse) w->reuse = r;
    w->ydirty = 1;
}

/* Constrained stationary point: solve (L D L') lam* = -d_W. */
static void compute_csp(work_t *w) {
    const int na = w->na;
    for (int i = w->reuse; i < na; i++) {
        const int j = w->WS[i];
        real acc = (w->sense[j] & SENSE_LOWER) ? -w->dlower[j] : -w->dupper[j];
        const real *li = &w->L[TRI(i)];
        for (int t = 0; t < i; t++) acc = RFMA(-li[t], w->xl[t], acc);
        w->xl[i] = acc;
    }
    for (int i = w->reuse; i < na; i++) w->zl[i] = w->xl[i] * w->Dinv[i];
    for (int i = na - 1; i >= 0; i--) {
        real acc = w->zl[i];
        for (int t = na - 1; t > i; t--) acc = RFMA(-w->L[TRI(t) + i], w->lam_star[t], acc);
        w->lam_star[i] = acc;
    }
    w->reuse = na;
}

/* Direction p with M_W' p = 0 and p_sing = +-1, stored in lam_star. */
static void singular_direction(work_t *w) {
    const int sg = w->sing;
    const real *ls = &w->L[TRI(sg)];
    for (int i = sg - 1; i >= 0; i--) {
        real acc = -ls[i];
        for (int t = sg - 1; t > i; t--) acc = RFMA(-w->L[TRI(t) + i], w->lam_star[t], acc);
        w->lam_star[i] = acc;
    }
    w->lam_star[sg] = 1.0;
    if (w->sense[w->WS[sg]] & SENSE_LOWER)
        for (int i = 0; i <= sg; i++) w->lam_star[i] = -w->lam_star[i];
}

static void primal_and_fval(work_t *w, const oracle_ldp *p, const oracle_settings *s) {
    const int n = w->n;
    real soft = 0.0;
    for (int k = 0; k < n; k++) w->u[k] = 0.0;
    for (int i = 0; i < w->na; i++) {
        const int j = w->WS[i];
        const real *mi = &p->M[(size_t)j * n];
        const real l = w->lam_star[i];
        for (int k = 0; k < n; k++) w->u[k] = RFMA(-mi[k], l, w->u[k]);
        if (w->sense[j] & SENSE_SOFT) soft = RFMA(l * l, s->rho_soft, soft);
    }
    real fv = 0.0;
    for (int k = 0; k < n; k++) fv = RFMA(w->u[k], w->u[k], fv);
    w->soft_slack = soft;
    w->fval = fv + soft;
}


/* ======================================================================= Gram-scan form (mode 1)
 * The twin of the wavefront kernel's GRAM instantiations (lmpc_set_option "gram_scan"): the same
 * algorithm taking the same decisions, with four sums formed differently so that no n-step chain is
 * left inside an iteration:
 *   - row values for the constraint scan from Gram columns,  M_j u = -sum_{i in W} G(j, W_i) lam*_i
 *     (|W| terms in working-set order instead of n terms; u itself is formed once, at the exit);
 *   - dual objective from the factorisation,  u'u + rho sum_soft lam*_i^2 = lam*' K lam* = sum_i y_i z_i
 *     with y = L^-1 rhs, z = D^-1 y  (a 64-leaf pairwise tree, see tree64);
 *   - the two dot products of a row append (new pivot, new entry of y) and the soft slack as pairwise
 *     trees over the working-set positions instead of serial chains.
 * Everything else (triangular sweeps, rank-one removal, ratio tests, selection, guards) is the code
 * above.  Results differ from mode 0 in the last bits only; a working set that would outgrow 64 rows
 * is re-solved in mode 0 from scratch (the kernel hands such a point to its slow path, which is mode 0).
 */
#define ORACLE_MODE_GRAM 1
#define EXIT_WSCAP (-7)

/* pairwise sum of 64 leaves in the order of the wavefront reduction: neighbours, pairs of pairs, ... up
 * to the four 16-lane rows, then (row0 + row1) + (row2 + row3) */
static real tree64(const real *p) {
    real a[32], b[16], c[8], d[4];
    for (int i = 0; i < 32; i++) a[i] = p[2 * i] + p[2 * i + 1];
    for (int i = 0; i < 16; i++) b[i] = a[2 * i] + a[2 * i + 1];
    for (int i = 0; i < 8; i++) c[i] = b[2 * i] + b[2 * i + 1];
    for (int i = 0; i < 4; i++) d[i] = c[2 * i] + c[2 * i + 1];
    return (d[0] + d[1]) + (d[2] + d[3]);
}

static void gram_prepare(work_t *w, const oracle_ldp *p) {
    const int n = p->n, m = p->m;
    if (w->Gf) return;
    w->Gf = (real *)calloc((size_t)m * m + 1, sizeof(real));
    w->Mu = (real *)calloc(m > 0 ? m : 1, sizeof(real));
    for (int a = 0; a < m; a++)
        for (int b = 0; b <= a; b++) {
            real acc = 0.0;
            for (int k = 0; k < n; k++) acc = RFMA(p->M[(size_t)a * n + k], p->M[(size_t)b * n + k], acc);
            w->Gf[(size_t)a * m + b] = acc;
            w->Gf[(size_t)b * m + a] = acc;
        }
}

static void ldl_add_gram(work_t *w, const oracle_settings *s, int j) {
    const int na = w->na, m = w->m;
    real *row = &w->L[TRI(na)];
    const real *gj = &w->Gf[(size_t)j * m];
    real pr[64], py[64];
    for (int i = 0; i < na; i++) row[i] = gj[w->WS[i]];
    real dnew = gj[j];
    if (w->sense[j] & SENSE_SOFT) dnew += s->rho_soft;
    for (int i = 0; i < na; i++) {          /* q = L \ g */
        real acc = row[i];
        const real *li = &w->L[TRI(i)];
        for (int t = 0; t < i; t++) acc = RFMA(-li[t], row[t], acc);
        row[i] = acc;
    }
    for (int i = 0; i < 64; i++) { pr[i] = 0.0; py[i] = 0.0; }
    for (int i = 0; i < na; i++) {
        const real q = row[i], l = q * w->Dinv[i];
        row[i] = l;
        pr[i] = l * q;
        py[i] = l * w->xl[i];
    }
    dnew = dnew - tree64(pr);
    const real rj = (w->sense[j] & SENSE_LOWER) ? -w->dlower[j] : -w->dupper[j];
    w->xl[na] = rj - tree64(py);            /* new entry of y = L^-1 rhs (meaningless while ydirty: redone then) */
    row[na] = 1.0;
    const int is_soft = (w->sense[j] & SENSE_SOFT) != 0;
    if (dnew < s->zero_tol || (!is_soft && na - w->nsoft_act >= w->n)) {
        w->D[na] = 0.0; w->Dinv[na] = 0.0; w->sing = na;
    } else {
        w->D[na] = dnew; w->Dinv[na] = 1.0 / dnew;
    }
    w->WS[na] = j;
    w->lam[na] = 0.0;
    w->lam_star[na] = 0.0;
    w->sense[j] |= SENSE_ACTIVE;
    w->nsoft_act += is_soft;
    w->na = na + 1;
}

static void compute_csp_gram(work_t *w) {
    const int na = w->na;
    real pf[64];
    if (w->ydirty) {                        /* a row left since the last stationary point: y from scratch */
        for (int i = 0; i < na; i++) {
            const int j = w->WS[i];
            real acc = (w->sense[j] & SENSE_LOWER) ? -w->dlower[j] : -w->dupper[j];
            const real *li = &w->L[TRI(i)];
            for (int t = 0; t < i; t++) acc = RFMA(-li[t], w->xl[t], acc);
            w->xl[i] = acc;
        }
        w->ydirty = 0;
    }
    for (int i = 0; i < 64; i++) pf[i] = 0.0;
    for (int i = 0; i < na; i++) { w->zl[i] = w->xl[i] * w->Dinv[i]; pf[i] = w->xl[i] * w->zl[i]; }
    w->fval = tree64(pf);
    for (int i = na - 1; i >= 0; i--) {
        real acc = w->zl[i];
        for (int t = na - 1; t > i; t--) acc = RFMA(-w->L[TRI(t) + i], w->lam_star[t], acc);
        w->lam_star[i] = acc;
    }
}

/* row values of every constraint at the stationary point, from Gram columns */
static void scan_gram(work_t *w) {
    const int m = w->m;
    for (int j = 0; j < m; j++) w->Mu[j] = 0.0;
    for (int i = 0; i < w->na; i++) {
        const real *gi = &w->Gf[(size_t)w->WS[i] * m];
        const real l = w->lam_star[i];
        for (int j = 0; j < m; j++) w->Mu[j] = RFMA(-gi[j], l, w->Mu[j]);
    }
}

static void soft_slack_gram(work_t *w, const oracle_settings *s) {
    real ps[64];
    w->soft_slack = 0.0;
    if (w->nsoft_act <= 0) return;
    for (int i = 0; i < 64; i++) ps[i] = 0.0;
    for (int i = 0; i < w->na; i++)
        if (w->sense[w->WS[i]] & SENSE_SOFT) ps[i] = (w->lam_star[i] * w->lam_star[i]) * s->rho_soft;
    w->soft_slack = tree64(ps);
}

/* u = -M_W' lam* over the working set as it stands (formed once, when the solve ends) */
static void primal_gram(work_t *w, const oracle_ldp *p) {
    const int n = w->n;
    for (int k = 0; k < n; k++) w->u[k] = 0.0;
    for (int i = 0; i < w->na; i++) {
        const real *mi = &p->M[(size_t)w->WS[i] * n];
        const real l = w->lam_star[i];
        for (int k = 0; k < n; k++) w->u[k] = RFMA(-mi[k], l, w->u[k]);
    }
}

static void write_outputs(const work_t *w, const oracle_ldp *p, const real *theta,
                          real *xout, uint64_t *active, int nwords) {
    const int n = p->n, nth = p->nth;
    for (int k = 0; k < p->nout; k++) {
        real xs = 0.0, sh = p->x0[k];
        for (int c = 0; c < n; c++) xs = RFMA(p->Rout[(size_t)k * n + c], w->u[c], xs);
        for (int t = 0; t < nth; t++) sh = RFMA(p->Xth[(size_t)k * nth + t], theta[t], sh);
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

/* dupper/dlower = du/dl + Dth*theta   (mpc_update_qp.c:1-10) */
static void shift_bounds(work_t *w, const oracle_ldp *p, const real *theta) {
    const int m = p->m, nth = p->nth;
    for (int j = 0; j < m; j++) {
        real sh = 0.0;
        for (int t = 0; t < nth; t++) sh = RFMA(p->Dth[(size_t)j * nth + t], theta[t], sh);
        w->dupper[j] = p->du0[j] + sh;
        w->dlower[j] = p->dl0[j] + sh;
    }
}

/* One LDP solve on the bounds already in w.  sense0[m] are the constraint flags of THIS solve: rows
 * flagged ACTIVE start in the working set (at their lower bound if also flagged LOWER) --
 * equalities, and the binaries a branch-and-bound node has fixed.
 *   forced < 0 : fresh start.  The initial working set is the rows flagged ACTIVE plus the rows of
 *     `warm` (NULL = cold).  two_pass == 0: one pass in row order (the warm start of a closed loop,
 *     /root/reference/codegen/mpc_update_qp.c:44-47).  two_pass != 0 (a B&B node restarted from its
 *     parent's working set): the flagged rows first, the warm rows after them, so that a warm row
 *     that turns out dependent is dropped instead of making a fixed row look over-determined.
 *   forced >= 0: continue IN PLACE from the optimal working set left in w by the previous call (the
 *     parent node): row forced>>1 has just been fixed (ACTIVE|IMMUTABLE, lower side if forced&1) and
 *     enters the working set as the first step, exactly as a violated row would -- the dual iterate
 *     of the parent stays feasible for the child, which is what makes a dual active-set method cheap
 *     inside branch and bound (the reference gets this from daqp_bnb, [EXT]).
 *   forced == -2: continue from the working set AND factorisation left in w by the previous solve of the same
 *     problem family (closed loop with a kept factor, oracle_simulate warm == 2): only the bounds have moved.
 *     This is libdaqp's DAQP_WARMSTART as the generated code uses it -- the workspace is not cleared between two
 *     calls, only the cached forward solve is (reuse_ind = 0; /root/reference/codegen/mpc_update_qp.c:44-54).
 *     The multipliers restart at zero (dual feasible for any bounds).
 * Leaves the iterate in w (u, fval, working set) and returns the DAQP-style exit flag. */
static int solve_core(work_t *w, const oracle_ldp *p, const oracle_settings *s, const int32_t *sense0,
                      const uint64_t *warm, int two_pass, int forced, int32_t *iters) {
    const int n = p->n, m = p->m;
    const int gram = s->mode == ORACLE_MODE_GRAM;
    int exitflag = EXIT_ITERLIMIT, iter = 1, cycle = 0;
    real best_fval = -1.0;

    const int capped = gram || s->cap_check;
    if (forced == -2) {
        w->sing = -1; w->reuse = 0; w->fval = 0.0; w->soft_slack = 0.0; w->ydirty = 1;
        for (int k = 0; k < n; k++) w->u[k] = 0.0;
        for (int i = 0; i < w->na; i++) { w->lam[i] = 0.0; w->lam_star[i] = 0.0; }
    } else if (forced < 0) {
        for (int j = 0; j < m; j++) w->sense[j] = sense0[j] & ~SENSE_LOWER;
        w->na = 0; w->sing = -1; w->reuse = 0; w->fval = 0.0; w->soft_slack = 0.0; w->nsoft_act = 0;
        w->ydirty = 0;
        for (int k = 0; k < n; k++) w->u[k] = 0.0;

        /* initial working set: rows flagged ACTIVE and, if warm, the given mask */
        for (int pass = 0; pass < (two_pass ? 2 : 1); pass++) {
            for (int j = 0; j < m; j++) {
                int want = (sense0[j] & SENSE_ACTIVE) != 0, lower = want && (sense0[j] & SENSE_LOWER);
                if (two_pass && pass == 1) { if (want) continue; }          /* flagged rows went in first */
                if (warm && !(sense0[j] & SENSE_IMMUTABLE) && !(two_pass && pass == 0)) {
                    if ((warm[j >> 6] >> (j & 63)) & 1) want = 1;
                    else if ((warm[(m + j) >> 6] >> ((m + j) & 63)) & 1) { want = 1; lower = 1; }
                }
                if (!want) { w->sense[j] &= ~SENSE_ACTIVE; continue; }
                if (lower) w->sense[j] |= SENSE_LOWER;
                if (capped && w->na >= w->cap64) { exitflag = EXIT_WSCAP; goto done; }
                if (gram) ldl_add_gram(w, s, j);
                else ldl_add(w, p, s, j);
                if (w->sing >= 0) {
                    if (sense0[j] & SENSE_IMMUTABLE) { exitflag = EXIT_OVERDETERMINED_INITIAL; goto done; }
                    /* dependent warm-start row: drop it again */
                    w->na--; w->sing = -1;
                    if (w->sense[j] & SENSE_SOFT) w->nsoft_act--;
                    w->sense[j] &= ~(SENSE_ACTIVE | SENSE_LOWER);
                }
            }
        }
    } else {
        const int jf = forced >> 1;
        w->sense[jf] = (sense0[jf] & ~SENSE_LOWER) & ~SENSE_ACTIVE;             /* ldl_add sets ACTIVE */
        if (gram && iter < s->iter_limit && w->na >= w->cap64) { exitflag = EXIT_WSCAP; goto done; }
    }

    for (; iter < s->iter_limit; iter++) {
        if (w->sing < 0) {
            int nblock = 0, rm = -1, add = -1, isupper = 0;
            real alpha = 0.0;
            if (forced >= 0) {
                add = forced >> 1; isupper = !(forced & 1);
                forced = -1;
            } else {
                if (gram) compute_csp_gram(w); else compute_csp(w);
                for (int i = 0; i < w->na; i++) {
                    const int j = w->WS[i];
                    if (w->sense[j] & SENSE_IMMUTABLE) continue;
                    if (w->sense[j] & SENSE_LOWER) { if (w->lam_star[i] < s->dual_tol) continue; }
                    else if (w->lam_star[i] > -s->dual_tol) continue;
                    const real cand = -w->lam[i] / (w->lam_star[i] - w->lam[i]);
                    if (nblock == 0 || cand < alpha) { alpha = cand; rm = i; }
                    nblock++;
                }
                if (nblock == 0) {
                    if (gram) scan_gram(w); else primal_and_fval(w, p, s);
                    if (w->fval > s->fval_bound) { exitflag = EXIT_INFEASIBLE; break; }
                    /* most violated constraint, primal_tol margin */
                    real min_val = -s->primal_tol;
                    int broken = 0;
                    for (int j = 0; j < m; j++) {
                        if (w->sense[j] & SENSE_IMMUTABLE) continue;
                        const real *mj = &p->M[(size_t)j * n];
                        real Mu = 0.0;
                        if (gram) Mu = w->Mu[j];
                        else for (int k = 0; k < n; k++) Mu = RFMA(mj[k], w->u[k], Mu);
                        const real vu = w->dupper[j] - Mu;
                        const real vl = -(w->dlower[j] - Mu);
                        if (w->sense[j] & SENSE_ACTIVE) {
                            /* a hard row of the working set sits ON its bound in exact arithmetic; if the
                             * iterate violates it by more than primal_tol the factorisation has broken
                             * down (typically an infeasible problem with a nearly dependent working set,
                             * multipliers ~1e15): never report that as optimal */
                            if (!(w->sense[j] & SENSE_SOFT) && (vu < -s->primal_tol || vl < -s->primal_tol)) broken = 1;
                            continue;
                        }
                        if (vu < min_val) { add = j; isupper = 1; min_val = vu; }
                        else if (vl < min_val) { add = j; isupper = 0; min_val = vl; }
                    }
                    if (add < 0) {
                        if (gram) soft_slack_gram(w, s);
                        if (broken) exitflag = EXIT_CYCLE;
                        else exitflag = (w->soft_slack > s->primal_tol) ? EXIT_SOFT_OPTIMAL : EXIT_OPTIMAL;
                        break;
                    }
                }
            }
            if (add >= 0) {
                for (int i = 0; i < w->na; i++) w->lam[i] = w->lam_star[i];
                if (!isupper) w->sense[add] |= SENSE_LOWER;
                if (capped && w->na >= w->cap64) { exitflag = EXIT_WSCAP; break; }
                if (gram) ldl_add_gram(w, s, add);
                else ldl_add(w, p, s, add);
                if (w->fval - best_fval < s->progress_tol) {
                    if (++cycle > s->cycle_tol) { exitflag = EXIT_CYCLE; break; }
                } else { best_fval = w->fval; cycle = 0; }
            } else {
                for (int i = 0; i < w->na; i++)
                    w->lam[i] = RFMA(alpha, w->lam_star[i] - w->lam[i], w->lam[i]);
                ldl_remove(w, s, rm);
            }
        } else {
            singular_direction(w);
            if (gram) for (int i = w->sing + 1; i < w->na; i++) w->lam_star[i] = 0.0;
            int nblock = 0, rm = -1;
            real alpha = 0.0;
            for (int i = 0; i < w->na; i++) {
                const int j = w->WS[i];
                if (w->sense[j] & SENSE_IMMUTABLE) continue;
                if (w->sense[j] & SENSE_LOWER) { if (w->lam_star[i] < s->dual_tol) continue; }
                else if (w->lam_star[i] > -s->dual_tol) continue;
                const real cand = -w->lam[i] / w->lam_star[i];
                if (nblock == 0 || cand < alpha) { alpha = cand; rm = i; }
                nblock++;
            }
            if (nblock == 0) { exitflag = EXIT_INFEASIBLE; break; }
            for (int i = 0; i < w->na; i++) w->lam[i] = RFMA(alpha, w->lam_star[i], w->lam[i]);
            ldl_remove(w, s, rm);
        }
    }
done:
    if (gram) primal_gram(w, p);
    if (iters) *iters = iter;
    return exitflag;
}

static int solve_one(work_t *w, const oracle_ldp *p, const oracle_settings *s,
                     const real *theta, const uint64_t *warm, real *xout,
                     int32_t *iters, uint64_t *active, int nwords) {
    shift_bounds(w, p, theta);
    int ef = solve_core(w, p, s, p->sense, warm, 0, -1, iters);
    if (ef == EXIT_WSCAP) {                 /* Gram-scan form only: more than 64 rows wanted -> mode 0 from scratch */
        oracle_settings s0 = *s;
        s0.mode = 0;
        ef = solve_core(w, p, &s0, p->sense, warm, 0, -1, iters);
    }
    write_outputs(w, p, theta, xout, active, nwords);
    return ef;
}

/* One step of a closed loop that keeps the factorisation (oracle_simulate warm == 2; the twin of the wavefront
 * kernel's kept state, lmpc_simulate_device on that path).  *kept != 0: w holds the final working set, L and D of this
 * scenario's previous step -> continue from them (solve_core forced == -2).  Otherwise the mask-based start of
 * solve_one.  A working set that would outgrow 64 rows is re-solved from the mask in the n-chain form without the
 * limit (the kernel hands such a point to its slow path) and nothing is kept after it; nothing is kept after a
 * failed solve either. */
static int solve_one_keep(work_t *w, const oracle_ldp *p, const oracle_settings *s, const real *theta,
                          const uint64_t *warm, int *kept, real *xout, int32_t *iters, uint64_t *active, int nwords) {
    oracle_settings sc = *s;
    sc.cap_check = 1;
    shift_bounds(w, p, theta);
    int ef = solve_core(w, p, &sc, p->sense, *kept ? NULL : warm, 0, *kept ? -2 : -1, iters);
    int fellback = 0;
    if (ef == EXIT_WSCAP) {
        sc.mode = 0; sc.cap_check = 0;
        ef = solve_core(w, p, &sc, p->sense, warm, 0, -1, iters);
        fellback = 1;
    }
    write_outputs(w, p, theta, xout, active, nwords);
    *kept = ef >= 1 && !fellback;
    return ef;
}

/* Depth-first branch and bound over the rows flagged BINARY (each must end up active at its upper
 * or at its lower bound), what the reference gets from daqp_bnb ([EXT] libdaqp, called at
 * /root/reference/codegen/mpc_update_qp.c:40-43 / utils.jl:277-282 when mpQP.has_binaries).
 * libdaqp's source is not available, so the search order is this file's own (the optimum of a
 * strictly convex MIQP does not depend on it):
 *   - a node = a set of binaries fixed to a side; its relaxation is the LDP with those rows as
 *     active immutable rows, with fval_bound = incumbent value, so the dual iterations stop as soon
 *     as the node is dominated;
 *   - branch on the lowest-index binary row that is not in the relaxation's final working set,
 *     first to the bound its row value M_j u is closer to, then to the other;
 *   - BOTH children continue in place from their parent's optimal working set, factorisation and
 *     multipliers (the newly fixed row enters like a violated row would: solve_core's `forced`): the
 *     first child right away, the second -- reached by backtracking, when the first child's subtree has
 *     overwritten that state -- from a SNAPSHOT of it taken when the node branched (one per depth).
 *     (Until round 3 the second child rebuilt the parent's working set row by row from a mask: for the
 *     satellite problem 41 such rebuilds of ~30 rows per parameter point, most of the search's time.)
 *   - a node whose binaries are all active is a leaf; it replaces the incumbent if strictly better.
 * iters returns the iterations summed over all nodes; the flag is 1 if an incumbent exists,
 * -1 if none, -4 if the node limit ran out first. */
#define BNB_NODE_LIMIT 100000
/* solver state of a node (what continuing in place needs): a deep copy between two work areas of equal shape */
static void work_copy(work_t *dst, const work_t *src) {
    const int cap = src->cap, n = src->n, m = src->m;
    dst->na = src->na; dst->sing = src->sing; dst->reuse = src->reuse; dst->nsoft_act = src->nsoft_act;
    dst->fval = src->fval; dst->soft_slack = src->soft_slack; dst->ydirty = src->ydirty;
    memcpy(dst->L, src->L, sizeof(real) * (size_t)TRI(cap + 1));
    memcpy(dst->D, src->D, sizeof(real) * (cap + 1));
    memcpy(dst->Dinv, src->Dinv, sizeof(real) * (cap + 1));
    memcpy(dst->lam, src->lam, sizeof(real) * (cap + 1));
    memcpy(dst->lam_star, src->lam_star, sizeof(real) * (cap + 1));
    memcpy(dst->xl, src->xl, sizeof(real) * (cap + 1));
    memcpy(dst->zl, src->zl, sizeof(real) * (cap + 1));
    memcpy(dst->u, src->u, sizeof(real) * (n > 0 ? n : 1));
    memcpy(dst->WS, src->WS, sizeof(int) * (cap + 1));
    memcpy(dst->sense, src->sense, sizeof(int32_t) * (m > 0 ? m : 1));
    if (src->Mu && dst->Mu) memcpy(dst->Mu, src->Mu, sizeof(real) * (m > 0 ? m : 1));
}

static int solve_bnb(work_t *w, const oracle_ldp *p, const oracle_settings *s, const real *theta,
                     real *xout, int32_t *iters, uint64_t *active, int nwords) {
    const int n = p->n, m = p->m;
    const int nw = nwords > 0 ? nwords : 1;
    int32_t *sense = (int32_t *)malloc(sizeof(int32_t) * (m > 0 ? m : 1));
    int *stk_j = (int *)malloc(sizeof(int) * (m + 1)), *stk_side = (int *)malloc(sizeof(int) * (m + 1)),
        *stk_tried = (int *)malloc(sizeof(int) * (m + 1));
    work_t **snap = (work_t **)calloc((size_t)m + 1, sizeof(work_t *));   /* snap[d]: state of the node that branched at depth d */
    int nsoft_rows = 0;
    for (int j = 0; j < m; j++) nsoft_rows += (p->sense[j] & SENSE_SOFT) != 0;
    real *ubest = (real *)calloc(n, sizeof(real));
    uint64_t *abest = (uint64_t *)calloc(nw, sizeof(uint64_t));
    oracle_settings sn = *s;
    real best = s->fval_bound;
    int have = 0, depth = 0, nodes = 0, total_it = 0, flag = EXIT_INFEASIBLE, inplace = 0;
    /* A search's working sets take one row more than a plain solve's: the row a node has just fixed enters on top of its
     * parent's final working set whatever that holds (n hard rows, the soft rows, the row that made it singular), and the
     * arrays are sized for it (work_new: cap + 1 rows).  The capacity-checked forms (Gram-scan; the kernels in both forms)
     * stop one row later accordingly: binary32 searches on random dense problems do reach n + 2 + #soft rows. */
    w->cap64 = w->cap + 1 < 64 ? w->cap + 1 : 64;
    shift_bounds(w, p, theta);
    for (;;) {
        if (nodes >= BNB_NODE_LIMIT) { flag = EXIT_ITERLIMIT; break; }
        for (int j = 0; j < m; j++) sense[j] = p->sense[j];
        for (int d = 0; d < depth; d++)
            sense[stk_j[d]] |= SENSE_ACTIVE | SENSE_IMMUTABLE | (stk_side[d] ? SENSE_LOWER : 0);
        sn.fval_bound = best;
        int32_t it = 0;
        int ef;
        if (inplace) ef = solve_core(w, p, &sn, sense, NULL, 0, 2 * stk_j[depth - 1] + stk_side[depth - 1], &it);
        else ef = solve_core(w, p, &sn, sense, NULL, 0, -1, &it);                   /* the root */
        nodes++;
        total_it += it;
        if (ef == EXIT_WSCAP) { flag = EXIT_WSCAP; have = 0; break; }     /* Gram-scan form: as the kernel does */
        int descend = 0;
        if (ef >= 1) {
            int jb = -1;
            for (int j = 0; j < m && jb < 0; j++)
                if ((p->sense[j] & SENSE_BINARY) && !(w->sense[j] & SENSE_ACTIVE)) jb = j;
            if (jb < 0) {                               /* leaf */
                if (!have || w->fval < best) {
                    have = 1; best = w->fval;
                    for (int k = 0; k < n; k++) ubest[k] = w->u[k];
                    for (int q = 0; q < nwords; q++) abest[q] = 0;
                    for (int i = 0; i < w->na; i++) {
                        const int j = w->WS[i];
                        const int bit = (w->sense[j] & SENSE_LOWER) ? m + j : j;
                        abest[bit >> 6] |= (uint64_t)1 << (bit & 63);
                    }
                }
            } else {
                const real *mj = &p->M[(size_t)jb * n];
                real Mu = 0.0;
                if (s->mode == ORACLE_MODE_GRAM) Mu = w->Mu[jb];
                else for (int k = 0; k < n; k++) Mu = RFMA(mj[k], w->u[k], Mu);
                const int lower_first = (Mu - w->dlower[jb]) < (w->dupper[jb] - Mu);
                stk_j[depth] = jb; stk_side[depth] = lower_first; stk_tried[depth] = 1;
                if (!snap[depth]) {                                 /* this node's final state, for its second child */
                    snap[depth] = work_new(n, m, nsoft_rows);
                    if (s->mode == ORACLE_MODE_GRAM) snap[depth]->Mu = (real *)calloc(m > 0 ? m : 1, sizeof(real));
                }
                work_copy(snap[depth], w);
                depth++;
                descend = 1;
            }
        }
        inplace = descend;
        if (!descend) {                                 /* backtrack to the next untried side */
            while (depth > 0 && stk_tried[depth - 1] == 2) depth--;
            if (depth == 0) break;
            stk_side[depth - 1] ^= 1;
            stk_tried[depth - 1] = 2;
            work_copy(w, snap[depth - 1]);              /* back to the parent's optimal state: second child, in place */
            inplace = 1;
        }
    }
    if (have) {
        if (flag != EXIT_ITERLIMIT) flag = EXIT_OPTIMAL;
        for (int k = 0; k < n; k++) w->u[k] = ubest[k];
    } else {
        for (int k = 0; k < n; k++) w->u[k] = 0.0;
    }
    w->na = 0;                                          /* write_outputs: mask comes from abest */
    write_outputs(w, p, theta, xout, NULL, 0);
    if (active) for (int q = 0; q < nwords; q++) active[q] = have ? abest[q] : 0;
    if (iters) *iters = total_it;
    for (int d = 0; d <= m; d++) if (snap[d]) work_free(snap[d]);
    (void)nw;
    free(snap); free(sense); free(stk_j); free(stk_side); free(stk_tried); free(ubest); free(abest);
    return flag;
}

/* ---------------------------------------------------------------- exported entry points */
int oracle_active_words(int m) { return (2 * m + 63) / 64; }

void oracle_default_settings(oracle_settings_abi *s) {
#ifdef ORACLE_F32
    s->primal_tol = 1e-4; s->dual_tol = 1e-6; s->zero_tol = 1e-6; s->progress_tol = 1e-4;
    s->fval_bound = 1e30; s->rho_soft = 1e-3; s->cycle_tol = 10; s->iter_limit = 10000;
    s->mode = 0; s->pad_ = 0;
#else
    s->primal_tol = 1e-6; s->dual_tol = 1e-12; s->zero_tol = 1e-11; s->progress_tol = 1e-6;
    s->fval_bound = 1e30; s->rho_soft = 1e-6; s->cycle_tol = 10; s->iter_limit = 10000;
    s->mode = 0; s->pad_ = 0;
#endif
}

/* theta: N rows of nth; X: N rows of nout; active: N rows of nwords (may be NULL);
 * warm: N rows of nwords initial working sets (may be NULL = cold start). */
void oracle_solve_batch(const oracle_ldp *p, const oracle_settings_abi *sabi, int64_t N,
                        const real *theta, const uint64_t *warm, real *X,
                        int32_t *exitflag, int32_t *iters, uint64_t *active) {
    const int nw = oracle_active_words(p->m);
    const oracle_settings sr = settings_from_abi(sabi), *s = &sr;
    int nsoft = 0;
    for (int j = 0; j < p->m; j++) nsoft += (p->sense[j] & SENSE_SOFT) != 0;
    work_t *w = work_new(p->n, p->m, nsoft);
    if (s->mode == ORACLE_MODE_GRAM) gram_prepare(w, p);
    int nbin = 0;
    for (int j = 0; j < p->m; j++) nbin += (p->sense[j] & SENSE_BINARY) != 0;
    for (int64_t i = 0; i < N; i++) {
        int32_t it = 0;
        int ef = nbin ? solve_bnb(w, p, s, theta + i * p->nth, X + i * p->nout, &it,
                                  active ? active + i * nw : NULL, nw)
                      : solve_one(w, p, s, theta + i * p->nth, warm ? warm + i * nw : NULL,
                                  X + i * p->nout, &it, active ? active + i * nw : NULL, nw);
        exitflag[i] = ef;
        if (iters) iters[i] = it;
    }
    work_free(w);
}

/* Timing front end (bench.py's cpu_baseline leg): the same batch `reps` times in one call, so that a
 * many-thread measurement is not paced by the caller's interpreter lock between calls. */
void ORACLE_NAME(oracle_solve_batch_repeat)(const oracle_ldp *p, const oracle_settings_abi *sabi, int64_t N,
                                            const real *theta, real *X, int32_t *exitflag, int32_t *iters,
                                            int32_t reps) {
    for (int32_t r = 0; r < reps; r++) oracle_solve_batch(p, sabi, N, theta, NULL, X, exitflag, iters, NULL);
}

/* Closed loop, one scenario after the other: the reference's Simulation loop without observer
 * (/root/reference/src/simulation.jl:93-113): theta = [x; r; uprev] (src/explicit.jl:54-63),
 * u = compute_control, x <- F x + G u (sums in index order, F then G), uprev <- u.
 * warm != 0 reuses the previous step's final working set (generated-C DAQP_WARMSTART,
 * /root/reference/codegen/mpc_update_qp.c:44-47); the first step is cold: warm == 1 re-appends its rows in
 * index order from the mask, warm == 2 continues from the kept factorisation in working-set order
 * (solve_one_keep).  nout must equal nu. */
void oracle_simulate(const oracle_ldp *p, const oracle_settings_abi *sabi, int64_t N, int32_t T, int32_t nx,
                     int32_t nr, int32_t nup, const real *F, const real *G, real *x,
                     const real *r, real *uprev, real *U, real *X, int32_t *flag_min,
                     int32_t warm) {
    const int nw = oracle_active_words(p->m), nu = p->nout, nth = p->nth;
    const oracle_settings sr = settings_from_abi(sabi), *s = &sr;
    int nsoft = 0, nbin = 0;
    for (int j = 0; j < p->m; j++) nsoft += (p->sense[j] & SENSE_SOFT) != 0;
    for (int j = 0; j < p->m; j++) nbin += (p->sense[j] & SENSE_BINARY) != 0;
    work_t *w = work_new(p->n, p->m, nsoft);
    if (s->mode == ORACLE_MODE_GRAM) gram_prepare(w, p);
    real *th = (real *)calloc(nth > 0 ? nth : 1, sizeof(real));
    real *u = (real *)calloc(nu, sizeof(real));
    real *xn = (real *)calloc(nx, sizeof(real));
    uint64_t *act = (uint64_t *)calloc(nw, sizeof(uint64_t));
    int kept = 0;
    for (int64_t i = 0; i < N; i++) {
        real *xi = x + i * nx;
        if (X) for (int a = 0; a < nx; a++) X[(size_t)i * nx + a] = xi[a];
        for (int k = 0; k < T; k++) {
            for (int a = 0; a < nx; a++) th[a] = xi[a];
            for (int a = 0; a < nr; a++) th[nx + a] = r ? r[i * nr + a] : 0.0;
            for (int a = 0; a < nup; a++) th[nx + nr + a] = uprev ? uprev[i * nup + a] : 0.0;
            int32_t it = 0;
            int ef;
            if (nbin) ef = solve_bnb(w, p, s, th, u, &it, act, nw);      /* B&B nodes start cold */
            else if (warm == 2) {
                if (k == 0) kept = 0;
                ef = solve_one_keep(w, p, s, th, k > 0 ? act : NULL, &kept, u, &it, act, nw);
            } else ef = solve_one(w, p, s, th, (warm && k > 0) ? act : NULL, u, &it, act, nw);
            for (int a = 0; a < nx; a++) {
                real acc = 0.0;
                for (int c = 0; c < nx; c++) acc = RFMA(F[a * nx + c], xi[c], acc);
                for (int l = 0; l < nu; l++) acc = RFMA(G[a * nu + l], u[l], acc);
                xn[a] = acc;
            }
            for (int a = 0; a < nx; a++) {
                xi[a] = xn[a];
                if (X) X[((size_t)(k + 1) * N + i) * nx + a] = xn[a];
            }
            for (int l = 0; l < nu; l++) {
                if (l < nup) uprev[i * nup + l] = u[l];
                if (U) U[((size_t)k * N + i) * nu + l] = u[l];
            }
            if (flag_min) flag_min[i] = (k == 0 || ef < flag_min[i]) ? ef : flag_min[i];
        }
    }
    free(th); free(u); free(xn); free(act);
    work_free(w);
}
