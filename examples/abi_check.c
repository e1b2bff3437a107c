/*
 * Plain-C client of liblmpc_hip.so: what a maintainer's FFI sees (no C++, no Python, no torch).
 * Built and run by tests/test_host.py::test_c_client_links_and_runs with gcc -std=c99 -Wall -Wextra -Werror.
 *
 * It sets up the reference's K1 problem shape in miniature (a 2-variable box-constrained QP with one
 * parameter), runs the host-only transform (reference codegen.jl:239-280 qp2ldp), and then:
 *   - without a GPU: checks that lmpc_setup refuses with LMPC_ERR_NOGPU (there is no CPU fallback);
 *   - with a GPU: solves three parameter points through lmpc_solve_batch and through the generated
 *     controller's entry point lmpc_compute_control, and checks them against the closed-form answer.
 * Exit code 0 = everything as expected; it prints which branch ran.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "lmpc_hip.h"

#define CHECK(cond, msg) do { if (!(cond)) { fprintf(stderr, "abi_check: %s\n", msg); return 1; } } while (0)

int main(void) {
    /* min 1/2 U'HU + (f_theta*theta)'U,  -1 <= U <= 1;  H = diag(2, 4), f_theta = [-2; -4]:
     * unconstrained optimum U = [theta; theta], clipped to the box (H diagonal: exact) */
    const double H[4] = {2.0, 0.0, 0.0, 4.0};          /* column-major, as Julia stores mpQP.H */
    const double f[2] = {0.0, 0.0};
    const double f_theta[2] = {-2.0, -4.0};
    const double bu[2] = {1.0, 1.0}, bl[2] = {-1.0, -1.0};
    const double W[2] = {0.0, 0.0};
    const int32_t sense[2] = {0, 0};
    const int n = 2, m = 2, ms = 2, nth = 1, nout = 2;

    CHECK(lmpc_abi_version() >= 1, "ABI version");
    lmpc_settings s;
    lmpc_default_settings(&s);
    CHECK(s.primal_tol == 1e-6 && s.iter_limit == 10000, "default settings");

    double M[4], du[2], dl[2], Dth[2], Rout[4], x0[2], Xth[2];
    int rc = lmpc_transform(n, m, ms, nth, nout, H, f, f_theta, NULL, bu, bl, W, sense, NULL, 0,
                            M, du, dl, Dth, Rout, x0, Xth);
    CHECK(rc == LMPC_OK, "lmpc_transform");
    /* rows of [I] R^-1 normalised: M = identity here (H diagonal) */
    CHECK(fabs(M[0] - 1.0) < 1e-15 && fabs(M[3] - 1.0) < 1e-15 && M[1] == 0.0 && M[2] == 0.0, "LDP rows");
    /* the unconstrained optimum as an affine map of theta: x = x0 + Xth*theta = theta */
    CHECK(fabs(Xth[0] - 1.0) < 1e-15 && fabs(Xth[1] - 1.0) < 1e-15 && x0[0] == 0.0, "output map");

    /* the two keywords of the reference's DAQP.setup call (src/setup.jl:11-13) are part of the ABI: host-side answers */
    CHECK(lmpc_abi_version() >= 2 && s.eps_prox == 0.0 && s.eta_prox == 1e-6, "settings v2");
    {
        const double Hn[4] = {2.0, 0.0, 0.5, 2.0};            /* column-major [[2, .5], [0, 2]]: not symmetric */
        const int32_t bp[2] = {1, 2};
        lmpc_handle *hx = NULL;
        CHECK(lmpc_setup_ex(&hx, n, m, ms, nth, nout, H, f, f_theta, NULL, bu, bl, W, sense, NULL, 0, &s, bp, 2, 0, 0)
                  == LMPC_ERR_UNSUPPORTED && hx == NULL, "prioritised constraints are refused");
        CHECK(lmpc_setup_ex(&hx, n, m, ms, nth, nout, Hn, f, f_theta, NULL, bu, bl, W, sense, NULL, 0, &s, NULL, 0, 0, 0)
                  == LMPC_ERR_BADARG, "is_avi = 0 with a non-symmetric H");
        double ML[4], MR[4], G[4];
        CHECK(lmpc_transform_avi(n, m, ms, nth, nout, Hn, f, f_theta, NULL, bu, bl, W, sense, NULL, 0, ML, MR, G,
                                 NULL, NULL, NULL, NULL, NULL, NULL) == LMPC_OK, "lmpc_transform_avi");
        CHECK(fabs(G[0] - 1.0) < 1e-14 && fabs(G[3] - 1.0) < 1e-14 && fabs(G[1] - G[2]) > 1e-3, "Gram matrix of the AVI pack");
    }

    lmpc_handle *h = NULL;
    rc = lmpc_setup(&h, n, m, ms, nth, nout, H, f, f_theta, NULL, bu, bl, W, sense, NULL, 0, &s, 0);
    if (rc == LMPC_ERR_NOGPU) {
        CHECK(h == NULL, "handle must stay NULL on failure");
        CHECK(strlen(lmpc_last_error(NULL)) > 0, "error text");
        printf("abi_check: no HIP device -> lmpc_setup refused with LMPC_ERR_NOGPU (no CPU fallback): ok\n");
        return 0;
    }
    CHECK(rc == LMPC_OK && h != NULL, "lmpc_setup");

    const double theta[3] = {0.25, 3.0, -0.6};
    double x[6];
    int32_t flag[3], iters[3];
    rc = lmpc_solve_batch(h, 3, theta, x, flag, iters, NULL, NULL);
    CHECK(rc == LMPC_OK, "lmpc_solve_batch");
    for (int i = 0; i < 3; i++) {
        const double want = theta[i] > 1.0 ? 1.0 : (theta[i] < -1.0 ? -1.0 : theta[i]);
        CHECK(flag[i] == 1, "exit flag");
        CHECK(fabs(x[2 * i] - want) < 1e-12 && fabs(x[2 * i + 1] - want) < 1e-12, "solution");
    }
    CHECK(iters[0] == 1 && iters[1] > 1, "iteration counts");

    /* the generated controller's call shape: theta = [state(1)], control in/out (2 entries) */
    lmpc_param_layout lay;
    memset(&lay, 0, sizeof lay);
    lay.n_state = 1;
    CHECK(lmpc_set_parameter_layout(h, &lay) == LMPC_OK, "lmpc_set_parameter_layout");
    double control[6] = {0, 0, 0, 0, 0, 0};
    rc = lmpc_compute_control(h, 3, control, theta, NULL, NULL, NULL, flag, 0);
    CHECK(rc == LMPC_OK, "lmpc_compute_control");
    for (int i = 0; i < 6; i++) CHECK(control[i] == x[i], "compute_control == solve_batch");

    lmpc_free(h);
    printf("abi_check: solved through the C ABI on a HIP device: ok\n");
    return 0;
}
