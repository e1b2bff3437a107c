/*
 * lmpc_hip.h -- C ABI of the MI355X (gfx950) batched QP backend for LinearMPC.jl's online path.
 *
 * Drop-in boundary: everything LinearMPC.jl does between "theta is formed" and "x*, exitflag
 * come back from DAQP" (reference /root/reference/src/utils.jl:268-283 `solve`, with the
 * one-time half at /root/reference/src/setup.jl:7-29 `setup!`), for N parameter points at once.
 * The reference-side binding a maintainer would add (Julia `ccall`) is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C, opaque handle, no exceptions; every entry point returns an int status.
 *  - "QP" arrays arrive exactly as Julia stores mpc.mpQP (COLUMN-major Float64, Cint senses),
 *    so the ccall passes the struct fields without copies or transposes.
 *  - batched arrays are "one problem per contiguous record": theta is nth x N column-major on
 *    the Julia side == N records of nth doubles; x is nout x N == N records of nout doubles.
 *  - exit flags per problem use DAQP's sign convention (reference asserts exitflag >= 1,
 *    utils.jl:46):  1 optimal, 2 soft-optimal, -1 infeasible, -2 cycle, -3 unbounded,
 *    -4 iteration limit, -5 non-convex, -6 over-determined initial working set.  One flag is this
 *    library's own: -7 = the working set outgrew what the kernels hold.  The wavefront kernel keeps 64 rows; a
 *    point that wants more is re-solved behind it by a one-problem-per-thread kernel with room for 256, so -7 is
 *    left only beyond 256 rows, for hybrid problems (branch and bound) with n + 1 + #SOFT > 64, and under
 *    lmpc_set_option("big_path", 0).  Like every flag < 1 it fails the reference's assertion.
 *  - sense bit flags are DAQP's (reference mpc2mpqp.jl:868-885): 1 ACTIVE, 2 LOWER,
 *    4 IMMUTABLE, 5 EQUALITY (=ACTIVE|IMMUTABLE), 8 SOFT, 16 BINARY.  A problem with BINARY rows
 *    (hybrid MPC, mpQP.has_binaries) is solved by branch and bound: every BINARY row ends up
 *    active at one of its two bounds; the flag is 1 with the best assignment found, -1 if none is
 *    feasible, -4 if the node limit (100000) was hit; `iters` is the sum over all nodes.
 *  - active-set masks: lmpc_active_words(h) 64-bit words per problem; bit j (0 <= j < m) set
 *    = constraint j active at its UPPER bound, bit m+j set = active at its LOWER bound.
 *  - a handle is bound to one GPU (the `device` given at setup); calls on one handle must not
 *    overlap in time, different handles are independent (the reference's DAQP workspace is
 *    one-per-MPC and not thread-safe either: types.jl:93-97,141).
 */
#ifndef LMPC_HIP_H
#define LMPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lmpc_handle lmpc_handle;

/* Solver settings == the DAQP settings LinearMPC.jl documents
 * (/root/reference/docs/src/manual/solver.md:49-56) plus the two DAQP-internal tolerances the
 * algorithm needs (zero_tol for singular pivots, fval_bound). rho_soft = 1/soft_weight
 * (setup.jl:26).  Replaces DAQP.settings(model, Dict(...)). */
typedef struct lmpc_settings {
    double primal_tol;   /* 1e-6  */
    double dual_tol;     /* 1e-12 */
    double zero_tol;     /* 1e-11 */
    double progress_tol; /* 1e-6  */
    double fval_bound;   /* 1e30  */
    double rho_soft;     /* 1e-6  */
    int32_t cycle_tol;   /* 10    */
    int32_t iter_limit;  /* 10000 */
    /* ABI version 2: DAQP's proximal-point settings.  (PARITY: UNPINNED -- libdaqp's own prox loop is not available to
     * restate and no reference test holds a vector for it; this is the textbook method the setting is named after,
     * checked against this build's own oracle twin and by KKT certificates of the original problem.)
     * eps_prox > 0 AT SETUP selects the proximal-point mode, the one
     * that accepts a merely positive SEMIdefinite (symmetric) H -- without it such an H is answered with -5, as by
     * DAQP.setup (/root/reference/src/setup.jl:18-19):  x_{k+1} = argmin 1/2 x'Hx + f(theta)'x + eps_prox/2 |x - x_k|^2
     * over the constraint set, x_0 = 0, until |x_{k+1} - x_k|_inf < eta_prox; every subproblem is solved by the
     * library's L D U kernel on the Hessian H + eps_prox I, warm from the previous one's working set; iteration counts
     * add up against iter_limit.  The limit is a KKT point of the ORIGINAL problem (not unique if H is singular).
     * Changing eps_prox on a live handle (lmpc_set_settings) is refused: the factorisation depends on it. */
    double eps_prox;     /* 0     */
    double eta_prox;     /* 1e-6  */
} lmpc_settings;
/* The layout every binding mirrors (integration/LmpcHipExt.jl::LmpcSettings, linearmpc.jl_amd/_cabi.py::Settings): ten
 * fields, 72 bytes, the two 32-bit counters side by side at offset 48.  Checked here at compile time and against the
 * Julia struct's text by tests/test_host.py::test_settings_layout_is_the_same_in_every_binding. */
typedef char lmpc_settings_is_72_bytes[(sizeof(lmpc_settings) == 72) ? 1 : -1];

/* status codes of the API itself (solver outcomes are the per-problem exit flags) */
#define LMPC_OK 1
#define LMPC_ERR_INFEASIBLE (-1)      /* setup: bl > bu somewhere (DAQP.setup flag -1)        */
#define LMPC_ERR_NONCONVEX (-5)       /* setup: H not positive definite (DAQP.setup flag -5)  */
#define LMPC_ERR_OVERDETERMINED (-6)  /* setup: equality rows linearly dependent (flag -6)    */
#define LMPC_ERR_BADARG (-100)
#define LMPC_ERR_NOGPU (-101)         /* no usable HIP device: there is NO CPU fallback       */
#define LMPC_ERR_HIP (-102)           /* a HIP runtime call failed, see lmpc_last_error       */
#define LMPC_ERR_UNSUPPORTED (-103)   /* problem shape outside what the kernels cover         */

/* Fills *s with the defaults listed above. */
void lmpc_default_settings(lmpc_settings *s);

/*
 * One-time setup from the mpQP, replacing DAQP.setup + DAQP.settings (setup.jl:11-13,26) and
 * doing what DAQP.update / qp2ldp precompute (utils.jl:272-281, codegen.jl:239-280):
 * Cholesky H = R'R, LDP rows M = [I_ms;A] R^-1 (normalised), Dth = W + M R^-T f_theta,
 * du/dl, the output maps, and the upload of that constant pack to GPU `device`.
 *
 *   n      decision variables (Nc*nu)            m     two-sided constraints (ms simple first)
 *   ms     leading simple bounds  bl<=U[i]<=bu   nth   length of theta = [x; r; d; uprev; p]
 *   nout   leading entries of U* returned per problem (nu for compute_control, n for
 *          compute_control_trajectory)
 *   H[n*n] f[n] f_theta[n*nth] A[(m-ms)*n] bu[m] bl[m] W[m*nth]   column-major, as in mpc.mpQP
 *   sense[m]  Cint flags        Kfb[nout*nx] or NULL  prestabilising feedback mpc.K
 *          (column-major): output k gets  - Kfb[k,:]*theta[0:nx]  (utils.jl:48-49)
 *   s      settings or NULL for defaults          device  HIP device ordinal
 *
 * Returns LMPC_OK (1) or a negative code; *out is NULL on failure.
 */
int lmpc_setup(lmpc_handle **out, int n, int m, int ms, int nth, int nout,
               const double *H, const double *f, const double *f_theta,
               const double *A, const double *bu, const double *bl, const double *W,
               const int32_t *sense, const double *Kfb, int nx,
               const lmpc_settings *s, int device);

/*
 * The same setup with the two keywords the reference passes to DAQP.setup next to the matrices
 * (/root/reference/src/setup.jl:11-13):
 *
 *   break_points[n_break_points]  mpQP.break_points (Cint; /root/reference/src/mpc2mpqp.jl:890-892): the ends of the
 *          constraint priority levels (`prio` of set_bounds! / add_constraint!, setup.jl:50-97).  Empty (NULL, 0) for
 *          every controller without priorities.  A non-empty list asks DAQP for its hierarchical mode, which the
 *          batched backend does not implement: LMPC_ERR_UNSUPPORTED (never silently ignored).
 *   is_avi  !mpQP.is_symmetric: H is NOT symmetric (several objectives, one per player: set_objective!(mpc, uids; ...),
 *          setup.jl:137-151, mpc2mpqp.jl:900-950) and the problem is the affine variational inequality
 *              find x in the constraint set with (H x + f + f_theta theta)'(y - x) >= 0 for every feasible y
 *          (H + H' must be positive definite: LMPC_ERR_NONCONVEX otherwise).  Solved by the library's AVI kernel
 *          (one problem per lane, recursive L D U factorisation; binary64; no BINARY rows); every entry point that
 *          takes a handle works on it except the binary32 ones and the Gram-scan option.  is_avi == 0 with a
 *          non-symmetric H is LMPC_ERR_BADARG.  (PARITY OF THE ITERATION: UNPINNED -- libdaqp's AVI mode is not
 *          restated; the solution of a strongly monotone AVI is unique, which is what the reference-held numbers
 *          -- the game-theoretic closed loop's end values, test/runtests.jl:1337-1358 -- and the KKT certificates
 *          pin; iteration counts and working-set sequences are this build's own.)
 * lmpc_setup itself decides is_avi from H exactly as the reference decides mpQP.is_symmetric
 * (isapprox(H, H', rtol = 1e-9), mpc2mpqp.jl:897) and has no priorities.
 * lmpc_is_avi: 1 / 0.  lmpc_get_avi: the AVI pack's second matrix MR[m*n] (row j = (H^-1 ML_j')', ML = what
 * lmpc_get_ldp returns as M) and the full Gram matrix G[m*m] = ML MR' (row-major; either may be NULL).
 */
int lmpc_setup_ex(lmpc_handle **out, int n, int m, int ms, int nth, int nout,
                  const double *H, const double *f, const double *f_theta,
                  const double *A, const double *bu, const double *bl, const double *W,
                  const int32_t *sense, const double *Kfb, int nx,
                  const lmpc_settings *s, const int32_t *break_points, int n_break_points,
                  int is_avi, int device);
int lmpc_is_avi(const lmpc_handle *h);
int lmpc_get_avi(const lmpc_handle *h, double *MR, double *G);
/* The extra arrays of a handle in proximal-point mode (lmpc_settings.eps_prox > 0 at setup; its subproblems' pack is
 * what lmpc_get_ldp / lmpc_get_avi return, for H + eps_prox I): Hinv[n*n] = (H + eps_prox I)^-1, x0f[n] and
 * Xthf[n*nth] = the full-length affine map of the unconstrained optimum, Kth[nout*nth] = the outputs' feedback term.
 * LMPC_ERR_BADARG on any other handle. */
int lmpc_get_prox(const lmpc_handle *h, double *Hinv, double *x0f, double *Xthf, double *Kth);

/*
 * Setup from an already-transformed least-distance problem -- the data the reference's code
 * generator writes into C arrays (codegen.jl:183-189: Dth, du, dl, Uth_offset, u_offset, plus
 * DAQP's M) -- all ROW-major:
 *   M[m*n] normalised rows, du[m], dl[m], Dth[m*nth], Rout[nout*n] (rows of R^-1),
 *   x0[nout] (= u_offset), Xth[nout*nth] (= Uth_offset incl. -K), sense[m].
 * Replaces the static workspace of generated mpc_workspace.c / mpc_compute_control
 * (codegen/mpc_update_qp.c:29-54).
 */
int lmpc_setup_ldp(lmpc_handle **out, int n, int m, int ms, int nth, int nout,
                   const double *M, const double *du, const double *dl, const double *Dth,
                   const double *Rout, const double *x0, const double *Xth,
                   const int32_t *sense, const lmpc_settings *s, int device);

/*
 * Host-only half of lmpc_setup: the QP -> LDP transform (reference codegen.jl:239-280 `qp2ldp`,
 * i.e. what LinearMPC.codegen writes into the generated C arrays) without touching a GPU.
 * Inputs as lmpc_setup, outputs row-major as lmpc_setup_ldp takes them; any output may be NULL.
 */
int lmpc_transform(int n, int m, int ms, int nth, int nout,
                   const double *H, const double *f, const double *f_theta,
                   const double *A, const double *bu, const double *bl, const double *W,
                   const int32_t *sense, const double *Kfb, int nx,
                   double *M, double *du, double *dl, double *Dth,
                   double *Rout, double *x0, double *Xth);

/* Host-only half of an is_avi setup (no GPU touched): the transform described at lmpc_setup_ex, outputs row-major --
 * ML[m*n], MR[m*n], G[m*m], du[m], dl[m], Dth[m*nth], Rout[nout*n] (leading rows of I), x0[nout], Xth[nout*nth];
 * any output may be NULL.  LMPC_ERR_NONCONVEX if H + H' is not positive definite. */
int lmpc_transform_avi(int n, int m, int ms, int nth, int nout,
                       const double *H, const double *f, const double *f_theta,
                       const double *A, const double *bu, const double *bl, const double *W,
                       const int32_t *sense, const double *Kfb, int nx,
                       double *ML, double *MR, double *G, double *du, double *dl, double *Dth,
                       double *Rout, double *x0, double *Xth);

/* Copies the constant pack held by the handle back to the caller (row-major, sizes as in
 * lmpc_setup_ldp); any pointer may be NULL.  Lets a test feed the SAME pack to the oracle. */
int lmpc_get_ldp(const lmpc_handle *h, double *M, double *du, double *dl, double *Dth,
                 double *Rout, double *x0, double *Xth, int32_t *sense);

/* dims[0..5] = n, m, ms, nth, nout, active words per problem */
int lmpc_get_dims(const lmpc_handle *h, int32_t dims[6]);
int lmpc_active_words(const lmpc_handle *h);

/* Replaces DAQP.settings on a live model. */
int lmpc_set_settings(lmpc_handle *h, const lmpc_settings *s);

/*
 * THE hot path.  Replaces, for N parameter points at once, the body of `solve`
 * (utils.jl:272-282: bu/bl/f update, DAQP.update, DAQP.solve) and the primal recovery
 * x = R^-1 u + offsets (codegen/mpc_update_qp.c:14-22).  Cold start per problem
 * (mpc_update_qp.c:44-47) unless `warm` is given.
 *
 *   theta     N records of nth doubles        x        N records of nout doubles (out)
 *   exitflag  N int32 (out)                   iters    N int32 (out) or NULL
 *   active    N records of lmpc_active_words words (out) or NULL
 *   warm      N records of the same layout: initial working sets (in) or NULL = cold
 *
 * lmpc_solve_batch: HOST pointers; synchronous.  Large batches move through a three-stage pipeline in chunks
 * (each half of what is left, the last one "host_chunk" problems): H2D copy of chunk k+1, kernels of chunk k,
 * D2H copy of chunk k-1 on three streams.  With arrays the caller pinned (lmpc_pin_host) every copy is
 * asynchronous and one thread drives the pipeline (1.17 ms per 10^6 pendulum problems, PCIe-bound: 56 MB in at
 * 52 GB/s); with pageable arrays, whose copy calls block, the upload and download sides run on two host threads
 * ("host_threads" 0: one chunk, one thread; 1.25 against 1.30 ms).  (The option "host_register", which pinned the
 * caller's arrays for the duration of each call, was removed in round 5 -- it ended in a GPU memory fault after a few
 * hundred calls -- and lmpc_set_option answers LMPC_ERR_UNSUPPORTED for it.)
 * lmpc_solve_batch_device: DEVICE pointers on the handle's GPU; enqueues the kernels on
 * `stream` (a hipStream_t passed as void*, NULL = default stream) and returns without
 * synchronising -- this is what bench.py times with inputs resident in HBM.
 * One exception, once per handle: the FIRST large batch (>= 65 536 points) of a fresh wavefront-kernel handle whose
 * working sets can exceed 40 rows first solves its own leading 16 384 points into scratch and WAITS for that launch
 * (the probe that decides the launch shape; lmpc_set_option "wave_probe" 0 switches it off).  Inside a stream capture
 * (hipStreamBeginCapture) the probe is skipped and left for the first call outside a capture.
 */
int lmpc_solve_batch(lmpc_handle *h, int64_t N, const double *theta, double *x,
                     int32_t *exitflag, int32_t *iters, uint64_t *active,
                     const uint64_t *warm);
/* lmpc_solve_batches_device: n_batches batches of N points each in ONE call -- theta / x / exitflag are HOST tables of
 * n_batches DEVICE pointers (the tables are read during the call, the batches' buffers need not be adjacent).  Cold
 * plain solves; results are those of n_batches calls of lmpc_solve_batch_device bit for bit.  On the handles the
 * one-launch kernel covers (small box-constrained problems: the headline configuration) up to eight batches go into
 * one kernel launch, in which the solving tail of a batch runs under the stream of the next: 10^6-point pendulum
 * batches cost ~16 us each in a call of three against 22 us one call at a time.  Elsewhere it enqueues the batches one
 * after the other.  Caller side: the same loop over parameter batches that calls solve (reference src/utils.jl:268-283). */
int lmpc_solve_batches_device(lmpc_handle *h, int32_t n_batches, int64_t N, const double *const *theta, double *const *x,
                              int32_t *const *exitflag, void *stream);
int lmpc_solve_batch_device(lmpc_handle *h, int64_t N, const double *theta, double *x,
                            int32_t *exitflag, int32_t *iters, uint64_t *active,
                            const uint64_t *warm, void *stream);

/*
 * The same hot path in binary32 -- the reference's single-precision build of this path
 * (codegen.jl:19,31-37,82: float_type = "float" makes `c_float` a float and compiles DAQP with
 * DAQP_SINGLE_PRECISION, so mpc_compute_control takes and returns floats).  theta and x are float
 * records; the constant pack is the handle's binary64 pack rounded to binary32 on the first call;
 * tolerances are the handle's settings rounded to binary32 -- give the handle settings that make
 * sense at that precision (lmpc_default_settings_f32: primal 1e-4, dual 1e-6, zero 1e-6,
 * progress 1e-4, rho_soft 1e-3; libdaqp's own single-precision defaults live in a header that is
 * not part of the reference tree).  Runs on the wavefront kernel (any handle whose problem it covers:
 * n <= 127, 1 <= m <= 1024), including branch and bound over BINARY rows (n <= 64)
 * (BASELINE config 5); LMPC_ERR_UNSUPPORTED otherwise.
 */
void lmpc_default_settings_f32(lmpc_settings *s);
int lmpc_solve_batch_f32(lmpc_handle *h, int64_t N, const float *theta, float *x,
                         int32_t *exitflag, int32_t *iters, uint64_t *active,
                         const uint64_t *warm);
int lmpc_solve_batch_f32_device(lmpc_handle *h, int64_t N, const float *theta, float *x,
                                int32_t *exitflag, int32_t *iters, uint64_t *active,
                                const uint64_t *warm, void *stream);

/* A caller that reuses its Theta / X / exitflag arrays from call to call (a Monte-Carlo loop, a closed
 * loop over many scenarios) can pin them ONCE: lmpc_solve_batch* then runs fully asynchronous copies on them.
 * Pinning works on whole pages: give every array pages of its own (a page-aligned allocation, e.g. mmap /
 * posix_memalign; Julia: Mmap.mmap(Matrix{Float64}, dims)) -- a range that shares a page with memory pinned
 * earlier is refused with LMPC_ERR_BADARG.  Unpin before the memory is freed. */
int lmpc_pin_host(void *p, size_t bytes);
int lmpc_unpin_host(void *p);

/*
 * Several GPUs behind ONE call -- the shape the reference's caller has (one process, one Theta:
 * /root/reference/src/utils.jl:268-283 `solve`; SURVEY.md section 8(b) `n_devices`, section 8(e)).
 *
 * lmpc_setup_multi: lmpc_setup on every listed device (devices == NULL or n_devices <= 0: all visible
 * devices, 0 .. count-1); the constant pack is replicated, each device gets its own handle.
 * lmpc_solve_batch_multi: HOST pointers as lmpc_solve_batch.  The batch is cut into contiguous shards
 * (lmpc_multi_partition: N/n_devices each, the remainder on the leading devices); every device moves its
 * shard through the same chunked H2D / kernels / D2H pipeline lmpc_solve_batch uses, all driven from the
 * calling thread, results written straight into the caller's x / exitflag / iters / active.
 * lmpc_solve_batch_multi_device: shards already RESIDENT on their GPUs (theta[d], x[d], exitflag[d] are
 * DEVICE pointers on device d, N_dev[d] problems).  Every device solves its shard on its own stream; if
 * x_root / exitflag_root (DEVICE pointers on the FIRST device, sum(N_dev) records) are given, the
 * per-shard solutions are gathered there over xGMI with RCCL (ncclCommInitAll in this process,
 * ncclSend / ncclRecv pairs: every device sends on its own link).  Returns after all devices finished.
 * The problems are independent: no collective runs inside the solve.
 */
typedef struct lmpc_multi lmpc_multi;
int lmpc_setup_multi(lmpc_multi **out, int n, int m, int ms, int nth, int nout,
                     const double *H, const double *f, const double *f_theta,
                     const double *A, const double *bu, const double *bl, const double *W,
                     const int32_t *sense, const double *Kfb, int nx,
                     const lmpc_settings *s, const int *devices, int n_devices);
int lmpc_multi_devices(const lmpc_multi *hm);
/* "transport": how lmpc_solve_batch_multi_device gathers the shards on the first device -- 0 (default) RCCL
 * ncclSend / ncclRecv pairs, every device on its own xGMI link; 1 = event-ordered peer copies issued by the first
 * device's stream (hipMemcpyPeerAsync: the copy engines; needs no RCCL).  With the environment variable
 * LMPC_MULTI_TRANSPORT=copy at setup time transport 1 is the default AND the device list may name a device more than
 * once: several shards, handles, streams and host threads on one GPU -- the way the n_devices > 1 code is exercised on
 * a machine with a single GPU (tests/test_gpu_parity.py::test_multi_device_control_flow_on_one_gpu). */
int lmpc_multi_set_option(lmpc_multi *hm, const char *name, int value);
lmpc_handle *lmpc_multi_handle(lmpc_multi *hm, int i);      /* device i's handle (options, inspection) */
void lmpc_multi_partition(int64_t N, int n_devices, int64_t *offsets /* n_devices + 1 */);
int lmpc_solve_batch_multi(lmpc_multi *hm, int64_t N, const double *theta, double *x,
                           int32_t *exitflag, int32_t *iters, uint64_t *active, const uint64_t *warm);
int lmpc_solve_batch_multi_device(lmpc_multi *hm, const int64_t *N_dev, const double *const *theta,
                                  double *const *x, int32_t *const *exitflag,
                                  double *x_root, int32_t *exitflag_root);
const char *lmpc_multi_last_error(const lmpc_multi *hm);
void lmpc_free_multi(lmpc_multi *hm);

/* N = 1 convenience with DAQP.solve's shape: returns the exit flag (or an LMPC_ERR_* <= -100),
 * x[nout] out.  What Simulation's per-step compute_control (simulation.jl:106) would call.
 * Ordering: the call runs on a stream of the handle's own, NOT ordered with the caller's streams, and
 * uses the handle's scratch (work lists, counters, warm-start state).  Everything enqueued earlier on this handle through
 * a *_device entry point must have completed before lmpc_solve_one (or a host-array lmpc_compute_control with a handful
 * of states, which takes the same route) is called; such a call also counts as one call for the "even number of calls
 * per captured graph" rule of the *_device entry points.  One thread at a time per handle, as for a DAQP workspace. */
int lmpc_solve_one(lmpc_handle *h, const double *theta, double *x);

/*
 * Batched closed-loop simulation: N independent scenarios advanced T steps on the GPU (by default each
 * scenario at its own pace -- the scenarios do not interact, so the arrays returned are those of the
 * step-by-step loop bit for bit; lmpc_set_option "sim_async").  That execution order reads a few
 * counters back between rounds: lmpc_simulate_device waits on `stream` a few times before it returns
 * (its outputs are still only complete once `stream` is), so it cannot be captured into a hipGraph;
 * with "sim_async" 0 it only enqueues.
 * Per step and scenario it does what one pass of the reference's Simulation loop does
 * (src/simulation.jl:93-113 without observer): theta = [x; r; uprev] (src/explicit.jl:54-63),
 * u = compute_control (the batched solve of this handle, which must have been set up with
 * nout = nu and, if used, the prestabilising feedback folded in), x <- F x + G u, uprev <- u.
 * warm != 0 starts each solve from the previous step's final working set (the reference's
 * DAQP_WARMSTART build, codegen/mpc_update_qp.c:44-47); the first step is cold.  On the wavefront-kernel
 * path (binary64, no binaries) the working set comes back WITH its factorisation, in its order -- libdaqp's
 * workspace under DAQP_WARMSTART is simply not cleared between two calls (mpc_update_qp.c:44-54) -- kept per
 * scenario in device memory (17 KB at the largest capacity); lmpc_set_option "sim_keep_factor" 0 re-appends the
 * rows of the previous step's mask instead, as the lane kernels and the binary32 loop do.  Either way the
 * inputs are those of the cold loop up to the tolerances (K6, test/runtests.jl:85-117).
 *
 *   nx + nr + nuprev must equal nth.  F[nx*nx], G[nx*nu] row-major HOST arrays (the plant).
 *   x      N records of nx: in = initial states, out = states after T steps
 *   r      N records of nr (constant reference per scenario) or NULL = 0
 *   uprev  N records of nuprev, in/out (NULL allowed when nuprev == 0)
 *   U_traj T*N*nu (step-major) or NULL;  X_traj (T+1)*N*nx or NULL
 *   flag_min N int32: smallest exit flag seen over the T steps (>= 1 means every solve succeeded) or NULL
 * lmpc_simulate takes HOST pointers; lmpc_simulate_device DEVICE pointers (except F, G) and a stream.
 */
int lmpc_simulate(lmpc_handle *h, int64_t N, int T, int nx, int nr, int nuprev,
                  const double *F, const double *G, double *x, const double *r, double *uprev,
                  double *U_traj, double *X_traj, int32_t *flag_min, int warm);
int lmpc_simulate_device(lmpc_handle *h, int64_t N, int T, int nx, int nr, int nuprev,
                         const double *F, const double *G, double *x, const double *r, double *uprev,
                         double *U_traj, double *X_traj, int32_t *flag_min, int warm, void *stream);

/*
 * The closed loop in binary32 -- the reference's generated controller built with float_type = "float"
 * (codegen.jl:19,31-37) inside a simulation: x, r, uprev and the trajectories are float records, the
 * plant (F, G: HOST binary64 arrays as everywhere) is rounded to binary32 like the constant pack, the
 * plant step is an fmaf chain in the same order.  Wavefront kernel (see lmpc_solve_batch_f32).
 */
int lmpc_simulate_f32(lmpc_handle *h, int64_t N, int T, int nx, int nr, int nuprev,
                      const double *F, const double *G, float *x, const float *r, float *uprev,
                      float *U_traj, float *X_traj, int32_t *flag_min, int warm);
int lmpc_simulate_f32_device(lmpc_handle *h, int64_t N, int T, int nx, int nr, int nuprev,
                             const double *F, const double *G, float *x, const float *r, float *uprev,
                             float *U_traj, float *X_traj, int32_t *flag_min, int warm, void *stream);

/*
 * Batched form_parameter on the device, previews included: theta_i = [x_i; r; d; uprev_i; p]
 * (reference explicit.jl:54-63) with r, d, p formatted as format_reference / format_disturbance /
 * format_affine_parameters do (utils.jl:78-261) and, inside a closed loop, as Simulation's
 * get_preview does (simulation.jl:100-103,128-134).
 *
 * A block is cut from a trajectory: `w` values per column, `T` columns stored column by column
 * (a Julia w x T matrix) -- one matrix per scenario, `stride` doubles apart, or one shared by all
 * scenarios (stride 0).  H == 0: the block is column k0 (a constant reference: T = 1, k0 = 0);
 * H > 0 (preview over H steps, H = mpc.Np): columns k0 .. k0+H-1, flattened column by column.
 * Columns past the end repeat the last one (how the reference pads a short trajectory); src == NULL
 * gives zeros (the reference's `nothing` default); w == 0 leaves the block out.
 * nx + width(r) + width(d) + nuprev + width(p) must equal the handle's nth.
 */
typedef struct lmpc_block {
    const double *src;   /* DEVICE pointer or NULL                                  */
    int64_t stride;      /* doubles between two scenarios' matrices, 0 = shared     */
    int32_t w;           /* values per column (ny, nd or the base parameter count)  */
    int32_t T;           /* columns available                                       */
    int32_t k0;          /* first column taken                                      */
    int32_t H;           /* 0 = one column, > 0 = preview over H columns            */
} lmpc_block;
int lmpc_form_parameter_device(lmpc_handle *h, int64_t N, double *theta, const double *x, int nx,
                               const lmpc_block *r, const lmpc_block *d, const double *uprev,
                               int nuprev, const lmpc_block *p, void *stream);

/*
 * Closed loop with a reference TRAJECTORY (reference simulation.jl:69-73,93-113: `rs` held at its
 * last column, and with settings.reference_preview the controller sees columns k+1 .. k+Np at
 * step k).  As lmpc_simulate_device, but the reference of step k (0-based) is cut from `r`:
 * r->H == 0: column k; r->H > 0: columns k+1 .. k+H (r->k0 is ignored).  DEVICE pointers.
 */
int lmpc_simulate_ref_device(lmpc_handle *h, int64_t N, int T, int nx, const lmpc_block *r, int nuprev,
                             const double *F, const double *G, double *x, double *uprev,
                             double *U_traj, double *X_traj, int32_t *flag_min, int warm, void *stream);

/*
 * The batched counterpart of the reference's GENERATED controller
 *   int mpc_compute_control(c_float* control, c_float* state, c_float* reference,
 *                           c_float* disturbance[, c_float* affine_parameter])
 * (reference src/codegen.jl:1-17, codegen/mpc_update_qp.h:3-7, body codegen/mpc_update_qp.c:29-54):
 * mpc_update_parameter (codegen/mpc_update_parameter.c:1-29) + mpc_update_qp + daqp_ldp/daqp_bnb +
 * mpc_get_solution for N problems at once.  The handle must have been set up with nout = N_CONTROL.
 *
 * lmpc_set_parameter_layout gives the handle the constants the generated header defines
 * (codegen.jl:154-165): N_STATE, N_REFERENCE, N_DISTURBANCE, N_CONTROL_PREV, N_AFFINE_PARAMETER --
 * they must add up to nth -- and, for settings.reference_condensation (codegen.jl:191-195),
 * N_PREVIEW_HORIZON with traj2setpoint[n_reference*n_preview_horizon*n_reference] exactly as the
 * generator writes it (mpc.traj2setpoint[:], HOST array, copied).
 *
 *   control   N records of nout doubles: in = previous control (first n_control_prev entries are
 *             read), out = u* -- the in/out convention of the generated function
 *   state     N records of n_state            reference  N records of n_reference doubles, or of
 *             n_reference*n_preview_horizon (an n_reference x n_preview_horizon trajectory, column by
 *             column) when condensing; disturbance / affine_parameter: N records of their widths.
 *             NULL for reference / disturbance / affine_parameter = zeros (runtests.jl:943 passes NULL
 *             for absent blocks).
 *   exitflag  N int32 (out) or NULL           warm != 0 = the DAQP_WARMSTART build
 *             (mpc_update_qp.c:44-47): each problem starts from the working set its previous call
 *             with the same N ended with; 0 = cold start every call.
 * lmpc_compute_control: HOST pointers, synchronous.  lmpc_compute_control_device: DEVICE pointers,
 * enqueued on `stream`, no synchronisation.
 */
typedef struct lmpc_param_layout {
    int32_t n_state, n_reference, n_disturbance, n_control_prev, n_affine_parameter;
    int32_t n_preview_horizon;       /* 0 unless settings.reference_condensation */
    const double *traj2setpoint;     /* HOST, or NULL when n_preview_horizon == 0 */
} lmpc_param_layout;
int lmpc_set_parameter_layout(lmpc_handle *h, const lmpc_param_layout *layout);
int lmpc_compute_control(lmpc_handle *h, int64_t N, double *control, const double *state,
                         const double *reference, const double *disturbance,
                         const double *affine_parameter, int32_t *exitflag, int warm);
int lmpc_compute_control_device(lmpc_handle *h, int64_t N, double *control, const double *state,
                                const double *reference, const double *disturbance,
                                const double *affine_parameter, int32_t *exitflag, int warm, void *stream);

/*
 * The reference's GENERATED state observer for N scenarios at once: mpc_predict_state(state, control,
 * disturbance) and mpc_correct_state(state, measurement, disturbance) (codegen/mpc_observer.c:1-28,
 * mpc_observer.h; Julia: predict! / correct! of the KalmanFilter, src/observer.jl:104-123), the other
 * two functions the generated controller exports next to mpc_compute_control.
 *
 * lmpc_set_observer takes the arrays exactly as the generator writes them (src/observer.jl:124-140):
 *   plant_dynamics        MPC_PLANT_DYNAMICS       n_state rows [f_offset_i, F_i(n_state), G_i(n_control), Gd_i(n_disturbance)]
 *   measurement_function  MPC_MEASUREMENT_FUNCTION n_measurement rows [h_offset_j, C_j(n_state), Dd_j(n_disturbance)]
 *   k_transpose           K_TRANSPOSE_OBSERVER     n_measurement rows of n_state (the Kalman gain, transposed)
 * (HOST arrays, copied).  state: N records of n_state, updated in place; control / measurement /
 * disturbance: N records of their widths; disturbance may be NULL (zeros), as in the reference's calls
 * (runtests.jl:942-946).  *_device variants take DEVICE pointers and a stream and do not synchronise.
 */
typedef struct lmpc_observer {
    int32_t n_state, n_control, n_disturbance, n_measurement;
    const double *plant_dynamics, *measurement_function, *k_transpose;
} lmpc_observer;
int lmpc_set_observer(lmpc_handle *h, const lmpc_observer *obs);
int lmpc_predict_state(lmpc_handle *h, int64_t N, double *state, const double *control, const double *disturbance);
int lmpc_correct_state(lmpc_handle *h, int64_t N, double *state, const double *measurement,
                       const double *disturbance);
int lmpc_predict_state_device(lmpc_handle *h, int64_t N, double *state, const double *control,
                              const double *disturbance, void *stream);
int lmpc_correct_state_device(lmpc_handle *h, int64_t N, double *state, const double *measurement,
                              const double *disturbance, void *stream);

/*
 * The generated offset-free observer's controller call for N scenarios (reference src/observer.jl:156-196):
 *   int mpc_compute_control_observer(control, observer_state, reference, measured_disturbance[, affine_parameter])
 * = mpc_get_estimated_state (state = observer_state[0:n_state]) + mpc_get_estimated_disturbance (disturbance =
 * [measured_disturbance or zeros; observer_state[n_state : n_state + n_offset_free]]) + mpc_compute_control.
 * n_offset_free = the layout's n_disturbance - n_measured_disturbance; observer_state: N records of
 * n_state + n_offset_free doubles.  DEVICE pointers, enqueued on `stream`.
 */
int lmpc_compute_control_observer_device(lmpc_handle *h, int64_t N, double *control, const double *observer_state,
                                         int n_measured_disturbance, const double *reference,
                                         const double *measured_disturbance, const double *affine_parameter,
                                         int32_t *exitflag, int warm, void *stream);

/* Which kernel variant the handle dispatches to (for benchmark reports), e.g. "lane<5>". */
const char *lmpc_kernel_name(const lmpc_handle *h);

/* Working-set statistics of the wavefront kernel, as the handle last saw them (cumulative over its launches, a launch or
 * two behind; read from mapped host memory, no synchronisation): out[0] problems the kernel finished, out[1..3] of them
 * those whose working set never held more than 24 / 32 / 48 rows, out[4] the working-set capacity of the first pass the
 * next call would use (0 = one pass).  This is what decides whether a batch first runs at a smaller capacity
 * (lmpc_set_option "wave_two_pass"); reported for benchmark logs and tests.  Returns LMPC_OK (all zeros before the
 * first wavefront-kernel launch). */
int lmpc_wave_stats(lmpc_handle *h, unsigned long long out[5]);

/* Device time per lmpc_solve_batch* call, measured with HIP events recorded on the launch
 * stream (switch on with lmpc_profile(h, 1); timing-only events, hipEventDisableSystemFence: a default event's
 * system-scope cache writeback added ~1.7 us to a 24 us call that rocprofv3's kernel trace does not see).  lmpc_profile_read waits for the recorded calls,
 * averages over the calls made since the last read and returns how many there were (<0: error):
 *   avg_ms[0] whole call (screening pass + iterating pass), avg_ms[1] screening kernel,
 *   avg_ms[2] iterating kernel. */
int lmpc_profile(lmpc_handle *h, int enable);
int lmpc_profile_read(lmpc_handle *h, double avg_ms[3]);

/* Tuning switches (none of them changes a result): "in_flight" k = a HINT: the caller keeps k independent batches in
 * flight on this GPU (one handle and one stream each; default 1).  From k = 2 on the kernels take the workgroup shapes
 * that suit a shared chip ("fast_nstr" 4, "fast_tiles" 28, "lane_block" 64) instead of the stand-alone ones.
 * "fast_dma" = how the one-launch kernel's streaming wavefronts take their records: 2 (default with up to three
 * streaming wavefronts per workgroup) = by LDS-DMA (global_load_lds_dwordx4, nontemporal) into a ring of two tiles per
 * wavefront, 0 = through registers.  "gram_scan" (default 0) is NOT one of them: it selects the wavefront kernel's
 * Gram-scan form, whose results differ from the default form in the last bits (see lmpc_wave_kernel.hpp; checker: the
 * oracle's mode 1).  "fast" (default 1) = small box-constrained problems
 * (m == ms == n <= 5, up to 16 parameters -- 8 for n = 5 --, cold start) are solved by ONE kernel that streams the batch and
 * runs the iterations underneath (0 = the two-kernel form below); "fast_nstr" 1..4 = streaming wavefronts per
 * workgroup of that kernel (default 3; 4 suits several batches in flight), "fast_tiles" = tiles of 64 problems per workgroup
 * (default: one resident round of workgroups; 28 suits three 10^6-point batches in flight); "lane_straight" (default 1) =
 * straight-line first tier in the boxed iterating kernels.  "screen" (default 1) = run cold-start batches through the streaming
 * screening pass before the iterating kernel; 0 = iterating kernel only.  "screen_wave" (default 1) = the same pass in
 * front of the wavefront kernel (binary64 problems without binary rows and without initially active rows, 1..32
 * parameters): it finishes every problem whose unconstrained optimum is feasible -- what the wavefront kernel's first
 * iteration would do -- and the wavefront kernel walks a work list of the others.  "big_path" (default 1) = points
 * whose working set outgrows the wavefront kernel's 64 rows are re-solved by the one-problem-per-thread kernel
 * (0 = they keep exit flag -7).  Results are
 * bit-identical either way.  Closed loop (lmpc_simulate*): "sim_async" (default 1) = scenarios
 * advance independently of each other (rounds of a streaming kernel and the iterating kernel; on the wavefront-kernel
 * path whenever "sim_run_ahead" applies, 2 = there in any case), 0 = all scenarios step by step together;
 * "sim_run_ahead" (default 1; wavefront-kernel path) = a scenario whose step ends with a non-empty working set stays
 * inside the wavefront kernel for its next step, warm on the factorisation as it stands (cold: from nothing) -- same
 * results as the step-synchronous loop with "sim_keep_factor" 1; "sim_blind" (default 2) = rounds enqueued between two
 * reads of the work-list counters; "sim_small" (default 1) = the all-in-registers instantiation
 * of the streaming kernel for single-input problems with nx <= 4, m <= 8; "sim_fused" (default 1) = plant step inside the solve's
 * kernels (lane kernels; round 3: the wavefront-kernel path's lock-step loop too).  All of them change the execution
 * order only, never a result.  "sim_keep_factor" (default 1; wavefront-kernel path, warm closed loop) = continue each
 * step from the kept factorisation instead of re-appending the previous mask's rows: same optimum up to the
 * tolerances, last bits of u may differ (checker: oracle_simulate warm = 2 against warm = 1).
 * Wavefront kernel, working-set capacity: "wave_two_pass" (default -1 = decided by the handle's statistics, see
 * lmpc_wave_stats; 0 = never; 1 = always, at "wave_cap1" rows, default 24) = run a batch first at a smaller capacity
 * (more wavefronts resident) and hand the points that outgrow it to a second launch at the full capacity;
 * "wave_cap" c (8 .. 64, 0 = the problem's own) = the full capacity itself, points beyond it go to the slow path.
 * Neither changes a result.
 * "row_kernel" (default -1): cold plain batches with n <= 64 variables, m <= 160 rows (from 8 192 problems on; 16 < n <= 32,
 * m <= 96: whatever the batch size) and branch-and-bound searches (m <= 64, up to 47 binary rows; whatever the batch
 * size: ONE search takes 1.0 ms there against 1.5 ms) run on the four-problems-per-wavefront kernel
 * (lmpc_row_kernel.hpp: one problem per 16-lane DPP row) where the handle's statistics let its working-set capacity --
 * 16 / 31 / 32 rows, searches 16 / 48 -- hold nearly all points; what outgrows it is listed for the wavefront kernel
 * (the two-pass protocol above).  -1 = where it measured faster (binary64 every shape, binary32 the two-slot shape and
 * the searches), 1 = wherever an instantiation covers the problem, whatever the batch size, 0 = never.  "row_blocks" =
 * its workgroups per CU (tuning).  Results identical either way (tools/fuzz_row.py, tools/fuzz_row_bnb.py).
 * "qp_tiers": problems with n = 2 .. 12 variables and up to 64 hard or SOFT rows (no other flags) can send cold plain
 * binary64 batches through a tiers pass (one problem per lane, append-only paths finished in registers) in front of the
 * wavefront / lane kernel.  2 = always, 0 = never (the screening pass as before), 1 (default) = yes -- and for batches of
 * 65 536 problems and more on the wavefront path whichever of the two the handle has measured faster (the best of three
 * calls each way, timed by events that later calls read without waiting; measured again every 512 calls).  Results identical either way.
 * Variational handles (is_avi) with n <= 8 simple bounds run a chain of register-resident kernels in front of the
 * generic one: "avi_tiers" (default 1; 0 = the generic kernel alone), "avi_tiers_first" (-1 = default: 3 up to n = 6,
 * else 2; 1 .. 3 = straight-line tiers of the pass over the whole batch; 0 = the complete lane kernel over the whole
 * batch), "avi_waves" (generic kernel: wavefronts per CU, default 16).  Results identical either way.
 * hipGraph capture of calls on one handle: the work-list and ticket counters alternate between two sets, each call
 * clearing the set of the next one -- capture an EVEN number of calls per handle. */
int lmpc_set_option(lmpc_handle *h, const char *name, int value);

/* Distinct optimal active sets of a solved batch, reduced on the device: the caller side of a sampling-based
 * explicit-MPC region discovery / complexity certificate (/root/reference/src/explicit.jl:23-48 hands the mpQP to
 * ParametricDAQP for an exact enumeration; what the batched backend contributes is the sample: solve, then one entry per
 * critical region the sample hit).  `active` and `exitflag` are lmpc_solve_batch_device's outputs (device pointers;
 * exitflag == NULL: every problem counts, else those with exitflag >= 1).  Outputs, device pointers, dense in
 * [0, *n_sets): set_masks[k * words ..] the mask, set_count[k] how many problems ended on it, set_first[k] the
 * smallest problem index that did.  `capacity` = room in the three arrays; more distinct sets than that raise the
 * overflow word (lmpc_distinct_active_sets_overflowed: 0 / nonzero, synchronises the stream) and *n_sets then counts
 * the claims, not the stored sets.  Order of the sets is unspecified (sort on the host).  Asynchronous on `stream`. */
int lmpc_distinct_active_sets_device(lmpc_handle *h, int64_t N, const uint64_t *active, const int32_t *exitflag,
                                     int32_t capacity, uint64_t *set_masks, int64_t *set_count, int64_t *set_first,
                                     int32_t *n_sets, void *stream);
int lmpc_distinct_active_sets_overflowed(lmpc_handle *h, void *stream);

/* One step of a sampling-based region discovery in ONE enqueue: lmpc_solve_batch_device on the resident sample with the
 * active-set masks kept on the device, lmpc_distinct_active_sets_device on them, and a last kernel that writes the
 * distinct sets into a block of mapped host memory owned by the handle -- no copy call, no synchronisation inside.
 * theta / x / exitflag / active and the four set arrays are DEVICE pointers as in the two calls it chains.
 * *result_host (valid until the next call of this function on the handle or lmpc_free) is read after ONE
 * synchronisation of `stream`: 64-bit words
 *     [0] sets found (more than `capacity`: the call overflowed, repeat with more room; -1: not finished yet)
 *     [1] overflow word   [2] problems counted (exit flag >= 1)   [3] unused
 *     then per set k < min([0], capacity): `words` words of mask, its count, the smallest problem index that hit it.
 * The sets come in no particular order (sort on the host).  Caller side of /root/reference/src/explicit.jl:23-48. */
int lmpc_discover_regions_device(lmpc_handle *h, int64_t N, const double *theta, double *x, int32_t *exitflag,
                                 uint64_t *active, int32_t capacity, uint64_t *set_masks, int64_t *set_count,
                                 int64_t *set_first, int32_t *n_sets, const long long **result_host, void *stream);

/* Per-problem exit flags are DAQP's (1 optimal, 2 soft optimal, -1 infeasible, -2 cycle, -4 iteration limit, -6
 * over-determined initial working set) plus two of this library's own:
 *   -7  the working set outgrew what the kernels hold (wavefront kernel: 64 rows, slow path: 256 rows);
 *   -8  LMPC_EXIT_UNFINISHED: the problem was queued inside the one-launch kernel and never solved because one of that
 *       kernel's bounded waits ran out (not expected to happen; the provisional flag every queued problem carries until
 *       its solving lane overwrites it -- so a failure can never be read as success, /root/reference/src/utils.jl:46
 *       `@assert exitflag>=1`).
 * The *_device entry points are asynchronous, so such a failure is reported by the handle afterwards: the next call on
 * the handle, lmpc_profile_read, lmpc_release_scratch and lmpc_check return LMPC_ERR_HIP once (text in
 * lmpc_last_error); the host-pointer entry points (lmpc_solve_batch, ...) report it from the call itself.
 * lmpc_check waits for the handle's GPU first, i.e. it answers for everything enqueued so far.
 * ("fast_spin_limit" k > 0 of lmpc_set_option is the test hook: the waits give up after k - 1 polls.) */
#define LMPC_EXIT_UNFINISHED (-8)
int lmpc_check(lmpc_handle *h);

/* lmpc_reserve: allocate NOW what the *_device entry points would allocate lazily inside their first call on a batch of
 * N problems (work lists, overflow lists, counters, the slow path's scratch), so that the first call enqueues kernels
 * and nothing else.  Optional; on a wavefront-kernel handle it also sends ONE dummy problem through the kernel on
 * `stream` and waits for it (the queue's first dispatch of that kernel costs ~4 ms), otherwise asynchronous (a few
 * memsets).  (The kernels' code objects are loaded by lmpc_setup*, not by the first solve.)  What a first call on a fresh
 * wavefront-kernel handle still does by itself, once: it solves the leading 16 384 points of its own batch in front
 * into scratch outputs, waits for that launch and reads the working-set sizes it saw, so that the batch runs in the
 * launch shape a warmed-up handle would choose (lmpc_set_option "wave_probe" 0: never; such a handle's first calls run
 * in one pass until its statistics exist).  The statistics behind that choice cover the most recent one to two
 * million problems, not the handle's whole life. */
int lmpc_reserve(lmpc_handle *h, int64_t N, void *stream);

/* Staging and scratch buffers of a handle grow with the largest batch it has seen and are kept between calls.
 * lmpc_release_scratch waits for the handle's GPU and gives them back (the constant pack stays; the next call
 * allocates what it needs again, and a DAQP_WARMSTART state kept by lmpc_compute_control* is dropped). */
int lmpc_release_scratch(lmpc_handle *h);

void lmpc_free(lmpc_handle *h);

/* Last error text of this handle (or of the failed setup call when h == NULL). */
const char *lmpc_last_error(const lmpc_handle *h);

/* Library/ABI version, bumped on any signature change. */
int lmpc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LMPC_HIP_H */
