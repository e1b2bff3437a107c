// C ABI of liblmpc_hip.so, second unit (round 5: split off lmpc_api.hip): everything that runs a LOOP around the batched solve
// on the device -- closed-loop simulation (lmpc_simulate*), the generated controller's entry points
// (lmpc_set_parameter_layout, lmpc_compute_control*, lmpc_form_parameter_device) and the generated observer
// (lmpc_set_observer, lmpc_predict_state*, lmpc_correct_state*).  Setup, the plain solves, options and the launch
// policy stay in lmpc_api.hip; this unit reaches them through the api_* functions of lmpc_internal.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lmpc_internal.hpp"
#include "lmpc_lane_kernel.hpp"
#include "lmpc_sim_kernels.hpp"

using namespace lmpc;

namespace lmpc {
// this unit's code object onto the device without a launch (setup; see preload_code in lmpc_api.hip)
void loop_preload() {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void *)form_theta_kernel<double>);
    (void)hipGetLastError();
}
}  // namespace lmpc

extern "C" {

// per-scenario kept closed-loop state of the wavefront path (working set + factorisation, lmpc_wave_kernel.hpp): makes
// room for N scenarios if the device has it to spare, marks every state "nothing kept", sets h->keepOn
static int ensure_keep(lmpc_handle *h, int64_t N, hipStream_t st) {
    const size_t keepR = (size_t)h->W.keepStride, keepI = 5 * 64;
    if (N > h->keepCap) {
        hipFree(h->dKeepR); hipFree(h->dKeepI); h->dKeepR = nullptr; h->dKeepI = nullptr; h->keepCap = 0;
        // (17 GB for 1e6 scenarios at capacity 64: only while it is at most half of what the device has free --
        // beyond that the loop runs on masks rather than crowding out the caller)
        size_t freeB = 0, totalB = 0;
        const size_t needB = (sizeof(double) * keepR + sizeof(int32_t) * keepI) * (size_t)N;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { freeB = 0; (void)hipGetLastError(); }
        if (needB <= freeB / 2 &&
            hipMalloc(&h->dKeepR, sizeof(double) * keepR * (size_t)N) == hipSuccess &&
            hipMalloc(&h->dKeepI, sizeof(int32_t) * keepI * (size_t)N) == hipSuccess) h->keepCap = N;
        else { hipFree(h->dKeepR); hipFree(h->dKeepI); h->dKeepR = nullptr; h->dKeepI = nullptr; (void)hipGetLastError(); }
    }
    if (N <= h->keepCap) {
        // nothing kept yet: the size word of every scenario's state to -1
        HIP_TRY(h, hipMemset2DAsync(h->dKeepI + 256, sizeof(int32_t) * keepI, 0xFF, sizeof(int32_t), (size_t)N, st));
        h->keepOn = true;
    }
    return LMPC_OK;
}

int lmpc_simulate_device(lmpc_handle *h, int64_t N, int T, int nx, int nr, int nuprev, const double *F,
                         const double *G, double *x, const double *r, double *uprev, double *U_traj,
                         double *X_traj, int32_t *flag_min, int warm, void *stream) {
    if (!h) return LMPC_ERR_BADARG;
    const int nu = h->P.nout;
    if (N < 0 || T < 0 || nx <= 0 || nx > 32 || nr < 0 || nuprev < 0 || nuprev > nu || !F || !G || (N > 0 && !x) ||
        (nuprev > 0 && N > 0 && !uprev) || nx + nr + nuprev != h->P.nth)
        return fail(h, LMPC_ERR_BADARG, "lmpc_simulate_device: theta = [x; r; uprev] must match the handle "
                                        "(nx + nr + nuprev == nth, nout == nu, nx <= 32)");
    if (N == 0 || T == 0) return LMPC_OK;
    LMPC_ENTER_DEVICE(h);
    hipStream_t st = (hipStream_t)stream;
    { const int rce = api_ensure_sim(h, N); if (rce != LMPC_OK) return rce; }
    HIP_TRY(h, hipMemcpyAsync(h->simFG, F, sizeof(double) * nx * nx, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->simFG + nx * nx, G, sizeof(double) * nx * nu, hipMemcpyHostToDevice, st));
    if (X_traj) HIP_TRY(h, hipMemcpyAsync(X_traj, x, sizeof(double) * (size_t)N * nx, hipMemcpyDeviceToDevice, st));
    const unsigned grid = (unsigned)((N + 255) / 256);
    if (h->useWave && !h->avi && !h->waveProbed && N >= 4 * kProbe) {
        // fresh handle: the working-set sizes of the first step's problems decide how the loop's launches are shaped
        // (wave_probe) -- on the records of the leading scenarios, formed here and formed again with all the others below
        hipLaunchKernelGGL(form_theta_kernel<double>, dim3((unsigned)((kProbe + 255) / 256)), dim3(256), 0, st, h->simTheta, x, r,
                           uprev, nx, nr, nuprev, (long long)kProbe);
        HIP_TRY(h, hipGetLastError());
        const int rcp = api_wave_probe(h, h->simTheta, N, st);
        if (rcp != LMPC_OK) return rcp;
    }
    // theta = [x; r; uprev] is formed once; from then on every scenario's state lives in its record
    // (the scenario-asynchronous loop forms it in its first streaming pass)
    // (round 3: also on the wavefront-kernel path -- soft rows, many rows: the same streaming half on the handle's
    // screening pack, the wavefront kernel as the iterating half; binary64, no branch and bound)
    const size_t simLds = sizeof(double) * ((((size_t)h->P.m + 7) & ~(size_t)7) * (h->L.nthp + 2) + (size_t)kMaxSimU * h->L.nthp + kMaxSimU + 64 + 8 * kMaxSimU);
    // It is OPT-IN there ("sim_async" 2): measured on the benchmark class (pendulum N = 50, 2e5 scenarios x 100 steps,
    // tools/sim_bench.py) the rounds lose to the lock-step loop -- 4.9e8 against 5.3e8 scenario-steps/s from
    // closed-loop-visited starts, 1.17e9 against 1.31e9 with 90 % of the scenarios at rest: what costs the time is the
    // wavefront kernel on the transient's problems, which both loops run, and a round adds a launch and a host round
    // trip where the lock-step loop adds a 25 us screening pass.
    // (end of round 3: with RUN-AHEAD -- a scenario's consecutive steps that need iterations stay inside the wavefront
    // kernel, warm on the factor as it stands in LDS -- the rounds win on every workload but the six-slot problems:
    // 1.28e9 against 8.7e8 on that benchmark, 2.7e9 against 2.1e9 with 90 % at rest, 1.6e9 against 1.2e9 at 1e6
    // scenarios.  Default from then on whenever run-ahead applies: warm with "sim_keep_factor" 1, or cold.)
    const bool runAhead = h->simRunAhead && (!warm || h->simKeep) && (h->P.m + 63) / 64 <= kWaveRunAheadSlots;
    const bool waveAsync = h->useWave && (h->simAsync >= 2 || (h->simAsync >= 1 && runAhead)) && !h->bnb && nu <= kMaxSimU &&
                           nx <= 8 && h->P.nth <= 16 && api_wave_screens(h, N) && simLds <= 48 * 1024;
    const bool asyncLoop = waveAsync ||
                           (!h->useWave && h->simFused && h->simAsync && nu <= kMaxSimU && nx <= 8 && h->P.nth <= 16 && api_will_screen(h, N));
    if (!asyncLoop)
        hipLaunchKernelGGL(form_theta_kernel<double>, dim3(grid), dim3(256), 0, st, h->simTheta, x, r, uprev, nx, nr,
                           nuprev, (long long)N);
    // Lane / screening kernels: the kernel that finishes a problem also advances its scenario and
    // writes the next step's record into the other theta buffer (SimFuse) -- a closed-loop step is
    // the solve's two launches and nothing else
    // Scenario-asynchronous loop (lmpc_simrun_kernel.hpp): scenarios are independent, so each one runs ahead
    // in registers through its unconstrained steps and only the steps that need iterations go through the
    // iterating kernel, one round per such step.  The host reads the work-list counters after every
    // streaming pass (one stream synchronisation per round) and stops when nothing is queued any more.
    if (asyncLoop) {
        h->asyncX = x; h->asyncR = r; h->asyncUp = nuprev > 0 ? uprev : nullptr;
        HIP_TRY(h, hipMemcpyAsync(h->dC + h->L.oFG, F, sizeof(double) * nx * nx, hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipMemcpyAsync(h->dC + h->L.oFG + nx * nx, G, sizeof(double) * nx * nu, hipMemcpyHostToDevice, st));
        if (!h->simK) HIP_TRY(h, hipMalloc(&h->simK, sizeof(int32_t) * (size_t)h->simCap));
        HIP_TRY(h, hipMemsetAsync(h->simK, 0, sizeof(int32_t) * (size_t)N, st));
        if (warm) HIP_TRY(h, hipMemsetAsync(h->simAct, 0, sizeof(uint64_t) * (size_t)N * (size_t)h->P.words(), st));   // first step is cold
        const bool prof = h->prof;
        h->prof = false;
        h->asyncT = T;
        h->L.sim = SimFuse{h->simFG, h->simTheta, flag_min, nullptr, nx, nu, nr, nuprev, 0, h->simK, U_traj, X_traj,
                           (long long)N};
        h->keepOn = false; h->raWarm = warm != 0;
        if (h->useWave && runAhead && warm && T > 1 && wave_first_pass_cap(h, N) > 0) {
            // (a first pass at a smaller capacity is in sight: it writes every scenario's state out after each step, so
            // that a step which outgrows it restarts exactly where the step-synchronous loop would)
            const int rck = ensure_keep(h, N, st);
            if (rck != LMPC_OK) {                     // (leave the handle as an ordinary solve expects it)
                h->L.sim = SimFuse{}; h->waveSim = WaveSim{}; h->raWarm = false; h->asyncT = 0; h->prof = prof;
                h->asyncX = h->asyncR = h->asyncUp = nullptr;
                return rck;
            }
        }
        if (h->useWave) h->waveSim = WaveSim{h->simFG, h->simK, U_traj, X_traj, flag_min, nx, nu, nr, nuprev, (long long)N, -1, runAhead ? T : 0};
        constexpr int kBurst = 2;   // steps a scenario of a (short) work list may run ahead before it is parked
        const size_t setLen = (size_t)kShards * kCountStride;
        std::vector<int32_t> hc(3 * setLen);
        uint64_t *masks = warm ? h->simAct : nullptr;
        int rc = LMPC_OK;
        h->asyncListIn = h->asyncCntIn = nullptr;
        h->asyncCap = T + 1;
        h->asyncResetPark = false;
        const bool dbg = std::getenv("LMPC_DEBUG_SIM") != nullptr;
        // pass 0 runs every scenario up to its first step that needs iterations; then: solve that step for the
        // listed scenarios, let them run ahead a little (most meet the next such step at once: the transient),
        // park the ones that broke free; when the list has drained, run the parked ones on, compacted.
        bool drained = false;
        for (int pass = 0; pass <= 2 * T + 4 && rc == LMPC_OK; pass++) {
            h->asyncPhase = 1;
            rc = api_launch(h, N, h->simTheta, nullptr, nullptr, nullptr, masks, masks, st);
            if (rc != LMPC_OK) break;
            if (hipMemcpyAsync(hc.data(), h->dCount, sizeof(int32_t) * hc.size(), hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) { rc = fail(h, LMPC_ERR_HIP, "lmpc_simulate: reading the work-list counters"); break; }
            const size_t off = (size_t)(h->asyncCntNow - h->dCount);
            long long queued = 0, longest = 0, parked = 0, plongest = 0;
            for (int sh = 0; sh < kShards; sh++) {
                const long long q = hc[off + (size_t)sh * kCountStride], pk = h->asyncResetPark ? 0 : hc[2 * setLen + (size_t)sh * kCountStride];
                queued += q; parked += pk;
                longest = q > longest ? q : longest;
                plongest = pk > plongest ? pk : plongest;
            }
            if (dbg) std::fprintf(stderr, "lmpc sim pass %d: %lld scenarios queued, %lld parked\n", pass, queued, parked);
            h->asyncResetPark = false;
            if (queued > 0) {
                h->asyncPhase = 2;
                rc = api_launch(h, N, h->simTheta, nullptr, nullptr, nullptr, masks, masks, st);
                h->asyncListIn = h->asyncListOut; h->asyncCntIn = h->asyncCntNow; h->asyncMaxIn = longest;
                h->asyncCap = kBurst;
                // a few more rounds without asking: a list never grows from one round to the next, so the grid of
                // `longest` covers them, and a round on an empty list costs less than the host round trip it saves
                for (int e = 0; e < h->simBlind && rc == LMPC_OK; e++) {
                    h->asyncPhase = 1;
                    rc = api_launch(h, N, h->simTheta, nullptr, nullptr, nullptr, masks, masks, st);
                    if (rc != LMPC_OK) break;
                    h->asyncPhase = 2;
                    rc = api_launch(h, N, h->simTheta, nullptr, nullptr, nullptr, masks, masks, st);
                    h->asyncListIn = h->asyncListOut; h->asyncCntIn = h->asyncCntNow;
                }
            } else if (parked > 0) {
                // no iterating kernel ran, so nobody cleared the counter set the next pass writes
                if (hipMemsetAsync(h->dCount, 0, sizeof(int32_t) * 2 * setLen, st) != hipSuccess) { rc = fail(h, LMPC_ERR_HIP, "lmpc_simulate: clearing the work-list counters"); break; }
                h->asyncListIn = h->dList3; h->asyncCntIn = h->dCount + 2 * setLen; h->asyncMaxIn = plongest;
                h->asyncCap = T + 1;
                h->asyncResetPark = true;
            } else {
                drained = true;
                break;
            }
        }
        // the pass cap is generous (every scenario needs at most T rounds), but running into it with scenarios
        // still queued or parked must not look like success: their states and trajectory slots are unfinished
        if (rc == LMPC_OK && !drained)
            rc = fail(h, LMPC_ERR_HIP, "lmpc_simulate: the scenario-asynchronous loop hit its pass limit with scenarios "
                                       "still queued (lmpc_set_option(\"sim_async\", 0) runs the step-synchronous loop)");
        h->asyncPhase = 0;
        h->asyncListIn = h->asyncCntIn = nullptr;
        // the last streaming pass queued nothing, so no iterating kernel cleared the other counter set
        if (h->dCount) { hipMemsetAsync(h->dCount, 0, sizeof(int32_t) * 3 * kShards * kCountStride, st); h->countSet = 0; }
        h->L.sim = SimFuse{};
        h->waveSim = WaveSim{};
        h->keepOn = false; h->raWarm = false;
        h->prof = prof;
        if (rc != LMPC_OK) return rc;
        hipLaunchKernelGGL(unpack_theta_kernel, dim3(grid), dim3(256), 0, st, h->simTheta, x, nuprev > 0 ? uprev : nullptr,
                           nx, nr, nuprev, (long long)N);
        HIP_TRY(h, hipGetLastError());
        return LMPC_OK;
    }
    if (!h->useWave && h->simFused && nu <= kMaxSimU) {
        HIP_TRY(h, hipMemcpyAsync(h->dC + h->L.oFG, F, sizeof(double) * nx * nx, hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipMemcpyAsync(h->dC + h->L.oFG + nx * nx, G, sizeof(double) * nx * nu, hipMemcpyHostToDevice, st));
        if (!h->simTheta2) HIP_TRY(h, hipMalloc(&h->simTheta2, sizeof(double) * (size_t)h->simCap * h->P.nth));
        double *cur = h->simTheta, *nxt = h->simTheta2;
        int rc = LMPC_OK;
        for (int k = 0; k < T && rc == LMPC_OK; k++) {
            h->L.sim = SimFuse{h->simFG, nxt, flag_min, X_traj ? X_traj + (size_t)(k + 1) * N * nx : nullptr,
                               nx, nu, nr, nuprev, k == 0 ? 1 : 0};
            const uint64_t *wm = (warm && k > 0) ? h->simAct : nullptr;
            // no input trajectory asked for: the kernels write neither u nor the per-step flags
            rc = api_launch(h, N, cur, U_traj ? U_traj + (size_t)k * N * nu : nullptr, nullptr, nullptr,
                        warm ? h->simAct : nullptr, wm, st);
            std::swap(cur, nxt);
        }
        h->L.sim = SimFuse{};
        if (rc != LMPC_OK) return rc;
        hipLaunchKernelGGL(unpack_theta_kernel, dim3(grid), dim3(256), 0, st, cur, x, nuprev > 0 ? uprev : nullptr, nx,
                           nr, nuprev, (long long)N);
        HIP_TRY(h, hipGetLastError());
        return LMPC_OK;
    }
    // Wavefront path, warm: every scenario's final working set stays on the device WITH its factorisation, in its
    // order (2 x 64 + cap (cap - 1) / 2 reals and 320 ints per scenario: 17 KB at cap 64), so that a warm step
    // starts from the factor as it stands -- what DAQP_WARMSTART means in libdaqp, whose workspace is simply not
    // cleared between two calls (codegen/mpc_update_qp.c:44-54) -- instead of re-appending the rows of the mask one
    // by one.  Option "sim_keep_factor" 0 (or no memory for it): the mask-based warm start of the other paths.
    h->keepOn = false;
    if (h->useWave && !h->avi && warm && !h->bnb && h->simKeep && T > 1) {
        const int rck = ensure_keep(h, N, st);
        if (rck != LMPC_OK) return rck;
    }
    // Wavefront path with its screening pass in front: the plant step is fused into the three kernels of a step like
    // on the lane path -- the screening pass advances the scenarios it finishes (SimFuse), the wavefront kernel and its
    // slow path advance theirs in place (WaveSim with the step number) -- so a step is those launches and nothing
    // else: no plant kernel (16 us at 2e5 scenarios), no input / flag arrays in between.
    if (h->useWave && h->simFused && !h->bnb && nu <= kMaxSimU && nx <= 8 && api_wave_screens(h, N)) {
        HIP_TRY(h, hipMemcpyAsync(h->dC + h->L.oFG, F, sizeof(double) * nx * nx, hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipMemcpyAsync(h->dC + h->L.oFG + nx * nx, G, sizeof(double) * nx * nu, hipMemcpyHostToDevice, st));
        int rc = LMPC_OK;
        for (int k = 0; k < T && rc == LMPC_OK; k++) {
            h->L.sim = SimFuse{h->simFG, h->simTheta, flag_min, X_traj ? X_traj + (size_t)(k + 1) * N * nx : nullptr,
                               nx, nu, nr, nuprev, k == 0 ? 1 : 0};
            h->waveSim = WaveSim{h->simFG, nullptr, nullptr, X_traj, flag_min, nx, nu, nr, nuprev, (long long)N, k};
            const uint64_t *wm = (warm && k > 0) ? h->simAct : nullptr;
            rc = api_launch(h, N, h->simTheta, U_traj ? U_traj + (size_t)k * N * nu : nullptr, nullptr, nullptr,
                        warm ? h->simAct : nullptr, wm, st);
        }
        h->L.sim = SimFuse{};
        h->waveSim = WaveSim{};
        h->keepOn = false;
        if (rc != LMPC_OK) return rc;
        hipLaunchKernelGGL(unpack_theta_kernel, dim3(grid), dim3(256), 0, st, h->simTheta, x, nuprev > 0 ? uprev : nullptr,
                           nx, nr, nuprev, (long long)N);
        HIP_TRY(h, hipGetLastError());
        return LMPC_OK;
    }
    for (int k = 0; k < T; k++) {
        // warm start = previous step's final working set (reference codegen DAQP_WARMSTART,
        // codegen/mpc_update_qp.c:44-47); the first step is always cold
        const uint64_t *wm = (warm && k > 0) ? h->simAct : nullptr;
        int rc = api_launch(h, N, h->simTheta, h->simU, h->simFlag, nullptr, warm ? h->simAct : nullptr, wm, st);
        if (rc != LMPC_OK) { h->keepOn = false; return rc; }
        const bool last = k == T - 1;
        hipLaunchKernelGGL(plant_theta_kernel<double>, dim3(grid), dim3(256), 0, st, h->simTheta, h->P.nth, nr, h->simU,
                           h->simFlag, h->simFG, nx, nu, nuprev,
                           X_traj ? X_traj + (size_t)(k + 1) * N * nx : nullptr,
                           U_traj ? U_traj + (size_t)k * N * nu : nullptr, flag_min, k == 0 ? 1 : 0,
                           last ? x : nullptr, (last && nuprev > 0) ? uprev : nullptr, (long long)N);
        if (hipGetLastError() != hipSuccess) { h->keepOn = false; return fail(h, LMPC_ERR_HIP, "lmpc_simulate_device: plant step launch"); }
    }
    h->keepOn = false;
    return LMPC_OK;
}

int lmpc_simulate_f32_device(lmpc_handle *h, int64_t N, int T, int nx, int nr, int nuprev, const double *F,
                             const double *G, float *x, const float *r, float *uprev, float *U_traj,
                             float *X_traj, int32_t *flag_min, int warm, void *stream) {
    if (!h) return LMPC_ERR_BADARG;
    const int nu = h->P.nout;
    if (N < 0 || T < 0 || nx <= 0 || nx > 32 || nr < 0 || nuprev < 0 || nuprev > nu || nu > 64 || !F || !G ||
        (N > 0 && !x) || (nuprev > 0 && N > 0 && !uprev) || nx + nr + nuprev != h->P.nth)
        return fail(h, LMPC_ERR_BADARG, "lmpc_simulate_f32_device: theta = [x; r; uprev] must match the handle "
                                        "(nx + nr + nuprev == nth, nout == nu, nx <= 32)");
    if (N == 0 || T == 0) return LMPC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(h, LMPC_ERR_NOGPU, "lmpc: no HIP device available (this library has no CPU path)");
    LMPC_ENTER_DEVICE(h);
    int rc = api_ensure_f32(h);
    if (rc != LMPC_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    { const int rce = api_ensure_sim(h, N); if (rce != LMPC_OK) return rce; }
    float *dTh = reinterpret_cast<float *>(h->simTheta), *dU = reinterpret_cast<float *>(h->simU),
          *dFG = reinterpret_cast<float *>(h->simFG);
    std::vector<float> fg((size_t)nx * nx + (size_t)nx * nu);       // the plant rounded to binary32, like the pack
    for (int i = 0; i < nx * nx; i++) fg[i] = (float)F[i];
    for (int i = 0; i < nx * nu; i++) fg[(size_t)nx * nx + i] = (float)G[i];
    HIP_TRY(h, hipMemcpy(dFG, fg.data(), sizeof(float) * fg.size(), hipMemcpyHostToDevice));
    if (X_traj) HIP_TRY(h, hipMemcpyAsync(X_traj, x, sizeof(float) * (size_t)N * nx, hipMemcpyDeviceToDevice, st));
    const unsigned grid = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(form_theta_kernel<float>, dim3(grid), dim3(256), 0, st, dTh, x, r, uprev, nx, nr, nuprev,
                       (long long)N);
    for (int k = 0; k < T; k++) {
        const uint64_t *wm = (warm && k > 0 && !h->bnb) ? h->simAct : nullptr;
        rc = api_launch_wave_f32(h, h->dCwf, N, dTh, dU, h->simFlag, nullptr, (warm && !h->bnb) ? h->simAct : nullptr,
                                  wm, st);
        if (rc != LMPC_OK) return rc;
        const bool last = k == T - 1;
        hipLaunchKernelGGL(plant_theta_kernel<float>, dim3(grid), dim3(256), 0, st, dTh, h->P.nth, nr, dU, h->simFlag,
                           dFG, nx, nu, nuprev, X_traj ? X_traj + (size_t)(k + 1) * N * nx : nullptr,
                           U_traj ? U_traj + (size_t)k * N * nu : nullptr, flag_min, k == 0 ? 1 : 0,
                           last ? x : nullptr, (last && nuprev > 0) ? uprev : nullptr, (long long)N);
        HIP_TRY(h, hipGetLastError());
    }
    return LMPC_OK;
}

int lmpc_simulate_f32(lmpc_handle *h, int64_t N, int T, int nx, int nr, int nuprev, const double *F, const double *G,
                      float *x, const float *r, float *uprev, float *U_traj, float *X_traj, int32_t *flag_min,
                      int warm) {
    if (!h) return LMPC_ERR_BADARG;
    if (N <= 0 || T <= 0) return N < 0 || T < 0 ? LMPC_ERR_BADARG : LMPC_OK;
    LMPC_ENTER_DEVICE(h);
    const int nu = h->P.nout;
    float *dx = nullptr, *dr = nullptr, *du = nullptr, *dU = nullptr, *dX = nullptr;
    int32_t *df = nullptr;
    auto cleanup = [&]() { hipFree(dx); hipFree(dr); hipFree(du); hipFree(dU); hipFree(dX); hipFree(df); };
#define SIMF_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); \
        return fail(h, LMPC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); } } while (0)
    SIMF_TRY(hipMalloc(&dx, sizeof(float) * (size_t)N * nx));
    SIMF_TRY(hipMemcpy(dx, x, sizeof(float) * (size_t)N * nx, hipMemcpyHostToDevice));
    if (r && nr > 0) {
        SIMF_TRY(hipMalloc(&dr, sizeof(float) * (size_t)N * nr));
        SIMF_TRY(hipMemcpy(dr, r, sizeof(float) * (size_t)N * nr, hipMemcpyHostToDevice));
    }
    if (nuprev > 0) {                                    // NULL = zeros, as in lmpc_simulate
        SIMF_TRY(hipMalloc(&du, sizeof(float) * (size_t)N * nuprev));
        if (uprev) SIMF_TRY(hipMemcpy(du, uprev, sizeof(float) * (size_t)N * nuprev, hipMemcpyHostToDevice));
        else SIMF_TRY(hipMemset(du, 0, sizeof(float) * (size_t)N * nuprev));
    }
    if (U_traj) SIMF_TRY(hipMalloc(&dU, sizeof(float) * (size_t)T * N * nu));
    if (X_traj) SIMF_TRY(hipMalloc(&dX, sizeof(float) * (size_t)(T + 1) * N * nx));
    if (flag_min) SIMF_TRY(hipMalloc(&df, sizeof(int32_t) * (size_t)N));
    int rc = lmpc_simulate_f32_device(h, N, T, nx, nr, nuprev, F, G, dx, dr, du, dU, dX, df, warm, nullptr);
    if (rc != LMPC_OK) { cleanup(); return rc; }
    SIMF_TRY(hipMemcpy(x, dx, sizeof(float) * (size_t)N * nx, hipMemcpyDeviceToHost));
    if (du && uprev) SIMF_TRY(hipMemcpy(uprev, du, sizeof(float) * (size_t)N * nuprev, hipMemcpyDeviceToHost));
    if (dU) SIMF_TRY(hipMemcpy(U_traj, dU, sizeof(float) * (size_t)T * N * nu, hipMemcpyDeviceToHost));
    if (dX) SIMF_TRY(hipMemcpy(X_traj, dX, sizeof(float) * (size_t)(T + 1) * N * nx, hipMemcpyDeviceToHost));
    if (df) SIMF_TRY(hipMemcpy(flag_min, df, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost));
#undef SIMF_TRY
    cleanup();
    return LMPC_OK;
}

namespace {
ThetaBlock to_block(const lmpc_block *b) {
    ThetaBlock t{nullptr, 0, 0, 1, 0, 0};
    if (b) { t.src = b->src; t.stride = b->stride; t.w = b->w; t.T = b->T > 0 ? b->T : 1; t.k0 = b->k0; t.H = b->H; }
    return t;
}
}  // namespace

int lmpc_form_parameter_device(lmpc_handle *h, int64_t N, double *theta, const double *x, int nx,
                               const lmpc_block *r, const lmpc_block *d, const double *uprev, int nuprev,
                               const lmpc_block *p, void *stream) {
    if (!h) return LMPC_ERR_BADARG;
    const ThetaBlock br = to_block(r), bd = to_block(d), bp = to_block(p);
    if (N < 0 || nx < 0 || nuprev < 0 || br.w < 0 || bd.w < 0 || bp.w < 0 || br.H < 0 || bd.H < 0 || bp.H < 0 ||
        (N > 0 && (!theta || (nx > 0 && !x))))
        return fail(h, LMPC_ERR_BADARG, "lmpc_form_parameter_device: NULL array or negative size");
    if (nx + br.width() + bd.width() + nuprev + bp.width() != h->P.nth)
        return fail(h, LMPC_ERR_BADARG, "lmpc_form_parameter_device: blocks do not add up to the handle's nth = " +
                                            std::to_string(h->P.nth));
    if (N == 0 || h->P.nth == 0) return LMPC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(h, LMPC_ERR_NOGPU, "lmpc: no HIP device available (this library has no CPU path)");
    LMPC_ENTER_DEVICE(h);
    const long long total = (long long)N * h->P.nth;
    hipLaunchKernelGGL(form_parameter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       theta, x, nx, br, bd, uprev, nuprev, bp, (long long)N);
    HIP_TRY(h, hipGetLastError());
    return LMPC_OK;
}

int lmpc_simulate_ref_device(lmpc_handle *h, int64_t N, int T, int nx, const lmpc_block *r, int nuprev,
                             const double *F, const double *G, double *x, double *uprev, double *U_traj,
                             double *X_traj, int32_t *flag_min, int warm, void *stream) {
    if (!h) return LMPC_ERR_BADARG;
    const int nu = h->P.nout;
    ThetaBlock br = to_block(r);
    if (N < 0 || T < 0 || nx <= 0 || nx > 32 || nuprev < 0 || nuprev > nu || !F || !G || (N > 0 && !x) ||
        (nuprev > 0 && N > 0 && !uprev) || br.w < 0 || br.H < 0 || nx + br.width() + nuprev != h->P.nth)
        return fail(h, LMPC_ERR_BADARG, "lmpc_simulate_ref_device: theta = [x; r-block; uprev] must match the handle "
                                        "(nx + width(r) + nuprev == nth, nout == nu, nx <= 32)");
    if (N == 0 || T == 0) return LMPC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(h, LMPC_ERR_NOGPU, "lmpc: no HIP device available (this library has no CPU path)");
    LMPC_ENTER_DEVICE(h);
    hipStream_t st = (hipStream_t)stream;
    { const int rce = api_ensure_sim(h, N); if (rce != LMPC_OK) return rce; }
    HIP_TRY(h, hipMemcpyAsync(h->simFG, F, sizeof(double) * nx * nx, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->simFG + nx * nx, G, sizeof(double) * nx * nu, hipMemcpyHostToDevice, st));
    if (X_traj) HIP_TRY(h, hipMemcpyAsync(X_traj, x, sizeof(double) * (size_t)N * nx, hipMemcpyDeviceToDevice, st));
    const unsigned grid = (unsigned)((N + 255) / 256);
    const long long total = (long long)N * h->P.nth;
    const ThetaBlock none{nullptr, 0, 0, 1, 0, 0};
    for (int k = 0; k < T; k++) {
        br.k0 = br.H > 0 ? k + 1 : k;                 // simulation.jl:101 get_preview(rs, k, Np) / rs[:,k]
        hipLaunchKernelGGL(form_parameter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           h->simTheta, x, nx, br, none, uprev, nuprev, none, (long long)N);
        const uint64_t *wm = (warm && k > 0) ? h->simAct : nullptr;
        int rc = api_launch(h, N, h->simTheta, h->simU, h->simFlag, nullptr, warm ? h->simAct : nullptr, wm, st);
        if (rc != LMPC_OK) return rc;
        {
#define LMPC_PK(NX) hipLaunchKernelGGL(plant_kernel<NX>, dim3(grid), dim3(256), 0, st, x, uprev, h->simU, h->simFlag, h->simFG, nx, \
                           nu, nuprev, X_traj ? X_traj + (size_t)(k + 1) * N * nx : nullptr, \
                           U_traj ? U_traj + (size_t)k * N * nu : nullptr, flag_min, k == 0 ? 1 : 0, (long long)N)
            switch (nx) {
                case 1: LMPC_PK(1); break; case 2: LMPC_PK(2); break; case 3: LMPC_PK(3); break; case 4: LMPC_PK(4); break;
                case 5: LMPC_PK(5); break; case 6: LMPC_PK(6); break; case 7: LMPC_PK(7); break; case 8: LMPC_PK(8); break;
                default: LMPC_PK(0); break;
            }
#undef LMPC_PK
        }
        HIP_TRY(h, hipGetLastError());
    }
    return LMPC_OK;
}

int lmpc_set_parameter_layout(lmpc_handle *h, const lmpc_param_layout *l) {
    if (!h || !l) return LMPC_ERR_BADARG;
    if (l->n_state < 0 || l->n_reference < 0 || l->n_disturbance < 0 || l->n_control_prev < 0 ||
        l->n_affine_parameter < 0 || l->n_preview_horizon < 0 || l->n_control_prev > h->P.nout ||
        (l->n_preview_horizon > 0 && !l->traj2setpoint))
        return fail(h, LMPC_ERR_BADARG, "lmpc_set_parameter_layout: negative size, n_control_prev > nout, or "
                                        "n_preview_horizon > 0 without traj2setpoint");
    if (l->n_state + l->n_reference + l->n_disturbance + l->n_control_prev + l->n_affine_parameter != h->P.nth)
        return fail(h, LMPC_ERR_BADARG, "lmpc_set_parameter_layout: blocks do not add up to the handle's nth = " +
                                            std::to_string(h->P.nth));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(h, LMPC_ERR_NOGPU, "lmpc: no HIP device available (this library has no CPU path)");
    LMPC_ENTER_DEVICE(h);
    hipFree(h->ccT2S);
    h->ccT2S = nullptr;
    if (l->n_preview_horizon > 0 && l->n_reference > 0) {
        const size_t cnt = (size_t)l->n_reference * l->n_reference * l->n_preview_horizon;
        HIP_TRY(h, hipMalloc(&h->ccT2S, sizeof(double) * cnt));
        HIP_TRY(h, hipMemcpy(h->ccT2S, l->traj2setpoint, sizeof(double) * cnt, hipMemcpyHostToDevice));
    }
    h->ccNx = l->n_state; h->ccNr = l->n_reference; h->ccNd = l->n_disturbance; h->ccNup = l->n_control_prev;
    h->ccNp = l->n_affine_parameter; h->ccNph = l->n_reference > 0 ? l->n_preview_horizon : 0;
    h->ccWarmN = -1;
    return LMPC_OK;
}

int lmpc_compute_control_device(lmpc_handle *h, int64_t N, double *control, const double *state,
                                const double *reference, const double *disturbance,
                                const double *affine_parameter, int32_t *exitflag, int warm, void *stream) {
    if (!h) return LMPC_ERR_BADARG;
    if (h->ccNx < 0) return fail(h, LMPC_ERR_BADARG, "lmpc_compute_control: call lmpc_set_parameter_layout first");
    if (N < 0 || (N > 0 && (!control || (h->ccNx > 0 && !state))))
        return fail(h, LMPC_ERR_BADARG, "lmpc_compute_control: NULL control/state or negative N");
    if (N == 0) return LMPC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(h, LMPC_ERR_NOGPU, "lmpc: no HIP device available (this library has no CPU path)");
    LMPC_ENTER_DEVICE(h);
    hipStream_t st = (hipStream_t)stream;
    const size_t w = (size_t)h->P.words();
    if (N > h->ccCap) {
        hipFree(h->ccTheta); hipFree(h->ccAct); hipFree(h->ccFlag);
        h->ccTheta = nullptr; h->ccAct = nullptr; h->ccFlag = nullptr; h->ccCap = 0; h->ccWarmN = -1;
        HIP_TRY(h, hipMalloc(&h->ccTheta, sizeof(double) * (size_t)N * (h->P.nth ? h->P.nth : 1)));
        HIP_TRY(h, hipMalloc(&h->ccAct, sizeof(uint64_t) * (size_t)N * w));
        HIP_TRY(h, hipMalloc(&h->ccFlag, sizeof(int32_t) * (size_t)N));
        h->ccCap = N;
    }
    // lane / screening path without reference condensation: the screening kernel assembles theta from the
    // five arrays itself (GatherArgs) and hands the records of the problems that need iterations to the
    // iterating kernel through ccTheta -- no theta buffer is written or read for the others
    if (api_will_screen(h, N) && h->ccNph == 0 && h->ccFused) {
        h->L.gat = GatherArgs{state, reference, disturbance, control, affine_parameter, h->ccTheta,
                              h->ccNx, h->ccNr, h->ccNd, h->ccNup, h->ccNp, h->P.nout};
        const bool use_warm_g = warm && h->ccWarmN == N;
        int rcg = api_launch(h, N, h->ccTheta, control, exitflag ? exitflag : h->ccFlag, nullptr, warm ? h->ccAct : nullptr,
                         use_warm_g ? h->ccAct : nullptr, st);
        h->L.gat = GatherArgs{};
        if (rcg != LMPC_OK) return rcg;
        h->ccWarmN = warm ? N : -1;
        return LMPC_OK;
    }
    const long long total = (long long)N * h->P.nth;
    if (total > 0) {
        hipLaunchKernelGGL(update_parameter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           h->ccTheta, control, h->P.nout, state, h->ccNx, reference, h->ccNr, h->ccNph, h->ccT2S,
                           disturbance, h->ccNd, h->ccNup, affine_parameter, h->ccNp, (long long)N);
        HIP_TRY(h, hipGetLastError());
    }
    // DAQP_WARMSTART build (codegen/mpc_update_qp.c:44-47): the working sets the previous call ended
    // with are the next call's starting point; otherwise every call starts cold
    const bool use_warm = warm && h->ccWarmN == N && !h->bnb;
    int rc = api_launch(h, N, h->ccTheta, control, exitflag ? exitflag : h->ccFlag, nullptr,
                    (warm && !h->bnb) ? h->ccAct : nullptr, use_warm ? h->ccAct : nullptr, st);
    if (rc != LMPC_OK) return rc;
    h->ccWarmN = (warm && !h->bnb) ? N : -1;
    return LMPC_OK;
}

int lmpc_compute_control(lmpc_handle *h, int64_t N, double *control, const double *state,
                         const double *reference, const double *disturbance, const double *affine_parameter,
                         int32_t *exitflag, int warm) {
    if (!h) return LMPC_ERR_BADARG;
    if (h->ccNx < 0) return fail(h, LMPC_ERR_BADARG, "lmpc_compute_control: call lmpc_set_parameter_layout first");
    if (N < 0 || (N > 0 && (!control || (h->ccNx > 0 && !state))))
        return fail(h, LMPC_ERR_BADARG, "lmpc_compute_control: NULL control/state or negative N");
    if (N == 0) return LMPC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(h, LMPC_ERR_NOGPU, "lmpc: no HIP device available (this library has no CPU path)");
    LMPC_ENTER_DEVICE(h);
    const int nu = h->P.nout;
    const size_t wr = (size_t)h->ccNr * (h->ccNph > 0 ? h->ccNph : 1);
    // one staging block per handle, kept between calls (a closed loop calls this once per time step):
    // [control | state | reference | disturbance | parameter] doubles, then the flags
    const size_t per = (size_t)nu + h->ccNx + wr + h->ccNd + h->ccNp;
    if ((size_t)N * per * sizeof(double) <= (size_t)64 * 1024) {
        // a handful of problems -- the generated controller's call, one state per time step (mpc_update_qp.c:29-54): the
        // block lives in MAPPED host memory, the kernels read the arguments and write control and flags straight
        // through it, the host waits once for the handle's own stream (seven blocking copies before: ~100 us a call)
        const size_t need = ((size_t)N * per * sizeof(double) + 63 & ~(size_t)63) + sizeof(int32_t) * (size_t)N + 4096;
        if (need > h->ccMapBytes) {
            if (h->ccMapHost) { (void)hipDeviceSynchronize(); (void)hipHostFree(h->ccMapHost); }
            h->ccMapHost = h->ccMapDev = nullptr; h->ccMapBytes = 0;
            char *hp = nullptr;
            HIP_TRY(h, hipHostMalloc(reinterpret_cast<void **>(&hp), need, hipHostMallocMapped));
            if (hipHostGetDevicePointer(reinterpret_cast<void **>(&h->ccMapDev), hp, 0) != hipSuccess) {
                (void)hipHostFree(hp); h->ccMapDev = nullptr;
                return fail(h, LMPC_ERR_HIP, "lmpc_compute_control: mapped block");
            }
            h->ccMapHost = hp; h->ccMapBytes = need;
        }
        if (!h->oneStream) HIP_TRY(h, hipStreamCreateWithFlags(&h->oneStream, hipStreamNonBlocking));
        size_t off = 0;
        auto put = [&](const double *src, size_t w) -> double * {
            if (!src || w == 0) return nullptr;
            std::memcpy(h->ccMapHost + off, src, sizeof(double) * (size_t)N * w);
            double *d = reinterpret_cast<double *>(h->ccMapDev + off);
            off += sizeof(double) * (size_t)N * w;
            return d;
        };
        double *dc = put(control, (size_t)nu), *ds = put(state, (size_t)h->ccNx), *dr = put(reference, wr),
               *dd = put(disturbance, (size_t)h->ccNd), *dp = put(affine_parameter, (size_t)h->ccNp);
        const size_t oF = (off + 63) & ~(size_t)63;
        int32_t *hf = reinterpret_cast<int32_t *>(h->ccMapHost + oF);
        for (int64_t i = 0; i < N; i++) hf[i] = LMPC_EXIT_UNFINISHED;
        const int rc = lmpc_compute_control_device(h, N, dc, ds, dr, dd, dp, reinterpret_cast<int32_t *>(h->ccMapDev + oF), warm,
                                                   h->oneStream);
        const hipError_t es = hipStreamSynchronize(h->oneStream);
        if (rc != LMPC_OK) return rc;
        if (es != hipSuccess) return fail(h, LMPC_ERR_HIP, std::string("lmpc_compute_control: ") + hipGetErrorString(es));
        const int rcf = check_fast_err(h);
        if (rcf != LMPC_OK) return rcf;
        std::memcpy(control, h->ccMapHost, sizeof(double) * (size_t)N * nu);
        if (exitflag) std::memcpy(exitflag, hf, sizeof(int32_t) * (size_t)N);
        return LMPC_OK;
    }
    if (N > h->ccStageCap || per > h->ccStagePer) {
        hipFree(h->ccStage); hipFree(h->ccStageFlag);       // (ccObsScratch is another entry point's buffer)
        h->ccStage = nullptr; h->ccStageFlag = nullptr; h->ccStageCap = 0; h->ccStagePer = 0;
        HIP_TRY(h, hipMalloc(&h->ccStage, sizeof(double) * (size_t)N * per));
        HIP_TRY(h, hipMalloc(&h->ccStageFlag, sizeof(int32_t) * (size_t)N));
        h->ccStageCap = N; h->ccStagePer = per;
    }
    double *cur = h->ccStage;
    auto up = [&](const double *src, size_t w, double **dst) -> hipError_t {
        *dst = nullptr;
        if (!src || w == 0) return hipSuccess;
        *dst = cur;
        cur += (size_t)N * w;
        return hipMemcpy(*dst, src, sizeof(double) * (size_t)N * w, hipMemcpyHostToDevice);
    };
    double *dc, *ds, *dr, *dd, *dp;
    HIP_TRY(h, up(control, (size_t)nu, &dc));
    HIP_TRY(h, up(state, (size_t)h->ccNx, &ds));
    HIP_TRY(h, up(reference, wr, &dr));
    HIP_TRY(h, up(disturbance, (size_t)h->ccNd, &dd));
    HIP_TRY(h, up(affine_parameter, (size_t)h->ccNp, &dp));
    int rc = lmpc_compute_control_device(h, N, dc, ds, dr, dd, dp, h->ccStageFlag, warm, nullptr);
    if (rc != LMPC_OK) return rc;
    HIP_TRY(h, hipMemcpy(control, dc, sizeof(double) * (size_t)N * nu, hipMemcpyDeviceToHost));
    if (exitflag) HIP_TRY(h, hipMemcpy(exitflag, h->ccStageFlag, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost));
    return LMPC_OK;
}

int lmpc_compute_control_observer_device(lmpc_handle *h, int64_t N, double *control, const double *observer_state,
                                         int n_measured_disturbance, const double *reference,
                                         const double *measured_disturbance, const double *affine_parameter,
                                         int32_t *exitflag, int warm, void *stream) {
    if (!h) return LMPC_ERR_BADARG;
    if (h->ccNx < 0) return fail(h, LMPC_ERR_BADARG, "lmpc_compute_control_observer: call lmpc_set_parameter_layout first");
    const int ndm = n_measured_disturbance, ndo = h->ccNd - ndm;
    if (N < 0 || ndm < 0 || ndo < 0 || (N > 0 && (!control || !observer_state)))
        return fail(h, LMPC_ERR_BADARG, "lmpc_compute_control_observer: NULL array, negative N, or more measured "
                                        "disturbances than the layout's n_disturbance");
    if (N == 0) return LMPC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(h, LMPC_ERR_NOGPU, "lmpc: no HIP device available (this library has no CPU path)");
    LMPC_ENTER_DEVICE(h);
    const size_t per = (size_t)h->ccNx + h->ccNd;
    if (N > h->ccObsCap) {
        hipFree(h->ccObsScratch);
        h->ccObsScratch = nullptr; h->ccObsCap = 0;
        HIP_TRY(h, hipMalloc(&h->ccObsScratch, sizeof(double) * (size_t)N * (per ? per : 1)));
        h->ccObsCap = N;
    }
    double *st_ = h->ccObsScratch, *di_ = h->ccObsScratch + (size_t)N * h->ccNx;
    const long long total = (long long)N * (long long)per;
    if (total > 0) {
        hipLaunchKernelGGL(split_observer_state_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, st_, di_, observer_state, measured_disturbance, h->ccNx, ndm, ndo,
                           (long long)N);
        HIP_TRY(h, hipGetLastError());
    }
    return lmpc_compute_control_device(h, N, control, st_, reference, h->ccNd > 0 ? di_ : nullptr, affine_parameter,
                                       exitflag, warm, stream);
}

int lmpc_set_observer(lmpc_handle *h, const lmpc_observer *o) {
    if (!h || !o) return LMPC_ERR_BADARG;
    if (o->n_state <= 0 || o->n_state > 32 || o->n_control < 0 || o->n_disturbance < 0 || o->n_measurement <= 0 ||
        !o->plant_dynamics || !o->measurement_function || !o->k_transpose)
        return fail(h, LMPC_ERR_BADARG, "lmpc_set_observer: sizes (1 <= n_state <= 32, n_measurement >= 1) or NULL array");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(h, LMPC_ERR_NOGPU, "lmpc: no HIP device available (this library has no CPU path)");
    LMPC_ENTER_DEVICE(h);
    const size_t nd_ = (size_t)o->n_state * (1 + o->n_state + o->n_control + o->n_disturbance);
    const size_t nm_ = (size_t)o->n_measurement * (1 + o->n_state + o->n_disturbance);
    const size_t nk_ = (size_t)o->n_measurement * o->n_state;
    hipFree(h->obsC);
    h->obsC = nullptr;
    HIP_TRY(h, hipMalloc(&h->obsC, sizeof(double) * (nd_ + nm_ + nk_)));
    HIP_TRY(h, hipMemcpy(h->obsC, o->plant_dynamics, sizeof(double) * nd_, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->obsC + nd_, o->measurement_function, sizeof(double) * nm_, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->obsC + nd_ + nm_, o->k_transpose, sizeof(double) * nk_, hipMemcpyHostToDevice));
    h->obsNx = o->n_state; h->obsNu = o->n_control; h->obsNd = o->n_disturbance; h->obsNy = o->n_measurement;
    return LMPC_OK;
}

int lmpc_predict_state_device(lmpc_handle *h, int64_t N, double *state, const double *control,
                              const double *disturbance, void *stream) {
    if (!h) return LMPC_ERR_BADARG;
    if (!h->obsC) return fail(h, LMPC_ERR_BADARG, "lmpc_predict_state: call lmpc_set_observer first");
    if (N < 0 || (N > 0 && (!state || (h->obsNu > 0 && !control))))
        return fail(h, LMPC_ERR_BADARG, "lmpc_predict_state: NULL state/control or negative N");
    if (N == 0) return LMPC_OK;
    LMPC_ENTER_DEVICE(h);
#define LMPC_PS(NX) hipLaunchKernelGGL(predict_state_kernel<NX>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, \
        (hipStream_t)stream, state, control, disturbance, h->obsC, h->obsNx, h->obsNu, h->obsNd, (long long)N)
    switch (h->obsNx) {
        case 1: LMPC_PS(1); break; case 2: LMPC_PS(2); break; case 3: LMPC_PS(3); break; case 4: LMPC_PS(4); break;
        case 5: LMPC_PS(5); break; case 6: LMPC_PS(6); break; case 7: LMPC_PS(7); break; case 8: LMPC_PS(8); break;
        default: LMPC_PS(0); break;
    }
#undef LMPC_PS
    HIP_TRY(h, hipGetLastError());
    return LMPC_OK;
}

int lmpc_correct_state_device(lmpc_handle *h, int64_t N, double *state, const double *measurement,
                              const double *disturbance, void *stream) {
    if (!h) return LMPC_ERR_BADARG;
    if (!h->obsC) return fail(h, LMPC_ERR_BADARG, "lmpc_correct_state: call lmpc_set_observer first");
    if (N < 0 || (N > 0 && (!state || !measurement)))
        return fail(h, LMPC_ERR_BADARG, "lmpc_correct_state: NULL state/measurement or negative N");
    if (N == 0) return LMPC_OK;
    LMPC_ENTER_DEVICE(h);
    const size_t nd_ = (size_t)h->obsNx * (1 + h->obsNx + h->obsNu + h->obsNd);
    const size_t nm_ = (size_t)h->obsNy * (1 + h->obsNx + h->obsNd);
#define LMPC_CS(NX) hipLaunchKernelGGL(correct_state_kernel<NX>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, \
        (hipStream_t)stream, state, measurement, disturbance, h->obsC + nd_, h->obsC + nd_ + nm_, h->obsNx, \
        h->obsNy, h->obsNd, (long long)N)
    switch (h->obsNx) {
        case 1: LMPC_CS(1); break; case 2: LMPC_CS(2); break; case 3: LMPC_CS(3); break; case 4: LMPC_CS(4); break;
        case 5: LMPC_CS(5); break; case 6: LMPC_CS(6); break; case 7: LMPC_CS(7); break; case 8: LMPC_CS(8); break;
        default: LMPC_CS(0); break;
    }
#undef LMPC_CS
    HIP_TRY(h, hipGetLastError());
    return LMPC_OK;
}

namespace {
// host-pointer wrapper shared by predict / correct: state in/out, one input array, optional disturbance
int observer_host(lmpc_handle *h, int64_t N, double *state, const double *in, int win, const double *dist, bool predict) {
    if (!h) return LMPC_ERR_BADARG;
    if (!h->obsC) return fail(h, LMPC_ERR_BADARG, "lmpc observer: call lmpc_set_observer first");
    if (N < 0 || (N > 0 && (!state || (win > 0 && !in)))) return fail(h, LMPC_ERR_BADARG, "lmpc observer: NULL array or negative N");
    if (N == 0) return LMPC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(h, LMPC_ERR_NOGPU, "lmpc: no HIP device available (this library has no CPU path)");
    LMPC_ENTER_DEVICE(h);
    double *ds = nullptr, *di = nullptr, *dd = nullptr;
    auto cleanup = [&]() { hipFree(ds); hipFree(di); hipFree(dd); };
#define OB_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); \
        return fail(h, LMPC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); } } while (0)
    const size_t nx = (size_t)h->obsNx, nd = (size_t)h->obsNd;
    OB_TRY(hipMalloc(&ds, sizeof(double) * N * nx));
    OB_TRY(hipMemcpy(ds, state, sizeof(double) * N * nx, hipMemcpyHostToDevice));
    if (win > 0) {
        OB_TRY(hipMalloc(&di, sizeof(double) * N * win));
        OB_TRY(hipMemcpy(di, in, sizeof(double) * N * win, hipMemcpyHostToDevice));
    }
    if (dist && nd > 0) {
        OB_TRY(hipMalloc(&dd, sizeof(double) * N * nd));
        OB_TRY(hipMemcpy(dd, dist, sizeof(double) * N * nd, hipMemcpyHostToDevice));
    }
    int rc = predict ? lmpc_predict_state_device(h, N, ds, di, dd, nullptr) : lmpc_correct_state_device(h, N, ds, di, dd, nullptr);
    if (rc != LMPC_OK) { cleanup(); return rc; }
    OB_TRY(hipMemcpy(state, ds, sizeof(double) * N * nx, hipMemcpyDeviceToHost));
#undef OB_TRY
    cleanup();
    return LMPC_OK;
}
}  // namespace

int lmpc_predict_state(lmpc_handle *h, int64_t N, double *state, const double *control, const double *disturbance) {
    return observer_host(h, N, state, control, h ? h->obsNu : 0, disturbance, true);
}

int lmpc_correct_state(lmpc_handle *h, int64_t N, double *state, const double *measurement, const double *disturbance) {
    return observer_host(h, N, state, measurement, h ? h->obsNy : 0, disturbance, false);
}

int lmpc_simulate(lmpc_handle *h, int64_t N, int T, int nx, int nr, int nuprev, const double *F, const double *G,
                  double *x, const double *r, double *uprev, double *U_traj, double *X_traj, int32_t *flag_min,
                  int warm) {
    if (!h) return LMPC_ERR_BADARG;
    if (N <= 0 || T <= 0) return N < 0 || T < 0 ? LMPC_ERR_BADARG : LMPC_OK;
    LMPC_ENTER_DEVICE(h);
    const int nu = h->P.nout;
    double *dx = nullptr, *dr = nullptr, *du = nullptr, *dU = nullptr, *dX = nullptr;
    int32_t *df = nullptr;
    auto cleanup = [&]() { hipFree(dx); hipFree(dr); hipFree(du); hipFree(dU); hipFree(dX); hipFree(df); };
#define SIM_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); \
        return fail(h, LMPC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); } } while (0)
    SIM_TRY(hipMalloc(&dx, sizeof(double) * (size_t)N * nx));
    SIM_TRY(hipMemcpy(dx, x, sizeof(double) * (size_t)N * nx, hipMemcpyHostToDevice));
    if (r && nr > 0) {
        SIM_TRY(hipMalloc(&dr, sizeof(double) * (size_t)N * nr));
        SIM_TRY(hipMemcpy(dr, r, sizeof(double) * (size_t)N * nr, hipMemcpyHostToDevice));
    }
    if (nuprev > 0) {
        SIM_TRY(hipMalloc(&du, sizeof(double) * (size_t)N * nuprev));
        if (uprev) SIM_TRY(hipMemcpy(du, uprev, sizeof(double) * (size_t)N * nuprev, hipMemcpyHostToDevice));
        else SIM_TRY(hipMemset(du, 0, sizeof(double) * (size_t)N * nuprev));
    }
    if (U_traj) SIM_TRY(hipMalloc(&dU, sizeof(double) * (size_t)T * N * nu));
    if (X_traj) SIM_TRY(hipMalloc(&dX, sizeof(double) * (size_t)(T + 1) * N * nx));
    if (flag_min) SIM_TRY(hipMalloc(&df, sizeof(int32_t) * (size_t)N));
    int rc = lmpc_simulate_device(h, N, T, nx, nr, nuprev, F, G, dx, dr, du, dU, dX, df, warm, nullptr);
    if (rc == LMPC_OK) {
        SIM_TRY(hipDeviceSynchronize());
        SIM_TRY(hipMemcpy(x, dx, sizeof(double) * (size_t)N * nx, hipMemcpyDeviceToHost));
        if (uprev && nuprev > 0) SIM_TRY(hipMemcpy(uprev, du, sizeof(double) * (size_t)N * nuprev, hipMemcpyDeviceToHost));
        if (U_traj) SIM_TRY(hipMemcpy(U_traj, dU, sizeof(double) * (size_t)T * N * nu, hipMemcpyDeviceToHost));
        if (X_traj) SIM_TRY(hipMemcpy(X_traj, dX, sizeof(double) * (size_t)(T + 1) * N * nx, hipMemcpyDeviceToHost));
        if (flag_min) SIM_TRY(hipMemcpy(flag_min, df, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost));
    }
#undef SIM_TRY
    cleanup();
    return rc;
}

}  // extern "C"
