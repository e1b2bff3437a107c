// Internals shared by the translation units of liblmpc_hip.so: the handle, error plumbing, the
// wave kernel's launch interface.  The wavefront kernel's instantiations are compiled in their own
// translation units (lmpc_wave_inst.hip, one per real type x branch-and-bound) so that the library
// builds in parallel.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lmpc_hip.h"
#include "lmpc_pack.hpp"
#include "lmpc_wave_layout.hpp"

namespace lmpc {
constexpr int kLaneSizes[] = {2, 3, 4, 5, 6, 8, 10, 12};
constexpr int kLaneMaxN = 12;
constexpr int kLaneMaxM = 64;
constexpr size_t kLdsMax = 160 * 1024;
constexpr int kWaveMaxN = 127, kWaveMaxCap = 64, kWaveMaxM = 1024;
constexpr int64_t kProbe = 16384;       // points a fresh wavefront-kernel handle solves in front of its first large batch (wave_probe)

struct EventTriple { hipEvent_t a, mid, b; };
// work-list mode of the wavefront kernel for one launch (list == nullptr: the whole batch)
struct WaveList { const int32_t *list = nullptr, *count = nullptr; int32_t *count_next = nullptr; long long seg_cap = 0; };
}  // namespace lmpc

struct lmpc_handle {
    lmpc::HostPack P;
    lmpc_settings S;
    int device = 0;
    int laneN = 0;              // lane-kernel instantiation (row stride of M/Rout on the device)
    lmpc::PackLayout L{};
    double *dC = nullptr;       // constant pack on the GPU
    size_t nC = 0;
    std::string err, kname;
    std::mutex errMu;           // `err` is written through fail() only: the host pipeline's upload thread and the caller's
                                // thread can both fail on one handle; everything else of a handle is
                                // single-caller state -- one thread at a time per handle, as for a DAQP workspace
    // staging for the host-pointer entry point
    double *sTheta = nullptr, *sX = nullptr;
    int32_t *sFlag = nullptr, *sIter = nullptr;
    uint64_t *sAct = nullptr, *sWarm = nullptr;
    int64_t sCap = 0;
    // ... and its three-stage pipeline: H2D copies, kernels, D2H copies on their own streams (lmpc_multi.hip)
    hipStream_t sUp = nullptr, sRun = nullptr, sDown = nullptr;
    std::vector<hipEvent_t> pipeEv;
    int hostChunk = 32768;      // tuning: smallest (= last) chunk of the pipeline's schedule ("host_chunk")
    int hostRegister = 0;       // tuning: pin the caller's arrays in place for the call ("host_register"; see lmpc_multi.hip)
    int hostThreads = 1;        // tuning: pageable arrays: upload and download sides on separate host threads ("host_threads")
    // work list of the problems the screening pass leaves for the iterating kernel
    int32_t *dList = nullptr, *dCount = nullptr;
    int32_t *dList2 = nullptr, *dList3 = nullptr;          // second work list of the scenario-asynchronous closed loop (lists alternate per round)
    int64_t listCap = 0;        // batch size the list buffer was sized for
    int countSet = 0;           // which of the two counter sets the next call uses
    bool screen = true;         // two-pass (screen + iterate) for cold starts; lmpc_set_option
    // general path: one QP per wavefront
    bool useWave = false;
    bool bnb = false;           // rows flagged BINARY: branch and bound in the wavefront kernel
    int waveCap = 0;            // tuning: wavefronts per CU for the wave kernel's grid (0 = 16)
    bool waveQueue = true;      // tuning: dynamic problem queue of the wave kernel (0 = static split)
    lmpc::WaveList waveList{};  // set around a launch that follows the screening pass
    lmpc::WaveSim waveSim{};    // scenario-asynchronous closed loop on the wavefront path (kstep == nullptr: off)
    int screenWave = 1;         // tuning: screening pass in front of the wavefront kernel where it applies ("screen_wave")
    bool screenPackOnly = false;   // the lane-style pack holds only what the screening pass reads (wave-only problem)
    int32_t *dQueue = nullptr;
    int waveOvfSet = 0;         // ... and which pair of overflow counters this CALL uses
    int wavePass = 0;           // set by launch_wave_inst around its launches: 0 one pass, 1 / 2 first / second of two
    int waveTwoPass = -1;       // option "wave_two_pass": -1 auto, 0 off, 1 whenever the capacity allows
    int waveCap1 = 24;          // option "wave_cap1": working-set capacity of the first pass when "wave_two_pass" is 1
    int32_t *dOvfList1 = nullptr;
    int64_t ovfCap1 = 0;
    unsigned long long *dStat = nullptr;             // working-set statistics of the wavefront kernel (device, cumulative)
    volatile unsigned long long *hStat = nullptr;    // ... their copy in mapped host memory, written by every launch
    unsigned long long *dStatHost = nullptr;         // ... and its device address
    bool waveProbed = false;    // a fresh handle's first large batch has been probed for its working-set sizes (wave_probe)
    int waveProbe = 1;          // option "wave_probe": 0 = never probe (first calls run in one pass until statistics exist)
    unsigned long long statA[4] = {0, 0, 0, 0}, statB[4] = {0, 0, 0, 0};   // window of the statistics: snapshots of the
                                // cumulative counters about 2^20 problems apart; decisions use (now - statA)
    int waveCtrSet = 0;         // which of the two (ticket, overflow) counter pairs the next wavefront-kernel launch uses
    int32_t *dRegTable = nullptr;  // hash table of lmpc_distinct_active_sets_device (lmpc_regions.hip): 16 control words + slots
    int regCap = 0;
    unsigned long long *dRegW1 = nullptr;   // one-word masks: key / count / first-index tables of regW1Cap slots each (clean
    int regW1Cap = 0;                       // between calls: the publishing kernel resets what it read)
    int regW1 = 1;                          // tuning: 0 = the ballot-loop kernels also for one-word masks ("region_lockfree")
    int regBlocks = 0;          // tuning: workgroups per CU of the two-level distinct-mask reduction ("region_blocks", 0 = 1)
    long long *hRegOut = nullptr;   // lmpc_discover_regions_device: result block in mapped host memory ...
    long long *dRegOut = nullptr;   // ... and its device address; regOutWords = its size in 64-bit words
    size_t regOutWords = 0;
    // slow path for working sets beyond the 64 lanes (lmpc_big_kernel.hpp): overflow list + counter, per-thread scratch
    int capFull = 0;            // n + 2 + #soft: the rows a working set of this problem can reach (one beyond the kernels' own n + 1 + #soft)
    int32_t *dOvfList = nullptr, *dOvfCount = nullptr, *dBigI = nullptr;
    void *dBigR = nullptr;
    int64_t ovfCap = 0;
    // branch and bound: per-wavefront snapshots of a node's state, one slot per search depth (lmpc_wave_kernel.hpp)
    void *dBnbR = nullptr;
    int32_t *dBnbI = nullptr;
    size_t bnbBytesR = 0, bnbBytesI = 0;
    void *dRowBnb = nullptr;    // ... of the row kernel's searches (lmpc_row_inst.hip)
    size_t rowBnbBytes = 0;
    // closed loop on the wavefront path: per-scenario working set + factor kept between two steps ("sim_keep_factor")
    void *dKeepR = nullptr;
    int32_t *dKeepI = nullptr;
    int64_t keepCap = 0;
    bool keepOn = false;
    int simKeep = 1;
    bool raWarm = false;        // the run-ahead loop in progress is a warm one
    int simRunAhead = 1;        // scenario-asynchronous loop on the wavefront path: consecutive steps of a scenario inside the kernel
    int nBinary = 0;            // rows flagged BINARY = the search's largest depth
    int bigPath = 1;            // tuning: 0 = leave such points at exit flag -7 ("big_path")
    int wavePacked = -1;        // tuning: layout of the wave kernel's factor (-1 automatic, 0 square, 1 packed)
    int waveLevel = -1;         // tuning: LDS staging level of the wave kernel (-1 = automatic)
    int waveGram = 0;           // Gram-scan form of the wavefront kernel ("gram_scan"; lmpc_wave_kernel.hpp GRAM)
    int waveNwv = 0;            // tuning: wavefronts per wave-kernel workgroup (0 = automatic)
    int laneBlock = 0;          // tuning: workgroup size of the lane kernel (0 = automatic)
    int lanePer = 0;            // tuning: work-list workgroups per shard (0 = one resident round)
    int laneTier = 1;           // tuning: first-tier capacity in the boxed lane kernels (results identical either way)
    int laneStraight = 1;       // tuning: straight-line first tier in the boxed lane kernels ("lane_straight", lmpc_tiers.hpp)
    int fastPath = 1;           // tuning: one-launch kernel for small boxed problems ("fast", lmpc_fast_kernel.hpp)
    int fastTiles = 0;          // tuning: tiles of 64 problems per workgroup of that kernel (0 = 24)
    int fastNstr = 0;           // tuning: streaming wavefronts per workgroup of that kernel, 1..4 (0 = 3)
    int fastDyn = -1;           // tuning: tiles per workgroup of that kernel handed out through global tickets (-1 = default 4)
    int32_t *dFastCtr = nullptr;   // ... their counters: two alternating sets of kFastCtrs, 128 bytes apart
    int fastCtrSet = 0;
    int fastDma = -1;           // tuning: LDS-DMA ring depth of its streaming wavefronts (-1 = default, 0 = registers, 2, 3)
    int32_t *dFastErr = nullptr;   // raised by that kernel if one of its bounded waits ran out (never expected):
    volatile int32_t *hFastErr = nullptr;   // ... a word of pinned host memory, dFastErr its device address
    int fastSpinLimit = 0;         // test hook ("fast_spin_limit"): k > 0 = the kernel's waits give up after k - 1 polls
    // affine variational inequality (non-symmetric H, is_avi): its own kernel, pack and scratch (lmpc_avi.hip)
    bool avi = false;
    lmpc::AviLayout A{};
    double *dCa = nullptr;
    int32_t *dSa = nullptr;
    double *dAviR = nullptr;
    int32_t *dAviI = nullptr;
    int aviSlabs = 0;           // wavefront slabs of scratch allocated
    int aviWaves = 0;           // tuning: wavefronts per CU of its grid ("avi_waves", 0 = 16)
    // ... small box-constrained problems: register-resident straight-line kernels in front of it (lmpc_avi_tiers_kernel.hpp)
    int aviTiersN = 0;          // n if the problem qualifies (2 .. 8), else 0
    int aviTiers = 1;           // tuning: 0 = the generic kernel alone ("avi_tiers"; results identical either way)
    int aviTiersFirst = -1;     // tuning: tiers of the pass over the whole batch, 0 .. 3 ("avi_tiers_first"; -1: 3 up to n = 6, else 2;
                                // 0: the lane kernel over the whole batch)
    int aviTiersOcc[2] = {0, 0};   // workgroups per CU the two instantiations keep resident (0 = not asked yet)
    int32_t *dAviList[2] = {nullptr, nullptr};   // the two work lists of the chain, kShards segments each
    int32_t *dAviCnt = nullptr;                  // ... their counters (two sets of kShards, kCountStride apart)
    int64_t aviListCap = 0;
    lmpc::WaveLayout W{};
    double *dCw = nullptr;
    float *dCwf = nullptr;      // binary32 copy of the wave kernel's pack, built on the first f32 solve
    int32_t *dSw = nullptr;
    int numCU = 256;
    // closed-loop simulation scratch
    double *simTheta = nullptr, *simTheta2 = nullptr, *simU = nullptr, *simFG = nullptr;
    int asyncPhase = 0, asyncT = 0;     // scenario-asynchronous closed loop: which half `launch` runs (0 = off)
    int32_t *asyncCntNow = nullptr, *asyncCntNext = nullptr;
    int32_t *asyncListOut = nullptr;    // ... the list the last streaming pass wrote (the iterating half reads it)
    const int32_t *asyncListIn = nullptr, *asyncCntIn = nullptr;   // ... the round before's list: the scenarios to continue
    const double *asyncX = nullptr, *asyncR = nullptr, *asyncUp = nullptr;   // ... first pass: form theta from these
    int simSmall = 1;                   // ... the all-in-registers instantiation of sim_run_kernel where it applies
    int simBlind = 2;                   // ... rounds enqueued between two reads of the work-list counters
    int asyncCap = 0;                   // ... steps a scenario may run ahead in this streaming pass
    bool asyncResetPark = false;        // ... this pass consumes the parked list: clear its counters afterwards
    long long asyncMaxIn = 0;           // ... its longest shard segment (sizes the grid)
    int32_t *simK = nullptr;            // per-scenario step counters
    int simAsync = 1;           // tuning: scenario-asynchronous closed loop ("sim_async")
    int ccFused = 1;            // tuning: lmpc_compute_control assembles theta inside the screening kernel ("cc_fused")
    int simFused = 1;           // tuning: plant step inside the lane / screening kernels (lmpc_set_option "sim_fused")
    int32_t *simFlag = nullptr;
    uint64_t *simAct = nullptr;
    int64_t simCap = 0;
    // generated-controller entry point (lmpc_compute_control*): layout of theta, scratch, warm-start state
    int ccNx = -1, ccNr = 0, ccNd = 0, ccNup = 0, ccNp = 0, ccNph = 0;
    double *ccT2S = nullptr, *ccTheta = nullptr;
    uint64_t *ccAct = nullptr;
    int32_t *ccFlag = nullptr;
    // generated observer (lmpc_set_observer): [MPC_PLANT_DYNAMICS | MPC_MEASUREMENT_FUNCTION | K_TRANSPOSE_OBSERVER]
    double *obsC = nullptr;
    int obsNx = 0, obsNu = 0, obsNd = 0, obsNy = 0;
    double *ccObsScratch = nullptr;     // lmpc_compute_control_observer: state and disturbance split from the observer state
    int64_t ccObsCap = 0;
    double *ccStage = nullptr;          // host-pointer entry point: device copies of the five argument arrays
    int32_t *ccStageFlag = nullptr;
    int64_t ccStageCap = 0;
    size_t ccStagePer = 0;
    int64_t ccCap = 0, ccWarmN = -1;    // ccWarmN: batch size whose final working sets ccAct holds
    // lmpc_solve_one: ONE record in mapped host memory (theta in, x and flag out), its device address, its own stream
    char *oneHost = nullptr, *oneDev = nullptr;
    hipStream_t oneStream = nullptr;
    // lmpc_compute_control with a handful of problems (the generated controller's one call per time step): its five
    // argument arrays and the flags in mapped host memory likewise
    char *ccMapHost = nullptr, *ccMapDev = nullptr;
    size_t ccMapBytes = 0;
    // small problems with many rows on the wavefront path: straight-line tiers, one problem per lane, in front of it
    // (lmpc_qp_tiers_kernel.hpp)
    bool qpTiersOk = false;     // the problem qualifies (n = 2 .. 12, m <= 64 hard or SOFT rows without other flags, nth <= 16)
    double *dQpScan = nullptr;  // ... its scan pack (rows of M with their bounds), built with the first launch
    int qpTiers = 1;            // "qp_tiers": 0 = screening pass + wavefront kernel as before, 2 = always the tiers pass, 1 (default) =
                                // the tiers pass, and for large batches whichever of the two the handle has MEASURED faster
    // ... that measurement: one large call each way, timed by events that are read (without waiting) by later calls;
    // taken again every 512 calls.  Variant 0 = with the pass, 1 = without.
    hipEvent_t qpAbEv[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    bool qpAbPending[2] = {false, false};
    int64_t qpAbN[2] = {0, 0};
    double qpAbNsPer[2] = {-1.0, -1.0};      // measured nanoseconds per problem (-1: not measured yet)
    double qpAbAcc[2] = {-1.0, -1.0};   // ... best sample of the current round of measurements, and how many it holds
    int qpAbCnt[2] = {0, 0};
    long long qpAbCalls = 0;
    // four problems per wavefront (lmpc_row_kernel.hpp) in front of / instead of the wavefront kernel
    int rowKernel = -1;         // "row_kernel": -1 = where it applies (large cold batches), 0 = never, 1 = whenever an instantiation covers
    int rowBlocks = 0;          // tuning: workgroups per CU of its grid ("row_blocks", 0 = what LDS and registers allow)
    bool waveWarmed = false;    // lmpc_reserve has sent its one dummy problem through the wavefront kernel
    bool preloadOnly = false;   // launch_wave in "load the code, launch nothing" mode (preload_code)
    // profiling
    bool prof = false;
    std::vector<lmpc::EventTriple> events;
    std::vector<hipEvent_t> eventPool;   // recycled by lmpc_profile_read
};

extern thread_local std::string g_setup_err;

namespace lmpc {

hipError_t pool_event(lmpc_handle *h, hipEvent_t *e);
int fail(lmpc_handle *h, int code, const std::string &msg);
// the one-launch kernel's error word (lmpc_fast_inst.hip): LMPC_OK, or LMPC_ERR_HIP once per raised error
int check_fast_err(lmpc_handle *h);

#define HIP_TRY(h, call)                                                                     \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess)                                                               \
            return lmpc::fail(h, LMPC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

// Every entry point works on the handle's GPU and hands the caller's current device back when it returns
// (a host application with several GPUs keeps its own notion of "current device").
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t enter(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev == dev) return hipSuccess;
        const hipError_t e = hipSetDevice(dev);
        switched = (e == hipSuccess) && prev >= 0;
        return e;
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};
#define LMPC_ENTER_DEVICE(h) lmpc::DeviceScope lmpc_dev_scope__; HIP_TRY(h, lmpc_dev_scope__.enter((h)->device))

// affine-variational-inequality mode (lmpc_avi.hip): pack upload, settings, launch, scratch release
int finalize_avi(lmpc_handle *h);
void avi_fill_settings(lmpc_handle *h);
int launch_avi(lmpc_handle *h, int64_t nprob, const double *theta, double *x, int32_t *flag, int32_t *iters,
               uint64_t *active, const uint64_t *warm, hipStream_t st);
void avi_release(lmpc_handle *h, bool pack_too);
int launch_avi_tiers(lmpc_handle *h, bool first, int kfirst, unsigned grid, hipStream_t st, const double *theta, double *x,
                     int32_t *flag, int32_t *iters, uint64_t *active, const int32_t *list_in, const int32_t *count_in,
                     int32_t *list_out, int32_t *count_out, int32_t *count_clear, long long seg_cap, long long nprob, int *occ);

// one-launch solver for small box-constrained problems (lmpc_fast_inst.hip)
bool fast_covers(const lmpc_handle *h);
void fast_preload(lmpc_handle *h);
size_t qp_tiers_lds_bytes(int n, int m);
int launch_qp_tiers(lmpc_handle *h, int64_t nprob, const double *theta, double *x, int32_t *flag, int32_t *iters,
                    uint64_t *active, int32_t *list, int32_t *count, long long seg_cap, hipStream_t st, bool preload);
void avi_preload(lmpc_handle *h);
int avi_reserve(lmpc_handle *h, int64_t nprob, hipStream_t st);
int launch_fast(lmpc_handle *h, int64_t nprob, const double *theta, double *x, int32_t *flag, int32_t *iters,
                uint64_t *active, hipStream_t st);
int launch_fast_multi(lmpc_handle *h, int nb, int64_t nprob, const double *const *theta, double *const *x, int32_t *const *flag,
                      hipStream_t st);

// capacity of the first of two passes the wavefront kernel would run a batch of nprob problems at (0: one pass;
// lmpc_wave_launch.hpp, compiled into the binary64 translation unit)
int wave_first_pass_cap(lmpc_handle *h, int64_t nprob);
void wave_stat_read(const lmpc_handle *h, unsigned long long out[4]);
int wave_reserve(lmpc_handle *h, int64_t nprob, hipStream_t st);

// lmpc_api.hip's launch policy and scratch for the second API unit (lmpc_api_loop.hip); that unit's code preload
int api_launch(lmpc_handle *h, int64_t nprob, const double *theta, double *x, int32_t *flag, int32_t *iters, uint64_t *active,
               const uint64_t *warm, hipStream_t st);
int api_ensure_sim(lmpc_handle *h, int64_t N);
int api_ensure_f32(lmpc_handle *h);
bool api_will_screen(const lmpc_handle *h, int64_t nprob);
bool api_wave_screens(const lmpc_handle *h, int64_t nprob);
int api_wave_probe(lmpc_handle *h, const double *theta, int64_t nprob, hipStream_t st);
int api_launch_wave_f32(lmpc_handle *h, const float *dC, int64_t nprob, const float *theta, float *x, int32_t *flag, int32_t *iters,
                        uint64_t *active, const uint64_t *warm, hipStream_t st);
void loop_preload();

// four problems per wavefront (lmpc_row_inst.hip): capacity the batch would run at on that kernel (0: it does not take
// the batch), and its launch as the only pass (pass 0) or the first of two (pass 1) of a wavefront-kernel call
int row_pass_cap(lmpc_handle *h, int64_t nprob, size_t rs, bool warm, bool gram, bool bnb);
template <typename R>
int launch_row(lmpc_handle *h, const R *dC, int64_t nprob, const R *theta, R *x, int32_t *flag, int32_t *iters,
               uint64_t *active, hipStream_t st, int cap, int pass);
extern template int launch_row<double>(lmpc_handle *, const double *, int64_t, const double *, double *, int32_t *, int32_t *,
                                       uint64_t *, hipStream_t, int, int);
extern template int launch_row<float>(lmpc_handle *, const float *, int64_t, const float *, float *, int32_t *, int32_t *,
                                      uint64_t *, hipStream_t, int, int);
// ... with branch and bound
int row_bnb_pass_cap(lmpc_handle *h, int64_t nprob, size_t rs);
template <typename R>
int launch_row_bnb(lmpc_handle *h, const R *dC, int64_t nprob, const R *theta, R *x, int32_t *flag, int32_t *iters,
                   uint64_t *active, hipStream_t st, int cap, int pass);
extern template int launch_row_bnb<float>(lmpc_handle *, const float *, int64_t, const float *, float *, int32_t *, int32_t *,
                                          uint64_t *, hipStream_t, int, int);
extern template int launch_row_bnb<double>(lmpc_handle *, const double *, int64_t, const double *, double *, int32_t *, int32_t *,
                                           uint64_t *, hipStream_t, int, int);

// launch of the wavefront kernel for one batch (defined in lmpc_wave_launch.hpp, instantiated once per
// (R, BNB) in lmpc_wave_inst.hip)
template <typename R, bool BNB, bool GRAM, bool SIM = true>
int launch_wave_inst(lmpc_handle *h, const R *dC, int64_t nprob, const R *theta, R *x, int32_t *flag,
                     int32_t *iters, uint64_t *active, const uint64_t *warm, hipStream_t st);
#define LMPC_WAVE_EXTERN4(R, B, G, S)                                                                              \
    extern template int launch_wave_inst<R, B, G, S>(lmpc_handle *, const R *, int64_t, const R *, R *, int32_t *, \
                                                     int32_t *, uint64_t *, const uint64_t *, hipStream_t);
#define LMPC_WAVE_EXTERN(R, B, G) LMPC_WAVE_EXTERN4(R, B, G, true)
LMPC_WAVE_EXTERN(double, false, false) LMPC_WAVE_EXTERN(double, true, false)
LMPC_WAVE_EXTERN(float, false, false) LMPC_WAVE_EXTERN(float, true, false)
LMPC_WAVE_EXTERN(double, false, true) LMPC_WAVE_EXTERN(double, true, true)
LMPC_WAVE_EXTERN(float, false, true) LMPC_WAVE_EXTERN(float, true, true)
// (binary64 without branch and bound also WITHOUT the closed-loop machinery: the plain batched solve, its own units)
LMPC_WAVE_EXTERN4(double, false, false, false) LMPC_WAVE_EXTERN4(double, false, true, false)
#undef LMPC_WAVE_EXTERN
#undef LMPC_WAVE_EXTERN4

}  // namespace lmpc
