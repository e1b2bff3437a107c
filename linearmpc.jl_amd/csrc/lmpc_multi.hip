// Host-pointer entry points and the multi-device layer of liblmpc_hip.so.
//
//  * lmpc_solve_batch / lmpc_solve_batch_f32 (what a LinearMPC.jl caller with Theta in host memory
//    uses; reference src/utils.jl:268-283 `solve`, one caller, one Theta): the batch is cut into chunks
//    that move through a three-stage pipeline -- H2D copy of chunk k+1, kernels of chunk k, D2H copy of
//    chunk k-1 -- on three HIP streams per device, the caller's arrays pinned in place for the duration
//    of the call (hipHostRegister) so that every copy is asynchronous and PCIe runs in both directions
//    at once.
//  * lmpc_setup_multi / lmpc_solve_batch_multi: ONE process, one handle per GPU, the batch split into
//    contiguous shards (independent problems, constant pack replicated), every device running the same
//    pipeline on its shard, results written straight into the caller's arrays.
//  * lmpc_solve_batch_multi_device: shards already resident on their GPUs; per-shard solutions are
//    gathered to device 0 over xGMI with RCCL (ncclCommInitAll, ncclSend / ncclRecv pairs: every rank
//    sends on its own link).  RCCL is loaded on first use (dlopen), the library has no link-time
//    dependency on it.
#include <dlfcn.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "lmpc_internal.hpp"

using namespace lmpc;

namespace {

// ranges pinned through lmpc_pin_host (page-rounded): a second registration that touches one of their pages must
// not reach the runtime, which aborts on a doubly registered page
std::mutex g_pin_mu;
std::map<uintptr_t, uintptr_t> g_pins;      // begin -> end

// ---------------------------------------------------------------- pinned view of the caller's arrays
// hipHostRegister pins the pages in place (what the runtime does internally for a large pageable copy,
// but then the copy call blocks); registered, hipMemcpyAsync returns at once and the stages overlap.
struct PinScope {
    std::vector<std::pair<uintptr_t, uintptr_t>> want;      // [begin, end) of the caller's arrays
    std::vector<void *> pinned;
    void pin(const void *p, size_t bytes) {
        if (p && bytes) want.emplace_back(reinterpret_cast<uintptr_t>(p), reinterpret_cast<uintptr_t>(p) + bytes);
    }
    // Pinning works on whole pages, and two of the caller's arrays may share one (exit flags and iteration
    // counts of a 30 000-problem batch come out of the same malloc arena): registering both would register that
    // page twice, which the runtime answers with abort().  So: page-round every range, merge what overlaps or
    // touches, register each merged range once.
    void commit() {
        const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
        for (auto &r : want) { r.first &= ~(page - 1); r.second = (r.second + page - 1) & ~(page - 1); }
        std::sort(want.begin(), want.end());
        std::vector<std::pair<uintptr_t, uintptr_t>> merged;
        for (auto &r : want) {
            if (!merged.empty() && r.first <= merged.back().second) merged.back().second = std::max(merged.back().second, r.second);
            else merged.push_back(r);
        }
        want.clear();
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (auto &r : merged) {
            bool user_pinned = false;            // (pinned by the caller through lmpc_pin_host: nothing to do)
            for (auto &u : g_pins) user_pinned = user_pinned || (r.first < u.second && u.first < r.second);
            if (user_pinned) continue;
            void *p = reinterpret_cast<void *>(r.first);
            if (hipHostRegister(p, r.second - r.first, hipHostRegisterPortable) == hipSuccess) pinned.push_back(p);
            else (void)hipGetLastError();     // already registered by the caller, or not pinnable: the copies still work
        }
    }
    ~PinScope() { for (void *p : pinned) (void)hipHostUnregister(p); }
};

int ensure_staging(lmpc_handle *h, int64_t N, bool warm) {
    if (N <= h->sCap && (!warm || h->sWarm)) return LMPC_OK;
    if (N > h->sCap) {
        hipFree(h->sTheta); hipFree(h->sX); hipFree(h->sFlag); hipFree(h->sIter); hipFree(h->sAct); hipFree(h->sWarm);
        h->sTheta = h->sX = nullptr; h->sFlag = h->sIter = nullptr; h->sAct = h->sWarm = nullptr;
        h->sCap = 0;
        const size_t w = (size_t)h->P.words();
        HIP_TRY(h, hipMalloc(&h->sTheta, sizeof(double) * (size_t)N * (h->P.nth ? h->P.nth : 1)));
        HIP_TRY(h, hipMalloc(&h->sX, sizeof(double) * (size_t)N * h->P.nout));
        HIP_TRY(h, hipMalloc(&h->sFlag, sizeof(int32_t) * (size_t)N));
        HIP_TRY(h, hipMalloc(&h->sIter, sizeof(int32_t) * (size_t)N));
        HIP_TRY(h, hipMalloc(&h->sAct, sizeof(uint64_t) * (size_t)N * w));
        h->sCap = N;
    }
    if (warm && !h->sWarm)
        HIP_TRY(h, hipMalloc(&h->sWarm, sizeof(uint64_t) * (size_t)h->sCap * h->P.words()));
    return LMPC_OK;
}

int ensure_pipe(lmpc_handle *h, size_t nev) {
    if (!h->sUp) HIP_TRY(h, hipStreamCreateWithFlags(&h->sUp, hipStreamNonBlocking));
    if (!h->sRun) HIP_TRY(h, hipStreamCreateWithFlags(&h->sRun, hipStreamNonBlocking));
    if (!h->sDown) HIP_TRY(h, hipStreamCreateWithFlags(&h->sDown, hipStreamNonBlocking));
    while (h->pipeEv.size() < nev) {
        hipEvent_t e;
        HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->pipeEv.push_back(e);
    }
    return LMPC_OK;
}

// one device's share of a host-pointer call (rs = bytes per real: 8 binary64, 4 binary32)
struct HostJob {
    lmpc_handle *h;
    int64_t N;
    const char *theta;
    char *x;
    int32_t *flag, *iters;
    uint64_t *active;
    const uint64_t *warm;
};

int launch_chunk(lmpc_handle *h, size_t rs, int64_t n, int64_t off, bool warm) {
    const size_t nth = (size_t)h->P.nth, nout = (size_t)h->P.nout, w = (size_t)h->P.words();
    char *dTh = reinterpret_cast<char *>(h->sTheta), *dX = reinterpret_cast<char *>(h->sX);
    if (rs == 8)
        return lmpc_solve_batch_device(h, n, reinterpret_cast<const double *>(dTh + rs * off * nth),
                                       reinterpret_cast<double *>(dX + rs * off * nout), h->sFlag + off, h->sIter + off,
                                       h->sAct + off * w, warm ? h->sWarm + off * w : nullptr, h->sRun);
    return lmpc_solve_batch_f32_device(h, n, reinterpret_cast<const float *>(dTh + rs * off * nth),
                                       reinterpret_cast<float *>(dX + rs * off * nout), h->sFlag + off, h->sIter + off,
                                       h->sAct + off * w, warm ? h->sWarm + off * w : nullptr, h->sRun);
}

// Chunk schedule of one shard: every chunk half of what is left, down to `smallest` problems.  The H2D
// copies are the longest stage (56 B in against 12 B out per pendulum problem), so what the pipeline adds to
// the plain copy time is its tail -- kernels and D2H copy of the LAST chunk -- and the per-chunk enqueue cost
// (~40 us of host time: equal chunks of 65536 made the HOST the bottleneck, 1.45 ms per 10^6 against 1.18 ms
// with 262144): few, large chunks first, a small one last.
std::vector<int64_t> chunk_offsets(int64_t N, int64_t smallest) {
    std::vector<int64_t> off{0};
    int64_t left = N;
    while (left > 0) {
        int64_t c = left / 2;
        if (c < smallest) c = std::min(left, smallest);
        if (left - c < smallest / 2) c = left;           // no crumbs
        off.push_back(off.back() + c);
        left -= c;
    }
    return off;
}

// is [p, p + bytes) inside memory the caller pinned through lmpc_pin_host?
bool caller_pinned(const void *p, size_t bytes) {
    if (!p || bytes == 0) return true;
    const uintptr_t b = reinterpret_cast<uintptr_t>(p), e = b + bytes;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (auto &r : g_pins)
        if (r.first <= b && e <= r.second) return true;
    return false;
}

// How the chunks of all jobs are driven:
//   ASYNC     every array of the call is pinned (by the caller through lmpc_pin_host, or for this call with
//             "host_register"): hipMemcpyAsync returns at once, ONE thread enqueues everything;
//   THREADED  pageable arrays: a copy call blocks until its bytes have moved, so the upload side (H2D copy +
//             kernel launches, one thread per device) and the download side (D2H copies, the calling thread) run
//             on separate host threads -- PCIe is busy in both directions although every single call blocks;
//   SINGLE    small calls: one chunk, one thread, no pipeline.
enum class HostMode { ASYNC, THREADED, SINGLE };

int run_host_jobs(std::vector<HostJob> &jobs, size_t rs, lmpc_handle *errh, HostMode mode) {
    int64_t maxN = 0;
    for (auto &j : jobs) maxN = std::max(maxN, j.N);
    if (maxN == 0) return LMPC_OK;
    lmpc_handle *h0 = jobs[0].h;
    // (pageable copies pay a pin / unpin inside the runtime per call: fewer, larger chunks -- measured best at 65536)
    const int64_t smallest = mode == HostMode::SINGLE ? maxN
                           : std::max<int64_t>(mode == HostMode::THREADED ? 65536 : 1024, h0->hostChunk);
    std::vector<std::vector<int64_t>> sched;
    int64_t nchunks = 0;
    for (auto &j : jobs) {
        sched.push_back(chunk_offsets(j.N, smallest));
        nchunks = std::max<int64_t>(nchunks, (int64_t)sched.back().size() - 1);
    }
    for (auto &j : jobs) {
        if (j.N == 0) continue;
        DeviceScope sc;
        HIP_TRY(j.h, sc.enter(j.h->device));
        int rc = ensure_staging(j.h, j.N, j.warm != nullptr);
        if (rc == LMPC_OK) rc = ensure_pipe(j.h, (size_t)(2 * nchunks));
        if (rc != LMPC_OK) { if (errh != j.h) errh->err = j.h->err; return rc; }
    }
    // upload side of chunk c of job ji: H2D copies, then the kernels behind them
    auto upload = [&](size_t ji, int64_t c) -> int {
        HostJob &j = jobs[ji];
        lmpc_handle *h = j.h;
        const int64_t off = sched[ji][c], n = sched[ji][c + 1] - off;
        const size_t nth = (size_t)h->P.nth, w = (size_t)h->P.words();
        DeviceScope sc;
        HIP_TRY(h, sc.enter(h->device));
        hipEvent_t evUp = h->pipeEv[2 * c], evRun = h->pipeEv[2 * c + 1];
        if (nth > 0)
            HIP_TRY(h, hipMemcpyAsync(reinterpret_cast<char *>(h->sTheta) + rs * off * nth, j.theta + rs * off * nth,
                                      rs * n * nth, hipMemcpyHostToDevice, h->sUp));
        if (j.warm)
            HIP_TRY(h, hipMemcpyAsync(h->sWarm + off * w, j.warm + off * w, sizeof(uint64_t) * n * w,
                                      hipMemcpyHostToDevice, h->sUp));
        HIP_TRY(h, hipEventRecord(evUp, h->sUp));
        HIP_TRY(h, hipStreamWaitEvent(h->sRun, evUp, 0));
        const int rc = launch_chunk(h, rs, n, off, j.warm != nullptr);
        if (rc != LMPC_OK) return rc;
        HIP_TRY(h, hipEventRecord(evRun, h->sRun));
        return LMPC_OK;
    };
    // download side: D2H copies behind the chunk's kernels
    auto download = [&](size_t ji, int64_t c) -> int {
        HostJob &j = jobs[ji];
        lmpc_handle *h = j.h;
        const int64_t off = sched[ji][c], n = sched[ji][c + 1] - off;
        const size_t nout = (size_t)h->P.nout, w = (size_t)h->P.words();
        DeviceScope sc;
        HIP_TRY(h, sc.enter(h->device));
        HIP_TRY(h, hipStreamWaitEvent(h->sDown, h->pipeEv[2 * c + 1], 0));
        HIP_TRY(h, hipMemcpyAsync(j.x + rs * off * nout, reinterpret_cast<char *>(h->sX) + rs * off * nout,
                                  rs * n * nout, hipMemcpyDeviceToHost, h->sDown));
        HIP_TRY(h, hipMemcpyAsync(j.flag + off, h->sFlag + off, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->sDown));
        if (j.iters)
            HIP_TRY(h, hipMemcpyAsync(j.iters + off, h->sIter + off, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->sDown));
        if (j.active)
            HIP_TRY(h, hipMemcpyAsync(j.active + off * w, h->sAct + off * w, sizeof(uint64_t) * n * w,
                                      hipMemcpyDeviceToHost, h->sDown));
        return LMPC_OK;
    };
    auto has = [&](size_t ji, int64_t c) { return c + 1 < (int64_t)sched[ji].size(); };
    int rc = LMPC_OK;
    if (mode != HostMode::THREADED) {
        for (int64_t c = 0; c < nchunks && rc == LMPC_OK; c++)
            for (size_t ji = 0; ji < jobs.size() && rc == LMPC_OK; ji++)
                if (has(ji, c)) { rc = upload(ji, c); if (rc == LMPC_OK) rc = download(ji, c); }
    } else {
        std::vector<std::atomic<int64_t>> launched(jobs.size());
        for (auto &a : launched) a.store(0);
        std::atomic<int> uerr{LMPC_OK};
        std::vector<std::thread> up;
        for (size_t ji = 0; ji < jobs.size(); ji++)
            up.emplace_back([&, ji]() {
                for (int64_t c = 0; has(ji, c) && uerr.load() == LMPC_OK; c++) {
                    const int r = upload(ji, c);
                    if (r != LMPC_OK) { uerr.store(r); break; }
                    launched[ji].store(c + 1, std::memory_order_release);
                }
            });
        for (int64_t c = 0; c < nchunks && rc == LMPC_OK; c++)
            for (size_t ji = 0; ji < jobs.size() && rc == LMPC_OK; ji++) {
                if (!has(ji, c)) continue;
                while (launched[ji].load(std::memory_order_acquire) <= c && uerr.load() == LMPC_OK) std::this_thread::yield();
                if (uerr.load() != LMPC_OK) { rc = uerr.load(); break; }
                rc = download(ji, c);
            }
        if (rc != LMPC_OK) uerr.store(rc);               // stops the upload threads at their next chunk
        for (auto &t : up) t.join();
    }
    if (rc != LMPC_OK) for (auto &j : jobs) if (errh != j.h && !j.h->err.empty()) errh->err = j.h->err;
    // drain every device's pipeline, also after an error (nothing may still write into the caller's arrays)
    for (auto &j : jobs) {
        if (j.N == 0) continue;
        DeviceScope sc;
        if (sc.enter(j.h->device) != hipSuccess) continue;
        hipError_t e1 = hipStreamSynchronize(j.h->sUp), e2 = hipStreamSynchronize(j.h->sRun),
                   e3 = hipStreamSynchronize(j.h->sDown);
        if (rc == LMPC_OK && (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess))
            rc = fail(errh, LMPC_ERR_HIP, std::string("lmpc: host pipeline: ") +
                                              hipGetErrorString(e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3)));
    }
    // everything this call enqueued has completed: the one-launch kernel's error word is final (lmpc_fast_inst.hip)
    for (auto &j : jobs) {
        if (rc != LMPC_OK || j.N == 0) continue;
        rc = check_fast_err(j.h);
        if (rc != LMPC_OK && errh != j.h) errh->err = j.h->err;
    }
    return rc;
}

// pinned for this call ("host_register" 1), pinned by the caller, or pageable?
HostMode host_mode(PinScope &pin, const lmpc_handle *h, size_t rs, int64_t N, const void *theta, void *x, int32_t *flag,
                   int32_t *iters, uint64_t *active, const uint64_t *warm);

HostMode host_mode(PinScope &pin, const lmpc_handle *h, size_t rs, int64_t N, const void *theta, void *x, int32_t *flag,
                   int32_t *iters, uint64_t *active, const uint64_t *warm) {
    const size_t w = (size_t)h->P.words();
    const size_t bth = rs * (size_t)N * h->P.nth, bx = rs * (size_t)N * h->P.nout, bf = sizeof(int32_t) * (size_t)N,
                 ba = sizeof(uint64_t) * (size_t)N * w;
    // small calls (a closed loop's single solve): a pipeline would cost more than it hides
    if (bth + bx + bf < ((size_t)4 << 20)) return HostMode::SINGLE;
    if (caller_pinned(theta, bth) && caller_pinned(x, bx) && caller_pinned(flag, bf) && caller_pinned(iters, iters ? bf : 0) &&
        caller_pinned(active, active ? ba : 0) && caller_pinned(warm, warm ? ba : 0))
        return HostMode::ASYNC;
    // (pinning the caller's arrays for the duration of a call -- "host_register" 1 of rounds 2-4 -- is gone: on this
    // runtime registering and unregistering ordinary heap memory call after call ended, after a few hundred calls in one
    // process, in a GPU memory access fault.  Caller-pinned, long-lived arrays (lmpc_pin_host) take the branch above.)
    return h->hostThreads ? HostMode::THREADED : HostMode::SINGLE;
}

int solve_host(lmpc_handle *h, size_t rs, int64_t N, const void *theta, void *x, int32_t *flag, int32_t *iters,
               uint64_t *active, const uint64_t *warm, const char *who) {
    if (!h) return LMPC_ERR_BADARG;
    if (N < 0 || (N > 0 && (!x || !flag || (h->P.nth > 0 && !theta))))
        return fail(h, LMPC_ERR_BADARG, std::string(who) + ": NULL array or negative N");
    if (N == 0) return LMPC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(h, LMPC_ERR_NOGPU, "lmpc: no HIP device available (this library has no CPU path)");
    LMPC_ENTER_DEVICE(h);
    PinScope pin;
    const HostMode mode = host_mode(pin, h, rs, N, theta, x, flag, iters, active, warm);
    std::vector<HostJob> jobs{HostJob{h, N, static_cast<const char *>(theta), static_cast<char *>(x), flag, iters, active, warm}};
    return run_host_jobs(jobs, rs, h, mode);
}

// ---------------------------------------------------------------- RCCL, loaded on first use
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string &err) {
        if (lib) return true;
        const char *env = std::getenv("LMPC_RCCL_LIB");
        const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            lib = dlopen(n, RTLD_LAZY | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("lmpc: cannot load RCCL (librccl.so.1): ") + (dlerror() ? dlerror() : ""); return false; }
#define LMPC_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(lib, name)); if (!field) { err = "lmpc: RCCL lacks " name; return false; }
        LMPC_SYM(CommInitAll, "ncclCommInitAll") LMPC_SYM(CommDestroy, "ncclCommDestroy")
        LMPC_SYM(GroupStart, "ncclGroupStart") LMPC_SYM(GroupEnd, "ncclGroupEnd") LMPC_SYM(Send, "ncclSend")
        LMPC_SYM(Recv, "ncclRecv") LMPC_SYM(GetErrorString, "ncclGetErrorString")
#undef LMPC_SYM
        return true;
    }
};
Rccl g_rccl;

}  // namespace

struct lmpc_multi {
    std::vector<lmpc_handle *> h;      // one per device, same problem
    std::vector<int> dev;
    std::vector<hipStream_t> stream;   // device-resident path: one stream per device
    std::vector<ncclComm_t> comm;      // created on the first gather
    std::vector<hipEvent_t> done;      // transport 1: "shard d is solved", recorded on stream d, awaited by stream 0
    int transport = 0;                 // 0: RCCL send / receive pairs; 1: event-ordered peer copies issued by device 0
    bool rcclSelf = false;             // test hook (LMPC_MULTI_TRANSPORT=rccl_self, one device): the local copy of shard 0
                                       // goes through RCCL too -- ncclCommInitAll, a send / receive pair to itself
    bool repeats = false;              // the device list names a device more than once (transport 1 only)
    std::string err;
};

namespace {
int mfail(lmpc_multi *hm, int code, const std::string &msg) {
    if (hm) hm->err = msg; else g_setup_err = msg;
    return code;
}
}  // namespace

extern "C" {

int lmpc_solve_batch(lmpc_handle *h, int64_t N, const double *theta, double *x, int32_t *exitflag,
                     int32_t *iters, uint64_t *active, const uint64_t *warm) {
    return solve_host(h, sizeof(double), N, theta, x, exitflag, iters, active, warm, "lmpc_solve_batch");
}

int lmpc_solve_batch_f32(lmpc_handle *h, int64_t N, const float *theta, float *x, int32_t *exitflag,
                         int32_t *iters, uint64_t *active, const uint64_t *warm) {
    return solve_host(h, sizeof(float), N, theta, x, exitflag, iters, active, warm, "lmpc_solve_batch_f32");
}

int lmpc_pin_host(void *p, size_t bytes) {
    if (!p || bytes == 0) return LMPC_ERR_BADARG;
    const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
    const uintptr_t b = reinterpret_cast<uintptr_t>(p) & ~(page - 1);
    const uintptr_t e = (reinterpret_cast<uintptr_t>(p) + bytes + page - 1) & ~(page - 1);
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (auto &r : g_pins)
        if (b < r.second && r.first < e)
            return mfail(nullptr, LMPC_ERR_BADARG, "lmpc_pin_host: the range shares a page with memory pinned earlier "
                                                   "(pin each allocation once, or allocate page-aligned)");
    const hipError_t err = hipHostRegister(reinterpret_cast<void *>(b), e - b, hipHostRegisterPortable);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        return mfail(nullptr, err == hipErrorNoDevice ? LMPC_ERR_NOGPU : LMPC_ERR_HIP, std::string("hipHostRegister: ") + hipGetErrorString(err));
    }
    g_pins[b] = e;
    return LMPC_OK;
}

int lmpc_unpin_host(void *p) {
    if (!p) return LMPC_ERR_BADARG;
    const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
    const uintptr_t b = reinterpret_cast<uintptr_t>(p) & ~(page - 1);
    std::lock_guard<std::mutex> lk(g_pin_mu);
    auto it = g_pins.find(b);
    if (it == g_pins.end()) return mfail(nullptr, LMPC_ERR_BADARG, "lmpc_unpin_host: not pinned by lmpc_pin_host");
    const hipError_t err = hipHostUnregister(reinterpret_cast<void *>(b));
    g_pins.erase(it);
    if (err == hipSuccess) return LMPC_OK;
    (void)hipGetLastError();
    return mfail(nullptr, LMPC_ERR_HIP, std::string("hipHostUnregister: ") + hipGetErrorString(err));
}

void lmpc_multi_partition(int64_t N, int n_devices, int64_t *offsets) {
    // contiguous shards, the remainder spread over the leading devices: offsets[d] .. offsets[d+1]
    if (!offsets || n_devices <= 0) return;
    const int64_t base = N > 0 ? N / n_devices : 0, rem = N > 0 ? N % n_devices : 0;
    offsets[0] = 0;
    for (int d = 0; d < n_devices; d++) offsets[d + 1] = offsets[d] + base + (d < rem ? 1 : 0);
}

int lmpc_setup_multi(lmpc_multi **out, int n, int m, int ms, int nth, int nout, const double *H, const double *f,
                     const double *f_theta, const double *A, const double *bu, const double *bl, const double *W,
                     const int32_t *sense, const double *Kfb, int nx, const lmpc_settings *s, const int *devices,
                     int n_devices) {
    if (!out) return mfail(nullptr, LMPC_ERR_BADARG, "lmpc_setup_multi: out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return mfail(nullptr, LMPC_ERR_NOGPU, "lmpc: no HIP device available (this library has no CPU path)");
    if (n_devices <= 0) { n_devices = ndev; devices = nullptr; }      // all visible devices
    lmpc_multi *hm = new lmpc_multi();
    // LMPC_MULTI_TRANSPORT=copy: the gather runs as peer copies instead of RCCL pairs (lmpc_multi_set_option
    // "transport" 1), and only then may a device appear more than once in the list -- several shards, handles, streams
    // and host threads on ONE GPU: how the n_devices > 1 control flow is exercised on a one-GPU machine
    const char *tenv = std::getenv("LMPC_MULTI_TRANSPORT");
    if (tenv && std::strcmp(tenv, "copy") == 0) hm->transport = 1;
    if (tenv && std::strcmp(tenv, "rccl_self") == 0) hm->rcclSelf = true;
    for (int d = 0; d < n_devices; d++) {
        const int dev = devices ? devices[d] : d;
        const bool seen = std::find(hm->dev.begin(), hm->dev.end(), dev) != hm->dev.end();
        if (dev < 0 || dev >= ndev || (seen && hm->transport != 1)) {
            lmpc_free_multi(hm);
            return mfail(nullptr, LMPC_ERR_BADARG, "lmpc_setup_multi: bad or repeated device ordinal");
        }
        hm->repeats = hm->repeats || seen;
        lmpc_handle *h = nullptr;
        const int rc = lmpc_setup(&h, n, m, ms, nth, nout, H, f, f_theta, A, bu, bl, W, sense, Kfb, nx, s, dev);
        if (rc != LMPC_OK) { lmpc_free_multi(hm); return rc; }       // (the text is lmpc_last_error(NULL))
        hm->h.push_back(h);
        hm->dev.push_back(dev);
    }
    *out = hm;
    return LMPC_OK;
}

int lmpc_multi_devices(const lmpc_multi *hm) { return hm ? (int)hm->h.size() : LMPC_ERR_BADARG; }

int lmpc_multi_set_option(lmpc_multi *hm, const char *name, int value) {
    if (!hm || !name) return LMPC_ERR_BADARG;
    if (std::strcmp(name, "transport") == 0) {
        if (value != 0 && value != 1) return mfail(hm, LMPC_ERR_BADARG, "lmpc_multi_set_option: transport must be 0 (RCCL) or 1 (peer copies)");
        if (value == 0 && hm->repeats)
            return mfail(hm, LMPC_ERR_BADARG, "lmpc_multi_set_option: RCCL needs distinct devices (this handle repeats one)");
        hm->transport = value;
        return LMPC_OK;
    }
    return mfail(hm, LMPC_ERR_BADARG, std::string("lmpc_multi_set_option: unknown option ") + name);
}

lmpc_handle *lmpc_multi_handle(lmpc_multi *hm, int i) {
    return (hm && i >= 0 && i < (int)hm->h.size()) ? hm->h[i] : nullptr;
}

const char *lmpc_multi_last_error(const lmpc_multi *hm) { return hm ? hm->err.c_str() : g_setup_err.c_str(); }

int lmpc_solve_batch_multi(lmpc_multi *hm, int64_t N, const double *theta, double *x, int32_t *exitflag,
                           int32_t *iters, uint64_t *active, const uint64_t *warm) {
    if (!hm || hm->h.empty()) return LMPC_ERR_BADARG;
    lmpc_handle *h0 = hm->h[0];
    if (N < 0 || (N > 0 && (!x || !exitflag || (h0->P.nth > 0 && !theta))))
        return mfail(hm, LMPC_ERR_BADARG, "lmpc_solve_batch_multi: NULL array or negative N");
    if (N == 0) return LMPC_OK;
    const int nd = (int)hm->h.size();
    const size_t nth = (size_t)h0->P.nth, nout = (size_t)h0->P.nout, w = (size_t)h0->P.words();
    std::vector<int64_t> off((size_t)nd + 1);
    lmpc_multi_partition(N, nd, off.data());
    PinScope pin;
    const HostMode mode = host_mode(pin, h0, sizeof(double), N, theta, x, exitflag, iters, active, warm);
    std::vector<HostJob> jobs;
    for (int d = 0; d < nd; d++) {
        const int64_t o = off[d];
        jobs.push_back(HostJob{hm->h[d], off[d + 1] - o, reinterpret_cast<const char *>(theta + o * nth),
                               reinterpret_cast<char *>(x + o * nout), exitflag + o, iters ? iters + o : nullptr,
                               active ? active + o * w : nullptr, warm ? warm + o * w : nullptr});
    }
    const int rc = run_host_jobs(jobs, sizeof(double), h0, mode);
    if (rc != LMPC_OK) hm->err = h0->err;
    return rc;
}

int lmpc_solve_batch_multi_device(lmpc_multi *hm, const int64_t *N_dev, const double *const *theta,
                                  double *const *x, int32_t *const *exitflag, double *x_root,
                                  int32_t *exitflag_root) {
    if (!hm || hm->h.empty() || !N_dev || !theta || !x || !exitflag) return LMPC_ERR_BADARG;
    const int nd = (int)hm->h.size();
    const size_t nout = (size_t)hm->h[0]->P.nout;
    if (hm->stream.empty()) {
        hm->stream.assign((size_t)nd, nullptr);
        for (int d = 0; d < nd; d++) {
            DeviceScope sc;
            if (sc.enter(hm->dev[d]) != hipSuccess || hipStreamCreateWithFlags(&hm->stream[d], hipStreamNonBlocking) != hipSuccess)
                return mfail(hm, LMPC_ERR_HIP, "lmpc_solve_batch_multi_device: cannot create the device streams");
        }
    }
    const bool gather = x_root != nullptr || exitflag_root != nullptr;
    if (gather && nd > 1 && hm->transport == 1 && hm->done.empty()) {
        hm->done.assign((size_t)nd, nullptr);
        for (int d = 0; d < nd; d++) {
            DeviceScope sc;
            if (sc.enter(hm->dev[d]) != hipSuccess || hipEventCreateWithFlags(&hm->done[d], hipEventDisableTiming) != hipSuccess) {
                hm->done.clear();
                return mfail(hm, LMPC_ERR_HIP, "lmpc_solve_batch_multi_device: cannot create the gather events");
            }
        }
    }
    if (gather && ((nd > 1 && hm->transport == 0) || (nd == 1 && hm->rcclSelf)) && hm->comm.empty()) {
        if (!g_rccl.load(hm->err)) return LMPC_ERR_UNSUPPORTED;
        hm->comm.assign((size_t)nd, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(hm->comm.data(), nd, hm->dev.data());
        if (r != ncclSuccess) {
            hm->comm.clear();
            return mfail(hm, LMPC_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
        }
    }
    // every device solves its own shard on its own stream (no data-path collective inside the solve)
    for (int d = 0; d < nd; d++) {
        if (N_dev[d] < 0) return mfail(hm, LMPC_ERR_BADARG, "lmpc_solve_batch_multi_device: negative shard size");
        if (N_dev[d] == 0) continue;
        const int rc = lmpc_solve_batch_device(hm->h[d], N_dev[d], theta[d], x[d], exitflag[d], nullptr, nullptr,
                                               nullptr, hm->stream[d]);
        if (rc != LMPC_OK) { hm->err = hm->h[d]->err; return rc; }
    }
    int rc = LMPC_OK;
    if (gather) {
        // the one exchange step of the sharded job: per-shard solutions to device 0.  Shard 0 is a local
        // copy; every other device sends its shard on its own xGMI link (ncclSend / ncclRecv pairs inside
        // one group), device 0 receives them behind its own solve.
        std::vector<int64_t> off((size_t)nd + 1, 0);
        for (int d = 0; d < nd; d++) off[d + 1] = off[d] + N_dev[d];
        {
            DeviceScope sc;
            if (sc.enter(hm->dev[0]) != hipSuccess) return mfail(hm, LMPC_ERR_HIP, "lmpc: hipSetDevice");
            if (nd == 1 && hm->rcclSelf && N_dev[0] > 0) {
                // one device, test hook: shard 0 reaches its place in the gathered arrays through RCCL -- the same calls,
                // data types and group structure the several-device branch below makes, with itself as the peer
                ncclResult_t r = g_rccl.GroupStart();
                if (x_root && x_root != x[0] && r == ncclSuccess) {
                    r = g_rccl.Send(x[0], (size_t)N_dev[0] * nout, ncclFloat64, 0, hm->comm[0], hm->stream[0]);
                    if (r == ncclSuccess) r = g_rccl.Recv(x_root, (size_t)N_dev[0] * nout, ncclFloat64, 0, hm->comm[0], hm->stream[0]);
                }
                if (exitflag_root && exitflag_root != exitflag[0] && r == ncclSuccess) {
                    r = g_rccl.Send(exitflag[0], (size_t)N_dev[0], ncclInt32, 0, hm->comm[0], hm->stream[0]);
                    if (r == ncclSuccess) r = g_rccl.Recv(exitflag_root, (size_t)N_dev[0], ncclInt32, 0, hm->comm[0], hm->stream[0]);
                }
                const ncclResult_t re = g_rccl.GroupEnd();
                if (r == ncclSuccess) r = re;
                if (r != ncclSuccess) rc = mfail(hm, LMPC_ERR_HIP, std::string("RCCL self gather: ") + g_rccl.GetErrorString(r));
            } else {
            if (x_root && N_dev[0] > 0 && x_root != x[0] &&
                hipMemcpyAsync(x_root, x[0], sizeof(double) * N_dev[0] * nout, hipMemcpyDeviceToDevice, hm->stream[0]) != hipSuccess)
                rc = mfail(hm, LMPC_ERR_HIP, "lmpc: gather: local copy of shard 0");
            if (exitflag_root && N_dev[0] > 0 && exitflag_root != exitflag[0] &&
                hipMemcpyAsync(exitflag_root, exitflag[0], sizeof(int32_t) * N_dev[0], hipMemcpyDeviceToDevice, hm->stream[0]) != hipSuccess)
                rc = mfail(hm, LMPC_ERR_HIP, "lmpc: gather: local copy of shard 0");
            }
        }
        if (rc == LMPC_OK && nd > 1 && hm->transport == 1) {
            // peer copies: shard d's solve is marked by an event on its stream; device 0's stream waits for it and pulls
            // the shard into its place (hipMemcpyPeerAsync: the copy engines over xGMI between two devices, a plain
            // device-to-device copy when both ends are the same GPU)
            for (int d = 1; d < nd && rc == LMPC_OK; d++) {
                if (N_dev[d] == 0) continue;
                {
                    DeviceScope sc;
                    if (sc.enter(hm->dev[d]) != hipSuccess || hipEventRecord(hm->done[d], hm->stream[d]) != hipSuccess) {
                        rc = mfail(hm, LMPC_ERR_HIP, "lmpc: gather: recording a shard's event");
                        break;
                    }
                }
                DeviceScope sc0;
                if (sc0.enter(hm->dev[0]) != hipSuccess || hipStreamWaitEvent(hm->stream[0], hm->done[d], 0) != hipSuccess) {
                    rc = mfail(hm, LMPC_ERR_HIP, "lmpc: gather: waiting for a shard's event");
                    break;
                }
                if (x_root && hipMemcpyPeerAsync(x_root + off[d] * nout, hm->dev[0], x[d], hm->dev[d],
                                                 sizeof(double) * (size_t)N_dev[d] * nout, hm->stream[0]) != hipSuccess)
                    rc = mfail(hm, LMPC_ERR_HIP, "lmpc: gather: peer copy of the solutions");
                if (rc == LMPC_OK && exitflag_root &&
                    hipMemcpyPeerAsync(exitflag_root + off[d], hm->dev[0], exitflag[d], hm->dev[d],
                                       sizeof(int32_t) * (size_t)N_dev[d], hm->stream[0]) != hipSuccess)
                    rc = mfail(hm, LMPC_ERR_HIP, "lmpc: gather: peer copy of the exit flags");
            }
        } else if (rc == LMPC_OK && nd > 1) {
            ncclResult_t r = g_rccl.GroupStart();
            for (int d = 1; d < nd && r == ncclSuccess; d++) {
                if (N_dev[d] == 0) continue;
                if (x_root) {
                    r = g_rccl.Send(x[d], (size_t)N_dev[d] * nout, ncclFloat64, 0, hm->comm[d], hm->stream[d]);
                    if (r == ncclSuccess)
                        r = g_rccl.Recv(x_root + off[d] * nout, (size_t)N_dev[d] * nout, ncclFloat64, d, hm->comm[0], hm->stream[0]);
                }
                if (exitflag_root && r == ncclSuccess) {
                    r = g_rccl.Send(exitflag[d], (size_t)N_dev[d], ncclInt32, 0, hm->comm[d], hm->stream[d]);
                    if (r == ncclSuccess)
                        r = g_rccl.Recv(exitflag_root + off[d], (size_t)N_dev[d], ncclInt32, d, hm->comm[0], hm->stream[0]);
                }
            }
            const ncclResult_t re = g_rccl.GroupEnd();
            if (r == ncclSuccess) r = re;
            if (r != ncclSuccess) rc = mfail(hm, LMPC_ERR_HIP, std::string("RCCL gather: ") + g_rccl.GetErrorString(r));
        }
    }
    for (int d = 0; d < nd; d++) {
        DeviceScope sc;
        if (sc.enter(hm->dev[d]) != hipSuccess) continue;
        const hipError_t e = hipStreamSynchronize(hm->stream[d]);
        if (e != hipSuccess && rc == LMPC_OK) rc = mfail(hm, LMPC_ERR_HIP, std::string("lmpc: device stream: ") + hipGetErrorString(e));
    }
    return rc;
}

void lmpc_free_multi(lmpc_multi *hm) {
    if (!hm) return;
    for (ncclComm_t c : hm->comm) if (c && g_rccl.CommDestroy) g_rccl.CommDestroy(c);
    for (size_t d = 0; d < hm->done.size(); d++) {
        DeviceScope sc;
        if (hm->done[d] && sc.enter(hm->dev[d]) == hipSuccess) hipEventDestroy(hm->done[d]);
    }
    for (size_t d = 0; d < hm->stream.size(); d++) {
        DeviceScope sc;
        if (hm->stream[d] && sc.enter(hm->dev[d]) == hipSuccess) hipStreamDestroy(hm->stream[d]);
    }
    for (lmpc_handle *h : hm->h) lmpc_free(h);
    delete hm;
}

}  // extern "C"
