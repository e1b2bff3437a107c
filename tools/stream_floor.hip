// Floor of the headline call's memory stream on this GPU: kernels that read 10^6 records of 7 doubles (56 MB) and write
// one double + one int per record (12 MB), nothing else, in the access shapes the solver kernels could use.
// Cold HBM: six rotating input buffers (336 MB between two uses of a line).  Event-timed like lmpc_profile (one
// call at a time).   build: hipcc -O3 --offload-arch=gfx950 -o stream_floor tools/stream_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int NT = 7;

// A: one record per lane, one tile per wavefront, 8-byte loads at a 56-byte stride (what the screening pass does)
__global__ __launch_bounds__(256) void rec_once(const double *__restrict__ th, double *__restrict__ x, int *__restrict__ f, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < NT; t++) s += th[i * NT + t];
    x[i] = s; f[i] = 1;
}
// B: the same, T tiles per wavefront with the next tile's loads in flight (what the one-launch kernel's streamers do)
template <int DEPTH>
__global__ __launch_bounds__(256) void rec_loop(const double *__restrict__ th, double *__restrict__ x, int *__restrict__ f, long long n, int tiles) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long t0 = wave * tiles, ntile = (n + 63) / 64;
    double buf[DEPTH][NT];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        const long long i = (t0 + d) * 64 + lane;
#pragma unroll
        for (int t = 0; t < NT; t++) buf[d][t] = (t0 + d < ntile && d < tiles && i < n) ? th[i * NT + t] : 0.0;
    }
    for (int k = 0; k < tiles; k += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const long long tl = t0 + k + d;
            if (k + d >= tiles || tl >= ntile) break;
            const long long i = tl * 64 + lane;
            double s = 0.0;
#pragma unroll
            for (int t = 0; t < NT; t++) s += buf[d][t];
            const long long j = (tl + DEPTH) * 64 + lane;
            const bool more = k + d + DEPTH < tiles && tl + DEPTH < ntile && j < n;
#pragma unroll
            for (int t = 0; t < NT; t++) buf[d][t] = more ? th[j * NT + t] : 0.0;
            if (i < n) { x[i] = s; f[i] = 1; }
        }
    }
}
// B': as B, but the tiles of a wavefront are spread over the batch (wave w: tiles w, w + W, w + 2W, ...): at any time
// the resident wavefronts read one contiguous window that sweeps forward, as the launch order of A makes them do
__global__ __launch_bounds__(256) void rec_loop_spread(const double *__restrict__ th, double *__restrict__ x, int *__restrict__ f, long long n, int tiles) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), W = (long long)gridDim.x * 4, ntile = (n + 63) / 64;
    double buf[NT];
    {
        const long long i = wave * 64 + lane;
#pragma unroll
        for (int t = 0; t < NT; t++) buf[t] = (wave < ntile && i < n) ? th[i * NT + t] : 0.0;
    }
    for (int k = 0; k < tiles; k++) {
        const long long tl = wave + k * W;
        if (tl >= ntile) break;
        const long long i = tl * 64 + lane;
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < NT; t++) s += buf[t];
        const long long j = (tl + W) * 64 + lane;
        const bool more = k + 1 < tiles && tl + W < ntile && j < n;
#pragma unroll
        for (int t = 0; t < NT; t++) buf[t] = more ? th[j * NT + t] : 0.0;
        if (i < n) { x[i] = s; f[i] = 1; }
    }
}
// P: B made pipelinable for the hardware's in-order memory counter: two register sets used alternately, the next
// tile's loads issued BEFORE the current one is consumed and unconditionally (past the end: the current tile again),
// the same number of store instructions on every path (lanes with nothing to store write to a dummy slot) -- the
// compiler can then wait with vmcnt(N > 0) instead of draining every load AND store at the top of each iteration
__global__ __launch_bounds__(256) void rec_pp(const double *__restrict__ th, double *__restrict__ x, int *__restrict__ f, long long n, int tiles,
                                              double *__restrict__ dx, int *__restrict__ df) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), ntile = (n + 63) / 64;
    long long tl = wave * tiles;
    if (tl >= ntile) return;
    const long long last = (tl + tiles < ntile ? tl + tiles : ntile) - 1;
    double a[NT], b[NT];
    auto ld = [&](long long t_, double (&dst)[NT]) {
        long long i = t_ * 64 + lane; i = i < n ? i : n - 1;
#pragma unroll
        for (int t = 0; t < NT; t++) dst[t] = th[i * NT + t];
    };
    auto st = [&](long long t_, const double (&src)[NT]) {
        const long long i = t_ * 64 + lane;
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < NT; t++) s += src[t];
        double *px = i < n ? x + i : dx + lane;
        int *pf = i < n ? f + i : df + lane;
        *px = s; *pf = 1;
    };
    ld(tl, a);
    for (;;) {
        ld(tl < last ? tl + 1 : tl, b);
        st(tl, a);
        if (tl >= last) break;
        tl++;
        ld(tl < last ? tl + 1 : tl, a);
        st(tl, b);
        if (tl >= last) break;
        tl++;
    }
}
// D: T tiles per wavefront, ALL their loads issued up front, then the sums and stores (no loop-carried wait)
template <int T>
__global__ __launch_bounds__(256) void rec_upfront(const double *__restrict__ th, double *__restrict__ x, int *__restrict__ f, long long n) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    double buf[T][NT];
#pragma unroll
    for (int d = 0; d < T; d++) {
        const long long i = (wave * T + d) * 64 + lane, ic = i < n ? i : n - 1;
#pragma unroll
        for (int t = 0; t < NT; t++) buf[d][t] = th[ic * NT + t];
    }
#pragma unroll
    for (int d = 0; d < T; d++) {
        const long long i = (wave * T + d) * 64 + lane;
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < NT; t++) s += buf[d][t];
        if (i < n) { x[i] = s; f[i] = 1; }
    }
}
// E: A with the register budget of the one-launch kernel (three wavefronts per SIMD)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void rec_once_occ3(const double *__restrict__ th, double *__restrict__ x, int *__restrict__ f, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < NT; t++) s += th[i * NT + t];
    x[i] = s; f[i] = 1;
}
// C: 16-byte pieces in address order (a plain copy's read shape), grid-stride
__global__ __launch_bounds__(256) void flat16(const double2 *__restrict__ th, double *__restrict__ x, int *__restrict__ f, long long n) {
    const long long np = n * NT / 2, stride = (long long)gridDim.x * 256;
    double s = 0.0;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < np; p += stride) { const double2 v = th[p]; s += v.x + v.y; }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { x[i] = s; f[i] = 1; }
}

int main() {
    const long long n = 1000000;
    const int NB = 6;
    std::vector<double *> th(NB);
    double *x; int *f;
    for (auto &p : th) { CK(hipMalloc(&p, sizeof(double) * n * NT)); CK(hipMemset(p, 0, sizeof(double) * n * NT)); }
    CK(hipMalloc(&x, sizeof(double) * n * NB)); CK(hipMalloc(&f, sizeof(int) * n * NB));
    double *dx; int *df; CK(hipMalloc(&dx, 64 * sizeof(double))); CK(hipMalloc(&df, 64 * sizeof(int)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char *name, auto launch) -> int {
        for (int k = 0; k < 12; k++) launch(k % NB);
        CK(hipDeviceSynchronize());
        double tot = 0.0; float best = 1e9f;
        const int reps = 60;
        for (int k = 0; k < reps; k++) {
            CK(hipEventRecord(a, 0)); launch(k % NB); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); tot += ms; best = ms < best ? ms : best;
        }
        std::printf("%-44s avg %6.2f us  min %6.2f us  (%.2f TB/s of 68 MB at the average)\n", name, 1e3 * tot / reps, 1e3 * best, 68e6 / (tot / reps * 1e-3) / 1e12);
        return 0;
    };
    const unsigned g1 = (unsigned)((n + 255) / 256);
    if (run("A record per lane, one tile per wave", [&](int k) { hipLaunchKernelGGL(rec_once, dim3(g1), dim3(256), 0, 0, th[k], x + k * n, f + k * n, n); })) return 1;
    for (int tiles : {2, 4, 7, 21}) {
        const unsigned g = (unsigned)(((n + 63) / 64 + 4 * tiles - 1) / (4 * tiles));
        char nm[96];
        std::snprintf(nm, sizeof nm, "B %2d tiles per wave, 1 ahead (%u workgroups)", tiles, g);
        if (run(nm, [&](int k) { hipLaunchKernelGGL(rec_loop<1>, dim3(g), dim3(256), 0, 0, th[k], x + k * n, f + k * n, n, tiles); })) return 1;
        std::snprintf(nm, sizeof nm, "B' %2d tiles per wave, spread over the batch", tiles);
        if (run(nm, [&](int k) { hipLaunchKernelGGL(rec_loop_spread, dim3(g), dim3(256), 0, 0, th[k], x + k * n, f + k * n, n, tiles); })) return 1;
        std::snprintf(nm, sizeof nm, "P %2d tiles per wave, pipelined for vmcnt", tiles);
        if (run(nm, [&](int k) { hipLaunchKernelGGL(rec_pp, dim3(g), dim3(256), 0, 0, th[k], x + k * n, f + k * n, n, tiles, dx, df); })) return 1;
        std::snprintf(nm, sizeof nm, "B %2d tiles per wave, 2 ahead", tiles);
        if (run(nm, [&](int k) { hipLaunchKernelGGL(rec_loop<2>, dim3(g), dim3(256), 0, 0, th[k], x + k * n, f + k * n, n, tiles); })) return 1;
    }
    if (run("D 2 tiles per wave, loads up front", [&](int k) { hipLaunchKernelGGL(rec_upfront<2>, dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, th[k], x + k * n, f + k * n, n); })) return 1;
    if (run("D 4 tiles per wave, loads up front", [&](int k) { hipLaunchKernelGGL(rec_upfront<4>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, th[k], x + k * n, f + k * n, n); })) return 1;
    if (run("E one tile per wave, 3 wavefronts per SIMD", [&](int k) { hipLaunchKernelGGL(rec_once_occ3, dim3(g1), dim3(256), 0, 0, th[k], x + k * n, f + k * n, n); })) return 1;
    for (unsigned g : {768u, 2048u, 4096u}) {
        char nm[96];
        std::snprintf(nm, sizeof nm, "C 16-byte pieces in address order, %u workgroups", g);
        if (run(nm, [&](int k) { hipLaunchKernelGGL(flat16, dim3(g), dim3(256), 0, 0, reinterpret_cast<const double2 *>(th[k]), x + k * n, f + k * n, n); })) return 1;
    }
    if (run("empty kernel (event-to-event floor)", [&](int k) { hipLaunchKernelGGL(rec_once, dim3(1), dim3(256), 0, 0, th[k], x, f, 0ll); })) return 1;
    return 0;
}
