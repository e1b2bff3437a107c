// Micro-probe (gfx950): semantics and issue cost of the DP-ALU DPP broadcast `row_newbcast` that the row kernel's
// serial chains are built on -- v_fmac_f64_dpp (inline asm) against v_mov_b64_dpp + v_fma_f64 (builtin) against the
// wavefront kernel's 2 x v_readlane + v_fma_f64.  Build: hipcc --offload-arch=gfx950 -O3 -o dpp_probe dpp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e__), __LINE__); exit(1); } } while (0)

template <int T> __device__ __forceinline__ double bc(double v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x150 + T, 0xF, 0xF, false);
}
template <int T> __device__ __forceinline__ void fmac_bc(double &acc, double src, double mul) {
    // acc += bcast_T(src) * mul
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(T));
}
template <int T> __device__ __forceinline__ void fmac_bc_neg0(double &acc, double src, double mul) {
    // one wait state here + the one hipcc puts behind every asm statement
    asm("s_nop 0\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(T));
}
template <int T> __device__ __forceinline__ void fmac_bc_neg(double &acc, double src, double mul) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(T));
}

// semantics: out[lane] = v[lane] - L[lane] * v[16 * (lane / 16) + T]
template <int MODE>
__global__ void sem(const double *v, const double *L, double *out) {
    const int l = threadIdx.x;
    double a = v[l];
    const double s = v[l], m = L[l];
    if (MODE == 0) a = __builtin_fma(-m, bc<5>(s), a);
    if (MODE == 1) fmac_bc_neg<5>(a, s, m);
    if (MODE == 2) { fmac_bc_neg<15>(a, s, m); }
    if (MODE == 3) { double t = a; fmac_bc_neg<0>(t, t, m); a = t; }   // source and accumulator the same register
    out[l] = a;
}

template <int T0> struct Chain {
    template <int MODE> static __device__ __forceinline__ void run(double &v, const double *Lr) {
        if constexpr (MODE == 0) v = __builtin_fma(-Lr[T0], bc<T0>(v), v);
        else if constexpr (MODE == 1) fmac_bc_neg<T0>(v, v, Lr[T0]);
        else if constexpr (MODE == 3) fmac_bc_neg0<T0>(v, v, Lr[T0]);
        else {
            const int lo = __builtin_amdgcn_readlane(__double2loint(v), T0), hi = __builtin_amdgcn_readlane(__double2hiint(v), T0);
            v = __builtin_fma(-Lr[T0], __hiloint2double(hi, lo), v);
        }
        if constexpr (T0 + 1 < 16) Chain<T0 + 1>::template run<MODE>(v, Lr);
    }
};

// timing: `reps` sweeps of 16 dependent chain steps per wavefront
template <int MODE>
__global__ __launch_bounds__(256) void timing(const double *Lg, double *out, int reps, long long *cyc) {
    const int l = threadIdx.x & 63;
    double Lr[16];
#pragma unroll
    for (int t = 0; t < 16; t++) Lr[t] = (l & 15) > t ? Lg[t * 64 + l] : 0.0;
    double v = 1.0 + 1e-3 * l;
    const long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        Chain<0>::template run<MODE>(v, Lr);
        v = v * 0.999;
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    double hv[64], hL[64], ho[64];
    for (int i = 0; i < 64; i++) { hv[i] = 1.0 + 0.37 * i; hL[i] = 0.01 * (i + 1); }
    double *dv, *dL, *dout;
    CK(hipMalloc(&dv, 512)); CK(hipMalloc(&dL, 512)); CK(hipMalloc(&dout, 512));
    CK(hipMemcpy(dv, hv, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dL, hL, 512, hipMemcpyHostToDevice));
    int bad = 0;
    for (int mode = 0; mode < 4; mode++) {
        if (mode == 0) sem<0><<<1, 64>>>(dv, dL, dout);
        if (mode == 1) sem<1><<<1, 64>>>(dv, dL, dout);
        if (mode == 2) sem<2><<<1, 64>>>(dv, dL, dout);
        if (mode == 3) sem<3><<<1, 64>>>(dv, dL, dout);
        CK(hipMemcpy(ho, dout, 512, hipMemcpyDeviceToHost));
        const int T = mode == 2 ? 15 : (mode == 3 ? 0 : 5);
        int b = 0;
        for (int i = 0; i < 64; i++) {
            const double want = fma(-hL[i], hv[16 * (i / 16) + T], hv[i]);
            if (want != ho[i]) b++;
        }
        printf("semantics mode %d (row_newbcast:%d): %d mismatches of 64\n", mode, T, b);
        bad += b;
    }
    // timing
    std::vector<double> Lg(16 * 64);
    for (size_t i = 0; i < Lg.size(); i++) Lg[i] = 1e-3 * ((i * 7) % 13);
    double *dLg, *dbig; long long *dc;
    CK(hipMalloc(&dLg, Lg.size() * 8)); CK(hipMemcpy(dLg, Lg.data(), Lg.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&dbig, 8 * 256 * 4096)); CK(hipMalloc(&dc, 8));
    const int reps = 20000;
    for (int mode = 0; mode < 4; mode++) {
        for (int wpb : {1, 2, 4, 8, 12, 16}) {      // wavefronts per workgroup = per CU (one workgroup per CU): 1/SIMD at 4, 2 at 8, ...
            hipEvent_t a, b;
            CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            for (int it = 0; it < 2; it++) {
                CK(hipEventRecord(a));
                if (mode == 0) timing<0><<<256, 64 * wpb>>>(dLg, dbig, reps, dc);
                if (mode == 1) timing<1><<<256, 64 * wpb>>>(dLg, dbig, reps, dc);
                if (mode == 2) timing<2><<<256, 64 * wpb>>>(dLg, dbig, reps, dc);
                if (mode == 3) timing<3><<<256, 64 * wpb>>>(dLg, dbig, reps, dc);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            }
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            long long c; CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
            printf("timing mode %d (%s) %2d wavefronts/CU: %.3f ms, %.2f clock64-ticks per chain step (wave 0), %.2f ns per step per wavefront\n", mode,
                   mode == 0 ? "mov_b64_dpp + fma" : (mode == 1 ? "s_nop 1 + fmac_f64_dpp" : (mode == 2 ? "2 readlane + fma" : "s_nop 0 + fmac_f64_dpp")), wpb, ms,
                   (double)c / (reps * 16.0), ms * 1e6 / (reps * 16.0));
        }
    }
    printf(bad ? "FAILED\n" : "OK\n");
    return bad ? 1 : 0;
}
