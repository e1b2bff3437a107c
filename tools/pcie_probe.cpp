// Probe: pageable vs registered host memory transfer rates and the cost of hipHostRegister.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t nb = 56ull << 20;
    char *h = (char *)malloc(nb), *h2 = (char *)malloc(nb);
    memset(h, 1, nb); memset(h2, 2, nb);
    void *d; hipMalloc(&d, nb);
    hipMemcpy(d, h, nb, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now(); hipMemcpy(d, h, nb, hipMemcpyHostToDevice); double t1 = now();
        hipMemcpy(h2, d, nb, hipMemcpyDeviceToHost); double t2 = now();
        printf("pageable H2D %.2f GB/s  D2H %.2f GB/s\n", nb / (t1 - t0) / 1e9, nb / (t2 - t1) / 1e9);
    }
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now(); hipError_t e = hipHostRegister(h, nb, hipHostRegisterDefault); double t1 = now();
        hipMemcpy(d, h, nb, hipMemcpyHostToDevice); double t2 = now();
        hipMemcpy(d, h, nb, hipMemcpyHostToDevice); double t3 = now();
        hipHostUnregister(h); double t4 = now();
        printf("register %.3f ms (err %d)  H2D first %.2f GB/s second %.2f GB/s  unregister %.3f ms\n", 1e3 * (t1 - t0), (int)e,
               nb / (t2 - t1) / 1e9, nb / (t3 - t2) / 1e9, 1e3 * (t4 - t3));
    }
    void *p; hipHostMalloc(&p, nb, hipHostMallocDefault);
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now(); memcpy(p, h, nb); double t1 = now();
        hipMemcpy(d, p, nb, hipMemcpyHostToDevice); double t2 = now();
        hipMemcpy(p, d, nb, hipMemcpyDeviceToHost); double t3 = now();
        printf("cpu memcpy to pinned %.2f GB/s  pinned H2D %.2f GB/s  pinned D2H %.2f GB/s\n", nb / (t1 - t0) / 1e9, nb / (t2 - t1) / 1e9, nb / (t3 - t2) / 1e9);
    }
    return 0;
}
