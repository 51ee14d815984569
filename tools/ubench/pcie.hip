// pcie.hip -- what the boundary of the matcher can count on: host->device and device->host copy rates
// for pageable, touched-pageable, registered (hipHostRegister) and pinned (hipHostMalloc) host memory, the cost of
// registering, and whether a pageable H2D copy overlaps with a kernel on another stream.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/pcie.hip -o tools/ubench/pcie     run: tools/ubench/pcie [MiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void spin(unsigned long long cycles, unsigned *out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = 1;
}

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 1024, n = mib << 20;
    void *d = nullptr;
    CK(hipMalloc(&d, n));
    CK(hipMemset(d, 1, n));
    char *pg = (char *)malloc(n);                       // untouched pageable
    char *pt = (char *)malloc(n); memset(pt, 2, n);     // touched pageable
    double t;
    t = now(); CK(hipMemcpy(pg, d, n, hipMemcpyDeviceToHost)); printf("D2H untouched pageable   %6.1f GB/s\n", n / (now() - t) / 1e9);
    t = now(); CK(hipMemcpy(pt, d, n, hipMemcpyDeviceToHost)); printf("D2H touched pageable     %6.1f GB/s\n", n / (now() - t) / 1e9);
    t = now(); CK(hipMemcpy(pt, d, n, hipMemcpyDeviceToHost)); printf("D2H touched pageable (2) %6.1f GB/s\n", n / (now() - t) / 1e9);
    t = now(); CK(hipMemcpy(d, pt, n, hipMemcpyHostToDevice)); printf("H2D touched pageable     %6.1f GB/s\n", n / (now() - t) / 1e9);
    t = now(); CK(hipHostRegister(pt, n, hipHostRegisterDefault)); printf("hipHostRegister          %6.1f ms for %zu MiB (%.1f GB/s)\n", (now() - t) * 1e3, mib, n / (now() - t) / 1e9);
    t = now(); CK(hipMemcpy(pt, d, n, hipMemcpyDeviceToHost)); printf("D2H registered           %6.1f GB/s\n", n / (now() - t) / 1e9);
    t = now(); CK(hipMemcpy(d, pt, n, hipMemcpyHostToDevice)); printf("H2D registered           %6.1f GB/s\n", n / (now() - t) / 1e9);
    t = now(); CK(hipHostUnregister(pt)); printf("hipHostUnregister        %6.1f ms\n", (now() - t) * 1e3);
    void *ph = nullptr;
    t = now(); CK(hipHostMalloc(&ph, n, hipHostMallocDefault)); printf("hipHostMalloc            %6.1f ms\n", (now() - t) * 1e3);
    t = now(); CK(hipMemcpy(ph, d, n, hipMemcpyDeviceToHost)); printf("D2H pinned               %6.1f GB/s\n", n / (now() - t) / 1e9);
    t = now(); CK(hipMemcpy(d, ph, n, hipMemcpyHostToDevice)); printf("H2D pinned               %6.1f GB/s\n", n / (now() - t) / 1e9);
    // several host threads copying disjoint slices of a touched pageable buffer at once
    for (int nt : {2, 4, 8}) {
        t = now();
        std::thread th[8];
        for (int k = 0; k < nt; k++) th[k] = std::thread([=]() { (void)hipMemcpy(pt + n / nt * k, (char *)d + n / nt * k, n / nt, hipMemcpyDeviceToHost); });
        for (int k = 0; k < nt; k++) th[k].join();
        printf("D2H touched pageable, %d threads %6.1f GB/s\n", nt, n / (now() - t) / 1e9);
    }
    // overlap: a kernel busy for ~20 ms on stream A; a pageable H2D copy issued meanwhile from the host
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    unsigned *flag;
    CK(hipMalloc(&flag, 4));
    int rate = 0;
    CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));   // kHz
    const unsigned long long cyc = (unsigned long long)rate * 20;          // 20 ms
    t = now(); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, sa, cyc, flag); CK(hipStreamSynchronize(sa)); const double tk = now() - t;
    t = now(); CK(hipMemcpyAsync(d, pg, n / 2, hipMemcpyHostToDevice, sb)); const double tissue = now() - t; CK(hipStreamSynchronize(sb)); const double tc = now() - t;
    t = now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, sa, cyc, flag);
    CK(hipMemcpyAsync(d, pg, n / 2, hipMemcpyHostToDevice, sb));
    CK(hipStreamSynchronize(sb));
    CK(hipStreamSynchronize(sa));
    printf("kernel alone %.1f ms, pageable H2D of %zu MiB alone %.1f ms (the call itself returned after %.1f ms), both at once %.1f ms\n", tk * 1e3, mib / 2, tc * 1e3, tissue * 1e3, (now() - t) * 1e3);
    // D2H on a second host thread while the main thread does H2D (full duplex?)
    t = now();
    std::thread down([=]() { (void)hipMemcpy(pt, (char *)d + n / 2, n / 2, hipMemcpyDeviceToHost); });
    CK(hipMemcpy(d, pg, n / 2, hipMemcpyHostToDevice));
    down.join();
    printf("H2D %zu MiB (main thread) + D2H %zu MiB (second thread) at once: %.1f ms = %.1f GB/s each way\n", mib / 2, mib / 2, (now() - t) * 1e3, n / 2 / (now() - t) / 1e9);
    return 0;
}
