// realloc.hip -- does re-allocating a lot of device memory get slow after a D2H copy into a (still live) pageable host buffer?
// (seen in tools/boundary_c3.py: the second context of a process needed 1.3 s for its 35 GB of index buffers when the result
// arrays of the first were reused.)   build: hipcc -O3 --offload-arch=gfx950 tools/ubench/realloc.hip -o tools/ubench/realloc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t GB = 1ull << 30, nbuf = 6, each = 6 * GB, hostn = 1 * GB;
    const int reuse_host = argc > 1 ? atoi(argv[1]) : 1;
    char *host = (char *)malloc(hostn);
    memset(host, 1, hostn);
    for (int round = 0; round < 4; round++) {
        std::vector<void *> bufs(nbuf);
        double t = now();
        for (auto &b : bufs) CK(hipMalloc(&b, each));
        const double ta = now() - t;
        t = now();
        for (auto &b : bufs) CK(hipMemset(b, round, each));
        CK(hipDeviceSynchronize());
        const double tm = now() - t;
        if (!reuse_host) { free(host); host = (char *)malloc(hostn); memset(host, 1, hostn); }
        t = now();
        CK(hipMemcpy(host, bufs[0], hostn, hipMemcpyDeviceToHost));
        const double td = now() - t;
        t = now();
        for (auto &b : bufs) CK(hipFree(b));
        const double tf = now() - t;
        printf("round %d: hipMalloc %zu x %zu GB %.1f ms, memset all %.1f ms, D2H 1 GB into %s host buffer %.1f ms, hipFree %.1f ms\n", round, nbuf, each / GB, ta * 1e3, tm * 1e3,
               reuse_host ? "the SAME" : "a NEW", td * 1e3, tf * 1e3);
    }
    return 0;
}
