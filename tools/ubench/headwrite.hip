// headwrite.hip -- what does writing the bucket heads cost by itself?  (round 5: three different finish kernels of the index build
// all take 3.2-3.4 ms per strand at C3 -- 3 GB of records in, 8.6 GB of 16-byte heads out -- whatever their instruction count and
// occupancy.)  2^29 heads of 16 bytes:
//   contiguous      one table per strand: 8.6 GB in one run
//   pair halves     the pair table as ONE strand's build writes it: 64 bytes written, 64 bytes skipped (17.2 GB span)
//   pair both       both strands' halves by one kernel: whole 128-byte lines (17.2 GB)
//   pair 2 blocks   both strands' halves by TWO blocks that the dispatcher puts on one XCD one after the other (block ids b and
//                   b + 8): do the halves of a line meet in that XCD's L2 and leave as one 128-byte write?
//   + read          each of the above while the same kernel also streams 3 GB in (the records)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
// mode 0 contiguous, 1 pair halves (strand 0), 2 pair both
__global__ void __launch_bounds__(512) k(ulonglong2 *__restrict__ head, uint64_t nheads, int mode, const ulonglong2 *__restrict__ rd, uint64_t nrd, uint64_t *out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (uint64_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    const uint64_t per = nheads / gridDim.x;            // a block owns a contiguous range of heads (a "partition")
    const uint64_t h0 = (uint64_t)blockIdx.x * per;
    for (uint64_t b = threadIdx.x; b < per; b += blockDim.x) {
        const uint64_t h = h0 + b;
        const ulonglong2 v = make_ulonglong2(h, ~h);
        if (mode == 0) head[h] = v;
        else if (mode == 1) head[((h & ~3ull) << 1) | (h & 3ull)] = v;
        else if (mode == 2) { head[2 * h] = v; head[2 * h + 1] = v; }
    }
    if (mode == 3) {   // 2 * 65536 blocks: x = b % 8 labels the XCD, j = b / 8: strand = j % 2, partition = (j / 2) * 8 + x
        const uint64_t x = blockIdx.x & 7u, j = blockIdx.x >> 3, strand = j & 1u, part = (j >> 1) * 8 + x;
        const uint64_t per2 = nheads / (gridDim.x / 2);
        for (uint64_t b = threadIdx.x; b < per2; b += blockDim.x) {
            const uint64_t h = part * per2 + b;
            head[(((h & ~3ull) << 1) | (h & 3ull)) + 4 * strand] = make_ulonglong2(h, ~h);
        }
    }
    if (rd) for (uint64_t i = tid; i < nrd; i += nt) { const ulonglong2 q = rd[i]; acc += q.x ^ q.y; }
    if (acc == 0x1234567ull) out[0] = acc;
}
static void run(const char *name, ulonglong2 *head, uint64_t nheads, int mode, const ulonglong2 *rd, uint64_t nrd, uint64_t *out, double gb, int grid = 65536) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, head, nheads, mode, rd, nrd, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-52s %6.2f ms  %6.2f TB/s\n", name, best, gb / best); fflush(stdout);
}
int main() {
    const uint64_t nheads = 1ull << 29;
    ulonglong2 *head, *rd; uint64_t *out;
    CK(hipMalloc(&head, nheads * 32)); CK(hipMalloc(&rd, 3ull << 30)); CK(hipMalloc(&out, 64));
    CK(hipMemset(head, 0, nheads * 32)); CK(hipMemset(rd, 1, 3ull << 30));
    const uint64_t nrd = (3ull << 30) / 16;
    run("contiguous 8.6 GB", head, nheads, 0, nullptr, 0, out, 8.59);
    run("pair halves: 64 B on, 64 B off (8.6 GB)", head, nheads, 1, nullptr, 0, out, 8.59);
    run("pair both: whole 128-B lines (17.2 GB)", head, nheads, 2, nullptr, 0, out, 17.18);
    run("contiguous 8.6 GB + 3.2 GB read", head, nheads, 0, rd, nrd, out, 8.59 + 3.22);
    run("pair halves 8.6 GB + 3.2 GB read", head, nheads, 1, rd, nrd, out, 8.59 + 3.22);
    run("pair both 17.2 GB + 6.4 GB read", head, nheads, 2, rd, 2 * nrd > (3ull << 30) / 16 ? nrd : 2 * nrd, out, 17.18 + 3.22);
    run("pair, 2 blocks on one XCD per partition (17.2 GB)", head, nheads, 3, nullptr, 0, out, 17.18, 131072);
    run("pair, 2 blocks on one XCD (17.2 GB) + 6.4 GB read", head, nheads, 3, rd, nrd, out, 17.18 + 3.22, 131072);
    return 0;
}
