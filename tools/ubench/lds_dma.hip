// lds_dma.hip -- what global_load_lds_{dword,ubyte} write where (gfx950): lanes 0 .. 7 of every wave load one dword / one byte each
// straight into LDS at a wave-uniform base; the kernel then dumps the LDS words.  Expected: lane L's data at base + 4 L, a byte
// zero-extended to a dword, lanes that are switched off write nothing.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k(const uint32_t *g, const uint8_t *gb, uint32_t *out) {
    __shared__ uint32_t words[4][16];
    __shared__ uint32_t bytes[4][16];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane < 16) { words[wv][lane] = 0xDEADBEEFu; bytes[wv][lane] = 0xDEADBEEFu; }
    __syncthreads();
    if (lane < 8) {
        __builtin_amdgcn_global_load_lds(g + wv * 100 + lane, &words[wv][8], 4, 0, 0);      // lanes 0..7 -> slots 8..15 ?
        __builtin_amdgcn_global_load_lds(gb + wv * 100 + lane, &bytes[wv][0], 1, 0, 0);     // lanes 0..7 -> slots 0..7 ?
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (lane < 16) { out[wv * 32 + lane] = words[wv][lane]; out[wv * 32 + 16 + lane] = bytes[wv][lane]; }
}
int main() {
    uint32_t h[1024]; uint8_t hb[1024];
    for (int i = 0; i < 1024; i++) { h[i] = 0x1000u + i; hb[i] = (uint8_t)(i + 1); }
    uint32_t *g, *out; uint8_t *gb;
    CK(hipMalloc(&g, sizeof h)); CK(hipMalloc(&gb, sizeof hb)); CK(hipMalloc(&out, 128 * 4));
    CK(hipMemcpy(g, h, sizeof h, hipMemcpyHostToDevice)); CK(hipMemcpy(gb, hb, sizeof hb, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, g, gb, out);
    uint32_t o[128]; CK(hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost));
    for (int w = 0; w < 4; w++) {
        printf("wave %d words:", w); for (int i = 0; i < 16; i++) printf(" %x", o[w * 32 + i]); printf("\n");
        printf("wave %d bytes:", w); for (int i = 0; i < 16; i++) printf(" %x", o[w * 32 + 16 + i]); printf("\n");
    }
    return 0;
}
