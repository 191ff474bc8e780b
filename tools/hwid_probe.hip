// Diagnostic: which SIMD does wave w of a workgroup land on?  (HW_REG_HW_ID, gfx9 layout:
// wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...)
// hipcc --offload-arch=gfx950 -O2 -o hwid_probe tools/hwid_probe.hip && ./hwid_probe [waves]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void probe(unsigned *out) {
    const unsigned id = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
    if ((threadIdx.x & 63) == 0)
        out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}
int main(int argc, char **argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 8, blocks = argc > 2 ? atoi(argv[2]) : 4;
    unsigned *d, h[1024];
    hipMalloc(&d, sizeof(h));
    probe<<<blocks, waves * 64>>>(d);
    hipMemcpy(h, d, sizeof(unsigned) * waves * blocks, hipMemcpyDeviceToHost);
    for (int b = 0; b < blocks; b++) {
        printf("block %d:", b);
        for (int w = 0; w < waves; w++) {
            const unsigned v = h[b * waves + w];
            printf("  w%d=simd%u/slot%u/cu%u", w, (v >> 4) & 3, v & 15, (v >> 8) & 15);
        }
        printf("\n");
    }
    return 0;
}
