// Diagnostic: fp64 VALU issue rate / dependent latency on gfx950 for one CU.
// chains = independent FMA chains per lane, waves = wavefronts per workgroup (one workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS, int OP>
__global__ void probe(double *out, int iters, unsigned long long *cycles) {
    double v[CHAINS];
    for (int j = 0; j < CHAINS; j++) v[j] = 1.0 + threadIdx.x * 1e-9 + j;
    const double a = 1.0000001, b = 1e-9;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < CHAINS; j++) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[j]) : "v"(a), "v"(b));
            if (OP == 1) asm volatile("v_rndne_f64 %0, %0" : "+v"(v[j]));
            if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[j]) : "v"(a));
            if (OP == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[j]) : "v"(b));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int j = 0; j < CHAINS; j++) s += v[j];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}
template <int CHAINS, int OP>
void run(const char *name, int waves) {
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 8);
    const int iters = 2000;
    probe<CHAINS, OP><<<1, waves * 64>>>(out, iters, cyc);
    hipDeviceSynchronize();
    probe<CHAINS, OP><<<1, waves * 64>>>(out, iters, cyc);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per_wave_instr = (double)h / (iters * CHAINS);
    const int waves_per_simd = (waves + 3) / 4;
    printf("%-12s chains=%d waves=%2d (%d/SIMD): %.2f cycles per instruction per wave, %.2f per SIMD issue slot\n", name,
           CHAINS, waves, waves_per_simd, per_wave_instr, per_wave_instr / waves_per_simd);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int waves : {1, 4, 8, 16}) {
        run<1, 0>("v_fma_f64", waves); run<2, 0>("v_fma_f64", waves); run<4, 0>("v_fma_f64", waves); run<8, 0>("v_fma_f64", waves);
    }
    for (int waves : {4, 8}) { run<4, 1>("v_rndne_f64", waves); run<4, 2>("v_mul_f64", waves); run<4, 3>("v_add_f64", waves); }
    return 0;
}
