// Diagnostic: issue cost of single VALU instructions on gfx950, one workgroup of 16 waves (4 per SIMD) on one CU,
// eight independent chains per lane: the pipe is full, so the figure is cycles of SIMD issue per wave-instruction
// relative to v_fma_f64 (= 1.00).   hipcc --offload-arch=gfx950 -O2 -o tools/issue_probe.exe tools/issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void probe(double *out, int iters, unsigned long long *cycles) {
    constexpr int CH = 8;
    double v[CH];
    int w[CH];
    for (int j = 0; j < CH; j++) {
        v[j] = 1.0 + threadIdx.x * 1e-9 + j;
        w[j] = threadIdx.x + j;
    }
    const double a = 1.0000001, b = 1e-9;
    const int mask = 0x000FFFFF, bias = 0x3FF00000;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < CH; j++) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[j]) : "v"(a), "v"(b));
            if (OP == 1) asm volatile("v_rcp_f64 %0, %0" : "+v"(v[j]));
            if (OP == 2) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(v[j]));
            if (OP == 3) asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(w[j]) : "v"(v[j]));
            if (OP == 4) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(w[j]) : "s"(mask), "v"(bias));
            if (OP == 5) asm volatile("v_bfe_u32 %0, %0, 20, 11" : "+v"(w[j]));
            if (OP == 6) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v[j]) : "v"(w[j]));
            if (OP == 7) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[j]) : "v"(a));
            if (OP == 8) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(w[j]) : "v"(mask), "v"(bias));
            if (OP == 9) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[j]) : "v"(b));
            if (OP == 10) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(v[j]) : "v"(w[j]));
            if (OP == 11) asm volatile("v_rsq_f64 %0, %0" : "+v"(v[j]));
            if (OP == 12) asm volatile("v_log_f32 %0, %0" : "+v"(w[j]));
            if (OP == 13) asm volatile("v_sqrt_f64 %0, %0" : "+v"(v[j]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int j = 0; j < CH; j++) s += v[j] + w[j];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}
template <int OP>
double run(const char *name, double ref) {
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 8);
    const int iters = 2000;
    probe<OP><<<1, 1024>>>(out, iters, cyc);
    hipDeviceSynchronize();
    probe<OP><<<1, 1024>>>(out, iters, cyc);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / (iters * 8) / 4; // per SIMD issue slot
    printf("%-22s %.3f ticks per wave-instruction and SIMD  (%.2f x v_fma_f64)\n", name, per, ref > 0 ? per / ref : 1.0);
    hipFree(out); hipFree(cyc);
    return per;
}
int main() {
    const double f = run<0>("v_fma_f64", 0);
    run<7>("v_mul_f64", f); run<9>("v_add_f64", f); run<1>("v_rcp_f64", f); run<11>("v_rsq_f64", f); run<13>("v_sqrt_f64", f);
    run<2>("v_frexp_mant_f64", f); run<3>("v_frexp_exp_i32_f64", f); run<6>("v_ldexp_f64", f); run<10>("v_cvt_f64_i32", f);
    run<4>("v_and_or_b32", f); run<5>("v_bfe_u32", f); run<8>("v_add3_u32", f); run<12>("v_log_f32", f);
    return 0;
}
