// Diagnostic: accuracy of v_rcp_f64 and of the refinements built on it (pt_device.h rcp_nr), against the
// host's correctly rounded 1 / b, in ulp of the result, over 4M values spread over 600 binades.
//   hipcc --offload-arch=gfx950 -O2 -o tools/rcp_probe.exe tools/rcp_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double *b, double *raw, double *nr2, double *cubic, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = b[i];
    double r = __builtin_amdgcn_rcp(x);
    raw[i] = r;
    double e = __builtin_fma(-x, r, 1.0);
    double r1 = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r1, 1.0);
    nr2[i] = __builtin_fma(r1, e, r1);
    e = __builtin_fma(-x, r, 1.0);
    const double t = __builtin_fma(e, e, e);
    cubic[i] = __builtin_fma(r, t, r);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> b(n), raw(n), nr2(n), cub(n);
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> u(1.0, 2.0);
    std::uniform_int_distribution<int> ex(-300, 300);
    for (int i = 0; i < n; i++) b[i] = std::ldexp(u(g), ex(g)) * ((i & 1) ? -1 : 1);
    double *db, *d1, *d2, *d3;
    hipMalloc(&db, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
    hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(db, d1, d2, d3, n);
    hipMemcpy(raw.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(nr2.data(), d2, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(cub.data(), d3, n * 8, hipMemcpyDeviceToHost);
    double m1 = 0, m2 = 0, m3 = 0;
    for (int i = 0; i < n; i++) {
        const long double ex1 = 1.0L / (long double)b[i];
        const double ulp = std::fabs(std::nextafter((double)ex1, INFINITY) - (double)ex1);
        m1 = std::fmax(m1, (double)(fabsl((long double)raw[i] - ex1) / ulp));
        m2 = std::fmax(m2, (double)(fabsl((long double)nr2[i] - ex1) / ulp));
        m3 = std::fmax(m3, (double)(fabsl((long double)cub[i] - ex1) / ulp));
    }
    printf("max error in ulp over %d values: v_rcp_f64 %.3g (2^%.1f), two Newton steps %.3f, one third-order step %.3f\n", n, m1,
           std::log2(m1), m2, m3);
    return 0;
}
