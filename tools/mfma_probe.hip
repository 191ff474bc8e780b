// Diagnostic (not product code): which lanes does an fp64 MFMA with a ones matrix add up?
// For every source lane s, A (or B) carries 1.0 in lane s and 0 elsewhere, the other operand is all
// ones; the output says which (lane, register) hold that lane's contribution.
//   hipcc --offload-arch=gfx950 -O2 -o tools/mfma_probe.exe tools/mfma_probe.hip && tools/mfma_probe.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void probe(double *out4a, double *out4b, double *out16a, double *out16b) {
    const int lane = threadIdx.x;
    for (int s = 0; s < 64; s++) {
        const double e = lane == s ? 1.0 : 0.0;
        const double d4a = __builtin_amdgcn_mfma_f64_4x4x4f64(e, 1.0, 0.0, 0, 0, 0);
        const double d4b = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, e, 0.0, 0, 0, 0);
        out4a[s * 64 + lane] = d4a;
        out4b[s * 64 + lane] = d4b;
        v4d z = {0, 0, 0, 0};
        const v4d da = __builtin_amdgcn_mfma_f64_16x16x4f64(e, 1.0, z, 0, 0, 0);
        const v4d db = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, e, z, 0, 0, 0);
        for (int r = 0; r < 4; r++) {
            out16a[(s * 64 + lane) * 4 + r] = da[r];
            out16b[(s * 64 + lane) * 4 + r] = db[r];
        }
    }
}

static void show(const char *name, const std::vector<double> &o, int regs) {
    printf("%s: source lane -> output lanes (reg) holding it\n", name);
    for (int s = 0; s < 64; s++) {
        printf("  s=%2d:", s);
        int n = 0;
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < regs; r++)
                if (o[(s * 64 + l) * regs + r] != 0) {
                    if (n < 20)
                        printf(" %d%s", l, regs > 1 ? (r == 0 ? "a" : r == 1 ? "b" : r == 2 ? "c" : "d") : "");
                    n++;
                }
        printf("  [%d]\n", n);
    }
}

int main() {
    double *d4a, *d4b, *d16a, *d16b;
    hipMalloc(&d4a, 64 * 64 * 8);
    hipMalloc(&d4b, 64 * 64 * 8);
    hipMalloc(&d16a, 64 * 64 * 4 * 8);
    hipMalloc(&d16b, 64 * 64 * 4 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d4a, d4b, d16a, d16b);
    if (hipDeviceSynchronize() != hipSuccess) {
        printf("kernel failed\n");
        return 1;
    }
    std::vector<double> a(64 * 64), b(64 * 64), c(64 * 64 * 4), d(64 * 64 * 4);
    hipMemcpy(a.data(), d4a, a.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d4b, b.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d16a, c.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(d.data(), d16b, d.size() * 8, hipMemcpyDeviceToHost);
    show("4x4x4 data in A", a, 1);
    show("4x4x4 data in B", b, 1);
    show("16x16x4 data in A", c, 4);
    show("16x16x4 data in B", d, 4);
    return 0;
}
