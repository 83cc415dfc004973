// Probe of the register layout of v_mfma_f64_16x16x4_f64 on gfx950 (used by k_time_modes_mfma).
// build: hipcc --offload-arch=gfx950 -O2 mfma_f64_layout.hip -o mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4 __attribute__((ext_vector_type(4)));
__global__ void probe(double *out) {
    const int l = threadIdx.x;
    // hypothesis for the inputs: A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16]
    const int i = l % 16, k = l / 16, j = l % 16;
    v4 acc = {0, 0, 0, 0};
    // pass 1: D[i][j] = i      (A[i][0] = i, B[0][j] = 1)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(k == 0 ? (double)i : 0.0, k == 0 ? 1.0 : 0.0, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
    // pass 2: D[i][j] = j      (A[i][0] = 1, B[0][j] = j)
    v4 acc2 = {0, 0, 0, 0};
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(k == 0 ? 1.0 : 0.0, k == 0 ? (double)j : 0.0, acc2, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[256 + l * 4 + r] = acc2[r];
    // pass 3: which k do the operands of lane group l/16 multiply?  A[i][k] = 1, B[k][j] = 10^k -> D = sum over k
    v4 acc3 = {0, 0, 0, 0};
    double p = 1.0;
    for (int q = 0; q < k; ++q) p *= 10.0;
    acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, p, acc3, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[512 + l * 4 + r] = acc3[r];
}
int main() {
    double *d, h[768];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("row index i held by (lane, r):\n");
    for (int l = 0; l < 64; l += 1) { if (l % 16 == 0 || l % 16 == 1) printf("lane %2d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
    printf("column index j held by (lane, r):\n");
    for (int l = 0; l < 64; l += 1) { if (l % 16 <= 1 || l % 16 == 15) printf("lane %2d: %g %g %g %g\n", l, h[256+l*4], h[256+l*4+1], h[256+l*4+2], h[256+l*4+3]); }
    printf("sum over k of 10^k (1111 = all four k used once): lane 0: %g  lane 17: %g\n", h[512], h[512 + 17 * 4]);
    return 0;
}
