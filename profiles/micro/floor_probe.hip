// Probe of launch/latency floors on MI355X for the PCG kernel shapes (sphere10k: V=10242, TP=32).
// hipcc --offload-arch=gfx950 -O3 -o floor_probe floor_probe.hip && ./floor_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_empty(int n) {}

template <int NB>
__global__ __launch_bounds__(NB) void k_stream(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ c, double* __restrict__ d, int n) {
    int i = blockIdx.x * NB + threadIdx.x;
    if (i < n) { double x = a[i], y = b[i]; c[i] = x + 0.5 * y; d[i] = x - y; }
}

// gather: 7 neighbours per row, rows of 32 doubles; col read from global directly
template <int NB>
__global__ __launch_bounds__(NB) void k_gather(const int* __restrict__ col, const double* __restrict__ val, const double* __restrict__ z, const double* __restrict__ p,
                                               double* __restrict__ pn, double* __restrict__ out, int V, double beta) {
    int e = blockIdx.x * NB + threadIdx.x;
    int v = e >> 5, t = e & 31;
    if (v >= V) return;
    double s = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        int u = col[v * 7 + j];
        double w = val[v * 7 + j];
        s += w * (z[u * 32 + t] + beta * p[u * 32 + t]);
    }
    pn[e] = z[e] + beta * p[e];
    out[e] = s;
}

template <int NB, int EPT>
__global__ __launch_bounds__(NB) void k_gather_ept(const int* __restrict__ col, const double* __restrict__ val, const double* __restrict__ z, const double* __restrict__ p,
                                                   double* __restrict__ pn, double* __restrict__ out, int V, double beta) {
    for (int q = 0; q < EPT; ++q) {
        int e = (blockIdx.x * EPT + q) * NB + threadIdx.x;
        int v = e >> 5, t = e & 31;
        if (v >= V) return;
        double s = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            int u = col[v * 7 + j];
            double w = val[v * 7 + j];
            s += w * (z[u * 32 + t] + beta * p[u * 32 + t]);
        }
        pn[e] = z[e] + beta * p[e];
        out[e] = s;
    }
}

// partial re-reduce prologue in two layouts
template <int NB, bool COL_MAJOR>
__global__ __launch_bounds__(NB) void k_prologue(const double* __restrict__ part, int G, double* out) {
    __shared__ double red[NB];
    __shared__ double tot[32];
    int tid = threadIdx.x, c = tid & 31, j = tid >> 5, J = NB >> 5;
    double s = 0;
    for (int g = j; g < G; g += J) s += COL_MAJOR ? part[c * G + g] : part[g * 32 + c];
    red[tid] = s;
    __syncthreads();
    if (j == 0) { double t = 0; for (int k = 0; k < J; ++k) t += red[c + 32 * k]; tot[c] = t; }
    __syncthreads();
    if (blockIdx.x == 0 && tid < 32) out[tid] = tot[tid];
}

template <typename F>
double timeit(F f, int reps, hipStream_t s) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) f();
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 1e3 * ms / reps;
}

int main(int argc, char** argv) {
    const int V = argc > 1 ? atoi(argv[1]) : 10242, TP = 32, N = V * TP;
    hipStream_t s; CK(hipStreamCreate(&s));
    double *a, *b, *c, *d, *val, *part, *o; int* col;
    CK(hipMalloc(&a, N * 8)); CK(hipMalloc(&b, N * 8)); CK(hipMalloc(&c, N * 8)); CK(hipMalloc(&d, N * 8));
    CK(hipMalloc(&val, V * 7 * 8)); CK(hipMalloc(&col, V * 7 * 4)); CK(hipMalloc(&part, 4096 * 32 * 8)); CK(hipMalloc(&o, 1024));
    CK(hipMemset(a, 0, N * 8)); CK(hipMemset(b, 0, N * 8)); CK(hipMemset(part, 0, 4096 * 32 * 8));
    std::vector<int> hc(V * 7); std::vector<double> hv(V * 7, 0.1);
    for (int v = 0; v < V; ++v) for (int j = 0; j < 7; ++j) { int u = v + (j - 3) * (V > 50000 ? 83 : 17); if (u < 0) u += V; if (u >= V) u -= V; hc[v * 7 + j] = u; }
    CK(hipMemcpy(col, hc.data(), V * 7 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(val, hv.data(), V * 7 * 8, hipMemcpyHostToDevice));
    const int reps = 2000;
    printf("empty  256thr x1284: %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(1284), dim3(256), 0, s, 0); }, reps, s));
    printf("empty 1024thr x 321: %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(321), dim3(1024), 0, s, 0); }, reps, s));
    printf("empty 1024thr x 321 lds20k: %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(321), dim3(1024), 20000, s, 0); }, reps, s));
    printf("stream 2r2w 256thr: %.2f us\n", timeit([&] { hipLaunchKernelGGL((k_stream<256>), dim3((N + 255) / 256), dim3(256), 0, s, a, b, c, d, N); }, reps, s));
    printf("stream 2r2w 1024thr: %.2f us\n", timeit([&] { hipLaunchKernelGGL((k_stream<1024>), dim3((N + 1023) / 1024), dim3(1024), 0, s, a, b, c, d, N); }, reps, s));
    printf("gather 256thr: %.2f us\n", timeit([&] { hipLaunchKernelGGL((k_gather<256>), dim3((N + 255) / 256), dim3(256), 0, s, col, val, a, b, c, d, V, 0.5); }, reps, s));
    printf("gather 1024thr: %.2f us\n", timeit([&] { hipLaunchKernelGGL((k_gather<1024>), dim3((N + 1023) / 1024), dim3(1024), 0, s, col, val, a, b, c, d, V, 0.5); }, reps, s));
    printf("gather 512thr: %.2f us\n", timeit([&] { hipLaunchKernelGGL((k_gather<512>), dim3((N + 511) / 512), dim3(512), 0, s, col, val, a, b, c, d, V, 0.5); }, reps, s));
    printf("gather 1024thr ept4: %.2f us\n", timeit([&] { hipLaunchKernelGGL((k_gather_ept<1024, 4>), dim3((N + 4095) / 4096), dim3(1024), 0, s, col, val, a, b, c, d, V, 0.5); }, reps, s));
    printf("gather 256thr ept4: %.2f us\n", timeit([&] { hipLaunchKernelGGL((k_gather_ept<256, 4>), dim3((N + 1023) / 1024), dim3(256), 0, s, col, val, a, b, c, d, V, 0.5); }, reps, s));
    for (int G : {328, 1024}) {
        printf("prologue col-major G=%d: %.2f us\n", G, timeit([&] { hipLaunchKernelGGL((k_prologue<1024, true>), dim3(321), dim3(1024), 0, s, part, G, o); }, reps, s));
        printf("prologue row-major G=%d: %.2f us\n", G, timeit([&] { hipLaunchKernelGGL((k_prologue<1024, false>), dim3(321), dim3(1024), 0, s, part, G, o); }, reps, s));
        printf("prologue row-major 256thr G=%d: %.2f us\n", G, timeit([&] { hipLaunchKernelGGL((k_prologue<256, false>), dim3(1284), dim3(256), 0, s, part, G, o); }, reps, s));
    }
    // alternating two dependent kernels (what a CG iteration does)
    printf("gather+stream pair: %.2f us per pair\n", timeit([&] {
        hipLaunchKernelGGL((k_gather<256>), dim3((N + 255) / 256), dim3(256), 0, s, col, val, a, b, c, d, V, 0.5);
        hipLaunchKernelGGL((k_stream<256>), dim3((N + 255) / 256), dim3(256), 0, s, c, d, a, b, N); }, reps, s));
    return 0;
}
