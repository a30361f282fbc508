// Microbenchmark: v_mfma_f64_4x4x4_4b_f64 on gfx950 -- pipe occupancy (alone and mixed with the
// 16x16x4 form, as the Hessian kernel would issue them) and the lane layout of A, B and C/D.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_4x4 mfma_f64_4x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error line %d\n", __LINE__); exit(1);} } while (0)

// MODE 0: NBIG 16x16x4 only; MODE 1: NSMALL 4x4x4 only; MODE 2: both per iteration
template <int NBIG, int NSMALL, int WG>
__global__ __launch_bounds__(WG) void k_rate(double* out, unsigned long long* clk, int iters, double a0, double b0) {
    f64x4 acc[NBIG > 0 ? NBIG : 1];
    double sm[NSMALL > 0 ? NSMALL : 1];
#pragma unroll
    for (int i = 0; i < NBIG; ++i) acc[i] = (f64x4){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NSMALL; ++i) sm[i] = 0.0;
    double a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NBIG; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NSMALL; ++i) sm[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, sm[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NBIG; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < NSMALL; ++i) s += sm[i];
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// one-hot probe: wave w = (la, lb): A = 1 in lane la, B = 1 in lane lb -> which D lanes are 1?
__global__ void k_probe(int* out) {
    int w = blockIdx.x;            // 0..4095
    int la = w >> 6, lb = w & 63, l = threadIdx.x;
    double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
    double c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[w * 64 + l] = c != 0.0 ? 1 : 0;
}

template <int NBIG, int NSMALL, int WG>
void run(int blocks, int iters, double* d, unsigned long long* clk) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate<NBIG, NSMALL, WG>), dim3(blocks), dim3(WG), 0, 0, d, clk, 10, 1.0, 2.0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rate<NBIG, NSMALL, WG>), dim3(blocks), dim3(WG), 0, 0, d, clk, iters, 1.0, 2.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double waves_per_simd = (double)blocks * (WG / 64) / 1024.0;
    double wall_cyc = ms * 1e-3 * 2.4e9 / ((double)iters * waves_per_simd);
    double flops = (double)blocks * (WG / 64) * iters * (NBIG * 2048.0 + NSMALL * 512.0);
    double ghz = (double)h[0] / (double)h[1] * 0.1;
    double cyc = (double)h[0] / ((double)iters * (waves_per_simd < 1 ? 1 : waves_per_simd));
    printf("big=%d small=%d wg=%d blocks=%d (%.1f waves/SIMD): %.3f ms  %.2f TFLOP/s  clock %.2f GHz  %.1f (wave 0 counter) / %.1f (wall @2.4 GHz) pipe cycles per iteration"
           "  (expect %d if 16x16x4 = 64 and 4x4x4 = 16)\n",
           NBIG, NSMALL, WG, blocks, waves_per_simd, ms, flops / ms / 1e9, ghz, cyc, wall_cyc, NBIG * 64 + NSMALL * 16);
}

int main() {
    double* d; CK(hipMalloc(&d, 1 << 24));
    unsigned long long* clk; CK(hipMalloc(&clk, 1 << 16));
    int iters = 20000;
    // one 768-thread workgroup per CU = 3 waves per SIMD, as the Hessian tile kernel runs
    run<7, 0, 768>(256, 3 * iters, d, clk);   // clock ramp
    run<7, 0, 768>(256, iters, d, clk);
    run<6, 0, 768>(256, iters, d, clk);
    run<5, 0, 768>(256, iters, d, clk);
    run<0, 7, 768>(256, iters, d, clk);
    run<6, 1, 768>(256, iters, d, clk);
    run<5, 2, 768>(256, iters, d, clk);
    run<5, 1, 768>(256, iters, d, clk);
    run<4, 3, 768>(256, iters, d, clk);
    run<3, 4, 768>(256, iters, d, clk);
    run<7, 0, 512>(256, iters, d, clk);
    run<5, 2, 512>(256, iters, d, clk);
    run<7, 0, 256>(256, iters, d, clk);
    run<5, 2, 256>(256, iters, d, clk);
    int* p; CK(hipMalloc(&p, 4096 * 64 * 4));
    hipLaunchKernelGGL(k_probe, dim3(4096), dim3(64), 0, 0, p);
    int* h = (int*)malloc(4096 * 64 * 4);
    CK(hipMemcpy(h, p, 4096 * 64 * 4, hipMemcpyDeviceToHost));
    // For each A lane la: the set of B lanes it pairs with (same block, same k) and the D lanes hit.
    // Hypothesis H: lane l -> block = l >> 4 ... print the raw structure compactly instead of guessing.
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d pairs with B lanes:", la);
        for (int lb = 0; lb < 64; ++lb) {
            int any = 0, first = -1;
            for (int l = 0; l < 64; ++l) if (h[(la * 64 + lb) * 64 + l]) { any++; if (first < 0) first = l; }
            if (any) printf(" %d->D%d%s", lb, first, any > 1 ? "+" : "");
        }
        printf("\n");
    }
    return 0;
}
