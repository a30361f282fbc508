// Microbenchmark: v_mfma_f64_16x16x4_f64 issue rate, dependent-chain latency, the clock
// the chip holds under an fp64-MFMA load, and the f64 C/D layout on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error line %d\n", __LINE__); exit(1);} } while (0)

template <int NACC, int WG, bool VALU>
__global__ __launch_bounds__(WG) void k_rate(double* out, unsigned long long* clk, int iters, double a0, double b0) {
    f64x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f64x4){0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (VALU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][r] = fma(a, b, acc[i][r]);
            } else {
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ void k_layout(double* out) {
    int l = threadIdx.x;
    double a = 100.0 * (l & 15) + (l >> 4);
    double b = ((l >> 4) == 0) ? (double)((l & 15) + 1) : 0.0;
    f64x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

template <int NACC, int WG, bool VALU>
void run(int blocks, int iters, double* d, unsigned long long* clk) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate<NACC, WG, VALU>), dim3(blocks), dim3(WG), 0, 0, d, clk, 10, 1.0, 2.0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rate<NACC, WG, VALU>), dim3(blocks), dim3(WG), 0, 0, d, clk, iters, 1.0, 2.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double ops = (double)blocks * (WG / 64) * iters * NACC;           // wave-level MFMAs (or 4 FMA instr)
    double flops = ops * 2048.0;
    double ghz = (double)h[0] / (double)h[1] * 0.1;
    double waves_per_simd = (double)blocks * (WG / 64) / 1024.0;
    double cyc = (double)h[0] / ((double)iters * NACC * (waves_per_simd < 1 ? 1 : waves_per_simd));
    printf("%s NACC=%d wg=%d blocks=%d (%.1f waves/SIMD): %.3f ms  %.2f TFLOP/s  clock %.2f GHz  %.1f cyc per %s per SIMD\n",
           VALU ? "VALU" : "MFMA", NACC, WG, blocks, waves_per_simd, ms, flops / ms / 1e9, ghz, cyc,
           VALU ? "4 v_fma_f64" : "MFMA");
}

int main() {
    double* d; CK(hipMalloc(&d, 1 << 24));
    unsigned long long* clk; CK(hipMalloc(&clk, 1 << 16));
    int iters = 20000;
    run<1, 256, false>(256, iters, d, clk);
    run<2, 256, false>(256, iters, d, clk);
    run<4, 256, false>(256, iters, d, clk);
    run<8, 256, false>(256, iters, d, clk);
    run<4, 256, false>(512, iters, d, clk);
    run<4, 256, false>(768, iters, d, clk);
    run<4, 256, false>(1024, iters, d, clk);
    run<1, 512, false>(256, iters, d, clk);
    run<2, 512, false>(256, iters, d, clk);
    run<4, 512, false>(256, iters, d, clk);
    run<1, 1024, false>(256, iters, d, clk);
    run<2, 1024, false>(256, iters, d, clk);
    run<4, 256, false>(64, iters, d, clk);
    run<4, 256, true>(256, iters, d, clk);
    run<4, 256, true>(512, iters, d, clk);
    run<4, 256, true>(1024, iters, d, clk);
    double* h = (double*)malloc(256 * 8);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d);
    CK(hipMemcpy(h, d, 256 * 8, hipMemcpyDeviceToHost));
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int i = (l >> 4) + 4 * r, j = l & 15;
        if (h[l * 4 + r] != 100.0 * i * (j + 1)) ok = 0;
    }
    printf("f64 C/D layout row=(lane>>4)+4*reg, col=lane&15: %s\n", ok ? "CONFIRMED" : "MISMATCH");
    return 0;
}
