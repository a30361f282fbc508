// Microbenchmark: does VALU work overlap v_mfma_f32_16x16x4_f32 on a gfx950 SIMD?
//  (a) other-wave: waves 0..3 (one per SIMD) run independent f32 MFMAs, waves 4..7 run VALU chains;
//  (b) same-wave: one wave per SIMD issues [1 MFMA, KV independent VALU ops] in program order.
// Times: MFMA only, VALU only, both.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_valu_overlap mfma_f32_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error line %d\n", __LINE__); exit(1);} } while (0)

// MODE bit0: MFMAs; bit1: VALU.  VT: 0 f64 fma, 1 f32 fma.  SAME: VALU in the MFMA wave itself, KV per MFMA.
template <int MODE, int VT, int SAME, int KV>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a0) {
    const int wid = threadIdx.x >> 6;
    float res = 0;
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = a0 * 0.5f;
    double xd[8];
    float xf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { xd[j] = a0 + threadIdx.x + j; xf[j] = a0 + threadIdx.x + j; }
    const double cd = a0 * 1e-3;
    const float cf = a0 * 1e-3f;
    if (SAME) {
        if (wid < 4) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (MODE & 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                    if (MODE & 2) {
#pragma unroll
                        for (int v = 0; v < KV; ++v) {
                            if (VT == 0) xd[(i * KV + v) & 7] = fma(xd[(i * KV + v) & 7], cd, 1.0);
                            else xf[(i * KV + v) & 7] = fmaf(xf[(i * KV + v) & 7], cf, 1.0f);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    } else {
        if (wid < 4) {
            if (MODE & 1)
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                }
        } else if (MODE & 2) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int v = 0; v < KV; ++v) {
                        if (VT == 0) xd[(i * KV + v) & 7] = fma(xd[(i * KV + v) & 7], cd, 1.0);
                        else xf[(i * KV + v) & 7] = fmaf(xf[(i * KV + v) & 7], cf, 1.0f);
                    }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) res += acc[i][0] + (float)xd[i] + xf[i];
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <int MODE, int VT, int SAME, int KV>
float run(float* d, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, VT, SAME, KV>), dim3(256), dim3(512), 0, 0, d, 10, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, VT, SAME, KV>), dim3(256), dim3(512), 0, 0, d, iters, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

template <int VT, int SAME, int KV>
void trio(float* d, int iters) {
    float m = run<1, VT, SAME, KV>(d, iters), v = run<2, VT, SAME, KV>(d, iters), b = run<3, VT, SAME, KV>(d, iters);
    printf("%s %s KV=%d: MFMA only %.3f ms (%.1f TF)  VALU only %.3f ms (%.2f cyc/op at 2.4 GHz)  both %.3f ms  (sum %.3f, max %.3f)\n",
           SAME ? "same-wave " : "other-wave", VT ? "f32 fma" : "f64 fma", KV, m, 256.0 * 4 * iters * 8 * 2048 / m / 1e9, v,
           v * 1e-3 * 2.4e9 / ((double)iters * 8 * KV), b, m + v, m > v ? m : v);
}

int main() {
    float* d; CK(hipMalloc(&d, 256 * 512 * 4));
    const int it = 40000;
    trio<1, 0, 2>(d, it); trio<1, 0, 4>(d, it); trio<1, 0, 6>(d, it);
    trio<0, 0, 2>(d, it); trio<0, 0, 4>(d, it);
    trio<1, 1, 2>(d, it); trio<1, 1, 4>(d, it); trio<1, 1, 6>(d, it);
    trio<0, 1, 2>(d, it); trio<0, 1, 4>(d, it);
    return 0;
}
