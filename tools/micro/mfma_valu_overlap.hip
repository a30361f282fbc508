// Microbenchmark: can VALU work of one wave overlap f64 MFMAs of the other waves of its SIMD?
// 1024-thread workgroups, one per CU: waves 0..11 run independent f64 MFMAs, waves 12..15 run a
// DEPENDENT chain of VALU ops (f64 fma / f32 fma / int).  Times: MFMA only, VALU only, both.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error line %d\n", __LINE__); exit(1);} } while (0)

// MODE bit0: MFMA waves active; bit1: VALU waves active.  VT: 0 f64, 1 f32, 2 int
template <int MODE, int VT, int INDEP>
__global__ __launch_bounds__(1024) void k(double* out, int iters_m, int iters_v, double a0) {
    const int wid = threadIdx.x >> 6;
    double res = 0;
    if (wid < 12) {
        if (MODE & 1) {
            f64x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (f64x4){0, 0, 0, 0};
            double a = a0 + threadIdx.x, b = a0 * 0.5;
            for (int it = 0; it < iters_m; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            }
            res = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        }
    } else if (MODE & 2) {
        if (VT == 0) {
            double x[INDEP];
#pragma unroll
            for (int j = 0; j < INDEP; ++j) x[j] = a0 + threadIdx.x + j;
            double c = a0 * 1e-3;
            for (int it = 0; it < iters_v; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int j = 0; j < INDEP; ++j) x[j] = fma(x[j], c, 1.0);
            }
            for (int j = 0; j < INDEP; ++j) res += x[j];
        } else if (VT == 1) {
            float x[INDEP];
#pragma unroll
            for (int j = 0; j < INDEP; ++j) x[j] = (float)a0 + threadIdx.x + j;
            float c = (float)a0 * 1e-3f;
            for (int it = 0; it < iters_v; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int j = 0; j < INDEP; ++j) x[j] = fmaf(x[j], c, 1.0f);
            }
            for (int j = 0; j < INDEP; ++j) res += x[j];
        } else {
            unsigned x[INDEP];
#pragma unroll
            for (int j = 0; j < INDEP; ++j) x[j] = threadIdx.x + j;
            unsigned c = (unsigned)a0 + 3;
            for (int it = 0; it < iters_v; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int j = 0; j < INDEP; ++j) x[j] = x[j] * c + 12345u;
            }
            for (int j = 0; j < INDEP; ++j) res += x[j];
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = res;
}

template <int MODE, int VT, int INDEP>
float run(double* d, int im, int iv) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, VT, INDEP>), dim3(256), dim3(1024), 0, 0, d, 10, 10, 1.0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, VT, INDEP>), dim3(256), dim3(1024), 0, 0, d, im, iv, 1.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

template <int VT, int INDEP>
void trio(double* d, const char* name, int im, int iv) {
    float m = run<1, VT, INDEP>(d, im, iv), v = run<2, VT, INDEP>(d, im, iv), b = run<3, VT, INDEP>(d, im, iv);
    printf("%-8s indep=%d: MFMA only %.3f ms (%.1f TF)  VALU only %.3f ms  both %.3f ms   (sum %.3f, max %.3f)\n", name, INDEP, m,
           256.0 * 12 * im * 4 * 2048 / m / 1e9, v, b, m + v, m > v ? m : v);
}

int main() {
    double* d; CK(hipMalloc(&d, 256 * 1024 * 8));
    const int im = 20000;   // 12 waves x 80000 MFMAs per CU
    trio<0, 1>(d, "f64 fma", im, 6000);
    trio<0, 4>(d, "f64 fma", im, 3000);
    trio<1, 1>(d, "f32 fma", im, 6000);
    trio<1, 4>(d, "f32 fma", im, 6000);
    trio<2, 1>(d, "int mad", im, 6000);
    trio<2, 4>(d, "int mad", im, 6000);
    return 0;
}
