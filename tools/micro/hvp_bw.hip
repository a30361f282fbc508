// Dev microbenchmark: streaming-HVP kernel variants (out = X^T (d * (X v))), X read once.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/hvp_bw.hip -o tools/micro/hvp_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef double f64x2u __attribute__((ext_vector_type(2), aligned(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ double sum64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void k_fill(double* p, int64_t n, double scale) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t s = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) p[i] = scale * (double)((i * 2654435761u) & 1023) / 1024.0 - 0.5 * scale;
}

template <int NT>
__device__ __forceinline__ f64x2u ld2(const double* p) {
    if constexpr (NT) return __builtin_nontemporal_load((const f64x2u*)p);
    else return *(const f64x2u*)p;
}

typedef const __attribute__((address_space(1))) char* gcptr;
typedef const __attribute__((address_space(1))) f64x2u* gv2ptr;
template <int NT>
__device__ __forceinline__ f64x2u ldo(const double* base, unsigned boff) {
    // pin the (uniform) row base into SGPRs so that the load uses the saddr + 32-bit voffset form
    uint64_t b = (uint64_t)base;
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    gcptr sb = (gcptr)(((uint64_t)hi << 32) | lo);
    if constexpr (NT) return __builtin_nontemporal_load((gv2ptr)(sb + boff));
    else return *(gv2ptr)(sb + boff);
}

// one wave per row; WG = 256 (4 rows per WG step).  MODE 0 = full, 1 = dot only.
template <int CPT, int PF, int MODE, int NT>
__global__ __launch_bounds__(256) void hvp64(const double* X, int64_t ldx, int64_t N, int D, const double* v,
                                             const double* dvec, double* part) {
    constexpr int NE = 2 * CPT;
    extern __shared__ double dyn[];
    const int tid = threadIdx.x, tt = tid & 63, team = tid >> 6;
    const int64_t T = (int64_t)gridDim.x * 4, g = (int64_t)blockIdx.x * 4 + team;
    unsigned coff[CPT];
    double vv[NE], acc[NE];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        int c0 = 2 * (tt + 64 * k);
        coff[k] = 8u * (unsigned)min(c0, D - 2);
#pragma unroll
        for (int e = 0; e < 2; ++e) { vv[2 * k + e] = c0 + e < D ? v[c0 + e] : 0.0; acc[2 * k + e] = 0.0; }
    }
    const int64_t iters = (N + T - 1) / T;
    f64x2u xn[CPT];
    double dn;
    {
        int64_t nc = min(g, N - 1);
#pragma unroll
        for (int k = 0; k < CPT; ++k) xn[k] = ldo<NT>(X + nc * ldx, coff[k]);
        dn = dvec[nc];
    }
    double dsum = 0.0;
    for (int64_t it = 0; it < iters; ++it) {
        int64_t n = g + it * T;
        f64x2u x[CPT];
        double dcur = n < N ? dn : 0.0;
#pragma unroll
        for (int k = 0; k < CPT; ++k) x[k] = xn[k];
        if (PF) {
            int64_t nc = min(n + T, N - 1);
#pragma unroll
            for (int k = 0; k < CPT; ++k) xn[k] = ldo<NT>(X + nc * ldx, coff[k]);
            dn = dvec[nc];
        }
        double dot = 0.0;
#pragma unroll
        for (int k = 0; k < CPT; ++k) dot = fma(x[k][0], vv[2 * k], fma(x[k][1], vv[2 * k + 1], dot));
        if (MODE == 1) { dsum += dot * dcur; }
        else {
            dot = sum64(dot);
            double t = dcur * dot;
#pragma unroll
            for (int k = 0; k < CPT; ++k) { acc[2 * k] = fma(t, x[k][0], acc[2 * k]); acc[2 * k + 1] = fma(t, x[k][1], acc[2 * k + 1]); }
        }
        if (!PF) {
            int64_t nc = min(n + T, N - 1);
#pragma unroll
            for (int k = 0; k < CPT; ++k) xn[k] = ldo<NT>(X + nc * ldx, coff[k]);
            dn = dvec[nc];
        }
    }
    if (MODE == 1) { acc[0] = dsum; }
    const int DP = 128 * CPT;
#pragma unroll
    for (int k = 0; k < CPT; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) dyn[team * DP + 2 * (tt + 64 * k) + e] = acc[2 * k + e];
    __syncthreads();
    double* out = part + (int64_t)blockIdx.x * (D + 1);
    for (int c = tid; c < D; c += 256) out[c] = dyn[c] + dyn[DP + c] + dyn[2 * DP + c] + dyn[3 * DP + c];
}

// 1024-thread team, one row at a time per CU; v re-read from L2 each row; x reloaded progressively.
template <int CPT, int NT, int VREG>
__global__ __launch_bounds__(1024) void hvp1024(const double* X, int64_t ldx, int64_t N, int D, const double* v,
                                                const double* dvec, double* part) {
    extern __shared__ double vlds[];
    double (*red)[16] = (double (*)[16])(vlds + (VREG == 2 ? ((D + 1) & ~1) : 0));
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t T = gridDim.x, g = blockIdx.x;
    if (VREG == 2) {
        for (int c = tid; c < D; c += 1024) vlds[c] = v[c];
        __syncthreads();
    }
    unsigned coff[CPT];
    double acc[2 * CPT];
    double vv[VREG == 1 ? 2 * CPT : 1];
    unsigned ok = 0;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        int c0 = 2 * (tid + 1024 * k);
        coff[k] = 8u * (unsigned)min(c0, D - 2);
        ok |= (c0 < D ? 1u : 0u) << k;
        acc[2 * k] = acc[2 * k + 1] = 0.0;
        if (VREG == 1) { vv[2 * k] = c0 < D ? v[c0] : 0.0; vv[2 * k + 1] = c0 + 1 < D ? v[c0 + 1] : 0.0; }
    }
    f64x2u x[CPT];
    {
        int64_t nc = min(g, N - 1);
#pragma unroll
        for (int k = 0; k < CPT; ++k) x[k] = ldo<NT>(X + nc * ldx, coff[k]);
    }
    const int64_t iters = (N + T - 1) / T;
    for (int64_t it = 0; it < iters; ++it) {
        int64_t n = g + it * T;
        double dcur = n < N ? dvec[min(n, N - 1)] : 0.0;
        double dot = 0.0;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            if (VREG == 1) dot = fma(x[k][0], vv[2 * k], fma(x[k][1], vv[2 * k + 1], dot));
            else {
                f64x2u vk = VREG == 2 ? *(const f64x2u*)((const char*)vlds + coff[k]) : ldo<0>(v, coff[k]);
                double p = fma(x[k][0], vk[0], x[k][1] * vk[1]);
                dot += ((ok >> k) & 1) ? p : 0.0;
            }
        }
        dot = sum64(dot);
        int p = it & 1;
        if (lane == 0) red[p][wv] = dot;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += red[p][i];
        double t = dcur * tot;
        int64_t nc = min(n + T, N - 1);
        const double* xr = X + nc * ldx;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            double tm = (VREG == 1 || ((ok >> k) & 1)) ? t : 0.0;
            acc[2 * k] = fma(tm, x[k][0], acc[2 * k]);
            acc[2 * k + 1] = fma(tm, x[k][1], acc[2 * k + 1]);
            x[k] = ldo<NT>(xr, coff[k]);
        }
    }
    double* out = part + (int64_t)blockIdx.x * (D + 1);
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        int c0 = 2 * (tid + 1024 * k);
        if (c0 < D) { out[c0] = acc[2 * k]; out[c0 + 1] = acc[2 * k + 1]; }
    }
}

template <typename F>
float timeit(F f, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

int main(int argc, char** argv) {
    int which = argc > 1 ? atoi(argv[1]) : 0;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cu = prop.multiProcessorCount;
    printf("CUs %d\n", cu);
    double *X, *v, *d, *part;
    if (which == 0) {
        int64_t N = 1000000; int D = 1000;
        CK(hipMalloc(&X, N * D * 8)); CK(hipMalloc(&v, D * 8)); CK(hipMalloc(&d, N * 8));
        CK(hipMalloc(&part, (size_t)cu * 16 * (D + 1) * 8));
        k_fill<<<4096, 256>>>(X, N * D, 1.0); k_fill<<<16, 256>>>(v, D, 1.0); k_fill<<<1024, 256>>>(d, N, 1.0);
        double bytes = 8.0 * N * D;
        std::vector<double> h(4);
#define RUN64(CPT, PF, MODE, NT, GM)                                                                           \
        { float ms = timeit([&] { hvp64<CPT, PF, MODE, NT><<<cu * GM, 256, 4 * 128 * CPT * 8>>>(X, D, N, D, v, d, part); }); \
          CK(hipMemcpy(h.data(), part, 32, hipMemcpyDeviceToHost));                                            \
          printf("hvp64 CPT=%d PF=%d MODE=%d NT=%d grid=%dxCU: %.3f ms %.0f GB/s  [%g %g]\n", CPT, PF, MODE, NT, GM, ms, bytes / ms / 1e6, h[0], h[1]); }
        RUN64(8, 0, 0, 0, 4) RUN64(8, 0, 0, 0, 8) RUN64(8, 0, 0, 0, 2)
        RUN64(8, 1, 0, 0, 4) RUN64(8, 1, 0, 0, 2) RUN64(8, 1, 0, 0, 3)
        RUN64(8, 0, 1, 0, 4) RUN64(8, 1, 1, 0, 4) RUN64(8, 0, 1, 0, 8)
        RUN64(8, 0, 0, 1, 4) RUN64(8, 1, 0, 1, 4) RUN64(8, 1, 1, 1, 4)
    } else {
        int64_t N = 100000; int D = 20000;
        CK(hipMalloc(&X, N * D * 8)); CK(hipMalloc(&v, D * 8)); CK(hipMalloc(&d, N * 8));
        CK(hipMalloc(&part, (size_t)cu * 4 * (D + 1) * 8));
        k_fill<<<4096, 256>>>(X, N * D, 1.0); k_fill<<<16, 256>>>(v, D, 1.0); k_fill<<<1024, 256>>>(d, N, 1.0);
        double bytes = 8.0 * N * D;
        std::vector<double> h(4);
#define RUN1024(CPT, NT, VREG)                                                                                 \
        { float ms = timeit([&] { hvp1024<CPT, NT, VREG><<<cu, 1024, (VREG == 2 ? (D + 2) * 8 : 0) + 256>>>(X, D, N, D, v, d, part); });            \
          CK(hipMemcpy(h.data(), part, 32, hipMemcpyDeviceToHost));                                            \
          printf("hvp1024 CPT=%d NT=%d VREG=%d: %.3f ms %.0f GB/s  [%g %g]\n", CPT, NT, VREG, ms, bytes / ms / 1e6, h[0], h[1]); }
        CK(hipFuncSetAttribute((const void*)hvp1024<10, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
        CK(hipFuncSetAttribute((const void*)hvp1024<10, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
        RUN1024(10, 0, 0) RUN1024(10, 1, 0) RUN1024(10, 0, 1) RUN1024(10, 0, 2) RUN1024(10, 1, 2)
    }
    return 0;
}
