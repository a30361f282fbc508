// Dev microbenchmark 2: buffer-load based streaming HVP, double-buffered rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
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

// Row descriptor: raw buffer over one row (uniform base), out-of-range lanes read 0.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const double* row, unsigned bytes) {
    uint64_t b = (uint64_t)row;
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
template <int AUX>
__device__ __forceinline__ f64x2 bld(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, AUX));
}

// TEAM = 64*TW threads per row.  WG = WGT threads (WGT/TEAM teams).  Rows double-buffered in registers.
template <int TW, int WGT, int CPT, int VLDS, int AUX, int P>
__global__ __launch_bounds__(WGT) void hvp(const double* X, int64_t ldx, int64_t N, int D, const double* v,
                                           const double* dvec, double* part) {
    constexpr int TEAM = 64 * TW, TEAMS = WGT / TEAM, NE = 2 * CPT;
    extern __shared__ double dyn[];
    double* vl = dyn;                                           // VLDS: v (zero padded to TEAM*NE)
    const int DZ = (D + 1) & ~1;                                  // index of a zero pair behind v
    double* red = dyn + (VLDS ? DZ + 2 : 0);                 // [2][TW] per team
    const int tid = threadIdx.x, tt = tid % TEAM, team = tid / TEAM, lane = tid & 63, wv = (tid >> 6) % TW;
    const int64_t T = (int64_t)gridDim.x * TEAMS, g = (int64_t)blockIdx.x * TEAMS + team;
    const unsigned voff = 16u * tt, rowbytes = 8u * D;
    double acc[NE], vv[VLDS ? 1 : NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) acc[i] = 0.0;
    if (VLDS) {
        for (int c = tid; c < DZ + 2; c += WGT) vl[c] = c < D ? v[c] : 0.0;
        __syncthreads();
    } else {
#pragma unroll
        for (int k = 0; k < CPT; ++k)
#pragma unroll
            for (int e = 0; e < 2; ++e) { int c = 2 * (tt + TEAM * k) + e; vv[2 * k + e] = c < D ? v[c] : 0.0; }
    }
    const int64_t iters = (N + T - 1) / T;
    f64x2 x[CPT], q[P > 0 ? P : 1];
    double dn;
    {
        bool ok = g < N; int64_t nc = ok ? g : 0;
        __amdgpu_buffer_rsrc_t r = row_rsrc(X + nc * ldx, ok ? rowbytes : 0);
#pragma unroll
        for (int k = 0; k < CPT; ++k) { if (k < P) q[k] = bld<AUX>(r, voff + 16u * TEAM * k); else x[k] = bld<AUX>(r, voff + 16u * TEAM * k); }
        dn = dvec[nc];
    }
    for (int64_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < P; ++k) x[k] = q[k];
        double dcur = dn;
        int64_t n = g + (it + 1) * T;
        bool ok = n < N; int64_t nc = ok ? n : 0;
        __amdgpu_buffer_rsrc_t r = row_rsrc(X + nc * ldx, ok ? rowbytes : 0);
#pragma unroll
        for (int k = 0; k < P; ++k) q[k] = bld<AUX>(r, voff + 16u * TEAM * k);
        dn = dvec[nc];
        double dot = 0.0;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            if (VLDS) {
                f64x2 vk = *(const f64x2*)(vl + min(2 * (tt + TEAM * k), DZ));
                dot = fma(x[k][0], vk[0], fma(x[k][1], vk[1], dot));
            } else dot = fma(x[k][0], vv[2 * k], fma(x[k][1], vv[2 * k + 1], dot));
        }
        dot = sum64(dot);
        if (TW > 1) {
            double* rd = red + ((it & 1) * TEAMS + team) * TW;
            if (lane == 0) rd[wv] = dot;
            __syncthreads();
            dot = 0.0;
#pragma unroll
            for (int i = 0; i < TW; ++i) dot += rd[i];
        }
        double t = dcur * dot;
#pragma unroll
        for (int k = P; k < CPT; ++k) {
            acc[2 * k] = fma(t, x[k][0], acc[2 * k]); acc[2 * k + 1] = fma(t, x[k][1], acc[2 * k + 1]);
            x[k] = bld<AUX>(r, voff + 16u * TEAM * k);
        }
#pragma unroll
        for (int k = 0; k < P; ++k) { acc[2 * k] = fma(t, x[k][0], acc[2 * k]); acc[2 * k + 1] = fma(t, x[k][1], acc[2 * k + 1]); }
    }
    // combine teams of the workgroup (reuse LDS) and write the partial
    double* out = part + (int64_t)blockIdx.x * (D + 1);
    if (TEAMS > 1) {
        __syncthreads();
        const int DP = TEAM * NE;
#pragma unroll
        for (int k = 0; k < CPT; ++k) *(f64x2*)(dyn + team * DP + 2 * (tt + TEAM * k)) = f64x2{acc[2 * k], acc[2 * k + 1]};
        __syncthreads();
        for (int c = tid; c < D; c += WGT) { double s = 0.0; for (int q = 0; q < TEAMS; ++q) s += dyn[q * DP + c]; out[c] = s; }
    } else {
#pragma unroll
        for (int k = 0; k < CPT; ++k) { int c = 2 * (tt + TEAM * k); if (c < D) { out[c] = acc[2 * k]; out[c + 1] = acc[2 * k + 1]; } }
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
    double *X, *v, *d, *part;
    int64_t N = which == 0 ? 1000000 : 100000; int D = which == 0 ? 1000 : 20000;
    CK(hipMalloc(&X, N * D * 8)); CK(hipMalloc(&v, D * 8)); CK(hipMalloc(&d, N * 8));
    CK(hipMalloc(&part, (size_t)cu * 16 * (D + 1) * 8));
    k_fill<<<4096, 256>>>(X, N * D, 1.0); k_fill<<<16, 256>>>(v, D, 1.0); k_fill<<<1024, 256>>>(d, N, 1.0);
    double bytes = 8.0 * N * D;
    std::vector<double> h(4);
#define RUN(TW, WGT, CPT, VLDS, AUX, MINB, GM) RUNP(TW, WGT, CPT, VLDS, AUX, CPT, GM)
#define RUNP(TW, WGT, CPT, VLDS, AUX, MINB, GM)                                                                  \
    { constexpr int TEAM = 64 * TW, TEAMS = WGT / TEAM;                                                         \
      size_t sh = (size_t)8 * ((VLDS ? D + 4 : 0) + 2 * TEAMS * TW);                                   \
      size_t sh2 = TEAMS > 1 ? (size_t)8 * TEAMS * TEAM * 2 * CPT : 0; if (sh2 > sh) sh = sh2;                  \
      CK(hipFuncSetAttribute((const void*)hvp<TW, WGT, CPT, VLDS, AUX, MINB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); \
      float ms = timeit([&] { hvp<TW, WGT, CPT, VLDS, AUX, MINB><<<cu * GM, WGT, sh>>>(X, D, N, D, v, d, part); }); \
      CK(hipMemcpy(h.data(), part, 32, hipMemcpyDeviceToHost));                                                 \
      printf("hvp TW=%d WG=%d CPT=%d VLDS=%d AUX=%d P=%d grid=%dxCU: %.3f ms %.0f GB/s  [%g %g]\n", TW, WGT, CPT, VLDS, AUX, MINB, GM, ms, bytes / ms / 1e6, h[0], h[1]); }
    if (which == 0) {
        RUN(1, 256, 8, 0, 0, 1, 2) RUN(1, 256, 8, 0, 0, 1, 3) RUN(1, 256, 8, 0, 0, 1, 4)
        RUN(1, 256, 8, 0, 2, 1, 3) RUN(1, 256, 8, 0, 3, 1, 3)
        RUN(4, 256, 2, 0, 0, 1, 4) RUN(4, 256, 2, 0, 0, 1, 8) RUN(4, 256, 2, 0, 2, 1, 8)
        RUN(8, 512, 1, 0, 0, 1, 2) RUN(8, 512, 1, 0, 0, 1, 4)
    } else {
        RUNP(8, 512, 20, 1, 0, 8, 1) RUNP(8, 512, 20, 1, 0, 10, 1) RUNP(8, 512, 20, 1, 0, 12, 1) RUNP(8, 512, 20, 1, 0, 14, 1) RUNP(8, 512, 20, 1, 2, 12, 1)
        RUNP(16, 1024, 10, 1, 0, 4, 1) RUNP(16, 1024, 10, 1, 0, 6, 1) RUNP(16, 1024, 10, 1, 0, 0, 1)
    }
    return 0;
}
