// Counter-based synthetic inputs (SURVEY.md section 8d "Synthetic inputs"): Philox-4x32-10 with key = seed and
// counter = (row, column pair, stream), so that ANY block of rows can be regenerated independently -- on another
// rank (row shards of one global data set), or on the host for a spot check (vittles_amd/synth.py implements the same
// generator in NumPy; the bits are identical, the normal transform agrees to a few ulps of libm).
// The reference has no generator of its own: its tests draw from np.random with fixed seeds
// (vittles/tests/test_sensitivity_lib.py:463-478, notebook cell 6); this replaces that set-up step for data sets that
// never touch the host (config 3: 80 GB of X).
#include "vt_common.h"

namespace {

struct Philox4 {
    unsigned x[4];
};

// Philox-4x32-10 (Salmon et al., SC'11; the constants of the Random123 reference implementation)
__host__ __device__ __forceinline__ Philox4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                                          unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}

// 53 random bits -> a double in (0, 1): (m + 0.5) 2^-53
__host__ __device__ __forceinline__ double u01(unsigned hi, unsigned lo) {
    const unsigned long long m = (((unsigned long long)hi << 32) | lo) >> 11;
    return ((double)m + 0.5) * 0x1.0p-53;
}

// X[n][2p], X[n][2p+1] = Box-Muller of the two uniforms of counter (row, p, stream)
template <typename T>
__global__ __launch_bounds__(256) void k_synth_normal(unsigned k0, unsigned k1, unsigned stream, int64_t row0, int64_t N,
                                                      int D, T* X, int64_t ldx) {
    const int P = (D + 1) >> 1;
    const int64_t total = N * P;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t n = e / P;
        const int p = (int)(e - n * P);
        const unsigned long long row = (unsigned long long)(row0 + n);
        const Philox4 r = philox4x32_10((unsigned)row, (unsigned)(row >> 32), (unsigned)p, stream, k0, k1);
        const double u1 = u01(r.x[0], r.x[1]), u2 = u01(r.x[2], r.x[3]);
        const double rad = sqrt(-2.0 * log(u1));
        double sn, cs;
        sincospi(2.0 * u2, &sn, &cs);
        T* xr = X + n * ldx + 2 * p;
        xr[0] = (T)(rad * cs);
        if (2 * p + 1 < D) xr[1] = (T)(rad * sn);
    }
}

// y_n from z_n = x_n . theta* and one uniform of counter (row, 0, stream 1):
//   logistic: 1{u < sigmoid(z)};  poisson: inverse CDF of Poisson(exp(z)) by sequential search;  gaussian: z + N(0, 1)
__global__ __launch_bounds__(256) void k_synth_response(unsigned k0, unsigned k1, int64_t row0, int64_t N, int family,
                                                        const double* z, double* y) {
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
        const unsigned long long row = (unsigned long long)(row0 + n);
        const Philox4 r = philox4x32_10((unsigned)row, (unsigned)(row >> 32), 0u, 1u, k0, k1);
        const double u = u01(r.x[0], r.x[1]);
        const double zn = z[n];
        double out;
        if (family == VT_FAMILY_LOGISTIC) {
            out = u < 1.0 / (1.0 + exp(-zn)) ? 1.0 : 0.0;
        } else if (family == VT_FAMILY_POISSON) {
            const double lam = exp(zn);
            double pk = exp(-lam), cdf = pk;
            int k = 0;
            while (cdf < u && k < 100000) {
                ++k;
                pk *= lam / k;
                cdf += pk;
            }
            out = (double)k;
        } else {
            const double u2 = u01(r.x[2], r.x[3]);
            out = zn + sqrt(-2.0 * log(u)) * cospi(2.0 * u2);
        }
        y[n] = out;
    }
}

template <typename T>
int synth_normal(uint64_t seed, unsigned stream_id, int64_t row0, int64_t N, int D, T* X, int64_t ldx, void* stream) {
    if (N < 0 || D < 1 || ldx < D || row0 < 0) return VT_ERR_BAD_ARG;
    if (N == 0) return VT_OK;
    if (!X) return VT_ERR_BAD_ARG;
    const int64_t total = N * ((D + 1) / 2);
    const int grid = (int)std::min<int64_t>((total + 255) / 256, (int64_t)vt_num_cu() * 16);
    hipLaunchKernelGGL((k_synth_normal<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned)seed,
                       (unsigned)(seed >> 32), stream_id, row0, N, D, X, ldx);
    return vt_launch_status();
}

}  // namespace

extern "C" int vt_synth_normal(uint64_t seed, unsigned stream_id, int64_t row0, int64_t N, int D, double* X, int64_t ldx,
                               void* stream) {
    return synth_normal<double>(seed, stream_id, row0, N, D, X, ldx, stream);
}

extern "C" int vt_synth_normal_f32(uint64_t seed, unsigned stream_id, int64_t row0, int64_t N, int D, float* X,
                                   int64_t ldx, void* stream) {
    return synth_normal<float>(seed, stream_id, row0, N, D, X, ldx, stream);
}

extern "C" int vt_synth_response(uint64_t seed, int64_t row0, int64_t N, int family, const double* z, double* y,
                                 void* stream) {
    if (N < 0 || row0 < 0 || family < 0 || family > 2) return VT_ERR_BAD_ARG;
    if (N == 0) return VT_OK;
    if (!z || !y) return VT_ERR_BAD_ARG;
    const int grid = (int)std::min<int64_t>((N + 255) / 256, (int64_t)vt_num_cu() * 16);
    hipLaunchKernelGGL(k_synth_response, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned)seed,
                       (unsigned)(seed >> 32), row0, N, family, z, y);
    return vt_launch_status();
}
