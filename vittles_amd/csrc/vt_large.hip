// Large-D (D > 128) dense path: tiled fp64-MFMA building blocks and the host-side blocked
// algorithms on top of them (BASELINE.json configs[2]: D = 1000).
//
//   k_syrk_tile   H tile (128x128) += X[:, I]^T diag(d) X[:, J] over a chunk of rows (split-K over N),
//                 operands loaded straight from global memory in MFMA layout (re-reads hit L2/MALL).
//   k_gemm_nt     C = beta C + alpha A B^T   (A: M x K, B: N x K, both row-major, K contiguous);
//                 the k index inside a group of 16 is permuted so that a lane owns 4 consecutive k
//                 (32-byte loads) -- A and B use the same permutation, so the product is unchanged.
//   blocked Cholesky (128-wide panels): diagonal block by k_potrf_pack_small, panel by the MFMA
//                 TRSM in forward-only mode, trailing update by k_gemm_nt.
//   blocked TRSM  S = alpha s .* B (L L^T)^{-1}: per 128-column block one k_gemm_nt update and one
//                 small-TRSM solve with the block's inverted-diagonal pack.
// Replaces the same reference lines as the small-D kernels (see include/vittles_hip.h).
#include "vt_common.h"

namespace {

constexpr int kT = 128;            // tile / panel width
constexpr int kPackStride = 2 * 36 * 256;   // doubles per diagonal-block pack (NB = 8)

int g_cu = 0;
int num_cu() {
    if (g_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        g_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return g_cu;
}

// ------------------------------------------------------------------ GEMM NT -------
struct GemmArgs {
    double* C;
    int64_t ldc;
    int64_t M;
    int N;
    const double* A;
    int64_t lda;
    const double* B;
    int64_t ldb;
    int K;
    double alpha, beta;
};

// 4 consecutive k of one row: vector loads when the group is inside [0, K), zero otherwise.
template <bool VEC>
__device__ __forceinline__ f64x4 load_k4(const double* row, int k, int K) {
    if constexpr (VEC) {
        int kc = min(k, K - 4);
        f64x2u lo = *(const f64x2u*)(row + kc);
        f64x2u hi = *(const f64x2u*)(row + kc + 2);
        f64x4 v = {lo[0], lo[1], hi[0], hi[1]};
        return k < K ? v : (f64x4){0.0, 0.0, 0.0, 0.0};
    } else {
        f64x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double x = row[min(k + e, K - 1)];
            v[e] = (k + e < K) ? x : 0.0;
        }
        return v;
    }
}

template <bool VEC, bool LOWER>
__global__ __launch_bounds__(256) void k_gemm_nt(GemmArgs a) {
    if (LOWER && blockIdx.y > blockIdx.x) return;   // tile strictly above the diagonal
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int64_t m_base = (int64_t)blockIdx.x * kT + (wave >> 1) * 64;
    const int n_base = blockIdx.y * kT + (wave & 1) * 64;
    if (m_base >= a.M || n_base >= a.N) return;

    const double* arow[4];
    const double* brow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t m = m_base + 16 * i + c;
        arow[i] = a.A + (m < a.M ? m : a.M - 1) * a.lda;
        int n = n_base + 16 * i + c;
        brow[i] = a.B + (int64_t)(n < a.N ? n : a.N - 1) * a.ldb;
    }
    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};

    for (int k0 = 0; k0 < a.K; k0 += 16) {
        f64x4 af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i] = load_k4<VEC>(arow[i], k0 + 4 * q, a.K);
            bf[i] = load_k4<VEC>(brow[i], k0 + 4 * q, a.K);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma_f64(af[i][s], bf[j][s], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = n_base + 16 * j + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int64_t m = m_base + 16 * i + q + 4 * r;
                if (m < a.M && n < a.N) {
                    double* p = a.C + m * a.ldc + n;
                    double v = a.alpha * acc[i][j][r];
                    if (a.beta != 0.0) v = fma(a.beta, *p, v);
                    *p = v;
                }
            }
        }
}

int gemm_nt(double* C, int64_t ldc, int64_t M, int N, const double* A, int64_t lda, const double* B,
            int64_t ldb, int K, double alpha, double beta, bool lower, hipStream_t st) {
    if (M <= 0 || N <= 0) return VT_OK;
    GemmArgs a{C, ldc, M, N, A, lda, B, ldb, K, alpha, beta};
    dim3 grid((unsigned)((M + kT - 1) / kT), (unsigned)((N + kT - 1) / kT));
    bool vec = (K % 4 == 0) && K >= 4;
    if (vec && lower) hipLaunchKernelGGL((k_gemm_nt<true, true>), grid, dim3(256), 0, st, a);
    else if (vec) hipLaunchKernelGGL((k_gemm_nt<true, false>), grid, dim3(256), 0, st, a);
    else if (lower) hipLaunchKernelGGL((k_gemm_nt<false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_gemm_nt<false, false>), grid, dim3(256), 0, st, a);
    return vt_launch_status();
}

// ------------------------------------------------------------------ SYRK tiles -------
struct SyrkArgs {
    const double* X;
    int64_t ldx;
    const double* d;
    int64_t N;
    int D;
    int nt;        // tiles per side
    int ksplit;
    double* part;  // [ksplit][ntl][128*128]
};

__global__ __launch_bounds__(256) void k_syrk_tile(SyrkArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15;
    // lower-triangular tile index -> (ti, tj), tj <= ti
    int t = blockIdx.x, ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - ti * (ti + 1) / 2;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = ti * kT + wm * 64, j0 = tj * kT + wn * 64;
    const int ntl = a.nt * (a.nt + 1) / 2;
    double* out = a.part + ((int64_t)blockIdx.y * ntl + t) * (kT * kT);
    const bool skip = (ti == tj && wn > wm) || i0 >= a.D || j0 >= a.D;   // wave strictly above the diagonal

    const int64_t nsteps = (a.N + 3) >> 2;
    const int64_t per = (nsteps + a.ksplit - 1) / a.ksplit;
    const int64_t s0 = (int64_t)blockIdx.y * per;
    const int64_t s1 = min(nsteps, s0 + per);

    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};

    if (!skip) {
        int ca[4], cb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ca[i] = min(i0 + 16 * i + c, a.D - 1);
            cb[i] = min(j0 + 16 * i + c, a.D - 1);
        }
        struct Buf { double xa[4], xb[4], dn; };
        auto load = [&](int64_t s, Buf& b) {
            int64_t n = s * 4 + q;
            bool ok = s < s1 && n < a.N;
            int64_t nc = n < a.N ? n : a.N - 1;
            const double* xr = a.X + nc * a.ldx;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                b.xa[i] = xr[ca[i]];
                b.xb[i] = xr[cb[i]];
            }
            double dv = a.d[nc];
            b.dn = ok ? dv : 0.0;
        };
        auto compute = [&](const Buf& b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double bj = b.dn * b.xb[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = mfma_f64(b.xa[i], bj, acc[i][j]);
            }
        };
        Buf b0, b1, b2;
        int64_t s = s0;
        load(s, b0);
        load(s + 1, b1);
        load(s + 2, b2);
        for (; s < s1; s += 3) {
            compute(b0);
            load(s + 3, b0);
            __builtin_amdgcn_sched_barrier(0);
            compute(b1);
            load(s + 4, b1);
            __builtin_amdgcn_sched_barrier(0);
            compute(b2);
            load(s + 5, b2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // columns beyond D inside the tile hold copies of column D-1: never read by the reduce.
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(wm * 64 + 16 * i + q + 4 * r) * kT + wn * 64 + 16 * j + c] = acc[i][j][r];
}

// H[i][j] = H[j][i] = sum_chunks part (+ prior on the diagonal), i >= j.
__global__ void k_syrk_reduce(const double* part, int nt, int ksplit, int D, const double* prior, double* H,
                              int ldh) {
    int t = blockIdx.x, ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - ti * (ti + 1) / 2;
    const int ntl = nt * (nt + 1) / 2;
    for (int e = threadIdx.x; e < kT * kT; e += blockDim.x) {
        int i = ti * kT + e / kT, j = tj * kT + e % kT;
        if (i >= D || j >= D || j > i) continue;
        double s = 0.0;
        for (int k = 0; k < ksplit; ++k) s += part[((int64_t)k * ntl + t) * (kT * kT) + e];
        if (i == j && prior) s += prior[i];
        H[(int64_t)i * ldh + j] = s;
        H[(int64_t)j * ldh + i] = s;
    }
}

// ------------------------------------------------------------------ small helpers -------
// Out[n][c] = alpha * s_n * B[n][c]
__global__ void k_scale_rows(const double* B, int64_t ldb, const double* s, double alpha, int64_t N, int D,
                             double* Out, int64_t ldo) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = N * D;
    for (; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t n = idx / D;
        int cidx = (int)(idx - n * D);
        double f = s ? alpha * s[n] : alpha;
        Out[n * ldo + cidx] = f * B[n * ldb + cidx];
    }
}

// Lc = tril(L) (ld D), LT = Lc^T
__global__ void k_copy_lower_and_transpose(const double* L, int ldl, int D, double* Lc, double* LT) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)D * D) return;
    int i = (int)(idx / D), j = (int)(idx - (int64_t)i * D);
    double v = j <= i ? L[(int64_t)i * ldl + j] : 0.0;
    Lc[(int64_t)i * D + j] = v;
    LT[(int64_t)j * D + i] = v;
}

__global__ void k_merge_info(int* info, const int* local, int offset) {
    if (*info == 0 && *local != 0) *info = offset + *local;
}

}  // namespace

// ------------------------------------------------------------------ host algorithms -------
size_t vt_large_potrf_ws_bytes(int D) { return (size_t)kPackStride * sizeof(double) + 256; }

int vt_large_potrf(double* A, int lda, int D, int* info, void* ws, hipStream_t st) {
    double* pack = (double*)ws;
    int* linfo = (int*)((char*)ws + (size_t)kPackStride * sizeof(double));
    if (hipMemsetAsync(info, 0, sizeof(int), st) != hipSuccess) return VT_ERR_HIP;
    const int nblk = (D + kT - 1) / kT;
    for (int J = 0; J < nblk; ++J) {
        const int o = J * kT, Db = std::min(kT, D - o);
        double* Ajj = A + (int64_t)o * lda + o;
        int rc = vt_potrf_pack_block(Ajj, lda, Db, linfo, pack, 1, st);
        if (rc != VT_OK) return rc;
        hipLaunchKernelGGL(k_merge_info, dim3(1), dim3(1), 0, st, info, (const int*)linfo, o);
        const int R = D - o - Db;
        if (R > 0) {
            double* P = A + (int64_t)(o + Db) * lda + o;   // panel below the diagonal block
            rc = vt_trsm_rows_mode(pack, Db, P, lda, nullptr, 1.0, R, P, lda, 1, st);   // P <- P L_JJ^{-T}
            if (rc != VT_OK) return rc;
            double* C = A + (int64_t)(o + Db) * lda + (o + Db);
            rc = gemm_nt(C, lda, R, R, P, lda, P, lda, Db, -1.0, 1.0, true, st);
            if (rc != VT_OK) return rc;
        }
    }
    return vt_launch_status();
}

size_t vt_large_pack_bytes(int D) {
    const int nblk = (D + kT - 1) / kT;
    return ((size_t)2 * D * D + (size_t)nblk * kPackStride) * sizeof(double);
}

int vt_large_pack(const double* L, int ldl, int D, void* pack, hipStream_t st) {
    double* Lc = (double*)pack;
    double* LT = Lc + (size_t)D * D;
    double* packs = LT + (size_t)D * D;
    int64_t tot = (int64_t)D * D;
    hipLaunchKernelGGL(k_copy_lower_and_transpose, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, L, ldl, D,
                       Lc, LT);
    const int nblk = (D + kT - 1) / kT;
    for (int J = 0; J < nblk; ++J) {
        const int o = J * kT, Db = std::min(kT, D - o);
        int rc = vt_potrf_pack_block(Lc + (int64_t)o * D + o, D, Db, nullptr, packs + (size_t)J * kPackStride, 0, st);
        if (rc != VT_OK) return rc;
    }
    return vt_launch_status();
}

int vt_large_trsm_rows(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale,
                       double alpha, int64_t N, double* Out, int64_t ldo, hipStream_t st) {
    const double* Lc = (const double*)pack;
    const double* LT = Lc + (size_t)D * D;
    const double* packs = LT + (size_t)D * D;
    const int nblk = (D + kT - 1) / kT;
    {
        int64_t tot = N * D;
        int64_t g = (tot + 255) / 256;
        if (g > 65535) g = 65535;
        hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)g), dim3(256), 0, st, B, ldb, rowscale, alpha, N, D, Out, ldo);
    }
    int rc;
    for (int J = 0; J < nblk; ++J) {   // forward:  Y L^T = G
        const int o = J * kT, Db = std::min(kT, D - o);
        if (J > 0) {
            rc = gemm_nt(Out + o, ldo, N, Db, Out, ldo, Lc + (int64_t)o * D, D, o, -1.0, 1.0, false, st);
            if (rc != VT_OK) return rc;
        }
        rc = vt_trsm_rows_mode(packs + (size_t)J * kPackStride, Db, Out + o, ldo, nullptr, 1.0, N, Out + o, ldo, 1, st);
        if (rc != VT_OK) return rc;
    }
    for (int J = nblk - 1; J >= 0; --J) {   // backward:  Z L = Y
        const int o = J * kT, Db = std::min(kT, D - o);
        const int rest = D - o - Db;
        if (rest > 0) {
            rc = gemm_nt(Out + o, ldo, N, Db, Out + o + Db, ldo, LT + (int64_t)o * D + o + Db, D, rest, -1.0, 1.0,
                         false, st);
            if (rc != VT_OK) return rc;
        }
        rc = vt_trsm_rows_mode(packs + (size_t)J * kPackStride, Db, Out + o, ldo, nullptr, 1.0, N, Out + o, ldo, 2, st);
        if (rc != VT_OK) return rc;
    }
    return vt_launch_status();
}

static int syrk_split(int ntl) {
    int want = 2 * num_cu();
    int ks = (want + ntl - 1) / ntl;
    if (ks < 1) ks = 1;
    if (ks > 64) ks = 64;
    return ks;
}

size_t vt_large_syrk_ws_bytes(int D) {
    const int nt = (D + kT - 1) / kT, ntl = nt * (nt + 1) / 2;
    return (size_t)syrk_split(ntl) * ntl * kT * kT * sizeof(double);
}

int vt_large_syrk(const double* X, int64_t ldx, const double* d, int64_t N, int D, const double* prior, double* H,
                  int ldh, void* ws, hipStream_t st) {
    const int nt = (D + kT - 1) / kT, ntl = nt * (nt + 1) / 2;
    int ks = syrk_split(ntl);
    int64_t nsteps = (N + 3) / 4;
    if (ks > nsteps) ks = (int)std::max<int64_t>(1, nsteps);
    SyrkArgs a{X, ldx, d, N, D, nt, ks, (double*)ws};
    hipLaunchKernelGGL(k_syrk_tile, dim3(ntl, ks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_syrk_reduce, dim3(ntl), dim3(1024), 0, st, (const double*)ws, nt, ks, D, prior, H, ldh);
    return vt_launch_status();
}
