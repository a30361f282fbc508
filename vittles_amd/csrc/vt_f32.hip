// fp32 storage mode of the sensitivity path (BASELINE.json north_star: "optional fp32 mode", tolerance 1e-4
// against the fp64 oracle).  X -- and the sensitivity matrix S, the two N x D arrays that set the HBM traffic --
// are float; theta, y, w, the per-row factors r and d, H, g and the Cholesky factor stay double.
//   vt_glm_hessian_f32 : H = X^T diag(d) X by v_mfma_f32_16x16x4_f32, row dot in f32, link in fp64
//   vt_trsm_pack_f32   : the fp64 TRSM operand pack (vt_potrf_pack) rounded to float, re-laid for the f32 MFMA
//   vt_trsm_rows_f32   : S = alpha diag(s) B (L L^T)^{-1} as an MFMA chain in f32
// The fp64 kernels replace the same reference lines (vittles/sensitivity_lib.py:354, 226); this file is the
// reduced-precision variant of vt_hessian.hip / vt_chol.hip and is never the headline number.
//
// Accumulation: an f32 MFMA is a k-ordered fmaf chain (no wider internal sum), so a wave adds at most
// kChunkSteps k-steps (4096 rows) into its f32 accumulators before they are folded into an fp64 slab in LDS:
// the rounding error of a sum stays <= ~kChunkSteps * 2^-24 relative, whatever N is.
#include "vt_common.h"
#include "vt_hess_stream.h"   // k_glm_hess_stream<float, ...>: the fused Hessian kernel shared with the narrow fp64 path

namespace {

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// ------------------------------------------------------------------------------------ Hessian
// Sum the P partial slabs in a fixed order, undo the f32 accumulator layout, mirror the triangle, add the prior.
__global__ __launch_bounds__(256) void k_hess_finalize_f32(const double* in, int L, int P, int NB, int D,
                                                           const double* prior, const double* theta, double* H,
                                                           int ldh, double* g) {
    __shared__ double sh[16][17];
    const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    double s = 0.0;
    if (e < L)
        for (int k = grp; k < P; k += 16) s += in[(int64_t)k * L + e];
    sh[grp][el] = s;
    __syncthreads();
    if (grp != 0 || e >= L) return;
    s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sh[k][el];
    const int NBT = NB * (NB + 1) / 2;
    if (e < NBT * 256) {
        const int blk = e >> 8, r = (e >> 6) & 3, lane = e & 63;
        int ba = 0;
        while ((ba + 1) * (ba + 2) / 2 <= blk) ++ba;
        const int bb = blk - ba * (ba + 1) / 2;
        const int i = 16 * ba + 4 * (lane >> 4) + r;
        const int j = 16 * bb + (lane & 15);
        if (i < D && j < D) {
            if (i == j && prior) s += prior[i];
            if (ba != bb || i >= j) {
                H[(int64_t)i * ldh + j] = s;
                H[(int64_t)j * ldh + i] = s;
            }
        }
    } else if (g) {
        const int t = e - NBT * 256;
        if (t < D) g[t] = s + (prior ? prior[t] * theta[t] : 0.0);
    }
}

constexpr int hess32_waves(int NB) { return NB <= 5 ? 8 : 4; }

template <int NB>
int launch_hess32(const HessStreamArgs<float>& a, int grid, hipStream_t st) {
    return hs_launch<float, NB, hess32_waves(NB)>(a, grid, st);
}

// ------------------------------------------------------------------------------------ TRSM
// f32 operand image: pack32[(PI * 64 + lane) * 4 + s] = M_PI[i = lane & 15][p = 4 (lane >> 4) + s]: the A
// operand of k-step s of block pair PI, one ds_read_b128 per product and lane.  M_PI are the matrices of the
// fp64 pack (vt_chol.hip: F_ji = -Linv_jj L_ji, B_ji = -Linv_jj^T L_ij^T, the diagonal ones the inverses).
// With the f32 C/D map a lane owns rows 4 q .. 4 q + 3 of an accumulator, i.e. 4 CONSECUTIVE parameters of one
// right-hand side (16 contiguous bytes of a row of X or S) and register s is k-step s of the next product:
// no index permutation is needed (the fp64 pack permutes its full blocks to get the same property).
__host__ __device__ __forceinline__ int perm16(bool permuted, int p) { return permuted ? (4 * (p & 3) + (p >> 2)) : p; }

__global__ __launch_bounds__(256) void k_trsm_pack_f32(const double* pack64, int D, float* pack32) {
    const int NB = (D + 15) >> 4, NBT = NB * (NB + 1) / 2;
    const bool lastperm = (D & 15) == 0;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < 2 * NBT * 256; e += gridDim.x * 256) {
        const int PI = e >> 8, lane = (e >> 2) & 63, s = e & 3;
        const int i = lane & 15, p = 4 * (lane >> 4) + s;
        const bool bwd = PI >= NBT;
        const int t = bwd ? PI - NBT : PI;
        int ba = 0;
        while ((ba + 1) * (ba + 2) / 2 <= t) ++ba;
        const int bb = t - ba * (ba + 1) / 2;
        const int rowblk = bwd ? bb : ba, srcblk = bwd ? ba : bb;   // forward tri(row, src), backward tri(src, row)
        const int ih = perm16(rowblk < NB - 1 || lastperm, i);
        const int kk = perm16(srcblk < NB - 1 || lastperm, p);
        pack32[e] = (float)pack64[((int64_t)PI * 4 + (kk >> 2)) * 64 + (kk & 3) * 16 + ih];
    }
}

struct Trsm32Args {
    const float* pack;
    const float* B;
    int64_t ldb;
    const double* rowscale;
    double alpha;
    int64_t N;
    int D;
    float* Out;
    int64_t ldo;
};

// product idx of the chain -> (row block j, source block i): forward j = 0..NB-1, i = j..0; backward
// j = NB-1..0, i = j..NB-1 (the diagonal product first: it reads V[j] before V[j] is overwritten)
template <int NB>
struct T32Sched {
    static constexpr int NBT = NB * (NB + 1) / 2;
    static constexpr bool bwd(int idx) { return idx >= NBT; }
    static constexpr int row(int idx) {
        if (idx < NBT) {
            int j = 0, k = idx;
            while (k > j) { k -= j + 1; ++j; }
            return j;
        }
        int k = idx - NBT, j = NB - 1;
        while (k >= NB - j) { k -= NB - j; --j; }
        return j;
    }
    static constexpr int src(int idx) {
        if (idx < NBT) {
            int j = 0, k = idx;
            while (k > j) { k -= j + 1; ++j; }
            return j - k;
        }
        int k = idx - NBT, j = NB - 1;
        while (k >= NB - j) { k -= NB - j; --j; }
        return j + k;
    }
    static constexpr bool first(int idx) { return src(idx) == row(idx); }
    static constexpr bool last(int idx) { return bwd(idx) ? src(idx) == NB - 1 : src(idx) == 0; }
    static constexpr int image(int idx) { return bwd(idx) ? NBT + vt_tri(src(idx), row(idx)) : vt_tri(row(idx), src(idx)); }
};

template <int NB, bool LASTFULL>
struct T32State {
    f32x4 V[NB];
    f32x4 acc, cur, nxt;
    const f32x4* lds;
    int lane, q, D;
    float* orow;
    bool store;
};

template <int J, int NB, bool LASTFULL>
__device__ __forceinline__ void t32_store(T32State<NB, LASTFULL>& st) {
    if constexpr (J < NB - 1 || LASTFULL) {
        *(f32x4u*)(st.orow + 16 * J + 4 * st.q) = st.acc;
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int col = 16 * J + 4 * st.q + s;
            if (col < st.D) st.orow[col] = st.acc[s];
        }
    }
}

template <int IDX, int END, int NB, bool LASTFULL>
__device__ __forceinline__ void t32_chain(T32State<NB, LASTFULL>& st) {
    using S = T32Sched<NB>;
    if constexpr (IDX < END) {
        if constexpr (IDX + 1 < END) {
            constexpr int img = S::image(IDX + 1);   // constexpr: otherwise the schedule is walked at run time
            st.nxt = st.lds[img * 64 + st.lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int i = S::src(IDX), j = S::row(IDX);
        if constexpr (S::first(IDX)) st.acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) st.acc = mfma_f32(st.cur[s], st.V[i][s], st.acc);
        if constexpr (S::last(IDX)) {
            st.V[j] = st.acc;
            if (S::bwd(IDX) && st.store) t32_store<j, NB, LASTFULL>(st);
        }
        __builtin_amdgcn_sched_barrier(0);
        st.cur = st.nxt;
        t32_chain<IDX + 1, END, NB, LASTFULL>(st);
    }
}

// Out[n,:] = alpha s_n (L L^T)^{-1} B[n,:] in f32: one 16-row tile per wave and pass, right-hand sides on the
// MFMA column axis, the whole substitution a chain of 2 NBT block products against the LDS-resident pack.
template <int NB, bool LASTFULL, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_trsm_rows_f32(Trsm32Args a) {
    using S = T32Sched<NB>;
    constexpr int NBT = S::NBT;
    __shared__ f32x4 lds[2 * NBT * 64];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int64_t W = (int64_t)gridDim.x * WAVES;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t ngroups = (a.N + 15) / 16;

    T32State<NB, LASTFULL> st;
    st.lds = lds;
    st.lane = lane;
    st.q = q;
    st.D = a.D;

    f32x4 Gn[NB];
    float coefn;
    // the row tiles are walked from the last one down (like k_trsm_rows_small: the rows the Hessian pass read last are
    // the ones still in the memory-side cache -- 0.4 GB of float X at the headline shape, 256 MB of cache)
    auto load_group = [&](int64_t grp) {
        const int64_t row = (ngroups - 1 - grp) * 16 + n;
        const bool rowok = row < a.N;
        const int64_t rowc = rowok ? row : a.N - 1;
        const float* br = a.B + rowc * a.ldb;
        const double sc = a.rowscale ? a.rowscale[rowc] : 1.0;
        coefn = rowok ? (float)(a.alpha * sc) : 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b < NB - 1 || LASTFULL) {
                Gn[b] = *(const f32x4u*)(br + 16 * b + 4 * q);
            } else {
                f32x4 v;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int col = 16 * b + 4 * q + s;
                    const float x = br[min(col, a.D - 1)];
                    v[s] = col < a.D ? x : 0.f;
                }
                Gn[b] = v;
            }
        }
    };

    int64_t grp = gw;
    if (grp < ngroups) load_group(grp);
    {
        const f32x4* src = (const f32x4*)a.pack;
        for (int i = threadIdx.x; i < 2 * NBT * 64; i += WAVES * 64) lds[i] = src[i];
    }
    __syncthreads();
    for (; grp < ngroups; grp += W) {
#pragma unroll
        for (int b = 0; b < NB; ++b) st.V[b] = Gn[b] * coefn;
        const int64_t row = (ngroups - 1 - grp) * 16 + n;
        st.store = row < a.N;
        st.orow = a.Out + (st.store ? row : 0) * a.ldo;
        load_group(grp + W < ngroups ? grp + W : grp);
        constexpr int img_f = S::image(0), img_b = S::image(NBT);
        st.cur = lds[img_f * 64 + lane];
        t32_chain<0, NBT, NB, LASTFULL>(st);
        st.cur = lds[img_b * 64 + lane];
        t32_chain<NBT, 2 * NBT, NB, LASTFULL>(st);
    }
}

constexpr int kTrsm32Waves = 8;

template <int NB>
void launch_trsm32(const Trsm32Args& a, hipStream_t st) {
    const int64_t ngroups = (a.N + 15) / 16;
    int64_t g = (ngroups + kTrsm32Waves - 1) / kTrsm32Waves;
    if (g > vt_num_cu()) g = vt_num_cu();
    if (g < 1) g = 1;
    if (a.D % 16 == 0)
        hipLaunchKernelGGL((k_trsm_rows_f32<NB, true, kTrsm32Waves>), dim3((int)g), dim3(kTrsm32Waves * 64), 0, st, a);
    else
        hipLaunchKernelGGL((k_trsm_rows_f32<NB, false, kTrsm32Waves>), dim3((int)g), dim3(kTrsm32Waves * 64), 0, st, a);
}

}  // namespace

// D > 128 in the fp32 storage mode: link + gradient by the float row kernel (one pass over the float X), then the
// tiled SYRK with float panels (vt_large.hip: k_syrk_lds<..., AT = float>) -- half the HBM bytes of the fp64 path,
// every product and sum in fp64.  The row kernels bound it at D <= 4096.
extern "C" size_t vt_rows_workspace_bytes(int D);
extern "C" int vt_glm_grad_f32(int family, const float* X, int64_t ldx, const double* theta, const double* y,
                               const double* w, const double* offset, double scale, const double* prior, int64_t N,
                               int D, double* g_out, double* r_out, double* d_out, void* ws, size_t ws_bytes,
                               void* stream);
constexpr int kF32LargeDMax = 4096;
static size_t large32_ws_rows(int D) { return (vt_rows_workspace_bytes(D) + 255) / 256 * 256; }

extern "C" size_t vt_glm_hessian_f32_workspace_bytes(int64_t N, int D) {
    if (D < 1 || D > kF32LargeDMax || N < 0) return 0;
    if (D > VT_SMALL_D_MAX)
        return large32_ws_rows(D) + ((size_t)(N + D + 64) * sizeof(double) + 255) / 256 * 256 + vt_large_syrk_ws_bytes(D);
    return ((size_t)vt_num_cu() * hs_slab_len((D + 15) / 16) + 1024) * sizeof(double);
}

extern "C" int vt_glm_hessian_f32(int family, const float* X, int64_t ldx, const double* theta, const double* y,
                                  const double* w, const double* offset, double scale, const double* prior,
                                  int64_t N, int D, double* H, int ldh, double* g, double* r, double* d, void* ws,
                                  size_t ws_bytes, void* stream) {
    if (!theta || !H || N < 0 || D < 1 || ldx < D || ldh < D) return VT_ERR_BAD_ARG;
    if (N > 0 && (!X || !y || !w)) return VT_ERR_BAD_ARG;
    if (family < 0 || family > 2) return VT_ERR_BAD_ARG;
    if (D > kF32LargeDMax) return VT_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < vt_glm_hessian_f32_workspace_bytes(N, D)) return VT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (D > VT_SMALL_D_MAX) {
        char* base = (char*)ws;
        double* dtmp = (double*)(base + large32_ws_rows(D));
        double* gtmp = dtmp + N;
        void* syrk_ws = (void*)(base + large32_ws_rows(D) + ((size_t)(N + D + 64) * sizeof(double) + 255) / 256 * 256);
        double* dd = d ? d : dtmp;
        int rc = vt_glm_grad_f32(family, X, ldx, theta, y, w, offset, scale, prior, N, D, g ? g : gtmp, r, dd, ws,
                                 large32_ws_rows(D), stream);
        if (rc != VT_OK) return rc;
        return vt_large_syrk_f32(X, ldx, dd, N, D, prior, H, ldh, syrk_ws, st);
    }
    const int NB = (D + 15) / 16;
    const int L = hs_slab_len(NB);
    double* part = (double*)ws;
    double* sink = part + (size_t)vt_num_cu() * L;
    int grid = 0;
    if (N > 0) {
        const int64_t nsteps = (N + 3) / 4;
        grid = (int)std::max<int64_t>(1, std::min<int64_t>((nsteps + 31) / 32, vt_num_cu()));   // >= 4 k-steps per wave
        HessStreamArgs<float> a{family, X, ldx, theta, y, w, offset, scale, N, D, r, d, part, sink};
        int rc = VT_ERR_UNSUPPORTED;
        // row-link kernel (vt_hess_stream.h): 32-bit descriptor offsets, so moderate leading dimensions only
        static const bool no_rowlink = getenv("VT_HESS_NO_ROWLINK") != nullptr;   // dev A/B switch
        // (N = 10^6: D = 8 0.036 -> 0.026 ms, 20 0.048 -> 0.035, 64 0.082 -> 0.075; from NB = 5 on the DPP-broadcast
        //  kernel is faster -- 0.106 against 0.115 ms at D = 80, and NB >= 7 spills here)
        const bool rowlink = !no_rowlink && NB <= 4 && ldx <= ((int64_t)1 << 22);
        if (rowlink) grid = rl_grid(N, NB, vt_num_cu());
#define VT_HESS32_CASE(K)                                                                               \
    case K:                                                                                             \
        if constexpr (K <= 4) {                                                                         \
            if (rowlink) {                                                                              \
                rc = rl_launch<float, K>(a, grid, st);                                                  \
                break;                                                                                  \
            }                                                                                           \
        }                                                                                               \
        rc = launch_hess32<K>(a, grid, st);                                                             \
        break;
        switch (NB) {
#ifdef VT_F32_DEV_NB   // dev builds (tools/build_dev32.sh): one block count only, to keep the compile short
            VT_HESS32_CASE(VT_F32_DEV_NB)
#else
            VT_HESS32_CASE(1)
            VT_HESS32_CASE(2)
            VT_HESS32_CASE(3)
            VT_HESS32_CASE(4)
            VT_HESS32_CASE(5)
            VT_HESS32_CASE(6)
            VT_HESS32_CASE(7)
            VT_HESS32_CASE(8)
#endif
        }
#undef VT_HESS32_CASE
        if (rc != VT_OK) return rc;
    }
    hipLaunchKernelGGL(k_hess_finalize_f32, dim3((L + 15) / 16), dim3(256), 0, st, (const double*)part, L, grid, NB, D,
                       prior, theta, H, ldh, g);
    return vt_launch_status();
}

extern "C" size_t vt_trsm_pack_f32_bytes(int D) {
    if (D < 1 || D > VT_SMALL_D_MAX) return 0;
    const int NB = (D + 15) / 16;
    return (size_t)2 * (NB * (NB + 1) / 2) * 256 * sizeof(float);
}

extern "C" int vt_trsm_pack_f32(const void* pack64, int D, void* pack32, void* stream) {
    if (!pack64 || !pack32 || D < 1) return VT_ERR_BAD_ARG;
    if (D > VT_SMALL_D_MAX) return VT_ERR_UNSUPPORTED;
    const int NB = (D + 15) / 16;
    hipLaunchKernelGGL(k_trsm_pack_f32, dim3(NB * (NB + 1)), dim3(256), 0, (hipStream_t)stream, (const double*)pack64, D,
                       (float*)pack32);
    return vt_launch_status();
}

extern "C" int vt_trsm_rows_f32(const void* pack32, int D, const float* B, int64_t ldb, const double* rowscale,
                                double alpha, int64_t N, float* Out, int64_t ldo, void* stream) {
    if (!pack32 || D < 1 || ldb < D || ldo < D || N < 0) return VT_ERR_BAD_ARG;
    if (D > VT_SMALL_D_MAX) return VT_ERR_UNSUPPORTED;
    if (N == 0) return VT_OK;
    if (!B || !Out) return VT_ERR_BAD_ARG;
    Trsm32Args a{(const float*)pack32, B, ldb, rowscale, alpha, N, D, Out, ldo};
    hipStream_t st = (hipStream_t)stream;
    switch ((D + 15) / 16) {
        case 1: launch_trsm32<1>(a, st); break;
        case 2: launch_trsm32<2>(a, st); break;
        case 3: launch_trsm32<3>(a, st); break;
        case 4: launch_trsm32<4>(a, st); break;
        case 5: launch_trsm32<5>(a, st); break;
        case 6: launch_trsm32<6>(a, st); break;
        case 7: launch_trsm32<7>(a, st); break;
        case 8: launch_trsm32<8>(a, st); break;
        default: return VT_ERR_UNSUPPORTED;
    }
    return vt_launch_status();
}

// ---- fp32 storage mode, D > 128: the sensitivity solve in row slabs -------------------------------------------
// Out (float) [n, :] = alpha * s_n * (L L^T)^{-1} B (float) [n, :]: a slab of rows is widened into a double scratch,
// solved in place by the fp64 chain / blocked kernels (vt_trsm_rows with the ordinary fp64 pack) and narrowed into
// the float result.  Storage (X and S: 4 N D bytes each) is what the mode halves; the solve itself is MFMA-bound at
// these D, and the two conversions add 24 bytes of traffic per element of a slab.
namespace {
__global__ void k_widen_rows(const float* B, int64_t ldb, int64_t n, int D, double* W, int64_t ldw) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tot = n * ldw;
    if (idx >= tot) return;
    const int64_t r = idx / ldw;
    const int c = (int)(idx - r * ldw);
    W[idx] = c < D ? (double)B[r * ldb + c] : 0.0;
}
__global__ void k_narrow_rows(const double* W, int64_t ldw, int64_t n, int D, float* Out, int64_t ldo) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * D) return;
    const int64_t r = idx / D;
    const int c = (int)(idx - r * D);
    Out[r * ldo + c] = (float)W[r * ldw + c];
}
constexpr int64_t kSlabRows = 131072;   // 512 rows per CU: the persistent solve kernels stay filled
}  // namespace

extern "C" int vt_trsm_rows(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha,
                            int64_t N, double* Out, int64_t ldo, void* stream);

extern "C" size_t vt_trsm_rows_f32_large_workspace_bytes(int64_t N, int D) {
    if (N < 0 || D <= VT_SMALL_D_MAX || D > VT_DENSE_D_MAX) return 0;
    return (size_t)std::min<int64_t>(std::max<int64_t>(N, 1), kSlabRows) * (size_t)(D + (D & 1)) * sizeof(double);
}

extern "C" int vt_trsm_rows_f32_large(const void* pack, int D, const float* B, int64_t ldb, const double* rowscale,
                                      double alpha, int64_t N, float* Out, int64_t ldo, void* ws, size_t ws_bytes,
                                      void* stream) {
    if (!pack || D < 1 || ldb < D || ldo < D || N < 0) return VT_ERR_BAD_ARG;
    if (D <= VT_SMALL_D_MAX || D > VT_DENSE_D_MAX) return VT_ERR_UNSUPPORTED;
    if (N == 0) return VT_OK;
    if (!B || !Out) return VT_ERR_BAD_ARG;
    const int64_t ldw = D + (D & 1);
    const int64_t slab = ws ? (int64_t)(ws_bytes / ((size_t)ldw * sizeof(double))) : 0;
    if (slab < std::min<int64_t>(N, 4096)) return VT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double* W = (double*)ws;
    for (int64_t r0 = 0; r0 < N; r0 += slab) {
        const int64_t n = std::min(slab, N - r0);
        hipLaunchKernelGGL(k_widen_rows, dim3((unsigned)((n * ldw + 255) / 256)), dim3(256), 0, st, B + r0 * ldb, ldb, n, D,
                           W, ldw);
        int rc = vt_trsm_rows(pack, D, W, ldw, rowscale ? rowscale + r0 : nullptr, alpha, n, W, ldw, stream);
        if (rc != VT_OK) return rc;
        hipLaunchKernelGGL(k_narrow_rows, dim3((unsigned)((n * D + 255) / 256)), dim3(256), 0, st, (const double*)W, ldw, n,
                           D, Out + r0 * ldo, ldo);
    }
    return vt_launch_status();
}
