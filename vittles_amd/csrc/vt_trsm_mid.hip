// Mid-range (129 <= D <= 384) TRSM: operand pack in CHAIN ORDER and the dispatch over the chain kernels of
// vt_trsm_mid.h (instantiated in vt_trsm_mid_{a,b,c,e,f,g,h,i,j}.hip; the halves of a full block in _d).  Replaces the same reference lines as vt_chol.hip
// (scipy.linalg.cho_solve with N right-hand sides, vittles/solver_lib.py:28-29, vittles/sensitivity_lib.py:226).
#include "vt_common.h"
#include "vt_trsm_sched.h"
#include <algorithm>
#include <cstdlib>

int vt_mid_trsm_a(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha,
                  int64_t N, double* Out, int64_t ldo, hipStream_t st);
int vt_mid_trsm_b(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha,
                  int64_t N, double* Out, int64_t ldo, hipStream_t st);
int vt_mid_trsm_c(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha,
                  int64_t N, double* Out, int64_t ldo, hipStream_t st);

#define VT_MID_DECL(X)                                                                                              \
    int vt_mid_trsm_##X(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha, \
                        int64_t N, double* Out, int64_t ldo, hipStream_t st);
VT_MID_DECL(e) VT_MID_DECL(f) VT_MID_DECL(g) VT_MID_DECL(h) VT_MID_DECL(i) VT_MID_DECL(j)   // NB = 17-18, 19-20, 21, 22, 23, 24
#undef VT_MID_DECL

int vt_mid_trsm_d(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha, int64_t N,
                  double* Out, int64_t ldo, int mode, hipStream_t st);

namespace {

// Linv_JJ of every 16 x 16 diagonal block (padding rows past D: identity).  One wave per block, lane c = column c:
//   x_i = (delta_ic - sum_{m<i} L[i][m] x_m) / L[i][i]
__global__ __launch_bounds__(64) void k_mid_linv(const double* L, int ldl, int D, double* linv) {
    const int J = blockIdx.x, o = 16 * J, c = threadIdx.x & 15;
    if (threadIdx.x >= 16) return;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const bool real = o + i < D;
        double sacc = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int m = 0; m < i; ++m) {
            const double lim = real ? L[(int64_t)(o + i) * ldl + o + m] : 0.0;
            sacc = fma(-lim, x[m], sacc);
        }
        const double lii = real ? L[(int64_t)(o + i) * ldl + o + i] : 1.0;
        x[i] = i < c ? 0.0 : sacc / lii;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) linv[J * 256 + i * 16 + c] = x[i];
}

// Image of chain product idx (workgroup idx, one wave): forward F_jj = Linv_jj, F_ji = -Linv_jj L_ji (i < j);
// backward B_jj = Linv_jj^T, B_ji = -Linv_jj^T L_ij^T (i > j); element (row, k) -> A-operand lane order with the
// index permutations of the chain (vt_perm: a full block keeps logical index 4q+s at register s of lane group q).
__global__ __launch_bounds__(64) void k_mid_pack(const double* L, int ldl, int D, const double* linv, double* pack) {
    const int NB = (D + 15) >> 4, NBT = NB * (NB + 1) / 2;
    const int idx = blockIdx.x;
    const bool bwd = idx >= NBT;
    int j, i;
    if (!bwd) {       // forward: j = 0..NB-1, i = j..0
        j = 0;
        int k = idx;
        while (k > j) { k -= j + 1; ++j; }
        i = j - k;
    } else {          // backward: j = NB-1..0, i = j..NB-1
        int k = idx - NBT;
        j = NB - 1;
        while (k >= NB - j) { k -= NB - j; --j; }
        i = j + k;
    }
    const int lane = threadIdx.x, q = lane >> 4, c = lane & 15;
    const double* lj = linv + j * 256;
    auto Lat = [&](int r, int cc) -> double { return (r < D && cc < D && cc <= r) ? L[(int64_t)r * ldl + cc] : 0.0; };
    double m[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = q + 4 * r;
        double v;
        if (i == j) {
            v = bwd ? lj[c * 16 + row] : lj[row * 16 + c];
        } else {
            v = 0.0;
            for (int k = 0; k < 16; ++k) {
                if (!bwd) v = fma(-lj[row * 16 + k], Lat(16 * j + k, 16 * i + c), v);
                else v = fma(-lj[k * 16 + row], Lat(16 * i + c, 16 * j + k), v);
            }
        }
        m[r] = v;
    }
    const bool lastperm = (D & 15) == 0;
    const bool pj = j < NB - 1 || lastperm, pi = i < NB - 1 || lastperm;
    const int kk = vt_perm(pi, c);
    double* dst = pack + ((int64_t)idx * 4 + (kk >> 2)) * 64 + (kk & 3) * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[vt_perm(pj, q + 4 * r)] = m[r];
}

inline int mid_images(int D) {
    const int NB = (D + 15) / 16;
    return NB * (NB + 1);
}

}  // namespace

// mode 0: full solve (any 129 <= D <= 384); 1 / 2: forward-only / backward-only (D = 384 or 256 only)
int vt_mid_trsm_rows_mode(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha,
                          int64_t N, double* Out, int64_t ldo, int mode, hipStream_t st) {
    if (mode == 0) return vt_mid_trsm_rows(pack, D, B, ldb, rowscale, alpha, N, Out, ldo, st);
    if (D != VT_TRSM_ONEPASS_D_MAX && D != VT_MID_D_MAX) return VT_ERR_UNSUPPORTED;
    if (N == 0) return VT_OK;
    return vt_mid_trsm_d(pack, D, B, ldb, rowscale, alpha, N, Out, ldo, mode, st);
}

// The one-pass chain holds a 16 x D tile of right-hand sides per wave: 8 D / 16 + ~60 registers of the 256 a wave has
// at two waves per SIMD, so it reaches past the Hessian kernels' D <= VT_MID_D_MAX, to NB = 24 (NB = 20: 226 VGPRs;
// NB = 24: all 256 and 9 spilled -- still the fastest solve of the library: 79 % of the fp64-MFMA peak at D = 384).
constexpr int kMidTrsmDMax = VT_TRSM_ONEPASS_D_MAX;
bool vt_mid_trsm_ok(int D) {
    static const bool off = getenv("VT_TRSM_NO_MID") != nullptr;   // dev A/B switches, read once
    static const int dmax = getenv("VT_TRSM_MID_MAX_D") ? atoi(getenv("VT_TRSM_MID_MAX_D")) : kMidTrsmDMax;
    return !off && D > VT_SMALL_D_MAX && D <= std::min(dmax, kMidTrsmDMax);
}

// [chain-order images | Linv blocks]; the image area doubles as the scratch of the blocked factorisation
size_t vt_mid_pack_bytes(int D) {
    const int NB = (D + 15) / 16;
    const size_t images = std::max((size_t)mid_images(D) * 256 * sizeof(double), vt_large_potrf_ws_bytes(D));
    return (images + 255) / 256 * 256 + (size_t)NB * 256 * sizeof(double);
}

int vt_mid_pack(const double* L, int ldl, int D, void* pack, hipStream_t st) {
    const int NB = (D + 15) / 16;
    const size_t images = std::max((size_t)mid_images(D) * 256 * sizeof(double), vt_large_potrf_ws_bytes(D));
    double* linv = (double*)((char*)pack + (images + 255) / 256 * 256);
    hipLaunchKernelGGL(k_mid_linv, dim3(NB), dim3(64), 0, st, L, ldl, D, linv);
    hipLaunchKernelGGL(k_mid_pack, dim3(mid_images(D)), dim3(64), 0, st, L, ldl, D, (const double*)linv, (double*)pack);
    return vt_launch_status();
}

int vt_mid_trsm_rows(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha,
                     int64_t N, double* Out, int64_t ldo, hipStream_t st) {
    if (N == 0) return VT_OK;
    const int NB = (D + 15) / 16;
    if (NB <= 12) return vt_mid_trsm_a(pack, D, B, ldb, rowscale, alpha, N, Out, ldo, st);
    if (NB <= 14) return vt_mid_trsm_b(pack, D, B, ldb, rowscale, alpha, N, Out, ldo, st);
    if (NB <= 16) return vt_mid_trsm_c(pack, D, B, ldb, rowscale, alpha, N, Out, ldo, st);
    switch (NB) {
        case 17: case 18: return vt_mid_trsm_e(pack, D, B, ldb, rowscale, alpha, N, Out, ldo, st);
        case 19: case 20: return vt_mid_trsm_f(pack, D, B, ldb, rowscale, alpha, N, Out, ldo, st);
        case 21: return vt_mid_trsm_g(pack, D, B, ldb, rowscale, alpha, N, Out, ldo, st);
        case 22: return vt_mid_trsm_h(pack, D, B, ldb, rowscale, alpha, N, Out, ldo, st);
        case 23: return vt_mid_trsm_i(pack, D, B, ldb, rowscale, alpha, N, Out, ldo, st);
        case 24: return vt_mid_trsm_j(pack, D, B, ldb, rowscale, alpha, N, Out, ldo, st);
    }
    return VT_ERR_UNSUPPORTED;
}
