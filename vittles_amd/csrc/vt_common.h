// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vittles_hip.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));
// 16-byte vector of two doubles that is only guaranteed 8-byte aligned (rows of X
// are 8-byte aligned for odd D); gfx950 global loads/stores need dword alignment.
typedef double f64x2u __attribute__((ext_vector_type(2), aligned(8)));

#define VT_WAVE 64

#define VT_CHECK_HIP(expr)                         \
    do {                                           \
        hipError_t _e = (expr);                    \
        if (_e != hipSuccess) return VT_ERR_HIP;   \
    } while (0)

static inline int vt_launch_status() {
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4) * B(4x16), one f64 of A and B per lane.
//   A operand: lane l holds A[i = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][j = l & 15]
//   C/D      : lane l, register r holds D[i = (l >> 4) + 4 r][j = l & 15]
// (f64 does NOT use the f32/bf16 C/D map; cdna_hip_programming.md section 3.)
__device__ __forceinline__ f64x4 mfma_f64(double a, double b, f64x4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global
// memory (s_waitcnt vmcnt(0)), which drains every prefetch in flight -- fatal inside a
// software-pipelined loop.
__device__ __forceinline__ void vt_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ int vt_nblocks(int D) { return (D + 15) >> 4; }

// v + (v rotated right by N lanes inside each row of 16 lanes), on the VALU (DPP row_ror):
// ~10x lower latency than the ds_bpermute a __shfl_xor costs.
template <int N>
__device__ __forceinline__ double vt_ror16_add(double v) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + N, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + N, 0xf, 0xf, true);
    return v + __hiloint2double(hi, lo);
}

// Sum over the 16 lanes that share (lane >> 4); every lane gets the total (DPP version).
__device__ __forceinline__ double vt_sum16_dpp(double v) {
    v = vt_ror16_add<8>(v);
    v = vt_ror16_add<4>(v);
    v = vt_ror16_add<2>(v);
    v = vt_ror16_add<1>(v);
    return v;
}

// 1/y to ~1 ulp without the IEEE division sequence (v_rcp_f64 + two Newton steps).
__device__ __forceinline__ double vt_rcp(double y) {
    double x = __builtin_amdgcn_rcp(y);
    x = fma(fma(-y, x, 1.0), x, x);
    x = fma(fma(-y, x, 1.0), x, x);
    return x;
}

// exp(x) in fp64 (<= 1 ulp: range reduction by ln2 in two pieces + degree-11 minimax
// polynomial + ldexp; overflow -> inf and underflow -> 0 come out of v_ldexp_f64).
// The coefficients sit in constant memory so that they are scalar-loaded into SGPRs:
// libm's exp() makes hipcc park 10 of them in VGPR pairs for the whole kernel, which the
// MFMA kernels cannot afford.
__constant__ double kVtExpC[14] = {
    1.4426950408889634,        // log2(e)
    -0.6931471805599453,       // -ln2 (high part)
    -2.3190468138462996e-17,   // -ln2 (low part)
    0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22, 0x1.71dee623fde64p-19, 0x1.a01997c89e6b0p-16,
    0x1.a01a014761f6ep-13, 0x1.6c16c1852b7b0p-10, 0x1.1111111122322p-7, 0x1.55555555502a1p-5,
    0x1.5555555555511p-3, 0x1.000000000000bp-1, 1.0};

__device__ __forceinline__ double vt_exp(double x) {
    const double* C = kVtExpC;
    double n = __builtin_rint(x * C[0]);
    double r = fma(n, C[1], x);
    r = fma(n, C[2], r);
    double p = fma(r, C[3], C[4]);
#pragma unroll
    for (int i = 5; i <= 12; ++i) p = fma(r, p, C[i]);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    // |x| beyond the fp64 exponent range: keep n representable for the int conversion
    n = fmin(fmax(n, -2200.0), 2200.0);
    return ldexp(p, (int)n);
}

// Per-row link derivatives: l1 = ell'(z), l2 = ell''(z).
__device__ __forceinline__ void vt_link(int family, double z, double y, double& l1, double& l2) {
    if (family == VT_FAMILY_LOGISTIC) {
        // p = sigmoid(z), evaluated on the stable side
        double e = vt_exp(-fabs(z));
        double inv = vt_rcp(1.0 + e);   // 1 + e in (1, 2]: no overflow / denormal cases
        double p = z >= 0.0 ? inv : e * inv;
        l1 = p - y;
        l2 = e * inv * inv;  // p (1 - p) without cancellation
    } else if (family == VT_FAMILY_POISSON) {
        double e = vt_exp(z);
        l1 = e - y;
        l2 = e;
    } else {
        l1 = z - y;
        l2 = 1.0;
    }
}

// Per-row higher link derivatives (independent of y): l2 = ell''(z), l3 = ell'''(z).
__device__ __forceinline__ void vt_link3(int family, double z, double& l2, double& l3) {
    if (family == VT_FAMILY_LOGISTIC) {
        double e = vt_exp(-fabs(z));
        double inv = vt_rcp(1.0 + e);
        l2 = e * inv * inv;                               // p (1 - p)
        double omp2 = (1.0 - e) * inv;                    // |1 - 2p| = (1 - e) / (1 + e)
        l3 = z >= 0.0 ? -l2 * omp2 : l2 * omp2;           // p (1 - p) (1 - 2p)
    } else if (family == VT_FAMILY_POISSON) {
        l2 = l3 = vt_exp(z);
    } else {
        l2 = 1.0;
        l3 = 0.0;
    }
}

// l2 = ell'', l3 = ell''', l4 = ell''''.
__device__ __forceinline__ void vt_link4(int family, double z, double& l2, double& l3, double& l4) {
    vt_link3(family, z, l2, l3);
    if (family == VT_FAMILY_LOGISTIC) {
        l4 = l2 * (1.0 - 6.0 * l2);          // p (1 - p) (1 - 6 p (1 - p))
    } else if (family == VT_FAMILY_POISSON) {
        l4 = l2;
    } else {
        l4 = 0.0;
    }
}

// Per-row loss value ell(z, y).
__device__ __forceinline__ double vt_loss(int family, double z, double y) {
    if (family == VT_FAMILY_LOGISTIC) {
        return fmax(z, 0.0) + log1p(exp(-fabs(z))) - y * z;
    } else if (family == VT_FAMILY_POISSON) {
        return exp(z) - y * z;
    } else {
        double t = z - y;
        return 0.5 * t * t;
    }
}

// Sum over the 16 lanes that share (lane >> 4); every lane gets the total.
__device__ __forceinline__ double vt_sum16(double v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// Sum over all 64 lanes; every lane gets the total.
__device__ __forceinline__ double vt_sum64(double v) {
    v = vt_sum16(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// Index of the (a, b) block, a >= b, in packed lower-triangular block order.
__host__ __device__ __forceinline__ constexpr int vt_tri(int a, int b) { return a * (a + 1) / 2 + b; }

// ---- internal (non-exported) entry points shared between translation units -------------------
#include <algorithm>
// Per-DEVICE host state (one process may drive several GPUs: `device=` is a public argument of the Python
// layer).  vt_num_cu(): compute units of the CURRENT device (256 when no device is visible, so that the
// workspace queries work on a CPU-only build host).  vt_allow_big_lds(): raise a kernel's dynamic-LDS limit
// once per (call site, device) -- function attributes are per device.
constexpr int VT_MAX_DEVICES = 64;
struct VtLdsAttr {
    bool done[VT_MAX_DEVICES] = {};
};
int vt_num_cu();
int vt_allow_big_lds(const void* func, VtLdsAttr& flag, int bytes);
int vt_potrf_pack_block(double* A, int lda, int D, int* info, double* pack, int do_factor, hipStream_t st);
int vt_trsm_rows_mode(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha,
                      int64_t N, double* Out, int64_t ldo, int mode, hipStream_t st);
size_t vt_large_potrf_ws_bytes(int D);
int vt_large_potrf(double* A, int lda, int D, int* info, void* ws, hipStream_t st);
size_t vt_large_pack_bytes(int D);
int vt_large_pack(const double* L, int ldl, int D, void* pack, hipStream_t st);
int vt_large_trsm_rows(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha,
                       int64_t N, double* Out, int64_t ldo, hipStream_t st);
// The one-pass solve chain (vt_trsm_mid.h) takes D <= VT_TRSM_ONEPASS_D_MAX: a wave holds a 16 x D tile of right-hand
// sides in registers (NB = 24: the 256 VGPRs of a wave at two waves per SIMD, 9 spilled).  Past it the blocked solve
// (vt_large.hip) uses diagonal blocks of that width -- 384 = three of the GEMM-NT updates' 128-column tiles (320-column
// blocks were measured: D = 1000 48.9 -> 56.9 ms, 2.5 tiles per block).
#define VT_TRSM_ONEPASS_D_MAX 384
bool vt_mid_trsm_ok(int D);
size_t vt_mid_pack_bytes(int D);
int vt_mid_pack(const double* L, int ldl, int D, void* pack, hipStream_t st);
int vt_mid_trsm_rows(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha,
                     int64_t N, double* Out, int64_t ldo, hipStream_t st);
int vt_mid_trsm_rows_mode(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha,
                          int64_t N, double* Out, int64_t ldo, int mode, hipStream_t st);
bool vt_hess_mid_ok(const double* X, int64_t ldx, int D);
size_t vt_hess_mid_ws_bytes(int D);
int vt_hess_mid(int family, const double* X, int64_t ldx, const double* theta, const double* y, const double* w,
                const double* offset, double scale, int64_t N, int D, double* r, double* d, double* part,
                int* nslabs, hipStream_t st);
size_t vt_large_syrk_ws_bytes(int D);
int vt_large_syrk_f32(const float* X, int64_t ldx, const double* d, int64_t N, int D, const double* prior, double* H,
                      int ldh, void* ws, hipStream_t st);
int vt_large_syrk(const double* X, int64_t ldx, const double* d, int64_t N, int D, const double* prior, double* H,
                  int ldh, void* ws, hipStream_t st);
