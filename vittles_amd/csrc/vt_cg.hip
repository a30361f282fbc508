// K6 driver pieces: the BLAS-1 part of one conjugate-gradient iteration, fused in ONE launch.
// Replaces the five NumPy vector operations per iteration inside scipy.sparse.linalg.cg as the
// reference calls it (vittles/solver_lib.py:91-97; SciPy's loop: rho = r.r; p = r + beta p;
// q = A p; alpha = rho / p.q; x += alpha p; r -= alpha q).  The matrix-vector product q = A p is
// the fused HVP kernel (vt_glm_hvp) or a user callable; everything else is this kernel.
//
// All vectors are replicated on every rank (D doubles: 160 KB at D = 20k), so no collective is
// needed here; a single 1024-thread workgroup walks them (deterministic reductions).
#include "vt_common.h"

namespace {

__device__ __forceinline__ double block_sum(double v, double* red) {
    v = vt_sum64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i];
    return s;
}

// scal[0] = rho (in: r.r of the current residual; out: of the new one), scal[1] = ||r_new||,
// scal[2] = alpha, scal[3] = p.q.
//   alpha = rho / p.q;  x += alpha p;  r -= alpha q;  rho' = r.r;  p = r + (rho'/rho) p
__global__ __launch_bounds__(1024) void k_cg_update(double* p, const double* q, double* x, double* r, int64_t D,
                                                    double* scal) {
    __shared__ double red[16];
    const int tid = threadIdx.x;
    double pq = 0.0;
    for (int64_t i = tid; i < D; i += 1024) pq = fma(p[i], q[i], pq);
    pq = block_sum(pq, red);
    const double rho = scal[0];
    const double alpha = rho / pq;
    double rr = 0.0;
    for (int64_t i = tid; i < D; i += 1024) {
        x[i] = fma(alpha, p[i], x[i]);
        double ri = fma(-alpha, q[i], r[i]);
        r[i] = ri;
        rr = fma(ri, ri, rr);
    }
    rr = block_sum(rr, red);
    const double beta = rr / rho;
    for (int64_t i = tid; i < D; i += 1024) p[i] = fma(beta, p[i], r[i]);
    if (tid == 0) {
        scal[0] = rr;
        scal[1] = sqrt(rr);
        scal[2] = alpha;
        scal[3] = pq;
    }
}

// Start of a solve: r = b - q0 (q0 = A x0, or NULL for x0 = 0), p = r, scal[0] = r.r, scal[1] = ||r||,
// scal[4] = ||b||.
__global__ __launch_bounds__(1024) void k_cg_init(const double* b, const double* q0, double* r, double* p,
                                                  int64_t D, double* scal) {
    __shared__ double red[16];
    const int tid = threadIdx.x;
    double rr = 0.0, bb = 0.0;
    for (int64_t i = tid; i < D; i += 1024) {
        double bi = b[i];
        double ri = q0 ? bi - q0[i] : bi;
        r[i] = ri;
        p[i] = ri;
        rr = fma(ri, ri, rr);
        bb = fma(bi, bi, bb);
    }
    rr = block_sum(rr, red);
    bb = block_sum(bb, red);
    if (tid == 0) {
        scal[0] = rr;
        scal[1] = sqrt(rr);
        scal[2] = 0.0;
        scal[3] = 0.0;
        scal[4] = sqrt(bb);
    }
}

}  // namespace

namespace {
// ---- K columns in lock-step (block CG with independent recurrences), optional diagonal preconditioner ----------
// Column k lives at offset k * ld of every vector array; scal holds 8 doubles per column:
//   [0] rho = r.z   [1] ||r||   [2] alpha   [3] p.q   [4] ||b||   [5] iterations done   [6] atol (set by the host
//   after the init: the column is FROZEN -- neither updated nor counted -- once ||r|| < atol)   [7] unused
// minv (D,) or NULL: z = minv .* r is SciPy's `M @ r` for a diagonal M ~ A^{-1} (scipy.sparse.linalg.cg's loop:
// z = psolve(r); rho = r.z; p = z + (rho/rho_prev) p; the stopping test stays on ||r||).
__global__ __launch_bounds__(1024) void k_cg_update_multi(double* P, const double* Q, double* Xs, double* R, int64_t D,
                                                          int64_t ld, const double* minv, double* scal_all) {
    __shared__ double red[16];
    const int tid = threadIdx.x;
    double* scal = scal_all + 8 * blockIdx.x;
    if (scal[1] < scal[6]) return;   // converged column (uniform over the workgroup)
    double* p = P + blockIdx.x * ld;
    const double* q = Q + blockIdx.x * ld;
    double* x = Xs + blockIdx.x * ld;
    double* r = R + blockIdx.x * ld;
    double pq = 0.0;
    for (int64_t i = tid; i < D; i += 1024) pq = fma(p[i], q[i], pq);
    pq = block_sum(pq, red);
    const double rho = scal[0];
    const double alpha = rho / pq;
    double rr = 0.0, rz = 0.0;
    for (int64_t i = tid; i < D; i += 1024) {
        x[i] = fma(alpha, p[i], x[i]);
        double ri = fma(-alpha, q[i], r[i]);
        r[i] = ri;
        rr = fma(ri, ri, rr);
        if (minv) rz = fma(ri * minv[i], ri, rz);
    }
    rr = block_sum(rr, red);
    if (minv) rz = block_sum(rz, red);
    else rz = rr;
    const double beta = rz / rho;
    for (int64_t i = tid; i < D; i += 1024) p[i] = fma(beta, p[i], minv ? minv[i] * r[i] : r[i]);
    if (tid == 0) {
        scal[0] = rz;
        scal[1] = sqrt(rr);
        scal[2] = alpha;
        scal[3] = pq;
        scal[5] += 1.0;
    }
}

__global__ __launch_bounds__(1024) void k_cg_init_multi(const double* B, const double* Q0, double* R, double* P,
                                                        int64_t D, int64_t ld, const double* minv, double* scal_all) {
    __shared__ double red[16];
    const int tid = threadIdx.x;
    double* scal = scal_all + 8 * blockIdx.x;
    const double* b = B + blockIdx.x * ld;
    const double* q0 = Q0 ? Q0 + blockIdx.x * ld : nullptr;
    double* r = R + blockIdx.x * ld;
    double* p = P + blockIdx.x * ld;
    double rr = 0.0, bb = 0.0, rz = 0.0;
    for (int64_t i = tid; i < D; i += 1024) {
        double bi = b[i];
        double ri = q0 ? bi - q0[i] : bi;
        double zi = minv ? minv[i] * ri : ri;
        r[i] = ri;
        p[i] = zi;
        rr = fma(ri, ri, rr);
        rz = fma(ri, zi, rz);
        bb = fma(bi, bi, bb);
    }
    rr = block_sum(rr, red);
    rz = block_sum(rz, red);
    bb = block_sum(bb, red);
    if (tid == 0) {
        scal[0] = rz;
        scal[1] = sqrt(rr);
        scal[2] = 0.0;
        scal[3] = 0.0;
        scal[4] = sqrt(bb);
        scal[5] = 0.0;
        scal[6] = 0.0;
        scal[7] = 0.0;
    }
}
}  // namespace

extern "C" int vt_cg_init_multi(const double* b, const double* q0, double* r, double* p, int64_t D, int64_t ld, int K,
                                const double* minv, double* scal, void* stream) {
    if (!b || !r || !p || !scal || D < 1 || K < 1 || (K > 1 && ld < D)) return VT_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_cg_init_multi, dim3(K), dim3(1024), 0, (hipStream_t)stream, b, q0, r, p, D, ld, minv, scal);
    return vt_launch_status();
}

extern "C" int vt_cg_update_multi(double* p, const double* q, double* x, double* r, int64_t D, int64_t ld, int K,
                                  const double* minv, double* scal, void* stream) {
    if (!p || !q || !x || !r || !scal || D < 1 || K < 1 || (K > 1 && ld < D)) return VT_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_cg_update_multi, dim3(K), dim3(1024), 0, (hipStream_t)stream, p, q, x, r, D, ld, minv, scal);
    return vt_launch_status();
}

extern "C" int vt_cg_init(const double* b, const double* q0, double* r, double* p, int64_t D, double* scal,
                          void* stream) {
    if (!b || !r || !p || !scal || D < 1) return VT_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_cg_init, dim3(1), dim3(1024), 0, (hipStream_t)stream, b, q0, r, p, D, scal);
    return vt_launch_status();
}

extern "C" int vt_cg_update(double* p, const double* q, double* x, double* r, int64_t D, double* scal,
                            void* stream) {
    if (!p || !q || !x || !r || !scal || D < 1) return VT_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_cg_update, dim3(1), dim3(1024), 0, (hipStream_t)stream, p, q, x, r, D, scal);
    return vt_launch_status();
}

// out[0] = sum_i a_i b_i (c_i): the small vector reductions of the host layer -- dot products, squared norms
// (a == b), the prior term sum_j prior_j theta_j^2 (c == b) -- in ONE workgroup with a fixed summation order
// (bit-reproducible, unlike a library dot that picks its grid by size).
namespace {
__global__ __launch_bounds__(1024) void k_dot3(const double* a, const double* b, const double* c, int64_t n,
                                               double* out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s = fma(c ? a[i] * c[i] : a[i], b[i], s);
    s = vt_sum64(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        out[0] = t;
    }
}
}  // namespace

extern "C" int vt_dot3(const double* a, const double* b, const double* c, int64_t n, double* out, void* stream) {
    if (!out || n < 0 || (n > 0 && (!a || !b))) return VT_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_dot3, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, b, c, n, out);
    return vt_launch_status();
}

// ---- sparse matrix-vector product (CSR, fp64 values, int32 indices) for K vectors at once ----------------------
// Out[k][i] = sum_p data[p] X[k][indices[p]],  p in [indptr[i], indptr[i+1]):  16 lanes walk a row, summed by DPP in a
// fixed order.  The `mat_times_vec` of the iterative route of get_sparse_cholesky_solver for sparse SPD matrices
// that are neither block-arrowhead nor small enough to densify (the reference hands them to SuperLU,
// vittles/solver_lib.py:46-48).
namespace {
__global__ __launch_bounds__(256) void k_spmv_csr(const int* indptr, const int* indices, const double* data, int n,
                                                  const double* X, int64_t ldx, double* Out, int64_t ldo) {
    const int row = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4), c = threadIdx.x & 15;
    const double* x = X + (int64_t)blockIdx.y * ldx;
    double s = 0.0;
    if (row < n) {
        const int p1 = indptr[row + 1];
        for (int p = indptr[row] + c; p < p1; p += 16) s = fma(data[p], x[indices[p]], s);
    }
    s = vt_sum16(s);
    if (row < n && c == 0) Out[(int64_t)blockIdx.y * ldo + row] = s;
}
}  // namespace

extern "C" int vt_spmv_csr(const int* indptr, const int* indices, const double* data, int n, const double* X,
                           int64_t ldx, int K, double* Out, int64_t ldo, void* stream) {
    if (!indptr || !X || !Out || n < 1 || K < 1 || (K > 1 && (ldx < n || ldo < n))) return VT_ERR_BAD_ARG;
    if (K > 65535) return VT_ERR_UNSUPPORTED;
    const unsigned gx = (unsigned)(((int64_t)n * 16 + 255) / 256);
    hipLaunchKernelGGL(k_spmv_csr, dim3(gx, K), dim3(256), 0, (hipStream_t)stream, indptr, indices, data, n, X, ldx, Out,
                       ldo);
    return vt_launch_status();
}
