/* vittles_hip.h -- C-ABI of the MI355X (gfx950) linear-response sensitivity hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / C++ types.
 * Every pointer named *_dev is a DEVICE pointer (hipMalloc'd or owned by a torch
 * tensor); the library never allocates or frees device memory -- scratch comes in
 * through a `ws` pointer whose size the matching *_workspace_bytes() query states.
 * All kernels are enqueued on the `stream` argument (a hipStream_t passed as void*;
 * NULL = the default stream) and return without synchronising.
 *
 * Return value: 0 = ok; <0 = argument / HIP error (vt_error_string() names it);
 * LAPACK-style `info` values (k>0: leading minor k not positive definite) are
 * written to an `int* info_dev` on the device so that no call forces a host sync.
 *
 * Each entry point cites the interface of the reference (rgiordan/vittles, paths
 * relative to /root/reference) that it replaces.  The reference has no native
 * code at all: what is replaced is the work that autograd / SciPy perform inside
 * the cited Python lines.
 *
 * Conventions
 *   X        : (N, D) row-major fp64, leading dimension ldx (elements, >= D).
 *   theta    : (D,) fp64.   y, w, offset : (N,) fp64 (offset may be NULL).
 *   objective: f(theta, w) = scale * sum_n w_n ell(x_n.theta + offset_n, y_n)
 *                            + 0.5 * sum_j prior_j theta_j^2
 *   family   : VT_FAMILY_LOGISTIC  ell = log(1+e^z) - y z
 *              VT_FAMILY_POISSON   ell = e^z - y z
 *              VT_FAMILY_GAUSSIAN  ell = (z - y)^2 / 2
 *   r_n = scale * ell'(z_n)          (cross-Hessian column n is r_n * x_n)
 *   d_n = scale * w_n * ell''(z_n)   (Hessian row weight)
 *   H   : (D, D) row-major, FULL symmetric storage, leading dimension ldh.
 *   L   : lower Cholesky factor, H = L L^T, row-major in the lower triangle
 *         (== the transpose of scipy.linalg.cho_factor's default upper factor).
 *   N   : may be 0 (an empty data set): the row pointers (X, y, w, ...) may then be NULL and the
 *         results are those of an empty sum (H = prior, g = prior .* theta, no rows of S).
 *   S   : (N, D) row-major, row n = d theta-hat / d w_n.  Viewed transposed this
 *         is the reference's (D, N) `_sens_mat`, F-contiguous exactly as
 *         scipy.linalg.cho_solve returns it.
 */
#ifndef VITTLES_HIP_H
#define VITTLES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VT_OK                  0
#define VT_ERR_BAD_ARG        -1
#define VT_ERR_UNSUPPORTED    -2   /* shape outside what the kernels cover */
#define VT_ERR_WORKSPACE      -3   /* ws_bytes too small */
#define VT_ERR_HIP            -4   /* a HIP runtime call failed */
#define VT_ERR_NO_DEVICE      -5
#define VT_ERR_SCRATCH_LIMIT  -6   /* input larger than the static scratch of vt_glm_hvp_noalloc */

#define VT_FAMILY_LOGISTIC 0
#define VT_FAMILY_POISSON  1
#define VT_FAMILY_GAUSSIAN 2

/* Largest D handled by the register/LDS-resident ("small-D") SYRK / Cholesky /
 * TRSM kernels.  Larger D (up to VT_DENSE_D_MAX) takes the tiled kernels; the
 * row-streaming entry points (link, grad, HVP, gemv, value) take D <= VT_ROWS_D_MAX.
 * Anything larger returns VT_ERR_UNSUPPORTED. */
#define VT_SMALL_D_MAX 128
/* VT_SMALL_D_MAX < D <= VT_MID_D_MAX: one-pass kernels with the block triangle / the solve chain dealt over the
 * waves of a workgroup (csrc/vt_hess_mid.hip, csrc/vt_trsm_mid.hip); beyond, the 128 x 128 tiled kernels (the
 * one-pass solve chain itself reaches D = 384: its tile of right-hand sides still fits a wave's registers). */
#define VT_MID_D_MAX 256
#define VT_DENSE_D_MAX 8192
#define VT_ROWS_D_MAX  20448

int         vt_version(void);
const char* vt_error_string(int code);
/* Number of visible HIP devices (<0 on error).  Product code calls this to fail
 * loudly when no GPU is present. */
int         vt_device_count(void);

/* ---- K0 + K1 + K3 fused: link, gradient and Hessian in ONE pass over X --------
 * Replaces: autograd.hessian(objective_fun, argnum=0)(theta, w)
 *             vittles/sensitivity_lib.py:381-383  (also lr_cov_lib.py:101-102)
 *           autograd.grad(objective_fun, argnum=0)(theta, w)
 *             vittles/sensitivity_lib.py:354, 203-209
 *           and the per-row factors of the cross-Hessian
 *             vittles/sensitivity_lib.py:180-181, 217-219
 * H_dev (D,D) <- X^T diag(d) X (+ diag(prior) if prior_dev != NULL)
 * g_dev (D,)  <- X^T (w .* r)  (+ prior .* theta if prior_dev != NULL)   [may be NULL]
 * r_dev, d_dev (N,) per-row factors                                     [may be NULL]
 * Row-sharded use: every rank calls this on its rows; only one rank passes prior. */
size_t vt_glm_hessian_workspace_bytes(int64_t N, int D);
int    vt_glm_hessian(int family, const double* X_dev, int64_t ldx,
                      const double* theta_dev, const double* y_dev, const double* w_dev,
                      const double* offset_dev, double scale, const double* prior_dev,
                      int64_t N, int D,
                      double* H_dev, int ldh, double* g_dev, double* r_dev, double* d_dev,
                      void* ws_dev, size_t ws_bytes, void* stream);

/* ---- K4: dense Cholesky -------------------------------------------------------
 * Replaces: scipy.linalg.cho_factor(h)          vittles/solver_lib.py:26-27
 * In place on the lower triangle of the row-major A (strict upper part untouched).
 * *info_dev <- 0, or k>0 if the leading minor of order k is not positive definite
 * (LAPACK dpotrf's info, which SciPy turns into LinAlgError). */
size_t vt_potrf_workspace_bytes(int D);
int    vt_potrf(double* A_dev, int lda, int D, int* info_dev,
                void* ws_dev, size_t ws_bytes, void* stream);

/* ---- K5 (generic): solve with an arbitrary-stride right-hand side ----------------
 * Replaces: scipy.linalg.cho_solve(h_chol, v)   vittles/solver_lib.py:28-29
 * B(i,k) lives at B_dev[i*ldb_row + k*ldb_col], i<D, k<nrhs; solved in place. */
int    vt_potrs(const double* L_dev, int ldl, int D,
                double* B_dev, int64_t ldb_row, int64_t ldb_col, int64_t nrhs, void* stream);

/* ---- K5 (MFMA TRSM): operand pack + row-major multi-RHS solve ---------------------
 * vt_trsm_pack turns L into the MFMA A-operand images (inverted 16x16 diagonal
 * blocks folded into the off-diagonal blocks) the TRSM kernel keeps in LDS. */
size_t vt_trsm_pack_bytes(int D);
int    vt_trsm_pack(const double* L_dev, int ldl, int D, void* pack_dev, void* stream);

/* K4 + pack in ONE launch (the inverted diagonal blocks are a by-product of the blocked
 * factorisation): A <- L in place, *info_dev as vt_potrf, pack_dev as vt_trsm_pack.
 * Replaces: scipy.linalg.cho_factor(h)   vittles/solver_lib.py:26-27 */
int    vt_potrf_pack(double* A_dev, int lda, int D, int* info_dev, void* pack_dev, void* stream);
/* The same, out of place: H_dev is left untouched (the reference keeps `self._hess0` for
 * get_hessian_at_opt(), vittles/sensitivity_lib.py:385, 411-412), L_dev gets the factor in its lower
 * triangle and zeros above it.  Saves the copy an in-place factorisation needs. */
int    vt_potrf_pack_from(const double* H_dev, int ldh, double* L_dev, int ldl, int D, int* info_dev,
                          void* pack_dev, void* stream);

/* Out[n,:] = alpha * s_n * H^{-1} B[n,:]   for n < N  (s_n = 1 if rowscale_dev NULL)
 * Replaces: `-1 * self._hess_solver(self._estimating_equation_jac0)`
 *             vittles/sensitivity_lib.py:226  (dpotrs with nrhs = N)
 * With B = X, rowscale = r, alpha = -1 this is K2+K5 fused: the (D,N) cross-Hessian
 * (sensitivity_lib.py:217-219) is never materialised. */
int    vt_trsm_rows(const void* pack_dev, int D,
                    const double* B_dev, int64_t ldb, const double* rowscale_dev, double alpha,
                    int64_t N, double* Out_dev, int64_t ldo, void* stream);

/* ---- Row-streaming kernels (HBM-bound; X read once) ------------------------------
 * All take a workspace of vt_rows_workspace_bytes(D) bytes. */
size_t vt_rows_workspace_bytes(int D);

/* K0: z_n = x_n.theta + offset_n, r_n, d_n (any of the three outputs may be NULL).
 * Replaces the forward pass autograd traces inside objective_fun
 *   (vittles/sensitivity_lib.py:354-383 evaluate it once per derivative sweep). */
int    vt_glm_link(int family, const double* X_dev, int64_t ldx, const double* theta_dev,
                   const double* y_dev, const double* w_dev, const double* offset_dev, double scale,
                   int64_t N, int D, double* z_dev, double* r_dev, double* d_dev,
                   void* ws_dev, size_t ws_bytes, void* stream);

/* K3: g = X^T (w .* r) (+ prior .* theta).
 * Replaces: autograd.grad(objective_fun, argnum=0)(theta, w)
 *             vittles/sensitivity_lib.py:354, 203-209; vittles/lr_cov_lib.py:110 */
int    vt_glm_grad(int family, const double* X_dev, int64_t ldx, const double* theta_dev,
                   const double* y_dev, const double* w_dev, const double* offset_dev, double scale,
                   const double* prior_dev, int64_t N, int D, double* g_dev, double* r_dev, double* d_dev,
                   void* ws_dev, size_t ws_bytes, void* stream);

/* K6: fused Hessian-vector product  out = X^T (d .* (X v)) + prior .* v
 * (prior_dev (D,) if non-NULL, else the scalar prior_scalar).
 * Replaces: the `mat_times_vec` callable a user hands to get_cg_solver
 *             vittles/solver_lib.py:70-98 (autograd.hessian_vector_product,
 *             vittles/lr_cov_lib.py:79-80), one call per CG iteration. */
int    vt_glm_hvp(const double* X_dev, int64_t ldx, const double* d_dev, const double* prior_dev,
                  double prior_scalar, const double* v_dev, int64_t N, int D, double* out_dev,
                  void* ws_dev, size_t ws_bytes, void* stream);

/* K6 for K vectors at once: Out[k, :] = X^T (d .* (X V[k, :])) + prior .* V[k, :]  (V, Out: K x D, one vector per
 * row, K <= 128): two passes of the MFMA tile kernels over X (Z = X V^T, then Z^T diag(d) X) -- 16 N D bytes for
 * the K products where K calls of vt_glm_hvp read 8 K N D.  Needs ldx even and X 16-byte aligned for the LDS-staged
 * kernels (other layouts take the direct-from-global kernels).
 * Replaces: the `mat_times_vec` of get_cg_solver applied to EVERY column of the cross-Hessian
 *             vittles/solver_lib.py:91-97 under vittles/sensitivity_lib.py:226 (`self._hess_solver(jac0)` with a
 *             (D, N) jac0: SciPy's cg takes one right-hand side, so the reference's CG route cannot run it; the
 *             block solver here is the build-side extension SURVEY.md section 3.3 names). */
size_t vt_glm_hvp_multi_workspace_bytes(int64_t N, int D, int K);
int    vt_glm_hvp_multi(const double* X_dev, int64_t ldx, const double* d_dev, const double* prior_dev,
                        double prior_scalar, const double* V_dev, int64_t ldv, int K, int64_t N, int D,
                        double* Out_dev, int64_t ldo, void* ws_dev, size_t ws_bytes, void* stream);

/* vt_glm_hvp without a workspace argument, for callers that cannot allocate.
 * No counterpart in the reference: this entry point exists for callers that must not call hipMalloc
 * (code under HIP graph capture, tight serving loops around the CG iteration, harnesses that forbid
 * allocation) and so cannot size and hand over `ws` either.  Same kernels and bit-identical results as
 * vt_glm_hvp; the per-workgroup partial slabs live in a static device array of VT_NOALLOC_SCRATCH_BYTES
 * (41.9 MB, held by every GPU the library runs on until the process ends), enough for every D the row
 * kernels take (D <= VT_ROWS_D_MAX on a GPU of <= 256 CUs); anything larger returns
 * VT_ERR_SCRATCH_LIMIT.  Calls no hipMalloc / hipFree.  All calls on one GPU share that GPU's array:
 * calls that overlap on one GPU (two streams) race.  The first call on a device resolves the array's
 * address (hipGetSymbolAddress): make it once outside any graph capture. */
#define VT_NOALLOC_SCRATCH_BYTES ((size_t)256 * (VT_ROWS_D_MAX + 1) * 8)
int    vt_glm_hvp_noalloc(const double* X_dev, int64_t ldx, const double* d_dev, const double* prior_dev,
                          double prior_scalar, const double* v_dev, int64_t N, int D, double* out_dev,
                          void* stream);

/* K9: out (D,) = A^T u for a row-major (N, D) A.
 * Replaces: `self._sens_mat @ (new_hyper_par_value - self._hyper_val0)`
 *             vittles/sensitivity_lib.py:245-247 (A = S stored (N, D), u = delta w). */
int    vt_gemv_t(const double* A_dev, int64_t lda, const double* u_dev, int64_t N, int D,
                 double* out_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* out (D,) = A^T (u1 .* u2): first-order Taylor right-hand side  (d^2 f / d theta d w) dw = X^T (r .* dw).
 * Replaces: the order-1 term of ParametricSensitivityTaylorExpansion._evaluate_dkinput_dhyperk
 *             vittles/sensitivity_lib.py:1209-1260 (forward-mode jvp of the estimating equation). */
int    vt_gemv_t2(const double* A_dev, int64_t lda, const double* u1_dev, const double* u2_dev, int64_t N, int D,
                  double* out_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* Second-order data-weight term of a GLM, one pass over X:
 *   out (D,) = scale * sum_n [ w_n ell'''(z_n) (x_n.u)^2 + 2 dw_n ell''(z_n) (x_n.u) ] x_n
 * = G_{theta theta}[u, u] + 2 G_{theta w}[u, dw] for the estimating equation G = grad_theta f (f is linear
 * in w, so G_{ww} = 0); with u = d theta-hat/dw . dw the second directional derivative of the optimum is
 * -H^{-1} out.  z_dev is vt_glm_link's z output.
 * Replaces: the order-2 terms summed by ParametricSensitivityTaylorExpansion._evaluate_dkinput_dhyperk
 *             vittles/sensitivity_lib.py:1209-1260 (terms from _get_taylor_base_terms / DerivativeTerm.differentiate,
 *             :495-688), each a nested autograd.make_jvp of the gradient. */
int    vt_glm_taylor2_rhs(int family, const double* X_dev, int64_t ldx, const double* z_dev, const double* w_dev,
                          const double* dw_dev, double scale, const double* u_dev, int64_t N, int D,
                          double* out_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* ANY order k of the data-weight Taylor expansion (2 <= k <= VT_TAYLOR_MAX_ORDER), as a per-row coefficient:
 *   c_n = scale * [ w_n  sum_{m=2..min(k, mio)}    ell^(m+1)(z_n) B_{k,m}(zeta_1, ..)
 *                 + k dw_n sum_{m=1..min(k-1, mio)} ell^(m+1)(z_n) B_{k-1,m}(zeta_1, ..) ]
 * with zeta_dev a (k-1) x N array (row j-1, leading dimension ldz: zeta_j = X d_j from vt_gemv), B_{n,m} the
 * partial Bell polynomials and mio = max_input_order (>= k: no truncation; the reference drops every term with
 * more input partial derivatives than that, :1247-1249).  The order-k derivative of the optimum is
 * -H^{-1} X^T c (vt_gemv_t + the Cholesky solve); orders 2 and 3 also have the one-pass kernels around this.
 * Replaces: ParametricSensitivityTaylorExpansion._evaluate_dkinput_dhyperk vittles/sensitivity_lib.py:1209-1260 at
 *           any order (the reference's nested autograd.make_jvp terms, :495-688, for the registered GLM objective). */
#define VT_TAYLOR_MAX_ORDER 8
int    vt_glm_taylor_coeff(int family, const double* z_dev, const double* w_dev, const double* dw_dev,
                           const double* zeta_dev, int64_t ldz, double scale, int order, int max_input_order,
                           int64_t N, double* c_dev, void* stream);

/* Third-order data-weight term of a GLM, one pass over X (zeta2_dev = X d2 from vt_gemv, u1 = d1):
 *   out = scale * sum_n [ w_n (ell'''' z1^3 + 3 ell''' z1 z2) + 3 dw_n (ell''' z1^2 + ell'' z2) ] x_n,
 *   z1 = x_n.d1, z2 = zeta2_n; the third directional derivative of the optimum is -H^{-1} out.
 * Replaces: the order-3 terms of ParametricSensitivityTaylorExpansion._evaluate_dkinput_dhyperk
 *             vittles/sensitivity_lib.py:1209-1260. */
int    vt_glm_taylor3_rhs(int family, const double* X_dev, int64_t ldx, const double* z_dev, const double* w_dev,
                          const double* dw_dev, const double* zeta2_dev, double scale, const double* u1_dev,
                          int64_t N, int D, double* out_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* K9, many weight vectors at once:  C (Da, Db) = alpha * A^T diag(s) B + 1 bias^T  for row-major
 * A (N, Da), B (N, Db); s (N,) and bias (Db,) may be NULL.  fp64 MFMA, N axis split over the chip,
 * fixed-order reduction.  With A = the matrix whose COLUMNS are new weight vectors (bootstrap draws,
 * K-fold / leave-one-out indicator weights), B = S and bias = theta-hat - S^T w0 this is
 * Replaces: `self._sens_mat @ (new_hyper_par_value - self._hyper_val0)`
 *             vittles/sensitivity_lib.py:245-247 called once per weight vector in a Python loop
 *             (docs/source/example_notebooks/mle_weight_sensitivity_example.ipynb cells 24-26;
 *              bootstrap weights: docs/source/example_notebooks/example_utils.py:19-20).
 * Da, Db <= VT_DENSE_D_MAX; workspace from vt_gemm_tn_workspace_bytes (0 = unsupported shape). */
size_t vt_gemm_tn_workspace_bytes(int Da, int Db);
int    vt_gemm_tn(const double* A_dev, int64_t lda, int Da, const double* B_dev, int64_t ldb, int Db,
                  const double* rowscale_dev, int64_t N, double alpha, const double* bias_dev,
                  double* C_dev, int64_t ldc, void* ws_dev, size_t ws_bytes, void* stream);

/* out (N,) = A v for a row-major (N, D) A (moment Jacobians times a vector, K8). */
int    vt_gemv(const double* A_dev, int64_t lda, const double* v_dev, int64_t N, int D,
               double* out_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* value_dev[0] = scale * sum_n w_n ell(z_n, y_n)  (the caller adds the prior term).
 * Replaces: evaluating objective_fun(theta, w) itself (vittles/sensitivity_lib.py:303-320). */
int    vt_glm_value(int family, const double* X_dev, int64_t ldx, const double* theta_dev,
                    const double* y_dev, const double* w_dev, const double* offset_dev, double scale,
                    int64_t N, int D, double* value_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* ---- fp32 STORAGE mode (BASELINE.json north_star: "optional fp32 mode", 1e-4 against the fp64 oracle) ------
 * The two N x D arrays that set the HBM traffic -- X and the sensitivity matrix S -- are float; theta, y, w,
 * offset, the per-row factors r and d, H, g, the Cholesky factor and every D-vector stay double.  The row
 * operators carry every sum in fp64 (they agree with fp64 arithmetic on the rounded X to ~1e-12); the two MFMA
 * kernels multiply in f32 (v_mfma_f32_16x16x4_f32), vt_glm_hessian_f32 also takes its row dot x_n . theta in f32
 * (its r, d, g are f32-accurate, ~1e-6: use vt_glm_grad_f32 / vt_glm_link_f32 where fp64 sums matter, e.g. an
 * optimiser's stopping test) and folds its f32 accumulators into fp64 every <= 4096 rows per wave.
 * D <= VT_SMALL_D_MAX for the f32-MFMA Hessian / TRSM kernels (beyond: the tiled route below, D <= 4096).  Row
 * operators: D <= 20448 when ldx is a multiple of 4 and X is 16-byte aligned (rows on a 16-byte pitch: four
 * floats per lane above 4096 columns -- configs[3]'s D = 20000 CG route on a float X); else D <= 4096 when D and
 * ldx are even and X is 8-byte aligned (two floats per lane), D <= 2048 otherwise (odd D, odd ldx or a 4-byte
 * aligned base take the scalar configurations); anything larger returns VT_ERR_UNSUPPORTED (use the fp64 entry
 * points, or pad the row pitch).  Each entry point replaces the same
 * reference lines as its fp64 namesake above and takes the same arguments, with float where noted. */
size_t vt_glm_hessian_f32_workspace_bytes(int64_t N, int D);
int    vt_glm_hessian_f32(int family, const float* X_dev, int64_t ldx, const double* theta_dev,
                          const double* y_dev, const double* w_dev, const double* offset_dev, double scale,
                          const double* prior_dev, int64_t N, int D, double* H_dev, int ldh, double* g_dev,
                          double* r_dev, double* d_dev, void* ws_dev, size_t ws_bytes, void* stream);
/* pack32_dev (vt_trsm_pack_f32_bytes(D) bytes) <- the operand pack of vt_potrf_pack / vt_trsm_pack (the
 * factorisation itself is always fp64), rounded to float and laid out for the f32 MFMA chain. */
size_t vt_trsm_pack_f32_bytes(int D);
int    vt_trsm_pack_f32(const void* pack_dev, int D, void* pack32_dev, void* stream);
/* Out[n, :] = alpha * rowscale[n] * (L L^T)^{-1} B[n, :]  with float B and Out (rowscale stays double). */
int    vt_trsm_rows_f32(const void* pack32_dev, int D, const float* B_dev, int64_t ldb,
                        const double* rowscale_dev, double alpha, int64_t N, float* Out_dev, int64_t ldo,
                        void* stream);
/* D > VT_SMALL_D_MAX in the fp32 mode (configs[2]'s D = 1000 with float X and S: half the 160 GB footprint):
 * vt_glm_hessian_f32 takes D <= 4096 (float row kernel for link + gradient, tiled SYRK with float panels, every
 * product and sum in fp64); the solve below takes the ORDINARY fp64 pack of vt_potrf_pack / vt_trsm_pack and a
 * scratch of vt_trsm_rows_f32_large_workspace_bytes(N, D) bytes (a slab of rows is widened to double, solved by
 * the fp64 kernels and narrowed into the float result; a smaller scratch means more, smaller slabs).
 * Replaces: the same lines as vt_trsm_rows (vittles/sensitivity_lib.py:226, vittles/solver_lib.py:28-29). */
size_t vt_trsm_rows_f32_large_workspace_bytes(int64_t N, int D);
int    vt_trsm_rows_f32_large(const void* pack_dev, int D, const float* B_dev, int64_t ldb,
                              const double* rowscale_dev, double alpha, int64_t N, float* Out_dev, int64_t ldo,
                              void* ws_dev, size_t ws_bytes, void* stream);
/* the row operators (workspace: vt_rows_workspace_bytes(D), as for the fp64 ones) */
int    vt_glm_link_f32(int family, const float* X_dev, int64_t ldx, const double* theta_dev,
                       const double* y_dev, const double* w_dev, const double* offset_dev, double scale,
                       int64_t N, int D, double* z_dev, double* r_dev, double* d_dev, void* ws_dev,
                       size_t ws_bytes, void* stream);
int    vt_glm_grad_f32(int family, const float* X_dev, int64_t ldx, const double* theta_dev,
                       const double* y_dev, const double* w_dev, const double* offset_dev, double scale,
                       const double* prior_dev, int64_t N, int D, double* g_dev, double* r_dev, double* d_dev,
                       void* ws_dev, size_t ws_bytes, void* stream);
int    vt_glm_hvp_f32(const float* X_dev, int64_t ldx, const double* d_dev, const double* prior_dev,
                      double prior_scalar, const double* v_dev, int64_t N, int D, double* out_dev, void* ws_dev,
                      size_t ws_bytes, void* stream);
int    vt_gemv_t_f32(const float* A_dev, int64_t lda, const double* u_dev, int64_t N, int D, double* out_dev,
                     void* ws_dev, size_t ws_bytes, void* stream);
int    vt_gemv_t2_f32(const float* A_dev, int64_t lda, const double* u1_dev, const double* u2_dev, int64_t N, int D,
                      double* out_dev, void* ws_dev, size_t ws_bytes, void* stream);
int    vt_gemv_f32(const float* A_dev, int64_t lda, const double* v_dev, int64_t N, int D, double* out_dev,
                   void* ws_dev, size_t ws_bytes, void* stream);
int    vt_glm_value_f32(int family, const float* X_dev, int64_t ldx, const double* theta_dev,
                        const double* y_dev, const double* w_dev, const double* offset_dev, double scale,
                        int64_t N, int D, double* value_dev, void* ws_dev, size_t ws_bytes, void* stream);
/* vt_gemm_tn with B stored in float (the fp32 mode's S): C (Da, Db) = alpha * A^T diag(s) B + 1 bias^T, A, s, the
 * products, the sums and C in double.  Same reference lines as vt_gemm_tn (sensitivity_lib.py:245-247 in a loop);
 * workspace from vt_gemm_tn_workspace_bytes. */
int    vt_gemm_tn_f32b(const double* A_dev, int64_t lda, int Da, const float* B_dev, int64_t ldb, int Db,
                       const double* rowscale_dev, int64_t N, double alpha, const double* bias_dev,
                       double* C_dev, int64_t ldc, void* ws_dev, size_t ws_bytes, void* stream);

/* ---- Synthetic inputs: counter-based generator ----------------------------------
 * Philox-4x32-10, key = seed, counter = (row, column pair, stream_id): any block of rows can be regenerated
 * independently (row shards of one global data set; host-side spot checks -- vittles_amd/synth.py implements the
 * same generator in NumPy, identical bits, normal transform equal to a few ulps).  No counterpart in the reference,
 * whose tests draw from np.random with fixed seeds (vittles/tests/test_sensitivity_lib.py:463-478).
 * vt_synth_normal: X[n][j] ~ N(0,1) for rows row0 .. row0+N-1 of the global matrix (Box-Muller of the two uniforms
 *   of counter (row0 + n, j / 2, stream_id); stream_id 0 = design matrix, 1 = responses, 2 = theta*).
 * vt_synth_response: y_n from z_n = x_n . theta* and the uniforms of counter (row0 + n, 0, 1):
 *   logistic 1{u < sigmoid(z)}; poisson: inverse CDF of Poisson(exp(z)); gaussian z + N(0,1). */
int    vt_synth_normal(uint64_t seed, unsigned stream_id, int64_t row0, int64_t N, int D, double* X_dev,
                       int64_t ldx, void* stream);
int    vt_synth_normal_f32(uint64_t seed, unsigned stream_id, int64_t row0, int64_t N, int D, float* X_dev,
                           int64_t ldx, void* stream);
int    vt_synth_response(uint64_t seed, int64_t row0, int64_t N, int family, const double* z_dev, double* y_dev,
                         void* stream);

/* ---- K6 driver: the BLAS-1 part of one CG iteration, fused into one launch -------------
 * Replaces: the loop body of scipy.sparse.linalg.cg as called by
 *             vittles/solver_lib.py:91-97 (SciPy _isolve/iterative.py: rho = r.r;
 *             p = r + beta p; q = A p; alpha = rho / p.q; x += alpha p; r -= alpha q).
 * scal_dev: 8 doubles on the device: [0] rho = r.r, [1] ||r||, [2] alpha, [3] p.q, [4] ||b||.
 * vt_cg_init : r = b - q0 (q0 = A x0; NULL means x0 = 0), p = r, fills scal.
 * vt_cg_update: given q = A p:  alpha = rho/p.q; x += alpha p; r -= alpha q;
 *               rho' = r.r; p = r + (rho'/rho) p.  The host reads scal[1] to test convergence. */
int    vt_cg_init(const double* b_dev, const double* q0_dev, double* r_dev, double* p_dev, int64_t D,
                  double* scal_dev, void* stream);
int    vt_cg_update(double* p_dev, const double* q_dev, double* x_dev, double* r_dev, int64_t D,
                    double* scal_dev, void* stream);
/* The same for K right-hand sides in lock-step (independent recurrences: column k of every array at offset k*ld)
 * with an optional DIAGONAL preconditioner minv_dev (D,) -- SciPy's `M` (cg_opts of vittles/solver_lib.py:93):
 * z = minv .* r, rho = r.z, p = z + (rho/rho_prev) p, stopping test on ||r||.
 * scal_dev: 8 doubles per column: [0] rho, [1] ||r||, [2] alpha, [3] p.q, [4] ||b||, [5] iterations done,
 * [6] atol, written by the host after the init: a column with ||r|| < atol is frozen (not updated, not counted).
 * Replaces: scipy.sparse.linalg.cg run once per column of the right-hand side (vittles/solver_lib.py:91-97 under
 *             vittles/sensitivity_lib.py:226). */
int    vt_cg_init_multi(const double* b_dev, const double* q0_dev, double* r_dev, double* p_dev, int64_t D,
                        int64_t ld, int K, const double* minv_dev, double* scal_dev, void* stream);
int    vt_cg_update_multi(double* p_dev, const double* q_dev, double* x_dev, double* r_dev, int64_t D,
                          int64_t ld, int K, const double* minv_dev, double* scal_dev, void* stream);

/* ---- multi-GPU: the path's one data-path collective through the C-ABI -------------------------------------------
 * One process (or thread) per GPU; every rank holds a contiguous row shard of (X, y, w) and calls the entry points
 * above on it with prior_dev = NULL on every rank but one.  The only exchange of the dense path is the sum
 * all-reduce of [H | g] (lay them out in one buffer of D*D + pad + D doubles and reduce it in place before
 * vt_potrf_pack_from; S stays sharded like X), of the CG route the D-vector of vt_glm_hvp (or the K x D block of
 * vt_glm_hvp_multi) once per iteration.  These four bind RCCL directly (librccl.so is dlopen'ed at the first call;
 * VT_ERR_UNSUPPORTED if it is not there); the Python mirror uses torch.distributed (backend "nccl" = RCCL) instead.
 *   vt_comm_unique_id: 128 bytes made on rank 0 (ncclGetUniqueId), handed to the other ranks by the host's own means.
 *   vt_comm_init     : ncclCommInitRank on the CURRENT device (hipSetDevice first); collective over all ranks.
 *   vt_allreduce_sum : in place, fp64, on `stream`; ordered with the kernels queued on that stream.
 * No counterpart in the reference (a single-process NumPy library); SURVEY.md section 8b/8e's `vt_set_comm` hook. */
int    vt_comm_unique_id(void* id128);
int    vt_comm_init(void** comm_out, int nranks, int rank, const void* id128);
int    vt_comm_destroy(void* comm);
int    vt_allreduce_sum(void* comm, double* buf_dev, int64_t count, void* stream);

/* Out[k][i] = sum_p data[p] X[k][indices[p]], p in [indptr[i], indptr[i+1]): CSR (int32 indices, fp64 values) times
 * K vectors stored as the rows of X (K x n, leading dimension ldx), fixed summation order.
 * Replaces: the solves scipy.sparse.linalg.factorized (SuperLU) does for a GENERAL sparse matrix in
 *             get_sparse_cholesky_solver (vittles/solver_lib.py:46-48): for a sparse SPD matrix that is neither
 *             block-arrowhead nor small enough for the dense device Cholesky the Python mirror runs the block CG
 *             above (Jacobi preconditioner, relative residual 1e-12) over this product. */
int    vt_spmv_csr(const int* indptr_dev, const int* indices_dev, const double* data_dev, int n,
                   const double* X_dev, int64_t ldx, int K, double* Out_dev, int64_t ldo, void* stream);

/* out_dev[0] = sum_i a_i b_i c_i  (c_dev NULL: sum_i a_i b_i), one workgroup, fixed summation order.
 * Replaces: the NumPy reductions around the path -- np.linalg.norm of the estimating equation / of the Newton
 *             step (vittles/sensitivity_lib.py:209, vittles/lr_cov_lib.py:112), and the prior term of the
 *             objective value -- so that no library reduction kernel sits in the optimiser's loop. */
int    vt_dot3(const double* a_dev, const double* b_dev, const double* c_dev, int64_t n, double* out_dev,
               void* stream);

/* ---- K7: hierarchical mean-field Gaussian VI objective (BASELINE.json configs[4]) --------
 * Model and parameter layout: see oracle/vi.py / vittles_amd/objectives.py.
 * eta = [m_0 v_0 ... m_{G-1} v_{G-1} m_mu v_mu], d = 2k(G+1); X is (G*ng, k) row-major.
 * global_terms: 1 = the whole objective.  0 = a SHARD of the groups (one rank of a group-sharded objective)
 * that does not own the terms of the global rows / of the value that are not sums over groups (mu's prior
 * and entropy): the shards' global rows (the last 2k entries of out) and values then add up to the whole
 * objective's -- one all-reduce of 2k doubles (collective C3).
 * Replaces: autograd.grad(f) and the forward-over-reverse HVP `_f_fwd_hess`
 *             vittles/sparse_hessian_lib.py:59-60 (vittles/sensitivity_lib.py:440-492),
 *             vittles/lr_cov_lib.py:77-80, for this registered objective. */
size_t vt_vi_workspace_bytes(int G, int k);
int    vt_vi_hvp(const double* X_dev, const double* w_dev, const double* eta_dev, const double* u_dev,
                 double tau_y, double tau_th, double tau_mu, int G, int ng, int k, int global_terms,
                 double* out_dev, void* ws_dev, size_t ws_bytes, void* stream);
int    vt_vi_grad(const double* X_dev, const double* y_dev, const double* w_dev, const double* eta_dev,
                  double tau_y, double tau_th, double tau_mu, int G, int ng, int k, int global_terms,
                  double* out_dev, void* ws_dev, size_t ws_bytes, void* stream);
int    vt_vi_value(const double* X_dev, const double* y_dev, const double* w_dev, const double* eta_dev,
                   double tau_y, double tau_th, double tau_mu, int G, int ng, int k, int global_terms,
                   double* value_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* ---- K7: SparseBlockHessian assembly ------------------------------------------------------
 * vt_block_colour_vec: v <- 0; v[inds[b, ib]] = 1 for all b
 *   Replaces: SparseBlockHessian._hess_summed_term   vittles/sparse_hessian_lib.py:62-67
 * vt_block_gather: after HVP number ib, COO triplets e = (ib*G + b)*M + r in the reference's
 *   order plus the dense image blocks[b][r][ib].
 *   Replaces: the Python double loop            vittles/sparse_hessian_lib.py:93-102
 * vt_global_gather: after the HVP of global index number jg: 2*nloc + 2*Kg triplets in the
 *   reference's order (two 0.5*h halves for global pairs) plus border[(b*M+r)][jg], corner[j2][jg].
 *   Replaces: get_global_hessian's loops         vittles/sparse_hessian_lib.py:130-156 */
int    vt_block_colour_vec(const int64_t* inds_dev, int G, int M, int ib, double* v_dev, int64_t d,
                           void* stream);
int    vt_block_gather(const double* hv_dev, const int64_t* inds_dev, int G, int M, int ib,
                       double* vals_dev, int64_t* rows_dev, int64_t* cols_dev, double* blocks_dev,
                       void* stream);
int    vt_global_gather(const double* hrow_dev, const int64_t* local_inds_dev, int64_t nloc,
                        const int64_t* ginds_dev, int Kg, int jg, double* vals_dev, int64_t* rows_dev,
                        int64_t* cols_dev, double* border_dev, double* corner_dev, void* stream);

/* ---- K8: batched block-arrowhead solver  H = [[blockdiag(B_g), C], [C^T, Gm]] ---------------
 * Replaces: scipy.sparse.linalg.factorized(h) (SuperLU)   vittles/solver_lib.py:46-48
 *           as used by vittles/lr_cov_lib.py:106, 111, 172.
 * vt_arrow_factor: B_g <- chol(B_g) in place, W_g = B_g^{-1} C_g, schur = Gm - sum_g C_g^T W_g
 *   (*info_dev: 0 or 1 + flat index of the first non-positive pivot).
 * vt_arrow_local_solve: x[inds] <- B^{-1} v[inds];  rhs_glob <- -sum_g C_g^T x_g  (Kg x nrhs);
 *   v, x are (d, nrhs) row-major.
 * vt_arrow_back: x[inds] -= W x_glob;  x[ginds] <- x_glob. */
int    vt_arrow_factor(double* blocks_dev, const double* border_dev, const double* corner_dev,
                       double* W_dev, double* spart_dev, double* schur_dev, int G, int M, int Kg,
                       int* info_dev, void* stream);
int    vt_arrow_local_solve(const double* blocks_dev, const double* border_dev, const int64_t* inds_dev,
                            const int64_t* ginds_dev, const double* v_dev, double* x_dev, double* tpart_dev,
                            double* rhs_glob_dev, int G, int M, int Kg, int nrhs, void* stream);
int    vt_arrow_back(const double* W_dev, const int64_t* inds_dev, const int64_t* ginds_dev,
                     const double* xg_dev, double* x_dev, int G, int M, int Kg, int nrhs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITTLES_HIP_H */
