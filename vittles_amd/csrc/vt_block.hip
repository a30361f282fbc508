// K7 / K8: hierarchical mean-field Gaussian VI objective (gradient / HVP / value), gather of M
// coloured HVPs into the block Hessian in the reference's COO order, and the batched
// block-arrowhead solver (small Cholesky per group + dense Schur complement on the globals).
//
// Replaces: vittles/sparse_hessian_lib.py:62-108 (M HVPs + the O(G*M) Python list appends),
//           :110-163 (global rows), vittles/solver_lib.py:46-48 (SuperLU on the COO matrix) and
//           vittles/lr_cov_lib.py:106-119,172 for block-structured objectives.
// These are HBM-bound and tiny next to the GLM path (G = 1e4 groups: ~10 MB); the design goal
// is one launch per logical step, deterministic reductions, no host loops over groups.
#include "vt_common.h"

namespace {

// Model (oracle/vi.py): y_gi ~ N(x_gi.theta_g, 1/tau_y), theta_g ~ N(mu, I/tau_th), mu ~ N(0, I/tau_mu);
// q(theta_g) = N(m_g, diag v_g), q(mu) = N(m_mu, diag v_mu).
// eta layout: [m_0 v_0 m_1 v_1 ... m_{G-1} v_{G-1} m_mu v_mu], each piece k long.
struct ViArgs {
    const double* X;   // (G*ng, k) row-major
    const double* y;   // (G*ng)
    const double* w;   // (G*ng)
    const double* eta; // (2k(G+1))
    const double* u;   // HVP direction (2k(G+1)) or NULL
    double tau_y, tau_th, tau_mu;
    int G, ng, k;
    double* out;       // (2k(G+1)) gradient / HVP
    double* gpart;     // (G, k+1): per-group contributions to the global rows / the value
    int global_terms;  // 1: the whole objective; 0: a shard of the groups that does NOT own the terms of the
                       //    global rows / the value that are not sums over groups (mu's prior and entropy)
};

enum { VI_HVP = 0, VI_GRAD = 1, VI_VALUE = 2 };

// one thread per (group g, coordinate j)
template <int MODE>
__global__ void k_vi_local(ViArgs a) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int g = (int)(t / a.k), j = (int)(t - (int64_t)g * a.k);
    if (g >= a.G) return;
    const int k = a.k, ng = a.ng;
    const double* Xg = a.X + (int64_t)g * ng * k;
    const double* wg = a.w + (int64_t)g * ng;
    const double* mg = a.eta + (int64_t)g * 2 * k;
    const double* vg = mg + k;
    const double* mmu = a.eta + (int64_t)a.G * 2 * k;
    if (MODE == VI_HVP) {
        const double* um = a.u + (int64_t)g * 2 * k;
        const double* uv = um + k;
        const double* ummu = a.u + (int64_t)a.G * 2 * k;
        double s = 0.0;
        for (int i = 0; i < ng; ++i) {
            double ti = 0.0;
            for (int l = 0; l < k; ++l) ti = fma(Xg[i * k + l], um[l], ti);
            s = fma(wg[i] * ti, Xg[i * k + j], s);
        }
        a.out[(int64_t)g * 2 * k + j] = a.tau_y * s + a.tau_th * (um[j] - ummu[j]);
        a.out[(int64_t)g * 2 * k + k + j] = 0.5 / (vg[j] * vg[j]) * uv[j];
        a.gpart[(int64_t)g * (k + 1) + j] = um[j];
    } else if (MODE == VI_GRAD) {
        const double* yg = a.y + (int64_t)g * ng;
        double s = 0.0, s2 = 0.0;
        for (int i = 0; i < ng; ++i) {
            double zi = 0.0;
            for (int l = 0; l < k; ++l) zi = fma(Xg[i * k + l], mg[l], zi);
            double xij = Xg[i * k + j];
            s = fma(wg[i] * (yg[i] - zi), xij, s);
            s2 = fma(wg[i] * xij, xij, s2);
        }
        a.out[(int64_t)g * 2 * k + j] = -a.tau_y * s + a.tau_th * (mg[j] - mmu[j]);
        a.out[(int64_t)g * 2 * k + k + j] = 0.5 * a.tau_y * s2 + 0.5 * a.tau_th - 0.5 / vg[j];
        a.gpart[(int64_t)g * (k + 1) + j] = mg[j] - mmu[j];
    } else {
        // value: thread j == 0 of the group does the likelihood; every j its own hier/entropy part
        const double* yg = a.y + (int64_t)g * ng;
        double val = 0.0;
        if (j == 0) {
            for (int i = 0; i < ng; ++i) {
                double zi = 0.0, xv = 0.0;
                for (int l = 0; l < k; ++l) {
                    double x = Xg[i * k + l];
                    zi = fma(x, mg[l], zi);
                    xv = fma(x * x, vg[l], xv);
                }
                double r = yg[i] - zi;
                val += 0.5 * a.tau_y * wg[i] * (r * r + xv);
            }
        }
        double dm = mg[j] - mmu[j];
        val += 0.5 * a.tau_th * (dm * dm + vg[j]) - 0.5 * log(vg[j]);
        a.gpart[(int64_t)g * (k + 1) + j] = val;
    }
}

// Deterministic column sums of gpart over groups + the global rows.  One workgroup.
template <int MODE>
__global__ __launch_bounds__(1024) void k_vi_global(ViArgs a, double* value_out) {
    __shared__ double red[1024];
    const int k = a.k, G = a.G, tid = threadIdx.x;
    const double* mmu = a.eta + (int64_t)G * 2 * k;
    const double* vmu = mmu + k;
    double total_val = 0.0;
    for (int j = 0; j < k; ++j) {
        double s = 0.0;
        for (int g = tid; g < G; g += 1024) s += a.gpart[(int64_t)g * (k + 1) + j];
        red[tid] = s;
        __syncthreads();
        for (int off = 512; off > 0; off >>= 1) {
            if (tid < off) red[tid] += red[tid + off];
            __syncthreads();
        }
        double tot = red[0];
        __syncthreads();
        if (tid == 0) {
            int64_t base = (int64_t)G * 2 * k;
            // a shard (global_terms == 0) contributes only what is a sum over its groups; the shards' global
            // rows / values then ADD UP to the whole objective's
            const double once = a.global_terms ? 1.0 : 0.0;
            const double tmu = once * a.tau_mu;
            if (MODE == VI_HVP) {
                const double* ummu = a.u + base;
                const double* uvmu = ummu + k;
                a.out[base + j] = -a.tau_th * tot + (G * a.tau_th + tmu) * ummu[j];
                a.out[base + k + j] = once * (0.5 / (vmu[j] * vmu[j]) * uvmu[j]);
            } else if (MODE == VI_GRAD) {
                a.out[base + j] = -a.tau_th * tot + tmu * mmu[j];
                a.out[base + k + j] = 0.5 * (G * a.tau_th + tmu) - once * (0.5 / vmu[j]);
            } else {
                total_val += tot + 0.5 * a.tau_th * G * vmu[j] + 0.5 * tmu * (mmu[j] * mmu[j] + vmu[j]) -
                             once * (0.5 * log(vmu[j]));
            }
        }
    }
    if (MODE == VI_VALUE && tid == 0) *value_out = total_val;
}

template <int MODE>
int run_vi(const ViArgs& a, double* value_out, hipStream_t st) {
    int64_t nt = (int64_t)a.G * a.k;
    hipLaunchKernelGGL(k_vi_local<MODE>, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_vi_global<MODE>, dim3(1), dim3(1024), 0, st, a, value_out);
    return vt_launch_status();
}

// ---- coloured direction vectors and gathers ---------------------------------------------------
// v[inds[b, ib]] = 1 for every block b (sparse_hessian_lib.py:65-66), everything else 0.
__global__ void k_colour_vec(const int64_t* inds, int G, int M, int ib, double* v) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < G) v[inds[(int64_t)b * M + ib]] = 1.0;
}

// After HVP number ib: entries e = (ib*G + b)*M + r of the COO triplets, reference order
// (sparse_hessian_lib.py:93-102), and the dense block image blocks[b][r][ib].
__global__ void k_block_gather(const double* hv, const int64_t* inds, int G, int M, int ib, double* vals,
                               int64_t* rows, int64_t* cols, double* blocks) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)G * M) return;
    int b = (int)(t / M), r = (int)(t - (int64_t)b * M);
    int64_t idx = inds[(int64_t)b * M + r];
    double v = hv[idx];
    int64_t e = ((int64_t)ib * G + b) * M + r;
    vals[e] = v;
    rows[e] = idx;
    cols[e] = inds[(int64_t)b * M + ib];
    blocks[((int64_t)b * M + r) * M + ib] = v;
}

// Global index number jg (value ig): COO entries in the reference order
// (sparse_hessian_lib.py:137-153) and the dense border / corner images.
__global__ void k_global_gather(const double* hrow, const int64_t* local_inds, int64_t nloc,
                                const int64_t* ginds, int Kg, int jg, double* vals, int64_t* rows,
                                int64_t* cols, double* border, double* corner) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = 2 * nloc + 2 * Kg;
    const int64_t base = (int64_t)jg * per;
    const int64_t ig = ginds[jg];
    if (t < nloc) {
        int64_t il = local_inds[t];
        double v = hrow[il];
        vals[base + 2 * t] = v;      rows[base + 2 * t] = il;      cols[base + 2 * t] = ig;
        vals[base + 2 * t + 1] = v;  rows[base + 2 * t + 1] = ig;  cols[base + 2 * t + 1] = il;
        border[t * Kg + jg] = v;
    } else if (t < nloc + Kg) {
        int j2 = (int)(t - nloc);
        int64_t ig2 = ginds[j2];
        double v = hrow[ig2];
        int64_t e = base + 2 * nloc + 2 * j2;
        vals[e] = 0.5 * v;      rows[e] = ig2;     cols[e] = ig;
        vals[e + 1] = 0.5 * v;  rows[e + 1] = ig;  cols[e + 1] = ig2;
        corner[(int64_t)j2 * Kg + jg] = v;
    }
}

// ---- batched block-arrowhead solver -------------------------------------------------------------
// H = [[blockdiag(B_g), C], [C^T, Gm]].  One thread per group; M <= 16, Kg <= 16.
// factor: B_g = L_g L_g^T in place (lower), W_g = B_g^{-1} C_g, spart[g] = C_g^T W_g (Kg x Kg).
__global__ void k_arrow_factor(double* blocks, const double* border, double* W, double* spart, int G, int M,
                               int Kg, int* info) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    double* B = blocks + (int64_t)g * M * M;
    for (int j = 0; j < M; ++j) {
        double d = B[j * M + j];
        for (int l = 0; l < j; ++l) d -= B[j * M + l] * B[j * M + l];
        if (!(d > 0.0)) {
            atomicCAS(info, 0, g * M + j + 1);
            d = 1.0;
        }
        d = sqrt(d);
        B[j * M + j] = d;
        for (int i = j + 1; i < M; ++i) {
            double s = B[i * M + j];
            for (int l = 0; l < j; ++l) s -= B[i * M + l] * B[j * M + l];
            B[i * M + j] = s / d;
        }
    }
    const double* C = border + (int64_t)g * M * Kg;
    double* Wg = W + (int64_t)g * M * Kg;
    for (int c = 0; c < Kg; ++c) {
        for (int i = 0; i < M; ++i) {
            double s = C[i * Kg + c];
            for (int l = 0; l < i; ++l) s -= B[i * M + l] * Wg[l * Kg + c];
            Wg[i * Kg + c] = s / B[i * M + i];
        }
        for (int i = M - 1; i >= 0; --i) {
            double s = Wg[i * Kg + c];
            for (int l = i + 1; l < M; ++l) s -= B[l * M + i] * Wg[l * Kg + c];
            Wg[i * Kg + c] = s / B[i * M + i];
        }
    }
    double* sp = spart + (int64_t)g * Kg * Kg;
    for (int a = 0; a < Kg; ++a)
        for (int c = 0; c < Kg; ++c) {
            double s = 0.0;
            for (int i = 0; i < M; ++i) s = fma(C[i * Kg + a], Wg[i * Kg + c], s);
            sp[a * Kg + c] = s;
        }
}

// out[e] = base[e] - sum_g part[g][e]   (e < L), one workgroup, fixed order.
__global__ __launch_bounds__(1024) void k_sub_group_sum(const double* base, const double* part, int G, int L,
                                                        double* out) {
    __shared__ double red[1024];
    const int tid = threadIdx.x;
    for (int e = 0; e < L; ++e) {
        double s = 0.0;
        for (int g = tid; g < G; g += 1024) s += part[(int64_t)g * L + e];
        red[tid] = s;
        __syncthreads();
        for (int off = 512; off > 0; off >>= 1) {
            if (tid < off) red[tid] += red[tid + off];
            __syncthreads();
        }
        if (tid == 0) out[e] = (base ? base[e] : 0.0) - red[0];
        __syncthreads();
    }
}

// local solve for nrhs right-hand sides stored (d, nrhs) row-major, local rows addressed through
// inds: y_g = B_g^{-1} v_g (written into x), tpart[g] = C_g^T y_g  (Kg x nrhs).
__global__ void k_arrow_local_solve(const double* blocks, const double* border, const int64_t* inds,
                                    const double* v, double* x, double* tpart, int G, int M, int Kg,
                                    int nrhs) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const double* B = blocks + (int64_t)g * M * M;
    const double* C = border + (int64_t)g * M * Kg;
    const int64_t* id = inds + (int64_t)g * M;
    for (int c = 0; c < nrhs; ++c) {
        for (int i = 0; i < M; ++i) {
            double s = v[id[i] * nrhs + c];
            for (int l = 0; l < i; ++l) s -= B[i * M + l] * x[id[l] * nrhs + c];
            x[id[i] * nrhs + c] = s / B[i * M + i];
        }
        for (int i = M - 1; i >= 0; --i) {
            double s = x[id[i] * nrhs + c];
            for (int l = i + 1; l < M; ++l) s -= B[l * M + i] * x[id[l] * nrhs + c];
            x[id[i] * nrhs + c] = s / B[i * M + i];
        }
        for (int a = 0; a < Kg; ++a) {
            double s = 0.0;
            for (int i = 0; i < M; ++i) s = fma(C[i * Kg + a], x[id[i] * nrhs + c], s);
            tpart[((int64_t)g * Kg + a) * nrhs + c] = s;
        }
    }
}

// x_g -= W_g x_glob;  x[ginds] = x_glob
__global__ void k_arrow_back(const double* W, const int64_t* inds, const int64_t* ginds, const double* xg,
                             double* x, int G, int M, int Kg, int nrhs) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < (int64_t)G * M) {
        int g = (int)(t / M), i = (int)(t - (int64_t)g * M);
        const double* Wg = W + ((int64_t)g * M + i) * Kg;
        int64_t row = inds[t];
        for (int c = 0; c < nrhs; ++c) {
            double s = x[row * nrhs + c];
            for (int a = 0; a < Kg; ++a) s -= Wg[a] * xg[a * nrhs + c];
            x[row * nrhs + c] = s;
        }
    } else if (t < (int64_t)G * M + Kg) {
        int a = (int)(t - (int64_t)G * M);
        for (int c = 0; c < nrhs; ++c) x[ginds[a] * nrhs + c] = xg[a * nrhs + c];
    }
}

}  // namespace

#define VI_ARGS ViArgs a{X, y, w, eta, u, tau_y, tau_th, tau_mu, G, ng, k, out, gpart, global_terms}

extern "C" size_t vt_vi_workspace_bytes(int G, int k) { return (size_t)G * (k + 1) * sizeof(double); }

extern "C" int vt_vi_hvp(const double* X, const double* w, const double* eta, const double* u, double tau_y,
                         double tau_th, double tau_mu, int G, int ng, int k, int global_terms, double* out, void* ws,
                         size_t ws_bytes, void* stream) {
    if (!X || !w || !eta || !u || !out || G < 1 || ng < 1 || k < 1) return VT_ERR_BAD_ARG;
    if (!ws || ws_bytes < vt_vi_workspace_bytes(G, k)) return VT_ERR_WORKSPACE;
    const double* y = nullptr;
    double* gpart = (double*)ws;
    VI_ARGS;
    return run_vi<VI_HVP>(a, nullptr, (hipStream_t)stream);
}

extern "C" int vt_vi_grad(const double* X, const double* y, const double* w, const double* eta, double tau_y,
                          double tau_th, double tau_mu, int G, int ng, int k, int global_terms, double* out, void* ws,
                          size_t ws_bytes, void* stream) {
    if (!X || !y || !w || !eta || !out || G < 1 || ng < 1 || k < 1) return VT_ERR_BAD_ARG;
    if (!ws || ws_bytes < vt_vi_workspace_bytes(G, k)) return VT_ERR_WORKSPACE;
    const double* u = nullptr;
    double* gpart = (double*)ws;
    VI_ARGS;
    return run_vi<VI_GRAD>(a, nullptr, (hipStream_t)stream);
}

extern "C" int vt_vi_value(const double* X, const double* y, const double* w, const double* eta, double tau_y,
                           double tau_th, double tau_mu, int G, int ng, int k, int global_terms, double* value_out,
                           void* ws,
                           size_t ws_bytes, void* stream) {
    if (!X || !y || !w || !eta || !value_out || G < 1 || ng < 1 || k < 1) return VT_ERR_BAD_ARG;
    if (!ws || ws_bytes < vt_vi_workspace_bytes(G, k)) return VT_ERR_WORKSPACE;
    const double* u = nullptr;
    double* out = nullptr;
    double* gpart = (double*)ws;
    VI_ARGS;
    return run_vi<VI_VALUE>(a, value_out, (hipStream_t)stream);
}

extern "C" int vt_block_colour_vec(const int64_t* inds, int G, int M, int ib, double* v, int64_t d,
                                   void* stream) {
    if (!inds || !v || G < 1 || M < 1 || ib < 0 || ib >= M) return VT_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(v, 0, (size_t)d * sizeof(double), st) != hipSuccess) return VT_ERR_HIP;
    hipLaunchKernelGGL(k_colour_vec, dim3((G + 255) / 256), dim3(256), 0, st, inds, G, M, ib, v);
    return vt_launch_status();
}

extern "C" int vt_block_gather(const double* hv, const int64_t* inds, int G, int M, int ib, double* vals,
                               int64_t* rows, int64_t* cols, double* blocks, void* stream) {
    if (!hv || !inds || !vals || !rows || !cols || !blocks || G < 1 || M < 1) return VT_ERR_BAD_ARG;
    int64_t nt = (int64_t)G * M;
    hipLaunchKernelGGL(k_block_gather, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, hv,
                       inds, G, M, ib, vals, rows, cols, blocks);
    return vt_launch_status();
}

extern "C" int vt_global_gather(const double* hrow, const int64_t* local_inds, int64_t nloc,
                                const int64_t* ginds, int Kg, int jg, double* vals, int64_t* rows,
                                int64_t* cols, double* border, double* corner, void* stream) {
    if (!hrow || !ginds || !vals || !rows || !cols || !border || !corner || Kg < 1) return VT_ERR_BAD_ARG;
    int64_t nt = nloc + Kg;
    hipLaunchKernelGGL(k_global_gather, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       hrow, local_inds, nloc, ginds, Kg, jg, vals, rows, cols, border, corner);
    return vt_launch_status();
}

extern "C" int vt_arrow_factor(double* blocks, const double* border, const double* corner, double* W,
                               double* spart, double* schur, int G, int M, int Kg, int* info, void* stream) {
    if (!blocks || !info || G < 1 || M < 1 || M > 64 || Kg < 0 || Kg > 64) return VT_ERR_BAD_ARG;
    if (Kg > 0 && (!border || !corner || !W || !spart || !schur)) return VT_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(info, 0, sizeof(int), st) != hipSuccess) return VT_ERR_HIP;
    hipLaunchKernelGGL(k_arrow_factor, dim3((G + 63) / 64), dim3(64), 0, st, blocks, border, W, spart, G, M, Kg,
                       info);
    if (Kg > 0)
        hipLaunchKernelGGL(k_sub_group_sum, dim3(1), dim3(1024), 0, st, corner, (const double*)spart, G, Kg * Kg,
                           schur);
    return vt_launch_status();
}

extern "C" int vt_arrow_local_solve(const double* blocks, const double* border, const int64_t* inds,
                                    const int64_t* ginds, const double* v, double* x, double* tpart,
                                    double* rhs_glob, int G, int M, int Kg, int nrhs, void* stream) {
    if (!blocks || !inds || !v || !x || G < 1 || M < 1 || nrhs < 1) return VT_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_arrow_local_solve, dim3((G + 63) / 64), dim3(64), 0, st, blocks, border, inds, v, x,
                       tpart, G, M, Kg, nrhs);
    if (Kg > 0)   // rhs_glob = -(sum_g C_g^T y_g); the caller adds v[ginds]
        hipLaunchKernelGGL(k_sub_group_sum, dim3(1), dim3(1024), 0, st, (const double*)nullptr,
                           (const double*)tpart, G, Kg * nrhs, rhs_glob);
    return vt_launch_status();
}

extern "C" int vt_arrow_back(const double* W, const int64_t* inds, const int64_t* ginds, const double* xg,
                             double* x, int G, int M, int Kg, int nrhs, void* stream) {
    if (!inds || !x || G < 1 || M < 1 || nrhs < 1) return VT_ERR_BAD_ARG;
    if (Kg == 0) return VT_OK;
    int64_t nt = (int64_t)G * M + Kg;
    hipLaunchKernelGGL(k_arrow_back, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                       inds, ginds, xg, x, G, M, Kg, nrhs);
    return vt_launch_status();
}
