// K0 + K1 + K3 fused for the weighted-GLM objective: ONE pass over X produces
//   H = X^T diag(d) X (+ prior),  g = X^T (w r) (+ prior theta),  r, d per row.
// Replaces autograd.hessian / autograd.grad of the reference
// (vittles/sensitivity_lib.py:354, 381-383) -- see include/vittles_hip.h.
//
// Small-D kernel (D <= 128, NB = ceil(D/16) <= 8 column blocks)
// -----------------------------------------------------------------
// A weighted SYRK whose K dimension is N: every wave owns the WHOLE lower block
// triangle of H in registers (NB(NB+1)/2 accumulators of v_mfma_f64_16x16x4_f64)
// and streams rows of X straight from HBM into the MFMA operand layout:
//   k-step = 4 consecutive rows; lane l = (q = l>>4, c = l&15) loads X[row q][16a + c]
//   for every column block a -- 16 lanes read 128 contiguous bytes of one row.
//   The same register is the A operand (x) and, scaled by d_n, the B operand.
// No LDS staging is needed and X is read exactly once.  The row dot x_n.theta
// (link), the gradient accumulation and the per-row factors ride along in VALU
// slots beside the MFMAs.  Partials are reduced deterministically (no fp64
// atomics): 4 waves -> 1 through LDS, then a fixed-order tree over workgroups.
#include "vt_common.h"

namespace {

struct HessArgs {
    int family;
    const double* X;
    int64_t ldx;
    const double* theta;
    const double* y;
    const double* w;
    const double* offset;
    double scale;
    int64_t N;
    int D;
    double* r_out;
    double* d_out;
    double* part;       // [gridDim.x][L] partial slabs, L = NBT*256 + NB*16
    double* wave_part;  // [gridDim.x][4][NBT*256] per-wave slabs
    double* sink;       // 1024 doubles nobody reads
};

// Loads are UNCONDITIONAL with clamped row / column indices (a per-element
// "load or zero" select makes hipcc branch around every load and drain vmcnt):
// out-of-range rows re-read row N-1 and are neutralised by d_n = 0, out-of-range
// columns re-read column D-1 and land in padded rows/columns that are never stored.
template <int NB>
struct StepData {
    double x[NB];
    double y, w, off;
};

template <int NB, bool HAS_OFF>
__device__ __forceinline__ void load_step(const HessArgs& a, int64_t step, int q, int c, StepData<NB>& d) {
    int64_t n = step * 4 + q;
    n = n < a.N ? n : a.N - 1;
#ifdef VT_ABLATE_LOAD
    n = q;   // dev ablation: always the same 4 rows (L1 hits)
#endif
    const double* xr = a.X + n * a.ldx;
    const int clast = min(16 * (NB - 1) + c, a.D - 1);
#pragma unroll
    for (int b = 0; b < NB - 1; ++b) d.x[b] = xr[16 * b + c];
    d.x[NB - 1] = xr[clast];
    // per-row scalars ride in the same prefetch slot: loading them at the point of
    // use exposes a full HBM miss in every k-step.
    d.y = a.y[n];
    d.w = a.w[n];
    d.off = HAS_OFF ? a.offset[n] : 0.0;
}

// One wave's share of the work: HALF selects which blocks of the triangle it owns
// (blocks with (tri index & 1) == HALF).  fp64 MFMA only reaches its issue rate
// with >= 2 waves per SIMD (one wave alone gets ~45 % of peak on gfx950, measured:
// tools/micro/mfma_f64_rate.hip), and one wave cannot hold the whole triangle in
// the 256 registers it gets at that occupancy -- so waves w and w+4 of a 512-thread
// workgroup stream the SAME rows (second reader hits L1/L2) and split the blocks.
template <int NB, int FAM, bool HAS_OFF, int HALF>
__device__ __forceinline__ void hess_small_body(const HessArgs& a, double* gl) {
    constexpr int NBT = NB * (NB + 1) / 2;
    constexpr int NH = (NBT + 1 - HALF) / 2;   // blocks owned by this half

    const int lane = threadIdx.x & 63;
    const int pair = (threadIdx.x >> 6) & 3;
    const int q = lane >> 4, c = lane & 15;
    const int64_t nsteps = (a.N + 3) >> 2;
    const int64_t W = (int64_t)gridDim.x * 4;
    const int64_t gw = (int64_t)blockIdx.x * 4 + pair;

    // theta lives in LDS (zero padded): 7 broadcast ds_reads per k-step instead of 14
    // long-lived VGPRs -- the register budget at 2 waves/SIMD is what limits NB.
    const double* th = gl + 4 * NB * 16;

    f64x4 acc[NH > 0 ? NH : 1];
#pragma unroll
    for (int i = 0; i < NH; ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
    double gacc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) gacc[b] = 0.0;

    // Software pipeline inside each wave: the link for step s+1 (row dot, 4 shuffle
    // rounds, exp, divide: a long dependent VALU/LDS chain) is issued in the same
    // scheduling region as the MFMAs of step s, so it fills the 64-cycle gaps between
    // MFMAs instead of idling the matrix pipe (measured: with link and MFMAs in
    // separate phases the two waves of a SIMD run in lockstep and the pipe is 46 % busy).
    struct Link { double dw, gwt; };
    auto link = [&](int64_t step, const StepData<NB>& sd) -> Link {
        const double (&x)[NB] = sd.x;
        int64_t n = step * 4 + q;
        bool rowok = (step < nsteps) && (n < a.N);
        double p = 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) p = fma(x[b], th[16 * b + c], p);
        double z = vt_sum16_dpp(p);
        double yv = sd.y, wv = sd.w;
        if (HAS_OFF) z += sd.off;
        double l1, l2;
        vt_link(FAM, z, yv, l1, l2);
        double rr = a.scale * l1;
        Link lk;
        lk.dw = rowok ? a.scale * wv * l2 : 0.0;
        lk.gwt = rowok ? wv * rr : 0.0;
        if constexpr (HALF == 0) {
            // branch-free per-row outputs: lanes with nothing to store write to a sink
            // (any control flow here makes hipcc drain vmcnt(0) every phase).
            bool st = rowok && c == 0;
            double* rp = (st && a.r_out) ? a.r_out + n : a.sink + threadIdx.x;
            double* dp = (st && a.d_out) ? a.d_out + n : a.sink + 512 + threadIdx.x;
            *rp = rr;
            *dp = lk.dw;
        }
        return lk;
    };
    auto mfmas = [&](const StepData<NB>& sd, const Link& lk) {
        const double (&x)[NB] = sd.x;
        if constexpr (HALF == 1) {
#pragma unroll
            for (int b = 0; b < NB; ++b) gacc[b] = fma(lk.gwt, x[b], gacc[b]);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double bj = lk.dw * x[j];
#pragma unroll
            for (int i = j; i < NB; ++i)
                if ((vt_tri(i, j) & 1) == HALF) {
#ifdef VT_ABLATE_MFMA
                    acc[vt_tri(i, j) >> 1][0] += x[i] * bj;   // dev ablation: no MFMA
#else
                    acc[vt_tri(i, j) >> 1] = mfma_f64(x[i], bj, acc[vt_tri(i, j) >> 1]);
#endif
                }
        }
    };

    // 4-deep register ring, two k-steps per scheduling region: the MFMAs of steps (k, k+1)
    // are followed by the links of steps (k+2, k+3) -- two independent dependent chains
    // that the scheduler interleaves -- and the two freed buffers are refilled at once.
    // While this wave walks its link chains the partner wave on the SIMD owns the matrix pipe.
    // One wave can issue an fp64 MFMA only every ~140 cycles (measured, tools/micro), the
    // instruction occupies the pipe for 64: the link's VALU work is placed BETWEEN this wave's
    // MFMAs in program order, where it is free.
    auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < 2 * NH; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);   // 7 VALU
        }
    };
    StepData<NB> x0, x1, x2, x3;
    int64_t s = gw;
    load_step<NB, HAS_OFF>(a, s, q, c, x0);
    load_step<NB, HAS_OFF>(a, s + W, q, c, x1);
    load_step<NB, HAS_OFF>(a, s + 2 * W, q, c, x2);
    load_step<NB, HAS_OFF>(a, s + 3 * W, q, c, x3);
    Link l0 = link(s, x0), l1 = link(s + W, x1), l2, l3;
    for (; s < nsteps; s += 4 * W) {
        mfmas(x0, l0);
        mfmas(x1, l1);
        l2 = link(s + 2 * W, x2);
        l3 = link(s + 3 * W, x3);
        interleave();
        load_step<NB, HAS_OFF>(a, s + 4 * W, q, c, x0);
        load_step<NB, HAS_OFF>(a, s + 5 * W, q, c, x1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(x2, l2);
        mfmas(x3, l3);
        l0 = link(s + 4 * W, x0);
        l1 = link(s + 5 * W, x1);
        interleave();
        load_step<NB, HAS_OFF>(a, s + 6 * W, q, c, x2);
        load_step<NB, HAS_OFF>(a, s + 7 * W, q, c, x3);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- gradient: sum the 4 row groups of the wave (half 1 only) ---------------
    if constexpr (HALF == 1) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            double v = gacc[b];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (q == 0) gl[pair * (NB * 16) + 16 * b + c] = v;
        }
    }

    // ---- H accumulators ------------------------------------------------------------
    // They leave the register file only through GLOBAL stores: a ds_write (or any VALU
    // op) on them makes hipcc keep all of them in VGPRs for the whole kernel and spill
    // everything else.  Each wave parks its blocks in the pair's slab in the workspace
    // (L2-resident); the workgroup then folds its 4 slabs into one.
    double* mine = a.wave_part + ((int64_t)blockIdx.x * 4 + pair) * (NBT * 256);
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
            if ((vt_tri(i, j) & 1) == HALF) {
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[(vt_tri(i, j) * 4 + r) * 64 + lane] = acc[vt_tri(i, j) >> 1][r];
            }
}

template <int NB, int FAM, bool HAS_OFF>
__global__ __launch_bounds__(512) void k_glm_hess_small(HessArgs a) {
    constexpr int NBT = NB * (NB + 1) / 2;
    constexpr int L = NBT * 256 + NB * 16;
    __shared__ double gl[5 * NB * 16];   // 4 gradient slots + theta
    for (int t = threadIdx.x; t < NB * 16; t += 512) gl[4 * NB * 16 + t] = t < a.D ? a.theta[t] : 0.0;
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0)
        hess_small_body<NB, FAM, HAS_OFF, 0>(a, gl);
    else
        hess_small_body<NB, FAM, HAS_OFF, 1>(a, gl);
    __syncthreads();
    const double* wslab = a.wave_part + (int64_t)blockIdx.x * 4 * (NBT * 256);
    double* out = a.part + (int64_t)blockIdx.x * L;
    for (int e = threadIdx.x; e < NBT * 256; e += 512)
        out[e] = (wslab[e] + wslab[NBT * 256 + e]) + (wslab[2 * NBT * 256 + e] + wslab[3 * NBT * 256 + e]);
    for (int t = threadIdx.x; t < NB * 16; t += 512)
        out[NBT * 256 + t] = (gl[t] + gl[NB * 16 + t]) + (gl[2 * NB * 16 + t] + gl[3 * NB * 16 + t]);
}

// Sum `per` consecutive partial slabs into one: out[p][e] = sum_{k<per} in[p*per + k][e].
__global__ void k_reduce_slabs(const double* in, double* out, int L, int P_in, int per) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    int p = blockIdx.y;
    if (e >= L) return;
    double s = 0.0;
    int k0 = p * per, k1 = min(P_in, k0 + per);
    for (int k = k0; k < k1; ++k) s += in[(int64_t)k * L + e];
    out[(int64_t)p * L + e] = s;
}

// Final: sum the remaining slabs, undo the accumulator layout, mirror the
// triangle, add the prior.  One thread per accumulator element / gradient entry.
__global__ void k_hess_finalize(const double* in, int L, int P, int NB, int D,
                                const double* prior, const double* theta,
                                double* H, int ldh, double* g) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L) return;
    double s = 0.0;
    for (int k = 0; k < P; ++k) s += in[(int64_t)k * L + e];
    int NBT = NB * (NB + 1) / 2;
    if (e < NBT * 256) {
        int blk = e >> 8, r = (e >> 6) & 3, lane = e & 63;
        int ba = 0;
        while ((ba + 1) * (ba + 2) / 2 <= blk) ++ba;
        int bb = blk - ba * (ba + 1) / 2;
        int i = 16 * ba + (lane >> 4) + 4 * r;
        int j = 16 * bb + (lane & 15);
        if (i < D && j < D) {
            if (i == j && prior) s += prior[i];
            if (ba != bb || i >= j) {
                H[(int64_t)i * ldh + j] = s;
                H[(int64_t)j * ldh + i] = s;
            }
        }
    } else if (g) {
        int t = e - NBT * 256;
        if (t < D) g[t] = s + (prior ? prior[t] * theta[t] : 0.0);
    }
}

int g_num_cu = 0;
int num_cu() {
    if (g_num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        g_num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return g_num_cu;
}

constexpr int kStage2 = 16;

inline int slab_len(int NB) { return NB * (NB + 1) / 2 * 256 + NB * 16; }

inline int hess_grid(int64_t N) {
    int64_t nsteps = (N + 3) / 4;
    int64_t g = (nsteps + 15) / 16;  // >= 4 k-steps per wave before adding workgroups
    int cu = num_cu();
    if (g > cu) g = cu;
    if (g < 1) g = 1;
    return (int)g;
}

template <int NB, int FAM>
void launch_hess_small2(const HessArgs& a, int grid, hipStream_t st) {
    if (a.offset)
        hipLaunchKernelGGL((k_glm_hess_small<NB, FAM, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_glm_hess_small<NB, FAM, false>), dim3(grid), dim3(512), 0, st, a);
}

template <int NB>
void launch_hess_small(const HessArgs& a, int grid, hipStream_t st) {
    switch (a.family) {
        case VT_FAMILY_LOGISTIC: launch_hess_small2<NB, VT_FAMILY_LOGISTIC>(a, grid, st); break;
        case VT_FAMILY_POISSON: launch_hess_small2<NB, VT_FAMILY_POISSON>(a, grid, st); break;
        default: launch_hess_small2<NB, VT_FAMILY_GAUSSIAN>(a, grid, st); break;
    }
}

}  // namespace

extern "C" size_t vt_rows_workspace_bytes(int D);
extern "C" int vt_glm_grad(int family, const double* X, int64_t ldx, const double* theta, const double* y,
                           const double* w, const double* offset, double scale, const double* prior,
                           int64_t N, int D, double* g_out, double* r_out, double* d_out, void* ws,
                           size_t ws_bytes, void* stream);

// D > 128: link + gradient by the row-streaming kernel (one pass), then the tiled SYRK.
static size_t large_ws_rows(int D) { return (vt_rows_workspace_bytes(D) + 255) / 256 * 256; }

extern "C" size_t vt_glm_hessian_workspace_bytes(int64_t N, int D) {
    if (D < 1) return 0;
    if (D > VT_SMALL_D_MAX)
        return large_ws_rows(D) + (size_t)(N + D + 64) * sizeof(double) + vt_large_syrk_ws_bytes(D);
    int NB = (D + 15) / 16;
    return ((size_t)(num_cu() + kStage2) * slab_len(NB) + (size_t)num_cu() * 4 * (NB * (NB + 1) / 2) * 256 + 1024) *
           sizeof(double);
}

extern "C" int vt_glm_hessian(int family, const double* X, int64_t ldx, const double* theta,
                              const double* y, const double* w, const double* offset, double scale,
                              const double* prior, int64_t N, int D, double* H, int ldh, double* g,
                              double* r, double* d, void* ws, size_t ws_bytes, void* stream) {
    if (!X || !theta || !y || !w || !H || N < 0 || D < 1 || ldx < D || ldh < D) return VT_ERR_BAD_ARG;
    if (family < 0 || family > 2) return VT_ERR_BAD_ARG;
    if (!ws || ws_bytes < vt_glm_hessian_workspace_bytes(N, D)) return VT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (D > VT_SMALL_D_MAX) {
        char* base = (char*)ws;
        double* dtmp = (double*)(base + large_ws_rows(D));
        double* gtmp = dtmp + N;
        size_t off = large_ws_rows(D) + (size_t)(N + D) * sizeof(double);
        void* syrk_ws = (void*)(base + (off + 255) / 256 * 256);
        double* dd = d ? d : dtmp;
        int rc = vt_glm_grad(family, X, ldx, theta, y, w, offset, scale, prior, N, D, g ? g : gtmp, r, dd, ws,
                             large_ws_rows(D), stream);
        if (rc != VT_OK) return rc;
        return vt_large_syrk(X, ldx, dd, N, D, prior, H, ldh, syrk_ws, st);
    }
    const int NB = (D + 15) / 16;
    const int L = slab_len(NB);
    const int grid = hess_grid(N);
    double* part = (double*)ws;
    double* part2 = part + (size_t)num_cu() * L;
    double* wave_part = part2 + (size_t)kStage2 * L;
    double* sink = wave_part + (size_t)num_cu() * 4 * (NB * (NB + 1) / 2) * 256;

    HessArgs a{family, X, ldx, theta, y, w, offset, scale, N, D, r, d, part, wave_part, sink};
    switch (NB) {
        case 1: launch_hess_small<1>(a, grid, st); break;
        case 2: launch_hess_small<2>(a, grid, st); break;
        case 3: launch_hess_small<3>(a, grid, st); break;
        case 4: launch_hess_small<4>(a, grid, st); break;
        case 5: launch_hess_small<5>(a, grid, st); break;
        case 6: launch_hess_small<6>(a, grid, st); break;
        case 7: launch_hess_small<7>(a, grid, st); break;
        case 8: launch_hess_small<8>(a, grid, st); break;
        default: return VT_ERR_UNSUPPORTED;
    }
    const double* fin = part;
    int P = grid;
    if (P > kStage2) {
        int per = (P + kStage2 - 1) / kStage2;
        int P2 = (P + per - 1) / per;
        hipLaunchKernelGGL(k_reduce_slabs, dim3((L + 255) / 256, P2), dim3(256), 0, st, part, part2, L, P, per);
        fin = part2;
        P = P2;
    }
    hipLaunchKernelGGL(k_hess_finalize, dim3((L + 255) / 256), dim3(256), 0, st, fin, L, P, NB, D, prior, theta,
                       H, ldh, g);
    return vt_launch_status();
}
