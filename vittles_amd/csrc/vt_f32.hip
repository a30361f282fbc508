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
#include <type_traits>
#include <utility>
#ifndef VT_ABL32
#define VT_ABL32 0   // dev-only compile-time ablation mask (tools/build_dev32.sh): 1 no MFMA, 2 no link, 4 no loads in the loop
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// v_mfma_f32_16x16x4_f32: A lane l = A[i = l & 15][k = l >> 4], B lane l = B[k = l >> 4][j = l & 15],
// C/D lane l register r = D[i = 4 (l >> 4) + r][j = l & 15]   (NOT the f64 map: row (l >> 4) + 4 r there).
__device__ __forceinline__ f32x4 mfma_f32(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------ Hessian
struct Hess32Args {
    int family;
    const float* X;
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
    double* part;   // [gridDim.x][L] partial slabs, L = NBT*256 + NB*16
    double* sink;   // 1024 doubles nobody reads
};

constexpr int kChunkSteps = 1024;   // k-steps a wave accumulates in f32 before folding into fp64 (multiple of 4)

// A GROUP is 4 consecutive k-steps of a wave = 16 rows.  x: lane (q, c) holds X[row q of step s][16 b + c];
// y, w, off: the per-row scalars of row (step c & 3, q) -- the row this lane LINKS (below).
template <int NB>
struct Grp32 {
    float x[4][NB];
    double y, w, off;
};

// v + (v rotated right by N lanes inside each row of 16 lanes): one v_add_f32 with a DPP operand
template <int N>
__device__ __forceinline__ float ror16_add_f32(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, true));
}

// v of lane S of each row of 16 lanes, in every lane of that row (DPP row_newbcast: one VALU move)
template <int S>
__device__ __forceinline__ float bcast16(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + S, 0xf, 0xf, true));
}

// Every wave owns the WHOLE block triangle in f32 accumulators (NBT * 4 <= 144 registers) and streams its own
// k-steps (4 rows each) straight from HBM into the MFMA operand layout -- the register-streaming design of
// k_glm_hess_small without the split of the triangle over two waves, which fp64 needs and f32 does not.
// WAVES = 8 (two per SIMD, 256 registers each) up to NB = 5; 4 (one per SIMD, accumulators in the AGPR half of
// its 512 registers) above: an f32 MFMA wave with independent accumulators reaches the issue rate alone.
//
// VALU instructions do NOT overlap v_mfma_f32_16x16x4_f32 on gfx950 -- neither from the same wave nor from another
// wave of the SIMD; the times add (tools/micro/mfma_f32_valu_overlap.hip: "both" = MFMA + VALU in every
// arrangement; the instruction runs at the f32 VECTOR rate) -- so the kernel is MFMA cycles (32 each) + VALU
// cycles (3.5 per f32 op, 5.5 .. 8 per fp64 op), and the VALU work per k-step is cut to the bone:
//   * the row dot x_n . theta is taken in f32 against theta rounded to float (7 FMAs + 4 DPP adds per k-step; in
//     fp64 it was 7 conversions + 7 FMAs + 12 DPP instructions at twice the cycles each).  z, and with it r and d,
//     are therefore f32-accurate (~1e-6): inside the mode's 1e-4, and the optimiser's gradient comes from
//     vt_glm_grad_f32, which sums in fp64;
//   * the link (exp, reciprocal, ... ~60 fp64 instructions) is evaluated ONCE per row instead of once per lane
//     of the row's 16-lane group: the row dots of a group's 4 k-steps leave z of row (s, q) in all 16 lanes of
//     lane row q; lane (q, c) keeps the one of step c & 3, links it, and the MFMAs of step s fetch d_n and
//     w_n r_n from lane (q, s) by DPP row broadcast.
template <int NB, int FAM, bool HAS_OFF, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_glm_hess_f32(Hess32Args a) {
    constexpr int NT = WAVES * 64;
    constexpr int NBT = NB * (NB + 1) / 2;
    constexpr int L = NBT * 256 + NB * 16;
    extern __shared__ double lds[];   // fp64 slab [NBT*256] | gradient [NB*16] | theta as float [NB*16]
    double* slab = lds;
    double* gsum = lds + NBT * 256;
    float* th = (float*)(lds + L);
    for (int t = threadIdx.x; t < L; t += NT) lds[t] = 0.0;
    for (int t = threadIdx.x; t < NB * 16; t += NT) th[t] = t < a.D ? (float)a.theta[t] : 0.f;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15, cs = lane & 3;
    const int clast = min(16 * (NB - 1) + c, a.D - 1);
    const int64_t nsteps = (a.N + 3) >> 2;
    const int64_t W = (int64_t)gridDim.x * WAVES;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t per_wave = (nsteps + W - 1) / W;                      // same for every wave of the grid
    const int64_t nchunks = (per_wave + kChunkSteps - 1) / kChunkSteps;

    f32x4 acc[NBT];
    float gacc[NB];

    for (int64_t ch = 0; ch < nchunks; ++ch) {
        const int64_t t0 = ch * kChunkSteps;
        const int64_t t1 = min(t0 + (int64_t)kChunkSteps, per_wave);
#pragma unroll
        for (int i = 0; i < NBT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NB; ++b) gacc[b] = 0.f;

        // The wave's t-th k-step is global k-step gw + W t.  Unconditional loads with clamped indices (see
        // vt_hessian.hip): rows past N re-read row N-1 and get d_n = 0, columns past D re-read column D-1 and land
        // in padding that is never stored.
        auto load_group = [&](int64_t tg, Grp32<NB>& d) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                int64_t n = (gw + W * (tg + s)) * 4 + q;
                n = n < a.N ? n : a.N - 1;
                const float* xr = a.X + n * a.ldx;
#pragma unroll
                for (int b = 0; b < NB - 1; ++b) d.x[s][b] = xr[16 * b + c];
                d.x[s][NB - 1] = xr[clast];
            }
            int64_t nl = (gw + W * (tg + cs)) * 4 + q;
            nl = nl < a.N ? nl : a.N - 1;
            d.y = a.y[nl];
            d.w = a.w[nl];
            d.off = HAS_OFF ? a.offset[nl] : 0.0;
        };
        struct Link { float dw, gwt; };   // of row (step c & 3, q)
        auto link_group = [&](int64_t tg, const Grp32<NB>& d) -> Link {
            float thv[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) thv[b] = th[16 * b + c];
            float z[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float p = 0.f;
#pragma unroll
                for (int b = 0; b < NB; ++b) p = fmaf(d.x[s][b], thv[b], p);
                p = ror16_add_f32<8>(p);
                p = ror16_add_f32<4>(p);
                p = ror16_add_f32<2>(p);
                z[s] = ror16_add_f32<1>(p);
            }
            double zz = (double)(cs == 0 ? z[0] : (cs == 1 ? z[1] : (cs == 2 ? z[2] : z[3])));
            const int64_t ts = tg + cs;
            const int64_t n = (gw + W * ts) * 4 + q;
            const bool rowok = ts < t1 && n < a.N;
            if (HAS_OFF) zz += d.off;
            double l1, l2;
            if (VT_ABL32 & 2) { l1 = zz; l2 = 1.0; } else
            vt_link(FAM, zz, d.y, l1, l2);
            const double rr = a.scale * l1;
            const double dw = rowok ? a.scale * d.w * l2 : 0.0;
            // branch-free per-row outputs: lanes with nothing to store write to a sink
            const bool st = rowok && c < 4;
            double* rp = (st && a.r_out) ? a.r_out + n : a.sink + threadIdx.x;
            double* dp = (st && a.d_out) ? a.d_out + n : a.sink + 512 + threadIdx.x;
            *rp = rr;
            *dp = dw;
            Link lk;
            lk.dw = (float)dw;
            lk.gwt = rowok ? (float)(d.w * rr) : 0.f;
            return lk;
        };
        auto mfma_step = [&](const float (&x)[NB], float dws, float gws) {
#pragma unroll
            for (int b = 0; b < NB; ++b) gacc[b] = fmaf(gws, x[b], gacc[b]);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float bj = dws * x[j];
#pragma unroll
                for (int i = j; i < NB; ++i) {
                    if (VT_ABL32 & 1) acc[vt_tri(i, j)][0] += x[i] * bj; else
                    acc[vt_tri(i, j)] = mfma_f32(x[i], bj, acc[vt_tri(i, j)]);
                }
            }
        };
        auto mfma_group = [&](const Grp32<NB>& d, const Link& lk) {
            mfma_step(d.x[0], bcast16<0>(lk.dw), bcast16<0>(lk.gwt));
            mfma_step(d.x[1], bcast16<1>(lk.dw), bcast16<1>(lk.gwt));
            mfma_step(d.x[2], bcast16<2>(lk.dw), bcast16<2>(lk.gwt));
            mfma_step(d.x[3], bcast16<3>(lk.dw), bcast16<3>(lk.gwt));
        };

        // Ring of G groups, one group per region: region h multiplies group h, links group h+1 and refills the
        // buffers it has just consumed with group h+G, which is linked G-1 regions later.  Two waves per SIMD
        // cover each other's HBM latency with G = 2; a wave that has its SIMD to itself gets G = 3.
        // k-steps past t1 that the ring touches are linked as invalid (d = 0) and contribute nothing.
        constexpr int G = WAVES == 8 ? 2 : 3;
        Grp32<NB> g[G];
#pragma unroll
        for (int i = 0; i < G; ++i) load_group(t0 + 4 * i, g[i]);
        Link cur = link_group(t0, g[0]);
        for (int64_t t = t0; t < t1; t += 4 * G) {
#pragma unroll
            for (int h = 0; h < G; ++h) {
                mfma_group(g[h], cur);
                cur = link_group(t + 4 * (h + 1), g[(h + 1) % G]);
                if (!(VT_ABL32 & 4)) load_group(t + 4 * (h + G), g[h]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- fold the chunk into the fp64 slab: in phase p wave w adds its blocks blk = w + p (mod WAVES), so the
        //      waves always touch disjoint blocks; the gradient is added by one wave per phase.
        double gv[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            double v = (double)gacc[b];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            gv[b] = v;
        }
        for (int p = 0; p < WAVES; ++p) {
#pragma unroll
            for (int blk = 0; blk < NBT; ++blk)
                if (((blk - wave) & (WAVES - 1)) == p) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[(blk * 4 + r) * 64 + lane] += (double)acc[blk][r];
                }
            if (p == wave && q == 0) {
#pragma unroll
                for (int b = 0; b < NB; ++b) gsum[16 * b + c] += gv[b];
            }
            __syncthreads();
        }
    }
    double* out = a.part + (int64_t)blockIdx.x * L;
    for (int e = threadIdx.x; e < L; e += NT) out[e] = lds[e];
}

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
inline int slab_len32(int NB) { return NB * (NB + 1) / 2 * 256 + NB * 16; }

template <int NB, int FAM, bool HAS_OFF>
int launch_hess32_3(const Hess32Args& a, int grid, hipStream_t st) {
    static VtLdsAttr attr;   // per kernel instantiation, per device
    const int bytes = (slab_len32(NB) + NB * 16) * (int)sizeof(double);
    constexpr int WAVES = hess32_waves(NB);
    if (vt_allow_big_lds((const void*)k_glm_hess_f32<NB, FAM, HAS_OFF, WAVES>, attr, 96 * 1024) != VT_OK)
        return VT_ERR_HIP;
    hipLaunchKernelGGL((k_glm_hess_f32<NB, FAM, HAS_OFF, WAVES>), dim3(grid), dim3(WAVES * 64), bytes, st, a);
    return VT_OK;
}

template <int NB, int FAM>
int launch_hess32_2(const Hess32Args& a, int grid, hipStream_t st) {
    return a.offset ? launch_hess32_3<NB, FAM, true>(a, grid, st) : launch_hess32_3<NB, FAM, false>(a, grid, st);
}

template <int NB>
int launch_hess32(const Hess32Args& a, int grid, hipStream_t st) {
    switch (a.family) {
        case VT_FAMILY_LOGISTIC: return launch_hess32_2<NB, VT_FAMILY_LOGISTIC>(a, grid, st);
        case VT_FAMILY_POISSON: return launch_hess32_2<NB, VT_FAMILY_POISSON>(a, grid, st);
        default: return launch_hess32_2<NB, VT_FAMILY_GAUSSIAN>(a, grid, st);
    }
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
    auto load_group = [&](int64_t grp) {
        const int64_t row = grp * 16 + n;
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
        const int64_t row = grp * 16 + n;
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

extern "C" size_t vt_glm_hessian_f32_workspace_bytes(int64_t N, int D) {
    (void)N;
    if (D < 1 || D > VT_SMALL_D_MAX) return 0;
    return ((size_t)vt_num_cu() * slab_len32((D + 15) / 16) + 1024) * sizeof(double);
}

extern "C" int vt_glm_hessian_f32(int family, const float* X, int64_t ldx, const double* theta, const double* y,
                                  const double* w, const double* offset, double scale, const double* prior,
                                  int64_t N, int D, double* H, int ldh, double* g, double* r, double* d, void* ws,
                                  size_t ws_bytes, void* stream) {
    if (!theta || !H || N < 0 || D < 1 || ldx < D || ldh < D) return VT_ERR_BAD_ARG;
    if (N > 0 && (!X || !y || !w)) return VT_ERR_BAD_ARG;
    if (family < 0 || family > 2) return VT_ERR_BAD_ARG;
    if (D > VT_SMALL_D_MAX) return VT_ERR_UNSUPPORTED;   // the fp32 mode covers the MFMA-resident shapes only
    if (!ws || ws_bytes < vt_glm_hessian_f32_workspace_bytes(N, D)) return VT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int NB = (D + 15) / 16;
    const int L = slab_len32(NB);
    double* part = (double*)ws;
    double* sink = part + (size_t)vt_num_cu() * L;
    int grid = 0;
    if (N > 0) {
        const int64_t nsteps = (N + 3) / 4;
        grid = (int)std::max<int64_t>(1, std::min<int64_t>((nsteps + 31) / 32, vt_num_cu()));   // >= 4 k-steps per wave
        Hess32Args a{family, X, ldx, theta, y, w, offset, scale, N, D, r, d, part, sink};
        int rc = VT_ERR_UNSUPPORTED;
        switch (NB) {
#ifdef VT_F32_DEV_NB   // dev builds (tools/build_dev32.sh): one block count only, to keep the compile short
            case VT_F32_DEV_NB: rc = launch_hess32<VT_F32_DEV_NB>(a, grid, st); break;
#else
            case 1: rc = launch_hess32<1>(a, grid, st); break;
            case 2: rc = launch_hess32<2>(a, grid, st); break;
            case 3: rc = launch_hess32<3>(a, grid, st); break;
            case 4: rc = launch_hess32<4>(a, grid, st); break;
            case 5: rc = launch_hess32<5>(a, grid, st); break;
            case 6: rc = launch_hess32<6>(a, grid, st); break;
            case 7: rc = launch_hess32<7>(a, grid, st); break;
            case 8: rc = launch_hess32<8>(a, grid, st); break;
#endif
        }
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
