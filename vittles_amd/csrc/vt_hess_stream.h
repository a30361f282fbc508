// Register-streaming fused Hessian kernel (K0 + K1 + K3), shared by the fp64 path for NARROW designs (D <= 48, where
// the pass is HBM-bound and the MFMA work is small) and by the fp32 storage mode (every D <= 128):
//   H = X^T diag(d) X,  g = X^T (w r),  r, d per row   in ONE pass over X.
// Replaces autograd.hessian / autograd.grad of the reference (vittles/sensitivity_lib.py:354, 381-383).
//
// Every wave owns the WHOLE block triangle in accumulators and streams its own k-steps (4 rows each) straight from
// HBM into the MFMA operand layout (lane (q, c) loads X[row q][16 b + c]: 16 lanes read 128 / 64 contiguous bytes of a
// row); the same register is the A operand and, scaled by d_n, the B operand.  No LDS staging, X is read exactly once.
//
// VALU instructions never overlap an fp64 or f32-input MFMA on gfx950 (tools/micro/mfma_valu_overlap.hip,
// mfma_f32_valu_overlap.hip: the times add), so the VALU work per row is what decides these kernels:
//   * a GROUP of 4 k-steps (16 rows) is linked at once: the row dots of the 4 steps leave z of row (s, q) in all 16
//     lanes of lane row q; lane (q, c) keeps the one of step c & 3 and evaluates the link (exp, reciprocal, ... ~60
//     fp64 instructions) ONCE per row instead of once per lane of the row's 16-lane group; the MFMAs of step s fetch
//     d_n and w_n r_n back from lane (q, s) by DPP row broadcast.  (k_glm_hess_small links every row 16 times:
//     N = 4e6, D = 20 took 0.50 ms = 1.3 TB/s with it.)
//   * float X: the row dot is taken in f32 against theta rounded to float (z, r, d are then f32-accurate), and the
//     f32 accumulators are folded into the fp64 slab every kChunk k-steps per wave, so that the rounding error of a
//     sum stays <= ~kChunk 2^-24 relative whatever N is.  double X: everything in fp64, one fold at the end.
// Partials are reduced deterministically: waves fold into one fp64 slab in LDS in a fixed rotation, the workgroups'
// slabs are summed in a fixed order by the finalize kernel of the including file.
#pragma once
#include "vt_common.h"
#include <type_traits>
#ifndef VT_ABL32
#define VT_ABL32 0   // dev-only compile-time ablation mask (tools/build_dev32.sh): 1 no MFMA, 2 no link, 4 no loads in the loop
#endif

#ifndef VT_HS_G64
#define VT_HS_G64 0   // dev-only: ring depth (groups of 16 rows in flight per wave) of the fp64 kernel; 0 = default
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: A lane l = A[i = l & 15][k = l >> 4], B lane l = B[k = l >> 4][j = l & 15],
// C/D lane l register r = D[i = 4 (l >> 4) + r][j = l & 15]   (NOT the f64 map: row (l >> 4) + 4 r there).
__device__ __forceinline__ f32x4 mfma_f32(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <typename T>
struct HessStreamArgs {
    int family;
    const T* X;
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
    double* part;   // [gridDim.x][L] partial slabs, L = NBT*256 + NB*16: element (blk*4 + r)*64 + lane, then g
    double* sink;   // 1024 doubles nobody reads
};

template <typename T> struct HsTraits;
template <> struct HsTraits<float> {
    typedef f32x4 acc_t;
    static constexpr int64_t kChunk = 1024;   // k-steps a wave accumulates in f32 before folding (multiple of 4)
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return mfma_f32(a, b, c); }
    template <int N> static __device__ __forceinline__ float ror_add(float v) {
        return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, true));
    }
    // v of lane S of each row of 16 lanes, in every lane of that row (DPP row_newbcast)
    template <int S> static __device__ __forceinline__ float bcast(float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + S, 0xf, 0xf, true));
    }
};
template <> struct HsTraits<double> {
    typedef f64x4 acc_t;
    static constexpr int64_t kChunk = (int64_t)1 << 40;   // fp64 accumulators: one fold at the end
    static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) { return mfma_f64(a, b, c); }
    template <int N> static __device__ __forceinline__ double ror_add(double v) { return vt_ror16_add<N>(v); }
    template <int S> static __device__ __forceinline__ double bcast(double v) {
        int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + S, 0xf, 0xf, true);
        int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + S, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    }
};

// A GROUP is 4 consecutive k-steps of a wave = 16 rows.  x: lane (q, c) holds X[row q of step s][16 b + c];
// y, w, off: the per-row scalars of row (step c & 3, q) -- the row this lane LINKS.
template <typename T, int NB, int NS>
struct HsGroup {
    T x[4][NB];
    T xs[4][NS > 0 ? NS : 1];   // strip operands (below)
    double y, w, off;
};

// NS > 0 (double only, D mod 16 in 1..8): the last block row has only 4 NS real rows, and its products are NS
// v_mfma_f64_4x4x4_4b strips (16 pipe cycles each) instead of one 64-cycle 16x16x4: the four 4x4 blocks of that
// instruction take the SAME 4 rows of A -- lane l holds X[row q][16 (NB-1) + 4 r + (l & 3)], loaded as one more
// (L1-resident) operand -- against the four 4-column groups of the ordinary B operand, and result lane L holds row
// 4 r + (L >> 4), column L & 15: register r of the 16x16 accumulator (see vt_hessian.hip).  D = 20: 96 instead of
// 192 MFMA cycles per k-step.
// WAVES = 8: two waves per SIMD (<= 256 registers each); 4: one per SIMD, accumulators in the AGPR half of its 512
// registers (float, NB >= 6: an f32 MFMA wave with independent accumulators reaches the issue rate alone).
template <typename T, int NB, int NS, int FAM, bool HAS_OFF, int WAVES>
__device__ __forceinline__ void hs_body(const HessStreamArgs<T>& a) {
    static_assert(NS == 0 || std::is_same<T, double>::value, "strips are fp64 only");
    using TR = HsTraits<T>;
    using acc_t = typename TR::acc_t;
    constexpr int NT = WAVES * 64;
    constexpr int NBT = NB * (NB + 1) / 2;
    constexpr int L = NBT * 256 + NB * 16;
    constexpr int64_t kChunk = TR::kChunk;
    extern __shared__ double lds[];   // fp64 slab [NBT*256] | gradient [NB*16] | theta as T [NB*16]
    double* slab = lds;
    double* gsum = lds + NBT * 256;
    T* th = (T*)(lds + L);
    for (int t = threadIdx.x; t < L; t += NT) lds[t] = 0.0;
    for (int t = threadIdx.x; t < NB * 16; t += NT) th[t] = t < a.D ? (T)a.theta[t] : (T)0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15, cs = lane & 3;
    const int clast = min(16 * (NB - 1) + c, a.D - 1);
    const int64_t nsteps = (a.N + 3) >> 2;
    const int64_t W = (int64_t)gridDim.x * WAVES;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t per_wave = (nsteps + W - 1) / W;                      // same for every wave of the grid
    const int64_t nchunks = (per_wave + kChunk - 1) / kChunk;

    acc_t acc[NBT];
    T gacc[NB];

    for (int64_t ch = 0; ch < nchunks; ++ch) {
        const int64_t t0 = ch * kChunk;
        const int64_t t1 = min(t0 + kChunk, per_wave);
#pragma unroll
        for (int i = 0; i < NBT; ++i) acc[i] = (acc_t){0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < NB; ++b) gacc[b] = (T)0;

        // The wave's t-th k-step is global k-step gw + W t.  Unconditional loads with clamped indices: rows past N
        // re-read row N-1 and get d_n = 0, columns past D re-read column D-1 and land in padding that is never stored
        // (a per-element "load or zero" select makes hipcc branch around every load and drain vmcnt).
        auto load_group = [&](int64_t tg, HsGroup<T, NB, NS>& d) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                int64_t n = (gw + W * (tg + s)) * 4 + q;
                n = n < a.N ? n : a.N - 1;
                const T* xr = a.X + n * a.ldx;
#pragma unroll
                for (int b = 0; b < NB - 1; ++b) d.x[s][b] = xr[16 * b + c];
                d.x[s][NB - 1] = xr[clast];
#pragma unroll
                for (int r = 0; r < NS; ++r) d.xs[s][r] = xr[min(16 * (NB - 1) + 4 * r + cs, a.D - 1)];
            }
            int64_t nl = (gw + W * (tg + cs)) * 4 + q;
            nl = nl < a.N ? nl : a.N - 1;
            d.y = a.y[nl];
            d.w = a.w[nl];
            d.off = HAS_OFF ? a.offset[nl] : 0.0;
        };
        struct Link { T dw, gwt; };   // of row (step c & 3, q)
        auto link_group = [&](int64_t tg, const HsGroup<T, NB, NS>& d) -> Link {
            T thv[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) thv[b] = th[16 * b + c];
            T z[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                T p = (T)0;
#pragma unroll
                for (int b = 0; b < NB; ++b) p = fma(d.x[s][b], thv[b], p);
                p = TR::template ror_add<8>(p);
                p = TR::template ror_add<4>(p);
                p = TR::template ror_add<2>(p);
                z[s] = TR::template ror_add<1>(p);
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
            lk.dw = (T)dw;
            lk.gwt = rowok ? (T)(d.w * rr) : (T)0;
            return lk;
        };
        auto mfma_step = [&](const T (&x)[NB], const T (&xs)[NS > 0 ? NS : 1], T dws, T gws) {
#pragma unroll
            for (int b = 0; b < NB; ++b) gacc[b] = fma(gws, x[b], gacc[b]);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const T bj = dws * x[j];
#pragma unroll
                for (int i = j; i < NB; ++i) {
                    if constexpr (NS > 0) {
                        if (i == NB - 1) {
#pragma unroll
                            for (int r = 0; r < NS; ++r)
                                acc[vt_tri(i, j)][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(xs[r], bj, acc[vt_tri(i, j)][r], 0, 0, 0);
                            continue;
                        }
                    }
                    if (VT_ABL32 & 1) acc[vt_tri(i, j)][0] += x[i] * bj; else
                    acc[vt_tri(i, j)] = TR::mfma(x[i], bj, acc[vt_tri(i, j)]);
                }
            }
        };
        auto mfma_group = [&](const HsGroup<T, NB, NS>& d, const Link& lk) {
            mfma_step(d.x[0], d.xs[0], TR::template bcast<0>(lk.dw), TR::template bcast<0>(lk.gwt));
            mfma_step(d.x[1], d.xs[1], TR::template bcast<1>(lk.dw), TR::template bcast<1>(lk.gwt));
            mfma_step(d.x[2], d.xs[2], TR::template bcast<2>(lk.dw), TR::template bcast<2>(lk.gwt));
            mfma_step(d.x[3], d.xs[3], TR::template bcast<3>(lk.dw), TR::template bcast<3>(lk.gwt));
        };

        // Ring of G groups, one group per region: region h multiplies group h, links group h+1 and refills the
        // buffers it has just consumed with group h+G, which is linked G-1 regions later.  Two waves per SIMD
        // cover each other's HBM latency with G = 2; a wave that has its SIMD to itself gets G = 3.
        // k-steps past t1 that the ring touches are linked as invalid (d = 0) and contribute nothing.
        // (deeper rings were measured for the narrow fp64 designs: G = 3 no change, G = 4, 6 slower)
        constexpr int G = std::is_same<T, double>::value ? (VT_HS_G64 > 0 ? VT_HS_G64 : 2) : (WAVES == 8 ? 2 : 3);
        HsGroup<T, NB, NS> g[G];
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

template <typename T, int NB, int NS, int FAM, bool HAS_OFF, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_glm_hess_stream(HessStreamArgs<T> a) {
    hs_body<T, NB, NS, FAM, HAS_OFF, WAVES>(a);
}

// (Held to 128 registers so that two workgroups share a CU, the narrow fp64 designs spill 16 .. 100 registers and
// run 1.4 .. 2.6x slower; deeper load rings, G = 3, 4, 6, do not help either: what a SIMD waits for at NB <= 2 are the
// dependent fp64 chains of the link, ~70 instructions of ~10 cycles latency per 16 rows and wave.)

// ---------------------------------------------------------------------------------------------------------------
// D <= 32 (NB <= 2), double X: the ROW-LINK kernel.  In k_glm_hess_stream the row dot costs a DPP reduction per k-step
// and the link's inputs sit in lane groups; for the narrowest designs that VALU work (not HBM, not the MFMA) is what
// the pass takes.  Here a wave works on 64 rows (16 k-steps) at a time:
//   1. its MFMA operand registers (lane (q, c) = X[row 4 s + q][16 b + c], buffer loads with SGPR base and zero-filled
//      tail: no index clamping) are also written to a wave-private LDS tile [64 rows][16 NB + 1];
//   2. LANE n LINKS ROW n: z = sum_j tile[n][j] theta[j] is a plain FMA chain (no cross-lane reduction), the link is
//      evaluated once per row with all 64 lanes busy, r and d are stored as 512 contiguous bytes, and d_n, w_n r_n go
//      to LDS;
//   3. the 16 k-steps multiply: d_n and w_n r_n come back as LDS broadcasts, and each operand register is refilled
//      with the next 64 rows right after its last use (a whole iteration ahead of its next one).
// VALU instructions per 64 rows: ~180 instead of ~880; N = 4e6: D = 20 0.22 -> see DESIGN.md section 4.
// NB = 3, 4 (D <= 64): the LDS tiles of 64 rows would not fit, so a wave takes RPW = 32 rows (8 k-steps) at a time and
// lanes n and n + 32 each take half of row n's dot (one cross-lane add); both link the row, lane n stores.
// NB = 5 (D <= 80): 16 rows, four lanes per row; beyond that one wave cannot hold the whole triangle (NBT * 8
// accumulator registers) at two waves per SIMD, which is what the tile kernel of vt_hessian.hip is for.
#ifndef VT_RL_RPW12
#define VT_RL_RPW12 64   // rows per wave and iteration of the row-link kernel for NB <= 2 (dev switch)
#endif
struct RowLinkLds {   // offsets in doubles; the tiles hold T (esz = sizeof(T))
    int slab, gsum, theta, waves, per_wave, ts, total;
    __host__ __device__ RowLinkLds(int NB, int esz) {
        const int RPW = NB <= 2 ? VT_RL_RPW12 : (NB <= 4 ? 32 : 16);
        const int NBT = NB * (NB + 1) / 2;
        slab = 0;
        gsum = NBT * 256;
        theta = gsum + NB * 16;
        waves = theta + NB * 16;
        ts = 16 * NB + 1;                 // odd row stride: lane-per-row reads are conflict free
        per_wave = ((RPW * ts + 128) * esz + 7) / 8;        // tile | d_n [64] | w_n r_n [64]
        total = waves + 8 * per_wave;
    }
};

template <typename T> struct RlLoad;
template <> struct RlLoad<double> {
    static __device__ __forceinline__ double ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    }
};
template <> struct RlLoad<float> {
    static __device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    }
};

// T = float (fp32 storage mode, vt_f32.hip): operands, tile, row dot and the MFMA are f32, the link is fp64, and the
// f32 accumulators are folded into the fp64 slab every HsTraits<float>::kChunk k-steps (all waves of a workgroup
// fold together: the rounds are counted on the workgroup-uniform bound ceil(nsg / W)).  NB <= 8: the f32 triangle is
// half the registers.
template <typename T, int NB, int NS, int FAM, bool HAS_OFF>
__global__ __launch_bounds__(512) void k_glm_hess_rowlink(HessStreamArgs<T> a) {
    static_assert(NS == 0 || std::is_same<T, double>::value, "strips are fp64 only");
    using TR = HsTraits<T>;
    using acc_t = typename TR::acc_t;
    constexpr int ES = (int)sizeof(T);
    constexpr int NT = 512, WAVES = 8;
    constexpr int NBT = NB * (NB + 1) / 2;
    constexpr int L = NBT * 256 + NB * 16;
    constexpr int TS = 16 * NB + 1;
    constexpr int RPW = NB <= 2 ? VT_RL_RPW12 : (NB <= 4 ? 32 : 16);     // rows per wave and iteration
    constexpr int KS = RPW / 4;                // k-steps per iteration
    constexpr int HALVES = 64 / RPW;           // lanes sharing a row's dot
    constexpr int CPL = 16 * NB / HALVES;      // columns per lane in the dot
        constexpr int64_t kIterChunk = TR::kChunk / KS;            // iterations between two folds
    extern __shared__ double lds[];
    const RowLinkLds lay(NB, ES);
    double* slab = lds + lay.slab;
    double* gsum = lds + lay.gsum;
    T* th = (T*)(lds + lay.theta);
    for (int t = threadIdx.x; t < L; t += NT) lds[t] = 0.0;
    for (int t = threadIdx.x; t < NB * 16; t += NT) th[t] = t < a.D ? (T)a.theta[t] : (T)0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, c = lane & 15, cs = lane & 3;
    T* tile = (T*)(lds + lay.waves + wave * lay.per_wave);
    T* dwl = tile + RPW * TS;
    T* gwl = dwl + 64;
    const int lrow = lane & (RPW - 1), lhalf = lane / RPW;     // the row this lane links, and its part of the dot
    const int64_t nsg = (a.N + RPW - 1) / RPW;                 // super-groups of RPW rows
    const int64_t W = (int64_t)gridDim.x * WAVES;
    const unsigned voff = (unsigned)((q * a.ldx + c) * ES);    // lane (q, c): row q of a k-step, column c of a block
    const unsigned step_bytes = (unsigned)(4 * a.ldx * ES);    // 4 rows

    // descriptor of the RPW rows of super-group sg: rows past N and everything of a super-group past the end read 0
    auto rsrc_of = [&](int64_t sg) {
        const int64_t r0 = sg * RPW;
        const int64_t rows = sg < nsg ? min((int64_t)RPW, a.N - r0) : 0;
        const unsigned bytes = rows > 0 ? (unsigned)(((rows - 1) * a.ldx + a.D) * ES) : 0u;
        const T* base = a.X + (rows > 0 ? r0 : 0) * a.ldx;
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
    };
    auto load_op = [&](__amdgpu_buffer_rsrc_t r, int s, int b) -> T {
        return RlLoad<T>::ld(r, voff + (unsigned)(16 * ES) * b, step_bytes * s);
    };

    acc_t acc[NBT];
    T gacc[NB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < NBT; ++i) acc[i] = (acc_t){0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < NB; ++b) gacc[b] = (T)0;
    };
    zero_acc();

    T xop[KS][NB];
    int64_t sg = (int64_t)blockIdx.x * WAVES + wave;
    {
        const __amdgpu_buffer_rsrc_t r0 = rsrc_of(sg);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int b = 0; b < NB; ++b) xop[s][b] = load_op(r0, s, b);
    }
    // the per-row scalars of lane n's row travel one iteration ahead as well (loaded where they are used they would
    // expose a full HBM miss per 64 rows)
    double yn, wn, offn;
    auto load_scalars = [&](int64_t g) {
        const int64_t rw = g * RPW + lrow;
        const int64_t rc = rw < a.N ? rw : a.N - 1;
        yn = a.y[rc];
        wn = a.w[rc];
        offn = HAS_OFF ? a.offset[rc] : 0.0;
    };
    load_scalars(sg);
    const int64_t iters_all = (nsg + W - 1) / W;               // the most iterations any wave runs (uniform)
    for (int64_t done = 0;;) {
        for (int64_t it = 0; it < kIterChunk && sg < nsg; ++it, sg += W) {
            const int64_t row = sg * RPW + lrow;
            const bool ok = row < a.N;
            const double yv = yn, wv = wn, offv = offn;
            // 1. operand registers -> the wave's LDS tile
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int b = 0; b < NB; ++b) tile[(4 * s + q) * TS + 16 * b + c] = xop[s][b];
            // 2. lane n links row n (columns past D meet theta = 0)
            T z0 = 0, z1 = 0;
#pragma unroll
            for (int j = 0; j < CPL; j += 2) {
                z0 = fma(tile[lrow * TS + lhalf * CPL + j], th[lhalf * CPL + j], z0);
                z1 = fma(tile[lrow * TS + lhalf * CPL + j + 1], th[lhalf * CPL + j + 1], z1);
            }
            T zt = z0 + z1;
            if constexpr (HALVES >= 2) zt += __shfl_xor(zt, 32, 64);
            if constexpr (HALVES == 4) zt += __shfl_xor(zt, 16, 64);
            double zz = (double)zt;
            if (HAS_OFF) zz += offv;
            double l1, l2;
            vt_link(FAM, zz, yv, l1, l2);
            const double rr = a.scale * l1;
            const double dw = ok ? a.scale * wv * l2 : 0.0;
            const bool st = ok && lhalf == 0;
            double* rp = (st && a.r_out) ? a.r_out + row : a.sink + threadIdx.x;
            double* dp = (st && a.d_out) ? a.d_out + row : a.sink + 512 + threadIdx.x;
            *rp = rr;
            *dp = dw;
            dwl[lrow] = (T)dw;                 // (all lanes of a row write the same value)
            gwl[lrow] = ok ? (T)(wv * rr) : (T)0;
            // 3. the k-steps; every operand register is refilled with the next super-group right after its last use
            const __amdgpu_buffer_rsrc_t rn = rsrc_of(sg + W);
            load_scalars(sg + W);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const T dws = dwl[4 * s + q], gws = gwl[4 * s + q];
                T xs[NS > 0 ? NS : 1];
#pragma unroll
                for (int r = 0; r < NS; ++r) xs[r] = tile[(4 * s + q) * TS + 16 * (NB - 1) + 4 * r + cs];
#pragma unroll
                for (int b = 0; b < NB; ++b) gacc[b] = fma(gws, xop[s][b], gacc[b]);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const T bj = dws * xop[s][j];
#pragma unroll
                    for (int i = j; i < NB; ++i) {
                        if constexpr (NS > 0) {
                            if (i == NB - 1) {
#pragma unroll
                                for (int r = 0; r < NS; ++r)
                                    acc[vt_tri(i, j)][r] =
                                        __builtin_amdgcn_mfma_f64_4x4x4f64(xs[r], bj, acc[vt_tri(i, j)][r], 0, 0, 0);
                                continue;
                            }
                        }
                        acc[vt_tri(i, j)] = TR::mfma(xop[s][i], bj, acc[vt_tri(i, j)]);
                    }
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) xop[s][b] = load_op(rn, s, b);
            }
        }

        // ---- fold into the fp64 slab (fixed rotation over the waves)
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
        done += kIterChunk;
        if (done >= iters_all) break;
        zero_acc();
    }
    // one slab per workgroup
    double* out = a.part + (int64_t)blockIdx.x * L;
    for (int e = threadIdx.x; e < L; e += NT) out[e] = lds[e];
}

template <typename T, int NB, int NS, int FAM, bool HAS_OFF>
int rl_launch3(const HessStreamArgs<T>& a, int grid, hipStream_t st) {
    static VtLdsAttr attr;   // per kernel instantiation, per device
    const int bytes = RowLinkLds(NB, (int)sizeof(T)).total * (int)sizeof(double);
    if (vt_allow_big_lds((const void*)k_glm_hess_rowlink<T, NB, NS, FAM, HAS_OFF>, attr, 160 * 1024) != VT_OK) return VT_ERR_HIP;
    hipLaunchKernelGGL((k_glm_hess_rowlink<T, NB, NS, FAM, HAS_OFF>), dim3(grid), dim3(512), bytes, st, a);
    return VT_OK;
}

template <typename T, int NB, int NS>
int rl_launch2(const HessStreamArgs<T>& a, int grid, hipStream_t st) {
    const bool off = a.offset != nullptr;
    switch (a.family) {
        case VT_FAMILY_LOGISTIC:
            return off ? rl_launch3<T, NB, NS, VT_FAMILY_LOGISTIC, true>(a, grid, st)
                       : rl_launch3<T, NB, NS, VT_FAMILY_LOGISTIC, false>(a, grid, st);
        case VT_FAMILY_POISSON:
            return off ? rl_launch3<T, NB, NS, VT_FAMILY_POISSON, true>(a, grid, st)
                       : rl_launch3<T, NB, NS, VT_FAMILY_POISSON, false>(a, grid, st);
        default:
            return off ? rl_launch3<T, NB, NS, VT_FAMILY_GAUSSIAN, true>(a, grid, st)
                       : rl_launch3<T, NB, NS, VT_FAMILY_GAUSSIAN, false>(a, grid, st);
    }
}

// strips of the last block row: D mod 16 in 1..8, double X only
template <typename T, int NB>
int rl_launch(const HessStreamArgs<T>& a, int grid, hipStream_t st) {
    if constexpr (std::is_same<T, double>::value) {
        const int rem = a.D & 15;
        if (rem >= 1 && rem <= 4) return rl_launch2<T, NB, 1>(a, grid, st);
        if (rem >= 5 && rem <= 8) return rl_launch2<T, NB, 2>(a, grid, st);
    }
    return rl_launch2<T, NB, 0>(a, grid, st);
}

// rows per wave and iteration of the row-link kernel, and its grid for N rows on cu compute units
inline int rl_rows_per_wave(int NB) { return NB <= 2 ? VT_RL_RPW12 : (NB <= 4 ? 32 : 16); }
inline int rl_grid(int64_t N, int NB, int cu, int wgs_per_cu = 1) {
    const int64_t nsg = (N + rl_rows_per_wave(NB) - 1) / rl_rows_per_wave(NB);
    return (int)std::max<int64_t>(1, std::min<int64_t>((nsg + 7) / 8, (int64_t)cu * wgs_per_cu));
}

inline int hs_slab_len(int NB) { return NB * (NB + 1) / 2 * 256 + NB * 16; }

template <typename T, int NB, int NS, int FAM, bool HAS_OFF, int WAVES>
int hs_launch3(const HessStreamArgs<T>& a, int grid, hipStream_t st) {
    static VtLdsAttr attr;   // per kernel instantiation, per device
    const int bytes = (hs_slab_len(NB) + NB * 16) * (int)sizeof(double);
    if (vt_allow_big_lds((const void*)k_glm_hess_stream<T, NB, NS, FAM, HAS_OFF, WAVES>, attr, 96 * 1024) != VT_OK)
        return VT_ERR_HIP;
    hipLaunchKernelGGL((k_glm_hess_stream<T, NB, NS, FAM, HAS_OFF, WAVES>), dim3(grid), dim3(WAVES * 64), bytes, st, a);
    return VT_OK;
}

template <typename T, int NB, int NS, int WAVES>
int hs_launch2(const HessStreamArgs<T>& a, int grid, hipStream_t st) {
    const bool off = a.offset != nullptr;
    switch (a.family) {
        case VT_FAMILY_LOGISTIC:
            return off ? hs_launch3<T, NB, NS, VT_FAMILY_LOGISTIC, true, WAVES>(a, grid, st)
                       : hs_launch3<T, NB, NS, VT_FAMILY_LOGISTIC, false, WAVES>(a, grid, st);
        case VT_FAMILY_POISSON:
            return off ? hs_launch3<T, NB, NS, VT_FAMILY_POISSON, true, WAVES>(a, grid, st)
                       : hs_launch3<T, NB, NS, VT_FAMILY_POISSON, false, WAVES>(a, grid, st);
        default:
            return off ? hs_launch3<T, NB, NS, VT_FAMILY_GAUSSIAN, true, WAVES>(a, grid, st)
                       : hs_launch3<T, NB, NS, VT_FAMILY_GAUSSIAN, false, WAVES>(a, grid, st);
    }
}

// strips of the last block row: D mod 16 in 1..8, double X only
template <typename T, int NB, int WAVES>
int hs_launch(const HessStreamArgs<T>& a, int grid, hipStream_t st) {
    if constexpr (std::is_same<T, double>::value) {
        const int rem = a.D & 15;
        if (rem >= 1 && rem <= 4) return hs_launch2<T, NB, 1, WAVES>(a, grid, st);
        if (rem >= 5 && rem <= 8) return hs_launch2<T, NB, 2, WAVES>(a, grid, st);
    }
    return hs_launch2<T, NB, 0, WAVES>(a, grid, st);
}

}  // namespace
