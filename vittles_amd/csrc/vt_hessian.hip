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
#include "vt_hess_stream.h"   // k_glm_hess_stream<double, ...>: narrow designs (NB <= kStreamNB), one link per row
#include <cstdlib>
#include <type_traits>
#include <utility>
#ifndef VT_ABL
#define VT_ABL 0   // dev-only compile-time ablation mask (tools/build_abl.cmd)
#endif

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

// ---------------------------------------------------------------------------------------
// Tile kernel (the one that runs for N >= kTileMinRows): 12 waves per CU share TR-row tiles of X through LDS.
//
// What shapes it (measured on MI355X, tools/micro/mfma_valu_overlap.hip, mfma_f64_rate.hip, mfma_f64_4x4.hip):
// v_mfma_f64_16x16x4 occupies the SIMD's DP pipe for 64 cycles; two or three waves of independent MFMAs reach
// 99 % of the 78.6 TFLOP/s peak (one wave alone issues one every ~140 cycles), but every VALU instruction of ANY
// wave on that SIMD adds its own ~4.4 cycles on top (nothing rides "beside" an fp64 MFMA).  So the kernel
// minimises the VALU instructions executed per MFMA, keeps every memory operation off the VALU, and keeps the
// waves that issue MFMAs free of everything else:
//   * X goes HBM -> LDS by buffer_load_dwordx4 ... lds (one instruction per row, no registers, no ds_write;
//     padding columns are zero-filled by the descriptor's range check), two periods ahead of its use, three
//     tile buffers in rotation; the per-row scalars y, w, offset ride along the same way (dword DMA), so NO
//     load of the loop has a VGPR destination and the compiler never parks a wave on s_waitcnt vmcnt(0).
//   * each SIMD owns a QUARTER of the block triangle (cost-balanced, column-wise: make_own); the rows of the
//     last block row that exist only as padding are skipped by computing that row with 4x4x4 MFMA strips.
//   * wave group 0 (one wave per SIMD) is a SERVICE wave: every period it links the SIMD's rows of tile p+1 --
//     row dot by DPP; exp, rcp, ... ONCE per row, one lane per row; gradient; per-row outputs; d_n -> LDS.
//     It issues no MFMA, holds no accumulator, and runs its own code region (its registers do not add to the
//     MFMA waves').
//   * wave groups 1 and 2 split the k-steps of tile p: A and B operands are ds_read_b64's of the tile (row
//     stride = 16 mod 32 doubles: conflict free), B scaled by d_n on the way.  Group 1 also issues the DMA of
//     tile p+2 at the start of the period (SALU + 15 DMA instructions) and waits for it BEFORE the period's
//     barrier, a whole period later -- which is what lets the service wave link tile p+2 in period p+1 without
//     a vmcnt wait of its own.
//   period p:  group 1: DMA(p+2) -> buf[(p+2)%3] | group 0: link(p+1) | groups 1, 2: MFMA(p) from buf[p%3] | barrier
// History (tools/ab_hess.sh, N = 10^6, D = 100, whole call): three rotating roles with 16x16x4 products only
// 0.307 ms; + strips 0.272; + column-wise ownership 0.261; + LINK wave without MFMAs 0.256; fixed service wave
// in its own code region 0.251; + DMA moved to an MFMA wave 0.244.  Did not pay: operand prefetch one k-step
// ahead, s_setprio for either kind of wave, 16 waves per CU (three MFMA waves per SIMD: same time), one
// descriptor per tile with the row in the scalar offset (the range check sees it).
// ---------------------------------------------------------------------------------------
typedef unsigned vt_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* vt_ldsptr;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const double* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}

constexpr int kTileNQ = 4;                     // owners of the block triangle (one per SIMD)
#ifndef VT_TILE_NM
#define VT_TILE_NM 2
#endif
constexpr int kTileNM = VT_TILE_NM;            // MFMA waves per SIMD (3 measured the same)
constexpr int kTileWaves = kTileNQ * (kTileNM + 1);
constexpr int kTileThreads = 64 * kTileWaves;
constexpr int kTileMinRows = 4096;

__host__ __device__ constexpr int tile_rs(int NB) { return (NB & 1) ? 16 * NB : 16 * NB + 16; }
__host__ __device__ constexpr int tile_rows(int NB) { return NB <= 7 ? 48 : 36; }   // 3 buffers must fit 160 KB
// Row strips of the LAST block row.  When D mod 16 is 1..8 only the first 4 (or 8) rows of block row NB-1
// are real, and its blocks are computed by NS = 1 (or 2) v_mfma_f64_4x4x4_4b products of 16 pipe cycles each
// instead of one 64-cycle 16x16x4 product: the four 4x4 blocks of that instruction take the SAME 4 rows as A
// (lane l: X[k = l>>4][16(NB-1) + 4s + (l&3)]) against the four 4-column groups of the ordinary B operand
// (lane l: d_k X[k][16j + (l&15)]), and their result lane L holds row 4s + (L>>4), column L&15 of the block:
// exactly register s of the 16x16 accumulator layout.  Measured (tools/micro/mfma_f64_4x4.hip): 16-17 pipe
// cycles, mixes with the 16x16x4 form at no cost.  NS = 0: block row NB-1 is an ordinary one.
__host__ __device__ constexpr int tile_strips(int D) { return (D % 16 >= 1 && D % 16 <= 8) ? (D % 16 + 3) / 4 : 0; }
__host__ __device__ constexpr int blk_cost(int NB, int NS, int i) { return (NS > 0 && i == NB - 1) ? 16 * NS : 64; }
// Block (i, j) of the lower block triangle -> bit / accumulator order.
__host__ __device__ constexpr int blk_ord(int NB, int i, int j) { return (NB * (NB + 1) - (i + 1) * (i + 2)) / 2 + j; }
// Which SIMD (quarter) owns which block.  What a quarter pays per k-step is the MFMA pipe cycles of its blocks
// plus one v_mul_f64 (d_n * x: the B operand) per distinct block COLUMN it touches, so: whole columns are dealt
// out longest-first to the least loaded quarter, then single blocks move from the most loaded quarter to
// another one for as long as that lowers the maximum.  D = 100 (7 block rows, last one a 4-row strip):
// {6 full | 5 full + 3 strips | 5 full + 2 strips | 5 full + 2 strips} = 384 / 368 / 352 / 352 pipe cycles with
// 1 / 3 / 2 / 2 multiplies (bottom-row-first runs of equal cost: 7 / 6 / 5 / 3 multiplies).
struct OwnTab { int own[8][8]; };
constexpr int kMulCycles = 5;
__host__ __device__ constexpr int own_load(const OwnTab& t, int NB, int NS, int q) {
    int pipe = 0, cols = 0;
    for (int j = 0; j < NB; ++j) {
        bool any = false;
        for (int i = j; i < NB; ++i)
            if (t.own[i][j] == q) { pipe += blk_cost(NB, NS, i); any = true; }
        cols += any ? 1 : 0;
    }
    return pipe + kMulCycles * cols;
}
__host__ __device__ constexpr OwnTab make_own(int NB, int NS, int NQ) {
    OwnTab t{};
    int colcost[8] = {}, loads[8] = {};
    bool used[8] = {};
    for (int j = 0; j < NB; ++j)
        for (int i = j; i < NB; ++i) colcost[j] += blk_cost(NB, NS, i);
    for (int n = 0; n < NB; ++n) {
        int jb = -1, qb = 0;
        for (int j = 0; j < NB; ++j)
            if (!used[j] && (jb < 0 || colcost[j] > colcost[jb])) jb = j;
        for (int q = 1; q < NQ; ++q)
            if (loads[q] < loads[qb]) qb = q;
        used[jb] = true;
        loads[qb] += colcost[jb];
        for (int i = jb; i < NB; ++i) t.own[i][jb] = qb;
    }
    for (int it = 0; it < 64; ++it) {
        int L[8] = {}, qmax = 0;
        long long mx = 0, sq = 0;
        for (int q = 0; q < NQ; ++q) {
            L[q] = own_load(t, NB, NS, q);
            if (L[q] > L[qmax]) qmax = q;
            sq += (long long)L[q] * L[q];
        }
        mx = L[qmax];
        long long bmx = mx, bsq = sq;
        int bi = -1, bj = -1, bq = -1;
        for (int j = 0; j < NB; ++j)
            for (int i = j; i < NB; ++i) {
                if (t.own[i][j] != qmax) continue;
                for (int q = 0; q < NQ; ++q) {
                    if (q == qmax) continue;
                    t.own[i][j] = q;
                    long long m2 = 0, s2 = 0;
                    for (int k = 0; k < NQ; ++k) {
                        const long long l = own_load(t, NB, NS, k);
                        m2 = l > m2 ? l : m2;
                        s2 += l * l;
                    }
                    t.own[i][j] = qmax;
                    if (m2 < bmx || (m2 == bmx && s2 < bsq)) { bmx = m2; bsq = s2; bi = i; bj = j; bq = q; }
                }
            }
        if (bi < 0) break;
        t.own[bi][bj] = bq;
    }
    return t;
}
// The masks the kernel tests (compile-time constants): OWN bit blk_ord(i, j): quarter Q owns block (i, j);
// NEED bit b: it reads the ordinary operand of column block b (A of a full block, or B); COLS bit j: it owns a
// block in block column j (needs the scaled B operand d * x_j); strip: it owns a block of the strip row.
__host__ __device__ constexpr unsigned long long own_mask(const OwnTab& t, int NB, int Q) {
    unsigned long long m = 0;
    for (int i = 0; i < NB; ++i)
        for (int j = 0; j <= i; ++j)
            if (t.own[i][j] == Q) m |= 1ull << blk_ord(NB, i, j);
    return m;
}
__host__ __device__ constexpr unsigned need_mask(const OwnTab& t, int NB, int NS, int Q) {
    unsigned m = 0;
    for (int i = 0; i < NB; ++i)
        for (int j = 0; j <= i; ++j)
            if (t.own[i][j] == Q) {
                m |= 1u << j;
                if (!(NS > 0 && i == NB - 1)) m |= 1u << i;
            }
    return m;
}
__host__ __device__ constexpr unsigned cols_mask(const OwnTab& t, int NB, int Q) {
    unsigned m = 0;
    for (int i = 0; i < NB; ++i)
        for (int j = 0; j <= i; ++j)
            if (t.own[i][j] == Q) m |= 1u << j;
    return m;
}
__host__ __device__ constexpr bool strip_needed(const OwnTab& t, int NB, int NS, int Q) {
    if (NS == 0) return false;
    for (int j = 0; j < NB; ++j)
        if (t.own[NB - 1][j] == Q) return true;
    return false;
}

template <int NB>
struct TileLds {   // offsets in doubles
    static constexpr int RS = tile_rs(NB);
    static constexpr int TR = tile_rows(NB);
    static constexpr int kTile = 0;                       // [3][TR][RS]
    static constexpr int kD = 3 * TR * RS;                // [3][TR]  d_n of the tile
    static constexpr int kTheta = kD + 3 * TR;            // [NB*16]
    static constexpr int kScratch = kTheta + NB * 16;     // [4 SIMDs][32]: z and w_n r_n of the duty wave's rows
    static constexpr int kY = kScratch + 4 * 32;          // [3][TR] y_n, [3][TR] w_n, [3][TR] offset_n of the tiles
    static constexpr int kW = kY + 3 * TR;
    static constexpr int kOff = kW + 3 * TR;
    static constexpr int kFlags = kOff + 3 * TR;          // [4 SIMDs][4] ints: tiles landed / linked / consumed by MFMA wave 1
    static constexpr int kTotal = kFlags + 8;
};

// SIMD-local hand-over counters of the k-step split (each SIMD's rows are private to its three waves, so nothing
// in the tile loop needs the workgroup-wide s_barrier): monotone counts in LDS, one writer each, polled with
// s_sleep in between.  Every wait is bounded (a wave that gives up runs on and the results are wrong, but the grid
// always drains).
// Both measured (round 3, N = 10^6, D = 100): they remove stalls -- SQ_WAVE_CYCLES per launch 4.16e8 -> 4.04e8 (flags)
// -> 3.97e8 (flags + prefetch), MFMA pipe 65.6 -> 68.7 % busy -- and the wall time stays where it was (0.243 /
// 0.243 / 0.241 ms): the chip answers the tighter issue stream with a lower clock (2.29 -> 2.17 GHz).  Off by
// default: the s_barrier version is the simpler program.
#ifndef VT_TILE_FLAGS
#define VT_TILE_FLAGS 0    // dev switch: SIMD-local hand-over counters instead of the workgroup barrier
#endif
#ifndef VT_TILE_PF
#define VT_TILE_PF 0       // dev switch: operand reads one k-step ahead of the products (two register sets)
#endif
enum { kFlagLanded = 0, kFlagLinked = 1, kFlagDoneB = 2 };
__device__ __forceinline__ void tile_flag_wait(unsigned addr, int target) {
    for (int spins = 0; spins < (1 << 20); ++spins) {
        int v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        if (__builtin_amdgcn_readfirstlane(v) >= target) return;
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void tile_flag_set(unsigned addr, int value) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // everything this wave read or wrote in LDS is done
    if ((threadIdx.x & 63) == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(value) : "memory");
}

// Which rows of X a workgroup's k-th tile holds (tiles are dealt round-robin over the grid).
template <int TR>
struct TileRows {
    const HessArgs& a;
    int64_t mine;
    __device__ __forceinline__ int64_t first_row(int64_t k) const { return (blockIdx.x + k * (int64_t)gridDim.x) * (int64_t)TR; }
    __device__ __forceinline__ int nrow(int64_t k) const {
        const int64_t left = (k >= 0 && k < mine) ? a.N - first_row(k) : 0;
        return (int)(left < TR ? (left > 0 ? left : 0) : TR);
    }
};

// SERVICE wave (wave group 0, one per SIMD): links the RPS rows [RPS*simd, RPS*(simd+1)) of tile p+1 in period p.
// The tile landed a period ago: MFMA wave 0 of this SIMD issued its DMA and waited for it before the barrier.
template <int NB, int FAM, int TURNS, bool FLAGS>
__device__ __forceinline__ void hess_tile_service(const HessArgs& a, double* lds, int64_t mine, int simd) {
    using LY = TileLds<NB>;
    constexpr int RS = LY::RS, TR = LY::TR;
    constexpr int NBT = NB * (NB + 1) / 2;
    constexpr int RPS = TR / 4;                        // rows a service wave links
    constexpr int PASSES = (RPS + 3) / 4;
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, c = lane & 15;
    const double* th = lds + LY::kTheta;
    double* zs = lds + LY::kScratch + simd * 32;
    // scalar stage of the link: lane t < RPS owns row RPS*simd + t
    const int trow = RPS * simd + (lane & 15);
    const bool tlane = lane < RPS;
    const int trow_c = trow < TR ? trow : TR - 1;       // in-bounds LDS index for the other lanes
    const TileRows<TR> tiles{a, mine};
    double gacc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) gacc[b] = 0.0;

    auto link = [&](int64_t k, int buf) {
        const int nrow = tiles.nrow(k);
        const int64_t rb = nrow > 0 ? tiles.first_row(k) : 0;
        double x[PASSES][NB];
        // stage 1: z_n = x_n . theta (16 lanes per row)
#pragma unroll
        for (int k2 = 0; k2 < PASSES; ++k2) {
            const int rl = min(4 * k2 + q, RPS - 1);
            const double* tr = lds + LY::kTile + buf * (TR * RS) + (RPS * simd + rl) * RS + c;
            double pz = 0.0;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                x[k2][b] = tr[16 * b];
                pz = fma(x[k2][b], th[16 * b + c], pz);
            }
            const double z = vt_sum16_dpp(pz);
            if (c == 0) zs[4 * k2 + q] = z;
        }
        // stage 2: the scalar chain, one lane per row
        {
            const double sy = lds[LY::kY + buf * TR + trow_c];
            const double sw = lds[LY::kW + buf * TR + trow_c];
            const double z = zs[lane & 15] + lds[LY::kOff + buf * TR + trow_c];
            double l1, l2;
            vt_link(FAM, z, sy, l1, l2);
            const bool rowok = tlane && trow < nrow;
            const double rr = a.scale * l1;
            const double dw = rowok ? a.scale * sw * l2 : 0.0;
            const double gwt = rowok ? sw * rr : 0.0;
            const unsigned so = tlane ? 8u * trow : 0xfffffff0u;    // other lanes: dropped by the range check
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(vt_u32x2, rr),
                                                  tile_rsrc(a.r_out + rb, a.r_out ? 8u * nrow : 0u), so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(vt_u32x2, dw),
                                                  tile_rsrc(a.d_out + rb, a.d_out ? 8u * nrow : 0u), so, 0, 0);
            if (tlane) lds[LY::kD + buf * TR + trow] = dw;
            if (lane < 16) zs[16 + lane] = gwt;
        }
        // stage 3: gradient
#pragma unroll
        for (int k2 = 0; k2 < PASSES; ++k2) {
            const double gwt = (4 * k2 + q < RPS) ? zs[16 + 4 * k2 + q] : 0.0;
#pragma unroll
            for (int b = 0; b < NB; ++b) gacc[b] = fma(gwt, x[k2][b], gacc[b]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    const unsigned fl = (unsigned)(uintptr_t)(vt_ldsptr)((int*)(lds + LY::kFlags) + 4 * simd);
    if (!FLAGS) vt_lds_barrier();      // tile 0 has landed (MFMA wave 0 fetched it and waited)
    int b1 = 0;                        // buffer of tile p+1
    for (int64_t p = -1; p < mine; ++p) {
        if (FLAGS) tile_flag_wait(fl + 4 * kFlagLanded, (int)p + 2);      // tile p+1 is in LDS
        if (!(VT_ABL & 4)) link(p + 1, b1);
        if (FLAGS) tile_flag_set(fl + 4 * kFlagLinked, (int)p + 2);       // tiles 0..p+1 are linked
        else vt_lds_barrier();
        b1 = b1 == 2 ? 0 : b1 + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every tile is dead
    // gradient: the 4 row groups of the wave, one slot per service wave
    double* gl = lds + NBT * 256;      // [4][NB*16]
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        double v = gacc[b];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (q == 0) gl[simd * (NB * 16) + 16 * b + c] = v;
    }
#pragma unroll
    for (int turn = 0; turn < TURNS; ++turn) vt_lds_barrier();   // the MFMA waves' reduction turns
}

// MFMA wave ks (wave group ks + 1) of a SIMD.  Two ways to deal out a tile's products:
//   KSP = false: the SIMD owns a QUARTER Q of the block triangle and its MFMA waves split the tile's k-steps;
//   KSP = true : the SIMD owns a quarter of the tile's K-STEPS (the very rows its service wave links and its wave 0
//                fetches) for the WHOLE triangle, which its MFMA waves split between them (Q = ks: a cost-balanced
//                part, column-wise).  Every SIMD's pipe then carries exactly the same work (the 64-cycle
//                granularity of whole blocks leaves the quarters of D = 100 at 384 / 368 / 352 / 352 cycles per
//                k-step), and a wave reads 7-9 operands for 13-15 products instead of 8 for 7.
// Wave 0 also feeds the pipeline: DMA of tile p+2 at the start of period p.
template <int NB, int NS, int Q, bool KSP>
__device__ __forceinline__ void hess_tile_mfma(const HessArgs& a, double* lds, int64_t mine, int ks, int simd_rt) {
    using LY = TileLds<NB>;
    constexpr int NQ = KSP ? kTileNM : kTileNQ, RS = LY::RS, TR = LY::TR;
    constexpr OwnTab TAB = make_own(NB, NS, NQ);
    const int simd = KSP ? simd_rt : Q;                // the SIMD this wave runs on (whose rows its wave 0 fetches)
    constexpr bool STRIP = strip_needed(TAB, NB, NS, Q);
    // ownership as compile-time bit masks (bit blk_ord(i, j) / bit b / bit j): the unrolled loops below test
    // plain constants -- left to fold make_own()'s loops per iteration, the optimiser sometimes does not
    constexpr unsigned long long OWN = own_mask(TAB, NB, Q);
    constexpr unsigned NEED = need_mask(TAB, NB, NS, Q), COLS = cols_mask(TAB, NB, Q);
    constexpr int NOWN = __builtin_popcountll(OWN);
    // accumulator slot of an owned block = its rank among the quarter's blocks
    auto slot = [](int ord) constexpr { return __builtin_popcountll(OWN & ((1ull << ord) - 1ull)); };
    constexpr int KSTEPS = TR / 4;
#ifndef VT_TILE_K0_DELTA
#define VT_TILE_K0_DELTA 0   // dev switch: k-steps taken from the DMA-issuing wave and given to the other one(s)
#endif
    constexpr int K0 = KSTEPS / kTileNM - VT_TILE_K0_DELTA;   // wave 0 (which also issues the DMA): 6 of 12, 4 of 9
    static_assert(KSP || (KSTEPS - K0) % (kTileNM - 1) == 0, "k-step split");
    static_assert(!KSP || KSTEPS % kTileNQ == 0, "k-steps per SIMD");
    constexpr int KR = (KSTEPS - K0) / (kTileNM - 1);
    constexpr int RPS = TR / 4;
    constexpr int CH = RS / 2;                         // 16-byte chunks per LDS row
    constexpr int DMAH = (CH + 63) / 64;               // DMA instructions per row
    static_assert(2 * RPS <= 64, "one DMA instruction moves a SIMD's per-row scalars");
    const int kbeg = KSP ? simd * (KSTEPS / kTileNQ) : (ks == 0 ? 0 : K0 + (ks - 1) * KR);
    const int kcnt = KSP ? KSTEPS / kTileNQ : (ks == 0 ? K0 : KR);
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, c = lane & 15;
    const TileRows<TR> tiles{a, mine};

    // the RPS rows of my SIMD of tile k -> buffer `buf`.  One descriptor per row (SALU only): lanes past the
    // row's D doubles and rows past N read 0, and the only VGPR is 16 * chunk.  The per-row scalars y, w and
    // offset of those rows ride along the same way, one dword per lane.
    auto dma = [&](int64_t k, int buf) {
        const int nrow = (VT_ABL & 2) ? 0 : tiles.nrow(k);
        const int64_t rb = nrow > 0 ? tiles.first_row(k) : 0;
        const double* xb = a.X + (rb + RPS * simd) * a.ldx;
#pragma unroll
        for (int rr = 0; rr < RPS; ++rr) {
            const int row = RPS * simd + rr;
            __amdgpu_buffer_rsrc_t xrs = tile_rsrc(xb + (int64_t)rr * a.ldx, row < nrow ? 8u * a.D : 0u);
            double* dst = lds + LY::kTile + buf * (TR * RS) + row * RS;
#pragma unroll
            for (int h = 0; h < DMAH; ++h) {
                const int chunk = 64 * h + lane;
                if (chunk < CH)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (vt_ldsptr)(dst + 128 * h), 16, 16u * chunk, 0, 0, 0);
            }
        }
        if (lane < 2 * RPS) {
            const unsigned vo = 8u * (RPS * simd) + 4u * lane;
            const int so = buf * TR + RPS * simd;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(tile_rsrc(a.y + rb, 8u * nrow), (vt_ldsptr)(lds + LY::kY + so), 4, vo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(tile_rsrc(a.w + rb, 8u * nrow), (vt_ldsptr)(lds + LY::kW + so), 4, vo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(tile_rsrc(a.offset + rb, a.offset ? 8u * nrow : 0u),
                                                     (vt_ldsptr)(lds + LY::kOff + so), 4, vo, 0, 0, 0);
        }
    };

    // accumulators: one f64x4 per owned block; a strip block only ever touches components 0..NS-1
    f64x4 acc[NOWN > 0 ? NOWN : 1];
#pragma unroll
    for (int i = 0; i < NOWN; ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const unsigned strip_lane = 8u * (unsigned)(c & 3) - 8u * (unsigned)c;   // byte offset: column c -> column c & 3

    constexpr bool FLAGS = KSP && VT_TILE_FLAGS;
    const unsigned fl = (unsigned)(uintptr_t)(vt_ldsptr)((int*)(lds + LY::kFlags) + 4 * simd);
    if (ks == 0) {                     // prologue: tile 0, landed before anybody links it
        dma(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FLAGS) tile_flag_set(fl + 4 * kFlagLanded, 1);
    }
    if (!FLAGS) vt_lds_barrier();
    int b0 = 2;                        // buffer of tile p (p = -1: unused)
    for (int64_t p = -1; p < mine; ++p) {
        // the buffer of tile p+2 is the one of tile p-1: the other MFMA wave must be done with it (this one is, and
        // the service wave linked it before either could start)
        if (FLAGS && ks == 0) tile_flag_wait(fl + 4 * kFlagDoneB, (int)p);
        if (ks == 0 && !(VT_ABL & 8)) dma(p + 2, b0 == 0 ? 2 : b0 - 1);   // buffer (p+2) % 3 == buffer of tile p-1
        if (FLAGS && p >= 0) tile_flag_wait(fl + 4 * kFlagLinked, (int)p + 1);   // tile p is linked: its d_n are in LDS
        if (p >= 0 && !(VT_ABL & 1)) {
            // The operand reads are inline asm: as compiler-visible LDS loads they make hipcc wait for this
            // wave's own LDS-DMA (vmcnt(0): it cannot tell the tile buffers apart).  A real loop, not unrolled:
            // its body is the whole instruction budget of the kernel.
            unsigned ta = (unsigned)(uintptr_t)(vt_ldsptr)(lds + LY::kTile + b0 * (TR * RS) + (q + 4 * kbeg) * RS + c);
            unsigned da = (unsigned)(uintptr_t)(vt_ldsptr)(lds + LY::kD + b0 * TR + q + 4 * kbeg);
            double xa[2][8];
            double xs[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
            double dq[2];
            // the operand reads of one k-step (4 rows) into register set `st`
            auto issue = [&](int st, unsigned ta_, unsigned da_) __attribute__((always_inline)) {
                asm volatile("ds_read_b64 %0, %1" : "=v"(dq[st]) : "v"(da_));
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if ((NEED >> b) & 1)   // column blocks this wave's products touch
                        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xa[st][b]) : "v"(ta_), "n"(128 * b));
                    else
                        xa[st][b] = 0.0;
                }
                if constexpr (STRIP) {
                    const unsigned tas = ta_ + strip_lane;
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2)
                        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xs[st][s2]) : "v"(tas), "n"(128 * (NB - 1) + 32 * s2));
                }
            };
            // everything issued so far has landed; every operand of set `st` is tied to a volatile asm BEHIND the wait
            // (volatile asms keep their order)
            auto landed = [&](int st) __attribute__((always_inline)) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dq[st]));
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if ((NEED >> b) & 1) asm volatile("" : "+v"(xa[st][b]));
                if constexpr (STRIP) {
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2) asm volatile("" : "+v"(xs[st][s2]));
                }
            };
            auto products = [&](int st) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if (!((COLS >> j) & 1)) continue;
                    const double bj = dq[st] * xa[st][j];
#pragma unroll
                    for (int i = j; i < NB; ++i)
                        if ((OWN >> blk_ord(NB, i, j)) & 1) {
                            f64x4& ac = acc[slot(blk_ord(NB, i, j))];
                            if (NS > 0 && i == NB - 1) {
#pragma unroll
                                for (int s2 = 0; s2 < NS; ++s2)
                                    ac[s2] = __builtin_amdgcn_mfma_f64_4x4x4f64(xs[st][s2], bj, ac[s2], 0, 0, 0);
                            } else {
                                ac = mfma_f64(xa[st][i], bj, ac);
                            }
                        }
                }
            };
            if constexpr (KSP && VT_TILE_PF) {
                // the SIMD's few k-steps, unrolled, two operand sets in rotation: the reads of k-step m + 1 are in
                // flight under the products of k-step m (while this wave waits for LDS the other MFMA wave of the
                // SIMD issues alone, at less than half the pipe's rate: hardware fact 1)
                constexpr int KC = KSTEPS / kTileNQ;
                issue(0, ta, da);
#pragma unroll
                for (int m = 0; m < KC; ++m) {
                    landed(m & 1);
                    if (m + 1 < KC) issue((m + 1) & 1, ta + (m + 1) * (4 * RS * 8), da + (m + 1) * (4 * 8));
                    products(m & 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma nounroll
                for (int m = 0; m < kcnt; ++m) {
                    issue(0, ta, da);
                    landed(0);
                    products(0);
                    ta += 4 * RS * 8;
                    da += 4 * 8;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // my DMA of tile p+2, issued a whole period ago in wall time: landed before the barrier, so the service
        // wave may link it in the next period
        if (ks == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FLAGS) {
            if (ks == 0) tile_flag_set(fl + 4 * kFlagLanded, (int)p + 3);        // tiles 0..p+2 are in LDS
            else tile_flag_set(fl + 4 * kFlagDoneB, (int)p + 1);                 // tiles 0..p are consumed
        } else {
            vt_lds_barrier();
        }
        b0 = b0 == 2 ? 0 : b0 + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every tile is dead
    // H: the waves that hold parts of the same blocks add up in LDS in a fixed order: the MFMA waves of a quarter
    // (ks = 0, 1, ...), or the SIMDs (0..3) when each of them holds the whole triangle
    double* hl = lds;                  // [NBT*256]
    constexpr int TURNS = KSP ? kTileNQ : kTileNM;
#pragma unroll
    for (int turn = 0; turn < TURNS; ++turn) {
        if ((KSP ? simd : ks) == turn) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j)
                    if ((OWN >> blk_ord(NB, i, j)) & 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            double* e = hl + (vt_tri(i, j) * 4 + r) * 64 + lane;
                            // strip blocks: rows past the last strip are padding (never stored in H): zeros
                            const bool live = !(NS > 0 && i == NB - 1) || r < NS;
                            const double v = live ? acc[slot(blk_ord(NB, i, j))][r] : 0.0;
                            *e = turn == 0 ? v : *e + v;
                        }
                    }
        }
        vt_lds_barrier();
    }
}

#ifndef VT_TILE_KSPLIT
#define VT_TILE_KSPLIT 1   // dev switch: 0 = quarters of the block triangle per SIMD for every NB
#endif
// k-steps per SIMD need 4 | TR / 4 (48-row tiles: NB <= 7) and half the triangle's accumulators in one wave
__host__ __device__ constexpr bool tile_ksplit(int NB) { return VT_TILE_KSPLIT && (tile_rows(NB) / 4) % kTileNQ == 0 && NB <= 7; }

template <int NB, int NS, int FAM>
__global__ __launch_bounds__(kTileThreads) void k_glm_hess_tile(HessArgs a) {
    using LY = TileLds<NB>;
    constexpr int NBT = NB * (NB + 1) / 2;
    constexpr int L = NBT * 256 + NB * 16;
    extern __shared__ double lds[];
    for (int t = threadIdx.x; t < NB * 16; t += kTileThreads) lds[LY::kTheta + t] = t < a.D ? a.theta[t] : 0.0;
    for (int t = threadIdx.x; t < 4 * 32; t += kTileThreads) lds[LY::kScratch + t] = 0.0;
    for (int t = threadIdx.x; t < 8; t += kTileThreads) lds[LY::kFlags + t] = 0.0;
    __syncthreads();
    const int64_t ntiles = (a.N + LY::TR - 1) / LY::TR;
    const int64_t mine = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int simd = wid & 3, grp = wid >> 2;
    constexpr bool KSP = tile_ksplit(NB);
    if (grp == 0) {
        hess_tile_service<NB, FAM, KSP ? kTileNQ : kTileNM, KSP && VT_TILE_FLAGS>(a, lds, mine, simd);
    } else if constexpr (KSP) {
        static_assert(!KSP || kTileNM == 2, "two MFMA waves per SIMD split the triangle");
        if (grp == 1) hess_tile_mfma<NB, NS, 0, true>(a, lds, mine, 0, simd);    // part = MFMA wave
        else hess_tile_mfma<NB, NS, 1, true>(a, lds, mine, 1, simd);
    } else {
        switch (simd) {   // quarter = SIMD
            case 0: hess_tile_mfma<NB, NS, 0, false>(a, lds, mine, grp - 1, 0); break;
            case 1: hess_tile_mfma<NB, NS, 1, false>(a, lds, mine, grp - 1, 1); break;
            case 2: hess_tile_mfma<NB, NS, 2, false>(a, lds, mine, grp - 1, 2); break;
            default: hess_tile_mfma<NB, NS, 3, false>(a, lds, mine, grp - 1, 3); break;
        }
    }
    double* out = a.part + (int64_t)blockIdx.x * L;
    for (int e = threadIdx.x; e < NBT * 256; e += kTileThreads) out[e] = lds[e];
    for (int t = threadIdx.x; t < NB * 16; t += kTileThreads) {
        double s2 = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) s2 += lds[NBT * 256 + w * (NB * 16) + t];
        out[NBT * 256 + t] = s2;
    }
}

inline size_t tile_lds_bytes(int NB) {
    const size_t loop = (size_t)3 * tile_rows(NB) * tile_rs(NB) + 12 * tile_rows(NB) + NB * 16 + 4 * 32 + 8;
    const size_t tail = (size_t)NB * (NB + 1) / 2 * 256 + 4 * NB * 16;
    return std::max(loop, tail) * sizeof(double);
}

// Final: sum ALL partial slabs (fixed order), undo the accumulator layout, mirror the triangle, add
// the prior -- one launch.  A block owns 16 consecutive slab elements; its 16 thread groups each sum
// every 16th slab (128-byte coalesced reads, four loads in flight), then group 0 adds the 16 partial
// sums in order.
__global__ __launch_bounds__(256) void k_hess_finalize(const double* in, int L, int P, int NB, int D,
                                                       const double* prior, const double* theta,
                                                       double* H, int ldh, double* g) {
    __shared__ double sh[16][17];
    const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (e < L) {
        const double* p = in + e;
        int k = grp;
        for (; k + 48 < P; k += 64) {
            s0 += p[(int64_t)k * L];
            s1 += p[(int64_t)(k + 16) * L];
            s2 += p[(int64_t)(k + 32) * L];
            s3 += p[(int64_t)(k + 48) * L];
        }
        for (; k < P; k += 16) s0 += p[(int64_t)k * L];
    }
    sh[grp][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp != 0 || e >= L) return;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += sh[q][el];
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

inline int num_cu() { return vt_num_cu(); }   // of the current device

constexpr int kStage2 = 16;
#ifndef VT_STREAM_NB
#define VT_STREAM_NB 5
#endif
constexpr int kStreamNB = VT_STREAM_NB;   // D <= 16 kStreamNB takes k_glm_hess_stream<double> (vt_hess_stream.h)

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

template <int NB, int NS, int FAM>
int launch_hess_tile3(const HessArgs& a, int grid, hipStream_t st) {
    static VtLdsAttr attr;   // per kernel instantiation, per device
    if (vt_allow_big_lds((const void*)k_glm_hess_tile<NB, NS, FAM>, attr, 160 * 1024) != VT_OK) return VT_ERR_HIP;
    hipLaunchKernelGGL((k_glm_hess_tile<NB, NS, FAM>), dim3(grid), dim3(kTileThreads), tile_lds_bytes(NB), st, a);
    return VT_OK;
}

template <int NB, int NS>
int launch_hess_tile2(const HessArgs& a, int grid, hipStream_t st) {
    switch (a.family) {
        case VT_FAMILY_LOGISTIC: return launch_hess_tile3<NB, NS, VT_FAMILY_LOGISTIC>(a, grid, st);
        case VT_FAMILY_POISSON: return launch_hess_tile3<NB, NS, VT_FAMILY_POISSON>(a, grid, st);
        default: return launch_hess_tile3<NB, NS, VT_FAMILY_GAUSSIAN>(a, grid, st);
    }
}

template <int NB>
int launch_hess_tile(const HessArgs& a, int grid, hipStream_t st) {
    switch (tile_strips(a.D)) {
        case 1: return launch_hess_tile2<NB, 1>(a, grid, st);
        case 2: return launch_hess_tile2<NB, 2>(a, grid, st);
        default: return launch_hess_tile2<NB, 0>(a, grid, st);
    }
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

// ---- 256 < D <= kSplitDMax: the one-pass kernel on the first 256 columns + a strip product -------------------------
// The accumulators of a one-pass kernel stop fitting at D = 256 and the tiled SYRK pays a stage loop of its own for
// every partial tile row (D = 257: 30 % of peak).  In between:  with X = [X1 | X2], X1 the first 256 columns,
//   z = X1 theta1 + (X2 theta2 + offset)        -> k_glm_hess_mid on X1 with the bracket as its offset: H11, g1, r, d
//   [H21 | H22] = X2^T diag(d) [X1 | X2]        -> vt_gemm_tn (the rectangular MFMA product, split over N)
//   g2 = X2^T (w .* r)                          -> vt_gemm_t2's row kernel
// 2.1 passes over X instead of the tiled route's ~3 panel reads; N = 2e6: D = 257 5.5 -> 4.1 ms, 320 6.1 -> 5.0,
// 384 7.4 -> 6.7; past 384 the strip needs a second row of 128-column tiles and loses (D = 400: 12.9 against 9.2 ms).  VT_HESS_NO_SPLIT=1: dev A/B switch.
extern "C" int vt_gemv(const double* X, int64_t ldx, const double* v, int64_t N, int D, double* out, void* ws,
                       size_t ws_bytes, void* stream);
extern "C" int vt_gemv_t2(const double* X, int64_t ldx, const double* u1, const double* u2, int64_t N, int D,
                          double* out, void* ws, size_t ws_bytes, void* stream);
constexpr int kSplitDMax = 384;

__global__ __launch_bounds__(256) void k_add_vec(double* a, const double* b, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] += b[i];
}
// H[0:D1, D1:D] <- (H[D1:D, 0:D1])^T, prior on the diagonal of the corner, g2 += prior2 .* theta2
__global__ __launch_bounds__(256) void k_split_finish(double* H, int ldh, int D1, int D, const double* prior,
                                                      const double* theta, double* g) {
    const int t = D - D1;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < t * D1; e += gridDim.x * blockDim.x) {
        const int i = e / D1, j = e - i * D1;
        H[(int64_t)j * ldh + D1 + i] = H[(int64_t)(D1 + i) * ldh + j];
    }
    // the corner came out of a general product (a_i * (d b_j) and a_j * (d b_i) round differently): keep its lower
    // triangle, so that H is exactly symmetric like every other route's
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < t * t; e += gridDim.x * blockDim.x) {
        const int i = e / t, j = e - i * t;
        if (j > i) H[(int64_t)(D1 + i) * ldh + D1 + j] = H[(int64_t)(D1 + j) * ldh + D1 + i];
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < t && prior) {
        const int i = D1 + threadIdx.x;
        H[(int64_t)i * ldh + i] += prior[i];
        if (g) g[i] += prior[i] * theta[i];
    }
}

static bool hess_split_ok(const double* X, int64_t ldx, int D) {
    static const bool off = getenv("VT_HESS_NO_SPLIT") != nullptr;
    return !off && D > VT_MID_D_MAX && D <= kSplitDMax && vt_hess_mid_ok(X, ldx, VT_MID_D_MAX);
}
struct SplitWs { size_t slabs, off, rtmp, dtmp, g2, rows, tn, total; };
static SplitWs hess_split_ws(int64_t N, int D) {
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    SplitWs w{};
    const int t = D - VT_MID_D_MAX;
    size_t o = 0;
    w.slabs = o; o += up(vt_hess_mid_ws_bytes(VT_MID_D_MAX));
    w.off = o;   o += up((size_t)N * sizeof(double));
    w.rtmp = o;  o += up((size_t)N * sizeof(double));
    w.dtmp = o;  o += up((size_t)N * sizeof(double));
    w.g2 = o;    o += up((size_t)kSplitDMax * sizeof(double));
    w.rows = o;  o += up(vt_rows_workspace_bytes(t));
    w.tn = o;    o += up(vt_gemm_tn_workspace_bytes(t, D));
    w.total = o;
    return w;
}

extern "C" size_t vt_glm_hessian_workspace_bytes(int64_t N, int D) {
    if (D < 1) return 0;
    if (D > VT_SMALL_D_MAX) {
        size_t b = std::max(large_ws_rows(D) + (size_t)(N + D + 64) * sizeof(double) + vt_large_syrk_ws_bytes(D),
                            vt_hess_mid_ws_bytes(D));
        if (D > VT_MID_D_MAX && D <= kSplitDMax) b = std::max(b, hess_split_ws(N, D).total);
        return b;
    }
    int NB = (D + 15) / 16;
    return ((size_t)(num_cu() + kStage2) * slab_len(NB) + (size_t)num_cu() * 4 * (NB * (NB + 1) / 2) * 256 + 1024) *
           sizeof(double);
}

extern "C" int vt_glm_hessian(int family, const double* X, int64_t ldx, const double* theta,
                              const double* y, const double* w, const double* offset, double scale,
                              const double* prior, int64_t N, int D, double* H, int ldh, double* g,
                              double* r, double* d, void* ws, size_t ws_bytes, void* stream) {
    // N == 0 (no rows: H = prior, g = prior .* theta) needs no row data
    if (!theta || !H || N < 0 || D < 1 || ldx < D || ldh < D) return VT_ERR_BAD_ARG;
    if (N > 0 && (!X || !y || !w)) return VT_ERR_BAD_ARG;
    if (family < 0 || family > 2) return VT_ERR_BAD_ARG;
    if (!ws || ws_bytes < vt_glm_hessian_workspace_bytes(N, D)) return VT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (N > 0 && hess_split_ok(X, ldx, D)) {
        const int D1 = VT_MID_D_MAX, t = D - D1, NB1 = D1 / 16;
        const SplitWs lay = hess_split_ws(N, D);
        char* base = (char*)ws;
        double* off = (double*)(base + lay.off);
        double* rr = r ? r : (double*)(base + lay.rtmp);
        double* dd = d ? d : (double*)(base + lay.dtmp);
        double* g2 = g ? g + D1 : (double*)(base + lay.g2);
        void* rows_ws = base + lay.rows;
        const size_t rows_bytes = vt_rows_workspace_bytes(t);
        // the columns past 256 enter the linear predictor through the offset of the one-pass kernel
        int rc = vt_gemv(X + D1, ldx, theta + D1, N, t, off, rows_ws, rows_bytes, stream);
        if (rc != VT_OK) return rc;
        if (offset) hipLaunchKernelGGL(k_add_vec, dim3(num_cu() * 4), dim3(256), 0, st, off, offset, N);
        int nslabs = 0;
        rc = vt_hess_mid(family, X, ldx, theta, y, w, off, scale, N, D1, rr, dd, (double*)(base + lay.slabs), &nslabs, st);
        if (rc != VT_OK) return rc;
        const int L1 = slab_len(NB1);
        hipLaunchKernelGGL(k_hess_finalize, dim3((L1 + 15) / 16), dim3(256), 0, st, (const double*)(base + lay.slabs), L1,
                           nslabs, NB1, D1, prior, theta, H, ldh, g);
        // rows 256.. of H: X2^T diag(d) [X1 | X2]
        rc = vt_gemm_tn(X + D1, ldx, t, X, ldx, D, dd, N, 1.0, nullptr, H + (int64_t)D1 * ldh, ldh, base + lay.tn,
                        vt_gemm_tn_workspace_bytes(t, D), stream);
        if (rc != VT_OK) return rc;
        rc = vt_gemv_t2(X + D1, ldx, rr, w, N, t, g2, rows_ws, rows_bytes, stream);
        if (rc != VT_OK) return rc;
        hipLaunchKernelGGL(k_split_finish, dim3(64), dim3(256), 0, st, H, ldh, D1, D, prior, theta, g ? g : nullptr);
        return vt_launch_status();
    }
    if (N > 0 && vt_hess_mid_ok(X, ldx, D)) {
        // 129 <= D <= 256: one fused pass (vt_hess_mid.hip), slabs summed by the same finalize launch
        const int NBm = (D + 15) / 16;
        int nslabs = 0;
        int rc = vt_hess_mid(family, X, ldx, theta, y, w, offset, scale, N, D, r, d, (double*)ws, &nslabs, st);
        if (rc != VT_OK) return rc;
        const int Lm = slab_len(NBm);
        hipLaunchKernelGGL(k_hess_finalize, dim3((Lm + 15) / 16), dim3(256), 0, st, (const double*)ws, Lm, nslabs, NBm, D,
                           prior, theta, H, ldh, g);
        return vt_launch_status();
    }
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
    // dev switches, read once per process: VT_HESS_NO_TILE=1 sends every shape to the register-streaming kernels,
    // VT_HESS_NO_STREAM=1 sends the narrow designs to k_glm_hess_small (the round-1 kernel: every row linked 16 times)
    static const bool no_tile = getenv("VT_HESS_NO_TILE") != nullptr;
    static const bool no_stream = getenv("VT_HESS_NO_STREAM") != nullptr;
    static const int stream_wgs_env = getenv("VT_HESS_STREAM_WGS") ? atoi(getenv("VT_HESS_STREAM_WGS")) : 0;
    const int stream_wgs = stream_wgs_env > 0 ? stream_wgs_env : 1;   // one 8-wave workgroup is resident per CU (registers)
    const bool tiled = N >= kTileMinRows && NB >= 4 && NB <= 8 && (ldx % 2 == 0) && ((uintptr_t)X % 16 == 0) && !no_tile;
    const int64_t ntiles = (N + tile_rows(NB) - 1) / tile_rows(NB);
    const int grid = tiled ? (int)std::min<int64_t>(ntiles, num_cu()) : hess_grid(N);
    double* part = (double*)ws;
    double* part2 = part + (size_t)num_cu() * L;
    double* wave_part = part2 + (size_t)kStage2 * L;
    double* sink = wave_part + (size_t)num_cu() * 4 * (NB * (NB + 1) / 2) * 256;

    if (N == 0) {   // nothing to stream: the finalize kernel sums zero slabs and adds the prior
        hipLaunchKernelGGL(k_hess_finalize, dim3((L + 15) / 16), dim3(256), 0, st, (const double*)part, L, 0, NB, D,
                           prior, theta, H, ldh, g);
        return vt_launch_status();
    }
    static const bool no_rowlink5 = getenv("VT_HESS_NO_ROWLINK") != nullptr;
    if (NB <= kStreamNB && !no_stream && !(NB >= 5 && (no_rowlink5 || ldx > ((int64_t)1 << 22)))) {
        // narrow design: HBM-bound; the slabs of up to 4 workgroups per CU fit the [part | part2 | wave_part] region
        const int64_t nsteps = (N + 3) / 4;
        const int sgrid = (int)std::max<int64_t>(1, std::min<int64_t>((nsteps + 31) / 32,
                                                                       (int64_t)num_cu() * std::min(std::max(stream_wgs, 1), 4)));
        HessStreamArgs<double> sa{family, X, ldx, theta, y, w, offset, scale, N, D, r, d, part, sink};
        int rc = VT_ERR_UNSUPPORTED;
        static const bool no_rowlink = getenv("VT_HESS_NO_ROWLINK") != nullptr;   // dev A/B switch
        // (the row-link kernel addresses 64 rows through one buffer descriptor with 32-bit offsets)
        if (!no_rowlink && ldx <= ((int64_t)1 << 22)) {
            // super-groups of 64 (NB <= 2) or 32 rows, one 8-wave workgroup per CU
            static const int rl_wgs_env = getenv("VT_RL_WGS") ? atoi(getenv("VT_RL_WGS")) : 0;   // dev switch
            const int rl_wgs = rl_wgs_env > 0 ? std::min(rl_wgs_env, 4) : 1;
            const int rgrid = rl_grid(N, NB, num_cu(), rl_wgs);
            switch (NB) {
                case 1: rc = rl_launch<double, 1>(sa, rgrid, st); break;
                case 2: rc = rl_launch<double, 2>(sa, rgrid, st); break;
                case 3: rc = rl_launch<double, 3>(sa, rgrid, st); break;
                case 4: rc = rl_launch<double, 4>(sa, rgrid, st); break;
                case 5: rc = rl_launch<double, 5>(sa, rgrid, st); break;
#if VT_STREAM_NB >= 6
                case 6: rc = rl_launch<double, 6>(sa, rgrid, st); break;
#endif
            }
            if (rc != VT_OK) return rc;
            hipLaunchKernelGGL(k_hess_finalize, dim3((L + 15) / 16), dim3(256), 0, st, (const double*)part, L, rgrid, NB, D,
                               prior, theta, H, ldh, g);
            return vt_launch_status();
        }
        switch (NB) {
            case 1: rc = hs_launch<double, 1, 8>(sa, sgrid, st); break;
            case 2: rc = hs_launch<double, 2, 8>(sa, sgrid, st); break;
            case 3: rc = hs_launch<double, 3, 8>(sa, sgrid, st); break;
            case 4: rc = hs_launch<double, 4, 8>(sa, sgrid, st); break;
        }
        if (rc != VT_OK) return rc;
        hipLaunchKernelGGL(k_hess_finalize, dim3((L + 15) / 16), dim3(256), 0, st, (const double*)part, L, sgrid, NB, D,
                           prior, theta, H, ldh, g);
        return vt_launch_status();
    }
    HessArgs a{family, X, ldx, theta, y, w, offset, scale, N, D, r, d, part, wave_part, sink};
    if (tiled) {
        int rc = VT_ERR_UNSUPPORTED;
        switch (NB) {
#ifdef VT_HESS_DEV_NB   // dev builds (tools/build_abl.cmd): one block count only, to keep the compile short
            case VT_HESS_DEV_NB: rc = launch_hess_tile<VT_HESS_DEV_NB>(a, grid, st); break;
#else
            case 4: rc = launch_hess_tile<4>(a, grid, st); break;
            case 5: rc = launch_hess_tile<5>(a, grid, st); break;
            case 6: rc = launch_hess_tile<6>(a, grid, st); break;
            case 7: rc = launch_hess_tile<7>(a, grid, st); break;
            case 8: rc = launch_hess_tile<8>(a, grid, st); break;
#endif
        }
        if (rc != VT_OK) return rc;
    } else switch (NB) {
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
    hipLaunchKernelGGL(k_hess_finalize, dim3((L + 15) / 16), dim3(256), 0, st, (const double*)part, L, grid, NB, D,
                       prior, theta, H, ldh, g);
    return vt_launch_status();
}
