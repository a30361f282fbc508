// K0 + K1 + K3 fused for MID-RANGE designs, 129 <= D <= 256 (NB = ceil(D/16) = 9..16 column blocks):
//   H = X^T diag(d) X,  g = X^T (w r),  r, d per row  in ONE pass over X.
// Replaces autograd.hessian / autograd.grad of the reference (vittles/sensitivity_lib.py:354, 381-383), like
// vt_hessian.hip -- see include/vittles_hip.h (vt_glm_hessian).
//
// Why a kernel of its own.  Up to D = 128 the tile kernel (vt_hessian.hip) gives every SIMD a quarter of the
// block triangle and lets two MFMA waves split the k-steps of it; beyond, a quarter of the triangle (34 blocks
// = 272 accumulator registers at D = 256) no longer fits two waves per SIMD, and the 128 x 128 tiled SYRK
// (vt_large.hip) paid a full tile for every partial one and read X three times (round 2: D = 129 ran 7x
// slower than D = 128).  Here:
//   * 8 waves per CU (two per SIMD, 256 registers each) are 8 OWNERS of the block triangle: the blocks are
//     walked column by column and cut into 8 runs of equal cost (make_mid_tab), so an owner touches at most
//     three block columns (B operands) and a contiguous run of rows, and holds 6 (D = 129) to 19 (D = 256)
//     accumulators.  No two waves share a block: each stores its accumulators straight to the workgroup's
//     slab in the workspace (no reduction through LDS).
//   * X goes HBM -> LDS by buffer_load ... lds (rows zero-filled past D by the descriptor's range check),
//     two periods ahead, three tile buffers in rotation, exactly as in the tile kernel; y, w, offset ride along.
//   * the LINK of the next tile (row dot by DPP, the exp / rcp chain once per row on one lane per row, gradient,
//     r_n and d_n out) is a long dependent VALU chain whose every instruction queues behind the fp64 MFMAs of the
//     SIMD.  SVC = true (NB <= kMidSvcNB: the owners fit 168 registers): 12 waves per CU, the third wave of every
//     SIMD is a SERVICE wave that only links (as in the tile kernel), so the chain's latency is off the MFMA waves'
//     path.  SVC = false (more accumulators per owner than three waves per SIMD leave room for): the two waves
//     of a SIMD have two roles, wave A issues the SIMD's share of the DMA, wave B links before its MFMAs of the
//     period and owns VT_MID_LINK_DELTA pipe cycles per k-step less than A (measured at N = 2e6: the link costs
//     about 1.8 us per period this way, D = 129 1.12 ms against 0.79 ms at D = 128).
//   * the last block row is computed with 4x4x4 MFMA strips when D mod 16 <= 8 (as in the tile kernel), so
//     D = 129 costs 8 full block rows + one 4-row strip, not 9 block rows.
#include "vt_common.h"
#include <algorithm>
#include <cstdlib>

#ifndef VT_MID_LINK_DELTA
#define VT_MID_LINK_DELTA 80
#endif
#ifndef VT_MABL
#define VT_MABL 0   // dev-only compile-time ablation mask: 1 no MFMA, 2 no DMA, 4 no link
#endif

namespace {

struct MidArgs {
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
    double* part;   // [gridDim.x][L] slabs, L = NBT*256 + NB*16 (the layout k_hess_finalize sums)
};

typedef unsigned mid_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* mid_ldsptr;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mid_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}

__host__ __device__ constexpr int mid_rs(int NB) { return (NB & 1) ? 16 * NB : 16 * NB + 16; }   // 16 mod 32 doubles
__host__ __device__ constexpr int mid_rows(int NB) { return NB <= 11 ? 32 : 24; }                // 3 buffers fit 160 KB
__host__ __device__ constexpr int mid_strips(int D) { return (D % 16 >= 1 && D % 16 <= 8) ? (D % 16 + 3) / 4 : 0; }
__host__ __device__ constexpr int mid_cost(int NB, int NS, int i) { return (NS > 0 && i == NB - 1) ? 16 * NS : 64; }
// position of block (i, j), i >= j, in the column-by-column walk of the lower block triangle
__host__ __device__ constexpr int mid_ord(int NB, int i, int j) { return j * NB - j * (j - 1) / 2 + (i - j); }

// Ownership: the walk is cut into 8 runs, run k going to the wave of type k & 1 (0 = A: DMA, 1 = B: link) on SIMD
// k >> 1; a B run is VT_MID_LINK_DELTA pipe cycles per k-step shorter than an A run.  Owner index O = 4 type + SIMD.
struct MidTab {
    unsigned long long own[8][3];   // bit mid_ord(i, j) of owner O
    unsigned rows[8];               // bit i: owns a FULL block of block row i (needs the A operand of row i)
    unsigned cols[8];               // bit j: owns a block of block column j (needs the scaled B operand)
    bool strip[8];                  // owns a block of the strip row
    int nown[8];
};
__host__ __device__ constexpr MidTab make_mid_tab(int NB, int NS, int delta_in) {
    MidTab t{};
    int total = 0;
    for (int j = 0; j < NB; ++j)
        for (int i = j; i < NB; ++i) total += mid_cost(NB, NS, i);
    int delta = delta_in;
    if (4 * delta > total / 2) delta = total / 8;
    int thr[8] = {};   // cumulative targets, in units of 1/8 pipe cycle
    int cum = 0;
    for (int k = 0; k < 8; ++k) {
        cum += total + ((k & 1) ? -4 * delta : 4 * delta);
        thr[k] = cum;
    }
    int pos = 0;
    for (int j = 0; j < NB; ++j)
        for (int i = j; i < NB; ++i) {
            const int c = mid_cost(NB, NS, i);
            const int mid = 8 * pos + 4 * c;
            int k = 0;
            while (k < 7 && mid >= thr[k]) ++k;
            const int O = 4 * (k & 1) + (k >> 1);
            const int ord = mid_ord(NB, i, j);
            t.own[O][ord >> 6] |= 1ull << (ord & 63);
            t.cols[O] |= 1u << j;
            if (NS > 0 && i == NB - 1) t.strip[O] = true;
            else t.rows[O] |= 1u << i;
            t.nown[O] += 1;
            pos += c;
        }
    return t;
}
__host__ __device__ constexpr bool mid_bit(unsigned long long w0, unsigned long long w1, unsigned long long w2, int ord) {
    return ord < 64 ? ((w0 >> ord) & 1ull) != 0 : ord < 128 ? ((w1 >> (ord - 64)) & 1ull) != 0 : ((w2 >> (ord - 128)) & 1ull) != 0;
}
__host__ __device__ constexpr int mid_rank(unsigned long long w0, unsigned long long w1, unsigned long long w2, int ord) {
    if (ord < 64) return __builtin_popcountll(w0 & ((1ull << ord) - 1ull));
    if (ord < 128) return __builtin_popcountll(w0) + __builtin_popcountll(w1 & ((1ull << (ord - 64)) - 1ull));
    return __builtin_popcountll(w0) + __builtin_popcountll(w1) + __builtin_popcountll(w2 & ((1ull << (ord - 128)) - 1ull));
}

template <int NB>
struct MidLds {   // offsets in doubles
    static constexpr int RS = mid_rs(NB);
    static constexpr int TR = mid_rows(NB);
    static constexpr int kTile = 0;                       // [3][TR][RS]
    static constexpr int kD = 3 * TR * RS;                // [3][TR]  d_n of the tile
    static constexpr int kTheta = kD + 3 * TR;            // [NB*16]
    static constexpr int kScratch = kTheta + NB * 16;     // [4 SIMDs][32]: z and w_n r_n of the link wave's rows
    static constexpr int kY = kScratch + 4 * 32;          // [3][TR] y_n, w_n, offset_n of the tiles
    static constexpr int kW = kY + 3 * TR;
    static constexpr int kOff = kW + 3 * TR;
    static constexpr int kTotal = kOff + 3 * TR;
};

// Which rows of X a workgroup's k-th tile holds (tiles are dealt round-robin over the grid).
template <int TR>
struct MidRows {
    const MidArgs& a;
    int64_t mine;
    __device__ __forceinline__ int64_t first_row(int64_t k) const { return (blockIdx.x + k * (int64_t)gridDim.x) * (int64_t)TR; }
    __device__ __forceinline__ int nrow(int64_t k) const {
        const int64_t left = (k >= 0 && k < mine) ? a.N - first_row(k) : 0;
        return (int)(left < TR ? (left > 0 ? left : 0) : TR);
    }
};

// LINK of the RPS rows [RPS*simd, RPS*(simd+1)) of a tile (a service wave, or wave B of the SIMD).
template <int NB, bool LOWREG>
struct MidLink {
    using LY = MidLds<NB>;
    static constexpr int RS = LY::RS, TR = LY::TR, RPS = TR / 4, PASSES = (RPS + 3) / 4;
    // gradient: lane l accumulates columns l, l + 64, ... over the rows this wave links (4 accumulators instead of
    // the NB a (row group, column) lane would need: the B waves have no registers to spare)
    static constexpr int CPL = (NB * 16 + 63) / 64;
    double gacc[CPL];

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int m = 0; m < CPL; ++m) gacc[m] = 0.0;
    }
    __device__ __forceinline__ void link(const MidArgs& a, double* lds, const MidRows<TR>& tiles, int simd, int64_t k, int buf) {
        const int lane = threadIdx.x & 63;
        const int q = lane >> 4, c = lane & 15;
        const double* th = lds + LY::kTheta;
        double* zs = lds + LY::kScratch + simd * 32;
        const int trow = RPS * simd + (lane & 15);
        const bool tlane = lane < RPS;
        const int trow_c = trow < TR ? trow : TR - 1;
        const int nrow = tiles.nrow(k);
        const int64_t rb = nrow > 0 ? tiles.first_row(k) : 0;
        // stage 1: z_n = x_n . theta (16 lanes per row)
#pragma unroll
        for (int k2 = 0; k2 < PASSES; ++k2) {
            const int rl = min(4 * k2 + q, RPS - 1);
            const double* tr = lds + LY::kTile + buf * (TR * RS) + (RPS * simd + rl) * RS + c;
            double pz0 = 0.0, pz1 = 0.0;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (!LOWREG && (b & 1)) pz1 = fma(tr[16 * b], th[16 * b + c], pz1);
                else pz0 = fma(tr[16 * b], th[16 * b + c], pz0);
                // (4 blocks of loads in flight at a time: hoisted to the top, the 2 NB operands of every pass cost
                // the B waves more registers than their accumulators)
                if (LOWREG && (b & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            const double z = vt_sum16_dpp(pz0 + pz1);
            if (c == 0) zs[4 * k2 + q] = z;
        }
        // stage 2: the scalar chain, one lane per row
        {
            const double sy = lds[LY::kY + buf * TR + trow_c];
            const double sw = lds[LY::kW + buf * TR + trow_c];
            const double z = zs[lane & 15] + lds[LY::kOff + buf * TR + trow_c];
            double l1, l2;
            vt_link(a.family, z, sy, l1, l2);
            const bool rowok = tlane && trow < nrow;
            const double rr = a.scale * l1;
            const double dw = rowok ? a.scale * sw * l2 : 0.0;
            const double gwt = rowok ? sw * rr : 0.0;
            const unsigned so = tlane ? 8u * trow : 0xfffffff0u;    // other lanes: dropped by the range check
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(mid_u32x2, rr),
                                                  mid_rsrc(a.r_out + rb, a.r_out ? 8u * nrow : 0u), so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(mid_u32x2, dw),
                                                  mid_rsrc(a.d_out + rb, a.d_out ? 8u * nrow : 0u), so, 0, 0);
            if (tlane) lds[LY::kD + buf * TR + trow] = dw;
            if (lane < 16) zs[16 + lane] = gwt;
        }
        // stage 3: gradient, g[col] += (w_n r_n) x[n][col] (the rows are read again, column-wise; columns past D are
        // zero in the tile, lanes past the row's stride read what follows it and are never stored)
#pragma unroll
        for (int rr = 0; rr < RPS; ++rr) {
            const double gwt = zs[16 + rr];
            const double* tr = lds + LY::kTile + buf * (TR * RS) + (RPS * simd + rr) * RS + lane;
#pragma unroll
            for (int m = 0; m < CPL; ++m) gacc[m] = fma(gwt, tr[64 * m], gacc[m]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // the wave's column sums -> its slot of gl [4][NB*16] (after every tile is dead)
    __device__ __forceinline__ void store(double* gl, int simd) const {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int m = 0; m < CPL; ++m)
            if (64 * m + lane < NB * 16) gl[simd * (NB * 16) + 64 * m + lane] = gacc[m];
    }
};

// SERVICE wave (SVC kernels: waves 8..11, one per SIMD): links tile p + 1 in period p, nothing else.
template <int NB>
__device__ __forceinline__ void mid_service(const MidArgs& a, double* lds, int64_t mine, int simd) {
    using LY = MidLds<NB>;
    const MidRows<LY::TR> tiles{a, mine};
    MidLink<NB, false> lk;
    lk.init();
    vt_lds_barrier();                  // tile 0 has landed
    if (!(VT_MABL & 4)) lk.link(a, lds, tiles, simd, 0, 0);
    vt_lds_barrier();
    int b1 = 1;                        // buffer of tile p + 1
    for (int64_t p = 0; p < mine; ++p) {
        if (!(VT_MABL & 4)) lk.link(a, lds, tiles, simd, p + 1, b1);
        vt_lds_barrier();
        b1 = b1 == 2 ? 0 : b1 + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every tile is dead
    lk.store(lds, simd);
}

// Owner wave O (0..7) on SIMD O & 3.  O < 4 also issues the SIMD's share of the DMA; without service waves O >= 4
// links the SIMD's rows before its MFMAs.
template <int NB, int NS, int O, bool SVC>
__device__ __forceinline__ void mid_wave(const MidArgs& a, double* lds, int64_t mine) {
    using LY = MidLds<NB>;
    constexpr int RS = LY::RS, TR = LY::TR;
    constexpr bool DMA = O < 4, LINK = !SVC && O >= 4;
    constexpr int SIMD = O & 3;
    constexpr int KSTEPS = TR / 4;
    constexpr int RPS = TR / 4;                        // rows of a tile per SIMD (DMA share, link share)
    constexpr int CH = RS / 2;                         // 16-byte chunks per LDS row
    constexpr int DMAH = (CH + 63) / 64;               // DMA instructions per row
    constexpr MidTab TAB = make_mid_tab(NB, NS, SVC ? 0 : VT_MID_LINK_DELTA);
    constexpr unsigned long long W0 = TAB.own[O][0], W1 = TAB.own[O][1], W2 = TAB.own[O][2];
    constexpr unsigned NEED = TAB.rows[O] | TAB.cols[O], COLS = TAB.cols[O];
    constexpr bool STRIP = TAB.strip[O];
    constexpr int NOWN = TAB.nown[O];
    static_assert(2 * RPS <= 64, "one DMA instruction moves a SIMD's per-row scalars");

    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, c = lane & 15;
    const MidRows<TR> tiles{a, mine};

    // the RPS rows of my SIMD of tile k -> buffer `buf` (one descriptor per row, SALU only)
    auto dma = [&](int64_t k, int buf) {
        const int nrow = (VT_MABL & 2) ? 0 : tiles.nrow(k);
        const int64_t rb = nrow > 0 ? tiles.first_row(k) : 0;
        const double* xb = a.X + (rb + RPS * SIMD) * a.ldx;
#pragma unroll
        for (int rr = 0; rr < RPS; ++rr) {
            const int row = RPS * SIMD + rr;
            __amdgpu_buffer_rsrc_t xrs = mid_rsrc(xb + (int64_t)rr * a.ldx, row < nrow ? 8u * a.D : 0u);
            double* dst = lds + LY::kTile + buf * (TR * RS) + row * RS;
#pragma unroll
            for (int h = 0; h < DMAH; ++h) {
                const int chunk = 64 * h + lane;
                if (chunk < CH)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (mid_ldsptr)(dst + 128 * h), 16, 16u * chunk, 0, 0, 0);
            }
        }
        if (lane < 2 * RPS) {
            const unsigned vo = 8u * (RPS * SIMD) + 4u * lane;
            const int so = buf * TR + RPS * SIMD;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(mid_rsrc(a.y + rb, 8u * nrow), (mid_ldsptr)(lds + LY::kY + so), 4, vo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(mid_rsrc(a.w + rb, 8u * nrow), (mid_ldsptr)(lds + LY::kW + so), 4, vo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(mid_rsrc(a.offset + rb, a.offset ? 8u * nrow : 0u),
                                                     (mid_ldsptr)(lds + LY::kOff + so), 4, vo, 0, 0, 0);
        }
    };

    MidLink<NB, true> lk;
    lk.init();

    // ---- accumulators: one f64x4 per owned block (a strip block only touches components 0..NS-1) ----
    f64x4 acc[NOWN > 0 ? NOWN : 1];
#pragma unroll
    for (int i = 0; i < (NOWN > 0 ? NOWN : 1); ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const unsigned strip_lane = 8u * (unsigned)(c & 3) - 8u * (unsigned)c;   // byte offset: column c -> column c & 3

    // period -1 (peeled, so that the accumulators live in ONE loop nest: with the `p >= 0` test inside the loop
    // hipcc kept a second copy of them for the first iteration and spilled): tile 0 lands, then tile 1 is
    // fetched while tile 0 is linked
    if (DMA) {
        dma(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    vt_lds_barrier();
    if (DMA) dma(1, 1);
    if (LINK && !(VT_MABL & 4)) lk.link(a, lds, tiles, SIMD, 0, 0);
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    vt_lds_barrier();
    int b0 = 0;                        // buffer of tile p
    int b1 = 1;                        // buffer of tile p + 1
    for (int64_t p = 0; p < mine; ++p) {
        if (DMA) dma(p + 2, b0 == 0 ? 2 : b0 - 1);   // buffer (p + 2) % 3 == buffer of tile p - 1
        if (LINK && !(VT_MABL & 4)) lk.link(a, lds, tiles, SIMD, p + 1, b1);
        if (NOWN > 0 && !(VT_MABL & 1)) {
            // operand reads are inline asm: as compiler-visible LDS loads they make hipcc wait for this wave's own
            // LDS-DMA (vmcnt(0): it cannot tell the tile buffers apart)
            unsigned ta = (unsigned)(uintptr_t)(mid_ldsptr)(lds + LY::kTile + b0 * (TR * RS) + q * RS + c);
            unsigned da = (unsigned)(uintptr_t)(mid_ldsptr)(lds + LY::kD + b0 * TR + q);
#pragma nounroll
            for (int m = 0; m < KSTEPS; ++m) {
                double xa[NB];
                double xs[2] = {0.0, 0.0};
                double dq;
                asm volatile("ds_read_b64 %0, %1" : "=v"(dq) : "v"(da));
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if ((NEED >> b) & 1)
                        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xa[b]) : "v"(ta), "n"(128 * b));
                    else
                        xa[b] = 0.0;
                }
                if constexpr (STRIP) {
                    const unsigned tas = ta + strip_lane;
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2)
                        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xs[s2]) : "v"(tas), "n"(128 * (NB - 1) + 32 * s2));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dq));
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if ((NEED >> b) & 1) asm volatile("" : "+v"(xa[b]));
                if constexpr (STRIP) {
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2) asm volatile("" : "+v"(xs[s2]));
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if (!((COLS >> j) & 1)) continue;
                    const double bj = dq * xa[j];
#pragma unroll
                    for (int i = j; i < NB; ++i)
                        if (mid_bit(W0, W1, W2, mid_ord(NB, i, j))) {
                            f64x4& ac = acc[mid_rank(W0, W1, W2, mid_ord(NB, i, j))];
                            if (NS > 0 && i == NB - 1) {
#pragma unroll
                                for (int s2 = 0; s2 < NS; ++s2)
                                    ac[s2] = __builtin_amdgcn_mfma_f64_4x4x4f64(xs[s2], bj, ac[s2], 0, 0, 0);
                            } else {
                                ac = mfma_f64(xa[i], bj, ac);
                            }
                        }
                }
                ta += 4 * RS * 8;
                da += 4 * 8;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // my DMA of tile p + 2 has had the whole period: landed before the barrier, so it can be linked next period
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        vt_lds_barrier();
        b0 = b0 == 2 ? 0 : b0 + 1;
        b1 = b1 == 2 ? 0 : b1 + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every tile is dead
    if (LINK) lk.store(lds, SIMD);
    // H: every block has ONE owner, so the accumulators go straight to the workgroup's slab (accumulator layout,
    // block order tri(i, j): what k_hess_finalize reads)
    constexpr int NBT = NB * (NB + 1) / 2;
    double* out = a.part + (int64_t)blockIdx.x * (NBT * 256 + NB * 16);
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = j; i < NB; ++i)
            if (mid_bit(W0, W1, W2, mid_ord(NB, i, j))) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool live = !(NS > 0 && i == NB - 1) || r < NS;   // strip rows past the last strip: padding
                    out[(vt_tri(i, j) * 4 + r) * 64 + lane] = live ? acc[mid_rank(W0, W1, W2, mid_ord(NB, i, j))][r] : 0.0;
                }
            }
}

template <int NB, int NS, bool SVC>
__global__ __launch_bounds__(SVC ? 768 : 512) void k_glm_hess_mid(MidArgs a) {
    using LY = MidLds<NB>;
    constexpr int NBT = NB * (NB + 1) / 2;
    constexpr int NT = SVC ? 768 : 512;
    extern __shared__ double lds[];
    for (int t = threadIdx.x; t < NB * 16; t += NT) lds[LY::kTheta + t] = t < a.D ? a.theta[t] : 0.0;
    for (int t = threadIdx.x; t < 4 * 32; t += NT) lds[LY::kScratch + t] = 0.0;
    __syncthreads();
    const int64_t ntiles = (a.N + LY::TR - 1) / LY::TR;
    const int64_t mine = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef VT_MID_ONLYO
    mid_wave<NB, NS, VT_MID_ONLYO, SVC>(a, lds, mine);
#else
    switch (wid) {   // owner = wave (waves go to the SIMDs in cyclic order)
        case 0: mid_wave<NB, NS, 0, SVC>(a, lds, mine); break;
        case 1: mid_wave<NB, NS, 1, SVC>(a, lds, mine); break;
        case 2: mid_wave<NB, NS, 2, SVC>(a, lds, mine); break;
        case 3: mid_wave<NB, NS, 3, SVC>(a, lds, mine); break;
        case 4: mid_wave<NB, NS, 4, SVC>(a, lds, mine); break;
        case 5: mid_wave<NB, NS, 5, SVC>(a, lds, mine); break;
        case 6: mid_wave<NB, NS, 6, SVC>(a, lds, mine); break;
        case 7: mid_wave<NB, NS, 7, SVC>(a, lds, mine); break;
        default: mid_service<NB>(a, lds, mine, wid & 3); break;
    }
#endif
    __syncthreads();
    double* out = a.part + (int64_t)blockIdx.x * (NBT * 256 + NB * 16);
    for (int t = threadIdx.x; t < NB * 16; t += NT) {
        double s2 = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) s2 += lds[w * (NB * 16) + t];
        out[NBT * 256 + t] = s2;
    }
}

template <int NB>
constexpr size_t mid_lds_bytes() { return (size_t)MidLds<NB>::kTotal * sizeof(double); }

#ifndef VT_MID_SVC_NB
#define VT_MID_SVC_NB 13
#endif
constexpr int kMidSvcNB = VT_MID_SVC_NB;   // largest NB whose owners fit the 168 registers of three waves per SIMD

template <int NB, int NS>
int launch_mid2(const MidArgs& a, int grid, hipStream_t st) {
    constexpr bool SVC = NB <= kMidSvcNB;
    static VtLdsAttr attr;   // per kernel instantiation, per device
    if (vt_allow_big_lds((const void*)k_glm_hess_mid<NB, NS, SVC>, attr, 160 * 1024) != VT_OK) return VT_ERR_HIP;
    hipLaunchKernelGGL((k_glm_hess_mid<NB, NS, SVC>), dim3(grid), dim3(SVC ? 768 : 512), mid_lds_bytes<NB>(), st, a);
    return VT_OK;
}

template <int NB>
int launch_mid(const MidArgs& a, int grid, hipStream_t st) {
    switch (mid_strips(a.D)) {
        case 1: return launch_mid2<NB, 1>(a, grid, st);
        case 2: return launch_mid2<NB, 2>(a, grid, st);
        default: return launch_mid2<NB, 0>(a, grid, st);
    }
}

}  // namespace

bool vt_hess_mid_ok(const double* X, int64_t ldx, int D) {
    static const bool off = getenv("VT_HESS_NO_MID") != nullptr;   // dev A/B switch, read once
    // dev switch: the smallest D the owner-wave kernel takes (default: above what the 12-wave tile kernel covers)
    static const int dmin = getenv("VT_HESS_MID_MIN_D") ? atoi(getenv("VT_HESS_MID_MIN_D")) : VT_SMALL_D_MAX + 1;
    return !off && D >= std::max(dmin, 113) && D <= VT_MID_D_MAX && ldx % 2 == 0 && ((uintptr_t)X % 16 == 0) &&
           ldx <= ((int64_t)1 << 27);
}

size_t vt_hess_mid_ws_bytes(int D) {
    if (D < 113 || D > VT_MID_D_MAX) return 0;
    const int NB = (D + 15) / 16;
    return (size_t)vt_num_cu() * (NB * (NB + 1) / 2 * 256 + NB * 16) * sizeof(double);
}

// One pass over X: per-workgroup slabs of [H blocks | g] in `part`; returns the number of slabs in *nslabs.
int vt_hess_mid(int family, const double* X, int64_t ldx, const double* theta, const double* y, const double* w,
                const double* offset, double scale, int64_t N, int D, double* r, double* d, double* part,
                int* nslabs, hipStream_t st) {
    const int NB = (D + 15) / 16;
    const int TR = mid_rows(NB);
    const int64_t ntiles = (N + TR - 1) / TR;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ntiles, vt_num_cu()));
    MidArgs a{family, X, ldx, theta, y, w, offset, scale, N, D, r, d, part};
    int rc = VT_ERR_UNSUPPORTED;
    switch (NB) {
#ifdef VT_MID_DEVNB
        case VT_MID_DEVNB: rc = launch_mid<VT_MID_DEVNB>(a, grid, st); break;
#else
        case 8: rc = launch_mid<8>(a, grid, st); break;
        case 9: rc = launch_mid<9>(a, grid, st); break;
        case 10: rc = launch_mid<10>(a, grid, st); break;
        case 11: rc = launch_mid<11>(a, grid, st); break;
        case 12: rc = launch_mid<12>(a, grid, st); break;
        case 13: rc = launch_mid<13>(a, grid, st); break;
        case 14: rc = launch_mid<14>(a, grid, st); break;
        case 15: rc = launch_mid<15>(a, grid, st); break;
        case 16: rc = launch_mid<16>(a, grid, st); break;
#endif
    }
    if (rc != VT_OK) return rc;
    *nslabs = grid;
    return vt_launch_status();
}
