// Large-D (D > 128) dense path: tiled fp64-MFMA building blocks and the host-side blocked
// algorithms on top of them (BASELINE.json configs[2]: D = 1000).
//
//   k_syrk_tile   H tile (128x128) += X[:, I]^T diag(d) X[:, J] over a chunk of rows (split-K over N),
//                 operands loaded straight from global memory in MFMA layout (re-reads hit L2/MALL).
//   k_gemm_nt     C = beta C + alpha A B^T   (A: M x K, B: N x K, both row-major, K contiguous);
//                 the k index inside a group of 16 is permuted so that a lane owns 4 consecutive k
//                 (32-byte loads) -- A and B use the same permutation, so the product is unchanged.
//   blocked Cholesky (128-wide panels): diagonal block by k_potrf_pack_small, panel by the MFMA
//                 TRSM in forward-only mode, trailing update by k_gemm_nt.
//   blocked TRSM  S = alpha s .* B (L L^T)^{-1}: per 128-column block one GEMM-NT update (whose epilogue
//                 also applies the row scale: alpha s .* B is never materialised) and one small-TRSM
//                 solve with the block's inverted-diagonal pack.
// Replaces the same reference lines as the small-D kernels (see include/vittles_hip.h).
#include "vt_common.h"
#include <cstdlib>

#ifndef VT_LABL
#define VT_LABL 0   // dev-only compile-time ablation mask of k_gemm_nt_lds: 1 no DMA after the prologue, 2 no MFMA
#endif

namespace {

constexpr int kT = 128;            // tile / panel width
constexpr int kPackStride = 2 * 36 * 256;   // doubles per diagonal-block pack (NB = 8)

inline int num_cu() { return vt_num_cu(); }   // of the current device

// ------------------------------------------------------------------ GEMM NT -------
struct GemmArgs {
    double* C;
    int64_t ldc;
    int64_t M;
    int N;
    const double* A;
    int64_t lda;
    const double* B;
    int64_t ldb;
    int K;
    double alpha, beta;
    // optional: take the beta term from another matrix, row-scaled:  C = alpha A B^T + cscale * rowscale[m] * Csrc[m][n]
    const double* Csrc;
    int64_t ldcs;
    const double* rowscale;
    double cscale;
};

// 4 consecutive k of one row: vector loads when the group is inside [0, K), zero otherwise.
template <bool VEC>
__device__ __forceinline__ f64x4 load_k4(const double* row, int k, int K) {
    if constexpr (VEC) {
        int kc = min(k, K - 4);
        f64x2u lo = *(const f64x2u*)(row + kc);
        f64x2u hi = *(const f64x2u*)(row + kc + 2);
        f64x4 v = {lo[0], lo[1], hi[0], hi[1]};
        return k < K ? v : (f64x4){0.0, 0.0, 0.0, 0.0};
    } else {
        f64x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double x = row[min(k + e, K - 1)];
            v[e] = (k + e < K) ? x : 0.0;
        }
        return v;
    }
}

template <bool VEC, bool LOWER>
__global__ __launch_bounds__(256) void k_gemm_nt(GemmArgs a) {
    if (LOWER && blockIdx.y > blockIdx.x) return;   // tile strictly above the diagonal
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int64_t m_base = (int64_t)blockIdx.x * kT + (wave >> 1) * 64;
    const int n_base = blockIdx.y * kT + (wave & 1) * 64;
    if (m_base >= a.M || n_base >= a.N) return;

    const double* arow[4];
    const double* brow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t m = m_base + 16 * i + c;
        arow[i] = a.A + (m < a.M ? m : a.M - 1) * a.lda;
        int n = n_base + 16 * i + c;
        brow[i] = a.B + (int64_t)(n < a.N ? n : a.N - 1) * a.ldb;
    }
    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};

    for (int k0 = 0; k0 < a.K; k0 += 16) {
        f64x4 af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i] = load_k4<VEC>(arow[i], k0 + 4 * q, a.K);
            bf[i] = load_k4<VEC>(brow[i], k0 + 4 * q, a.K);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma_f64(af[i][s], bf[j][s], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = n_base + 16 * j + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int64_t m = m_base + 16 * i + q + 4 * r;
                if (m < a.M && n < a.N) {
                    double* p = a.C + m * a.ldc + n;
                    double v = a.alpha * acc[i][j][r];
                    if (a.Csrc) v = fma(a.cscale * (a.rowscale ? a.rowscale[m] : 1.0), a.Csrc[m * a.ldcs + n], v);
                    else if (a.beta != 0.0) v = fma(a.beta, *p, v);
                    *p = v;
                }
            }
        }
}

// ------------------------------------------------------------------ LDS-staged kernels -------
// Same products, operands staged through LDS by buffer_load_dwordx4 ... lds (no registers, no VALU:
// VALU instructions are not free next to fp64 MFMAs, tools/micro/mfma_valu_overlap.hip), three
// stages of 16 k in rotation, 8 waves (2 per SIMD) of 64x32 accumulators per 128x128 tile.
typedef __attribute__((address_space(3))) void* vt_ldsptr;
constexpr int kKC = 16;                        // k per stage

__device__ __forceinline__ __amdgpu_buffer_rsrc_t lg_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ void lg_wait4_barrier() {
    // the stage about to be used has landed (only the newest 4 DMA instructions of this wave may
    // still be in flight), then everybody's part of it
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// C = beta C + alpha A B^T on 128 x 128 output tiles: 8 waves (2 per SIMD) of 64 x 32, two stages of 16 k (72 KB of
// LDS, so two workgroups share a CU and cover each other's barrier and DMA waits: +9-12 % over a 3-stage single
// workgroup).
//   * LINE-WISE DMA: one instruction moves 8 rows x 16 k -- lane 8 kp + r reads the kp-th 16-byte piece of row r, so
//     the 8 lanes of a row ask for ONE contiguous 128-byte line, once per stage.  (The first layout had every lane of
//     an instruction read 16 bytes of a DIFFERENT row and eight instructions of different waves come back for the other
//     pieces of each line: 64 line requests per KB moved; blocked TRSM at N = 1.25e6, D = 1000: 53.3 -> 49.5 ms.)
//     LDS image per 8-row block: [kp 8][row 8][2 k] = 1 KB, blocks 1152 bytes apart, so that the 32 lanes of a half
//     wave (16 rows x 2 k) hit 64 different banks when they read MFMA operands (lane = row, k).
//   * PERSISTENT over the row tiles of a tall product (the blocked TRSM's updates: M = N rows): the workgroup walks
//     tiles blockIdx.x, + gridDim.x, ... and issues stage 0 of the NEXT tile during the last stage of the current one,
//     so the epilogue's stores and the next prologue's DMA latency overlap instead of idling the MFMA pipe twice
//     per tile (K is as short as 128 for the first column blocks: 8 stages per tile).
// 256 x 128 tiles (8 waves of 64 x 64, or 16 waves of 64 x 32) were built and measured SLOWER (66.0 / 56.4 ms
// against 53.6 for the same layout): one workgroup per CU has nobody to fill the MFMA pipe while all its waves
// re-read their operands after the stage barrier.  DESIGN.md section 8.
constexpr int kRB = 144;   // doubles between the LDS images of consecutive 8-row blocks (1 KB + 128 B)
constexpr int kGemmStage = 2 * (kT / 8) * kRB;                          // doubles per stage (A panel + B panel)
constexpr int kGemmLdsBytes = 2 * kGemmStage * (int)sizeof(double);     // 73 728

template <bool LOWER>
__global__ __launch_bounds__(512) void k_gemm_nt_lds(GemmArgs a) {
    if (LOWER && blockIdx.y > blockIdx.x) return;   // (LOWER products are launched one tile per workgroup)
    extern __shared__ double lds[];            // [2 stages][A panel | B panel]
    constexpr int PA = (kT / 8) * kRB;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, c = lane & 15;
    const int n0 = blockIdx.y * kT;
    const int nrem = min(kT, a.N - n0);
    // Wave grid over the 128 x nrem tile (workgroup-uniform): 2 x 4 waves of 64 x 32; a remainder panel of at most
    // 64 / 32 columns is dealt as 4 x 2 waves of 32 x 32 / 8 x 1 waves of 16 x 32, so that every SIMD keeps work and
    // nobody multiplies the zero padding (round 2 computed the full 128 columns for a 1-column remainder).
    const int narrow = nrem > 64 ? 0 : nrem > 32 ? 1 : 2;
    const int wm = narrow == 0 ? wave >> 2 : narrow == 1 ? wave >> 1 : wave;
    const int wn = narrow == 0 ? wave & 3 : narrow == 1 ? wave & 1 : 0;
    const int rpw = 64 >> narrow;                                    // rows per wave
    const int ni = 4 >> narrow;                                      // 16-row blocks per wave
    const int nj = min(2, max(0, (nrem - 32 * wn + 15) >> 4));       // 16-column blocks per wave
    const int nk = (a.K + kKC - 1) / kKC;
    const int64_t ntiles = (a.M + kT - 1) / kT;

    // lane 8 kp + r of a DMA instruction: row r of the block, k pair kp
    const unsigned voffA = (unsigned)(((lane & 7) * a.lda + 2 * (lane >> 3)) * 8);
    const unsigned voffB = (unsigned)(((lane & 7) * a.ldb + 2 * (lane >> 3)) * 8);
    const int ktail = a.K - (nk - 1) * kKC;    // k of the last stage (16 unless K is not a multiple of 16)
    const bool lane_in_tail = 2 * (lane >> 3) < ktail;

    // this wave's 4 DMA instructions of stage t of row tile `tile`: ids wave*4 .. wave*4+3 of 32 (A first, then B)
    auto issue = [&](int64_t tile, int t, int buf) {
        const bool tail = t == nk - 1 && ktail < kKC;   // uniform
        const int64_t m0 = tile * kT;
        const int mrem = (int)min((int64_t)kT, a.M - m0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int id = wave * 4 + e;
            const bool isb = id >= kT / 8;
            const int blk = id & (kT / 8 - 1);
            const int rem = isb ? nrem : mrem;
            const int64_t ld = isb ? a.ldb : a.lda;
            const double* base = (isb ? a.B + (int64_t)n0 * a.ldb : a.A + m0 * a.lda) + t * kKC;
            // rows past the tile read 0 (range check); k past K: the lanes of those pieces are sent out of range
            const int kv = tail ? ktail : kKC;
            const unsigned bytes = (unsigned)(((int64_t)(rem - 1) * ld + kv) * 8);
            unsigned voff = isb ? voffB : voffA;
            if (tail) voff = lane_in_tail ? voff : 0x7ffffff0u;
            double* dst = lds + buf * kGemmStage + (isb ? PA : 0) + blk * kRB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(lg_rsrc(base, bytes), (vt_ldsptr)dst, 16, voff,
                                                     (unsigned)(blk * 8 * ld) * 8u, 0, 0);
        }
    };

    int64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    issue(tile, 0, 0);
    int buf = 0;
    while (true) {
        const int64_t next = tile + gridDim.x;
        f64x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};

        for (int t = 0; t < nk; ++t) {
            // the stage about to be used has landed, everybody's part of it; then the DMA of the stage after it
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (!(VT_LABL & 1)) {
                if (t + 1 < nk) issue(tile, t + 1, buf ^ 1);
                else if (next < ntiles) issue(next, 0, buf ^ 1);
            }
            // odd K: the 16-byte lane that holds column K - 1 also fetched column K (inside the descriptor's range
            // for every row but the last): the last stage zeroes that element of all 2 x 128 panel rows in LDS
            // (kept out of the k-step loop: a masked copy of the loop cost 86 VGPRs and the second workgroup per CU)
            if ((a.K & 1) && t == nk - 1) {
                if (threadIdx.x < 2 * kT) {
                    const int r = threadIdx.x & (kT - 1);
                    lds[buf * kGemmStage + (threadIdx.x >= kT ? PA : 0) + (r >> 3) * kRB + (ktail >> 1) * 16 + (r & 7) * 2 + 1] = 0.0;
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            // operand of lane (c = row, q = k) for k-step s: plane kp = 2 s + (q >> 1), element q & 1
            const double* pa = lds + buf * kGemmStage + ((rpw >> 3) * wm + (c >> 3)) * kRB + (q >> 1) * 16 + (c & 7) * 2 + (q & 1);
            const double* pb = lds + buf * kGemmStage + PA + (4 * wn + (c >> 3)) * kRB + (q >> 1) * 16 + (c & 7) * 2 + (q & 1);
            if (nj > 0) {   // (scalar branches: a remainder panel multiplies only its real blocks)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    double af[4], bf[2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i] = i < ni ? pa[s * 32 + 2 * kRB * i] : 0.0;
#pragma unroll
                    for (int j = 0; j < 2; ++j) bf[j] = pb[s * 32 + 2 * kRB * j];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            if (VT_LABL & 2) acc[i][j][0] += af[i] * bf[j]; else
                            if (i < ni && j < nj) acc[i][j] = mfma_f64(af[i], bf[j], acc[i][j]);
                        }
                }
            }
            buf ^= 1;
        }
        // epilogue of this tile (the next tile's first stage is in flight)
        const int64_t m0 = tile * kT;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (i >= ni || j >= nj) continue;
                const int n = n0 + 32 * wn + 16 * j + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t m = m0 + rpw * wm + 16 * i + q + 4 * r;
                    if (m < a.M && n < a.N) {
                        double* p = a.C + m * a.ldc + n;
                        double v = a.alpha * acc[i][j][r];
                        if (a.Csrc) v = fma(a.cscale * (a.rowscale ? a.rowscale[m] : 1.0), a.Csrc[m * a.ldcs + n], v);
                        else if (a.beta != 0.0) v = fma(a.beta, *p, v);
                        *p = v;
                    }
                }
            }
        if (next >= ntiles) break;
        tile = next;
    }
}

// dev switches, read ONCE per process (never on the hot path)
static bool large_no_lds() {
    static const bool v = getenv("VT_LARGE_NO_LDS") != nullptr;
    return v;
}

static int large_stages() {
    // SYRK stages, default 2: measured 48.6 vs 43.2 TFLOP/s (N = 1.25e6, D = 1000); VT_LARGE_STAGES=3 for A/B runs
    static const int st = getenv("VT_LARGE_STAGES") ? atoi(getenv("VT_LARGE_STAGES")) : 2;
    return st == 3 ? 3 : 2;
}

bool lds_path_ok(const double* A, int64_t lda, const double* B, int64_t ldb, int K) {
    // (the DMA addresses the 128 rows of a panel through one buffer descriptor with 32-bit offsets)
    (void)K;   // (odd K: the kernel masks the half-used last k pair)
    return lda % 2 == 0 && ldb % 2 == 0 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) &&
           lda <= ((int64_t)1 << 21) && ldb <= ((int64_t)1 << 21) && !large_no_lds();
}

int gemm_nt(double* C, int64_t ldc, int64_t M, int N, const double* A, int64_t lda, const double* B,
            int64_t ldb, int K, double alpha, double beta, bool lower, hipStream_t st,
            const double* Csrc = nullptr, int64_t ldcs = 0, const double* rowscale = nullptr, double cscale = 1.0) {
    if (M <= 0 || N <= 0) return VT_OK;
    GemmArgs a{C, ldc, M, N, A, lda, B, ldb, K, alpha, beta, Csrc, ldcs, rowscale, cscale};
    dim3 grid((unsigned)((M + kT - 1) / kT), (unsigned)((N + kT - 1) / kT));
    if (lds_path_ok(A, lda, B, ldb, K)) {
        static VtLdsAttr attr[2];
        if (vt_allow_big_lds((const void*)k_gemm_nt_lds<false>, attr[0], kGemmLdsBytes) != VT_OK ||
            vt_allow_big_lds((const void*)k_gemm_nt_lds<true>, attr[1], kGemmLdsBytes) != VT_OK)
            return VT_ERR_HIP;
        if (lower) {
            hipLaunchKernelGGL((k_gemm_nt_lds<true>), grid, dim3(512), kGemmLdsBytes, st, a);
        } else {
            // persistent over the row tiles: two workgroups per CU and column block
            // gx a multiple of 8: workgroup x + y gx runs on XCD (x + y gx) % 8, so the column tiles y of one row
            // tile x share an XCD and with it the L2 lines of the A rows they all read (VT_GEMM_NT_GX_ANY=1: dev A/B;
            // measured at D = 512 / 1000 / 2000: no change in wall time -- the product is not bound by that traffic)
            static const bool gx_any = getenv("VT_GEMM_NT_GX_ANY") != nullptr;
            unsigned gx = (unsigned)std::min<int64_t>(grid.x, std::max(1, 2 * num_cu() / (int)grid.y));
            if (!gx_any && gx >= 8) gx &= ~7u;
            hipLaunchKernelGGL((k_gemm_nt_lds<false>), dim3(gx, grid.y), dim3(512), kGemmLdsBytes, st, a);
        }
        return vt_launch_status();
    }
    bool vec = (K % 4 == 0) && K >= 4;
    if (vec && lower) hipLaunchKernelGGL((k_gemm_nt<true, true>), grid, dim3(256), 0, st, a);
    else if (vec) hipLaunchKernelGGL((k_gemm_nt<true, false>), grid, dim3(256), 0, st, a);
    else if (lower) hipLaunchKernelGGL((k_gemm_nt<false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_gemm_nt<false, false>), grid, dim3(256), 0, st, a);
    return vt_launch_status();
}

// ------------------------------------------------------------------ SYRK tiles -------
// (tile, rows) enumeration of a launch.
//   * XCD-AWARE: workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one), and every tile's
//     two 16 x 128 panels per stage are re-read by every other tile of its tile row / column.  Round 2 numbered the
//     workgroups tile-major, so the tiles that read the same rows of X sat on different XCDs: PMC at N = 1.25e6,
//     D = 1000 showed an L2 hit rate of 2 % and 82 GB through the fabric for a 10 GB X (3.4 TB/s: the kernel was bound
//     there, not on the MFMA pipe).  Now XCD x = b & 7 owns the x-th EIGHTH OF THE ROWS for every tile, so its
//     resident workgroups walk the same rows of X at the same time and share them through the XCD's L2.
//   * CLASSES: the tiles of the lower triangle fall into four classes of equal cost -- A: full off-diagonal, B: full
//     diagonal, C: off-diagonal of a partial last tile row, D: the partial last diagonal tile -- and a class's tiles
//     split their XCD's rows s[c] ways, in proportion to what one k-step of such a tile costs the busiest SIMD
//     (syrk_tile_cost), so that the workgroups of a launch carry about the same work.  (Round 2 gave every tile the
//     same split: the launch lasted as long as its fullest tile, and D = 257 paid six full tiles for three.)
//   Workgroup b = slab b:  x = b & 7, j = b >> 3 in [0, per_xcd);  class c owns j in [off[c], off[c] + n[c] s[c]),
//   sub-range k = (j - off[c]) / n[c], tile (j - off[c]) % n[c]  (sub-range-major: neighbours in time share rows).
// The two-operand product (GEN) is one class of nta x ntb rectangular tiles.
// nx = 8: the row partition over the XCDs described above; nx = 1 (more tiles than an XCD has resident slots, so a
// tile's workgroups cannot be co-resident anyway): every tile splits ALL the rows, one slab per tile and sub-range
// instead of eight (D = 8192: 2080 slabs of 128 KB, not 16 640).
struct SyrkPlan {
    int n[4];
    int s[4];
    int off[4];
    int per_xcd;
    int nx;
};

struct SyrkArgs {
    const void* X;   // double, or float in the fp32 storage mode (k_syrk_*<..., AT = float>)
    int64_t ldx;
    const double* d;
    int64_t N;
    int D;
    int nt;        // tiles per side
    SyrkPlan plan;
    double* part;  // [plan.nx plan.per_xcd][128*128]
    // GEN (two-operand) mode:  C = X^T diag(d) B  with B (N x Db); nt x ntb rectangular tile grid
    const void* B;   // double, or float for the fp32 storage mode's S (k_syrk_*<true, ..., float>)
    int64_t ldb;
    int Db;
    int ntb;
};

// workgroup -> (class, tile within the class) and its rows [r0, r1) (multiples of 16 but for the very end)
__host__ __device__ __forceinline__ void plan_locate(const SyrkPlan& p, int64_t N, int wg, int& cls, int& local,
                                                     int64_t& r0, int64_t& r1) {
    const int x = wg % p.nx, j = wg / p.nx;
    cls = 0;
    while (cls < 3 && j >= p.off[cls] + p.n[cls] * p.s[cls]) ++cls;
    const int r = j - p.off[cls];
    const int k = r / p.n[cls];
    local = r - k * p.n[cls];
    const int64_t len8 = ((N + p.nx - 1) / p.nx + 15) / 16 * 16;
    const int64_t x0 = min((int64_t)x * len8, N), x1 = min(N, x0 + len8);
    const int64_t per = ((x1 - x0 + p.s[cls] - 1) / p.s[cls] + 15) / 16 * 16;
    r0 = min(x1, x0 + (int64_t)k * per);
    r1 = min(x1, r0 + per);
}
// class-order tile index u in [0, sum n) -> (class, local)
__host__ __device__ __forceinline__ void plan_tile(const SyrkPlan& p, int u, int& cls, int& local) {
    cls = 0;
    while (cls < 3 && u >= p.n[cls]) u -= p.n[cls++];
    local = u;
}

// (class, local) -> (ti, tj): lower triangle (SYRK) or ti * ntb + tj (two-operand product)
template <bool GEN>
__host__ __device__ __forceinline__ void tile_of(int nt, int ntb, int cls, int local, int& ti, int& tj) {
    if constexpr (GEN) {
        ti = local / ntb;
        tj = local - ti * ntb;
    } else if (cls == 0) {       // off-diagonal tiles of the full tile rows: row ti holds ti of them
        ti = 1;
        while (ti * (ti + 1) / 2 <= local) ++ti;
        tj = local - ti * (ti - 1) / 2;
    } else if (cls == 1) {
        ti = tj = local;
    } else if (cls == 2) {
        ti = nt - 1;
        tj = local;
    } else {
        ti = tj = nt - 1;
    }
}

// What wave `wave` (0..7, SIMD wave & 3) of a 128 x 128 tile computes: the 4 x 2 blocks (16 x 16 each) at rows
// 64 wm + 16 i, columns 32 wn + 16 j whose bit 2 i + j is set in `mask` -- the blocks inside icols x jcols and, on
// a diagonal tile, not strictly above the diagonal.  Diagonal tiles deal their six non-empty sub-tiles so that the
// four SIMDs carry 8 / 8 / 10 / 10 blocks (the plain 2 x 4 grid: 16 / 16 / 8 / 8, as long as a full tile).
struct WaveJob {
    int wm, wn;
    unsigned mask;
};
__host__ __device__ __forceinline__ WaveJob syrk_wave_job(bool diag, int icols, int jcols, int wave) {
    WaveJob w{wave >> 2, wave & 3, 0u};
    if (diag) {
        //                 wave:  0  1  2  3   4   5  6  7
        const int wmT = 0x10000111, wnT = 0x31000210;     // nibbles, wave 0 lowest; waves 4 and 5 idle
        w.wm = (wmT >> (4 * wave)) & 15;
        w.wn = (wnT >> (4 * wave)) & 15;
        if (wave == 4 || wave == 5) return WaveJob{0, 0, 0u};
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j) {
            const int r = 64 * w.wm + 16 * i, c = 32 * w.wn + 16 * j;
            if (r < icols && c < jcols && (!diag || c <= r)) w.mask |= 1u << (2 * i + j);
        }
    return w;
}
// MFMA blocks per k-step on the busiest SIMD, plus the stage's fixed share (DMA issue, barrier, operand reads) in
// the same unit
inline int syrk_tile_cost(bool diag, int icols, int jcols) {
    static const int c0 = getenv("VT_SYRK_C0") ? atoi(getenv("VT_SYRK_C0")) : 6;   // dev switch, read once
    int simd[4] = {0, 0, 0, 0};
    for (int wv = 0; wv < 8; ++wv) simd[wv & 3] += __builtin_popcount(syrk_wave_job(diag, icols, jcols, wv).mask);
    return std::max(std::max(simd[0], simd[1]), std::max(simd[2], simd[3])) + c0;
}

template <bool GEN, typename BT = double, typename AT = double>
__global__ __launch_bounds__(256) void k_syrk_tile(SyrkArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15;
    int cls, local, ti, tj;
    int64_t r0, r1;
    plan_locate(a.plan, a.N, blockIdx.x, cls, local, r0, r1);
    tile_of<GEN>(a.nt, a.ntb, cls, local, ti, tj);
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = ti * kT + wm * 64, j0 = tj * kT + wn * 64;
    const int Dj = GEN ? a.Db : a.D;
    double* out = a.part + (int64_t)blockIdx.x * (kT * kT);
    // wave strictly above the diagonal (SYRK) or entirely in the padding
    const bool skip = (!GEN && ti == tj && wn > wm) || i0 >= a.D || j0 >= Dj;

    // 4-row steps (a non-empty range starts at a multiple of 16; the last step is masked by n < N)
    const int64_t s0 = r1 > r0 ? r0 >> 2 : 0, s1 = r1 > r0 ? (r1 + 3) >> 2 : 0;

    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};

    if (!skip) {
        int ca[4], cb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ca[i] = min(i0 + 16 * i + c, a.D - 1);
            cb[i] = min(j0 + 16 * i + c, Dj - 1);
        }
        struct Buf { double xa[4], xb[4], dn; };
        auto load = [&](int64_t s, Buf& b) {
            int64_t n = s * 4 + q;
            bool ok = s < s1 && n < a.N;
            int64_t nc = n < a.N ? n : a.N - 1;
            const AT* xr = (const AT*)a.X + nc * a.ldx;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                b.xa[i] = (double)xr[ca[i]];
                if constexpr (GEN) b.xb[i] = (double)((const BT*)a.B + nc * a.ldb)[cb[i]];
                else b.xb[i] = (double)xr[cb[i]];
            }
            double dv = a.d ? a.d[nc] : 1.0;
            b.dn = ok ? dv : 0.0;
        };
        auto compute = [&](const Buf& b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double bj = b.dn * b.xb[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = mfma_f64(b.xa[i], bj, acc[i][j]);
            }
        };
        Buf b0, b1, b2;
        int64_t s = s0;
        load(s, b0);
        load(s + 1, b1);
        load(s + 2, b2);
        for (; s < s1; s += 3) {
            compute(b0);
            load(s + 3, b0);
            __builtin_amdgcn_sched_barrier(0);
            compute(b1);
            load(s + 4, b1);
            __builtin_amdgcn_sched_barrier(0);
            compute(b2);
            load(s + 5, b2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // columns beyond D inside the tile hold copies of column D-1: never read by the reduce.
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(wm * 64 + 16 * i + q + 4 * r) * kT + wn * 64 + 16 * j + c] = acc[i][j][r];
}

// LDS-staged version: panels [n][col] (row stride 144 doubles = 16 mod 32: conflict-free operand
// reads), one DMA instruction per row of a panel, d_n fetched a stage ahead.
// STAGES = 3: one workgroup per CU, DMA two stages ahead.  STAGES = 2: 72 KB of LDS, so TWO workgroups
// share a CU (4 waves per SIMD) and cover each other's barrier and DMA waits.
// BT = float (GEN only): the second operand is stored in float (the fp32 storage mode's S); its panel rows are
// 512 bytes of the same LDS slots and the B operand is converted as it is read.
// AT = float: the first operand (the SYRK's only one) is stored in float -- the fp32 storage mode's X for D > 128:
// half the DMA bytes, products and sums still fp64 (the operands are widened as they are read, so the result is the
// fp64 Hessian of the rounded data).
template <bool GEN, int STAGES, typename BT = double, typename AT = double>
__global__ __launch_bounds__(512) void k_syrk_lds(SyrkArgs a) {
    static_assert(GEN || std::is_same<BT, double>::value, "the SYRK has one operand (its type is AT)");
    using JT = typename std::conditional<GEN, BT, AT>::type;   // storage type of the column-side panel
    extern __shared__ double lds[];            // [STAGES][I: 16 x 144 | J: 16 x 144]
    constexpr int RSTR = kT + 16, PANEL = kKC * RSTR;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, c = lane & 15;
    int cls, local, ti, tj;
    int64_t r0, r1;
    plan_locate(a.plan, a.N, blockIdx.x, cls, local, r0, r1);
    tile_of<GEN>(a.nt, a.ntb, cls, local, ti, tj);
    double* out = a.part + (int64_t)blockIdx.x * (kT * kT);
    const int nk = r1 > r0 ? (int)((r1 - r0 + kKC - 1) / kKC) : 0;
    const int icols = min(kT, a.D - ti * kT), jcols = min(kT, (GEN ? a.Db : a.D) - tj * kT);
    // this wave's blocks of the tile (wave-uniform: everything below branches on scalars)
    const WaveJob job = syrk_wave_job(!GEN && ti == tj, icols, jcols, wave);
    const int wm = job.wm, wn = job.wn;
    const unsigned mask = __builtin_amdgcn_readfirstlane(job.mask);

    auto issue = [&](int t, int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int id = wave * 4 + e;
            const bool isj = id >= 16;
            const int n = id & 15;
            const int64_t row = r0 + (int64_t)t * kKC + n;
            const bool ok = t < nk && row < r1;
            const void* base = (GEN && isj) ? (const void*)((const BT*)a.B + (ok ? row : 0) * a.ldb + tj * kT)
                                            : (const void*)((const AT*)a.X + (ok ? row : 0) * a.ldx + (isj ? tj : ti) * kT);
            const unsigned bytes = ok ? ((GEN && isj) ? (unsigned)sizeof(BT) : (unsigned)sizeof(AT)) * (isj ? jcols : icols) : 0u;
            double* dst = lds + buf * (2 * PANEL) + (isj ? PANEL : 0) + n * RSTR;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(lg_rsrc(base, bytes), (vt_ldsptr)dst, 16, 16u * lane, 0, 0, 0);
        }
    };
    auto load_d = [&](int t, double (&dv)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            int64_t row = r0 + (int64_t)t * kKC + 4 * s + q;
            dv[s] = (GEN && !a.d) ? 1.0 : a.d[row < a.N ? row : a.N - 1];   // rows past the chunk meet zero-filled X
        }
    };

    f64x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};

    double dcur[4], dnxt[4];
    load_d(0, dcur);
    issue(0, 0);
    if constexpr (STAGES == 3) issue(1, 1);
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
        if constexpr (STAGES == 3) {
            lg_wait4_barrier();
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        load_d(t + 1, dnxt);                       // before the DMA issue: the newest 4 stay the DMAs
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (STAGES == 3) {
            const int b2 = buf == 0 ? 2 : buf - 1;
            issue(t + 2, b2);
        } else {
            issue(t + 1, buf ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (mask != 0u) {
            constexpr int BR = (int)(sizeof(double) / sizeof(JT));   // panel elements per double of LDS
            constexpr int AR = (int)(sizeof(double) / sizeof(AT));
            const AT* pa = (const AT*)(lds + buf * (2 * PANEL) + q * RSTR) + 64 * wm + c;
            const JT* pb = (const JT*)(lds + buf * (2 * PANEL) + PANEL + q * RSTR) + 32 * wn + c;
            {   // only the blocks of the mask (scalar branches; a second, unmasked copy of the loop for full tiles cost
                // 74 more VGPRs and the second workgroup per CU)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    double af[4], bf[2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i] = (double)pa[4 * s * RSTR * AR + 16 * i];
#pragma unroll
                    for (int j = 0; j < 2; ++j) bf[j] = dcur[s] * (double)pb[4 * s * RSTR * BR + 16 * j];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            if ((mask >> (2 * i + j)) & 1u) acc[i][j] = mfma_f64(af[i], bf[j], acc[i][j]);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) dcur[s] = dnxt[s];
        if constexpr (STAGES == 3) buf = buf == 2 ? 0 : buf + 1;
        else buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (blocks outside the mask are neither written nor read: the reduce kernels stay inside the tile's real rows and
    // columns and, for a diagonal tile, on or below the diagonal)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if ((mask >> (2 * i + j)) & 1u) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[(wm * 64 + 16 * i + q + 4 * r) * kT + wn * 32 + 16 * j + c] = acc[i][j][r];
            }
}

// H[i][j] = H[j][i] = sum_chunks part (+ prior on the diagonal), i >= j.  One thread per element,
// slabs summed in a fixed order.
__global__ __launch_bounds__(256) void k_syrk_reduce(const double* part, int nt, SyrkPlan plan, int D,
                                                     const double* prior, double* H, int ldh) {
    int cls, local, ti, tj;
    plan_tile(plan, blockIdx.y, cls, local);
    tile_of<false>(nt, 0, cls, local, ti, tj);
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int i = ti * kT + e / kT, j = tj * kT + e % kT;
    if (i >= D || j >= D || j > i) return;
    // the tile's slabs: workgroups x + 8 (off + k n + local), sub-range k, XCD x -- summed in that fixed order
    double sx[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) sx[x] = 0.0;
    for (int k = 0; k < plan.s[cls]; ++k) {
        const double* p = part + (int64_t)(plan.nx * (plan.off[cls] + k * plan.n[cls] + local)) * (kT * kT) + e;
#pragma unroll
        for (int x = 0; x < 8; ++x)
            if (x < plan.nx) sx[x] += p[(int64_t)x * (kT * kT)];
    }
    double s = ((sx[0] + sx[1]) + (sx[2] + sx[3])) + ((sx[4] + sx[5]) + (sx[6] + sx[7]));
    if (i == j && prior) s += prior[i];
    H[(int64_t)i * ldh + j] = s;
    H[(int64_t)j * ldh + i] = s;
}

// C[i][j] = alpha * sum_chunks part + bias[j]   (i < Da, j < Db): the two-operand product's epilogue.
// One thread per output element (a single tile may be split 512 ways: 64 MB of slabs), slabs summed
// in a fixed order.
__global__ __launch_bounds__(256) void k_gemm_tn_reduce(const double* part, int ntb, int ntl, int nsub, int nx, int Da,
                                                        int Db, double alpha, const double* bias, double* C,
                                                        int64_t ldc) {
    const int t = blockIdx.y, ti = t / ntb, tj = t - ti * ntb;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int i = ti * kT + e / kT, j = tj * kT + e % kT;
    if (i >= Da || j >= Db) return;
    // one class of ntl tiles (SyrkPlan): slabs x + 8 (k ntl + t), sub-range k, XCD x
    double sx[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) sx[x] = 0.0;
    for (int k = 0; k < nsub; ++k) {
        const double* p = part + (int64_t)(nx * (k * ntl + t)) * (kT * kT) + e;
#pragma unroll
        for (int x = 0; x < 8; ++x)
            if (x < nx) sx[x] += p[(int64_t)x * (kT * kT)];
    }
    double s = alpha * (((sx[0] + sx[1]) + (sx[2] + sx[3])) + ((sx[4] + sx[5]) + (sx[6] + sx[7])));
    if (bias) s += bias[j];
    C[(int64_t)i * ldc + j] = s;
}

// ------------------------------------------------------------------ small helpers -------
// Lc = tril(L), LT = Lc^T, both with the even leading dimension image_ld(D) (16-byte aligned rows for the LDS-DMA
// of the update products, whatever the parity of D); the pad column is zero.
__host__ __device__ constexpr int image_ld(int D) { return D + (D & 1); }
__global__ void k_copy_lower_and_transpose(const double* L, int ldl, int D, double* Lc, double* LT) {
    const int Dl = image_ld(D);
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)D * Dl) return;
    int i = (int)(idx / Dl), j = (int)(idx - (int64_t)i * Dl);
    if (j >= D) {   // pad column of row i of both images
        Lc[(int64_t)i * Dl + j] = 0.0;
        LT[(int64_t)i * Dl + j] = 0.0;
        return;
    }
    double v = j <= i ? L[(int64_t)i * ldl + j] : 0.0;
    Lc[(int64_t)i * Dl + j] = v;
    LT[(int64_t)j * Dl + i] = v;
}

__global__ void k_merge_info(int* info, const int* local, int offset) {
    if (*info == 0 && *local != 0) *info = offset + *local;
}

}  // namespace

// ------------------------------------------------------------------ host algorithms -------
size_t vt_large_potrf_ws_bytes(int D) { return (size_t)kPackStride * sizeof(double) + 256; }

int vt_large_potrf(double* A, int lda, int D, int* info, void* ws, hipStream_t st) {
    double* pack = (double*)ws;
    int* linfo = (int*)((char*)ws + (size_t)kPackStride * sizeof(double));
    if (hipMemsetAsync(info, 0, sizeof(int), st) != hipSuccess) return VT_ERR_HIP;
    const int nblk = (D + kT - 1) / kT;
    for (int J = 0; J < nblk; ++J) {
        const int o = J * kT, Db = std::min(kT, D - o);
        double* Ajj = A + (int64_t)o * lda + o;
        int rc = vt_potrf_pack_block(Ajj, lda, Db, linfo, pack, 1, st);
        if (rc != VT_OK) return rc;
        hipLaunchKernelGGL(k_merge_info, dim3(1), dim3(1), 0, st, info, (const int*)linfo, o);
        const int R = D - o - Db;
        if (R > 0) {
            double* P = A + (int64_t)(o + Db) * lda + o;   // panel below the diagonal block
            rc = vt_trsm_rows_mode(pack, Db, P, lda, nullptr, 1.0, R, P, lda, 1, st);   // P <- P L_JJ^{-T}
            if (rc != VT_OK) return rc;
            double* C = A + (int64_t)(o + Db) * lda + (o + Db);
            rc = gemm_nt(C, lda, R, R, P, lda, P, lda, Db, -1.0, 1.0, true, st);
            if (rc != VT_OK) return rc;
        }
    }
    return vt_launch_status();
}

// ---- blocked solve, D > 256 -----------------------------------------------------------------------------------
// Diagonal blocks of kBW = 384 columns (256 until the one-pass chain reached NB = 24), solved by the chain kernels
// of vt_trsm_mid.h (a full block: forward-only /
// backward-only halves; the last, partial block: the whole chain of vt_trsm_mid / vt_chol), off-diagonal panels
// by GEMM-NT updates.  Round 2 used 128-column blocks and the small chain: twice the sweeps over S (each an
// HBM-bound read + write of N x 128) and twice the update launches; a 256-column block does 4x the MFMA work per
// byte of its sweep (MFMA-bound at 73 % of peak) and moves a quarter of the flops out of the GEMM.
// Pack: [Lc | LT | one chain pack per 256-column block].  VT_TRSM_NO_MID=1 keeps the 128-column scheme (dev A/B).
constexpr int kBW = VT_TRSM_ONEPASS_D_MAX;
static bool wide_blocks() { return vt_mid_trsm_ok(kBW); }
static size_t wide_pack_stride() {   // doubles per block pack: the larger of a chain pack and a small pack
    return (std::max(vt_mid_pack_bytes(kBW), (size_t)kPackStride * sizeof(double)) + 255) / 256 * 256 / sizeof(double);
}
// Diagonal blocks of the blocked solve: 384 columns while more than 448 are left; a rest of 385..448 columns is cut
// as 256 + (129..192) -- a 384-column block would pay its two half-chains and two update passes for a few columns
// (D = 385: 15.8 ms against 9.5 ms at D = 384) -- and a rest of <= 384 is one block (the whole chain in one launch).
constexpr int kMaxWideBlocks = VT_DENSE_D_MAX / kBW + 3;
static int wide_partition(int D, int* off) {   // off[0..n]: column offsets, returns n
    int n = 0, o = 0;
    off[0] = 0;
    static const int rest_env = getenv("VT_WIDE_REST") ? atoi(getenv("VT_WIDE_REST")) : 64;   // dev switch
    const int rest = std::min(std::max(rest_env, 0), VT_MID_D_MAX);
    while (D - o > kBW + rest) { o += kBW; off[++n] = o; }
    if (D - o > kBW) { o += VT_MID_D_MAX; off[++n] = o; }
    off[++n] = D;
    return n;
}

size_t vt_large_pack_bytes(int D) {
    int off[kMaxWideBlocks + 1];
    const int nblk = (D + kT - 1) / kT, nbw = wide_partition(D, off);
    const size_t images = (size_t)2 * D * image_ld(D);
    return (images + std::max((size_t)nblk * kPackStride, (size_t)nbw * wide_pack_stride())) * sizeof(double);
}

int vt_large_pack(const double* L, int ldl, int D, void* pack, hipStream_t st) {
    const int Dl = image_ld(D);
    double* Lc = (double*)pack;
    double* LT = Lc + (size_t)D * Dl;
    double* packs = LT + (size_t)D * Dl;
    int64_t tot = (int64_t)D * Dl;
    hipLaunchKernelGGL(k_copy_lower_and_transpose, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, L, ldl, D,
                       Lc, LT);
    if (wide_blocks()) {
        int off[kMaxWideBlocks + 1];
        const int nbw = wide_partition(D, off);
        for (int J = 0; J < nbw; ++J) {
            const int o = off[J], Db = off[J + 1] - o;
            double* pk = packs + (size_t)J * wide_pack_stride();
            int rc = Db > VT_SMALL_D_MAX ? vt_mid_pack(Lc + (int64_t)o * Dl + o, Dl, Db, pk, st)
                                         : vt_potrf_pack_block(Lc + (int64_t)o * Dl + o, Dl, Db, nullptr, pk, 0, st);
            if (rc != VT_OK) return rc;
        }
        return vt_launch_status();
    }
    const int nblk = (D + kT - 1) / kT;
    for (int J = 0; J < nblk; ++J) {
        const int o = J * kT, Db = std::min(kT, D - o);
        int rc = vt_potrf_pack_block(Lc + (int64_t)o * Dl + o, Dl, Db, nullptr, packs + (size_t)J * kPackStride, 0, st);
        if (rc != VT_OK) return rc;
    }
    return vt_launch_status();
}

static int large_trsm_rows_wide(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale,
                                double alpha, int64_t N, double* Out, int64_t ldo, hipStream_t st) {
    const int Dl = image_ld(D);
    const double* Lc = (const double*)pack;
    const double* LT = Lc + (size_t)D * Dl;
    const double* packs = LT + (size_t)D * Dl;
    int off[kMaxWideBlocks + 1];
    const int nbw = wide_partition(D, off);
    int rc;
    // forward:  Y L^T = G with G = alpha * s .* B never materialised (block 0: the chain's row scale; later blocks:
    // the update's epilogue); the last block runs its forward AND backward chain in one launch
    for (int J = 0; J < nbw; ++J) {
        const int o = off[J], Db = off[J + 1] - o;
        const double* pk = packs + (size_t)J * wide_pack_stride();
        const double* src = Out + o;
        int64_t lds_ = ldo;
        const double* sc = nullptr;
        double al = 1.0;
        if (J == 0) {
            src = B; lds_ = ldb; sc = rowscale; al = alpha;
        } else {
            rc = gemm_nt(Out + o, ldo, N, Db, Out, ldo, Lc + (int64_t)o * Dl, Dl, o, -1.0, 0.0, false, st, B + o, ldb,
                         rowscale, alpha);
            if (rc != VT_OK) return rc;
        }
        if (J < nbw - 1) rc = vt_mid_trsm_rows_mode(pk, Db, src, lds_, sc, al, N, Out + o, ldo, 1, st);
        else if (Db > VT_SMALL_D_MAX) rc = vt_mid_trsm_rows_mode(pk, Db, src, lds_, sc, al, N, Out + o, ldo, 0, st);
        else rc = vt_trsm_rows_mode(pk, Db, src, lds_, sc, al, N, Out + o, ldo, 0, st);
        if (rc != VT_OK) return rc;
    }
    for (int J = nbw - 2; J >= 0; --J) {   // backward:  Z L = Y
        const int o = off[J], Db = off[J + 1] - o, rest = D - o - Db;
        rc = gemm_nt(Out + o, ldo, N, Db, Out + o + Db, ldo, LT + (int64_t)o * Dl + o + Db, Dl, rest, -1.0, 1.0,
                     false, st);
        if (rc != VT_OK) return rc;
        rc = vt_mid_trsm_rows_mode(packs + (size_t)J * wide_pack_stride(), Db, Out + o, ldo, nullptr, 1.0, N, Out + o,
                                   ldo, 2, st);
        if (rc != VT_OK) return rc;
    }
    return vt_launch_status();
}

int vt_large_trsm_rows(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale,
                       double alpha, int64_t N, double* Out, int64_t ldo, hipStream_t st) {
    if (wide_blocks()) return large_trsm_rows_wide(pack, D, B, ldb, rowscale, alpha, N, Out, ldo, st);
    const int Dl = image_ld(D);
    const double* Lc = (const double*)pack;
    const double* LT = Lc + (size_t)D * Dl;
    const double* packs = LT + (size_t)D * Dl;
    const int nblk = (D + kT - 1) / kT;
    // forward:  Y L^T = G with G = alpha * s .* B never materialised: block 0 takes it through the
    // small solve's row scale, the later blocks through the update's epilogue.
    int rc;
    for (int J = 0; J < nblk; ++J) {
        const int o = J * kT, Db = std::min(kT, D - o);
        if (J == 0) {
            rc = vt_trsm_rows_mode(packs, Db, B, ldb, rowscale, alpha, N, Out, ldo, 1, st);
        } else {
            rc = gemm_nt(Out + o, ldo, N, Db, Out, ldo, Lc + (int64_t)o * Dl, Dl, o, -1.0, 0.0, false, st, B + o, ldb,
                         rowscale, alpha);
            if (rc != VT_OK) return rc;
            rc = vt_trsm_rows_mode(packs + (size_t)J * kPackStride, Db, Out + o, ldo, nullptr, 1.0, N, Out + o, ldo, 1, st);
        }
        if (rc != VT_OK) return rc;
    }
    for (int J = nblk - 1; J >= 0; --J) {   // backward:  Z L = Y
        const int o = J * kT, Db = std::min(kT, D - o);
        const int rest = D - o - Db;
        if (rest > 0) {
            rc = gemm_nt(Out + o, ldo, N, Db, Out + o + Db, ldo, LT + (int64_t)o * Dl + o + Db, Dl, rest, -1.0, 1.0,
                         false, st);
            if (rc != VT_OK) return rc;
        }
        rc = vt_trsm_rows_mode(packs + (size_t)J * kPackStride, Db, Out + o, ldo, nullptr, 1.0, N, Out + o, ldo, 2, st);
        if (rc != VT_OK) return rc;
    }
    return vt_launch_status();
}

constexpr int kXcds = 8;
static int plan_nx(int ntl) { return ntl <= 2 * (2 * num_cu() / kXcds) ? kXcds : 1; }   // two rounds of an XCD's slots

static SyrkPlan uniform_plan(int ntl, int nsub) {   // one class: every tile splits each XCD's rows nsub ways
    SyrkPlan p{};
    p.nx = plan_nx(ntl);
    p.n[0] = ntl;
    for (int c = 0; c < 4; ++c) p.s[c] = nsub;
    p.off[1] = p.off[2] = p.off[3] = p.per_xcd = ntl * nsub;
    return p;
}

// SYRK: at most `rounds` rounds of an XCD's 64 resident workgroups (32 CUs x 2), dealt to the tile classes in
// proportion to n_c cost_c.  The launch must FIT whole rounds -- one workgroup too many starts another round, as
// long as the first -- so shares are rounded down and the class with the most work per workgroup then gets one
// more sub-range per tile while the whole still fits.
static SyrkPlan syrk_plan_for(int D, int64_t N, int nx, int target, int& maxper_out, const int (&cost)[4]) {
    const int nt = (D + kT - 1) / kT, wl = D - (nt - 1) * kT;
    const bool partial = wl < kT;
    const int nfull = partial ? nt - 1 : nt;
    SyrkPlan p{};
    p.nx = nx;
    p.n[0] = nfull * (nfull - 1) / 2;
    p.n[1] = nfull;
    p.n[2] = partial ? nt - 1 : 0;
    p.n[3] = partial ? 1 : 0;
    double W = 0.0;
    for (int c = 0; c < 4; ++c) W += (double)p.n[c] * cost[c];
    const int64_t max_sub = std::max<int64_t>(1, (N / nx + 4 * kKC - 1) / (4 * kKC));   // at least 4 stages per workgroup
    int wg = 0;
    for (int c = 0; c < 4; ++c) {
        p.s[c] = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)(target * (double)cost[c] / W), max_sub));
        wg += p.n[c] * p.s[c];
    }
    while (true) {
        int best = -1;
        double worst = 0.0, heaviest = 0.0;
        for (int c = 0; c < 4; ++c) {
            if (p.n[c] == 0) continue;
            const double per_wg = (double)cost[c] / p.s[c];
            heaviest = std::max(heaviest, per_wg);
            if (p.s[c] >= max_sub || wg + p.n[c] > target) continue;
            if (per_wg > worst) {
                worst = per_wg;
                best = c;
            }
        }
        if (best < 0 || worst < heaviest) break;   // (only while it shortens the launch)
        p.s[best] += 1;
        wg += p.n[best];
    }
    wg = 0;
    double maxper = 0.0;
    for (int c = 0; c < 4; ++c) {
        p.off[c] = wg;
        wg += p.n[c] * p.s[c];
        if (p.n[c] > 0) maxper = std::max(maxper, 1000.0 * cost[c] / p.s[c]);
    }
    p.per_xcd = wg;
    maxper_out = (int)maxper;
    return p;
}
constexpr int kSyrkMaxRounds = 2;
static SyrkPlan syrk_plan(int D, int64_t N) {
    static const int rounds_env = getenv("VT_SYRK_ROUNDS") ? atoi(getenv("VT_SYRK_ROUNDS")) : 0;   // dev switch
    const int nt = (D + kT - 1) / kT, wl = D - (nt - 1) * kT;
    const int cost[4] = {syrk_tile_cost(false, kT, kT), syrk_tile_cost(true, kT, kT), syrk_tile_cost(false, wl, kT),
                         syrk_tile_cost(true, wl, wl)};
    const int ntl = nt * (nt + 1) / 2;
    const int nx = plan_nx(ntl);
    const int slots = 2 * num_cu() / nx;
    SyrkPlan best{};
    double best_est = 0.0;
    for (int r = 1; r <= kSyrkMaxRounds; ++r) {
        if (rounds_env > 0 && r != std::min(rounds_env, kSyrkMaxRounds)) continue;
        int maxper;
        const SyrkPlan p = syrk_plan_for(D, N, nx, r * slots, maxper, cost);
        double W = 0.0;
        for (int c = 0; c < 4; ++c) W += 1000.0 * p.n[c] * cost[c];
        // one round: as long as its heaviest workgroup; more: the mean load of a slot plus half a workgroup of tail
        const double est = p.per_xcd <= slots ? (double)maxper : W / slots + 0.5 * maxper;
        if (best.per_xcd == 0 || est < best_est) {
            best = p;
            best_est = est;
        }
    }
    return best;
}

size_t vt_large_syrk_ws_bytes(int D) {
    const int nt = (D + kT - 1) / kT, ntl = nt * (nt + 1) / 2;
    // a plan fits kSyrkMaxRounds rounds of the chip unless every tile is already unsplit
    return (size_t)std::max(kSyrkMaxRounds * 2 * num_cu(), plan_nx(ntl) * ntl) * kT * kT * sizeof(double);
}

template <typename AT>
static int large_syrk_impl(const AT* X, int64_t ldx, const double* d, int64_t N, int D, const double* prior, double* H,
                           int ldh, void* ws, hipStream_t st) {
    const int nt = (D + kT - 1) / kT, ntl = nt * (nt + 1) / 2;
    SyrkPlan plan = syrk_plan(D, N);
    SyrkArgs a{X, ldx, d, N, D, nt, plan, (double*)ws, nullptr, 0, 0, 0};
    constexpr int kPer16 = 16 / (int)sizeof(AT);   // elements per 16-byte DMA lane
    if (N == 0) {
        for (int c = 0; c < 4; ++c) plan.s[c] = 0;   // no rows: the reduce below sums zero slabs and adds the prior
    } else if (ldx % kPer16 == 0 && (uintptr_t)X % 16 == 0 && !large_no_lds()) {
        const size_t shmem = (size_t)large_stages() * 2 * kKC * (kT + 16) * sizeof(double);   // 108 / 72 KB
        static VtLdsAttr attr[2];
        if (vt_allow_big_lds((const void*)k_syrk_lds<false, 3, double, AT>, attr[0], (int)shmem) != VT_OK ||
            vt_allow_big_lds((const void*)k_syrk_lds<false, 2, double, AT>, attr[1], (int)shmem) != VT_OK)
            return VT_ERR_HIP;
        if (large_stages() == 2)
            hipLaunchKernelGGL((k_syrk_lds<false, 2, double, AT>), dim3(plan.nx * plan.per_xcd), dim3(512), shmem, st, a);
        else
            hipLaunchKernelGGL((k_syrk_lds<false, 3, double, AT>), dim3(plan.nx * plan.per_xcd), dim3(512), shmem, st, a);
    } else {
        hipLaunchKernelGGL((k_syrk_tile<false, double, AT>), dim3(plan.nx * plan.per_xcd), dim3(256), 0, st, a);
    }
    hipLaunchKernelGGL(k_syrk_reduce, dim3(kT * kT / 256, ntl), dim3(256), 0, st, (const double*)ws, nt, plan, D, prior,
                       H, ldh);
    return vt_launch_status();
}

int vt_large_syrk(const double* X, int64_t ldx, const double* d, int64_t N, int D, const double* prior, double* H,
                  int ldh, void* ws, hipStream_t st) {
    return large_syrk_impl<double>(X, ldx, d, N, D, prior, H, ldh, ws, st);
}
// fp32 storage mode, D > 128: X in float (half the bytes), every product and sum in fp64
int vt_large_syrk_f32(const float* X, int64_t ldx, const double* d, int64_t N, int D, const double* prior, double* H,
                      int ldh, void* ws, hipStream_t st) {
    return large_syrk_impl<float>(X, ldx, d, N, D, prior, H, ldh, ws, st);
}

// ------------------------------------------------------------------ two-operand product -------
// C (Da x Db) = alpha * A^T diag(s) B + 1 bias^T for row-major A (N x Da), B (N x Db): the SYRK
// kernels with a second operand and a rectangular tile grid; the N axis is split so that the
// grid fills the chip even when there is a single output tile.
// (sub-ranges of an XCD's rows per tile: one round of the XCD's 64 resident workgroups when the tiles are fewer)
static int gemm_tn_split(int ntl) { return std::max(1, (2 * num_cu() / plan_nx(ntl)) / ntl); }

static size_t gemm_tn_ws_bytes(int Da, int Db) {
    const int nta = (Da + kT - 1) / kT, ntb = (Db + kT - 1) / kT;
    return (size_t)plan_nx(nta * ntb) * gemm_tn_split(nta * ntb) * nta * ntb * kT * kT * sizeof(double);
}

extern "C" size_t vt_gemm_tn_workspace_bytes(int Da, int Db) {
    if (Da < 1 || Db < 1 || Da > VT_DENSE_D_MAX || Db > VT_DENSE_D_MAX) return 0;
    return gemm_tn_ws_bytes(Da, Db);
}

template <typename BT>
static int gemm_tn_impl(const double* A, int64_t lda, int Da, const BT* B, int64_t ldb, int Db, const double* rowscale,
                        int64_t N, double alpha, const double* bias, double* C, int64_t ldc, void* ws, size_t ws_bytes,
                        void* stream, int db_max = VT_DENSE_D_MAX) {
    if (!C || N < 0 || Da < 1 || Db < 1 || lda < Da || ldb < Db || ldc < Db) return VT_ERR_BAD_ARG;
    if (N > 0 && (!A || !B)) return VT_ERR_BAD_ARG;
    if (Da > VT_DENSE_D_MAX || Db > db_max) return VT_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < gemm_tn_ws_bytes(Da, Db)) return VT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nta = (Da + kT - 1) / kT, ntb = (Db + kT - 1) / kT, ntl = nta * ntb;
    int ks = gemm_tn_split(ntl);   // sub-ranges per XCD
    const int nx = plan_nx(ntl);
    const int64_t nsteps = (N / nx + 4 * kKC - 1) / (4 * kKC);   // at least 4 stages per workgroup
    if (ks > nsteps) ks = (int)std::max<int64_t>(1, nsteps);
    SyrkArgs a{A, lda, rowscale, N, Da, nta, uniform_plan(ntl, ks), (double*)ws, B, ldb, Db, ntb};
    constexpr int kBPer16 = 16 / (int)sizeof(BT);   // B elements per 16-byte DMA lane
    if (N == 0) {
        ks = 0;   // no rows: C = bias
    } else if (lda % 2 == 0 && ldb % kBPer16 == 0 && (uintptr_t)A % 16 == 0 && (uintptr_t)B % 16 == 0 && !large_no_lds()) {
        const size_t shmem = (size_t)large_stages() * 2 * kKC * (kT + 16) * sizeof(double);   // 108 / 72 KB
        static VtLdsAttr attr[2];
        if (vt_allow_big_lds((const void*)k_syrk_lds<true, 3, BT>, attr[0], (int)shmem) != VT_OK ||
            vt_allow_big_lds((const void*)k_syrk_lds<true, 2, BT>, attr[1], (int)shmem) != VT_OK)
            return VT_ERR_HIP;
        if (large_stages() == 2) hipLaunchKernelGGL((k_syrk_lds<true, 2, BT>), dim3(nx * ntl * ks), dim3(512), shmem, st, a);
        else hipLaunchKernelGGL((k_syrk_lds<true, 3, BT>), dim3(nx * ntl * ks), dim3(512), shmem, st, a);
    } else {
        hipLaunchKernelGGL((k_syrk_tile<true, BT>), dim3(nx * ntl * ks), dim3(256), 0, st, a);
    }
    hipLaunchKernelGGL(k_gemm_tn_reduce, dim3(kT * kT / 256, ntl), dim3(256), 0, st, (const double*)ws, ntb, ntl, ks,
                       nx, Da, Db, alpha, bias, C, ldc);
    return vt_launch_status();
}

extern "C" int vt_gemm_tn(const double* A, int64_t lda, int Da, const double* B, int64_t ldb, int Db,
                          const double* rowscale, int64_t N, double alpha, const double* bias, double* C,
                          int64_t ldc, void* ws, size_t ws_bytes, void* stream) {
    return gemm_tn_impl<double>(A, lda, Da, B, ldb, Db, rowscale, N, alpha, bias, C, ldc, ws, ws_bytes, stream);
}

// fp32 storage mode: B (the sensitivity matrix S) is float; A, the sums and C stay double.
extern "C" int vt_gemm_tn_f32b(const double* A, int64_t lda, int Da, const float* B, int64_t ldb, int Db,
                               const double* rowscale, int64_t N, double alpha, const double* bias, double* C,
                               int64_t ldc, void* ws, size_t ws_bytes, void* stream) {
    return gemm_tn_impl<float>(A, lda, Da, B, ldb, Db, rowscale, N, alpha, bias, C, ldc, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------ multi-vector HVP -------
// Out[k, :] = X^T (d .* (X V[k, :])) + prior .* V[k, :]  for K vectors at once (V, Out: K x D, one vector per row).
// The single-vector kernel (vt_rows.hip) keeps v in LDS and the row in registers: K vectors do not fit either, so the
// K products are two passes of the MFMA tile kernels over X -- Z (N x K) = X V^T (GEMM-NT, a remainder panel of K
// columns: the narrow wave grid) and Out = Z^T diag(d) X (the two-operand product with a K-row first operand) --
// 16 N D bytes for any K <= 128 where K single products read 8 K N D.  Build-side extension:
// the reference applies its solver to all columns of the cross-Hessian at once (vittles/sensitivity_lib.py:226) and
// its CG route (vittles/solver_lib.py:91-97) would call the HVP once per column and iteration.
namespace {
__global__ void k_add_prior_rows(const double* prior, double prior_scalar, const double* V, int64_t ldv, int K, int D,
                                 double* Out, int64_t ldo) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)K * D) return;
    const int k = (int)(idx / D), j = (int)(idx - (int64_t)k * D);
    const double pj = prior ? prior[j] : prior_scalar;
    Out[k * ldo + j] = fma(pj, V[k * ldv + j], Out[k * ldo + j]);
}
inline int hvp_multi_kp(int K) { return (K + 1) & ~1; }   // Z rows on a 16-byte pitch
}  // namespace

extern "C" size_t vt_glm_hvp_multi_workspace_bytes(int64_t N, int D, int K) {
    if (N < 0 || D < 1 || K < 1 || K > kT || D > VT_ROWS_D_MAX) return 0;
    return ((size_t)N * hvp_multi_kp(K) * sizeof(double) + 255) / 256 * 256 + gemm_tn_ws_bytes(K, D);
}

extern "C" int vt_glm_hvp_multi(const double* X, int64_t ldx, const double* d, const double* prior, double prior_scalar,
                                const double* V, int64_t ldv, int K, int64_t N, int D, double* Out, int64_t ldo,
                                void* ws, size_t ws_bytes, void* stream) {
    if (!V || !Out || N < 0 || D < 1 || K < 1 || ldx < D || ldv < D || ldo < D) return VT_ERR_BAD_ARG;
    if (N > 0 && (!X || !d)) return VT_ERR_BAD_ARG;
    if (K > kT || D > VT_ROWS_D_MAX) return VT_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < vt_glm_hvp_multi_workspace_bytes(N, D, K)) return VT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int Kp = hvp_multi_kp(K);
    double* Z = (double*)ws;
    void* tn_ws = (char*)ws + ((size_t)N * Kp * sizeof(double) + 255) / 256 * 256;
    if (Kp > K && N > 0) VT_CHECK_HIP(hipMemsetAsync(Z, 0, (size_t)N * Kp * sizeof(double), st));   // the pad column
    int rc = gemm_nt(Z, Kp, N, K, X, ldx, V, ldv, D, 1.0, 0.0, false, st);
    if (rc != VT_OK) return rc;
    rc = gemm_tn_impl<double>(Z, Kp, K, X, ldx, D, d, N, 1.0, nullptr, Out, ldo, tn_ws, gemm_tn_ws_bytes(K, D), stream,
                              VT_ROWS_D_MAX);
    if (rc != VT_OK) return rc;
    const int64_t tot = (int64_t)K * D;
    hipLaunchKernelGGL(k_add_prior_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, prior, prior_scalar, V, ldv,
                       K, D, Out, ldo);
    return vt_launch_status();
}
