// K4 / K5: dense Cholesky, generic triangular solve, and the MFMA multi-RHS TRSM.
// Replaces scipy.linalg.cho_factor / cho_solve as the reference calls them
// (vittles/solver_lib.py:26-29) and the N-right-hand-side solve behind
// `-1 * self._hess_solver(jac0)` (vittles/sensitivity_lib.py:226).
#include "vt_common.h"
#include "vt_trsm_sched.h"
#ifndef VT_PABL
#define VT_PABL 0   // dev-only compile-time ablation mask
#endif
#ifndef VT_TRSM_NT_STORE
#define VT_TRSM_NT_STORE 0   // dev-only: non-temporal stores of the solved rows (measured: solve 0.378 -> 0.624 ms, the 16-byte
                             // pieces of a row no longer merge in L2; the next Hessian pass gains 1.5 %)
#endif
#ifndef VT_TRSM_PF_ALL
#define VT_TRSM_PF_ALL 0   // dev-only: force the prefetch depth of the solve kernel for every NB (0 = by NB)
#endif

namespace {

constexpr int kSmallLd = VT_SMALL_D_MAX + 1;  // odd stride: column walks are bank-conflict free

// ------------------------------------------------------------------ potrf -------
// One workgroup, matrix resident in LDS (D <= 128: 132 KB of the CU's 160 KB).
// Latency-bound by construction (D^3/3 flops is nothing), so the budget is barriers and
// dependent LDS round trips.  Blocked right-looking Cholesky with 16-wide panels:
//   (a) wave 0 factors the 16x16 diagonal block entirely in registers (lane = row; cross-lane
//       traffic is v_readlane with compile-time lane ids, no LDS),
//   (b) one thread per row below solves its 16-entry panel row by substitution against the block,
//   (c) the trailing blocks are updated with v_mfma_f64_16x16x4_f64 straight from LDS -- wave 0
//       takes the next diagonal block first and factors it (look-ahead) while the other waves do
//       the rest of the update and wave 1 inverts the block just used: the inverted diagonal blocks
//       are what the TRSM operand pack needs, and they never sit on the critical path.
// 2 barriers per panel.  The same kernel emits the pack (16x16x16 products on the MFMA).
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// do_factor: factor the matrix at Ain (leading dimension ldin) into A (Ain == A: in place; otherwise A
// also gets a zero strict upper triangle), else A already holds L;  pack != NULL: also emit the TRSM pack.
// ASMP: the panel substitution with hand-scheduled LDS reads (below); false: the same arithmetic in the same order as
// plain C++ (VT_POTRF_C_PANEL=1 selects it at run time; tests/test_gpu_chol.py compares the two bit for bit, so a
// toolchain whose register allocation breaks the asm variant's in-flight-read assumption is caught).
template <bool ASMP>
__global__ __launch_bounds__(256) void k_potrf_pack_small(const double* Ain, int ldin, double* A, int lda, int D,
                                                         int* info, double* pack, int do_factor) {
    __shared__ double a[VT_SMALL_D_MAX * kSmallLd];
    __shared__ double linv[8 * 16 * 17];
    __shared__ double rdiag_s[VT_SMALL_D_MAX];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NB = (D + 15) >> 4, NP = NB * 16, NBT = NB * (NB + 1) / 2;
    if (tid == 0) s_bad = 0;
    // zero-padded lower triangle -> LDS (identity on the padding diagonal); 128 threads per row, 32 rows'
    // loads in flight per thread (unconditional, clamped: a predicated load drains vmcnt every iteration),
    // so the whole matrix arrives in at most two memory latencies
    {
        const int j = tid & 127;
        const int jc = j < D ? j : D - 1;
        for (int i0 = tid >> 7; i0 < NP; i0 += 64) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int i = i0 + 2 * u;
                v[u] = Ain[(int64_t)(i < D ? i : D - 1) * ldin + jc];
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int i = i0 + 2 * u;
                if (i < NP && j < NP) a[i * kSmallLd + j] = (i < D && j <= i) ? v[u] : ((i >= D && j == i) ? 1.0 : 0.0);
            }
        }
    }
    __syncthreads();

    // (a) diagonal block J: factor in registers, lane r = row r (wave 0).  Leaves L_JJ in LDS and the
    //     reciprocal pivots in rdiag_s.
    auto diag_factor = [&](int J) __attribute__((always_inline)) {
        const int o = 16 * J;
        const int r = lane & 15;
        double row[16];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) row[cc] = a[(o + r) * kSmallLd + o + cc];
        int bad = 0;
        double rdiag[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            double akk = readlane_f64(row[k], k);
            if (!(akk > 0.0) && bad == 0 && o + k < D) bad = o + k + 1;
            // 1/sqrt by v_rsq_f64 + two Newton steps (the IEEE sqrt and divide sequences are
            // ~80 dependent instructions per column of this 16-long serial chain)
            const double ap = akk > 0.0 ? akk : 1.0;
            double rp = __builtin_amdgcn_rsq(ap);
            rp = rp * fma(-0.5 * ap * rp, rp, 1.5);
            rp = rp * fma(-0.5 * ap * rp, rp, 1.5);
            double piv = ap * rp;
            piv = fma(fma(-piv, piv, ap), 0.5 * rp, piv);   // one correction step of sqrt itself
            rdiag[k] = rp;
            double lk = r > k ? row[k] * rp : 0.0;
            row[k] = r == k ? piv : lk;
#pragma unroll
            for (int j = k + 1; j < 16; ++j) {
                double ljk = readlane_f64(lk, j);
                row[j] = fma(-lk, ljk, row[j]);
            }
        }
        if (lane < 16) {
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) a[(o + r) * kSmallLd + o + cc] = cc <= r ? row[cc] : 0.0;
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) rdiag_s[o + k] = rdiag[k];
        }
        if (bad && lane == 0) s_bad = bad;
    };
    // inverse of the factored block J (any wave): lane c holds column c of L_JJ^{-1},
    //   x_i = (delta_ic - sum_{m<i} L[i][m] x_m) / L[i][i];  L[i][m] are uniform (broadcast) LDS reads
    auto diag_invert = [&](int J) __attribute__((always_inline)) {
        const int o = 16 * J;
        const int c = lane & 15;
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double sacc = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int m = 0; m < i; ++m) sacc = fma(-a[(o + i) * kSmallLd + o + m], x[m], sacc);
            x[i] = i < c ? 0.0 : sacc * rdiag_s[o + i];
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) linv[(J * 16 + i) * 17 + c] = x[i];
        }
    };
    // (c) one trailing tile  A_ik -= L_iJ L_kJ^T  on the MFMA, straight from LDS
    auto trailing_tile = [&](int o, int bi, int bk) {
        const int q = lane >> 4, cl = lane & 15;
        const int ri = o + 16 + 16 * bi, rk = o + 16 + 16 * bk;
        f64x4 cacc;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) cacc[rr] = a[(ri + q + 4 * rr) * kSmallLd + rk + cl];
        double av[4], bv[4];
#pragma unroll
        for (int sstep = 0; sstep < 4; ++sstep) {
            av[sstep] = -a[(ri + cl) * kSmallLd + o + 4 * sstep + q];
            bv[sstep] = a[(rk + cl) * kSmallLd + o + 4 * sstep + q];
        }
#pragma unroll
        for (int sstep = 0; sstep < 4; ++sstep) cacc = mfma_f64(av[sstep], bv[sstep], cacc);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) a[(ri + q + 4 * rr) * kSmallLd + rk + cl] = cacc[rr];
    };

    // ---- TRSM operand pack (see vt_trsm_pack), one block pair at a time on the MFMA:
    //      forward  F_ji = -Linv_jj L_ji,  backward  B_ji = -Linv_jj^T L_ij^T  (16x16x16 products), the
    //      accumulator element (row, col) scattered to its slot of the A-operand image.  Everything a
    //      pair of block row j needs (Linv_jj, block row j and block column j of L) is final one panel
    //      step after j, so the rows are emitted inside the factorisation by the waves that idle there.
    const bool lastperm = (D & 15) == 0;
    auto emit_pair = [&](int j, int i, bool bwd) __attribute__((always_inline)) {
        const int q = lane >> 4, c = lane & 15;
        const int PI = bwd ? NBT + vt_tri(i, j) : vt_tri(j, i);
        f64x4 m = {0.0, 0.0, 0.0, 0.0};
        if (i == j) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                m[r] = bwd ? linv[(j * 16 + c) * 17 + q + 4 * r] : linv[(j * 16 + q + 4 * r) * 17 + c];
        } else {
            double av[4], bv[4];
#pragma unroll
            for (int sstep = 0; sstep < 4; ++sstep) {
                const int k = 4 * sstep + q;
                if (!bwd) {
                    av[sstep] = -linv[(j * 16 + c) * 17 + k];
                    bv[sstep] = a[(16 * j + k) * kSmallLd + 16 * i + c];
                } else {
                    av[sstep] = -linv[(j * 16 + k) * 17 + c];
                    bv[sstep] = a[(16 * i + c) * kSmallLd + 16 * j + k];
                }
            }
#pragma unroll
            for (int sstep = 0; sstep < 4; ++sstep) m = mfma_f64(av[sstep], bv[sstep], m);
        }
        const bool pj = j < NB - 1 || lastperm, pi = i < NB - 1 || lastperm;
        const int kk = vt_perm(pi, c);                     // vt_perm is an involution
        double* dst = pack + ((int64_t)PI * 4 + (kk >> 2)) * 64 + (kk & 3) * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[vt_perm(pj, q + 4 * r)] = m[r];
    };
    // the NB + 1 pairs of block row j: forward (j, 0..j), backward (j, j..NB-1); slot of nslots waves
    auto emit_row = [&](int j, int slot, int nslots) {
        for (int e = slot; e < NB + 1; e += nslots) {
            if (e <= j) emit_pair(j, e, false);
            else emit_pair(j, e - 1, true);
        }
    };

    if (!do_factor) {
        // A already holds L: only the inverted diagonal blocks are needed
        if (tid < NP) rdiag_s[tid] = vt_rcp(a[tid * kSmallLd + tid]);
        __syncthreads();
        for (int J = wave; J < NB; J += 4) diag_invert(J);
        __syncthreads();
        if (pack != nullptr)
            for (int j = 0; j < NB; ++j) emit_row(j, wave, 4);
    } else {
        if (wave == 0 && !(VT_PABL & 1)) diag_factor(0);
        __syncthreads();
        for (int J = 0; J < NB && !s_bad; ++J) {
            const int o = 16 * J;
            if (J + 1 == NB) break;
            // ---- (b) panel: row R of the blocks below:  t L_JJ^T = A[R][o..o+15]  by substitution ----
            const int R = o + 16 + tid;
            if (R < NP && !(VT_PABL & 2)) {
                // Column-oriented, with the column of L_JJ one step ahead in registers: left to the compiler this
                // became 16 dot products, each behind its own LDS round trip and with a dependent FMA chain as long
                // as the row (D = 100: 45.1 -> 43.2 us for the kernel).  The reads are uniform (broadcast) and inline
                // asm so that column m + 1 is in flight under the updates of column m.
                double t[16], rd[16];
#pragma unroll
                for (int m = 0; m < 16; ++m) t[m] = a[R * kSmallLd + o + m];
#pragma unroll
                for (int m = 0; m < 16; ++m) rd[m] = rdiag_s[o + m];
                if constexpr (ASMP) {
                typedef __attribute__((address_space(3))) void* ldsptr_t;
                const unsigned lb = (unsigned)(uintptr_t)(ldsptr_t)(a + o * kSmallLd + o);   // L_JJ[0][0]
                double Lc[2][16];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int cc = 1; cc < 16; ++cc)
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(Lc[0][cc]) : "v"(lb), "n"(cc * kSmallLd * 8));
#pragma unroll
                for (int m = 0; m < 15; ++m) {
#pragma unroll
                    for (int cc = m + 2; cc < 16; ++cc)   // column m + 1
                        asm volatile("ds_read_b64 %0, %1 offset:%2"
                                     : "=v"(Lc[(m + 1) & 1][cc])
                                     : "v"(lb), "n"((cc * kSmallLd + m + 1) * 8));
                    // LDS returns in order: all but the 14 - m reads just issued have landed
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(14 - m) : "memory");
#pragma unroll
                    for (int cc = m + 1; cc < 16; ++cc) asm volatile("" : "+v"(Lc[m & 1][cc]));
                    t[m] *= rd[m];
#pragma unroll
                    for (int cc = m + 1; cc < 16; ++cc) t[cc] = fma(-t[m], Lc[m & 1][cc], t[cc]);
                }
                } else {
#pragma unroll
                    for (int m = 0; m < 15; ++m) {
                        t[m] *= rd[m];
#pragma unroll
                        for (int cc = m + 1; cc < 16; ++cc) t[cc] = fma(-t[m], a[(o + cc) * kSmallLd + o + m], t[cc]);
                    }
                }
                t[15] *= rd[15];
#pragma unroll
                for (int cc = 0; cc < 16; ++cc) a[R * kSmallLd + o + cc] = t[cc];
            }
            __syncthreads();
            // ---- (c) trailing update with look-ahead: wave 0 updates the next diagonal block first and
            //      factors it while waves 1..3 update the rest of the trailing matrix and wave 1 inverts
            //      the block the panel was just solved against (needed by the TRSM pack only) --------------
            const int nb = NB - J - 1;
            const int nt = nb * (nb + 1) / 2;
            if (wave == 0) {
                if (!(VT_PABL & 4)) trailing_tile(o, 0, 0);
                if (!(VT_PABL & 1)) diag_factor(J + 1);
            } else {
                if (!(VT_PABL & 4)) {
                    for (int t = wave; t < nt; t += 3) {    // tiles 1..nt-1 over waves 1..3
                        int bi = 0;
                        while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
                        trailing_tile(o, bi, t - bi * (bi + 1) / 2);
                    }
                }
                if (pack != nullptr && !(VT_PABL & 8)) {
                    if (wave == 1) diag_invert(J);
                    else if (J >= 1) emit_row(J - 1, wave - 2, 2);   // block row J-1 has been final for a step
                }
            }
            __syncthreads();
        }
        // the last diagonal block was factored by the final look-ahead (or is block 0 when NB = 1);
        // block rows NB-2 and NB-1 of the pack are still to be emitted
        if (pack != nullptr && !s_bad && !(VT_PABL & 8)) {
            if (wave == 0) diag_invert(NB - 1);
            else if (NB >= 2) emit_row(NB - 2, wave - 1, 3);
            __syncthreads();
            emit_row(NB - 1, wave, 4);
        }
    }
    __syncthreads();
    const int bad = s_bad;
    if (do_factor) {
        const bool oop = Ain != A;
        for (int i = tid >> 7; i < D; i += 2) {
            const int j = tid & 127;
            if (j <= i) A[(int64_t)i * lda + j] = a[i * kSmallLd + j];
            else if (oop && j < D) A[(int64_t)i * lda + j] = 0.0;
        }
        if (tid == 0 && info) *info = bad;
    }
}

// ------------------------------------------------------------------ potrs -------
// Generic (any strides, any D that fits LDS): one wave per right-hand side, the
// solution vector lives in LDS, the 64 lanes split each row's dot product.
__global__ __launch_bounds__(64) void k_potrs_generic(const double* L, int ldl, int D, double* B,
                                                      int64_t ldr, int64_t ldc, int64_t nrhs) {
    extern __shared__ double x[];
    const int lane = threadIdx.x;
    for (int64_t k = blockIdx.x; k < nrhs; k += gridDim.x) {
        double* b = B + k * ldc;
        for (int i = lane; i < D; i += 64) x[i] = b[(int64_t)i * ldr];
        __syncthreads();
        for (int i = 0; i < D; ++i) {  // L y = b
            const double* Li = L + (int64_t)i * ldl;
            double p = 0.0;
            for (int j = lane; j < i; j += 64) p = fma(Li[j], x[j], p);
            p = vt_sum64(p);
            __syncthreads();
            if (lane == 0) x[i] = (x[i] - p) / Li[i];
            __syncthreads();
        }
        for (int i = D - 1; i >= 0; --i) {  // L^T z = y
            double p = 0.0;
            for (int j = i + 1 + lane; j < D; j += 64) p = fma(L[(int64_t)j * ldl + i], x[j], p);
            p = vt_sum64(p);
            __syncthreads();
            if (lane == 0) x[i] = (x[i] - p) / L[(int64_t)i * ldl + i];
            __syncthreads();
        }
        for (int i = lane; i < D; i += 64) b[(int64_t)i * ldr] = x[i];
        __syncthreads();
    }
}

// -------------------------------------------------------------- TRSM pack -------
// Logical -> hardware index maps inside a 16-wide block (vt_perm).  A "permuted" (full)
// block keeps logical index 4q+s at accumulator register s of lane group q, so a lane owns
// 4 CONSECUTIVE columns (32 contiguous bytes of a row of X / S); the trailing partial block
// keeps the native MFMA order 4s+q so that its all-zero k-steps can be skipped.
//
// pack[(PI*4 + s)*64 + lane]: A operand of k-step s for block pair PI.
//   forward  pair (j,i), i<=j : PI = tri(j,i)        F_jj = Linv_jj,  F_ji = -Linv_jj L_ji
//   backward pair (j,i), i>=j : PI = NBT + tri(i,j)  B_jj = Linv_jj^T, B_ji = -Linv_jj^T L_ij^T
// so that   Y_j = sum_{i<=j} F_ji (G_i | Y_i)   and   Z_j = B_jj Y_j + sum_{i>j} B_ji Z_i
// are pure MFMA accumulations (no divisions, no subtractions in the solve kernel).
// Emitted by k_potrf_pack_small above.

// -------------------------------------------------------------- TRSM rows -------
struct TrsmArgs {
    const double* pack;
    const double* B;
    int64_t ldb;
    const double* rowscale;
    double alpha;
    int64_t N;
    int D;
    double* Out;
    int64_t ldo;
    int mode;   // 0: full solve (L L^T)^{-1};  1: forward only  L^{-1};  2: backward only  L^{-T}
    int rev;    // walk the row tiles from the last one down (see launch_trsm)
};

// Out[n,:] = alpha * s_n * (L L^T)^{-1} B[n,:].   D <= 128.
// Each wave solves one tile of 16 right-hand sides at a time.  The right-hand sides
// sit on the MFMA "column" (lane & 15) axis, the parameter index on the "row" axis,
// so a solved block Y_i (an accumulator) IS the B operand of the following
// products -- register r of the accumulator is k-step r -- and the whole forward /
// backward substitution is a chain of v_mfma_f64_16x16x4_f64 with A operands read
// from the LDS-resident pack.  Nothing but X (or B) is read from HBM and nothing
// but S is written.
//
// The chain is a compile-time schedule of 2*NBT block products; product k+1's A
// operands are read from LDS while product k's MFMAs run, and sched_barriers keep
// hipcc from hoisting all ~200 ds_reads to the top (which spills hundreds of VGPRs).
template <int NB, int KL, bool LASTPERM>
struct TrsmState {
    static constexpr int NSET = TrsmSched2<NB, KL, LASTPERM>::NSET;
    f64x4 V[NB];
    f64x4 acc;
    double cur[4][NSET], nxt[4][NSET];
    const double* lds;
    int lane, q;
    int slane[NSET];   // lane of the A image a strip product reads
    double* orow;  // output row of this lane
    bool store_fwd, store_bwd;   // store Y after the forward chain (forward-only solves) / Z after the backward chain
    int D;
};

template <int J, int NB, int KL, bool LASTPERM>
__device__ __forceinline__ void trsm_store(TrsmState<NB, KL, LASTPERM>& st) {
    using S = TrsmSched<NB, KL, LASTPERM>;
    if constexpr (S::permuted(J)) {
#if VT_TRSM_NT_STORE
        __builtin_nontemporal_store((f64x2u){st.acc[0], st.acc[1]}, (f64x2u*)(st.orow + 16 * J + 4 * st.q));
        __builtin_nontemporal_store((f64x2u){st.acc[2], st.acc[3]}, (f64x2u*)(st.orow + 16 * J + 4 * st.q + 2));
#else
        *(f64x2u*)(st.orow + 16 * J + 4 * st.q) = (f64x2u){st.acc[0], st.acc[1]};
        *(f64x2u*)(st.orow + 16 * J + 4 * st.q + 2) = (f64x2u){st.acc[2], st.acc[3]};
#endif
    } else {
#pragma unroll
        for (int s = 0; s < KL; ++s) {
            int col = 16 * J + 4 * s + st.q;
            if (col < st.D) st.orow[col] = st.acc[s];
        }
    }
}

// A operand(s) of product IDX -> dst
template <int IDX, int NB, int KL, bool LASTPERM>
__device__ __forceinline__ void trsm_fetch(TrsmState<NB, KL, LASTPERM>& st, double (&dst)[4][TrsmState<NB, KL, LASTPERM>::NSET]) {
    using S = TrsmSched2<NB, KL, LASTPERM>;
    constexpr int img = S::image(IDX);
    constexpr int nks = S::ks(S::src(IDX));
    if constexpr (S::strip_row(IDX)) {
#pragma unroll
        for (int s = 0; s < nks; ++s)
#pragma unroll
            for (int r = 0; r < S::STRIPS; ++r) dst[s][r] = st.lds[(img * 4 + s) * 64 + st.slane[r]];
    } else {
#pragma unroll
        for (int s = 0; s < nks; ++s) dst[s][0] = st.lds[(img * 4 + s) * 64 + st.lane];
    }
}

// Products [IDX, END) of the schedule; st.cur holds the A operand of product IDX on entry.
template <int IDX, int END, int NB, int KL, bool LASTPERM>
__device__ __forceinline__ void trsm_chain(TrsmState<NB, KL, LASTPERM>& st) {
    using S = TrsmSched2<NB, KL, LASTPERM>;
    if constexpr (IDX < END) {
        if constexpr (IDX + 1 < END) trsm_fetch<IDX + 1, NB, KL, LASTPERM>(st, st.nxt);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int i = S::src(IDX), j = S::row(IDX);
        constexpr int cks = S::ks(i);
        constexpr bool is_first = S::first(IDX), is_last = S::last(IDX), is_bwd = S::bwd(IDX);
        if constexpr (is_first) st.acc = (f64x4){0.0, 0.0, 0.0, 0.0};
        if constexpr (S::strip_row(IDX)) {
#pragma unroll
            for (int s = 0; s < cks; ++s)
#pragma unroll
                for (int r = 0; r < S::STRIPS; ++r)
                    st.acc[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(st.cur[s][r], st.V[i][s], st.acc[r], 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < cks; ++s) st.acc = mfma_f64(st.cur[s][0], st.V[i][s], st.acc);
        }
        if constexpr (is_last) {
            st.V[j] = st.acc;
            if (is_bwd ? st.store_bwd : st.store_fwd) trsm_store<j, NB, KL, LASTPERM>(st);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int r = 0; r < S::NSET; ++r) st.cur[s][r] = st.nxt[s][r];
        trsm_chain<IDX + 1, END, NB, KL, LASTPERM>(st);
    }
}

template <int NB, int KL, bool LASTPERM, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_trsm_rows_small(TrsmArgs a) {
    using S = TrsmSched<NB, KL, LASTPERM>;
    constexpr int NBT = S::NBT;
    __shared__ double lds[2 * NBT * 256];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int64_t W = (int64_t)gridDim.x * WAVES;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t ngroups = (a.N + 15) / 16;

    TrsmState<NB, KL, LASTPERM> st;
    st.lds = lds;
    st.lane = lane;
    st.q = q;
    st.D = a.D;
#pragma unroll
    for (int r = 0; r < TrsmState<NB, KL, LASTPERM>::NSET; ++r) st.slane[r] = (lane & 48) | (4 * r + (lane & 3));

    // Right-hand sides of the next PF tiles (prefetch ring, Gn[0] = the next one).  Narrow designs are HBM-bound and a
    // 16-row tile is only 128 NB bytes per lane group: one tile ahead leaves 20 KB in flight per CU at NB = 2, so
    // they look further ahead (NB <= 2: 3 tiles, NB <= 4: 2) -- the registers are there.
    constexpr int PF = VT_TRSM_PF_ALL ? VT_TRSM_PF_ALL : (NB <= 2 ? 3 : NB <= 4 ? 2 : 1);
    f64x4 Gn[PF][NB];
    double coefn[PF];

    const int64_t last = ngroups - 1;       // (groups past the end re-read the last one: loaded, never used)
    // Unconditional loads with clamped indices (no branch per load): rows past N
    // re-read row N-1 and are zeroed through coef = 0.
    auto load_group = [&](int64_t grp, f64x4 (&G)[NB], double& coef) {
        if (a.rev) grp = last - grp;
        int64_t row = grp * 16 + n;
        bool rowok = row < a.N;
        int64_t rowc = rowok ? row : a.N - 1;
        const double* br = a.B + rowc * a.ldb;
        double sc = a.rowscale ? a.rowscale[rowc] : 1.0;
        coef = rowok ? a.alpha * sc : 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            f64x4 v = {0.0, 0.0, 0.0, 0.0};
            if constexpr (S::permuted(NB - 1)) {
                f64x2u lo = *(const f64x2u*)(br + 16 * b + 4 * q);
                f64x2u hi = *(const f64x2u*)(br + 16 * b + 4 * q + 2);
                v = (f64x4){lo[0], lo[1], hi[0], hi[1]};
            } else {
                if (b < NB - 1) {
                    f64x2u lo = *(const f64x2u*)(br + 16 * b + 4 * q);
                    f64x2u hi = *(const f64x2u*)(br + 16 * b + 4 * q + 2);
                    v = (f64x4){lo[0], lo[1], hi[0], hi[1]};
                } else {
#pragma unroll
                    for (int s = 0; s < KL; ++s) {
                        int col = 16 * b + 4 * s + q;
                        double x = br[min(col, a.D - 1)];
                        v[s] = col < a.D ? x : 0.0;
                    }
                }
            }
            G[b] = v;
        }
    };

    constexpr bool PREFETCH = WAVES <= 8;   // more waves per SIMD: no register room for (nor need of) a prefetched tile
    int64_t grp = gw;
    if constexpr (PREFETCH) {
#pragma unroll
        for (int f = 0; f < PF; ++f) load_group(min(grp + f * W, last), Gn[f], coefn[f]);
    }
    {   // the pack -> LDS, behind the first tiles' loads
        const f64x2u* src = (const f64x2u*)a.pack;
        f64x2u* dst = (f64x2u*)lds;
        for (int i = threadIdx.x; i < NBT * 256; i += WAVES * 64) dst[i] = src[i];
    }
    __syncthreads();
    // one tile: its right-hand side leaves the ring slot, the slot is refilled PF tiles ahead at once (the ring is
    // walked by UNROLLING, not by moving registers: a move would wait for the load it moves)
    auto tile = [&](int64_t g, f64x4 (&G)[NB], double& coef) {
        if constexpr (!PREFETCH) load_group(g, G, coef);
#pragma unroll
        for (int b = 0; b < NB; ++b) st.V[b] = G[b] * coef;
        if constexpr (PREFETCH) load_group(min(g + PF * W, last), G, coef);
        if (a.rev) g = last - g;
        int64_t row = g * 16 + n;
        const bool rowok = row < a.N;
        st.store_fwd = rowok && a.mode == 1;
        st.store_bwd = rowok && a.mode != 1;
        st.orow = a.Out + (rowok ? row : 0) * a.ldo;
        if (a.mode != 2) {   // forward substitution  L Y = G
            trsm_fetch<0, NB, KL, LASTPERM>(st, st.cur);
            trsm_chain<0, NBT, NB, KL, LASTPERM>(st);
        }
        if (a.mode != 1) {   // backward substitution  L^T Z = Y
            trsm_fetch<NBT, NB, KL, LASTPERM>(st, st.cur);
            trsm_chain<NBT, 2 * NBT, NB, KL, LASTPERM>(st);
        }
    };
    while (grp < ngroups) {
#pragma unroll
        for (int f = 0; f < PF; ++f) {
            if (grp < ngroups) {
                tile(grp, Gn[f], coefn[f]);
                grp += W;
            }
        }
    }
}

inline int num_cu() { return vt_num_cu(); }   // of the current device

// Two waves per SIMD, one 16-row tile each: everything a wave keeps outside the
// accumulator must fit the 256 architectural VGPRs, and the partner wave's MFMA
// chain fills this wave's load / dependency stalls.
// dev A/B (tools/build_abl.cmd): 12 / 16 waves (3 / 4 per SIMD, no tile prefetch, 168 / 128 VGPRs) measured
// 0.415 / 0.438 ms against 0.400 ms for 8 waves with the prefetched tile at N = 1e6, D = 100.
#ifndef VT_TRSM_WAVES
#define VT_TRSM_WAVES 8
#endif
constexpr int kTrsmWaves = VT_TRSM_WAVES;

template <int NB, int KL, bool LASTPERM>
void launch_trsm(const TrsmArgs& a, hipStream_t st) {
    int64_t ngroups = (a.N + 15) / 16;
    int64_t g = (ngroups + kTrsmWaves - 1) / kTrsmWaves;
    static const int wgs_env = getenv("VT_TRSM_WGS") ? atoi(getenv("VT_TRSM_WGS")) : 0;   // dev switch
    // one 8-wave workgroup per CU is what the chains want; the one-block solves (D <= 16) hardly use the MFMA pipe
    // and gain from more loads in flight (N = 4e6: D = 4 0.096 -> 0.049 ms at 4 per CU, D = 8 0.121 -> 0.103 at 2;
    // from D = 20 on more workgroups measured 2-10 % slower)
    const int wgs = wgs_env > 0 ? wgs_env : (NB == 1 ? (a.D <= 4 ? 4 : 2) : 1);
    if (g > (int64_t)num_cu() * wgs) g = (int64_t)num_cu() * wgs;
    if (g < 1) g = 1;
    hipLaunchKernelGGL((k_trsm_rows_small<NB, KL, LASTPERM, kTrsmWaves>), dim3((int)g),
                       dim3(kTrsmWaves * 64), 0, st, a);
}

template <int NB>
int dispatch_trsm_kl(const TrsmArgs& a, hipStream_t st) {
    int rem = a.D - 16 * (NB - 1);  // 1..16 columns in the last block
    if (rem == 16) {
        launch_trsm<NB, 4, true>(a, st);
        return VT_OK;
    }
    switch ((rem + 3) / 4) {
        case 1: launch_trsm<NB, 1, false>(a, st); break;
        case 2: launch_trsm<NB, 2, false>(a, st); break;
        case 3: launch_trsm<NB, 3, false>(a, st); break;
        default: launch_trsm<NB, 4, false>(a, st); break;
    }
    return VT_OK;
}

}  // namespace

static void launch_potrf_pack_small(const double* Ain, int ldin, double* A, int lda, int D, int* info, double* pack,
                                    int do_factor, hipStream_t st) {
    static const bool c_panel = getenv("VT_POTRF_C_PANEL") != nullptr;   // checking switch, read once per process
    if (c_panel)
        hipLaunchKernelGGL(k_potrf_pack_small<false>, dim3(1), dim3(256), 0, st, Ain, ldin, A, lda, D, info, pack, do_factor);
    else
        hipLaunchKernelGGL(k_potrf_pack_small<true>, dim3(1), dim3(256), 0, st, Ain, ldin, A, lda, D, info, pack, do_factor);
}

int vt_potrf_pack_block(double* A, int lda, int D, int* info, double* pack, int do_factor, hipStream_t st) {
    launch_potrf_pack_small((const double*)A, lda, A, lda, D, info, pack, do_factor, st);
    return vt_launch_status();
}

int vt_trsm_rows_mode(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha,
                      int64_t N, double* Out, int64_t ldo, int mode, hipStream_t st) {
    if (N == 0) return VT_OK;
    // Row tiles from the last one down: the pass that produced H (and, in a loop of steps, the next one) walks X from
    // the first row up, so the rows it touched last are the ones still in the 256 MB memory-side cache.
    static const bool fwd_order = getenv("VT_TRSM_ASCENDING") != nullptr;   // dev A/B switch, read once
    TrsmArgs a{(const double*)pack, B, ldb, rowscale, alpha, N, D, Out, ldo, mode, fwd_order ? 0 : 1};
    switch ((D + 15) / 16) {
        case 1: dispatch_trsm_kl<1>(a, st); break;
        case 2: dispatch_trsm_kl<2>(a, st); break;
        case 3: dispatch_trsm_kl<3>(a, st); break;
        case 4: dispatch_trsm_kl<4>(a, st); break;
        case 5: dispatch_trsm_kl<5>(a, st); break;
        case 6: dispatch_trsm_kl<6>(a, st); break;
        case 7: dispatch_trsm_kl<7>(a, st); break;
        case 8: dispatch_trsm_kl<8>(a, st); break;
        default: return VT_ERR_UNSUPPORTED;
    }
    return vt_launch_status();
}

constexpr int kLargeDMax = 8192;   // D x D images of L and L^T live in the pack

extern "C" size_t vt_potrf_workspace_bytes(int D) {
    return D > VT_SMALL_D_MAX ? vt_large_potrf_ws_bytes(D) : 0;
}

extern "C" int vt_potrf(double* A, int lda, int D, int* info, void* ws, size_t ws_bytes, void* stream) {
    if (!A || !info || D < 1 || lda < D) return VT_ERR_BAD_ARG;
    if (D > kLargeDMax) return VT_ERR_UNSUPPORTED;
    if (D > VT_SMALL_D_MAX) {
        if (!ws || ws_bytes < vt_large_potrf_ws_bytes(D)) return VT_ERR_WORKSPACE;
        return vt_large_potrf(A, lda, D, info, ws, (hipStream_t)stream);
    }
    return vt_potrf_pack_block(A, lda, D, info, nullptr, 1, (hipStream_t)stream);
}

extern "C" int vt_potrs(const double* L, int ldl, int D, double* B, int64_t ldb_row, int64_t ldb_col,
                        int64_t nrhs, void* stream) {
    if (!L || !B || D < 1 || ldl < D || nrhs < 0) return VT_ERR_BAD_ARG;
    if (nrhs == 0) return VT_OK;
    size_t shmem = (size_t)D * sizeof(double);
    if (shmem > 160 * 1024) return VT_ERR_UNSUPPORTED;
    if (shmem > 64 * 1024)
        VT_CHECK_HIP(hipFuncSetAttribute((const void*)k_potrs_generic, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)shmem));
    int64_t grid = nrhs < 4096 ? nrhs : 4096;
    hipLaunchKernelGGL(k_potrs_generic, dim3((int)grid), dim3(64), shmem, (hipStream_t)stream, L, ldl, D, B,
                       ldb_row, ldb_col, nrhs);
    return vt_launch_status();
}

extern "C" size_t vt_trsm_pack_bytes(int D) {
    if (D < 1 || D > kLargeDMax) return 0;
    if (vt_mid_trsm_ok(D)) return std::max(vt_mid_pack_bytes(D), vt_large_pack_bytes(D));
    if (D > VT_SMALL_D_MAX) return vt_large_pack_bytes(D);
    int NB = (D + 15) / 16;
    return (size_t)2 * (NB * (NB + 1) / 2) * 256 * sizeof(double);
}

extern "C" int vt_trsm_pack(const double* L, int ldl, int D, void* pack, void* stream) {
    if (!L || !pack || D < 1 || ldl < D) return VT_ERR_BAD_ARG;
    if (D > kLargeDMax) return VT_ERR_UNSUPPORTED;
    if (vt_mid_trsm_ok(D)) return vt_mid_pack(L, ldl, D, pack, (hipStream_t)stream);
    if (D > VT_SMALL_D_MAX) return vt_large_pack(L, ldl, D, pack, (hipStream_t)stream);
    return vt_potrf_pack_block((double*)L, ldl, D, nullptr, (double*)pack, 0, (hipStream_t)stream);
}

extern "C" int vt_potrf_pack(double* A, int lda, int D, int* info, void* pack, void* stream) {
    if (!A || !info || !pack || D < 1 || lda < D) return VT_ERR_BAD_ARG;
    if (D > kLargeDMax) return VT_ERR_UNSUPPORTED;
    if (D > VT_SMALL_D_MAX) {
        // the L / L^T images of the pack (2 D^2 doubles; the chain images of a mid-range pack) are free until the
        // pack kernels fill them: they serve as the factorisation's scratch.
        int rc = vt_large_potrf(A, lda, D, info, pack, (hipStream_t)stream);
        if (rc != VT_OK) return rc;
        if (vt_mid_trsm_ok(D)) return vt_mid_pack(A, lda, D, pack, (hipStream_t)stream);
        return vt_large_pack(A, lda, D, pack, (hipStream_t)stream);
    }
    return vt_potrf_pack_block(A, lda, D, info, (double*)pack, 1, (hipStream_t)stream);
}

extern "C" int vt_potrf_pack_from(const double* H, int ldh, double* L, int ldl, int D, int* info, void* pack,
                                  void* stream) {
    if (!H || !L || !info || !pack || D < 1 || ldh < D || ldl < D) return VT_ERR_BAD_ARG;
    if (D > kLargeDMax) return VT_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (D > VT_SMALL_D_MAX || (const double*)L == H) {
        if ((const double*)L != H)
            VT_CHECK_HIP(hipMemcpy2DAsync(L, (size_t)ldl * sizeof(double), H, (size_t)ldh * sizeof(double),
                                          (size_t)D * sizeof(double), D, hipMemcpyDeviceToDevice, st));
        return vt_potrf_pack(L, ldl, D, info, pack, stream);
    }
    launch_potrf_pack_small(H, ldh, L, ldl, D, info, (double*)pack, 1, st);
    return vt_launch_status();
}

extern "C" int vt_trsm_rows(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale,
                            double alpha, int64_t N, double* Out, int64_t ldo, void* stream) {
    if (!pack || D < 1 || ldb < D || ldo < D || N < 0) return VT_ERR_BAD_ARG;
    if (D > kLargeDMax) return VT_ERR_UNSUPPORTED;
    if (N == 0) return VT_OK;
    if (!B || !Out) return VT_ERR_BAD_ARG;
    if (vt_mid_trsm_ok(D)) return vt_mid_trsm_rows(pack, D, B, ldb, rowscale, alpha, N, Out, ldo, (hipStream_t)stream);
    if (D > VT_SMALL_D_MAX)
        return vt_large_trsm_rows(pack, D, B, ldb, rowscale, alpha, N, Out, ldo, (hipStream_t)stream);
    return vt_trsm_rows_mode(pack, D, B, ldb, rowscale, alpha, N, Out, ldo, 0, (hipStream_t)stream);
}
