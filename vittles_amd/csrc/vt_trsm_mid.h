// K2 + K5 for MID-RANGE designs, 129 <= D <= 256 (NB = 9..16):  S[n,:] = alpha s_n (L L^T)^{-1} B[n,:]  in ONE pass
// over B -- every row is read once and its solution written once (16 N D bytes), like the D <= 128 kernel of
// vt_chol.hip.  Replaces the N-right-hand-side cho_solve behind `-1 * self._hess_solver(jac0)`
// (vittles/sensitivity_lib.py:226, vittles/solver_lib.py:28-29) -- see include/vittles_hip.h (vt_trsm_rows).
//
// Round 2 solved these sizes block by block (128 columns at a time: a GEMM-NT update and a small-TRSM sweep per
// block, forward and backward), which read and wrote S six times and paid a full 128-column tile for a 1-column
// remainder: D = 129 ran at 19 % of the fp64-MFMA peak, D = 128 at 65 %.  Here the D <= 128 scheme is kept -- a
// wave holds the 16 x D tile of right-hand sides in registers (the right-hand sides on the MFMA column axis, so a
// solved block IS the B operand of the products that follow), and forward + backward substitution are one
// compile-time chain of 2 NBT block products whose A operands are the packed images F_ji = -Linv_jj L_ji,
// B_ji = -Linv_jj^T L_ij^T -- but the pack (2 NBT images of 2 KB: 557 KB at D = 256) no longer fits the LDS.
// It is laid out in the ORDER OF THE CHAIN and streamed through an LDS ring of 3 slots x 16 images by LDS-DMA
// (buffer_load ... lds from L2), one s_barrier per 16 products: the 8 waves of a workgroup run the same chain on
// their own 16 rows, so a chunk is fetched once per 128 rows of B (4 B/clk/CU of L2 traffic).
//   boundary before chunk c:  my pieces of chunk c have landed (vmcnt) and my reads of chunk c-1 are back (lgkmcnt)
//                             -> s_barrier -> DMA of chunk c+2 into the slot of chunk c-1.
// The operand reads are inline asm (compiler-visible LDS loads would wait for the wave's own LDS-DMA), one product
// ahead of the MFMAs, exactly as in the resident kernel.
#pragma once
#include "vt_common.h"
#include "vt_trsm_sched.h"

namespace {

struct MidTrsmArgs {
    const double* pack;   // [2 NBT][4][64] images in chain order
    const double* B;
    int64_t ldb;
    const double* rowscale;
    double alpha;
    int64_t N;
    int D;
    double* Out;
    int64_t ldo;
    int npass;            // passes of a workgroup (uniform: every wave of the grid runs the same chain count)
};

typedef __attribute__((address_space(3))) void* midt_ldsptr;
constexpr int kMidCH = 16;                  // images per ring slot
constexpr int kMidSlot = kMidCH * 256;      // doubles per slot (32 KB)
constexpr int kMidRingBytes = 3 * kMidSlot * (int)sizeof(double);

template <int NB, int KL, bool LASTPERM, int MODE = 0>
struct MidTrsmState {
    using S2 = TrsmSched2<NB, KL, LASTPERM>;
    static constexpr int NSET = S2::NSET;
    // MODE 0: the whole chain (forward + backward substitution); 1: forward only (Y = L^-1 G stored); 2: backward only
    // (Z = L^-T Y) -- the two halves of a 256-column diagonal block of the blocked solve for D > 256.  The pack
    // pointer the kernel gets starts at image BEG; ring slots and chunks count from there.
    static constexpr int BEG = MODE == 2 ? S2::NBT : 0;
    static constexpr int END = MODE == 1 ? S2::NBT : S2::M;
    static constexpr int M = END - BEG;
    static constexpr int NCH = (M + kMidCH - 1) / kMidCH;
    f64x4 V[NB];
    f64x4 acc;
    double cur[4][NSET], nxt[4][NSET];
    unsigned rd[3];            // LDS byte address of this lane's element of image 0 of the slot of chunk (c % 3)
    unsigned rds[3][NSET];     // the same for the strip products' lane map
    int sb[3];                 // ring slot (0..2) of chunk (c % 3) of this pass
    double* ring;
    const double* pack;
    int wave, lane, q, D;
    double* orow;
    bool store;
    bool has_next;

    // this wave's 4 of the 32 1-KB pieces of chunk `chunk` of the stream -> ring slot `slot`
    __device__ __forceinline__ void issue(int chunk, int slot) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pack, 0, (unsigned)(M * 2048), 0x00020000);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int piece = wave * 4 + e;
            double* dst = ring + slot * kMidSlot + piece * 128;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (midt_ldsptr)dst, 16, 16u * lane,
                                                     (unsigned)(chunk * (kMidCH * 2048) + piece * 1024), 0, 0);
        }
    }
};

template <int J, int NB, int KL, bool LASTPERM, int MODE>
__device__ __forceinline__ void midt_store(MidTrsmState<NB, KL, LASTPERM, MODE>& st) {
    using S = TrsmSched<NB, KL, LASTPERM>;
    if constexpr (S::permuted(J)) {
        *(f64x2u*)(st.orow + 16 * J + 4 * st.q) = (f64x2u){st.acc[0], st.acc[1]};
        *(f64x2u*)(st.orow + 16 * J + 4 * st.q + 2) = (f64x2u){st.acc[2], st.acc[3]};
    } else {
#pragma unroll
        for (int s = 0; s < KL; ++s) {
            int col = 16 * J + 4 * s + st.q;
            if (col < st.D) st.orow[col] = st.acc[s];
        }
    }
}

// before the first read of chunk C of this pass
template <int C, int NB, int KL, bool LASTPERM, int MODE>
__device__ __forceinline__ void midt_boundary(MidTrsmState<NB, KL, LASTPERM, MODE>& st) {
    constexpr int NCH = MidTrsmState<NB, KL, LASTPERM, MODE>::NCH;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    constexpr int cn = C + 2;
    if constexpr (cn < NCH) {
        st.issue(cn, st.sb[cn % 3]);
    } else {
        if (st.has_next) st.issue(cn - NCH, st.sb[cn % 3]);   // chunks 0, 1 of the next pass
    }
}

// A operand(s) of product IDX -> dst (reads in flight: midt_land waits for them)
template <int IDX, int NB, int KL, bool LASTPERM, int MODE>
__device__ __forceinline__ void midt_fetch(MidTrsmState<NB, KL, LASTPERM, MODE>& st,
                                           double (&dst)[4][MidTrsmState<NB, KL, LASTPERM, MODE>::NSET]) {
    using S = TrsmSched2<NB, KL, LASTPERM>;
    constexpr int REL = IDX - MidTrsmState<NB, KL, LASTPERM, MODE>::BEG;
    constexpr int nks = S::ks(S::src(IDX));
    constexpr int slot = (REL / kMidCH) % 3;
    constexpr int base = (REL % kMidCH) * 2048;
    if constexpr (S::strip_row(IDX)) {
#pragma unroll
        for (int s = 0; s < nks; ++s)
#pragma unroll
            for (int r = 0; r < S::STRIPS; ++r)
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst[s][r]) : "v"(st.rds[slot][r]), "n"(base + 512 * s));
    } else {
#pragma unroll
        for (int s = 0; s < nks; ++s)
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst[s][0]) : "v"(st.rd[slot]), "n"(base + 512 * s));
    }
}
template <int IDX, int NB, int KL, bool LASTPERM, int MODE>
__device__ __forceinline__ void midt_land(MidTrsmState<NB, KL, LASTPERM, MODE>& st,
                                          double (&dst)[4][MidTrsmState<NB, KL, LASTPERM, MODE>::NSET]) {
    using S = TrsmSched2<NB, KL, LASTPERM>;
    constexpr int nks = S::ks(S::src(IDX));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // every operand is tied to a volatile asm BEHIND the wait (volatile asms keep their order)
#pragma unroll
    for (int s = 0; s < nks; ++s) {
        if constexpr (S::strip_row(IDX)) {
#pragma unroll
            for (int r = 0; r < S::STRIPS; ++r) asm volatile("" : "+v"(dst[s][r]));
        } else {
            asm volatile("" : "+v"(dst[s][0]));
        }
    }
}

// Products [IDX, END) of the pass; st.cur holds the A operand of product IDX on entry.
template <int IDX, int NB, int KL, bool LASTPERM, int MODE>
__device__ __forceinline__ void midt_chain(MidTrsmState<NB, KL, LASTPERM, MODE>& st) {
    using S = TrsmSched2<NB, KL, LASTPERM>;
    using ST = MidTrsmState<NB, KL, LASTPERM, MODE>;
    constexpr int END = ST::END, BEG = ST::BEG;
    if constexpr (IDX < END) {
        constexpr bool more = IDX + 1 < END;
        constexpr bool edge = more && ((IDX + 1 - BEG) % kMidCH == 0);   // product IDX + 1 opens a chunk
        if constexpr (more && !edge) midt_fetch<IDX + 1, NB, KL, LASTPERM, MODE>(st, st.nxt);
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
            if ((is_bwd || MODE == 1) && st.store) midt_store<j, NB, KL, LASTPERM, MODE>(st);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (edge) {
            // (the MFMAs just issued cover the barrier; the chunk's first operands are read behind it)
            midt_boundary<(IDX + 1 - BEG) / kMidCH, NB, KL, LASTPERM, MODE>(st);
            midt_fetch<IDX + 1, NB, KL, LASTPERM, MODE>(st, st.nxt);
        }
        if constexpr (more) {
            midt_land<IDX + 1, NB, KL, LASTPERM, MODE>(st, st.nxt);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int r = 0; r < S::NSET; ++r) st.cur[s][r] = st.nxt[s][r];
        }
        midt_chain<IDX + 1, NB, KL, LASTPERM, MODE>(st);
    }
}

template <int NB, int KL, bool LASTPERM, int MODE = 0>
__global__ __launch_bounds__(512) void k_trsm_rows_mid(MidTrsmArgs a) {
    using S = TrsmSched<NB, KL, LASTPERM>;
    using ST = MidTrsmState<NB, KL, LASTPERM, MODE>;
    constexpr int NCH = ST::NCH;
    static_assert(NCH >= 3, "the ring holds three chunks");
    extern __shared__ double ring[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, n = lane & 15;
    const int64_t ngroups = (a.N + 15) / 16;

    ST st;
    st.ring = ring;
    st.pack = a.pack;
    st.wave = wave;
    st.lane = lane;
    st.q = q;
    st.D = a.D;
    unsigned slot_rd[3], slot_rds[3][ST::NSET];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        slot_rd[t] = (unsigned)(uintptr_t)(midt_ldsptr)(ring + t * kMidSlot + lane);
#pragma unroll
        for (int r = 0; r < ST::NSET; ++r)
            slot_rds[t][r] = (unsigned)(uintptr_t)(midt_ldsptr)(ring + t * kMidSlot + ((lane & 48) | (4 * r + (lane & 3))));
    }
    int rot = 0;   // ring slot of chunk 0 of the current pass

    constexpr bool PREFETCH = NB <= 12;   // beyond, two tiles do not fit the 256 registers of two waves per SIMD
    f64x4 Gn[PREFETCH ? NB : 1];
    double coefn = 0.0;
    // Unconditional loads with clamped indices (no branch per load): rows past N re-read row N-1 and are zeroed
    // through coef = 0.
    auto load_group = [&](int64_t grp, f64x4* G, double& coef) {
        int64_t row = grp * 16 + n;
        bool rowok = row < a.N;
        int64_t rowc = rowok ? row : a.N - 1;
        const double* br = a.B + rowc * a.ldb;
        double sc = a.rowscale ? a.rowscale[rowc] : 1.0;
        coef = rowok ? a.alpha * sc : 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            f64x4 v = {0.0, 0.0, 0.0, 0.0};
            if (S::permuted(b)) {
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
            G[b] = v;
        }
    };
    auto group_of = [&](int it) -> int64_t { return ((int64_t)it * gridDim.x + blockIdx.x) * 8 + wave; };

    // prologue: chunks 0 and 1 of the stream -> slots 0 and 1
    st.sb[0] = 0; st.sb[1] = 1; st.sb[2] = 2;
    st.issue(0, 0);
    st.issue(1, 1);
    if constexpr (PREFETCH) load_group(group_of(0), Gn, coefn);

    for (int it = 0; it < a.npass; ++it) {
        const int64_t grp = group_of(it);
        // ring slots of this pass's chunks: chunk c sits in slot (rot + c) % 3
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int sl = (rot + t) % 3;
            st.sb[t] = sl;
            st.rd[t] = sl == 0 ? slot_rd[0] : sl == 1 ? slot_rd[1] : slot_rd[2];
#pragma unroll
            for (int r = 0; r < ST::NSET; ++r)
                st.rds[t][r] = sl == 0 ? slot_rds[0][r] : sl == 1 ? slot_rds[1][r] : slot_rds[2][r];
        }
        st.has_next = it + 1 < a.npass;
        if constexpr (PREFETCH) {
#pragma unroll
            for (int b = 0; b < NB; ++b) st.V[b] = Gn[b] * coefn;
        } else {
            double coef;
            load_group(grp, st.V, coef);
#pragma unroll
            for (int b = 0; b < NB; ++b) st.V[b] = st.V[b] * coef;
        }
        const int64_t row = grp * 16 + n;
        st.store = row < a.N;
        st.orow = a.Out + (st.store ? row : 0) * a.ldo;
        midt_boundary<0, NB, KL, LASTPERM, MODE>(st);
        if constexpr (PREFETCH) {   // (behind the boundary: its vmcnt(0) would wait for these loads)
            const int64_t gnext = group_of(it + 1);
            load_group(st.has_next && gnext < ngroups ? gnext : (grp < ngroups ? grp : 0), Gn, coefn);
        }
        midt_fetch<ST::BEG, NB, KL, LASTPERM, MODE>(st, st.cur);
        midt_land<ST::BEG, NB, KL, LASTPERM, MODE>(st, st.cur);
        midt_chain<ST::BEG, NB, KL, LASTPERM, MODE>(st);
        rot = (rot + NCH) % 3;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

template <int NB, int KL, bool LASTPERM, int MODE = 0>
int midt_launch(const MidTrsmArgs& a0, hipStream_t st) {
    MidTrsmArgs a = a0;
    a.pack += (size_t)MidTrsmState<NB, KL, LASTPERM, MODE>::BEG * 256;   // the stream starts at the mode's first image
    static VtLdsAttr attr;   // per kernel instantiation, per device
    if (vt_allow_big_lds((const void*)k_trsm_rows_mid<NB, KL, LASTPERM, MODE>, attr, kMidRingBytes) != VT_OK) return VT_ERR_HIP;
    const int64_t ngroups = (a.N + 15) / 16;
    int64_t g = (ngroups + 7) / 8;
    if (g > vt_num_cu()) g = vt_num_cu();
    if (g < 1) g = 1;
    a.npass = (int)((ngroups + g * 8 - 1) / (g * 8));
    hipLaunchKernelGGL((k_trsm_rows_mid<NB, KL, LASTPERM, MODE>), dim3((int)g), dim3(512), kMidRingBytes, st, a);
    return VT_OK;
}

template <int NB>
int midt_dispatch(const MidTrsmArgs& a, hipStream_t st) {
    const int rem = a.D - 16 * (NB - 1);   // 1..16 columns in the last block
    if (rem == 16) return midt_launch<NB, 4, true>(a, st);
    // NB >= 17: one variant for every partial last block (its all-zero k-steps are computed: 1 / NB of the chain's
    // last block row; the strip variants would double the build time of these long schedules)
    if constexpr (NB >= 17) return midt_launch<NB, 4, false>(a, st);
    switch ((rem + 3) / 4) {
        case 1: return midt_launch<NB, 1, false>(a, st);
        case 2: return midt_launch<NB, 2, false>(a, st);
        default: return midt_launch<NB, 4, false>(a, st);   // (9..12 columns: the all-zero fourth k-step is computed)
    }
}

}  // namespace
