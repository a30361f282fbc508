// Compile-time schedule of the MFMA multi-RHS TRSM chains (shared by vt_chol.hip: D <= 128, pack resident in LDS,
// and vt_trsm_mid.hip: 129 <= D <= 256, pack streamed through an LDS ring).  See vt_chol.hip for the pack layout.
#pragma once
#include "vt_common.h"

namespace {

__host__ __device__ __forceinline__ int vt_perm(bool permuted, int p) {
    return permuted ? (4 * (p & 3) + (p >> 2)) : p;
}

template <int NB, int KL, bool LASTPERM>
struct TrsmSched {
    static constexpr int NBT = NB * (NB + 1) / 2;
    static constexpr int M = 2 * NBT;
    static constexpr bool permuted(int b) { return b < NB - 1 || LASTPERM; }
    static constexpr int ks(int b) { return permuted(b) ? 4 : KL; }
    // product idx -> (backward?, row block j, source block i)
    static constexpr bool bwd(int idx) { return idx >= NBT; }
    static constexpr int row(int idx) {
        if (idx < NBT) {  // forward: j = 0..NB-1, i = j..0
            int j = 0, k = idx;
            while (k > j) { k -= j + 1; ++j; }
            return j;
        }
        int k = idx - NBT, j = NB - 1;  // backward: j = NB-1..0, i = j..NB-1
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
    // index of the pack image for this product
    static constexpr int image(int idx) {
        return bwd(idx) ? NBT + vt_tri(src(idx), row(idx)) : vt_tri(row(idx), src(idx));
    }
};

// Row strips of the LAST block row.  When the trailing partial block has KL <= 2 k-steps (D mod 16 in 1..8) only
// the first 4 KL parameter rows of block row NB-1 exist, and its products are computed by KL
// v_mfma_f64_4x4x4_4b per k-step (16 pipe cycles each) instead of one 64-cycle 16x16x4 product: the four 4x4
// blocks of that instruction take the SAME 4 rows of the A image (lane l reads the image at lane
// (l & 48) | (4 r + (l & 3))) against the four 4-column groups of the ordinary B operand, and result lane L holds
// row 4 r + (L >> 4), right-hand side L & 15: exactly register r of the 16x16 accumulator (see vt_hessian.hip).
// D = 100: 26 of the 200 products of a tile shrink from 64 to 16 cycles (-9.75 % of the MFMA pipe time).
template <int NB, int KL, bool LASTPERM>
struct TrsmSched2 : TrsmSched<NB, KL, LASTPERM> {
    static constexpr int STRIPS = (!LASTPERM && KL <= 2) ? KL : 0;
    static constexpr bool strip_row(int idx) { return STRIPS > 0 && TrsmSched<NB, KL, LASTPERM>::row(idx) == NB - 1; }
    static constexpr int NSET = STRIPS > 1 ? STRIPS : 1;     // A-operand sets per k-step
};

}  // namespace
