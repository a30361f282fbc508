// Forward-only / backward-only halves of the mid-range TRSM chain (vt_trsm_mid.h) for a FULL 256-column block
// (NB = 16): the diagonal-block solves of the blocked solve for D > 256 (vt_large.hip).
#include "vt_trsm_mid.h"

int vt_mid_trsm_d(const void* pack, const double* B, int64_t ldb, const double* rowscale, double alpha, int64_t N,
                  double* Out, int64_t ldo, int mode, hipStream_t st) {
    MidTrsmArgs a{(const double*)pack, B, ldb, rowscale, alpha, N, 256, Out, ldo, 0};
    int rc = VT_ERR_UNSUPPORTED;
    if (mode == 1) rc = midt_launch<16, 4, true, 1>(a, st);
    else if (mode == 2) rc = midt_launch<16, 4, true, 2>(a, st);
    if (rc != VT_OK) return rc;
    return vt_launch_status();
}
