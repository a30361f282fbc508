// Forward-only / backward-only halves of the one-pass TRSM chain (vt_trsm_mid.h) for the FULL diagonal blocks of the
// blocked solve for D > 384 (vt_large.hip): VT_TRSM_ONEPASS_D_MAX = 384 columns (NB = 24) and VT_MID_D_MAX = 256 (NB = 16).
#include "vt_trsm_mid.h"

int vt_mid_trsm_d(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha, int64_t N,
                  double* Out, int64_t ldo, int mode, hipStream_t st) {
    static_assert(VT_TRSM_ONEPASS_D_MAX == 384 && VT_MID_D_MAX == 256, "the halves are instantiated for NB = 24 and 16");
    MidTrsmArgs a{(const double*)pack, B, ldb, rowscale, alpha, N, D, Out, ldo, 0};
    int rc = VT_ERR_UNSUPPORTED;
    if (D == VT_TRSM_ONEPASS_D_MAX) {
        if (mode == 1) rc = midt_launch<24, 4, true, 1>(a, st);
        else if (mode == 2) rc = midt_launch<24, 4, true, 2>(a, st);
    } else if (D == VT_MID_D_MAX) {
        if (mode == 1) rc = midt_launch<16, 4, true, 1>(a, st);
        else if (mode == 2) rc = midt_launch<16, 4, true, 2>(a, st);
    }
    if (rc != VT_OK) return rc;
    return vt_launch_status();
}
