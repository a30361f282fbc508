// Instantiations of the one-pass TRSM chain (vt_trsm_mid.h) for NB = 19, 20 (D up to 320): past the Hessian kernels'
// D <= 256 the solve still holds its 16 x D tile in registers (see vt_common.h, VT_TRSM_ONEPASS_D_MAX).
#include "vt_trsm_mid.h"

int vt_mid_trsm_f(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha,
                  int64_t N, double* Out, int64_t ldo, hipStream_t st) {
    MidTrsmArgs a{(const double*)pack, B, ldb, rowscale, alpha, N, D, Out, ldo, 0};
    int rc = VT_ERR_UNSUPPORTED;
    switch ((D + 15) / 16) {
        case 19: rc = midt_dispatch<19>(a, st); break;
        case 20: rc = midt_dispatch<20>(a, st); break;
    }
    if (rc != VT_OK) return rc;
    return vt_launch_status();
}
