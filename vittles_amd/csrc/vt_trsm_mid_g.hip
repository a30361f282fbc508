// Instantiations of the one-pass TRSM chain (vt_trsm_mid.h) for NB = 19 (D up to 304): past the Hessian kernels'
// D <= 256 the solve still holds its 16 x D tile in registers (see vt_trsm_mid.hip, kMidTrsmDMax).
#include "vt_trsm_mid.h"

int vt_mid_trsm_g(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha,
                  int64_t N, double* Out, int64_t ldo, hipStream_t st) {
    MidTrsmArgs a{(const double*)pack, B, ldb, rowscale, alpha, N, D, Out, ldo, 0};
    if ((D + 15) / 16 != 19) return VT_ERR_UNSUPPORTED;
    int rc = midt_dispatch<19>(a, st);
    if (rc != VT_OK) return rc;
    return vt_launch_status();
}
