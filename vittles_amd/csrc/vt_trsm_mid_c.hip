// Instantiations of the mid-range TRSM chain (vt_trsm_mid.h) for NB = 15..16: the chains are long compile-time
// schedules, so they are spread over several translation units to build in parallel.
#include "vt_trsm_mid.h"

int vt_mid_trsm_c(const void* pack, int D, const double* B, int64_t ldb, const double* rowscale, double alpha,
                   int64_t N, double* Out, int64_t ldo, hipStream_t st) {
    MidTrsmArgs a{(const double*)pack, B, ldb, rowscale, alpha, N, D, Out, ldo, 0};
    int rc = VT_ERR_UNSUPPORTED;
    switch ((D + 15) / 16) {
        case 15: rc = midt_dispatch<15>(a, st); break;
        case 16: rc = midt_dispatch<16>(a, st); break;
    }
    if (rc != VT_OK) return rc;
    return vt_launch_status();
}
