// Row-streaming kernels (HBM-bound): K0 link, K3 gradient, K6 fused HVP, K9 predict,
// objective value.  All share one shape:
//     for every row n:  dot_n = x_n . v     (optional)
//                       t_n   = op(dot_n, per-row scalars)      -> per-row outputs
//                       out  += t_n * x_n   (optional, D-vector)
//                       val  += s_n         (optional, scalar)
// with X read exactly ONCE: a TEAM of threads holds the whole row in registers
// between the dot and the rank-1 accumulation (16-byte coalesced loads, one row =
// TEAM * CPT * 16 bytes), the D-vector accumulator lives in registers for the whole
// kernel, and partial results are combined by a fixed-order two-stage reduction
// (no fp64 atomics -> bitwise reproducible).
//
// Replaces, for registered GLM objectives:
//   autograd.grad(objective)(theta, w)                 vittles/sensitivity_lib.py:354,203-209
//   the user's `mat_times_vec` HVP handed to get_cg_solver  vittles/solver_lib.py:70-98
//   `self._sens_mat @ (new_hyper - hyper0)`            vittles/sensitivity_lib.py:245-247
#include "vt_common.h"
#include <mutex>
#include <type_traits>

namespace {

enum RowOp { OP_LINK = 0, OP_GRAD = 1, OP_HVP = 2, OP_GEMVT = 3, OP_VALUE = 4, OP_GEMV = 5, OP_GEMVT2 = 6,
             OP_TAYLOR2 = 7, OP_TAYLOR3 = 8 };

struct RowArgs {
    const void* X;         // (N, ldx) row-major, double -- or float for the *_f32 entry points (fp32 storage mode:
                           // half the HBM bytes of these HBM-bound kernels; every sum is still carried in fp64)
    int64_t ldx;
    int64_t N;
    int D;
    const double* v;       // D-vector for the dot (theta / CG direction); may be NULL
    const double* y;       // per-row inputs (meaning depends on op)
    const double* w;
    const double* offset;
    const double* dvec;    // HVP: precomputed d_n;  GEMVT: the N-vector
    double scale;
    int family;
    double* z_out;         // per-row outputs
    double* r_out;
    double* d_out;
    double* part;          // [gridDim.x][D (+1)] partial column sums (+ value)
};

template <int OP>
struct OpTraits {
    static constexpr bool kDot = OP != OP_GEMVT && OP != OP_GEMVT2;
    static constexpr bool kCols = OP == OP_GRAD || OP == OP_HVP || OP == OP_GEMVT || OP == OP_GEMVT2 || OP == OP_TAYLOR2 ||
                                  OP == OP_TAYLOR3;
    static constexpr bool kVal = OP == OP_VALUE;
};

// Per-row scalars, fetched together with the row (ahead of the dot, so that their
// latency is hidden behind the row stream).
struct RowScal {
    double y, w, off, dv;
};

template <int OP>
__device__ __forceinline__ RowScal load_scal(const RowArgs& a, int64_t n) {
    RowScal s{0.0, 0.0, 0.0, 0.0};
    if constexpr (OP == OP_GEMVT || OP == OP_HVP) {
        s.dv = a.dvec[n];
    } else if constexpr (OP == OP_GEMVT2) {
        s.dv = a.dvec[n];
        s.y = a.y[n];
    } else if constexpr (OP == OP_TAYLOR2 || OP == OP_TAYLOR3) {
        s.y = a.y[n];      // z_n (from vt_glm_link)
        s.w = a.w[n];
        s.dv = a.dvec[n];  // delta w_n
        if constexpr (OP == OP_TAYLOR3) s.off = a.offset[n];   // zeta2_n = x_n . d2
    } else if constexpr (OP != OP_GEMV) {
        s.y = a.y[n];
        s.w = a.w[n];
        s.off = a.offset ? a.offset[n] : 0.0;
    }
    return s;
}

// Per-row scalar stage.  Returns the coefficient of x_n in the column accumulation.
template <int OP>
__device__ __forceinline__ double row_op(const RowArgs& a, int64_t n, const RowScal& s, double dot, bool lead,
                                         double& val) {
    if constexpr (OP == OP_GEMVT) {
        return s.dv;
    } else if constexpr (OP == OP_GEMVT2) {
        return s.dv * s.y;
    } else if constexpr (OP == OP_TAYLOR2) {
        double l2, l3;
        vt_link3(a.family, s.y, l2, l3);
        return a.scale * (s.w * l3 * dot + 2.0 * s.dv * l2) * dot;
    } else if constexpr (OP == OP_TAYLOR3) {
        // third total derivative of (w + t dw) ell'(z + t zeta1 + t^2/2 zeta2 + ...) without the ell'' zeta3 term
        double l2, l3, l4;
        vt_link4(a.family, s.y, l2, l3, l4);
        const double z1 = dot, z2 = s.off;
        return a.scale * (s.w * (l4 * z1 * z1 * z1 + 3.0 * l3 * z1 * z2) + 3.0 * s.dv * (l3 * z1 * z1 + l2 * z2));
    } else if constexpr (OP == OP_GEMV) {
        if (lead) a.z_out[n] = dot;
        return 0.0;
    } else if constexpr (OP == OP_HVP) {
        return s.dv * dot;
    } else {
        double z = dot + s.off;
        double yv = s.y, wv = s.w;
        if constexpr (OP == OP_VALUE) {
            if (lead) val += wv * vt_loss(a.family, z, yv);
            return 0.0;
        } else {
            double l1, l2;
            vt_link(a.family, z, yv, l1, l2);
            double rr = a.scale * l1;
            if (lead) {
                if (a.z_out) a.z_out[n] = z;
                if (a.r_out) a.r_out[n] = rr;
                if (a.d_out) a.d_out[n] = a.scale * wv * l2;
            }
            return wv * rr;
        }
    }
}

// TEAM threads per row, each owning CPT chunks of VEC doubles: column of element e of
// chunk k = VEC * (tt + TEAM * k) + e.   WG = max(TEAM, 256) threads.
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
template <typename XT> struct RowVec2;
template <> struct RowVec2<double> { typedef f64x2u type; };
template <> struct RowVec2<float> { typedef f32x2u type; };

template <int OP, int TEAM, int CPT, int VEC, int NT, typename XT>
__global__ __launch_bounds__(256) void k_rows(RowArgs a) {
    constexpr int WG = 256;
    constexpr int TEAMS = WG / TEAM;
    constexpr int NE = CPT * VEC;
    using T = OpTraits<OP>;
    __shared__ double red[WG / 64 + 1];
    extern __shared__ double dyn[];  // cross-team column reduction (TEAMS > 1)

    const int tid = threadIdx.x;
    const int tt = tid % TEAM;
    const int team = tid / TEAM;
    const int lane = tid & 63;
    const int64_t total_teams = (int64_t)gridDim.x * TEAMS;
    const int64_t gteam = (int64_t)blockIdx.x * TEAMS + team;

    // column offsets (clamped so that every load is in-bounds) and the dot vector
    int coff[CPT];
    double vv[NE];
    double acc[NE];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        int c0 = VEC * (tt + TEAM * k);
        coff[k] = min(c0, a.D - VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            vv[k * VEC + e] = (T::kDot && c0 + e < a.D) ? a.v[c0 + e] : 0.0;
            acc[k * VEC + e] = 0.0;
        }
    }
    double val = 0.0;

    const int64_t iters = (a.N + total_teams - 1) / total_teams;
    for (int64_t it = 0; it < iters; ++it) {
        int64_t n = gteam + it * total_teams;
        bool valid = n < a.N;
        int64_t nc = valid ? n : a.N - 1;
        const XT* xr = (const XT*)a.X + nc * a.ldx;
        double x[NE];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            if constexpr (VEC == 2) {
                using V2 = typename RowVec2<XT>::type;
                V2 t = NT ? __builtin_nontemporal_load((const V2*)(xr + coff[k])) : *(const V2*)(xr + coff[k]);
                x[2 * k] = (double)t[0];
                x[2 * k + 1] = (double)t[1];
            } else {
                x[k] = (double)(NT ? __builtin_nontemporal_load(xr + coff[k]) : xr[coff[k]]);
            }
        }
        RowScal sc = load_scal<OP>(a, nc);
        double dot = 0.0;
        if constexpr (T::kDot) {
#pragma unroll
            for (int i = 0; i < NE; ++i) dot = fma(x[i], vv[i], dot);
            if constexpr (TEAM == 16) {
                dot = vt_sum16(dot);
            } else {
                dot = vt_sum64(dot);
                if constexpr (TEAM > 64) {
                    __syncthreads();
                    if (lane == 0) red[tid >> 6] = dot;
                    __syncthreads();
                    dot = 0.0;
#pragma unroll
                    for (int i = 0; i < TEAM / 64; ++i) dot += red[i];
                }
            }
        }
        double dummy = 0.0;
        double t = row_op<OP>(a, nc, sc, dot, valid && tt == 0, valid ? val : dummy);
        if (!valid) t = 0.0;
        if constexpr (T::kCols) {
#pragma unroll
            for (int i = 0; i < NE; ++i) acc[i] = fma(t, x[i], acc[i]);
        }
    }

    // ---- combine within the workgroup, write the partial ------------------------
    const int PL = a.D + 1;
    double* out = a.part + (int64_t)blockIdx.x * PL;
    if constexpr (T::kCols) {
        if constexpr (TEAMS > 1) {
            const int DP = TEAM * CPT * VEC;  // padded row length covered by a team
#pragma unroll
            for (int k = 0; k < CPT; ++k)
#pragma unroll
                for (int e = 0; e < VEC; ++e) dyn[team * DP + VEC * (tt + TEAM * k) + e] = acc[k * VEC + e];
            __syncthreads();
            for (int c = tid; c < a.D; c += WG) {
                double s = 0.0;
                for (int g = 0; g < TEAMS; ++g) s += dyn[g * DP + c];
                out[c] = s;
            }
        } else {
#pragma unroll
            for (int k = 0; k < CPT; ++k)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    int c = VEC * (tt + TEAM * k) + e;
                    if (c < a.D) out[c] = acc[k * VEC + e];
                }
        }
    }
    if constexpr (T::kVal) {
        // only team leaders accumulated `val`
        double s = vt_sum64(val);
        __syncthreads();
        if (lane == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int i = 0; i < WG / 64; ++i) tot += red[i];
            out[a.D] = tot;
        }
    }
}

// ---- wide rows (D > 4096): one 512-thread team per CU --------------------------------
// A row no longer fits a wave's registers next to a second copy of itself, and one row
// in flight per CU is what HBM latency needs (160 KB at D = 20000).  So: the dot vector
// lives in LDS, the accumulator and the current row in registers, and the NEXT row is
// fetched in two parts -- chunks [0,P) into a prefetch buffer before the dot, chunks
// [P,CPT) straight into the current row's registers as the rank-1 update releases them --
// so that >= P chunks per lane are always outstanding.  Rows are read through a raw
// buffer descriptor (uniform base in SGPRs, 32-bit lane offsets, out-of-range lanes read
// 0: no clamps, no masks) with the streaming (slc) hint.  Measured: 7.0 TB/s at D = 20000.
typedef unsigned vt_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned vt_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const void* row, unsigned bytes) {
    uint64_t b = (uint64_t)row;
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <int VEC>
struct Chunk;
template <>
struct Chunk<2> {
    f64x2u v;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        v = __builtin_bit_cast(f64x2u, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 2));
    }
    __device__ __forceinline__ double get(int e) const { return v[e]; }
};
template <>
struct Chunk<1> {
    double v;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        v = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 2));
    }
    __device__ __forceinline__ double get(int) const { return v; }
};

// fp32 storage mode, wide rows: four floats (16 bytes) per lane and chunk, widened as they are used
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
struct ChunkF4 {
    f32x4u v;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        v = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 2));
    }
    __device__ __forceinline__ double get(int e) const { return (double)v[e]; }
};
template <int VEC, typename XT>
struct ChunkOf { using type = Chunk<VEC>; };
template <>
struct ChunkOf<4, float> { using type = ChunkF4; };

constexpr int kWideTeam = 512;

// XT = float: VEC = 4 (the fp32 storage mode's X for D > 4096, e.g. configs[3]'s D = 20000 at half the bytes); v, the
// accumulators and every sum stay double.
template <int OP, int VEC, int CPT, int P, typename XT = double>
__global__ __launch_bounds__(kWideTeam) void k_rows_wide(RowArgs a) {
    static_assert(std::is_same<XT, double>::value ? VEC <= 2 : VEC == 4, "double: 1 or 2 per lane, float: 4");
    using T = OpTraits<OP>;
    using ChunkT = typename ChunkOf<VEC, XT>::type;
    constexpr int TW = kWideTeam / 64;
    constexpr unsigned ES = (unsigned)sizeof(XT);
    extern __shared__ double dyn[];
    const int D = a.D;
    constexpr int ZP = VEC == 4 ? 4 : 2;              // zeros behind v for lanes past the row end: vl[DZ .. DZ + ZP)
    const int DZ = (D + ZP - 1) / ZP * ZP;
    double* vl = dyn;
    double* red = dyn + (T::kDot ? DZ + ZP : 0);  // [2][TW]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t Tn = gridDim.x, g = blockIdx.x;
    const unsigned voff = ES * VEC * tid, rowbytes = ES * D;
    constexpr unsigned kStep = ES * VEC * kWideTeam;

    double acc[T::kCols ? CPT * VEC : 1];
    if constexpr (T::kCols) {
#pragma unroll
        for (int i = 0; i < CPT * VEC; ++i) acc[i] = 0.0;
    }
    if constexpr (T::kDot) {
        for (int c = tid; c < DZ + ZP; c += kWideTeam) vl[c] = c < D ? a.v[c] : 0.0;
        __syncthreads();
    }
    double val = 0.0;
    const int64_t iters = (a.N + Tn - 1) / Tn;
    ChunkT x[CPT], q[P];
    RowScal sn;
    __amdgpu_buffer_rsrc_t r;
    {
        bool ok = g < a.N;
        int64_t nc = ok ? g : 0;
        r = row_rsrc((const XT*)a.X + nc * a.ldx, ok ? rowbytes : 0);
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            if (k < P) q[k].load(r, voff + kStep * k);
            else x[k].load(r, voff + kStep * k);
        }
        sn = load_scal<OP>(a, nc);
    }
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t n = g + it * Tn;  // < N by construction of iters only when valid
        const bool valid = n < a.N;
#pragma unroll
        for (int k = 0; k < P; ++k) x[k] = q[k];
        RowScal sc = sn;
        {
            int64_t n1 = n + Tn;
            bool ok = n1 < a.N;
            int64_t nc = ok ? n1 : 0;
            r = row_rsrc((const XT*)a.X + nc * a.ldx, ok ? rowbytes : 0);
#pragma unroll
            for (int k = 0; k < P; ++k) q[k].load(r, voff + kStep * k);
            sn = load_scal<OP>(a, nc);
        }
        double dot = 0.0;
        if constexpr (T::kDot) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int c = VEC * (tid + kWideTeam * k);
                if constexpr (VEC == 4) {
                    f64x2u v0 = *(const f64x2u*)(vl + min(c, DZ));
                    f64x2u v1 = *(const f64x2u*)(vl + min(c, DZ) + 2);
                    dot = fma(x[k].get(0), v0[0], fma(x[k].get(1), v0[1], dot));
                    dot = fma(x[k].get(2), v1[0], fma(x[k].get(3), v1[1], dot));
                } else if constexpr (VEC == 2) {
                    f64x2u vk = *(const f64x2u*)(vl + min(c, DZ));
                    dot = fma(x[k].get(0), vk[0], fma(x[k].get(1), vk[1], dot));
                } else {
                    dot = fma(x[k].get(0), vl[min(c, DZ)], dot);
                }
            }
            dot = vt_sum64(dot);
            double* rd = red + (it & 1) * TW;
            if (lane == 0) rd[wv] = dot;
            __syncthreads();
            dot = 0.0;
#pragma unroll
            for (int i = 0; i < TW; ++i) dot += rd[i];
        }
        double dummy = 0.0;
        double t = row_op<OP>(a, valid ? n : 0, sc, dot, valid && tid == 0, valid ? val : dummy);
        if (!valid) t = 0.0;
#pragma unroll
        for (int k = P; k < CPT; ++k) {
            if constexpr (T::kCols) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[k * VEC + e] = fma(t, x[k].get(e), acc[k * VEC + e]);
            }
            x[k].load(r, voff + kStep * k);
        }
        if constexpr (T::kCols) {
#pragma unroll
            for (int k = 0; k < P; ++k)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[k * VEC + e] = fma(t, x[k].get(e), acc[k * VEC + e]);
        }
    }
    double* out = a.part + (int64_t)blockIdx.x * (D + 1);
    if constexpr (T::kCols) {
#pragma unroll
        for (int k = 0; k < CPT; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                int c = VEC * (tid + kWideTeam * k) + e;
                if (c < D) out[c] = acc[k * VEC + e];
            }
    }
    if constexpr (T::kVal) {
        if (tid == 0) out[D] = val;  // only the team leader accumulated
    }
}

// out[c] = alpha * sum_p part[p][c] + beta[c] * bvec[c]   (c < L);  fixed order:
// 32 columns per workgroup, 32 lanes-groups each summing every 32nd slab, then the
// groups in order.
__global__ __launch_bounds__(1024) void k_rows_finalize(const double* part, int PL, int P, int L, int off,
                                                        double alpha, const double* beta, const double* bvec,
                                                        double beta_scalar, double* out) {
    __shared__ double sh[32][33];
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s = 0.0;
    if (c < L)
        for (int p = grp; p < P; p += 32) s += part[(int64_t)p * PL + off + c];
    sh[grp][cl] = s;
    __syncthreads();
    if (grp == 0 && c < L) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < 32; ++q) tot += sh[q][cl];
        tot *= alpha;
        if (bvec) tot += (beta ? beta[c] : beta_scalar) * bvec[c];
        out[c] = tot;
    }
}

inline int num_cu() { return vt_num_cu(); }   // of the current device

struct RowCfg {
    int team, cpt, vec, wg, teams;
};

// Largest D the wide kernel takes: v (+ zero pad) and the reduction scratch share 160 KB of LDS.
constexpr int kWideDMax = 20448;

// Pick the smallest team that covers D with the compiled chunk counts.
bool pick_cfg(int D, bool vec2, RowCfg& c, bool float_vec4 = false) {
    const int vec = vec2 ? 2 : 1;
    struct Opt { int team, cpt; };
    static const Opt opts[] = {{16, 1}, {16, 2}, {16, 4}, {64, 2}, {64, 4}, {64, 8}, {256, 4}, {256, 8}};
    for (const Opt& o : opts) {
        if ((int64_t)o.team * o.cpt * vec >= D) {
            c = RowCfg{o.team, o.cpt, vec, 256, 256 / o.team};
            return true;
        }
    }
    if (D > kWideDMax) return false;
    if (float_vec4) {   // float rows on a 16-byte pitch: chunks of four floats, 5 or 10 per lane
        for (int st : {5, 10}) {
            if ((int64_t)kWideTeam * st * 4 >= D) {
                c = RowCfg{kWideTeam, st, 4, kWideTeam, 1};
                return true;
            }
        }
        return false;
    }
    // wide team: bytes per lane in steps of 80 (5 x 16 B)
    static const int steps[] = {5, 10, 15, 20};
    for (int st : steps) {
        if ((int64_t)kWideTeam * st * 2 >= D) {
            c = RowCfg{kWideTeam, vec2 ? st : 2 * st, vec, kWideTeam, 1};
            return true;
        }
    }
    return false;
}

int max_grid() { return num_cu() * 4; }

template <int OP, int TEAM, int CPT, int VEC, typename XT>
void launch_one(const RowArgs& a, int grid, bool nt, hipStream_t st) {
    constexpr int WG = 256;
    constexpr int TEAMS = WG / TEAM;
    size_t shmem = TEAMS > 1 ? (size_t)TEAMS * TEAM * CPT * VEC * sizeof(double) : 0;
    if (nt) hipLaunchKernelGGL((k_rows<OP, TEAM, CPT, VEC, 1, XT>), dim3(grid), dim3(WG), shmem, st, a);
    else hipLaunchKernelGGL((k_rows<OP, TEAM, CPT, VEC, 0, XT>), dim3(grid), dim3(WG), shmem, st, a);
}

template <int OP, int VEC, int CPT, int P, typename XT = double>
int launch_wide(const RowArgs& a, int grid, hipStream_t st) {
    constexpr int ZP = VEC == 4 ? 4 : 2;
    size_t shmem = (size_t)8 * ((a.D + ZP - 1) / ZP * ZP + ZP + 2 * (kWideTeam / 64));
    static VtLdsAttr attr;  // per instantiation, per device
    if (vt_allow_big_lds((const void*)k_rows_wide<OP, VEC, CPT, P, XT>, attr, 160 * 1024) != VT_OK) return VT_ERR_HIP;
    hipLaunchKernelGGL((k_rows_wide<OP, VEC, CPT, P, XT>), dim3(grid), dim3(kWideTeam), shmem, st, a);
    return VT_OK;
}

template <int OP, int VEC, typename XT>
int launch_cfg(const RowArgs& a, const RowCfg& c, int grid, bool nt, hipStream_t st) {
#define VT_CASE(TEAM, CPT)                                   \
    if (c.team == TEAM && c.cpt == CPT) {                    \
        launch_one<OP, TEAM, CPT, VEC, XT>(a, grid, nt, st); \
        return VT_OK;                                        \
    }
    VT_CASE(16, 1) VT_CASE(16, 2) VT_CASE(16, 4) VT_CASE(64, 2) VT_CASE(64, 4) VT_CASE(64, 8)
    VT_CASE(256, 4) VT_CASE(256, 8)
#undef VT_CASE
    if constexpr (std::is_same<XT, float>::value) {
        // wide float rows (the fp32 mode at D > 4096): four floats per lane and chunk, OP by OP what the mode exposes
        if (c.team == kWideTeam && c.vec == 4) {
            if constexpr (OP == OP_LINK || OP == OP_GRAD || OP == OP_HVP || OP == OP_GEMV || OP == OP_GEMVT ||
                          OP == OP_GEMVT2 || OP == OP_VALUE) {
                if (c.cpt == 5) return launch_wide<OP, 4, 5, 5, float>(a, grid, st);
                if (c.cpt == 10) return launch_wide<OP, 4, 10, (OP == OP_GRAD ? 6 : 10), float>(a, grid, st);
            }
        }
        return VT_ERR_UNSUPPORTED;
    }
    if constexpr (std::is_same<XT, double>::value) if (c.team == kWideTeam) {   // wide rows of doubles
        // (chunks, prefetched chunks): full double buffering while the registers allow it
        if constexpr (VEC == 2) {
            if (c.cpt == 5) return launch_wide<OP, 2, 5, 5>(a, grid, st);
            if (c.cpt == 10) return launch_wide<OP, 2, 10, 10>(a, grid, st);
            if (c.cpt == 15) return launch_wide<OP, 2, 15, 15>(a, grid, st);
            if (c.cpt == 20) return launch_wide<OP, 2, 20, (OP == OP_GRAD || OP == OP_TAYLOR2 || OP == OP_TAYLOR3 ? 6 : 10)>(a, grid, st);
        } else {
            if (c.cpt == 10) return launch_wide<OP, 1, 10, 10>(a, grid, st);
            if (c.cpt == 20) return launch_wide<OP, 1, 20, 20>(a, grid, st);
            if (c.cpt == 30) return launch_wide<OP, 1, 30, 28>(a, grid, st);
            if (c.cpt == 40) return launch_wide<OP, 1, 40, (OP == OP_GRAD || OP == OP_TAYLOR2 || OP == OP_TAYLOR3 ? 8 : 20)>(a, grid, st);
        }
    }
    return VT_ERR_UNSUPPORTED;
}

// Run one row op.  Returns the number of partial slabs written (>0) or an error (<0).  max_slabs > 0: the
// workspace holds that many slabs -- a launch that would write more is refused BEFORE anything is launched.
template <int OP, typename XT = double>
int run_rows(RowArgs a, hipStream_t st, int max_slabs = 0) {
    bool vec2 = (a.D % 2 == 0) && (a.ldx % 2 == 0) && (((uintptr_t)a.X) % (2 * sizeof(XT)) == 0);
    // float rows past what the 256-thread teams cover: the wide kernel on 16-byte lanes (rows on a 16-byte pitch)
    const bool f4 = std::is_same<XT, float>::value && a.D > (vec2 ? 4096 : 2048) && a.ldx % 4 == 0 &&
                    ((uintptr_t)a.X) % 16 == 0;
    RowCfg c;
    if (!pick_cfg(a.D, vec2, c, f4)) return VT_ERR_UNSUPPORTED;
    int64_t want = (a.N + c.teams - 1) / c.teams;
    // 512-thread teams: one per CU; 256-thread workgroups: up to 4 per CU.
    int cap = c.wg == kWideTeam ? num_cu() : max_grid();
    int grid = (int)(want < cap ? (want < 1 ? 1 : want) : cap);
    if (max_slabs > 0 && grid > max_slabs) return VT_ERR_SCRATCH_LIMIT;
    // stream X past the caches once it is larger than the Infinity Cache can hold
    bool nt = (double)a.N * a.D * (double)sizeof(XT) > 192.0e6;
    int rc = (vec2 || c.vec == 4) ? launch_cfg<OP, 2, XT>(a, c, grid, nt, st) : launch_cfg<OP, 1, XT>(a, c, grid, nt, st);
    if (rc != VT_OK) return rc;
    if (hipGetLastError() != hipSuccess) return VT_ERR_HIP;
    return grid;
}

int finalize(const double* part, int D, int P, int L, int off, double alpha, const double* beta,
             const double* bvec, double beta_scalar, double* out, hipStream_t st) {
    hipLaunchKernelGGL(k_rows_finalize, dim3((L + 31) / 32), dim3(1024), 0, st, part, D + 1, P, L, off, alpha,
                       beta, bvec, beta_scalar, out);
    return vt_launch_status();
}

}  // namespace

extern "C" size_t vt_rows_workspace_bytes(int D) {
    return (size_t)max_grid() * (D + 1) * sizeof(double);
}

#define VT_ROWS_PROLOGUE                                                                  \
    if ((!X && N > 0) || N < 0 || D < 1 || ldx < D) return VT_ERR_BAD_ARG;                \
    if (!ws || ws_bytes < vt_rows_workspace_bytes(D)) return VT_ERR_WORKSPACE;            \
    hipStream_t st = (hipStream_t)stream;

extern "C" int vt_glm_link(int family, const double* X, int64_t ldx, const double* theta, const double* y,
                           const double* w, const double* offset, double scale, int64_t N, int D,
                           double* z_out, double* r_out, double* d_out, void* ws, size_t ws_bytes,
                           void* stream) {
    VT_ROWS_PROLOGUE
    if (N == 0) return VT_OK;
    if (!theta || !y || !w) return VT_ERR_BAD_ARG;
    RowArgs a{X, ldx, N, D, theta, y, w, offset, nullptr, scale, family, z_out, r_out, d_out, (double*)ws};
    int P = run_rows<OP_LINK>(a, st);
    return P < 0 ? P : VT_OK;
}

extern "C" int vt_glm_grad(int family, const double* X, int64_t ldx, const double* theta, const double* y,
                           const double* w, const double* offset, double scale, const double* prior,
                           int64_t N, int D, double* g_out, double* r_out, double* d_out, void* ws,
                           size_t ws_bytes, void* stream) {
    VT_ROWS_PROLOGUE
    if (!theta || !g_out || (N > 0 && (!y || !w))) return VT_ERR_BAD_ARG;
    int P = 0;
    if (N > 0) {
        RowArgs a{X, ldx, N, D, theta, y, w, offset, nullptr, scale, family, nullptr, r_out, d_out, (double*)ws};
        P = run_rows<OP_GRAD>(a, st);
        if (P < 0) return P;
    }
    return finalize((const double*)ws, D, P, D, 0, 1.0, prior, prior ? theta : nullptr, 0.0, g_out, st);
}

extern "C" int vt_glm_hvp(const double* X, int64_t ldx, const double* d, const double* prior,
                          double prior_scalar, const double* v, int64_t N, int D, double* out, void* ws,
                          size_t ws_bytes, void* stream) {
    VT_ROWS_PROLOGUE
    if (!v || !out || (N > 0 && !d)) return VT_ERR_BAD_ARG;
    int P = 0;
    if (N > 0) {
        RowArgs a{X, ldx, N, D, v, nullptr, nullptr, nullptr, d, 1.0, 0, nullptr, nullptr, nullptr, (double*)ws};
        P = run_rows<OP_HVP>(a, st);
        if (P < 0) return P;
    }
    bool has_prior = prior != nullptr || prior_scalar != 0.0;
    return finalize((const double*)ws, D, P, D, 0, 1.0, prior, has_prior ? v : nullptr, prior_scalar, out, st);
}

extern "C" int vt_gemv_t(const double* X, int64_t ldx, const double* u, int64_t N, int D, double* out,
                         void* ws, size_t ws_bytes, void* stream) {
    VT_ROWS_PROLOGUE
    if (!out || (N > 0 && !u)) return VT_ERR_BAD_ARG;
    int P = 0;
    if (N > 0) {
        RowArgs a{X, ldx, N, D, nullptr, nullptr, nullptr, nullptr, u, 1.0, 0, nullptr, nullptr, nullptr, (double*)ws};
        P = run_rows<OP_GEMVT>(a, st);
        if (P < 0) return P;
    }
    return finalize((const double*)ws, D, P, D, 0, 1.0, nullptr, nullptr, 0.0, out, st);
}

extern "C" int vt_gemv_t2(const double* X, int64_t ldx, const double* u1, const double* u2, int64_t N, int D,
                          double* out, void* ws, size_t ws_bytes, void* stream) {
    VT_ROWS_PROLOGUE
    if (!out || (N > 0 && (!u1 || !u2))) return VT_ERR_BAD_ARG;
    int P = 0;
    if (N > 0) {
        RowArgs a{X, ldx, N, D, nullptr, u2, nullptr, nullptr, u1, 1.0, 0, nullptr, nullptr, nullptr, (double*)ws};
        P = run_rows<OP_GEMVT2>(a, st);
        if (P < 0) return P;
    }
    return finalize((const double*)ws, D, P, D, 0, 1.0, nullptr, nullptr, 0.0, out, st);
}

extern "C" int vt_glm_taylor2_rhs(int family, const double* X, int64_t ldx, const double* z, const double* w,
                                  const double* dw, double scale, const double* u, int64_t N, int D, double* out,
                                  void* ws, size_t ws_bytes, void* stream) {
    VT_ROWS_PROLOGUE
    if (!u || !out || (N > 0 && (!z || !w || !dw))) return VT_ERR_BAD_ARG;
    int P = 0;
    if (N > 0) {
        RowArgs a{X, ldx, N, D, u, z, w, nullptr, dw, scale, family, nullptr, nullptr, nullptr, (double*)ws};
        P = run_rows<OP_TAYLOR2>(a, st);
        if (P < 0) return P;
    }
    return finalize((const double*)ws, D, P, D, 0, 1.0, nullptr, nullptr, 0.0, out, st);
}

extern "C" int vt_glm_taylor3_rhs(int family, const double* X, int64_t ldx, const double* z, const double* w,
                                  const double* dw, const double* zeta2, double scale, const double* u1,
                                  int64_t N, int D, double* out, void* ws, size_t ws_bytes, void* stream) {
    VT_ROWS_PROLOGUE
    if (!u1 || !out || (N > 0 && (!z || !w || !dw || !zeta2))) return VT_ERR_BAD_ARG;
    int P = 0;
    if (N > 0) {
        RowArgs a{X, ldx, N, D, u1, z, w, zeta2, dw, scale, family, nullptr, nullptr, nullptr, (double*)ws};
        P = run_rows<OP_TAYLOR3>(a, st);
        if (P < 0) return P;
    }
    return finalize((const double*)ws, D, P, D, 0, 1.0, nullptr, nullptr, 0.0, out, st);
}

extern "C" int vt_gemv(const double* X, int64_t ldx, const double* v, int64_t N, int D, double* out,
                       void* ws, size_t ws_bytes, void* stream) {
    VT_ROWS_PROLOGUE
    if (N == 0) return VT_OK;
    if (!v || !out) return VT_ERR_BAD_ARG;
    RowArgs a{X, ldx, N, D, v, nullptr, nullptr, nullptr, nullptr, 1.0, 0, out, nullptr, nullptr, (double*)ws};
    int P = run_rows<OP_GEMV>(a, st);
    return P < 0 ? P : VT_OK;
}

extern "C" int vt_glm_value(int family, const double* X, int64_t ldx, const double* theta, const double* y,
                            const double* w, const double* offset, double scale, int64_t N, int D,
                            double* value_out, void* ws, size_t ws_bytes, void* stream) {
    VT_ROWS_PROLOGUE
    if (!theta || !value_out || (N > 0 && (!y || !w))) return VT_ERR_BAD_ARG;
    int P = 0;
    if (N > 0) {
        RowArgs a{X, ldx, N, D, theta, y, w, offset, nullptr, scale, family, nullptr, nullptr, nullptr, (double*)ws};
        P = run_rows<OP_VALUE>(a, st);
        if (P < 0) return P;
    }
    // value = scale * sum_n w_n ell_n   (the prior term is added by the caller)
    return finalize((const double*)ws, D, P, 1, D, scale, nullptr, nullptr, 0.0, value_out, st);
}

// ---- the one entry point that allocates nothing (no counterpart in the reference) -----------------
// vt_glm_hvp for callers that cannot hand over (or allocate) a workspace: code under HIP graph capture,
// tight serving loops, harnesses that forbid hipMalloc.  The per-workgroup partial slabs that vt_glm_hvp
// keeps in the caller's `ws` live in a file-scope device array instead.  Its size covers every D the row
// kernels take: D <= 4096 uses at most 4 workgroups per CU of 256 CUs (1024 slabs of D + 1 doubles),
// 4096 < D <= 20448 one 512-thread team per CU (256 slabs): 256 * 20449 doubles = 41.9 MB per GPU, held
// until the process ends.  Every call on a GPU shares that GPU's array (calls that overlap on one GPU race);
// nothing in it has to start at zero: every slab the finalize kernel reads is fully written by the row
// kernel of the same call.
namespace {
constexpr size_t kScratchDoubles = (size_t)256 * (kWideDMax + 1);
static __device__ __align__(256) double scratch[kScratchDoubles];

constexpr int kMaxDevices = 64;
double* g_scratch_ptr[kMaxDevices];          // per device: the address of ITS copy of `scratch`
std::mutex g_scratch_mutex;

// `scratch` is a host-side stand-in in host code, not a device address: resolve it once per device, with
// that device current, and use the resolved pointer everywhere.
int scratch_for_current_device(double** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return VT_ERR_HIP;
    if (dev < 0 || dev >= kMaxDevices) return VT_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        double* ptr = nullptr;
        if (hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr) return VT_ERR_HIP;
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return VT_OK;
}
}  // namespace

static_assert(kScratchDoubles * sizeof(double) == VT_NOALLOC_SCRATCH_BYTES, "header states the scratch size");

extern "C" int vt_glm_hvp_noalloc(const double* X, int64_t ldx, const double* d, const double* prior,
                                  double prior_scalar, const double* v, int64_t N, int D, double* out,
                                  void* stream) {
    if ((!X && N > 0) || N < 0 || D < 1 || ldx < D) return VT_ERR_BAD_ARG;
    if (!v || !out || (N > 0 && !d)) return VT_ERR_BAD_ARG;
    if (D > kWideDMax) return VT_ERR_SCRATCH_LIMIT;
    // slabs one call can write: the grid cap of the kernel family that takes this D
    size_t slabs = D > 4096 ? (size_t)num_cu() : (size_t)max_grid();
    if (slabs * (size_t)(D + 1) > kScratchDoubles) return VT_ERR_SCRATCH_LIMIT;
    double* ptr = nullptr;
    int rc = scratch_for_current_device(&ptr);
    if (rc != VT_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    int P = 0;
    if (N > 0) {
        RowArgs a{X, ldx, N, D, v, nullptr, nullptr, nullptr, d, 1.0, 0, nullptr, nullptr, nullptr, ptr};
        P = run_rows<OP_HVP>(a, st, (int)slabs);   // (the slab count is checked before the launch)
        if (P < 0) return P;
    }
    bool has_prior = prior != nullptr || prior_scalar != 0.0;
    return finalize(ptr, D, P, D, 0, 1.0, prior, has_prior ? v : nullptr, prior_scalar, out, st);
}

// ---- fp32 storage mode (BASELINE.json north_star: "optional fp32 mode", 1e-4): X (or any streamed matrix) is
// float (D <= 4096, or <= 20448 with rows on a 16-byte pitch); vectors, per-row outputs and every accumulation stay fp64.  Same kernels, XT = float.
extern "C" int vt_glm_link_f32(int family, const float* X, int64_t ldx, const double* theta, const double* y,
                               const double* w, const double* offset, double scale, int64_t N, int D,
                               double* z_out, double* r_out, double* d_out, void* ws, size_t ws_bytes,
                               void* stream) {
    VT_ROWS_PROLOGUE
    if (N == 0) return VT_OK;
    if (!theta || !y || !w) return VT_ERR_BAD_ARG;
    RowArgs a{X, ldx, N, D, theta, y, w, offset, nullptr, scale, family, z_out, r_out, d_out, (double*)ws};
    int P = run_rows<OP_LINK, float>(a, st);
    return P < 0 ? P : VT_OK;
}

extern "C" int vt_glm_grad_f32(int family, const float* X, int64_t ldx, const double* theta, const double* y,
                               const double* w, const double* offset, double scale, const double* prior,
                               int64_t N, int D, double* g_out, double* r_out, double* d_out, void* ws,
                               size_t ws_bytes, void* stream) {
    VT_ROWS_PROLOGUE
    if (!theta || !g_out || (N > 0 && (!y || !w))) return VT_ERR_BAD_ARG;
    int P = 0;
    if (N > 0) {
        RowArgs a{X, ldx, N, D, theta, y, w, offset, nullptr, scale, family, nullptr, r_out, d_out, (double*)ws};
        P = run_rows<OP_GRAD, float>(a, st);
        if (P < 0) return P;
    }
    return finalize((const double*)ws, D, P, D, 0, 1.0, prior, prior ? theta : nullptr, 0.0, g_out, st);
}

extern "C" int vt_glm_hvp_f32(const float* X, int64_t ldx, const double* d, const double* prior,
                              double prior_scalar, const double* v, int64_t N, int D, double* out, void* ws,
                              size_t ws_bytes, void* stream) {
    VT_ROWS_PROLOGUE
    if (!v || !out || (N > 0 && !d)) return VT_ERR_BAD_ARG;
    int P = 0;
    if (N > 0) {
        RowArgs a{X, ldx, N, D, v, nullptr, nullptr, nullptr, d, 1.0, 0, nullptr, nullptr, nullptr, (double*)ws};
        P = run_rows<OP_HVP, float>(a, st);
        if (P < 0) return P;
    }
    bool has_prior = prior != nullptr || prior_scalar != 0.0;
    return finalize((const double*)ws, D, P, D, 0, 1.0, prior, has_prior ? v : nullptr, prior_scalar, out, st);
}

extern "C" int vt_gemv_t_f32(const float* X, int64_t ldx, const double* u, int64_t N, int D, double* out,
                             void* ws, size_t ws_bytes, void* stream) {
    VT_ROWS_PROLOGUE
    if (!out || (N > 0 && !u)) return VT_ERR_BAD_ARG;
    int P = 0;
    if (N > 0) {
        RowArgs a{X, ldx, N, D, nullptr, nullptr, nullptr, nullptr, u, 1.0, 0, nullptr, nullptr, nullptr, (double*)ws};
        P = run_rows<OP_GEMVT, float>(a, st);
        if (P < 0) return P;
    }
    return finalize((const double*)ws, D, P, D, 0, 1.0, nullptr, nullptr, 0.0, out, st);
}

extern "C" int vt_gemv_t2_f32(const float* X, int64_t ldx, const double* u1, const double* u2, int64_t N, int D,
                              double* out, void* ws, size_t ws_bytes, void* stream) {
    VT_ROWS_PROLOGUE
    if (!out || (N > 0 && (!u1 || !u2))) return VT_ERR_BAD_ARG;
    int P = 0;
    if (N > 0) {
        RowArgs a{X, ldx, N, D, nullptr, u2, nullptr, nullptr, u1, 1.0, 0, nullptr, nullptr, nullptr, (double*)ws};
        P = run_rows<OP_GEMVT2, float>(a, st);
        if (P < 0) return P;
    }
    return finalize((const double*)ws, D, P, D, 0, 1.0, nullptr, nullptr, 0.0, out, st);
}

extern "C" int vt_gemv_f32(const float* X, int64_t ldx, const double* v, int64_t N, int D, double* out,
                           void* ws, size_t ws_bytes, void* stream) {
    VT_ROWS_PROLOGUE
    if (N == 0) return VT_OK;
    if (!v || !out) return VT_ERR_BAD_ARG;
    RowArgs a{X, ldx, N, D, v, nullptr, nullptr, nullptr, nullptr, 1.0, 0, out, nullptr, nullptr, (double*)ws};
    int P = run_rows<OP_GEMV, float>(a, st);
    return P < 0 ? P : VT_OK;
}

extern "C" int vt_glm_value_f32(int family, const float* X, int64_t ldx, const double* theta, const double* y,
                                const double* w, const double* offset, double scale, int64_t N, int D,
                                double* value_out, void* ws, size_t ws_bytes, void* stream) {
    VT_ROWS_PROLOGUE
    if (!theta || !value_out || (N > 0 && (!y || !w))) return VT_ERR_BAD_ARG;
    int P = 0;
    if (N > 0) {
        RowArgs a{X, ldx, N, D, theta, y, w, offset, nullptr, scale, family, nullptr, nullptr, nullptr, (double*)ws};
        P = run_rows<OP_VALUE, float>(a, st);
        if (P < 0) return P;
    }
    return finalize((const double*)ws, D, P, 1, D, scale, nullptr, nullptr, 0.0, value_out, st);
}

// ---- data-weight Taylor expansion, ANY order (orders 2 and 3 have the fused one-pass kernels above) ---------------
// Order k of the expansion of theta(w + eps dw):  H d_k = -X^T c,
//   c_n = scale * [ w_n  sum_{m=2..min(k, mio)}   ell^(m+1)(z_n) B_{k,m}(zeta_1..)
//                 + k dw_n sum_{m=1..min(k-1, mio)} ell^(m+1)(z_n) B_{k-1,m}(zeta_1..) ],   zeta_j = x_n . d_j,
// B_{n,m} the partial Bell polynomials (Faa di Bruno for ell'(z(eps)); the m = 1 term of the first sum is H d_k
// itself).  A term with ell^(m+1) is an m-th input partial derivative of the estimating equation, which is what
// max_input_order (mio) cuts.  Per row: the link derivatives (logistic: polynomials in sigma from the compile-time
// table P_{m+1} = P_m' (s - s^2); Poisson: e^z; Gaussian: ell'' = 1) and the Bell triangle by its recurrence
// B_{n,m} = sum_i C(n-1, i-1) zeta_i B_{n-i,m-1}.  An N-vector pass: the two passes over X are vt_gemv / vt_gemv_t.
namespace {
constexpr int kTaylorMax = VT_TAYLOR_MAX_ORDER;
struct TaylorTabs {
    double lg[kTaylorMax + 2][kTaylorMax + 3];   // lg[m][i]: coefficient of sigma^i in d^m/dz^m log(1 + e^z), m >= 1
    double binom[kTaylorMax + 1][kTaylorMax + 1];
};
constexpr TaylorTabs make_taylor_tabs() {
    TaylorTabs t{};
    t.lg[1][1] = 1.0;
    for (int m = 1; m <= kTaylorMax; ++m)
        for (int i = 1; i <= m; ++i) {
            t.lg[m + 1][i] += i * t.lg[m][i];
            t.lg[m + 1][i + 1] -= i * t.lg[m][i];
        }
    for (int n = 0; n <= kTaylorMax; ++n) {
        t.binom[n][0] = 1.0;
        for (int r = 1; r <= n; ++r) t.binom[n][r] = t.binom[n - 1][r - 1] + (r <= n - 1 ? t.binom[n - 1][r] : 0.0);
    }
    return t;
}
__constant__ TaylorTabs c_taylor = make_taylor_tabs();

__global__ __launch_bounds__(256) void k_glm_taylor_coeff(int family, const double* z, const double* w, const double* dw,
                                                          const double* zeta, int64_t ldz, double scale, int k, int mio,
                                                          int64_t N, double* c) {
    for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        double ld[kTaylorMax + 2];               // ld[m] = ell^(m)(z_n), m = 2..k+1
        const double zn = z[n];
        if (family == VT_FAMILY_LOGISTIC) {
            const double e = exp(-fabs(zn));
            const double sg = zn >= 0.0 ? 1.0 / (1.0 + e) : e / (1.0 + e);
            for (int m = 2; m <= k + 1; ++m) {
                double p = 0.0;
                for (int i = m; i >= 1; --i) p = fma(p, sg, c_taylor.lg[m][i]);
                ld[m] = p * sg;
            }
        } else if (family == VT_FAMILY_POISSON) {
            const double ez = exp(zn);
            for (int m = 2; m <= k + 1; ++m) ld[m] = ez;
        } else {
            for (int m = 2; m <= k + 1; ++m) ld[m] = m == 2 ? 1.0 : 0.0;
        }
        double x[kTaylorMax + 1];
        for (int j = 1; j <= k; ++j) x[j] = j < k ? zeta[(int64_t)(j - 1) * ldz + n] : 0.0;
        double B[kTaylorMax + 1][kTaylorMax + 1];
        for (int a = 0; a <= k; ++a)
            for (int m = 0; m <= k; ++m) B[a][m] = 0.0;
        B[0][0] = 1.0;
        for (int a = 1; a <= k; ++a)
            for (int m = 1; m <= a; ++m) {
                double sacc = 0.0;
                for (int i = 1; i <= a - m + 1; ++i) sacc = fma(c_taylor.binom[a - 1][i - 1] * x[i], B[a - i][m - 1], sacc);
                B[a][m] = sacc;
            }
        double s1 = 0.0, s2 = 0.0;
        for (int m = 2; m <= k && m <= mio; ++m) s1 = fma(ld[m + 1], B[k][m], s1);
        for (int m = 1; m <= k - 1 && m <= mio; ++m) s2 = fma(ld[m + 1], B[k - 1][m], s2);
        c[n] = scale * (w[n] * s1 + (double)k * dw[n] * s2);
    }
}
}  // namespace

extern "C" int vt_glm_taylor_coeff(int family, const double* z, const double* w, const double* dw, const double* zeta,
                                   int64_t ldz, double scale, int order, int max_input_order, int64_t N, double* c,
                                   void* stream) {
    if (N < 0 || family < 0 || family > 2) return VT_ERR_BAD_ARG;
    if (order < 2 || order > kTaylorMax) return VT_ERR_UNSUPPORTED;
    if (N == 0) return VT_OK;
    if (!z || !w || !dw || !zeta || !c || ldz < N || max_input_order < 1) return VT_ERR_BAD_ARG;
    const int grid = (int)std::min<int64_t>((N + 255) / 256, (int64_t)num_cu() * 8);
    hipLaunchKernelGGL(k_glm_taylor_coeff, dim3(grid), dim3(256), 0, (hipStream_t)stream, family, z, w, dw, zeta, ldz,
                       scale, order, max_input_order, N, c);
    return vt_launch_status();
}
