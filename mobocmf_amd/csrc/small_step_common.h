// Device helpers shared by the one-launch steps of small surrogates (tiny_step.hip: one workgroup per surrogate, M <= 32;
// coop_step.hip: several workgroups per surrogate, M <= 128): wavefront / workgroup reductions, the layer kernels' covariance
// function with its gradient terms (gram.hip), and the theta / omega factor gradients of the conditioned iteration.
// Internal; every including file gets its own copies (anonymous namespace).
#pragma once
#include "common.h"

namespace {

constexpr int NWMAX = 8;                         // wavefronts per workgroup, at most (256 or 512 threads)
constexpr int TLM = MOBOCMF_TINY_MAX_LAYERS;
constexpr int DBT = MOBOCMF_TINY_MAX_D;          // x columns of a staged inducing row (zero-padded)
constexpr int ZW = DBT + 1;                      // + the f column
constexpr int HS = 5 + 2 * DBT;                  // packed hyper-parameters of a layer, at most
constexpr int NVEC = 11;                         // per-column vectors of a layer kept in `work`
constexpr int64_t CPL_DOUBLES = 256 * 17;           // scratch of coupling_seeds (mode 4)
constexpr int NSEG = 32;                         // parameter tensors of a model, at most (3 layers x 9 + 3 noise)
constexpr double MINV = 1e-10;                   // gpytorch.settings.min_variance (float64)
constexpr double LOG2PI = 1.8378770664093453;

// Sum over the wavefront by DPP moves (quad permutes, row shifts, row broadcasts: ~10 cycles a step; __shfl_xor is a
// ds_bpermute round trip through LDS per step and half).  The total is valid in LANE 63 only.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    union { double d; int i[2]; } u, r;
    u.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, ROW_MASK, 0xf, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, ROW_MASK, 0xf, true);
    return v + r.d;
}
__device__ __forceinline__ double wsum63(double v) {
    v = dpp_add<0xb1, 0xf>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4e, 0xf>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x114, 0xf>(v);     // row_shr:4
    v = dpp_add<0x118, 0xf>(v);     // row_shr:8   -> lane 15 of every row holds the row's sum
    v = dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1, 3
    v = dpp_add<0x143, 0xc>(v);     // row_bcast:31 into rows 2, 3
    return v;
}
// sum over the workgroup's NW wavefronts, handed to every thread (two barriers; sh: NW doubles nobody else touches meanwhile)
template <int NW>
__device__ __forceinline__ double bsum(double v, double* sh) {
    v = wsum63(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += sh[w];
    return t;
}
template <int NW>
__device__ __forceinline__ double red_sum(const double* red, int stride, int t) {      // the wavefronts' partials of slot t
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[w * stride + t];
    return v;
}
__device__ __forceinline__ double rdlane(double v, int l) {      // l: wave-uniform
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], l);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
    return u.d;
}

struct KV { double k, E1, E2, Ef, fd; };
// k(a, b) of DESIGN.md / gram.hip: a = (xa[0..d), fa), b = the staged inducing row zb (x part zero-padded to DBT, then f at
// zb[DBT]).  hy: packed constrained hyper-parameters, il: inverse lengthscales (il[k], il[DBT + k]; zero for k >= d, so the
// distance loops run over all DBT columns without a test)
__device__ __forceinline__ void kern_eval(int kind, int d, const double* xa, double fa, const double* zb, const double* hy,
                                          const double* il, KV& o) {
    double d1 = 0.0, d2 = 0.0;
#pragma unroll
    for (int k = 0; k < DBT; ++k) {
        const double df = (k < d ? xa[k] : 0.0) - zb[k];
        const double t1 = df * il[k], t2 = df * il[DBT + k];
        d1 += t1 * t1;
        d2 += t2 * t2;
    }
    o.E1 = exp(-0.5 * d1);
    if (!kind) { o.E2 = o.Ef = o.fd = 0.0; o.k = hy[0] * o.E1; return; }
    o.E2 = exp(-0.5 * d2);
    const double zf = zb[DBT];
    o.fd = (fa - zf) / hy[4];
    o.Ef = exp(-0.5 * o.fd * o.fd);
    o.k = hy[0] * o.E1 * (hy[2] * fa * zf + hy[1] * o.Ef) + hy[3] * o.E2;
}
// Gradient terms of one kernel value with upstream G: hyper-parameters accumulated into h[] in the FIXED layout
// [a1 | af | nu | a2 | lsf | ls1[DBT] | ls2[DBT]] (kind 0: alpha in slot 0, its lengthscales in the ls1 slots) -- compile-time
// indices, so h[] stays in registers; slot_of() maps a packed position to its slot.  dfa = dk/dfa, dzf = dk/dzf (times G).
__device__ __forceinline__ int slot_of(int kind, int d, int t) {
    if (!kind) return t == 0 ? 0 : 5 + (t - 1);
    return t < 5 ? t : (t < 5 + d ? t : 5 + DBT + (t - 5 - d));
}
__device__ __forceinline__ bool slot_used(int kind, int d, int t) {
    if (t < 5) return kind ? true : t == 0;
    if (t < 5 + DBT) return t - 5 < d;
    return kind && t - 5 - DBT < d;
}
__device__ __forceinline__ void kern_back(int kind, int d, const double* xa, double fa, const double* zb, const double* hy,
                                          const double* il, double G, double (&h)[HS], double& dfa, double& dzf) {
    KV o;
    kern_eval(kind, d, xa, fa, zb, hy, il, o);
    if (!kind) {
        h[0] += G * o.E1;
        const double W1 = G * hy[0] * o.E1;
#pragma unroll
        for (int k = 0; k < DBT; ++k) {
            const double t = ((k < d ? xa[k] : 0.0) - zb[k]) * il[k];
            h[5 + k] += W1 * t * t * il[k];
        }
        dfa = dzf = 0.0;
        return;
    }
    const double a1 = hy[0], af = hy[1], nu = hy[2], a2 = hy[3], ilf = 1.0 / hy[4], zf = zb[DBT];
    const double aE1 = a1 * o.E1, inner = nu * fa * zf + af * o.Ef, cb = aE1 * af * ilf, T = G * o.Ef * o.fd;
    h[0] += G * o.E1 * inner;
    h[1] += G * aE1 * o.Ef;
    h[2] += G * aE1 * fa * zf;
    h[3] += G * o.E2;
    h[4] += cb * T * o.fd;
    const double W1 = G * aE1 * inner, W2 = G * a2 * o.E2;
#pragma unroll
    for (int k = 0; k < DBT; ++k) {
        const double df = (k < d ? xa[k] : 0.0) - zb[k];
        const double t1 = df * il[k], t2 = df * il[DBT + k];
        h[5 + k] += W1 * t1 * t1 * il[k];
        h[5 + DBT + k] += W2 * t2 * t2 * il[DBT + k];
    }
    dfa = G * aE1 * nu * zf - T * cb;
    dzf = G * aE1 * nu * fa + T * cb;
}

// the same for the INPUTS of the data side only: dxa[k] += G dk/dx_k, dfa += G dk/dfa (parameters are constants)
__device__ __forceinline__ void kern_back_in(int kind, int d, const double* xa, double fa, const double* zb, const double* hy,
                                             const double* il, double G, double (&dxa)[DBT], double& dfa) {
    KV o;
    kern_eval(kind, d, xa, fa, zb, hy, il, o);
    double W1, W2 = 0.0;
    if (!kind) {
        W1 = G * hy[0] * o.E1;
    } else {
        const double aE1 = hy[0] * o.E1, zf = zb[DBT];
        W1 = G * aE1 * (hy[2] * fa * zf + hy[1] * o.Ef);
        W2 = G * hy[3] * o.E2;
        dfa += G * aE1 * hy[2] * zf - G * o.Ef * o.fd * aE1 * hy[1] / hy[4];
    }
#pragma unroll
    for (int k = 0; k < DBT; ++k) {
        const double df = (k < d ? xa[k] : 0.0) - zb[k];
        dxa[k] -= W1 * df * il[k] * il[k] + W2 * df * il[DBT + k] * il[DBT + k];
    }
}

__device__ __forceinline__ double ncdf_t(double z) { return 0.5 * (1.0 + erf(z * 0.7071067811865476)); }
__device__ __forceinline__ double npdf_t(double z) { return 0.3989422804014327 * exp(-0.5 * z * z); }
// a value another workgroup of this launch wrote before the grid barrier (agent-scope load: not served from a stale line)
__device__ __forceinline__ double peer(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Mode 4, after the grid barrier: the factor gradients of THIS model (blackbox_mfdgp_fitter.py:227-243; the algebra of
// elementwise.hip cond_factors_kernel) into its seed arrays -- zero on the batch columns, the theta factors on a constraint's
// Pareto columns, the omega factors (all models' moments at x~) on the x~ columns.  part: >= 256 * (1 + 2 n_obj) doubles of
// workgroup scratch.  NT threads; all of them call it.
template <int NT>
__device__ __forceinline__ void coupling_seeds(const mobocmf_tiny_model* models, const mobocmf_tiny_model& md, int ncol_top, double* part,
                               double* sh) {
    const mobocmf_tiny_coupling& cp = *md.coupling;
    const int tid = threadIdx.x, P = cp.P, T = cp.T, no = cp.n_obj, nc = cp.n_con;
    const int stride = 1 + 2 * no;
    double* sgm = const_cast<double*>(md.seed_gmean);      // (inputs of modes 1 / 3; in mode 4 the launch fills them itself)
    double* sgv = const_cast<double*>(md.seed_gvar);
    for (int c = tid; c < ncol_top; c += NT) { sgm[c] = 0.0; sgv[c] = 0.0; }
    // ---- omega: sum_p over the Pareto points, split among NG thread groups per point t (T <= NT)
    int NG = 256 / T;      // (T <= 256: the binding's limit) NG T <= 256 rows of `part`
    if (NG > P) NG = P;
    const int tl = tid % T, grp = tid / T;
    const bool on = grp < NG;
    if (on) {
        // (loops over the <= 8 objectives with compile-time bounds and a predicate: the per-objective arrays then stay in
        // registers; indexed by a run-time bound they were 448 bytes of scratch memory per lane)
        double isd[8], fmv[8], fvv[8], gm[8], gv[8], osum = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            fmv[j] = fvv[j] = isd[j] = gm[j] = gv[j] = 0.0;
            if (j < no) {
                const mobocmf_tiny_model& mj = models[cp.obj_model[j]];
                fmv[j] = peer(mj.top_mean + P + tl);
                fvv[j] = peer(mj.top_var + P + tl);
                isd[j] = 1.0 / sqrt(fvv[j]);
            }
        }
        for (int p = grp; p < P; p += NG) {
            double ph[8], u[8], O = 1.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                u[j] = 0.0;
                ph[j] = 1.0;
                if (j < no) {
                    u[j] = (cp.front[(int64_t)p * no + j] - fmv[j]) * isd[j];
                    ph[j] = ncdf_t(u[j]);
                    O *= ph[j];
                }
            }
            osum += O;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < no) {
                    double rest = 1.0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) rest *= (q == j || q >= no) ? 1.0 : ph[q];
                    const double pd = npdf_t(u[j]);
                    gm[j] += rest * pd * (-isd[j]);
                    gv[j] += rest * pd * (-0.5 * u[j] / fvv[j]);
                }
            }
        }
        double* pp = part + (int64_t)(grp * T + tl) * stride;
        pp[0] = osum;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < no) { pp[1 + j] = gm[j]; pp[1 + no + j] = gv[j]; }
    }
    __syncthreads();
    const double dldc = cp.log_eps - cp.log_1m_eps;      // omega: coef_c = log eps, coef_1mc = log(1 - eps)
    double acc = 0.0;
    if (tid < T) {
        const int t = tid;
        double osum = 0.0, gm = 0.0, gv = 0.0;
        const int j0 = md.role == 0 ? md.role_index : 0;
        for (int g2 = 0; g2 < NG; ++g2) {
            const double* pp = part + (int64_t)(g2 * T + t) * stride;
            osum += pp[0];
            if (no) { gm += pp[1 + j0]; gv += pp[1 + no + j0]; }      // (LDS / global reads with a run-time index: no private array)
        }
        double phic[8], dzm[8], dzv[8], C = 1.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            phic[k] = 1.0;
            dzm[k] = dzv[k] = 0.0;
            if (k < nc) {
                const mobocmf_tiny_model& mk = models[cp.con_model[k]];
                const double cm = peer(mk.top_mean + P + t), cv = peer(mk.top_var + P + t);
                const double sd = sqrt(cv), z = (cm - cp.thresholds[k]) / sd, pd = npdf_t(z);
                phic[k] = ncdf_t(z);
                dzm[k] = pd / sd;
                dzv[k] = -0.5 * pd * z / cv;
                C *= phic[k];
            }
        }
        acc = dldc * C * osum + cp.log_1m_eps * (double)P;
        if (md.role == 0) {
            sgm[P + t] = dldc * C * gm;
            sgv[P + t] = dldc * C * gv;
        } else {
            double rest = osum, zm = 0.0, zv = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                rest *= (q == md.role_index || q >= nc) ? 1.0 : phic[q];
                if (q == md.role_index) { zm = dzm[q]; zv = dzv[q]; }
            }
            sgm[P + t] = dldc * rest * zm;
            sgv[P + t] = dldc * rest * zv;
        }
    }
    if (blockIdx.x == 0) {      // (uniform) the omega term itself, once per launch
        acc = bsum<NT / 64>(acc, sh);
        if (tid == 0) cp.losses[nc] = acc;
    }
    // ---- theta (a constraint's own moments at the Pareto points): coef_c = log(1 - eps), coef_1mc = log eps
    if (md.role == 1) {
        double tacc = 0.0;
        const double thr = cp.thresholds[md.role_index], dl = cp.log_1m_eps - cp.log_eps;
        for (int pcol = tid; pcol < P; pcol += NT) {
            const double mu = md.top_mean[pcol], var = md.top_var[pcol], sd = sqrt(var), z = (mu - thr) / sd, pd = npdf_t(z);
            tacc += dl * ncdf_t(z) + cp.log_eps;
            sgm[pcol] = dl * pd / sd;
            sgv[pcol] = dl * (-0.5 * pd * z / var);
        }
        tacc = bsum<NT / 64>(tacc, sh);
        if (tid == 0) cp.losses[md.role_index] = tacc;
    }
    __syncthreads();
}

__device__ __forceinline__ int seg_len(int l, int s, int d) { return l == 0 ? (s == 0 ? 1 : d) : (s < 5 ? 1 : d); }

}  // namespace
