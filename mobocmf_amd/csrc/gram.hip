// Gram matrices of the MFDGP kernels and their backward, gfx950.
//
//   kind 0:  k = alpha * exp(-1/2 sum_k ((x_k - z_k)/ls_k)^2)                       (mfdgp_hidden_layer.py:43-47)
//   kind 1:  k = a1 E1(x,z) (nu f f' + af Ef(f,f')) + a2 E2(x,z)                     (mfdgp_hidden_layer.py:68-88,115)
//
// Output K is [Mp x Np] row-major: row m = inducing input m, column n = data row n (coalesced along n).
// One thread owns one BASE data row n0 (x row) and walks its xdiv sample replicas n = n0*xdiv + s, so the
// x-only factors E1, E2 (2 of the 3 exponentials) are evaluated once per (m, n0) and shared by the S samples.
// The inducing rows of the tile (32 x (d+1) doubles + inverse lengthscales) sit in LDS and are read as
// wave-wide broadcasts; the thread's x row lives in registers.
#include "common.h"

#define GT 128   // threads per block = base rows per block
#define GM 32    // inducing rows per block


// XD > 0: compile-time xdiv (the S sample replicas of a row): f values live in registers, the replica loop is
// unrolled and K is written with 16-byte stores.  XD == 0: run-time xdiv.
typedef double v2f64_t __attribute__((ext_vector_type(2)));

template <int KIND, int DB, int XD>
__global__ __launch_bounds__(GT) void gram_fwd_kernel(GramArgs g) {
    // replica path: a thread owns the XD adjacent columns of its base row -- written straight from registers that is 16 bytes
    // per lane at a stride of 8 XD bytes, four (eight) store instructions each touching 64 lines partially: the kernel was
    // bound by those stores (26.9 us at 512 x 16384 with 17 us of FP64 VALU work under them).  The wavefront's 64 XD values
    // of a row go through LDS instead (own region per wavefront, no barrier: LDS is in order within a wavefront) and leave
    // as 1 KB contiguous per store instruction.
    __shared__ __attribute__((aligned(16))) double stg[XD ? (GT / 64) * 64 * (XD + 2) : 1];
    __shared__ double zs[GM][DB];
    __shared__ double zfs[GM];
    __shared__ double il1[DB], il2[DB];
    const int d = g.d;
    const int tid = threadIdx.x;
    const int gm = g.gm;            // inducing rows of this block (<= GM; fewer for small grids: more, shorter workgroups)
    const int m0 = blockIdx.y * gm;
    const double* hyp = g.hyp;
    double a1, af = 0, nu = 0, a2 = 0, ilf = 0;
    if (KIND == 0) {
        a1 = hyp[0];
        if (tid < DB) il1[tid] = tid < d ? 1.0 / hyp[1 + tid] : 0.0;
    } else {
        a1 = hyp[0]; af = hyp[1]; nu = hyp[2]; a2 = hyp[3]; ilf = 1.0 / hyp[4];
        if (tid < DB) { il1[tid] = tid < d ? 1.0 / hyp[5 + tid] : 0.0; il2[tid] = tid < d ? 1.0 / hyp[5 + d + tid] : 0.0; }
    }
    // columns k >= d of the DB-wide tiles are zero (inducing rows, inverse lengthscales, the thread's x row): the distance
    // loops below run over all DB columns WITHOUT a test -- an `if (k < d)` per column made every column its own basic
    // block with its own LDS read + s_waitcnt lgkmcnt(0): 16 serialised LDS round trips per inducing row
    for (int e = tid; e < gm * DB; e += GT) {
        int mm = e / DB, k = e % DB;
        int m = m0 + mm;
        zs[mm][k] = (m < g.M && k < d) ? g.Zx[(int64_t)(m / g.zdiv) * d + k] : 0.0;
    }
    if (KIND == 1 && tid < gm) zfs[tid] = (m0 + tid < g.M) ? g.zf[m0 + tid] : 0.0;
    __syncthreads();

    const int64_t n0 = (int64_t)blockIdx.x * GT + tid;
    const int xdiv = XD ? XD : g.xdiv;
    const int64_t c0 = n0 * xdiv;
    if (c0 >= g.Np) return;
    const bool real = n0 < g.nbase;
    double xr[DB];
#pragma unroll
    for (int k = 0; k < DB; ++k) xr[k] = (real && k < d) ? g.x[n0 * d + k] : 0.0;
    double fr[XD ? XD : 1];
    if (XD && KIND == 1) {
#pragma unroll
        for (int s = 0; s < XD; s += 2) {
            v2f64_t v = real ? *(const v2f64_t*)(g.f + c0 + s) : (v2f64_t){0.0, 0.0};
            fr[s] = v[0];
            fr[s + 1] = v[1];
        }
    }

    // all 64 lanes of the wavefront are here and own real rows: the row stores may go through the staging region
    const bool full_wave = XD && __ballot(real) == ~0ull;
    (void)full_wave;
    if (g.knn && blockIdx.y == 0) {
        for (int s = 0; s < xdiv; ++s) {
            int64_t n = c0 + s;
            if (n >= g.Np) break;
            double v = 0.0;
            if (real) {
                if (KIND == 0) v = a1;
                else { double fn = g.f[n]; v = a1 * (nu * fn * fn + af) + a2; }
            }
            g.knn[n] = v;
        }
    }
    for (int mm = 0; mm < gm; ++mm) {
        const int m = m0 + mm;
        double* krow = g.K + (int64_t)m * g.ldk;
        if (!real || m >= g.M) {
            for (int s = 0; s < xdiv; ++s) {
                int64_t n = c0 + s;
                if (n >= g.Np) break;
                krow[n] = (g.is_kmm && n == m) ? 1.0 : 0.0;
            }
            continue;
        }
        double d1 = 0.0, d2 = 0.0;
#pragma unroll
        for (int k = 0; k < DB; ++k) {
            double df = xr[k] - zs[mm][k];
            double t1 = df * il1[k];
            d1 += t1 * t1;
            if (KIND == 1) { double t2 = df * il2[k]; d2 += t2 * t2; }
        }
        const double E1 = exp(-0.5 * d1);
        if (KIND == 0) {
            double v = a1 * E1;
            if (g.is_kmm && c0 == m) v += g.jitter;
            krow[c0] = v;   // kind 0 layers always have xdiv == 1 for Kmm; general loop below for replicas
            for (int s = 1; s < xdiv; ++s) krow[c0 + s] = a1 * E1;
        } else if (XD) {
            const double E2 = exp(-0.5 * d2);
            const double zfm = zfs[mm];
            const double c1 = a1 * E1, c2 = a2 * E2;
            const double cp = c1 * nu * zfm, cq = c1 * af;      // k = cq Ef + (cp f + c2): two FMAs per replica
            const int lane = tid & 63;
            double* my = stg + (tid >> 6) * 64 * (XD + 2);
#pragma unroll
            for (int s = 0; s < XD; s += 2) {
                v2f64_t v;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const double fn = fr[s + e];
                    const double fd = (fn - zfm) * ilf;
                    v[e] = cq * exp(-0.5 * fd * fd) + (cp * fn + c2);
                }
                if (full_wave) *(v2f64_t*)(my + lane * (XD + 2) + s) = v;
                else *(v2f64_t*)(krow + c0 + s) = v;      // a wavefront with padding rows: straight from the registers
            }
            if (full_wave) {
                // element e of the wavefront's 64 XD contiguous columns sits at (e / XD) (XD + 2) + e % XD
                double* wrow = krow + (c0 - (int64_t)lane * XD);
#pragma unroll
                for (int i = 0; i < XD / 2; ++i) {
                    const int e = (i * 64 + lane) * 2;
                    *(v2f64_t*)(wrow + e) = *(const v2f64_t*)(my + (e / XD) * (XD + 2) + e % XD);
                }
            }
        } else {
            const double E2 = exp(-0.5 * d2);
            const double zfm = zfs[mm];
            const double c1 = a1 * E1, c2 = a2 * E2;
            for (int s = 0; s < xdiv; ++s) {
                const int64_t n = c0 + s;
                const double fn = g.f[n];
                const double fd = (fn - zfm) * ilf;
                const double Ef = exp(-0.5 * fd * fd);
                double v = c1 * (nu * fn * zfm + af * Ef) + c2;
                if (g.is_kmm && n == m) v += g.jitter;
                krow[n] = v;
            }
        }
    }
}

__device__ __forceinline__ double wave_sum(double v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// Backward: consumes G = dL/dK (and gknn) and produces per-block partial sums (deterministic two-stage
// reduction, no float atomics across workgroups).
template <int KIND, int DB, bool WANT_DX, int XD>
__global__ __launch_bounds__(GT) void gram_bwd_kernel(GramArgs g) {
    __shared__ double zs[GM][DB];
    __shared__ double zfs[GM];
    __shared__ double il1[DB], il2[DB];
    __shared__ double dzf_s[GM];
    __shared__ double red[2][8 + 2 * DB];
    extern __shared__ double dfs[];   // [xdiv][GT] per-thread private df accumulators (kind 1)
    const int d = g.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gm = g.gm;            // inducing rows of this block (<= GM; fewer for small grids: more, shorter workgroups)
    const int m0 = blockIdx.y * gm;
    const double* hyp = g.hyp;
    double a1, af = 0, nu = 0, a2 = 0, ilf = 0;
    if (KIND == 0) {
        a1 = hyp[0];
        if (tid < DB) il1[tid] = tid < d ? 1.0 / hyp[1 + tid] : 0.0;
    } else {
        a1 = hyp[0]; af = hyp[1]; nu = hyp[2]; a2 = hyp[3]; ilf = 1.0 / hyp[4];
        if (tid < DB) { il1[tid] = tid < d ? 1.0 / hyp[5 + tid] : 0.0; il2[tid] = tid < d ? 1.0 / hyp[5 + d + tid] : 0.0; }
    }
    // columns k >= d of the DB-wide tiles are zero (inducing rows, inverse lengthscales, the thread's x row): the distance
    // loops below run over all DB columns WITHOUT a test -- an `if (k < d)` per column made every column its own basic
    // block with its own LDS read + s_waitcnt lgkmcnt(0): 16 serialised LDS round trips per inducing row
    for (int e = tid; e < gm * DB; e += GT) {
        int mm = e / DB, k = e % DB;
        int m = m0 + mm;
        zs[mm][k] = (m < g.M && k < d) ? g.Zx[(int64_t)(m / g.zdiv) * d + k] : 0.0;
    }
    if (tid < gm) {
        dzf_s[tid] = 0.0;
        if (KIND == 1) zfs[tid] = (m0 + tid < g.M) ? g.zf[m0 + tid] : 0.0;
    }
    __syncthreads();

    const int64_t n0 = (int64_t)blockIdx.x * GT + tid;
    const int xdiv = XD ? XD : g.xdiv;
    const int64_t c0 = n0 * xdiv;
    const bool real = n0 < g.nbase;
    // Columns in inactive 128-column blocks of G (GramArgs.colact) were never written and count as exact zeros.  amask: bit s
    // set = replica column c0 + s is live.  XD > 0 divides 128: a base row's replicas share one block (all or nothing).
    static_assert(XD == 0 || 128 % XD == 0, "replica runs must not straddle a column block");
    unsigned long long amask = ~0ull;
    if (g.colact && real) {
        if (XD) amask = g.colact[c0 >> 7] != 0 ? ~0ull : 0ull;
        else {
            amask = 0ull;
            for (int s = 0; s < xdiv; ++s) amask |= (unsigned long long)(g.colact[(c0 + s) >> 7] != 0 ? 1 : 0) << s;
        }
    }
    const bool live = real && amask != 0ull;      // no live column: the thread only writes its zero partials
    double fr[XD ? XD : 1], dfa[XD ? XD : 1];   // XD > 0: f and the df accumulators live in registers
    if (XD) {
#pragma unroll
        for (int s = 0; s < XD; s += 2) {
            v2f64_t v = real ? *(const v2f64_t*)(g.f + c0 + s) : (v2f64_t){0.0, 0.0};
            fr[s] = v[0];
            fr[s + 1] = v[1];
            dfa[s] = dfa[s + 1] = 0.0;
        }
    }
    double xr[DB], aL1[DB], aL2[KIND == 1 ? DB : 1], aX[WANT_DX ? DB : 1];
#pragma unroll
    for (int k = 0; k < DB; ++k) {
        xr[k] = (real && k < d) ? g.x[n0 * d + k] : 0.0;
        aL1[k] = 0.0;
        if (KIND == 1) aL2[k] = 0.0;
        if (WANT_DX) aX[k] = 0.0;
    }
    double s_a1 = 0, s_af = 0, s_nu = 0, s_a2 = 0, s_lsf = 0;
    if (KIND == 1 && !XD)
        for (int s = 0; s < xdiv; ++s) dfs[s * GT + tid] = 0.0;

    // diagonal k_nn terms (once per column)
    if (g.gknn && blockIdx.y == 0 && real) {
        for (int s = 0; s < xdiv; ++s) {
            const int64_t n = c0 + s;
            const double gk = g.gknn[n];
            if (KIND == 0) s_a1 += gk;
            else {
                const double fn = g.f[n];
                s_a1 += gk * (nu * fn * fn + af);
                s_nu += gk * a1 * fn * fn;
                s_af += gk * a1;
                s_a2 += gk;
                // d knn / d f accumulated into df_part row 0 below via dfdiag
            }
        }
    }

    for (int mm = 0; mm < gm; ++mm) {
        const int m = m0 + mm;
        if (m >= g.M) break;   // uniform across the block
        double dzf_loc = 0.0;
        if (live) {
            const double* grow = g.G + (int64_t)m * g.ldk;
            double d1 = 0.0, d2 = 0.0;
#pragma unroll
            for (int k = 0; k < DB; ++k) {
                double df = xr[k] - zs[mm][k];
                double t1 = df * il1[k];
                d1 += t1 * t1;
                if (KIND == 1) { double t2 = df * il2[k]; d2 += t2 * t2; }
            }
            const double E1 = exp(-0.5 * d1);
            double W1 = 0.0, W2 = 0.0;
            if (KIND == 0) {
                double Gsum = 0.0;
                for (int s = 0; s < xdiv; ++s) Gsum += ((amask >> s) & 1ull) ? grow[c0 + s] : 0.0;
                s_a1 += Gsum * E1;
                W1 = Gsum * a1 * E1;
            } else {
                const double E2 = exp(-0.5 * d2);
                const double zfm = zfs[mm];
                // Per replica only the five sums S0..S4 over (G, G f, G Ef, G Ef fd, G Ef fd^2) and the column's own f
                // gradient are formed; every hyper-parameter / zf term is a product of those with factors that do not
                // depend on the replica (14 + exp instead of 34 + exp flops per element).
                const double aE1 = a1 * E1;
                const double c_a = aE1 * nu * zfm, c_b = aE1 * af * ilf;
                double G2s = 0.0, S1 = 0.0, S2 = 0.0, S3 = 0.0, S4 = 0.0;
                auto body = [&](double Gv, double fn, double& dfacc) {
                    const double fd = (fn - zfm) * ilf;
                    const double GEf = Gv * exp(-0.5 * fd * fd);
                    const double T = GEf * fd;
                    G2s += Gv;
                    S1 += Gv * fn;
                    S2 += GEf;
                    S3 += T;
                    S4 += T * fd;
                    dfacc += Gv * c_a - T * c_b;                      // per-column f gradient over this block's rows
                };
                if (XD) {
#pragma unroll
                    for (int s = 0; s < XD; s += 2) {
                        const v2f64_t Gp = *(const v2f64_t*)(grow + c0 + s);
                        body(Gp[0], fr[s], dfa[s]);
                        body(Gp[1], fr[s + 1], dfa[s + 1]);
                    }
                } else {
                    for (int s = 0; s < xdiv; ++s) {
                        double acc = 0.0;
                        body(((amask >> s) & 1ull) ? grow[c0 + s] : 0.0, g.f[c0 + s], acc);
                        dfs[s * GT + tid] += acc;
                    }
                }
                const double inner = nu * zfm * S1 + af * S2;          // sum_s G (nu f zf + af Ef)
                s_a1 += E1 * inner;
                s_nu += aE1 * zfm * S1;
                s_af += aE1 * S2;
                s_lsf += c_b * S4;                                      // G a1 E1 af Ef fd^2 / lsf
                W1 = aE1 * inner;
                dzf_loc = aE1 * nu * S1 + c_b * S3;                     // G a1 E1 (nu f + af Ef (f - zf)/lsf^2)
                s_a2 += G2s * E2;
                W2 = G2s * a2 * E2;
            }
#pragma unroll
            for (int k = 0; k < DB; ++k) {
                double df = xr[k] - zs[mm][k];
                double t1 = df * il1[k];
                aL1[k] += W1 * t1 * t1;
                double gx = W1 * t1 * il1[k];
                if (KIND == 1) {
                    double t2 = df * il2[k];
                    aL2[k] += W2 * t2 * t2;
                    gx += W2 * t2 * il2[k];
                }
                if (WANT_DX) aX[k] -= gx;
            }
        }
        if (KIND == 1) {
            double tot = wave_sum(dzf_loc);
            if (lane == 0) atomicAdd(&dzf_s[mm], tot);   // LDS ds_add_f64, 2 adders per slot
        }
    }
    // ---- diagonal df term + zero-fill df_part for padded / non-real columns
    if (KIND == 1) {
        for (int s = 0; s < xdiv; ++s) {
            const int64_t n = c0 + s;
            if (n >= g.Np) break;
            double v = real ? (XD ? dfa[XD ? s : 0] : dfs[s * GT + tid]) : 0.0;
            if (real && g.gknn && blockIdx.y == 0) v += g.gknn[n] * a1 * 2.0 * nu * g.f[n];
            g.df_part[(int64_t)blockIdx.y * g.Np + n] = v;
        }
    }
    if (WANT_DX && real) {
        double* dxp = g.dx_part + ((int64_t)blockIdx.y * g.nbase + n0) * d;
#pragma unroll
        for (int k = 0; k < DB; ++k)
            if (k < d) dxp[k] = aX[k];
    }
    // ---- block reduction of the hyper-parameter sums
    const int H = (KIND == 0) ? 1 + d : 5 + 2 * d;
    {
        double v;
        v = wave_sum(s_a1); if (lane == 0) red[wave][0] = v;
        if (KIND == 1) {
            v = wave_sum(s_af); if (lane == 0) red[wave][1] = v;
            v = wave_sum(s_nu); if (lane == 0) red[wave][2] = v;
            v = wave_sum(s_a2); if (lane == 0) red[wave][3] = v;
            v = wave_sum(s_lsf); if (lane == 0) red[wave][4] = v;
        }
#pragma unroll
        for (int k = 0; k < DB; ++k) {
            if (k < d) {
                v = wave_sum(aL1[k]);
                if (lane == 0) red[wave][(KIND == 0 ? 1 : 5) + k] = v * il1[k];
                if (KIND == 1) {
                    v = wave_sum(aL2[k]);
                    if (lane == 0) red[wave][5 + d + k] = v * il2[k];
                }
            }
        }
    }
    __syncthreads();
    double* hp = g.hyp_part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * H;
    if (tid < H) hp[tid] = red[0][tid] + red[1][tid];
    if (KIND == 1 && tid < gm) g.dzf_part[(int64_t)blockIdx.x * g.Mp + m0 + tid] = dzf_s[tid];
}

// out[j] (+)= scale * sum_p part[p*stride + j]
__global__ void sum_partials_kernel(const double* part, int64_t P, int64_t stride, double* out, int64_t len,
                                    double scale, int accumulate) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= len) return;
    double v = 0.0;
    for (int64_t p = 0; p < P; ++p) v += part[p * stride + j];
    v *= scale;
    out[j] = accumulate ? out[j] + v : v;
}

// many partials, few outputs: one block per output j, threads stride over the partials
__global__ void sum_partials_wide_kernel(const double* part, int64_t P, int64_t stride, double* out, int64_t len,
                                         double scale, int accumulate) {
    __shared__ double sh[4];
    const int64_t j = blockIdx.x;
    double v = 0.0;
    for (int64_t p = threadIdx.x; p < P; p += 256) v += part[p * stride + j];
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = (sh[0] + sh[1] + sh[2] + sh[3]) * scale;
        out[j] = accumulate ? out[j] + v : v;
    }
}

// out[b] = sum_{s<div} in[b*div + s]
__global__ void group_sum_kernel(const double* in, double* out, int64_t nout, int div, int accumulate) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nout) return;
    double v = 0.0;
    for (int s = 0; s < div; ++s) v += in[j * div + s];
    out[j] = accumulate ? out[j] + v : v;
}

#define SUM_MULTI_MAX 8
// Several reductions in ONE launch: consecutive block ranges belong to consecutive tasks; a task with many partials per
// output and few outputs is reduced "wide" (one block per output, threads stride over the partials), the others one
// thread per output.  (Each separate launch is 3-5 us of pure latency on the step's critical path.)
struct SumTasksArg { SumTask t[SUM_MULTI_MAX]; int blk0[SUM_MULTI_MAX + 1]; int wide[SUM_MULTI_MAX]; int n; };
__global__ void sum_partials_multi_kernel(SumTasksArg T) {
    __shared__ double sh[4];
    int k = 0;
    while (k + 1 < T.n && (int)blockIdx.x >= T.blk0[k + 1]) ++k;
    const SumTask& t = T.t[k];
    if (T.wide[k] == 1) {
        const int64_t j = (int64_t)(blockIdx.x - T.blk0[k]);
        double v = 0.0;
        for (int64_t p = threadIdx.x; p < t.P; p += 256) v += t.part[p * t.stride + j];
        if (t.part2)
            for (int64_t p = threadIdx.x; p < t.P2; p += 256) v += t.part2[p * t.stride2 + j];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            v = sh[0] + sh[1] + sh[2] + sh[3];
            if (t.extra) v += t.extra[j];
            t.out[j] = t.accumulate ? t.out[j] + v : v;
        }
        return;
    }
    if (T.wide[k] == 2) {
        // many outputs AND a few dozen partials each: 64 outputs per block, four wavefronts share an output's partial rows
        // (p = q, q + 4, ...) and combine through LDS in a fixed order -- a quarter of the dependent loads per thread
        __shared__ double sq[4][64];
        const int jj = threadIdx.x & 63, q = threadIdx.x >> 6;
        const int64_t j = (int64_t)(blockIdx.x - T.blk0[k]) * 64 + jj;
        double v = 0.0;
        if (j < t.len) {
            for (int64_t p = q; p < t.P; p += 4) v += t.part[p * t.stride + j];
            if (t.part2)
                for (int64_t p = q; p < t.P2; p += 4) v += t.part2[p * t.stride2 + j];
        }
        sq[q][jj] = v;
        __syncthreads();
        if (q == 0 && j < t.len) {
            v = (sq[0][jj] + sq[1][jj]) + (sq[2][jj] + sq[3][jj]);
            if (t.extra) v += t.extra[j];
            t.out[j] = t.accumulate ? t.out[j] + v : v;
        }
        return;
    }
    const int64_t j = (int64_t)(blockIdx.x - T.blk0[k]) * blockDim.x + threadIdx.x;
    if (j >= t.len) return;
    double v = 0.0;
    for (int64_t p = 0; p < t.P; ++p) v += t.part[p * t.stride + j];
    if (t.part2)
        for (int64_t p = 0; p < t.P2; ++p) v += t.part2[p * t.stride2 + j];
    if (t.extra) v += t.extra[j];
    t.out[j] = t.accumulate ? t.out[j] + v : v;
}

int launch_sum_partials(const double* part, int64_t P, int64_t stride, double* out, int64_t len, double scale,
                        int accumulate, hipStream_t s);

int launch_sum_partials_multi(const SumTask* tasks, int n, hipStream_t s) {
    if (n > SUM_MULTI_MAX) {
        for (int i = 0; i < n; i += SUM_MULTI_MAX) {
            int rc = launch_sum_partials_multi(tasks + i, n - i < SUM_MULTI_MAX ? n - i : SUM_MULTI_MAX, s);
            if (rc) return rc;
        }
        return MOBOCMF_OK;
    }
    SumTasksArg T = {};
    int64_t nb = 0;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (tasks[i].len <= 0) continue;
        T.t[m] = tasks[i];
        T.blk0[m] = (int)nb;
        const int64_t pmax = tasks[i].P > (tasks[i].part2 ? tasks[i].P2 : 0) ? tasks[i].P : tasks[i].P2;
        T.wide[m] = (pmax >= 64 && tasks[i].len <= 4096) ? 1 : (pmax >= 16 ? 2 : 0);
        nb += T.wide[m] == 1 ? tasks[i].len : T.wide[m] == 2 ? (tasks[i].len + 63) / 64 : (tasks[i].len + 255) / 256;
        ++m;
    }
    T.blk0[m] = (int)nb;
    T.n = m;
    if (m == 0) return MOBOCMF_OK;
    if (nb > 0x7fffffff) return MOBOCMF_BAD_ARG;
    hipLaunchKernelGGL(sum_partials_multi_kernel, dim3((unsigned)nb), dim3(256), 0, s, T);
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}

int launch_sum_partials(const double* part, int64_t P, int64_t stride, double* out, int64_t len, double scale,
                        int accumulate, hipStream_t s) {
    if (len <= 0) return MOBOCMF_OK;
    if (P >= 64 && len <= 4096) {
        hipLaunchKernelGGL(sum_partials_wide_kernel, dim3((unsigned)len), dim3(256), 0, s, part, P, stride, out, len,
                           scale, accumulate);
        return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
    }
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, part, P, stride, out,
                       len, scale, accumulate);
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}

static inline int d_bucket(int d) { return d <= 2 ? 2 : (d <= 8 ? 8 : 32); }

// Inducing rows per workgroup
static int gram_rows_per_block(const GramArgs& g, bool fwd = false) {
    int64_t nb_cols = (g.Np + g.xdiv - 1) / g.xdiv;
    int64_t gx = (nb_cols + GT - 1) / GT;
    // GM, halved down to GM / 4 while the grid stays under 1024 two-wavefront workgroups (one wavefront per SIMD): the
    // threads walk their inducing rows serially, a thin grid is latency-bound (2048 base rows x 8 replicas, M = 512:
    // 256 workgroups x 32 rows took 61 us in backward for 67 MB of G)
    int gm = GM;
    while (gm > GM / 4 && gx * (g.Mp / gm) < 1024) gm /= 2;
    // the forward writes no partials, so shorter workgroups cost nothing.  For K_mm (a few hundred columns: 256 workgroups at
    // M = 512) two rows per workgroup take the launch from 7.4 to 5.2 us; on the panels (exp-bound: ~13 us of FP64 VALU work
    // beside 13 us of row stores at 512 x 16384) a finer grid changed nothing and is not used
    if (fwd && g.is_kmm)
        while (gm > 2 && gx * (g.Mp / gm) < 4096) gm /= 2;
    return gm;
}
void gram_grid(const GramArgs& g, dim3* grid) {
    int64_t nb_cols = (g.Np + g.xdiv - 1) / g.xdiv;   // base rows incl. the ones that only own padded columns
    *grid = dim3((unsigned)((nb_cols + GT - 1) / GT), (unsigned)(g.Mp / gram_rows_per_block(g)), 1);
}

int launch_gram_fwd(const GramArgs& g0, hipStream_t s) {
    GramArgs g = g0;
    g.gm = gram_rows_per_block(g0, true);
    if (g.d < 1 || g.d > 32) return MOBOCMF_BAD_ARG;
    dim3 grid;
    gram_grid(g, &grid);
    grid.y = (unsigned)(g.Mp / g.gm);
    const int db = d_bucket(g.d);
#define GF(K, D, X) hipLaunchKernelGGL((gram_fwd_kernel<K, D, X>), grid, dim3(GT), 0, s, g)
    // fast replica paths (f in registers, 16-byte stores) for the usual S = 8 / 16 of layers >= 1
    const bool vec_ok = g.kind == 1 && !g.is_kmm && ((uintptr_t)g.f % 16 == 0) && ((uintptr_t)g.K % 16 == 0) && (g.ldk % 2 == 0);
    if (g.kind == 0) { if (db == 2) GF(0, 2, 0); else if (db == 8) GF(0, 8, 0); else GF(0, 32, 0); }
    else if (vec_ok && g.xdiv == 8) { if (db == 2) GF(1, 2, 8); else if (db == 8) GF(1, 8, 8); else GF(1, 32, 8); }
    else if (vec_ok && g.xdiv == 16) { if (db == 2) GF(1, 2, 16); else if (db == 8) GF(1, 8, 16); else GF(1, 32, 16); }
    else { if (db == 2) GF(1, 2, 0); else if (db == 8) GF(1, 8, 0); else GF(1, 32, 0); }
#undef GF
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}

int launch_gram_bwd(const GramArgs& g0, bool want_dx, hipStream_t s) {
    GramArgs g = g0;
    g.gm = gram_rows_per_block(g0);
    if (g.d < 1 || g.d > 32) return MOBOCMF_BAD_ARG;
    dim3 grid;
    gram_grid(g, &grid);
    const int db = d_bucket(g.d);
    if (g.xdiv > 48) return MOBOCMF_BAD_ARG;
    const size_t shm = g.kind == 1 ? (size_t)g.xdiv * GT * sizeof(double) : 0;
#define GB(K, D, X)                                                                                    \
    do {                                                                                               \
        if (want_dx) hipLaunchKernelGGL((gram_bwd_kernel<K, D, true, X>), grid, dim3(GT), (X) ? 0 : shm, s, g);   \
        else hipLaunchKernelGGL((gram_bwd_kernel<K, D, false, X>), grid, dim3(GT), (X) ? 0 : shm, s, g);          \
    } while (0)
    const bool vec_ok = g.kind == 1 && ((uintptr_t)g.f % 16 == 0) && ((uintptr_t)g.G % 16 == 0) && (g.ldk % 2 == 0);
    if (g.kind == 0) { if (db == 2) GB(0, 2, 0); else if (db == 8) GB(0, 8, 0); else GB(0, 32, 0); }
    else if (vec_ok && g.xdiv == 8) { if (db == 2) GB(1, 2, 8); else if (db == 8) GB(1, 8, 8); else GB(1, 32, 8); }
    else if (vec_ok && g.xdiv == 16) { if (db == 2) GB(1, 2, 16); else if (db == 8) GB(1, 8, 16); else GB(1, 32, 16); }
    else { if (db == 2) GB(1, 2, 0); else if (db == 8) GB(1, 8, 0); else GB(1, 32, 0); }
#undef GB
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}
