// Random-Fourier-feature function samples evaluated on a grid: the Pareto extraction of a sampled problem
// evaluates every sampled objective / constraint on 1000 d^2 + N points, recursing through the layers
// (mobocmf/layers/mfdgp_hidden_layer.py:326-337, :402-444; mobocmf/util/moop.py:232-272).
//
//   kind 0:  f(x) = sum_j theta[j] s0 cos(W1[j].x + b1[j])
//   kind 1:  f(x) = sum_j  theta[j]      s1 fprev cos(W1[j].x + b1[j])                 (s1 includes sqrt(nu): linear kernel)
//                        + theta[F+j]    s1f cos(W1[j].x + Wf[j] fprev + b1[j])         (RBF on [x, f], shares W1, b1)
//                        + theta[2F+j]   s2 cos(W2[j].x + b2[j])
//
// One thread per grid point (its x row in registers), features streamed through LDS in chunks and read as wave-wide
// broadcasts; the F x n feature matrix the reference materialises (and a library matmul + cos + matvec would, 256 MB per
// block at d = 8) never exists.  The argument W1.x is shared by the first two blocks of kind 1.  FP64 cos-bound.
#include "common.h"

#define RFF_T 256     // threads = grid points per workgroup
#define RFF_FC 64     // features per LDS chunk

template <int KIND, int DB>
__global__ __launch_bounds__(RFF_T) void rff_eval_kernel(int d, int F, int64_t n, const double* __restrict__ x,
                                                         const double* __restrict__ fprev, const double* __restrict__ W1,
                                                         const double* __restrict__ b1, const double* __restrict__ Wf,
                                                         const double* __restrict__ W2, const double* __restrict__ b2,
                                                         const double* __restrict__ theta, double s0, double s1, double s2,
                                                         double* __restrict__ out) {
    __shared__ double w1s[RFF_FC][DB], w2s[KIND ? RFF_FC : 1][DB];
    __shared__ double b1s[RFF_FC], b2s[RFF_FC], wfs[RFF_FC], t0s[RFF_FC], t1s[RFF_FC], t2s[RFF_FC];
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * RFF_T + tid;
    const bool live = i < n;
    double xr[DB];
#pragma unroll
    for (int k = 0; k < DB; ++k) xr[k] = (live && k < d) ? x[i * d + k] : 0.0;
    const double fp = (KIND == 1 && live) ? fprev[i] : 0.0;
    double acc = 0.0;
    for (int f0 = 0; f0 < F; f0 += RFF_FC) {
        const int fc = F - f0 < RFF_FC ? F - f0 : RFF_FC;
        __syncthreads();
        for (int e = tid; e < fc * d; e += RFF_T) {
            const int j = e / d, k = e % d;
            w1s[j][k] = W1[(int64_t)(f0 + j) * d + k];
            if (KIND == 1) w2s[j][k] = W2[(int64_t)(f0 + j) * d + k];
        }
        if (tid < fc) {
            b1s[tid] = b1[f0 + tid];
            t0s[tid] = theta[f0 + tid];
            if (KIND == 1) {
                b2s[tid] = b2[f0 + tid];
                wfs[tid] = Wf[f0 + tid];
                t1s[tid] = theta[F + f0 + tid];
                t2s[tid] = theta[2 * F + f0 + tid];
            }
        }
        __syncthreads();
        for (int j = 0; j < fc; ++j) {
            double a1 = b1s[j], a2 = KIND == 1 ? b2s[j] : 0.0;
#pragma unroll
            for (int k = 0; k < DB; ++k) {
                if (k < d) {
                    a1 += w1s[j][k] * xr[k];
                    if (KIND == 1) a2 += w2s[j][k] * xr[k];
                }
            }
            if (KIND == 0) {
                acc += t0s[j] * cos(a1);
            } else {
                acc += t0s[j] * (s0 * fp) * cos(a1) + t1s[j] * s1 * cos(a1 + wfs[j] * fp) + t2s[j] * s2 * cos(a2);
            }
        }
    }
    if (live) out[i] = KIND == 0 ? s0 * acc : acc;
}

extern "C" int mobocmf_rff_eval(int32_t kind, int32_t d, int32_t F, int64_t n, const double* x, const double* fprev,
                                const double* W1, const double* b1, const double* Wf, const double* W2, const double* b2,
                                const double* theta, double s0, double s1, double s2, double* out,
                                mobocmf_stream_t stream) {
    if ((kind != 0 && kind != 1) || d < 1 || d > MOBOCMF_MAX_D || F < 1 || n < 1 || !x || !W1 || !b1 || !theta || !out)
        return MOBOCMF_BAD_ARG;
    if (kind == 1 && (!fprev || !Wf || !W2 || !b2)) return MOBOCMF_BAD_ARG;
    const dim3 grid((unsigned)((n + RFF_T - 1) / RFF_T)), block(RFF_T);
    hipStream_t s = (hipStream_t)stream;
#define RFF_GO(K, D) hipLaunchKernelGGL((rff_eval_kernel<K, D>), grid, block, 0, s, d, F, n, x, fprev, W1, b1, Wf, W2, b2, theta, s0, s1, s2, out)
    if (kind == 0) { if (d <= 2) RFF_GO(0, 2); else if (d <= 8) RFF_GO(0, 8); else RFF_GO(0, 32); }
    else { if (d <= 2) RFF_GO(1, 2); else if (d <= 8) RFF_GO(1, 8); else RFF_GO(1, 32); }
#undef RFF_GO
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}
