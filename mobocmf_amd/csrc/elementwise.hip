// Small / memory-bound kernels of the MFDGP path (gfx950): padding, transposes, M x M glue of the
// Cholesky backward chain, predictive-moment finish, KL, sample propagation, ELBO data term,
// acquisition moments, fused Adam.  All HBM-bound: coalesced, one pass, reductions by wavefront shuffles.
#include "common.h"

#define LOG_2PI 1.8378770664093454835606594728112

__device__ __forceinline__ double wave_sum(double v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// block-wide sum (blockDim.x = 256), result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    return t;
}

// the same for any block size up to 1024 threads (sh: 16 doubles)
__device__ __forceinline__ double block_sum_wide(double v, double* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) t += sh[w];
    return t;
}

#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256)
#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR)

__global__ void zero32_kernel(uint32_t* p, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
int launch_zero32(void* ptr, int64_t nwords, hipStream_t s) {
    if (nwords <= 0) return MOBOCMF_OK;
    hipLaunchKernelGGL(zero32_kernel, GRID1(nwords), 0, s, (uint32_t*)ptr, nwords);
    return CHECK_LAUNCH();
}
// Layer batching of the M x M chain (several layers' chains in one launch): blockIdx.z = layer, every workspace pointer of
// layer z is the layer-0 pointer + z*zs (the layers' chain blocks lie a constant stride apart); per-layer USER tensors
// (g_kl, info, ...) travel as small pointer tables.
#define GRIDZ(n, nz) dim3((unsigned)(((n) + 255) / 256), 1, (unsigned)(nz)), dim3(256)
struct ScalZ { const double* p[MAX_ZL]; };
__global__ void zero32_z_kernel(uint32_t* p, int64_t n, int64_t zs_words) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[blockIdx.z * zs_words + i] = 0u;
}
int launch_zero32_z(void* ptr, int64_t nwords, int nz, int64_t zs_bytes, hipStream_t s) {
    if (nwords <= 0) return MOBOCMF_OK;
    hipLaunchKernelGGL(zero32_z_kernel, GRIDZ(nwords, nz), 0, s, (uint32_t*)ptr, nwords, zs_bytes / 4);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ M x M helpers
// dst (Mp x Mp) = tril(src (M x M, ld lds)) zero padded
__global__ void pad_tril_kernel(const double* src, int64_t lds, int M, double* dst, int Mp) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Mp * Mp) return;
    int i = (int)(idx / Mp), j = (int)(idx % Mp);
    dst[idx] = (i < M && j <= i) ? src[(int64_t)i * lds + j] : 0.0;
}
int launch_pad_tril(const double* src, int64_t lds, int M, double* dst, int Mp, hipStream_t s) {
    hipLaunchKernelGGL(pad_tril_kernel, GRID1((int64_t)Mp * Mp), 0, s, src, lds, M, dst, Mp);
    return CHECK_LAUNCH();
}

__global__ void pad_vec_kernel(const double* src, int64_t n, double* dst, int64_t np) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) dst[i] = (src && i < n) ? src[i] : 0.0;
}
int launch_pad_vec(const double* src, int64_t n, double* dst, int64_t np, hipStream_t s) {
    hipLaunchKernelGGL(pad_vec_kernel, GRID1(np), 0, s, src, n, dst, np);
    return CHECK_LAUNCH();
}

// the same for every layer of a chain batch in ONE launch: LSp + z*zs = tril(L_S[z]) padded, mp + z*zs = m[z] padded
struct PadZ { const double* LS[MAX_ZL]; const double* m[MAX_ZL]; };
__global__ void pad_params_z_kernel(PadZ t, int M, double* LSp, double* mp, int Mp, int64_t zs) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Mp * Mp) return;
    const int z = blockIdx.z;
    int i = (int)(idx / Mp), j = (int)(idx % Mp);
    LSp[z * zs + idx] = (i < M && j <= i) ? t.LS[z][(int64_t)i * M + j] : 0.0;
    if (idx < Mp) mp[z * zs + idx] = idx < M ? t.m[z][idx] : 0.0;
}
int launch_pad_params_z(const double* const* LS, const double* const* m, int M, double* LSp, double* mp, int Mp, int nz,
                        int64_t zs, hipStream_t s) {
    PadZ t = {};
    for (int z = 0; z < nz; ++z) { t.LS[z] = LS[z]; t.m[z] = m[z]; }
    hipLaunchKernelGGL(pad_params_z_kernel, GRIDZ((int64_t)Mp * Mp, nz), 0, s, t, M, LSp, mp, Mp, zs);
    return CHECK_LAUNCH();
}

// out[c][r] = in[r][c]   (rows x cols -> cols x rows), 32x32 LDS tiles
__global__ void transpose_kernel(const double* in, int64_t ldi, double* out, int64_t ldo, int64_t rows, int64_t cols,
                                 int64_t zs) {
    __shared__ double t[32][33];
    in += blockIdx.z * zs;
    out += blockIdx.z * zs;
    int64_t c = (int64_t)blockIdx.x * 32 + threadIdx.x, r0 = (int64_t)blockIdx.y * 32;
    for (int k = threadIdx.y; k < 32; k += 8)
        if (r0 + k < rows && c < cols) t[k][threadIdx.x] = in[(r0 + k) * ldi + c];
    __syncthreads();
    int64_t r = r0 + threadIdx.x, c0 = (int64_t)blockIdx.x * 32;
    for (int k = threadIdx.y; k < 32; k += 8)
        if (c0 + k < cols && r < rows) out[(c0 + k) * ldo + r] = t[threadIdx.x][k];
}
int launch_transpose_z(const double* in, int64_t ldi, double* out, int64_t ldo, int64_t rows, int64_t cols, int nz,
                       int64_t zs, hipStream_t s) {
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32), (unsigned)nz);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, s, in, ldi, out, ldo, rows, cols, zs);
    return CHECK_LAUNCH();
}
// Two pairs of independent launches of the chain forward as ONE launch each (the chain is a string of latency-bound launches:
// at the reference's own sizes every one of them is ~4.5 us of a ~300 us step):
//   transpose_pad_z:   out = in^T (Mp x Mp, L^-1 -> L^-T)    +  the padded copies of the user tensors L_S, m of every layer
//   transpose_gemv_z:  out = in^T (Mp x Mp, U -> U^T)        +  vout = Mat vec (a = L^-1 m), one wavefront per row
__global__ void transpose_pad_z_kernel(const double* in, double* out, int Mp, int64_t zs, PadZ t, int M, double* LSp, double* mp) {
    __shared__ double tl[32][33];
    const int z = blockIdx.z;
    in += z * zs; out += z * zs;
    const int64_t c = (int64_t)blockIdx.x * 32 + threadIdx.x, r0 = (int64_t)blockIdx.y * 32;
    for (int k = threadIdx.y; k < 32; k += 8) {
        const int64_t i = r0 + k;
        tl[k][threadIdx.x] = in[i * Mp + c];
        LSp[z * zs + i * Mp + c] = (i < M && c <= i) ? t.LS[z][i * M + c] : 0.0;
        if (i == 0) mp[z * zs + c] = c < M ? t.m[z][c] : 0.0;
    }
    __syncthreads();
    const int64_t r = r0 + threadIdx.x, c0 = (int64_t)blockIdx.x * 32;
    for (int k = threadIdx.y; k < 32; k += 8) out[(c0 + k) * Mp + r] = tl[threadIdx.x][k];
}
int launch_transpose_pad_z(const double* in, double* out, int Mp, const double* const* LS, const double* const* m, int M,
                           double* LSp, double* mp, int nz, int64_t zs, hipStream_t s) {
    PadZ t = {};
    for (int z = 0; z < nz; ++z) { t.LS[z] = LS[z]; t.m[z] = m[z]; }
    hipLaunchKernelGGL(transpose_pad_z_kernel, dim3((unsigned)(Mp / 32), (unsigned)(Mp / 32), (unsigned)nz), dim3(32, 8), 0, s, in,
                       out, Mp, zs, t, M, LSp, mp);
    return CHECK_LAUNCH();
}
__global__ void transpose_gemv_z_kernel(const double* in, double* out, int Mp, int64_t zs, const double* Mat, const double* vec,
                                        double* vout) {
    __shared__ double tl[32][33];
    const int z = blockIdx.z;
    in += z * zs; out += z * zs; Mat += z * zs; vec += z * zs; vout += z * zs;
    const int64_t c = (int64_t)blockIdx.x * 32 + threadIdx.x, r0 = (int64_t)blockIdx.y * 32;
    for (int k = threadIdx.y; k < 32; k += 8) tl[k][threadIdx.x] = in[(r0 + k) * Mp + c];
    // the matrix-vector product: this block's share of the rows, a wavefront per row
    {
        const int tid = threadIdx.y * 32 + threadIdx.x, wave = tid >> 6, lane = tid & 63;
        const int nblk = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
        for (int row = 4 * b + wave; row < Mp; row += 4 * nblk) {
            const double* p = Mat + (int64_t)row * Mp;
            double sacc = 0.0;
            for (int j = lane; j < Mp; j += 64) sacc += p[j] * vec[j];
            sacc = wave_sum(sacc);
            if (lane == 0) vout[row] = sacc;
        }
    }
    __syncthreads();
    const int64_t r = r0 + threadIdx.x, c0 = (int64_t)blockIdx.x * 32;
    for (int k = threadIdx.y; k < 32; k += 8) out[(c0 + k) * Mp + r] = tl[threadIdx.x][k];
}
int launch_transpose_gemv_z(const double* in, double* out, int Mp, const double* Mat, const double* vec, double* vout, int nz,
                            int64_t zs, hipStream_t s) {
    hipLaunchKernelGGL(transpose_gemv_z_kernel, dim3((unsigned)(Mp / 32), (unsigned)(Mp / 32), (unsigned)nz), dim3(32, 8), 0, s, in,
                       out, Mp, zs, Mat, vec, vout);
    return CHECK_LAUNCH();
}
int launch_transpose(const double* in, int64_t ldi, double* out, int64_t ldo, int64_t rows, int64_t cols, hipStream_t s) {
    return launch_transpose_z(in, ldi, out, ldo, rows, cols, 1, 0, s);
}

// out[i] = sum_j Mat[i][j] * vec[j]   (one wavefront per row), optional accumulate
__global__ void gemv_rows_kernel(const double* Mat, int64_t ld, const double* vec, double* out, int rows, int64_t cols,
                                 double scale, int accumulate, int64_t zs) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= rows) return;
    Mat += blockIdx.z * zs; vec += blockIdx.z * zs; out += blockIdx.z * zs;
    const double* p = Mat + (int64_t)row * ld;
    double s = 0.0;
    for (int64_t j = lane; j < cols; j += 64) s += p[j] * vec[j];
    s = wave_sum(s) * scale;
    if (lane == 0) out[row] = accumulate ? out[row] + s : s;
}
int launch_gemv_rows_z(const double* Mat, int64_t ld, const double* vec, double* out, int rows, int64_t cols, double scale,
                       int accumulate, int nz, int64_t zs, hipStream_t s) {
    hipLaunchKernelGGL(gemv_rows_kernel, dim3((rows + 3) / 4, 1, nz), dim3(256), 0, s, Mat, ld, vec, out, rows, cols, scale,
                       accumulate, zs);
    return CHECK_LAUNCH();
}
int launch_gemv_rows(const double* Mat, int64_t ld, const double* vec, double* out, int rows, int64_t cols, double scale,
                     int accumulate, hipStream_t s) {
    return launch_gemv_rows_z(Mat, ld, vec, out, rows, cols, scale, accumulate, 1, 0, s);
}

// KL = 0.5 * (2 sum log L_ii - sum log LS_ii^2 + |U|_F^2 + |a|^2 - M)      
// stage 1: one block per 4 rows of U (lower triangle) -> partial sums; stage 2: one block adds them
__global__ void kl_part_kernel(const double* L, const double* LSp, const double* U, const double* a, int M, int Mp,
                               double* part, int64_t zs) {
    __shared__ double sh[4];
    double s = 0.0;
    const int r0 = blockIdx.x * 4;
    L += blockIdx.z * zs; LSp += blockIdx.z * zs; U += blockIdx.z * zs; a += blockIdx.z * zs; part += blockIdx.z * zs;
    for (int rr = 0; rr < 4; ++rr) {
        const int i = r0 + rr;
        if (i >= M) break;
        const double* u = U + (int64_t)i * Mp;
        for (int j = threadIdx.x; j <= i; j += 256) s += u[j] * u[j];
        if (threadIdx.x == 0) {
            double l = L[(int64_t)i * Mp + i], ls = LSp[(int64_t)i * Mp + i];
            s += 2.0 * log(l) - log(ls * ls) + a[i] * a[i];
        }
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
struct KlZ { double* p[MAX_ZL]; };      // per-layer KL outputs (user tensors: not strided)
__global__ void kl_final_kernel(const double* part, int np, int M, KlZ kl, int64_t zs) {
    __shared__ double sh[4];
    double s = 0.0;
    part += blockIdx.x * zs;
    for (int i = threadIdx.x; i < np; i += 256) s += part[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) kl.p[blockIdx.x][0] = 0.5 * (s - (double)M);
}
// small M (<= 128: the reference's own sizes): one block of 1024 threads per layer does both stages -- one launch instead of
// two.  Element-parallel with the loads unrolled (a first version walked a row per wavefront: 32 dependent row reads, 41 us).
__global__ __launch_bounds__(1024) void kl_one_kernel(const double* L, const double* LSp, const double* U, const double* a, int M,
                                                      int Mp, KlZ kl, int64_t zs) {
    __shared__ double sh[16];
    const int z = blockIdx.x;
    L += z * zs; LSp += z * zs; U += z * zs; a += z * zs;
    double s = 0.0;
    const int total = M * Mp;
#pragma unroll 4
    for (int e = threadIdx.x; e < total; e += 1024) {
        const int i = e / Mp, j = e - i * Mp;
        const double u = U[e];
        s += j <= i ? u * u : 0.0;
    }
    for (int i = threadIdx.x; i < M; i += 1024) {
        const double l = L[(int64_t)i * Mp + i], ls = LSp[(int64_t)i * Mp + i];
        s += 2.0 * log(l) - log(ls * ls) + a[i] * a[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += sh[w];
        kl.p[z][0] = 0.5 * (t - (double)M);
    }
}
int launch_kl_z(const double* L, const double* LSp, const double* U, const double* a, int M, int Mp, double* const* kl,
                double* part, int nz, int64_t zs, hipStream_t s) {
    if (M <= 128) {
        KlZ kz1 = {};
        for (int z = 0; z < nz; ++z) kz1.p[z] = kl[z];
        hipLaunchKernelGGL(kl_one_kernel, dim3(nz), dim3(1024), 0, s, L, LSp, U, a, M, Mp, kz1, zs);
        return CHECK_LAUNCH();
    }
    const int nb = (M + 3) / 4;
    hipLaunchKernelGGL(kl_part_kernel, dim3(nb, 1, nz), dim3(256), 0, s, L, LSp, U, a, M, Mp, part, zs);
    KlZ kz = {};
    for (int z = 0; z < nz; ++z) kz.p[z] = kl[z];      // kl[z] is a user tensor
    hipLaunchKernelGGL(kl_final_kernel, dim3(nz), dim3(256), 0, s, (const double*)part, nb, M, kz, zs);
    return CHECK_LAUNCH();
}
int launch_kl(const double* L, const double* LSp, const double* U, const double* a, int M, int Mp, double* kl,
              double* part, hipStream_t s) {
    double* one[1] = {kl};
    return launch_kl_z(L, LSp, U, a, M, Mp, one, part, 1, 0, s);
}

// ------------------------------------------------------------------ predictive moments
// q = sum_rb qpart, mean = sum_rb mupart, r = sum_rb rpart; var_raw = (branch ? knn - q : max(knn - q, 0)) + r
__global__ void moments_finish_kernel(const double* qpart, const double* mupart, const double* rpart, int nrb,
                                      int64_t Np, int64_t N, const double* knn, int branch, double min_var, double* q,
                                      double* r, double* varraw, double* mean, double* var, int32_t* zero_word) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0 && zero_word) zero_word[0] = 0;      // the backward's clamped-column counter (no zero launch there)
    if (n >= Np) return;
    double qs = 0, ms = 0, rs = 0;
    for (int b = 0; b < nrb; ++b) {
        qs += qpart[(int64_t)b * Np + n];
        ms += mupart[(int64_t)b * Np + n];
        rs += rpart[(int64_t)b * Np + n];
    }
    double sres = knn[n] - qs;
    if (!branch && sres < 0.0) sres = 0.0;
    double v = sres + rs;
    q[n] = qs;
    r[n] = rs;
    varraw[n] = v;
    if (n < N) {
        mean[n] = ms;
        var[n] = v < min_var ? min_var : v;
    }
}
int launch_moments_finish(const double* qpart, const double* mupart, const double* rpart, int nrb, int64_t Np, int64_t N,
                          const double* knn, int branch, double min_var, double* q, double* r, double* varraw,
                          double* mean, double* var, int32_t* zero_word, hipStream_t s) {
    hipLaunchKernelGGL(moments_finish_kernel, GRID1(Np), 0, s, qpart, mupart, rpart, nrb, Np, N, knn, branch, min_var, q,
                       r, varraw, mean, var, zero_word);
    return CHECK_LAUNCH();
}

// backward prep: gmu (padded), gv = g_var * [varraw > min_var], gv2 = 2*gv, cgv = gv * [branch || knn - q > 0]
__global__ void moments_bwd_prep_kernel(const double* g_mean, const double* g_var, const double* knn, const double* q,
                                        const double* varraw, int branch, double min_var, int64_t N, int64_t Np,
                                        double* gmu, double* gv, double* gv2, double* cgv, int32_t* nclamped, int32_t* blkact) {
    // 256 threads = two 128-column blocks; Np is a multiple of 128
    __shared__ int any_w[4];
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double gm = 0.0, g = 0.0, c = 0.0;
    if (n < N) {
        gm = g_mean ? g_mean[n] : 0.0;
        g = (g_var && varraw[n] > min_var) ? g_var[n] : 0.0;   // clamp_min passes gradient only above the floor
        c = (branch || knn[n] - q[n] > 0.0) ? g : 0.0;
        if (c != g) atomicAdd(nclamped, 1);                      // columns where clamp(k_nn - q, 0) is active (rare)
    }
    if (n < Np) {
        gmu[n] = gm;
        gv[n] = g;
        gv2[n] = 2.0 * g;
        cgv[n] = c;
    }
    if (blkact) {
        // a column block is ACTIVE if any upstream gradient of its columns is non-zero (NaN counts: it compares unequal);
        // the backward products skip inactive blocks, whose share is exactly zero (GemmArgs.colact / kact)
        const bool any = !(gm == 0.0) || !(g == 0.0);      // c is g or 0
        const unsigned long long b = __ballot(any);
        if ((threadIdx.x & 63) == 0) any_w[threadIdx.x >> 6] = b != 0ull ? 1 : 0;
        __syncthreads();
        if (threadIdx.x < 2) {
            const int64_t blk = (int64_t)blockIdx.x * 2 + threadIdx.x;
            if (blk * 128 < Np) blkact[blk] = any_w[2 * threadIdx.x] | any_w[2 * threadIdx.x + 1];
        }
    }
}
int launch_moments_bwd_prep(const double* g_mean, const double* g_var, const double* knn, const double* q,
                            const double* varraw, int branch, double min_var, int64_t N, int64_t Np, double* gmu,
                            double* gv, double* gv2, double* cgv, int32_t* nclamped, int zeroed, int32_t* blkact, hipStream_t s) {
    // zeroed: the forward's moments_finish cleared the counter (a second backward over the same forward then adds to a
    // non-zero count, which reads the same: only zero / non-zero matters)
    if (!zeroed && launch_zero32(nclamped, 1, s)) return MOBOCMF_HIP_ERROR;
    hipLaunchKernelGGL(moments_bwd_prep_kernel, GRID1(Np), 0, s, g_mean, g_var, knn, q, varraw, branch, min_var, N, Np,
                       gmu, gv, gv2, cgv, nclamped, blkact);
    return CHECK_LAUNCH();
}

// sum of one element over ns slabs, eight loads in flight at a time, ALWAYS added in the order 0, 1, 2, ... (the result does not
// depend on the batching: graph replay == eager, bit for bit).  The plain loop issued one load per trip and waited for it.
__device__ __forceinline__ double slab_sum(const double* p, int64_t slab_stride, int ns) {
    double v = 0.0;
    int z = 0;
    for (; z + 8 <= ns; z += 8) {
        double t[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) t[b] = p[(z + b) * slab_stride];
#pragma unroll
        for (int b = 0; b < 8; ++b) v += t[b];
    }
    for (; z < ns; ++z) v += p[z * slab_stride];
    return v;
}

// H (full symmetric) from the slabs of a lower_out syrk: out[i][j] = sum_z slab[z][max-tile order].  If `flag` is
// given and *flag == 0 the slabs were never written (kernel skipped): out = fallback instead.
__global__ void reduce_slabs_sym_kernel(const double* slabs, int64_t slab_stride, int nslab, int nslab_diag, double* out,
                                        int Mp, const int32_t* flag, const double* fallback) {
    // one thread per element of the LOWER tiles (coalesced slab reads); it also writes the mirrored element
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Mp * Mp) return;
    int i = (int)(idx / Mp), j = (int)(idx % Mp);
    if (flag && *flag == 0) {
        out[idx] = fallback[idx];
        return;
    }
    if (j / TILE > i / TILE) return;
    // diagonal tiles: the 16 x 16 blocks strictly above the diagonal are not computed by the syrk (GemmArgs.sym_out) -- every
    // element above the diagonal takes its mirror image, which also makes the result exactly symmetric
    const double* p = j > i ? slabs + (int64_t)j * Mp + i : slabs + (int64_t)i * Mp + j;
    const int ns = (i / TILE == j / TILE) ? nslab_diag : nslab;      // diagonal tiles may have their own slice count
    const double v = slab_sum(p, slab_stride, ns);
    out[idx] = v;
    if (j / TILE < i / TILE) out[(int64_t)j * Mp + i] = v;     // strictly-upper tiles mirror the lower ones
}
// the pair of a dual syrk launch: out = sum of slabs (always), out2 = sum of slabs2 if *flag != 0, else = out
__global__ void reduce_slabs_sym2_kernel(const double* slabs, const double* slabs2, int64_t slab_stride, int nslab,
                                         int nslab_diag, double* out, double* out2, int Mp, const int32_t* flag) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Mp * Mp) return;
    int i = (int)(idx / Mp), j = (int)(idx % Mp);
    if (j / TILE > i / TILE) return;
    const int64_t off = j > i ? (int64_t)j * Mp + i : (int64_t)i * Mp + j;
    const int ns = (i / TILE == j / TILE) ? nslab_diag : nslab;
    const double v = slab_sum(slabs + off, slab_stride, ns);
    double v2 = v;
    if (*flag != 0) v2 = slab_sum(slabs2 + off, slab_stride, ns);
    out[idx] = v;
    out2[idx] = v2;
    if (j / TILE < i / TILE) { out[(int64_t)j * Mp + i] = v; out2[(int64_t)j * Mp + i] = v2; }
}
int launch_reduce_slabs_sym2(const double* slabs, const double* slabs2, int64_t slab_stride, int nslab, int nslab_diag,
                             double* out, double* out2, int Mp, const int32_t* flag, hipStream_t s) {
    hipLaunchKernelGGL(reduce_slabs_sym2_kernel, GRID1((int64_t)Mp * Mp), 0, s, slabs, slabs2, slab_stride, nslab, nslab_diag,
                       out, out2, Mp, flag);
    return CHECK_LAUNCH();
}
int launch_reduce_slabs_sym(const double* slabs, int64_t slab_stride, int nslab, int nslab_diag, double* out, int Mp,
                            const int32_t* flag, const double* fallback, hipStream_t s) {
    hipLaunchKernelGGL(reduce_slabs_sym_kernel, GRID1((int64_t)Mp * Mp), 0, s, slabs, slab_stride, nslab, nslab_diag, out, Mp,
                       flag, fallback);
    return CHECK_LAUNCH();
}

// dU_tot = 2 tril(X) + gkl U with X = H U = (U^T H)^T = G1^T (H is exactly symmetric: the slab reduction mirrors it) -- the
// product X is never formed, its lower triangle is read out of G1 transposed (32 x 32 tiles through LDS);
// da_tot = da + gkl a;  Y = 2 G2 - 2 Hc + a da^T + da_tot a^T.  One launch for all three, all layers (blockIdx.z).
__global__ void dutot_y_kernel(const double* G1, const double* G2, const double* Hc, const double* U, const double* da,
                               const double* a, ScalZ gklz, int Mp, double* dU, double* da_tot, double* Y, int64_t zs) {
    __shared__ double tl[32][33];
    const double* gkl = gklz.p[blockIdx.z];
    const double g = gkl ? gkl[0] : 0.0;
    const int64_t zo = blockIdx.z * zs;
    G1 += zo; G2 += zo; Hc += zo; U += zo; da += zo; a += zo; dU += zo; da_tot += zo; Y += zo;
    const int nt = Mp / 32, ti = blockIdx.x / nt, tj = blockIdx.x % nt;      // tile (ti, tj) of the outputs
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                   // 32 x 8 threads
    if (tj <= ti) {      // the tile touches the lower triangle: stage G1's tile (tj, ti), coalesced along its rows
#pragma unroll
        for (int r = ty; r < 32; r += 8) tl[r][tx] = G1[(int64_t)(tj * 32 + r) * Mp + ti * 32 + tx];
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int i = ti * 32 + r, j = tj * 32 + tx;
        const int64_t idx = (int64_t)i * Mp + j;
        const double ai = a[i], aj = a[j], dai = da[i], daj = da[j];
        dU[idx] = j <= i ? 2.0 * tl[tx][r] + g * U[idx] : 0.0;      // X[i][j] = G1[j][i]
        Y[idx] = 2.0 * (G2[idx] - Hc[idx]) + ai * daj + (dai + g * ai) * aj;
        if (tj == 0 && tx == 0) da_tot[i] = dai + g * ai;
    }
}
int launch_dutot_y_z(const double* G1, const double* G2, const double* Hc, const double* U, const double* da, const double* a,
                     const double* const* gkl, int Mp, double* dU, double* da_tot, double* Y, int nz, int64_t zs, hipStream_t s) {
    ScalZ t = {};
    for (int z = 0; z < nz; ++z) t.p[z] = gkl[z];
    const int nt = Mp / 32;
    hipLaunchKernelGGL(dutot_y_kernel, dim3((unsigned)(nt * nt), 1, (unsigned)nz), dim3(256), 0, s, G1, G2, Hc, U, da, a, t, Mp,
                       dU, da_tot, Y, zs);
    return CHECK_LAUNCH();
}


// ------------------------------------------------------------------ Cholesky-chain backward glue (Mp x Mp)
// dU_tot = dU + gkl*U ;  da_tot = da + gkl*a        (in place on dU, da)
__global__ void add_kl_terms_kernel(double* dU, const double* U, double* da, const double* a, const double* gkl, int Mp) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double g = gkl ? gkl[0] : 0.0;
    if (idx < (int64_t)Mp * Mp) dU[idx] += g * U[idx];
    if (idx < Mp) da[idx] += g * a[idx];
}
int launch_add_kl_terms(double* dU, const double* U, double* da, const double* a, const double* gkl, int Mp, hipStream_t s) {
    hipLaunchKernelGGL(add_kl_terms_kernel, GRID1((int64_t)Mp * Mp), 0, s, dU, U, da, a, gkl, Mp);
    return CHECK_LAUNCH();
}

// X[i][j] += da[i] * m[j]   (rank-1, lower part is what matters)
__global__ void rank1_add_kernel(double* X, const double* u, const double* v, int Mp) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Mp * Mp) return;
    int i = (int)(idx / Mp), j = (int)(idx % Mp);
    X[idx] += u[i] * v[j];
}
int launch_rank1_add(double* X, const double* u, const double* v, int Mp, hipStream_t s) {
    hipLaunchKernelGGL(rank1_add_kernel, GRID1((int64_t)Mp * Mp), 0, s, X, u, v, Mp);
    return CHECK_LAUNCH();
}

// dL = -tril(T2) + gkl * diag(1/L_ii)  (rows/cols < M only; zero elsewhere)
__global__ void dl_from_t2_kernel(const double* T2, const double* L, ScalZ gklz, int M, int Mp, double* dL, int64_t zs) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Mp * Mp) return;
    const double* gkl = gklz.p[blockIdx.z];
    T2 += blockIdx.z * zs; L += blockIdx.z * zs; dL += blockIdx.z * zs;
    int i = (int)(idx / Mp), j = (int)(idx % Mp);
    double v = 0.0;
    if (i < M && j <= i) {
        v = -T2[idx];
        if (i == j && gkl) v += gkl[0] / L[idx];
    }
    dL[idx] = v;
}
int launch_dl_from_t2_z(const double* T2, const double* L, const double* const* gkl, int M, int Mp, double* dL, int nz,
                        int64_t zs, hipStream_t s) {
    ScalZ t = {};
    for (int z = 0; z < nz; ++z) t.p[z] = gkl[z];
    hipLaunchKernelGGL(dl_from_t2_kernel, GRIDZ((int64_t)Mp * Mp, nz), 0, s, T2, L, t, M, Mp, dL, zs);
    return CHECK_LAUNCH();
}
int launch_dl_from_t2(const double* T2, const double* L, const double* gkl, int M, int Mp, double* dL, hipStream_t s) {
    const double* one[1] = {gkl};
    return launch_dl_from_t2_z(T2, L, one, M, Mp, dL, 1, 0, s);
}

// P = Phi(T3): lower triangle with halved diagonal
__global__ void phi_kernel(const double* T3, int Mp, double* P, int64_t zs) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Mp * Mp) return;
    T3 += blockIdx.z * zs; P += blockIdx.z * zs;
    int i = (int)(idx / Mp), j = (int)(idx % Mp);
    P[idx] = j < i ? T3[idx] : (j == i ? 0.5 * T3[idx] : 0.0);
}
int launch_phi_z(const double* T3, int Mp, double* P, int nz, int64_t zs, hipStream_t s) {
    hipLaunchKernelGGL(phi_kernel, GRIDZ((int64_t)Mp * Mp, nz), 0, s, T3, Mp, P, zs);
    return CHECK_LAUNCH();
}
int launch_phi(const double* T3, int Mp, double* P, hipStream_t s) { return launch_phi_z(T3, Mp, P, 1, 0, s); }

// G = (S + S^T)/2
__global__ void symmetrize_kernel(const double* S, int Mp, double* G, int64_t zs) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Mp * Mp) return;
    S += blockIdx.z * zs; G += blockIdx.z * zs;
    int i = (int)(idx / Mp), j = (int)(idx % Mp);
    G[idx] = 0.5 * (S[idx] + S[(int64_t)j * Mp + i]);
}
int launch_symmetrize_z(const double* S, int Mp, double* G, int nz, int64_t zs, hipStream_t s) {
    hipLaunchKernelGGL(symmetrize_kernel, GRIDZ((int64_t)Mp * Mp, nz), 0, s, S, Mp, G, zs);
    return CHECK_LAUNCH();
}
int launch_symmetrize(const double* S, int Mp, double* G, hipStream_t s) { return launch_symmetrize_z(S, Mp, G, 1, 0, s); }

// g_LS (M x M, ld M) = tril(X (Mp x Mp)) - gkl * diag(1/LS_ii)
__global__ void gls_out_kernel(const double* X, const double* LSp, const double* gkl, int M, int Mp, double* gLS) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * M) return;
    int i = (int)(idx / M), j = (int)(idx % M);
    double v = 0.0;
    if (j <= i) {
        v = X[(int64_t)i * Mp + j];
        if (i == j && gkl) v -= gkl[0] / LSp[(int64_t)i * Mp + i];
    }
    gLS[idx] = v;
}
int launch_gls_out(const double* X, const double* LSp, const double* gkl, int M, int Mp, double* gLS, hipStream_t s) {
    hipLaunchKernelGGL(gls_out_kernel, GRID1((int64_t)M * M), 0, s, X, LSp, gkl, M, Mp, gLS);
    return CHECK_LAUNCH();
}

// Both user-tensor outputs of the chain backward for every layer in ONE launch (blockIdx.y = layer):
//   g_LS[z] (M x M, ld M) = tril(X_z) - gkl_z diag(1 / LS_ii)      (blocks [0, nb_ls))
//   g_m[z]  (M)           = L^-T_z da_tot_z                         (blocks [nb_ls, ...): one wavefront per row)
struct ChainOutZ { const double* gkl[MAX_ZL]; double* gLS[MAX_ZL]; double* gm[MAX_ZL]; };
__global__ void chain_outputs_z_kernel(const double* X, const double* LSp, const double* LinvT, const double* da_tot, int M,
                                       int Mp, ChainOutZ t, int nb_ls, int64_t zs) {
    const int z = blockIdx.y;
    X += z * zs; LSp += z * zs; LinvT += z * zs; da_tot += z * zs;
    if ((int)blockIdx.x < nb_ls) {
        const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (idx >= (int64_t)M * M) return;
        const int i = (int)(idx / M), j = (int)(idx % M);
        double v = 0.0;
        if (j <= i) {
            v = X[(int64_t)i * Mp + j];
            if (i == j && t.gkl[z]) v -= t.gkl[z][0] / LSp[(int64_t)i * Mp + i];
        }
        t.gLS[z][idx] = v;
        return;
    }
    const int row = ((int)blockIdx.x - nb_ls) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const double* p = LinvT + (int64_t)row * Mp;
    double sacc = 0.0;
    for (int j = lane; j < Mp; j += 64) sacc += p[j] * da_tot[j];
    sacc = wave_sum(sacc);
    if (lane == 0) t.gm[z][row] = sacc;
}
int launch_chain_outputs_z(const double* X, const double* LSp, const double* LinvT, const double* da_tot, int M, int Mp,
                           const double* const* gkl, double* const* gLS, double* const* gm, int nz, int64_t zs, hipStream_t s) {
    ChainOutZ t = {};
    for (int z = 0; z < nz; ++z) { t.gkl[z] = gkl[z]; t.gLS[z] = gLS[z]; t.gm[z] = gm[z]; }
    const int nb_ls = (int)(((int64_t)M * M + 255) / 256), nb_m = (M + 3) / 4;
    hipLaunchKernelGGL(chain_outputs_z_kernel, dim3((unsigned)(nb_ls + nb_m), (unsigned)nz), dim3(256), 0, s, X, LSp, LinvT, da_tot,
                       M, Mp, t, nb_ls, zs);
    return CHECK_LAUNCH();
}

__global__ void copy_block_kernel(const double* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    int64_t i = idx / cols, j = idx % cols;
    dst[i * ldd + j] = src[i * lds + j];
}
int launch_copy_block(const double* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols, hipStream_t s) {
    hipLaunchKernelGGL(copy_block_kernel, GRID1(rows * cols), 0, s, src, lds, dst, ldd, rows, cols);
    return CHECK_LAUNCH();
}

// dst (symmetric, n_real x n_real, ldd) <- the lower triangle of a column panel: src [rows x cols] holds the block of rows
// row0.. and columns row0..row0+cols of a symmetric matrix, valid on and below the panel's own diagonal.  Every 32 x 32
// tile on or below that diagonal is written once as it is and once transposed (through LDS: both writes coalesced).
__global__ __launch_bounds__(256) void mirror_lower_kernel(const double* src, int64_t lds_, double* dst, int64_t ldd,
                                                           int64_t row0, int64_t rows, int64_t cols, int64_t n_real) {
    __shared__ double t[32][33];
    const int64_t ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int64_t i = ti * 32 + r, j = tj * 32 + tx;
        double v = 0.0;
        if (i < rows && j < cols) v = src[i * lds_ + j];
        t[r][tx] = v;
        if (i >= j && row0 + i < n_real && row0 + j < n_real && j < cols) dst[(row0 + i) * ldd + row0 + j] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {      // transposed: element (i, j) of the tile goes to dst[j][i]
        const int64_t j = tj * 32 + r, i = ti * 32 + tx;
        if (i > j && row0 + i < n_real && row0 + j < n_real && j < cols && i < rows) dst[(row0 + j) * ldd + row0 + i] = t[tx][r];
    }
}
int launch_mirror_lower(const double* src, int64_t lds_, double* dst, int64_t ldd, int64_t row0, int64_t rows, int64_t cols,
                        int64_t n_real, hipStream_t s) {
    const dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    hipLaunchKernelGGL(mirror_lower_kernel, grid, dim3(256), 0, s, src, lds_, dst, ldd, row0, rows, cols, n_real);
    return CHECK_LAUNCH();
}

// ---- exact-GP comparison baselines (mobocmf/models/mfgp.py:145-184, mfgp_lin.py:101-189): the multi-fidelity kernels are
// element-wise combinations of TWO ARD-RBF Gram matrices,  K[i][j] = s1[i] s2[j] Ks[i][j] + ntab[min(l1[i], l2[j])] Kn[i][j]
// (+ diag on the diagonal): MFKernel has s = 1, ntab[t] = t; MFKernel_lin has s = cumulative products of rho and
// ntab[t] = the noise factor of level t.  Output zero-padded to (rows_p x cols_p).
__global__ void mf_combine_kernel(const double* Ks, const double* Kn, int64_t ld, const double* s1, const double* s2,
                                  const int32_t* l1, const int32_t* l2, const double* ntab, int64_t n1, int64_t n2,
                                  double diag, double* out, int64_t ldo, int64_t rows_p, int64_t cols_p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows_p * cols_p) return;
    const int64_t i = idx / cols_p, j = idx % cols_p;
    double v = 0.0;
    if (i < n1 && j < n2) {
        const int a = l1[i], b = l2[j];
        v = (s1 ? s1[i] : 1.0) * (s2 ? s2[j] : 1.0) * Ks[i * ld + j] + ntab[a < b ? a : b] * Kn[i * ld + j];
        if (i == j) v += diag;
    } else if (i == j && diag != 0.0) {
        v = 1.0;      // identity on the padded diagonal of a matrix that is factorised afterwards
    }
    out[i * ldo + j] = v;
}
int launch_mf_combine(const double* Ks, const double* Kn, int64_t ld, const double* s1, const double* s2, const int32_t* l1,
                      const int32_t* l2, const double* ntab, int64_t n1, int64_t n2, double diag, double* out, int64_t ldo,
                      int64_t rows_p, int64_t cols_p, hipStream_t s) {
    hipLaunchKernelGGL(mf_combine_kernel, GRID1(rows_p * cols_p), 0, s, Ks, Kn, ld, s1, s2, l1, l2, ntab, n1, n2, diag, out,
                       ldo, rows_p, cols_p);
    return CHECK_LAUNCH();
}
// dst (np x np) = [[src (n x n), 0], [0, I]]: a matrix to factorise, identity on the padded diagonal
__global__ void copy_pad_identity_kernel(const double* src, int64_t lds_, int n, double* dst, int np) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)np * np) return;
    const int i = (int)(idx / np), j = (int)(idx % np);
    dst[idx] = (i < n && j < n) ? src[(int64_t)i * lds_ + j] : (i == j ? 1.0 : 0.0);
}
int launch_copy_pad_identity(const double* src, int64_t lds_, int n, double* dst, int np, hipStream_t s) {
    hipLaunchKernelGGL(copy_pad_identity_kernel, GRID1((int64_t)np * np), 0, s, src, lds_, n, dst, np);
    return CHECK_LAUNCH();
}
// mll = -1/2 |a|^2 - sum_i log L_ii - n/2 log 2 pi   (a = L^-1 y; one block)
__global__ void exact_gp_mll_kernel(const double* L, int64_t ld, const double* a, int n, double* mll) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += -0.5 * a[i] * a[i] - log(L[(int64_t)i * ld + i]);
    s = block_sum(s, sh);
    if (threadIdx.x == 0) mll[0] = s - 0.5 * n * 1.8378770664093453;
}
int launch_exact_gp_mll(const double* L, int64_t ld, const double* a, int n, double* mll, hipStream_t s) {
    hipLaunchKernelGGL(exact_gp_mll_kernel, dim3(1), dim3(256), 0, s, L, ld, a, n, mll);
    return CHECK_LAUNCH();
}
// mean[j] = sum_p mupart[p][j], var[j] = kss[j] - sum_p qpart[p][j]   (j < nt)
__global__ void exact_gp_finish_kernel(const double* qpart, const double* mupart, int nparts, int64_t ntp, int64_t nt,
                                       const double* kss, double* mean, double* var) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nt) return;
    double q = 0.0, m = 0.0;
    for (int p = 0; p < nparts; ++p) { q += qpart[(int64_t)p * ntp + j]; m += mupart[(int64_t)p * ntp + j]; }
    mean[j] = m;
    var[j] = kss[j] - q;
}
int launch_exact_gp_finish(const double* qpart, const double* mupart, int nparts, int64_t ntp, int64_t nt, const double* kss,
                           double* mean, double* var, hipStream_t s) {
    hipLaunchKernelGGL(exact_gp_finish_kernel, GRID1(nt), 0, s, qpart, mupart, nparts, ntp, nt, kss, mean, var);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ public elementwise entry points
__global__ void propagate_fwd_kernel(const double* mean, const double* var, const double* eps, double* f, int64_t n,
                                     int div) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t b = i / div;
    f[i] = mean[b] + sqrt(var[b]) * eps[i];
}
// ---- the same with eps drawn INSIDE the launch (mfdgp_hidden_layer.py:272-274 draws it with torch.normal): counter-based
// Philox4x32-10 keyed by (seed, call counter, row) + Box-Muller in float64.  rng_state = {seed, calls, ticket} (device int64
// x 3, owned by the layer): every block reads `calls`, the block that takes the last ticket advances it -- a captured step
// replays with fresh eps and without torch's generator (whose graph support costs two fill launches per replay and one
// launch for the draw).  eps is written out for the backward pass.
__global__ void propagate_rng_fwd_kernel(const double* mean, const double* var, int64_t* rng_state, double* f, double* eps_out,
                                         int64_t n, int div) {
    const uint64_t seed = (uint64_t)rng_state[0], call = (uint64_t)__atomic_load_n(&rng_state[1], __ATOMIC_RELAXED);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int64_t b = i / div;
        const double e = philox_normal(seed, call, (uint64_t)i);
        eps_out[i] = e;
        f[i] = mean[b] + sqrt(var[b]) * e;
    }
    __syncthreads();      // every thread of the block has read `calls`
    if (threadIdx.x == 0) {
        const unsigned long long t = atomicAdd((unsigned long long*)&rng_state[2], 1ull);
        if (t == (unsigned long long)gridDim.x - 1) {      // every block has read `calls`: advance it for the next launch
            __atomic_store_n(&rng_state[2], (int64_t)0, __ATOMIC_RELAXED);
            __atomic_store_n(&rng_state[1], (int64_t)(call + 1), __ATOMIC_RELAXED);
        }
    }
}
// rows b >= nbase of the previous layer (up to nprev) fed nothing forward (prefix propagation): zero gradient
// add_mean / add_var (may be null): gradients the SAME moments receive from their other consumer (the ELBO's data term of the
// previous layer's own fidelity) -- added here, in the launch that writes the result anyway, instead of by two autograd adds
__global__ void propagate_bwd_kernel(const double* var, const double* eps, const double* gf, double* gmean, double* gvar,
                                     int64_t nbase, int div, int64_t nprev, const double* add_mean, const double* add_var) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nprev) return;
    const double am = add_mean ? add_mean[b] : 0.0, av = add_var ? add_var[b] : 0.0;
    if (b >= nbase) {
        gmean[b] = am;
        gvar[b] = av;
        return;
    }
    double sm = 0.0, sv = 0.0;
    for (int s = 0; s < div; ++s) {
        double g = gf[b * div + s];
        sm += g;
        sv += g * eps[b * div + s];
    }
    gmean[b] = sm + am;
    gvar[b] = sv * 0.5 / sqrt(var[b]) + av;
}

__device__ __forceinline__ double elp_term(double y, double mu, double v, double tau, double ltau) {
    double dlt = y - mu;
    return -0.5 * ((dlt * dlt + v) / tau + ltau + LOG_2PI);
}
__global__ void elbo_fwd_kernel(const double* mean, const double* var, const double* y, const double* fid,
                                const double* tau_p, double lo, double hi, double level, int64_t n, int div, double* part) {
    __shared__ double sh[4];
    // hi > lo: tau_p holds the RAW noise parameter of an Interval constraint, tau = lo + (hi - lo) sigmoid(raw)
    const double tau = hi > lo ? lo + (hi - lo) / (1.0 + exp(-tau_p[0])) : tau_p[0], ltau = log(tau);
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t b = i / div;
        if (fid[b] == level) s += elp_term(y[b], mean[i], var[i], tau, ltau);
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void final_sum_kernel(const double* part, int np, double scale, double* out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) s += part[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) out[0] = s * scale;
}
__global__ void elbo_bwd_kernel(const double* mean, const double* var, const double* y, const double* fid,
                                const double* tau_p, double lo, double hi, double level, int64_t n, int div,
                                const double* gout, double* gmean, double* gvar, double* part) {
    __shared__ double sh[4];
    const double tau = hi > lo ? lo + (hi - lo) / (1.0 + exp(-tau_p[0])) : tau_p[0];
    const double g = gout[0] / div;
    double st = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t b = i / div;
        double gm = 0.0, gvv = 0.0;
        if (fid[b] == level) {
            double dlt = y[b] - mean[i];
            gm = g * dlt / tau;
            gvv = -0.5 * g / tau;
            st += 0.5 * ((dlt * dlt + var[i]) / (tau * tau) - 1.0 / tau);
        }
        gmean[i] = gm;
        gvar[i] = gvv;
    }
    st = block_sum(st, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = st;
}
__global__ void final_sum_scaled_kernel(const double* part, int np, const double* gout, double scale, double* out,
                                        const double* raw, double lo, double hi) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) s += part[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) {
        double chain = 1.0;                   // d tau / d raw of the Interval transform
        if (hi > lo) {
            const double sg = 1.0 / (1.0 + exp(-raw[0]));
            chain = (hi - lo) * sg * (1.0 - sg);
        }
        out[0] = s * scale * gout[0] * chain;
    }
}

// ---- The whole ELBO of variational_elbo_mf.py:24-51 in TWO launches each way (was: two launches per fidelity + the tail,
// and as many again in backward): blockIdx.y = layer (fidelity level), every block reduces its share of the layer's masked
// expected log-likelihood; a one-block tail adds the partial sums, the KL tail and writes the results.  (A single launch
// whose last-finishing block does the tail needs a device ticket and __threadfence() -- an L2 write-back + invalidate on
// gfx950 -- in every block: see the note at adam_multi_kernel.)
#define ELBO_BLOCKS 512
#define ELBO_MAX_LAYERS 8
struct ElboTable {
    const double* mean[ELBO_MAX_LAYERS]; const double* var[ELBO_MAX_LAYERS]; const double* raw[ELBO_MAX_LAYERS];
    double* gmean[ELBO_MAX_LAYERS]; double* gvar[ELBO_MAX_LAYERS]; double* graw[ELBO_MAX_LAYERS];
    const double* kl[ELBO_MAX_LAYERS];
    double lo[ELBO_MAX_LAYERS], hi[ELBO_MAX_LAYERS];
    int div[ELBO_MAX_LAYERS];
    int64_t rows[ELBO_MAX_LAYERS];      // base rows layer l holds: the FIRST rows[l] of the batch (<= B)
};
__global__ void elbo_all_fwd_kernel(ElboTable t, const double* y, const double* fid, int64_t B, double* part) {
    __shared__ double sh[4];
    const int l = blockIdx.y;
    double s = 0.0;
    if (t.mean[l]) {
        const double lo = t.lo[l], hi = t.hi[l], raw = t.raw[l][0];
        const double tau = hi > lo ? lo + (hi - lo) / (1.0 + exp(-raw)) : raw, ltau = log(tau);
        const int div = t.div[l];
        const int64_t n = t.rows[l] * div;
        const double level = (double)l;
        const double* mean = t.mean[l];
        const double* var = t.var[l];
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t b = i / div;
            if (fid[b] == level) s += elp_term(y[b], mean[i], var[i], tau, ltau);
        }
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) part[(int64_t)l * gridDim.x + blockIdx.x] = s;
}
// one block: the partial sums of every layer, the KL tail, the three results
__global__ void elbo_all_fwd_tail_kernel(ElboTable t, int L, int nb, int nkl, double scale, const double* part, double* out) {
    __shared__ double sh[4];
    double data = 0.0;
    for (int ll = 0; ll < L; ++ll) {
        double p = 0.0;
        for (int i = threadIdx.x; i < nb; i += blockDim.x) p += part[(int64_t)ll * nb + i];
        p = block_sum(p, sh);
        __syncthreads();
        if (t.mean[ll]) data += p / t.div[ll];
    }
    if (threadIdx.x == 0) {
        double kl = 0.0;
        for (int j = 0; j < nkl; ++j) kl += t.kl[j][0];
        out[0] = data - scale * kl;
        out[1] = scale * kl;
        out[2] = -(data - scale * kl);
    }
}
// Small problems (every layer <= 4096 rows: the reference's own sizes): the whole ELBO in ONE block and one launch -- the
// layers one after the other, then the KL tail (two launches of ~4.5 us each are more than the work itself there).
__global__ __launch_bounds__(1024) void elbo_all_fwd_one_kernel(ElboTable t, int L, int nkl, double scale, const double* y, const double* fid,
                                        double* out) {
    __shared__ double sh[16];
    double data = 0.0;      // thread 0's
    for (int l = 0; l < L; ++l) {
        double s = 0.0;
        if (t.mean[l]) {
            const double lo = t.lo[l], hi = t.hi[l], raw = t.raw[l][0];
            const double tau = hi > lo ? lo + (hi - lo) / (1.0 + exp(-raw)) : raw, ltau = log(tau);
            const int div = t.div[l];
            const int64_t n = t.rows[l] * div;
            const double level = (double)l;
            const double* mean = t.mean[l];
            const double* var = t.var[l];
            for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
                const int64_t b = i / div;
                if (fid[b] == level) s += elp_term(y[b], mean[i], var[i], tau, ltau);
            }
        }
        s = block_sum_wide(s, sh);
        if (t.mean[l]) data += s / t.div[l];
    }
    if (threadIdx.x == 0) {
        double kl = 0.0;
        for (int j = 0; j < nkl; ++j) kl += t.kl[j][0];
        out[0] = data - scale * kl;
        out[1] = scale * kl;
        out[2] = -(data - scale * kl);
    }
}
__global__ __launch_bounds__(1024) void elbo_all_bwd_one_kernel(ElboTable t, int L, double scale, const double* y, const double* fid,
                                        const double* g_elbo, const double* g_skl, double* gkl) {
    __shared__ double sh[16];
    const double ge = g_elbo ? g_elbo[0] : 0.0;
    for (int l = 0; l < L; ++l) {
        double st = 0.0;
        if (t.mean[l]) {
            const double lo = t.lo[l], hi = t.hi[l], raw = t.raw[l][0];
            const double tau = hi > lo ? lo + (hi - lo) / (1.0 + exp(-raw)) : raw;
            const int div = t.div[l];
            const int64_t n = t.rows[l] * div;
            const double level = (double)l, g = ge / div;
            const double* mean = t.mean[l];
            const double* var = t.var[l];
            double* gmean = t.gmean[l];
            double* gvar = t.gvar[l];
            for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
                const int64_t b = i / div;
                double gm = 0.0, gvv = 0.0;
                if (fid[b] == level) {
                    const double dlt = y[b] - mean[i];
                    gm = g * dlt / tau;
                    gvv = -0.5 * g / tau;
                    st += 0.5 * ((dlt * dlt + var[i]) / (tau * tau) - 1.0 / tau);
                }
                gmean[i] = gm;
                gvar[i] = gvv;
            }
        }
        st = block_sum_wide(st, sh);
        if (threadIdx.x == 0 && t.mean[l] && t.graw[l]) {
            double chain = 1.0;
            if (t.hi[l] > t.lo[l]) {
                const double sg = 1.0 / (1.0 + exp(-t.raw[l][0]));
                chain = (t.hi[l] - t.lo[l]) * sg * (1.0 - sg);
            }
            t.graw[l][0] = st / t.div[l] * ge * chain;
        }
    }
    if (threadIdx.x == 0) gkl[0] = scale * ((g_skl ? g_skl[0] : 0.0) - ge);
}
// gradients: g_mean / g_var per layer row, g_raw (the noise parameter of each layer, chain rule of the Interval transform
// included), gkl[0] = d / d(every KL) = scale * (g_skl - g_elbo)
__global__ void elbo_all_bwd_kernel(ElboTable t, const double* y, const double* fid, int64_t B, const double* g_elbo,
                                    double* part) {
    __shared__ double sh[4];
    const int l = blockIdx.y;
    const double ge = g_elbo ? g_elbo[0] : 0.0;
    double st = 0.0;
    if (t.mean[l]) {
        const double lo = t.lo[l], hi = t.hi[l], raw = t.raw[l][0];
        const double tau = hi > lo ? lo + (hi - lo) / (1.0 + exp(-raw)) : raw;
        const int div = t.div[l];
        const int64_t n = t.rows[l] * div;
        const double level = (double)l, g = ge / div;
        const double* mean = t.mean[l];
        const double* var = t.var[l];
        double* gmean = t.gmean[l];
        double* gvar = t.gvar[l];
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t b = i / div;
            double gm = 0.0, gvv = 0.0;
            if (fid[b] == level) {
                const double dlt = y[b] - mean[i];
                gm = g * dlt / tau;
                gvv = -0.5 * g / tau;
                st += 0.5 * ((dlt * dlt + var[i]) / (tau * tau) - 1.0 / tau);
            }
            gmean[i] = gm;
            gvar[i] = gvv;
        }
    }
    st = block_sum(st, sh);
    if (threadIdx.x == 0) part[(int64_t)l * gridDim.x + blockIdx.x] = st;
}
__global__ void elbo_all_bwd_tail_kernel(ElboTable t, int L, int nb, double scale, const double* g_elbo, const double* g_skl,
                                         const double* part, double* gkl) {
    __shared__ double sh[4];
    const double ge = g_elbo ? g_elbo[0] : 0.0;
    for (int ll = 0; ll < L; ++ll) {
        double p = 0.0;
        for (int i = threadIdx.x; i < nb; i += blockDim.x) p += part[(int64_t)ll * nb + i];
        p = block_sum(p, sh);
        __syncthreads();
        if (threadIdx.x == 0 && t.mean[ll] && t.graw[ll]) {
            double chain = 1.0;
            if (t.hi[ll] > t.lo[ll]) {
                const double sg = 1.0 / (1.0 + exp(-t.raw[ll][0]));
                chain = (t.hi[ll] - t.lo[ll]) * sg * (1.0 - sg);
            }
            t.graw[ll][0] = p / t.div[ll] * ge * chain;
        }
    }
    if (threadIdx.x == 0) gkl[0] = scale * ((g_skl ? g_skl[0] : 0.0) - ge);
}

__global__ void acq_fwd_kernel(const double* mu_t, const double* var_t, double* mus, double* vars, int64_t T, int S) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    double sm = 0.0, s2 = 0.0;
    for (int s = 0; s < S; ++s) {
        double m = mu_t[t * S + s];
        sm += m;
        s2 += var_t[t * S + s] + m * m;
    }
    sm /= S;
    s2 /= S;
    mus[t] = sm;
    vars[t] = s2 - sm * sm;
}
__global__ void acq_bwd_kernel(const double* mu_t, const double* gmus, const double* gvars, double* gmu_t,
                               double* gvar_t, int64_t T, int S) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    double sm = 0.0;
    for (int s = 0; s < S; ++s) sm += mu_t[t * S + s];
    sm /= S;
    const double gm = gmus ? gmus[t] : 0.0, gvv = gvars ? gvars[t] : 0.0;
    for (int s = 0; s < S; ++s) {
        double m = mu_t[t * S + s];
        gmu_t[t * S + s] = gm / S + gvv * (2.0 * m / S - 2.0 * sm / S);
        gvar_t[t * S + s] = gvv / S;
    }
}
__global__ void jes_kernel(const double* vu, const double* vc, double* acq, int64_t T) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    double v = log(vu[t]) - log(vc[t]);
    acq[t] = 0.5 * (v > 0.0 ? v : 0.0);
}
__global__ void adam_kernel(double* p, const double* g, double* m, double* v, const double* mask, int64_t n, double lr,
                            double b1, double b2, double eps, double bc1, double bc2_sqrt) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mask && mask[i] == 0.0) return;
    double gi = g[i];
    double mi = b1 * m[i] + (1.0 - b1) * gi;
    double vi = b2 * v[i] + (1.0 - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    // torch.optim.Adam: p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
    p[i] -= (lr / bc1) * mi / (sqrt(vi) / bc2_sqrt + eps);
}

// All parameter tensors of a model in ONE launch (torch's capturable Adam is seven foreach launches of ~28 us each at C3):
// blockIdx.y = tensor, the pointer table travels by value in the kernel arguments (so a captured graph replays it), the
// count of completed steps lives on the device and is advanced by a one-thread launch that follows in stream order.
#define ADAM_MAX_TENSORS 40
struct AdamTable {
    double* p[ADAM_MAX_TENSORS];
    const double* g[ADAM_MAX_TENSORS];
    double* m[ADAM_MAX_TENSORS];
    double* v[ADAM_MAX_TENSORS];
    int64_t n[ADAM_MAX_TENSORS];
};
// (The step count is advanced by a trailing one-thread launch.  Letting the last-finishing workgroup do it through a device
// ticket -- here and in the ELBO reduction below -- was tried and withdrawn: together the two hand-overs cost 0.11 ms per C3
// surrogate step and 1.05 ms per C5 step in a same-box A/B, far more than the two launches they saved.)
__global__ void adam_multi_kernel(AdamTable t, double lr, double b1, double b2, double eps, const int64_t* steps_done) {
    const int64_t step = steps_done[0] + 1;
    const int ti = blockIdx.y;
    const int64_t n = t.n[ti];
    const double bc1 = 1.0 - pow(b1, (double)step), bc2s = sqrt(1.0 - pow(b2, (double)step));
    double* p = t.p[ti];
    const double* g = t.g[ti];
    double* m = t.m[ti];
    double* v = t.v[ti];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double gi = g[i];
        const double mi = b1 * m[i] + (1.0 - b1) * gi;
        const double vi = b2 * v[i] + (1.0 - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= (lr / bc1) * mi / (sqrt(vi) / bc2s + eps);
    }
}
__global__ void adam_count_kernel(int64_t* steps_done) { steps_done[0] += 1; }

// GPyTorch's shortcut (inputs identical to the inducing inputs -> q(u) itself): marginal variances of
// S = L_S L_S^T, var[i] = max(sum_{j<=i} L[i][j]^2, min_var), one row per 64 threads; and its gradient w.r.t. L_S.
__global__ void shortcut_var_fwd_kernel(const double* L, int M, double min_var, double* var) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    double s = 0.0;
    for (int j = lane; j <= row; j += 64) {
        const double v = L[(int64_t)row * M + j];
        s += v * v;
    }
    s = wave_sum(s);
    if (lane == 0) var[row] = s > min_var ? s : min_var;
}
__global__ void shortcut_var_bwd_kernel(const double* L, const double* var, const double* g, int M, double min_var,
                                        double* gL) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * M) return;
    const int i = (int)(idx / M), j = (int)(idx % M);
    gL[idx] = (j <= i && var[i] > min_var) ? 2.0 * L[idx] * g[i] : 0.0;
}

// ELBO tail: out[0] = sum_i a_i - scale * sum_j b_j,  out[1] = scale * sum_j b_j  (data terms a, layer KLs b): one launch
// instead of a chain of scalar adds / muls / subs, each of which is a ~4 us launch (and as many again in backward).
#define COMBINE_MAX 8
struct CombineTable { const double* a[COMBINE_MAX]; const double* b[COMBINE_MAX]; };
__global__ void elbo_combine_fwd_kernel(CombineTable t, int na, int nb, double scale, double* out) {
    double sa = 0.0, sb = 0.0;
    for (int i = 0; i < na; ++i) sa += t.a[i][0];
    for (int j = 0; j < nb; ++j) sb += t.b[j][0];
    out[0] = sa - scale * sb;
    out[1] = scale * sb;
}
// g[0] = d/da_i = g_elbo ;  g[1] = d/db_j = scale * (g_skl - g_elbo)     (either upstream gradient may be NULL = 0)
__global__ void elbo_combine_bwd_kernel(const double* g_elbo, const double* g_skl, double scale, double* g) {
    const double ge = g_elbo ? g_elbo[0] : 0.0, gs = g_skl ? g_skl[0] : 0.0;
    g[0] = ge;
    g[1] = scale * (gs - ge);
}

extern "C" {

int mobocmf_shortcut_var_forward(const double* L_S, int32_t M, double min_var, double* var, mobocmf_stream_t stream) {
    if (!L_S || !var || M < 1) return MOBOCMF_BAD_ARG;
    hipLaunchKernelGGL(shortcut_var_fwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, L_S, M,
                       min_var, var);
    return CHECK_LAUNCH();
}

int mobocmf_shortcut_var_backward(const double* L_S, const double* var, const double* g_var, int32_t M, double min_var,
                                  double* g_LS, mobocmf_stream_t stream) {
    if (!L_S || !var || !g_var || !g_LS || M < 1) return MOBOCMF_BAD_ARG;
    hipLaunchKernelGGL(shortcut_var_bwd_kernel, GRID1((int64_t)M * M), 0, (hipStream_t)stream, L_S, var, g_var, M, min_var,
                       g_LS);
    return CHECK_LAUNCH();
}

int mobocmf_elbo_combine_forward(int32_t n_data, const double* const* data_terms, int32_t n_kl, const double* const* kls,
                                 double scale, double* out2, mobocmf_stream_t stream) {
    if (n_data < 0 || n_kl < 0 || n_data > COMBINE_MAX || n_kl > COMBINE_MAX || !out2 || (n_data && !data_terms) ||
        (n_kl && !kls))
        return MOBOCMF_BAD_ARG;
    CombineTable t = {};
    for (int i = 0; i < n_data; ++i) { if (!data_terms[i]) return MOBOCMF_BAD_ARG; t.a[i] = data_terms[i]; }
    for (int j = 0; j < n_kl; ++j) { if (!kls[j]) return MOBOCMF_BAD_ARG; t.b[j] = kls[j]; }
    hipLaunchKernelGGL(elbo_combine_fwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, t, n_data, n_kl, scale, out2);
    return CHECK_LAUNCH();
}

int mobocmf_elbo_combine_backward(const double* g_elbo, const double* g_skl, double scale, double* g2,
                                  mobocmf_stream_t stream) {
    if (!g2) return MOBOCMF_BAD_ARG;
    hipLaunchKernelGGL(elbo_combine_bwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, g_elbo, g_skl, scale, g2);
    return CHECK_LAUNCH();
}

int mobocmf_adam_multi(int32_t n_tensors, double* const* params, const double* const* grads, double* const* exp_avg,
                       double* const* exp_avg_sq, const int64_t* sizes, double lr, double beta1, double beta2, double eps,
                       int64_t* step_state, mobocmf_stream_t stream) {
    if (n_tensors < 0 || !step_state || (n_tensors && (!params || !grads || !exp_avg || !exp_avg_sq || !sizes)))
        return MOBOCMF_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < n_tensors; base += ADAM_MAX_TENSORS) {
        AdamTable t;
        int cnt = n_tensors - base < ADAM_MAX_TENSORS ? n_tensors - base : ADAM_MAX_TENSORS;
        int64_t nmax = 0;
        for (int i = 0; i < cnt; ++i) {
            t.p[i] = params[base + i]; t.g[i] = grads[base + i]; t.m[i] = exp_avg[base + i]; t.v[i] = exp_avg_sq[base + i];
            t.n[i] = sizes[base + i];
            if (t.n[i] < 0 || !t.p[i] || !t.g[i] || !t.m[i] || !t.v[i]) return MOBOCMF_BAD_ARG;
            if (t.n[i] > nmax) nmax = t.n[i];
        }
        int64_t bx = (nmax + 1023) / 1024;       // 4 elements per thread in the grid-stride loop
        if (bx < 1) bx = 1;
        if (bx > 4096) bx = 4096;
        hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)bx, (unsigned)cnt), dim3(256), 0, s, t, lr, beta1, beta2, eps,
                           (const int64_t*)step_state);
    }
    hipLaunchKernelGGL(adam_count_kernel, dim3(1), dim3(1), 0, s, step_state);
    return CHECK_LAUNCH();
}

// Constrained hyper-parameters of a layer: softplus of every raw parameter tensor, packed into one vector in the C-ABI
// order (and the backward of that), one launch each.  The tensors are tiny (<= 5 + 2 d entries in total): one workgroup.
#define PACK_MAX_TENSORS 16
struct PackTable {
    const double* raw[PACK_MAX_TENSORS];
    double* g_raw[PACK_MAX_TENSORS];
    const double* g_src[PACK_MAX_TENSORS];      // _backward_v: upstream gradient of each tensor's packed slice (null: zero)
    int32_t off[PACK_MAX_TENSORS + 1];
};
__global__ void softplus_pack_kernel(PackTable t, int nt, double* out) {
    for (int e = threadIdx.x; e < t.off[nt]; e += blockDim.x) {
        int ti = 0;
        while (e >= t.off[ti + 1]) ++ti;
        const double x = t.raw[ti][e - t.off[ti]];
        out[e] = x > 20.0 ? x : log1p(exp(x));      // torch.nn.functional.softplus (beta 1, threshold 20)
    }
}
__global__ void softplus_pack_bwd_kernel(PackTable t, int nt, const double* g_out) {
    for (int e = threadIdx.x; e < t.off[nt]; e += blockDim.x) {
        int ti = 0;
        while (e >= t.off[ti + 1]) ++ti;
        const double x = t.raw[ti][e - t.off[ti]];
        t.g_raw[ti][e - t.off[ti]] = x > 20.0 ? g_out[e] : g_out[e] / (1.0 + exp(-x));
    }
}
// the same with one upstream-gradient pointer PER TENSOR (segments of the packed vector handed out as separate tensors)
__global__ void softplus_pack_bwd_v_kernel(PackTable t, int nt) {
    for (int e = threadIdx.x; e < t.off[nt]; e += blockDim.x) {
        int ti = 0;
        while (e >= t.off[ti + 1]) ++ti;
        const int k = e - t.off[ti];
        const double x = t.raw[ti][k];
        const double g = t.g_src[ti] ? t.g_src[ti][k] : 0.0;
        t.g_raw[ti][k] = x > 20.0 ? g : g / (1.0 + exp(-x));
    }
}
static int fill_pack_table(PackTable& t, int32_t n, const double* const* raw, double* const* g_raw, const int32_t* sizes) {
    if (n < 1 || n > PACK_MAX_TENSORS || !raw || !sizes) return MOBOCMF_BAD_ARG;
    t.off[0] = 0;
    for (int i = 0; i < n; ++i) {
        if (!raw[i] || sizes[i] < 1 || (g_raw && !g_raw[i])) return MOBOCMF_BAD_ARG;
        t.raw[i] = raw[i];
        t.g_raw[i] = g_raw ? g_raw[i] : nullptr;
        t.off[i + 1] = t.off[i] + sizes[i];
    }
    return MOBOCMF_OK;
}
int mobocmf_softplus_pack(int32_t n_tensors, const double* const* raw, const int32_t* sizes, double* out,
                          mobocmf_stream_t stream) {
    PackTable t;
    if (!out) return MOBOCMF_BAD_ARG;
    const int rc = fill_pack_table(t, n_tensors, raw, nullptr, sizes);
    if (rc) return rc;
    hipLaunchKernelGGL(softplus_pack_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, t, n_tensors, out);
    return CHECK_LAUNCH();
}
int mobocmf_softplus_pack_backward(int32_t n_tensors, const double* const* raw, const int32_t* sizes, const double* g_out,
                                   double* const* g_raw, mobocmf_stream_t stream) {
    PackTable t;
    if (!g_out || !g_raw) return MOBOCMF_BAD_ARG;
    const int rc = fill_pack_table(t, n_tensors, raw, g_raw, sizes);
    if (rc) return rc;
    hipLaunchKernelGGL(softplus_pack_bwd_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, t, n_tensors, g_out);
    return CHECK_LAUNCH();
}

int mobocmf_softplus_pack_backward_v(int32_t n_tensors, const double* const* raw, const int32_t* sizes,
                                     const double* const* g_out, double* const* g_raw, mobocmf_stream_t stream) {
    PackTable t;
    if (!g_out || !g_raw) return MOBOCMF_BAD_ARG;
    const int rc = fill_pack_table(t, n_tensors, raw, g_raw, sizes);
    if (rc) return rc;
    for (int i = 0; i < n_tensors; ++i) t.g_src[i] = g_out[i];
    hipLaunchKernelGGL(softplus_pack_bwd_v_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, t, n_tensors);
    return CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------------------------------
// Conditioned training (blackbox_mfdgp_fitter.py:227-243): the theta / omega factor losses and the glue
// around them.  The arithmetic is a few hundred flops; as framework ops it was ~100 element-wise launches per iteration
// (cdf, products, sums, the backward of every slice / stack / scalar subtraction), i.e. a third of an iteration that is
// bound by its launch count.  One block each.
// ---------------------------------------------------------------------------------------------------------------------
#define FACT_MAX 8
struct FactTable {
    const double* fm[FACT_MAX]; const double* fv[FACT_MAX]; const double* cm[FACT_MAX]; const double* cv[FACT_MAX];
    double* gfm[FACT_MAX]; double* gfv[FACT_MAX]; double* gcm[FACT_MAX]; double* gcv[FACT_MAX];
};
__device__ __forceinline__ double ncdf(double z) { return 0.5 * (1.0 + erf(z * 0.7071067811865476)); }
__device__ __forceinline__ double npdf(double z) { return 0.3989422804014327 * exp(-0.5 * z * z); }
// loss = sum_{p < P, t < T} [ coef_c c(p,t) + coef_1mc (1 - c(p,t)) ],
//   c(p,t) = prod_k Phi((cm_k[t] - thr[k]) / sqrt(cv_k[t])) * prod_j Phi((front[p][j] - fm_j[t]) / sqrt(fv_j[t]));
// the gradients w.r.t. every fm_j[t], fv_j[t], cm_k[t], cv_k[t] (for an upstream gradient of 1) come out of the same pass.
// Threads are spread over (t, p): NG = 256 / min(T, 256) groups share a point t and split the Pareto points p among them (the
// omega factors are 10 points x 50 Pareto points x n_obj erf / exp pairs: one thread per t walked all 50 serially, 80 us --
// longer than the whole one-launch step that feeds it); the groups' partial sums over p meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void cond_factors_kernel(FactTable tb, int n_obj, int n_con, int P, int T, const double* front,
                                                           const double* thr, double coef_c, double coef_1mc, double* loss) {
    __shared__ double sh[4];
    __shared__ double part[256][2 * FACT_MAX + 1];      // per (group, t): sum_p O, gm[j], gv[j]
    double acc = 0.0;
    const double dldc = coef_c - coef_1mc;
    const int Tt = T < 256 ? T : 256;                   // points of a tile
    int NG = 256 / Tt;
    if (NG > P) NG = P;
    const int tl = threadIdx.x % Tt, grp = threadIdx.x / Tt;
    for (int t0 = 0; t0 < T; t0 += Tt) {
        const int t = t0 + tl;
        const bool on = grp < NG && t < T;
        double isd[FACT_MAX], gm[FACT_MAX], gv[FACT_MAX], osum = 0.0;
        if (on) {
            for (int j = 0; j < n_obj; ++j) { isd[j] = 1.0 / sqrt(tb.fv[j][t]); gm[j] = 0.0; gv[j] = 0.0; }
            for (int p = grp; p < P; p += NG) {
                double ph[FACT_MAX], u[FACT_MAX], O = 1.0;
                for (int j = 0; j < n_obj; ++j) {
                    u[j] = (front[(int64_t)p * n_obj + j] - tb.fm[j][t]) * isd[j];
                    ph[j] = ncdf(u[j]);
                    O *= ph[j];
                }
                osum += O;
                for (int j = 0; j < n_obj; ++j) {
                    double rest = 1.0;      // O / Phi(u_j), formed without the division
                    for (int q = 0; q < n_obj; ++q) rest *= q == j ? 1.0 : ph[q];
                    const double pd = npdf(u[j]);
                    gm[j] += rest * pd * (-isd[j]);
                    gv[j] += rest * pd * (-0.5 * u[j] / tb.fv[j][t]);
                }
            }
            double* pp = part[grp * Tt + tl];
            pp[0] = osum;
            for (int j = 0; j < n_obj; ++j) { pp[1 + j] = gm[j]; pp[1 + FACT_MAX + j] = gv[j]; }
        }
        __syncthreads();
        if (on && grp == 0) {
            for (int g2 = 1; g2 < NG; ++g2) {
                const double* pp = part[g2 * Tt + tl];
                osum += pp[0];
                for (int j = 0; j < n_obj; ++j) { gm[j] += pp[1 + j]; gv[j] += pp[1 + FACT_MAX + j]; }
            }
            double phic[FACT_MAX], dzm[FACT_MAX], dzv[FACT_MAX];      // Phi(z_k), dPhi/dcm_k, dPhi/dcv_k
            double C = 1.0;
            for (int k = 0; k < n_con; ++k) {
                const double sd = sqrt(tb.cv[k][t]), z = (tb.cm[k][t] - thr[k]) / sd, pd = npdf(z);
                phic[k] = ncdf(z);
                dzm[k] = pd / sd;
                dzv[k] = -0.5 * pd * z / tb.cv[k][t];
                C *= phic[k];
            }
            acc += dldc * C * osum + coef_1mc * (double)P;
            for (int j = 0; j < n_obj; ++j) { tb.gfm[j][t] = dldc * C * gm[j]; tb.gfv[j][t] = dldc * C * gv[j]; }
            for (int k = 0; k < n_con; ++k) {
                double rest = osum;
                for (int q = 0; q < n_con; ++q) rest *= q == k ? 1.0 : phic[q];
                tb.gcm[k][t] = dldc * rest * dzm[k];
                tb.gcv[k][t] = dldc * rest * dzv[k];
            }
        }
        __syncthreads();
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) loss[0] = acc;
}
int mobocmf_cond_factors_forward(int32_t n_obj, int32_t n_con, int32_t P, int32_t T, const double* const* fs_mean,
                                 const double* const* fs_var, const double* const* cs_mean, const double* const* cs_var,
                                 const double* pareto_front, const double* thresholds, double coef_c, double coef_1mc,
                                 double* loss, double* const* g_fs_mean, double* const* g_fs_var, double* const* g_cs_mean,
                                 double* const* g_cs_var, mobocmf_stream_t stream) {
    if (n_obj < 0 || n_obj > FACT_MAX || n_con < 0 || n_con > FACT_MAX || n_obj + n_con < 1 || P < 1 || T < 1 || !loss ||
        (n_obj && (!fs_mean || !fs_var || !g_fs_mean || !g_fs_var || !pareto_front)) ||
        (n_con && (!cs_mean || !cs_var || !g_cs_mean || !g_cs_var || !thresholds)))
        return MOBOCMF_BAD_ARG;
    FactTable tb = {};
    for (int j = 0; j < n_obj; ++j) {
        if (!fs_mean[j] || !fs_var[j] || !g_fs_mean[j] || !g_fs_var[j]) return MOBOCMF_BAD_ARG;
        tb.fm[j] = fs_mean[j]; tb.fv[j] = fs_var[j]; tb.gfm[j] = g_fs_mean[j]; tb.gfv[j] = g_fs_var[j];
    }
    for (int k = 0; k < n_con; ++k) {
        if (!cs_mean[k] || !cs_var[k] || !g_cs_mean[k] || !g_cs_var[k]) return MOBOCMF_BAD_ARG;
        tb.cm[k] = cs_mean[k]; tb.cv[k] = cs_var[k]; tb.gcm[k] = g_cs_mean[k]; tb.gcv[k] = g_cs_var[k];
    }
    hipLaunchKernelGGL(cond_factors_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tb, n_obj, n_con, P, T, pareto_front,
                       thresholds, coef_c, coef_1mc, loss);
    return CHECK_LAUNCH();
}

// Segment glue, one launch each:  scale: out_i = g[0] * in_i (the backward of anything whose forward already formed its own
// gradient for an upstream 1);  gather: out = the segments back to back, zeros where a segment is absent (the backward of a
// split into row ranges);  combine: out[0] = sum_i coef_i * x_i[0] (a loss assembled from scalar terms).
#define SEG_MAX 32
struct SegTable { const double* in[SEG_MAX]; double* out[SEG_MAX]; int64_t off[SEG_MAX + 1]; double coef[SEG_MAX]; };
__global__ void seg_scale_kernel(SegTable t, int n, const double* g) {
    const double gg = g ? g[0] : 1.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < t.off[n]; e += (int64_t)gridDim.x * blockDim.x) {
        int i = 0;
        while (e >= t.off[i + 1]) ++i;
        t.out[i][e - t.off[i]] = gg * t.coef[i] * t.in[i][e - t.off[i]];
    }
}
__global__ void seg_gather_kernel(SegTable t, int n, double* out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < t.off[n]; e += (int64_t)gridDim.x * blockDim.x) {
        int i = 0;
        while (e >= t.off[i + 1]) ++i;
        out[e] = t.in[i] ? t.in[i][e - t.off[i]] : 0.0;
    }
}
__global__ void scalar_combine_kernel(SegTable t, int n, double* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double v = 0.0;
        for (int i = 0; i < n; ++i) v += t.coef[i] * t.in[i][0];
        out[0] = v;
    }
}
static int fill_seg(SegTable& t, int32_t n, const double* const* in, double* const* out, const int64_t* sizes, const double* coef,
                    bool in_may_be_null) {
    if (n < 1 || n > SEG_MAX || !in || !sizes) return MOBOCMF_BAD_ARG;
    t.off[0] = 0;
    for (int i = 0; i < n; ++i) {
        if (sizes[i] < 0 || (!in[i] && !in_may_be_null && sizes[i] > 0) || (out && !out[i] && sizes[i] > 0)) return MOBOCMF_BAD_ARG;
        t.in[i] = in[i];
        t.out[i] = out ? out[i] : nullptr;
        t.off[i + 1] = t.off[i] + sizes[i];
        t.coef[i] = coef ? coef[i] : 1.0;
    }
    return MOBOCMF_OK;
}
static unsigned seg_blocks(int64_t total) {
    const int64_t nb = (total + 255) / 256;
    return (unsigned)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
}
int mobocmf_scale_segments(int32_t n, const double* const* in, double* const* out, const int64_t* sizes, const double* coef,
                           const double* g, mobocmf_stream_t stream) {
    SegTable t;
    if (!out) return MOBOCMF_BAD_ARG;
    const int rc = fill_seg(t, n, in, out, sizes, coef, false);
    if (rc) return rc;
    if (t.off[n] == 0) return MOBOCMF_OK;
    hipLaunchKernelGGL(seg_scale_kernel, dim3(seg_blocks(t.off[n])), dim3(256), 0, (hipStream_t)stream, t, n, g);
    return CHECK_LAUNCH();
}
int mobocmf_gather_segments(int32_t n, const double* const* in, const int64_t* sizes, double* out, mobocmf_stream_t stream) {
    SegTable t;
    if (!out) return MOBOCMF_BAD_ARG;
    const int rc = fill_seg(t, n, in, nullptr, sizes, nullptr, true);
    if (rc) return rc;
    if (t.off[n] == 0) return MOBOCMF_OK;
    hipLaunchKernelGGL(seg_gather_kernel, dim3(seg_blocks(t.off[n])), dim3(256), 0, (hipStream_t)stream, t, n, out);
    return CHECK_LAUNCH();
}
int mobocmf_scalar_combine(int32_t n, const double* const* x, const double* coef, double* out, mobocmf_stream_t stream) {
    SegTable t;
    if (!out || !coef) return MOBOCMF_BAD_ARG;
    int64_t ones[SEG_MAX];
    for (int i = 0; i < SEG_MAX; ++i) ones[i] = 1;
    const int rc = fill_seg(t, n, x, nullptr, ones, coef, false);
    if (rc) return rc;
    hipLaunchKernelGGL(scalar_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, n, out);
    return CHECK_LAUNCH();
}

int mobocmf_propagate_forward(const double* mean, const double* var, const double* eps, double* f_out, int64_t n_out,
                              int32_t div, mobocmf_stream_t stream) {
    if (n_out < 0 || div < 1) return MOBOCMF_BAD_ARG;
    if (n_out == 0) return MOBOCMF_OK;
    hipLaunchKernelGGL(propagate_fwd_kernel, GRID1(n_out), 0, (hipStream_t)stream, mean, var, eps, f_out, n_out, div);
    return CHECK_LAUNCH();
}

int mobocmf_propagate_rng_forward(const double* mean, const double* var, int64_t* rng_state, double* f_out, double* eps_out,
                                  int64_t n_out, int32_t div, mobocmf_stream_t stream) {
    if (n_out < 1 || div < 1 || !mean || !var || !rng_state || !f_out || !eps_out) return MOBOCMF_BAD_ARG;
    hipLaunchKernelGGL(propagate_rng_fwd_kernel, GRID1(n_out), 0, (hipStream_t)stream, mean, var, rng_state, f_out, eps_out,
                       n_out, div);
    return CHECK_LAUNCH();
}

int mobocmf_propagate_backward(const double* var, const double* eps, const double* g_f, double* g_mean, double* g_var,
                               int64_t n_out, int32_t div, mobocmf_stream_t stream) {
    return mobocmf_propagate_backward_prefix(var, eps, g_f, g_mean, g_var, n_out, div, div > 0 ? n_out / div : 0, nullptr, nullptr,
                                             stream);
}

int mobocmf_propagate_backward_prefix(const double* var, const double* eps, const double* g_f, double* g_mean, double* g_var,
                                      int64_t n_out, int32_t div, int64_t n_prev, const double* add_mean, const double* add_var,
                                      mobocmf_stream_t stream) {
    if (n_out < 0 || div < 1 || n_out % div || n_prev < n_out / div) return MOBOCMF_BAD_ARG;
    if (n_prev == 0) return MOBOCMF_OK;
    hipLaunchKernelGGL(propagate_bwd_kernel, GRID1(n_prev), 0, (hipStream_t)stream, var, eps, g_f, g_mean, g_var,
                       n_out / div, div, n_prev, add_mean, add_var);
    return CHECK_LAUNCH();
}

int mobocmf_elbo_data_forward(const double* mean, const double* var, const double* y, const double* fid,
                              const double* tau, double level, int64_t n_rows, int32_t div, double* out, void* scratch,
                              size_t scratch_bytes, mobocmf_stream_t stream) {
    return mobocmf_elbo_data_interval_forward(mean, var, y, fid, tau, 0.0, 0.0, level, n_rows, div, out, scratch,
                                              scratch_bytes, stream);
}

int mobocmf_elbo_data_interval_forward(const double* mean, const double* var, const double* y, const double* fid,
                                       const double* raw_noise, double lo, double hi, double level, int64_t n_rows,
                                       int32_t div, double* out, void* scratch, size_t scratch_bytes,
                                       mobocmf_stream_t stream) {
    const double* tau = raw_noise;
    if (n_rows < 0 || div < 1) return MOBOCMF_BAD_ARG;
    if (scratch_bytes < ELBO_BLOCKS * sizeof(double)) return MOBOCMF_WORKSPACE_TOO_SMALL;
    int nb = (int)((n_rows + 255) / 256);
    nb = nb < 1 ? 1 : (nb > ELBO_BLOCKS ? ELBO_BLOCKS : nb);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(elbo_fwd_kernel, dim3(nb), dim3(256), 0, s, mean, var, y, fid, tau, lo, hi, level, n_rows, div,
                       (double*)scratch);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, (const double*)scratch, nb, 1.0 / div, out);
    return CHECK_LAUNCH();
}

int mobocmf_elbo_data_backward(const double* mean, const double* var, const double* y, const double* fid,
                               const double* tau, double level, int64_t n_rows, int32_t div, const double* g_out,
                               double* g_mean, double* g_var, double* g_tau, void* scratch, size_t scratch_bytes,
                               mobocmf_stream_t stream) {
    return mobocmf_elbo_data_interval_backward(mean, var, y, fid, tau, 0.0, 0.0, level, n_rows, div, g_out, g_mean, g_var,
                                               g_tau, scratch, scratch_bytes, stream);
}

int mobocmf_elbo_data_interval_backward(const double* mean, const double* var, const double* y, const double* fid,
                                        const double* raw_noise, double lo, double hi, double level, int64_t n_rows,
                                        int32_t div, const double* g_out, double* g_mean, double* g_var, double* g_tau,
                                        void* scratch, size_t scratch_bytes, mobocmf_stream_t stream) {
    const double* tau = raw_noise;
    if (n_rows < 0 || div < 1) return MOBOCMF_BAD_ARG;
    if (scratch_bytes < ELBO_BLOCKS * sizeof(double)) return MOBOCMF_WORKSPACE_TOO_SMALL;
    int nb = (int)((n_rows + 255) / 256);
    nb = nb < 1 ? 1 : (nb > ELBO_BLOCKS ? ELBO_BLOCKS : nb);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(elbo_bwd_kernel, dim3(nb), dim3(256), 0, s, mean, var, y, fid, tau, lo, hi, level, n_rows, div,
                       g_out, g_mean, g_var, (double*)scratch);
    hipLaunchKernelGGL(final_sum_scaled_kernel, dim3(1), dim3(256), 0, s, (const double*)scratch, nb, g_out, 1.0 / div,
                       g_tau, tau, lo, hi);
    return CHECK_LAUNCH();
}

static int elbo_blocks(int32_t L, const double* const* mean, const int32_t* div, int64_t B) {
    int64_t nmax = 1;
    for (int l = 0; l < L; ++l)
        if (mean[l] && B * div[l] > nmax) nmax = B * div[l];
    int64_t nb = (nmax + 1023) / 1024;
    return (int)(nb < 1 ? 1 : (nb > ELBO_BLOCKS ? ELBO_BLOCKS : nb));
}

int mobocmf_elbo_forward(int32_t L, const double* const* mean, const double* const* var, const int32_t* div,
                         const double* const* raw_noise, const double* lo, const double* hi, const double* y,
                         const double* fid, int64_t B, const int64_t* rows, int32_t n_kl, const double* const* kls, double scale,
                         double* out3, void* scratch, size_t scratch_bytes, mobocmf_stream_t stream) {
    if (L < 1 || L > ELBO_MAX_LAYERS || n_kl < 0 || n_kl > ELBO_MAX_LAYERS || !mean || !var || !div || !raw_noise || !lo ||
        !hi || !y || !fid || B < 0 || !out3 || !scratch || (n_kl && !kls))
        return MOBOCMF_BAD_ARG;
    if (scratch_bytes < (size_t)ELBO_MAX_LAYERS * ELBO_BLOCKS * sizeof(double)) return MOBOCMF_WORKSPACE_TOO_SMALL;
    ElboTable t = {};
    for (int l = 0; l < L; ++l) {
        if (mean[l] && (!var[l] || !raw_noise[l] || div[l] < 1)) return MOBOCMF_BAD_ARG;
        if (rows && (rows[l] < 0 || rows[l] > B)) return MOBOCMF_BAD_ARG;
        t.mean[l] = mean[l]; t.var[l] = var[l]; t.raw[l] = raw_noise[l]; t.lo[l] = lo[l]; t.hi[l] = hi[l]; t.div[l] = div[l];
        t.rows[l] = rows ? rows[l] : B;
    }
    for (int j = 0; j < n_kl; ++j) { if (!kls[j]) return MOBOCMF_BAD_ARG; t.kl[j] = kls[j]; }
    const int nb = elbo_blocks(L, mean, div, B);
    if (nb <= 4) {      // small: one block, one launch
        hipLaunchKernelGGL(elbo_all_fwd_one_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, t, L, n_kl, scale, y, fid, out3);
        return CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(elbo_all_fwd_kernel, dim3(nb, L), dim3(256), 0, (hipStream_t)stream, t, y, fid, B, (double*)scratch);
    hipLaunchKernelGGL(elbo_all_fwd_tail_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, t, L, nb, n_kl, scale,
                       (const double*)scratch, out3);
    return CHECK_LAUNCH();
}

int mobocmf_elbo_backward(int32_t L, const double* const* mean, const double* const* var, const int32_t* div,
                          const double* const* raw_noise, const double* lo, const double* hi, const double* y,
                          const double* fid, int64_t B, const int64_t* rows, double scale, const double* g_elbo,
                          const double* g_skl, double* const* g_mean, double* const* g_var, double* const* g_raw_noise,
                          double* g_kl, void* scratch, size_t scratch_bytes, mobocmf_stream_t stream) {
    if (L < 1 || L > ELBO_MAX_LAYERS || !mean || !var || !div || !raw_noise || !lo || !hi || !y || !fid || B < 0 ||
        !g_mean || !g_var || !g_raw_noise || !g_kl || !scratch)
        return MOBOCMF_BAD_ARG;
    if (scratch_bytes < (size_t)ELBO_MAX_LAYERS * ELBO_BLOCKS * sizeof(double)) return MOBOCMF_WORKSPACE_TOO_SMALL;
    ElboTable t = {};
    for (int l = 0; l < L; ++l) {
        if (mean[l] && (!var[l] || !raw_noise[l] || div[l] < 1 || !g_mean[l] || !g_var[l])) return MOBOCMF_BAD_ARG;
        if (rows && (rows[l] < 0 || rows[l] > B)) return MOBOCMF_BAD_ARG;
        t.mean[l] = mean[l]; t.var[l] = var[l]; t.raw[l] = raw_noise[l]; t.lo[l] = lo[l]; t.hi[l] = hi[l]; t.div[l] = div[l];
        t.rows[l] = rows ? rows[l] : B;
        t.gmean[l] = g_mean[l]; t.gvar[l] = g_var[l]; t.graw[l] = g_raw_noise[l];
    }
    const int nb = elbo_blocks(L, mean, div, B);
    if (nb <= 4) {
        hipLaunchKernelGGL(elbo_all_bwd_one_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, t, L, scale, y, fid, g_elbo, g_skl,
                           g_kl);
        return CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(elbo_all_bwd_kernel, dim3(nb, L), dim3(256), 0, (hipStream_t)stream, t, y, fid, B, g_elbo,
                       (double*)scratch);
    hipLaunchKernelGGL(elbo_all_bwd_tail_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, t, L, nb, scale, g_elbo, g_skl,
                       (const double*)scratch, g_kl);
    return CHECK_LAUNCH();
}

int mobocmf_acq_moments_forward(const double* mu_t, const double* var_t, double* mus, double* vars, int64_t T, int32_t S,
                                mobocmf_stream_t stream) {
    if (T < 0 || S < 1) return MOBOCMF_BAD_ARG;
    if (T == 0) return MOBOCMF_OK;
    hipLaunchKernelGGL(acq_fwd_kernel, GRID1(T), 0, (hipStream_t)stream, mu_t, var_t, mus, vars, T, S);
    return CHECK_LAUNCH();
}

int mobocmf_acq_moments_backward(const double* mu_t, const double* g_mus, const double* g_vars, double* g_mu_t,
                                 double* g_var_t, int64_t T, int32_t S, mobocmf_stream_t stream) {
    if (T < 0 || S < 1) return MOBOCMF_BAD_ARG;
    if (T == 0) return MOBOCMF_OK;
    hipLaunchKernelGGL(acq_bwd_kernel, GRID1(T), 0, (hipStream_t)stream, mu_t, g_mus, g_vars, g_mu_t, g_var_t, T, S);
    return CHECK_LAUNCH();
}

int mobocmf_jes_forward(const double* v_uncond, const double* v_cond, double* acq, int64_t T, mobocmf_stream_t stream) {
    if (T < 0) return MOBOCMF_BAD_ARG;
    if (T == 0) return MOBOCMF_OK;
    hipLaunchKernelGGL(jes_kernel, GRID1(T), 0, (hipStream_t)stream, v_uncond, v_cond, acq, T);
    return CHECK_LAUNCH();
}

int mobocmf_adam_step(double* param, const double* grad, double* exp_avg, double* exp_avg_sq, const double* mask,
                      int64_t n, double lr, double beta1, double beta2, double eps, int64_t step,
                      mobocmf_stream_t stream) {
    if (n < 0 || step < 1) return MOBOCMF_BAD_ARG;
    if (n == 0) return MOBOCMF_OK;
    double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, GRID1(n), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, mask, n, lr, beta1,
                       beta2, eps, bc1, sqrt(bc2));
    return CHECK_LAUNCH();
}

}  // extern "C"
