// FP64 MFMA GEMM for gfx950 (v_mfma_f64_16x16x4_f64), the dominant kernel of the MFDGP layer.
//
//   C[Mr x Nc] (+)= alpha * A[Mr x Kd] * B        B_T=0: B is [Kd x Nc] (n contiguous)
//                                                 B_T=1: B is [Nc x Kd] (k contiguous)  -> A * B^T
//
// Workgroup = 256 threads = 4 wavefronts (2 x 2), tile 128 x 128, K step 16, register-prefetched
// double-buffered LDS (2 x 36 KiB -> 2 workgroups per CU).  Each wavefront owns a 64 x 64 block =
// 4 x 4 MFMA tiles (16 accumulators x 4 f64 = 128 VGPRs).  Operand fragments for the 16x16x4 f64 MFMA:
// lane l holds A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; accumulator register r of lane l is
// C[row = (l>>4) + 4r][col = l&15].
// LDS images are padded so every ds_read_b64 of a fragment is bank-conflict free:
//   A / B_T tiles [128][16+2]  (row stride 36 dwords = 4*odd mod 64)
//   B tiles       [16][128+16] (row stride 288 dwords = 32 mod 64)
// Triangular operands (A = L^-1 lower, or upper) only visit the non-zero k range of their row block.
// blockIdx.x -> tile mapping is XCD-aware: blocks b, b+8, ... share an XCD (and its L2), so the row
// blocks that re-read the same 128-column panel of B are dealt to the same XCD back to back.
#include "common.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

#define BM 128
#define BN 128
#define BK 16
#define LDA_S (BK + 2)
#define LDB_S (BN + 16)
#define AS_ELEMS (BM * LDA_S)                                   // 2304 doubles
#define BS_ELEMS ((BK * LDB_S) > (BN * LDA_S) ? (BK * LDB_S) : (BN * LDA_S))  // 2304 doubles

template <bool B_T>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g, int nrb, int64_t ncb, int splitk) {
    __shared__ __attribute__((aligned(16))) double lds[2 * (AS_ELEMS + BS_ELEMS)];
    double* As0 = lds;
    double* Bs0 = lds + 2 * AS_ELEMS;

    // ---- XCD-aware tile mapping: id -> (xcd, slot); slot -> (local column block, row block)
    int64_t id = blockIdx.x;
    int64_t ntile = (int64_t)nrb * ncb;
    int rb;
    int64_t cb;
    if ((ncb & 7) == 0) {
        int64_t xcd = id & 7, slot = id >> 3;
        cb = (slot / nrb) * 8 + xcd;
        rb = nrb - 1 - (int)(slot % nrb);   // long (triangular) row blocks first
    } else {
        cb = id / nrb;
        rb = nrb - 1 - (int)(id % nrb);
    }
    (void)ntile;
    if (g.lower_out && cb > rb) return;

    const int z = blockIdx.z;
    const double* A = g.A;
    const double* B = g.B;
    double* C = g.C;
    int64_t k0 = 0, k1 = g.Kd;
    if (g.batched) {
        A += z * g.strideA;
        B += z * g.strideB;
        C += z * g.strideC;
    } else if (splitk > 1) {
        int64_t nk = g.Kd / BK;
        int64_t per = (nk + splitk - 1) / splitk;
        k0 = z * per * BK;
        k1 = k0 + per * BK;
        if (k1 > g.Kd) k1 = g.Kd;
        C += z * g.slab_stride;
    }
    if (g.tri & TRI_LOWER_A) {
        int64_t e = (int64_t)(rb + 1) * BM;
        if (k1 > e) k1 = e;
    }
    if (g.tri & TRI_UPPER_A) {
        int64_t b = (int64_t)rb * BM;
        if (k0 < b) k0 = b;
    }
    if (g.tri & TRI_LOWER_B) {
        int64_t b = cb * BN;
        if (k0 < b) k0 = b;
    }
    if (g.tri & TRI_UPPER_B) {
        int64_t e = (cb + 1) * BN;
        if (k1 > e) k1 = e;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 15, lk = lane >> 4;

    // global -> register staging maps
    const int a_row = tid >> 3, a_chk = tid & 7;        // A (and B_T): 4 x (32 rows apart), 16 B chunk of the 128 B row
    const int b_row = tid >> 6, b_chk = tid & 63;       // B: 4 x (4 rows apart), 16 B chunk of the 1 KiB row
    const double* Ag = A + ((int64_t)rb * BM + a_row) * g.lda + a_chk * 2;
    const double* Bg;
    if (B_T) Bg = B + (cb * BN + a_row) * g.ldb + a_chk * 2;
    else Bg = B + (int64_t)b_row * g.ldb + cb * BN + b_chk * 2;

    v2f64 ra[4], rbv[4];
    auto load_stage = [&](int64_t k) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ra[r] = *(const v2f64*)(Ag + (int64_t)(32 * r) * g.lda + k);
        if (B_T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) rbv[r] = *(const v2f64*)(Bg + (int64_t)(32 * r) * g.ldb + k);
            if (g.bscale) {
                v2f64 sc = *(const v2f64*)(g.bscale + k + a_chk * 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) rbv[r] *= sc;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) rbv[r] = *(const v2f64*)(Bg + (k + 4 * r) * g.ldb);
            if (g.bscale) {
                v2f64 sc = *(const v2f64*)(g.bscale + cb * BN + b_chk * 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) rbv[r] *= sc;
            }
        }
    };
    auto store_stage = [&](int buf) {
        double* As = As0 + buf * AS_ELEMS;
        double* Bs = Bs0 + buf * BS_ELEMS;
#pragma unroll
        for (int r = 0; r < 4; ++r) *(v2f64*)(As + (a_row + 32 * r) * LDA_S + a_chk * 2) = ra[r];
        if (B_T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) *(v2f64*)(Bs + (a_row + 32 * r) * LDA_S + a_chk * 2) = rbv[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) *(v2f64*)(Bs + (b_row + 4 * r) * LDB_S + b_chk * 2) = rbv[r];
        }
    };

    v4f64 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};

    const int64_t nk = (k1 > k0) ? (k1 - k0) / BK : 0;
    if (nk > 0) {
        load_stage(k0);
        store_stage(0);
    }
    __syncthreads();
    for (int64_t kt = 0; kt < nk; ++kt) {
        const int buf = (int)(kt & 1);
        if (kt + 1 < nk) load_stage(k0 + (kt + 1) * BK);
        const double* As = As0 + buf * AS_ELEMS + (wr * 64 + li) * LDA_S + lk;
        const double* Bs = B_T ? (Bs0 + buf * BS_ELEMS + (wc * 64 + li) * LDA_S + lk)
                               : (Bs0 + buf * BS_ELEMS + lk * LDB_S + wc * 64 + li);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            double af[4], bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) af[t] = As[t * 16 * LDA_S + ks * 4];
#pragma unroll
            for (int t = 0; t < 4; ++t) bf[t] = B_T ? Bs[t * 16 * LDA_S + ks * 4] : Bs[ks * 4 * LDB_S + t * 16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_stage(buf ^ 1);
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    const int64_t row0 = (int64_t)rb * BM + wr * 64 + lk;     // + mt*16 + 4*r
    const int64_t col0 = cb * BN + wc * 64 + li;               // + nt*16
    if (g.epi == EPI_DA) {
        // dA = alpha*acc + avec[i]*gmu[n] - 2*Aaux[i][n]*cgv[n]
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int64_t col = col0 + nt * 16;
            const double gm = g.gmu[col], cg = g.cgv[col];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = row0 + mt * 16 + 4 * r;
                    C[row * g.ldc + col] =
                        g.alpha * acc[mt][nt][r] + g.avec[row] * gm - 2.0 * g.Aaux[row * g.ldc + col] * cg;
                }
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + mt * 16 + 4 * r;
                const int64_t col = col0 + nt * 16;
                double v = g.alpha * acc[mt][nt][r];
                if (g.accumulate) v += C[row * g.ldc + col];
                acc[mt][nt][r] = v;
                C[row * g.ldc + col] = v;
            }
    if (g.epi == EPI_COLSTATS) {
        // partial column sums over this tile's 128 rows: per lane over its 16 rows, then across the
        // 4 lane groups of the wavefront (shuffles), then across the two row-wavefronts (LDS).
        double* red = lds;   // [2 stats][2 wr][128 cols]   (main-loop LDS is dead after the last barrier)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            double sq = 0.0, dt = 0.0;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = acc[mt][nt][r];
                    sq += v * v;
                    if (g.coldot_part) dt += g.avec[row0 + mt * 16 + 4 * r] * v;
                }
            sq += __shfl_xor(sq, 16);
            sq += __shfl_xor(sq, 32);
            dt += __shfl_xor(dt, 16);
            dt += __shfl_xor(dt, 32);
            if (lk == 0) {
                red[(0 * 2 + wr) * BN + wc * 64 + nt * 16 + li] = sq;
                red[(1 * 2 + wr) * BN + wc * 64 + nt * 16 + li] = dt;
            }
        }
        __syncthreads();
        if (tid < BN) {
            g.colsq_part[(int64_t)rb * g.Nc + cb * BN + tid] = red[tid] + red[BN + tid];
            if (g.coldot_part)
                g.coldot_part[(int64_t)rb * g.Nc + cb * BN + tid] = red[2 * BN + tid] + red[3 * BN + tid];
        }
    }
}

int launch_gemm(const GemmArgs& g, bool B_T, int splitk, hipStream_t s) {
    if (g.Mr % BM || g.Nc % BN || g.Kd % BK) return MOBOCMF_BAD_ARG;
    int nrb = g.Mr / BM;
    int64_t ncb = g.Nc / BN;
    int zdim = g.batched ? g.batched : (splitk > 1 ? splitk : 1);
    dim3 grid((unsigned)(nrb * ncb), 1, (unsigned)zdim);
    if (B_T)
        hipLaunchKernelGGL(gemm_f64_kernel<true>, grid, dim3(256), 0, s, g, nrb, ncb, g.batched ? 1 : splitk);
    else
        hipLaunchKernelGGL(gemm_f64_kernel<false>, grid, dim3(256), 0, s, g, nrb, ncb, g.batched ? 1 : splitk);
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}

__global__ void reduce_slabs_kernel(const double* slabs, int64_t slab_stride, int nslab, double* out, int64_t ld,
                                    int Mr, double scale, int lower_only, int accumulate) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Mr * Mr) return;
    int i = (int)(idx / Mr), j = (int)(idx % Mr);
    double v = 0.0;
    // tiles strictly above the diagonal were never written by a lower_out GEMM: never read them
    if (!lower_only || (j / TILE) <= (i / TILE)) {
        const double* p = slabs + (int64_t)i * Mr + j;
        for (int z = 0; z < nslab; ++z) v += p[z * slab_stride];
        v *= scale;
        if (lower_only && j > i) v = 0.0;
    }
    if (accumulate) v += out[(int64_t)i * ld + j];
    out[(int64_t)i * ld + j] = v;
}

int launch_reduce_slabs(const double* slabs, int64_t slab_stride, int nslab, double* out, int64_t ld, int Mr,
                        double scale, int lower_only, int accumulate, hipStream_t s) {
    int64_t n = (int64_t)Mr * Mr;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slabs, slab_stride,
                       nslab, out, ld, Mr, scale, lower_only, accumulate);
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}
