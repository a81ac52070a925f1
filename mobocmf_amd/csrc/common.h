// Shared declarations for the gfx950 MFDGP kernels (internal; the public C-ABI is include/mobocmf_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mobocmf_hip.h"

#define TILE 128   // every internal matrix is zero-padded to multiples of TILE in both dimensions
#define NB 64      // Cholesky / triangular-inverse panel width (one wavefront = one 64x64 diagonal block)

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return MOBOCMF_HIP_ERROR; \
    } while (0)

// ------------------------------------------------------------------ tuning of the call in progress (api.hip)
// The kernel-selection knobs (include/mobocmf_hip.h: mobocmf_tuning) travel with every C-ABI call.  An entry point opens a
// TuneScope on the pointer it was given (NULL = defaults); the launch helpers below it read tune().  The scope is a
// thread-local pointer that lives for the duration of the call only: nothing survives the call, two threads never see each
// other's values.
const mobocmf_tuning& tune();
bool tuning_ok(const mobocmf_tuning* t);      // NULL, or a struct of the right size with every field in range
void probe_at(int i, hipStream_t s);          // records the i-th probe event of the call in progress, if it has any
struct TuneScope {
    const mobocmf_tuning* prev_t;
    void* const* prev_p;
    explicit TuneScope(const mobocmf_tuning* t, void* const* probe = nullptr);
    ~TuneScope();
    TuneScope(const TuneScope&) = delete;
    TuneScope& operator=(const TuneScope&) = delete;
};

// ------------------------------------------------------------------ GEMM (gemm_f64.hip)
// bit mask: which k range of a tile is structurally non-zero (A is Mr x Kd, logical B is Kd x Nc)
enum { TRI_NONE = 0, TRI_LOWER_A = 1, TRI_UPPER_A = 2, TRI_LOWER_B = 4, TRI_UPPER_B = 8 };
enum { EPI_STORE = 0, EPI_COLSTATS = 1, EPI_DA = 2 };

struct GemmArgs {
    const double* A;   // [Mr x Kd] row-major, lda
    const double* B;   // B_T ? [Nc x Kd] : [Kd x Nc]
    double* C;         // [Mr x Nc]
    int64_t lda, ldb, ldc;
    int Mr;            // rows (multiple of TILE)
    int64_t Nc;        // cols (multiple of TILE)
    int64_t Kd;        // contraction length (multiple of 16)
    int tri;           // TRI_*: restricts the k range per row block (A triangular, square A: Kd == Mr)
    int lower_out;     // 1: skip tiles strictly above the diagonal (square outputs)
    int sym_out;       // 1 (with lower_out, A B^T with B = A: a syrk): the result is symmetric, so the 16 x 16 blocks
                       // strictly above the diagonal of every DIAGONAL tile are not computed either (left as zeros; the
                       // slab reduction mirrors them from below the diagonal)
    int sym_full;      // 1 (small A B^T products only, with sym_out): the small-operand kernel writes the FULL symmetric result
                       // straight into C (every block on or below the block diagonal stores its mirror image too) -- no slab,
                       // no reduction launch; with C2 / bscale2 set the same launch also forms the twin product (blockIdx.z = 1)
    int splitk_diag;   // > 0 (k-sliced sym_out launches): the diagonal tiles -- 10/16 of a full tile's MFMA work on their
                       // critical wavefronts -- take this many k slices instead of `splitk`, so that all workgroups run
                       // equally long; slab z of a tile is only written for z < its own slice count
    double alpha;
    int accumulate;    // 1: C += alpha*A*B  (else C = ...)
    int rm;            // tile height: 0 / 128 = 128 x 128 tiles; 64 = 64 x 128 tiles, three workgroups per CU (plain A B panel
                       // products only: gemm_f64.hip tile_rows(); anything else silently runs on 128-row tiles)
    int pair_mode;     // row-block pairing of triangular launches: 0 = automatic, 1 = never, 2 = always (tuning / tests)
    // split-K: gridDim.z slices of the contraction, slice z writes C + z*slab_stride (reduced by reduce_slabs)
    int64_t slab_stride;
    // batching (only without split-K): blockIdx.z selects the batch
    int64_t strideA, strideB, strideC;
    int batched;
    // layer batching (chains of several layers in one launch): blockIdx.z = batch * zlayers + layer; layer l works on
    // A + l*zsA, B + l*zsB, C + l*zsC (0 / 1 layers: off).  Not with split-K.
    int zlayers;
    int64_t zsA, zsB, zsC;
    // B_T=1: weights of the contraction index, sum_k A[i][k] bscale[k] B[j][k];  EPI_DA: column scale of the product
    const double* bscale;
    int64_t Kreal;                 // > 0: rows >= Kreal of B are zero padding (the real inner dimension, e.g. M of an
                                   // Mp-padded operand): kernels that honour it stop the contraction there
    double* rowdot_part;           // EPI_DA (optional): partial row sums of Aaux[row][col] * gmu[col], [slice][Mr] with
                                   // gemm_rowdot_parts(g) column slices -- da = A g_mean without another pass over A
    int stream_out;                // write C with non-temporal stores: for a panel whose next reader is far away (it
                                   // would only evict what this and the next launches re-read; a panel consumed by the
                                   // NEXT launch must stay cacheable -- streaming A cost the step 4 % although the
                                   // kernel alone got 6 % faster)
    const int32_t* skip_if_zero;   // device word: the whole launch is a no-op when it is 0 (rarely needed passes)
    // Dual launch (k-sliced A B^T only): the grid is doubled; the second half computes the SAME product with the contraction
    // weights bscale2 into the slabs at C2 -- but only if *dual_flag != 0, otherwise its workgroups exit at once.  One launch
    // instead of a real one plus a conditional one that is a no-op nearly always (the clamped-column syrk of the backward).
    const int32_t* dual_flag;
    const double* bscale2;
    double* C2;
    // Column-block activity (one int32 per 128 columns of the N' side; moments_bwd_prep writes it): block j is inactive when
    // every upstream gradient of its columns is exactly zero, i.e. the block's share of every backward product is exactly
    // zero.  A B form (colact): an inactive column block of C is NOT computed (its tiles exit at once; an EPI_DA tile still
    // zeroes its row-dot partials) -- consumers of C must not read those columns.  k-sliced A B^T form (kact): the
    // contraction runs over the active 128-column blocks of K only (compacted in the kernel's prologue); the slices split
    // the ACTIVE steps evenly.  NULL = dense.
    const int32_t* colact;
    const int32_t* kact;
    // epilogues
    int epi;
    double* colsq_part;    // EPI_COLSTATS: [2*Mr/TILE][Nc] partial column sums of C^2 (two wavefront rows per row block)
    double* coldot_part;   // EPI_COLSTATS: [2*Mr/TILE][Nc] partial column sums of avec[i]*C[i][n]   (may be null)
    const double* avec;    // EPI_COLSTATS / EPI_DA: length Mr
    const double* gmu;     // EPI_DA: length Nc
    const double* cgv;     // EPI_DA: length Nc
    const double* Aaux;    // EPI_DA: [Mr x Nc], ld = ldc
};

int launch_gemm(const GemmArgs& g, bool B_T, int splitk, hipStream_t s);
// out[j] (+)= sum_p part[p*stride + j] (+ sum_p part2[p*stride2 + j]); several such reductions in ONE launch when every
// partial count is small (small problems: each separate launch is ~4.5 us of pure latency)
struct SumTask {
    const double* part; int64_t P, stride;
    const double* part2; int64_t P2, stride2;      // optional second source (part2 == nullptr: none)
    double* out; int64_t len; int accumulate;
    const double* extra;                           // optional single row added to the result (nullptr: none)
};
int launch_sum_partials_multi(const SumTask* tasks, int n, hipStream_t s);
int gemm_nt_slabs(const GemmArgs& g, int splitk);   // k-slices (slabs) an A B^T launch will really write
bool gemm_nt_is_small(const GemmArgs& g);           // the product runs on the small-operand kernel (GemmArgs.sym_full applies)
int gemm_rowdot_parts(const GemmArgs& g);   // number of column slices EPI_DA writes to rowdot_part
int gemm_colstat_rows(const GemmArgs& g);   // number of partial rows EPI_COLSTATS writes to colsq_part / coldot_part
int launch_gemm_auto(const GemmArgs& g, bool B_T, double* ws, int64_t ws_elems, hipStream_t s);
int launch_gemm_auto_pair(const GemmArgs& a, const GemmArgs& b, double* ws, int64_t ws_elems, hipStream_t s);
// out[i][j] = scale * sum_z slabs[z][i][j]  (lower_only: zero above the diagonal), Mr x Mr
int launch_reduce_slabs(const double* slabs, int64_t slab_stride, int nslab, double* out, int64_t ld, int Mr,
                        double scale, int lower_only, int accumulate, hipStream_t s);

// zero-fill by a kernel (32-bit words); used instead of hipMemsetAsync so that a captured step contains only
// kernel nodes (a replayed memset node was observed to leave 0xFE bytes in a 4-byte word on ROCm 7.2)
int launch_zero32(void* ptr, int64_t nwords, hipStream_t s);
int launch_zero32_z(void* ptr, int64_t nwords, int nz, int64_t zs_bytes, hipStream_t s);   // nz regions, zs_bytes apart
#define MAX_ZL 4   // layers whose chains one batched call may carry

// ------------------------------------------------------------------ Gram (gram.hip)
struct GramArgs {
    int kind, d, xdiv, zdiv;
    const double* x;      // [nbase x d]
    const double* f;      // [nbase*xdiv] (kind 1)
    int64_t nbase;
    const double* Zx;     // [M/zdiv x d]   row side
    const double* zf;     // [M] (kind 1)
    int M;
    const double* hyp;
    double* K;            // [Mp x Np]
    int64_t ldk;
    int Mp;
    int64_t Np;
    double* knn;          // [Np] or null
    double jitter;
    int is_kmm;           // add jitter on the diagonal, identity on the padded diagonal
    int gm;               // inducing rows per workgroup (set by the launchers: gram_grid's rule)
    // backward
    const double* G;      // dL/dK [Mp x Np], ld = ldk
    const double* gknn;   // [Np] or null
    double* hyp_part;     // [gridDim.y*gridDim.x][H]
    double* df_part;      // [gridDim.y][Np]        (kind 1)
    double* dzf_part;     // [gridDim.x][Mp]        (kind 1)
    double* dx_part;      // [gridDim.y][nbase*d]   (want_dx)
    const int32_t* colact; // per 128 columns of G (GemmArgs.colact): inactive blocks of G were never written and count as zero
};

// ------------------------------------------------------------------ hyper-parameter packing
// kind 0: [alpha, ls[0..d)]                                   (1 + d doubles)
// kind 1: [a1, af, nu, a2, lsf, ls1[0..d), ls2[0..d)]         (5 + 2d doubles)
static inline int hyp_len(int kind, int d) { return kind == 0 ? 1 + d : 5 + 2 * d; }
#define MAX_D MOBOCMF_MAX_D

// ------------------------------------------------------------------ eps of the training branch (elementwise.hip, tiny_step.hip)
// counter-based Philox4x32-10 keyed by (seed, call counter, row) + Box-Muller in float64: the draw of
// mobocmf_propagate_rng_forward, shared with the one-launch step so that both produce the same eps from the same state
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ double philox_normal(uint64_t seed, uint64_t call, uint64_t idx) {
    uint32_t r[4];
    philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)call, (uint32_t)(call >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    // two uniforms in (0, 1) with 53 random bits each, then Box-Muller (the cosine branch)
    const double u1 = ((double)(((uint64_t)(r[0] >> 5) << 26) | (uint64_t)(r[1] >> 6)) + 0.5) * 1.1102230246251565e-16;
    const double u2 = ((double)(((uint64_t)(r[2] >> 5) << 26) | (uint64_t)(r[3] >> 6)) + 0.5) * 1.1102230246251565e-16;
    return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}
__device__ __forceinline__ double philox_uniform(uint64_t seed, uint64_t call, uint64_t idx) {      // (0, 1), 53 random bits
    uint32_t r[4];
    philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)call, (uint32_t)(call >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    return ((double)(((uint64_t)(r[0] >> 5) << 26) | (uint64_t)(r[1] >> 6)) + 0.5) * 1.1102230246251565e-16;
}
