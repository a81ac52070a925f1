// Blocked in-place Cholesky of K_mm and triangular inverse, gfx950.
//
// Right-looking, panel width NB = 64 = one wavefront.  The 64x64 diagonal block is factorised by ONE
// wavefront with the block held row-per-lane in registers; column broadcasts (pivot, L_kj) are
// wavefront cross-lane reads (v_readlane), no LDS and no barriers.  The same wavefront-shuffle scheme
// solves the 64-row panel blocks below the diagonal (X L_jj^T = A_ij) and inverts L_jj (kept for the
// blocked triangular inverse).  The trailing update A22 -= L21 L21^T runs on the f64 MFMA.
// Reference behaviour replaced: gpytorch psd_safe_cholesky(K_mm) -> torch.linalg.cholesky_ex.
#include "common.h"
#include "tile16.h"
#include <atomic>
#include <cstdlib>

typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double bcast(double v, int src) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// The 64x64 routines below hold a block row-per-lane in registers.  Column j of L is published once per step in LDS and
// read back as wave-uniform 16-byte broadcasts (two k per ds_read_b128) -- half the instructions of fetching every
// multiplier with a pair of v_readlane; only the pivot is a cross-lane register read.
typedef double v2f64c __attribute__((ext_vector_type(2)));

// 1 / sqrt(a): v_rsq_f64 seed + two Newton steps (the IEEE sqrt + divide pair is ~60 dependent instructions on the pivot
// chain of every elimination step); relative error ~1 ulp, a <= 0 or NaN gives inf / NaN as before (the caller tests a).
__device__ __forceinline__ double rsqrt_nr(double a) {
    double y = __builtin_amdgcn_rsq(a);
    y = y * (1.5 - (0.5 * a) * y * y);
    // last step in residual form (fused): e = 1 - a y^2 is exact to one rounding, so y + (y/2) e is within ~1 ulp
    const double t = a * y;
    const double e = __builtin_fma(-t, y, 1.0);
    return __builtin_fma(0.5 * y, e, y);
}

struct InfoZ { int32_t* p[MAX_ZL]; };      // per-layer status words (user tensors: not strided)

// Two wavefronts per workgroup: wavefront 0 factorises the 64 x 64 diagonal block, wavefront 1 runs the right-looking
// triangular solve of its 64-row block TWO elimination steps behind (step t needs column t of L and 1 / L_tt only, which
// wavefront 0 publishes in LDS in its step t; one workgroup barrier per step hands them over).  The panel's critical path
// is then the factorisation alone instead of factorisation + solve back to back.
// Both roles are the SAME step -- scale the pivot column entry, eliminate it from the columns to the right with the
// multipliers L[k][j] read from LDS -- so they share one unrolled loop: wavefront 1 keeps its row shifted by two columns
// (v[c + 2]) and reads the LDS image through a pointer shifted by two rows and two columns, which makes its step t the
// loop's iteration t + 2 with the same register indices (and the 16-byte alignment of the multiplier pairs intact).  Two
// role-specific unrolled loops in one kernel made hipcc spill ~2000 registers.
// Block 0 solves against the identity: its wavefront 1 ends with row i of L^-T, i.e. column i of L^-1 (the inverse of the
// diagonal block that the blocked triangular inverse needs), written transposed.
// The loop body is STRAIGHT-LINE code: role differences are selects and pointer choices, never branches.  With basic blocks
// inside the unrolled loop LLVM sinks each update v[k] -= m * L[k][j] down to the block that finally uses v[k], i.e. turns
// the right-looking elimination into a left-looking one that keeps every multiplier ever loaded in registers (2000 spills).
// Iterations in which a role has nothing to do (wavefront 1 in the first two, wavefront 0 in the last two) run on zero
// padding: the LDS image and the reciprocal-pivot vector carry two extra rows / entries on either side.
#define P2_LAG 2
__global__ __launch_bounds__(128) void potrf_panel2_kernel(double* A, int64_t ld, int jb, double* Dinv, double* Ld,
                                                           InfoZ infoz, int64_t zs) {
    __shared__ __attribute__((aligned(16))) double LTp[NB + 2 * P2_LAG + 1][NB];      // rows -2 .. NB + 2 of the image
    __shared__ double rdp[NB + 2 * P2_LAG];
    __shared__ double sink[NB + 2 * P2_LAG];                                           // wavefront 1's "publications"
    __shared__ __attribute__((aligned(16))) double sinkrow[NB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar role
    const int bi = blockIdx.x;
    A += blockIdx.y * zs; Dinv += blockIdx.y * zs; Ld += blockIdx.y * zs;      // layer batching
    int32_t* info = infoz.p[blockIdx.y];
    const int64_t j0 = (int64_t)jb * NB;
    const int lag = wave * P2_LAG;
    double (*LT)[NB] = LTp + P2_LAG;
    double* rd = rdp + P2_LAG;
    // zero padding read by the idle iterations: rows -2, -1 (wavefront 1's first two) and NB .. NB + 2 (wavefront 0's last two)
    for (int e = threadIdx.x; e < P2_LAG * NB; e += 128) (&LTp[0][0])[e] = 0.0;
    for (int e = threadIdx.x; e < (P2_LAG + 1) * NB; e += 128) (&LT[NB][0])[e] = 0.0;
    if (threadIdx.x < P2_LAG) rdp[threadIdx.x] = 0.0;
    double v[NB + P2_LAG];
    double* prow = A + (j0 + (int64_t)bi * NB + lane) * ld + j0;      // wavefront 1: its block's row
    {
        const bool ident = wave == 1 && bi == 0;
        const double* src = wave == 0 ? A + (j0 + lane) * ld + j0 : prow;
#pragma unroll
        for (int c = 0; c < NB + P2_LAG; ++c) v[c] = 0.0;
        if (wave == 0) {
#pragma unroll
            for (int c = 0; c < NB; ++c) v[c] = src[c];
        } else {
#pragma unroll
            for (int c = 0; c < NB; ++c) v[c + P2_LAG] = ident ? (c == lane ? 1.0 : 0.0) : src[c];
        }
    }
    const double* LTw = &LT[0][0] - lag * NB - lag;      // LTw[i*NB + k] = LT[i - lag][k - lag]
    const double* rdw = rd - lag;
    double* pubrow = wave == 0 ? &LT[0][0] + lane : sinkrow + lane;      // + i*NB for wavefront 0 (stride 0 for the sink)
    const int pubstride = wave == 0 ? NB : 0;
    double* pubrd = wave == 0 ? rd : sink + P2_LAG;
    int fail = 0;
    double rinv = rsqrt_nr(bcast(v[0], 0));
    if (!(bcast(v[0], 0) > 0.0)) fail = 1;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NB + P2_LAG; ++i) {
        // both roles take the reciprocal pivot from LDS (wavefront 0 reads back what it has just published: ~100 cycles on
        // its pivot chain, but a role-dependent source compiles to a branch, and any branch in this loop brings the
        // sinking described above back)
        pubrd[i] = rinv;
        const double r = rdw[i];
        const double m = v[i] * r;          // factorisation: L[lane][i];  solve: X[lane][i - lag]
        v[i] = m;
        pubrow[i * pubstride] = lane >= i ? m : 0.0;      // column i of L (wavefront 1: into the sink)
        if (i + 1 < NB + P2_LAG) {
            v[i + 1] -= m * LTw[i * NB + i + 1];
            // the next pivot's reciprocal square root starts now and runs under the updates below (wavefront 1 computes
            // one too and never uses it: a branch here would cost more than the ~20 instructions)
            const double an = bcast(v[i + 1], (i + 1) & 63);
            fail = (fail == 0 && i + 1 < NB && !(an > 0.0)) ? i + 2 : fail;
            rinv = rsqrt_nr(an);
        }
        if ((i + 2) & 1) {
            if (i + 2 < NB + P2_LAG) v[i + 2] -= m * LTw[i * NB + i + 2];
        }
#pragma unroll
        for (int k = (i + 3) & ~1; k + 1 < NB + P2_LAG; k += 2) {
            const v2f64c c = *(const v2f64c*)&LTw[i * NB + k];
            v[k] -= m * c[0];
            v[k + 1] -= m * c[1];
        }
        __syncthreads();          // iteration i's column of L and reciprocal pivot are in LDS for iteration i + lag
        __builtin_amdgcn_sched_barrier(0);
    }
    if (wave == 0) {
        if (bi == 0) {
            double* wrow = Ld + (int64_t)jb * NB * NB + lane * NB;
#pragma unroll
            for (int c = 0; c < NB; ++c) wrow[c] = (c <= lane) ? v[c] : 0.0;
            // the first panel (re)sets the status word, later ones only report the first failure: no separate zero launch
            if (lane == 0 && (jb == 0 || (fail && *info == 0))) *info = fail ? (int32_t)(j0 + fail) : 0;
        }
    } else if (bi == 0) {      // v[k + lag] = L^-T[lane][k] = L^-1[k][lane]: store transposed (one coalesced row per k)
        double* inv = Dinv + (int64_t)jb * NB * NB;
#pragma unroll
        for (int k = 0; k < NB; ++k) inv[k * NB + lane] = (k >= lane) ? v[k + P2_LAG] : 0.0;
    } else {
#pragma unroll
        for (int c = 0; c < NB; ++c) prow[c] = v[c + P2_LAG];
    }
}

// ---- Four columns per hand-over (potrf_panel4_kernel).  The two-role scheme of potrf_panel2_kernel with the elimination
// advancing FOUR columns per iteration: the 4 x 4 pivot block is broadcast out of wavefront 0's registers, factorised
// redundantly in every lane (T, reciprocal diagonal), published, and each lane -- of either role -- turns its four entries
// v[i..i+3] into the four multipliers m = v T^-T by forward substitution and subtracts m L[k][i..i+3]^T from the columns
// k > i + 3.  Same flops and LDS traffic as one column at a time, but ONE publish / barrier / read-back round trip per four
// columns instead of four: the panel is bound by that round trip (~1000 cycles per column with one column per round).
// Wavefront 1 lags ONE iteration (four columns) behind; as in potrf_panel2_kernel both roles are the same straight-line
// code on shifted registers (v[c + 4]) and shifted LDS pointers, idle iterations run on zero padding.
#define P4_LAG 4
#define P4_STEPS (NB / 4)
__global__ __launch_bounds__(128) void potrf_panel4_kernel(double* A, int64_t ld, int jb, double* Dinv, double* Ld,
                                                           InfoZ infoz, int64_t zs) {
    __shared__ __attribute__((aligned(16))) double LTp[NB + 2 * P4_LAG][NB];      // columns -4 .. NB + 4 of the image [col][row]
    __shared__ __attribute__((aligned(16))) double Tp[P4_STEPS + 2][16];          // steps -1 .. 16: T10 T20 T21 T30 T31 T32 r0 r1 r2 r3
    __shared__ __attribute__((aligned(16))) double sinkT[16];
    __shared__ __attribute__((aligned(16))) double sinkcol[4 * NB + 96];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar role
    const int bi = blockIdx.x;
    A += blockIdx.y * zs; Dinv += blockIdx.y * zs; Ld += blockIdx.y * zs;      // layer batching
    int32_t* info = infoz.p[blockIdx.y];
    const int64_t j0 = (int64_t)jb * NB;
    const int lag = wave * P4_LAG;
    double (*LT)[NB] = LTp + P4_LAG;
    for (int e = threadIdx.x; e < P4_LAG * NB; e += 128) { (&LTp[0][0])[e] = 0.0; (&LT[NB][0])[e] = 0.0; }
    if (threadIdx.x < 16) { Tp[0][threadIdx.x] = 0.0; Tp[P4_STEPS + 1][threadIdx.x] = 0.0; }
    double v[NB + P4_LAG];
    double* prow = A + (j0 + (int64_t)bi * NB + lane) * ld + j0;      // wavefront 1: its block's row
    {
        const bool ident = wave == 1 && bi == 0;
        const double* src = wave == 0 ? A + (j0 + lane) * ld + j0 : prow;
#pragma unroll
        for (int c = 0; c < NB + P4_LAG; ++c) v[c] = 0.0;
        if (wave == 0) {
#pragma unroll
            for (int c = 0; c < NB; ++c) v[c] = src[c];
        } else {
#pragma unroll
            for (int c = 0; c < NB; ++c) v[c + P4_LAG] = ident ? (c == lane ? 1.0 : 0.0) : src[c];
        }
    }
    const double* LTw = &LT[0][0] - lag * NB - lag;      // LTw[c*NB + k] = LT[c - lag][k - lag]
    const double* Tw = &Tp[1][0] - (lag / 4) * 16;       // Tw[16*it + e]: step it - lag/4
    double* pubcol = wave == 0 ? &LT[0][0] + lane : sinkcol + lane;      // + c*NB for wavefront 0 (stride 0 rows for the sink)
    const int pubstride = wave == 0 ? NB : 0;
    int fail = 0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < P4_STEPS + 1; ++it) {
        const int i = 4 * it;
        // ---- the pivot block out of wavefront 0's registers (wavefront 1 computes on its own numbers and discards the result)
        const int l0 = i & 63, l1 = (i + 1) & 63, l2 = (i + 2) & 63, l3 = (i + 3) & 63;
        const double p00 = bcast(v[i], l0);
        const double p10 = bcast(v[i], l1), p11 = bcast(v[i + 1], l1);
        const double p20 = bcast(v[i], l2), p21 = bcast(v[i + 1], l2), p22 = bcast(v[i + 2], l2);
        const double p30 = bcast(v[i], l3), p31 = bcast(v[i + 1], l3), p32 = bcast(v[i + 2], l3), p33 = bcast(v[i + 3], l3);
        const double r0 = rsqrt_nr(p00);
        const double T10 = p10 * r0, T20 = p20 * r0, T30 = p30 * r0;
        const double q11 = __builtin_fma(-T10, T10, p11);
        const double r1 = rsqrt_nr(q11);
        const double T21 = __builtin_fma(-T20, T10, p21) * r1, T31 = __builtin_fma(-T30, T10, p31) * r1;
        const double q22 = __builtin_fma(-T21, T21, __builtin_fma(-T20, T20, p22));
        const double r2 = rsqrt_nr(q22);
        const double T32 = __builtin_fma(-T31, T21, __builtin_fma(-T30, T20, p32)) * r2;
        const double q33 = __builtin_fma(-T32, T32, __builtin_fma(-T31, T31, __builtin_fma(-T30, T30, p33)));
        const double r3 = rsqrt_nr(q33);
        if (it < P4_STEPS) {
            int f = 0;
            f = !(q33 > 0.0) ? i + 4 : f;
            f = !(q22 > 0.0) ? i + 3 : f;
            f = !(q11 > 0.0) ? i + 2 : f;
            f = !(p00 > 0.0) ? i + 1 : f;
            fail = (fail == 0) ? f : fail;
        }
        {   // published branch-free (a basic block inside this loop makes LLVM sink the updates: see potrf_panel2_kernel):
            // lane e < 10 stores entry e, every other lane stores into its own slot of the sink
            double* tw = (wave == 0 && it < P4_STEPS) ? &Tp[1 + it][0] : sinkT;
            double tv = T10;
            tv = lane == 1 ? T20 : tv; tv = lane == 2 ? T21 : tv; tv = lane == 3 ? T30 : tv; tv = lane == 4 ? T31 : tv;
            tv = lane == 5 ? T32 : tv; tv = lane == 6 ? r0 : tv; tv = lane == 7 ? r1 : tv; tv = lane == 8 ? r2 : tv;
            tv = lane == 9 ? r3 : tv;
            double* dst = lane < 10 ? tw + lane : sinkcol + 16 + lane;
            *dst = tv;
        }
        // ---- both roles: the step's T from LDS (wavefront 0 reads back what it has just published)
        const double* t = Tw + 16 * it;
        const double a10 = t[0], a20 = t[1], a21 = t[2], a30 = t[3], a31 = t[4], a32 = t[5];
        const double s0 = t[6], s1 = t[7], s2 = t[8], s3 = t[9];
        const double m0 = v[i] * s0;
        const double m1 = __builtin_fma(-m0, a10, v[i + 1]) * s1;
        const double m2 = __builtin_fma(-m1, a21, __builtin_fma(-m0, a20, v[i + 2])) * s2;
        const double m3 = __builtin_fma(-m2, a32, __builtin_fma(-m1, a31, __builtin_fma(-m0, a30, v[i + 3]))) * s3;
        v[i] = m0; v[i + 1] = m1; v[i + 2] = m2; v[i + 3] = m3;
        pubcol[(i + 0) * pubstride] = lane >= i + 0 ? m0 : 0.0;
        pubcol[(i + 1) * pubstride] = lane >= i + 1 ? m1 : 0.0;
        pubcol[(i + 2) * pubstride] = lane >= i + 2 ? m2 : 0.0;
        pubcol[(i + 3) * pubstride] = lane >= i + 3 ? m3 : 0.0;
#pragma unroll
        for (int k = i + 4; k + 1 < NB + P4_LAG; k += 2) {
            const v2f64c c0 = *(const v2f64c*)&LTw[(i + 0) * NB + k];
            const v2f64c c1 = *(const v2f64c*)&LTw[(i + 1) * NB + k];
            const v2f64c c2 = *(const v2f64c*)&LTw[(i + 2) * NB + k];
            const v2f64c c3 = *(const v2f64c*)&LTw[(i + 3) * NB + k];
            v[k] -= m0 * c0[0] + m1 * c1[0] + m2 * c2[0] + m3 * c3[0];
            v[k + 1] -= m0 * c0[1] + m1 * c1[1] + m2 * c2[1] + m3 * c3[1];
        }
        __syncthreads();          // this iteration's T and columns of L are in LDS for wavefront 1's next iteration
        __builtin_amdgcn_sched_barrier(0);
    }
    if (wave == 0) {
        if (bi == 0) {
            double* wrow = Ld + (int64_t)jb * NB * NB + lane * NB;
#pragma unroll
            for (int c = 0; c < NB; ++c) wrow[c] = (c <= lane) ? v[c] : 0.0;
            if (lane == 0 && (jb == 0 || (fail && *info == 0))) *info = fail ? (int32_t)(j0 + fail) : 0;
        }
    } else if (bi == 0) {      // v[k + lag] = L^-T[lane][k] = L^-1[k][lane]: store transposed (one coalesced row per k)
        double* inv = Dinv + (int64_t)jb * NB * NB;
#pragma unroll
        for (int k = 0; k < NB; ++k) inv[k * NB + lane] = (k >= lane) ? v[k + P4_LAG] : 0.0;
    } else {
#pragma unroll
        for (int c = 0; c < NB; ++c) prow[c] = v[c + P4_LAG];
    }
}

// ---- padded variants (NACT = 16: rows/columns >= NACT of the block are identity padding and are left alone; only
// this small instantiation is used -- 32 and 48 made hipcc spill, and so did a templated <64>, so the full block keeps
// the plain routines above).
// a[] = row `lane` of a symmetric 64x64 block; on exit a[c] (c <= lane) = L[lane][c]; LT[j][i] = L[i][j] (0 above the
// diagonal), rd[j] = 1 / L[j][j].  Returns the first failed pivot (1-based) or 0.
// NACT (multiple of 16): rows/columns >= NACT of the block are padding (identity) and are left alone.
template <int NACT>
__device__ __forceinline__ int chol64_pad_rows(double (&a)[NB], int lane, double (*LT)[NB], double* rd) {
    // The reciprocal pivot of step j+1 is started inside step j, right after column j+1 (alone) has received step j's
    // update: its sqrt + divide (~200 cycles of dependent latency) then run under the remaining updates of step j
    // instead of in front of step j+1 (measured: the 64 steps were ~750 cycles each, 20 us of the 45 us panel kernel).
    int fail = 0;
    double rinv;
    {
        const double a00 = bcast(a[0], 0);
        if (!(a00 > 0.0)) fail = 1;
        rinv = rsqrt_nr(a00);
    }
#pragma unroll
    for (int j = 0; j < NACT; ++j) {
        const double lij = a[j] * rinv;          // lanes >= j: L[lane][j]  (lane j: sqrt(ajj))
        a[j] = lij;
        LT[j][lane] = lane >= j ? lij : 0.0;
        if (lane == 0) rd[j] = rinv;
        double rinv_next = 0.0;
        if (j + 1 < NACT) {
            a[j + 1] -= lij * LT[j][j + 1];       // a[k] -= L[lane][j] * L[k][j]   (valid for lanes >= k)
            const double an = bcast(a[j + 1], j + 1);
            if (!(an > 0.0) && fail == 0) fail = j + 2;
            rinv_next = rsqrt_nr(an);
        }
        if ((j + 2) & 1) {
            if (j + 2 < NACT) a[j + 2] -= lij * LT[j][j + 2];
        }
#pragma unroll
        for (int k = (j + 3) & ~1; k < NACT; k += 2) {
            const v2f64c c = *(const v2f64c*)&LT[j][k];
            a[k] -= lij * c[0];
            a[k + 1] -= lij * c[1];
        }
        rinv = rinv_next;
        __builtin_amdgcn_sched_barrier(0);
    }
    return fail;
}

// x[] = row `lane` of L^-1 given L rows in a[] (both lower triangular)
template <int NACT>
__device__ __forceinline__ void trinv64_pad_rows(const double (&a)[NB], double (&x)[NB], int lane) {
#pragma unroll
    for (int k = NB - 1; k >= NACT; --k) x[k] = (lane == k) ? 1.0 : 0.0;      // identity padding
#pragma unroll
    for (int k = NACT - 1; k >= 0; --k) {
        double s = (lane == k) ? 1.0 : 0.0;
#pragma unroll
        for (int t = k + 1; t < NACT; ++t) s -= x[t] * bcast(a[k], t);          // L[t][k] = 0 for padded rows t
        x[k] = s / bcast(a[k], k);
    }
}

// row `lane` of B <- B L^-T  (forward substitution along the row)
template <int NACT>
__device__ __forceinline__ void trsm64_pad_rows(const double (&a)[NB], double (&b)[NB]) {
#pragma unroll
    for (int k = 0; k < NACT; ++k) {
        double s = b[k];
#pragma unroll
        for (int t = 0; t < k; ++t) s -= b[t] * bcast(a[t], k);
        b[k] = s / bcast(a[k], k);
    }
}

// grid.x = number of 64-row blocks at/below the diagonal of panel jb; block = 64 threads (1 wavefront)
// NOTE: every workgroup re-factorises the diagonal block from A, so block 0 must NOT overwrite it in place
// (a workgroup that is scheduled late -- e.g. when other streams occupy the CUs -- would read L_jj instead of
// A_jj).  The factor goes to the side buffer Ld; finish_l_kernel copies it into the diagonal at the end.
template <int NACT>
__global__ __launch_bounds__(64) void potrf_panel_pad_kernel(double* A, int64_t ld, int jb, double* Dinv, double* Ld,
                                                         InfoZ infoz, int64_t zs) {
    __shared__ __attribute__((aligned(16))) double LT[NB][NB];
    __shared__ double rd[NB];
    const int lane = threadIdx.x;
    const int bi = blockIdx.x;
    A += blockIdx.y * zs; Dinv += blockIdx.y * zs; Ld += blockIdx.y * zs;      // layer batching
    int32_t* info = infoz.p[blockIdx.y];
    const int64_t j0 = (int64_t)jb * NB;
    double a[NB];
    const double* drow = A + (j0 + lane) * ld + j0;
#pragma unroll
    for (int c = 0; c < NB; ++c) a[c] = drow[c];
    int fail = chol64_pad_rows<NACT>(a, lane, LT, rd);
    if (bi == 0) {
        double* wrow = Ld + (int64_t)jb * NB * NB + lane * NB;
#pragma unroll
        for (int c = 0; c < NB; ++c) wrow[c] = (c <= lane) ? a[c] : 0.0;
        if (lane == 0 && (jb == 0 || (fail && *info == 0))) *info = fail ? (int32_t)(j0 + fail) : 0;
        double x[NB];
        trinv64_pad_rows<NACT>(a, x, lane);
        double* irow = Dinv + (int64_t)jb * NB * NB + lane * NB;
#pragma unroll
        for (int c = 0; c < NB; ++c) irow[c] = (c <= lane) ? x[c] : 0.0;
    } else {
        double b[NB];
        double* prow = A + (j0 + (int64_t)bi * NB + lane) * ld + j0;
#pragma unroll
        for (int c = 0; c < NB; ++c) b[c] = prow[c];
        trsm64_pad_rows<NACT>(a, b);
#pragma unroll
        for (int c = 0; c < NB; ++c) prow[c] = b[c];
    }
}

// ---------------------------------------------------------------------------------- 64x64x64 MFMA product
#define LD64 66
// acc (2x2 MFMA tiles of the wavefront's 32x32 quadrant) += As[i][k] * (BT ? Bs[j][k] : Bs[k][j])
template <bool BT>
__device__ __forceinline__ void mm64(const double* As, const double* Bs, v4f64 (&acc)[2][2], int wr, int wc, int lane) {
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        double af[2], bf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) af[t] = As[(wr * 32 + t * 16 + li) * LD64 + ks * 4 + lk];
#pragma unroll
        for (int t = 0; t < 2; ++t)
            bf[t] = BT ? Bs[(wc * 32 + t * 16 + li) * LD64 + ks * 4 + lk] : Bs[(ks * 4 + lk) * LD64 + wc * 32 + t * 16 + li];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
}

__device__ __forceinline__ void load64(double* dst, const double* src, int64_t ld, int tid) {
    // 64x64 block, 256 threads: thread -> (row = tid/4 (+0), 16 contiguous doubles)
    const int r = tid >> 2, c0 = (tid & 3) * 16;
#pragma unroll
    for (int c = 0; c < 16; ++c) dst[r * LD64 + c0 + c] = src[(int64_t)r * ld + c0 + c];
}

// trailing update of panel jb: A[ti][tj] -= P_ti * P_tj^T for 64-blocks ti >= tj > jb (P = panel columns)
__global__ __launch_bounds__(256) void syrk64_update_kernel(double* A, int64_t ld, int jb, int64_t zs) {
    const int ti = jb + 1 + blockIdx.x, tj = jb + 1 + blockIdx.y;
    if (tj > ti) return;
    A += blockIdx.z * zs;
    __shared__ double Pi[NB * LD64], Pj[NB * LD64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int64_t j0 = (int64_t)jb * NB;
    load64(Pi, A + (int64_t)ti * NB * ld + j0, ld, tid);
    load64(Pj, A + (int64_t)tj * NB * ld + j0, ld, tid);
    __syncthreads();
    v4f64 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4f64){0, 0, 0, 0};
    mm64<true>(Pi, Pj, acc, wr, wc, lane);
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int64_t row = (int64_t)ti * NB + wr * 32 + i * 16 + lk + 4 * r;
                int64_t col = (int64_t)tj * NB + wc * 32 + j * 16 + li;
                A[row * ld + col] -= acc[i][j][r];
            }
}

// ---------------------------------------------------------------------------------- the factorisation as ONE launch
// potrf_coop_kernel: all 64-column steps of the right-looking factorisation inside one launch, G workgroups per layer, resident
// together.  Workgroup 0 (the PANEL workgroup) owns the critical path: it factorises the 64 x 64 diagonal block of step jb in
// LDS -- four 16-wide pivot tiles row-per-lane in registers by wavefront 0 (chol_inv_tile16), the updates between them on the
// matrix cores by wavefronts 1-3, the block's inverse row by row by wavefronts 4-7 as the pivots complete -- then (look-ahead)
// the next panel block L[jb+1, jb] = A[jb+1, jb] L_jj^-T and with it the next diagonal block, and goes straight on to step
// jb + 1; the two blocks of the next block row are fetched by wavefronts 4-7 WHILE the pivots run.  The TRAILING workgroups do
// the rest of step jb meanwhile (X: the panel blocks below and the update of column jb + 1, Y: the update of the columns beyond),
// and the INVERSE workgroups form L^-1 block row by block row behind them.  Hand-over is by monotonic words per layer (zeroed
// before the launch): `f` (steps published by the panel workgroup), `pri` (look-ahead blocks of the next step updated), `xd`
// (panel column complete), and an arrival counter per group.  Every wait is bounded: a workgroup that gives up writes info = -1
// and leaves, and so do its peers.
// Replaces the launch pair potrf_panel4_kernel + syrk64_update_kernel per 64 columns (launch_potrf_z below).
#define PC_T 512
#define PC_SPIN (1 << 22)
#define PC_S (NB * LD64)
#define PC_LDS_DOUBLES (4 * PC_S)      // the others' four dense blocks; the panel workgroup's 62 tiles (15 872 doubles) fit in it
#define PC_WORDS 8                     // sync words per layer

__device__ __forceinline__ bool pc_wait_ge(unsigned long long* w, unsigned long long target, int* flag_lds) {
    __syncthreads();
    if (threadIdx.x == 0) {
        int spins = 0, ok = 1;
        while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > PC_SPIN) { ok = 0; break; }
        }
        __threadfence();
        *flag_lds = ok;
    }
    __syncthreads();
    return *flag_lds != 0;
}
// publish what the workgroup wrote so far: all its stores are issued (barrier), then one agent-scope release by thread 0
__device__ __forceinline__ void pc_signal_store(unsigned long long* w, unsigned long long v) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void pc_signal_add(unsigned long long* w) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(w, 1ull);
    }
}
__device__ __forceinline__ bool pc_barrier(unsigned long long* cnt, unsigned n, int* flag_lds) {
    __syncthreads();
    if (n > 1) {
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned long long old = atomicAdd(cnt, 1ull), target = (old / n + 1ull) * n;
            int spins = 0, ok = 1;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > PC_SPIN) { ok = 0; break; }
            }
            __threadfence();
            *flag_lds = ok;
        }
        __syncthreads();
        return *flag_lds != 0;
    }
    return true;
}
#ifdef PC_STAMPS
// diagnostic build (tools/build_variant.sh pcstamps -DPC_STAMPS; tools/chol_stamps.py): wall-clock stamps of the panel workgroup
__device__ double pc_stamp_buf[1024];
extern "C" int mobocmf_debug_potrf_stamps(double* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pc_stamp_buf), sizeof(double) * 1024) == hipSuccess ? 0 : 3;
}
#define PCSTAMP(id) do { if (tid == 0 && z == 0 && nst < 510) { pc_stamp_buf[2 * nst + 2] = (double)(id); pc_stamp_buf[2 * nst + 3] = (double)wall_clock64(); ++nst; pc_stamp_buf[0] = (double)nst; } } while (0)
#else
#define PCSTAMP(id) do { } while (0)
#endif
typedef const double __attribute__((address_space(1)))* pc_gc;
typedef double __attribute__((address_space(1)))* pc_gw;
// a 64 x 64 block (row-major, ld) -> LDS [64][LD64], 512 threads; a wavefront reads one 512-byte row per instruction, all 8
// requests of a thread in flight before the first LDS store
__device__ __forceinline__ void pc_load64(double* dst, const double* src, int64_t ld, int tid) {
    pc_gc g = (pc_gc)src;
    double v[8];
    const int c = tid & 63, r0 = tid >> 6;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = g[(int64_t)(r0 + 8 * q) * ld + c];
#pragma unroll
    for (int q = 0; q < 8; ++q) dst[(r0 + 8 * q) * LD64 + c] = v[q];
}
// 64 x 64 x 64 on eight wavefronts: wavefront (wr = wave / 4, wc = wave % 4) forms rows 32 wr .. +32, columns 16 wc .. +16 of
// As[i][k] * (BT ? Bs[j][k] : Bs[k][j])
template <bool BT>
__device__ __forceinline__ void pc_mm64(const double* As, const double* Bs, v4f64 (&acc)[2], int wr, int wc, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    acc[0] = acc[1] = (v4f64){0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const double a0 = As[(wr * 32 + li) * LD64 + ks * 4 + lk], a1 = As[(wr * 32 + 16 + li) * LD64 + ks * 4 + lk];
        const double bf = BT ? Bs[(wc * 16 + li) * LD64 + ks * 4 + lk] : Bs[(ks * 4 + lk) * LD64 + wc * 16 + li];
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bf, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bf, acc[1], 0, 0, 0);
    }
}
// dst (global 64 x 64 block) += sgn * acc: all loads first, then the stores
__device__ __forceinline__ void pc_add_global(double* blk, int64_t ld, const v4f64 (&acc)[2], double sgn, int wr, int wc, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    pc_gw g = (pc_gw)blk;
    double v[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[t][r] = g[(int64_t)(wr * 32 + t * 16 + lk + 4 * r) * ld + wc * 16 + li];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) g[(int64_t)(wr * 32 + t * 16 + lk + 4 * r) * ld + wc * 16 + li] = __builtin_fma(sgn, acc[t][r], v[t][r]);
}
// tile (a, b <= a) number u of a packed lower triangle, u < 10
__device__ __forceinline__ void pc_tile_of(int u, int& a, int& b) {
    a = (u >= 1) + (u >= 3) + (u >= 6);
    b = u - a * (a + 1) / 2;
}
// the 10 lower 16 x 16 tiles of a 64 x 64 block -> swizzled tiles (256 threads, p = r * 16 + c)
__device__ __forceinline__ void pc_fetch_lower(double* dst, const double* blk, int64_t ld, int p) {
    pc_gc g = (pc_gc)blk;
    double v[10];
    const int r = p >> 4, c = p & 15;
#pragma unroll
    for (int t = 0; t < 10; ++t) {
        int ti, tj;
        pc_tile_of(t, ti, tj);
        v[t] = g[(int64_t)(ti * 16 + r) * ld + tj * 16 + c];
    }
#pragma unroll
    for (int t = 0; t < 10; ++t) dst[t * 256 + tel(r, c)] = v[t];
}
// all 16 tiles (tile (a, k) at (4 a + k) * 256)
__device__ __forceinline__ void pc_fetch_full(double* dst, const double* blk, int64_t ld, int p) {
    pc_gc g = (pc_gc)blk;
    double v[16];
    const int r = p >> 4, c = p & 15;
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = g[(int64_t)((t >> 2) * 16 + r) * ld + (t & 3) * 16 + c];
#pragma unroll
    for (int t = 0; t < 16; ++t) dst[t * 256 + tel(r, c)] = v[t];
}

// C += sgn * A B over a list of 64-block tasks.  The destination values are requested together with the operands, so the
// read-modify-write of C costs no round trip of its own after the matrix instructions.  (Tried: requesting task u + step's
// operands and destination before the matrix instructions of task u -- 24 more live registers per thread put 65 values into
// scratch and the launch got slower, 225 -> 250 us at n = 512.)
struct PcTask { const double* a; int64_t lda; const double* b; int64_t ldb; double* c; int64_t ldc; };
template <bool BT, class GetTask>
__device__ __forceinline__ void pc_tasks(int first, int ntask, int step, GetTask get, double sgn, double* Sa, double* Sb, int tid,
                                         int wr, int wc, int lane) {
    const int li = lane & 15, lk = lane >> 4, c = tid & 63, r0 = tid >> 6;
    for (int u = first; u < ntask; u += step) {
        const PcTask t = get(u);
        pc_gc ga = (pc_gc)t.a;
        pc_gc gb = (pc_gc)t.b;
        pc_gw gd = (pc_gw)t.c;
        double va[8], vb[8], vc[2][4];
#pragma unroll
        for (int q = 0; q < 8; ++q) va[q] = ga[(int64_t)(r0 + 8 * q) * t.lda + c];
#pragma unroll
        for (int q = 0; q < 8; ++q) vb[q] = gb[(int64_t)(r0 + 8 * q) * t.ldb + c];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) vc[h][r] = gd[(int64_t)(wr * 32 + h * 16 + lk + 4 * r) * t.ldc + wc * 16 + li];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            Sa[(r0 + 8 * q) * LD64 + c] = va[q];
            Sb[(r0 + 8 * q) * LD64 + c] = vb[q];
        }
        __syncthreads();
        v4f64 acc[2];
        pc_mm64<BT>(Sa, Sb, acc, wr, wc, lane);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                gd[(int64_t)(wr * 32 + h * 16 + lk + 4 * r) * t.ldc + wc * 16 + li] = __builtin_fma(sgn, acc[h][r], vc[h][r]);
        __syncthreads();
    }
}

__global__ __launch_bounds__(PC_T) void potrf_coop_kernel(double* A, int64_t ld, int nreal, double* Dinv, double* Ld, InfoZ infoz,
                                                           int64_t zs, unsigned long long* sync, double* Linv, int nblk, int NT) {
    extern __shared__ __attribute__((aligned(16))) double pc_lds[];
    __shared__ int flag, giveup;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int z = blockIdx.y, w = blockIdx.x, NI = (int)gridDim.x - 1 - NT;      // 1 panel + NT trailing + NI inverse workgroups
    A += z * zs; Dinv += z * zs; Ld += z * zs;
    if (Linv) Linv += z * zs;
    int32_t* info = infoz.p[z];
    unsigned long long* const F = sync + z * PC_WORDS;
    unsigned long long* const PRI = F + 1;       // steps whose look-ahead blocks the first trailing workgroup has updated
    unsigned long long* const BART = F + 2;      // barrier of the trailing workgroups
    unsigned long long* const XD = F + 3;        // steps whose panel L[:, jb] is complete
    unsigned long long* const BARI = F + 4;      // barrier of the inverse workgroups
    if (tid == 0) giveup = 0;
#ifdef PC_STAMPS
    int nst = 0;
#endif
    if (w == 0) {
        // ------------------------------------------------------------------ the panel workgroup
        double* const T0 = pc_lds;               // the diagonal block of this step / of the next one, 10 swizzled lower tiles each
        double* const T1 = T0 + 10 * 256;
        double* const Lip = T1 + 10 * 256;       // the block's inverse, same layout
        double* const An = Lip + 10 * 256;       // A[jb+1, jb], 16 tiles
        double* const Ln = An + 16 * 256;        // L[jb+1, jb], 16 tiles
        int fail = 0, cur = 0;
        if (tid < 256) pc_fetch_lower(T0, A, ld, tid);
        __syncthreads();
        for (int jb = 0; jb < nreal; ++jb) {
            double* const Lp = cur ? T1 : T0;
            double* const Dn = cur ? T0 : T1;
            const bool need = jb + 1 < nreal;
            bool fetched = !need || wave < 4;
            const unsigned long long target = (unsigned long long)jb;
            // wavefronts 4-7: the next block row's two blocks, as soon as the other workgroups are through with them
            auto try_fetch = [&](bool block) {
                if (fetched) return;
                bool ready = jb == 0;
                if (!ready) {
                    int spins = 0;
                    for (;;) {
                        ready = __hip_atomic_load(PRI, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
                        if (ready || !block) break;
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > PC_SPIN) { giveup = 1; break; }
                    }
                }
                if (ready) {
                    __threadfence();
                    const double* row = A + (int64_t)(jb + 1) * NB * ld;
                    pc_fetch_full(An, row + (int64_t)jb * NB, ld, tid - 256);
                    pc_fetch_lower(Dn, row + (int64_t)(jb + 1) * NB, ld, tid - 256);
                    fetched = true;
                }
            };
            // (A) the 64 x 64 diagonal block: 16-wide right-looking with look-ahead (coop_step.hip ph_chain, four tiles)
            PCSTAMP(1);
            if (wave == 0) {
                const int f = chol_inv_tile16(Lp + tix(0, 0), Lip + tix(0, 0), lane);
                if (f && !fail) fail = jb * NB + f;
            }
            try_fetch(false);
            __syncthreads();
#pragma unroll 1
            for (int s = 0; s < 3; ++s) {
                const double* Bi = Lip + tix(s, s);
                {
                    const int i = s + 1 + wave;      // panel tile (i, s): L_is = A_is L_ss^-T
                    if (i < 4) {
                        double* T = Lp + tix(i, s);
                        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc = mfma(T[tel(li, 4 * q + lk)], Bi[tel(li, 4 * q + lk)], acc);
#pragma unroll
                        for (int r = 0; r < 4; ++r) T[tel(4 * r + lk, li)] = acc[r];
                    }
                }
                __syncthreads();
                if (wave < 4) {
                    const int nrem = 3 - s, ntr = nrem * (nrem + 1) / 2;
                    for (int u = wave; u < ntr; u += 3 + (wave == 0 ? ntr : 0)) {      // A_ab -= L_as L_bs^T; u = 0: the next pivot tile, wavefront 0 alone
                        int a, b;
                        pc_tile_of(u, a, b);
                        const double* P = Lp + tix(s + 1 + a, s);
                        const double* Q = Lp + tix(s + 1 + b, s);
                        double* D = Lp + tix(s + 1 + a, s + 1 + b);
                        v4d acc;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] = D[tel(4 * r + lk, li)];
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc = mfma(-P[tel(li, 4 * q + lk)], Q[tel(li, 4 * q + lk)], acc);
#pragma unroll
                        for (int r = 0; r < 4; ++r) D[tel(4 * r + lk, li)] = acc[r];
                        if (u == 0) {
                            const int f = chol_inv_tile16(D, Lip + tix(s + 1, s + 1), lane);
                            if (f && !fail) fail = jb * NB + (s + 1) * 16 + f;
                        }
                    }
                } else {
                    // row s of the block's inverse (its pivots are complete): X_sc = -L_ss^-1 sum_{t = c .. s-1} L_st X_tc, column c
                    // by wavefront 4 + c
                    const int c = wave - 4;
                    if (c < s) {
                        v4d acc = {0.0, 0.0, 0.0, 0.0};
                        for (int t = c; t < s; ++t) {
                            const double* P = Lp + tix(s, t);
                            const double* Q = Lip + tix(t, c);
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc = mfma(P[tel(li, 4 * q + lk)], Q[tel(4 * q + lk, li)], acc);
                        }
                        const double* Aii = Lip + tix(s, s);
                        v4d d2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int q = 0; q < 4; ++q) d2 = mfma(Aii[tel(li, 4 * q + lk)], acc[q], d2);      // the accumulator IS the B fragment
                        double* X = Lip + tix(s, c);
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[tel(4 * r + lk, li)] = -d2[r];
                    }
                    try_fetch(false);
                }
                __syncthreads();
            }
            PCSTAMP(2);
            if (wave >= 4) {      // the last row of the inverse
                const int c = wave - 4;
                if (c < 3) {
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
                    for (int t = c; t < 3; ++t) {
                        const double* P = Lp + tix(3, t);
                        const double* Q = Lip + tix(t, c);
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc = mfma(P[tel(li, 4 * q + lk)], Q[tel(4 * q + lk, li)], acc);
                    }
                    const double* Aii = Lip + tix(3, 3);
                    v4d d2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < 4; ++q) d2 = mfma(Aii[tel(li, 4 * q + lk)], acc[q], d2);
                    double* X = Lip + tix(3, c);
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[tel(4 * r + lk, li)] = -d2[r];
                }
                try_fetch(true);
            }
            __syncthreads();
            if (giveup) {
                if (tid == 0) *info = -1;
                return;
            }
            PCSTAMP(3);
            // (C) L_jj and its inverse out (dense, zeros above the diagonal)
            {
                pc_gw Ldb = (pc_gw)(Ld + (int64_t)jb * NB * NB);
                pc_gw Dib = (pc_gw)(Dinv + (int64_t)jb * NB * NB);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = tid + PC_T * q, r = e >> 6, c = e & 63, ti = r >> 4, tj = c >> 4;
                    double vl = 0.0, vi = 0.0;
                    if (tj <= ti) {
                        vl = Lp[tix(ti, tj) + tel(r & 15, c & 15)];
                        vi = Lip[tix(ti, tj) + tel(r & 15, c & 15)];
                    }
                    Ldb[e] = vl;
                    Dib[e] = vi;
                }
            }
            PCSTAMP(4);
            if (!need) {
                if (Linv) pc_signal_store(F, (unsigned long long)(jb + 1));      // (the others invert the last block row)
                break;
            }
            // (D) look-ahead: L[jb+1, jb] = A[jb+1, jb] L_jj^-T, tile (a, tj) from the k tiles <= tj (the inverse is lower triangular);
            // wavefront w: a = w / 2 and the tile pair {0, 3} or {1, 2} -- five k tiles each
            {
                const int a = wave >> 1;
                pc_gw Pn = (pc_gw)(A + (int64_t)(jb + 1) * NB * ld + (int64_t)jb * NB);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int tj = (wave & 1) ? 1 + h : 3 * h;
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
                    for (int tk = 0; tk <= tj; ++tk) {
                        const double* P = An + (4 * a + tk) * 256;
                        const double* Q = Lip + tix(tj, tk);
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc = mfma(P[tel(li, 4 * q + lk)], Q[tel(li, 4 * q + lk)], acc);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        Ln[(4 * a + tj) * 256 + tel(4 * r + lk, li)] = acc[r];
                        Pn[(int64_t)(a * 16 + 4 * r + lk) * ld + tj * 16 + li] = acc[r];
                    }
                }
            }
            PCSTAMP(5);
            __syncthreads();      // Ln is complete for everybody, and the stores of L[jb+1, jb], L_jj^-1 are issued
            PCSTAMP(6);
            // the next diagonal block: D_ab -= sum_k Ln_ak Ln_bk^T.  Wavefront 0 takes the next pivot tile alone and goes on with it;
            // wavefront 4 -- on wavefront 0's SIMD -- only publishes the step (a release fence costs ~1 us), the other six share
            // the remaining nine tiles
            if (wave == 4) {
                if (lane == 0) {
                    __threadfence();
                    __hip_atomic_store(F, (unsigned long long)(jb + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const int slot = wave == 0 ? 0 : wave < 4 ? wave : wave - 1;      // 0 | 1 2 3 | (4: none) 4 5 6
            for (int u = wave == 4 ? 10 : slot; u < 10; u += wave == 0 ? 10 : 6) {
                int a, b;
                pc_tile_of(u, a, b);
                double* D = Dn + tix(a, b);
                v4d acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = D[tel(4 * r + lk, li)];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double* P = Ln + (4 * a + k) * 256;
                    const double* Q = Ln + (4 * b + k) * 256;
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = mfma(-P[tel(li, 4 * q + lk)], Q[tel(li, 4 * q + lk)], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) D[tel(4 * r + lk, li)] = acc[r];
            }
            cur ^= 1;
            PCSTAMP(7);
        }
        if (tid == 0) *info = fail;
        return;
    }
    double* S0 = pc_lds;
    double* S1 = S0 + PC_S;      // L_jj^-1
    double* S2 = S1 + PC_S;      // L[jb+1, jb]
    double* S3 = S2 + PC_S;
    const int wr = wave >> 2, wc = wave & 3;
    if (w <= NT) {
        // ------------------------------------------------------------------ the trailing workgroups
        // Step jb: X: L[i, jb] = A[i, jb] L_jj^-T and A[i, jb+1] -= L[i, jb] L[jb+1, jb]^T (i >= jb+2); barrier; Y: A[i, c] -=
        // L[i, jb] L[c, jb]^T for the columns beyond (c >= jb+2); barrier.  The FIRST of them takes row jb+2 and, straight after
        // it, the diagonal block (jb+2, jb+2) -- it needs L[jb+2, jb] only, which is in its LDS -- and publishes `pri`: those are
        // the two blocks the panel workgroup fetches next, so its wait ends ~12 us after it published the step, not after the
        // whole update (~23 us: five hand-overs across XCDs at 2-3 us each).
        // The first trailing workgroup does nothing else but what the look-ahead of the NEXT steps hangs on -- after the barrier
        // the two blocks of row jb+3 that its own next priority row reads, (jb+3, jb+2) and (jb+3, jb+3) -- and does not join the
        // others' second barrier: it never waits for the slowest block of Y.
        const int tw = w - 1, NR = NT - 1;      // NR: the trailing workgroups but the first
        auto do_row = [&](int jb, int i, bool pri) {
            double* Pi = A + (int64_t)i * NB * ld + (int64_t)jb * NB;
            pc_load64(S0, Pi, ld, tid);
            double cc[2][4], cd[2][4];      // A[i, jb+1] (and A[i, i] on the priority row): requested now, used after the products
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t ro = (int64_t)(wr * 32 + t * 16 + lk + 4 * r) * ld + wc * 16 + li;
                    cc[t][r] = ((pc_gc)Pi)[ro + NB];
                    cd[t][r] = pri ? ((pc_gc)Pi)[ro + 2 * NB] : 0.0;
                }
            __syncthreads();
            v4f64 acc[2];
            pc_mm64<true>(S0, S1, acc, wr, wc, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wr * 32 + t * 16 + lk + 4 * r, col = wc * 16 + li;
                    ((pc_gw)Pi)[(int64_t)row * ld + col] = acc[t][r];
                    S3[row * LD64 + col] = acc[t][r];
                }
            __syncthreads();
            pc_mm64<true>(S3, S2, acc, wr, wc, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ((pc_gw)Pi)[(int64_t)(wr * 32 + t * 16 + lk + 4 * r) * ld + NB + wc * 16 + li] = cc[t][r] - acc[t][r];
            if (pri) {
                pc_mm64<true>(S3, S3, acc, wr, wc, lane);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ((pc_gw)Pi)[(int64_t)(wr * 32 + t * 16 + lk + 4 * r) * ld + 2 * NB + wc * 16 + li] = cd[t][r] - acc[t][r];
                pc_signal_store(PRI, (unsigned long long)(jb + 1));
            }
            __syncthreads();
        };
        for (int jb = 0; jb + 2 < nreal; ++jb) {
            if (!pc_wait_ge(F, (unsigned long long)(jb + 1), &flag)) {
                if (tid == 0) *info = -1;
                return;
            }
            const int64_t j0 = (int64_t)jb * NB;
            pc_load64(S1, Dinv + (int64_t)jb * NB * NB, NB, tid);
            pc_load64(S2, A + (int64_t)(jb + 1) * NB * ld + j0, ld, tid);
            if (tw == 0) {
                do_row(jb, jb + 2, true);
                if (NR == 0)
                    for (int i = jb + 3; i < nreal; ++i) do_row(jb, i, false);
            } else {
                for (int i = jb + 3 + tw - 1; i < nreal; i += NR) do_row(jb, i, false);
            }
            if (!pc_barrier(BART, (unsigned)NT, &flag)) {
                if (tid == 0) *info = -1;
                return;
            }
            if (tw == 0 && NI > 0) pc_signal_store(XD, (unsigned long long)(jb + 1));      // (every L[i, jb] is written: the barrier)
            // Y, column by column: task u of the packed lower triangle of order nrem.  Task 0 -- block (jb+2, jb+2) -- is done;
            // tasks 1 and nrem -- (jb+3, jb+2), (jb+3, jb+3) -- are the first workgroup's, the rest the others'
            const int nrem = nreal - jb - 2, ntr = nrem * (nrem + 1) / 2;
            auto task = [&](int u) {
                int b = 0, rest = u;
                while (rest >= nrem - b) { rest -= nrem - b; ++b; }
                const int ci = jb + 2 + b, ri = ci + rest;
                return PcTask{A + (int64_t)ri * NB * ld + j0, ld, A + (int64_t)ci * NB * ld + j0, ld,
                              A + (int64_t)ri * NB * ld + (int64_t)ci * NB, ld};
            };
            const int nspecial = nrem >= 2 ? 3 : 1;
            auto rest_task = [&](int v) {
                int u = v + (nrem >= 2 ? 2 : 1);
                if (nrem >= 2 && u >= nrem) ++u;
                return task(u);
            };
            if (tw == 0) {
                if (nrem >= 2) pc_tasks<true>(0, 2, 1, [&](int v) { return task(v ? nrem : 1); }, -1.0, S0, S3, tid, wr, wc, lane);
                if (NR == 0) pc_tasks<true>(0, ntr - nspecial, 1, rest_task, -1.0, S0, S3, tid, wr, wc, lane);
            } else {
                pc_tasks<true>(tw - 1, ntr - nspecial, NR, rest_task, -1.0, S0, S3, tid, wr, wc, lane);
                if (jb + 3 < nreal && !pc_barrier(BART + 3, (unsigned)NR, &flag)) {      // this step's Y before the next step's X
                    if (tid == 0) *info = -1;
                    return;
                }
            }
        }
        return;
    }
    // ---------------------------------------------------------------------- the inverse workgroups
    // X = L^-1 (nblk 64-blocks square, ld = 64 nblk), right-looking in 64-blocks: S_ic = sum_{t < i} L_it X_tc is accumulated in
    // X's own storage as the block rows of X complete -- Zf(jb): X_jb,c = -L_jj^-1 S_jb,c (c < jb), X_jb,jb = L_jj^-1; Zu(jb):
    // S_ic += L_i,jb X_jb,c for i > jb, c <= jb -- so the inverse is complete one product after the last step's pivots (it
    // replaces trtri_level0_kernel and the merge products of launch_trtri_z).
    const int iw = w - NT - 1;
    const int64_t ldi = (int64_t)nblk * NB;
    // X <- 0, identity on the diagonal blocks of the padding rows
    for (int u = iw; u < nblk * nblk; u += NI) {
        const int bi = u / nblk, bj = u - bi * nblk;
        pc_gw g = (pc_gw)(Linv + (int64_t)bi * NB * ldi + (int64_t)bj * NB);
        const bool eye = bi == bj && bi >= nreal;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = (tid >> 6) + 8 * q, c = tid & 63;
            g[(int64_t)r * ldi + c] = eye && r == c ? 1.0 : 0.0;
        }
    }
    if (!pc_barrier(BARI, (unsigned)NI, &flag)) {
        if (tid == 0) *info = -1;
        return;
    }
    for (int jb = 0; jb < nreal; ++jb) {
        if (!pc_wait_ge(F, (unsigned long long)(jb + 1), &flag) ||
            (jb + 2 < nreal && !pc_wait_ge(XD, (unsigned long long)(jb + 1), &flag))) {
            if (tid == 0) *info = -1;
            return;
        }
        const int64_t j0 = (int64_t)jb * NB;
        pc_load64(S1, Dinv + (int64_t)jb * NB * NB, NB, tid);
        __syncthreads();
        // Zf: block row jb of the inverse
        for (int c = iw; c <= jb; c += NI) {
            double* Xc = Linv + (int64_t)jb * NB * ldi + (int64_t)c * NB;
            if (c == jb) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = (tid >> 6) + 8 * q, cc = tid & 63;
                    ((pc_gw)Xc)[(int64_t)r * ldi + cc] = S1[r * LD64 + cc];
                }
            } else {
                pc_load64(S0, Xc, ldi, tid);
                __syncthreads();
                v4f64 acc[2];
                pc_mm64<false>(S1, S0, acc, wr, wc, lane);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ((pc_gw)Xc)[(int64_t)(wr * 32 + t * 16 + lk + 4 * r) * ldi + wc * 16 + li] = -acc[t][r];
                __syncthreads();
            }
        }
        if (jb + 1 == nreal) break;
        if (!pc_barrier(BARI, (unsigned)NI, &flag)) {
            if (tid == 0) *info = -1;
            return;
        }
        // Zu: S_ic += L[i, jb] X[jb, c], i > jb, c <= jb
        const int nc = jb + 1, ntask = (nreal - jb - 1) * nc;
        pc_tasks<false>(iw, ntask, NI,
                        [&](int u) {
                            const int i = jb + 1 + u / nc, c = u % nc;
                            return PcTask{A + (int64_t)i * NB * ld + j0, ld, Linv + (int64_t)jb * NB * ldi + (int64_t)c * NB, ldi,
                                          Linv + (int64_t)i * NB * ldi + (int64_t)c * NB, ldi};
                        },
                        1.0, S0, S3, tid, wr, wc, lane);
        if (!pc_barrier(BARI, (unsigned)NI, &flag)) {
            if (tid == 0) *info = -1;
            return;
        }
    }
}

// zero the strict upper triangle
__global__ void tril_inplace_kernel(double* A, int64_t ld, int n) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * n) return;
    int i = (int)(idx / n), j = (int)(idx % n);
    if (j > i) A[(int64_t)i * ld + j] = 0.0;
}

int launch_tril_inplace(double* A, int64_t ld, int n, hipStream_t s) {
    int64_t n2 = (int64_t)n * n;
    hipLaunchKernelGGL(tril_inplace_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, A, ld, n);
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}

// strict upper triangle <- 0, diagonal 64x64 blocks <- the factors kept in Ld; z0 / z1 (n x n each, may be null) <- 0: the
// buffers the triangular inverse and U = L^-1 L_S fill only on and below the block diagonal (no separate zero launches)
// Diagonal blocks >= nreal lie entirely in the identity padding of K_mm (no panel touched them): their Ld / Dinv blocks are
// set to the identity HERE (round 3: a launch of their own) and the diagonal of A with them.
__global__ void finish_l_kernel(double* A, int64_t ld, int n, double* Ld, double* Dinv, int nreal, int64_t zs, double* z0,
                                double* z1) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * n) return;
    A += blockIdx.z * zs; Ld += blockIdx.z * zs; Dinv += blockIdx.z * zs;
    if (z0) z0[blockIdx.z * zs + idx] = 0.0;
    if (z1) z1[blockIdx.z * zs + idx] = 0.0;
    int i = (int)(idx / n), j = (int)(idx % n);
    if (i / NB == j / NB) {
        const int b = i / NB;
        const int64_t e = (int64_t)b * NB * NB + (i % NB) * NB + (j % NB);
        if (b >= nreal) {
            const double v = i == j ? 1.0 : 0.0;
            Ld[e] = v;
            Dinv[e] = v;
            A[(int64_t)i * ld + j] = v;
            return;
        }
        A[(int64_t)i * ld + j] = j > i ? 0.0 : Ld[e];
        return;
    }
    if (j > i) A[(int64_t)i * ld + j] = 0.0;
}

// columns per hand-over of the 64-wide panel kernel: tune().potrf_cols = 4 (potrf_panel4_kernel, default) or 1 (potrf_panel2_kernel)
// Dinv and Ld: (Mp/64) x 64 x 64 doubles each.  M = real order: rows/columns >= M of A are identity padding, which the
// factorisation leaves alone -- a 16-point problem padded to 128 costs 16 elimination steps, not 128.
// workgroups per layer of the one-launch form: the panel workgroup + nt trailing + ni inverse.  What the others' steps cost is
// hand-overs, and a barrier grows with its participants: beyond 32 trailing and 16 inverse workgroups the launch gets SLOWER
// (n = 1024: 32 + 16 -> 446 us, 51 + 32 -> 491, 77 + 48 -> 576; 25 + 8 -> 558: the inverse group then trails the pivots;
// profiles/r05_chol_one_launch.txt).  Capped so that three such launches -- the surrogates of a step, one stream each -- fit the
// 256 CUs side by side: all workgroups of a launch must be resident together.
static void potrf_coop_workgroups(int nreal, int nz, bool inverse, int& nt, int& ni) {
    int cap = 84 / nz - 1;
    cap = cap < 15 ? 15 : cap;
    const int y0 = (nreal - 2) * (nreal - 1) / 2, z0 = inverse ? ((nreal + 1) / 2) * (nreal / 2) : 0;
    ni = z0 > 16 ? 16 : z0;
    if (ni > cap / 3 + 2) ni = cap / 3 + 2;
    nt = y0 < 1 ? 1 : y0 > 32 ? 32 : y0;
    if (nt > cap - ni) nt = cap - ni;
#ifdef PC_STAMPS
    if (const char* e = getenv("MOBOCMF_DEBUG_POTRF_NT")) nt = atoi(e);
    if (const char* e = getenv("MOBOCMF_DEBUG_POTRF_NI")) ni = inverse ? atoi(e) : 0;
#endif
}
// bytes of `sync` the one-launch form needs for nz layers
int64_t potrf_sync_bytes(int nz) { return (int64_t)nz * PC_WORDS * 8; }

int launch_potrf_z(double* A, int64_t ld, int Mp, int M, double* Dinv, double* Ld, int32_t* const* info, int nz, int64_t zs,
                   double* zero0, double* zero1, void* sync, int* inverse_done, hipStream_t s) {
    // nz layers (same M): layer z works on A + z*zs, Dinv + z*zs, Ld + z*zs (doubles) and reports through info[z]
    // sync: potrf_sync_bytes(nz) of device memory for the one-launch form (tune().potrf_cols == 0), or null: one launch pair per step
    // inverse_done (may be null): the caller wants L^-1 in zero0 (Mp x Mp, ld Mp) next and would call launch_trtri_z for it;
    // set to 1 when the one-launch form has formed it already (then no launch_trtri_z), else to 0
    if (inverse_done) *inverse_done = 0;
    const int nblk = Mp / NB;
    const int nreal = (M + NB - 1) / NB;          // 64-blocks that hold real rows
    InfoZ iz = {};
    for (int z = 0; z < nz; ++z) iz.p[z] = info[z];      // (re)set by the first panel
    bool one_launch = false;
    if (sync && tune().potrf_cols == 0 && nreal >= 3 && nreal <= 16) {
        double* Linv = inverse_done && zero0 ? zero0 : nullptr;
        int NT = 1, NI = 0;
        potrf_coop_workgroups(nreal, nz, Linv != nullptr, NT, NI);
        const int G = 1 + NT + NI;
        const size_t shm = (size_t)PC_LDS_DOUBLES * sizeof(double);
        static std::atomic<uint64_t> granted{0};      // one write-once bit per device: the dynamic-LDS attribute was set there
        int devid = 0, cus = 0;
        HIP_TRY(hipGetDevice(&devid));
        const uint64_t bit = devid >= 0 && devid < 64 ? 1ull << devid : 0ull;
        if (!(granted.load() & bit)) {
            HIP_TRY(hipFuncSetAttribute((const void*)potrf_coop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            granted.fetch_or(bit);
        }
        // every workgroup of the launch waits for its peers inside the launch: all of them must be resident at once (an
        // ordinary launch, not a cooperative one: the bound is checked here, the waits inside are bounded)
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)potrf_coop_kernel, PC_T, shm));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid));
        if (per_cu >= 1 && (int64_t)G * nz <= (int64_t)per_cu * cus) {
            HIP_TRY(hipMemsetAsync(sync, 0, (size_t)potrf_sync_bytes(nz), s));
            hipLaunchKernelGGL(potrf_coop_kernel, dim3(G, nz), dim3(PC_T), shm, s, A, ld, nreal, Dinv, Ld, iz, zs,
                               (unsigned long long*)sync, Linv, Mp / NB, NT);
            one_launch = true;
            if (Linv) {
                *inverse_done = 1;
                zero0 = nullptr;      // (written in full by the launch: not cleared below)
            }
        }
    }
    for (int jb = 0; jb < nreal && !one_launch; ++jb) {
        int nact = M - jb * NB;
        nact = nact >= NB ? NB : (nact + 15) & ~15;
        const dim3 grid(nreal - jb, nz);          // blocks below the real rows are zero in these columns and stay zero
        if (nact == 16) hipLaunchKernelGGL(potrf_panel_pad_kernel<16>, grid, dim3(64), 0, s, A, ld, jb, Dinv, Ld, iz, zs);
        else if (tune().potrf_cols == 1)  // (0: the one-launch form where it applies, the four-column panel elsewhere)
            hipLaunchKernelGGL(potrf_panel2_kernel, grid, dim3(128), 0, s, A, ld, jb, Dinv, Ld, iz, zs);
        else hipLaunchKernelGGL(potrf_panel4_kernel, grid, dim3(128), 0, s, A, ld, jb, Dinv, Ld, iz, zs);
        int nt = nreal - jb - 1;
        if (nt > 0) hipLaunchKernelGGL(syrk64_update_kernel, dim3(nt, nt, nz), dim3(256), 0, s, A, ld, jb, zs);
    }
    (void)nblk;
    int64_t n2 = (int64_t)Mp * Mp;
    hipLaunchKernelGGL(finish_l_kernel, dim3((unsigned)((n2 + 255) / 256), 1, nz), dim3(256), 0, s, A, ld, Mp, Ld, Dinv, nreal,
                       zs, zero0, zero1);
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}

int launch_potrf(double* A, int64_t ld, int Mp, int M, double* Dinv, double* Ld, int32_t* info, hipStream_t s) {
    int32_t* one[1] = {info};
    return launch_potrf_z(A, ld, Mp, M, Dinv, Ld, one, 1, 0, nullptr, nullptr, nullptr, nullptr, s);
}

// ---------------------------------------------------------------------------------- triangular inverse
// level 0: the 128x128 diagonal blocks of L^-1 from the 64x64 inverses:  [[D0,0],[-D1 L10 D0, D1]]
__global__ __launch_bounds__(256) void trtri_level0_kernel(const double* L, int64_t ld, const double* Dinv, double* Linv,
                                                           int64_t ldi, int64_t zs) {
    const int b = blockIdx.x;
    L += blockIdx.y * zs; Dinv += blockIdx.y * zs; Linv += blockIdx.y * zs;      // layer batching
    __shared__ double S0[NB * LD64], S1[NB * LD64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    const double* D0 = Dinv + (int64_t)(2 * b) * NB * NB;
    const double* D1 = Dinv + (int64_t)(2 * b + 1) * NB * NB;
    const int64_t o = (int64_t)b * 128;
    load64(S0, L + (o + 64) * ld + o, ld, tid);   // L10
    load64(S1, D0, NB, tid);
    __syncthreads();
    v4f64 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4f64){0, 0, 0, 0};
    mm64<false>(S0, S1, acc, wr, wc, lane);   // T = L10 * D0
    __syncthreads();
    // T -> S0 ; D1 -> S1 ; also write D0 / D1 / zero block to Linv
    {
        const int r = tid >> 2, c0 = (tid & 3) * 16;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            Linv[(o + r) * ldi + o + c0 + c] = S1[r * LD64 + c0 + c];
            Linv[(o + r) * ldi + o + 64 + c0 + c] = 0.0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) S0[(wr * 32 + i * 16 + lk + 4 * r) * LD64 + wc * 32 + j * 16 + li] = acc[i][j][r];
    load64(S1, D1, NB, tid);
    __syncthreads();
    {
        const int r = tid >> 2, c0 = (tid & 3) * 16;
#pragma unroll
        for (int c = 0; c < 16; ++c) Linv[(o + 64 + r) * ldi + o + 64 + c0 + c] = S1[r * LD64 + c0 + c];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4f64){0, 0, 0, 0};
    mm64<false>(S1, S0, acc, wr, wc, lane);   // D1 * T
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Linv[(o + 64 + wr * 32 + i * 16 + lk + 4 * r) * ldi + o + wc * 32 + j * 16 + li] = -acc[i][j][r];
}

// Linv (Mp x Mp, pre-zeroed above the block diagonal by the caller) = L^-1.  T = scratch Mp x Mp.
// Power-of-two block counts: recursive doubling, [[A,0],[B,C]]^-1 = [[A^-1,0],[-C^-1 B A^-1, C^-1]], every level
// is two (batched) MFMA GEMMs; otherwise block row by block row.
int launch_trtri_z(const double* L, int64_t ld, int Mp, const double* Dinv, double* Linv, double* T, double* ws,
                   int64_t ws_elems, int nz, int64_t zs, hipStream_t s) {
    // nz layers: every operand of layer z is the layer-0 pointer + z*zs (doubles)
    const int nb = Mp / TILE;
    const int zl = nz > 1 ? nz : 0;
    auto layered = [&](GemmArgs& g) { g.zlayers = zl; g.zsA = g.zsB = g.zsC = zs; };
    hipLaunchKernelGGL(trtri_level0_kernel, dim3(nb, nz), dim3(256), 0, s, L, ld, Dinv, Linv, (int64_t)Mp, zs);
    if ((nb & (nb - 1)) == 0) {
        for (int sz = TILE; sz < Mp; sz *= 2) {
            const int nmerge = Mp / (2 * sz);
            const int64_t bs = (int64_t)2 * sz * Mp + 2 * sz;     // origin stride between merges (Linv / T, ld Mp)
            GemmArgs g = {};
            g.A = L + (int64_t)sz * ld;  g.lda = ld;              // L21
            g.B = Linv;                  g.ldb = Mp;              // Linv11 (lower)
            g.C = T + (int64_t)sz * Mp;  g.ldc = Mp;
            g.Mr = sz; g.Nc = sz; g.Kd = sz; g.tri = TRI_LOWER_B; g.alpha = 1.0;
            GemmArgs h = {};
            h.A = Linv + (int64_t)sz * Mp + sz; h.lda = Mp;       // Linv22 (lower)
            h.B = T + (int64_t)sz * Mp;         h.ldb = Mp;
            h.C = Linv + (int64_t)sz * Mp;      h.ldc = Mp;
            h.Mr = sz; h.Nc = sz; h.Kd = sz; h.tri = TRI_LOWER_A; h.alpha = -1.0;
            layered(g);
            layered(h);
            int rc;
            if (nmerge > 1) {
                g.batched = h.batched = nmerge;
                g.strideA = (int64_t)2 * sz * ld + 2 * sz; g.strideB = bs; g.strideC = bs;
                h.strideA = bs; h.strideB = bs; h.strideC = bs;
                rc = launch_gemm(g, false, 1, s);
                if (rc) return rc;
                rc = launch_gemm(h, false, 1, s);
            } else {
                rc = launch_gemm_auto(g, false, ws, ws_elems, s);
                if (rc) return rc;
                rc = launch_gemm_auto(h, false, ws, ws_elems, s);
            }
            if (rc) return rc;
        }
        return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
    }
    for (int i = 1; i < nb; ++i) {
        // T[128 x i*128] = L[i, 0:i] * Linv[0:i, 0:i]       (B lower triangular)
        GemmArgs g = {};
        g.A = L + (int64_t)i * TILE * ld;
        g.lda = ld;
        g.B = Linv;
        g.ldb = Mp;
        g.C = T;
        g.ldc = Mp;
        g.Mr = TILE;
        g.Nc = (int64_t)i * TILE;
        g.Kd = (int64_t)i * TILE;
        g.tri = TRI_LOWER_B;
        g.alpha = 1.0;
        layered(g);
        int rc = launch_gemm_auto(g, false, ws, ws_elems, s);
        if (rc) return rc;
        // Linv[i, 0:i] = -Linv[i,i] * T
        GemmArgs h = {};
        h.A = Linv + (int64_t)i * TILE * Mp + (int64_t)i * TILE;
        h.lda = Mp;
        h.B = T;
        h.ldb = Mp;
        h.C = Linv + (int64_t)i * TILE * Mp;
        h.ldc = Mp;
        h.Mr = TILE;
        h.Nc = (int64_t)i * TILE;
        h.Kd = TILE;
        h.alpha = -1.0;
        layered(h);
        rc = zl ? launch_gemm_auto(h, false, ws, ws_elems, s) : launch_gemm(h, false, 1, s);
        if (rc) return rc;
    }
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}

int launch_trtri(const double* L, int64_t ld, int Mp, const double* Dinv, double* Linv, double* T, double* ws,
                 int64_t ws_elems, hipStream_t s) {
    return launch_trtri_z(L, ld, Mp, Dinv, Linv, T, ws, ws_elems, 1, 0, s);
}
