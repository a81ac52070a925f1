// The whole ELBO step of a MID-SIZE surrogate (32 < M <= 128 inducing points; any M <= 128 is accepted) in ONE launch by SEVERAL
// workgroups per surrogate (include/mobocmf_hip.h: mobocmf_coop_elbo_step), gfx950.
//
// tiny_step.hip runs a step as barrier-separated phases of ONE workgroup with element-parallel products: beyond M = 32 a
// product is thousands of dependent FMAs per thread and the chain state no longer fits a CU's LDS.  Here k workgroups share
// one surrogate; zero_grad + MFDGP.forward (mfdgp.py:174-196) + VariationalELBOMF (variational_elbo_mf.py:24-51) + backward +
// Adam (blackbox_mfdgp_fitter.py:161-171) are ~11 phases separated by an in-launch barrier of those k workgroups (an arrival
// counter in device memory + agent-scope fences: ~1-3 us, against ~4.7 us per kernel boundary and 57 launches per step on the
// layer path).  The algebra is DESIGN.md 1 line for line; every product runs on v_mfma_f64_16x16x4_f64 with one
// 16 x 16 output tile per wavefront.  Three kinds of phases:
//   * the M x M chain forward of a layer (K_mm, blocked Cholesky, triangular inverse, U = L^-1 L_S, a = L^-1 m, KL) by ONE
//     workgroup per layer, L and L^-1 held in LDS as swizzled 16 x 16 tiles of the lower triangle (2 x 72 KB at M = 128), the
//     16 x 16 diagonal blocks factorised row-per-lane in registers by one wavefront (tiny_step.hip's chol_inv_wave pattern);
//   * column-block phases: a workgroup owns 16 columns of a layer's M x N' panel and runs Gram -> A = L^-1 K -> C = U^T A ->
//     moments (forward) or dA -> dK = L^-T dA -> Gram backward (backward) on them without leaving LDS -- the accumulator
//     layout of the 16x16x4 instruction is written to an [k][16] LDS block, which is the next product's B operand;
//   * tile-parallel phases for everything M x M in the backward (the weighted syrk H = A diag(gv) A^T, k-sliced; the chain
//     backward's products), operands read from L2 in k-major form (4 rows x 128 contiguous bytes per load instruction), which
//     is why L^-1, U are kept in both orientations.
// Rows ordered by descending fidelity, layer l on the first rows[l] of them, as the one-workgroup kernel.
// Modes (do_update): 0 gradients only, 1 the step, 2 forward only, 4 the conditioned iteration in one launch (all models'
// workgroups meet once more after the forward and form the theta / omega factor gradients, as tiny_step.hip's mode 4).
#include <atomic>

#include "common.h"
#include "small_step_common.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int CT = 256;                          // threads per workgroup
constexpr int CNW = CT / 64;
constexpr int CMAXM = MOBOCMF_COOP_MAX_M;
constexpr int KSMAX = 4;                         // k-slices of the weighted syrk, at most
constexpr int XLD = 17;                          // leading dimension of an [Mp][16] column block in LDS
constexpr int NMAT = 7 + 2 * KSMAX;              // M x M matrices of a layer in `work`
constexpr int PHEAD = 4;                         // leading scalars of a column block's partial record
constexpr int SPIN_LIMIT = 1 << 18;

struct CGeom {
    int L, M, Mp, nt, ntri, d, S;
    int ncol[TLM], ncp[TLM], ncb[TLM], H[TLM], ks[TLM];
    int64_t flat_off[TLM], flat_noise, flat_len;
    int64_t mat[TLM], pan[TLM], vec[TLM], sml[TLM], part[TLM], hpart[TLM], cpl_off, work_len;
    int pstr;
};
__host__ __device__ inline void cgeom_of(const mobocmf_tiny_model& md, CGeom& g) {
    g.L = md.L; g.M = md.M; g.d = md.d; g.S = md.S;
    g.Mp = (md.M + 15) & ~15;
    g.nt = g.Mp / 16;
    g.ntri = g.nt * (g.nt + 1) / 2;
    g.pstr = PHEAD + HS + 2 * g.Mp;
    int64_t fo = 0;
    for (int l = 0; l < TLM; ++l) {
        g.ncol[l] = l < md.L ? md.rows[l] * (l ? md.S : 1) : 0;
        g.ncp[l] = (g.ncol[l] + 15) & ~15;
        g.ncb[l] = g.ncp[l] / 16;
        g.H[l] = l == 0 ? 1 + md.d : 5 + 2 * md.d;
        int ks = (g.ncp[l] + 255) / 256;      // ~256 columns (64 MFMAs) per k-slice
        g.ks[l] = ks < 1 ? 1 : (ks > KSMAX ? KSMAX : ks);
        g.flat_off[l] = fo;
        if (l < md.L) fo += g.H[l] + md.M + (int64_t)md.M * md.M;
    }
    g.flat_noise = fo;
    g.flat_len = fo + md.L;
    int64_t wo = (g.flat_len + 15) & ~(int64_t)15;
    const int64_t mm = (int64_t)g.Mp * g.Mp;
    for (int l = 0; l < TLM; ++l) {
        g.mat[l] = wo;
        if (l < md.L) wo += NMAT * mm;
        g.pan[l] = wo;
        wo += 2 * (int64_t)g.ncp[l] * g.Mp;
        g.vec[l] = wo;
        wo += (int64_t)NVEC * g.ncp[l];
        g.sml[l] = wo;
        if (l < md.L) wo += 5 * g.Mp + 16;
        g.part[l] = wo;
        wo += (int64_t)g.ncb[l] * g.pstr;
        g.hpart[l] = wo;
        if (l < md.L) wo += (int64_t)g.nt * HS;
        wo = (wo + 15) & ~(int64_t)15;
    }
    g.cpl_off = wo;
    g.work_len = wo + CPL_DOUBLES;
}
// matrices of a layer (offsets in units of Mp * Mp from mat[l])
enum { M_L = 0, M_LI = 1, M_LIT = 2, M_U = 3, M_UT = 4, M_G1 = 5, M_GT = 6, M_HS = 7, M_HCS = 7 + KSMAX,
       M_Y = M_HS, M_T4T = M_G1, M_T5 = M_GT };
// small vectors of a layer (offsets in units of Mp from sml[l]; the KL sits behind them)
enum { S_AV = 0, S_DAV = 1, S_DAT = 2, S_GMA = 3, S_GMB = 4 };
// per-column vectors of a layer (units of ncp from vec[l])
enum { V_F = 0, V_EPS = 1, V_MEAN = 2, V_VAR = 3, V_KNN = 4, V_Q = 5, V_RAW = 6, V_GMU = 7, V_GV = 8, V_CGV = 9, V_GF = 10 };

__device__ __forceinline__ v4d mfma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// ---- swizzled 16 x 16 tiles of a lower-triangular matrix in LDS: tile (ti, tj <= ti) at tix * 256, element (r, c) at
// r * 16 + (c ^ r): rows, columns, and the matrix instruction's operand fragments of a tile are all read without bank conflicts
__device__ __forceinline__ int tix(int ti, int tj) { return (ti * (ti + 1) / 2 + tj) * 256; }
__device__ __forceinline__ int tel(int r, int c) { return r * 16 + (c ^ r); }

// Cholesky + inverse of the 16 x 16 tile T (swizzled; lower part used) by the lanes 0..15 of one wavefront: lane i holds row i,
// right-looking, multipliers broadcast by v_readlane (tiny_step.hip chol_inv_wave).  T receives L (zeros above the diagonal),
// Ti its inverse.  Returns the 1-based failed pivot or 0 (wave-uniform).
__device__ int chol_inv_tile16(double* T, double* Ti, int lane) {
    double row[16];
    const int ln = lane & 15;
#pragma unroll
    for (int k = 0; k < 16; ++k) row[k] = T[tel(ln, k)];
    int fail = 0;
    double rr[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double djj = rdlane(row[j], j);
        if (!(djj > 0.0) && !fail) fail = j + 1;
        double r = __builtin_amdgcn_rsq(djj);
        r = r * (1.5 - (0.5 * djj) * r * r);
        r = __builtin_fma(0.5 * r, __builtin_fma(-(djj * r), r, 1.0), r);
        rr[j] = r;
        const double lij = row[j] * r;
        row[j] = lij;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) row[k] -= lij * rdlane(lij, k);
    }
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double s = i == ln ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) s -= rdlane(row[k], i) * x[k];
        x[i] = s * rr[i];
    }
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            T[tel(lane, k)] = k <= lane ? row[k] : 0.0;
            Ti[tel(k, lane)] = x[k];
        }
    }
    return fail;
}

// The in-launch barrier of the k workgroups of one surrogate (or of the whole grid): monotonic arrival counter, agent-scope
// fences on both sides (L2 write-back before arriving, invalidate after leaving: the workgroups sit on different XCDs).
// A wait that does not end (a workgroup of the launch not resident) is abandoned: returns false, the caller leaves the kernel.
__device__ __forceinline__ bool group_barrier(unsigned long long* cnt, unsigned n, int* flag_lds) {
    __syncthreads();
    if (n > 1) {
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned long long old = atomicAdd(cnt, 1ull), target = (old / n + 1ull) * n;
            int spins = 0, ok = 1;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_LIMIT) { ok = 0; break; }
            }
            __threadfence();
            *flag_lds = ok;
        }
        __syncthreads();
        return *flag_lds != 0;
    }
    return true;
}

// One 16 x 16 tile of T X for a column block X [k][XLD] in LDS and a matrix T given k-major in global memory (Tk[k * ld + row]
// = T[row][k]): k tiles kt0 .. kt1-1.  The 4 x 16 fragments of T are fetched one k tile ahead of their use.
__device__ __forceinline__ v4d tile_tx(const double* Tk, int ld, const double* X, int t, int kt0, int kt1, int lane) {
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    if (kt0 >= kt1) return acc;
    const int li = lane & 15, lk = lane >> 4;
    const double* tp = Tk + (int64_t)lk * ld + t * 16 + li;
    const double* xp = X + lk * XLD + li;
    double a[4], an[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = tp[(int64_t)(kt0 * 16 + 4 * q) * ld];
    for (int kt = kt0; kt < kt1; ++kt) {
        const int kn = kt + 1 < kt1 ? kt + 1 : kt;
#pragma unroll
        for (int q = 0; q < 4; ++q) an[q] = tp[(int64_t)(kn * 16 + 4 * q) * ld];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = mfma(a[q], xp[(kt * 16 + 4 * q) * XLD], acc);
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = an[q];
    }
    return acc;
}
__device__ __forceinline__ void store_x(double* X, int t, int lane, v4d acc) {      // accumulator tile t -> X[k][XLD]
    const int lj = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) X[(t * 16 + 4 * r + lk) * XLD + lj] = acc[r];
}
// row tiles of a column-block product dealt to the CNW wavefronts in pairs (t, nt-1-t): a triangular T gives every pair the same
// number of k tiles
__device__ __forceinline__ int pair_tile(int idx, int nt) { return (idx & 1) ? nt - 1 - (idx >> 1) : (idx >> 1); }

// the row tiles a wavefront takes in a column-block product with nt row tiles (at most 2 for nt <= 8)
__device__ __forceinline__ int wave_tiles(int wave, int nt, int (&t)[2]) {
    if (nt <= CNW) { t[0] = wave; return wave < nt ? 1 : 0; }
    const int p = wave;
    if (p >= (nt + 1) / 2) return 0;
    t[0] = p;
    t[1] = nt - 1 - p;
    return t[1] != p ? 2 : 1;
}
__device__ __forceinline__ void tile_of(int t, int& ti, int& tj) {      // index in the packed lower triangle -> (ti, tj <= ti)
    ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    tj = t - ti * (ti + 1) / 2;
}

constexpr int XFW = DBT + 2;      // a staged data row of a column block: x (zero padded), f, valid flag

//KERNEL//

}  // namespace
