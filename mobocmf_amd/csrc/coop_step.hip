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
// Modes (do_update): 0 gradients only, 1 the step, 2 forward only, 3 input gradients (acquisition search: `grad` <- d/dx of the
// seeded top-layer moments, nothing else written), 4 the conditioned iteration in one launch (all models'
// workgroups meet once more after the forward and form the theta / omega factor gradients, as tiny_step.hip's mode 4).
#include <atomic>

#include "common.h"
#include "small_step_common.h"
#include "tile16.h"

namespace {


constexpr int CT = 512;                          // threads per workgroup: two wavefronts per SIMD (the phases are latency-bound)
constexpr int CNW = CT / 64;
constexpr int CMAXM = MOBOCMF_COOP_MAX_M;
constexpr int KSMAX = 8;                         // slabs of the weighted syrk H (k-slices, or one per column block), at most
constexpr int XLD = 17;                          // leading dimension of an [Mp][16] column block in LDS
constexpr int NMAT = 7 + 2 * KSMAX;              // M x M matrices of a layer in `work`
constexpr int PHEAD = 4;                         // leading scalars of a column block's partial record
constexpr int SPIN_LIMIT = 1 << 21;      // polls of an in-launch barrier before it is abandoned (~1.5 s): the wait is for peers of the SAME launch,
                                         // which are resident (checked on the host) -- it ends unless the counters were tampered with

struct CGeom {
    int L, M, Mp, nt, ntri, d, S;
    int ncol[TLM], ncp[TLM], ncb[TLM], H[TLM], ks[TLM], hblk[TLM];
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
        // H = A diag(gv) A^T: a layer of few column blocks (the reference's own sizes: N' = N <= 75 + Pareto points) forms it
        // block by block inside the backward column phase, one slab per block -- no syrk phase, no barrier for it; a wider
        // one k-slices it over the whole surrogate in a phase of its own, ~256 columns (64 MFMAs of each product) per slice
        g.hblk[l] = g.ncb[l] <= KSMAX ? 1 : 0;
        int ks = (g.ncp[l] + 255) / 256;
        g.ks[l] = g.hblk[l] ? (g.ncb[l] < 1 ? 1 : g.ncb[l]) : (ks < 1 ? 1 : (ks > 4 ? 4 : ks));
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
        if (l < md.L) wo += 6 * g.Mp + HS + 24;
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
// small vectors of a layer (offsets in units of Mp from sml[l]): a, da, da_tot, L^-T da_tot, the K_mm Gram backward's share of
// g_m[l-1], the column blocks' summed d/dzf rows (S_GZ); behind them the KL (1), and the summed hyper-parameter partials + the
// noise-gradient sum (RED: HS + 1)
enum { S_AV = 0, S_DAV = 1, S_DAT = 2, S_GMA = 3, S_GMB = 4, S_GZ = 5 };
// per-column vectors of a layer (units of ncp from vec[l])
enum { V_F = 0, V_EPS = 1, V_MEAN = 2, V_VAR = 3, V_KNN = 4, V_Q = 5, V_RAW = 6, V_GMU = 7, V_GV = 8, V_CGV = 9, V_GF = 10 };

// global-memory views of generic pointers (global_load / global_store instead of flat accesses; and the compiler then knows
// that they cannot alias LDS, so the loads of a batch are issued together instead of one per LDS store)
typedef const __attribute__((address_space(1))) double* gcd;
typedef __attribute__((address_space(1))) double* gwd;
#define GC(p) ((gcd)(p))
#define GW(p) ((gwd)(p))

// n elements dealt to the CT threads, B per thread at a time: all B loads are issued before the first store (a loop that loads,
// waits and stores element by element pays one L2 round trip per element: ~1 us each right after a barrier's invalidate)
template <int B, class LoadF, class StoreF>
__device__ __forceinline__ void batched(int n, int tid, LoadF ld, StoreF st) {
    for (int base = tid; base < n; base += CT * B) {
        double v[B];
#pragma unroll
        for (int b = 0; b < B; ++b) { const int e = base + b * CT; v[b] = ld(e < n ? e : n - 1); }
#pragma unroll
        for (int b = 0; b < B; ++b) { const int e = base + b * CT; if (e < n) st(e, v[b]); }
    }
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
// = T[row][k]): k tiles kt0 .. kt1-1.  The 4 x 16 fragments of FOUR k tiles are requested together, before their MFMAs: one L2
// round trip per four k tiles (all eight at once -- 64 VGPRs of fragments -- was the register-pressure peak of the column phases
// and put 150 values into scratch memory: with four the kernel needs no scratch at all, M = N = 64: 150 -> 132 us per step).
__device__ __forceinline__ v4d tile_tx(const double* Tk_, int ld, const double* X, int t, int kt0, int kt1, int lane) {
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    const int li = lane & 15, lk = lane >> 4;
    gcd tp = GC(Tk_) + (int64_t)lk * ld + t * 16 + li;
    const double* xp = X + lk * XLD + li;
    for (int kb = kt0; kb < kt1; kb += 4) {
        double a[4][4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int kt = kb + kk < kt1 ? kb + kk : kt1 - 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) a[kk][q] = tp[(int64_t)(kt * 16 + 4 * q) * ld];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kb + kk < kt1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = mfma(a[kk][q], xp[((kb + kk) * 16 + 4 * q) * XLD], acc);
            }
        }
    }
    return acc;
}
__device__ __forceinline__ double xor_add(double v, int mask) { return v + __shfl_xor(v, mask); }
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
__device__ __forceinline__ void tile_of_slow(int t, int& ti, int& tj) {      // index in the packed lower triangle -> (ti, tj <= ti)
    ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    tj = t - ti * (ti + 1) / 2;
}
#define tile_of(t, ti, tj) do { const int tm_ = tmap[t]; ti = tm_ >> 4; tj = tm_ & 15; } while (0)

constexpr int XFW = DBT + 5;      // a staged data row of a column block: x (zero padded), f, valid flag, y, fidelity, row weight


// ---- Every phase is a function of its own with its own scope: as ONE 1400-line body the kernel made the compiler hoist the
// address arithmetic of all phases to the top of every loop nest and spill it (hundreds of scratch round trips, ~15 us in front
// of a phase).  What the phases share lives in LDS: the surrogate's descriptor, its geometry, and this context.
// (PHASE_FN: inlined.  As functions of their own the phases each saved and restored ~110 callee-saved VGPRs through scratch
// memory at entry and exit -- 4-7 us per call; inlined, they still get their own scopes, and nothing of one phase is live in
// the next: what they share is re-read from LDS.)
#define PHASE_FN __device__ __forceinline__
// a wave-uniform value read from LDS lands in a VGPR (two for a pointer or a double); through v_readfirstlane it lives in SGPRs
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
    u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
    return u.d;
}
template <class T>
__device__ __forceinline__ T* uni(T* p) {
    union { T* p; int i[2]; } u;
    u.p = p;
    u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
    u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
    return u.p;
}
struct Ctx {
    const mobocmf_tiny_model* models;
    double* lds;                        // behind descriptor, geometry and context
    unsigned long long *mcnt, *gcnt;    // arrival counters: this surrogate's workgroups / the whole grid
    double lr, b1, b2, aeps;
    int k, wj, do_update, n_stamp;
};
constexpr int MDW = (sizeof(mobocmf_tiny_model) + 15) / 16 * 2;      // (doubles)
constexpr int GEW = (sizeof(CGeom) + 15) / 16 * 2;
constexpr int CXW = (sizeof(Ctx) + 15) / 16 * 2;

#ifdef COOP_STAMPS
// (diagnostic build: workgroup 0 of a surrogate writes (id, 100 MHz wall clock) pairs behind its workspace; tools/coop_stamps.py)
#define CSTAMP(id) do { if (tid == 0) { if (wj == 0 && cx->n_stamp < 120) { double* st_ = md.work + g.work_len; st_[2 * cx->n_stamp] = (double)(id); st_[2 * cx->n_stamp + 1] = (double)wall_clock64(); } cx->n_stamp += 1; } } while (0)
#else
#define CSTAMP(id) do { } while (0)
#endif

// the locals every phase starts from (what it does not use costs nothing)
#define CTX_LOCALS                                                                                                        \
    extern __shared__ __attribute__((aligned(16))) double lds_all[];                                                      \
    const mobocmf_tiny_model& md = *(const mobocmf_tiny_model*)lds_all;                                                   \
    const CGeom& g = *(const CGeom*)(lds_all + MDW);                                                                      \
    Ctx* cx = (Ctx*)(lds_all + MDW + GEW);                                                                                \
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;                       \
    const int k = uni(cx->k), wj = uni(cx->wj), do_update = uni(cx->do_update);                                           \
    const mobocmf_tiny_model* models = uni(cx->models);                                                                   \
    const int L = uni(g.L), M = uni(g.M), Mp = uni(g.Mp), nt = uni(g.nt), d = uni(g.d), S = uni(g.S);                     \
    const int64_t mm = (int64_t)Mp * Mp;                                                                                  \
    double* W = uni(md.work);                                                                                             \
    double* gflat = W;                                                                                                    \
    double* hy = lds_all + MDW + GEW + CXW;        /* [TLM][HS] constrained hyper-parameters (derived from the __shared__ symbol: a pointer read back from memory would be a generic one and every LDS access a flat one) */                           \
    double* il = hy + TLM * HS;                    /* [TLM][2 DBT] inverse lengthscales */                                \
    double* sc = il + TLM * 2 * DBT;               /* [32]: tau[l] 0..2, log tau 8..10, Adam's bias terms 12, 13, bsum scratch 16..23, barrier flag 30 */ \
    double** seg_ptr = (double**)(sc + 32);        /* [NSEG] parameter tensors in flat-vector order (trainable ones; else null) */ \
    int* seg_end = (int*)(seg_ptr + NSEG);         /* [NSEG] */                                                           \
    int* tmap = seg_end + NSEG;                    /* [40] packed lower-triangle index -> (ti << 4) | tj */               \
    double* big = sc + 32 + NSEG + NSEG / 2 + 20;                                                                         \
    double* ztc = big;                             /* chain view: [Mp][ZW] Z~ of the layer; later m and diag(L) */        \
    double* Lp = ztc + Mp * ZW;                    /* packed swizzled tiles of L (later: of tril(L_S)) */                 \
    double* Lip = Lp + g.ntri * 256;               /* ... of L^-1 */                                                      \
    double* zta = big;                             /* column view: [L][Mp][ZW] Z~ of every layer */                       \
    double* X0 = zta + L * Mp * ZW;                /* three [Mp][XLD] column blocks */                                    \
    double* X1 = X0 + Mp * XLD;                                                                                           \
    double* X2 = X1 + Mp * XLD;                                                                                           \
    double* ava = X2 + Mp * XLD;                   /* [L][Mp] a = L^-1 m of every layer */                                \
    double* xf = ava + L * Mp;                     /* [16][XFW] the block's data rows */                                  \
    double* gcol = xf + 16 * XFW;                  /* [4][16] per-column scalars of the block */                          \
    double* red = gcol + 64;                       /* [CT / 16][16][3] partial column sums */                             \
    double* redh = red + (CT / 16) * 16 * 3;       /* [CNW][HS + 1] wavefront partials */                                 \
    const double gkl = uni(md.kl_scale), ge = -1.0;                                                                       \
    const double lr = uni(cx->lr), b1 = uni(cx->b1), b2 = uni(cx->b2), aeps = uni(cx->aeps);                              \
    auto MAT = [&](int l, int m) -> double* { return W + uni((int)g.mat[l]) + (int64_t)m * mm; };                         \
    auto VEC = [&](int l, int v) -> double* { return W + uni((int)g.vec[l]) + (int64_t)v * uni(g.ncp[l]); };              \
    auto SML = [&](int l, int v) -> double* { return W + uni((int)g.sml[l]) + (int64_t)v * Mp; };                         \
    auto PART = [&](int l, int cb) -> double* { return W + uni((int)g.part[l]) + (int64_t)cb * uni(g.pstr); };            \
    (void)lane; (void)wave; (void)li; (void)lk; (void)k; (void)wj; (void)do_update; (void)models; (void)L; (void)M;      \
    (void)nt; (void)d; (void)S; (void)mm; (void)gflat; (void)hy; (void)il; (void)sc; (void)seg_ptr; (void)seg_end;        \
    (void)tmap; (void)ztc; (void)Lp; (void)Lip; (void)zta; (void)X0; (void)X1; (void)X2; (void)ava; (void)xf; (void)gcol; \
    (void)red; (void)redh; (void)gkl; (void)ge; (void)lr; (void)b1; (void)b2; (void)aeps; (void)MAT; (void)VEC;          \
    (void)SML; (void)PART

#define FRAG(dst, P, k0, x0) _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) dst[q_] = GC(P)[(int64_t)((k0) + 4 * q_ + lk) * Mp + (x0) + li]
#define COPY4(dst, src) _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) dst[q_] = src[q_]
#define PIN() __builtin_amdgcn_sched_barrier(0)
#define DEF_STAGE_ROWS \
    auto stage_rows = [&](int l, int c0, bool forward) { \
        const int kind = l > 0, div = l ? S : 1, nc = g.ncol[l]; \
        if (tid < 16) { \
            const int c = c0 + tid; \
            const bool valid = c < nc; \
            double f = 0.0; \
            if (kind && valid) { \
                if (forward) { \
                    const int fdiv = l == 1 ? S : 1, cp = c / fdiv; \
                    const int64_t* rng = md.rng[l]; \
                    const double ev = md.eps[l] ? md.eps[l][c] : philox_normal((uint64_t)rng[0], (uint64_t)rng[1], (uint64_t)c); \
                    f = VEC(l - 1, V_MEAN)[cp] + sqrt(VEC(l - 1, V_VAR)[cp]) * ev; \
                    VEC(l, V_F)[c] = f; \
                    VEC(l, V_EPS)[c] = ev; \
                } else { \
                    f = VEC(l, V_F)[c]; \
                } \
            } \
            const int b = valid ? c / div : 0; \
            xf[tid * XFW + DBT] = f; \
            xf[tid * XFW + DBT + 1] = valid ? 1.0 : 0.0; \
            xf[tid * XFW + DBT + 2] = md.y[b]; \
            xf[tid * XFW + DBT + 3] = md.fid[b]; \
            xf[tid * XFW + DBT + 4] = md.row_weight ? md.row_weight[b] : 1.0; \
        } \
        if (tid >= 64 && tid < 64 + 16 * DBT) { \
            const int j = (tid - 64) / DBT, kk = (tid - 64) % DBT, c = c0 + j; \
            xf[j * XFW + kk] = (c < nc && kk < d) ? md.x[(int64_t)(c / div) * d + kk] : 0.0; \
        } \
    };


PHASE_FN void ph_setup() {
    CTX_LOCALS;
    // ---- P0 (every workgroup for itself): constrained hyper-parameters, inverse lengthscales, noise, parameter table
    for (int e = tid; e < TLM * HS; e += CT) {
        const int l = e / HS, t = e % HS;
        double v = 0.0;
        if (l < L && t < g.H[l]) {
            int s = 0, off = 0;
            while (t >= off + seg_len(l, s, d)) { off += seg_len(l, s, d); ++s; }
            const double x = md.raw[l][s][t - off];
            v = x > 20.0 ? x : log1p(exp(x));
        }
        hy[e] = v;
    }
    if (tid >= 64 && tid < 64 + NSEG) {
        const int kk = tid - 64;
        double* ptr = nullptr;
        int end = 0x7fffffff, j = 0;
        for (int l = 0; l < L; ++l) {
            const uint32_t tr = md.trainable[l];
            const int ns = l == 0 ? 2 : 7;
            int off = (int)g.flat_off[l];
            for (int s2 = 0; s2 < ns; ++s2, ++j) {
                off += seg_len(l, s2, d);
                if (j == kk) { ptr = ((tr >> s2) & 1u) ? md.raw[l][s2] : nullptr; end = off; }
            }
            off += M;
            if (j++ == kk) { ptr = ((tr >> 7) & 1u) ? md.m[l] : nullptr; end = off; }
            off += M * M;
            if (j++ == kk) { ptr = ((tr >> 8) & 1u) ? md.L_S[l] : nullptr; end = off; }
        }
        for (int l = 0; l < L; ++l)
            if (j++ == kk) { ptr = ((md.trainable[l] >> 9) & 1u) ? md.raw_noise[l] : nullptr; end = (int)g.flat_noise + l + 1; }
        seg_ptr[kk] = ptr;
        seg_end[kk] = end;
    }
    if (tid >= 128 && tid < 128 + 40) {
        int ti = 0, tj = 0;
        if (tid - 128 < g.ntri) tile_of_slow(tid - 128, ti, tj);
        tmap[tid - 128] = (ti << 4) | tj;
    }
    if (tid == CT - 1 && (do_update == 1 || do_update == 4)) {
        const double step = (double)(md.steps_done[0] + 1);
        sc[12] = 1.0 - pow(b1, step);
        sc[13] = sqrt(1.0 - pow(b2, step));
    }
    if (tid < L) {
        const double lo = md.noise_lo[tid], hi = md.noise_hi[tid], r = md.raw_noise[tid][0];
        sc[tid] = hi > lo ? lo + (hi - lo) / (1.0 + exp(-r)) : r;
        sc[8 + tid] = log(sc[tid]);
    }
    if ((do_update == 2 || do_update == 4) && md.xrng && md.rand_rows > 0 && wj == 0) {
        // the x~ of this iteration (blackbox_mfdgp_fitter.py:276): every model of the launch draws the SAME points
        const uint64_t seed = (uint64_t)md.xrng[0], call = (uint64_t)md.xrng[1];
        double* xw = const_cast<double*>(md.x) + (int64_t)md.rand_row0 * d;
        for (int e = tid; e < md.rand_rows * d; e += CT) xw[e] = philox_uniform(seed, call, (uint64_t)e);
    }
    __syncthreads();
    for (int e = tid; e < TLM * 2 * DBT; e += CT) {
        const int l = e / (2 * DBT), kk = e % (2 * DBT), k2 = kk % DBT;
        double v = 0.0;
        if (l < L && k2 < d) {
            if (l == 0) v = kk < DBT ? 1.0 / hy[l * HS + 1 + k2] : 0.0;
            else v = 1.0 / hy[l * HS + 5 + (kk < DBT ? 0 : d) + k2];
        }
        il[e] = v;
    }
    __syncthreads();
    CSTAMP(1);
}


// K_mm + jitter I of layer l (lower tiles; identity beyond M): elements first, first + step, ... of the packed tiles, into the
// swizzled LDS tiles (to_lds) or into the row-major matrix L of the workspace
PHASE_FN void kmm_tiles(int l, int first, int step, bool to_lds) {
    CTX_LOCALS;
    const int kind = l > 0;
    const double* hyl = hy + l * HS;
    const double* ill = il + l * 2 * DBT;
    for (int e = tid; e < Mp * ZW; e += CT) ztc[e] = 0.0;
    __syncthreads();
    {
        gcd zx = GC(md.Zx);
        batched<4>(M * d, tid, [&](int e) { return zx[e]; }, [&](int e, double v) { ztc[(e / d) * ZW + e % d] = v; });
        if (l > 0) {
            gcd mp = GC(md.m[l - 1]);
            batched<1>(M, tid, [&](int e) { return mp[e]; }, [&](int e, double v) { ztc[e * ZW + DBT] = v; });
        }
    }
    __syncthreads();
    gwd Kg = GW(MAT(l, M_L));
#pragma unroll 1
    for (int e = first + tid; e < g.ntri * 256; e += step) {
        const int t = e >> 8, r = (e >> 4) & 15, c = e & 15;
        int ti, tj;
        tile_of(t, ti, tj);
        const int i = ti * 16 + r, j = tj * 16 + c;
        double v = 0.0;
        if (i < M && j <= i) {
            KV o;
            kern_eval(kind, d, ztc + i * ZW, ztc[i * ZW + DBT], ztc + j * ZW, hyl, ill, o);
            v = o.k + (i == j ? md.jitter : 0.0);
        } else if (i == j) {
            v = 1.0;
        }
        if (to_lds) Lp[t * 256 + tel(r, c)] = v;
        else Kg[(int64_t)i * Mp + j] = v;
    }
}
// beyond 64 inducing points the K_mm of every layer is formed by ALL workgroups of the surrogate (thousands of covariance
// evaluations, three exponentials each: 14 us on one CU at M = 128) and handed to the chain's workgroup through the workspace
PHASE_FN void ph_kmm() {
    CTX_LOCALS;
    const int l = wj % L, r = wj / L, kl = (k - l + L - 1) / L;      // the kl workgroups wj = l, l + L, ... share layer l
    kmm_tiles(l, r * CT, kl * CT, false);
}

PHASE_FN void ph_chain(bool kmm_done) {
    CTX_LOCALS;
    // ---- P1: the M x M chain forward of layer l by ONE workgroup, L and L^-1 in LDS
    for (int l_it = wj; l_it < L; l_it += k) {
        int l = l_it;
        asm volatile("" : "+s"(l));      // (as in the column phases: the layer's addresses are formed where they are used)
        if (kmm_done) {
            gcd Kg = GC(MAT(l, M_L));
            batched<8>(g.ntri * 256, tid,
                       [&](int e) {
                           int ti, tj;
                           tile_of(e >> 8, ti, tj);
                           return Kg[(int64_t)(ti * 16 + ((e >> 4) & 15)) * Mp + tj * 16 + (e & 15)];
                       },
                       [&](int e, double v) { Lp[(e >> 8) * 256 + tel((e >> 4) & 15, e & 15)] = v; });
        } else {
            kmm_tiles(l, 0, CT, true);
        }
        __syncthreads();
        CSTAMP(2);
        // blocked right-looking Cholesky, 16-wide, with look-ahead: the diagonal tile is factorised (and inverted) row-per-lane in
        // registers by wavefront 0 -- ~3 us of dependent instructions, the critical path -- so in step s wavefront 0 forms ONLY the
        // next pivot block (panel tile (s+1, s), its update of the diagonal tile (s+1, s+1), factorisation) while the other
        // wavefronts run the rest of the panel and of the trailing update on the MFMA
        int fail = 0;
        if (wave == 0) fail = chol_inv_tile16(Lp + tix(0, 0), Lip + tix(0, 0), lane);
        __syncthreads();
        for (int s = 0; s + 1 < nt; ++s) {
            const double* Bi = Lip + tix(s, s);      // L_ss^-1
            if (wave == 0) {
                double* A = Lp + tix(s + 1, s);      // L_{s+1,s} = A_{s+1,s} L_ss^-T
                v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = mfma(A[tel(li, 4 * q + lk)], Bi[tel(li, 4 * q + lk)], acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) A[tel(4 * r + lk, li)] = acc[r];
            } else {
                for (int i = s + 1 + wave; i < nt; i += CNW - 1) {      // the other panel tiles
                    double* A = Lp + tix(i, s);
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = mfma(A[tel(li, 4 * q + lk)], Bi[tel(li, 4 * q + lk)], acc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) A[tel(4 * r + lk, li)] = acc[r];
                }
            }
            __syncthreads();
            if (wave == 0) {
                // the next pivot block: A_{s+1,s+1} -= L_{s+1,s} L_{s+1,s}^T, then its factorisation
                const double* P = Lp + tix(s + 1, s);
                double* D = Lp + tix(s + 1, s + 1);
                v4d acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = D[tel(4 * r + lk, li)];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = mfma(-P[tel(li, 4 * q + lk)], P[tel(li, 4 * q + lk)], acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) D[tel(4 * r + lk, li)] = acc[r];
                const int f = chol_inv_tile16(D, Lip + tix(s + 1, s + 1), lane);
                if (f && !fail) fail = (s + 1) * 16 + f;
            } else {
                const int nrem = nt - s - 1, ntr = nrem * (nrem + 1) / 2;
                for (int u = wave; u < ntr; u += CNW - 1) {      // A_ij -= L_is L_js^T, all tiles but (s+1, s+1) (u = 0)
                    int a, b;
                    tile_of(u, a, b);
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
                }
            }
            __syncthreads();
        }
        if (tid == 0) md.info[l] = fail;
        CSTAMP(3);
        // off-diagonal tiles of L^-1, one block column per wavefront: X_is = -L_ii^-1 sum_{t = s .. i-1} L_it X_ts
        for (int s = wave; s < nt; s += CNW) {
            {
                for (int i = s + 1; i < nt; ++i) {
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
                    for (int t = s; t < i; ++t) {
                        const double* A = Lp + tix(i, t);
                        const double* B = Lip + tix(t, s);
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc = mfma(A[tel(li, 4 * q + lk)], B[tel(4 * q + lk, li)], acc);
                    }
                    const double* Aii = Lip + tix(i, i);
                    v4d d2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < 4; ++q) d2 = mfma(Aii[tel(li, 4 * q + lk)], acc[q], d2);      // the accumulator IS the B fragment
                    double* X = Lip + tix(i, s);
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[tel(4 * r + lk, li)] = -d2[r];
                }
            }
        }
        __syncthreads();
        CSTAMP(4);
        // L, L^-1, L^-T to global (the other workgroups read them k-major); diag(L) and m kept; L's tiles then hold tril(L_S)
        {
            gwd Lg = GW(MAT(l, M_L));
            gwd Lig = GW(MAT(l, M_LI));
            gwd LiTg = GW(MAT(l, M_LIT));
            for (int e = tid; e < g.ntri * 256; e += CT) {
                const int t = e >> 8, r = (e >> 4) & 15, c = e & 15;
                int ti, tj;
                tile_of(t, ti, tj);
                Lg[(int64_t)(ti * 16 + r) * Mp + tj * 16 + c] = Lp[t * 256 + tel(r, c)];
                Lig[(int64_t)(ti * 16 + r) * Mp + tj * 16 + c] = Lip[t * 256 + tel(r, c)];
                LiTg[(int64_t)(tj * 16 + r) * Mp + ti * 16 + c] = Lip[t * 256 + tel(c, r)];
            }
            for (int i = tid; i < Mp; i += CT) {
                ztc[i] = 0.0;
                ztc[Mp + i] = Lp[tix(i >> 4, i >> 4) + tel(i & 15, i & 15)];
            }
        }
        __syncthreads();
        {
            gcd mp = GC(md.m[l]);
            batched<1>(M, tid, [&](int e) { return mp[e]; }, [&](int e, double v) { ztc[e] = v; });
            gcd ls = GC(md.L_S[l]);
            batched<8>(g.ntri * 256, tid,
                       [&](int e) {
                           const int t = e >> 8, r = (e >> 4) & 15, c = e & 15;
                           int ti, tj;
                           tile_of(t, ti, tj);
                           const int i = ti * 16 + r, j = tj * 16 + c;
                           return ls[(int64_t)(i < M ? i : M - 1) * M + (j < M ? j : M - 1)];
                       },
                       [&](int e, double v) {
                           const int t = e >> 8, r = (e >> 4) & 15, c = e & 15;
                           int ti, tj;
                           tile_of(t, ti, tj);
                           const int i = ti * 16 + r, j = tj * 16 + c;
                           Lp[t * 256 + tel(r, c)] = (i < M && j <= i) ? v : 0.0;
                       });
        }
        __syncthreads();
        CSTAMP(5);
        // U = L^-1 L_S (lower tiles), both orientations to global; a = L^-1 m; KL = 1/2 [2 sum log L_ii - sum log L_S,ii^2 + |U|^2 + |a|^2 - M]
        double klacc = 0.0;
        {
            gwd Ug = GW(MAT(l, M_U));
            gwd UTg = GW(MAT(l, M_UT));
            for (int u = wave; u < g.ntri; u += CNW) {
                int ti, tj;
                tile_of(u, ti, tj);
                v4d acc = {0.0, 0.0, 0.0, 0.0};
                for (int tk = tj; tk <= ti; ++tk) {
                    const double* A = Lip + tix(ti, tk);
                    const double* B = Lp + tix(tk, tj);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = mfma(A[tel(li, 4 * q + lk)], B[tel(4 * q + lk, li)], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = ti * 16 + 4 * r + lk, j = tj * 16 + li;
                    Ug[(int64_t)i * Mp + j] = acc[r];
                    UTg[(int64_t)j * Mp + i] = acc[r];
                    klacc += 0.5 * acc[r] * acc[r];
                }
            }
            double* avg = SML(l, S_AV);
            {
                const int i = tid >> 2, part = tid & 3;      // four threads per row (CT / 4 >= Mp)
                double a = 0.0;
                if (i < Mp) {
                    const int ti = i >> 4, r = i & 15;
                    for (int kk = part; kk <= i; kk += 4) a += Lip[tix(ti, kk >> 4) + tel(r, kk & 15)] * ztc[kk];
                }
                a = xor_add(xor_add(a, 1), 2);
                if (i < Mp && part == 0) {
                    const int ti = i >> 4, r = i & 15;
                    avg[i] = a;
                    if (i < M) klacc += 0.5 * a * a + log(ztc[Mp + i] / fabs(Lp[tix(ti, ti) + tel(r, r)])) - 0.5;
                }
            }
        }
        const double kl = bsum<CNW>(klacc, sc + 16);
        if (tid == 0) W[g.sml[l] + 6 * Mp] = kl;
        __syncthreads();
    }
}


PHASE_FN void ph_stage() {
    CTX_LOCALS;
    // ---- forward, layer by layer: a workgroup per block of 16 columns
    // Z~_l = [Z_x, m_{l-1}] and a_l of EVERY layer into LDS, once (parameters and chain results: fixed for the rest of the step)
    for (int e = tid; e < L * Mp * ZW; e += CT) zta[e] = 0.0;
    __syncthreads();
    {
        gcd zx = GC(md.Zx);
        for (int l = 0; l < L; ++l) {
            double* ztl = zta + l * Mp * ZW;
            batched<4>(M * d, tid, [&](int e) { return zx[e]; }, [&](int e, double v) { ztl[(e / d) * ZW + e % d] = v; });
            if (l > 0) {
                gcd mp = GC(md.m[l - 1]);
                batched<1>(M, tid, [&](int e) { return mp[e]; }, [&](int e, double v) { ztl[e * ZW + DBT] = v; });
            }
            gcd avg = GC(SML(l, S_AV));
            double* avl_ = ava + l * Mp;
            batched<1>(Mp, tid, [&](int e) { return avg[e]; }, [&](int e, double v) { avl_[e] = v; });
        }
    }
    __syncthreads();
    CSTAMP(7);
    // the block's data rows: x (zero padded to DBT), f, valid flag; layer >= 1 draws / reads f = mean + sqrt(var) eps
}


PHASE_FN void ph_forward(int l_in) {
    CTX_LOCALS;
    DEF_STAGE_ROWS
        const bool has_tile = wave < nt;
        for (int cb = wj; cb < g.ncb[l_in]; cb += k) {
            // (the layer number goes through an opaque move in every iteration: what depends on it -- a dozen workspace addresses --
            // is then formed where it is used instead of being hoisted in front of the loop and spilled across it)
            int l = l_in;
            asm volatile("" : "+s"(l));
            const int kind = l > 0, div = l ? S : 1, nc = g.ncol[l];
            const double* hyl = hy + l * HS;
            const double* ill = il + l * 2 * DBT;
            const double* zt = zta + l * Mp * ZW;
            const double* avl = ava + l * Mp;
            const double* LiTg = MAT(l, M_LIT);
            const double* Ug = MAT(l, M_U);
            double* ATg = W + g.pan[l];
            double* CTg = ATg + (int64_t)g.ncp[l] * Mp;
            CSTAMP(50);
            const int c0 = cb * 16;
            stage_rows(l, c0, true);
            __syncthreads();
            CSTAMP(51);
            // F1: K block
#pragma unroll 1
            for (int e = tid; e < Mp * 16; e += CT) {
                const int m = e >> 4, j = e & 15;
                double v = 0.0;
                if (m < M && xf[j * XFW + DBT + 1] != 0.0) {
                    KV o;
                    kern_eval(kind, d, xf + j * XFW, xf[j * XFW + DBT], zt + m * ZW, hyl, ill, o);
                    v = o.k;
                }
                X0[m * XLD + j] = v;
            }
            __syncthreads();
            CSTAMP(52);
            // F2: A = L^-1 K
            if (has_tile) store_x(X1, wave, lane, tile_tx(LiTg, Mp, X0, wave, 0, wave + 1, lane));
            __syncthreads();
            CSTAMP(53);
            // F3: C = U^T A
            if (has_tile) store_x(X2, wave, lane, tile_tx(Ug, Mp, X1, wave, wave, nt, lane));
            __syncthreads();
            CSTAMP(54);
            // F4: moments, the block's share of the data term; A, C to global (column-major: the backward's operand layout)
            {
                const int j = tid & 15, part = tid >> 4;
                double q = 0.0, mu = 0.0, r = 0.0;
                for (int m = part; m < Mp; m += CT / 16) {
                    const double a = X1[m * XLD + j], c = X2[m * XLD + j];
                    q += a * a;
                    mu += avl[m] * a;
                    r += c * c;
                }
                q = xor_add(xor_add(q, 16), 32);
                mu = xor_add(xor_add(mu, 16), 32);
                r = xor_add(xor_add(r, 16), 32);
                if (lane < 16) {
                    red[(wave * 16 + lane) * 3 + 0] = q;
                    red[(wave * 16 + lane) * 3 + 1] = mu;
                    red[(wave * 16 + lane) * 3 + 2] = r;
                }
            }
            for (int e = tid; e < 16 * Mp; e += CT) {
                const int j = e / Mp, m = e % Mp;
                GW(ATg)[(int64_t)(c0 + j) * Mp + m] = X1[m * XLD + j];
                GW(CTg)[(int64_t)(c0 + j) * Mp + m] = X2[m * XLD + j];
            }
            __syncthreads();
            CSTAMP(55);
            if (wave == 0) {
                double dterm = 0.0;
                if (lane < 16 && c0 + lane < nc) {
                    const int c = c0 + lane;
                    double q = 0.0, mu = 0.0, r = 0.0;
#pragma unroll
                    for (int p = 0; p < CNW; ++p) {
                        q += red[(p * 16 + lane) * 3 + 0];
                        mu += red[(p * 16 + lane) * 3 + 1];
                        r += red[(p * 16 + lane) * 3 + 2];
                    }
                    const double fn = xf[lane * XFW + DBT];
                    const double knn = kind ? hyl[0] * (hyl[2] * fn * fn + hyl[1]) + hyl[3] : hyl[0];
                    double sres = knn - q;
                    if (!md.branch && sres < 0.0) sres = 0.0;
                    const double vr = sres + r, var = vr < MINV ? MINV : vr;
                    GW(VEC(l, V_MEAN))[c] = mu; GW(VEC(l, V_VAR))[c] = var; GW(VEC(l, V_KNN))[c] = knn; GW(VEC(l, V_Q))[c] = q;
                    GW(VEC(l, V_RAW))[c] = vr;
                    if (xf[lane * XFW + DBT + 3] == (double)l) {
                        const double tau = sc[l], dlt = xf[lane * XFW + DBT + 2] - mu, w = xf[lane * XFW + DBT + 4];
                        dterm = w * (-0.5 * ((dlt * dlt + var) / tau + sc[8 + l] + LOG2PI)) / div;
                    }
                    if (l == L - 1 && md.top_mean) { GW(md.top_mean)[c] = mu; GW(md.top_var)[c] = var; }
                }
                dterm = wsum63(dterm);
                if (lane == 63) PART(l, cb)[0] = dterm;
            }
            __syncthreads();
        }
}


PHASE_FN void ph_elbo() {
    CTX_LOCALS;
    // ---- the ELBO (one wavefront of the surrogate's first workgroup; nobody waits for it)
    if (wj == 0 && wave == 0) {
        double data = 0.0, kl = 0.0;
        for (int l = 0; l < L; ++l) {
            for (int cb = lane; cb < g.ncb[l]; cb += 64) data += PART(l, cb)[0];
            if (lane == 0) kl += W[g.sml[l] + 6 * Mp];
        }
        data = wsum63(data);
        kl = wsum63(kl);
        if (lane == 63) {
            md.out[0] = data - gkl * kl;
            md.out[1] = gkl * kl;
            md.out[2] = -(data - gkl * kl);
        }
    }
}


PHASE_FN bool ph_couple() {
    CTX_LOCALS;
        // the conditioned iteration in one launch: all models' top-layer moments are in memory; the whole grid meets, the first
        // workgroup of every model forms the theta / omega factor gradients of its model (tiny_step.hip coupling_seeds), the
        // model's workgroups meet again
        const mobocmf_tiny_coupling& cpl = *md.coupling;
        if (cpl.n_models * k != (int)gridDim.x || cpl.T < 1 || cpl.T > 256 || cpl.P < 1) {
            if (tid == 0 && wj == 0) { atomicOr(cpl.status, 2); md.info[0] = -2; md.out[2] = __builtin_nan(""); }
            return false;
        }
        if (!group_barrier(cx->gcnt, gridDim.x, (int*)(sc + 30))) {
            if (tid == 0 && wj == 0) { atomicOr(cpl.status, 1); md.info[0] = -1; md.out[2] = __builtin_nan(""); }
            return false;
        }
        if (wj == 0) coupling_seeds<CT>(models, md, g.ncol[L - 1], W + g.cpl_off, sc + 16);
    return true;
}

    // ---- the weighted syrk H = A diag(gv) A^T, Hc = A diag(cgv) A^T of layer lh, k-sliced: one (tile, slice) per wavefront

PHASE_FN void ph_syrk(int lh) {
    CTX_LOCALS;
        const double* ATg = W + g.pan[lh];
        const double* vgv = VEC(lh, V_GV);
        const double* vcg = VEC(lh, V_CGV);
        const int ks = g.ks[lh], ncp = g.ncp[lh];
        const int len = ((ncp / 16 + ks - 1) / ks) * 16;
        for (int u = wj * CNW + wave; u < g.ntri * ks; u += k * CNW) {
            const int ch = u / g.ntri;
            int ti, tj;
            tile_of(u % g.ntri, ti, tj);
            const int cbeg = ch * len, cend = cbeg + len < ncp ? cbeg + len : ncp;
            v4d ah = {0.0, 0.0, 0.0, 0.0}, ac = {0.0, 0.0, 0.0, 0.0};
            gcd pa = GC(ATg) + (int64_t)lk * Mp + ti * 16 + li;
            gcd pb = GC(ATg) + (int64_t)lk * Mp + tj * 16 + li;
            gcd pw1 = GC(vgv) + lk;
            gcd pw2 = GC(vcg) + lk;
            // 16 columns (4 MFMA steps of each product) per stage, two stages requested ahead of the one being multiplied
            double a[3][4], b[3][4], w1[3][4], w2[3][4];
#define SYRK_LOAD(st, c_)                                                                                           \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                 \
        a[st][q] = pa[(int64_t)((c_) + 4 * q) * Mp]; b[st][q] = pb[(int64_t)((c_) + 4 * q) * Mp];                   \
        w1[st][q] = pw1[(c_) + 4 * q]; w2[st][q] = pw2[(c_) + 4 * q];                                               \
    }
            if (cbeg < cend) {
                SYRK_LOAD(0, cbeg);
                SYRK_LOAD(1, (cbeg + 16 < cend ? cbeg + 16 : cbeg));
            }
            for (int c = cbeg; c < cend; c += 16) {
                const int cn = c + 32 < cend ? c + 32 : c;
                SYRK_LOAD(2, cn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ah = mfma(a[0][q], b[0][q] * w1[0][q], ah);
                    ac = mfma(a[0][q], b[0][q] * w2[0][q], ac);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[0][q] = a[1][q]; b[0][q] = b[1][q]; w1[0][q] = w1[1][q]; w2[0][q] = w2[1][q];
                    a[1][q] = a[2][q]; b[1][q] = b[2][q]; w1[1][q] = w1[2][q]; w2[1][q] = w2[2][q];
                }
            }
#undef SYRK_LOAD
            double* Hs = MAT(lh, M_HS + ch);
            double* Hcs = MAT(lh, M_HCS + ch);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + 4 * r + lk, j = tj * 16 + li;
                Hs[(int64_t)i * Mp + j] = ah[r];
                Hcs[(int64_t)i * Mp + j] = ac[r];
                if (ti != tj) {
                    Hs[(int64_t)j * Mp + i] = ah[r];
                    Hcs[(int64_t)j * Mp + i] = ac[r];
                }
            }
        }
}

// INGRAD: mode 3 (input gradients; a compile-time variant so that the training step's code is what it was without it)
template <bool INGRAD>
PHASE_FN void ph_backward(int l_in) {
    CTX_LOCALS;
    DEF_STAGE_ROWS
        const bool has_tile = wave < nt;
        for (int cb = wj; cb < g.ncb[l_in]; cb += k) {
            int l = l_in;
            asm volatile("" : "+s"(l));
            const int kind = l > 0, div = l ? S : 1, nc = g.ncol[l];
            const double* hyl = hy + l * HS;
            const double* ill = il + l * 2 * DBT;
            const double* zt = zta + l * Mp * ZW;
            const double* avl = ava + l * Mp;
            const double* Lig = MAT(l, M_LI);
            const double* UTg = MAT(l, M_UT);
            const double* ATg = W + g.pan[l];
            const double* CTg = ATg + (int64_t)g.ncp[l] * Mp;
            CSTAMP(60);
            const int c0 = cb * 16;
            stage_rows(l, c0, false);
            {
                gcd ap = GC(ATg) + (int64_t)c0 * Mp;
                gcd cp = GC(CTg) + (int64_t)c0 * Mp;
                batched<4>(16 * Mp, tid, [&](int e) { return ap[e]; }, [&](int e, double v) { X1[(e % Mp) * XLD + e / Mp] = v; });
                batched<4>(16 * Mp, tid, [&](int e) { return cp[e]; }, [&](int e, double v) { X2[(e % Mp) * XLD + e / Mp] = v; });
            }
            // B1: upstream gradients of the block's moments: its own data term + what the next layer sent back through f
            if (wave == 1) {
                double st = 0.0;
                if (lane < 16) {
                    const int c = c0 + lane;
                    double gm = 0.0, gvc = 0.0, cg = 0.0;
                    if (c < nc) {
                        const int b = c / div;
                        const double tau = sc[l], gd = ge / div, vmean = VEC(l, V_MEAN)[c], vvar = VEC(l, V_VAR)[c];
                        double gvv = 0.0;
                        if (md.fid[b] == (double)l) {
                            const double dlt = md.y[b] - vmean, w = md.row_weight ? md.row_weight[b] : 1.0;
                            gm = w * gd * dlt / tau;
                            gvv = -0.5 * w * gd / tau;
                            st = w * 0.5 * ((dlt * dlt + vvar) / (tau * tau) - 1.0 / tau);
                        }
                        if (l == L - 1 && md.seed_gmean) {
                            gm += md.seed_scale * md.seed_gmean[c];
                            gvv += md.seed_scale * md.seed_gvar[c];
                        }
                        if (l + 1 < L) {
                            const int ncn = g.ncol[l + 1], fdiv = l == 0 ? S : 1;
                            if (c * fdiv < ncn) {
                                const double* ngf = VEC(l + 1, V_GF);
                                const double* nep = VEC(l + 1, V_EPS);
                                double sm = 0.0, sv = 0.0;
                                for (int s2 = 0; s2 < fdiv; ++s2) {
                                    const double g2 = ngf[c * fdiv + s2];
                                    sm += g2;
                                    sv += g2 * nep[c * fdiv + s2];
                                }
                                gm += sm;
                                gvv += sv * 0.5 / sqrt(vvar);
                            }
                        }
                        gvc = VEC(l, V_RAW)[c] > MINV ? gvv : 0.0;
                        cg = (md.branch || VEC(l, V_KNN)[c] - VEC(l, V_Q)[c] > 0.0) ? gvc : 0.0;
                    }
                    VEC(l, V_GMU)[c] = gm; VEC(l, V_GV)[c] = gvc; VEC(l, V_CGV)[c] = cg;
                    gcol[lane] = gm; gcol[16 + lane] = gvc; gcol[32 + lane] = cg;
                }
                st = wsum63(st);
                if (lane == 63) PART(l, cb)[1] = st;
            }
            __syncthreads();
            CSTAMP(61);
            if (!INGRAD && g.hblk[l]) {
                // the block's share of H = A diag(gv) A^T and Hc = A diag(cgv) A^T: slab `cb` (lower tiles and their mirrors)
                gwd Hs = GW(MAT(l, M_HS + cb));
                gwd Hcs = GW(MAT(l, M_HCS + cb));
                for (int u = wave; u < g.ntri; u += CNW) {
                    int ti, tj;
                    tile_of(u, ti, tj);
                    v4d ah = {0.0, 0.0, 0.0, 0.0}, ac = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double a = X1[(ti * 16 + li) * XLD + 4 * q + lk], b = X1[(tj * 16 + li) * XLD + 4 * q + lk];
                        ah = mfma(a, b * gcol[16 + 4 * q + lk], ah);
                        ac = mfma(a, b * gcol[32 + 4 * q + lk], ac);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = ti * 16 + 4 * r + lk, j = tj * 16 + li;
                        Hs[(int64_t)i * Mp + j] = ah[r];
                        Hcs[(int64_t)i * Mp + j] = ac[r];
                        if (ti != tj) {
                            Hs[(int64_t)j * Mp + i] = ah[r];
                            Hcs[(int64_t)j * Mp + i] = ac[r];
                        }
                    }
                }
            }
            // B2: dA = 2 U (C diag gv) + a g_mean^T - 2 A diag(cgv)   (the column scale commutes with the product)
            if (has_tile) {
                const v4d acc = tile_tx(UTg, Mp, X2, wave, 0, wave + 1, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = wave * 16 + 4 * r + lk;
                    X0[i * XLD + li] = 2.0 * gcol[16 + li] * acc[r] + avl[i] * gcol[li] - 2.0 * X1[i * XLD + li] * gcol[32 + li];
                }
            }
            for (int i = tid; i < Mp; i += CT) {      // the block's share of da = A g_mean
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < 16; ++j) s += X1[i * XLD + j] * gcol[j];
                PART(l, cb)[PHEAD + HS + Mp + i] = s;
            }
            __syncthreads();
            CSTAMP(62);
            // B3: dK = L^-T dA
            if (has_tile) store_x(X2, wave, lane, tile_tx(Lig, Mp, X0, wave, wave, nt, lane));
            __syncthreads();
            CSTAMP(63);
            if constexpr (INGRAD) {
                // B4' (input gradients, the parameters are constants): d/dx and d/df of the block's columns from dK.  A thread
                // keeps one column (CT is a multiple of 16) and sums over its inducing rows; the 32 threads of a column are
                // reduced through the wavefront (lanes j, j+16, j+32, j+48) and LDS.  The column's d/dx goes where its column
                // of A was (the block has it in LDS and nothing reads it again in this mode).
                double dxa[DBT], dfs = 0.0;
#pragma unroll
                for (int t = 0; t < DBT; ++t) dxa[t] = 0.0;
#pragma unroll 1
                for (int e = tid; e < Mp * 16; e += CT) {
                    const int m = e >> 4, j = e & 15;
                    if (m < M && xf[j * XFW + DBT + 1] != 0.0)
                        kern_back_in(kind, d, xf + j * XFW, xf[j * XFW + DBT], zt + m * ZW, hyl, ill, X2[m * XLD + j], dxa, dfs);
                }
#pragma unroll
                for (int t = 0; t <= DBT; ++t) {
                    const double v = xor_add(xor_add(t < DBT ? dxa[t < DBT ? t : 0] : dfs, 16), 32);
                    if (lane < 16) red[(wave * 16 + lane) * (DBT + 1) + t] = v;
                }
                __syncthreads();
                if (tid < 16 * (DBT + 1)) {
                    const int j = tid / (DBT + 1), t = tid % (DBT + 1), c = c0 + j;
                    double sum = 0.0;
#pragma unroll
                    for (int p = 0; p < CNW; ++p) sum += red[(p * 16 + j) * (DBT + 1) + t];
                    if (t < DBT) W[g.pan[l] + (int64_t)c * Mp + t] = c < nc ? sum : 0.0;
                    else if (kind) VEC(l, V_GF)[c] = c < nc ? sum + gcol[32 + j] * hyl[0] * 2.0 * hyl[2] * xf[j * XFW + DBT] : 0.0;
                }
                __syncthreads();
                continue;
            }
            // B4: Gram backward of (dK, dk_nn = cgv), element by element
            double hacc[HS];
#pragma unroll
            for (int t = 0; t < HS; ++t) hacc[t] = 0.0;
#pragma unroll 1
            for (int e = tid; e < Mp * 16; e += CT) {
                const int m = e >> 4, j = e & 15;
                double dfa = 0.0, dzf = 0.0;
                if (m < M && xf[j * XFW + DBT + 1] != 0.0) {
                    const double fn = xf[j * XFW + DBT];
                    kern_back(kind, d, xf + j * XFW, fn, zt + m * ZW, hyl, ill, X2[m * XLD + j], hacc, dfa, dzf);
                    if (m == 0) {
                        const double gk = gcol[32 + j];
                        if (!kind) hacc[0] += gk;
                        else {
                            hacc[0] += gk * (hyl[2] * fn * fn + hyl[1]);
                            hacc[1] += gk * hyl[0];
                            hacc[2] += gk * hyl[0] * fn * fn;
                            hacc[3] += gk;
                        }
                    }
                }
                X0[m * XLD + j] = dfa;
                X1[m * XLD + j] = dzf;
            }
#pragma unroll
            for (int t = 0; t < HS; ++t) {
                if (slot_used(kind, d, t)) {
                    const double v = wsum63(hacc[t]);
                    if (lane == 63) redh[wave * (HS + 1) + t] = v;
                }
            }
            __syncthreads();
            CSTAMP(64);
            // B5: the block's partial sums: hyper-parameters (fixed slot layout), d/df of its columns, d/dzf rows
            if (tid < HS) PART(l, cb)[PHEAD + tid] = slot_used(kind, d, tid) ? red_sum<CNW>(redh, HS + 1, tid) : 0.0;
            if (kind) {
                {
                    const int j = tid & 15, part = tid >> 4;
                    double s = 0.0;
                    for (int m = part; m < Mp; m += CT / 16) s += X0[m * XLD + j];
                    s = xor_add(xor_add(s, 16), 32);
                    if (lane < 16) red[wave * 16 + lane] = s;
                }
                for (int m = tid; m < Mp; m += CT) {
                    double s = 0.0;
#pragma unroll
                    for (int j = 0; j < 16; ++j) s += X1[m * XLD + j];
                    PART(l, cb)[PHEAD + HS + m] = s;
                }
                __syncthreads();
                if (tid < 16) {
                    const int c = c0 + tid;
                    double s = 0.0;
#pragma unroll
                    for (int p = 0; p < CNW; ++p) s += red[p * 16 + tid];
                    VEC(l, V_GF)[c] = c < nc ? s + gcol[32 + tid] * hyl[0] * 2.0 * hyl[2] * xf[tid * XFW + DBT] : 0.0;
                }
            }
            __syncthreads();
        }
        CSTAMP(20 + l_in);
}

// mode 3, last phase: d/dx of a base row = the sum over the layers that hold the row and over its sample columns
PHASE_FN void ph_dx() {
    CTX_LOCALS;
    gwd out = GW(md.grad);
    for (int e = wj * CT + tid; e < md.N * d; e += k * CT) {
        const int n = e / d, t = e - n * d;
        double sum = 0.0;
        for (int l = 0; l < L; ++l) {
            if (n < md.rows[l]) {
                const int div = l ? S : 1;
                for (int s2 = 0; s2 < div; ++s2) sum += W[g.pan[l] + (int64_t)(n * div + s2) * Mp + t];
            }
        }
        out[e] = sum;
    }
}

    // ---- the M x M chain backward, every layer at once, tile-parallel: one 16 x 16 output tile per wavefront.
    // Operand fragments are k-major (rows k0 + 4 q + lk, q = 0..3; 16 contiguous columns from x0): 4 rows x 128 contiguous bytes
    // per load instruction; the fragments of k tile kt + 1 are requested before the MFMAs of k tile kt are issued.

PHASE_FN void ph_cb1() {
    CTX_LOCALS;
    const int gw = wj * CNW + wave, nwv = k * CNW, nt2 = nt * nt;
    // CB1: G1 = U^T H, GT = H U (= G1^T: H is symmetric -- both orientations, so that every later read is k-major); two tile
    // products per task, H read as the sum of its k-slices.  Beside them: the column blocks' partial sums of a layer added up
    // (hyper-parameters, noise, d/dzf rows), one wavefront per 64 of them.
    for (int u = gw; u < 2 * L * nt2; u += nwv) {
        const int which = u / (L * nt2), l = (u % (L * nt2)) / nt2, ti = (u % nt2) / nt, tj = u % nt;
        const double* Ug = MAT(l, M_U);
        const int ks = g.ks[l];
        // which 0: G1[i][j] = sum_{k >= i-block} U[k][i] H[k][j];  1: GT[i][j] = sum_{k >= j-block} H[k][i] U[k][j]
        const int ucol = which ? tj * 16 : ti * 16, hcol = which ? ti * 16 : tj * 16, kt0 = which ? tj : ti;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        // (the fragments of U and of every slab of H for TWO k tiles -- one when there are more than four slabs -- are requested
        // together; the slabs are added in order 0, 1, ... on the way into the MFMA)
        const double* Hb = MAT(l, M_HS);
        const int kb = ks <= 4 ? 2 : 1;
        for (int kt = kt0; kt < nt; kt += kb) {
            double uf[2][4], hf[8][4];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                if (c2 < kb) {
                    const int kc = kt + c2 < nt ? kt + c2 : nt - 1;
                    FRAG(uf[c2], Ug, kc * 16, ucol);
#pragma unroll
                    for (int s2 = 0; s2 < 8; ++s2)
                        if (s2 < ks && (c2 == 0 || s2 < 4)) { FRAG(hf[c2 * 4 + s2], Hb + s2 * mm, kc * 16, hcol); }
                }
            }
            PIN();
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                if (c2 < kb && kt + c2 < nt) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        double h = hf[c2 * 4][q];
#pragma unroll
                        for (int s2 = 1; s2 < 8; ++s2)
                            if (s2 < ks && (c2 == 0 || s2 < 4)) h += hf[c2 * 4 + s2][q];
                        acc = which ? mfma(h, uf[c2][q], acc) : mfma(uf[c2][q], h, acc);
                    }
                }
            }
            PIN();
        }
        gwd G = GW(MAT(l, which ? M_GT : M_G1));
#pragma unroll
        for (int r = 0; r < 4; ++r) G[(int64_t)(ti * 16 + 4 * r + lk) * Mp + tj * 16 + li] = acc[r];
    }
    {
        const int nel = PHEAD + HS + Mp, nch = (nel + 63) / 64, first = (2 * L * nt2) % nwv;
        for (int u = (gw - first + nwv) % nwv; u < L * nch; u += nwv) {
            const int l = u / nch, e = (u % nch) * 64 + lane, ncb = g.ncb[l];
            if (e < nel) {
                gcd pp = GC(W + g.part[l]) + e;
                double s = 0.0;
                for (int cb = 0; cb < ncb; cb += 8) {
                    double v[8];
#pragma unroll
                    for (int b = 0; b < 8; ++b) v[b] = pp[(int64_t)(cb + b < ncb ? cb + b : ncb - 1) * g.pstr];
#pragma unroll
                    for (int b = 0; b < 8; ++b) s += cb + b < ncb ? v[b] : 0.0;
                }
                // (layout of the sums: [PHEAD scalars | HS hyper-parameter slots] behind the KL, the d/dzf rows in S_GZ)
                if (e < PHEAD + HS) W[g.sml[l] + 6 * Mp + 1 + e] = s;
                else SML(l, S_GZ)[e - PHEAD - HS] = s;
            }
        }
        // da = A g_mean of every layer (the column blocks' partial rows), da_tot = da + gkl a: the same way, for CB2+3
        const int nch2 = (Mp + 63) / 64, first2 = (first + L * nch) % nwv;
        for (int u = (gw - first2 + nwv) % nwv; u < L * nch2; u += nwv) {
            const int l = u / nch2, i = (u % nch2) * 64 + lane, ncb = g.ncb[l];
            if (i < Mp) {
                gcd pp = GC(W + g.part[l]) + PHEAD + HS + Mp + i;
                double s = 0.0;
                for (int cb = 0; cb < ncb; cb += 8) {
                    double v[8];
#pragma unroll
                    for (int b = 0; b < 8; ++b) v[b] = pp[(int64_t)(cb + b < ncb ? cb + b : ncb - 1) * g.pstr];
#pragma unroll
                    for (int b = 0; b < 8; ++b) s += cb + b < ncb ? v[b] : 0.0;
                }
                SML(l, S_DAV)[i] = s;
                SML(l, S_DAT)[i] = s + gkl * GC(SML(l, S_AV))[i];
            }
        }
    }
}


PHASE_FN void ph_cb23() {
    CTX_LOCALS;
    const int gw = wj * CNW + wave, nwv = k * CNW, nt2 = nt * nt;
    // CB2+3: Y = 2 (U G1 - Hc) + a da^T + da_tot a^T + dU_tot U^T with dU_tot = 2 tril(G1^T) + gkl U (one task per tile);
    //        g_LS = tril(L^-T dU_tot) - gkl diag(1 / L_S,ii) (one task per lower tile)
    for (int u = gw; u < L * nt2 + L * g.ntri + L * nt; u += nwv) {
        if (u >= L * nt2 + L * g.ntri) {      // g_m += L^-T da_tot: a task per 16 rows (da_tot rides as column 0 of a B operand)
            const int u2 = u - L * nt2 - L * g.ntri, l = u2 / nt, t = u2 % nt;
            const double* Lig = MAT(l, M_LI);
            gcd dat = GC(SML(l, S_DAT));
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            for (int kt = t; kt < nt; ++kt) {
                double a[4], b[4];
                FRAG(a, Lig, kt * 16, t * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) b[q] = dat[kt * 16 + 4 * q + lk];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = mfma(a[q], li == 0 ? b[q] : 0.0, acc);
            }
            if (li == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) SML(l, S_GMA)[t * 16 + 4 * r + lk] = acc[r];
            }
            continue;
        }
        if (u >= L * nt2) {
            const int l = (u - L * nt2) / g.ntri;
            int ti, tj;
            tile_of((u - L * nt2) % g.ntri, ti, tj);
            const double* Ug = MAT(l, M_U);
            const double* GT = MAT(l, M_GT);
            const double* Lig = MAT(l, M_LI);
            v4d s3 = {0.0, 0.0, 0.0, 0.0};
            // (the fragments of up to four k tiles are requested together: one L2 round trip per four tiles, not one per tile)
            for (int kb = ti; kb < nt; kb += 4) {
                double la[4][4], gt[4][4], ub[4][4];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int kt = kb + s4 < nt ? kb + s4 : nt - 1;
                    FRAG(la[s4], Lig, kt * 16, ti * 16);      // L^-1[k][i]
                    FRAG(gt[s4], GT, kt * 16, tj * 16);       // G1[j][k]
                    FRAG(ub[s4], Ug, kt * 16, tj * 16);       // U[k][j]
                }
                PIN();
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    if (kb + s4 < nt) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int kk = (kb + s4) * 16 + 4 * q + lk, j = tj * 16 + li;
                            const double du = j <= kk ? 2.0 * gt[s4][q] + gkl * ub[s4][q] : 0.0;      // dU_tot[k][j]
                            s3 = mfma(la[s4][q], du, s3);
                        }
                    }
                }
            }
            gwd gls = GW(gflat + g.flat_off[l] + g.H[l] + M);
            gcd ls = GC(md.L_S[l]);
            const int j = tj * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + 4 * r + lk;
                if (i < M && j < M) {
                    double t = 0.0;
                    if (j <= i) {
                        t = s3[r];
                        if (i == j) t -= gkl / ls[(int64_t)i * M + i];
                    }
                    gls[(int64_t)i * M + j] = t;
                    if (ti != tj) gls[(int64_t)j * M + i] = 0.0;      // the upper triangle of g_LS: zeros
                }
            }
            continue;
        }
        const int l = u / nt2, ti = (u % nt2) / nt, tj = u % nt;
        const double* UTg = MAT(l, M_UT);
        const double* G1 = MAT(l, M_G1);
        const int ks = g.ks[l];
        v4d s1 = {0.0, 0.0, 0.0, 0.0}, s2 = {0.0, 0.0, 0.0, 0.0};
        const int kmin = ti < tj ? ti : tj;
        for (int kb = 0; kb <= ti; kb += 4) {
            double ua[4][4], gb[4][4], gi[4][4], ub[4][4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int kt = kb + s4 <= ti ? kb + s4 : ti, k2 = kt <= kmin ? kt : kmin;
                FRAG(ua[s4], UTg, kt * 16, ti * 16);      // U[i][k]
                FRAG(gb[s4], G1, kt * 16, tj * 16);       // G1[k][j]
                FRAG(gi[s4], G1, k2 * 16, ti * 16);       // G1[k][i]   (k <= min(i, j) only)
                FRAG(ub[s4], UTg, k2 * 16, tj * 16);      // U[j][k]
            }
            PIN();
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int kt = kb + s4;
                if (kt <= ti) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) s1 = mfma(ua[s4][q], gb[s4][q], s1);
                    if (kt <= kmin) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int kk = kt * 16 + 4 * q + lk, i = ti * 16 + li;
                            const double du = kk <= i ? 2.0 * gi[s4][q] + gkl * ua[s4][q] : 0.0;      // dU_tot[i][k]
                            s2 = mfma(du, ub[s4][q], s2);
                        }
                    }
                }
            }
        }
        gcd avg = GC(SML(l, S_AV));
        gcd dav = GC(SML(l, S_DAV));
        gcd dat = GC(SML(l, S_DAT));
        gwd Y = GW(MAT(l, M_Y));
        const int j = tj * 16 + li;
        const double aj = avg[j], dj = dav[j];
        double hc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sl = 0; sl < KSMAX; ++sl) {
            if (sl < ks) {
#pragma unroll
                for (int r = 0; r < 4; ++r) hc[r] += GC(MAT(l, M_HCS) + sl * mm)[(int64_t)(ti * 16 + 4 * r + lk) * Mp + j];
            }
        }
        // (Y overwrites slice 0 of H, which CB1 was the last to read; Hc is a different matrix)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ti * 16 + 4 * r + lk;
            Y[(int64_t)i * Mp + j] = 2.0 * (s1[r] - hc[r]) + avg[i] * dj + dat[i] * aj + s2[r];
        }
    }
}


PHASE_FN void ph_cb456() {
    CTX_LOCALS;
    // CB4-6, one workgroup per block of 16 columns of Y: dL = -tril(L^-T Y) + gkl diag(1 / L_ii), P = Phi(L^T dL), T4 = L^-T P;
    for (int u = wj; u < L * nt; u += k) {
        const int l = u / nt, cb = u % nt;
        const double* Lg = MAT(l, M_L);
        const double* Lig = MAT(l, M_LI);
        const double* Y = MAT(l, M_Y);
        __syncthreads();
        const int ntl = wave < nt ? 1 : 0, t = wave;
        {
            gcd yp = GC(Y) + cb * 16;
            batched<4>(Mp * 16, tid, [&](int e) { return yp[(int64_t)(e >> 4) * Mp + (e & 15)]; },
                       [&](int e, double v) { X0[(e >> 4) * XLD + (e & 15)] = v; });
        }
        __syncthreads();
        if (ntl) {
            const v4d acc = tile_tx(Lig, Mp, X0, t, t, nt, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = t * 16 + 4 * r + lk, j = cb * 16 + li;
                double v = j <= i ? -acc[r] : 0.0;
                if (i == j && i < M) v += gkl / GC(Lg)[(int64_t)i * Mp + i];
                X1[i * XLD + li] = v;
            }
        }
        __syncthreads();
        if (ntl) {
            const v4d acc = tile_tx(Lg, Mp, X1, t, t, nt, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = t * 16 + 4 * r + lk, j = cb * 16 + li;
                X2[i * XLD + li] = j < i ? acc[r] : (j == i ? 0.5 * acc[r] : 0.0);
            }
        }
        __syncthreads();
        if (ntl) store_x(X0, t, lane, tile_tx(Lig, Mp, X2, t, t > cb ? t : cb, nt, lane));
        __syncthreads();
        gwd T4T = GW(MAT(l, M_T4T));
        for (int e = tid; e < 16 * Mp; e += CT) {
            const int j = e / Mp, m = e % Mp;
            T4T[(int64_t)(cb * 16 + j) * Mp + m] = X0[m * XLD + j];
        }
    }
}


PHASE_FN void ph_cb8() {
    CTX_LOCALS;
    // CB7+8: T5 = T4 L^-1 and the Gram backward of dK_mm = sym(T5), a workgroup per 16 rows (both arguments are Z~: a pair's f
    // gradient counts twice).  The workgroup forms the tile row AND the tile column of T5 that its 16 rows of sym(T5) are made of
    // (2 nt - 1 tile products, a wavefront each, into LDS) instead of reading them back after one more barrier.
    double* Gs = X0;      // [16][Mp + 1]: T5[i][j] + T5[j][i] of the workgroup's rows i (fits: 16 (Mp + 1) <= 3 Mp XLD)
    const int GLD = Mp + 1;
    for (int u = wj; u < L * nt; u += k) {
        const int l = u / nt, ti = u % nt, kind = l > 0;
        const double* zt = zta + l * Mp * ZW;
        const double* T4T = MAT(l, M_T4T);
        const double* Lig = MAT(l, M_LI);
        __syncthreads();
        for (int e = tid; e < 16 * GLD; e += CT) Gs[e] = 0.0;
        __syncthreads();
        for (int pass = 0; pass < 2; ++pass) {
            // pass 0: tiles (ti, tj), added as they are; pass 1: tiles (tj, ti), added transposed (the diagonal tile takes both)
            for (int tj = wave; tj < nt; tj += CNW) {
                const int ta = pass ? tj : ti, tb = pass ? ti : tj;      // T5 tile (ta, tb) = sum_k T4[ta-rows][k] L^-1[k][tb-cols]
                v4d acc = {0.0, 0.0, 0.0, 0.0};
                for (int kb = tb; kb < nt; kb += 4) {
                    double a[4][4], b[4][4];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int kt = kb + s4 < nt ? kb + s4 : nt - 1;
                        FRAG(a[s4], T4T, kt * 16, ta * 16);
                        FRAG(b[s4], Lig, kt * 16, tb * 16);
                    }
                    PIN();
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        if (kb + s4 < nt) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc = mfma(a[s4][q], b[s4][q], acc);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ra = 4 * r + lk, cb2 = li;      // element (ra, cb2) of tile (ta, tb)
                    if (!pass) Gs[ra * GLD + tj * 16 + cb2] += acc[r];
                    else Gs[cb2 * GLD + tj * 16 + ra] += acc[r];
                }
            }
            __syncthreads();      // (a wavefront's two passes touch the same 16 x 16 patch of Gs: in order)
        }
        const double* hyl = hy + l * HS;
        const double* ill = il + l * 2 * DBT;
        double hacc[HS];
#pragma unroll
        for (int t = 0; t < HS; ++t) hacc[t] = 0.0;
        const int i = ti * 16 + (tid & 15);
        double rs = 0.0;
        if (i < M) {
#pragma unroll 1
            for (int j = tid >> 4; j < M; j += CT / 16) {
                const double G = 0.5 * Gs[(tid & 15) * GLD + j];
                double dfa, dzf;
                kern_back(kind, d, zt + i * ZW, zt[i * ZW + DBT], zt + j * ZW, hyl, ill, G, hacc, dfa, dzf);
                rs += dfa;
            }
        }
        red[(tid >> 4) * 16 + (tid & 15)] = rs;
#pragma unroll
        for (int t = 0; t < HS; ++t) {
            if (slot_used(kind, d, t)) {
                const double v = wsum63(hacc[t]);
                if (lane == 63) redh[wave * (HS + 1) + t] = v;
            }
        }
        __syncthreads();
        if (tid < HS) W[g.hpart[l] + (int64_t)ti * HS + tid] = slot_used(kind, d, tid) ? red_sum<CNW>(redh, HS + 1, tid) : 0.0;
        if (tid >= 64 && tid < 80) {
            double s = 0.0;
            for (int p = 0; p < CT / 16; ++p) s += red[p * 16 + (tid - 64)];
            SML(l, S_GMB)[ti * 16 + (tid - 64)] = 2.0 * s;
        }
        __syncthreads();
    }
}


PHASE_FN void ph_adam() {
    CTX_LOCALS;
    // ---- raw-parameter gradients assembled from the (pre-reduced) partial sums, element by element of the flat vector, and Adam
    // (torch.optim.Adam: p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)); g_LS is in the flat vector already.  Four elements
    // per thread at a time: their loads are in flight together.
    {
        const double bc1 = sc[12], bc2s = sc[13];
        const bool upd = do_update == 1 || do_update == 4;
        const int flat = (int)g.flat_len;
        auto grad_of = [&](int e) -> double {
            if (e >= (int)g.flat_noise) {
                const int l = e - (int)g.flat_noise, div = l ? S : 1;
                const double s = GC(W)[g.sml[l] + 6 * Mp + 1 + 1];
                double chain = 1.0;
                if (md.noise_hi[l] > md.noise_lo[l]) {
                    const double sg = 1.0 / (1.0 + exp(-md.raw_noise[l][0]));
                    chain = (md.noise_hi[l] - md.noise_lo[l]) * sg * (1.0 - sg);
                }
                return s / div * ge * chain;
            }
            int l = 0;
            while (l + 1 < L && e >= (int)g.flat_off[l + 1]) ++l;
            const int t = e - (int)g.flat_off[l], Hl = g.H[l];
            if (t < Hl) {
                const int slot = slot_of(l > 0, d, t);
                double s = GC(W)[g.sml[l] + 6 * Mp + 1 + PHEAD + slot];
                for (int ti = 0; ti < nt; ++ti) s += GC(W)[g.hpart[l] + (int64_t)ti * HS + slot];
                int sgm = 0, off = 0;
                while (t >= off + seg_len(l, sgm, d)) { off += seg_len(l, sgm, d); ++sgm; }
                const double x = md.raw[l][sgm][t - off];
                return x > 20.0 ? s : s / (1.0 + exp(-x));
            }
            if (t < Hl + M) {
                // (the layer is LANE-varying here -- a wavefront of this loop can hold the m entries of two layers when M^2 is
                // small: M <= 6 -- so the layer's base is formed per lane, not through SML(), which makes it wave-uniform)
                const int i = t - Hl;
                double s = GC(W)[g.sml[l] + (int64_t)S_GMA * Mp + i];
                if (l + 1 < L) s += GC(W)[g.sml[l + 1] + (int64_t)S_GMB * Mp + i] + GC(W)[g.sml[l + 1] + (int64_t)S_GZ * Mp + i];
                return s;
            }
            return GC(gflat)[e];
        };
        for (int base = wj * CT + tid; base < flat; base += 4 * k * CT) {
            double gi[4], am[4], av[4], pv[4];
            double* pp[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int e0 = base + b * k * CT, e = e0 < flat ? e0 : flat - 1;
                int kk = 0;
#pragma unroll
                for (int sft = NSEG / 2; sft > 0; sft >>= 1)
                    if (seg_end[kk + sft - 1] <= e) kk += sft;
                double* p = seg_ptr[kk];
                pp[b] = (upd && p && e0 < flat) ? p + (e - (kk ? seg_end[kk - 1] : 0)) : nullptr;
                gi[b] = grad_of(e);
                am[b] = GC(md.adam_m)[e];
                av[b] = GC(md.adam_v)[e];
                pv[b] = pp[b] ? GC(pp[b])[0] : 0.0;
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int e = base + b * k * CT;
                if (e >= flat) continue;
                GW(gflat)[e] = gi[b];
                if (md.grad) GW(md.grad)[e] = gi[b];
                if (!pp[b]) continue;
                const double mi2 = b1 * am[b] + (1.0 - b1) * gi[b];
                const double vi = b2 * av[b] + (1.0 - b2) * gi[b] * gi[b];
                GW(md.adam_m)[e] = mi2;
                GW(md.adam_v)[e] = vi;
                GW(pp[b])[0] = pv[b] - (lr / bc1) * mi2 / (sqrt(vi) / bc2s + aeps);
            }
        }
        CSTAMP(40);
        if (upd && wj == 0 && tid == 0) {
            md.steps_done[0] += 1;
            for (int l = 1; l < L; ++l)
                if (!md.eps[l] && md.rng[l]) md.rng[l][1] += 1;
            if (blockIdx.x == 0 && md.xrng) md.xrng[1] += 1;
        }
    }
}


#define MODEL_BARRIER(id) do { CSTAMP(id); if (!group_barrier(cx->mcnt, (unsigned)k, (int*)(sc + 30))) { if (tid == 0) { md.info[0] = -1; md.out[2] = __builtin_nan(""); } return; } CSTAMP(99); } while (0)

// PREDICT: the instantiation for an acquisition search (mode 3, and mode 2 with MOBOCMF_STEP_CHAIN_VALID) -- a kernel of its own, so
// that the training step's kernel carries none of its code (with the input-gradient phases inlined into ONE kernel the training
// step lost ~1.5 %: 316.5 -> 321.5 us at C2)
template <bool PREDICT>
__global__ __launch_bounds__(CT) void coop_step_kernel(const mobocmf_tiny_model* models_, int k_, unsigned long long* sync_words,
                                                       double lr_, double b1_, double b2_, double aeps_, int do_update_) {
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    {
        // the surrogate's descriptor, its geometry and the phases' context: into LDS once (a descriptor field read through the
        // global pointer is an L2 round trip whenever the compiler cannot prove that no store in between changed it; the
        // geometry's per-layer arrays, indexed by the run-time layer, would live in scratch memory as a private struct)
        const int mi = blockIdx.x / k_;
        if (threadIdx.x < (int)(sizeof(mobocmf_tiny_model) / 8))
            ((uint64_t*)lds_all)[threadIdx.x] = ((const uint64_t*)(models_ + mi))[threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) {
            cgeom_of(*(const mobocmf_tiny_model*)lds_all, *(CGeom*)(lds_all + MDW));
            Ctx* c0 = (Ctx*)(lds_all + MDW + GEW);
            c0->models = models_;
            c0->lds = lds_all + MDW + GEW + CXW;
            c0->mcnt = sync_words + 16 * (int64_t)mi;
            c0->gcnt = sync_words + 16 * (int64_t)(gridDim.x / k_);
            c0->lr = lr_; c0->b1 = b1_; c0->b2 = b2_; c0->aeps = aeps_;
            c0->k = k_; c0->wj = blockIdx.x % k_; c0->do_update = do_update_ & 15; c0->n_stamp = 0;
        }
        __syncthreads();
    }
    // (only what the barriers and the phase sequence need: everything else is re-derived inside the phases)
    const mobocmf_tiny_model& md = *(const mobocmf_tiny_model*)lds_all;
    const CGeom& g = *(const CGeom*)(lds_all + MDW);
    Ctx* cx = (Ctx*)(lds_all + MDW + GEW);
    const int tid = threadIdx.x, k = uni(cx->k), wj = uni(cx->wj), do_update = uni(cx->do_update), L = uni(g.L), nt = uni(g.nt);
    double* sc = lds_all + MDW + GEW + CXW + TLM * HS + TLM * 2 * DBT;
    (void)wj;
    CSTAMP(0);
    ph_setup();
    if (!PREDICT || !(do_update_ & MOBOCMF_STEP_CHAIN_VALID)) {      // (set: L^-1, U, a, KL of these parameters are in `work` from an earlier launch)
        const bool spread_kmm = nt > 4 && k >= 4;      // (the same in every workgroup of the surrogate: one barrier more)
        if (spread_kmm) {
            ph_kmm();
            MODEL_BARRIER(8);
        }
        ph_chain(spread_kmm);
        MODEL_BARRIER(6);
    }
    ph_stage();
    for (int l = 0; l < L; ++l) {
        ph_forward(l);
        MODEL_BARRIER(10 + l);
    }
    ph_elbo();
    if (do_update == 2) return;
    if constexpr (PREDICT) {      // input gradients: the backward column phases only, then the rows' sums
        for (int l = L - 1; l >= 0; --l) {
            ph_backward<true>(l);
            MODEL_BARRIER(25 + l);
        }
        ph_dx();
        return;
    } else {
    if (do_update == 4) {
        if (!ph_couple()) return;
        MODEL_BARRIER(15);
    }
    for (int l = L - 1; l >= 0; --l) {
        ph_backward<false>(l);
        if (l + 1 < L && !g.hblk[l + 1]) ph_syrk(l + 1);
        MODEL_BARRIER(25 + l);
    }
    if (!g.hblk[0]) {      // (a layer of few column blocks left its H in the backward column phase: no phase, no barrier here)
        ph_syrk(0);
        MODEL_BARRIER(30);
    }
    ph_cb1();
    MODEL_BARRIER(31);
    ph_cb23();
    MODEL_BARRIER(32);
    ph_cb456();
    MODEL_BARRIER(33);
    ph_cb8();
    MODEL_BARRIER(35);
    ph_adam();
    }
}
#undef FRAG
#undef COPY4
#undef PIN
#undef MODEL_BARRIER

#undef CSTAMP

size_t coop_lds_bytes(int Mp) {
    const int ntri = (Mp / 16) * (Mp / 16 + 1) / 2;
    const size_t common = (sizeof(mobocmf_tiny_model) + 15) / 16 * 2 + (sizeof(CGeom) + 15) / 16 * 2 + (size_t)TLM * HS + (size_t)TLM * 2 * DBT + 32 + NSEG + NSEG / 2 + 20;
    const size_t chain = (size_t)Mp * ZW + 2 * (size_t)ntri * 256;
    const size_t col = (size_t)TLM * Mp * ZW + 3 * (size_t)Mp * XLD + (size_t)TLM * Mp + 16 * XFW + 64 + (CT / 16) * 16 * 3 + CNW * (HS + 1);
    return (common + (chain > col ? chain : col)) * sizeof(double);
}

bool coop_valid_model(const mobocmf_tiny_model& m) {
    if (m.L < 1 || m.L > TLM || m.M < 1 || m.M > CMAXM || m.d < 1 || m.d > DBT || m.S < 1 || m.N < 1) return false;
    if (m.rows[0] != m.N) return false;
    for (int l = 0; l < m.L; ++l) {
        if (m.rows[l] < 1 || (l && m.rows[l] > m.rows[l - 1])) return false;
        if ((int64_t)m.rows[l] * m.S > (1 << 20)) return false;
        const int ns = l == 0 ? 2 : 7;
        for (int s = 0; s < ns; ++s)
            if (!m.raw[l][s]) return false;
        if (!m.m[l] || !m.L_S[l] || !m.raw_noise[l]) return false;
        if (l && !m.eps[l] && !m.rng[l]) return false;
    }
    if ((m.seed_gmean == nullptr) != (m.seed_gvar == nullptr) || (m.top_mean == nullptr) != (m.top_var == nullptr)) return false;
    if (m.xrng && (m.rand_row0 < 0 || m.rand_rows < 0 || m.rand_row0 + m.rand_rows > m.N)) return false;
    if (m.branch != 0 && m.branch != 1) return false;
    return m.x && m.y && m.fid && m.Zx && m.adam_m && m.adam_v && m.steps_done && m.work && m.out && m.info;
}

}  // namespace

extern "C" {

int mobocmf_coop_work_bytes(const mobocmf_tiny_model* model, size_t* bytes) {
    if (!model || !bytes || model->L < 1 || model->L > TLM || model->M < 1 || model->M > CMAXM || model->d < 1 || model->S < 1)
        return MOBOCMF_BAD_ARG;
    for (int l = 0; l < model->L; ++l)
        if (model->rows[l] < 1) return MOBOCMF_BAD_ARG;
    CGeom g;
    cgeom_of(*model, g);
    size_t n = (size_t)g.work_len;
#ifdef COOP_STAMPS
    n += 256;
#endif
    *bytes = n * sizeof(double);
    return MOBOCMF_OK;
}

int mobocmf_coop_elbo_step(const mobocmf_tiny_model* host_models, const mobocmf_tiny_model* dev_models, int32_t n_models,
                           int32_t wgs_per_model, int64_t* sync_words, double lr, double beta1, double beta2, double eps,
                           int32_t do_update, int32_t* wgs_used, mobocmf_stream_t stream) {
    if (!host_models || !dev_models || !sync_words || n_models < 1 || n_models > 256 || wgs_per_model < 0 || wgs_per_model > 64)
        return MOBOCMF_BAD_ARG;
    const int chain_valid = do_update & MOBOCMF_STEP_CHAIN_VALID;
    do_update &= ~MOBOCMF_STEP_CHAIN_VALID;
    if (do_update < 0 || do_update > 4 || (chain_valid && do_update != 2 && do_update != 3)) return MOBOCMF_BAD_ARG;
    int mpmax = 0, want = 1;
    for (int i = 0; i < n_models; ++i) {
        const mobocmf_tiny_model& m = host_models[i];
        if (!coop_valid_model(m)) return MOBOCMF_BAD_ARG;
        if (do_update == 4) {
            if (!m.coupling || m.S != 1 || !m.seed_gmean || !m.top_mean || m.role < 0 || m.role > 1 || m.role_index < 0 ||
                m.role_index > 7 || m.coupling != host_models[0].coupling)
                return MOBOCMF_BAD_ARG;
        }
        if (do_update == 3 && (!m.grad || !m.seed_gmean)) return MOBOCMF_BAD_ARG;      // (grad receives N x d input gradients)
        CGeom g;
        cgeom_of(m, g);
        if (g.Mp > mpmax) mpmax = g.Mp;
        // what this model's widest phase can use: the chain backward's column blocks of all layers, half the column blocks of a
        // layer's panel (two rounds of column blocks cost less than the barrier among twice the workgroups)
        int w = m.L * g.nt;
        for (int l = 0; l < m.L; ++l)
            if ((g.ncb[l] + 1) / 2 > w) w = (g.ncb[l] + 1) / 2;
        if (w > want) want = w;
        if (g.work_len >= ((int64_t)1 << 31)) return MOBOCMF_BAD_ARG;      // (workspace offsets are formed in 32 bits)
    }
    const size_t shm = coop_lds_bytes(mpmax);
    const bool predict = do_update == 3 || chain_valid;
    const void* kfn = predict ? (const void*)coop_step_kernel<true> : (const void*)coop_step_kernel<false>;
    static std::atomic<uint64_t> granted[2] = {{0}, {0}};      // one write-once bit per device and instantiation: the dynamic-LDS attribute was set there
    int devid = 0;
    HIP_TRY(hipGetDevice(&devid));
    const uint64_t bit = devid >= 0 && devid < 64 ? 1ull << devid : 0ull;
    if (shm > 64 * 1024 && !(granted[predict].load() & bit)) {
        HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        granted[predict].fetch_or(bit);
    }
    // every workgroup of the launch waits for its peers inside the launch: all of them must be resident at once
    int per_cu = 0, cus = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, CT, shm));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid));
    const int64_t resident = (int64_t)per_cu * cus;
    int k = wgs_per_model;
    if (k == 0) {
        k = want < 32 ? want : 32;
        const int64_t cap = resident < 160 ? resident : 160;      // (the barrier's fences cost with the workgroups of the whole launch)
        while (k > 1 && (int64_t)k * n_models > cap) --k;
    }
    if ((int64_t)k * n_models > resident || k < 1) return MOBOCMF_BAD_ARG;
    if (wgs_used) *wgs_used = k;
    if (predict)
        hipLaunchKernelGGL(coop_step_kernel<true>, dim3((unsigned)(n_models * k)), dim3(CT), shm, (hipStream_t)stream, dev_models, k,
                           (unsigned long long*)sync_words, lr, beta1, beta2, eps, do_update | chain_valid);
    else
        hipLaunchKernelGGL(coop_step_kernel<false>, dim3((unsigned)(n_models * k)), dim3(CT), shm, (hipStream_t)stream, dev_models, k,
                           (unsigned long long*)sync_words, lr, beta1, beta2, eps, do_update);
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}

}  // extern "C"
