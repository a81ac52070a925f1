// C-ABI orchestration of one variational MFDGP layer (forward, backward, predictive covariance).
// See include/mobocmf_hip.h for the contract and DESIGN.md for the algebra.  Every function only
// enqueues kernels on the caller's stream; all memory is caller-provided (saved / scratch).
#include "common.h"

// ---- kernels / launchers defined in the other translation units
int launch_gram_fwd(const GramArgs& g, hipStream_t s);
int launch_gram_bwd(const GramArgs& g, bool want_dx, hipStream_t s);
void gram_grid(const GramArgs& g, dim3* grid);
int launch_sum_partials(const double* part, int64_t P, int64_t stride, double* out, int64_t len, double scale,
                        int accumulate, hipStream_t s);
int launch_potrf(double* A, int64_t ld, int Mp, int M, double* Dinv, double* Ld, int32_t* info, hipStream_t s);
int launch_trtri(const double* L, int64_t ld, int Mp, const double* Dinv, double* Linv, double* T, double* ws,
                 int64_t ws_elems, hipStream_t s);
int launch_pad_tril(const double* src, int64_t lds, int M, double* dst, int Mp, hipStream_t s);
int launch_pad_vec(const double* src, int64_t n, double* dst, int64_t np, hipStream_t s);
int launch_transpose(const double* in, int64_t ldi, double* out, int64_t ldo, int64_t rows, int64_t cols, hipStream_t s);
int launch_gemv_rows(const double* Mat, int64_t ld, const double* vec, double* out, int rows, int64_t cols, double scale,
                     int accumulate, hipStream_t s);
int launch_kl(const double* L, const double* LSp, const double* U, const double* a, int M, int Mp, double* kl,
              double* part, hipStream_t s);
int launch_moments_finish(const double* qpart, const double* mupart, const double* rpart, int nrb, int64_t Np, int64_t N,
                          const double* knn, int branch, double min_var, double* q, double* r, double* varraw,
                          double* mean, double* var, int32_t* zero_word, hipStream_t s);
int launch_moments_bwd_prep(const double* g_mean, const double* g_var, const double* knn, const double* q,
                            const double* varraw, int branch, double min_var, int64_t N, int64_t Np, double* gmu,
                            double* gv, double* gv2, double* cgv, int32_t* nclamped, int zeroed, int32_t* blkact, hipStream_t s);
int launch_reduce_slabs_sym(const double* slabs, int64_t slab_stride, int nslab, int nslab_diag, double* out, int Mp, const int32_t* flag,
                            const double* fallback, hipStream_t s);
int launch_reduce_slabs_sym2(const double* slabs, const double* slabs2, int64_t slab_stride, int nslab, int nslab_diag,
                             double* out, double* out2, int Mp, const int32_t* flag, hipStream_t s);
int launch_add_kl_terms(double* dU, const double* U, double* da, const double* a, const double* gkl, int Mp, hipStream_t s);
int launch_rank1_add(double* X, const double* u, const double* v, int Mp, hipStream_t s);
int launch_dl_from_t2(const double* T2, const double* L, const double* gkl, int M, int Mp, double* dL, hipStream_t s);
int launch_phi(const double* T3, int Mp, double* P, hipStream_t s);
int launch_symmetrize(const double* S, int Mp, double* G, hipStream_t s);
int launch_gls_out(const double* X, const double* LSp, const double* gkl, int M, int Mp, double* gLS, hipStream_t s);
int launch_copy_block(const double* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols, hipStream_t s);
int launch_tril_inplace(double* A, int64_t ld, int n, hipStream_t s);
int launch_mf_combine(const double* Ks, const double* Kn, int64_t ld, const double* s1, const double* s2, const int32_t* l1,
                      const int32_t* l2, const double* ntab, int64_t n1, int64_t n2, double diag, double* out, int64_t ldo,
                      int64_t rows_p, int64_t cols_p, hipStream_t s);
int launch_copy_pad_identity(const double* src, int64_t lds, int n, double* dst, int np, hipStream_t s);
int launch_exact_gp_mll(const double* L, int64_t ld, const double* a, int n, double* mll, hipStream_t s);
int launch_exact_gp_finish(const double* qpart, const double* mupart, int nparts, int64_t ntp, int64_t nt, const double* kss,
                           double* mean, double* var, hipStream_t s);
int launch_mirror_lower(const double* src, int64_t lds, double* dst, int64_t ldd, int64_t row0, int64_t rows, int64_t cols,
                        int64_t n_real, hipStream_t s);
// layer-batched forms (blockIdx.z = layer, workspace pointers + z*zs doubles, user tensors as tables)
int launch_potrf_z(double* A, int64_t ld, int Mp, int M, double* Dinv, double* Ld, int32_t* const* info, int nz, int64_t zs,
                   double* zero0, double* zero1, void* sync, int* inverse_done, hipStream_t s);
int launch_chain_outputs_z(const double* X, const double* LSp, const double* LinvT, const double* da_tot, int M, int Mp,
                           const double* const* gkl, double* const* gLS, double* const* gm, int nz, int64_t zs, hipStream_t s);
int launch_pad_params_z(const double* const* LS, const double* const* m, int M, double* LSp, double* mp, int Mp, int nz,
                        int64_t zs, hipStream_t s);
int launch_trtri_z(const double* L, int64_t ld, int Mp, const double* Dinv, double* Linv, double* T, double* ws,
                   int64_t ws_elems, int nz, int64_t zs, hipStream_t s);
int launch_transpose_z(const double* in, int64_t ldi, double* out, int64_t ldo, int64_t rows, int64_t cols, int nz,
                       int64_t zs, hipStream_t s);
int launch_gemv_rows_z(const double* Mat, int64_t ld, const double* vec, double* out, int rows, int64_t cols, double scale,
                       int accumulate, int nz, int64_t zs, hipStream_t s);
int launch_transpose_pad_z(const double* in, double* out, int Mp, const double* const* LS, const double* const* m, int M,
                           double* LSp, double* mp, int nz, int64_t zs, hipStream_t s);
int launch_transpose_gemv_z(const double* in, double* out, int Mp, const double* Mat, const double* vec, double* vout, int nz,
                            int64_t zs, hipStream_t s);
int launch_kl_z(const double* L, const double* LSp, const double* U, const double* a, int M, int Mp, double* const* kl,
                double* part, int nz, int64_t zs, hipStream_t s);
int launch_dutot_y_z(const double* G1, const double* G2, const double* Hc, const double* U, const double* da, const double* a,
                     const double* const* gkl, int Mp, double* dU, double* da_tot, double* Y, int nz, int64_t zs, hipStream_t s);
int launch_dl_from_t2_z(const double* T2, const double* L, const double* const* gkl, int M, int Mp, double* dL, int nz,
                        int64_t zs, hipStream_t s);
int launch_phi_z(const double* T3, int Mp, double* P, int nz, int64_t zs, hipStream_t s);
int launch_symmetrize_z(const double* S, int Mp, double* G, int nz, int64_t zs, hipStream_t s);

// ---- the tuning / probe events of the C-ABI call this thread is inside (common.h: TuneScope).  Call-scoped, thread-local:
// the library keeps no state between calls (include/mobocmf_hip.h preamble).
static const mobocmf_tuning kDefaultTuning = {(uint32_t)sizeof(mobocmf_tuning), 384, 512, 0, 0, 1024, 32, 0, 1, 0};
static thread_local const mobocmf_tuning* t_tuning = nullptr;
static thread_local void* const* t_probe = nullptr;
const mobocmf_tuning& tune() { return t_tuning ? *t_tuning : kDefaultTuning; }
bool tuning_ok(const mobocmf_tuning* t) {
    if (!t) return true;
    return t->struct_size == sizeof(mobocmf_tuning) && t->small_gemm_max >= 1 && t->small_gemm_max <= 512 &&
           t->small_panel_max >= 1 && t->small_panel_max <= 512 && (t->tile_rows == 0 || t->tile_rows == 64 || t->tile_rows == 128) &&
           t->pair_mode >= 0 && t->pair_mode <= 2 && t->mid_gemm_max >= 0 && t->mid_gemm_max <= 4096 &&
           (t->mid_gemm_waves == 4 || t->mid_gemm_waves == 8 || t->mid_gemm_waves == 32) &&
           (t->syrk_workgroups == 0 || (t->syrk_workgroups >= 16 && t->syrk_workgroups <= 4096)) &&
           (t->sparse_backward == 0 || t->sparse_backward == 1) && (t->potrf_cols == 0 || t->potrf_cols == 1 || t->potrf_cols == 4);
}
TuneScope::TuneScope(const mobocmf_tuning* t, void* const* probe) : prev_t(t_tuning), prev_p(t_probe) {
    t_tuning = t;
    t_probe = probe;
}
TuneScope::~TuneScope() {
    t_tuning = prev_t;
    t_probe = prev_p;
}
// Diagnostic probe (bench.py's per_kernel_instep_ms; mobocmf_layer_desc.probe_events): caller-created HIP events recorded
// around the grid-filling launches of the PANEL halves of the layer call in progress -- per-kernel durations INSIDE a step,
// in the driver's own run, without a profiler.  Off unless the descriptor carries events.
void probe_at(int i, hipStream_t s) {
    if (t_probe && i < MOBOCMF_PROBE_EVENTS && t_probe[i]) (void)hipEventRecord((hipEvent_t)t_probe[i], s);
}
// Tile height of the M x N' panel products (gemm_f64.hip: 128 x 128 tiles, or 64 x 128 tiles with three workgroups per
// CU and the triangular operand resolved in 64-row blocks).  0 = automatic: launch_gemm picks it from the shape
// (gemm_f64.hip tile_rows); 64 / 128 force it for every panel product of the call (mobocmf_tuning.tile_rows).
static void set_pairing(GemmArgs& g) { g.pair_mode = tune().pair_mode; }
static int panel_tile_rows(int, int64_t) { return tune().tile_rows; }

#define TRY(x)              \
    do {                    \
        int _rc = (x);      \
        if (_rc) return _rc; \
    } while (0)

namespace {

struct Bump {
    uintptr_t base;      // 0: size query (nothing is dereferenced, take() returns nullptr)
    size_t off, cap;
    bool ok;
    Bump(void* p, size_t c) : base((uintptr_t)p), off(0), cap(c), ok(true) {}
    double* take(int64_t n) {
        size_t bytes = ((size_t)n * sizeof(double) + 255) & ~(size_t)255;
        if (bytes > cap || off > cap - bytes) { ok = false; off += bytes; return nullptr; }
        double* r = base ? (double*)(base + off) : nullptr;
        off += bytes;
        return r;
    }
};

struct Dims {
    int M, Mp, nrb, H;
    int64_t N, Np, nbase;
    int splitk;
    dim3 ggrid_mn, ggrid_mm;
};

GemmArgs gemm_args(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, int Mr, int64_t Nc,
                   int64_t Kd, int tri, double alpha) {
    GemmArgs g = {};
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.Mr = Mr; g.Nc = Nc; g.Kd = Kd; g.tri = tri; g.alpha = alpha;
    return g;
}

// k slices of the weighted syrk H = A diag(w) A^T (Mp x Mp from Mp x Np): sF for the tiles below the diagonal, *sD for the
// diagonal ones (0: one class).  A diagonal tile's critical wavefronts issue 10/16 of a full tile's MFMAs, so it gets
// fewer, longer slices; together they fill one round of <= 512 resident workgroups (2 per CU).  Multiples of 8 keep every
// k-slice on one XCD (gemm_f64.hip) and are preferred unless another pair is > 10 % shorter.
int syrk_splitk(int Mp, int64_t Np, int* sD) {
    const int nrb = Mp / TILE;
    const int64_t nF = (int64_t)nrb * (nrb - 1) / 2, nD = nrb, ksteps = Np / 16;
    int cap = (int)(ksteps / 8);
    if (cap > 128) cap = 128;
    if (cap < 1) cap = 1;
    *sD = 0;
    if (nF == 0) {      // a single (diagonal) tile
        int sk = 512 < cap ? 512 : cap;
        if (sk >= 8) sk &= ~7;
        return sk;
    }
    // one round of resident workgroups: two per CU -- one per CU for short contractions (N' <= 16384: the slices are a few
    // dozen K steps, half as many slabs to write and add again costs less than the thinner grid; r3 sweep, profiles/)
    int budget = tune().syrk_workgroups;
    if (budget <= 0) budget = Np <= 16384 ? 256 : 512;
    double best = 1e30;
    int bF = 1, bD = 1;
    for (int f = 1; f <= cap; ++f)
        for (int d = 1; d <= f; ++d) {
            if (nF * f + nD * d > budget) break;
            const double cf = (double)((ksteps + f - 1) / f), cd = 0.625 * (double)((ksteps + d - 1) / d);
            double c = cf > cd ? cf : cd;
            if ((f & 7) || (d & 7)) c *= 1.10;
            c += 1e-3 * (f + d);      // ties: fewer slabs to add
            if (c < best) { best = c; bF = f; bD = d; }
        }
    *sD = bD;
    return bF;
}
int64_t syrk_slab_elems(int Mp, int64_t Np) {
    int sD;
    const int sk = syrk_splitk(Mp, Np, &sD);
    return (int64_t)(sk > 16 ? sk : 16) * Mp * Mp;
}
// H (full, symmetric) = A diag(w) A^T; skip (device word, may be NULL): *skip == 0 -> nothing is computed and H = fallback
int weighted_syrk(const double* A, int64_t lda, const double* w, int Mp, int64_t Np, double* slabs, double* H,
                  const int32_t* skip, const double* fallback, const int32_t* kact, hipStream_t s) {
    const int64_t mm = (int64_t)Mp * Mp;
    GemmArgs ga = gemm_args(A, lda, A, lda, slabs, Mp, Mp, Mp, Np, TRI_NONE, 1.0);
    ga.bscale = w; ga.lower_out = 1; ga.sym_out = 1; ga.slab_stride = mm; ga.skip_if_zero = skip;
    ga.kact = kact;      // w is zero throughout the inactive 128-column blocks: the contraction leaves them out
    int sD = 0;
    const int sF = syrk_splitk(Mp, Np, &sD);
    const int nsl = gemm_nt_slabs(ga, sF);      // 1: a small problem goes through whole, no k-slicing
    if (gemm_nt_is_small(ga) && !skip) {      // ... and writes the full symmetric H itself: one launch, no slab, no reduction
        ga.C = H; ga.ldc = Mp; ga.sym_full = 1;
        return launch_gemm(ga, true, 1, s);
    }
    if (nsl > 1) ga.splitk_diag = sD;
    TRY(launch_gemm(ga, true, nsl, s));
    return launch_reduce_slabs_sym(slabs, mm, nsl, (nsl > 1 && sD > 0) ? sD : nsl, H, Mp, skip, fallback, s);
}

// H = A diag(w) A^T and Hc = A diag(w2) A^T, where Hc differs from H only if *flag != 0 (w2 = w except in clamped columns):
// ONE dual launch (the twin half of the grid exits at once when the flag is zero) + ONE reduction writing both.
int weighted_syrk_pair(const double* A, int64_t lda, const double* w, const double* w2, int Mp, int64_t Np, double* slabs,
                       double* slabs2, double* H, double* Hc, const int32_t* flag, const int32_t* kact, hipStream_t s) {
    const int64_t mm = (int64_t)Mp * Mp;
    GemmArgs ga = gemm_args(A, lda, A, lda, slabs, Mp, Mp, Mp, Np, TRI_NONE, 1.0);
    ga.bscale = w; ga.lower_out = 1; ga.sym_out = 1; ga.slab_stride = mm;
    int sD = 0;
    const int sF = syrk_splitk(Mp, Np, &sD);
    const int nsl = gemm_nt_slabs(ga, sF);
    if (gemm_nt_is_small(ga)) {      // small problem: ONE launch of the small-operand kernel forms both products (blockIdx.z)
        // and writes them as full symmetric matrices -- where round 3 spent four launches (two products into slabs, two
        // mirror / copy passes).  The twin is always computed here: at these sizes it costs less than the launch it saves.
        ga.C = H; ga.ldc = Mp; ga.sym_full = 1;
        ga.bscale2 = w2; ga.C2 = Hc;
        return launch_gemm(ga, true, 1, s);
    }
    if (nsl <= 1) {      // a short contraction on the tiled kernel (one slab): the plain pair of launches
        TRY(weighted_syrk(A, lda, w, Mp, Np, slabs, H, nullptr, nullptr, kact, s));
        return weighted_syrk(A, lda, w2, Mp, Np, slabs, Hc, flag, H, kact, s);
    }
    ga.splitk_diag = sD;
    ga.dual_flag = flag; ga.bscale2 = w2; ga.C2 = slabs2; ga.kact = kact;
    TRY(launch_gemm(ga, true, nsl, s));
    return launch_reduce_slabs_sym2(slabs, slabs2, mm, nsl, sD > 0 ? sD : nsl, H, Hc, Mp, flag, s);
}

bool valid_desc(const mobocmf_layer_desc* d) {
    return d && tuning_ok(d->tuning) && (d->kind == 0 || d->kind == 1) && d->d >= 1 && d->d <= MOBOCMF_MAX_D && d->M >= 1 && d->xdiv >= 1 &&
           d->xdiv <= MOBOCMF_MAX_XDIV && d->Np >= 1 && d->Np % d->xdiv == 0 && (d->branch == 0 || d->branch == 1) && d->phase >= 0 &&
           d->phase <= MOBOCMF_PHASE_PANEL_INPUTS;
}

Dims dims_of(const mobocmf_layer_desc* d) {
    Dims D;
    D.M = d->M;
    D.Mp = (int)round_up(d->M, TILE);
    D.nrb = D.Mp / TILE;
    D.N = d->Np;
    D.Np = round_up(d->Np, TILE);
    D.nbase = d->Np / d->xdiv;
    D.H = hyp_len(d->kind, d->d);
    { int sD; D.splitk = syrk_splitk(D.Mp, D.Np, &sD); }
    GramArgs g = {};
    g.xdiv = d->xdiv;
    g.Np = D.Np;
    g.Mp = D.Mp;
    gram_grid(g, &D.ggrid_mn);
    g.xdiv = 1;
    g.Np = D.Mp;
    gram_grid(g, &D.ggrid_mm);
    return D;
}

// ---------------------------------------------------------------------------------------------------------------
// Workspaces.  A layer call has a CHAIN half (the M x M work that depends on the parameters alone) and a PANEL half (the
// M x N' work).  Everything the chain half owns is M x M-sized and lives in ChainWs; with several layers of equal M the
// chain blocks lie a constant stride apart and ONE z-batched sequence of launches runs all their chains (the chain is a
// serial string of latency-bound kernels: batching the layers cuts its length by the number of layers).
//   * single-layer entry points (mobocmf_layer_forward / _backward): ChainWs is carved out of `saved` and `scratch`;
//   * multi-layer entry points (mobocmf_layers_chain_*, mobocmf_layer_panel_*): out of the caller's chain blocks.
// ---------------------------------------------------------------------------------------------------------------
struct ChainWs {
    // state: forward chain -> panel halves and backward chain
    double *L, *Linv, *LinvT, *U, *UT, *LSp, *a, *mp;
    // forward scratch
    double *Dinv, *Ld, *T, *ws, *klpart;
    int64_t ws_elems;
    // panel backward -> chain backward
    double *H, *Hc, *da, *flag;
    // backward scratch
    double *W[8], *da_tot, *slabs, *hyp_part2, *df_part2, *dzf_part2;
    int64_t slab_elems;
};

// K_mn is dead once A = L^-1 K_mn exists (the Gram backward recomputes its exponentials from x, f, Z~): it lives in the
// forward SCRATCH, not in the state kept for backward -- one M x N' panel less per layer (268 MB at C3, 8.6 GB at C4)
struct PanelSaved { double *A, *C, *knn, *q, *r, *varraw; };
struct PanelFwd { double *K, *qpart, *mupart, *rpart; };
struct PanelBwd { double *gmu, *gv, *gv2, *cgv, *dA, *dK, *slabs, *slabs2, *dapart, *hyp_part, *df_part, *dzf_part, *dx_part; int64_t slab_elems; int32_t* blkact; };

void carve_chain_state(Bump& b, const Dims& D, ChainWs& S) {
    int64_t mm = (int64_t)D.Mp * D.Mp;
    S.L = b.take(mm); S.Linv = b.take(mm); S.LinvT = b.take(mm); S.U = b.take(mm); S.UT = b.take(mm); S.LSp = b.take(mm);
    S.a = b.take(D.Mp); S.mp = b.take(D.Mp);
}
void carve_panel_saved(Bump& b, const Dims& D, PanelSaved& S) {
    int64_t mn = (int64_t)D.Mp * D.Np;
    S.A = b.take(mn); S.C = b.take(mn);
    S.knn = b.take(D.Np); S.q = b.take(D.Np); S.r = b.take(D.Np); S.varraw = b.take(D.Np);
}
void carve_chain_fwd(Bump& b, const Dims& D, ChainWs& S) {
    S.Dinv = b.take((int64_t)(D.Mp / NB) * NB * NB);
    S.Ld = b.take((int64_t)(D.Mp / NB) * NB * NB);
    S.T = b.take((int64_t)D.Mp * D.Mp);
    S.ws_elems = (int64_t)16 * D.Mp * D.Mp;
    S.ws = b.take(S.ws_elems);
    S.klpart = b.take(D.Mp);
}
void carve_panel_fwd(Bump& b, const Dims& D, PanelFwd& S) {
    S.K = b.take((int64_t)D.Mp * D.Np);
    S.qpart = b.take((int64_t)4 * D.nrb * D.Np);      // two partial rows per row block of the tile height (64-row tiles:
    S.mupart = b.take((int64_t)4 * D.nrb * D.Np);     // 4 per 128 rows; gemm_f64.hip, EPI_COLSTATS / gemm_colstat_rows)
    S.rpart = b.take((int64_t)4 * D.nrb * D.Np);
}
void carve_chain_bwd_in(Bump& b, const Dims& D, ChainWs& S) {
    int64_t mm = (int64_t)D.Mp * D.Mp;
    S.H = b.take(mm); S.Hc = b.take(mm); S.da = b.take(D.Mp); S.flag = b.take(4);
}
// kind-independent sizes (hyp_part2 for the largest hyper-parameter vector, the f-column partials always): the chain
// blocks of all layers of a model have ONE size
void carve_chain_bwd(Bump& b, const Dims& D, ChainWs& S) {
    int64_t mm = (int64_t)D.Mp * D.Mp;
    for (int i = 0; i < 8; ++i) S.W[i] = b.take(mm);
    S.da_tot = b.take(D.Mp);
    S.hyp_part2 = b.take((int64_t)D.ggrid_mm.x * D.ggrid_mm.y * hyp_len(1, MOBOCMF_MAX_D));
    S.df_part2 = b.take((int64_t)D.ggrid_mm.y * D.Mp);
    S.dzf_part2 = b.take((int64_t)D.ggrid_mm.x * D.Mp);
}
void carve_panel_bwd(Bump& b, const Dims& D, const mobocmf_layer_desc* d, PanelBwd& S) {
    int64_t mm = (int64_t)D.Mp * D.Mp, mn = (int64_t)D.Mp * D.Np;
    S.gmu = b.take(D.Np); S.gv = b.take(D.Np); S.gv2 = b.take(D.Np); S.cgv = b.take(D.Np);
    S.blkact = (int32_t*)b.take((D.Np / TILE + 1) / 2);      // one word per 128 columns (moments_bwd_prep)
    S.dA = b.take(mn); S.dK = b.take(mn);
    S.slab_elems = syrk_slab_elems(D.Mp, D.Np);
    S.slabs = b.take(S.slab_elems);
    S.slabs2 = d->branch == 0 ? b.take(S.slab_elems) : nullptr;      // the clamped-column twin of the syrk (training branch)
    S.dapart = b.take((D.Np <= 8192 ? D.Np / 16 : D.Np / 64) * (int64_t)D.Mp);   // row-dot partials of the dA epilogue (gemm_rowdot_parts)
    S.hyp_part = b.take((int64_t)D.ggrid_mn.x * D.ggrid_mn.y * D.H);
    S.df_part = S.dzf_part = nullptr;
    if (d->kind == 1) {
        S.df_part = b.take((int64_t)D.ggrid_mn.y * D.Np);
        S.dzf_part = b.take((int64_t)D.ggrid_mn.x * D.Mp);
    }
    S.dx_part = d->want_dx ? b.take((int64_t)D.ggrid_mn.y * D.nbase * d->d) : nullptr;
}

// single-layer layout: saved = [chain state | panel saved]; scratch (forward) = [chain fwd | panel fwd];
// scratch (backward) = [panel bwd | chain bwd in | chain bwd] (the chain's k-slice slabs alias the panel's)
bool carve_single_saved(void* saved, size_t bytes, const Dims& D, ChainWs& c, PanelSaved& p, size_t* used = nullptr) {
    Bump b(saved, bytes);
    carve_chain_state(b, D, c);
    carve_panel_saved(b, D, p);
    if (used) *used = b.off;
    return b.ok;
}
bool carve_single_fwd(void* scratch, size_t bytes, const Dims& D, ChainWs& c, PanelFwd& p, size_t* used = nullptr) {
    Bump b(scratch, bytes);
    carve_chain_fwd(b, D, c);
    carve_panel_fwd(b, D, p);
    if (used) *used = b.off;
    return b.ok;
}
bool carve_single_bwd(void* scratch, size_t bytes, const Dims& D, const mobocmf_layer_desc* d, ChainWs& c, PanelBwd& p,
                      size_t* used = nullptr) {
    Bump b(scratch, bytes);
    carve_panel_bwd(b, D, d, p);
    carve_chain_bwd_in(b, D, c);
    carve_chain_bwd(b, D, c);
    c.slabs = p.slabs;
    c.slab_elems = p.slab_elems;
    if (used) *used = b.off;
    return b.ok;
}
// multi-layer layout: one chain block per layer = [state | fwd scratch | bwd in | bwd scratch]; the k-slice slabs of the
// backward chain alias the forward scratch `ws` (dead by then)
bool carve_chain_block(void* block, size_t bytes, const Dims& D, ChainWs& c, size_t* state_bytes = nullptr,
                       size_t* used = nullptr) {
    Bump b(block, bytes);
    carve_chain_state(b, D, c);
    if (state_bytes) *state_bytes = b.off;
    carve_chain_fwd(b, D, c);
    carve_chain_bwd_in(b, D, c);
    carve_chain_bwd(b, D, c);
    c.slabs = c.ws;
    c.slab_elems = c.ws_elems;
    if (used) *used = b.off;
    return b.ok;
}

// per-layer user tensors of a chain call (tables of n entries)
struct ChainIO {
    const mobocmf_layer_desc* const* desc;
    const double* const* Zx; const double* const* zf; const double* const* hyp;
    const double* const* m; const double* const* L_S;
    double* const* kl; int32_t* const* info;                                   // forward
    const double* const* g_kl; double* const* g_zf; double* const* g_hyp;       // backward
    double* const* g_m; double* const* g_LS;
};

GramArgs kmm_gram(const mobocmf_layer_desc* d, const Dims& D, const double* Zx, const double* zf, const double* hyp) {
    GramArgs g = {};
    g.kind = d->kind; g.d = d->d; g.xdiv = 1; g.zdiv = 1;
    g.x = Zx; g.f = zf; g.nbase = D.M; g.Zx = Zx; g.zf = zf; g.M = D.M; g.hyp = hyp;
    g.ldk = D.Mp; g.Mp = D.Mp; g.Np = D.Mp; g.knn = nullptr; g.jitter = d->jitter; g.is_kmm = 1;
    return g;
}

// M x M product of every layer of the batch (operands in the chain blocks)
int chain_gemm(GemmArgs g, bool B_T, const ChainWs& c, int n, int64_t zs, hipStream_t s) {
    if (n > 1) { g.zlayers = n; g.zsA = g.zsB = g.zsC = zs; }
    return launch_gemm_auto(g, B_T, c.slabs, c.slab_elems, s);
}

// CHAIN half, forward, of n layers with the same M: K_mm + jitter, its Cholesky and inverse, U = L^-1 L_S, a = L^-1 m, KL.
// c = workspace of layer 0; layer z works on every pointer + z*zs.
int chain_forward(int n, const ChainWs& c, int64_t zs, const ChainIO& io, const Dims& D, hipStream_t s) {
    const int Mp = D.Mp;
    const int64_t mm = (int64_t)Mp * Mp;
    for (int z = 0; z < n; ++z) {      // the Gram kernels differ per layer (kernel kind, user tensors): one launch each
        GramArgs g = kmm_gram(io.desc[z], D, io.Zx[z], io.zf[z], io.hyp[z]);
        g.K = c.L + z * zs;
        TRY(launch_gram_fwd(g, s));
    }
    // (the last launch of the factorisation also clears L^-1 and U, which are filled on and below the block diagonal only)
    // (the product scratch `ws` is not in use before the triangular inverse: its first bytes are the one-launch factorisation's
    // hand-over words)
    int inv_done = 0;
    TRY(launch_potrf_z(c.L, Mp, Mp, D.M, c.Dinv, c.Ld, io.info, n, zs, c.Linv, c.U, c.ws, &inv_done, s));
    if (!inv_done) TRY(launch_trtri_z(c.L, Mp, Mp, c.Dinv, c.Linv, c.T, c.ws, c.ws_elems, n, zs, s));
    // L^-T, and the user tensors L_S, m -> padded copies (all layers): two independent jobs, one launch
    TRY(launch_transpose_pad_z(c.Linv, c.LinvT, Mp, io.L_S, io.m, D.M, c.LSp, c.mp, n, zs, s));
    // U = L^-1 L_S (lower x lower), a = L^-1 m
    {
        GemmArgs ga = gemm_args(c.Linv, Mp, c.LSp, Mp, c.U, Mp, Mp, Mp, Mp, TRI_LOWER_A | TRI_LOWER_B, 1.0);
        ga.lower_out = 1;
        ga.Kreal = D.M;      // the contraction's padded tail multiplies zeros (small problems)
        GemmArgs gz = ga;
        if (n > 1) { gz.zlayers = n; gz.zsA = gz.zsB = gz.zsC = zs; }
        TRY(launch_gemm_auto(gz, false, c.ws, c.ws_elems, s));
    }
    TRY(launch_transpose_gemv_z(c.U, c.UT, Mp, c.Linv, c.mp, c.a, n, zs, s));      // U^T and a = L^-1 m: one launch
    TRY(launch_kl_z(c.L, c.LSp, c.U, c.a, D.M, Mp, io.kl, c.klpart, n, zs, s));
    return MOBOCMF_OK;
}

// PANEL half, forward: K_mn, A = L^-1 K (+ q, mean partials), C = U^T A (+ r partials), moments
int panel_forward(const mobocmf_layer_desc* desc, const Dims& D, const ChainWs& c, const PanelSaved& P, const PanelFwd& F,
                  const double* x, const double* f, const double* Zx, const double* zf, const double* hyp, double* mean,
                  double* var, bool block_layout, hipStream_t s) {
    const int Mp = D.Mp;
    const int64_t Np = D.Np;
    GramArgs g = {};
    g.kind = desc->kind; g.d = desc->d; g.zdiv = 1; g.Zx = Zx; g.zf = zf; g.M = D.M; g.hyp = hyp; g.Mp = Mp;
    g.xdiv = desc->xdiv; g.x = x; g.f = f; g.nbase = D.nbase; g.jitter = desc->jitter;
    g.K = F.K; g.ldk = Np; g.Np = Np; g.knn = P.knn; g.is_kmm = 0;
    probe_at(9, s);
    TRY(launch_gram_fwd(g, s));
    probe_at(0, s);
    GemmArgs ga = gemm_args(c.Linv, Mp, F.K, Np, P.A, Np, Mp, Np, Mp, TRI_LOWER_A, 1.0);
    ga.epi = EPI_COLSTATS; ga.colsq_part = F.qpart; ga.coldot_part = F.mupart; ga.avec = c.a;
    ga.Kreal = D.M;      // rows >= M of K_mn (and of A, C, dA below) are zero padding
    ga.rm = panel_tile_rows(Mp, Np);
    set_pairing(ga);
    TRY(launch_gemm(ga, false, 1, s));
    probe_at(1, s);
    GemmArgs gc = gemm_args(c.UT, Mp, P.A, Np, P.C, Np, Mp, Np, Mp, TRI_UPPER_A, 1.0);
    gc.epi = EPI_COLSTATS; gc.colsq_part = F.rpart; gc.coldot_part = nullptr; gc.avec = c.a;
    gc.Kreal = D.M;
    gc.stream_out = (desc->branch == 0 && Np * Mp * 8 >= ((int64_t)64 << 20)) ? 1 : 0;   // C is next read in backward
    gc.rm = ga.rm; gc.pair_mode = ga.pair_mode;
    TRY(launch_gemm(gc, false, 1, s));
    probe_at(2, s);
    // chain-block layout: the clamped-column counter of the backward lives in the block and is cleared here
    TRY(launch_moments_finish(F.qpart, F.mupart, F.rpart, gemm_colstat_rows(ga), Np, D.N, P.knn, desc->branch, desc->min_var, P.q, P.r,
                              P.varraw, mean, var, block_layout ? (int32_t*)c.flag : nullptr, s));
    return MOBOCMF_OK;
}

// PANEL half, backward: dA, the weighted syrk(s) H / Hc and da (unless the parameters are constants), dK, Gram backward of
// K_mn and k_nn.  Writes g_f, g_x and -- overwriting -- the K_mn share of g_hyp / g_zf.
int panel_backward(const mobocmf_layer_desc* desc, const Dims& D, const ChainWs& c, const PanelSaved& P, const PanelBwd& B,
                   bool inputs_only, bool block_layout, const double* x, const double* f, const double* Zx, const double* zf,
                   const double* hyp, const double* g_mean, const double* g_var, double* g_f, double* g_zf, double* g_hyp,
                   double* g_x, hipStream_t s) {
    const int Mp = D.Mp;
    const int64_t Np = D.Np, mm = (int64_t)Mp * Mp;
    int32_t* nclamped = (int32_t*)c.flag;
    double* Hc = desc->branch != 0 ? c.H : c.Hc;
    // act: one word per 128 columns, zero where every upstream gradient of the block is exactly zero (the top layer of a
    // multi-fidelity model only gets gradient from the rows scored at ITS fidelity): those blocks' shares of dA, H, da,
    // dK and the Gram backward are exactly zero and are not computed
    const int32_t* act = tune().sparse_backward ? B.blkact : nullptr;
    TRY(launch_moments_bwd_prep(g_mean, g_var, P.knn, P.q, P.varraw, desc->branch, desc->min_var, D.N, Np, B.gmu, B.gv,
                                B.gv2, B.cgv, nclamped, block_layout ? 1 : 0, (int32_t*)act, s));
    // dA = 2 U (C diag(gv)) + a gmu^T - 2 A diag(cgv)
    int da_parts = 0;
    {
        GemmArgs ga = gemm_args(c.U, Mp, P.C, Np, B.dA, Np, Mp, Np, Mp, TRI_LOWER_A, 2.0);
        ga.bscale = B.gv; ga.epi = EPI_DA; ga.avec = c.a; ga.gmu = B.gmu; ga.cgv = B.cgv; ga.Aaux = P.A;
        ga.Kreal = D.M;
        ga.rowdot_part = inputs_only ? nullptr : B.dapart;      // da = A gmu rides in the epilogue (it reads A anyway)
        ga.colact = act;
        ga.rm = panel_tile_rows(Mp, Np);
        set_pairing(ga);
        probe_at(3, s);
        TRY(launch_gemm(ga, false, 1, s));
        probe_at(4, s);
        da_parts = inputs_only ? 0 : gemm_rowdot_parts(ga);      // summed in the call's last launch (below): da is read by the
    }                                                           // CHAIN half only
    // H = A diag(gv) A^T  (weighted syrk, split-K over N').  Both M x M contractions of the backward reduce to it:
    //   dU = 2 tril(A diag(gv) C^T) = 2 tril(H U),   dA A^T = 2 U U^T H + a da^T - 2 Hc,  Hc = A diag(cgv) A^T.
    // Hc differs from H only when clamp(k_nn - q, 0) is active in some column: its syrk is skipped on the device
    // (skip_if_zero) when no column is clamped.
    if (!inputs_only) {
        probe_at(5, s);
        if (desc->branch == 0) TRY(weighted_syrk_pair(P.A, Np, B.gv, B.cgv, Mp, Np, B.slabs, B.slabs2, c.H, Hc, nclamped, act, s));
        else TRY(weighted_syrk(P.A, Np, B.gv, Mp, Np, B.slabs, c.H, nullptr, nullptr, act, s));
        probe_at(6, s);
    }
    // dK = L^-T dA
    {
        GemmArgs ga = gemm_args(c.LinvT, Mp, B.dA, Np, B.dK, Np, Mp, Np, Mp, TRI_UPPER_A, 1.0);
        ga.Kreal = D.M;
        ga.rm = panel_tile_rows(Mp, Np);
        set_pairing(ga);
        ga.colact = act;      // inactive column blocks of dK stay unwritten: the Gram backward below does not read them
        probe_at(7, s);
        TRY(launch_gemm(ga, false, 1, s));
        probe_at(8, s);
    }
    // Gram backward of K_mn and k_nn
    GramArgs g = {};
    g.kind = desc->kind; g.d = desc->d; g.zdiv = 1; g.Zx = Zx; g.zf = zf; g.M = D.M; g.hyp = hyp; g.Mp = Mp;
    g.xdiv = desc->xdiv; g.x = x; g.f = f; g.nbase = D.nbase;
    g.ldk = Np; g.Np = Np; g.G = B.dK; g.gknn = B.cgv; g.colact = act;
    g.hyp_part = B.hyp_part; g.df_part = B.df_part; g.dzf_part = B.dzf_part; g.dx_part = B.dx_part;
    TRY(launch_gram_bwd(g, desc->want_dx != 0, s));
    probe_at(10, s);
    SumTask tk[5];
    int nt = 0;
    if (da_parts > 0) tk[nt++] = {B.dapart, da_parts, Mp, nullptr, 0, 0, c.da, Mp, 0};      // da = A g_mean (row-dot partials of dA)
    tk[nt++] = {B.hyp_part, (int64_t)D.ggrid_mn.x * D.ggrid_mn.y, D.H, nullptr, 0, 0, g_hyp, D.H, 0};
    if (desc->kind == 1) {
        tk[nt++] = {B.df_part, D.ggrid_mn.y, Np, nullptr, 0, 0, g_f, D.N, 0};
        tk[nt++] = {B.dzf_part, D.ggrid_mn.x, Mp, nullptr, 0, 0, g_zf, D.M, 0};
    }
    if (desc->want_dx)
        tk[nt++] = {B.dx_part, D.ggrid_mn.y, D.nbase * desc->d, nullptr, 0, 0, g_x, D.nbase * desc->d, 0};
    return launch_sum_partials_multi(tk, nt, s);
}

// CHAIN half, backward, of n layers:  dL = -tril(L^-T [dA A^T + dU_tot U^T + da_tot a^T]) + gkl diag(1/L_ii), g_m, g_LS,
// Cholesky backward, Gram backward of K_mm.  H, Hc, da come from the PANEL halves (zero[z]: layer z had none -- no
// upstream mean / var gradient).  acc[z]: add to g_hyp / g_zf (the PANEL half wrote its share there) instead of overwriting.
// fold[z]: zf of layer z IS the variational mean of layer z - 1 of this batch (Z~_z = [Z_x, m_{z-1}], mfdgp_hidden_layer.py:
// 555-556): its whole gradient -- the K_mm share formed here plus the PANEL half's share waiting in g_zf[z] if acc[z] -- is
// added to g_m[z - 1] in the last launch, and g_zf[z] is not written (autograd would add the two tensors in a launch of its own).
int chain_backward(int n, const ChainWs& c, int64_t zs, const ChainIO& io, const Dims& D, const bool* zero, const int* acc,
                   const bool* fold, hipStream_t s) {
    const int Mp = D.Mp;
    const int64_t mm = (int64_t)Mp * Mp;
    double *G1 = c.W[0], *G2 = c.W[1], *X = c.W[2], *dU = c.W[3], *Y = c.W[4], *T1 = c.W[5], *T2 = c.W[6], *LT = c.W[7];
    double *dL = G1, *T4 = G2, *Gm = X;   // reused once their first content is dead
    bool any_clamp_branch = false;
    for (int z = 0; z < n; ++z) {
        if (zero[z]) {
            TRY(launch_zero32(c.H + z * zs, mm * 2, s));
            TRY(launch_zero32(c.Hc + z * zs, mm * 2, s));
            TRY(launch_zero32(c.da + z * zs, (int64_t)Mp * 2, s));
        }
        any_clamp_branch = any_clamp_branch || io.desc[z]->branch == 0;
    }
    // layers on the eval branch keep Hc == H: the batched launches read Hc, so copy it there
    for (int z = 0; z < n; ++z)
        if (io.desc[z]->branch != 0 && !zero[z] && (n > 1 || any_clamp_branch))
            TRY(launch_copy_block(c.H + z * zs, Mp, c.Hc + z * zs, Mp, Mp, Mp, s));
    const double* Hc = (n == 1 && io.desc[0]->branch != 0) ? c.H : c.Hc;
    {
        GemmArgs g1 = gemm_args(c.UT, Mp, c.H, Mp, G1, Mp, Mp, Mp, Mp, TRI_UPPER_A, 1.0);          // G1 = U^T H
        g1.Kreal = D.M;      // the contraction's padded tail multiplies zeros (small problems)
        TRY(chain_gemm(g1, false, c, n, zs, s));
        GemmArgs g2 = gemm_args(c.U, Mp, G1, Mp, G2, Mp, Mp, Mp, Mp, TRI_LOWER_A, 1.0);            // G2 = U U^T H
        g2.Kreal = D.M;
        TRY(chain_gemm(g2, false, c, n, zs, s));
        // X = H U is G1^T (H symmetric): dU_tot = 2 tril(X) + gkl U reads it out of G1, in the launch that also forms Y
        TRY(launch_dutot_y_z(G1, G2, Hc, c.U, c.da, c.a, io.g_kl, Mp, dU, c.da_tot, Y, n, zs, s));
    }
    // Y += dU_tot U^T (as dU_tot x UT: both products in the A B form) and T1 = L^-T dU_tot are independent: one launch for
    // the pair where the mid-size kernel takes them.  g_LS = tril(T1) - gkl diag(1/LS_ii) and g_m = L^-T da_tot: the user
    // tensors of all layers in one launch
    {
        GemmArgs g4 = gemm_args(dU, Mp, c.UT, Mp, Y, Mp, Mp, Mp, Mp, TRI_LOWER_A | TRI_UPPER_B, 1.0);
        g4.accumulate = 1;
        g4.Kreal = D.M;
        GemmArgs ga = gemm_args(c.LinvT, Mp, dU, Mp, T1, Mp, Mp, Mp, Mp, TRI_UPPER_A | TRI_LOWER_B, 1.0);
        ga.lower_out = 1;
        ga.Kreal = D.M;
        if (n > 1) {
            g4.zlayers = ga.zlayers = n;
            g4.zsA = g4.zsB = g4.zsC = ga.zsA = ga.zsB = ga.zsC = zs;
        }
        TRY(launch_gemm_auto_pair(g4, ga, c.slabs, c.slab_elems, s));
        TRY(launch_chain_outputs_z(T1, c.LSp, c.LinvT, c.da_tot, D.M, Mp, io.g_kl, io.g_LS, io.g_m, n, zs, s));
    }
    // dL
    {
        GemmArgs ga = gemm_args(c.LinvT, Mp, Y, Mp, T2, Mp, Mp, Mp, Mp, TRI_UPPER_A, 1.0);
        ga.lower_out = 1;
        ga.Kreal = D.M;
        TRY(chain_gemm(ga, false, c, n, zs, s));
        TRY(launch_dl_from_t2_z(T2, c.L, io.g_kl, D.M, Mp, dL, n, zs, s));
    }
    // Cholesky backward: dKmm = sym(L^-T Phi(L^T dL) L^-1)
    {
        TRY(launch_transpose_z(c.L, Mp, LT, Mp, Mp, Mp, n, zs, s));
        GemmArgs ga = gemm_args(LT, Mp, dL, Mp, T2, Mp, Mp, Mp, Mp, TRI_UPPER_A | TRI_LOWER_B, 1.0);
        ga.Kreal = D.M;
        TRY(chain_gemm(ga, false, c, n, zs, s));
        TRY(launch_phi_z(T2, Mp, T1, n, zs, s));
        GemmArgs gb = gemm_args(c.LinvT, Mp, T1, Mp, T4, Mp, Mp, Mp, Mp, TRI_UPPER_A | TRI_LOWER_B, 1.0);
        gb.Kreal = D.M;
        TRY(chain_gemm(gb, false, c, n, zs, s));
        GemmArgs gc = gemm_args(T4, Mp, c.Linv, Mp, T2, Mp, Mp, Mp, Mp, TRI_LOWER_B, 1.0);
        gc.Kreal = D.M;
        TRY(chain_gemm(gc, false, c, n, zs, s));
        TRY(launch_symmetrize_z(T2, Mp, Gm, n, zs, s));
    }
    // Gram backward of K_mm (both arguments are Z~): one launch per layer (kernel kind, user tensors), then the partial
    // sums of all layers in one
    SumTask tk[2 * MAX_ZL];
    int nt = 0;
    for (int z = 0; z < n; ++z) {
        const mobocmf_layer_desc* d = io.desc[z];
        GramArgs g = kmm_gram(d, D, io.Zx[z], io.zf[z], io.hyp[z]);
        g.G = Gm + z * zs; g.gknn = nullptr;
        g.hyp_part = c.hyp_part2 + z * zs; g.df_part = c.df_part2 + z * zs; g.dzf_part = c.dzf_part2 + z * zs;
        g.dx_part = nullptr;
        TRY(launch_gram_bwd(g, false, s));
        const int H = hyp_len(d->kind, d->d);
        tk[nt++] = {g.hyp_part, (int64_t)D.ggrid_mm.x * D.ggrid_mm.y, H, nullptr, 0, 0, io.g_hyp[z], H, acc[z]};
        if (d->kind == 1) {    // both arguments of K_mm are Z~: the row-side and the column-side partials land in g_zf
            if (fold[z])       // ... or, with the PANEL half's share, on top of g_m[z - 1] (written by chain_outputs above)
                tk[nt++] = {g.df_part, D.ggrid_mm.y, Mp, g.dzf_part, D.ggrid_mm.x, Mp, io.g_m[z - 1], D.M, 1,
                            acc[z] ? io.g_zf[z] : nullptr};
            else
                tk[nt++] = {g.df_part, D.ggrid_mm.y, Mp, g.dzf_part, D.ggrid_mm.x, Mp, io.g_zf[z], D.M, acc[z], nullptr};
        }
    }
    return launch_sum_partials_multi(tk, nt, s);
}

bool same_chain_shape(int n, const mobocmf_layer_desc* const* d) {
    for (int z = 0; z < n; ++z)
        if (!valid_desc(d[z]) || d[z]->M != d[0]->M) return false;
    return true;
}

}  // namespace

// ---- K10: the full eval-branch predictive covariance  cov = K_nn - A^T A + C^T C  (A = L^-1 K_mn, C = U^T A).
// Everything is formed TRANSPOSED, rows = data points, so that both contractions over the M inducing points are A B^T
// products with the contraction index contiguous (no transposes of M x N' panels):
//   K_nm (Gram with the data rows on the row side);  A^T = K_nm L^-T  (A B^T with B = L^-1, logical B upper-triangular);
//   C^T = A^T U  (A B with B = U lower-triangular);  cov = K_nn - A^T (A^T)^T + C^T (C^T)^T  on the LOWER 128-tiles only
// (symmetric: half the flops of the two dense products), one column panel of width covpanel_cols at a time -- the
// scratch is O(N' x panel), not O(N'^2), so N' is not capped -- and mirrored into the caller's matrix.
namespace {
int64_t gcd64(int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; }
// panel width: a multiple of 128 (tiles) and of xdiv (a panel starts at a base row of the replicated inputs), ~2048 columns
int64_t covpanel_cols(const mobocmf_layer_desc* d, int64_t Np) {
    const int64_t l = (int64_t)TILE / gcd64(TILE, d->xdiv) * d->xdiv;
    int64_t w = l * (2048 / l > 1 ? 2048 / l : 1);
    return w < Np ? w : Np;
}
struct CovWs { double *Knm, *AT, *CT, *Cv; };
bool carve_cov(void* scratch, size_t bytes, const mobocmf_layer_desc* d, const Dims& D, CovWs& w, size_t* used = nullptr) {
    Bump b(scratch, bytes);
    const int64_t nm = D.Np * D.Mp;
    w.Knm = b.take(nm); w.AT = b.take(nm); w.CT = b.take(nm);
    w.Cv = b.take(D.Np * covpanel_cols(d, D.Np));
    if (used) *used = b.off;
    return b.ok;
}
}  // namespace

extern "C" {

int mobocmf_version(void) { return 200; }

int mobocmf_device_arch_ok(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    const char* a = p.gcnArchName;
    return (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? 1 : 0;
}

int mobocmf_layer_workspace_bytes(const mobocmf_layer_desc* desc, size_t* saved_bytes, size_t* scratch_bytes) {
    if (!valid_desc(desc) || !saved_bytes || !scratch_bytes) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(desc->tuning, desc->probe_events);
    Dims D = dims_of(desc);
    const size_t big = ~(size_t)0 >> 1;
    ChainWs c;
    PanelSaved P;
    PanelFwd F;
    PanelBwd B;
    size_t sv = 0, sf = 0, sb = 0;
    carve_single_saved(nullptr, big, D, c, P, &sv);
    carve_single_fwd(nullptr, big, D, c, F, &sf);
    carve_single_bwd(nullptr, big, D, desc, c, B, &sb);
    *saved_bytes = sv;
    *scratch_bytes = sf > sb ? sf : sb;
    return MOBOCMF_OK;
}

int mobocmf_layer_chain_state_bytes(const mobocmf_layer_desc* desc, size_t* bytes) {
    if (!valid_desc(desc) || !bytes) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(desc->tuning, desc->probe_events);
    Dims D = dims_of(desc);
    Bump b(nullptr, ~(size_t)0 >> 1);
    ChainWs c;
    carve_chain_state(b, D, c);
    *bytes = b.off;
    return MOBOCMF_OK;
}

int mobocmf_layer_forward(const mobocmf_layer_desc* desc, const double* x, const double* f, const double* Zx,
                          const double* zf, const double* hyp, const double* m, const double* L_S, double* mean,
                          double* var, double* kl, int32_t* info, void* saved, size_t saved_bytes, void* scratch,
                          size_t scratch_bytes, mobocmf_stream_t stream) {
    if (!valid_desc(desc) || desc->phase == MOBOCMF_PHASE_CHAIN_ONLY || !Zx || !hyp || !saved || !scratch)
        return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(desc->tuning, desc->probe_events);
    const bool do_chain = desc->phase == MOBOCMF_PHASE_ALL || desc->phase == MOBOCMF_PHASE_CHAIN;
    const bool do_panel = desc->phase != MOBOCMF_PHASE_CHAIN;
    if (do_chain && (!m || !L_S || !kl || !info)) return MOBOCMF_BAD_ARG;
    if (do_panel && (!x || !mean || !var)) return MOBOCMF_BAD_ARG;
    if (desc->kind == 1 && (!zf || (do_panel && !f))) return MOBOCMF_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    Dims D = dims_of(desc);
    ChainWs c = {};
    PanelSaved P;
    PanelFwd F;
    if (!carve_single_saved(saved, saved_bytes, D, c, P) || !carve_single_fwd(scratch, scratch_bytes, D, c, F))
        return MOBOCMF_WORKSPACE_TOO_SMALL;
    if (do_chain) {
        ChainIO io = {};
        io.desc = &desc; io.Zx = &Zx; io.zf = &zf; io.hyp = &hyp; io.m = &m; io.L_S = &L_S; io.kl = &kl; io.info = &info;
        TRY(chain_forward(1, c, 0, io, D, s));
    }
    if (!do_panel) return MOBOCMF_OK;
    return panel_forward(desc, D, c, P, F, x, f, Zx, zf, hyp, mean, var, false, s);
}

int mobocmf_layer_backward(const mobocmf_layer_desc* desc, const double* x, const double* f, const double* Zx,
                           const double* zf, const double* hyp, const double* m, const double* L_S,
                           const double* g_mean, const double* g_var, const double* g_kl, double* g_f, double* g_zf,
                           double* g_hyp, double* g_m, double* g_LS, double* g_x, void* saved, size_t saved_bytes,
                           void* scratch, size_t scratch_bytes, mobocmf_stream_t stream) {
    if (!valid_desc(desc) || !Zx || !hyp || !g_hyp || !saved || !scratch) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(desc->tuning, desc->probe_events);
    const bool inputs_only = desc->phase == MOBOCMF_PHASE_PANEL_INPUTS;   // parameters are constants: no H / Hc / da
    const bool do_panel = desc->phase == MOBOCMF_PHASE_ALL || desc->phase == MOBOCMF_PHASE_PANEL || inputs_only;
    const bool do_chain = desc->phase == MOBOCMF_PHASE_ALL || desc->phase == MOBOCMF_PHASE_CHAIN ||
                          desc->phase == MOBOCMF_PHASE_CHAIN_ONLY;
    if (do_panel && (!x || !g_mean || !g_var || (desc->want_dx && !g_x))) return MOBOCMF_BAD_ARG;
    if (do_chain && (!g_kl || !g_m || !g_LS)) return MOBOCMF_BAD_ARG;
    if (desc->kind == 1 && (!zf || !g_zf || (do_panel && (!f || !g_f)))) return MOBOCMF_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    Dims D = dims_of(desc);
    ChainWs c = {};
    PanelSaved P;
    PanelBwd B;
    if (!carve_single_saved(saved, saved_bytes, D, c, P) || !carve_single_bwd(scratch, scratch_bytes, D, desc, c, B))
        return MOBOCMF_WORKSPACE_TOO_SMALL;
    if (do_panel)
        TRY(panel_backward(desc, D, c, P, B, inputs_only, false, x, f, Zx, zf, hyp, g_mean, g_var, g_f, g_zf, g_hyp, g_x, s));
    if (!do_chain) return MOBOCMF_OK;
    ChainIO io = {};
    io.desc = &desc; io.Zx = &Zx; io.zf = &zf; io.hyp = &hyp; io.m = &m; io.L_S = &L_S;
    io.g_kl = &g_kl; io.g_zf = &g_zf; io.g_hyp = &g_hyp; io.g_m = &g_m; io.g_LS = &g_LS;
    const bool zero = desc->phase == MOBOCMF_PHASE_CHAIN_ONLY;     // no upstream mean/var gradient: H = Hc = 0, da = 0
    const int acc = desc->phase == MOBOCMF_PHASE_ALL ? 1 : 0;       // split: the chain half reports its own g_hyp / g_zf
    const bool fold = false;
    return chain_backward(1, c, 0, io, D, &zero, &acc, &fold, s);
}

// ------------------------------------------------------------------------------------------------- multi-layer forms
int mobocmf_chain_block_bytes(const mobocmf_layer_desc* desc, size_t* block_bytes, size_t* state_bytes) {
    if (!valid_desc(desc) || !block_bytes) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(desc->tuning, desc->probe_events);
    Dims D = dims_of(desc);
    ChainWs c;
    size_t st = 0, used = 0;
    carve_chain_block(nullptr, ~(size_t)0 >> 1, D, c, &st, &used);
    *block_bytes = used;
    if (state_bytes) *state_bytes = st;
    return MOBOCMF_OK;
}

int mobocmf_panel_workspace_bytes(const mobocmf_layer_desc* desc, size_t* saved_bytes, size_t* scratch_bytes) {
    if (!valid_desc(desc) || !saved_bytes || !scratch_bytes) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(desc->tuning, desc->probe_events);
    Dims D = dims_of(desc);
    const size_t big = ~(size_t)0 >> 1;
    Bump bs(nullptr, big), bf(nullptr, big), bb(nullptr, big);
    PanelSaved P;
    PanelFwd F;
    PanelBwd B;
    carve_panel_saved(bs, D, P);
    carve_panel_fwd(bf, D, F);
    carve_panel_bwd(bb, D, desc, B);
    *saved_bytes = bs.off;
    *scratch_bytes = bf.off > bb.off ? bf.off : bb.off;
    return MOBOCMF_OK;
}

int mobocmf_layers_chain_forward(int32_t n, const mobocmf_layer_desc* const* desc, const double* const* Zx,
                                 const double* const* zf, const double* const* hyp, const double* const* m,
                                 const double* const* L_S, double* const* kl, int32_t* const* info, void* blocks,
                                 size_t block_stride, size_t blocks_bytes, mobocmf_stream_t stream) {
    if (n < 1 || n > MAX_ZL || !desc || !Zx || !zf || !hyp || !m || !L_S || !kl || !info || !blocks || (block_stride & 255))
        return MOBOCMF_BAD_ARG;
    if (!same_chain_shape(n, desc)) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(desc[0]->tuning);      // one z-batched sequence of launches: the first layer's tuning serves all
    if (blocks_bytes / (size_t)n < block_stride) return MOBOCMF_WORKSPACE_TOO_SMALL;      // layer z works at blocks + z * stride
    for (int z = 0; z < n; ++z)
        if (!Zx[z] || !hyp[z] || !m[z] || !L_S[z] || !kl[z] || !info[z] || (desc[z]->kind == 1 && !zf[z])) return MOBOCMF_BAD_ARG;
    Dims D = dims_of(desc[0]);
    ChainWs c = {};
    if (!carve_chain_block(blocks, block_stride, D, c)) return MOBOCMF_WORKSPACE_TOO_SMALL;
    ChainIO io = {};
    io.desc = desc; io.Zx = Zx; io.zf = zf; io.hyp = hyp; io.m = m; io.L_S = L_S; io.kl = kl; io.info = info;
    return chain_forward(n, c, (int64_t)(block_stride / sizeof(double)), io, D, (hipStream_t)stream);
}

int mobocmf_layers_chain_backward(int32_t n, const mobocmf_layer_desc* const* desc, const double* const* Zx,
                                  const double* const* zf, const double* const* hyp, const double* const* g_kl,
                                  const int32_t* had_panel, double* const* g_zf, double* const* g_hyp, double* const* g_m,
                                  double* const* g_LS, void* blocks, size_t block_stride, size_t blocks_bytes,
                                  mobocmf_stream_t stream) {
    if (n < 1 || n > MAX_ZL || !desc || !Zx || !zf || !hyp || !g_kl || !had_panel || !g_zf || !g_hyp || !g_m || !g_LS ||
        !blocks || (block_stride & 255))
        return MOBOCMF_BAD_ARG;
    if (!same_chain_shape(n, desc)) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(desc[0]->tuning);      // one z-batched sequence of launches: the first layer's tuning serves all
    if (blocks_bytes / (size_t)n < block_stride) return MOBOCMF_WORKSPACE_TOO_SMALL;
    bool zero[MAX_ZL], fold[MAX_ZL];
    int acc[MAX_ZL];
    for (int z = 0; z < n; ++z) {
        if (!Zx[z] || !hyp[z] || !g_kl[z] || !g_hyp[z] || !g_m[z] || !g_LS[z] || (desc[z]->kind == 1 && (!zf[z] || !g_zf[z])))
            return MOBOCMF_BAD_ARG;
        const int hp = had_panel[z] & 3;
        if (hp == 3 || (had_panel[z] & ~7)) return MOBOCMF_BAD_ARG;
        zero[z] = hp == 0;
        acc[z] = hp == 2 ? 1 : 0;      // 2: g_hyp / g_zf hold the PANEL half's share already -- add to it
        fold[z] = (had_panel[z] & 4) != 0;
        if (fold[z] && (z == 0 || desc[z]->kind != 1)) return MOBOCMF_BAD_ARG;
    }
    Dims D = dims_of(desc[0]);
    ChainWs c = {};
    if (!carve_chain_block(blocks, block_stride, D, c)) return MOBOCMF_WORKSPACE_TOO_SMALL;
    ChainIO io = {};
    io.desc = desc; io.Zx = Zx; io.zf = zf; io.hyp = hyp;
    io.g_kl = g_kl; io.g_zf = g_zf; io.g_hyp = g_hyp; io.g_m = g_m; io.g_LS = g_LS;
    return chain_backward(n, c, (int64_t)(block_stride / sizeof(double)), io, D, zero, acc, fold, (hipStream_t)stream);
}

int mobocmf_layer_panel_forward(const mobocmf_layer_desc* desc, const double* x, const double* f, const double* Zx,
                                const double* zf, const double* hyp, double* mean, double* var, void* chain_block,
                                size_t block_bytes, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                                mobocmf_stream_t stream) {
    if (!valid_desc(desc) || !x || !Zx || !hyp || !mean || !var || !chain_block || !saved || !scratch) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(desc->tuning, desc->probe_events);
    if (desc->kind == 1 && (!zf || !f)) return MOBOCMF_BAD_ARG;
    Dims D = dims_of(desc);
    ChainWs c = {};
    if (!carve_chain_block(chain_block, block_bytes, D, c)) return MOBOCMF_WORKSPACE_TOO_SMALL;
    Bump bs(saved, saved_bytes), bf(scratch, scratch_bytes);
    PanelSaved P;
    PanelFwd F;
    carve_panel_saved(bs, D, P);
    carve_panel_fwd(bf, D, F);
    if (!bs.ok || !bf.ok) return MOBOCMF_WORKSPACE_TOO_SMALL;
    return panel_forward(desc, D, c, P, F, x, f, Zx, zf, hyp, mean, var, true, (hipStream_t)stream);
}

int mobocmf_layer_panel_backward(const mobocmf_layer_desc* desc, const double* x, const double* f, const double* Zx,
                                 const double* zf, const double* hyp, const double* g_mean, const double* g_var,
                                 double* g_f, double* g_zf, double* g_hyp, double* g_x, void* chain_block,
                                 size_t block_bytes, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                                 mobocmf_stream_t stream) {
    if (!valid_desc(desc) || !x || !Zx || !hyp || !g_mean || !g_var || !g_hyp || !chain_block || !saved || !scratch ||
        (desc->want_dx && !g_x))
        return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(desc->tuning, desc->probe_events);
    if (desc->kind == 1 && (!zf || !f || !g_f || !g_zf)) return MOBOCMF_BAD_ARG;
    Dims D = dims_of(desc);
    ChainWs c = {};
    if (!carve_chain_block(chain_block, block_bytes, D, c)) return MOBOCMF_WORKSPACE_TOO_SMALL;
    Bump bs(saved, saved_bytes), bb(scratch, scratch_bytes);
    PanelSaved P;
    PanelBwd B;
    carve_panel_saved(bs, D, P);
    carve_panel_bwd(bb, D, desc, B);
    if (!bs.ok || !bb.ok) return MOBOCMF_WORKSPACE_TOO_SMALL;
    return panel_backward(desc, D, c, P, B, desc->phase == MOBOCMF_PHASE_PANEL_INPUTS, true, x, f, Zx, zf, hyp, g_mean, g_var,
                          g_f, g_zf, g_hyp, g_x, (hipStream_t)stream);
}

int mobocmf_predictive_covariance_workspace_bytes(const mobocmf_layer_desc* desc, size_t* scratch_bytes) {
    if (!valid_desc(desc) || !scratch_bytes) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(desc->tuning, desc->probe_events);
    Dims D = dims_of(desc);
    CovWs w;
    carve_cov(nullptr, ~(size_t)0 >> 1, desc, D, w, scratch_bytes);
    return MOBOCMF_OK;
}

int mobocmf_predictive_covariance(const mobocmf_layer_desc* desc, const double* x, const double* f, const double* Zx,
                                  const double* zf, const double* hyp, double* cov, int64_t ldcov, const void* chain_state,
                                  size_t chain_state_bytes, void* scratch, size_t scratch_bytes, mobocmf_stream_t stream) {
    if (!valid_desc(desc) || !x || !Zx || !hyp || !cov || !chain_state || !scratch || ldcov < desc->Np) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(desc->tuning, desc->probe_events);
    if (desc->kind == 1 && (!f || !zf)) return MOBOCMF_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    Dims D = dims_of(desc);
    ChainWs cw = {};
    {
        Bump b((void*)chain_state, chain_state_bytes);
        carve_chain_state(b, D, cw);
        if (!b.ok) return MOBOCMF_WORKSPACE_TOO_SMALL;
    }
    CovWs w;
    if (!carve_cov(scratch, scratch_bytes, desc, D, w)) return MOBOCMF_WORKSPACE_TOO_SMALL;
    const int Mp = D.Mp;
    const int64_t Np = D.Np;
    if (Np > 0x7fffff80) return MOBOCMF_BAD_ARG;
    // K_nm: data rows on the row side (replicated zdiv-fold like the layer's columns), Z~ on the column side
    GramArgs g = {};
    g.kind = desc->kind; g.d = desc->d; g.xdiv = 1; g.zdiv = desc->xdiv;
    g.x = Zx; g.f = zf; g.nbase = D.M; g.Zx = x; g.zf = f; g.M = (int)D.N; g.hyp = hyp;
    g.K = w.Knm; g.ldk = Mp; g.Mp = (int)Np; g.Np = Mp; g.knn = nullptr; g.jitter = 0.0; g.is_kmm = 0;
    TRY(launch_gram_fwd(g, s));
    {   // A^T[n][j] = sum_k K_nm[n][k] L^-1[j][k]
        GemmArgs ga = gemm_args(w.Knm, Mp, cw.Linv, Mp, w.AT, Mp, (int)Np, Mp, Mp, TRI_UPPER_B, 1.0);
        TRY(launch_gemm(ga, true, 1, s));
        // C^T[n][j] = sum_k A^T[n][k] U[k][j]
        GemmArgs gc = gemm_args(w.AT, Mp, cw.U, Mp, w.CT, Mp, (int)Np, Mp, Mp, TRI_LOWER_B, 1.0);
        TRY(launch_gemm(gc, false, 1, s));
    }
    const int64_t W = covpanel_cols(desc, Np);
    for (int64_t J0 = 0; J0 < Np && J0 < D.N; J0 += W) {
        const int64_t Wp = Np - J0 < W ? Np - J0 : W, R = Np - J0, b0 = J0 / desc->xdiv;
        // K_nn block: rows J0.. (row side), columns J0..J0+Wp (column side); J0 is a multiple of xdiv
        GramArgs k = {};
        k.kind = desc->kind; k.d = desc->d; k.xdiv = desc->xdiv; k.zdiv = desc->xdiv;
        k.x = x + b0 * desc->d; k.f = f ? f + J0 : nullptr; k.nbase = D.nbase - b0;
        k.Zx = x + b0 * desc->d; k.zf = f ? f + J0 : nullptr; k.M = (int)(D.N - J0); k.hyp = hyp;
        k.K = w.Cv; k.ldk = Wp; k.Mp = (int)R; k.Np = Wp; k.knn = nullptr; k.jitter = 0.0; k.is_kmm = 0;
        TRY(launch_gram_fwd(k, s));
        GemmArgs ga = gemm_args(w.AT + J0 * Mp, Mp, w.AT + J0 * Mp, Mp, w.Cv, Wp, (int)R, Wp, Mp, TRI_NONE, -1.0);
        ga.accumulate = 1; ga.lower_out = 1;
        TRY(launch_gemm(ga, true, 1, s));
        GemmArgs gc = gemm_args(w.CT + J0 * Mp, Mp, w.CT + J0 * Mp, Mp, w.Cv, Wp, (int)R, Wp, Mp, TRI_NONE, 1.0);
        gc.accumulate = 1; gc.lower_out = 1;
        TRY(launch_gemm(gc, true, 1, s));
        TRY(launch_mirror_lower(w.Cv, Wp, cov, ldcov, J0, R, Wp, D.N, s));
    }
    return MOBOCMF_OK;
}

}  // extern "C"

// ---- Exact-GP comparison baselines on the layer's kernels ("exact-GP baselines reuse K1 / K3 / K4"):
// the Gram matrices come from mobocmf_gram_forward (K1), the factorisation and the triangular inverse are the chain's
// (K3), the predictive moments are the layer's A = L^-1 K product with the column-statistics epilogue (K4 / K5).
namespace {
struct ExactState { double *L, *Linv, *a, *yp; };
struct ExactFactorWs { double *Dinv, *Ld, *T, *ws; int64_t ws_elems; };
void carve_exact_state(Bump& b, int np, ExactState& S) {
    const int64_t mm = (int64_t)np * np;
    S.L = b.take(mm); S.Linv = b.take(mm); S.a = b.take(np); S.yp = b.take(np);
}
void carve_exact_factor(Bump& b, int np, ExactFactorWs& W) {
    W.Dinv = b.take((int64_t)(np / NB) * NB * NB);
    W.Ld = b.take((int64_t)(np / NB) * NB * NB);
    W.T = b.take((int64_t)np * np);
    W.ws_elems = (int64_t)16 * np * np;
    W.ws = b.take(W.ws_elems);
}
struct ExactPredictWs { double *Kp, *A, *qpart, *mupart; };
void carve_exact_predict(Bump& b, int np, int64_t ntp, ExactPredictWs& W) {
    W.Kp = b.take((int64_t)np * ntp); W.A = b.take((int64_t)np * ntp);
    W.qpart = b.take((int64_t)4 * (np / TILE) * ntp); W.mupart = b.take((int64_t)4 * (np / TILE) * ntp);
}
}  // namespace

extern "C" {

int mobocmf_mf_kernel_combine(int64_t n1, int64_t n2, const double* Ks, const double* Kn, int64_t ld, const double* s1,
                              const double* s2, const int32_t* l1, const int32_t* l2, const double* ntab, double diag,
                              double* out, int64_t ldo, int64_t rows_p, int64_t cols_p, mobocmf_stream_t stream) {
    if (n1 < 1 || n2 < 1 || !Ks || !Kn || !l1 || !l2 || !ntab || !out || ld < n2 || rows_p < n1 || cols_p < n2 || ldo < cols_p)
        return MOBOCMF_BAD_ARG;
    return launch_mf_combine(Ks, Kn, ld, s1, s2, l1, l2, ntab, n1, n2, diag, out, ldo, rows_p, cols_p, (hipStream_t)stream);
}

int mobocmf_exact_gp_workspace_bytes(int32_t n, int64_t nt, size_t* state_bytes, size_t* scratch_bytes) {
    if (n < 1 || nt < 0 || !state_bytes || !scratch_bytes) return MOBOCMF_BAD_ARG;
    const int np = (int)round_up(n, TILE);
    const int64_t ntp = round_up(nt > 0 ? nt : 1, TILE);
    const size_t big = ~(size_t)0 >> 1;
    Bump bs(nullptr, big), bf(nullptr, big), bp(nullptr, big);
    ExactState S; ExactFactorWs F; ExactPredictWs P;
    carve_exact_state(bs, np, S);
    carve_exact_factor(bf, np, F);
    carve_exact_predict(bp, np, ntp, P);
    *state_bytes = bs.off;
    *scratch_bytes = bf.off > bp.off ? bf.off : bp.off;
    return MOBOCMF_OK;
}

/* K: the n x n training covariance INCLUDING the noise on its diagonal (ld ldk); y[n].  state <- L, L^-1, a = L^-1 y;
 * mll[0] = log N(y | 0, K); info as in the layer calls (0 or the failed pivot). */
int mobocmf_exact_gp_factor(int32_t n, const double* K, int64_t ldk, const double* y, double* mll, int32_t* info, void* state,
                            size_t state_bytes, void* scratch, size_t scratch_bytes, const mobocmf_tuning* tuning,
                            mobocmf_stream_t stream) {
    if (n < 1 || !K || ldk < n || !y || !mll || !info || !state || !scratch || !tuning_ok(tuning)) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(tuning);
    hipStream_t s = (hipStream_t)stream;
    const int np = (int)round_up(n, TILE);
    Bump bs(state, state_bytes), bf(scratch, scratch_bytes);
    ExactState S; ExactFactorWs F;
    carve_exact_state(bs, np, S);
    carve_exact_factor(bf, np, F);
    if (!bs.ok || !bf.ok) return MOBOCMF_WORKSPACE_TOO_SMALL;
    // padded copy of K (identity on the padded diagonal), padded y
    TRY(launch_copy_pad_identity(K, ldk, n, S.L, np, s));
    TRY(launch_pad_vec(y, n, S.yp, np, s));
    int inv_done = 0;
    TRY(launch_potrf_z(S.L, np, np, n, F.Dinv, F.Ld, &info, 1, 0, S.Linv, nullptr, F.ws, &inv_done, s));
    if (!inv_done) TRY(launch_trtri(S.L, np, np, F.Dinv, S.Linv, F.T, F.ws, F.ws_elems, s));
    TRY(launch_gemv_rows(S.Linv, np, S.yp, S.a, np, np, 1.0, 0, s));
    return launch_exact_gp_mll(S.L, np, S.a, n, mll, s);
}

/* Posterior moments of the latent function at nt points: Kts [n x nt] = k(train, test) (ld), kss[nt] = prior variances:
 * mean = Kts^T K^-1 y = (L^-1 Kts)^T a,  var = kss - colsum((L^-1 Kts)^2)  -- the layer's triangular MFMA product with its
 * column-statistics epilogue. */
int mobocmf_exact_gp_predict(int32_t n, int64_t nt, const double* Kts, int64_t ld, const double* kss, double* mean,
                             double* var, const void* state, size_t state_bytes, void* scratch, size_t scratch_bytes,
                             const mobocmf_tuning* tuning, mobocmf_stream_t stream) {
    if (n < 1 || nt < 1 || !Kts || ld < nt || !kss || !mean || !var || !state || !scratch || !tuning_ok(tuning)) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(tuning);
    hipStream_t s = (hipStream_t)stream;
    const int np = (int)round_up(n, TILE);
    const int64_t ntp = round_up(nt, TILE);
    Bump bs((void*)state, state_bytes), bp(scratch, scratch_bytes);
    ExactState S; ExactPredictWs P;
    carve_exact_state(bs, np, S);
    carve_exact_predict(bp, np, ntp, P);
    if (!bs.ok || !bp.ok) return MOBOCMF_WORKSPACE_TOO_SMALL;
    TRY(launch_zero32(P.Kp, (int64_t)np * ntp * 2, s));
    TRY(launch_copy_block(Kts, ld, P.Kp, ntp, n, nt, s));
    GemmArgs ga = gemm_args(S.Linv, np, P.Kp, ntp, P.A, ntp, np, ntp, np, TRI_LOWER_A, 1.0);
    ga.epi = EPI_COLSTATS; ga.colsq_part = P.qpart; ga.coldot_part = P.mupart; ga.avec = S.a;
    ga.Kreal = n;
    TRY(launch_gemm(ga, false, 1, s));
    return launch_exact_gp_finish(P.qpart, P.mupart, gemm_colstat_rows(ga), ntp, nt, kss, mean, var, s);
}

}  // extern "C"

extern "C" {

int mobocmf_gemm_f64(int32_t tri, int32_t trans_b, int32_t Mr, int64_t Nc, int64_t Kd, const double* A, int64_t lda,
                     const double* B, int64_t ldb, double* C, int64_t ldc, double alpha, int32_t accumulate,
                     const mobocmf_tuning* tuning, mobocmf_stream_t stream) {
    if (!A || !B || !C || Mr <= 0 || Nc <= 0 || Kd <= 0 || (lda & 1) || (ldb & 1) || !tuning_ok(tuning)) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(tuning);
    GemmArgs g = gemm_args(A, lda, B, ldb, C, ldc, Mr, Nc, Kd, tri, alpha);
    g.accumulate = accumulate;
    // small / mid-size operands take the kernels the M x M chain uses for them; no workspace: never k-sliced
    return launch_gemm_auto(g, trans_b != 0, nullptr, 0, (hipStream_t)stream);
}

int mobocmf_gram_forward(int32_t kind, int32_t d, const double* x1, const double* f1, int64_t n1, const double* x2,
                         const double* f2, int64_t n2, const double* hyp, double* K, int64_t ldk, mobocmf_stream_t stream) {
    if ((kind != 0 && kind != 1) || d < 1 || d > MOBOCMF_MAX_D || !x1 || !x2 || !hyp || !K || n1 < 1 || n2 < 1 ||
        n1 > 0x7fffffff || ldk < n2 || (kind == 1 && (!f1 || !f2)))
        return MOBOCMF_BAD_ARG;
    GramArgs g = {};
    g.kind = kind; g.d = d; g.xdiv = 1; g.zdiv = 1;
    g.x = x2; g.f = f2; g.nbase = n2; g.Zx = x1; g.zf = f1; g.M = (int)n1; g.hyp = hyp;
    g.K = K; g.ldk = ldk; g.Mp = (int)round_up(n1, 32); g.Np = n2; g.knn = nullptr; g.jitter = 0.0; g.is_kmm = 0;
    return launch_gram_fwd(g, (hipStream_t)stream);
}

int mobocmf_gemm_f64_epilogue(int32_t tri, int32_t epi, int32_t Mr, int64_t Nc, int64_t Kd, const double* A, int64_t lda,
                              const double* B, int64_t ldb, double* C, int64_t ldc, double alpha, int32_t stream_out,
                              double* colsq_part, double* coldot_part, const double* avec, const double* bscale,
                              const double* gmu, const double* cgv, const double* Aaux, double* rowdot_part,
                              const int32_t* col_activity, const mobocmf_tuning* tuning, mobocmf_stream_t stream) {
    if (!A || !B || !C || Mr <= 0 || Nc <= 0 || Kd <= 0 || (lda & 1) || (ldb & 1) || !tuning_ok(tuning)) return MOBOCMF_BAD_ARG;
    if (epi < EPI_STORE || epi > EPI_DA) return MOBOCMF_BAD_ARG;
    if (epi == EPI_COLSTATS && (!colsq_part || !avec)) return MOBOCMF_BAD_ARG;
    if (epi == EPI_DA && (!avec || !gmu || !cgv || !Aaux)) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(tuning);
    GemmArgs g = gemm_args(A, lda, B, ldb, C, ldc, Mr, Nc, Kd, tri, alpha);
    g.epi = epi; g.stream_out = stream_out;
    g.colsq_part = colsq_part; g.coldot_part = coldot_part; g.avec = avec;
    g.bscale = epi == EPI_DA ? bscale : nullptr; g.gmu = gmu; g.cgv = cgv; g.Aaux = Aaux; g.rowdot_part = rowdot_part;
    g.rm = panel_tile_rows(Mr, Nc);
    set_pairing(g);
    g.colact = col_activity;
    return launch_gemm(g, false, 1, (hipStream_t)stream);
}

int mobocmf_gemm_colstat_rows(int32_t tri, int32_t Mr, int64_t Nc, int64_t Kd, const mobocmf_tuning* tuning, int32_t* rows) {
    if (!rows || Mr <= 0 || Nc <= 0 || Kd <= 0 || !tuning_ok(tuning)) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(tuning);
    GemmArgs g = gemm_args(nullptr, Kd, nullptr, Nc, nullptr, Nc, Mr, Nc, Kd, tri, 1.0);
    g.epi = EPI_COLSTATS;
    g.rm = panel_tile_rows(Mr, Nc);
    *rows = gemm_colstat_rows(g);
    return MOBOCMF_OK;
}

int mobocmf_syrk_workspace_bytes(int32_t Mr, int64_t Kd, const mobocmf_tuning* tuning, size_t* bytes) {
    if (!bytes || Mr <= 0 || Kd <= 0 || Mr % TILE || Kd % TILE || !tuning_ok(tuning)) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(tuning);
    *bytes = (size_t)syrk_slab_elems(Mr, Kd) * sizeof(double);
    return MOBOCMF_OK;
}

int mobocmf_syrk_weighted_f64(int32_t Mr, int64_t Kd, const double* A, int64_t lda, const double* w, double* H,
                              void* workspace, int64_t workspace_bytes, const int32_t* k_activity,
                              const mobocmf_tuning* tuning, mobocmf_stream_t stream) {
    if (!A || !w || !H || !workspace || Mr <= 0 || Kd <= 0 || Mr % TILE || Kd % TILE || (lda & 1) || !tuning_ok(tuning))
        return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(tuning);
    if (workspace_bytes < syrk_slab_elems(Mr, Kd) * (int64_t)sizeof(double)) return MOBOCMF_WORKSPACE_TOO_SMALL;
    return weighted_syrk(A, lda, w, Mr, Kd, (double*)workspace, H, nullptr, nullptr, k_activity, (hipStream_t)stream);
}

int mobocmf_tuning_init(mobocmf_tuning* t) {
    if (!t) return MOBOCMF_BAD_ARG;
    *t = kDefaultTuning;
    return MOBOCMF_OK;
}

#ifdef MOBOCMF_HOST_FUZZ
// Host-sanitizer build only (tools/asan_host.sh; never in the product library): carves every workspace layout out of HOST
// buffers of the given sizes and writes the first and the last double of every region, so that AddressSanitizer sees any
// region that reaches past the size the *_bytes functions reported.  No kernel is launched, no GPU is needed.
static void touch(double* p, int64_t n, int* cnt) {
    if (!p || n <= 0) return;
    p[0] = 1.0;
    p[n - 1] = 2.0;
    ++*cnt;
}
int mobocmf_debug_touch_workspaces(const mobocmf_layer_desc* desc, void* saved, size_t saved_bytes, void* scratch,
                                   size_t scratch_bytes, void* block, size_t block_bytes, void* psaved, size_t psaved_bytes,
                                   void* pscratch, size_t pscratch_bytes, void* cov, size_t cov_bytes, int32_t* regions) {
    if (!valid_desc(desc) || !regions) return MOBOCMF_BAD_ARG;
    TuneScope tune_scope(desc->tuning, desc->probe_events);
    Dims D = dims_of(desc);
    const int64_t mm = (int64_t)D.Mp * D.Mp, mn = (int64_t)D.Mp * D.Np;
    int n = 0;
    auto chain_state = [&](ChainWs& c) {
        touch(c.L, mm, &n); touch(c.Linv, mm, &n); touch(c.LinvT, mm, &n); touch(c.U, mm, &n); touch(c.UT, mm, &n);
        touch(c.LSp, mm, &n); touch(c.a, D.Mp, &n); touch(c.mp, D.Mp, &n);
    };
    auto chain_fwd = [&](ChainWs& c) {
        touch(c.Dinv, (int64_t)(D.Mp / NB) * NB * NB, &n); touch(c.Ld, (int64_t)(D.Mp / NB) * NB * NB, &n);
        touch(c.T, mm, &n); touch(c.ws, c.ws_elems, &n); touch(c.klpart, D.Mp, &n);
    };
    auto chain_bwd = [&](ChainWs& c) {
        touch(c.H, mm, &n); touch(c.Hc, mm, &n); touch(c.da, D.Mp, &n); touch(c.flag, 4, &n);
        for (int i = 0; i < 8; ++i) touch(c.W[i], mm, &n);
        touch(c.da_tot, D.Mp, &n);
        touch(c.hyp_part2, (int64_t)D.ggrid_mm.x * D.ggrid_mm.y * hyp_len(1, MOBOCMF_MAX_D), &n);
        touch(c.df_part2, (int64_t)D.ggrid_mm.y * D.Mp, &n); touch(c.dzf_part2, (int64_t)D.ggrid_mm.x * D.Mp, &n);
        touch(c.slabs, c.slab_elems, &n);
    };
    auto panel_saved = [&](PanelSaved& P) {
        touch(P.A, mn, &n); touch(P.C, mn, &n); touch(P.knn, D.Np, &n); touch(P.q, D.Np, &n); touch(P.r, D.Np, &n);
        touch(P.varraw, D.Np, &n);
    };
    auto panel_fwd = [&](PanelFwd& F) {
        touch(F.K, mn, &n); touch(F.qpart, (int64_t)2 * D.nrb * D.Np, &n); touch(F.mupart, (int64_t)2 * D.nrb * D.Np, &n);
        touch(F.rpart, (int64_t)2 * D.nrb * D.Np, &n);
    };
    auto panel_bwd = [&](PanelBwd& B) {
        touch(B.gmu, D.Np, &n); touch(B.gv, D.Np, &n); touch(B.gv2, D.Np, &n); touch(B.cgv, D.Np, &n);
        touch((double*)B.blkact, (D.Np / TILE + 1) / 2, &n);
        touch(B.dA, mn, &n); touch(B.dK, mn, &n); touch(B.slabs, B.slab_elems, &n); touch(B.slabs2, B.slab_elems, &n);
        touch(B.dapart, (D.Np <= 8192 ? D.Np / 16 : D.Np / 64) * (int64_t)D.Mp, &n);
        touch(B.hyp_part, (int64_t)D.ggrid_mn.x * D.ggrid_mn.y * D.H, &n);
        if (desc->kind == 1) { touch(B.df_part, (int64_t)D.ggrid_mn.y * D.Np, &n); touch(B.dzf_part, (int64_t)D.ggrid_mn.x * D.Mp, &n); }
        if (desc->want_dx) touch(B.dx_part, (int64_t)D.ggrid_mn.y * D.nbase * desc->d, &n);
    };
    if (saved && scratch) {       // single-layer layout (mobocmf_layer_workspace_bytes)
        ChainWs c = {}; PanelSaved P; PanelFwd F; PanelBwd B;
        if (!carve_single_saved(saved, saved_bytes, D, c, P) || !carve_single_fwd(scratch, scratch_bytes, D, c, F))
            return MOBOCMF_WORKSPACE_TOO_SMALL;
        chain_state(c); panel_saved(P); chain_fwd(c); panel_fwd(F);
        ChainWs c2 = {};
        if (!carve_single_saved(saved, saved_bytes, D, c2, P) || !carve_single_bwd(scratch, scratch_bytes, D, desc, c2, B))
            return MOBOCMF_WORKSPACE_TOO_SMALL;
        panel_bwd(B); chain_bwd(c2);
    }
    if (block) {                  // multi-layer layout (mobocmf_chain_block_bytes / mobocmf_panel_workspace_bytes)
        ChainWs c = {};
        if (!carve_chain_block(block, block_bytes, D, c)) return MOBOCMF_WORKSPACE_TOO_SMALL;
        chain_state(c); chain_fwd(c); chain_bwd(c);
    }
    if (psaved && pscratch) {
        Bump bs(psaved, psaved_bytes), bf(pscratch, pscratch_bytes), bb(pscratch, pscratch_bytes);
        PanelSaved P; PanelFwd F; PanelBwd B;
        carve_panel_saved(bs, D, P); carve_panel_fwd(bf, D, F); carve_panel_bwd(bb, D, desc, B);
        if (!bs.ok || !bf.ok || !bb.ok) return MOBOCMF_WORKSPACE_TOO_SMALL;
        panel_saved(P); panel_fwd(F); panel_bwd(B);
    }
    if (cov) {                    // predictive covariance (mobocmf_predictive_covariance_workspace_bytes)
        CovWs w;
        if (!carve_cov(cov, cov_bytes, desc, D, w)) return MOBOCMF_WORKSPACE_TOO_SMALL;
        touch(w.Knm, mn, &n); touch(w.AT, mn, &n); touch(w.CT, mn, &n); touch(w.Cv, D.Np * covpanel_cols(desc, D.Np), &n);
    }
    *regions = n;
    return MOBOCMF_OK;
}
#endif

int mobocmf_check_info(const int32_t* info, int32_t* pivot, mobocmf_stream_t stream) {
    int32_t h = 0;
    if (hipMemcpyAsync(&h, info, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
        return MOBOCMF_HIP_ERROR;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return MOBOCMF_HIP_ERROR;
    if (pivot) *pivot = h;
    return h == 0 ? MOBOCMF_OK : MOBOCMF_NOT_PD;
}

}  // extern "C"
