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
                          double* mean, double* var, hipStream_t s);
int launch_moments_bwd_prep(const double* g_mean, const double* g_var, const double* knn, const double* q,
                            const double* varraw, int branch, double min_var, int64_t N, int64_t Np, double* gmu,
                            double* gv, double* gv2, double* cgv, int32_t* nclamped, hipStream_t s);
int launch_reduce_slabs_sym(const double* slabs, int64_t slab_stride, int nslab, double* out, int Mp, const int32_t* flag,
                            const double* fallback, hipStream_t s);
int launch_dutot(const double* X, const double* U, const double* da, const double* a, const double* gkl, int Mp,
                 double* dU, double* da_tot, hipStream_t s);
int launch_y_combine(const double* G2, const double* Hc, const double* a, const double* da, const double* da_tot, int Mp,
                     double* Y, hipStream_t s);
int launch_add_kl_terms(double* dU, const double* U, double* da, const double* a, const double* gkl, int Mp, hipStream_t s);
int launch_rank1_add(double* X, const double* u, const double* v, int Mp, hipStream_t s);
int launch_dl_from_t2(const double* T2, const double* L, const double* gkl, int M, int Mp, double* dL, hipStream_t s);
int launch_phi(const double* T3, int Mp, double* P, hipStream_t s);
int launch_symmetrize(const double* S, int Mp, double* G, hipStream_t s);
int launch_gls_out(const double* X, const double* LSp, const double* gkl, int M, int Mp, double* gLS, hipStream_t s);
int launch_copy_block(const double* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols, hipStream_t s);
int launch_tril_inplace(double* A, int64_t ld, int n, hipStream_t s);

#define TRY(x)              \
    do {                    \
        int _rc = (x);      \
        if (_rc) return _rc; \
    } while (0)

namespace {

struct Bump {
    char* base;
    size_t off, cap;
    bool ok;
    Bump(void* p, size_t c) : base((char*)p), off(0), cap(c), ok(true) {}
    double* take(int64_t n) {
        size_t bytes = ((size_t)n * sizeof(double) + 255) & ~(size_t)255;
        if (off + bytes > cap) { ok = false; off += bytes; return nullptr; }
        double* r = (double*)(base + off);
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

bool valid_desc(const mobocmf_layer_desc* d) {
    return d && (d->kind == 0 || d->kind == 1) && d->d >= 1 && d->d <= MOBOCMF_MAX_D && d->M >= 1 && d->xdiv >= 1 &&
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
    int ntl = D.nrb * (D.nrb + 1) / 2;
    int64_t ksteps = D.Np / 16;
    int sk = 512 / ntl;   // one round of <= 512 resident workgroups (2 per CU): fewer, longer slices = fewer slabs to add
    if (sk > ksteps / 8) sk = (int)(ksteps / 8);
    if (sk > 128) sk = 128;
    // multiples of 8 keep every k-slice on one XCD (gemm_f64.hip); not at the price of leaving > 1/4 of the slots empty
    // (36 tiles at M = 1024: 14 slices fill 504 of 512 slots, 8 only 288)
    if (sk >= 8 && 4 * (sk & ~7) >= 3 * sk) sk &= ~7;
    if (sk < 1) sk = 1;
    D.splitk = sk;
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

struct Saved {
    double *L, *Linv, *LinvT, *U, *UT, *LSp, *a, *mp, *K, *A, *C, *knn, *q, *r, *varraw;
};

// The CHAIN half's state comes first and does not depend on N': its leading bytes may be copied between the `saved`
// buffers of calls that share the parameters (mobocmf_layer_chain_state_bytes).
void carve_chain_state(Bump& b, const Dims& D, Saved& S) {
    int64_t mm = (int64_t)D.Mp * D.Mp;
    S.L = b.take(mm); S.Linv = b.take(mm); S.LinvT = b.take(mm); S.U = b.take(mm); S.UT = b.take(mm); S.LSp = b.take(mm);
    S.a = b.take(D.Mp); S.mp = b.take(D.Mp);
}

bool carve_saved(Bump& b, const Dims& D, Saved& S) {
    int64_t mn = (int64_t)D.Mp * D.Np;
    carve_chain_state(b, D, S);
    S.K = b.take(mn); S.A = b.take(mn); S.C = b.take(mn);
    S.knn = b.take(D.Np); S.q = b.take(D.Np); S.r = b.take(D.Np); S.varraw = b.take(D.Np);
    return b.ok;
}

struct ScratchF {
    double *Dinv, *Ld, *T, *qpart, *mupart, *rpart, *ws;
    int64_t ws_elems;
};
bool carve_scratch_fwd(Bump& b, const Dims& D, ScratchF& S) {
    S.Dinv = b.take((int64_t)(D.Mp / NB) * NB * NB);
    S.Ld = b.take((int64_t)(D.Mp / NB) * NB * NB);
    S.T = b.take((int64_t)D.Mp * D.Mp);
    S.qpart = b.take((int64_t)D.nrb * D.Np);
    S.mupart = b.take((int64_t)D.nrb * D.Np);
    S.rpart = b.take((int64_t)D.nrb * D.Np);
    S.ws_elems = (int64_t)16 * D.Mp * D.Mp;
    S.ws = b.take(S.ws_elems);
    return b.ok;
}

struct ScratchB {
    double *gmu, *gv, *gv2, *cgv, *dA, *dK, *slabs, *W[10], *da, *da_tot, *flag, *hyp_part, *df_part, *dzf_part, *dx_part,
        *hyp_part2, *df_part2, *dzf_part2, *gzf_tmp, *dapart;
};
bool carve_scratch_bwd(Bump& b, const Dims& D, const mobocmf_layer_desc* d, ScratchB& S) {
    int64_t mm = (int64_t)D.Mp * D.Mp, mn = (int64_t)D.Mp * D.Np;
    S.gmu = b.take(D.Np); S.gv = b.take(D.Np); S.gv2 = b.take(D.Np); S.cgv = b.take(D.Np);
    S.dA = b.take(mn); S.dK = b.take(mn);
    S.slabs = b.take((int64_t)(D.splitk > 16 ? D.splitk : 16) * mm);
    for (int i = 0; i < 10; ++i) S.W[i] = b.take(mm);
    S.da = b.take(D.Mp);
    S.da_tot = b.take(D.Mp);
    S.flag = b.take(4);
    S.dapart = b.take((D.Np <= 8192 ? D.Np / 16 : D.Np / 64) * (int64_t)D.Mp);   // row-dot partials of the dA epilogue (gemm_rowdot_parts)
    S.hyp_part = b.take((int64_t)D.ggrid_mn.x * D.ggrid_mn.y * D.H);
    S.hyp_part2 = b.take((int64_t)D.ggrid_mm.x * D.ggrid_mm.y * D.H);
    S.df_part = S.dzf_part = S.df_part2 = S.dzf_part2 = S.gzf_tmp = nullptr;
    if (d->kind == 1) {
        S.df_part = b.take((int64_t)D.ggrid_mn.y * D.Np);
        S.dzf_part = b.take((int64_t)D.ggrid_mn.x * D.Mp);
        S.df_part2 = b.take((int64_t)D.ggrid_mm.y * D.Mp);
        S.dzf_part2 = b.take((int64_t)D.ggrid_mm.x * D.Mp);
        S.gzf_tmp = b.take(D.Mp);
    }
    S.dx_part = d->want_dx ? b.take((int64_t)D.ggrid_mn.y * D.nbase * d->d) : nullptr;
    return b.ok;
}

GemmArgs gemm_args(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, int Mr, int64_t Nc,
                   int64_t Kd, int tri, double alpha) {
    GemmArgs g = {};
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.Mr = Mr; g.Nc = Nc; g.Kd = Kd; g.tri = tri; g.alpha = alpha;
    return g;
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
    Dims D = dims_of(desc);
    Bump bs(nullptr, ~(size_t)0 >> 1);
    Saved S;
    carve_saved(bs, D, S);
    *saved_bytes = bs.off;
    Bump bf(nullptr, ~(size_t)0 >> 1), bb(nullptr, ~(size_t)0 >> 1);
    ScratchF F;
    ScratchB B;
    carve_scratch_fwd(bf, D, F);
    carve_scratch_bwd(bb, D, desc, B);
    // predictive covariance scratch: A^T, C^T (Np x Mp) + padded cov (Np x Np), only sized for Np <= 16384
    size_t cov = 0;
    if (D.Np <= 16384) cov = (size_t)(2 * D.Np * D.Mp + D.Np * D.Np) * sizeof(double) + 1024;
    size_t m = bf.off > bb.off ? bf.off : bb.off;
    *scratch_bytes = m > cov ? m : cov;
    return MOBOCMF_OK;
}

int mobocmf_layer_chain_state_bytes(const mobocmf_layer_desc* desc, size_t* bytes) {
    if (!valid_desc(desc) || !bytes) return MOBOCMF_BAD_ARG;
    Dims D = dims_of(desc);
    Bump b(nullptr, ~(size_t)0 >> 1);
    Saved S;
    carve_chain_state(b, D, S);
    *bytes = b.off;
    return MOBOCMF_OK;
}

int mobocmf_layer_forward(const mobocmf_layer_desc* desc, const double* x, const double* f, const double* Zx,
                          const double* zf, const double* hyp, const double* m, const double* L_S, double* mean,
                          double* var, double* kl, int32_t* info, void* saved, size_t saved_bytes, void* scratch,
                          size_t scratch_bytes, mobocmf_stream_t stream) {
    if (!valid_desc(desc) || desc->phase == MOBOCMF_PHASE_CHAIN_ONLY || !Zx || !hyp || !saved || !scratch)
        return MOBOCMF_BAD_ARG;
    const bool do_chain = desc->phase == MOBOCMF_PHASE_ALL || desc->phase == MOBOCMF_PHASE_CHAIN;
    const bool do_panel = desc->phase != MOBOCMF_PHASE_CHAIN;
    if (do_chain && (!m || !L_S || !kl || !info)) return MOBOCMF_BAD_ARG;
    if (do_panel && (!x || !mean || !var)) return MOBOCMF_BAD_ARG;
    if (desc->kind == 1 && (!zf || (do_panel && !f))) return MOBOCMF_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    Dims D = dims_of(desc);
    Bump bs(saved, saved_bytes), bf(scratch, scratch_bytes);
    Saved S;
    ScratchF F;
    if (!carve_saved(bs, D, S) || !carve_scratch_fwd(bf, D, F)) return MOBOCMF_WORKSPACE_TOO_SMALL;
    const int Mp = D.Mp;
    const int64_t Np = D.Np, mm = (int64_t)Mp * Mp;

    GramArgs g = {};
    g.kind = desc->kind; g.d = desc->d; g.xdiv = 1; g.zdiv = 1;
    g.x = Zx; g.f = zf; g.nbase = D.M; g.Zx = Zx; g.zf = zf; g.M = D.M; g.hyp = hyp;
    g.K = S.L; g.ldk = Mp; g.Mp = Mp; g.Np = Mp; g.knn = nullptr; g.jitter = desc->jitter; g.is_kmm = 1;
    if (do_chain) {
        // K_mm + jitter -> L (in place Cholesky)
        TRY(launch_gram_fwd(g, s));
        TRY(launch_potrf(S.L, Mp, Mp, D.M, F.Dinv, F.Ld, info, s));
        TRY(launch_zero32(S.Linv, mm * 2, s));
        TRY(launch_trtri(S.L, Mp, Mp, F.Dinv, S.Linv, F.T, F.ws, F.ws_elems, s));
        TRY(launch_transpose(S.Linv, Mp, S.LinvT, Mp, Mp, Mp, s));
        TRY(launch_pad_tril(L_S, D.M, D.M, S.LSp, Mp, s));
        TRY(launch_pad_vec(m, D.M, S.mp, Mp, s));
        // U = L^-1 L_S (lower x lower), a = L^-1 m
        TRY(launch_zero32(S.U, mm * 2, s));
        {
            GemmArgs ga = gemm_args(S.Linv, Mp, S.LSp, Mp, S.U, Mp, Mp, Mp, Mp, TRI_LOWER_A | TRI_LOWER_B, 1.0);
            ga.lower_out = 1;
            ga.Kreal = D.M;      // the contraction's padded tail multiplies zeros (small problems)
            TRY(launch_gemm_auto(ga, false, F.ws, F.ws_elems, s));
        }
        TRY(launch_transpose(S.U, Mp, S.UT, Mp, Mp, Mp, s));
        TRY(launch_gemv_rows(S.Linv, Mp, S.mp, S.a, Mp, Mp, 1.0, 0, s));
        TRY(launch_kl(S.L, S.LSp, S.U, S.a, D.M, Mp, kl, F.qpart, s));
    }
    if (!do_panel) return MOBOCMF_OK;

    // K_mn, k_nn
    g.xdiv = desc->xdiv; g.x = x; g.f = f; g.nbase = D.nbase;
    g.K = S.K; g.ldk = Np; g.Np = Np; g.knn = S.knn; g.is_kmm = 0;
    TRY(launch_gram_fwd(g, s));
    // A = L^-1 K_mn  (+ q, mean partials);  C = U^T A (+ r partials)
    {
        GemmArgs ga = gemm_args(S.Linv, Mp, S.K, Np, S.A, Np, Mp, Np, Mp, TRI_LOWER_A, 1.0);
        ga.epi = EPI_COLSTATS; ga.colsq_part = F.qpart; ga.coldot_part = F.mupart; ga.avec = S.a;
        ga.Kreal = D.M;      // rows >= M of K_mn (and of A, C, dA below) are zero padding
        TRY(launch_gemm(ga, false, 1, s));
        GemmArgs gc = gemm_args(S.UT, Mp, S.A, Np, S.C, Np, Mp, Np, Mp, TRI_UPPER_A, 1.0);
        gc.epi = EPI_COLSTATS; gc.colsq_part = F.rpart; gc.coldot_part = nullptr; gc.avec = S.a;
        gc.Kreal = D.M;
        gc.stream_out = (desc->branch == 0 && Np * Mp * 8 >= ((int64_t)64 << 20)) ? 1 : 0;   // C is next read in backward
        TRY(launch_gemm(gc, false, 1, s));
    }
    TRY(launch_moments_finish(F.qpart, F.mupart, F.rpart, D.nrb, Np, D.N, S.knn, desc->branch, desc->min_var, S.q, S.r,
                              S.varraw, mean, var, s));
    return MOBOCMF_OK;
}

int mobocmf_layer_backward(const mobocmf_layer_desc* desc, const double* x, const double* f, const double* Zx,
                           const double* zf, const double* hyp, const double* m, const double* L_S,
                           const double* g_mean, const double* g_var, const double* g_kl, double* g_f, double* g_zf,
                           double* g_hyp, double* g_m, double* g_LS, double* g_x, void* saved, size_t saved_bytes,
                           void* scratch, size_t scratch_bytes, mobocmf_stream_t stream) {
    if (!valid_desc(desc) || !Zx || !hyp || !g_hyp || !saved || !scratch) return MOBOCMF_BAD_ARG;
    const bool inputs_only = desc->phase == MOBOCMF_PHASE_PANEL_INPUTS;   // parameters are constants: no H / Hc / da
    const bool do_panel = desc->phase == MOBOCMF_PHASE_ALL || desc->phase == MOBOCMF_PHASE_PANEL || inputs_only;
    const bool do_chain = desc->phase == MOBOCMF_PHASE_ALL || desc->phase == MOBOCMF_PHASE_CHAIN ||
                          desc->phase == MOBOCMF_PHASE_CHAIN_ONLY;
    const int acc = desc->phase == MOBOCMF_PHASE_ALL ? 1 : 0;   // split: the chain half reports its own g_hyp / g_zf
    if (do_panel && (!x || !g_mean || !g_var || (desc->want_dx && !g_x))) return MOBOCMF_BAD_ARG;
    if (do_chain && (!g_kl || !g_m || !g_LS)) return MOBOCMF_BAD_ARG;
    if (desc->kind == 1 && (!zf || !g_zf || (do_panel && (!f || !g_f)))) return MOBOCMF_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    Dims D = dims_of(desc);
    Bump bs(saved, saved_bytes), bb(scratch, scratch_bytes);
    Saved S;
    ScratchB B;
    if (!carve_saved(bs, D, S) || !carve_scratch_bwd(bb, D, desc, B)) return MOBOCMF_WORKSPACE_TOO_SMALL;
    const int Mp = D.Mp;
    const int64_t Np = D.Np, mm = (int64_t)Mp * Mp;
    double *H = B.W[0], *Hc = B.W[1], *G1 = B.W[2], *G2 = B.W[3], *X = B.W[4], *dU = B.W[5], *Y = B.W[6], *T1 = B.W[7],
           *T2 = B.W[8], *LT = B.W[9];
    double *dL = G1, *T4 = G2, *Gm = X;   // reused once their first content is dead
    int32_t* nclamped = (int32_t*)B.flag;
    const int64_t slab_elems = (int64_t)(D.splitk > 16 ? D.splitk : 16) * mm;

    if (desc->branch != 0) Hc = H;
    GramArgs g = {};
    g.kind = desc->kind; g.d = desc->d; g.zdiv = 1; g.Zx = Zx; g.zf = zf; g.M = D.M; g.hyp = hyp; g.Mp = Mp;
    if (do_panel) {
        TRY(launch_moments_bwd_prep(g_mean, g_var, S.knn, S.q, S.varraw, desc->branch, desc->min_var, D.N, Np, B.gmu, B.gv,
                                    B.gv2, B.cgv, nclamped, s));
        // dA = 2 U (C diag(gv)) + a gmu^T - 2 A diag(cgv)
        {
            GemmArgs ga = gemm_args(S.U, Mp, S.C, Np, B.dA, Np, Mp, Np, Mp, TRI_LOWER_A, 2.0);
            ga.bscale = B.gv; ga.epi = EPI_DA; ga.avec = S.a; ga.gmu = B.gmu; ga.cgv = B.cgv; ga.Aaux = S.A;
            ga.Kreal = D.M;
            ga.rowdot_part = inputs_only ? nullptr : B.dapart;      // da = A gmu rides in the epilogue (it reads A anyway)
            TRY(launch_gemm(ga, false, 1, s));
            if (!inputs_only) TRY(launch_sum_partials(B.dapart, gemm_rowdot_parts(ga), Mp, B.da, Mp, 1.0, 0, s));
        }
        // H = A diag(gv) A^T  (weighted syrk, split-K over N').  Both M x M contractions of the backward reduce to it:
        //   dU = 2 tril(A diag(gv) C^T) = 2 tril(H U),   dA A^T = 2 U U^T H + a da^T - 2 Hc,  Hc = A diag(cgv) A^T.
        // Hc differs from H only when clamp(k_nn - q, 0) is active in some column: its syrk is skipped on the device
        // (skip_if_zero) when no column is clamped.
        if (!inputs_only) {
            GemmArgs ga = gemm_args(S.A, Np, S.A, Np, B.slabs, Mp, Mp, Mp, Np, TRI_NONE, 1.0);
            ga.bscale = B.gv; ga.lower_out = 1; ga.sym_out = 1; ga.slab_stride = mm;
            const int nsl = gemm_nt_slabs(ga, D.splitk);      // 1: a small problem goes through whole, no k-slicing
            TRY(launch_gemm(ga, true, nsl, s));
            TRY(launch_reduce_slabs_sym(B.slabs, mm, nsl, H, Mp, nullptr, nullptr, s));
            if (desc->branch == 0) {
                ga.bscale = B.cgv; ga.skip_if_zero = nclamped;
                TRY(launch_gemm(ga, true, nsl, s));
                TRY(launch_reduce_slabs_sym(B.slabs, mm, nsl, Hc, Mp, nclamped, H, s));
            }
        }
        // dK = L^-T dA
        {
            GemmArgs ga = gemm_args(S.LinvT, Mp, B.dA, Np, B.dK, Np, Mp, Np, Mp, TRI_UPPER_A, 1.0);
            ga.Kreal = D.M;
            TRY(launch_gemm(ga, false, 1, s));
        }
        // Gram backward of K_mn and k_nn
        g.xdiv = desc->xdiv; g.x = x; g.f = f; g.nbase = D.nbase;
        g.ldk = Np; g.Np = Np; g.G = B.dK; g.gknn = B.cgv;
        g.hyp_part = B.hyp_part; g.df_part = B.df_part; g.dzf_part = B.dzf_part; g.dx_part = B.dx_part;
        TRY(launch_gram_bwd(g, desc->want_dx != 0, s));
        {
            SumTask tk[4];
            int nt = 0;
            tk[nt++] = {B.hyp_part, (int64_t)D.ggrid_mn.x * D.ggrid_mn.y, D.H, nullptr, 0, 0, g_hyp, D.H, 0};
            if (desc->kind == 1) {
                tk[nt++] = {B.df_part, D.ggrid_mn.y, Np, nullptr, 0, 0, g_f, D.N, 0};
                tk[nt++] = {B.dzf_part, D.ggrid_mn.x, Mp, nullptr, 0, 0, g_zf, D.M, 0};
            }
            if (desc->want_dx)
                tk[nt++] = {B.dx_part, D.ggrid_mn.y, D.nbase * desc->d, nullptr, 0, 0, g_x, D.nbase * desc->d, 0};
            TRY(launch_sum_partials_multi(tk, nt, s));
        }
    }
    if (!do_chain) return MOBOCMF_OK;
    if (desc->phase == MOBOCMF_PHASE_CHAIN_ONLY) {      // no upstream mean/var gradient: H = Hc = 0, da = 0
        TRY(launch_zero32(H, mm * 2, s));
        if (Hc != H) TRY(launch_zero32(Hc, mm * 2, s));
        TRY(launch_zero32(B.da, (int64_t)Mp * 2, s));
    }

    // ---- M x M chain:  dL = -tril(L^-T [dA A^T + dU_tot U^T + da_tot a^T]) + gkl diag(1/L_ii)
    {
        GemmArgs g1 = gemm_args(S.UT, Mp, H, Mp, G1, Mp, Mp, Mp, Mp, TRI_UPPER_A, 1.0);          // G1 = U^T H
        g1.Kreal = D.M;      // the contraction's padded tail multiplies zeros (small problems)
        TRY(launch_gemm_auto(g1, false, B.slabs, slab_elems, s));
        GemmArgs g2 = gemm_args(S.U, Mp, G1, Mp, G2, Mp, Mp, Mp, Mp, TRI_LOWER_A, 1.0);          // G2 = U U^T H
        g2.Kreal = D.M;      // the contraction's padded tail multiplies zeros (small problems)
        TRY(launch_gemm_auto(g2, false, B.slabs, slab_elems, s));
        GemmArgs g3 = gemm_args(H, Mp, S.U, Mp, X, Mp, Mp, Mp, Mp, TRI_LOWER_B, 1.0);            // X = H U (lower tiles)
        g3.lower_out = 1;
        g3.Kreal = D.M;      // the contraction's padded tail multiplies zeros (small problems)
        TRY(launch_gemm_auto(g3, false, B.slabs, slab_elems, s));
        TRY(launch_dutot(X, S.U, B.da, S.a, g_kl, Mp, dU, B.da_tot, s));
        TRY(launch_y_combine(G2, Hc, S.a, B.da, B.da_tot, Mp, Y, s));
        GemmArgs g4 = gemm_args(dU, Mp, S.U, Mp, Y, Mp, Mp, Mp, Mp, TRI_LOWER_A | TRI_UPPER_B, 1.0);   // Y += dU_tot U^T
        g4.accumulate = 1;
        g4.Kreal = D.M;      // the contraction's padded tail multiplies zeros (small problems)
        TRY(launch_gemm_auto(g4, true, B.slabs, slab_elems, s));
    }
    // g_m = L^-T da_tot
    TRY(launch_gemv_rows(S.LinvT, Mp, B.da_tot, g_m, D.M, Mp, 1.0, 0, s));
    // g_LS = tril(L^-T dU_tot) - gkl diag(1/LS_ii)
    {
        GemmArgs ga = gemm_args(S.LinvT, Mp, dU, Mp, T1, Mp, Mp, Mp, Mp, TRI_UPPER_A | TRI_LOWER_B, 1.0);
        ga.lower_out = 1;
        ga.Kreal = D.M;      // the contraction's padded tail multiplies zeros (small problems)
        TRY(launch_gemm_auto(ga, false, B.slabs, slab_elems, s));
        TRY(launch_gls_out(T1, S.LSp, g_kl, D.M, Mp, g_LS, s));
    }
    // dL
    {
        GemmArgs ga = gemm_args(S.LinvT, Mp, Y, Mp, T2, Mp, Mp, Mp, Mp, TRI_UPPER_A, 1.0);
        ga.lower_out = 1;
        ga.Kreal = D.M;      // the contraction's padded tail multiplies zeros (small problems)
        TRY(launch_gemm_auto(ga, false, B.slabs, slab_elems, s));
        TRY(launch_dl_from_t2(T2, S.L, g_kl, D.M, Mp, dL, s));
    }
    // Cholesky backward: dKmm = sym(L^-T Phi(L^T dL) L^-1)
    {
        TRY(launch_transpose(S.L, Mp, LT, Mp, Mp, Mp, s));
        GemmArgs ga = gemm_args(LT, Mp, dL, Mp, T2, Mp, Mp, Mp, Mp, TRI_UPPER_A | TRI_LOWER_B, 1.0);
        ga.Kreal = D.M;      // the contraction's padded tail multiplies zeros (small problems)
        TRY(launch_gemm_auto(ga, false, B.slabs, slab_elems, s));
        TRY(launch_phi(T2, Mp, T1, s));
        GemmArgs gb = gemm_args(S.LinvT, Mp, T1, Mp, T4, Mp, Mp, Mp, Mp, TRI_UPPER_A | TRI_LOWER_B, 1.0);
        gb.Kreal = D.M;      // the contraction's padded tail multiplies zeros (small problems)
        TRY(launch_gemm_auto(gb, false, B.slabs, slab_elems, s));
        GemmArgs gc = gemm_args(T4, Mp, S.Linv, Mp, T2, Mp, Mp, Mp, Mp, TRI_LOWER_B, 1.0);
        gc.Kreal = D.M;      // the contraction's padded tail multiplies zeros (small problems)
        TRY(launch_gemm_auto(gc, false, B.slabs, slab_elems, s));
        TRY(launch_symmetrize(T2, Mp, Gm, s));
    }
    // Gram backward of K_mm (both arguments are Z~)
    g.xdiv = 1; g.x = Zx; g.f = zf; g.nbase = D.M; g.ldk = Mp; g.Np = Mp; g.G = Gm; g.gknn = nullptr;
    g.hyp_part = B.hyp_part2; g.df_part = B.df_part2; g.dzf_part = B.dzf_part2; g.dx_part = nullptr;
    TRY(launch_gram_bwd(g, false, s));
    {
        SumTask tk[2];
        int nt = 0;
        tk[nt++] = {B.hyp_part2, (int64_t)D.ggrid_mm.x * D.ggrid_mm.y, D.H, nullptr, 0, 0, g_hyp, D.H, acc};
        if (desc->kind == 1)      // both arguments of K_mm are Z~: the row-side and the column-side partials land in g_zf
            tk[nt++] = {B.df_part2, D.ggrid_mm.y, Mp, B.dzf_part2, D.ggrid_mm.x, Mp, g_zf, D.M, acc};
        TRY(launch_sum_partials_multi(tk, nt, s));
    }
    return MOBOCMF_OK;
}

int mobocmf_predictive_covariance(const mobocmf_layer_desc* desc, const double* x, const double* f, const double* hyp,
                                  double* cov, int64_t ldcov, void* saved, size_t saved_bytes, void* scratch,
                                  size_t scratch_bytes, mobocmf_stream_t stream) {
    if (!valid_desc(desc) || !x || !hyp || !cov || !saved || !scratch || ldcov < desc->Np) return MOBOCMF_BAD_ARG;
    if (desc->kind == 1 && !f) return MOBOCMF_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    Dims D = dims_of(desc);
    if (D.Np > 16384) return MOBOCMF_BAD_ARG;
    Bump bs(saved, saved_bytes), bc(scratch, scratch_bytes);
    Saved S;
    if (!carve_saved(bs, D, S)) return MOBOCMF_WORKSPACE_TOO_SMALL;
    const int Mp = D.Mp;
    const int64_t Np = D.Np;
    double* AT = bc.take(Np * Mp);
    double* CT = bc.take(Np * Mp);
    double* Cv = bc.take(Np * Np);
    if (!bc.ok) return MOBOCMF_WORKSPACE_TOO_SMALL;
    // K_nn: Gram with the data rows on both sides
    GramArgs g = {};
    g.kind = desc->kind; g.d = desc->d; g.xdiv = desc->xdiv; g.zdiv = desc->xdiv;
    g.x = x; g.f = f; g.nbase = D.nbase; g.Zx = x; g.zf = f; g.M = (int)D.N; g.hyp = hyp;
    g.K = Cv; g.ldk = Np; g.Mp = (int)Np; g.Np = Np; g.knn = nullptr; g.jitter = 0.0; g.is_kmm = 0;
    TRY(launch_gram_fwd(g, s));
    TRY(launch_transpose(S.A, Np, AT, Mp, Mp, Np, s));
    TRY(launch_transpose(S.C, Np, CT, Mp, Mp, Np, s));
    GemmArgs ga = gemm_args(AT, Mp, S.A, Np, Cv, Np, (int)Np, Np, Mp, TRI_NONE, -1.0);
    ga.accumulate = 1;
    TRY(launch_gemm(ga, false, 1, s));
    GemmArgs gc = gemm_args(CT, Mp, S.C, Np, Cv, Np, (int)Np, Np, Mp, TRI_NONE, 1.0);
    gc.accumulate = 1;
    TRY(launch_gemm(gc, false, 1, s));
    TRY(launch_copy_block(Cv, Np, cov, ldcov, D.N, D.N, s));
    return MOBOCMF_OK;
}

int mobocmf_gemm_f64(int32_t tri, int32_t trans_b, int32_t Mr, int64_t Nc, int64_t Kd, const double* A, int64_t lda,
                     const double* B, int64_t ldb, double* C, int64_t ldc, double alpha, int32_t accumulate,
                     mobocmf_stream_t stream) {
    if (!A || !B || !C || Mr <= 0 || Nc <= 0 || Kd <= 0 || (lda & 1) || (ldb & 1)) return MOBOCMF_BAD_ARG;
    GemmArgs g = gemm_args(A, lda, B, ldb, C, ldc, Mr, Nc, Kd, tri, alpha);
    g.accumulate = accumulate;
    return launch_gemm(g, trans_b != 0, 1, (hipStream_t)stream);
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
                              mobocmf_stream_t stream) {
    if (!A || !B || !C || Mr <= 0 || Nc <= 0 || Kd <= 0 || (lda & 1) || (ldb & 1)) return MOBOCMF_BAD_ARG;
    if (epi < EPI_STORE || epi > EPI_DA) return MOBOCMF_BAD_ARG;
    if (epi == EPI_COLSTATS && (!colsq_part || !avec)) return MOBOCMF_BAD_ARG;
    if (epi == EPI_DA && (!avec || !gmu || !cgv || !Aaux)) return MOBOCMF_BAD_ARG;
    GemmArgs g = gemm_args(A, lda, B, ldb, C, ldc, Mr, Nc, Kd, tri, alpha);
    g.epi = epi; g.stream_out = stream_out;
    g.colsq_part = colsq_part; g.coldot_part = coldot_part; g.avec = avec;
    g.bscale = epi == EPI_DA ? bscale : nullptr; g.gmu = gmu; g.cgv = cgv; g.Aaux = Aaux; g.rowdot_part = rowdot_part;
    return launch_gemm(g, false, 1, (hipStream_t)stream);
}

int mobocmf_check_info(const int32_t* info, int32_t* pivot, mobocmf_stream_t stream) {
    int32_t h = 0;
    if (hipMemcpyAsync(&h, info, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
        return MOBOCMF_HIP_ERROR;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return MOBOCMF_HIP_ERROR;
    if (pivot) *pivot = h;
    return h == 0 ? MOBOCMF_OK : MOBOCMF_NOT_PD;
}

}  // extern "C"
