// The whole ELBO step of a SMALL surrogate in one launch (include/mobocmf_hip.h: mobocmf_tiny_elbo_step), gfx950.
//
// At the reference's own sizes (M = N = tens of points, examples/example_acquisition_mfdgp_forrester/...py:51-62) a step
// through the layer entry points is ~50 dependent launches of ~4.7 us each whatever they compute.  Here ONE
// workgroup of 256 (wide panels: 512) threads runs zero_grad + MFDGP.forward (mfdgp.py:174-196) + VariationalELBOMF (variational_elbo_mf.py:
// 24-51) + backward + Adam (blackbox_mfdgp_fitter.py:161-171) as ~45 phases separated by workgroup barriers (~0.1 us each
// instead of a kernel boundary); several surrogates are several workgroups of the same launch.  The algebra is DESIGN.md 1
// line for line -- explicit L^-1, A = L^-1 K, C = U^T A, the single weighted syrk H, Murray's Cholesky backward -- as
// element-parallel products: a thread owns an output element and walks its (triangular) contraction range.  The M x M chain
// state of every layer lives in LDS; the M x N' panels join it there when they fit, else they live in `work` (a workgroup
// reads its own global writes after a barrier: one CU, one L1).  Only the Cholesky + triangular inverse is serial: one wavefront per layer, the matrix row-per-lane in
// registers, pivots and multipliers broadcast by v_readlane (no LDS round trip, no barrier), the layers side by side.
// Modes (do_update): 0 gradients only, 1 the step, 2 forward only (acquisition moments; the first launch of a three-launch
// conditioned iteration), 3 input gradients (acquisition search), 4 the conditioned iteration in one launch (the models'
// workgroups meet at an arrival counter after their forward and form the theta / omega factor gradients themselves).
#include <atomic>

#include "common.h"
#include "small_step_common.h"

namespace {

struct Geom {
    int L, M, d, S;
    int ncol[TLM], H[TLM];
    int64_t flat_off[TLM], flat_noise, flat_len;
    int64_t work_off[TLM], scratch_off, pool_base, pool_len, cpl_off, work_len;
    int ncmax, srows;
};
__host__ __device__ inline void geom_of(const mobocmf_tiny_model& md, Geom& g) {
    g.L = md.L; g.M = md.M; g.d = md.d; g.S = md.S;
    int64_t fo = 0;
    g.ncmax = 0;
    for (int l = 0; l < TLM; ++l) {
        g.ncol[l] = l < md.L ? md.rows[l] * (l ? md.S : 1) : 0;
        g.H[l] = l == 0 ? 1 + md.d : 5 + 2 * md.d;
        g.flat_off[l] = fo;
        if (l < md.L) fo += g.H[l] + md.M + (int64_t)md.M * md.M;
        if (g.ncol[l] > g.ncmax) g.ncmax = g.ncol[l];
    }
    g.flat_noise = fo;
    g.flat_len = fo + md.L;
    // the panel pool (every layer's A, C and per-column vectors, three scratch panels): in LDS when it fits the launch,
    // otherwise in `work` behind the flat gradient; offsets are relative to the pool
    g.pool_base = (g.flat_len + 1) & ~(int64_t)1;
    int64_t wo = 0;
    for (int l = 0; l < TLM; ++l) {
        g.work_off[l] = wo;
        wo += (int64_t)g.ncol[l] * (2 * md.M + NVEC);
    }
    g.scratch_off = wo;
    g.srows = md.M > DBT ? md.M : DBT;      // rows of a scratch panel (mode 3 keeps DBT values per column in one)
    g.pool_len = wo + 3 * (int64_t)g.srows * g.ncmax;
    g.cpl_off = g.pool_base + g.pool_len;      // mode 4: partial sums of the factor terms, 256 x (1 + 2 x 8) doubles
    g.work_len = g.cpl_off + CPL_DOUBLES;
#ifdef TINY_STAMPS
    g.work_len += 128;      // phase stamps (tools/tiny_stamps.py): the last 128 doubles of `work`
#endif
}

// Cholesky of the MM x MM matrix in Lm (lower part used; identity beyond the real size) and the inverse of the factor, by ONE
// wavefront: lane i holds row i in registers.  Right-looking: column j's multipliers l_ij = a_ij / sqrt(a_jj) are formed in
// every lane at once, lane k's multiplier is broadcast (v_readlane, compile-time lane) for the update of column k.  The
// inverse is a forward substitution with lane c owning column c of L^-1 (L_ik broadcast the same way).  No LDS traffic
// inside, no barrier; Lm / Li receive L and L^-1 (zeros above the diagonal) at the end.
template <int MM>
__device__ __forceinline__ void chol_inv_wave(double* Lm, double* Li, int LD, int lane, int32_t* info) {
    double row[MM];
    const int ln = lane < MM ? lane : 0;
#pragma unroll
    for (int k = 0; k < MM; ++k) row[k] = Lm[ln * LD + k];
    int fail = 0;
    // 1 / L_jj (wave-uniform) is parked in row 0 of Li until the inverse needs it (Li is written at the very end): MM values in
    // registers per lane next to row[] and x[] pushed the 32-wide instantiation over its register budget and into scratch
#pragma unroll
    for (int j = 0; j < MM; ++j) {
        const double djj = rdlane(row[j], j);
        if (!(djj > 0.0) && !fail) fail = j + 1;
        // 1 / sqrt(d): v_rsq_f64 + two Newton steps (the IEEE sqrt + divide pair is ~200 dependent cycles per column)
        double r = __builtin_amdgcn_rsq(djj);
        r = r * (1.5 - (0.5 * djj) * r * r);
        r = __builtin_fma(0.5 * r, __builtin_fma(-(djj * r), r, 1.0), r);      // residual form: ~1 ulp (chol.hip rsqrt_nr)
        if (lane == 0) Li[j] = r;
        const double lij = row[j] * r;      // lane j: d / sqrt(d) = L_jj
        row[j] = lij;
#pragma unroll
        for (int k = j + 1; k < MM; ++k) row[k] -= lij * rdlane(lij, k);
    }
    // L goes to LDS now: the inverse reads L_ik back as wave-uniform broadcasts, so row[] is dead while x[] is alive (both in
    // registers were 2 x 2 MM VGPRs: at MM = 32 all 256 of them)
    if (lane < MM) {
#pragma unroll
        for (int k = 0; k < MM; ++k) Lm[lane * LD + k] = k <= lane ? row[k] : 0.0;
    }
    double x[MM];
#pragma unroll
    for (int i = 0; i < MM; ++i) {
        double s = i == lane ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) s -= Lm[i * LD + k] * x[k];
        x[i] = s * Li[i];
    }
    if (lane < MM) {
#pragma unroll
        for (int k = 0; k < MM; ++k) Li[k * LD + lane] = x[k];
    }
    if (lane == 0) info[0] = fail;
}

template <int MR, int TT>
__global__ __launch_bounds__(TT) void tiny_step_kernel(const mobocmf_tiny_model* models, double lr, double b1, double b2,
                                                       double aeps, int do_update, int pool_in_lds) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const mobocmf_tiny_model& md = models[blockIdx.x];
    constexpr int LD = MR + 1, MS = MR * LD;
    constexpr int NW = TT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the geometry -- per-layer sizes and workspace offsets, indexed by the run-time layer -- lives in LDS, formed by one thread: as
    // a private struct its arrays sat in scratch memory (624 B per lane) and every g.ncol[l] / g.work_off[l] at the head of a
    // phase was a scratch round trip
    __shared__ Geom g_lds;
    if (tid == 0) geom_of(md, g_lds);
    __syncthreads();
    const Geom& g = g_lds;
    const int L = g.L, M = g.M, d = g.d, S = g.S;
    double* Lm = lds;                       // [TLM][MS]  L
    double* Li = Lm + TLM * MS;             // [TLM][MS]  L^-1
    double* Um = Li + TLM * MS;             // [TLM][MS]  U = L^-1 L_S
    double* W0 = Um + TLM * MS;             // three M x M temporaries of the backward
    double* W1 = W0 + MS;
    double* W2 = W1 + MS;
    double* zt = W2 + MS;                   // [TLM][MR][ZW]  Z~_l
    double* hy = zt + TLM * MR * ZW;        // [TLM][HS] constrained hyper-parameters
    double* il = hy + TLM * HS;             // [TLM][2 DBT] inverse lengthscales
    double* av = il + TLM * 2 * DBT;        // [TLM][MR]  a = L^-1 m
    double* gmv = av + TLM * MR;            // [TLM][MR]  d loss / d m
    double* ghy = gmv + TLM * MR;           // [TLM][HS]  d loss / d (constrained hyper-parameters)
    double* dav = ghy + TLM * HS;           // [MR] da
    double* dat = dav + MR;                 // [MR] da_tot
    double* mst = dat + MR;                 // [TLM][MR] m of every layer (staged)
    double* red = mst + TLM * MR;           // [NWMAX][HS + 1] wavefront partials
    double* sc = red + NWMAX * (HS + 1);    // [24]: tau[l] (0..2), g_noise[l] (4..6), Adam's bias terms (12, 13), bsum scratch (16..23)
    double** seg_ptr = (double**)(sc + 24); // [NSEG] parameter tensors in flat-vector order (trainable ones; else null)
    int* seg_end = (int*)(seg_ptr + NSEG);  // [NSEG] end offset of each tensor in the flat vector
    double* gflat = md.work;
    // the panels: LDS behind the chain state when the launch reserved room for them (a dependent global round trip per
    // contraction step is what a phase costs otherwise), else the caller's workspace; generic pointers either way
    double* work = pool_in_lds ? sc + 24 + NSEG + NSEG / 2 : md.work + g.pool_base;
    double* Kb = work + g.scratch_off;      // M x ncmax scratch panels
    double* S1 = Kb + (int64_t)g.srows * g.ncmax;
    double* S2 = S1 + (int64_t)g.srows * g.ncmax;
#ifdef TINY_STAMPS
    double* stamps = md.work + g.work_len - 128;
    int n_stamp = 0;
#define STAMP() do { if (tid == 0) stamps[n_stamp] = (double)wall_clock64(); ++n_stamp; } while (0)
#else
#define STAMP() do { } while (0)
#endif
    STAMP();
    const double gkl = md.kl_scale;         // d(-ELBO) / d KL_l
    const double ge = -1.0;                 // d(-ELBO) / d ELBO

    // ---- P0: constrained hyper-parameters, Z~, tau; gradient accumulators cleared
    for (int e = tid; e < TLM * HS; e += TT) {
        const int l = e / HS, t = e % HS;
        double v = 0.0;
        if (l < L && t < g.H[l]) {
            int s = 0, off = 0;
            while (t >= off + seg_len(l, s, d)) { off += seg_len(l, s, d); ++s; }
            const double x = md.raw[l][s][t - off];
            v = x > 20.0 ? x : log1p(exp(x));
        }
        hy[e] = v;
        ghy[e] = 0.0;
    }
    for (int e = tid; e < TLM * MR; e += TT) gmv[e] = 0.0;
    for (int e = tid; e < TLM * MR * ZW; e += TT) {
        const int l = e / (MR * ZW), m = (e / ZW) % MR, k = e % ZW;
        double v = 0.0;
        if (l < L && m < M) {
            if (k < d) v = md.Zx[m * d + k];
            else if (k == DBT && l > 0) v = md.m[l - 1][m];
        }
        zt[e] = v;
    }
    // L_S and m of every layer staged in LDS (W_l is free until the backward): P3 walks them k by k
    for (int e = tid; e < L * M * M; e += TT) {
        const int l = e / (M * M), i = (e / M) % M, j = e % M;
        W0[l * MS + i * LD + j] = j <= i ? md.L_S[l][i * M + j] : 0.0;
    }
    for (int e = tid; e < L * M; e += TT) mst[(e / M) * MR + e % M] = md.m[e / M][e % M];
    if (do_update == 3)
        for (int e = tid; e < md.N * d; e += TT) md.grad[e] = 0.0;
    if ((do_update == 2 || do_update == 4) && md.xrng && md.rand_rows > 0) {
        // the x~ of this iteration (:276): every model of the launch draws the SAME points from the shared stream
        const uint64_t seed = (uint64_t)md.xrng[0], call = (uint64_t)md.xrng[1];
        double* xw = const_cast<double*>(md.x) + (int64_t)md.rand_row0 * d;
        for (int e = tid; e < md.rand_rows * d; e += TT) xw[e] = philox_uniform(seed, call, (uint64_t)e);
    }
    if (tid >= 64 && tid < 64 + NSEG) {      // the parameter tensors in flat-vector order: thread 64 + k fills entry k
        const int k = tid - 64;
        double* ptr = nullptr;
        int end = 0x7fffffff, j = 0;
        for (int l = 0; l < L; ++l) {
            const uint32_t tr = md.trainable[l];
            const int ns = l == 0 ? 2 : 7;
            int off = (int)g.flat_off[l];
            for (int s2 = 0; s2 < ns; ++s2, ++j) {
                off += seg_len(l, s2, d);
                if (j == k) { ptr = ((tr >> s2) & 1u) ? md.raw[l][s2] : nullptr; end = off; }
            }
            off += M;
            if (j++ == k) { ptr = ((tr >> 7) & 1u) ? md.m[l] : nullptr; end = off; }
            off += M * M;
            if (j++ == k) { ptr = ((tr >> 8) & 1u) ? md.L_S[l] : nullptr; end = off; }
        }
        for (int l = 0; l < L; ++l)
            if (j++ == k) { ptr = ((md.trainable[l] >> 9) & 1u) ? md.raw_noise[l] : nullptr; end = (int)g.flat_noise + l + 1; }
        seg_ptr[k] = ptr;
        seg_end[k] = end;
    }
    if (tid == TT - 1 && (do_update == 1 || do_update == 4)) {      // Adam's bias corrections (two pow calls): once, off the critical path
        const double step = (double)(md.steps_done[0] + 1);
        sc[12] = 1.0 - pow(b1, step);
        sc[13] = sqrt(1.0 - pow(b2, step));
    }
    if (tid < L) {
        const double lo = md.noise_lo[tid], hi = md.noise_hi[tid], r = md.raw_noise[tid][0];
        sc[tid] = hi > lo ? lo + (hi - lo) / (1.0 + exp(-r)) : r;
    }
    __syncthreads();
    STAMP();
    for (int e = tid; e < TLM * 2 * DBT; e += TT) {
        const int l = e / (2 * DBT), k = e % (2 * DBT), kk = k % DBT;
        double v = 0.0;
        if (l < L && kk < d) {
            if (l == 0) v = k < DBT ? 1.0 / hy[l * HS + 1 + kk] : 0.0;
            else v = 1.0 / hy[l * HS + 5 + (k < DBT ? 0 : d) + kk];
        }
        il[e] = v;
    }
    __syncthreads();
    STAMP();

    // ---- P1: K_mm + jitter I of every layer (lower part; identity beyond M)
    for (int e = tid; e < L * MR * MR; e += TT) {
        const int l = e / (MR * MR), i = (e / MR) % MR, j = e % MR;
        double v = 0.0;
        if (i < M && j <= i) {
            KV o;
            kern_eval(l > 0, d, zt + (l * MR + i) * ZW, zt[(l * MR + i) * ZW + DBT], zt + (l * MR + j) * ZW, hy + l * HS,
                      il + l * 2 * DBT, o);
            v = o.k + (i == j ? md.jitter : 0.0);
        } else if (i == j) {
            v = 1.0;
        }
        Lm[l * MS + i * LD + j] = v;
    }
    __syncthreads();
    STAMP();
    // ---- P2: Cholesky + inverse, one wavefront per layer
    if (wave < L) chol_inv_wave<MR>(Lm + wave * MS, Li + wave * MS, LD, lane, md.info + wave);
    __syncthreads();
    STAMP();
    // ---- P3: U = L^-1 L_S, a = L^-1 m, KL = 1/2 [2 sum log L_ii - sum log L_S,ii^2 + |U|^2 + |a|^2 - M]
    double klacc = 0.0;
    for (int e = tid; e < L * M * M; e += TT) {
        const int l = e / (M * M), i = (e / M) % M, j = e % M;
        const double* li = Li + l * MS + i * LD;
        const double* ls = W0 + l * MS;
        double u = 0.0;
        if (j <= i) {
#pragma unroll 4
            for (int k = j; k <= i; ++k) u += li[k] * ls[k * LD + j];
            klacc += 0.5 * u * u;
        }
        Um[l * MS + i * LD + j] = u;
        if (j == 0) {
            const double* mv = mst + l * MR;
            double a = 0.0;
#pragma unroll 4
            for (int k = 0; k <= i; ++k) a += li[k] * mv[k];
            av[l * MR + i] = a;
            const double lsii = ls[i * LD + i];
            klacc += 0.5 * a * a + log(Lm[l * MS + i * LD + i] / fabs(lsii)) - 0.5;      // log L_ii - 1/2 log L_S,ii^2
        }
    }
    __syncthreads();
    STAMP();

    // ---- forward, layer by layer
    double dacc = 0.0;
    for (int l = 0; l < L; ++l) {
        const int kind = l > 0, div = l ? S : 1, nc = g.ncol[l];
        double* wl = work + g.work_off[l];
        double* A = wl;
        double* C = A + (int64_t)M * nc;
        double* vf = C + (int64_t)M * nc;      // f, eps, mean, var, knn, q, vraw, gmu, gv, cgv, gf
        double *vmean = vf + 2 * nc, *vvar = vf + 3 * nc, *vknn = vf + 4 * nc, *vq = vf + 5 * nc, *vraw = vf + 6 * nc;
        const double* hyl = hy + l * HS;
        const double* ill = il + l * 2 * DBT;
        // F1: K_mn
        for (int e = tid; e < M * nc; e += TT) {
            const int m = e / nc, c = e % nc, b = c / div;
            const double fn = kind ? vf[c] : 0.0;
            KV o;
            kern_eval(kind, d, md.x + (int64_t)b * d, fn, zt + (l * MR + m) * ZW, hyl, ill, o);
            Kb[(int64_t)m * nc + c] = o.k;
            if (m == 0) vknn[c] = kind ? hyl[0] * (hyl[2] * fn * fn + hyl[1]) + hyl[3] : hyl[0];
        }
        __syncthreads();
        STAMP();
        // F2: A = L^-1 K
        for (int e = tid; e < M * nc; e += TT) {
            const int i = e / nc, c = e % nc;
            const double* li = Li + l * MS + i * LD;
            double s = 0.0;
#pragma unroll 4
            for (int k = 0; k <= i; ++k) s += li[k] * Kb[(int64_t)k * nc + c];
            A[(int64_t)i * nc + c] = s;
        }
        __syncthreads();
        STAMP();
        // F3: C = U^T A
        for (int e = tid; e < M * nc; e += TT) {
            const int j = e / nc, c = e % nc;
            const double* u = Um + l * MS + j;
            double s = 0.0;
#pragma unroll 4
            for (int i = j; i < M; ++i) s += u[i * LD] * A[(int64_t)i * nc + c];
            C[(int64_t)j * nc + c] = s;
        }
        __syncthreads();
        STAMP();
        // F4: moments and the layer's data term
        {
            const double tau = sc[l], ltau = log(tau);
            for (int c = tid; c < nc; c += TT) {
                double q = 0.0, mu = 0.0, r = 0.0;
#pragma unroll 4
                for (int i = 0; i < M; ++i) {
                    const double a = A[(int64_t)i * nc + c], cc = C[(int64_t)i * nc + c];
                    q += a * a;
                    mu += av[l * MR + i] * a;
                    r += cc * cc;
                }
                double sres = vknn[c] - q;
                if (!md.branch && sres < 0.0) sres = 0.0;      // train branch: clamp(k_nn - q, 0); eval branch: none
                const double vr = sres + r, var = vr < MINV ? MINV : vr;
                vmean[c] = mu; vvar[c] = var; vq[c] = q; vraw[c] = vr;
                const int b = c / div;
                if (md.fid[b] == (double)l) {
                    const double dlt = md.y[b] - mu, w = md.row_weight ? md.row_weight[b] : 1.0;
                    dacc += w * (-0.5 * ((dlt * dlt + var) / tau + ltau + LOG2PI)) / div;
                }
                if (l == L - 1 && md.top_mean) { md.top_mean[c] = mu; md.top_var[c] = var; }
            }
        }
        __syncthreads();
        STAMP();
        // F5: the next layer's inputs f = mean + sqrt(var) eps, one thread per column of the NEXT layer (one draw each)
        if (l + 1 < L) {
            const int ncn = g.ncol[l + 1], fdiv = l == 0 ? S : 1;
            double* nf = work + g.work_off[l + 1] + 2 * (int64_t)M * ncn;
            const double* xeps = md.eps[l + 1];
            const int64_t* rng = md.rng[l + 1];
            const uint64_t seed = rng ? (uint64_t)rng[0] : 0, call = rng ? (uint64_t)rng[1] : 0;
            for (int cn = tid; cn < ncn; cn += TT) {
                const int c = cn / fdiv;
                const double ev = xeps ? xeps[cn] : philox_normal(seed, call, (uint64_t)cn);
                nf[cn] = vmean[c] + sqrt(vvar[c]) * ev;
                nf[ncn + cn] = ev;
            }
            __syncthreads();
            STAMP();
        }
    }
    {
        const double data = bsum<NW>(dacc, sc + 16);
        const double kl = bsum<NW>(klacc, sc + 16);
        if (tid == 0) {
            md.out[0] = data - gkl * kl;
            md.out[1] = gkl * kl;
            md.out[2] = -(data - gkl * kl);
        }
    }

    if (do_update == 2) return;
    if (do_update == 4) {
        // the conditioned iteration in one launch: every model's top-layer moments are published, the grid meets, every
        // workgroup forms the theta / omega factor gradients of its own model (its own moments it reads back from L1 / LDS
        // order; the others' with agent-scope loads)
        // The barrier: a monotonic arrival counter in the coupling record (never reset: launch k waits for k n_models
        // arrivals).  The n_models <= 64 workgroups of a launch are resident together on a 256-CU device whatever else runs
        // (a waiting workgroup holds one CU's slot; the missing ones get theirs as soon as ANY slot frees), so the wait ends;
        // should it not within ~0.2 s, the workgroup gives up, flags the model (info = -1, the coupling's sticky status word)
        // and leaves without touching its parameters instead of hanging the device.  This is an ORDINARY launch, not a
        // cooperative one (hipLaunchCooperativeKernel + grid.sync() does the same job with a 23 us dispatch gap per launch
        // and cannot be captured into a graph): co-residency is checked on the host against the device's occupancy for this
        // kernel and LDS size (mobocmf_tiny_elbo_step refuses mode 4 otherwise).
        // A launch whose shape does not match the record (another n_models than the record was made for: the arrival counter
        // would no longer be a multiple of the grid at launch start; T outside what coupling_seeds divides by) stops HERE, in
        // every workgroup alike, before anyone arrives: nothing hangs, nothing is updated, the sticky status word says why.
        const mobocmf_tiny_coupling& cpl = *md.coupling;
        if (cpl.n_models != (int)gridDim.x || cpl.T < 1 || cpl.T > 256 || cpl.P < 1) {
            if (tid == 0) {
                atomicOr(cpl.status, 2);
                md.info[0] = -2;
                md.out[2] = __builtin_nan("");
            }
            return;
        }
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            unsigned long long* cnt = (unsigned long long*)cpl.barrier;
            const unsigned long long n = gridDim.x;
            const unsigned long long old = atomicAdd(cnt, 1ull), target = (old / n + 1ull) * n;
            int spins = 0, gave_up = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1 << 21)) { gave_up = 1; break; }
            }
            __threadfence();
            sc[14] = gave_up ? 1.0 : 0.0;
            if (gave_up) {      // sticky (only ever OR'd; the next launch rewrites info / out): TinyConditionedStep.check() reads it
                atomicOr(cpl.status, 1);
                md.info[0] = -1;
                md.out[2] = __builtin_nan("");
            }
        }
        __syncthreads();
        // a workgroup that gave up has no right to its peers' moments: it leaves BEFORE the backward and the update -- its
        // parameters, optimiser state, step count and random streams stay as they were (its peers may still complete: they
        // read what this workgroup published before it arrived)
        if (sc[14] != 0.0) return;
        coupling_seeds<TT>(models, md, g.ncol[L - 1], md.work + g.cpl_off, sc + 16);
    }
    // ---- backward, top layer first
    for (int l = L - 1; l >= 0; --l) {
        const int kind = l > 0, div = l ? S : 1, nc = g.ncol[l], Hl = g.H[l];
        double* wl = work + g.work_off[l];
        double* A = wl;
        double* C = A + (int64_t)M * nc;
        double* vf = C + (int64_t)M * nc;
        double *vmean = vf + 2 * nc, *vvar = vf + 3 * nc, *vknn = vf + 4 * nc, *vq = vf + 5 * nc, *vraw = vf + 6 * nc;
        double *vgmu = vf + 7 * nc, *vgv = vf + 8 * nc, *vcgv = vf + 9 * nc, *vgf = vf + 10 * nc;
        const double* hyl = hy + l * HS;
        const double* ill = il + l * 2 * DBT;
        const double* Ul = Um + l * MS;
        const double* Lil = Li + l * MS;
        const double* Ll = Lm + l * MS;
        const double* al = av + l * MR;
        // B1: upstream gradients of the layer's moments: its own data term + what the next layer sent back through f
        double st = 0.0;
        {
            const double tau = sc[l], gd = ge / div;
            const int ncn = l + 1 < L ? g.ncol[l + 1] : 0, fdiv = l == 0 ? S : 1;
            const double* nvf = l + 1 < L ? work + g.work_off[l + 1] + 2 * (int64_t)M * ncn : nullptr;
            for (int c = tid; c < nc; c += TT) {
                const int b = c / div;
                double gm = 0.0, gvv = 0.0;
                if (md.fid[b] == (double)l) {
                    const double dlt = md.y[b] - vmean[c], w = md.row_weight ? md.row_weight[b] : 1.0;
                    gm = w * gd * dlt / tau;
                    gvv = -0.5 * w * gd / tau;
                    st += w * 0.5 * ((dlt * dlt + vvar[c]) / (tau * tau) - 1.0 / tau);
                }
                if (l == L - 1 && md.seed_gmean) {
                    gm += md.seed_scale * md.seed_gmean[c];
                    gvv += md.seed_scale * md.seed_gvar[c];
                }
                if (c * fdiv < ncn) {
                    double sm = 0.0, sv = 0.0;
                    for (int s = 0; s < fdiv; ++s) {
                        const double g2 = nvf[10 * ncn + c * fdiv + s];
                        sm += g2;
                        sv += g2 * nvf[ncn + c * fdiv + s];
                    }
                    gm += sm;
                    gvv += sv * 0.5 / sqrt(vvar[c]);
                }
                const double gvc = vraw[c] > MINV ? gvv : 0.0;
                vgmu[c] = gm;
                vgv[c] = gvc;
                vcgv[c] = (md.branch || vknn[c] - vq[c] > 0.0) ? gvc : 0.0;
            }
        }
        __syncthreads();
        STAMP();
        // B2: dA = 2 U (C diag gv) + a g_mean^T - 2 A diag(cgv)
        for (int e = tid; e < M * nc; e += TT) {
            const int i = e / nc, c = e % nc;
            const double* u = Ul + i * LD;
            double s = 0.0;
#pragma unroll 4
            for (int j = 0; j <= i; ++j) s += u[j] * C[(int64_t)j * nc + c];
            S1[(int64_t)i * nc + c] = 2.0 * s * vgv[c] + al[i] * vgmu[c] - 2.0 * A[(int64_t)i * nc + c] * vcgv[c];
        }
        __syncthreads();
        STAMP();
        // B3: dK = L^-T dA
        for (int e = tid; e < M * nc; e += TT) {
            const int m = e / nc, c = e % nc;
            double s = 0.0;
#pragma unroll 4
            for (int i = m; i < M; ++i) s += Lil[i * LD + m] * S1[(int64_t)i * nc + c];
            Kb[(int64_t)m * nc + c] = s;
        }
        __syncthreads();
        STAMP();
        if (do_update == 3) {
            // input gradients only (the parameters are constants: acquisition search): per column d loss / d f, handed to
            // the layer below, and d loss / d x, summed over the row's columns into grad[b][k]
            for (int c = tid; c < nc; c += TT) {
                const int b = c / div;
                const double fn = kind ? vf[c] : 0.0;
                double dxa[DBT], dfs = 0.0;
#pragma unroll
                for (int k = 0; k < DBT; ++k) dxa[k] = 0.0;
                for (int m = 0; m < M; ++m)
                    kern_back_in(kind, d, md.x + (int64_t)b * d, fn, zt + (l * MR + m) * ZW, hyl, ill, Kb[(int64_t)m * nc + c], dxa, dfs);
                if (kind) vgf[c] = dfs + vcgv[c] * hyl[0] * 2.0 * hyl[2] * fn;
#pragma unroll
                for (int k = 0; k < DBT; ++k) S1[(int64_t)c * DBT + k] = dxa[k];
            }
            __syncthreads();
            for (int e = tid; e < md.rows[l] * d; e += TT) {
                const int b = e / d, k = e % d;
                double sx = 0.0;
                for (int s2 = 0; s2 < div; ++s2) sx += S1[(int64_t)(b * div + s2) * DBT + k];
                md.grad[e] += sx;
            }
            __syncthreads();
            continue;
        }
        // B4: Gram backward of (dK, dk_nn = cgv), element by element; H = A diag(gv) A^T, Hc = A diag(cgv) A^T, da = A g_mean
        double hacc[HS];
#pragma unroll
        for (int t = 0; t < HS; ++t) hacc[t] = 0.0;
        for (int e = tid; e < M * nc; e += TT) {
            const int m = e / nc, c = e % nc, b = c / div;
            const double fn = kind ? vf[c] : 0.0;
            double dfa, dzf;
            kern_back(kind, d, md.x + (int64_t)b * d, fn, zt + (l * MR + m) * ZW, hyl, ill, Kb[(int64_t)m * nc + c], hacc, dfa, dzf);
            S1[(int64_t)m * nc + c] = dfa;
            S2[(int64_t)m * nc + c] = dzf;
            if (m == 0) {
                const double gk = vcgv[c];
                if (!kind) hacc[0] += gk;
                else {
                    hacc[0] += gk * (hyl[2] * fn * fn + hyl[1]);
                    hacc[1] += gk * hyl[0];
                    hacc[2] += gk * hyl[0] * fn * fn;
                    hacc[3] += gk;
                }
            }
        }
        for (int e = tid; e < M * M; e += TT) {
            const int i = e / M, j = e % M;
            if (j > i) continue;
            double h = 0.0, hc = 0.0;
#pragma unroll 4
            for (int c = 0; c < nc; ++c) {
                const double p = A[(int64_t)i * nc + c] * A[(int64_t)j * nc + c];
                h += p * vgv[c];
                hc += p * vcgv[c];
            }
            W0[i * LD + j] = W0[j * LD + i] = h;
            W1[i * LD + j] = W1[j * LD + i] = hc;
            if (j == 0) {
                double s = 0.0;
#pragma unroll 4
                for (int c = 0; c < nc; ++c) s += A[(int64_t)i * nc + c] * vgmu[c];
                dav[i] = s;
            }
        }
#pragma unroll
        for (int t = 0; t < HS; ++t) {
            if (slot_used(kind, d, t)) {      // uniform
                const double v = wsum63(hacc[t]);
                if (lane == 63) red[wave * (HS + 1) + t] = v;
            }
        }
        {
            const double v = wsum63(st);
            if (lane == 63) red[wave * (HS + 1) + HS] = v;
        }
        __syncthreads();
        STAMP();
        // B5: column / row sums of the Gram backward; the wavefront partials
        if (kind) {
            for (int c = tid; c < nc; c += TT) {
                double s = 0.0;
#pragma unroll 4
                for (int m = 0; m < M; ++m) s += S1[(int64_t)m * nc + c];
                vgf[c] = s + vcgv[c] * hyl[0] * 2.0 * hyl[2] * vf[c];
            }
            for (int m = tid; m < M; m += TT) {
                double s = 0.0;
#pragma unroll 4
                for (int c = 0; c < nc; ++c) s += S2[(int64_t)m * nc + c];
                gmv[(l - 1) * MR + m] += s;
            }
        }
        if (tid < Hl) {
            const int t = slot_of(kind, d, tid);
            ghy[l * HS + tid] += red_sum<NW>(red, HS + 1, t);
        }
        if (tid == TT - 1) {
            const double s = red_sum<NW>(red, HS + 1, HS);
            double chain = 1.0;
            if (md.noise_hi[l] > md.noise_lo[l]) {
                const double sg = 1.0 / (1.0 + exp(-md.raw_noise[l][0]));
                chain = (md.noise_hi[l] - md.noise_lo[l]) * sg * (1.0 - sg);
            }
            sc[4 + l] = s / div * ge * chain;
        }
        __syncthreads();
        STAMP();
        // ---- the M x M chain backward: W0 = H, W1 = Hc, dav = da
        // CB1: G1 = U^T H
        for (int e = tid; e < M * M; e += TT) {
            const int i = e / M, j = e % M;
            double s = 0.0;
#pragma unroll 4
            for (int k = i; k < M; ++k) s += Ul[k * LD + i] * W0[k * LD + j];
            W2[i * LD + j] = s;
        }
        __syncthreads();
        STAMP();
        // CB2: Y = 2 (U G1 - Hc) + a da^T + da_tot a^T (-> W1), dU_tot = 2 tril(G1^T) + gkl U (-> W0), da_tot
        for (int e = tid; e < M * M; e += TT) {
            const int i = e / M, j = e % M;
            double s = 0.0;
#pragma unroll 4
            for (int k = 0; k <= i; ++k) s += Ul[i * LD + k] * W2[k * LD + j];
            const double dti = dav[i] + gkl * al[i];
            W1[i * LD + j] = 2.0 * (s - W1[i * LD + j]) + al[i] * dav[j] + dti * al[j];
            W0[i * LD + j] = j <= i ? 2.0 * W2[j * LD + i] + gkl * Ul[i * LD + j] : 0.0;
            if (j == 0) dat[i] = dti;
        }
        __syncthreads();
        STAMP();
        // CB3: Y += dU_tot U^T;  g_LS = tril(L^-T dU_tot) - gkl diag(1 / L_S,ii);  g_m += L^-T da_tot
        {
            double* gls = gflat + g.flat_off[l] + Hl + M;
            const double* ls = md.L_S[l];
            for (int e = tid; e < M * M; e += TT) {
                const int i = e / M, j = e % M, kmax = i < j ? i : j;
                double s = 0.0;
#pragma unroll 4
                for (int k = 0; k <= kmax; ++k) s += W0[i * LD + k] * Ul[j * LD + k];
                W1[i * LD + j] += s;
                double t = 0.0;
                if (j <= i) {
#pragma unroll 4
                    for (int k = i; k < M; ++k) t += Lil[k * LD + i] * W0[k * LD + j];
                    if (i == j) t -= gkl / ls[i * M + i];
                }
                gls[i * M + j] = t;
                if (j == 0) {
                    double u = 0.0;
#pragma unroll 4
                    for (int k = i; k < M; ++k) u += Lil[k * LD + i] * dat[k];
                    gmv[l * MR + i] += u;
                }
            }
        }
        __syncthreads();
        STAMP();
        // CB4: dL = -tril(L^-T Y) + gkl diag(1 / L_ii)  (-> W2)
        for (int e = tid; e < M * M; e += TT) {
            const int i = e / M, j = e % M;
            double v = 0.0;
            if (j <= i) {
#pragma unroll 4
                for (int k = i; k < M; ++k) v -= Lil[k * LD + i] * W1[k * LD + j];
                if (i == j) v += gkl / Ll[i * LD + i];
            }
            W2[i * LD + j] = v;
        }
        __syncthreads();
        STAMP();
        // CB5: P = Phi(L^T dL)  (-> W0)
        for (int e = tid; e < M * M; e += TT) {
            const int i = e / M, j = e % M;
            double v = 0.0;
            if (j <= i) {
#pragma unroll 4
                for (int k = i; k < M; ++k) v += Ll[k * LD + i] * W2[k * LD + j];
                if (i == j) v *= 0.5;
            }
            W0[i * LD + j] = v;
        }
        __syncthreads();
        STAMP();
        // CB6: T4 = L^-T P  (-> W1)
        for (int e = tid; e < M * M; e += TT) {
            const int i = e / M, j = e % M;
            double v = 0.0;
#pragma unroll 4
            for (int k = i > j ? i : j; k < M; ++k) v += Lil[k * LD + i] * W0[k * LD + j];
            W1[i * LD + j] = v;
        }
        __syncthreads();
        STAMP();
        // CB7: T5 = T4 L^-1  (-> W2)
        for (int e = tid; e < M * M; e += TT) {
            const int i = e / M, j = e % M;
            double v = 0.0;
#pragma unroll 4
            for (int k = j; k < M; ++k) v += W1[i * LD + k] * Lil[k * LD + j];
            W2[i * LD + j] = v;
        }
        __syncthreads();
        STAMP();
        // CB8: Gram backward of dK_mm = sym(T5): both arguments are Z~ -- a pair's f gradient counts twice (-> W0)
#pragma unroll
        for (int t = 0; t < HS; ++t) hacc[t] = 0.0;
        for (int e = tid; e < M * M; e += TT) {
            const int i = e / M, j = e % M;
            const double G = 0.5 * (W2[i * LD + j] + W2[j * LD + i]);
            double dfa, dzf;
            kern_back(kind, d, zt + (l * MR + i) * ZW, zt[(l * MR + i) * ZW + DBT], zt + (l * MR + j) * ZW, hyl, ill, G, hacc,
                      dfa, dzf);
            W0[i * LD + j] = dfa;
        }
#pragma unroll
        for (int t = 0; t < HS; ++t) {
            if (slot_used(kind, d, t)) {
                const double v = wsum63(hacc[t]);
                if (lane == 63) red[wave * (HS + 1) + t] = v;
            }
        }
        __syncthreads();
        STAMP();
        if (tid < Hl) {
            const int t = slot_of(kind, d, tid);
            ghy[l * HS + tid] += red_sum<NW>(red, HS + 1, t);
        }
        if (kind)
            for (int i = tid; i < M; i += TT) {
                double s = 0.0;
#pragma unroll 4
                for (int j = 0; j < M; ++j) s += W0[i * LD + j];
                gmv[(l - 1) * MR + i] += 2.0 * s;
            }
        __syncthreads();
        STAMP();
    }

    if (do_update == 3) return;
    // ---- raw-parameter gradients into the flat vector (g_LS is there already)
    for (int e = tid; e < L * HS; e += TT) {
        const int l = e / HS, t = e % HS;
        if (t < g.H[l]) {
            int s = 0, off = 0;
            while (t >= off + seg_len(l, s, d)) { off += seg_len(l, s, d); ++s; }
            const double x = md.raw[l][s][t - off], gv = ghy[e];
            gflat[g.flat_off[l] + t] = x > 20.0 ? gv : gv / (1.0 + exp(-x));
        }
    }
    for (int e = tid; e < L * M; e += TT) gflat[g.flat_off[e / M] + g.H[e / M] + e % M] = gmv[(e / M) * MR + e % M];
    if (tid < L) gflat[g.flat_noise + tid] = sc[4 + tid];
    __syncthreads();
    STAMP();
    if (md.grad)
        for (int64_t e = tid; e < g.flat_len; e += TT) md.grad[e] = gflat[e];
    if (!do_update) return;
    // ---- Adam (torch.optim.Adam: p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)), every trainable tensor
    {
        const int64_t step = md.steps_done[0] + 1;
        const double bc1 = sc[12], bc2s = sc[13];
        // one pass over the flat vector (a pass per tensor is a dependent global round trip per tensor: 7 us for 15 tensors)
        for (int e = tid; e < (int)g.flat_len; e += TT) {
            int k = 0;
#pragma unroll
            for (int sft = NSEG / 2; sft > 0; sft >>= 1)      // first tensor whose end lies beyond e
                if (seg_end[k + sft - 1] <= e) k += sft;
            double* p = seg_ptr[k];
            if (!p) continue;
            const int start = k ? seg_end[k - 1] : 0;
            const double gi = gflat[e];
            const double mi = b1 * md.adam_m[e] + (1.0 - b1) * gi;
            const double vi = b2 * md.adam_v[e] + (1.0 - b2) * gi * gi;
            md.adam_m[e] = mi;
            md.adam_v[e] = vi;
            p[e - start] -= (lr / bc1) * mi / (sqrt(vi) / bc2s + aeps);
        }
        STAMP();
        __syncthreads();      // every thread has read steps_done / the rng call counters
        if (tid == 0) {
            md.steps_done[0] = step;
            for (int l = 1; l < L; ++l)
                if (!md.eps[l] && md.rng[l]) md.rng[l][1] += 1;
            if (blockIdx.x == 0 && md.xrng) md.xrng[1] += 1;
        }
    }
}

constexpr size_t LDS_BUDGET = 160 * 1024 - 512;      // LDS of a gfx950 CU (one workgroup per CU may hold all of it), less the kernel's static part (the geometry)
size_t lds_bytes(int MR) {
    const int LD = MR + 1, MS = MR * LD;
    const size_t n = (size_t)(3 * TLM + 3) * MS + (size_t)TLM * MR * ZW + 2 * (size_t)TLM * HS + (size_t)TLM * 2 * DBT +
                     3 * (size_t)TLM * MR + 2 * (size_t)MR + NWMAX * (HS + 1) + 24 + NSEG + NSEG / 2;
    return n * sizeof(double);
}

bool valid_model(const mobocmf_tiny_model& m) {
    if (m.L < 1 || m.L > TLM || m.M < 1 || m.M > MOBOCMF_TINY_MAX_M || m.d < 1 || m.d > DBT || m.S < 1 || m.N < 1) return false;
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

int mobocmf_tiny_flat_len(const mobocmf_tiny_model* model, int64_t* len) {
    if (!model || !len || model->L < 1 || model->L > TLM || model->M < 1 || model->d < 1) return MOBOCMF_BAD_ARG;
    Geom g;
    geom_of(*model, g);
    *len = g.flat_len;
    return MOBOCMF_OK;
}

int mobocmf_tiny_work_bytes(const mobocmf_tiny_model* model, size_t* bytes) {
    if (!model || !bytes || model->L < 1 || model->L > TLM || model->M < 1 || model->d < 1 || model->S < 1) return MOBOCMF_BAD_ARG;
    for (int l = 0; l < model->L; ++l)
        if (model->rows[l] < 1) return MOBOCMF_BAD_ARG;
    Geom g;
    geom_of(*model, g);
    *bytes = (size_t)g.work_len * sizeof(double);
    return MOBOCMF_OK;
}

int mobocmf_tiny_elbo_step(const mobocmf_tiny_model* host_models, const mobocmf_tiny_model* dev_models, int32_t n_models,
                           double lr, double beta1, double beta2, double eps, int32_t do_update, mobocmf_stream_t stream) {
    if (!host_models || !dev_models || n_models < 1 || n_models > 65535 || do_update < 0 || do_update > 4) return MOBOCMF_BAD_ARG;
    int mmax = 0;
    int64_t pmax = 0;
    int cmax = 0;
    for (int i = 0; i < n_models; ++i) {
        if (!valid_model(host_models[i]) || (do_update == 3 && !host_models[i].grad)) return MOBOCMF_BAD_ARG;
        if (do_update == 4) {      // S = 1, seeds and top moments present, a coupling (checked in depth by the caller's binding)
            const mobocmf_tiny_model& m = host_models[i];
            if (!m.coupling || m.S != 1 || !m.seed_gmean || !m.top_mean || m.role < 0 || m.role > 1 || m.role_index < 0 ||
                m.role_index > 7 || m.coupling != host_models[0].coupling)
                return MOBOCMF_BAD_ARG;
        }
        if (host_models[i].M > mmax) mmax = host_models[i].M;
        Geom g;
        geom_of(host_models[i], g);
        if (g.pool_len > pmax) pmax = g.pool_len;
        if (g.ncmax > cmax) cmax = g.ncmax;
    }
    hipStream_t s = (hipStream_t)stream;
    const int MR = mmax <= 16 ? 16 : 32;
    size_t shm = lds_bytes(MR);
    const size_t pool = (size_t)pmax * sizeof(double);
    const int pool_in_lds = shm + pool <= LDS_BUDGET ? 1 : 0;
    if (pool_in_lds) shm += pool;
    // 512 threads when a phase has more than two elements per thread of a 256-thread workgroup (wide panels: conditioned
    // training, acquisition): the phases are latency-bound per thread, more threads walk more elements at once
    const bool wide = (int64_t)mmax * cmax > 2 * 256;
    const int slot = (MR == 32 ? 2 : 0) + (wide ? 1 : 0);
    const void* fn = MR == 16 ? (wide ? (const void*)tiny_step_kernel<16, 512> : (const void*)tiny_step_kernel<16, 256>)
                              : (wide ? (const void*)tiny_step_kernel<32, 512> : (const void*)tiny_step_kernel<32, 256>);
    // more than the default 64 KB of dynamic LDS needs the function attribute, which is per DEVICE: one write-once bit per
    // (kernel instantiation, device) remembers that it was set (devices beyond 63: set on every call, it is idempotent)
    static std::atomic<uint64_t> granted[4] = {{0}, {0}, {0}, {0}};
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess) return MOBOCMF_HIP_ERROR;
    const uint64_t bit = devid >= 0 && devid < 64 ? 1ull << devid : 0ull;
    if (shm > 64 * 1024 && !(granted[slot].load() & bit)) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET) != hipSuccess)
            return MOBOCMF_HIP_ERROR;
        granted[slot].fetch_or(bit);
    }
    if (do_update == 4) {
        // the in-launch barrier needs every workgroup of the launch resident at once: bound the grid by what THIS device holds
        // of this kernel with this much LDS (a partitioned or smaller device holds fewer; MOBOCMF_BAD_ARG sends the caller to
        // the three-launch form)
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, wide ? 512 : 256, shm) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid) != hipSuccess)
            return MOBOCMF_HIP_ERROR;
        if (n_models > 64 || (int64_t)n_models > (int64_t)per_cu * cus) return MOBOCMF_BAD_ARG;
    }
#define LAUNCH(MR_, TT_)                                                                                              \
    hipLaunchKernelGGL((tiny_step_kernel<MR_, TT_>), dim3((unsigned)n_models), dim3(TT_), shm, s, dev_models, lr, beta1, \
                       beta2, eps, do_update, pool_in_lds)
    if (MR == 16) { if (wide) LAUNCH(16, 512); else LAUNCH(16, 256); }
    else { if (wide) LAUNCH(32, 512); else LAUNCH(32, 256); }
#undef LAUNCH
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}

}  // extern "C"
