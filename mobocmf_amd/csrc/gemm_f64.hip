// FP64 MFMA GEMM for gfx950, the dominant kernel of the MFDGP layer.
//
//   C[Mr x Nc] (+)= alpha * A[Mr x Kd] * B        B_T=0: B is [Kd x Nc] (n contiguous)
//                                                 B_T=1: B is [Nc x Kd] (k contiguous)  -> A * B^T
//
// Matrix instruction: v_mfma_f64_16x16x4_f64 (GEMM_MI == 16, the default since round 4).  Measured on MI355X with the
// instruction issued on VGPR tuples (tools/mfma_peak.hip, profiles/r04_mfma_peak.txt): 77.1 TFLOP/s, against 73.3 for
// v_mfma_f64_4x4x4_4b_f64 and 67-71 for v_fma_f64.  (Rounds 1-3 ran the 4x4x4 form -- -DGEMM_MI=4 still builds it -- because the
// builtin's accumulator copies made the 16x16x4 microbenchmark read 36 TFLOP/s: an artefact)  Lane map of
// the 16x16x4 form: lane (li = l & 15, lk = l >> 4) feeds A[row li][k = 4 lk + kq] to MFMA kq of a K step and holds the result
// rows 4 r + lk, column li.  The 4x4x4 form uses its four independent blocks as four column groups of one 4 x 16 output strip:
// lane (kk = l>>4, b = (l>>2)&3, i = l&3) holds A[row i][k] -- the same 4 rows in every block, an LDS broadcast read -- and
// B[k][col 4b+j]; result lane (i = l>>4, col = l&15) (layout verified with one-hot operands, tools/probe_mfma444.hip).
//
// Workgroup = 256 threads = 4 wavefronts (2 x 2), tile 128 x 128, K step 16.  Each wavefront owns 64 x 64 =
// 64 accumulator registers (128 VGPRs).  Tiles are staged global -> LDS by global_load_lds_dwordx4 (LDS-DMA,
// no staging VGPRs, no ds_write), double buffered: stage t+1 is in flight while stage t is multiplied.
// LDS-DMA writes 64 lanes x 16 B contiguously, so the LDS images are unpadded; bank conflicts of the fragment
// reads are removed by an XOR swizzle of the 16-byte chunk applied on the SOURCE address of the DMA and on
// the fragment read (same involution on both sides):
//   A / B_T image [128 rows][16 k]   : chunk ^= ((row >> 1) & 1) << 2
//   B image       [16 k][128 cols]   : none -- every lane owns ADJACENT column pairs (columns 2*li, 2*li+1 and 32 + 2*li,
//                                      33 + 2*li of its wavefront's 64), so a B fragment is one ds_read_b128 per k and
//                                      column pair, and the lane groups the LDS services a b128 read in ({0-3,12-15,
//                                      20-27}, ...) already hit 16 different 16-byte bank groups; the epilogues move 16
//                                      bytes per lane (half the store / load instructions of one column per lane)
// Within a K step lane group kk handles k = 4*kk + ks (ks = 0..3), so that per-k weights are contiguous.
// Triangular operands only visit the non-zero k range of their tile.  blockIdx -> tile mapping is XCD-aware.
#include "common.h"
#include <type_traits>
#include <atomic>

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

#define BM 128
#define BN 128
#define BK 16
#define KL_MAX 1024   // active K blocks one k-slice of an A B^T product can hold (GemmArgs.kact; launch_gemm checks)
#define TILE_ELEMS (BM * BK)   // 2048 doubles = 16 KiB per operand tile

// Matrix instruction of gemm_f64_kernel's main loop.  16: v_mfma_f64_16x16x4_f64 (round 4; 77 TFLOP/s sustained, one fragment
// value feeds 2048 flops).  4: v_mfma_f64_4x4x4_4b_f64 (rounds 1-3; 73 TFLOP/s, 512 flops per fragment value) -- kept for
// A/B builds (EXTRA_HIPCC_FLAGS=-DGEMM_MI=4).  Accumulator lane map is the SAME for both: acc[mt][nt][r] of lane (li, lk) is
// row mt*32 + wr*16 + 4r + lk, so every epilogue serves either.  What differs is the operand side:
//   4x4x4:   lane (kk = l>>4, i = l&3) feeds A[4r + i][k = kk] to MFMA r (four MFMAs cover 16 rows), 16 ds_read_b128 of A per kpair
//   16x16x4: lane (li, lk) feeds A[li][k = lk] -- one MFMA covers the 16 rows x 16 columns x 4 k; with MFMA kq of a K step
//            taking k = 4*lk + kq a lane's four A values are 32 contiguous bytes of its image row: two ds_read_b128 per
//            16-row group and K step (16 b128 per K step and wavefront where the 4x4x4 form issues 40, 64 MFMAs instead of 256).
// Image swizzle (XOR on the 16-byte chunk index of a [rows][16 k] image row, applied on the DMA source and on the read):
// the four 16-lane groups a ds_read_b128 is serviced in ({0-3,12-15,20-27}, ...) must hit 16 different 16-byte slots of the
// 256-byte bank row.  16x16x4: a group holds 8 rows x chunk c and 8 other rows x chunk c^2; slot = (row&1)*8 + chunk, so the
// four rows of one parity need four different XOR values that keep bit 1 clear: s = bit1(row) | bit3(row) << 2.
#ifndef GEMM_MI
#define GEMM_MI 16
#endif
#if GEMM_MI == 16
#define IMG_SWZ(row) ((((row) >> 1) & 1) | ((((row) >> 3) & 1) << 2))
#else
#define IMG_SWZ(row) ((((row) >> 1) & 1) << 2)
#endif


// Cross-lane sums without the LDS crossbar (ds_bpermute round trips): DPP moves inside a 16-lane row, permlane swaps across
// rows.  row16_sum: every lane ends with the sum over its row of 16 lanes (quad butterflies, then the mirrored half, then
// the mirrored row).  rows4_sum: every lane ends with the sum over the 4 lanes {l mod 16 + 16 r} (v_permlane16_swap pairs
// odd with even rows, v_permlane32_swap the two halves: fed the same register twice they yield the two butterfly operands).
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);      // row_half_mirror
    v += dpp_mov<0x140>(v);      // row_mirror
    return v;
}
__device__ __forceinline__ double rows4_sum(double v) {
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
    }
    return v;
}

__device__ __forceinline__ void glds16(const double* gsrc, double* lds_wave_base) {
    // 64 lanes x 16 B: lane l lands at lds_wave_base + 16*l bytes
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Diagnostic build only (-DGEMM_STAMPS, tools/gemm_stamps.py): wall-clock stamps (100 MHz s_memrealtime) of one lane per
// workgroup at the phase boundaries, written to a debug buffer that nothing else reads.  Compiled out of the product.
#ifdef GEMM_STAMPS
__device__ unsigned long long* g_stamp_buf = nullptr;
extern "C" int mobocmf_debug_set_stamps(unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : 3;
}
#define STAMP(i)                                                                                            \
    do {                                                                                                    \
        if (g_stamp_buf && threadIdx.x == 0) g_stamp_buf[(int64_t)blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define STAMP_ID()                                                                                          \
    do {                                                                                                    \
        if (g_stamp_buf && threadIdx.x == 0) {                                                              \
            unsigned hw, xcc;                                                                               \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                               \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                             \
            g_stamp_buf[(int64_t)blockIdx.x * 32 + 15] = ((unsigned long long)xcc << 32) | hw;              \
        }                                                                                                   \
    } while (0)
// stamps INSIDE one light and one dense K step of part 0, kept in scalar registers until the kernel's end (a store per
// stamp would sit in the vector-memory queue the step's own s_waitcnt vmcnt(0) drains)
#define STEP_STAMPS_DECL unsigned long long sst[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define STEP_STAMP(i) do { if (sbase >= 0) sst[sbase + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define STEP_STAMPS_FLUSH                                                                                   \
    do {                                                                                                    \
        if (g_stamp_buf && threadIdx.x == 0)                                                                \
            for (int q = 0; q < 10; ++q) g_stamp_buf[(int64_t)blockIdx.x * 32 + 16 + q] = sst[q];           \
    } while (0)
#else
#define STAMP(i)
#define STAMP_ID()
#define STEP_STAMPS_DECL
#define STEP_STAMP(i)
#define STEP_STAMPS_FLUSH
#endif

// RM = rows of a tile (128 or 64; columns always 128).  RM = 64 (NN form only): 64 x 128 tiles, each wavefront 32 x 64, half
// the MFMAs per K step, 48 KB of LDS and <= 168 VGPRs per workgroup -> THREE workgroups per CU.  A triangular A operand is
// resolved in 64-row blocks then: the k range of a row block ends (lower) / starts (upper) at a 64 boundary, so the product
// walks n(n+1)/2 = 36 half-height block steps for M = 512 where 128-row tiles walk 10 full-height ones (= 40), of an ideal 32;
// and N' = 8192 makes 512 tiles instead of 256 -- every CU busy after pairing.
template <bool B_T, bool TRI, int EPI, int RM>
__global__ __launch_bounds__(256, RM == 64 ? 3 : 2) void gemm_f64_kernel(GemmArgs g, int nrb, int64_t ncb, int splitk, int pair) {
    static_assert(RM == 128 || (RM == 64 && !B_T), "tile heights: 128, or 64 for the A B form");
    constexpr int MT = RM / 32;               // 16-row groups per wavefront (interleaved between the two row-wavefronts)
    constexpr int A_ELEMS = RM * BK;          // A image [RM rows][16 k]
    constexpr int STAGE_ELEMS = A_ELEMS + TILE_ELEMS;
    __shared__ __attribute__((aligned(16))) double lds[2 * STAGE_ELEMS];   // [buf][A | B]
    __shared__ int kl[B_T ? KL_MAX : 1];      // A B^T over active K blocks (GemmArgs.kact): block ids of this slice
    __shared__ int kl_wsum[4];
    if (g.skip_if_zero && *g.skip_if_zero == 0) return;
    STAMP(0);
    STAMP_ID();
    // ---- block id -> (tile, k-slice).  Blocks b, b+8, ... share an XCD (and its L2):
    //  * no split-K: the row blocks that re-read the same 128-column panel of B run back to back on one XCD;
    //  * split-K: all tiles of one k-slice (they share the slice's rows of A and B) run back to back on one XCD.
    int64_t id = blockIdx.x;
    if (g.dual_flag) {      // dual launch: the second half of the grid is the conditional twin (common.h)
        const int64_t half = (int64_t)(gridDim.x >> 1);
        if (id >= half) {
            if (*g.dual_flag == 0) return;
            id -= half;
            g.bscale = g.bscale2;
            g.C = g.C2;
        }
    }
    int rb, z = 0, sk_eff = splitk;      // sk_eff: k slices of THIS tile
    int64_t cb;
    if (g.batched) {
        z = blockIdx.z;
        cb = id / nrb;
        rb = nrb - 1 - (int)(id % nrb);
    } else if (splitk > 1 && g.lower_out && g.splitk_diag > 0) {
        // two tile classes with their own slice counts: the nrb (nrb - 1) / 2 tiles below the diagonal first, then the nrb
        // diagonal ones; inside a class as below (XCD-aware when the counts allow it)
        const int64_t nF = (int64_t)nrb * (nrb - 1) / 2, nblkF = nF * splitk;
        const bool diag = id >= nblkF;
        const int64_t id2 = diag ? id - nblkF : id, nT = diag ? nrb : nF;
        sk_eff = diag ? g.splitk_diag : splitk;
        int64_t t;
        if ((sk_eff & 7) == 0 && (nblkF & 7) == 0) {
            const int64_t xcd = id2 & 7, seq = id2 >> 3;
            t = seq % nT;
            z = (int)((seq / nT) * 8 + xcd);
        } else {
            t = id2 % nT;
            z = (int)(id2 / nT);
        }
        if (diag) {
            rb = (int)t;
            cb = t;
        } else {
            int r = 1;
            while ((int64_t)r * (r + 1) / 2 <= t) ++r;
            rb = r;
            cb = t - (int64_t)r * (r - 1) / 2;
        }
    } else if (splitk > 1) {
        const int64_t ntile = g.lower_out ? (int64_t)nrb * (nrb + 1) / 2 : (int64_t)nrb * ncb;
        int64_t t;
        if ((splitk & 7) == 0) {
            const int64_t xcd = id & 7, seq = id >> 3;
            t = seq % ntile;
            z = (int)((seq / ntile) * 8 + xcd);
        } else {
            t = id % ntile;
            z = (int)(id / ntile);
        }
        if (g.lower_out) {
            int r = 0;
            while ((int64_t)(r + 1) * (r + 2) / 2 <= t) ++r;
            rb = r;
            cb = t - (int64_t)r * (r + 1) / 2;
        } else {
            cb = t / nrb;
            rb = (int)(t % nrb);
        }
    } else {
        // plain launch: nslot row-block slots per column block.  With a triangular A the work of row block rb is
        // proportional to rb+1 (lower) / nrb-rb (upper); workgroups are dealt to CUs round-robin, so unequal tiles
        // leave most CUs idle (measured: triangular as slow as dense).  `pair` makes every workgroup do the two
        // row blocks (p, nrb-1-p): equal work for all.
        const int nslot = pair ? (nrb + 1) / 2 : nrb;
        int sl;
        if ((ncb & 7) == 0) {
            int64_t xcd = id & 7, slot = id >> 3;
            cb = (slot / nslot) * 8 + xcd;
            sl = (int)(slot % nslot);
        } else {
            cb = id / nslot;
            sl = (int)(id % nslot);
        }
        rb = pair ? sl : nrb - 1 - sl;
    }
    if (g.lower_out && cb > rb) return;

    const double* A = g.A;
    const double* B = g.B;
    double* C = g.C;
    if (g.batched) {
        const int zl = g.zlayers > 1 ? z % g.zlayers : 0;      // layer, batch index inside the layer
        const int zb = g.zlayers > 1 ? z / g.zlayers : z;
        A += zb * g.strideA + zl * g.zsA;
        B += zb * g.strideB + zl * g.zsB;
        C += zb * g.strideC + zl * g.zsC;
    }
    if (!g.batched && splitk > 1) {
        C += z * g.slab_stride;
        if (g.zlayers > 1) {      // k-sliced AND layer-batched: blockIdx.z = layer (slabs of layer l at C + l*zsC)
            A += blockIdx.z * g.zsA;
            B += blockIdx.z * g.zsB;
            C += blockIdx.z * g.zsC;
        }
    }
    const int nparts = (pair && rb != nrb - 1 - rb) ? 2 : 1;
    const int rb_first = rb;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // wave-uniform, and told so: the LDS destinations of the stage's DMA instructions (M0) and the wavefront's tile offsets are
    // then scalar arithmetic instead of a vector computation + v_readfirstlane per DMA instruction and K step (round 4: -10
    // VGPRs, 1-4 % per launch at the 512-row shapes, C3 996 -> 1006 steps/s in a same-box A/B)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 15, lk = lane >> 4;

    if constexpr (!B_T) {
        // inactive column block (GemmArgs.colact): its share of the product is exactly zero and is not computed
        if (g.colact && g.colact[cb] == 0) {
            if (EPI == EPI_DA && g.rowdot_part) {      // the row-dot partials are summed over ALL column blocks: zeros
                for (int part = 0; part < nparts; ++part) {
                    const int rbp = part == 0 ? rb_first : nrb - 1 - rb_first;
                    for (int i = tid; i < 2 * RM; i += 256)
                        g.rowdot_part[((int64_t)cb * 2 + i / RM) * g.Mr + (int64_t)rbp * RM + i % RM] = 0.0;
                }
            }
            return;
        }
    }

    // ---- LDS-DMA staging maps (one instruction = 64 lanes x 16 B = 1 KiB of the image)
    // A / B_T image: lane -> (row = 8*wave + l/8 (+32 per round), physical chunk l%8); source chunk swizzled
    const int a_row = tid >> 3;
    const int a_lchk = (tid & 7) ^ IMG_SWZ(a_row);
    const double* Ag0 = A + (int64_t)a_row * g.lda + a_lchk * 2;
    // B image: lane -> (k = wave (+4 per round), chunk l)
    const double* Bg = B_T ? B + (cb * BN + a_row) * g.ldb + a_lchk * 2
                           : B + (int64_t)wave * g.ldb + cb * BN;
    auto stage = [&](const double* Ag, int64_t k, int buf) {
        double* As = lds + buf * STAGE_ELEMS;
        double* Bs = As + A_ELEMS;
#pragma unroll
        for (int r = 0; r < RM / 32; ++r) glds16(Ag + (int64_t)(32 * r) * g.lda + k, As + (r * 4 + wave) * 128);
        if (B_T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) glds16(Bg + (int64_t)(32 * r) * g.ldb + k, Bs + (r * 4 + wave) * 128);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                glds16(Bg + (k + 4 * r) * g.ldb + lane * 2, Bs + (r * 4 + wave) * 128);
        }
    };

    // ---- fragment read offsets (doubles).  k pairs (2*kp', 2*kp'+1) are adjacent: one ds_read_b128 per pair.
    // kpair p (0,1) covers ks = 2p, 2p+1, i.e. k = 4*lk + 2p + {0,1}: logical 16-B chunk 2*lk + p.
#if GEMM_MI == 16
    // 16x16x4: lane (li, lk) feeds row li of a 16-row group with k = 4*lk + kq (kq = 0..3: MFMA number kq of the K step takes
    // lane group lk's k = 4*lk + kq on both operands), i.e. logical 16-byte chunks 2*lk and 2*lk + 1 of its row.
    const int swA = IMG_SWZ(li);
#else
    const int swA = ((lane >> 1) & 1) << 2;
#endif
    const int colP0 = ((2 * lk + 0) ^ swA) << 1, colP1 = ((2 * lk + 1) ^ swA) << 1;
    // the two row-wavefronts own INTERLEAVED 16-row groups (group 2*mt + wr): inside a triangular diagonal block both
    // then skip a similar share of structurally-zero groups (critical path 20/32 of a dense block instead of 26/32)
#if GEMM_MI == 16
    const int a_base = (wr * 16 + li) * BK;                         // + mt*32*BK + colP
#else
    const int a_base = (wr * 16 + (lane & 3)) * BK;                 // + (mt*32 + 4r)*BK + colP
#endif
    // A B^T: the two column-wavefronts own INTERLEAVED 16-column groups (group 2*nt + wc), like the row-wavefronts their row
    // groups: in a diagonal tile of a symmetric product the 16 x 16 blocks strictly above the diagonal (column group > row
    // group) then spread evenly over the four wavefronts
    const int bt_base = (wc * 16 + li) * BK;                        // B_T: + nt*32*BK + colP
    // B: k row = 4*lk + ks; accumulator column nt of a lane = wc*64 + (nt>>1)*32 + 2*li + (nt&1)
    const int bn_base = (4 * lk) * BN + wc * 64 + 2 * li;           // + ks*BN + (nt>>1)*32

  STEP_STAMPS_DECL
  for (int part = 0; part < nparts; ++part) {
    // Paired row blocks: the LONG one first, and the two workgroups that share a column block of B -- pairs (0, nrb-1)
    // and (1, nrb-2), dispatched back to back on one XCD -- walk k in the SAME direction at the same time: the first
    // (long) parts from the end of k they have in common, the second parts back towards it.  The slab of B one of them
    // has just pulled into the XCD's L2 is then a hit for the other: B is fetched ~1.5x instead of 2.5x (PMC FETCH_SIZE).
    const bool upper = (g.tri & TRI_UPPER_A) != 0;
    const bool long_first_is_high = !upper;          // lower: row block nrb-1-p has the long k range; upper: row block p
    rb = ((part == 0) == long_first_is_high) ? nrb - 1 - rb_first : rb_first;
    if (nparts == 1) rb = rb_first;
    const bool rev = TRI && nparts == 2 && (upper ? part == 0 : part == 1);   // walk k downwards
    int64_t k0 = 0, k1 = g.Kd;
    if (g.tri & TRI_LOWER_A) {
        int64_t e = (int64_t)(rb + 1) * RM;
        if (k1 > e) k1 = e;
    }
    if (g.tri & TRI_UPPER_A) {
        int64_t b = (int64_t)rb * RM;
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
    bool compact = false;      // K steps of this slice come from the active-block list kl[] (GemmArgs.kact)
    int64_t ct0 = 0;           // compact: index of the slice's first step among all active steps
    int co0 = 0;               //          ordinal of the active block kl[0]
    if (!g.batched && splitk > 1) {   // slice the tile's own non-zero k range
        bool sliced = false;
        if constexpr (B_T) {
            if (g.kact) {
                // Active 128-column blocks of K, in ascending order: every thread counts a run of blocks, a workgroup scan
                // gives each run its first ordinal; the slice takes steps [z per, (z+1) per) of the 8 nact active steps.
                const int nb = (int)(g.Kd / 128), run = (nb + 255) / 256;
                const int rb0 = tid * run < nb ? tid * run : nb, rb1 = rb0 + run < nb ? rb0 + run : nb;
                int cnt = 0;
                for (int b = rb0; b < rb1; ++b) cnt += g.kact[b] != 0 ? 1 : 0;
                int inc = cnt;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int v = __shfl_up(inc, off);
                    if (lane >= off) inc += v;
                }
                if (lane == 63) kl_wsum[wave] = inc;
                __syncthreads();
                int ord = inc - cnt;
                for (int w = 0; w < wave; ++w) ord += kl_wsum[w];
                const int64_t nkt = (int64_t)(kl_wsum[0] + kl_wsum[1] + kl_wsum[2] + kl_wsum[3]) * (128 / BK);
                const int64_t per = (nkt + sk_eff - 1) / sk_eff;
                const int64_t tb = z * per < nkt ? z * per : nkt, te = tb + per < nkt ? tb + per : nkt;
                ct0 = tb;
                co0 = (int)(tb / (128 / BK));
                const int co1 = te > tb ? (int)((te - 1) / (128 / BK)) : co0 - 1;
                for (int b = rb0; b < rb1; ++b)
                    if (g.kact[b] != 0) {
                        if (ord >= co0 && ord <= co1 && ord - co0 < KL_MAX) kl[ord - co0] = b;
                        ++ord;
                    }
                __syncthreads();
                compact = true;
                sliced = true;
                k0 = 0;
                k1 = (te - tb) * BK;      // only the step COUNT matters below: offsets come from KOFF
            }
        }
        if (!sliced) {
            int64_t nkt = k1 > k0 ? (k1 - k0) / BK : 0;
            int64_t per = (nkt + sk_eff - 1) / sk_eff;
            int64_t b = k0 + z * per * BK, e = b + per * BK;
            k0 = b < k1 ? b : k1;
            k1 = e < k1 ? e : k1;
        }
    }
    const double* Ag = Ag0 + (int64_t)rb * RM * g.lda;

    v4f64 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};

    const int64_t nk = (k1 > k0) ? (k1 - k0) / BK : 0;
#define KSTEP(KT) (rev ? nk - 1 - (KT) : (KT))   /* iteration -> K step of the tile */
    // iteration -> offset along K
    auto koff = [&](int64_t kt) -> int64_t {
        if (B_T && compact) {
            const int64_t t = ct0 + kt;
            return (int64_t)kl[(int)(t / (128 / BK)) - co0] * 128 + (t % (128 / BK)) * BK;
        }
        return k0 + KSTEP(kt) * BK;
    };
#define KOFF(KT) koff(KT)
    if (nk > 0) stage(Ag, KOFF(0), 0);

    // LDS-DMA data is ordered for other wavefronts' ds_reads only by the issuing wavefront's vmcnt wait followed by
    // a barrier; the waits are written out (hipcc adds them only when it sees the DMA in the same scheduling scope)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(1 + 4 * part);      // first stage landed
    v4f64 w_nxt = (v4f64){1.0, 1.0, 1.0, 1.0};
    if (B_T && g.bscale && nk > 0) w_nxt = *(const v4f64*)(g.bscale + KOFF(0) + 4 * lk);
    // Software pipeline of one K step over 8 groups g = (kpair p, row tile mt): the A fragments of group g+1
    // (4 x ds_read_b128) and, at a kpair boundary, the B fragments of the next kpair are read while the 32 MFMAs of
    // group g issue.  sched_barrier(0) pins the group boundaries so the register allocator sees two fragment sets, not
    // eight.  (Literal indices through macros: lambdas / late-unrolled loops left the fragment arrays in scratch.)
#if GEMM_MI == 16
    // ---- 16x16x4 main loop.  All fragments of a K step are read up front (B: 8 ds_read_b128, A: 2 per 16-row group), the
    // next stage's DMA is issued behind the first of them, then the MFMAs go row group by row group -- a wavefront issues one
    // 64-cycle MFMA after the other, so the reads of the later groups return under the first group's MFMAs and the
    // co-resident wavefronts' bursts; no software pipeline of fragment sets is needed (the 4x4x4 form's eight groups).
    v2f64 fa[MT][2];             // A: [row group][chunk]: k = 4*lk + 2*chunk + {0, 1}
    double fb[4][4];             // B: [kq][nt]: k = 4*lk + kq
#define LOAD_A16(MT_)                                                                                       \
    fa[MT_][0] = *(const v2f64*)(As + a_base + (MT_) * 32 * BK + colP0);                                    \
    fa[MT_][1] = *(const v2f64*)(As + a_base + (MT_) * 32 * BK + colP1);
#define LOAD_B16                                                                                            \
    if (B_T) {                                                                                              \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                    \
            const v2f64 v0 = *(const v2f64*)(Bs + bt_base + t * 32 * BK + colP0);                           \
            const v2f64 v1 = *(const v2f64*)(Bs + bt_base + t * 32 * BK + colP1);                           \
            fb[0][t] = v0[0];                                                                               \
            fb[1][t] = v0[1];                                                                               \
            fb[2][t] = v1[0];                                                                               \
            fb[3][t] = v1[1];                                                                               \
        }                                                                                                   \
        if (g.bscale) {                                                                                     \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                  \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) fb[q][t] *= w4[q];                           \
        }                                                                                                   \
    } else {                                                                                                \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                      \
            _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                \
                const v2f64 v = *(const v2f64*)(Bs + bn_base + q * BN + h * 32);                            \
                fb[q][2 * h] = v[0];                                                                        \
                fb[q][2 * h + 1] = v[1];                                                                    \
            }                                                                                               \
    }
// the first JN column groups of row group MT_ over the four k quads of the step
#define MMA16(MT_, JN)                                                                                      \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                          \
        _Pragma("unroll") for (int j = 0; j < (JN); ++j)                                                   \
            acc[MT_][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[MT_][q >> 1][q & 1], fb[q][j], acc[MT_][j], 0, 0, 0);
#define MMA_ALL(MT_) MMA16(MT_, 4)
#define MMA_IF(MT_)                                                                                         \
    if (act & (1 << (MT_))) { MMA16(MT_, 4) }
#endif
#if GEMM_MI != 16
    v2f64 a0[4], a1[4];          // A fragment sets (even / odd group)
    double b0[2][4], b1[2][4];   // B fragment sets (kpair 0 / 1): [ks&1][nt]
#define LOAD_A(dst, P, MT)                                                                                  \
    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                          \
        dst[r] = *(const v2f64*)(As + a_base + ((MT) * 32 + 4 * r) * BK + ((P) ? colP1 : colP0));
#define LOAD_B(dst, P)                                                                                      \
    if (B_T) {                                                                                              \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                    \
            v2f64 v = *(const v2f64*)(Bs + bt_base + t * 32 * BK + ((P) ? colP1 : colP0));                  \
            dst[0][t] = v[0];                                                                               \
            dst[1][t] = v[1];                                                                               \
        }                                                                                                   \
        if (g.bscale) {                                                                                     \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                \
                dst[0][t] *= w4[2 * (P)];                                                                   \
                dst[1][t] *= w4[2 * (P) + 1];                                                               \
            }                                                                                               \
        }                                                                                                   \
    } else {                                                                                                \
        _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                      \
            _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                \
                const v2f64 v = *(const v2f64*)(Bs + bn_base + (2 * (P) + e) * BN + h * 32);                \
                dst[e][2 * h] = v[0];                                                                       \
                dst[e][2 * h + 1] = v[1];                                                                   \
            }                                                                                               \
    }
#define MMA_DO(asrc, bsrc, MT)                                                                              \
    _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                          \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                  \
                acc[MT][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(asrc[r][e], bsrc[e][j], acc[MT][j][r], 0, 0, 0);
#define MMA_ALL(asrc, bsrc, MT) MMA_DO(asrc, bsrc, MT) __builtin_amdgcn_sched_barrier(0);
#define MMA_IF(asrc, bsrc, MT)                                                                              \
    if (act & (1 << (MT))) { MMA_DO(asrc, bsrc, MT) }                                                       \
    __builtin_amdgcn_sched_barrier(0);
#endif
// The LDS-DMA of the NEXT stage is issued behind the step's first fragment reads (it writes the other buffer): in front
// of them its address arithmetic and eight issues sat on the critical path between the barrier and the first MFMA.
#define STAGE_NEXT                                                                                          \
        if (kt + 1 < nk) {                                                                                  \
            /* the weight load is issued BEFORE the DMA: waiting for it never drains the DMA (vmcnt is in order) */ \
            const int64_t knext = KOFF(kt + 1);                                                             \
            if (B_T && g.bscale) w_nxt = *(const v4f64*)(g.bscale + knext + 4 * lk);                        \
            stage(Ag, knext, buf ^ 1);                                                                      \
        }                                                                                                   \
        STEP_STAMP(1);
#if GEMM_MI == 16
#define KSTEP_STD(MMA)                                                                                      \
        LOAD_B16                                                                                            \
        LOAD_A16(0)                                                                                         \
        STAGE_NEXT                                                                                          \
        LOAD_A16(MT - 1)                                                                                    \
        if constexpr (MT == 4) {                                                                            \
            LOAD_A16(1)                                                                                     \
            LOAD_A16(2)                                                                                     \
        }                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        MMA(0)                                                                                              \
        if constexpr (MT == 4) {                                                                            \
            MMA(1)                                                                                          \
            MMA(2)                                                                                          \
        }                                                                                                   \
        MMA(MT - 1)                                                                                         \
        /* the MFMAs stay IN FRONT of the step's closing wait + barrier: they only touch registers, so without this fence   \
           hipcc hoists "s_waitcnt vmcnt(0); s_barrier" to right behind the first MFMA -- the wavefront then waits for the   \
           next stage's DMA (a full L2 / HBM round trip) before it has issued anything that could hide it (first build of   \
           this loop: 5-10 % slower than the 4x4x4 form at every shape) */                                                 \
        __builtin_amdgcn_sched_barrier(0);
// Diagonal tile of a symmetric A B^T product (sym_out): with row group 2*mt + wr and column group 2*nt + wc a 16 x 16
// block lies on or below the diagonal iff 2*nt + wc <= 2*mt + wr, i.e. nt <= mt for three of the wavefronts (KSTEP_LE:
// 10 of 16 blocks) and nt < mt for (wr, wc) = (0, 1) (KSTEP_LT: 6 of 16); the slab reduction mirrors the rest.
#define KSTEP_LE                                                                                            \
        LOAD_B16                                                                                            \
        LOAD_A16(0)                                                                                         \
        STAGE_NEXT                                                                                          \
        LOAD_A16(1) LOAD_A16(2) LOAD_A16(3)                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        MMA16(0, 1) MMA16(1, 2) MMA16(2, 3) MMA16(3, 4)                                                     \
        __builtin_amdgcn_sched_barrier(0);
#define KSTEP_LT                                                                                            \
        LOAD_B16                                                                                            \
        LOAD_A16(1)                                                                                         \
        STAGE_NEXT                                                                                          \
        LOAD_A16(2) LOAD_A16(3)                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        MMA16(1, 1) MMA16(2, 2) MMA16(3, 3)                                                                 \
        __builtin_amdgcn_sched_barrier(0);
#else
// The standard K step: all four row groups, MMA = MMA_ALL or the skipping MMA_IF.
#define KSTEP_STD(MMA)                                                                                      \
        LOAD_B(b0, 0)                                                                                       \
        LOAD_A(a0, 0, 0)                                                                                    \
        STAGE_NEXT                                                                                          \
        if constexpr (MT == 4) {                                                                            \
            LOAD_A(a1, 0, 1) MMA(a0, b0, 0)                                                                 \
            LOAD_A(a0, 0, 2) MMA(a1, b0, 1)                                                                 \
            LOAD_A(a1, 0, 3) MMA(a0, b0, 2)                                                                 \
            LOAD_A(a0, 1, 0) LOAD_B(b1, 1) MMA(a1, b0, 3)                                                   \
            LOAD_A(a1, 1, 1) MMA(a0, b1, 0)                                                                 \
            LOAD_A(a0, 1, 2) MMA(a1, b1, 1)                                                                 \
            LOAD_A(a1, 1, 3) MMA(a0, b1, 2)                                                                 \
            MMA(a1, b1, 3)                                                                                  \
        } else {      /* 64-row tiles: four groups of 32 MFMAs */                                           \
            LOAD_A(a1, 0, MT - 1) MMA(a0, b0, 0)                                                            \
            LOAD_A(a0, 1, 0) LOAD_B(b1, 1) MMA(a1, b0, MT - 1)                                              \
            LOAD_A(a1, 1, MT - 1) MMA(a0, b1, 0)                                                            \
            MMA(a1, b1, MT - 1)                                                                             \
        }
// Diagonal tile of a symmetric A B^T product (sym_out): with row group 2*mt + wr and column group 2*nt + wc a 16 x 16
// block lies on or below the diagonal iff 2*nt + wc <= 2*mt + wr, i.e. nt <= mt for three of the wavefronts (KSTEP_LE:
// 10 of 16 blocks) and nt < mt for (wr, wc) = (0, 1) (KSTEP_LT: 6 of 16); the slab reduction mirrors the rest.
// GRPJ = the first JN column groups of row group MT, compile-time counts, no branch inside the K step.
#define GRPJ(asrc, bsrc, MT, JN)                                                                            \
    _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                          \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                      \
            _Pragma("unroll") for (int j = 0; j < (JN); ++j)                                               \
                acc[MT][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(asrc[r][e], bsrc[e][j], acc[MT][j][r], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);
#define KSTEP_LE                                                                                            \
        LOAD_B(b0, 0)                                                                                       \
        LOAD_A(a0, 0, 0)                                                                                    \
        STAGE_NEXT                                                                                          \
        LOAD_A(a1, 0, 1) GRPJ(a0, b0, 0, 1)                                                                 \
        LOAD_A(a0, 0, 2) GRPJ(a1, b0, 1, 2)                                                                 \
        LOAD_A(a1, 0, 3) GRPJ(a0, b0, 2, 3)                                                                 \
        LOAD_A(a0, 1, 0) LOAD_B(b1, 1) GRPJ(a1, b0, 3, 4)                                                   \
        LOAD_A(a1, 1, 1) GRPJ(a0, b1, 0, 1)                                                                 \
        LOAD_A(a0, 1, 2) GRPJ(a1, b1, 1, 2)                                                                 \
        LOAD_A(a1, 1, 3) GRPJ(a0, b1, 2, 3)                                                                 \
        GRPJ(a1, b1, 3, 4)
#define KSTEP_LT                                                                                            \
        LOAD_B(b0, 0)                                                                                       \
        LOAD_A(a0, 0, 1)                                                                                    \
        STAGE_NEXT                                                                                          \
        LOAD_A(a1, 0, 2) GRPJ(a0, b0, 1, 1)                                                                 \
        LOAD_A(a0, 0, 3) GRPJ(a1, b0, 2, 2)                                                                 \
        LOAD_A(a1, 1, 1) LOAD_B(b1, 1) GRPJ(a0, b0, 3, 3)                                                   \
        LOAD_A(a0, 1, 2) GRPJ(a1, b1, 1, 1)                                                                 \
        LOAD_A(a1, 1, 3) GRPJ(a0, b1, 2, 2)                                                                 \
        GRPJ(a1, b1, 3, 3)
#endif
// K steps [KT0, KT1) of the pipeline with the K-step body BODY.  COND = 1: inside a triangular diagonal block, 16-row
// groups that are structurally zero for the step are skipped (bit mt of `act`, used by MMA_IF).
#define STAGE_LOOP(KT0, KT1, COND, BODY)                                                                    \
    for (int64_t kt = (KT0); kt < (KT1); ++kt) {                                                            \
        const int buf = (int)(kt & 1);                                                                      \
        const v4f64 w4 = w_nxt;    /* contraction weights of k = 4*lk + ks of this step (B_T) */            \
        const int sbase = (part == 0 && kt == nk - 2) ? 0 : (part == 0 && kt == 4) ? 5 : -1;                \
        (void)sbase;                                                                                        \
        STEP_STAMP(0);                                                                                      \
        const double* As = lds + buf * STAGE_ELEMS;                                                         \
        const double* Bs = As + A_ELEMS;                                                                    \
        int act = 15;                                                                                       \
        if ((COND) == 1) {                                                                                  \
            const int64_t kk = k0 + KSTEP(kt) * BK;                                                         \
            const int64_t r0 = (int64_t)rb * RM + wr * 16;                                                  \
            act = 0;                                                                                        \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                            \
                const bool nz = (g.tri & TRI_LOWER_A) ? (kk <= r0 + mt * 32 + 15) : (kk + 15 >= r0 + mt * 32); \
                act |= nz ? (1 << mt) : 0;                                                                  \
            }                                                                                               \
            act = __builtin_amdgcn_readfirstlane(act);                                                      \
        }                                                                                                   \
        (void)act;                                                                                          \
        BODY                                                                                                \
        STEP_STAMP(2);                                                                                      \
        /* all LDS reads of buf returned (lgkmcnt) and this wavefront's DMA into buf^1 landed (vmcnt) */    \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                         \
        STEP_STAMP(3);                                                                                      \
        __syncthreads();                                                                                    \
        STEP_STAMP(4);                                                                                      \
    }
    if (TRI) {
        // the 8 K steps of the diagonal block (last for a lower-, first for an upper-triangular A) take the skipping
        // body, every other step the branch-free one
        const int64_t nd = nk < RM / BK ? nk : RM / BK;
        const bool diag_first = upper != rev;     // the diagonal block is the lowest k of an upper-, the highest of a lower-
        const int64_t d0 = ((g.tri & (TRI_LOWER_A | TRI_UPPER_A)) && diag_first) ? nd : 0;
        const int64_t d1 = ((g.tri & (TRI_LOWER_A | TRI_UPPER_A)) && !diag_first) ? nk - nd : nk;
        STAGE_LOOP(0, d0, 1, KSTEP_STD(MMA_IF))
        STAMP(2 + 4 * part);
        STAGE_LOOP(d0, d1, 0, KSTEP_STD(MMA_ALL))
        STAMP(3 + 4 * part);
        STAGE_LOOP(d1, nk, 1, KSTEP_STD(MMA_IF))
    } else {
        bool done = false;
        if constexpr (B_T) {
            if (g.sym_out && g.lower_out && rb == cb) {
                // symmetric output, diagonal tile: the 16 x 16 blocks strictly above the diagonal are not computed
                const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // scalar branch: whole loops per wavefront role
                if (wave_u == 1) {      // (wr, wc) = (0, 1)
                    STAGE_LOOP(0, nk, 0, KSTEP_LT)
                } else {
                    STAGE_LOOP(0, nk, 0, KSTEP_LE)
                }
                done = true;
            }
        }
        if (!done) { STAGE_LOOP(0, nk, 0, KSTEP_STD(MMA_ALL)) }
    }
#undef STAGE_LOOP
#undef KSTEP
#undef KOFF
#undef KSTEP_STD
#undef STAGE_NEXT
#undef KSTEP_LE
#undef KSTEP_LT
#undef GRPJ
#undef MMA_IF
#undef MMA_ALL
#undef MMA_DO
#undef LOAD_A
#undef LOAD_B
#undef LOAD_A16
#undef LOAD_B16
#undef MMA16

    // ------------------------------------------------------------------ epilogue
    // Every operand of an epilogue is fetched up front in batches (the row vector `avec` through LDS, staged at the top
    // of the part; column vectors and the Aaux tile by unconditional loads): a conditional load inside the unrolled
    // element loop compiles to one branch + load + s_waitcnt vmcnt(0) PER ELEMENT (64 dependent round trips, ~24 us per
    // epilogue -- the round-1 column-statistics epilogue did exactly that).
    STAMP(4 + 4 * part);      // main loop done

    // accumulator acc[mt][nt][r] of a lane: row = row0 + mt*32 + 4*r; column = col0 + COLOFF(nt).  NN: the lane's columns
    // come in adjacent pairs (nt = 2h, 2h+1 -> col0 + 32h, +1): 16-byte accesses; A B^T keeps one column per 16-lane group
    // (column group 2*nt + wc).
    const int64_t row0 = (int64_t)rb * RM + wr * 16 + lk;
    const int64_t col0 = cb * BN + (B_T ? wc * 16 + li : wc * 64 + 2 * li);
    const int ccol = B_T ? wc * 16 + li : wc * 64 + 2 * li;    // the same inside the tile
#define COLOFF(nt) (B_T ? (nt) * 32 : ((nt) >> 1) * 32 + ((nt) & 1))
    if (EPI == EPI_DA && !B_T) {
        // dA = alpha*acc + avec[i]*gmu[n] - 2*Aaux[i][n]*cgv[n];   optionally rd[row] = sum_n Aaux[row][n] * gmu[n].
        // No LDS, no workgroup barrier (see the column-statistics epilogue): avec comes by broadcast loads.  The Aaux tile
        // is fetched one 16-row group at a time (all four at once would need 128 VGPRs next to the 128 accumulators).
        v2f64 gm[2], cg[2], cs[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            gm[h] = *(const v2f64*)(g.gmu + col0 + 32 * h);
            cg[h] = *(const v2f64*)(g.cgv + col0 + 32 * h);
            cs[h] = (v2f64){1.0, 1.0};
        }
        if (g.bscale) {      // column scaling commutes with A*
#pragma unroll
            for (int h = 0; h < 2; ++h) cs[h] = *(const v2f64*)(g.bscale + col0 + 32 * h);
        }
        // row dots: after the butterfly over li all 16 lanes of a lane group hold the sum of row (mt, r); lane li keeps the
        // one with 4*mt + r == li -- one register for the wavefront's 16 x 4 row sums
        double rdkeep = 0.0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            v2f64 av[2][4];
            double ar[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ar[r] = g.avec[row0 + mt * 32 + 4 * r];
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    av[h][r] = *(const v2f64*)(g.Aaux + (row0 + mt * 32 + 4 * r) * g.ldc + col0 + 32 * h);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double rd = 0.0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const v2f64 a = av[h][r];
                    v2f64 o;
                    o[0] = g.alpha * cs[h][0] * acc[mt][2 * h][r] + ar[r] * gm[h][0] - 2.0 * a[0] * cg[h][0];
                    o[1] = g.alpha * cs[h][1] * acc[mt][2 * h + 1][r] + ar[r] * gm[h][1] - 2.0 * a[1] * cg[h][1];
                    *(v2f64*)(C + (row0 + mt * 32 + 4 * r) * g.ldc + col0 + 32 * h) = o;
                    rd += a[0] * gm[h][0] + a[1] * gm[h][1];
                }
                if (g.rowdot_part) {
                    rd = row16_sum(rd);
                    if (li == 4 * mt + r) rdkeep = rd;
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // one 16-row group at a time: bounded live ranges, no spills
        }
        if (g.rowdot_part && li < 4 * MT)      // lane li: row (mt, r) = (li >> 2, li & 3) of this wavefront's 64-column slice
            g.rowdot_part[((int64_t)cb * 2 + wc) * g.Mr + row0 + (li >> 2) * 32 + 4 * (li & 3)] = rdkeep;
        STAMP(9 + part);
        continue;
    }
    if (EPI == EPI_COLSTATS) {
        // Column statistics FIRST, the C stores last: a register reload from scratch (the epilogue's address arithmetic
        // may spill a value) is a vector-memory operation, and its s_waitcnt vmcnt(0) would wait for every C store issued
        // before it (measured: 17 us per tile at M = 1024 with the stores in front).
        // Partial sums stay PER WAVEFRONT: per lane over its 16 rows, then across the wavefront's 4 lane groups (two
        // shuffles); partial row 2*rb + wr of colsq_part / coldot_part.  No workgroup barrier: right after the main loop a
        // wavefront may sit behind an MFMA burst of the co-resident workgroup, and a barrier here made all four wait for
        // the unluckiest one (3.6 us of a 7 us epilogue, in-kernel stamps).
        double sq[4] = {0.0, 0.0, 0.0, 0.0}, dt[4] = {0.0, 0.0, 0.0, 0.0};
        const bool want_dot = g.coldot_part != nullptr;
        double ar[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ar[mt][r] = 0.0;
        if (want_dot) {      // 16 broadcast loads (L2 hits), all in flight together; no store is outstanding here
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ar[mt][r] = g.avec[row0 + mt * 32 + 4 * r];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = g.alpha * acc[mt][nt][r];
                    sq[nt] += v * v;
                    dt[nt] += ar[mt][r] * v;
                }
        if (part == 0) STAMP(11);      // sums done
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            sq[nt] = rows4_sum(sq[nt]);
            if (want_dot) dt[nt] = rows4_sum(dt[nt]);
        }
        if (lk == 0) {
            const int64_t prow = ((int64_t)rb * 2 + wr) * g.Nc + col0;
            if (B_T) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    g.colsq_part[prow + COLOFF(nt)] = sq[nt];
                    if (want_dot) g.coldot_part[prow + COLOFF(nt)] = dt[nt];
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    *(v2f64*)(g.colsq_part + prow + 32 * h) = (v2f64){sq[2 * h], sq[2 * h + 1]};
                    if (want_dot) *(v2f64*)(g.coldot_part + prow + 32 * h) = (v2f64){dt[2 * h], dt[2 * h + 1]};
                }
            }
        }
        if (part == 0) STAMP(12);      // partials written
        __builtin_amdgcn_sched_barrier(0);
    }
    // One running per-lane pointer walks the tile's rows (stride 4 rows inside a 16-row group, 32 rows between groups):
    // precomputed row addresses cost 32 VGPRs on top of the 128 accumulators and spill -- and every reload from scratch
    // between two stores is an s_waitcnt vmcnt(0), i.e. a full drain of the stores issued so far (~10 us per tile).
#define STORE_TILE(ST, ST2, ACCUM)                                                                          \
    {                                                                                                       \
        double* cp = C + row0 * g.ldc + col0;                                                               \
        const int64_t step4 = 4 * g.ldc, step_mt = 32 * g.ldc - 16 * g.ldc;                                 \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                \
                if (B_T) {                                                                                  \
                    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                     \
                        double v = g.alpha * acc[mt][nt][r];                                                \
                        if (ACCUM) v += cp[COLOFF(nt)];                                                     \
                        ST(v, cp + COLOFF(nt));                                                             \
                    }                                                                                       \
                } else {                                                                                    \
                    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                        \
                        v2f64 v;                                                                            \
                        v[0] = g.alpha * acc[mt][2 * h][r];                                                 \
                        v[1] = g.alpha * acc[mt][2 * h + 1][r];                                             \
                        if (ACCUM) v += *(const v2f64*)(cp + 32 * h);                                       \
                        ST2(v, (v2f64*)(cp + 32 * h));                                                      \
                    }                                                                                       \
                }                                                                                           \
                cp += step4;                                                                                \
            }                                                                                               \
            cp += step_mt;                                                                                  \
        }                                                                                                   \
    }
#define ST_PLAIN(v, p) (*(p) = (v))
#define ST_STREAM(v, p) __builtin_nontemporal_store((v), (p))
    if (g.accumulate) { STORE_TILE(ST_PLAIN, ST_PLAIN, 1) }
    else if (g.stream_out) { STORE_TILE(ST_STREAM, ST_STREAM, 0) }
    else { STORE_TILE(ST_PLAIN, ST_PLAIN, 0) }
#undef STORE_TILE
    if (part == 0) STAMP(13);          // C stores issued

#undef COLOFF
    STAMP(9 + part);          // epilogue issued
  }   // parts
  STEP_STAMPS_FLUSH;
}

static bool small_panel_ok(const GemmArgs& g, bool B_T, int splitk);
template <int EPI> static int launch_small_panel(const GemmArgs& g, hipStream_t s);
static bool small_gemm_ok(const GemmArgs& g, bool B_T);
static int launch_small_gemm(const GemmArgs& g, bool B_T, hipStream_t s);

// 64-row tiles serve the plain M x N' panel products only: A B form, no k-slicing, no batching, no lower-triangular output;
// anything else runs on 128-row tiles.  GemmArgs::rm = 0 picks the height from the shape (tools/tile_sweep.py, MI355X,
// triangular A, colstats / dA epilogues, ms):
//     M x N'        128 unpaired  128 paired  64 unpaired  64 paired
//     512 x 8192       0.070        0.086       0.074       0.052      <- 256 tiles of 128 rows: one per CU, the longest one
//     512 x 16384      0.124        0.090       0.119       0.084         walks 32 dependent K steps; paired they fill half
//     512 x 32768      0.243        0.158       0.230       0.165         the chip.  64-row tiles: twice the tiles, pairs of
//     512 x 65536      0.480        0.328       0.421       0.326         equal work on every CU
//     1024 x 8192      0.247        0.158       0.209       0.154
//     1024 x 65536     1.547        1.106       1.346       1.126
//     256 x 8192       0.036        0.052       0.025       0.030
//     768 x 16384      0.272        0.213       0.244       0.168
// Rule: 128-row tiles (paired) once their pairs fill two workgroups per CU (>= 512); below that 64-row tiles, paired when the
// pairs still give every CU one workgroup (>= 256), else unpaired.
static int tile_rows(const GemmArgs& g, bool B_T, int splitk) {
    if (B_T || splitk > 1 || g.batched || g.lower_out || g.zlayers > 1) return BM;
    if (g.rm == 64 || g.rm == BM) return g.rm;
    if (!(g.tri & (TRI_LOWER_A | TRI_UPPER_A))) return BM;
    const int64_t pairs128 = (int64_t)((g.Mr / BM + 1) / 2) * (g.Nc / BN);
    return pairs128 >= 512 ? BM : 64;
}

int launch_gemm(const GemmArgs& g0, bool B_T, int splitk, hipStream_t s) {
    GemmArgs g = g0;
    if (g.Mr % BM || g.Nc % BN || g.Kd % BK) return MOBOCMF_BAD_ARG;
    // block activity is honoured where the kernels implement it, and ignored (dense product: the same result) elsewhere
    if (B_T || g.batched || splitk > 1) g.colact = nullptr;
    if (g.kact) {
        const int sk_min = (g.lower_out && g.splitk_diag > 0 && g.splitk_diag < splitk) ? g.splitk_diag : splitk;
        const int64_t nb = g.Kd / 128;
        if (!B_T || g.batched || splitk <= 1 || g.zlayers > 1 || g.Kd % 128 || (g.tri & (TRI_LOWER_B | TRI_UPPER_B)) ||
            (nb + sk_min - 1) / sk_min + 2 > KL_MAX)
            g.kact = nullptr;
    }
    if (!B_T && ((g.ldc & 1) || ((uintptr_t)g.C & 15))) return MOBOCMF_BAD_ARG;   // 16-byte epilogue accesses (A B form)
    if (small_panel_ok(g, B_T, splitk)) {
        if (g.epi == EPI_COLSTATS) return launch_small_panel<EPI_COLSTATS>(g, s);
        if (g.epi == EPI_DA) return launch_small_panel<EPI_DA>(g, s);
        return launch_small_panel<EPI_STORE>(g, s);
    }
    if (B_T && splitk <= 1 && small_gemm_ok(g, true)) return launch_small_gemm(g, true, s);   // small weighted syrk
    // batched small products (the 128 / 256-wide merges of the blocked triangular inverse): a handful of tiles each, pure
    // latency on the 128 x 128 x 16 pipeline (19-23 us per launch at M = 512), ~7 us on the small-operand kernel
    if (!B_T && g.batched > 1 && splitk <= 1 && small_gemm_ok(g, false)) return launch_small_gemm(g, false, s);
    const int rm = tile_rows(g, B_T, splitk);
    int nrb = g.Mr / rm;
    int64_t ncb = g.Nc / BN;
    dim3 grid;
    int pair = 0;
    if (g.batched) {
        grid = dim3((unsigned)(nrb * ncb), 1, (unsigned)(g.batched * (g.zlayers > 1 ? g.zlayers : 1)));
        splitk = 1;
    } else if (splitk > 1) {
        int64_t ntile = g.lower_out ? (int64_t)nrb * (nrb + 1) / 2 : (int64_t)nrb * ncb;
        if (g.lower_out && g.splitk_diag > 0) ntile = 0;      // two classes: counted below
        const int64_t nblk = ntile ? ntile * splitk : (int64_t)nrb * (nrb - 1) / 2 * splitk + (int64_t)nrb * g.splitk_diag;
        grid = dim3((unsigned)(g.dual_flag ? 2 * nblk : nblk), 1, (unsigned)(g.zlayers > 1 ? g.zlayers : 1));
    } else {
        // pairing balances the work per workgroup; it only pays when the pairs still fill the chip (table above), otherwise
        // the longest single tile is the critical path and pairing lengthens it
        const bool can_pair = (g.tri & (TRI_LOWER_A | TRI_UPPER_A)) && !g.lower_out && nrb > 1;
        if (rm == 64) pair = (can_pair && (int64_t)(nrb / 2) * ncb >= 256) ? 1 : 0;      // pairs still cover every CU
        else pair = (can_pair && (int64_t)nrb * ncb > 512) ? 1 : 0;
        if (g.pair_mode == 1) pair = 0;
        if (g.pair_mode == 2 && (g.tri & (TRI_LOWER_A | TRI_UPPER_A)) && !g.lower_out && nrb > 1) pair = 1;
        grid = dim3((unsigned)((pair ? (nrb + 1) / 2 : nrb) * ncb), 1, 1);
        splitk = 1;
    }
    // TRI instantiation: per-step skipping of structurally-zero row groups.  A dense plain-store product takes it too
    // (its diagonal-block loops are empty then): hipcc's register allocation of the <false, false, EPI_STORE>
    // instantiation spills loop-invariant LDS addresses and reloads them -- a scratch round trip -- in every K step
    // (dense 512 x 65536 x 512: 0.62 ms against 0.53 ms through this instantiation)
    const bool tri = (g.tri & (TRI_LOWER_A | TRI_UPPER_A)) != 0 || g.epi == EPI_STORE;
#ifdef GEMM_LDS_PAD64
    // experiment (tools/build_variant.sh): dynamic LDS on the 64-row launches caps the workgroups a CU takes (48 KB static:
    // three fit in 160 KB; + 32 KB -> two, + 64 KB -> one), to see how the dispatcher spreads a grid smaller than the slots
    const size_t dyn64 = GEMM_LDS_PAD64;
#define LAUNCH(BT, TR, EP, RM_)                                                                                         \
    do {                                                                                                                \
        const size_t dyn = (RM_) == 64 ? dyn64 : 0;                                                                     \
        if (dyn) (void)hipFuncSetAttribute((const void*)gemm_f64_kernel<BT, TR, EP, RM_>,                               \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);                       \
        hipLaunchKernelGGL((gemm_f64_kernel<BT, TR, EP, RM_>), grid, dim3(256), dyn, s, g, nrb, ncb, splitk, pair);      \
    } while (0)
#else
#define LAUNCH(BT, TR, EP, RM_) hipLaunchKernelGGL((gemm_f64_kernel<BT, TR, EP, RM_>), grid, dim3(256), 0, s, g, nrb, ncb, splitk, pair)
#endif
    if (B_T) {
        if (g.epi != EPI_STORE) return MOBOCMF_BAD_ARG;
        LAUNCH(true, false, EPI_STORE, 128);
    } else if (rm == 64) {      // 64-row tiles: always the TRI instantiation (a dense operand has empty diagonal-block loops)
        if (g.epi == EPI_COLSTATS) LAUNCH(false, true, EPI_COLSTATS, 64);
        else if (g.epi == EPI_DA) LAUNCH(false, true, EPI_DA, 64);
        else LAUNCH(false, true, EPI_STORE, 64);
    } else if (tri) {
        if (g.epi == EPI_COLSTATS) LAUNCH(false, true, EPI_COLSTATS, 128);
        else if (g.epi == EPI_DA) LAUNCH(false, true, EPI_DA, 128);
        else LAUNCH(false, true, EPI_STORE, 128);
    } else {
        if (g.epi == EPI_COLSTATS) LAUNCH(false, false, EPI_COLSTATS, 128);
        else if (g.epi == EPI_DA) LAUNCH(false, false, EPI_DA, 128);
        else LAUNCH(false, false, EPI_STORE, 128);
    }
#undef LAUNCH
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}

// partial rows (of Nc entries each) an EPI_COLSTATS launch writes to colsq_part / coldot_part: two per row block of the tile
// height the launch will use (the small-panel kernel: two per 128 rows, the second one zero)
int gemm_colstat_rows(const GemmArgs& g) {
    if (small_panel_ok(g, false, 1)) return 2 * (g.Mr / BM);
    return 2 * (g.Mr / tile_rows(g, false, 1));
}

// out[i][j] (+)= sum_z slabs[z][i][j]   (rows x cols, slabs dense with ld = cols); lower_only: tiles above the
// diagonal are neither read nor written, elements above the diagonal inside diagonal tiles are zeroed if `tril`
__global__ void reduce_slabs_kernel(const double* slabs, int64_t slab_stride, int nslab, double* out, int64_t ld,
                                    int rows, int64_t cols, double scale, int lower_only, int tril, int accumulate,
                                    int64_t zs_slabs, int64_t zs_out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)rows * cols) return;
    slabs += blockIdx.z * zs_slabs;      // layer batching
    out += blockIdx.z * zs_out;
    int i = (int)(idx / cols);
    int64_t j = idx % cols;
    if (lower_only && (j / TILE) > (i / TILE)) {
        if (tril && !accumulate) out[(int64_t)i * ld + j] = 0.0;
        return;
    }
    double v = 0.0;
    const double* p = slabs + (int64_t)i * cols + j;
    for (int z = 0; z < nslab; ++z) v += p[z * slab_stride];
    v *= scale;
    if (tril && j > i) v = 0.0;
    if (accumulate) v += out[(int64_t)i * ld + j];
    out[(int64_t)i * ld + j] = v;
}

int launch_reduce_slabs(const double* slabs, int64_t slab_stride, int nslab, double* out, int64_t ld, int Mr,
                        double scale, int lower_only, int accumulate, hipStream_t s) {
    int64_t n = (int64_t)Mr * Mr;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slabs, slab_stride,
                       nslab, out, ld, Mr, (int64_t)Mr, scale, lower_only, lower_only, accumulate, (int64_t)0, (int64_t)0);
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}

// Hand-over to the tiled MFMA kernels (largest dimension), from tools/size_sweep.py + the C3 bench: the 16x16-block
// product wins up to 384 (a 512^3 product is LDS-bound at ~27 us and costs C3 3 %), the whole-block panel kernel up to
// K = 512 when the launch stays within 512 workgroups (M = N = 300: 1.98 -> 1.54 ms per step, 450: 2.47 -> 2.2).
static int small_gemm_limit() { return tune().small_gemm_max; }
static int small_panel_limit() { return tune().small_panel_max; }
// ---------------------------------------------------------------------------------- small operands
// All dimensions <= 256 (the M x M chain of a surrogate with M <= 256 -- the sizes the reference's own BO runs live at):
// the 128x128x16 MFMA pipeline is pure latency there (one to four workgroups walking 8-16 dependent K steps, ~20 us a
// product, ~100 products per training step).  Here every workgroup owns a 16 x 16 block of C (one element per thread),
// streams 16 x 16 operand blocks through LDS and applies the structural zeros of triangular operands on load, so the
// result never depends on what lies in the unused triangle.  ~4 us a product.
template <bool B_T>
__device__ __forceinline__ void small_gemm_body(GemmArgs g, const int bz) {
    // K is walked in chunks of 128: all 16 loads of a thread for a chunk (8 of A, 8 of B) are issued together, i.e. one
    // global-load latency per chunk instead of one per 16-wide K step (8 dependent steps made a 128^3 product ~7 us)
    constexpr int KC = 128;
    __shared__ double As[16][KC + 1], Bs[KC][17];
    if (g.skip_if_zero && *g.skip_if_zero == 0) return;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t r0 = (int64_t)blockIdx.y * 16, c0 = (int64_t)blockIdx.x * 16;
    const bool sym_full = B_T && g.sym_full;
    if (sym_full) {
        // the weighted syrk of a small layer backward, H = A diag(w) A^T (and its twin with the clamped-column weights,
        // bz = 1), written as the full symmetric matrix by this launch: blocks on or below the block diagonal only
        if (c0 > r0) return;
        if (bz == 1) { g.bscale = g.bscale2; g.C = g.C2; }
    } else {
    if (g.lower_out && c0 / BM > r0 / BM) return;        // same contract as the tiled kernel: lower 128-tiles only
    }
    if (!sym_full) {   // bz = batch * layers + layer, as in the tiled kernel (batched merges of the triangular inverse, layer
        // batching of the chains, or both)
        const int nzl = g.zlayers > 1 ? g.zlayers : 1;
        const int zl = bz % nzl, zb = bz / nzl;
        g.A += zb * g.strideA + zl * g.zsA;
        g.B += zb * g.strideB + zl * g.zsB;
        g.C += zb * g.strideC + zl * g.zsC;
    }
    int64_t k0 = 0, k1 = g.Kd;
    if (g.tri & TRI_LOWER_A) k1 = k1 < r0 + 16 ? k1 : r0 + 16;
    if (g.tri & TRI_UPPER_A) k0 = k0 > r0 ? k0 : r0;
    if (g.tri & TRI_LOWER_B) k0 = k0 > c0 ? k0 : c0;
    if (g.tri & TRI_UPPER_B) k1 = k1 < c0 + 16 ? k1 : c0 + 16;
    if (g.Kreal > 0) {       // the contraction index >= Kreal only meets zero padding of one of the operands
        const int64_t kend = (g.Kreal + 15) & ~(int64_t)15;
        if (k1 > kend) k1 = kend;
    }
    double acc = 0.0;
    for (int64_t kc = k0; kc < k1; kc += KC) {
        const int kn = (int)(k1 - kc < KC ? k1 - kc : KC);       // multiple of 16
        double va[KC / 16], vb[KC / 16];
#pragma unroll
        for (int u = 0; u < KC / 16; ++u) {
            va[u] = vb[u] = 0.0;
            if (u * 16 < kn) {
                const int64_t i = r0 + ty, kk = kc + u * 16 + tx;                 // A[r0 + ty][kk]
                double v = g.A[i * g.lda + kk];
                if (((g.tri & TRI_LOWER_A) && kk > i) || ((g.tri & TRI_UPPER_A) && kk < i)) v = 0.0;
                va[u] = v;
                if (B_T) {                                                        // B[c0 + ty][kk]  ->  Bs[k][col]
                    const int64_t j = c0 + ty;
                    double w = g.B[j * g.ldb + kk];
                    if (((g.tri & TRI_LOWER_B) && kk < j) || ((g.tri & TRI_UPPER_B) && kk > j)) w = 0.0;
                    if (g.bscale) w *= g.bscale[kk];      // contraction weights (the weighted syrk A diag(w) A^T)
                    vb[u] = w;
                } else {                                                          // B[kc + 16u + ty][c0 + tx]
                    const int64_t kb = kc + u * 16 + ty, j = c0 + tx;
                    double w = g.B[kb * g.ldb + j];
                    if (((g.tri & TRI_LOWER_B) && kb < j) || ((g.tri & TRI_UPPER_B) && kb > j)) w = 0.0;
                    vb[u] = w;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < KC / 16; ++u) {
            As[ty][u * 16 + tx] = va[u];
            if (B_T) Bs[u * 16 + tx][ty] = vb[u];
            else Bs[u * 16 + ty][tx] = vb[u];
        }
        __syncthreads();
        for (int q = 0; q < kn; q += 4) {
            acc += As[ty][q] * Bs[q][tx];
            acc += As[ty][q + 1] * Bs[q + 1][tx];
            acc += As[ty][q + 2] * Bs[q + 2][tx];
            acc += As[ty][q + 3] * Bs[q + 3][tx];
        }
        __syncthreads();
    }
    double* c = g.C + (r0 + ty) * g.ldc + c0 + tx;
    double v = g.alpha * acc;
    if (sym_full) {
        // element (i, j) and its mirror image get the SAME number (the one computed for the lower triangle): the result is
        // exactly symmetric, as the slab reduction of the tiled path makes it (the chain backward reads X = H U as (U^T H)^T)
        if (c0 < r0 || tx <= ty) {
            *c = v;
            g.C[(c0 + tx) * g.ldc + r0 + ty] = v;
        }
        return;
    }
    if (g.accumulate) v += *c;
    *c = v;
}

template <bool B_T>
__global__ __launch_bounds__(256) void small_gemm_kernel(GemmArgs g) {
    small_gemm_body<B_T>(g, (int)blockIdx.z);
}
// two independent A B products of the same shape and layer batching in one launch: even blockIdx.z -> g0, odd -> g1
__global__ __launch_bounds__(256) void small_gemm2_kernel(GemmArgs g0, GemmArgs g1) {
    if (blockIdx.z & 1) small_gemm_body<false>(g1, (int)(blockIdx.z >> 1));
    else small_gemm_body<false>(g0, (int)(blockIdx.z >> 1));
}

// M x N' panel products of a small problem (K = Mp <= 256, at most two workgroups per CU in all): one workgroup per
// (128-row block, 16 columns), 8 rows per thread, same epilogues and partial-sum layout as the tiled kernel.  The tiled
// kernel -- and a first version of this one that walked K in 16-wide steps -- pays one global-load latency per K step
// (8 dependent steps = ~28 us for a 128 x 128 x 128 product).  Here a workgroup pulls its whole 128 x 128 block of A and
// 128 x 16 block of B into LDS at once (149 KB of the CU's 160 KB: every load of the block is in flight together),
// then multiplies out of LDS.
#define SP_LDT 132     // leading dimension of the TRANSPOSED A image [k][row] (16-byte aligned rows of 4, padded)
#define SP_LDB 18
#define SP_LDS_BYTES ((BM * SP_LDT + BM * SP_LDB) * 8)
// Thread tile 4 rows x 2 columns: per k one 32-byte run of the transposed A image (two ds_read_b128) and one 16-byte pair of
// B feed eight FMAs -- three LDS instructions per eight FMAs where one output column per thread needed nine (the loop is
// LDS-issue bound: 128 x 16 outputs per workgroup, K = 128, everything resident in LDS).
template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void small_panel_kernel(GemmArgs g) {
    extern __shared__ double sp_smem[];
    double* AsT = sp_smem;                   // [128 k][SP_LDT rows]
    double* Bs = sp_smem + BM * SP_LDT;      // [128 k][SP_LDB cols]
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;       // staging map: 16 k x 16 rows
    const int tc = threadIdx.x & 7, tr = threadIdx.x >> 3;        // compute map: columns 2 tc, 2 tc + 1; rows 4 tr .. 4 tr + 3
    const int rb = blockIdx.y;
    const int64_t r0 = (int64_t)rb * BM, c0 = (int64_t)blockIdx.x * 16;
    if (g.colact && g.colact[c0 >> 7] == 0) {      // inactive column block (GemmArgs.colact): not computed
        if (EPI == EPI_DA && g.rowdot_part && threadIdx.x < BM) g.rowdot_part[(int64_t)blockIdx.x * g.Mr + r0 + threadIdx.x] = 0.0;
        return;
    }
    int64_t k0 = 0, k1 = g.Kd;               // multiples of 128 (Mr, Kd are)
    if (g.tri & TRI_LOWER_A) k1 = k1 < r0 + BM ? k1 : r0 + BM;
    if (g.tri & TRI_UPPER_A) k0 = r0;
    // B's rows >= Kreal are zero padding (a 16-point problem padded to 128): the contraction stops at the real size
    const int64_t kend = g.Kreal > 0 ? ((g.Kreal + 15) & ~(int64_t)15) : g.Kd;
    if (k1 > kend) k1 = kend;
    double acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = 0.0;
    for (int64_t kc = k0; kc < k1; kc += BM) {
        const int kn = (int)(k1 - kc < BM ? k1 - kc : BM);      // multiple of 16: K steps of this chunk that matter
        double va[8][8], vb[8];               // every load of the block in flight before the first use
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t row = r0 + ty + 16 * i;
#pragma unroll
            for (int c = 0; c < 8; ++c) va[i][c] = c * 16 < kn ? g.A[row * g.lda + kc + c * 16 + tx] : 0.0;
            vb[i] = 16 * i < kn ? g.B[(kc + ty + 16 * i) * g.ldb + c0 + tx] : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t row = r0 + ty + 16 * i;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int64_t kk = kc + c * 16 + tx;
                double v = va[i][c];
                if (((g.tri & TRI_LOWER_A) && kk > row) || ((g.tri & TRI_UPPER_A) && kk < row)) v = 0.0;
                AsT[(c * 16 + tx) * SP_LDT + ty + 16 * i] = v;
            }
            Bs[(ty + 16 * i) * SP_LDB + tx] = vb[i];
        }
        __syncthreads();
        const double* ap = AsT + 4 * tr;
        const double* bp = Bs + 2 * tc;
#pragma unroll 8
        for (int q = 0; q < kn; ++q) {
            const v2f64 a01 = *(const v2f64*)(ap + q * SP_LDT), a23 = *(const v2f64*)(ap + q * SP_LDT + 2);
            const v2f64 b = *(const v2f64*)(bp + q * SP_LDB);
            acc[0][0] += a01[0] * b[0]; acc[0][1] += a01[0] * b[1];
            acc[1][0] += a01[1] * b[0]; acc[1][1] += a01[1] * b[1];
            acc[2][0] += a23[0] * b[0]; acc[2][1] += a23[0] * b[1];
            acc[3][0] += a23[1] * b[0]; acc[3][1] += a23[1] * b[1];
        }
        __syncthreads();
    }
    const int64_t col = c0 + 2 * tc, row0 = r0 + 4 * tr;
    if (EPI == EPI_DA) {
        const v2f64 gm = *(const v2f64*)(g.gmu + col), cg = *(const v2f64*)(g.cgv + col);
        v2f64 cs = (v2f64){g.alpha, g.alpha};
        if (g.bscale) { const v2f64 bs = *(const v2f64*)(g.bscale + col); cs[0] *= bs[0]; cs[1] *= bs[1]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = row0 + i;
            const v2f64 av = *(const v2f64*)(g.Aaux + row * g.ldc + col);
            const double ar = g.avec[row];
            v2f64 o;
            o[0] = cs[0] * acc[i][0] + ar * gm[0] - 2.0 * av[0] * cg[0];
            o[1] = cs[1] * acc[i][1] + ar * gm[1] - 2.0 * av[1] * cg[1];
            *(v2f64*)(g.C + row * g.ldc + col) = o;
            if (g.rowdot_part) {       // the 8 lanes tc of a row hold its 16 columns
                double v = av[0] * gm[0] + av[1] * gm[1];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                if (tc == 0) g.rowdot_part[(int64_t)blockIdx.x * g.Mr + row] = v;
            }
        }
        return;
    }
    double sq[2] = {0.0, 0.0}, dt[2] = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = row0 + i;
        v2f64 v = (v2f64){g.alpha * acc[i][0], g.alpha * acc[i][1]};
        if (g.accumulate) { const v2f64 old = *(const v2f64*)(g.C + row * g.ldc + col); v[0] += old[0]; v[1] += old[1]; }
        *(v2f64*)(g.C + row * g.ldc + col) = v;
        if (EPI == EPI_COLSTATS) {
            sq[0] += v[0] * v[0]; sq[1] += v[1] * v[1];
            if (g.coldot_part) { const double ar = g.avec[row]; dt[0] += ar * v[0]; dt[1] += ar * v[1]; }
        }
    }
    if (EPI == EPI_COLSTATS) {
        // column sums: over the 8 row groups of a wavefront (lane bits 3..5), then over the 4 wavefronts through LDS
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            sq[b] += __shfl_xor(sq[b], 8); sq[b] += __shfl_xor(sq[b], 16); sq[b] += __shfl_xor(sq[b], 32);
            dt[b] += __shfl_xor(dt[b], 8); dt[b] += __shfl_xor(dt[b], 16); dt[b] += __shfl_xor(dt[b], 32);
        }
        double* red = sp_smem;               // [2][4 waves][16 cols]; the operand blocks are dead after the last barrier
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane < 8) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                red[(0 * 4 + wave) * 16 + 2 * tc + b] = sq[b];
                red[(1 * 4 + wave) * 16 + 2 * tc + b] = dt[b];
            }
        }
        __syncthreads();
        if (threadIdx.x < 16) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                s0 += red[(0 * 4 + w) * 16 + threadIdx.x];
                s1 += red[(1 * 4 + w) * 16 + threadIdx.x];
            }
            const int64_t cc = c0 + threadIdx.x;
            // same partial layout as the tiled kernel (two rows per 128-row block): the second one is zero here
            g.colsq_part[(int64_t)(2 * rb) * g.Nc + cc] = s0;
            g.colsq_part[(int64_t)(2 * rb + 1) * g.Nc + cc] = 0.0;
            if (g.coldot_part) {
                g.coldot_part[(int64_t)(2 * rb) * g.Nc + cc] = s1;
                g.coldot_part[(int64_t)(2 * rb + 1) * g.Nc + cc] = 0.0;
            }
        }
    }
}

template <int EPI>
static int launch_small_panel(const GemmArgs& g, hipStream_t s) {
    // the dynamic-LDS attribute is per device: one bit per device ordinal (first call per device happens in an eager
    // warm-up, never under stream capture); setting it twice from two threads is harmless, so a relaxed fetch_or is enough
    static std::atomic<uint64_t> configured{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MOBOCMF_HIP_ERROR;
    if (!(configured.load(std::memory_order_acquire) & (1ull << dev))) {
        if (hipFuncSetAttribute((const void*)small_panel_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                SP_LDS_BYTES) != hipSuccess)
            return MOBOCMF_HIP_ERROR;
        configured.fetch_or(1ull << dev, std::memory_order_release);
    }
    const dim3 grid((unsigned)(g.Nc / 16), (unsigned)(g.Mr / BM));
    hipLaunchKernelGGL(small_panel_kernel<EPI>, grid, dim3(256), SP_LDS_BYTES, s, g);
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}

static bool small_panel_ok(const GemmArgs& g, bool B_T, int splitk) {
    return !B_T && splitk <= 1 && !g.batched && !g.skip_if_zero && !g.lower_out && !(g.tri & (TRI_LOWER_B | TRI_UPPER_B)) &&
           g.Kd <= small_panel_limit() && g.Mr <= small_panel_limit() && g.Mr % BM == 0 && g.Nc % 16 == 0 && g.Kd % BM == 0 &&
           (g.Nc / 16) * (g.Mr / BM) <= 512 && (g.epi == EPI_DA || !g.bscale);
}

int gemm_rowdot_parts(const GemmArgs& g) {
    return small_panel_ok(g, false, 1) ? (int)(g.Nc / 16) : (int)(2 * (g.Nc / BN));
}

static bool small_gemm_ok(const GemmArgs& g, bool B_T) {
    const int L = small_gemm_limit();
    if (g.Mr > L || g.Nc > L || g.Kd > L) return false;
    return g.epi == EPI_STORE && (B_T || (!g.bscale && !g.skip_if_zero)) && g.Mr <= 512 && g.Nc <= 512 &&
           g.Kd <= 512 && g.Mr % 16 == 0 && g.Nc % 16 == 0 && g.Kd % 16 == 0;
}

static int launch_small_gemm(const GemmArgs& g, bool B_T, hipStream_t s) {
    if (B_T && g.sym_full && (g.batched > 1 || g.zlayers > 1 || g.Mr != g.Nc || g.accumulate)) return MOBOCMF_BAD_ARG;
    const dim3 grid((unsigned)(g.Nc / 16), (unsigned)(g.Mr / 16),
                    (B_T && g.sym_full) ? (g.C2 ? 2u : 1u)
                                        : (unsigned)((g.batched > 1 ? g.batched : 1) * (g.zlayers > 1 ? g.zlayers : 1)));
    if (B_T) hipLaunchKernelGGL(small_gemm_kernel<true>, grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL(small_gemm_kernel<false>, grid, dim3(256), 0, s, g);
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}

// slabs a k-sliced A B^T product will write: 1 when the small-operand kernel takes it whole
int gemm_nt_slabs(const GemmArgs& g, int splitk) { return small_gemm_ok(g, true) ? 1 : splitk; }
// true: an A B^T product of this shape runs on the small-operand kernel (which can write a symmetric result in full, sym_full)
bool gemm_nt_is_small(const GemmArgs& g) { return small_gemm_ok(g, true); }

// ---------------------------------------------------------------------------------- mid-size operands (the M x M chain)
// M x M x M products with 384 < M <= 1024 (the chain of a C3 / C5 surrogate: U = L^-1 L_S, the panels of the triangular inverse,
// eight products of the chain backward).  On the 128 x 128 pipeline they are 10-16 tiles: to occupy the chip they are k-sliced
// 8-16 ways into slabs -- one or two K steps per workgroup between a prologue and a 128 KB slab store, then a reduction
// launch: 19 + 7 us for 4 us of MFMA work, 64 MB of slab traffic, and 320-512 workgroups that the other streams' panel
// kernels share the chip with.  Here: 64 x 64 tiles (four wavefronts of 32 x 32), the whole contraction in one workgroup,
// ONE launch and no slabs -- 64 workgroups per layer at M = 512.  Operands are cache-resident, so plain global loads
// (two stages ahead, in registers) replace the LDS-DMA pipeline; the four k-groups of a stage are software-pipelined
// (fragments of group G + 1 read while the 32 MFMAs of group G issue).  Triangular operands bound the k range per 64-block;
// inside a diagonal block the stored zeros of the unused triangle do the rest (all chain operands hold them).  Fragment /
// accumulator lane maps as in gemm_f64_kernel.
// Three forms, same lane maps (mobocmf_set_mid_gemm_waves):
//   32 (default)  32 x 64 tiles, four wavefronts of 16 x 32, 64-k stages: 19.5 us per product at M = 512, 38 at 1024
//    8            64 x 64 tiles, eight wavefronts of 32 x 16, 32-k stages: 27 / 51 us
//    4            64 x 64 tiles, four wavefronts of 32 x 32, 32-k stages, fragments software-pipelined: 29 / 55 us
// (k-sliced 128 x 128 pipeline + slab reduction: 19 + 7 us at M = 512, 45-80 + 16 at 1024.)  What a stage costs beside
// its MFMAs at one workgroup per CU -- two barriers, the LDS hand-over, the first fragment reads: ~0.5-0.7 us, measured
// with the MFMAs / the fragment reads compiled out -- does not overlap with them (one wavefront per SIMD executes in
// order; eight wavefronts or a deeper prefetch changed little), so the form with the fewest stages per k and the most
// workgroups wins.  tools/mid_k_sweep.py.
#define MD_BM 64
#define MD_BN 64
#define MD_BK 32
#define MD_LDA (MD_BK + 2)      // k contiguous; 272-byte rows: 16-byte aligned, off the 256-byte bank period
#define MD_LDB (MD_BN + 2)      // columns contiguous (A B form)
template <bool B_T>
__device__ __forceinline__ void gemm_mid_body(const GemmArgs& g, int nrb, int ncb) {
    __shared__ __attribute__((aligned(16))) double As[MD_BM * MD_LDA];
    __shared__ __attribute__((aligned(16))) double Bs[B_T ? MD_BN * MD_LDA : MD_BK * MD_LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 15, lk = lane >> 4;
    // row blocks from the bottom: with a lower-triangular A (or lower_out) the long tiles start first
    const int rb = nrb - 1 - (int)(blockIdx.x / ncb), cb = (int)(blockIdx.x % ncb);
    if (g.lower_out && cb > (rb | 1)) return;      // the 128 x 128 lower tiles, as the tiled kernel writes them
    const int z = blockIdx.z;
    const double* A = g.A + (g.zlayers > 1 ? z * g.zsA : 0);
    const double* B = g.B + (g.zlayers > 1 ? z * g.zsB : 0);
    double* C = g.C + (g.zlayers > 1 ? z * g.zsC : 0);
    int64_t k0 = 0, k1 = g.Kd;
    if (g.Kreal > 0) { const int64_t ke = (g.Kreal + MD_BK - 1) / MD_BK * MD_BK; if (k1 > ke) k1 = ke; }
    if (g.tri & TRI_LOWER_A) { const int64_t e = (int64_t)(rb + 1) * MD_BM; if (k1 > e) k1 = e; }
    if (g.tri & TRI_UPPER_A) { const int64_t b = (int64_t)rb * MD_BM; if (k0 < b) k0 = b; }
    if (g.tri & TRI_LOWER_B) { const int64_t b = (int64_t)cb * MD_BN; if (k0 < b) k0 = b; }
    if (g.tri & TRI_UPPER_B) { const int64_t e = (int64_t)(cb + 1) * MD_BN; if (k1 > e) k1 = e; }
    const int nst = k1 > k0 ? (int)((k1 - k0) / MD_BK) : 0;

    // staging maps: A (and B^T) image 64 rows x 32 k -- thread = (row t/4, 8 k); B image 32 k x 64 columns -- (k t/8, 8 columns)
    const int ar = tid >> 2, ak = (tid & 3) * 8;
    const int bk = tid >> 3, bc = (tid & 7) * 8;
    const double* Ag = A + ((int64_t)rb * MD_BM + ar) * g.lda + ak;
    const double* Bg = B_T ? B + ((int64_t)cb * MD_BN + ar) * g.ldb + ak : B + (int64_t)bk * g.ldb + (int64_t)cb * MD_BN + bc;
    // two register sets: stages st + 1 and st + 2 are in flight while stage st multiplies (one workgroup per CU, one wavefront
    // per SIMD: nothing else hides the L2 round trip, ~1.3 us against 0.85 us of MFMAs per stage)
    v2f64 ra0[4], rb0[4], ra1[4], rb1[4];
#define MD_FETCH(ST, RA, RB)                                                                                \
    {                                                                                                       \
        const int64_t k = k0 + (int64_t)(ST) * MD_BK;                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) RA[i] = *(const v2f64*)(Ag + k + 2 * i);             \
        if (B_T) {                                                                                          \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) RB[i] = *(const v2f64*)(Bg + k + 2 * i);         \
        } else {                                                                                            \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) RB[i] = *(const v2f64*)(Bg + k * g.ldb + 2 * i); \
        }                                                                                                   \
    }
    v4f64 acc[2][2];      // [mt][nt][r]: row = wr*16 + mt*32 + 4r + lk, column = wc*32 + 2 li + nt
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
    if (nst > 0) {
        MD_FETCH(0, ra0, rb0)
        MD_FETCH((nst > 1 ? 1 : 0), ra1, rb1)
    }
    const int a_off = (wr * 16 + (lane & 3)) * MD_LDA + 4 * lk;
    const int bn_off = (4 * lk) * MD_LDB + wc * 32 + 2 * li;
    const int bt_off = (wc * 32 + 2 * li) * MD_LDA + 4 * lk;
#define MD_STAGE(ST, RA, RB)                                                                                \
    {                                                                                                       \
        __syncthreads();      /* the previous stage's fragment reads are done */                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) *(v2f64*)(As + ar * MD_LDA + ak + 2 * i) = RA[i];    \
        if (B_T) {                                                                                          \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) *(v2f64*)(Bs + ar * MD_LDA + ak + 2 * i) = RB[i]; \
        } else {                                                                                            \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) *(v2f64*)(Bs + bk * MD_LDB + bc + 2 * i) = RB[i]; \
        }                                                                                                   \
        __syncthreads();                                                                                    \
        /* unconditional (the last two stages re-fetch the final one): with a conditional fetch the number of loads in     \
           flight depends on the path and hipcc waits for ALL of them (vmcnt(0)) before the next LDS store -- that  \
           halved the prefetch distance; the operands come from the Infinity Cache / HBM at ~1.5 us a round trip */  \
        MD_FETCH(((ST) + 2 < nst ? (ST) + 2 : nst - 1), RA, RB)                                             \
        /* fragment sets of the 4 k-groups (kk, p) of the stage, software-pipelined: group G + 1 is read from LDS while  \
           the 32 MFMAs of group G issue (one wavefront per SIMD: nothing else overlaps LDS with the matrix pipe) */      \
        MD_LOADF(0, fa0, fb0)                                                                               \
        MD_LOADF(1, fa1, fb1) MD_MMA(fa0, fb0) __builtin_amdgcn_sched_barrier(0);                           \
        MD_LOADF(2, fa0, fb0) MD_MMA(fa1, fb1) __builtin_amdgcn_sched_barrier(0);                           \
        MD_LOADF(3, fa1, fb1) MD_MMA(fa0, fb0) __builtin_amdgcn_sched_barrier(0);                           \
        MD_MMA(fa1, fb1)                                                                                    \
    }
    v2f64 fa0[2][4], fa1[2][4];
    double fb0[2][2], fb1[2][2];      // [e][nt]
#define MD_LOADF(G, FA, FB)                                                                                 \
    {                                                                                                       \
        constexpr int kq = ((G) >> 1) * 16 + ((G) & 1) * 2;      /* k offset of the group inside the stage */ \
        if (B_T) {                                                                                          \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                             \
                const v2f64 v = *(const v2f64*)(Bs + bt_off + nt * MD_LDA + kq);                            \
                FB[0][nt] = v[0];                                                                           \
                FB[1][nt] = v[1];                                                                           \
            }                                                                                               \
        } else {                                                                                            \
            _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                \
                const v2f64 v = *(const v2f64*)(Bs + bn_off + (kq + e) * MD_LDB);                           \
                FB[e][0] = v[0];                                                                            \
                FB[e][1] = v[1];                                                                            \
            }                                                                                               \
        }                                                                                                   \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                   \
            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                  \
                FA[mt][r] = *(const v2f64*)(As + a_off + (mt * 32 + 4 * r) * MD_LDA + kq);                  \
    }
/* 16 independent accumulators between two uses of one */
#define MD_MMA(FA, FB)                                                                                      \
    _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                          \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                   \
            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                  \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                           \
                    acc[mt][nt][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(FA[mt][r][e], FB[e][nt], acc[mt][nt][r], 0, 0, 0);
    // pairs of stages without a branch between them (see MD_FETCH above: the load count in flight must not depend on the path)
    int st = 0;
    for (; st + 1 < nst; st += 2) {
        MD_STAGE(st, ra0, rb0)
        MD_STAGE(st + 1, ra1, rb1)
    }
    if (st < nst) MD_STAGE(st, ra0, rb0)
#undef MD_STAGE
#undef MD_FETCH
#undef MD_LOADF
#undef MD_MMA
    double* cp = C + ((int64_t)rb * MD_BM + wr * 16 + lk) * g.ldc + (int64_t)cb * MD_BN + wc * 32 + 2 * li;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* q = cp + (int64_t)(mt * 32 + 4 * r) * g.ldc;
            v2f64 v = (v2f64){g.alpha * acc[mt][0][r], g.alpha * acc[mt][1][r]};
            if (g.accumulate) v += *(const v2f64*)q;
            *(v2f64*)q = v;
        }
}

// The same tile on EIGHT wavefronts (2 x 4 of 32 x 16): two wavefronts per SIMD, so one's barriers, LDS stores and fragment
// reads run under the other's MFMAs -- the four-wavefront form executes them back to back (measured per 32-k stage:
// 0.77 us of MFMAs + 0.46 us of everything else, additive).
template <bool B_T>
__device__ __forceinline__ void gemm_mid8_body(const GemmArgs& g, int nrb, int ncb) {
    __shared__ __attribute__((aligned(16))) double As[MD_BM * MD_LDA];
    __shared__ __attribute__((aligned(16))) double Bs[B_T ? MD_BN * MD_LDA : MD_BK * MD_LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3, li = lane & 15, lk = lane >> 4;
    const int rb = nrb - 1 - (int)(blockIdx.x / ncb), cb = (int)(blockIdx.x % ncb);
    if (g.lower_out && cb > (rb | 1)) return;
    const int z = blockIdx.z;
    const double* A = g.A + (g.zlayers > 1 ? z * g.zsA : 0);
    const double* B = g.B + (g.zlayers > 1 ? z * g.zsB : 0);
    double* C = g.C + (g.zlayers > 1 ? z * g.zsC : 0);
    int64_t k0 = 0, k1 = g.Kd;
    if (g.Kreal > 0) { const int64_t ke = (g.Kreal + MD_BK - 1) / MD_BK * MD_BK; if (k1 > ke) k1 = ke; }
    if (g.tri & TRI_LOWER_A) { const int64_t e = (int64_t)(rb + 1) * MD_BM; if (k1 > e) k1 = e; }
    if (g.tri & TRI_UPPER_A) { const int64_t b = (int64_t)rb * MD_BM; if (k0 < b) k0 = b; }
    if (g.tri & TRI_LOWER_B) { const int64_t b = (int64_t)cb * MD_BN; if (k0 < b) k0 = b; }
    if (g.tri & TRI_UPPER_B) { const int64_t e = (int64_t)(cb + 1) * MD_BN; if (k1 > e) k1 = e; }
    const int nst = k1 > k0 ? (int)((k1 - k0) / MD_BK) : 0;
    // staging maps (512 threads): A / B^T image 64 rows x 32 k -- (row t/8, 4 k); B image 32 k x 64 columns -- (k t/16, 4 columns)
    const int ar = tid >> 3, ak = (tid & 7) * 4;
    const int bk = tid >> 4, bc = (tid & 15) * 4;
    const double* Ag = A + ((int64_t)rb * MD_BM + ar) * g.lda + ak;
    const double* Bg = B_T ? B + ((int64_t)cb * MD_BN + ar) * g.ldb + ak : B + (int64_t)bk * g.ldb + (int64_t)cb * MD_BN + bc;
    v2f64 ra[2], rbv[2];
    auto fetch = [&](int st) {
        const int64_t k = k0 + (int64_t)st * MD_BK;
        ra[0] = *(const v2f64*)(Ag + k);
        ra[1] = *(const v2f64*)(Ag + k + 2);
        const double* bp = B_T ? Bg + k : Bg + k * g.ldb;
        rbv[0] = *(const v2f64*)bp;
        rbv[1] = *(const v2f64*)(bp + 2);
    };
    v4f64 acc[2];      // [mt][r]: row = wr*16 + mt*32 + 4r + lk, column = wc*16 + li
    acc[0] = acc[1] = (v4f64){0.0, 0.0, 0.0, 0.0};
    if (nst > 0) fetch(0);
    const int a_off = (wr * 16 + (lane & 3)) * MD_LDA + 4 * lk;
    const int bn_off = (4 * lk) * MD_LDB + wc * 16 + li;
    const int bt_off = (wc * 16 + li) * MD_LDA + 4 * lk;
    for (int st = 0; st < nst; ++st) {
        __syncthreads();      // the previous stage's fragment reads are done
        *(v2f64*)(As + ar * MD_LDA + ak) = ra[0];
        *(v2f64*)(As + ar * MD_LDA + ak + 2) = ra[1];
        if (B_T) {
            *(v2f64*)(Bs + ar * MD_LDA + ak) = rbv[0];
            *(v2f64*)(Bs + ar * MD_LDA + ak + 2) = rbv[1];
        } else {
            *(v2f64*)(Bs + bk * MD_LDB + bc) = rbv[0];
            *(v2f64*)(Bs + bk * MD_LDB + bc + 2) = rbv[1];
        }
        __syncthreads();
        if (st + 1 < nst) fetch(st + 1);      // in flight while this stage multiplies
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            const int kq = (G >> 1) * 16 + (G & 1) * 2;
            double fb[2];
            if (B_T) {
                const v2f64 v = *(const v2f64*)(Bs + bt_off + kq);
                fb[0] = v[0];
                fb[1] = v[1];
            } else {
                fb[0] = Bs[bn_off + kq * MD_LDB];
                fb[1] = Bs[bn_off + (kq + 1) * MD_LDB];
            }
            v2f64 fa[2][4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) fa[mt][r] = *(const v2f64*)(As + a_off + (mt * 32 + 4 * r) * MD_LDA + kq);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[mt][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[mt][r][e], fb[e], acc[mt][r], 0, 0, 0);
        }
    }
    double* cp = C + ((int64_t)rb * MD_BM + wr * 16 + lk) * g.ldc + (int64_t)cb * MD_BN + wc * 16 + li;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* q = cp + (int64_t)(mt * 32 + 4 * r) * g.ldc;
            double v = g.alpha * acc[mt][r];
            if (g.accumulate) v += *q;
            *q = v;
        }
}

// 32 x 64 tiles, 64-k stages (four wavefronts of 16 x 32): twice the workgroups of the 64 x 64 form, and half the barriers,
// LDS hand-overs and exposed fragment reads per k -- what a stage costs beside its MFMAs at one workgroup per CU.
#define MS_BM 32
#define MS_BK 64
#define MS_LDA (MS_BK + 2)
#define MS_ASWZ(row) ((((row) & 1) << 1) | ((((row) >> 1) & 1) << 4))      // doubles: {0, 2, 16, 18} for row % 4 = 0 .. 3
template <bool B_T>
__device__ __forceinline__ void gemm_mid32_body(const GemmArgs& g, int nrb, int ncb) {
    // A image: rows of exactly 64 doubles, the 16-byte column groups XOR-swizzled by the row (MS_ASWZ): a wavefront's fragment read
    // has 16 distinct addresses -- rows j = lane % 4, contraction offsets 4 lk -- and with padded rows (stride = 4 banks mod 64) the
    // four lane groups' offsets (8 banks apart) land on each other's rows' banks; swizzled, the 16 addresses cover the 64 banks once
    __shared__ __attribute__((aligned(16))) double As[MS_BM * MS_BK];
    __shared__ __attribute__((aligned(16))) double Bs[B_T ? MD_BN * MS_LDA : MS_BK * MD_LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 15, lk = lane >> 4;
    const int rb = nrb - 1 - (int)(blockIdx.x / ncb), cb = (int)(blockIdx.x % ncb);      // 32-row blocks, 64-column blocks
    if (g.lower_out && cb > (rb >> 1 | 1)) return;      // the 128 x 128 lower tiles, as the tiled kernel writes them
    const int z = blockIdx.z;
    const double* A = g.A + (g.zlayers > 1 ? z * g.zsA : 0);
    const double* B = g.B + (g.zlayers > 1 ? z * g.zsB : 0);
    double* C = g.C + (g.zlayers > 1 ? z * g.zsC : 0);
    int64_t k0 = 0, k1 = g.Kd;
    if (g.Kreal > 0) { const int64_t ke = (g.Kreal + MS_BK - 1) / MS_BK * MS_BK; if (k1 > ke) k1 = ke; }
    // triangular operands: k range per 64-block (the rows of this tile lie in 64-block rb / 2)
    if (g.tri & TRI_LOWER_A) { const int64_t e = (int64_t)(rb / 2 + 1) * 64; if (k1 > e) k1 = e; }
    if (g.tri & TRI_UPPER_A) { const int64_t b = (int64_t)(rb / 2) * 64; if (k0 < b) k0 = b; }
    if (g.tri & TRI_LOWER_B) { const int64_t b = (int64_t)cb * MD_BN; if (k0 < b) k0 = b; }
    if (g.tri & TRI_UPPER_B) { const int64_t e = (int64_t)(cb + 1) * MD_BN; if (k1 > e) k1 = e; }
    const int nst = k1 > k0 ? (int)((k1 - k0) / MS_BK) : 0;
    // staging maps (round 5): every image row is 64 doubles = 512 contiguous bytes; thread t takes the 16 bytes number t % 32 of
    // row t / 32 + 8 i -- a wavefront's load instruction reads two whole rows, 16 sectors of 64 bytes.  Until round 5 a thread
    // took 64 contiguous bytes of a row as four 16-byte loads: every one of its load instructions touched 64 separate sectors,
    // four times the address-unit work for the same bytes, 768 sector accesses per wavefront and stage -- ~1.3 us of each
    // 2.3 us stage, whatever else in the kernel was changed.
    const int fr = tid >> 5, fc = (tid & 31) * 2;
    const double* Ag = A + ((int64_t)rb * MS_BM + fr) * g.lda + fc;
    const double* Bg = B_T ? B + ((int64_t)cb * MD_BN + fr) * g.ldb + fc : B + (int64_t)fr * g.ldb + (int64_t)cb * MD_BN + fc;
    v2f64 ra[4], rbv[8];
    auto fetch = [&](int st) {
        const int64_t k = k0 + (int64_t)st * MS_BK;
        const double* bp = B_T ? Bg + k : Bg + k * g.ldb;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *(const v2f64*)(Ag + (int64_t)(8 * i) * g.lda + k);
#pragma unroll
        for (int i = 0; i < 8; ++i) rbv[i] = *(const v2f64*)(bp + (int64_t)(8 * i) * g.ldb);
    };
    v4f64 acc[2];      // [nt][r]: row = wr*16 + 4r + lk, column = wc*32 + 2 li + nt
    acc[0] = acc[1] = (v4f64){0.0, 0.0, 0.0, 0.0};
    if (nst > 0) fetch(0);
    const int a_row = (wr * 16 + (lane & 3)) * MS_BK, a_swz = MS_ASWZ(lane);
    const int bn_off = (4 * lk) * MD_LDB + wc * 32 + 2 * li;
    const int bt_off = (wc * 32 + 2 * li) * MS_LDA + 4 * lk;
    for (int st = 0; st < nst; ++st) {
        __syncthreads();      // the previous stage's fragment reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) *(v2f64*)(As + (fr + 8 * i) * MS_BK + (fc ^ MS_ASWZ(fr))) = ra[i];      // ((fr + 8 i) % 4 == fr % 4)
#pragma unroll
        for (int i = 0; i < 8; ++i) *(v2f64*)(Bs + (fr + 8 * i) * (B_T ? MS_LDA : MD_LDB) + fc) = rbv[i];
        __syncthreads();
        fetch(st + 1 < nst ? st + 1 : st);      // unconditional: a path-dependent load count costs a full drain (see above)
        // eight groups of 16 MFMAs, (kq, p) = (16 (g / 2), g % 2); the six 16-byte fragment reads of group g + 1 are issued BEFORE
        // the MFMAs of group g (round 5): left to the compiler every group opened by waiting for its own reads, and with one
        // wavefront per SIMD nothing hides them (worth 4 % of the launch; the coalesced staging loads above were worth 16 %)
        v2f64 fa[2][4];
        v2f64 fb[2][2];      // B^T form: [nt] = (k, k+1) of column nt; A B form: [e] = columns (nt 0, nt 1) of row k + e
        auto frag = [&](auto SET, int gi) {
            constexpr int S = decltype(SET)::value;
            const int kk = 16 * (gi >> 1) + 2 * (gi & 1);
            if (B_T) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) fb[S][nt] = *(const v2f64*)(Bs + bt_off + nt * MS_LDA + kk);
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e) fb[S][e] = *(const v2f64*)(Bs + bn_off + (kk + e) * MD_LDB);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) fa[S][r] = *(const v2f64*)(As + a_row + (4 * r) * MS_BK + ((4 * lk + kk) ^ a_swz));
        };
        auto mma = [&](auto SET) {
            constexpr int S = decltype(SET)::value;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[nt][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[S][r][e], B_T ? fb[S][nt][e] : fb[S][e][nt], acc[nt][r], 0, 0, 0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        frag(I0{}, 0);
#pragma unroll
        for (int gp = 0; gp < 4; ++gp) {
            frag(I1{}, 2 * gp + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(I0{});
            __builtin_amdgcn_sched_barrier(0);
            if (gp < 3) frag(I0{}, 2 * gp + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(I1{});
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    double* cp = C + ((int64_t)rb * MS_BM + wr * 16 + lk) * g.ldc + (int64_t)cb * MD_BN + wc * 32 + 2 * li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double* q = cp + (int64_t)(4 * r) * g.ldc;
        v2f64 v = (v2f64){g.alpha * acc[0][r], g.alpha * acc[1][r]};
        if (g.accumulate) v += *(const v2f64*)q;
        *(v2f64*)q = v;
    }
}
template <bool B_T>
__global__ __launch_bounds__(256) void gemm_mid32_kernel(GemmArgs g, int nrb, int ncb) {
    gemm_mid32_body<B_T>(g, nrb, ncb);
}
__global__ __launch_bounds__(256) void gemm_mid32x2_kernel(GemmArgs g0, GemmArgs g1, int nrb, int ncb) {
    if (blockIdx.y == 0) gemm_mid32_body<false>(g0, nrb, ncb);
    else gemm_mid32_body<false>(g1, nrb, ncb);
}

template <bool B_T>
__global__ __launch_bounds__(256) void gemm_mid_kernel(GemmArgs g, int nrb, int ncb) {
    gemm_mid_body<B_T>(g, nrb, ncb);
}
template <bool B_T>
__global__ __launch_bounds__(512) void gemm_mid8_kernel(GemmArgs g, int nrb, int ncb) {
    gemm_mid8_body<B_T>(g, nrb, ncb);
}
__global__ __launch_bounds__(512) void gemm_mid8x2_kernel(GemmArgs g0, GemmArgs g1, int nrb, int ncb) {
    if (blockIdx.y == 0) gemm_mid8_body<false>(g0, nrb, ncb);
    else gemm_mid8_body<false>(g1, nrb, ncb);
}
// two independent A B products of the same shape in one launch (blockIdx.y picks the problem): half-empty grids side by side
__global__ __launch_bounds__(256) void gemm_mid2_kernel(GemmArgs g0, GemmArgs g1, int nrb, int ncb) {
    if (blockIdx.y == 0) gemm_mid_body<false>(g0, nrb, ncb);
    else gemm_mid_body<false>(g1, nrb, ncb);
}

static bool mid_gemm_ok(const GemmArgs& g) {
    const int L = tune().mid_gemm_max;
    return g.epi == EPI_STORE && g.batched <= 1 && !g.bscale && !g.skip_if_zero && !g.sym_out && !g.dual_flag && !g.colact &&
           !g.kact && g.Mr <= L && g.Nc <= L && g.Kd <= L && g.Mr % MD_BM == 0 && g.Nc % MD_BN == 0 && g.Kd % MD_BK == 0 &&
           !(g.lda & 1) && !(g.ldb & 1) && !(g.ldc & 1) && !((uintptr_t)g.A & 15) && !((uintptr_t)g.B & 15) &&
           !((uintptr_t)g.C & 15) && (g.zlayers <= 1 || (!(g.zsA & 1) && !(g.zsB & 1) && !(g.zsC & 1)));
}
static int launch_mid_gemm(const GemmArgs& g, bool B_T, hipStream_t s) {
    const int nrb = g.Mr / MD_BM, ncb = (int)(g.Nc / MD_BN);
    const dim3 grid((unsigned)(nrb * ncb), 1, (unsigned)(g.zlayers > 1 ? g.zlayers : 1));
    if (tune().mid_gemm_waves == 32 && g.Kd % MS_BK == 0) {
        const int nrb32 = g.Mr / MS_BM;
        const dim3 grid32((unsigned)(nrb32 * ncb), 1, (unsigned)(g.zlayers > 1 ? g.zlayers : 1));
        if (B_T) hipLaunchKernelGGL(gemm_mid32_kernel<true>, grid32, dim3(256), 0, s, g, nrb32, ncb);
        else hipLaunchKernelGGL(gemm_mid32_kernel<false>, grid32, dim3(256), 0, s, g, nrb32, ncb);
    } else if (tune().mid_gemm_waves == 8) {
        if (B_T) hipLaunchKernelGGL(gemm_mid8_kernel<true>, grid, dim3(512), 0, s, g, nrb, ncb);
        else hipLaunchKernelGGL(gemm_mid8_kernel<false>, grid, dim3(512), 0, s, g, nrb, ncb);
    } else {
        if (B_T) hipLaunchKernelGGL(gemm_mid_kernel<true>, grid, dim3(256), 0, s, g, nrb, ncb);
        else hipLaunchKernelGGL(gemm_mid_kernel<false>, grid, dim3(256), 0, s, g, nrb, ncb);
    }
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}

// Two independent plain A B products (same shape, same layer batching) -- one launch on the mid-size kernel when both qualify,
// otherwise one after the other through launch_gemm_auto.
int launch_gemm_auto_pair(const GemmArgs& a, const GemmArgs& b, double* ws, int64_t ws_elems, hipStream_t s) {
    const int nza = a.zlayers > 1 ? a.zlayers : 1, nzb = b.zlayers > 1 ? b.zlayers : 1;
    GemmArgs a1 = a, b1 = b;
    a1.batched = b1.batched = 1;
    const bool small = small_gemm_ok(nza > 1 ? a1 : a, false) || small_gemm_ok(nzb > 1 ? b1 : b, false);
    if (small && nza == nzb && a.Mr == b.Mr && a.Nc == b.Nc && a.batched <= 1 && b.batched <= 1 &&
        small_gemm_ok(nza > 1 ? a1 : a, false) && small_gemm_ok(nzb > 1 ? b1 : b, false) && !a.skip_if_zero && !b.skip_if_zero) {
        // both on the small-operand kernel: one launch, the problems interleaved in blockIdx.z
        GemmArgs x = nza > 1 ? a1 : a, y = nzb > 1 ? b1 : b;
        const dim3 grid((unsigned)(x.Nc / 16), (unsigned)(x.Mr / 16), (unsigned)(2 * nza));
        hipLaunchKernelGGL(small_gemm2_kernel, grid, dim3(256), 0, s, x, y);
        return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
    }
    if (!small && nza == nzb && a.Mr == b.Mr && a.Nc == b.Nc && a.batched <= 1 && b.batched <= 1 && mid_gemm_ok(a) && mid_gemm_ok(b)) {
        const int nrb = a.Mr / MD_BM, ncb = (int)(a.Nc / MD_BN);
        if (tune().mid_gemm_waves == 32 && a.Kd % MS_BK == 0 && b.Kd % MS_BK == 0)
            hipLaunchKernelGGL(gemm_mid32x2_kernel, dim3((unsigned)((a.Mr / MS_BM) * ncb), 2, (unsigned)nza), dim3(256), 0, s, a, b,
                               a.Mr / MS_BM, ncb);
        else if (tune().mid_gemm_waves == 8)
            hipLaunchKernelGGL(gemm_mid8x2_kernel, dim3((unsigned)(nrb * ncb), 2, (unsigned)nza), dim3(512), 0, s, a, b, nrb, ncb);
        else
            hipLaunchKernelGGL(gemm_mid2_kernel, dim3((unsigned)(nrb * ncb), 2, (unsigned)nza), dim3(256), 0, s, a, b, nrb, ncb);
        return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
    }
    int rc = launch_gemm_auto(a, false, ws, ws_elems, s);
    if (rc) return rc;
    return launch_gemm_auto(b, false, ws, ws_elems, s);
}

// Small-grid GEMMs (M x M operands: a handful of 128x128 tiles on 256 CUs) are bound by one CU's MFMA rate:
// slice k over more workgroups into slabs, then add the slabs.  ws must hold splitk * Mr * Nc doubles.
int launch_gemm_auto(const GemmArgs& g0, bool B_T, double* ws, int64_t ws_elems, hipStream_t s) {
    // zlayers > 1: the same product for several layers (operands + l*zs*), blockIdx.z = layer; ws = the slab buffer of
    // layer 0, the other layers' slab buffers lie zsC apart like every other chain operand
    const int nz = g0.zlayers > 1 ? g0.zlayers : 1;
    if (nz > 1 && g0.batched > 1) return launch_gemm(g0, B_T, 1, s);      // batch x layers: plain z-batched launch
    if (nz > 1) {
        GemmArgs g1 = g0;
        g1.batched = 1;
        if (small_gemm_ok(g1, B_T)) return launch_small_gemm(g1, B_T, s);
    } else if (small_gemm_ok(g0, B_T)) return launch_small_gemm(g0, B_T, s);
    if (mid_gemm_ok(g0)) return launch_mid_gemm(g0, B_T, s);      // one launch, no slabs (the M x M chain at 384 < M <= 1024)
    GemmArgs g = g0;
    const int nrb = g.Mr / BM;
    const int64_t ncb = g.Nc / BN;
    const int64_t ntile = g.lower_out ? (int64_t)nrb * (nrb + 1) / 2 : (int64_t)nrb * ncb;
    int64_t nk = g.Kd / BK;
    if (g.tri) nk = (nk + 1) / 2;
    int sk = 1;
    if (!g.batched && g.epi == EPI_STORE && ntile * nz < 96 * nz && ntile < 96 && nk >= 8) {
        sk = (int)(512 / (ntile * nz));      // up to one round of resident workgroups
        if (sk > nk / 2) sk = (int)(nk / 2);
        if (sk >= 8) sk &= ~7;
        if ((int64_t)sk * g.Mr * g.Nc > ws_elems) sk = (int)(ws_elems / ((int64_t)g.Mr * g.Nc));
        if (sk >= 8) sk &= ~7;
    }
    if (sk < 2) {
        if (nz > 1) { g.batched = 1; return launch_gemm(g, B_T, 1, s); }
        return launch_gemm(g0, B_T, 1, s);
    }
    g.C = ws;
    g.ldc = g.Nc;
    g.slab_stride = (int64_t)g.Mr * g.Nc;
    g.accumulate = 0;
    // (layer l: slabs at ws + l*zsC -- zsC is the chain-block stride, the slab buffer sits inside the block)
    int rc = launch_gemm(g, B_T, sk, s);
    if (rc) return rc;
    int64_t n = (int64_t)g.Mr * g.Nc;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n + 255) / 256), 1, (unsigned)nz), dim3(256), 0, s,
                       (const double*)ws, g.slab_stride, sk, g0.C, g0.ldc, g.Mr, g.Nc, 1.0, g0.lower_out, 0, g0.accumulate,
                       nz > 1 ? g0.zsC : (int64_t)0, nz > 1 ? g0.zsC : (int64_t)0);
    return hipGetLastError() == hipSuccess ? MOBOCMF_OK : MOBOCMF_HIP_ERROR;
}
