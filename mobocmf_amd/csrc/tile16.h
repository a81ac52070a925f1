// 16 x 16 tiles of a lower-triangular matrix in LDS and their factorisation by one wavefront (gfx950): shared by the cooperative
// step (coop_step.hip) and the one-launch Cholesky (chol.hip, potrf_coop_kernel).
#pragma once
#include <hip/hip_runtime.h>

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4d mfma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

__device__ __forceinline__ double t16_rdlane(double v, int l) {      // l: wave-uniform
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], l);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
    return u.d;
}

// ---- swizzled 16 x 16 tiles of a lower-triangular matrix in LDS: tile (ti, tj <= ti) at tix * 256, element (r, c) at
// r * 16 + (c ^ r): rows, columns, and the matrix instruction's operand fragments of a tile are all read without bank conflicts
__device__ __forceinline__ int tix(int ti, int tj) { return (ti * (ti + 1) / 2 + tj) * 256; }
__device__ __forceinline__ int tel(int r, int c) { return r * 16 + (c ^ r); }

// Cholesky + inverse of the 16 x 16 tile T (swizzled; lower part used) by one wavefront: lane i of every 16-lane row holds row i
// (the four rows of the wavefront do the same work), right-looking, multipliers broadcast by v_readlane (tiny_step.hip
// chol_inv_wave).  T receives L (zeros above the diagonal), Ti its inverse.  Returns the 1-based failed pivot or 0 (wave-uniform).
// (Tried: taking the pivot block's inverse off the critical path -- the panel below it solved by substitution in registers, 136
// in-lane operations against wave-uniform LDS reads of L_ss, the inverses of all pivot blocks formed afterwards side by side --
// is SLOWER: the substitution costs what the inverse cost, 16.5 -> 21.8 us at M = 64, 36 -> 46.5 at M = 128.)
// (Tried: the panel tiles below the pivot tile riding in the factorisation's own column loop -- lane groups 1-3 of the wavefront
// hold their rows, scaling and elimination are the same instructions for them, the multipliers come out of the diagonal lanes by
// v_readlane either way -- so that nothing inside the factorisation waits for L_ss^-1 (inverted by an idle wavefront one pivot
// later): parity-green, and NO faster -- Cholesky phase 16.8 -> 16.0 us at M = 64, 35.7 -> 35.3 at M = 128, the step unchanged.
// The 3.3 us of this routine are its 16 dependent column steps (~0.2 us each: two v_readlane, v_rsq_f64, six dependent f64
// operations, then the eliminations that feed the next pivot), not the inverse behind them.)
// (Tried: pinning the eliminations where they are written -- an empty asm on the updated values after every column step: hipcc
// otherwise defers every update row[k] -= ... to column k's own step, a left-looking loop whose columns open with a chain of k
// dependent FMAs and whose multipliers wait in SGPRs, some spilled through v_writelane -- gives the right-looking instruction
// order and is NOT faster: Cholesky phase 16.0 -> 17.1 us at M = 64, the one-launch factorisation 214 -> 226 us at n = 512.)
// (Tried: the broadcasts as DPP row_newbcast moves -- no trip through an SGPR -- are SLOWER: 17.6 -> 21 us for the four tiles of
// M = 64, profiles/r05_coop_step.txt; the DPP move waits for its source's write-back where v_readlane's result is forwarded.)
__device__ __forceinline__ int chol_inv_tile16(double* T, double* Ti, int lane) {
    double row[16];
    const int ln = lane & 15;
#pragma unroll
    for (int k = 0; k < 16; ++k) row[k] = T[tel(ln, k)];
    int fail = 0;
    double rr[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double djj = t16_rdlane(row[j], j);
        if (!(djj > 0.0) && !fail) fail = j + 1;
        double r = __builtin_amdgcn_rsq(djj);
        r = r * (1.5 - (0.5 * djj) * r * r);
        r = __builtin_fma(0.5 * r, __builtin_fma(-(djj * r), r, 1.0), r);
        rr[j] = r;
        const double lij = row[j] * r;
        row[j] = lij;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) row[k] -= lij * t16_rdlane(lij, k);
    }
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double s = i == ln ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) s -= t16_rdlane(row[k], i) * x[k];
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

