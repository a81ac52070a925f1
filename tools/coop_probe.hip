// Design probe for the cooperative small-surrogate step (csrc/coop_step.hip): what an in-launch barrier between k workgroups
// costs on this chip, and what one tile-parallel MFMA phase (16x16 output tile per wavefront, operands read straight from
// L2 in the matrix instruction's lane layout) costs between two such barriers.
//   hipcc --offload-arch=gfx950 -O3 tools/coop_probe.hip -o tools/coop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void group_barrier(unsigned long long* cnt, unsigned k, unsigned long long& epoch, int fence) {
    __syncthreads();
    if (k > 1) {
        if (threadIdx.x == 0) {
            if (fence) __threadfence();
            epoch += k;
            atomicAdd(cnt, 1ull);
            int spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) break;
            }
            if (fence) __threadfence();
        }
        __syncthreads();
    }
}

// mapping 0: group g = blockIdx / k (its workgroups on consecutive XCDs); 1: group g's workgroups all on XCD g % 8
__device__ __forceinline__ void who(int k, int mapping, int& g, int& j) {
    const int b = blockIdx.x;
    if (!mapping) { g = b / k; j = b % k; }
    else { const int x = b & 7, s = b >> 3; g = x + 8 * (s / k); j = s % k; }
}

__global__ void barrier_kernel(unsigned long long* cnts, int k, int groups, int nbar, int mapping, int fence, double* sink) {
    int g, j;
    who(k, mapping, g, j);
    if (g >= groups) return;
    unsigned long long epoch = 0;
    double v = threadIdx.x;
    for (int i = 0; i < nbar; ++i) {
        v = v * 1.0000001 + 1.0;
        group_barrier(cnts + 32 * g, k, epoch, fence);
    }
    if (v == 12345.678) sink[0] = v;
}

// one wavefront: D(16x16) = sum_k A[i0 + i][k] B[k][j0 + j]; TA: A given transposed (A[k][i]); TB: B transposed (B[j][k])
template <bool TA, bool TB>
__device__ __forceinline__ v4f64 mma_tile(const double* A, int lda, const double* B, int ldb, int i0, int j0, int k0, int k1,
                                          int lane) {
    v4f64 acc = {0, 0, 0, 0};
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll 8
    for (int k = k0; k < k1; k += 4) {
        const double a = TA ? A[(size_t)(k + lk) * lda + i0 + li] : A[(size_t)(i0 + li) * lda + k + lk];
        const double b = TB ? B[(size_t)(j0 + li) * ldb + k + lk] : B[(size_t)(k + lk) * ldb + j0 + li];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ void store_tile(double* C, int ldc, int i0, int j0, int lane, v4f64 acc) {
    const int lj = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(size_t)(i0 + 4 * r + lk) * ldc + j0 + lj] = acc[r];
}

// nphase times: C = A B (phase p even) / A = C^T B ... just alternate two buffers so that every phase depends on the last
template <int TT>
__global__ __launch_bounds__(TT) void phase_kernel(double* X, double* Y, const double* Bm, unsigned long long* cnts, int M, int k,
                                                   int groups, int nphase, int mapping, int variant) {
    int g, j;
    who(k, mapping, g, j);
    if (g >= groups) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = TT / 64;
    const int wg = j * NW + wave, nw = k * NW;
    double* x = X + (size_t)g * M * M;
    double* y = Y + (size_t)g * M * M;
    const int nt = M / 16;
    unsigned long long epoch = 0;
    for (int p = 0; p < nphase; ++p) {
        for (int t = wg; t < nt * nt; t += nw) {
            const int ti = t / nt, tj = t % nt;
            v4f64 acc;
            if (variant == 0) acc = mma_tile<false, false>(x, M, Bm, M, ti * 16, tj * 16, 0, M, lane);
            else if (variant == 1) acc = mma_tile<true, false>(x, M, Bm, M, ti * 16, tj * 16, 0, M, lane);
            else acc = mma_tile<false, true>(x, M, Bm, M, ti * 16, tj * 16, 0, M, lane);
            store_tile(y, M, ti * 16, tj * 16, lane, acc);
        }
        group_barrier(cnts + 32 * g, k, epoch, 1);
        double* t2 = x; x = y; y = t2;
    }
}

static float time_ms(hipEvent_t e0, hipEvent_t e1) {
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    unsigned long long* cnts;
    double* sink;
    hipMalloc(&cnts, 8 * 32 * 64);
    hipMalloc(&sink, 64);
    // ---- A: barrier cost
    printf("# in-launch barrier among k workgroups (256 threads each), us per barrier; groups = concurrent independent groups\n");
    printf("# k groups mapping fence us_per_barrier\n");
    const int nbar = 400;
    for (int groups : {1, 4}) {
        for (int k : {1, 2, 4, 8, 16, 32, 64}) {
            for (int mapping = 0; mapping < 2; ++mapping) {
                for (int fence = 0; fence < 2; ++fence) {
                    if (mapping == 1 && k > 32) continue;
                    const int grid = mapping ? 8 * k * ((groups + 7) / 8) : groups * k;
                    float best = 1e30f;
                    for (int rep = 0; rep < 3; ++rep) {
                        hipMemset(cnts, 0, 8 * 32 * 64);
                        hipDeviceSynchronize();
                        hipEventRecord(e0);
                        hipLaunchKernelGGL(barrier_kernel, dim3(grid), dim3(256), 0, 0, cnts, k, groups, nbar, mapping, fence, sink);
                        hipEventRecord(e1);
                        const float ms = time_ms(e0, e1);
                        if (ms < best) best = ms;
                    }
                    printf("%2d %d %s %s %.3f\n", k, groups, mapping ? "same-xcd" : "spread", fence ? "fence" : "nofence",
                           best * 1e3 / nbar);
                }
            }
        }
    }
    // ---- B: layout check of the direct-from-memory MFMA tile product
    for (int M : {64, 128}) {
        std::vector<double> hA(M * M), hB(M * M), hC(M * M), ref(M * M);
        srand(1);
        for (auto& v : hA) v = rand() / (double)RAND_MAX - 0.5;
        for (auto& v : hB) v = rand() / (double)RAND_MAX - 0.5;
        double *dX, *dY, *dB;
        hipMalloc(&dX, 8 * M * M * 8);
        hipMalloc(&dY, 8 * M * M * 8);
        hipMalloc(&dB, M * M * 8);
        hipMemcpy(dB, hB.data(), M * M * 8, hipMemcpyHostToDevice);
        for (int variant = 0; variant < 3; ++variant) {
            hipMemcpy(dX, hA.data(), M * M * 8, hipMemcpyHostToDevice);
            hipMemset(cnts, 0, 8 * 32 * 64);
            hipLaunchKernelGGL(phase_kernel<256>, dim3(4), dim3(256), 0, 0, dX, dY, dB, cnts, M, 4, 1, 1, 0, variant);
            hipMemcpy(hC.data(), dY, M * M * 8, hipMemcpyDeviceToHost);
            double err = 0;
            for (int i = 0; i < M; ++i)
                for (int j2 = 0; j2 < M; ++j2) {
                    double s = 0;
                    for (int kk = 0; kk < M; ++kk) {
                        const double a = variant == 1 ? hA[kk * M + i] : hA[i * M + kk];
                        const double b = variant == 2 ? hB[j2 * M + kk] : hB[kk * M + j2];
                        s += a * b;
                    }
                    err = fmax(err, fabs(s - hC[i * M + j2]));
                }
            printf("# layout check M=%d variant %d (0: A B, 1: A^T B, 2: A B^T): max abs err %.3e\n", M, variant, err);
        }
        // ---- C: phase cost = product + barrier, chained
        printf("# M=%d: us per phase (M x M x M product, one 16x16 tile per wavefront, + barrier)\n# threads k groups mapping variant us\n", M);
        const int nphase = 200;
        for (int groups : {1, 4}) {
            for (int k : {1, 2, 4, 8, 16, 32}) {
                for (int mapping = 0; mapping < 2; ++mapping) {
                    for (int tt : {256, 512, 1024}) {
                        for (int variant = 0; variant < 3; ++variant) {
                            if (variant && !(tt == 512)) continue;
                            const int grid = mapping ? 8 * k * ((groups + 7) / 8) : groups * k;
                            float best = 1e30f;
                            for (int rep = 0; rep < 3; ++rep) {
                                hipMemset(cnts, 0, 8 * 32 * 64);
                                hipMemset(dX, 0, 8 * M * M * 8);
                                hipDeviceSynchronize();
                                hipEventRecord(e0);
                                if (tt == 256)
                                    hipLaunchKernelGGL(phase_kernel<256>, dim3(grid), dim3(256), 0, 0, dX, dY, dB, cnts, M, k, groups, nphase, mapping, variant);
                                else if (tt == 512)
                                    hipLaunchKernelGGL(phase_kernel<512>, dim3(grid), dim3(512), 0, 0, dX, dY, dB, cnts, M, k, groups, nphase, mapping, variant);
                                else
                                    hipLaunchKernelGGL(phase_kernel<1024>, dim3(grid), dim3(1024), 0, 0, dX, dY, dB, cnts, M, k, groups, nphase, mapping, variant);
                                hipEventRecord(e1);
                                const float ms = time_ms(e0, e1);
                                if (ms < best) best = ms;
                            }
                            printf("%4d %2d %d %s %d %.3f\n", tt, k, groups, mapping ? "same-xcd" : "spread", variant, best * 1e3 / nphase);
                        }
                    }
                }
            }
        }
        hipFree(dX); hipFree(dY); hipFree(dB);
    }
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
