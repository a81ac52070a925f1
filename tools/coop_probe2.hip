// Probe 2 for csrc/coop_step.hip: how the workgroups of a launch should hand matrices to each other between phases.
//   COH 0: plain loads / stores + __threadfence() on both sides of the barrier (L2 write-back + invalidate per workgroup)
//   COH 1: agent-scope relaxed atomic loads / stores (sc1: coherent across the XCDs' L2s), no fence, s_waitcnt before arriving
// Every run is CHECKED against the same recurrence on the host (x <- x^T B, B orthogonal), so a stale read shows.
//   hipcc --offload-arch=gfx950 -O3 tools/coop_probe2.hip -o tools/coop_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int COH>
__device__ __forceinline__ double ld(const double* p) {
    if (COH == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <int COH>
__device__ __forceinline__ void st(double* p, double v) {
    if (COH == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

template <int COH>
__device__ __forceinline__ void group_barrier(unsigned long long* cnt, unsigned k, unsigned long long& epoch) {
    if (COH == 1) __builtin_amdgcn_s_waitcnt(0);      // this wavefront's stores have left
    __syncthreads();
    if (k > 1) {
        if (threadIdx.x == 0) {
            if (COH == 0) __threadfence();
            epoch += k;
            __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) break;
            }
            if (COH == 0) __threadfence();
        }
        __syncthreads();
    }
}

// y = x^T B  (both operands k-major: every load instruction reads 4 rows x 128 contiguous bytes)
template <int TT, int COH, int M>
__global__ __launch_bounds__(TT) void phase_kernel(double* X, double* Y, const double* Bm, unsigned long long* cnts, int k,
                                                   int nphase) {
    const int g = blockIdx.x / k, j = blockIdx.x % k;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = TT / 64;
    const int wg = j * NW + wave, nw = k * NW;
    double* x = X + (size_t)g * M * M;
    double* y = Y + (size_t)g * M * M;
    constexpr int nt = M / 16;
    const int li = lane & 15, lk = lane >> 4;
    unsigned long long epoch = 0;
    for (int p = 0; p < nphase; ++p) {
        for (int t = wg; t < nt * nt; t += nw) {
            const int i0 = (t / nt) * 16, j0 = (t % nt) * 16;
            double a[M / 4], b[M / 4];
#pragma unroll
            for (int s = 0; s < M / 4; ++s) {
                a[s] = ld<COH>(x + (size_t)(4 * s + lk) * M + i0 + li);
                b[s] = Bm[(size_t)(4 * s + lk) * M + j0 + li];
            }
            v4f64 acc = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < M / 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) st<COH>(y + (size_t)(i0 + 4 * r + lk) * M + j0 + li, acc[r]);
        }
        group_barrier<COH>(cnts + 32 * g, k, epoch);
        double* t2 = x; x = y; y = t2;
    }
}

template <int TT, int COH, int M>
void run(int k, int groups, const std::vector<double>& hX, const std::vector<double>& ref, double* dX, double* dY, double* dB,
         unsigned long long* cnts, int nphase) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    double err = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(cnts, 0, 8 * 32 * 64);
        for (int g = 0; g < groups; ++g) hipMemcpy(dX + (size_t)g * M * M, hX.data(), M * M * 8, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((phase_kernel<TT, COH, M>), dim3(groups * k), dim3(TT), 0, 0, dX, dY, dB, cnts, k, nphase);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        std::vector<double> out(M * M);
        for (int g = 0; g < groups; ++g) {
            hipMemcpy(out.data(), (nphase % 2 ? dY : dX) + (size_t)g * M * M, M * M * 8, hipMemcpyDeviceToHost);
            for (int e = 0; e < M * M; ++e) err = fmax(err, fabs(out[e] - ref[e]));
        }
    }
    printf("%3d %4d %2d %d %s %.3f %.2e\n", M, TT, k, groups, COH ? "sc1-nofence" : "plain-fence", best * 1e3 / nphase, err);
}

template <int M>
void sweep(unsigned long long* cnts) {
    const int nphase = 200;
    std::vector<double> hX(M * M), hB(M * M, 0.0), ref, tmp(M * M);
    srand(3);
    for (auto& v : hX) v = rand() / (double)RAND_MAX - 0.5;
    // B = product of Givens rotations on a signed permutation: orthogonal, dense enough to mix
    for (int i = 0; i < M; ++i) hB[i * M + (i * 7 + 3) % M] = (i & 1) ? -1.0 : 1.0;
    for (int r = 0; r < 3 * M; ++r) {
        const int p = rand() % M, q = (p + 1 + rand() % (M - 1)) % M;
        const double th = rand() / (double)RAND_MAX * 6.28, c = cos(th), s = sin(th);
        for (int i = 0; i < M; ++i) {
            const double u = hB[i * M + p], v = hB[i * M + q];
            hB[i * M + p] = c * u - s * v;
            hB[i * M + q] = s * u + c * v;
        }
    }
    ref = hX;
    for (int p = 0; p < nphase; ++p) {
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < M; ++j) {
                double s = 0;
                for (int kk = 0; kk < M; ++kk) s += ref[kk * M + i] * hB[kk * M + j];
                tmp[i * M + j] = s;
            }
        ref.swap(tmp);
    }
    double *dX, *dY, *dB;
    hipMalloc(&dX, 8 * M * M * 8);
    hipMalloc(&dY, 8 * M * M * 8);
    hipMalloc(&dB, M * M * 8);
    hipMemcpy(dB, hB.data(), M * M * 8, hipMemcpyHostToDevice);
    for (int groups : {1, 4})
        for (int k : {1, 2, 4, 8, 16, 32, 64}) {
            if (groups * k > 256) continue;
            run<256, 0, M>(k, groups, hX, ref, dX, dY, dB, cnts, nphase);
            run<256, 1, M>(k, groups, hX, ref, dX, dY, dB, cnts, nphase);
            if (k <= 4) {
                run<1024, 0, M>(k, groups, hX, ref, dX, dY, dB, cnts, nphase);
                run<1024, 1, M>(k, groups, hX, ref, dX, dY, dB, cnts, nphase);
            }
        }
    hipFree(dX); hipFree(dY); hipFree(dB);
}

int main() {
    unsigned long long* cnts;
    hipMalloc(&cnts, 8 * 32 * 64);
    printf("# us per phase: M x M x M product (k-major operands, all loads of a tile issued up front) + barrier among k workgroups\n");
    printf("# M threads k groups coherence us_per_phase max_abs_err_vs_host_after_200_phases\n");
    sweep<64>(cnts);
    sweep<128>(cnts);
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
