// Measures what pure instruction streams sustain on this chip (no memory traffic): the practical FP64
// ceilings (DVFS included) that the GEMM kernel's roofline fraction should be read against.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));

// mode 0: v_mfma_f64_16x16x4_f64   mode 1: v_mfma_f64_4x4x4_4b_f64   mode 2: v_fma_f64
// mode 3: waves 0,1 MFMA 16x16x4, waves 2,3 v_fma_f64 (co-issue test, one wave per SIMD)
// mode 4: half of the waves 4x4x4 MFMA, half v_fma_f64, chosen by (wave + blockIdx) parity: with two workgroups per CU every
//         SIMD hosts one wave of each kind -- do the matrix and the vector FP64 pipes add up, or are they one datapath?
template <int MODE>
__global__ __launch_bounds__(256) void loop_kernel(double* out, int iters, double a0, double b0) {
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    double s = 0;
    const int wave = threadIdx.x >> 6;
    bool do_mfma = MODE == 0 || (MODE == 3 && wave < 2);
    if (MODE == 1 || (MODE == 4 && ((wave + blockIdx.x) & 1))) {
        double acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i];
    } else if (do_mfma) {
        v4f64 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (v4f64){0, 0, 0, 0};
        // Round 4: through the builtin hipcc keeps these accumulators in AGPRs across the back edge and copies all of them to
        // VGPRs and back in every iteration (128 v_accvgpr_write + 128 v_accvgpr_read per 16 MFMAs): the round-2/3 figure of
        // 36 TFLOP/s measured those copies.  The instruction is written out with the accumulator tied to one VGPR tuple, so
        // the loop body is 16 MFMAs and the loop counter (checked with hipcc -S: no v_accvgpr_* in the loop).
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < (MODE == 4 ? 2 : 8); ++r)
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = __builtin_fma(a, b, acc[i]);
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) s += acc[i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double flops_per_wave_iter_mfma, double flops_per_wave_iter_valu, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        int grid = 256 * wgs_per_cu, iters = 2048;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(loop_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 0.5);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        double fl;
        if (MODE == 3 || MODE == 4) fl = 10.0 * grid * iters * (2 * flops_per_wave_iter_mfma + 2 * flops_per_wave_iter_valu);
        else fl = 10.0 * grid * 4 * iters * (MODE == 2 ? flops_per_wave_iter_valu : flops_per_wave_iter_mfma);
        printf("%-34s waves/SIMD=%d: %.1f TFLOP/s (%.2f ms)\n", name, wgs_per_cu, fl / best / 1e9, best);
    }
}

int main() {
    double* out;
    hipMalloc(&out, 8 * 256 * 2048);
    run<0>("v_mfma_f64_16x16x4_f64", 16.0 * 2048, 0, out);
    run<1>("v_mfma_f64_4x4x4_4b_f64", 16.0 * 512, 0, out);
    run<2>("v_fma_f64", 0, 8.0 * 32 * 128, out);
    run<3>("2 waves MFMA + 2 waves v_fma_f64", 16.0 * 2048, 8.0 * 32 * 128, out);
    run<4>("4x4x4 MFMA waves + v_fma_f64 waves mixed", 16.0 * 512, 2.0 * 32 * 128, out);
    return 0;
}
