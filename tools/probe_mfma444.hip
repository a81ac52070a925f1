// Discovers the lane maps of v_mfma_f64_4x4x4_4b_f64 (and its cbsz/abid A-broadcast) with one-hot operands.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int CBSZ, int ABID>
__global__ void probe(int* out) {   // out[p*64+q] = lane of D that became 1 (or -1 / -2 if none / many)
    int lane = threadIdx.x;
    for (int p = 0; p < 64; ++p)
        for (int q = 0; q < 64; ++q) {
            double a = lane == p ? 1.0 : 0.0, b = lane == q ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
            unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) {
                int cnt = __popcll(m);
                out[p * 64 + q] = cnt == 0 ? -1 : (cnt == 1 ? __ffsll((long long)m) - 1 : -100 - cnt);
            }
        }
}

template <int CBSZ, int ABID>
void run(const char* name) {
    int* d;
    hipMalloc(&d, 64 * 64 * 4);
    hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, d);
    static int h[4096];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("== %s\n", name);
    for (int p = 0; p < 64; ++p) {
        printf("A lane %2d:", p);
        for (int q = 0; q < 64; ++q)
            if (h[p * 64 + q] != -1) printf(" B%d->D%d", q, h[p * 64 + q]);
        printf("\n");
    }
}

int main() {
    run<0, 0>("cbsz=0 abid=0");
    run<2, 0>("cbsz=2 abid=0");
    run<2, 1>("cbsz=2 abid=1");
    run<2, 3>("cbsz=2 abid=3");
    return 0;
}
