// micro-benchmark: workgroup dispatch rate on MI355X as a function of WG size and LDS use
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(int *out) { if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = 1; }
__global__ void k_lds(int *out) { extern __shared__ int s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); if (s[(threadIdx.x + 1) % blockDim.x] == 0x7fffffff) out[0] = 1; }
int main() {
    int *d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 100000;
    for (int threads : {64, 256, 512}) for (int lds : {0, 4096, 16384, 65536}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (lds == 0) hipLaunchKernelGGL(k_empty, dim3(N), dim3(threads), 0, 0, d);
            else hipLaunchKernelGGL(k_lds, dim3(N), dim3(threads), lds, 0, d);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("threads %4d lds %6d : %8.1f us for %d WGs -> %.0f WG/us\n", threads, lds, ms * 1e3, N, N / (ms * 1e3));
        }
    }
    return 0;
}
