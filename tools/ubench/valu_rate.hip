// micro-benchmark: integer VALU issue rate per CU (wave64) for a few opcodes, full occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short pk16 __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(256) void k(int *out, int n, int seed) {
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    const int b = seed | 1;
    for (int i = 0; i < n; i++) {
#define STEP(x) if (OP == 0) x = min(x, b + i); else if (OP == 1) x = x + (b ^ i); else if (OP == 2) { pk16 v = *(pk16*)&x; pk16 w = {(short)b, (short)i}; v = __builtin_elementwise_min(v, w); x = *(int*)&v; } else if (OP == 3) x = x * b + i; else if (OP == 4) x = (int)__builtin_amdgcn_alignbyte(x, b, 1) ; else x = __builtin_amdgcn_ubfe(x, 3, 8) + b;
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP> void run(const char *name, int *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, n = 4096;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, n, 3);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double winstr = (double)blocks * 4 * n * 8; // wave-level instructions of the measured op
        if (rep) printf("%-14s %8.1f us  %.2f wave-instr/cycle/CU (at 2.4 GHz, 256 CUs)\n", name, ms * 1e3, winstr / (ms * 1e-3) / 2.4e9 / 256);
    }
}
int main() {
    int *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_min_i32", d); run<1>("v_add_u32", d); run<2>("v_pk_min_i16", d); run<3>("v_mad (mul_lo)", d); run<4>("v_alignbyte", d); run<5>("v_bfe+add", d);
    return 0;
}
