// mfma_i8_layout.hip -- operand lane maps of v_mfma_i32_16x16x64_i8 on gfx950, found with exact integer data (the guide gives the bf16 maps
// and says to check other dtypes).  Each lane supplies 16 bytes of A and of B; the host tries candidate (lane, byte) -> (row / col, k) maps
// against a plain triple loop.   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/mfma_i8_layout tools/ubench/mfma_i8_layout.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const v4i *a, const v4i *b, v4i *d)
{
    const v4i c = {0, 0, 0, 0};
    d[threadIdx.x] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
}
int main()
{
    std::vector<int8_t> A(16 * 64), B(64 * 16); // A[m][k], B[k][n]
    srand(3);
    for (auto &v : A) v = (int8_t)(rand() % 255 - 127);
    for (auto &v : B) v = (int8_t)(rand() % 255 - 127);
    int ref[16][16];
    for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) { int s = 0; for (int kk = 0; kk < 64; kk++) s += (int)A[m * 64 + kk] * (int)B[kk * 16 + n]; ref[m][n] = s; }
    v4i *da, *db, *dd;
    hipMalloc(&da, 64 * 16); hipMalloc(&db, 64 * 16); hipMalloc(&dd, 64 * 16);
    for (int hyp = 0; hyp < 3; hyp++) {
        std::vector<int8_t> fa(64 * 16), fb(64 * 16);
        for (int l = 0; l < 64; l++) for (int j = 0; j < 16; j++) {
            int kk = hyp == 0 ? 16 * (l >> 4) + j : hyp == 1 ? 8 * (l >> 4) + (j & 7) + 32 * (j >> 3) : 4 * (l >> 4) + (j & 3) + 16 * (j >> 2);
            fa[l * 16 + j] = A[(l & 15) * 64 + kk];
            fb[l * 16 + j] = B[kk * 16 + (l & 15)];
        }
        hipMemcpy(da, fa.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, fb.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
        int out[64 * 4];
        hipMemcpy(out, dd, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) bad += out[l * 4 + i] != ref[4 * (l >> 4) + i][l & 15]; // C/D: col = lane & 15, row = 4 (lane >> 4) + reg
        printf("hypothesis %d (k of byte j of lane l = %s): %d of 256 outputs differ\n", hyp,
               hyp == 0 ? "16 (l >> 4) + j" : hyp == 1 ? "8 (l >> 4) + (j & 7) + 32 (j >> 3)" : "4 (l >> 4) + (j & 3) + 16 (j >> 2)", bad);
    }
    return 0;
}
