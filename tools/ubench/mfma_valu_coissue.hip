// mfma_valu_coissue.hip -- round 5: does a wave that issues v_mfma_i32_16x16x64_i8 slow the VALU instruction stream of ANOTHER wave on the
// same SIMD?  (The riding-blur-on-the-matrix-pipe experiment, tools/experiments/blur_mfma_r05.patch, removed ~55 % of the riding blur's VALU
// instructions and FAST's launch did not get shorter; this measures why.)
// One workgroup per CU of 2 or 4 waves per SIMD (HW_ID's SIMD field is recorded to prove the spread).  The first n waves of each
// SIMD run role A, the rest role B; roles: 0 = exit at once, 1 = a stream of independent v_add_u32 (8 accumulators), 2 = a stream of
// independent MFMAs (4 accumulators), 3 = the blur's mix: 3 MFMAs then 16 v_add_u32.  Output: cycles per instruction of each role, alone
// and beside the other, from s_memtime around the loop (per wave, median over waves).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu_coissue tools/ubench/mfma_valu_coissue.hip && /tmp/mfma_valu_coissue > profiles/r05_mfma_valu_coissue.json
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void valu_block(unsigned &a0, unsigned &a1, unsigned &a2, unsigned &a3, unsigned &a4, unsigned &a5, unsigned &a6, unsigned &a7, unsigned b)
{
#pragma unroll
    for (int u = 0; u < 8; u++) {
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(b));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a1) : "v"(b));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(b));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a3) : "v"(b));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a4) : "v"(b));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a5) : "v"(b));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a6) : "v"(b));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a7) : "v"(b));
    }
}

__global__ __launch_bounds__(1024) void coissue_kernel(unsigned *sink, long long *cycles, unsigned *simd_of, int role_a, int role_b, int n_first, int iters, unsigned seed)
{
    const int wave = threadIdx.x >> 6;
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const int simd = (hwid >> 4) & 3;
    // the first wave to land on a SIMD takes role A, the second role B: decided per SIMD through LDS
    __shared__ int s_seen[4];
    if (threadIdx.x < 4) s_seen[threadIdx.x] = 0;
    __syncthreads();
    int order = 0;
    if ((threadIdx.x & 63) == 0) order = atomicAdd(&s_seen[simd], 1);
    order = __builtin_amdgcn_readfirstlane(order);
    __syncthreads();
    const int role = order < n_first ? role_a : role_b;
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 11u, a5 = a0 + 13u, a6 = a0 + 17u, a7 = a0 + 19u, b = seed | 1u;
    v4i m0 = {(int)a0, 1, 2, 3}, m1 = {(int)a1, 1, 2, 3}, m2 = {(int)a2, 1, 2, 3}, m3 = {(int)a3, 1, 2, 3};
    const v4i fa = {(int)(a0 & 0x01010101u), 0x01000100, 0x00010001, (int)b}, fb = {0x01010101, (int)(a1 & 0x01010101u), 0x01000001, 0x00010100};
    long long t0 = 0, t1 = 0;
    long long n_inst = 0;
    if (role != 0) {
        t0 = (long long)__builtin_readcyclecounter();
        if (role == 1) {
            for (int i = 0; i < iters; i++) valu_block(a0, a1, a2, a3, a4, a5, a6, a7, b);
            n_inst = (long long)iters * 64;
        } else if (role == 2) {
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    m0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, m0, 0, 0, 0);
                    m1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, m1, 0, 0, 0);
                    m2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, m2, 0, 0, 0);
                    m3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, m3, 0, 0, 0);
                }
            }
            n_inst = (long long)iters * 16;
        } else {
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    m0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, m0, 0, 0, 0);
                    m1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, m1, 0, 0, 0);
                    m2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, m2, 0, 0, 0);
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(b));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a1) : "v"(b));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(b));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a3) : "v"(b));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a4) : "v"(b));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a5) : "v"(b));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a6) : "v"(b));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a7) : "v"(b));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(b));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a1) : "v"(b));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(b));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a3) : "v"(b));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a4) : "v"(b));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a5) : "v"(b));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a6) : "v"(b));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a7) : "v"(b));
                }
            }
            n_inst = (long long)iters * 4 * 19;
        }
        t1 = (long long)__builtin_readcyclecounter();
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + wave;
        cycles[3 * w + 0] = role;
        cycles[3 * w + 1] = t1 - t0;
        cycles[3 * w + 2] = n_inst;
        simd_of[w] = (unsigned)(simd | ((order < n_first ? 0 : 1) << 4));
    }
    sink[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(m0.x + m1.y + m2.z + m3.w);
}

int main()
{
    const int n_wg = 256, iters = 4000;
    unsigned *sink, *simd_of;
    long long *cyc;
    hipMalloc(&sink, n_wg * 1024 * 4);
    hipMalloc(&simd_of, n_wg * 16 * 4);
    hipMalloc(&cyc, n_wg * 16 * 3 * 8);
    std::vector<long long> h(n_wg * 16 * 3);
    std::vector<unsigned> hs(n_wg * 16);
    static const char *names[4] = {"idle", "valu", "mfma", "mix_3mfma_16valu"};
    // {role of the first n_first waves per SIMD, role of the rest, n_first, waves per SIMD}
    const int cases[][4] = {{1, 0, 1, 2}, {1, 1, 1, 2}, {1, 2, 1, 2}, {2, 0, 1, 2}, {2, 2, 1, 2}, {3, 0, 1, 2}, {3, 3, 1, 2}, {1, 3, 1, 2},
                            {1, 0, 3, 4}, {1, 1, 3, 4}, {1, 2, 3, 4}, {1, 3, 3, 4}, {1, 2, 2, 4}, {1, 3, 2, 4}};
    printf("{\"what\": \"cycles per instruction of one wave's stream, alone on its SIMD and beside a second wave (2 or 4 waves per SIMD, every CU loaded); s_memtime cycles\",\n \"cases\": [\n");
    const int nc = (int)(sizeof(cases) / sizeof(cases[0]));
    for (int c = 0; c < nc; c++) {
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(coissue_kernel, dim3(n_wg), dim3(64 * 4 * cases[c][3]), 0, 0, sink, cyc, simd_of, cases[c][0], cases[c][1], cases[c][2], iters, 12345u + rep);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hs.data(), simd_of, hs.size() * 4, hipMemcpyDeviceToHost);
        std::vector<double> cpi[2];
        int paired = 0;
        for (int wg = 0; wg < n_wg; wg++) {
            const int nw = 4 * cases[c][3];
            int per_simd[4] = {0, 0, 0, 0};
            for (int w = 0; w < nw; w++) per_simd[hs[wg * nw + w] & 3]++;
            paired += per_simd[0] == cases[c][3] && per_simd[1] == cases[c][3] && per_simd[2] == cases[c][3] && per_simd[3] == cases[c][3];
            for (int w = 0; w < nw; w++) {
                const long long *e = &h[3 * (wg * nw + w)];
                if (e[0] != 0) cpi[(hs[wg * nw + w] >> 4) & 1].push_back((double)e[1] / (double)e[2]);
            }
        }
        auto med = [](std::vector<double> &v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("  {\"waves_per_simd\": %d, \"first\": \"%d x %s\", \"rest\": \"%d x %s\", \"first_cycles_per_inst\": %.3f, \"rest_cycles_per_inst\": %.3f, \"workgroups_evenly_spread\": %d}%s\n",
               cases[c][3], cases[c][2], names[cases[c][0]], cases[c][3] - cases[c][2], names[cases[c][1]], med(cpi[0]), med(cpi[1]), paired, c + 1 < nc ? "," : "");
    }
    printf(" ]}\n");
    return 0;
}
