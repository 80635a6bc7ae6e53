// valu_peak.hip -- what does one gfx950 SIMD issue per cycle, per opcode and per occupancy?
//
// Reconciles bench.py's VALU-issue peak with MI355X_MICROARCH.md ("wave64 VALU = 2 cycles on a SIMD-32 with more than one
// wave resident, 4 for a lone wave"; measured there for v_fma_f32).  For every opcode the front-end's hot kernels use
// (packed int16 min / max / sub, v_perm_b32, v_dot4, alignbyte, mbcnt, ...), fp32 controls (v_fma_f32, v_pk_fma_f32) and
// the packed-f16 candidates (v_pk_min/max_f16, v_pk_maximum3_f16) this runs a long stream of INDEPENDENT instructions
// (8 accumulators, inline asm so the compiler cannot fold them) at 1 / 2 / 4 / 8 waves per SIMD on every CU and reports
//   chip-wide wave-instructions per second = all instructions / (last wave's end - first wave's start), s_memrealtime ticks
//   the shader clock actually held         = d(s_memtime) / d(s_memrealtime) x 100 MHz (median over the waves)
//   cycles per wave-instruction per SIMD   = 256 CUs x 4 SIMDs x clock / that rate
// NOT run under rocprofv3 (profiled passes clock lower).  Output: one JSON document on stdout.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_peak tools/ubench/valu_peak.hip && /tmp/valu_peak > profiles/r02_valu_peak.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define UNROLL 16 // x 8 accumulators = 128 instructions per loop iteration

enum Op { ADD_U32, MIN_I32, PK_MIN_I16, PK_MAX_I16, PK_SUB_I16, PERM_B32, ALIGNBYTE, DOT4_U32_U8, MAD_U32_U24, LSHL_OR, MBCNT_LO, BFE_U32,
          AND_B32, XOR_B32, LSHLREV_B32, SUB_U32, MAX_I32, MIN_U32, CNDMASK_B32, MOV_B32, ADD3_U32, MUL_LO_U32, MUL_U32_U24, SAD_U8, MAX_F32, MAX3_F32, MUL_F32, ADD_F32, PK_ADD_U16, PK_MAD_I16, PK_LSHLREV_B16, CVT_F32_UBYTE0, MED3_I32, ADD_LSHL_U32,
          FMA_F32, PK_FMA_F32, PK_MIN_F16, PK_MAX_F16, PK_MAXIMUM3_F16, PK_MINIMUM3_F16, PK_ADD_F16, MAX3_I32, MIN3_I32, NUM_OPS };
static const char *k_names[NUM_OPS] = {"v_add_u32", "v_min_i32", "v_pk_min_i16", "v_pk_max_i16", "v_pk_sub_i16", "v_perm_b32", "v_alignbyte_b32",
                                       "v_dot4_u32_u8", "v_mad_u32_u24", "v_lshl_or_b32", "v_mbcnt_lo_u32_b32", "v_bfe_u32",
                                       "v_and_b32", "v_xor_b32", "v_lshlrev_b32", "v_sub_u32", "v_max_i32", "v_min_u32", "v_cndmask_b32", "v_mov_b32", "v_add3_u32",
                                       "v_mul_lo_u32", "v_mul_u32_u24", "v_sad_u8", "v_max_f32", "v_max3_f32", "v_mul_f32", "v_add_f32", "v_pk_add_u16", "v_pk_mad_i16",
                                       "v_pk_lshlrev_b16", "v_cvt_f32_ubyte0", "v_med3_i32", "v_add_lshl_u32",
                                       "v_fma_f32", "v_pk_fma_f32", "v_pk_min_f16", "v_pk_max_f16", "v_pk_maximum3_f16", "v_pk_minimum3_f16",
                                       "v_pk_add_f16", "v_max3_i32", "v_min3_i32"};

#define ONE(ins, a) asm volatile(ins : "+v"(a) : "v"(b), "v"(c));
#define EIGHT(ins) ONE(ins, a0) ONE(ins, a1) ONE(ins, a2) ONE(ins, a3) ONE(ins, a4) ONE(ins, a5) ONE(ins, a6) ONE(ins, a7)

template <int OP>
__global__ __launch_bounds__(256) void issue_kernel(unsigned *sink, long long *stamps, int iters, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 11u, a5 = a0 + 13u, a6 = a0 + 17u, a7 = a0 + 19u;
    unsigned b = seed | 0x01010101u, c = 0x07060504u ^ (seed << 8);
    // 64-bit accumulators for v_pk_fma_f32
    double d0 = (double)a0, d1 = (double)a1, d2 = (double)a2, d3 = (double)a3, d4 = (double)a4, d5 = (double)a5, d6 = (double)a6, d7 = (double)a7;
    double db = 1.0000001, dc = 1e-9;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (OP == ADD_U32) { EIGHT("v_add_u32 %0, %0, %1") }
            else if (OP == MIN_I32) { EIGHT("v_min_i32 %0, %0, %1") }
            else if (OP == PK_MIN_I16) { EIGHT("v_pk_min_i16 %0, %0, %1") }
            else if (OP == PK_MAX_I16) { EIGHT("v_pk_max_i16 %0, %0, %1") }
            else if (OP == PK_SUB_I16) { EIGHT("v_pk_sub_i16 %0, %0, %1") }
            else if (OP == PERM_B32) { EIGHT("v_perm_b32 %0, %0, %1, %2") }
            else if (OP == ALIGNBYTE) { EIGHT("v_alignbyte_b32 %0, %0, %1, 1") }
            else if (OP == DOT4_U32_U8) { EIGHT("v_dot4_u32_u8 %0, %1, %2, %0") }
            else if (OP == MAD_U32_U24) { EIGHT("v_mad_u32_u24 %0, %0, %1, %2") }
            else if (OP == LSHL_OR) { EIGHT("v_lshl_or_b32 %0, %0, 1, %1") }
            else if (OP == MBCNT_LO) { EIGHT("v_mbcnt_lo_u32_b32 %0, %1, %0") }
            else if (OP == BFE_U32) { EIGHT("v_bfe_u32 %0, %0, 1, 31") }
            else if (OP == AND_B32) { EIGHT("v_and_b32 %0, %0, %1") }
            else if (OP == XOR_B32) { EIGHT("v_xor_b32 %0, %0, %1") }
            else if (OP == LSHLREV_B32) { EIGHT("v_lshlrev_b32 %0, 1, %0") }
            else if (OP == SUB_U32) { EIGHT("v_sub_u32 %0, %0, %1") }
            else if (OP == MAX_I32) { EIGHT("v_max_i32 %0, %0, %1") }
            else if (OP == MIN_U32) { EIGHT("v_min_u32 %0, %0, %1") }
            else if (OP == CNDMASK_B32) { EIGHT("v_cndmask_b32 %0, %0, %1, vcc") }
            else if (OP == MOV_B32) { EIGHT("v_mov_b32 %0, %1") }
            else if (OP == ADD3_U32) { EIGHT("v_add3_u32 %0, %0, %1, %2") }
            else if (OP == MUL_LO_U32) { EIGHT("v_mul_lo_u32 %0, %0, %1") }
            else if (OP == MUL_U32_U24) { EIGHT("v_mul_u32_u24 %0, %0, %1") }
            else if (OP == SAD_U8) { EIGHT("v_sad_u8 %0, %1, %2, %0") }
            else if (OP == MAX_F32) { EIGHT("v_max_f32 %0, %0, %1") }
            else if (OP == MAX3_F32) { EIGHT("v_max3_f32 %0, %0, %1, %2") }
            else if (OP == MUL_F32) { EIGHT("v_mul_f32 %0, %0, %1") }
            else if (OP == ADD_F32) { EIGHT("v_add_f32 %0, %0, %1") }
            else if (OP == PK_ADD_U16) { EIGHT("v_pk_add_u16 %0, %0, %1") }
            else if (OP == PK_MAD_I16) { EIGHT("v_pk_mad_i16 %0, %0, %1, %2") }
            else if (OP == PK_LSHLREV_B16) { EIGHT("v_pk_lshlrev_b16 %0, 1, %0") }
            else if (OP == CVT_F32_UBYTE0) { EIGHT("v_cvt_f32_ubyte0 %0, %0") }
            else if (OP == MED3_I32) { EIGHT("v_med3_i32 %0, %0, %1, %2") }
            else if (OP == ADD_LSHL_U32) { EIGHT("v_add_lshl_u32 %0, %0, %1, 1") }
            else if (OP == FMA_F32) { EIGHT("v_fma_f32 %0, %0, %1, %2") }
            else if (OP == PK_MIN_F16) { EIGHT("v_pk_min_f16 %0, %0, %1") }
            else if (OP == PK_MAX_F16) { EIGHT("v_pk_max_f16 %0, %0, %1") }
            else if (OP == PK_MAXIMUM3_F16) { EIGHT("v_pk_maximum3_f16 %0, %0, %1, %2") }
            else if (OP == PK_MINIMUM3_F16) { EIGHT("v_pk_minimum3_f16 %0, %0, %1, %2") }
            else if (OP == PK_ADD_F16) { EIGHT("v_pk_add_f16 %0, %0, %1") }
            else if (OP == MAX3_I32) { EIGHT("v_max3_i32 %0, %0, %1, %2") }
            else if (OP == MIN3_I32) { EIGHT("v_min3_i32 %0, %0, %1, %2") }
            else if (OP == PK_FMA_F32) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d0) : "v"(db), "v"(dc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d1) : "v"(db), "v"(dc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d2) : "v"(db), "v"(dc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d3) : "v"(db), "v"(dc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d4) : "v"(db), "v"(dc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d5) : "v"(db), "v"(dc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d6) : "v"(db), "v"(dc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d7) : "v"(db), "v"(dc));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (OP == PK_FMA_F32) x ^= (unsigned)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    sink[blockIdx.x * 256 + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[4 * w + 0] = t0; stamps[4 * w + 1] = t1; stamps[4 * w + 2] = r0; stamps[4 * w + 3] = r1;
    }
}

struct Result { double cyc_per_inst, clock_mhz, ginst_per_s, wall_ms; };

template <int OP>
static Result run(int waves_per_simd, unsigned *d_sink, long long *d_stamps, int n_cu)
{
    const int blocks = n_cu * waves_per_simd, iters = 1024;
    std::vector<long long> st((size_t)blocks * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    Result best = {1e30, 0, 0, 0};
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(issue_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_sink, d_stamps, iters, 3u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(st.data(), d_stamps, sizeof(long long) * 16 * blocks, hipMemcpyDeviceToHost);
        // chip-wide span of the instruction streams in wall ticks (s_memrealtime, 100 MHz; the counter is shared by the chip) and the
        // shader clock each wave saw (its own cycles / its own ticks); waves of a SIMD do not all overlap for their whole life, so
        // a per-wave cycle count over-states the issue rate -- the span of ALL waves does not
        long long r_first = st[2], r_last = st[3];
        std::vector<double> clk;
        for (int w = 0; w < 4 * blocks; w++) {
            r_first = std::min(r_first, st[4 * w + 2]); r_last = std::max(r_last, st[4 * w + 3]);
            const double dc = (double)(st[4 * w + 1] - st[4 * w + 0]), dr = (double)(st[4 * w + 3] - st[4 * w + 2]);
            if (dr > 0) clk.push_back(dc / dr * 100.0);
        }
        std::sort(clk.begin(), clk.end());
        const double med_clk = clk.empty() ? 0.0 : clk[clk.size() / 2];
        const double span_s = (double)(r_last - r_first) * 1e-8;
        const double insts_total = (double)iters * UNROLL * 8 * 4.0 * blocks;
        const double rate = insts_total / span_s;                        // wave-instructions per second, whole chip
        const double cpi = (double)n_cu * 4.0 * med_clk * 1e6 / rate;    // shader cycles per wave-instruction per SIMD
        if (rep > 0 && cpi < best.cyc_per_inst) best = {cpi, med_clk, rate / 1e9, (double)ms};
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best;
}

template <int OP>
static void report(unsigned *d_sink, long long *d_stamps, int n_cu, bool last)
{
    printf("  \"%s\": {", k_names[OP]);
    const int occ[4] = {1, 2, 4, 8};
    for (int i = 0; i < 4; i++) {
        const Result r = run<OP>(occ[i], d_sink, d_stamps, n_cu);
        printf("\"%d\": {\"cycles_per_wave_inst_per_simd\": %.3f, \"clock_mhz\": %.0f, \"chip_G_wave_inst_per_s\": %.1f, \"wall_ms\": %.3f}%s",
               occ[i], r.cyc_per_inst, r.clock_mhz, r.ginst_per_s, r.wall_ms, i < 3 ? ", " : "");
    }
    printf("}%s\n", last ? "" : ",");
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int n_cu = prop.multiProcessorCount;
    unsigned *d_sink; long long *d_stamps;
    hipMalloc(&d_sink, (size_t)n_cu * 8 * 256 * 4);
    hipMalloc(&d_stamps, (size_t)n_cu * 8 * 16 * sizeof(long long));
    printf("{\n \"device\": \"%s\", \"compute_units\": %d, \"method\": \"independent inline-asm instruction streams, 8 accumulators, %d instructions per wave; "
           "keys 1/2/4/8 = waves per SIMD on every CU; rate = all instructions / (last end - first start) in s_memrealtime ticks (100 MHz), clock = median of d(s_memtime)/d(s_memrealtime) x 100 MHz per wave, cycles_per_wave_inst_per_simd = CUs x 4 x clock / rate; not under rocprofv3\",\n \"ops\": {\n",
           prop.name, n_cu, 1024 * UNROLL * 8);
    report<ADD_U32>(d_sink, d_stamps, n_cu, false);
    report<MIN_I32>(d_sink, d_stamps, n_cu, false);
    report<PK_MIN_I16>(d_sink, d_stamps, n_cu, false);
    report<PK_MAX_I16>(d_sink, d_stamps, n_cu, false);
    report<PK_SUB_I16>(d_sink, d_stamps, n_cu, false);
    report<PERM_B32>(d_sink, d_stamps, n_cu, false);
    report<ALIGNBYTE>(d_sink, d_stamps, n_cu, false);
    report<DOT4_U32_U8>(d_sink, d_stamps, n_cu, false);
    report<MAD_U32_U24>(d_sink, d_stamps, n_cu, false);
    report<LSHL_OR>(d_sink, d_stamps, n_cu, false);
    report<MBCNT_LO>(d_sink, d_stamps, n_cu, false);
    report<BFE_U32>(d_sink, d_stamps, n_cu, false);
    report<MAX3_I32>(d_sink, d_stamps, n_cu, false);
    report<MIN3_I32>(d_sink, d_stamps, n_cu, false);
    report<AND_B32>(d_sink, d_stamps, n_cu, false);
    report<XOR_B32>(d_sink, d_stamps, n_cu, false);
    report<LSHLREV_B32>(d_sink, d_stamps, n_cu, false);
    report<SUB_U32>(d_sink, d_stamps, n_cu, false);
    report<MAX_I32>(d_sink, d_stamps, n_cu, false);
    report<MIN_U32>(d_sink, d_stamps, n_cu, false);
    report<CNDMASK_B32>(d_sink, d_stamps, n_cu, false);
    report<MOV_B32>(d_sink, d_stamps, n_cu, false);
    report<ADD3_U32>(d_sink, d_stamps, n_cu, false);
    report<MUL_LO_U32>(d_sink, d_stamps, n_cu, false);
    report<MUL_U32_U24>(d_sink, d_stamps, n_cu, false);
    report<SAD_U8>(d_sink, d_stamps, n_cu, false);
    report<MAX_F32>(d_sink, d_stamps, n_cu, false);
    report<MAX3_F32>(d_sink, d_stamps, n_cu, false);
    report<MUL_F32>(d_sink, d_stamps, n_cu, false);
    report<ADD_F32>(d_sink, d_stamps, n_cu, false);
    report<PK_ADD_U16>(d_sink, d_stamps, n_cu, false);
    report<PK_MAD_I16>(d_sink, d_stamps, n_cu, false);
    report<PK_LSHLREV_B16>(d_sink, d_stamps, n_cu, false);
    report<CVT_F32_UBYTE0>(d_sink, d_stamps, n_cu, false);
    report<MED3_I32>(d_sink, d_stamps, n_cu, false);
    report<ADD_LSHL_U32>(d_sink, d_stamps, n_cu, false);
    report<FMA_F32>(d_sink, d_stamps, n_cu, false);
    report<PK_FMA_F32>(d_sink, d_stamps, n_cu, false);
    report<PK_MIN_F16>(d_sink, d_stamps, n_cu, false);
    report<PK_MAX_F16>(d_sink, d_stamps, n_cu, false);
    report<PK_ADD_F16>(d_sink, d_stamps, n_cu, false);
    report<PK_MAXIMUM3_F16>(d_sink, d_stamps, n_cu, false);
    report<PK_MINIMUM3_F16>(d_sink, d_stamps, n_cu, true);
    printf(" }\n}\n");
    return 0;
}
