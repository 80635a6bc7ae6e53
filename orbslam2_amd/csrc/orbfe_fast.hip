// orbfe_fast.hip -- cell-wise cv::FAST with two thresholds (src/ORBextractor.cc:783-823) + quadtree bucket accumulation.
#include "orbfe_common.hpp"
#include "orbfe_blur_wave.hpp"

// ---------------------------------------------------------------------------
// FAST-9/16 per cell: score map + 3x3 NMS inside the cell + two-threshold select
// ---------------------------------------------------------------------------
__device__ __forceinline__ int fast_score16(const uint8_t *t, int pitch, int minth)
{
    const int v = t[0];
    int d[16];
    d[0] = v - t[3 * pitch];       d[1] = v - t[3 * pitch + 1];   d[2] = v - t[2 * pitch + 2];   d[3] = v - t[pitch + 3];
    d[4] = v - t[3];               d[5] = v - t[-pitch + 3];      d[6] = v - t[-2 * pitch + 2];  d[7] = v - t[-3 * pitch + 1];
    d[8] = v - t[-3 * pitch];      d[9] = v - t[-3 * pitch - 1];  d[10] = v - t[-2 * pitch - 2]; d[11] = v - t[-pitch - 3];
    d[12] = v - t[-3];             d[13] = v - t[pitch - 3];      d[14] = v - t[2 * pitch - 2];  d[15] = v - t[3 * pitch - 1];
    int mn2[16], mx2[16], mn4[16], mx4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn2[k] = min(d[k], d[(k + 1) & 15]); mx2[k] = max(d[k], d[(k + 1) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; k++) { mn4[k] = min(mn2[k], mn2[(k + 2) & 15]); mx4[k] = max(mx2[k], mx2[(k + 2) & 15]); }
    int a = -512, b = 512;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int mn9 = min(min(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
        const int mx9 = max(max(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
        a = max(a, mn9);
        b = min(b, mx9);
    }
    return max(minth, max(a, -b)) - 1;
}

// One wave (64-thread workgroup) per FAST cell (src/ORBextractor.cc:783-810): no inter-wave barriers,
// and wave-ordered ballot compaction keeps every queue in row-major order, which is cv::FAST's
// emission order.  Phases:
//  0  aligned 32-bit loads of the (w+6)x(h+6) cell tile into LDS;
//  A  a necessary test on the four even opposite ring pairs for every interior pixel (a 9-arc holds one pixel of every
//     opposite pair) -> ordered queue (+ a side list for pixels that pass both polarities)
//  C  exact threshold-independent score (closed form of cornerScore<16>) for the queue
//  D  strict 3x3 NMS inside the cell interior; iniThFAST, or minThFAST if that leaves nothing
//  E  ordered compaction of the survivors into the cell's slot.
__device__ __forceinline__ void load_ring(const uint8_t *t, int pitch, int d[16])
{
    const int v = t[0];
    d[0] = v - t[3 * pitch];       d[1] = v - t[3 * pitch + 1];   d[2] = v - t[2 * pitch + 2];   d[3] = v - t[pitch + 3];
    d[4] = v - t[3];               d[5] = v - t[-pitch + 3];      d[6] = v - t[-2 * pitch + 2];  d[7] = v - t[-3 * pitch + 1];
    d[8] = v - t[-3 * pitch];      d[9] = v - t[-3 * pitch - 1];  d[10] = v - t[-2 * pitch - 2]; d[11] = v - t[-pitch - 3];
    d[12] = v - t[-3];             d[13] = v - t[pitch - 3];      d[14] = v - t[2 * pitch - 2];  d[15] = v - t[3 * pitch - 1];
}

// number of set bits of m below this lane, plus acc (v_mbcnt_lo/hi accumulate form)
__device__ __forceinline__ unsigned mbcnt64(unsigned long long m, unsigned acc)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, acc));
}

// Two adjacent bytes P, P+1 of a 12-byte row window (w[0] | w[1] | w[2]), zero-extended into the two
// 16-bit halves of a register: one v_perm_b32 with a constant selector (0x0c selects 0x00).
template <int P> __device__ __forceinline__ pk16 row_pair(const unsigned (&w)[3])
{
    if constexpr (P + 1 <= 7)
        return __builtin_bit_cast(pk16, __builtin_amdgcn_perm(w[1], w[0], (unsigned)P | 0x0c00u | ((unsigned)(P + 1) << 16) | 0x0c000000u));
    else
        return __builtin_bit_cast(pk16, __builtin_amdgcn_perm(w[2], w[1], (unsigned)(P - 4) | 0x0c00u | ((unsigned)(P - 3) << 16) | 0x0c000000u));
}

// x + (bit `lane` of m): one v_addc_co_u32 with the ballot as carry-in (the compiler's own `x += pred` is a select plus an add)
__device__ __forceinline__ int add_lane_bit(int x, unsigned long long m)
{
    int r;
    unsigned long long c;
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r), "=s"(c) : "v"(x), "s"(m));
    return r;
}

// ballots of "the low / high 16-bit half is negative", one compare each (spelled in C they cost an extraction per low half, and
// a ballot plus an `if` on the same condition is lowered twice), and a 16-bit LDS store under such a ballot
__device__ __forceinline__ unsigned long long neg_lo16(unsigned v) { unsigned long long m; asm("v_cmp_gt_i16_e64 %0, 0, %1" : "=s"(m) : "v"(v)); return m; }
__device__ __forceinline__ unsigned long long neg_hi16(unsigned v) { unsigned long long m; asm("v_cmp_gt_i32_e64 %0, 0, %1" : "=s"(m) : "v"(v)); return m; }
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void *)p; }
__device__ __forceinline__ void lds_store_lo16(unsigned long long m, unsigned addr, unsigned v)
{
    unsigned long long save;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b16 %2, %3\n\ts_or_b64 exec, exec, %0" : "=&s"(save) : "s"(m), "v"(addr), "v"(v) : "memory", "scc");
}
__device__ __forceinline__ void lds_store_hi16(unsigned long long m, unsigned addr, unsigned v)
{
    unsigned long long save;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b16_d16_hi %2, %3\n\ts_or_b64 exec, exec, %0" : "=&s"(save) : "s"(m), "v"(addr), "v"(v) : "memory", "scc");
}

// ---- packed half-precision min / max for the exact score (phase C) ------------------------------------------------------
// gfx950 issues every min / max opcode -- integer, packed, float alike -- once per ~4 cycles per SIMD, and a few plain ops (add, sub,
// and, xor, mov, fp32 fma) once per ~2 (profiles/r02_valu_peak.json).  The packed opcode that does more per issue slot is
// CDNA4's three-operand v_pk_maximum3_f16 / v_pk_minimum3_f16: two comparisons in each 16-bit half.  A pixel p becomes the
// half-precision number 1024 + p by XOR-ing 0x6400 into its zero-extended 16-bit lane (integers up to 2048 are exact in f16, and
// so is every difference formed below) -- the XOR that sign-normalises the entry anyway, so the conversion costs nothing.
// Inline asm: the f16 min / max builtins make the compiler quiet signalling NaNs first (an extra op per operand).
__device__ __forceinline__ unsigned h2min(unsigned a, unsigned b) { unsigned d; asm("v_pk_min_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ unsigned h2max(unsigned a, unsigned b) { unsigned d; asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ unsigned h2min3(unsigned a, unsigned b, unsigned c) { unsigned d; asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ unsigned h2max3(unsigned a, unsigned b, unsigned c) { unsigned d; asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ unsigned h2add(unsigned a, unsigned b) { unsigned d; asm("v_pk_add_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ unsigned h2sub(unsigned a, unsigned b) { unsigned d; asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }

// TP = tile pitch in bytes as a compile-time constant (0: run-time value): with it every ring / row offset folds
// into the immediate offset field of the LDS instructions instead of costing address VALU.
// BK: also accumulate the quadtree bucket counts / best keys of the survivors (orbfe_octree3.hip) -- aggregated per
// cell in LDS, then stored to the cell's own entries of DeviceBuffers::bk_part.
// the waves of a workgroup are independent; a wave's own LDS traffic only needs its outstanding LDS operations retired
#define FAST_WAVE_SYNC() do { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); } while (0)
template <int TP, bool BK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void fast_cell_kernel(DeviceConfig cfg, DeviceBuffers buf, int n_images, int tile_pitch_rt, int tile_bytes, int sc_bytes, int q_bytes, int lds_per_wave ORBFE_CUT_PARAM)
{
    const int tile_pitch = TP ? TP : tile_pitch_rt;
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem_all[];
    // XCD-aware block -> (image, cell) map (same scheme as describe_kernel): consecutive cells of one image run
    // on one XCD, so the 128-B lines that horizontally / vertically adjacent cell tiles share (a 37-row tile
    // uses ~44 B of each line) are served by that XCD's L2 instead of being re-fetched from HBM by 8 XCDs.
    // A workgroup is four independent waves = four consecutive cells (no workgroup barriers: FAST_WAVE_SYNC): horizontally
    // adjacent cells share the 128-B lines of their tile rows, and on one CU those lines are fetched from L2 once.
    const int bpi_cells = (cfg.cells_total + 3) >> 2;
    const int bpi = bpi_cells + ((cfg.blur_tiles_total - cfg.fast_blur_t0 + 3) >> 2);
    int img, blk;
    if (!xcd_map_magic(bpi, n_images, cfg.xcd_magic, img, blk)) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (__builtin_expect(blk >= bpi_cells, 0)) { // the image's last workgroups: blur tiles riding in this launch
        const int u = cfg.fast_blur_t0 + (blk - bpi_cells) * 4 + wave;
        if (u < cfg.blur_tiles_total) blur_wave(cfg, buf, img, u);
        return;
    }
    const int cell = blk * 4 + wave;
    if (cell >= cfg.cells_total) return;
    uint8_t *s_mem = s_mem_all + wave * lds_per_wave;
    // the cell's level, position and clipped tile size from the host-built table (one scalar load instead of the level search,
    // a division and the clipping of src/ORBextractor.cc:783-800 behind a chain of dependent scalar loads)
    const uint4 cinfo = const_load_u32x4(buf.cell_info + cell);
    const uint32_t aux_g = const_load_u32((const uint32_t *)buf.cell_aux + 2 * cell), aux_c = const_load_u32((const uint32_t *)buf.cell_aux + 2 * cell + 1); // host-divided lane maps
    // phase A's per-lane constants of this cell's shape, built on the host (35 vector instructions per wave otherwise): issued now, used after the staging
    const uint32_t *lane_tab = buf.fast_lane_tab + ((aux_g & 0x1ffffu) << 9); // uniform
    const uint4 lt_a = *(const uint4 *)(lane_tab + ((threadIdx.x & 63) << 3));
    const uint2 lt_b = *(const uint2 *)(lane_tab + ((threadIdx.x & 63) << 3) + 4);
    const int level = (int)(cinfo.x & 0xffu);
    const LevelInfo &L = cfg.lv[level];
    const int ci = (int)cinfo.w;
    const int lane = threadIdx.x & 63;
    int *cnt_out = buf.cell_cnt + (size_t)img * cfg.cells_total + cell;
    if (!(cinfo.x & 0x100u)) { // no FAST call for this cell (src/ORBextractor.cc:788-798)
        if (lane == 0) *cnt_out = 0;
        return;
    }
    const int ini_x = (int)(cinfo.y & 0xffffu), ini_y = (int)(cinfo.y >> 16);
    const int max_x = ini_x + (int)(cinfo.z & 0xffu);
    const int tw = (int)(cinfo.z & 0xffu), th = (int)((cinfo.z >> 8) & 0xffu);
    const int iw = tw - 6, ih = th - 6;
    const int cell_x0 = ini_x - cfg.min_border, cell_y0 = ini_y - cfg.min_border; // j * wCell, i * hCell
    // bucket tables of this cell's columns / rows (BK): issued now, consumed in phase E
    unsigned tabx = 0u, taby = 0u, bk_off = ~0u;
    if (BK) {
        bk_off = const_load_u32(buf.bk_off + cell);
        const uint32_t *bx_tab = buf.bk_tab + L.bk_xoff + 3 + cell_x0; // survivor x = c + 3 + j * wCell
        const uint32_t *by_tab = buf.bk_tab + L.bk_yoff + 3 + cell_y0;
        // (scalar base, 32-bit lane byte offset) addressing: no 64-bit vector add
        tabx = *(const uint32_t *)((const uint8_t *)bx_tab + min((unsigned)lane << 2, (unsigned)(iw - 1) << 2));
        taby = *(const uint32_t *)((const uint8_t *)by_tab + min((unsigned)lane << 2, (unsigned)(ih - 1) << 2));
    }
    // LDS layout (sizes fixed by the host from the largest cell): tile | scores | queue.  Queue 2 is
    // compacted in place over queue 1 (writes never pass the read cursor); the per-entry flags of
    // phase D reuse the tile, which is dead after phase C.  5.2 KB per wave at KITTI's cell size: seven workgroups per CU, which is also what
    // the kernel's 70 VGPRs allow (round 5; 72 VGPRs and six until the lane maps moved to host tables).
    uint8_t *s_tile = s_mem;                           // [th][tile_pitch], column 0 = pixel xa (4-aligned)
    uint8_t *s_sc = s_mem + tile_bytes;                // [(ih+2)][(iw+2)], zero border
    uint16_t *s_q1 = (uint16_t *)(s_sc + sc_bytes);    // packed (r << 8 | c), row-major ascending
    uint16_t *s_q2 = s_q1;
    uint8_t *s_qf = s_tile;                            // per queue-2 entry: 0 / 1 (local max, >= minTh) / 2 (>= iniTh)
    const int scp = iw + 2;

    // tile column 0 is pixel ini_x: the staging loads are UNALIGNED dwords (free on this memory system), so interior pixel c
    // always sits at tile byte c + 3 and a cell of up to 32 columns is eight 4-pixel groups per row whatever ini_x & 3 is
    const int xa = ini_x, ox = 0;
    const int wpr = (max_x - xa + 3) >> 2; // words per tile row
    int src_pitch;
    const uint8_t *src = level_image(cfg, buf, img, level, src_pitch) + (size_t)ini_y * src_pitch + xa; // level 0 may be the caller's packed image (any row alignment: the staging loads are unaligned anyway)
    // The reference runs FAST at iniThFAST and, only when that yields no keypoint in the cell, again at minThFAST
    // (src/ORBextractor.cc:803-810).  Same here: the first attempt queues and scores only what passes the quick test at
    // iniTh (about 0.6 of what passes at minTh on the benchmark images, so phases C-E shrink accordingly); a cell without a
    // survivor re-stages its tile (phase D may have overwritten it with flags) and repeats everything at minTh.  Both
    // attempts compute exactly cv::FAST(cell, t, true) for their t: scores are threshold-free and a neighbour below t can
    // never suppress a corner at t.
    int t = cfg.ini_th;
    int n2 = 0;
    unsigned rcq[4];
    int fq[4];
    for (int attempt = 0; attempt < 2; attempt++) {
    n2 = 0;
    {
        // 16-byte chunks (unaligned 128-bit loads are fine on this memory system; the LDS side is aligned: the tile pitch is a
        // multiple of 16): lane = (row of the pass, chunk), 64 / cpr rows per load instruction -- two loads for a 37-row tile
        // of three chunks instead of twelve dword loads.  Lanes beyond the last whole row of a pass and rows beyond the tile
        // repeat an element (same value to the same LDS bytes), so nothing is predicated.  The last chunk of a row may read up
        // to 15 bytes past the tile (inside the pyramid row's margin, or -- level 0 read in place -- the first bytes of the next row: a
        // tile ends >= 10 rows above the image's last row) into LDS bytes no pixel test uses.
        const int cpr = (wpr + 3) >> 2;
        const int rpi = (int)((aux_c >> 17) & 0x7fu);                       // 64 / cpr
        int rl0 = (int)(__umul24((unsigned)lane, aux_c & 0x1ffffu) >> 16);  // lane / cpr
        rl0 = rl0 < rpi ? rl0 : rpi - 1;
        const int ch16 = (lane - rl0 * cpr < cpr ? lane - rl0 * cpr : 0) << 4;
        for (int rb = 0; rb < th; rb += 2 * rpi) { // two loads in flight per lane
            uint4 v[2];
            int dst[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                int r = rb + u * rpi + rl0;
                r = r < th ? r : th - 1;
                dst[u] = __mul24(r, tile_pitch) + ch16;
                v[u] = load16_unaligned(src + (unsigned)(__mul24(r, src_pitch) + ch16));
            }
#pragma unroll
            for (int u = 0; u < 2; u++) *(uint4 *)(s_tile + dst[u]) = v[u];
        }
    }
    for (int i = lane; i < sc_bytes / 16; i += 64) ((uint4 *)s_sc)[i] = make_uint4(0u, 0u, 0u, 0u); // sc_bytes is a multiple of 16
    FAST_WAVE_SYNC();
    if (ORBFE_CUT(1)) { if (lane == 0) *cnt_out = 0; return; }
    // Phases A and C work on TWO pixels per lane, one in each 16-bit half of a register, with packed
    // v_pk_{sub,min,max}_i16 (ring differences are in [-255, 255]): integer VALU issue is what bounds
    // this kernel (measured ~1 wave64 instruction / cycle / CU), so instructions are what is saved.
    int roff[16]; // ring offsets inside the LDS tile (uniform)
    roff[0] = 3 * tile_pitch;      roff[1] = 3 * tile_pitch + 1;   roff[2] = 2 * tile_pitch + 2;   roff[3] = tile_pitch + 3;
    roff[4] = 3;                   roff[5] = -tile_pitch + 3;      roff[6] = -2 * tile_pitch + 2;  roff[7] = -3 * tile_pitch + 1;
    roff[8] = -3 * tile_pitch;     roff[9] = -3 * tile_pitch - 1;  roff[10] = -2 * tile_pitch - 2; roff[11] = -tile_pitch - 3;
    roff[12] = -3;                 roff[13] = tile_pitch - 3;      roff[14] = 2 * tile_pitch - 2;  roff[15] = 3 * tile_pitch - 1;
    // ---- C (defined first: phase A calls it when its side list fills up): exact score for the entry's polarity: max over the
    //      16 arcs of the min of the arc's 9 differences v - r_k = v' - min over arcs of (max of the arc's ring values); the
    //      maximum of a 9-arc is the maximum of three 3-arc maxima, so the whole score is 16 + 16 three-way maxima and 8 three-way
    //      minima (v_pk_maximum3_f16 / v_pk_minimum3_f16, two queue entries per lane, one in each half) instead of 80 two-way
    //      ones.  Bright entries are negated (centre and ring), which maps their score onto the dark formula; both the negation
    //      and the byte -> f16 conversion are ONE xor.  Entries: c | (r + 1) << 8 | bright << 15. ----
    const unsigned tt16 = attempt == 0 ? cfg.ini_th_h2 : cfg.min_th_h2; // t as a half-precision number in both halves (from the host: a conversion and a quarter-rate multiply per wave otherwise)
    auto score_entries = [&](int count, auto fetch) {
        for (int q0 = 0; q0 < count; q0 += 128) {
            const int qa = q0 + lane, qb = q0 + 64 + lane;
            const bool va = qa < count, vb = qb < count;
            const unsigned ea = fetch(va ? qa : 0), eb = fetch(vb ? qb : 0);
            const int ra = (ea >> 8) & 63, ca = ea & 255, rb = (eb >> 8) & 63, cb = eb & 255; // ra, rb = row + 1
            // base = the ring's lowest address (row - 3, column - 3), so every ring offset is a non-negative immediate
            const uint8_t *pa = &s_tile[(ra - 1) * tile_pitch + ca + ox];
            const uint8_t *pb = &s_tile[(rb - 1) * tile_pitch + cb + ox];
            // 0x6400: p -> 1024 + p; 0x8000 more for a bright entry: -> -(1024 + p)
            const unsigned fx = 0x64006400u ^ ((ea & 0x8000u) ? 0x8000u : 0u) ^ ((eb & 0x8000u) ? 0x80000000u : 0u);
            // one v_perm_b32 packs the two bytes into the two halves, one full-rate v_xor_b32 converts / negates (spelled this
            // way because the compiler otherwise picks a shift + a three-input or: two half-rate ops)
            auto pack = [&](unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x0c040c00u) ^ fx; };
            const int ctr = 3 * tile_pitch + 3;
            const unsigned vv = pack(pa[ctr], pb[ctr]);
            unsigned e[16];
#pragma unroll
            for (int k = 0; k < 16; k++) e[k] = pack(pa[ctr + roff[k]], pb[ctr + roff[k]]);
            unsigned m3[16], m9[16];
#pragma unroll
            for (int k = 0; k < 16; k++) m3[k] = h2max3(e[k], e[(k + 1) & 15], e[(k + 2) & 15]);
#pragma unroll
            for (int k = 0; k < 16; k++) m9[k] = h2max3(m3[k], m3[(k + 3) & 15], m3[(k + 6) & 15]);
            unsigned worst = h2min3(m9[0], m9[1], m9[2]);
            worst = h2min3(worst, m9[3], m9[4]); worst = h2min3(worst, m9[5], m9[6]); worst = h2min3(worst, m9[7], m9[8]);
            worst = h2min3(worst, m9[9], m9[10]); worst = h2min3(worst, m9[11], m9[12]); worst = h2min3(worst, m9[13], m9[14]);
            worst = h2min(worst, m9[15]);
            // best = max(v' - worst, t): an exact f16 integer in [t, 255]; + 1023 puts score = best - 1 into the low mantissa bits
            const unsigned best = h2max(h2sub(vv, worst), tt16);
            const unsigned sc2 = h2add(best, 0x63fe63feu);
            const int sa = (int)(sc2 & 0xffu), sb = (int)((sc2 >> 16) & 0xffu);
            if (va && sa >= t) s_sc[ra * scp + ca + 1] = (uint8_t)sa;
            if (vb && sb >= t) s_sc[rb * scp + cb + 1] = (uint8_t)sb;
        }
    };
    // ---- A: a quick NECESSARY test for every interior pixel on the four EVEN opposite ring pairs (0/8, 2/10, 4/12, 6/14): a dark
    //      (bright) 9-arc holds one pixel of every opposite pair, so it needs one darker (brighter) pixel in each of the four.
    //      cv::FAST's own quick test uses all eight pairs; half of them admit 15 % more pixels on the benchmark images (all of
    //      which the exact score of phase C then rejects) for half the extraction and min / max work, and only tile rows
    //      0, 1, 3, 5, 6 of a pixel's 7-row window are read.  With four pairs a pixel can pass BOTH polarities and still be a
    //      corner of one of them (with eight it cannot): it is queued as bright, and once more as dark on a side list that phase
    //      C scores too but phases D / E never see (0.2 % of the pixels).  Survivors are queued in row-major order.
    //      A lane owns FOUR horizontally adjacent pixels whose tile bytes are 3..6 of a 12-byte window (3 aligned LDS words per
    //      ring row): every ring column x-3..x+3 of the four pixels lies inside the window, so each ring position is two
    //      v_perm_b32 with constant selectors, and the test runs on raw ring values (with d = v - r:
    //      min_k max(d_k, d_k+8) = v - max_k min(r_k, r_k+8)); a zero-extended byte is also a half-precision denormal that
    //      orders like the byte, so the two three-way reductions are one v_pk_maximum3_f16 / v_pk_minimum3_f16 each. ----
    uint16_t *s_xq = (uint16_t *)(s_mem + tile_bytes + sc_bytes + q_bytes); // side list, 256 entries (phase E's accumulators later)
    int nx = 0;
    {
        // lane -> (row rl of the iteration's band, group jg) with jg FIXED per lane: a band is dr = 64 / ng whole rows (ng <= 16
        // because FAST cells are < 60 px wide, src/ORBextractor.cc:766-775; 8-10 for every level of the usual cameras, so 94-100 %
        // of the lanes work), which makes the column range checks, the entry's column part and the LDS column offset loop
        // invariants and leaves one add each for the address and the entries per iteration
        const int dr = (int)((aux_g >> 17) & 0x7fu); // 64 / ng, ng = (iw + 3) >> 2 groups per interior row; group j covers c = 4 j .. 4 j + 3
        const int pw = tile_pitch >> 2;
        const pk16 tp = {(short)t, (short)t};
        // From fast_lane_tab (orbfe_api.hip builds it with these formulas): rl = lane / ng, jg = lane - rl * ng, c0 = 4 jg; the lane works when rl < dr;
        // vm01 / vm23: which of the lane's four pixels exist, as sign bits of the halves (a ballot of `value & mask != 0` is one compare; a
        // ballot of a boolean expression is a select plus a compare); vl01 / vl23: the same in the last band, which the cell's bottom may cut;
        // e01: entries c | (r + 1) << 8 of pixels 0 1 in the two halves (the row bias keeps the word positive)
        unsigned vm01 = lt_a.x, vm23 = lt_a.y;
        const unsigned vl01 = lt_a.z, vl23 = lt_a.w;
        const int r0_last = (int)(aux_g >> 24); // ((ih - 1) / dr) * dr
        unsigned e01 = lt_b.x;
        const unsigned e_step = (unsigned)(dr << 8) * 0x10001u;
        const uint32_t *tw4 = (const uint32_t *)s_tile + lt_b.y; // rl * pw + jg
        for (int r0 = 0; r0 < ih; r0 += dr, tw4 += dr * pw, e01 += e_step) {
            if (r0 == r0_last) { vm01 = vl01; vm23 = vl23; } // rows past the cell read LDS beyond the tile (still this wave's region)
            unsigned w0[3], w1[3], w3[3], w5[3], w6[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                w0[i] = tw4[i]; w1[i] = tw4[pw + i]; w3[i] = tw4[3 * pw + i]; w5[i] = tw4[5 * pw + i]; w6[i] = tw4[6 * pw + i];
            }
            pk16 n01[4], x01[4], n23[4], x23[4];
#define FAST_PAIR(i, wa, sa, wb, sb)                                                                              \
    {                                                                                                             \
        const pk16 a01 = row_pair<sa>(wa), a23 = row_pair<sa + 2>(wa);                                           \
        const pk16 b01 = row_pair<sb>(wb), b23 = row_pair<sb + 2>(wb);                                           \
        n01[i] = __builtin_elementwise_min(a01, b01); x01[i] = __builtin_elementwise_max(a01, b01);               \
        n23[i] = __builtin_elementwise_min(a23, b23); x23[i] = __builtin_elementwise_max(a23, b23);               \
    }
            // ring pairs (k, k+8): (dx, dy) -> window start byte 3 + dx, tile row 3 + dy
            FAST_PAIR(0, w6, 3, w0, 3)  // ( 0, 3) / ( 0,-3)
            FAST_PAIR(1, w5, 5, w1, 1)  // ( 2, 2) / (-2,-2)
            FAST_PAIR(2, w3, 6, w3, 0)  // ( 3, 0) / (-3, 0)
            FAST_PAIR(3, w1, 5, w5, 1)  // ( 2,-2) / (-2, 2)
#undef FAST_PAIR
            auto u = [](pk16 v) { return __builtin_bit_cast(unsigned, v); };
            auto k = [](unsigned v) { return __builtin_bit_cast(pk16, v); };
            const pk16 A01 = __builtin_elementwise_max(k(h2max3(u(n01[0]), u(n01[1]), u(n01[2]))), n01[3]);
            const pk16 A23 = __builtin_elementwise_max(k(h2max3(u(n23[0]), u(n23[1]), u(n23[2]))), n23[3]);
            const pk16 B01 = __builtin_elementwise_min(k(h2min3(u(x01[0]), u(x01[1]), u(x01[2]))), x01[3]);
            const pk16 B23 = __builtin_elementwise_min(k(h2min3(u(x23[0]), u(x23[1]), u(x23[2]))), x23[3]);
            const pk16 v01 = row_pair<3>(w3), v23 = row_pair<5>(w3);
            // sign bit of a half: dq = A + t - v < 0 <=> v - A > t (dark), bq = v + t - B < 0 <=> B - v > t (bright)
            const unsigned dq01 = u((A01 + tp) - v01), bq01 = u((v01 + tp) - B01);
            const unsigned dq23 = u((A23 + tp) - v23), bq23 = u((v23 + tp) - B23);
            const unsigned o01 = (dq01 | bq01) & vm01, o23 = (dq23 | bq23) & vm23;
            const unsigned long long m0 = neg_lo16(o01), m1 = neg_hi16(o01), m2 = neg_lo16(o23), m3 = neg_hi16(o23);
            if ((m0 | m1 | m2 | m3) == 0ull) continue; // no pixel of the band passes (a scalar test): nothing to queue -- the flat regions of real images
            // bright flag straight from bq's sign bits
            const unsigned q01 = (bq01 & 0x80008000u) | e01, q23 = (bq23 & 0x80008000u) | (e01 + 0x20002u);
            const int pos0 = n2 + (int)mbcnt64(m3, mbcnt64(m2, mbcnt64(m1, mbcnt64(m0, 0u))));
            const int pos1 = add_lane_bit(pos0, m0), pos2 = add_lane_bit(pos1, m1), pos3 = add_lane_bit(pos2, m2);
            lds_store_lo16(m0, lds_addr(s_q2 + pos0), q01);
            lds_store_hi16(m1, lds_addr(s_q2 + pos1), q01);
            lds_store_lo16(m2, lds_addr(s_q2 + pos2), q23);
            lds_store_hi16(m3, lds_addr(s_q2 + pos3), q23);
            n2 += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
            // both polarities passed: once more, as dark, on the side list (rare: uniform branch)
            const unsigned xb01 = dq01 & bq01 & vm01, xb23 = dq23 & bq23 & vm23;
            if (__builtin_amdgcn_ballot_w64((xb01 | xb23) != 0u) != 0ull) {
                const bool x0 = (xb01 & 0x8000u) != 0u, x1 = (int)xb01 < 0, x2 = (xb23 & 0x8000u) != 0u, x3 = (int)xb23 < 0;
                const unsigned long long y0 = __builtin_amdgcn_ballot_w64(x0), y1 = __builtin_amdgcn_ballot_w64(x1), y2 = __builtin_amdgcn_ballot_w64(x2), y3 = __builtin_amdgcn_ballot_w64(x3);
                const int add = __popcll(y0) + __popcll(y1) + __popcll(y2) + __popcll(y3);
                if (nx + add > 256) { // side list full: score what it holds now
                    FAST_WAVE_SYNC();
                    score_entries(nx, [&](int q) { return (unsigned)s_xq[q]; });
                    FAST_WAVE_SYNC();
                    nx = 0;
                }
                int xp = nx + (int)mbcnt64(y3, mbcnt64(y2, mbcnt64(y1, mbcnt64(y0, 0u))));
                if (x0) s_xq[xp] = (uint16_t)(q01 & 0x7fffu);
                xp += x0;
                if (x1) s_xq[xp] = (uint16_t)((q01 >> 16) & 0x7fffu);
                xp += x1;
                if (x2) s_xq[xp] = (uint16_t)(q23 & 0x7fffu);
                xp += x2;
                if (x3) s_xq[xp] = (uint16_t)((q23 >> 16) & 0x7fffu);
                nx += add;
            }
        }
    }
    FAST_WAVE_SYNC();
    if (ORBFE_CUT(3)) { if (lane == 0) *cnt_out = 0; return; }
    // the side list goes behind the ordered queue when it fits (it always does on real images: the queue has room for every
    // pixel of the cell), so one call scores both; phases D / E only walk the first n2 entries
    int n_sc = n2;
    if (nx) {
        if (n2 + nx <= (q_bytes >> 1)) {
            for (int i = lane; i < nx; i += 64) s_q2[n2 + i] = s_xq[i];
            n_sc = n2 + nx;
            FAST_WAVE_SYNC();
        } else {
            score_entries(nx, [&](int q) { return (unsigned)s_xq[q]; });
        }
    }
    score_entries(n_sc, [&](int q) { return (unsigned)s_q2[q]; });
    FAST_WAVE_SYNC();
    if (ORBFE_CUT(4)) { if (lane == 0) *cnt_out = 0; return; }
    // ---- D: NMS + threshold choice.  The first 256 queue entries (all of them for ordinary cells) keep their flag and
    //      coordinates in registers for the compaction of phase E; later ones go through the LDS flag array. ----
    bool any = false;
    auto nms_flag = [&](unsigned rc) {
        const uint8_t *p = &s_sc[(rc >> 8) * scp + (rc & 255) + 1]; // rc >> 8 = row + 1
        const int s = p[0];
        // all nine reads issued together (a short-circuit chain would be nine dependent LDS round trips)
        const int n0 = p[-scp - 1], n1 = p[-scp], n2_ = p[-scp + 1], n3 = p[-1], n4 = p[1], n5 = p[scp - 1], n6 = p[scp], n7 = p[scp + 1];
        const int mx = max(max(max(n0, n1), max(n2_, n3)), max(max(n4, n5), max(n6, n7)));
        return ((s > 0) & (s > mx)) ? 1 : 0; // every stored score is >= t
    };
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int q = lane + 64 * k;
        rcq[k] = 0u; fq[k] = 0;
        if (q < n2) {
            rcq[k] = s_q2[q] & 0x7fffu;
            fq[k] = nms_flag(rcq[k]);
            any |= (fq[k] != 0);
        }
    }
    for (int q = lane + 256; q < n2; q += 64) {
        const int f = nms_flag(s_q2[q] & 0x7fffu);
        any |= (f != 0);
        s_qf[q] = (uint8_t)f;
    }
    FAST_WAVE_SYNC();
    if (__ballot(any) != 0ull || attempt == 1) break;
    t = cfg.min_th; // nothing at iniTh: FAST(cell, minThFAST, true)
    } // attempts
    const int need = 1;
    (void)q_bytes;
    // ---- E: ordered emission ----
    uint32_t *oxy = buf.cell_xy + ((size_t)img * cfg.cells_total + cell) * cfg.cell_cap;
    uint8_t *osc = buf.cell_sc + ((size_t)img * cfg.cells_total + cell) * cfg.cell_cap;
    // BK: the cell's survivors fall into the bucket columns gx0..gx1 and rows by0..by1 (tables are monotone).  When that
    // rectangle has at most 64 buckets (the host decides: bk_off) their counts and best keys are accumulated in LDS and stored
    // to the cell's own entries of bk_part: plain stores, no global atomics (a device-scope atomic is a memory-side request
    // of its own on this multi-die part); the quadtree kernel buckets the candidates of the other cells itself
    unsigned *s_ac = (unsigned *)(s_mem + tile_bytes + sc_bytes + q_bytes), *s_ab = s_ac + 64;
    int gx0 = 0, by0 = 0, ncols = 1, nb = 0;
    const bool part = BK && bk_off != ~0u;
    if (part) {
        gx0 = (int)(__builtin_amdgcn_readfirstlane(tabx) >> 16);
        by0 = (int)(__builtin_amdgcn_readfirstlane(taby) >> 16);
        ncols = (int)(__builtin_amdgcn_readlane(tabx, 63) >> 16) - gx0 + 1; // lanes >= iw hold the last column / row
        nb = ncols * ((int)(__builtin_amdgcn_readlane(taby, 63) >> 16) - by0 + 1);
        s_ac[lane] = 0u; s_ab[lane] = 0u;
        FAST_WAVE_SYNC();
    }
    // survivors are first compacted in place over the queue (a write never passes this iteration's reads), then
    // emitted densely: one pass of 64 lanes per 64 survivors instead of one per 64 queue entries
    int run = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (64 * k < n2) {
            const bool pred = fq[k] >= need; // fq = 0 beyond the queue
            const unsigned long long m = __ballot(pred);
            if (pred) s_q2[run + (int)mbcnt64(m, 0u)] = (uint16_t)rcq[k];
            run += __popcll(m);
        }
    }
    for (int q0 = 256; q0 < n2; q0 += 64) {
        const int q = q0 + lane;
        const bool pred = q < n2 && s_qf[q] >= need;
        const unsigned long long m = __ballot(pred);
        const uint16_t rc = q < n2 ? s_q2[q] : (uint16_t)0;
        if (pred) s_q2[run + (int)mbcnt64(m, 0u)] = rc;
        run += __popcll(m);
    }
    FAST_WAVE_SYNC();
    const int n_out = run < cfg.cell_cap ? run : cfg.cell_cap;
    for (int p0 = 0; p0 < n_out; p0 += 64) {
        const int pos = p0 + lane;
        const bool v = pos < n_out;
        const unsigned rc = v ? (s_q2[pos] & 0x7fffu) : 0u;
        const int r = (int)(rc >> 8) - 1, c = rc & 255;
        unsigned tx = 0u, ty = 0u;
        if (part) { tx = (unsigned)__shfl((int)tabx, c, 64); ty = (unsigned)__shfl((int)taby, r, 64); }
        if (v) {
            // cell-local FAST coords (c+3, r+3) + (j*wCell, i*hCell): src/ORBextractor.cc:816-817
            const unsigned x = (unsigned)(c + 3 + cell_x0);
            const unsigned y = (unsigned)(r + 3 + cell_y0);
            const unsigned sc = s_sc[(r + 1) * scp + c + 1];
            oxy[pos] = x | (y << 16);
            osc[pos] = (uint8_t)sc;
            if (part) {
                const int li = ((int)(ty >> 16) - by0) * ncols + ((int)(tx >> 16) - gx0);
                atomicAdd(&s_ac[li], 1u);
                atomicMax(&s_ab[li], ORBFE_BK_KEY(sc, (unsigned)ci, (unsigned)pos));
            }
        }
    }
    if (lane == 0) *cnt_out = run < cfg.cell_cap ? run : cfg.cell_cap;
    if (part && !ORBFE_CUT(5)) { // every frame rewrites all nb entries (zeros included): nothing to clear between frames
        FAST_WAVE_SYNC();
        if (lane < nb) buf.bk_part[(size_t)img * cfg.bk_part_total + bk_off + lane] = ORBFE_BK_PART(s_ac[lane], s_ab[lane]);
    }
}


static inline int max_cell_w(const DeviceConfig &cfg) { int m = 0; for (int l = 0; l < cfg.nlevels; l++) m = cfg.lv[l].w_cell > m ? cfg.lv[l].w_cell : m; return m; }
static inline int max_cell_h(const DeviceConfig &cfg) { int m = 0; for (int l = 0; l < cfg.nlevels; l++) m = cfg.lv[l].h_cell > m ? cfg.lv[l].h_cell : m; return m; }

int orbfe_fast_tile_pitch(const DeviceConfig &cfg) { return (max_cell_w(cfg) + 6 + 15) & ~15; } // whole 16-byte chunks (the staging stores 128 bits at a time)

void orbfe_launch_fast(const DeviceConfig &cfg_in, const DeviceBuffers &buf, int n_images, bool buckets, hipStream_t s, int blur_first_level)
{
    DeviceConfig cfg = cfg_in;
    cfg.fast_blur_t0 = blur_first_level < cfg.nlevels ? cfg.lv[blur_first_level].blur_tile_off : cfg.blur_tiles_total;
    const int mw = max_cell_w(cfg), mh = max_cell_h(cfg);
    const int tile_pitch = orbfe_fast_tile_pitch(cfg);
    const int tile_rows = mh + 6;
    const int tile_bytes = (tile_pitch * tile_rows + 15) & ~15;
    const int sc_bytes = ((mw + 2) * (mh + 2) + 15) & ~15;
    const int q_bytes = (2 * mw * mh + 15) & ~15;
    // flags alias the tile: it must hold one byte per interior pixel
    // flags alias the tile region: tile_bytes passed to the kernel covers both; + 2 x 64 words of bucket accumulators
    const int tile_region = tile_bytes > mw * mh ? tile_bytes : ((mw * mh + 15) & ~15);
    const int lds_per_wave = tile_region + sc_bytes + q_bytes + 512;
    const size_t lds = (size_t)4 * lds_per_wave;
    const int bpi = (cfg.cells_total + 3) / 4 + (cfg.blur_tiles_total - cfg.fast_blur_t0 + 3) / 4;
    cfg.xcd_magic = xcd_map_magic_host(bpi, n_images);
    dim3 grid(xcd_grid(bpi, n_images));
#define FAST_LAUNCH(TP)                                                                                                                   \
    do {                                                                                                                              \
        if (buckets) hipLaunchKernelGGL((fast_cell_kernel<TP, true>), grid, dim3(256), lds, s, cfg, buf, n_images, tile_pitch, tile_region, sc_bytes, q_bytes, lds_per_wave ORBFE_CUT_ARG("ORBFE_FAST_DBG")); \
        else hipLaunchKernelGGL((fast_cell_kernel<TP, false>), grid, dim3(256), lds, s, cfg, buf, n_images, tile_pitch, tile_region, sc_bytes, q_bytes, lds_per_wave ORBFE_CUT_ARG("ORBFE_FAST_DBG")); \
    } while (0)
    switch (tile_pitch) {
    case 32: FAST_LAUNCH(32); break;
    case 48: FAST_LAUNCH(48); break;
    case 64: FAST_LAUNCH(64); break;
    default: FAST_LAUNCH(0); break;
    }
#undef FAST_LAUNCH
}
