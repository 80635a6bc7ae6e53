// orbfe_octree3.hip -- DistributeOctTree (reference src/ORBextractor.cc:533-757) on bucket pyramids.
//
// The split geometry of the reference is data independent: a node's box is halved at
// x0 + ceil((x1-x0)/2), y0 + ceil((y1-y0)/2) whatever its points are, so the path of a point through the
// tree (root index, then one quadrant per depth) depends on its coordinates only.  What IS data
// dependent -- which nodes are split, in which order, and when the process stops -- needs nothing but
// the number of points in a node and in its four children.  Hence:
//   1. fast_cell_kernel (phase E) looks up every survivor's root and quadrant path down to depth 5
//      ("bucket") in two host-built tables (the path is separable in x and y) and accumulates per-bucket
//      counts and the per-bucket best key  score << 24 | ~(cell << 12 | slot)  (max score, first in
//      cv::FAST emission order = the reference's "first maximum wins") -- per cell in LDS, then
//      plain stores of the cell's partial (count, best key) entries (bucket partials, DeviceBuffers::bk_part: a cell's
//      survivors fall into a small rectangle of buckets); this kernel adds them up per bucket in LDS;
//   2. counts and best keys are summed / maximised up the quadrant pyramid (depths 4..0);
//   3. the split passes work on the node list alone (<= max_nodes entries in LDS): a node is
//      (box, depth, path), its child counts are pyramid look-ups, and the list-order bookkeeping, the
//      "largest node first" phase and the stop rules are those of tests/octree_model.py, which is
//      validated against the literal std::list restatement (and were the point-parallel kernel's, which
//      round 5 removed).  No pass touches the points again;
//   4. the surviving point of a node is a best-key look-up.
// Nodes deeper than the bucket depth (clustered candidates with a generous quota) take a slow path:
// the candidates are counting-sorted by bucket once (HBM scratch), and a deep node classifies the few
// points of its bucket by replaying their paths.
// Candidates are never gathered into emission order on this path; orbfe_fetch_candidates runs
// candidates_gather_kernel on demand (parity tap).
#include "orbfe_common.hpp"
#include <cstdlib>

// Round 4: 256 threads and <= 40 KB of LDS per workgroup, so that all 1024 workgroups of a 64-pair step are resident at once
// (four per CU: the kernel needs 118 VGPRs = 4 waves per SIMD = 16 per CU; with 512 threads and 80 KB that was two workgroups
// per CU and two rounds of 25 us each).  What left the LDS: the node boxes (never read: a node's children come from the count
// pyramid, its box is implied by root + path), the best-key pyramid after the pyramid step (copied to an HBM scratch slice, read
// back by the final selection: one more dependent load per kept node), half of the count pyramid (depths 4 and 5 are 16 bit:
// a bucket holds at most a few thousand strict-NMS survivors, orbfe_create checks), the sort keys (they alias the node array
// that is being built).
#ifndef OT3_THREADS
#define OT3_THREADS 256
#endif
#define OT3_WAVES (OT3_THREADS / 64)
#define OT3_DB 5                       // bucket depth
#define OT3_ROOTS 4                    // root slots per level (n_ini <= 4)
#define OT3_PYR (OT3_ROOTS * 1365)     // sum_{d=0..5} 4^d = 1365 entries per root
#define OT3_HI (OT3_ROOTS * 85)        // entries of depths 0..3 (32-bit counts); depths 4 and 5 follow as 16-bit counts
static_assert(OT3_PYR == ORBFE_BK_PYR, "the best-key scratch of orbfe_api.hip holds one pyramid per (image, level)");
#define OT3_BUCKETS (OT3_ROOTS * 1024)
// best key: score (8 bits) << 24 | ~(cell (12 bits) << 12 | slot (12 bits)); the host checks the field widths
#define OT3_REF_MASK ORBFE_BK_REF_MASK
#define OT3_KEY(sc, cell, slot) ORBFE_BK_KEY(sc, cell, slot)
static_assert(OT3_DB == ORBFE_BK_DEPTH && OT3_BUCKETS == ORBFE_BK_BUCKETS, "bucket geometry is shared with fast_cell_kernel");

// entries of depths < d; entry (d, root, path) = ot3_off(d) + (root << 2d) + path, its children are
// ot3_off(d+1) + 4 * ((root << 2d) + path) + quadrant
__device__ __forceinline__ int ot3_off(int d) { return OT3_ROOTS * (((1 << (2 * d)) - 1) / 3); }

__device__ __forceinline__ int ot3_wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// exclusive block scan of 4 values per thread; totals in tot[4].  s_w: 4*OT3_WAVES ints.
__device__ __forceinline__ void ot3_block_scan4(int v[4], int tot[4], int *s_w)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) inc[c] = ot3_wave_incl_scan(v[c], lane);
    if (lane == 63) {
#pragma unroll
        for (int c = 0; c < 4; c++) s_w[4 * wave + c] = inc[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; c++) {
        int base = 0, t = 0;
        for (int w = 0; w < OT3_WAVES; w++) {
            const int x = s_w[4 * w + c];
            if (w < wave) base += x;
            t += x;
        }
        tot[c] = t;
        v[c] = base + inc[c] - v[c];
    }
    __syncthreads();
}

// in-place exclusive scan of an LDS int array a[0..n); returns the total.
__device__ __forceinline__ int ot3_scan_array(int *a, int n, int *s_w)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + OT3_THREADS - 1) / OT3_THREADS;
    const int b = tid * per < n ? tid * per : n, e = (b + per < n) ? b + per : n;
    int sum = 0;
    for (int i = b; i < e; i++) sum += a[i];
    const int inc = ot3_wave_incl_scan(sum, lane);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < OT3_WAVES; w++) {
        const int x = s_w[w];
        if (w < wave) base += x;
        total += x;
    }
    int run = base + inc - sum;
    for (int i = b; i < e; i++) {
        const int v = a[i];
        a[i] = run;
        run += v;
    }
    __syncthreads();
    return total;
}

struct Ot3Nodes {
    int *cnt;
    unsigned *path;          // quadrant path, 2 bits per depth (depth <= 15)
    int *dr;                 // depth | root << 4
};

// bytes of one node array: three words per node, and room for the sort keys of the "largest node first" phase, which are built
// in the array that is not in use (the next pass's nodes are written after the ranking)
__host__ __device__ __forceinline__ size_t ot3_slot_bytes(int cap, int sort_cap)
{
    const size_t a = 3 * sizeof(int) * (size_t)cap, k = sizeof(unsigned long long) * (size_t)sort_cap;
    return ((a > k ? a : k) + 15) & ~(size_t)15;
}

__device__ __forceinline__ void ot3_bind(Ot3Nodes &n, uint8_t *&p, int cap, int sort_cap)
{
    uint8_t *q = p;
    n.cnt = (int *)q; q += sizeof(int) * cap;
    n.path = (unsigned *)q; q += sizeof(unsigned) * cap;
    n.dr = (int *)q;
    p += ot3_slot_bytes(cap, sort_cap);
}

// bytes of the node tables (two node arrays, per-node bookkeeping) of one workgroup
size_t orbfe_octree3_node_bytes(int max_nodes, int sort_cap)
{
    const size_t cap = (size_t)max_nodes;
    return ((2 * ot3_slot_bytes(max_nodes, sort_cap) + sizeof(int) * cap * (4 + 1 + 1 + 1 + 1) + 64) + 255) & ~(size_t)255;
}

// dynamic LDS of octree3_kernel: the count pyramid (32-bit entries for depths 0..3, 16-bit for depths 4 and 5), then one region
// that holds the best-key pyramid while the buckets are summed up and the node tables afterwards (unless those live in HBM)
size_t orbfe_octree3_lds_bytes(int max_nodes, int sort_cap, bool nodes_in_hbm)
{
    const size_t cnt = sizeof(int) * OT3_HI + sizeof(uint16_t) * (OT3_PYR - OT3_HI), best = sizeof(int) * OT3_PYR;
    const size_t nodes = nodes_in_hbm ? 0 : orbfe_octree3_node_bytes(max_nodes, sort_cap);
    return ((cnt + 15) & ~(size_t)15) + (best > nodes ? best : nodes) + 64;
}

// root and quadrant path of a point down to `depth` (src/ORBextractor.cc:537-564 for the root, :145-209 for a split)
__device__ __forceinline__ unsigned ot3_path(int x, int y, int depth, float hx, int n_ini, int region_h, int &root)
{
    int b = (int)__fdiv_rn((float)x, hx);
    b = b < 0 ? 0 : (b >= n_ini ? n_ini - 1 : b);
    root = b;
    int x0 = (int)__fmul_rn(hx, (float)b), x1 = (int)__fmul_rn(hx, (float)(b + 1)), y0 = 0, y1 = region_h;
    unsigned path = 0;
    for (int d = 0; d < depth; d++) {
        const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
        const int cx = x < mx ? 0 : 1, cy = y < my ? 0 : 1;
        path = (path << 2) | (unsigned)(cx + 2 * cy);
        x0 = cx ? mx : x0; x1 = cx ? x1 : mx;
        y0 = cy ? my : y0; y1 = cy ? y1 : my;
    }
    return path;
}

// Walk over the level's cell slots: wave w takes cells w, w+8, ...; lanes take slots.  The first two
// 64-slot chunks of a cell are loaded without waiting for the cell's count (slots beyond it are
// allocated but unused), so a wave keeps six independent loads in flight.  f(xy, score, cell, slot).
template <typename F>
__device__ __forceinline__ void ot3_for_each_point(const int *cell_cnt, const uint32_t *cell_xy, const uint8_t *cell_sc, int n_cells, int cell_cap, F f)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < n_cells; c += OT3_WAVES) {
        const size_t base = (size_t)c * cell_cap;
        const int k1 = lane + 64;
        const int cnt = cell_cnt[c];
        const uint32_t xa = lane < cell_cap ? cell_xy[base + lane] : 0u;
        const uint8_t sa = lane < cell_cap ? cell_sc[base + lane] : (uint8_t)0;
        const uint32_t xb = k1 < cell_cap ? cell_xy[base + k1] : 0u;
        const uint8_t sb = k1 < cell_cap ? cell_sc[base + k1] : (uint8_t)0;
        if (lane < cnt) f(xa, (unsigned)sa, c, lane);
        if (k1 < cnt) f(xb, (unsigned)sb, c, k1);
        for (int k = lane + 128; k < cnt; k += 64) f(cell_xy[base + k], (unsigned)cell_sc[base + k], c, k);
    }
}

template <bool NODES_IN_HBM>
__global__ __launch_bounds__(OT3_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void octree3_kernel(DeviceConfig cfg, DeviceBuffers buf, int sort_cap, size_t node_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
    __shared__ int s_w[4 * OT3_WAVES];
    __shared__ int s_scal[8];
    __shared__ __attribute__((aligned(16))) int s_bin[256]; // step 4a (processing order for describe_kernel): zeroed here, used at the end
    // longest first: the workgroups of level 0 (largest quota, most split passes) are dispatched before those of level 1, ...
    const int img = blockIdx.x, level = (int)blockIdx.y;
    for (int i = threadIdx.x; i < 256; i += OT3_THREADS) s_bin[i] = 0;
    const LevelInfo &L = cfg.lv[level];
    const int tid = threadIdx.x;
#ifdef ORBFE_PROFILE_CUTS // tools/octree3_timeline.py (`make cuts` build only): start / end of the first 2048 workgroups and the phase
                          // boundaries of image 0's workgroups, 100 MHz clock
    struct Stamp {
        long long *p;
        __device__ Stamp(long long *q) : p(q) { if (p) p[0] = (long long)__builtin_amdgcn_s_memrealtime(); }
        __device__ ~Stamp() { if (p) p[1] = (long long)__builtin_amdgcn_s_memrealtime(); }
    } stamp(tid == 0 && (img * cfg.nlevels + level) < 2048 ? buf.dbg_ts + 2 * (img * cfg.nlevels + level) : nullptr);
    long long *ph = tid == 0 && img == 0 && level < 8 ? buf.dbg_ts + 2048 + 64 * level : nullptr;
    int ph_n = 0;
#define OT3_PHASE() do { if (ph && ph_n < 63) ph[ph_n++] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define OT3_PHASE() do { } while (0)
#endif
    OT3_PHASE();
    const int MAXN = cfg.max_nodes;

    // LDS: counts of depths 0..3 (int), of depths 4 and 5 (uint16: entry e >= OT3_HI at s_c16[e - OT3_HI]), then the shared region:
    // best-key pyramid until the pyramid step is done, node tables afterwards
    uint8_t *p = s_raw;
    int *s_chi = (int *)p; p += sizeof(int) * OT3_HI;
    uint16_t *s_c16 = (uint16_t *)p; p += sizeof(uint16_t) * (OT3_PYR - OT3_HI);
    p = (uint8_t *)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    unsigned *s_best = (unsigned *)p; // best key, all depths (steps 1 and 2 only)
    unsigned *g_best = buf.bk_best + ((size_t)img * cfg.nlevels + level) * OT3_PYR; // ... and where it lives afterwards
    auto cnt_at = [&](int e) -> int { return e < OT3_HI ? s_chi[e] : (int)s_c16[e - OT3_HI]; };
    // deep path only: s_bend[1 + b] = end of bucket b in the sorted arrays.  Kept in HBM scratch so that the LDS
    // footprint lets several workgroups share a CU.
    int *s_bend = buf.bk_end + ((size_t)img * cfg.nlevels + level) * (OT3_BUCKETS + 1);
    // Node tables: LDS when they fit beside the pyramids (NODES_IN_HBM = false); otherwise the workgroup's slice of an HBM
    // scratch buffer -- many features on few levels (a level's node capacity ~ its quota) must not be a create-time error.
    if (NODES_IN_HBM) p = buf.ot3_scratch + ((size_t)img * cfg.nlevels + level) * node_bytes;
    Ot3Nodes A, B;
    ot3_bind(A, p, MAXN, sort_cap);
    ot3_bind(B, p, MAXN, sort_cap);
    int *s_ccnt = (int *)p; p += sizeof(int) * 4 * MAXN; // child counts per node
    int *s_rank = (int *)p; p += sizeof(int) * MAXN;     // processing rank (-1: not processed this pass)
    int *s_plist = (int *)p; p += sizeof(int) * MAXN;    // processing order -> node
    int *s_kk = (int *)p; p += sizeof(int) * MAXN;
    int *s_un = (int *)p; p += sizeof(int) * MAXN;
    int *s_n = &s_scal[0], *s_total_k = &s_scal[1], *s_nproc = &s_scal[2], *s_nexpand = &s_scal[3], *s_mode = &s_scal[4], *s_done = &s_scal[5],
        *s_deep = &s_scal[6];

    const size_t ib = (size_t)img;
    const int *cell_cnt = buf.cell_cnt + ib * cfg.cells_total + L.cell_off;
    const uint32_t *cell_xy = buf.cell_xy + (ib * cfg.cells_total + L.cell_off) * cfg.cell_cap;
    const uint8_t *cell_sc = buf.cell_sc + (ib * cfg.cells_total + L.cell_off) * cfg.cell_cap;
    const size_t coff = ib * cfg.cand_total + L.cand_off;
    uint32_t *deep_xy = buf.ot_xy2 + coff;  // deep path: candidates counting-sorted by bucket
    uint32_t *deep_key = buf.idx0 + coff;
    int *sel_cnt = buf.sel_cnt + ib * cfg.nlevels + level;
    uint32_t *sel_xy = buf.sel_xy + ib * cfg.sel_total + L.sel_off;
    uint8_t *sel_sc = buf.sel_sc + ib * cfg.sel_total + L.sel_off;
    int *status = buf.status + img;
    const int region_h = (L.h - cfg.edge_threshold + 3) - cfg.min_border;
    const int n_ini = L.n_ini, quota = L.quota;
    const float hx = L.hx;
    const int db = L.bk_depth; // this level's bucket depth: 5, or 4 (small levels: orbfe_create picks it so that every FAST cell has bucket partials)

    // ---- 1. buckets: the per-cell partial (count, best key) entries of fast_cell_kernel<.., true> (phase E), summed / maximised
    //      per bucket in LDS; bk_emap (host-built) names the bucket of every entry.  The candidates of cells without entries
    //      (more than 64 buckets under one cell: levels whose buckets are ~3 px) are bucketed here from the cell slots. ----
    {
        // depth-5 counts: 16 bit each, summed with 32-bit LDS atomics on the word that holds the pair (a bucket's total stays below
        // 2^16, so nothing carries into the neighbour)
        unsigned *cnt5w = (unsigned *)(s_c16 + (ot3_off(db) - OT3_HI)); // (names from the depth-5 case)
        unsigned *best5 = s_best + ot3_off(db);
        auto cnt5_add = [&](int b, int v) { atomicAdd(&cnt5w[b >> 1], (unsigned)v << (16 * (b & 1))); };
        // 24 entries per thread in flight as six 128-bit loads of each array (one batch = 6144 entries covers level 0 of a KITTI frame;
        // with ten dword loads per array and 256 threads that level was three dependent load rounds: 7 us of the launch's critical
        // path); orbfe_create starts every level's entries on a 16-byte boundary (padding entries are zero and never written); the
        // first batch is issued before the LDS arrays are cleared
        const uint32_t *part = buf.bk_part + ib * cfg.bk_part_total + L.bk_part_off;
        const uint32_t *emap = buf.bk_emap + L.bk_part_off;
        const int n_part = L.bk_part_n;
        constexpr int U = 6;
        uint4 pe[U], be[U];
        auto load = [&](int e0) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int e = e0 + 4 * (u * OT3_THREADS + tid);
                pe[u] = make_uint4(0u, 0u, 0u, 0u); be[u] = pe[u];
                if (e < n_part) { pe[u] = *(const uint4 *)(part + e); be[u] = *(const uint4 *)(emap + e); } // whole quads exist: the level's range is padded to a multiple of 4
            }
        };
        auto apply1 = [&](uint32_t p1, uint32_t b1) {
            if (p1 & 0xfffu) {
                cnt5_add((int)(b1 & 0xffffu), (int)(p1 & 0xfffu));
                atomicMax(&best5[b1 & 0xffffu], ORBFE_BK_PART_KEY(p1, b1 >> 16));
            }
        };
        auto apply = [&]() {
#pragma unroll
            for (int u = 0; u < U; u++) { apply1(pe[u].x, be[u].x); apply1(pe[u].y, be[u].y); apply1(pe[u].z, be[u].z); apply1(pe[u].w, be[u].w); }
        };
        load(0);
        const uint4 zero = {0u, 0u, 0u, 0u};
        const int n_bk = OT3_ROOTS << (2 * db); // 4096 or 1024 buckets
        for (int i = tid; i < n_bk / 4; i += OT3_THREADS) ((uint4 *)best5)[i] = zero;
        for (int i = tid; i < n_bk / 8; i += OT3_THREADS) ((uint4 *)cnt5w)[i] = zero;
        __syncthreads();
        apply();
        for (int e0 = 4 * U * OT3_THREADS; e0 < n_part; e0 += 4 * U * OT3_THREADS) { load(e0); apply(); }
        OT3_PHASE();
        if (L.bk_points) {
            // sixteen lanes per cell (these levels have a handful of candidates per cell), everything a lane needs first loaded
            // without waiting for the cell's count: slots beyond it are allocated but unused.  On the benchmark's dense images this
            // phase is 14-18 us of the workgroups of levels 4-7, which end the launch (tools/octree3_timeline.py).  Round 4 measured
            // three rewrites that all stayed at 14-19 us: the bucket from the host tables instead of ot3_path, four / six items per
            // thread with every load issued first, 32 lanes x 2 preloaded slots per cell -- each trades dependent round trips (~2 us
            // under this launch's load) against VALU issue (the four workgroups of a CU are of one level and in this phase together)
            const uint32_t *bk_off = buf.bk_off + L.cell_off;
            auto put = [&](uint32_t xy, unsigned sc, int c, int k) {
                int root;
                const unsigned path = ot3_path((int)(xy & 0xffffu), (int)(xy >> 16), db, hx, n_ini, region_h, root);
                const int b = (root << (2 * db)) + (int)path;
                cnt5_add(b, 1);
                atomicMax(&best5[b], OT3_KEY(sc, c, k));
            };
            for (int idx = tid; idx < 16 * L.n_cells; idx += OT3_THREADS) {
                const int c = idx >> 4, k0 = idx & 15;
                const size_t base = (size_t)c * cfg.cell_cap;
                const uint32_t off = bk_off[c];
                const int cnt = cell_cnt[c];
                const uint32_t xy0 = k0 < cfg.cell_cap ? cell_xy[base + k0] : 0u;
                const unsigned sc0 = k0 < cfg.cell_cap ? cell_sc[base + k0] : 0u;
                if (off != ~0u) continue;
                if (k0 < cnt) put(xy0, sc0, c, k0);
                for (int k = k0 + 16; k < cnt; k += 16) put(cell_xy[base + k], (unsigned)cell_sc[base + k], c, k);
            }
        }
    }
    __syncthreads();
    OT3_PHASE();
    // ---- 2. pyramid ----
    for (int d = db - 1; d >= 0; d--) {
        const int n_e = OT3_ROOTS << (2 * d);
        const int o = ot3_off(d), oc = ot3_off(d + 1);
        for (int e = tid; e < n_e; e += OT3_THREADS) {
            const int c = oc + 4 * e;
            const int sum = cnt_at(c) + cnt_at(c + 1) + cnt_at(c + 2) + cnt_at(c + 3);
            if (o + e < OT3_HI) s_chi[o + e] = sum; else s_c16[o + e - OT3_HI] = (uint16_t)sum;
            const unsigned b0 = s_best[c], b1 = s_best[c + 1], b2 = s_best[c + 2], b3 = s_best[c + 3];
            const unsigned m01 = b0 > b1 ? b0 : b1, m23 = b2 > b3 ? b2 : b3;
            s_best[o + e] = m01 > m23 ? m01 : m23;
        }
        __syncthreads();
    }
    // the best keys leave the LDS: the final selection reads them back from this workgroup's slice (its own stores: same L2),
    // and their LDS region becomes the node tables
    for (int i = tid; i < OT3_PYR / 4; i += OT3_THREADS) ((uint4 *)g_best)[i] = ((const uint4 *)s_best)[i];
    // the region may be overwritten once every wave has READ its part (the stores carry the data in registers): wait for the LDS
    // reads only, not for the stores' round trip (__syncthreads waits for vmcnt(0): 4 us here); the stores have long landed when
    // the final selection reads the slice, many barriers later
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    OT3_PHASE();
    // ---- roots (src/ORBextractor.cc:537-581) ----
    if (tid == 0) {
        int n = 0, nc = 0;
        for (int b = 0; b < n_ini; b++) {
            const int c = s_chi[b];
            nc += c;
            if (c > 0) {
                A.cnt[n] = c; A.path[n] = 0u; A.dr[n] = b << 4; // the root's box (:541-557) is implied by b: see ot3_path
                n++;
            }
        }
        *s_n = n;
        *s_done = 0;
        *s_deep = 0;
        buf.lvl_ncand[ib * cfg.nlevels + level] = nc;
        s_scal[7] = nc;
    }
    __syncthreads();
    const int nc = s_scal[7];
    if (nc == 0) {
        if (tid == 0) *sel_cnt = 0;
        return;
    }
    const bool deep_ok = nc <= L.cand_cap; // the sorted arrays hold cand_cap entries
    if (!deep_ok && tid == 0) *status = 1;
    bool deep_ready = false;

    // child counts of a node deeper than the buckets: replay the paths of its bucket's points
    auto deep_children = [&](int dr, unsigned path, int c[4]) {
        const int d = dr & 15, root = dr >> 4;
        const int b = (root << (2 * db)) + (int)(path >> (2 * (d - db)));
        c[0] = c[1] = c[2] = c[3] = 0;
        for (int j = s_bend[b]; j < s_bend[b + 1]; j++) {
            const uint32_t xy = deep_xy[j];
            int r;
            const unsigned pp = ot3_path((int)(xy & 0xffffu), (int)(xy >> 16), d + 1, hx, n_ini, region_h, r);
            if ((pp >> 2) == path) c[pp & 3u]++;
        }
    };

    // ---- 3. split passes ----
    Ot3Nodes cur = A, nxt = B;
    int sorted_phase = 0;
    for (int iter = 0; iter < 100000; iter++) { // n grows every pass, so this ends at n >= quota at the latest
        const int n = *s_n;
        OT3_PHASE();
        // child k of node i as new node q
        auto emit = [&](int q, int i, int k, int cnt) {
            nxt.cnt[q] = cnt;
            nxt.path[q] = (cur.path[i] << 2) | (unsigned)k;
            nxt.dr[q] = (cur.dr[i] & 15) < 15 ? cur.dr[i] + 1 : cur.dr[i]; // depth 15 = 1-px boxes: never multi-point
        };
        auto copy_node = [&](int q, int i) { nxt.cnt[q] = cur.cnt[i]; nxt.path[q] = cur.path[i]; nxt.dr[q] = cur.dr[i]; };
        if (!sorted_phase && n <= OT3_THREADS) {
            // Express full pass (round 4): every multi-point node is split and thread i owns node i, so the child counts stay in
            // registers, one 4-value block scan yields the ranks and the pass totals, every thread knows the stop rule's inputs, and the
            // pass needs ONE barrier after the scan's two instead of six (the first passes -- 4, 16, 64 nodes -- are nothing but
            // barriers and LDS round trips: 2.6 us each on the launch's critical path).  A node deeper than the bucket pyramid
            // (counted in the scan's spare bits) sends the pass down the general path below.
            int v4[4] = {0, 0, 0, 0}, c0 = 0, c1 = 0, c2 = 0, c3 = 0, k = 0, multi = 0;
            if (tid < n) {
                multi = cur.cnt[tid] > 1;
                if (multi) {
                    const int dr = cur.dr[tid], d = dr & 15, root = dr >> 4;
                    if (d < db) {
                        const int c = ot3_off(d + 1) + 4 * ((root << (2 * d)) + (int)cur.path[tid]);
                        c0 = cnt_at(c); c1 = cnt_at(c + 1); c2 = cnt_at(c + 2); c3 = cnt_at(c + 3);
                        k = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
                        v4[0] = 1; v4[1] = k; v4[3] = (c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1);
                    } else {
                        v4[0] = 1 << 16; // deep
                    }
                } else {
                    v4[2] = 1;
                }
            }
            int tot[4];
            ot3_block_scan4(v4, tot, s_w); // exclusive prefixes in list order
            if ((tot[0] >> 16) == 0) {
                const int total_k = tot[1], n_new = total_k + tot[2];
                if (n_new > MAXN) {
                    if (tid == 0) { *status = 2; *sel_cnt = 0; }
                    return;
                }
                if (tid < n) {
                    const int i = tid;
                    if (!multi) {
                        copy_node(total_k + v4[2], i);
                    } else {
                        int q = total_k - (v4[1] + k); // blocks of later-processed parents sit nearer the front; children n4..n1
                        if (c3 > 0) emit(q++, i, 3, c3);
                        if (c2 > 0) emit(q++, i, 2, c2);
                        if (c1 > 0) emit(q++, i, 1, c1);
                        if (c0 > 0) emit(q++, i, 0, c0);
                    }
                }
                // stop logic (src/ORBextractor.cc:661-731): uniform values, written for the general path's readers
                const bool done = n_new >= quota || n_new == n;
                const int mode = n_new + 3 * tot[3] > quota ? 1 : 0;
                if (tid == 0) { *s_n = n_new; *s_deep = 0; *s_done = done ? 1 : 0; *s_mode = mode; }
                __syncthreads();
                { Ot3Nodes t = cur; cur = nxt; nxt = t; }
                if (done) break;
                sorted_phase = mode;
                continue;
            }
        }
        // child counts
        for (int i = tid; i < n; i += OT3_THREADS) {
            int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            const int multi = cur.cnt[i] > 1;
            if (multi) {
                const int dr = cur.dr[i], d = dr & 15, root = dr >> 4;
                if (d < db) {
                    const int c = ot3_off(d + 1) + 4 * ((root << (2 * d)) + (int)cur.path[i]);
                    c0 = cnt_at(c); c1 = cnt_at(c + 1); c2 = cnt_at(c + 2); c3 = cnt_at(c + 3);
                } else {
                    *s_deep = 1;
                }
            }
            s_ccnt[4 * i] = c0; s_ccnt[4 * i + 1] = c1; s_ccnt[4 * i + 2] = c2; s_ccnt[4 * i + 3] = c3;
            s_rank[i] = -1;
            s_kk[i] = multi;
        }
        __syncthreads();
        if (*s_deep) {
            if (!deep_ready && deep_ok) {
                // counting sort of the candidates by bucket: s_bend[1 + b] runs from the bucket's start to its end
                const uint16_t *cnt5 = s_c16 + (ot3_off(db) - OT3_HI);
                const int n_bk = OT3_ROOTS << (2 * db);
                for (int b = tid; b < n_bk; b += OT3_THREADS) s_bend[1 + b] = (int)cnt5[b];
                if (tid == 0) s_bend[0] = 0;
                __syncthreads();
                ot3_scan_array(s_bend + 1, n_bk, s_w);
                ot3_for_each_point(cell_cnt, cell_xy, cell_sc, L.n_cells, cfg.cell_cap, [&](uint32_t xy, unsigned sc, int cell, int slot) {
                    int root;
                    const unsigned path = ot3_path((int)(xy & 0xffffu), (int)(xy >> 16), db, hx, n_ini, region_h, root);
                    const int b = (root << (2 * db)) + (int)path;
                    const int pos = atomicAdd(&s_bend[1 + b], 1);
                    deep_xy[pos] = xy;
                    deep_key[pos] = OT3_KEY(sc, cell, slot);
                });
                __syncthreads();
                deep_ready = true;
            }
            for (int i = tid; i < n; i += OT3_THREADS) {
                const int dr = cur.dr[i];
                if (cur.cnt[i] > 1 && (dr & 15) >= db) {
                    int c[4] = {0, 0, 0, 0};
                    if (deep_ready && (dr & 15) < 15) deep_children(dr, cur.path[i], c);
                    else { c[0] = cur.cnt[i]; } // cannot be refined (capacity guards only): keep the node whole
                    s_ccnt[4 * i] = c[0]; s_ccnt[4 * i + 1] = c[1]; s_ccnt[4 * i + 2] = c[2]; s_ccnt[4 * i + 3] = c[3];
                }
            }
            __syncthreads();
        }

        int n_new, nexpand_fast = -1;
        if (!sorted_phase && n <= OT3_THREADS) {
            // full pass (every multi-point node is split, list order = processing order): thread i owns node i and
            // one 4-value block scan yields its rank, the children ahead of it, the unsplit nodes ahead of it and
            // the pass totals
            int v4[4] = {0, 0, 0, 0}, c0 = 0, c1 = 0, c2 = 0, c3 = 0, k = 0, multi = 0;
            if (tid < n) {
                multi = cur.cnt[tid] > 1;
                if (multi) {
                    c0 = s_ccnt[4 * tid]; c1 = s_ccnt[4 * tid + 1]; c2 = s_ccnt[4 * tid + 2]; c3 = s_ccnt[4 * tid + 3];
                    k = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
                    v4[0] = 1; v4[1] = k; v4[3] = (c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1);
                } else {
                    v4[2] = 1;
                }
            }
            int tot[4];
            ot3_block_scan4(v4, tot, s_w); // exclusive prefixes in list order
            const int total_k = tot[1];
            n_new = total_k + tot[2];
            nexpand_fast = tot[3];
            if (n_new > MAXN) {
                if (tid == 0) { *status = 2; *sel_cnt = 0; }
                return;
            }
            if (tid < n) {
                const int i = tid;
                if (!multi) {
                    copy_node(total_k + v4[2], i);
                } else {
                    int q = total_k - (v4[1] + k); // blocks of later-processed parents sit nearer the front; children n4..n1
                    if (c3 > 0) emit(q++, i, 3, c3);
                    if (c2 > 0) emit(q++, i, 2, c2);
                    if (c1 > 0) emit(q++, i, 1, c1);
                    if (c0 > 0) emit(q++, i, 0, c0);
                }
            }
        } else {
            // processing order of the multi-point nodes
            const int m = ot3_scan_array(s_kk, n, s_w); // s_kk[i] = rank among multi nodes (list order)
            if (sorted_phase) OT3_PHASE(); // scan kk
            if (!sorted_phase) {
                for (int i = tid; i < n; i += OT3_THREADS)
                    if (cur.cnt[i] > 1) s_plist[s_kk[i]] = i;
                __syncthreads();
            } else {
                // (size, pointer) order of the reference under contract Q3: count descending, list position ascending.  The keys are
                // built in the node array of the NEXT pass (written only after the ranking; ot3_slot_bytes makes room)
                unsigned long long *s_key = (unsigned long long *)nxt.cnt;
                int P = 1;
                while (P < m) P <<= 1;
                if (m <= 4 * OT3_THREADS && nc < 65536) {
                    // few keys (all distinct), counts and list positions below 2^16: 32-bit keys ranked by counting -- every thread
                    // compares its key(s) with all m, sixteen per step (four 128-bit broadcast reads in flight; with one key per
                    // step the loop is bound by LDS latency: 7-10 us of a workgroup's 30, tools/octree3_timeline.py); the rank is
                    // the node's place in s_plist
                    uint32_t *k32 = (uint32_t *)s_key;
                    const int m16 = (m + 15) & ~15; // <= 2 * sort_cap words
                    for (int i = tid; i < m16; i += OT3_THREADS) k32[i] = ~0u;
                    __syncthreads();
                    for (int i = tid; i < n; i += OT3_THREADS)
                        if (cur.cnt[i] > 1) k32[s_kk[i]] = ((0xffffu - (unsigned)cur.cnt[i]) << 16) | (unsigned)i;
                    __syncthreads();
                    auto below = [](const uint4 &q, uint32_t v) { return (int)(q.x < v) + (int)(q.y < v) + (int)(q.z < v) + (int)(q.w < v); };
                    for (int u = 0; u * OT3_THREADS < m; u++) { // one round per 512 keys
                        const bool has = tid + u * OT3_THREADS < m;
                        const uint32_t mine = has ? k32[tid + u * OT3_THREADS] : ~0u;
                        int rk = 0;
                        for (int j = 0; j < m16; j += 16) {
                            const uint4 a = *(const uint4 *)(k32 + j), b = *(const uint4 *)(k32 + j + 4), c = *(const uint4 *)(k32 + j + 8), d = *(const uint4 *)(k32 + j + 12);
                            rk += below(a, mine) + below(b, mine) + below(c, mine) + below(d, mine);
                        }
                        if (has) s_plist[rk] = (int)(mine & 0xffffu);
                    }
                    __syncthreads();
                    goto ranked;
                }
                for (int i = tid; i < P; i += OT3_THREADS) s_key[i] = ~0ull;
                __syncthreads();
                for (int i = tid; i < n; i += OT3_THREADS)
                    if (cur.cnt[i] > 1)
                        s_key[s_kk[i]] = ((unsigned long long)(0xffffffffu - (unsigned)cur.cnt[i]) << 32) | (unsigned)i;
                __syncthreads();
                for (int k = 2; k <= P; k <<= 1) {
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int i = tid; i < P; i += OT3_THREADS) {
                            const int ixj = i ^ j;
                            if (ixj > i) {
                                const unsigned long long a = s_key[i], b = s_key[ixj];
                                const bool up = ((i & k) == 0);
                                if ((a > b) == up) { s_key[i] = b; s_key[ixj] = a; }
                            }
                        }
                        __syncthreads();
                    }
                }
                for (int i = tid; i < m; i += OT3_THREADS) s_plist[i] = (int)(s_key[i] & 0xffffffffu);
                __syncthreads();
            ranked:;
            }
            if (sorted_phase) OT3_PHASE(); // ranked
            // k = non-empty children per processing rank; exclusive prefix in s_un
            for (int r = tid; r < m; r += OT3_THREADS) {
                const int i = s_plist[r];
                const int k = (s_ccnt[4 * i] > 0) + (s_ccnt[4 * i + 1] > 0) + (s_ccnt[4 * i + 2] > 0) + (s_ccnt[4 * i + 3] > 0);
                s_kk[r] = k;
                s_un[r] = k;
            }
            __syncthreads();
            ot3_scan_array(s_un, m, s_w);
            if (sorted_phase) OT3_PHASE(); // scan un
            if (tid == 0) { *s_nproc = m; *s_nexpand = 0; }
            __syncthreads();
            if (sorted_phase) { // first r with n + sum_{r'<=r}(k-1) >= quota (src/ORBextractor.cc:724-725); monotone in r
                for (int r = tid; r < m; r += OT3_THREADS)
                    if (n + s_un[r] + s_kk[r] - (r + 1) >= quota) atomicMin(s_nproc, r + 1);
                __syncthreads();
            }
            if (tid == 0) {
                const int nproc = *s_nproc;
                *s_total_k = nproc > 0 ? s_un[nproc - 1] + s_kk[nproc - 1] : 0;
            }
            __syncthreads();
            const int nproc = *s_nproc, total_k = *s_total_k;
            if (sorted_phase) OT3_PHASE(); // nproc
            for (int r = tid; r < nproc; r += OT3_THREADS) s_rank[s_plist[r]] = r;
            __syncthreads();
            for (int i = tid; i < n; i += OT3_THREADS) s_plist[i] = (s_rank[i] < 0) ? 1 : 0; // reuse: unprocessed flags
            __syncthreads();
            const int n_un = ot3_scan_array(s_plist, n, s_w);
            if (sorted_phase) OT3_PHASE(); // scan flags
            n_new = total_k + n_un;
            if (n_new > MAXN) { // cannot happen for max_nodes >= max(quota+3, 4*n_ini); guard anyway
                if (tid == 0) { *status = 2; *sel_cnt = 0; }
                return;
            }
            // new node array in list order: blocks of later-processed parents nearer the front, children n4..n1
            for (int i = tid; i < n; i += OT3_THREADS) {
                const int r = s_rank[i];
                if (r < 0) {
                    copy_node(total_k + s_plist[i], i);
                    continue;
                }
                const int c0 = s_ccnt[4 * i], c1 = s_ccnt[4 * i + 1], c2 = s_ccnt[4 * i + 2], c3 = s_ccnt[4 * i + 3];
                const int k = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
                int q = total_k - (s_un[r] + k);
                if (c3 > 0) emit(q++, i, 3, c3);
                if (c2 > 0) emit(q++, i, 2, c2);
                if (c1 > 0) emit(q++, i, 1, c1);
                if (c0 > 0) emit(q++, i, 0, c0);
                const int nexp = (c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1);
                if (nexp) atomicAdd(s_nexpand, nexp);
            }
        }
        __syncthreads();
        // stop logic (src/ORBextractor.cc:661-731)
        if (tid == 0) {
            const int prev = n;
            *s_n = n_new;
            *s_deep = 0;
            if (n_new >= quota || n_new == prev) *s_done = 1;
            else if (!sorted_phase && n_new + 3 * (nexpand_fast >= 0 ? nexpand_fast : *s_nexpand) > quota) *s_mode = 1;
            else *s_mode = sorted_phase;
        }
        __syncthreads();
        { Ot3Nodes t = cur; cur = nxt; nxt = t; }
        if (*s_done) break;
        sorted_phase = *s_mode;
        __syncthreads();
    }

    OT3_PHASE();
    // ---- 4. keep the best response per node, first wins (src/ORBextractor.cc:735-754) ----
    const int n = *s_n;
    const int n_out = n < L.sel_cap ? n : L.sel_cap;
    if (n > L.sel_cap && tid == 0) *status = 3;
    // ---- 4a. describe_kernel's processing order: the level's kept nodes counting-sorted into row-major order of coarse boxes --
    //      bin = (top po_rb row bits of the node's quadrant path, root, top po_cb column bits): rows ~40 px tall, columns as fine as 256
    //      bins allow; order inside a bin as the atomics fall (the order of PROCESSING changes no result: every keypoint is written
    //      to its own slot).  Keypoints that share cache lines then run in the same describe_kernel workgroups, which is bound by
    //      its L1 fills.  One barrier: every wave scans all 256 bins for itself (same values from every wave). ----
    const bool ordered = !NODES_IN_HBM && cfg.proc_order;
    unsigned key_pre = 0u; // the best key of the thread's first node: requested now, so that the ordering step runs under this load's latency
    if (tid < n_out) {
        const int dr = cur.dr[tid], d = dr & 15;
        if (d <= db) key_pre = g_best[ot3_off(d) + ((dr >> 4) << (2 * d)) + (int)cur.path[tid]];
    }
    if (ordered) {
        const int rb = L.po_rb, cb = L.po_cb;
        int *s_base = s_chi; // the count pyramid is dead (16-byte aligned: the start of the dynamic LDS)
        for (int i = tid; i < n_out; i += OT3_THREADS) {
            const int dr = cur.dr[i], d = dr & 15, root = dr >> 4;
            unsigned path = cur.path[i]; // x bits even, y bits odd, the first split highest
            path = d >= 6 ? path >> (2 * (d - 6)) : path << (2 * (6 - d)); // six levels of it
            const unsigned x6 = ((path >> 5) & 32u) | ((path >> 4) & 16u) | ((path >> 3) & 8u) | ((path >> 2) & 4u) | ((path >> 1) & 2u) | (path & 1u);
            const unsigned y6 = ((path >> 6) & 32u) | ((path >> 5) & 16u) | ((path >> 4) & 8u) | ((path >> 3) & 4u) | ((path >> 2) & 2u) | ((path >> 1) & 1u);
            const int bin = (int)((((y6 >> (6 - rb)) * OT3_ROOTS + (unsigned)root) << cb) | (x6 >> (6 - cb))); // (alternating the direction from row to row: no difference)
            s_rank[i] = bin | (atomicAdd(&s_bin[bin], 1) << 8);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // LDS only: not the wait for key_pre that __syncthreads() would add
        { // exclusive scan of the 256 bins by every wave: lane l owns bins 4 l .. 4 l + 3
            const int lane = tid & 63;
            const int4 v = ((const int4 *)s_bin)[lane];
            const int sum = v.x + v.y + v.z + v.w;
            const int ex = ot3_wave_incl_scan(sum, lane) - sum;
            ((int4 *)s_base)[lane] = make_int4(ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z); // every wave stores the same values
            __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the wave's own stores
            __builtin_amdgcn_wave_barrier();
        }
    }
    uint32_t *proc_xy = buf.proc_xy + ib * cfg.sel_total + L.sel_off, *proc_meta = buf.proc_meta + ib * cfg.sel_total + L.sel_off;
    for (int i = tid; i < n_out; i += OT3_THREADS) {
        const int dr = cur.dr[i], d = dr & 15, root = dr >> 4;
        const unsigned path = cur.path[i];
        unsigned key = 0u;
        if (d <= db) {
            key = i == tid ? key_pre : g_best[ot3_off(d) + (root << (2 * d)) + (int)path];
        } else { // only reachable through deep_children, i.e. with the sorted arrays built
            const int b = (root << (2 * db)) + (int)(path >> (2 * (d - db)));
            for (int j = s_bend[b]; j < s_bend[b + 1]; j++) {
                const uint32_t xy = deep_xy[j];
                int r;
                const unsigned pp = ot3_path((int)(xy & 0xffffu), (int)(xy >> 16), d, hx, n_ini, region_h, r);
                const unsigned kj = deep_key[j];
                if (pp == path && kj > key) key = kj;
            }
        }
        const unsigned ref = OT3_REF_MASK - (key & OT3_REF_MASK);
        const uint32_t xy = cell_xy[(size_t)(ref >> 12) * cfg.cell_cap + (ref & 4095u)];
        sel_xy[i] = xy;
        sel_sc[i] = (uint8_t)(key >> 24);
        if (cfg.proc_order) {
            int pos = i;
            if (ordered) { const int r = s_rank[i]; pos = s_chi[r & 255] + (r >> 8); }
            proc_xy[pos] = xy;
            proc_meta[pos] = (uint32_t)(L.sel_off + i) | (key & 0xff000000u);
        }
    }
    if (tid == 0) *sel_cnt = n_out;
    OT3_PHASE();
}

// Parity tap (orbfe_fetch_candidates): the candidates of every level in cv::FAST emission order
// (cell-row-major, in-cell order).  One workgroup per (image, level); not part of the frame path.
__global__ __launch_bounds__(OT3_THREADS) void candidates_gather_kernel(DeviceConfig cfg, DeviceBuffers buf)
{
    __shared__ int s_w[OT3_WAVES];
    const int level = blockIdx.x, img = blockIdx.y;
    const LevelInfo &L = cfg.lv[level];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t ib = (size_t)img;
    const int *cell_cnt = buf.cell_cnt + ib * cfg.cells_total + L.cell_off;
    int *cell_base = buf.cell_base + ib * cfg.cells_total + L.cell_off;
    const uint32_t *cell_xy = buf.cell_xy + (ib * cfg.cells_total + L.cell_off) * cfg.cell_cap;
    const uint8_t *cell_sc = buf.cell_sc + (ib * cfg.cells_total + L.cell_off) * cfg.cell_cap;
    const size_t coff = ib * cfg.cand_total + L.cand_off;
    const int n = L.n_cells;
    const int per = (n + OT3_THREADS - 1) / OT3_THREADS;
    const int b = tid * per < n ? tid * per : n, e = (b + per < n) ? b + per : n;
    int sum = 0;
    for (int i = b; i < e; i++) sum += cell_cnt[i];
    const int inc = ot3_wave_incl_scan(sum, lane);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < OT3_WAVES; w++) {
        const int x = s_w[w];
        if (w < wave) base += x;
        total += x;
    }
    int run = base + inc - sum;
    for (int i = b; i < e; i++) {
        const int v = cell_cnt[i];
        cell_base[i] = run;
        run += v;
    }
    __syncthreads();
    int nc = total;
    if (nc > L.cand_cap) nc = L.cand_cap;
    if (tid == 0) buf.lvl_ncand[ib * cfg.nlevels + level] = nc;
    for (int c = wave; c < n; c += OT3_WAVES) {
        const int cb = cell_base[c], cnt = cell_cnt[c];
        for (int k = lane; k < cnt; k += 64)
            if (cb + k < nc) {
                buf.cand_xy[coff + cb + k] = cell_xy[(size_t)c * cfg.cell_cap + k];
                buf.cand_sc[coff + cb + k] = cell_sc[(size_t)c * cfg.cell_cap + k];
            }
    }
}

void orbfe_launch_octree3(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, int sort_cap, size_t lds, bool nodes_in_hbm, hipStream_t s)
{
    dim3 grid(n_images, cfg.nlevels);
    const size_t node_bytes = orbfe_octree3_node_bytes(cfg.max_nodes, sort_cap);
    if (nodes_in_hbm) hipLaunchKernelGGL(octree3_kernel<true>, grid, dim3(OT3_THREADS), lds, s, cfg, buf, sort_cap, node_bytes);
    else hipLaunchKernelGGL(octree3_kernel<false>, grid, dim3(OT3_THREADS), lds, s, cfg, buf, sort_cap, node_bytes);
}

void orbfe_launch_candidates_gather(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, hipStream_t s)
{
    dim3 grid(cfg.nlevels, n_images);
    hipLaunchKernelGGL(candidates_gather_kernel, grid, dim3(OT3_THREADS), 0, s, cfg, buf);
}

int orbfe_octree3_prepare(size_t lds, bool nodes_in_hbm)
{
    if (lds <= 64 * 1024) return 0;
    const void *f = nodes_in_hbm ? (const void *)octree3_kernel<true> : (const void *)octree3_kernel<false>;
    return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 0 : -1;
}
