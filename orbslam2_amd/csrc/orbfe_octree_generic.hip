// orbfe_octree_generic.hip -- DistributeOctTree (src/ORBextractor.cc:533-757), generic node-parallel kernel: the fallback of orbfe_octree3.hip (geometries beyond the bucket-pyramid kernel's limits).
#include "orbfe_common.hpp"

#define OT_THREADS 512

// ---------------------------------------------------------------------------
// DistributeOctTree, generic node-parallel kernel (any n_ini; fallback of orbfe_octree3.hip): one workgroup per (image, level)
// ---------------------------------------------------------------------------
// Array formulation validated on the CPU by tests/octree_model.py:
//  * nodes live in an array kept in std::list order (front -> back);
//  * a pass splits a set of multi-point nodes; their non-empty children are written
//    n4,n3,n2,n1 at the front, blocks of later-processed parents nearer the front;
//  * every node owns a contiguous segment [beg, beg+cnt) of a candidate-index
//    permutation (two ping-pong buffers); a split is a stable 4-way partition of the
//    segment, so the per-node point order stays the FAST emission order;
//  * the "expand the biggest node first" phase sorts on (count desc, position asc),
//    which equals the reference's (size, pointer) ordering under contract Q3.
struct OtNodes {
    short *x0, *y0, *x1, *y1;
    int *beg, *cnt;
    uint8_t *bf;
};

__device__ __forceinline__ void ot_bind(OtNodes &n, uint8_t *&p, int cap)
{
    n.beg = (int *)p; p += sizeof(int) * cap;
    n.cnt = (int *)p; p += sizeof(int) * cap;
    n.x0 = (short *)p; p += sizeof(short) * cap;
    n.y0 = (short *)p; p += sizeof(short) * cap;
    n.x1 = (short *)p; p += sizeof(short) * cap;
    n.y1 = (short *)p; p += sizeof(short) * cap;
    n.bf = p; p += ((cap + 7) / 8) * 8;
}

__global__ __launch_bounds__(OT_THREADS) void octree_generic_kernel(DeviceConfig cfg, DeviceBuffers buf, int sort_cap)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
    __shared__ int s_scan[OT_THREADS];
    __shared__ int s_n, s_total_k, s_nproc, s_nexpand, s_mode, s_done;
    const int level = blockIdx.x, img = blockIdx.y;
    const LevelInfo &L = cfg.lv[level];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = OT_THREADS / 64;
    const int MAXN = cfg.max_nodes;

    uint8_t *p = s_raw;
    unsigned long long *s_key = (unsigned long long *)p; p += sizeof(unsigned long long) * sort_cap;
    OtNodes A, B;
    ot_bind(A, p, MAXN);
    ot_bind(B, p, MAXN);
    int *s_ccnt = (int *)p; p += sizeof(int) * 4 * MAXN;   // child counts per old node
    int *s_rank = (int *)p; p += sizeof(int) * MAXN;       // processing rank of old node (-1: not processed)
    int *s_plist = (int *)p; p += sizeof(int) * MAXN;      // processing order -> old node
    int *s_kk = (int *)p; p += sizeof(int) * MAXN;         // scan scratch
    int *s_un = (int *)p; p += sizeof(int) * MAXN;         // scan scratch (unprocessed flags)

    const size_t ib = (size_t)img;
    int *cell_cnt = buf.cell_cnt + ib * cfg.cells_total + L.cell_off;
    int *cell_base = buf.cell_base + ib * cfg.cells_total + L.cell_off;
    const uint32_t *cell_xy = buf.cell_xy + (ib * cfg.cells_total + L.cell_off) * cfg.cell_cap;
    const uint8_t *cell_sc = buf.cell_sc + (ib * cfg.cells_total + L.cell_off) * cfg.cell_cap;
    uint32_t *cxy = buf.cand_xy + ib * cfg.cand_total + L.cand_off;
    uint8_t *csc = buf.cand_sc + ib * cfg.cand_total + L.cand_off;
    uint32_t *idx[2] = {buf.idx0 + ib * cfg.cand_total + L.cand_off, buf.idx1 + ib * cfg.cand_total + L.cand_off};
    int *sel_cnt = buf.sel_cnt + ib * cfg.nlevels + level;
    uint32_t *sel_xy = buf.sel_xy + ib * cfg.sel_total + L.sel_off;
    uint8_t *sel_sc = buf.sel_sc + ib * cfg.sel_total + L.sel_off;

    // ---- gather the per-cell candidates into emission order ----
    int nc = block_excl_scan(cell_cnt, cell_base, L.n_cells, s_scan);
    if (nc > L.cand_cap) { nc = L.cand_cap; if (tid == 0) buf.status[img] = 1; }
    if (tid == 0) buf.lvl_ncand[ib * cfg.nlevels + level] = nc;
    for (int i = tid; i < nc; i += OT_THREADS) {
        int lo = 0, hi = L.n_cells - 1; // last cell with cell_base <= i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (cell_base[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const int k = i - cell_base[lo];
        cxy[i] = cell_xy[(size_t)lo * cfg.cell_cap + k];
        csc[i] = cell_sc[(size_t)lo * cfg.cell_cap + k];
    }
    __syncthreads();
    if (nc == 0) {
        if (tid == 0) *sel_cnt = 0;
        return;
    }

    // ---- roots: stable partition by int(x / hX) (src/ORBextractor.cc:537-564) ----
    const int n_ini = L.n_ini;
    const int region_h = (L.h - cfg.edge_threshold + 3) - cfg.min_border;
    for (int i = tid; i < n_ini; i += OT_THREADS) s_kk[i] = 0;
    __syncthreads();
    for (int i = tid; i < nc; i += OT_THREADS) {
        int b = (int)__fdiv_rn((float)(cxy[i] & 0xffffu), L.hx);
        b = b < 0 ? 0 : (b >= n_ini ? n_ini - 1 : b);
        atomicAdd(&s_kk[b], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0, n = 0;
        for (int b = 0; b < n_ini; b++) {
            const int c = s_kk[b];
            s_un[b] = run; // segment begin of bucket b
            if (c > 0) {
                A.x0[n] = (short)(int)__fmul_rn(L.hx, (float)b);
                A.x1[n] = (short)(int)__fmul_rn(L.hx, (float)(b + 1));
                A.y0[n] = 0;
                A.y1[n] = (short)region_h;
                A.beg[n] = run; A.cnt[n] = c; A.bf[n] = 0;
                n++;
            }
            run += c;
        }
        s_n = n;
        s_done = 0;
    }
    __syncthreads();
    for (int b = wave; b < n_ini; b += nwaves) {
        int run = s_un[b];
        for (int i0 = 0; i0 < nc; i0 += 64) {
            const int i = i0 + lane;
            bool pred = false;
            if (i < nc) {
                int bb = (int)__fdiv_rn((float)(cxy[i] & 0xffffu), L.hx);
                bb = bb < 0 ? 0 : (bb >= n_ini ? n_ini - 1 : bb);
                pred = (bb == b);
            }
            const unsigned long long m = __ballot(pred);
            if (pred) idx[0][run + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
            run += __popcll(m);
        }
    }
    __syncthreads();

    // ---- split passes ----
    OtNodes cur = A, nxt = B;
    int sorted_phase = 0;
    for (int iter = 0; iter < 100000; iter++) { // n grows every pass, so this ends at n >= quota at the latest
        const int n = s_n;
        // (A) child counts of every multi-point node
        for (int i = wave; i < n; i += nwaves) {
            const int cnt = cur.cnt[i];
            if (cnt > 1) {
                const int mx = cur.x0[i] + ((cur.x1[i] - cur.x0[i] + 1) >> 1);
                const int my = cur.y0[i] + ((cur.y1[i] - cur.y0[i] + 1) >> 1);
                const uint32_t *src = idx[cur.bf[i]] + cur.beg[i];
                int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                for (int j = lane; j < cnt; j += 64) {
                    const uint32_t xy = cxy[src[j]];
                    const int cls = ((int)(xy & 0xffffu) < mx ? 0 : 1) + ((int)(xy >> 16) < my ? 0 : 2);
                    c0 += cls == 0; c1 += cls == 1; c2 += cls == 2; c3 += cls == 3;
                }
                c0 = wave_sum_i32(c0); c1 = wave_sum_i32(c1); c2 = wave_sum_i32(c2); c3 = wave_sum_i32(c3);
                if (lane == 0) { s_ccnt[4 * i] = c0; s_ccnt[4 * i + 1] = c1; s_ccnt[4 * i + 2] = c2; s_ccnt[4 * i + 3] = c3; }
            }
        }
        for (int i = tid; i < n; i += OT_THREADS) { s_rank[i] = -1; s_kk[i] = cur.cnt[i] > 1 ? 1 : 0; }
        __syncthreads();
        // (B) processing order
        int m;
        if (!sorted_phase) {
            m = block_excl_scan(s_kk, s_kk, n, s_scan); // s_kk[i] = rank among multi nodes
            for (int i = tid; i < n; i += OT_THREADS)
                if (cur.cnt[i] > 1) s_plist[s_kk[i]] = i;
            __syncthreads();
        } else {
            m = block_excl_scan(s_kk, s_kk, n, s_scan);
            int P = 1;
            while (P < m) P <<= 1;
            for (int i = tid; i < P; i += OT_THREADS) s_key[i] = ~0ull;
            __syncthreads();
            for (int i = tid; i < n; i += OT_THREADS)
                if (cur.cnt[i] > 1)
                    s_key[s_kk[i]] = ((unsigned long long)(0xffffffffu - (unsigned)cur.cnt[i]) << 32) | (unsigned)i;
            __syncthreads();
            for (int k = 2; k <= P; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < P; i += OT_THREADS) {
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
            for (int i = tid; i < m; i += OT_THREADS) s_plist[i] = (int)(s_key[i] & 0xffffffffu);
            __syncthreads();
        }
        // k (non-empty children) per processing rank; inclusive prefix decides the stop
        for (int r = tid; r < m; r += OT_THREADS) {
            const int i = s_plist[r];
            s_kk[r] = (s_ccnt[4 * i] > 0) + (s_ccnt[4 * i + 1] > 0) + (s_ccnt[4 * i + 2] > 0) + (s_ccnt[4 * i + 3] > 0);
        }
        __syncthreads();
        block_excl_scan(s_kk, s_un, m, s_scan); // s_un[r] = sum of k over ranks < r
        if (tid == 0) {
            int nproc = m;
            if (sorted_phase) {
                // first r with n + sum_{r'<=r}(k-1) >= quota (src/ORBextractor.cc:724-725)
                nproc = m;
                for (int r = 0; r < m; r++) {
                    const int incl = s_un[r] + s_kk[r];
                    if (n + incl - (r + 1) >= L.quota) { nproc = r + 1; break; }
                }
            }
            s_nproc = nproc;
            s_total_k = nproc > 0 ? s_un[nproc - 1] + s_kk[nproc - 1] : 0;
        }
        __syncthreads();
        const int nproc = s_nproc, total_k = s_total_k;
        for (int r = tid; r < nproc; r += OT_THREADS) s_rank[s_plist[r]] = r;
        __syncthreads();
        // unprocessed old nodes keep their relative order behind the new blocks
        for (int i = tid; i < n; i += OT_THREADS) s_plist[i] = (s_rank[i] < 0) ? 1 : 0; // reuse as flag array
        __syncthreads();
        const int n_un = block_excl_scan(s_plist, s_plist, n, s_scan);
        const int n_new = total_k + n_un;
        if (n_new > MAXN) { // cannot happen for max_nodes >= max(quota+3, 4*n_ini); guard anyway
            if (tid == 0) { buf.status[img] = 2; *sel_cnt = 0; }
            return;
        }
        if (tid == 0) s_nexpand = 0;
        __syncthreads();
        // (D) emit new node array + scatter the points of processed nodes
        for (int i = wave; i < n; i += nwaves) {
            const int r = s_rank[i];
            if (r < 0) {
                if (lane == 0) {
                    const int q = total_k + s_plist[i];
                    nxt.x0[q] = cur.x0[i]; nxt.y0[q] = cur.y0[i]; nxt.x1[q] = cur.x1[i]; nxt.y1[q] = cur.y1[i];
                    nxt.beg[q] = cur.beg[i]; nxt.cnt[q] = cur.cnt[i]; nxt.bf[q] = cur.bf[i];
                }
                continue;
            }
            const int x0 = cur.x0[i], y0 = cur.y0[i], x1 = cur.x1[i], y1 = cur.y1[i];
            const int mx = x0 + ((x1 - x0 + 1) >> 1);
            const int my = y0 + ((y1 - y0 + 1) >> 1);
            const int cnt = cur.cnt[i], beg = cur.beg[i], sb = cur.bf[i];
            const int c0 = s_ccnt[4 * i], c1 = s_ccnt[4 * i + 1], c2 = s_ccnt[4 * i + 2], c3 = s_ccnt[4 * i + 3];
            const int k = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
            // block of this parent starts after the blocks of all later-processed parents
            int q = total_k - (s_un[r] + k);
            if (lane == 0) {
                int nexp = 0;
                const int b0 = beg, b1 = beg + c0, b2 = b1 + c1, b3 = b2 + c2;
                if (c3 > 0) { nxt.x0[q] = (short)mx; nxt.y0[q] = (short)my; nxt.x1[q] = (short)x1; nxt.y1[q] = (short)y1; nxt.beg[q] = b3; nxt.cnt[q] = c3; nxt.bf[q] = (uint8_t)(1 - sb); q++; nexp += c3 > 1; }
                if (c2 > 0) { nxt.x0[q] = (short)x0; nxt.y0[q] = (short)my; nxt.x1[q] = (short)mx; nxt.y1[q] = (short)y1; nxt.beg[q] = b2; nxt.cnt[q] = c2; nxt.bf[q] = (uint8_t)(1 - sb); q++; nexp += c2 > 1; }
                if (c1 > 0) { nxt.x0[q] = (short)mx; nxt.y0[q] = (short)y0; nxt.x1[q] = (short)x1; nxt.y1[q] = (short)my; nxt.beg[q] = b1; nxt.cnt[q] = c1; nxt.bf[q] = (uint8_t)(1 - sb); q++; nexp += c1 > 1; }
                if (c0 > 0) { nxt.x0[q] = (short)x0; nxt.y0[q] = (short)y0; nxt.x1[q] = (short)mx; nxt.y1[q] = (short)my; nxt.beg[q] = b0; nxt.cnt[q] = c0; nxt.bf[q] = (uint8_t)(1 - sb); q++; nexp += c0 > 1; }
                if (nexp) atomicAdd(&s_nexpand, nexp);
            }
            const uint32_t *src = idx[sb] + beg;
            uint32_t *dst = idx[1 - sb];
            int r0 = beg, r1 = beg + c0, r2 = r1 + c1, r3 = r2 + c2;
            const unsigned long long lt = (1ull << lane) - 1ull;
            for (int j0 = 0; j0 < cnt; j0 += 64) {
                const int j = j0 + lane;
                int cls = -1;
                uint32_t id = 0;
                if (j < cnt) {
                    id = src[j];
                    const uint32_t xy = cxy[id];
                    cls = ((int)(xy & 0xffffu) < mx ? 0 : 1) + ((int)(xy >> 16) < my ? 0 : 2);
                }
                const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1),
                                         m2 = __ballot(cls == 2), m3 = __ballot(cls == 3);
                if (cls == 0) dst[r0 + __popcll(m0 & lt)] = id;
                else if (cls == 1) dst[r1 + __popcll(m1 & lt)] = id;
                else if (cls == 2) dst[r2 + __popcll(m2 & lt)] = id;
                else if (cls == 3) dst[r3 + __popcll(m3 & lt)] = id;
                r0 += __popcll(m0); r1 += __popcll(m1); r2 += __popcll(m2); r3 += __popcll(m3);
            }
        }
        __syncthreads();
        // (E) stop logic (src/ORBextractor.cc:661-731)
        if (tid == 0) {
            const int prev = n;
            s_n = n_new;
            if (n_new >= L.quota || n_new == prev) s_done = 1;
            else if (!sorted_phase && n_new + 3 * s_nexpand > L.quota) s_mode = 1;
            else s_mode = sorted_phase;
        }
        __syncthreads();
        { OtNodes t = cur; cur = nxt; nxt = t; }
        if (s_done) break;
        sorted_phase = s_mode;
        __syncthreads();
    }

    // ---- keep the best response per node, first wins (src/ORBextractor.cc:735-754) ----
    const int n = s_n;
    int n_out = n < L.sel_cap ? n : L.sel_cap;
    if (n > L.sel_cap && tid == 0) buf.status[img] = 3;
    for (int i = wave; i < n_out; i += nwaves) {
        const uint32_t *src = idx[cur.bf[i]] + cur.beg[i];
        const int cnt = cur.cnt[i];
        unsigned best = 0xffffffffu; // (255-score)<<24 | position in node  (cnt < 2^24)
        for (int j = lane; j < cnt; j += 64) {
            const unsigned key = ((unsigned)(255 - csc[src[j]]) << 24) | (unsigned)j;
            best = key < best ? key : best;
        }
        best = wave_min_u32(best);
        if (lane == 0) {
            const uint32_t id = src[best & 0xffffffu];
            sel_xy[i] = cxy[id];
            sel_sc[i] = csc[id];
        }
    }
    if (tid == 0) *sel_cnt = n_out;
}


static inline int ot_sort_cap(const DeviceConfig &cfg) { int p = 1; while (p < cfg.max_nodes) p <<= 1; return p; }

size_t orbfe_octree_lds_bytes(const DeviceConfig &cfg)
{
    const int cap = cfg.max_nodes;
    const size_t node = 2 * sizeof(int) * cap + 4 * sizeof(short) * cap + ((cap + 7) / 8) * 8;
    return sizeof(unsigned long long) * ot_sort_cap(cfg) + 2 * node + sizeof(int) * 4 * cap + 4 * sizeof(int) * cap;
}

void orbfe_launch_octree_generic(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, hipStream_t s)
{
    dim3 grid(cfg.nlevels, n_images);
    // node tables beyond the default 64 KB of dynamic LDS (wide-aspect images with large quotas); the attribute is per
    // device, so it is remembered per device ordinal
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void *)octree_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(octree_generic_kernel, grid, dim3(OT_THREADS), orbfe_octree_lds_bytes(cfg), s, cfg, buf, ot_sort_cap(cfg));
}
