// orbfe_octree.hip -- point-parallel DistributeOctTree (reference src/ORBextractor.cc:533-757).
//
// One 512-thread workgroup per (image, level).  Same array formulation as the generic kernel in
// orbfe_octree_generic.hip (validated on the CPU by tests/octree_model.py), but the per-pass work is
// parallel over POINTS instead of over nodes:
//   * the candidates of the level live in one position-ordered array of (candidate id | owning node
//     << 16) words; a node owns a contiguous range of positions.  When the level has at most
//     `lds_pts` candidates the array and the id-indexed xy / score tables sit in LDS (single buffer:
//     a thread stages its <= 16 positions in registers before the scatter); bigger levels use
//     ping-pong arrays in HBM (L2 resident);
//   * a pass is two "chunk walks": thread t owns positions [t*per, (t+1)*per).  Walk 1 classifies
//     every point of a multi-point node into its quadrant, counts classes per chunk and per node;
//     a block scan of the chunk totals gives E(j)[c] = #points of class c before position j, so the
//     stable rank of a point inside its (node, class) is E(j)[c] - E(node begin)[c];
//   * node bookkeeping (which nodes are split this pass, list order of the children, the
//     "largest node first" phase with its (count desc, list position asc) sort) is done per node by
//     one thread each; walk 2 scatters every point to its child range (unsplit nodes stay in place).
// Points are never visited through per-node serial chains, so a pass costs O(nc / 512) per thread
// plus a handful of barriers, whatever the node structure.
#include "orbfe_device.h"
#include <cstdlib>

#define OT2_THREADS 512
#define OT2_WAVES (OT2_THREADS / 64)
#define OT2_ID_BITS 20 // position word = candidate id (20 bits) | owning node (12 bits)
#define OT2_ID_MASK 0xfffffu
// bring-up aid: thread 0 of (image 0, the level in g_ts_level) stamps the shader clock
#define OT2_TS(slot) do { if (ts && threadIdx.x == 0) ts[(slot)] = clock64(); } while (0)

__device__ __forceinline__ int ot2_wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// exclusive block scan of 4 values per thread; totals in tot[4].  s_w: 4*OT2_WAVES ints.
__device__ __forceinline__ void ot2_block_scan4(int v[4], int tot[4], int *s_w)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) inc[c] = ot2_wave_incl_scan(v[c], lane);
    if (lane == 63) {
#pragma unroll
        for (int c = 0; c < 4; c++) s_w[4 * wave + c] = inc[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; c++) {
        int base = 0, t = 0;
        for (int w = 0; w < OT2_WAVES; w++) {
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
__device__ __forceinline__ int ot2_scan_array(int *a, int n, int *s_w)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + OT2_THREADS - 1) / OT2_THREADS;
    const int b = tid * per, e = (b + per < n) ? b + per : n;
    int sum = 0;
    for (int i = b; i < e; i++) sum += a[i];
    const int inc = ot2_wave_incl_scan(sum, lane);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < OT2_WAVES; w++) {
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

struct Ot2Nodes {
    int *beg, *cnt;
    short *x0, *y0, *x1, *y1;
};

__device__ __forceinline__ void ot2_bind(Ot2Nodes &n, uint8_t *&p, int cap)
{
    n.beg = (int *)p; p += sizeof(int) * cap;
    n.cnt = (int *)p; p += sizeof(int) * cap;
    n.x0 = (short *)p; p += sizeof(short) * cap;
    n.y0 = (short *)p; p += sizeof(short) * cap;
    n.x1 = (short *)p; p += sizeof(short) * cap;
    n.y1 = (short *)p; p += sizeof(short) * cap;
    p = (uint8_t *)(((uintptr_t)p + 7) & ~(uintptr_t)7);
}

size_t orbfe_octree2_lds_bytes(int max_nodes, int sort_cap, int lds_pts)
{
    const size_t cap = (size_t)max_nodes;
    const size_t node = 2 * sizeof(int) * cap + 4 * sizeof(short) * cap + 8;
    return sizeof(unsigned long long) * sort_cap + 2 * node + sizeof(int) * cap * (4 + 4 + 4 + 1 + 1 + 1 + 1 + 1 + 1) +
           sizeof(int) * 4 * OT2_THREADS + (size_t)lds_pts * 9 + 64;
}

struct Ot2Ctx {
    // LDS tables
    unsigned long long *s_key;
    Ot2Nodes A, B;
    int *s_ccnt, *s_nbl, *s_newidx, *s_nbt, *s_newun, *s_rank, *s_plist, *s_kk, *s_un, *s_ct, *s_w;
    int *s_n, *s_total_k, *s_nproc, *s_nexpand, *s_mode, *s_done;
    int sort_cap, max_nodes;
};

// LDSP: point array + xy/score tables in LDS (nc <= 16 * OT2_THREADS); else ping-pong in HBM.
template <bool LDSP>
__device__ __forceinline__ void ot2_body(long long *ts, const Ot2Ctx &K, const LevelInfo &L, int region_h, int nc, int quota,
                         const uint32_t *xy_tab, const uint8_t *sc_tab, // indexed by candidate id
                         uint32_t *ip_c, uint32_t *ip_n,                // (id | node << 16) by position
                         uint32_t *sel_xy, uint8_t *sel_sc, int *sel_cnt, int *status)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int MAXN = K.max_nodes;
    int *s_ccnt = K.s_ccnt, *s_nbl = K.s_nbl, *s_newidx = K.s_newidx, *s_nbt = K.s_nbt, *s_newun = K.s_newun;
    int *s_rank = K.s_rank, *s_plist = K.s_plist, *s_kk = K.s_kk, *s_un = K.s_un, *s_ct = K.s_ct, *s_w = K.s_w;
    unsigned long long *s_key = K.s_key;
    const int per = (nc + OT2_THREADS - 1) / OT2_THREADS; // LDSP: <= 16
    const int jb = tid * per < nc ? tid * per : nc;
    const int je = jb + per < nc ? jb + per : nc;

    // ---- roots: stable partition by int(x / hX) (src/ORBextractor.cc:537-564); n_ini <= 4 here ----
    const int n_ini = L.n_ini;
    {
        int lc[4] = {0, 0, 0, 0};
        for (int j = jb; j < je; j++) {
            int b = (int)__fdiv_rn((float)(xy_tab[j] & 0xffffu), L.hx);
            b = b < 0 ? 0 : (b >= n_ini ? n_ini - 1 : b);
            lc[0] += b == 0; lc[1] += b == 1; lc[2] += b == 2; lc[3] += b == 3;
        }
        int tot[4];
        ot2_block_scan4(lc, tot, s_w); // lc = E(chunk start)
        if (tid == 0) {
            int run = 0, n = 0;
            for (int b = 0; b < n_ini; b++) {
                s_un[b] = run;   // first position of bucket b
                s_kk[b] = n;     // node index of bucket b (valid if non-empty)
                if (tot[b] > 0) {
                    K.A.x0[n] = (short)(int)__fmul_rn(L.hx, (float)b);
                    K.A.x1[n] = (short)(int)__fmul_rn(L.hx, (float)(b + 1));
                    K.A.y0[n] = 0;
                    K.A.y1[n] = (short)region_h;
                    K.A.beg[n] = run; K.A.cnt[n] = tot[b];
                    n++;
                }
                run += tot[b];
            }
            *K.s_n = n;
            *K.s_done = 0;
        }
        __syncthreads();
        for (int j = jb; j < je; j++) { // emission order: id == position
            int b = (int)__fdiv_rn((float)(xy_tab[j] & 0xffffu), L.hx);
            b = b < 0 ? 0 : (b >= n_ini ? n_ini - 1 : b);
            const int r = b == 0 ? lc[0]++ : (b == 1 ? lc[1]++ : (b == 2 ? lc[2]++ : lc[3]++));
            ip_c[s_un[b] + r] = (uint32_t)j | ((uint32_t)s_kk[b] << OT2_ID_BITS);
        }
        __syncthreads();
    }

    OT2_TS(3);
    // ---- split passes ----
    Ot2Nodes cur = K.A, nxt = K.B;
    int sorted_phase = 0;
    for (int iter = 0; iter < 100000; iter++) { // n grows every pass, so this ends at n >= quota at the latest
        const int n = *K.s_n;
        if (iter < 20) OT2_TS(8 + 8 * iter);
        for (int i = tid; i < 4 * n; i += OT2_THREADS) s_ccnt[i] = 0;
        for (int i = tid; i < n; i += OT2_THREADS) { s_rank[i] = -1; s_kk[i] = cur.cnt[i] > 1 ? 1 : 0; }
        __syncthreads();
        // (1) walk 1: classify, count per chunk / per node.  LDS mode: the chunk's position words and
        //     xy are first staged in registers with back-to-back LDS reads, so the serial node-run logic
        //     below never waits on a dependent LDS round trip per point.
        int lc[4] = {0, 0, 0, 0};
        uint32_t stage[16], sxy[16];
        if (LDSP) {
#pragma unroll
            for (int k = 0; k < 16; k++) stage[k] = (jb + k < je) ? ip_c[jb + k] : 0u;
#pragma unroll
            for (int k = 0; k < 16; k++) sxy[k] = xy_tab[stage[k] & OT2_ID_MASK];
        }
        {
            int run_nd = -1, mx = 0, my = 0, multi = 0;
            int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
            auto flush = [&]() {
                if (multi) {
                    if (r0) atomicAdd(&s_ccnt[4 * run_nd], r0);
                    if (r1) atomicAdd(&s_ccnt[4 * run_nd + 1], r1);
                    if (r2) atomicAdd(&s_ccnt[4 * run_nd + 2], r2);
                    if (r3) atomicAdd(&s_ccnt[4 * run_nd + 3], r3);
                }
            };
            // final flush: while nodes are few and large a whole wave sits inside one node and 64 lanes would
            // serialise on the same four LDS counters; reduce across the wave first in that case
            auto flush_final = [&]() {
                const int first = __builtin_amdgcn_readfirstlane(run_nd);
                if (__all(run_nd == first)) {
                    int a0 = multi ? r0 : 0, a1 = multi ? r1 : 0, a2 = multi ? r2 : 0, a3 = multi ? r3 : 0;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64);
                        a2 += __shfl_xor(a2, o, 64); a3 += __shfl_xor(a3, o, 64);
                    }
                    if ((threadIdx.x & 63) == 0 && first >= 0) {
                        if (a0) atomicAdd(&s_ccnt[4 * first], a0);
                        if (a1) atomicAdd(&s_ccnt[4 * first + 1], a1);
                        if (a2) atomicAdd(&s_ccnt[4 * first + 2], a2);
                        if (a3) atomicAdd(&s_ccnt[4 * first + 3], a3);
                    }
                } else {
                    flush();
                }
            };
            auto step1 = [&](int j, uint32_t ip, uint32_t xy_staged) {
                const int nd = (int)(ip >> OT2_ID_BITS);
                if (nd != run_nd) {
                    flush();
                    run_nd = nd; r0 = r1 = r2 = r3 = 0;
                    multi = cur.cnt[nd] > 1;
                    mx = cur.x0[nd] + ((cur.x1[nd] - cur.x0[nd] + 1) >> 1);
                    my = cur.y0[nd] + ((cur.y1[nd] - cur.y0[nd] + 1) >> 1);
                    if (multi && j == cur.beg[nd]) {
                        s_nbl[4 * nd] = lc[0]; s_nbl[4 * nd + 1] = lc[1]; s_nbl[4 * nd + 2] = lc[2]; s_nbl[4 * nd + 3] = lc[3];
                        s_nbt[nd] = tid;
                    }
                }
                if (multi) {
                    const uint32_t xy = LDSP ? xy_staged : xy_tab[ip & OT2_ID_MASK];
                    const int cls = ((int)(xy & 0xffffu) < mx ? 0 : 1) + ((int)(xy >> 16) < my ? 0 : 2);
                    lc[0] += cls == 0; lc[1] += cls == 1; lc[2] += cls == 2; lc[3] += cls == 3;
                    r0 += cls == 0; r1 += cls == 1; r2 += cls == 2; r3 += cls == 3;
                }
            };
            if (LDSP) {
#pragma unroll
                for (int k = 0; k < 16; k++)
                    if (jb + k < je) step1(jb + k, stage[k], sxy[k]);
            } else {
#pragma unroll 4
                for (int j = jb; j < je; j++) step1(j, ip_c[j], 0u);
            }
            flush_final();
        }
        if (iter < 20) OT2_TS(9 + 8 * iter);
        int tot4[4];
        ot2_block_scan4(lc, tot4, s_w); // lc = E(chunk start)[c]
        s_ct[4 * tid] = lc[0]; s_ct[4 * tid + 1] = lc[1]; s_ct[4 * tid + 2] = lc[2]; s_ct[4 * tid + 3] = lc[3];
        int n_new, nexpand_fast = -1;
        if (!sorted_phase && n <= OT2_THREADS) {
            // Streamlined full pass (every multi-point node is split, list order = processing order): thread i
            // owns node i and one 4-value block scan yields its rank, the children ahead of it, the unsplit nodes
            // ahead of it and the pass totals -- 7 barriers per pass instead of 18.
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
            ot2_block_scan4(v4, tot, s_w); // exclusive prefixes in list order
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
                    const int q = total_k + v4[2];
                    nxt.x0[q] = cur.x0[i]; nxt.y0[q] = cur.y0[i]; nxt.x1[q] = cur.x1[i]; nxt.y1[q] = cur.y1[i];
                    nxt.beg[q] = cur.beg[i]; nxt.cnt[q] = cur.cnt[i];
                    s_newun[i] = q;
                    s_rank[i] = -1;
                } else {
                    s_rank[i] = v4[0];
                    const int x0 = cur.x0[i], y0 = cur.y0[i], x1 = cur.x1[i], y1 = cur.y1[i];
                    const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
                    const int beg = cur.beg[i];
                    int q = total_k - (v4[1] + k);
                    const int b0 = beg, b1 = beg + c0, b2 = b1 + c1, b3 = b2 + c2;
                    if (c3 > 0) { nxt.x0[q] = (short)mx; nxt.y0[q] = (short)my; nxt.x1[q] = (short)x1; nxt.y1[q] = (short)y1; nxt.beg[q] = b3; nxt.cnt[q] = c3; s_newidx[4 * i + 3] = q; q++; }
                    if (c2 > 0) { nxt.x0[q] = (short)x0; nxt.y0[q] = (short)my; nxt.x1[q] = (short)mx; nxt.y1[q] = (short)y1; nxt.beg[q] = b2; nxt.cnt[q] = c2; s_newidx[4 * i + 2] = q; q++; }
                    if (c1 > 0) { nxt.x0[q] = (short)mx; nxt.y0[q] = (short)y0; nxt.x1[q] = (short)x1; nxt.y1[q] = (short)my; nxt.beg[q] = b1; nxt.cnt[q] = c1; s_newidx[4 * i + 1] = q; q++; }
                    if (c0 > 0) { nxt.x0[q] = (short)x0; nxt.y0[q] = (short)y0; nxt.x1[q] = (short)mx; nxt.y1[q] = (short)my; nxt.beg[q] = b0; nxt.cnt[q] = c0; s_newidx[4 * i] = q; q++; }
                }
            }
        } else {
            // (2) processing order of the multi-point nodes
            const int m = ot2_scan_array(s_kk, n, s_w); // s_kk[i] = rank among multi nodes (list order)
            if (!sorted_phase) {
                for (int i = tid; i < n; i += OT2_THREADS)
                    if (cur.cnt[i] > 1) s_plist[s_kk[i]] = i;
                __syncthreads();
            } else {
                int P = 1;
                while (P < m) P <<= 1;
                for (int i = tid; i < P; i += OT2_THREADS) s_key[i] = ~0ull;
                __syncthreads();
                for (int i = tid; i < n; i += OT2_THREADS)
                    if (cur.cnt[i] > 1)
                        s_key[s_kk[i]] = ((unsigned long long)(0xffffffffu - (unsigned)cur.cnt[i]) << 32) | (unsigned)i;
                __syncthreads();
                for (int k = 2; k <= P; k <<= 1) {
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int i = tid; i < P; i += OT2_THREADS) {
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
                for (int i = tid; i < m; i += OT2_THREADS) s_plist[i] = (int)(s_key[i] & 0xffffffffu);
                __syncthreads();
            }
            if (iter < 20) OT2_TS(10 + 8 * iter);
            // k = non-empty children per processing rank; exclusive prefix in s_un
            for (int r = tid; r < m; r += OT2_THREADS) {
                const int i = s_plist[r];
                const int k = (s_ccnt[4 * i] > 0) + (s_ccnt[4 * i + 1] > 0) + (s_ccnt[4 * i + 2] > 0) + (s_ccnt[4 * i + 3] > 0);
                s_kk[r] = k;
                s_un[r] = k;
            }
            __syncthreads();
            ot2_scan_array(s_un, m, s_w);
            if (tid == 0) {
                int nproc = m;
                if (sorted_phase) { // first r with n + sum_{r'<=r}(k-1) >= quota (src/ORBextractor.cc:724-725)
                    for (int r = 0; r < m; r++) {
                        const int incl = s_un[r] + s_kk[r];
                        if (n + incl - (r + 1) >= quota) { nproc = r + 1; break; }
                    }
                }
                *K.s_nproc = nproc;
                *K.s_total_k = nproc > 0 ? s_un[nproc - 1] + s_kk[nproc - 1] : 0;
                *K.s_nexpand = 0;
            }
            __syncthreads();
            const int nproc = *K.s_nproc, total_k = *K.s_total_k;
            for (int r = tid; r < nproc; r += OT2_THREADS) s_rank[s_plist[r]] = r;
            __syncthreads();
            for (int i = tid; i < n; i += OT2_THREADS) s_plist[i] = (s_rank[i] < 0) ? 1 : 0; // reuse: unprocessed flags
            __syncthreads();
            const int n_un = ot2_scan_array(s_plist, n, s_w);
            n_new = total_k + n_un;
            if (n_new > MAXN) { // cannot happen for max_nodes >= max(quota+3, 4*n_ini); guard anyway
                if (tid == 0) { *status = 2; *sel_cnt = 0; }
                return;
            }
            if (iter < 20) OT2_TS(11 + 8 * iter);
            // (3) new node array in list order: blocks of later-processed parents nearer the front, children n4..n1
            for (int i = tid; i < n; i += OT2_THREADS) {
                const int r = s_rank[i];
                if (r < 0) {
                    const int q = total_k + s_plist[i];
                    nxt.x0[q] = cur.x0[i]; nxt.y0[q] = cur.y0[i]; nxt.x1[q] = cur.x1[i]; nxt.y1[q] = cur.y1[i];
                    nxt.beg[q] = cur.beg[i]; nxt.cnt[q] = cur.cnt[i];
                    s_newun[i] = q;
                    continue;
                }
                const int x0 = cur.x0[i], y0 = cur.y0[i], x1 = cur.x1[i], y1 = cur.y1[i];
                const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
                const int beg = cur.beg[i];
                const int c0 = s_ccnt[4 * i], c1 = s_ccnt[4 * i + 1], c2 = s_ccnt[4 * i + 2], c3 = s_ccnt[4 * i + 3];
                const int k = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
                int q = total_k - (s_un[r] + k);
                int nexp = 0;
                const int b0 = beg, b1 = beg + c0, b2 = b1 + c1, b3 = b2 + c2;
                if (c3 > 0) { nxt.x0[q] = (short)mx; nxt.y0[q] = (short)my; nxt.x1[q] = (short)x1; nxt.y1[q] = (short)y1; nxt.beg[q] = b3; nxt.cnt[q] = c3; s_newidx[4 * i + 3] = q; q++; nexp += c3 > 1; }
                if (c2 > 0) { nxt.x0[q] = (short)x0; nxt.y0[q] = (short)my; nxt.x1[q] = (short)mx; nxt.y1[q] = (short)y1; nxt.beg[q] = b2; nxt.cnt[q] = c2; s_newidx[4 * i + 2] = q; q++; nexp += c2 > 1; }
                if (c1 > 0) { nxt.x0[q] = (short)mx; nxt.y0[q] = (short)y0; nxt.x1[q] = (short)x1; nxt.y1[q] = (short)my; nxt.beg[q] = b1; nxt.cnt[q] = c1; s_newidx[4 * i + 1] = q; q++; nexp += c1 > 1; }
                if (c0 > 0) { nxt.x0[q] = (short)x0; nxt.y0[q] = (short)y0; nxt.x1[q] = (short)mx; nxt.y1[q] = (short)my; nxt.beg[q] = b0; nxt.cnt[q] = c0; s_newidx[4 * i] = q; q++; nexp += c0 > 1; }
                if (nexp) atomicAdd(K.s_nexpand, nexp);
            }
        }
        // (4) walk 2: scatter every point (children ranges subdivide the parent's range; others stay put)
        // (LDS mode: single buffer; every thread staged its chunk at the top of the pass, before any write)
        __syncthreads();
        if (iter < 20) OT2_TS(12 + 8 * iter);
        {
            int e0 = s_ct[4 * tid], e1 = s_ct[4 * tid + 1], e2 = s_ct[4 * tid + 2], e3 = s_ct[4 * tid + 3]; // E(j)[c], running
            int run_nd = -1, mx = 0, my = 0, multi = 0, rk = -1, cb0 = 0, cb1 = 0, cb2 = 0, cb3 = 0, nun = 0;
            auto step = [&](int j, uint32_t ip, uint32_t xy_staged) {
                const int nd = (int)(ip >> OT2_ID_BITS);
                if (nd != run_nd) {
                    run_nd = nd;
                    multi = cur.cnt[nd] > 1;
                    rk = s_rank[nd];
                    if (multi) {
                        mx = cur.x0[nd] + ((cur.x1[nd] - cur.x0[nd] + 1) >> 1);
                        my = cur.y0[nd] + ((cur.y1[nd] - cur.y0[nd] + 1) >> 1);
                    }
                    if (rk >= 0) { // child range start minus E(node begin)[c]
                        const int t0 = s_nbt[nd];
                        const int beg = cur.beg[nd];
                        const int c0 = s_ccnt[4 * nd], c1 = s_ccnt[4 * nd + 1], c2 = s_ccnt[4 * nd + 2];
                        cb0 = beg - (s_ct[4 * t0] + s_nbl[4 * nd]);
                        cb1 = beg + c0 - (s_ct[4 * t0 + 1] + s_nbl[4 * nd + 1]);
                        cb2 = beg + c0 + c1 - (s_ct[4 * t0 + 2] + s_nbl[4 * nd + 2]);
                        cb3 = beg + c0 + c1 + c2 - (s_ct[4 * t0 + 3] + s_nbl[4 * nd + 3]);
                    } else {
                        nun = s_newun[nd];
                    }
                }
                int pos = j, pn = nun;
                if (multi) {
                    const uint32_t xy = LDSP ? xy_staged : xy_tab[ip & OT2_ID_MASK];
                    const int cls = ((int)(xy & 0xffffu) < mx ? 0 : 1) + ((int)(xy >> 16) < my ? 0 : 2);
                    if (rk >= 0) {
                        pos = cls == 0 ? cb0 + e0 : (cls == 1 ? cb1 + e1 : (cls == 2 ? cb2 + e2 : cb3 + e3));
                        pn = s_newidx[4 * nd + cls];
                    }
                    e0 += cls == 0; e1 += cls == 1; e2 += cls == 2; e3 += cls == 3;
                }
                ip_n[pos] = (ip & OT2_ID_MASK) | ((uint32_t)pn << OT2_ID_BITS);
            };
            if (LDSP) {
#pragma unroll
                for (int k = 0; k < 16; k++)
                    if (jb + k < je) step(jb + k, stage[k], sxy[k]);
            } else {
                for (int j = jb; j < je; j++) step(j, ip_c[j], 0u);
            }
        }
        if (iter < 20) OT2_TS(13 + 8 * iter);
        // (5) stop logic (src/ORBextractor.cc:661-731)
        if (tid == 0) {
            const int prev = n;
            *K.s_n = n_new;
            if (n_new >= quota || n_new == prev) *K.s_done = 1;
            else if (!sorted_phase && n_new + 3 * (nexpand_fast >= 0 ? nexpand_fast : *K.s_nexpand) > quota) *K.s_mode = 1;
            else *K.s_mode = sorted_phase;
        }
        __syncthreads();
        { Ot2Nodes t = cur; cur = nxt; nxt = t; }
        if (!LDSP) { uint32_t *t = ip_c; ip_c = ip_n; ip_n = t; }
        if (*K.s_done) break;
        sorted_phase = *K.s_mode;
        __syncthreads();
    }

    OT2_TS(4);
    // ---- keep the best response per node, first wins (src/ORBextractor.cc:735-754) ----
    const int n = *K.s_n;
    const int n_out = n < L.sel_cap ? n : L.sel_cap;
    if (n > L.sel_cap && tid == 0) *status = 3;
    for (int i = tid; i < n_out; i += OT2_THREADS) {
        const int cnt = cur.cnt[i], beg = cur.beg[i];
        if (cnt > 128) continue; // big nodes: cooperative loop below
        int best = (int)(ip_c[beg] & OT2_ID_MASK), bs = sc_tab[best];
        for (int j = 1; j < cnt; j++) {
            const int id = (int)(ip_c[beg + j] & OT2_ID_MASK);
            const int s = sc_tab[id];
            if (s > bs) { bs = s; best = id; }
        }
        sel_xy[i] = xy_tab[best];
        sel_sc[i] = (uint8_t)bs;
    }
    for (int i = wave; i < n_out; i += OT2_WAVES) {
        const int cnt = cur.cnt[i], beg = cur.beg[i];
        if (cnt <= 128) continue;
        unsigned best = 0xffffffffu; // (255-score)<<24 | position in node
        for (int j = lane; j < cnt; j += 64) {
            const unsigned key = ((unsigned)(255 - sc_tab[ip_c[beg + j] & OT2_ID_MASK]) << 24) | (unsigned)j;
            best = key < best ? key : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned t = (unsigned)__shfl_xor((int)best, o, 64);
            best = t < best ? t : best;
        }
        if (lane == 0) {
            const int id = (int)(ip_c[beg + (best & 0xffffffu)] & OT2_ID_MASK);
            sel_xy[i] = xy_tab[id];
            sel_sc[i] = sc_tab[id];
        }
    }
    if (tid == 0) *sel_cnt = n_out;
    OT2_TS(5);
}

__device__ __forceinline__ void ot2_run_level(const DeviceConfig &cfg, const DeviceBuffers &buf, int level, int img, int sort_cap, int lds_pts, int dbg_stop,
                              uint8_t *s_raw, int *s_w, int *s_scal)
{
    const LevelInfo &L = cfg.lv[level];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int MAXN = cfg.max_nodes;
    long long *ts = (dbg_stop >= 100 && img == 0 && level == dbg_stop - 100) ? buf.dbg_ts : nullptr;
    OT2_TS(0);

    Ot2Ctx K;
    uint8_t *p = s_raw;
    K.s_key = (unsigned long long *)p; p += sizeof(unsigned long long) * sort_cap;
    ot2_bind(K.A, p, MAXN);
    ot2_bind(K.B, p, MAXN);
    K.s_ccnt = (int *)p; p += sizeof(int) * 4 * MAXN;   // child counts per node
    K.s_nbl = (int *)p; p += sizeof(int) * 4 * MAXN;    // class counts inside the owner chunk before the node's first point
    K.s_newidx = (int *)p; p += sizeof(int) * 4 * MAXN; // new node index of child (node, class)
    K.s_nbt = (int *)p; p += sizeof(int) * MAXN;        // chunk (thread) that holds the node's first point
    K.s_newun = (int *)p; p += sizeof(int) * MAXN;      // new node index of an unprocessed node
    K.s_rank = (int *)p; p += sizeof(int) * MAXN;       // processing rank (-1: not processed this pass)
    K.s_plist = (int *)p; p += sizeof(int) * MAXN;      // processing order -> node
    K.s_kk = (int *)p; p += sizeof(int) * MAXN;
    K.s_un = (int *)p; p += sizeof(int) * MAXN;
    K.s_ct = (int *)p; p += sizeof(int) * 4 * OT2_THREADS; // per-chunk class prefix E(chunk start)[c]
    uint32_t *s_xy = (uint32_t *)p; p += sizeof(uint32_t) * lds_pts;
    uint32_t *s_ip = (uint32_t *)p; p += sizeof(uint32_t) * lds_pts;
    uint8_t *s_sc = p;
    K.s_w = s_w;
    K.s_n = &s_scal[0]; K.s_total_k = &s_scal[1]; K.s_nproc = &s_scal[2]; K.s_nexpand = &s_scal[3];
    K.s_mode = &s_scal[4]; K.s_done = &s_scal[5];
    K.sort_cap = sort_cap; K.max_nodes = MAXN;

    const size_t ib = (size_t)img;
    const int *cell_cnt = buf.cell_cnt + ib * cfg.cells_total + L.cell_off;
    int *cell_base = buf.cell_base + ib * cfg.cells_total + L.cell_off;
    const uint32_t *cell_xy = buf.cell_xy + (ib * cfg.cells_total + L.cell_off) * cfg.cell_cap;
    const uint8_t *cell_sc = buf.cell_sc + (ib * cfg.cells_total + L.cell_off) * cfg.cell_cap;
    const size_t coff = ib * cfg.cand_total + L.cand_off;
    uint32_t *xy_a = buf.cand_xy + coff;
    uint8_t *sc_a = buf.cand_sc + coff;
    int *sel_cnt = buf.sel_cnt + ib * cfg.nlevels + level;
    uint32_t *sel_xy = buf.sel_xy + ib * cfg.sel_total + L.sel_off;
    uint8_t *sel_sc = buf.sel_sc + ib * cfg.sel_total + L.sel_off;

    // ---- gather the per-cell candidates into emission order (cell-row-major, then in-cell order) ----
    int nc;
    {
        const int n = L.n_cells;
        const int per = (n + OT2_THREADS - 1) / OT2_THREADS;
        const int b = tid * per, e = (b + per < n) ? b + per : n;
        int sum = 0;
        for (int i = b; i < e; i++) sum += cell_cnt[i];
        const int inc = ot2_wave_incl_scan(sum, lane);
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int base = 0, total = 0;
        for (int w = 0; w < OT2_WAVES; w++) {
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
        nc = total;
        __syncthreads();
    }
    OT2_TS(1);
    if (nc > L.cand_cap) { nc = L.cand_cap; if (tid == 0) buf.status[img] = 1; }
    if (nc > (1 << OT2_ID_BITS)) { nc = 1 << OT2_ID_BITS; if (tid == 0) buf.status[img] = 4; } // host routes such levels to the generic kernel
    if (tid == 0) buf.lvl_ncand[ib * cfg.nlevels + level] = nc;
    const bool ldsp = nc <= lds_pts && nc <= 16 * OT2_THREADS;
    // position-parallel copy: the cell of position i is found by a binary search over the scanned cell
    // offsets (staged in LDS when they fit), so the reads of one cell's slot are contiguous
    {
        int *s_cb = K.s_ct; // 4*OT2_THREADS ints, free until the first pass
        const bool cb_lds = L.n_cells <= 4 * OT2_THREADS;
        if (cb_lds)
            for (int c = tid; c < L.n_cells; c += OT2_THREADS) s_cb[c] = cell_base[c];
        __syncthreads();
        const int *cb = cb_lds ? s_cb : cell_base;
#pragma unroll 2
        for (int i = tid; i < nc; i += OT2_THREADS) {
            int lo = 0, hi = L.n_cells - 1; // last cell with cell_base <= i
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (cb[mid] <= i) lo = mid; else hi = mid - 1;
            }
            const int k = i - cb[lo];
            const uint32_t xy = cell_xy[(size_t)lo * cfg.cell_cap + k];
            const uint8_t sc = cell_sc[(size_t)lo * cfg.cell_cap + k];
            xy_a[i] = xy; sc_a[i] = sc; // emission-order copy (orbfe_fetch_candidates, HBM path)
            if (ldsp) { s_xy[i] = xy; s_sc[i] = sc; }
        }
    }
    __syncthreads();
    OT2_TS(2);
    if (nc == 0 || dbg_stop == 1) {
        if (tid == 0) *sel_cnt = 0;
        return;
    }
    const int region_h = (L.h - cfg.edge_threshold + 3) - cfg.min_border;
    if (ldsp)
        ot2_body<true>(ts, K, L, region_h, nc, (dbg_stop >= 2 && dbg_stop < 100) ? dbg_stop - 1 : L.quota, s_xy, s_sc, s_ip, s_ip, sel_xy, sel_sc, sel_cnt, buf.status + img);
    else
        ot2_body<false>(ts, K, L, region_h, nc, L.quota, xy_a, sc_a, buf.idx0 + coff, buf.ot_xy2 + coff, sel_xy, sel_sc, sel_cnt,
                        buf.status + img);
}

// A workgroup handles levels p and nlevels-1-p of one image back to back: candidates shrink ~1.44x per
// level, so the pairs (0,7), (1,6), ... are balanced, and a batch of 32 stereo pairs is exactly one
// workgroup per CU (the LDS footprint allows only one) instead of two unbalanced rounds.
__global__ __launch_bounds__(OT2_THREADS) void octree2_kernel(DeviceConfig cfg, DeviceBuffers buf, int sort_cap, int lds_pts ORBFE_CUT_PARAM)
{
#ifdef ORBFE_PROFILE_CUTS
    const int dbg_stop = dbg;
#else
    const int dbg_stop = 0; // the stop / timestamp branches fold away in the shipped build
#endif
    extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
    __shared__ int s_w[4 * OT2_WAVES];
    __shared__ int s_scal[8];
    const int la = blockIdx.x, lb = cfg.nlevels - 1 - (int)blockIdx.x, img = blockIdx.y;
    ot2_run_level(cfg, buf, la, img, sort_cap, lds_pts, dbg_stop, s_raw, s_w, s_scal);
    if (lb != la) {
        __syncthreads();
        ot2_run_level(cfg, buf, lb, img, sort_cap, lds_pts, dbg_stop, s_raw, s_w, s_scal);
    }
}

void orbfe_launch_octree2(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, int sort_cap, int lds_pts, size_t lds, hipStream_t s)
{
    dim3 grid((cfg.nlevels + 1) / 2, n_images);
    hipLaunchKernelGGL(octree2_kernel, grid, dim3(OT2_THREADS), lds, s, cfg, buf, sort_cap, lds_pts ORBFE_CUT_ARG("ORBFE_OT2_STOP"));
}

int orbfe_octree2_prepare(size_t lds)
{
    if (lds <= 64 * 1024) return 0;
    return hipFuncSetAttribute((const void *)octree2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 0 : -1;
}
