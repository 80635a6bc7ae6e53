// tests/asan/resolve_harness.cpp -- TEST-ONLY: runs the HOST half of liborbfe.so's Tracking matchers
// (orbslam2_amd/csrc/orbfe_match_resolve.h: query builders, Frame::isInFrustum, greedy replays, full-list fallback) on the CPU
// under -fsanitize=address,undefined, with the device's part -- grid + window query + Hamming + top-K selection -- replaced by
// a brute-force stand-in written here.  tests/test_host_resolve_asan.py compares the results with the C oracle, so the product's
// host logic is checked without a GPU, and every index it forms is checked by the sanitizers.
//   usage: resolve_harness <scene_dir>      (scene files of tests/scene_files.py)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../orbslam2_amd/csrc/orbfe_match_resolve.h"

using namespace orbfe_resolve;

static std::string g_dir;
template <class T> static std::vector<T> rd(const char *name)
{
    std::ifstream f(g_dir + "/" + name, std::ios::binary | std::ios::ate);
    if (!f) { std::cerr << "missing " << name << "\n"; std::exit(2); }
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<T> v((size_t)n / sizeof(T));
    f.read((char *)v.data(), n);
    return v;
}
template <class T> static void wr(const char *name, const std::vector<T> &v)
{
    std::ofstream f(g_dir + "/" + name, std::ios::binary);
    f.write((const char *)v.data(), (std::streamsize)(v.size() * sizeof(T)));
}

// ---- stand-in for the device: Frame grid (AssignFeaturesToGrid) + window query + Hamming, as window_candidates_kernel defines
// its output (one key per hit; a hit that fails the mvuRight gate keeps its place with dist = 511) ----
struct BruteFrame {
    std::vector<orbfe_keypoint> keys;
    std::vector<uint8_t> desc;
    std::vector<float> ur; // empty: monocular
    float min_x, min_y, inv_w, inv_h;
    std::vector<std::vector<int> > cell; // [ix * 48 + iy]
    void build(float minx, float maxx, float miny, float maxy)
    {
        min_x = minx; min_y = miny;
        inv_w = 64.0f / (maxx - minx); inv_h = 48.0f / (maxy - miny);
        cell.assign(64 * 48, std::vector<int>());
        for (size_t i = 0; i < keys.size(); i++) {
            const int px = (int)roundf((keys[i].x - min_x) * inv_w), py = (int)roundf((keys[i].y - min_y) * inv_h);
            if (px >= 0 && px < 64 && py >= 0 && py < 48) cell[px * 48 + py].push_back((int)i);
        }
    }
    void query(const MatchQuery &Q, const uint8_t *qd, std::vector<ckey_t> &out) const
    {
        out.clear();
        if (!(Q.flags & 1)) return;
        int v = (int)floorf((Q.u - min_x - Q.r) * inv_w);
        const int min_cx = v > 0 ? v : 0;
        v = (int)ceilf((Q.u - min_x + Q.r) * inv_w);
        const int max_cx = v < 63 ? v : 63;
        v = (int)floorf((Q.v - min_y - Q.r) * inv_h);
        const int min_cy = v > 0 ? v : 0;
        v = (int)ceilf((Q.v - min_y + Q.r) * inv_h);
        const int max_cy = v < 47 ? v : 47;
        if (!(min_cx < 64 && max_cx >= 0 && min_cy < 48 && max_cy >= 0)) return;
        const bool check_levels = (Q.min_level > 0) || (Q.max_level >= 0);
        for (int ix = min_cx; ix <= max_cx; ix++)
            for (int iy = min_cy; iy <= max_cy; iy++)
                for (int idx : cell[ix * 48 + iy]) {
                    const orbfe_keypoint &kp = keys[idx];
                    if (check_levels && (kp.octave < Q.min_level || (Q.max_level >= 0 && kp.octave > Q.max_level))) continue;
                    if (!(fabsf(kp.x - Q.u) < Q.r && fabsf(kp.y - Q.v) < Q.r)) continue;
                    unsigned dist = 0;
                    for (int b = 0; b < 32; b++) dist += (unsigned)__builtin_popcount((unsigned)(qd[b] ^ desc[(size_t)32 * idx + b]));
                    if ((Q.flags & 2) && !ur.empty() && ur[idx] > 0 && fabsf(Q.ur - ur[idx]) > Q.ur_rad) dist = 511;
                    out.push_back(((ckey_t)dist << 36) | ((ckey_t)ix << 30) | ((ckey_t)iy << 24) | ((ckey_t)idx << 8) | (ckey_t)(kp.octave & 255));
                }
    }
};

// what the device hands over for one matcher call: full lists + top-K prefixes under the call's static filters
struct DeviceStandIn {
    std::vector<std::vector<ckey_t> > lists;
    std::vector<ckey_t> topk;
    std::vector<int32_t> n_static;
    int fallbacks = 0;
    void run(const BruteFrame &F, const std::vector<MatchQuery> &q, const std::vector<uint8_t> &qd, const uint8_t *blocked0, bool drop_gated, int starve_every)
    {
        const int nq = (int)q.size();
        lists.assign(nq, std::vector<ckey_t>());
        topk.assign((size_t)nq * TOPK, NO_KEY);
        n_static.assign(nq, 0);
        for (int i = 0; i < nq; i++) {
            F.query(q[i], &qd[(size_t)32 * i], lists[i]);
            std::vector<ckey_t> ok;
            for (ckey_t k : lists[i]) {
                if (drop_gated && key_dist(k) >= 256) continue;
                if (blocked0 && blocked0[key_idx(k)]) continue;
                ok.push_back(k);
            }
            std::sort(ok.begin(), ok.end());
            if (starve_every > 0 && i % starve_every == 0 && !ok.empty()) { n_static[i] = INT_MAX; continue; } // "list too long for the LDS stage"
            n_static[i] = (int32_t)ok.size();
            for (size_t k = 0; k < ok.size() && k < (size_t)TOPK; k++) topk[(size_t)i * TOPK + k] = ok[k];
        }
    }
    static int full_cb(void *user, int q, std::vector<ckey_t> &out)
    {
        DeviceStandIn *d = (DeviceStandIn *)user;
        d->fallbacks++;
        out = d->lists[q]; // unfiltered, unsorted: the device's list
        return 0;
    }
    CandidateSource source(const uint8_t *blocked0, bool drop_gated)
    {
        CandidateSource s;
        s.topk = topk.data(); s.n_static = n_static.data(); s.blocked0 = blocked0; s.drop_gated = drop_gated; s.user = this; s.full = full_cb;
        return s;
    }
};

int main(int argc, char **argv)
{
    if (argc != 2) { std::cerr << "usage: resolve_harness <scene_dir>\n"; return 2; }
    g_dir = argv[1];
    const std::vector<float> cam = rd<float>("cam.f32"), bounds = rd<float>("bounds.f32");
    orbfe_params P = {};
    P.fx = cam[0]; P.fy = cam[1]; P.cx = cam[2]; P.cy = cam[3]; P.bf = cam[4]; P.nlevels = 8; P.scale_factor = 1.2f;
    const Camera C = camera_of(&P);
    float sf[8];
    sf[0] = 1.0f;
    for (int i = 1; i < 8; i++) sf[i] = (float)((double)sf[i - 1] * (double)1.2f);
    const float log_sf = logf((float)(double)P.scale_factor);

    BruteFrame cur;
    cur.keys = rd<orbfe_keypoint>("cur_k.bin"); cur.desc = rd<uint8_t>("cur_d.bin"); cur.ur = rd<float>("cur_ur.bin");
    cur.build(bounds[0], bounds[1], bounds[2], bounds[3]);
    const int N = (int)cur.keys.size();
    const std::vector<uint8_t> cur_has = rd<uint8_t>("cur_has_obs.bin"), last_desc = rd<uint8_t>("last_desc.bin"), found = rd<uint8_t>("already_found.bin");
    const std::vector<float> pos = rd<float>("last_pos.bin"), ang = rd<float>("last_ang.bin"), T_last = rd<float>("T_last.bin"), T_cur = rd<float>("T_cur.bin");
    const std::vector<float> normal = rd<float>("normal.bin"), max_d = rd<float>("max_d.bin"), min_d = rd<float>("min_d.bin");
    const std::vector<int32_t> oct = rd<int32_t>("last_oct.bin"), valid3 = rd<int32_t>("last_valid.bin"), obs = rd<int32_t>("last_obs.bin");
    const int M = (int)oct.size();
    std::vector<int32_t> usable(M);
    for (int i = 0; i < M; i++) usable[i] = valid3[i] == 1;
    int total_fallbacks = 0;

    for (int starve = 0; starve <= 3; starve += 3) { // second round: every third query is starved of its prefix (forced full-list replay)
        const std::string tag = starve ? "_starved" : "";
        DeviceStandIn dev;
        std::vector<MatchQuery> q;
        std::vector<uint8_t> qd;
        // 1. SearchByProjection(CurrentFrame, LastFrame, 7, false), orientation check on
        {
            if (build_queries_last(C, sf, 8, bounds[0], bounds[1], bounds[2], bounds[3], T_cur.data(), T_last.data(), M, pos.data(), last_desc.data(),
                                   usable.data(), oct.data(), 7.0f, 0, q, qd) != 0) return 3;
            std::vector<uint8_t> has(cur_has), blocked0(cur_has);
            dev.run(cur, q, qd, blocked0.data(), true, starve);
            CandidateSource src = dev.source(blocked0.data(), true);
            std::vector<int32_t> match(N);
            const int nm = resolve_last(src, M, obs.data(), ang.data(), N, &cur.keys[0].angle, sizeof(orbfe_keypoint), has, 1, match.data());
            wr(("h_last" + tag + ".bin").c_str(), match); wr(("hn_last" + tag + ".bin").c_str(), std::vector<int32_t>(1, nm));
        }
        // 2. isInFrustum + SearchByProjection(F, vpMapPoints, 3), nnratio 0.8
        {
            std::vector<orbfe_track_point> tp(M);
            is_in_frustum(C, 8, log_sf, T_cur.data(), bounds[0], bounds[1], bounds[2], bounds[3], M, pos.data(), normal.data(), max_d.data(), min_d.data(), 0.5f, tp.data());
            if (!starve) wr("h_tp.bin", tp);
            if (build_queries_points(sf, 8, M, tp.data(), last_desc.data(), 3.0f, q, qd) != 0) return 4;
            std::vector<uint8_t> has(cur_has), blocked0(cur_has);
            dev.run(cur, q, qd, blocked0.data(), true, starve);
            CandidateSource src = dev.source(blocked0.data(), true);
            std::vector<int32_t> match(N);
            const int nm = resolve_points(src, M, obs.data(), N, has, 0.8f, match.data());
            wr(("h_pts" + tag + ".bin").c_str(), match); wr(("hn_pts" + tag + ".bin").c_str(), std::vector<int32_t>(1, nm));
        }
        // 3. SearchByProjection(CurrentFrame, pKF, sAlreadyFound, 10, 100)
        {
            std::vector<int32_t> kf_ok(M);
            for (int i = 0; i < M; i++) kf_ok[i] = usable[i] && !found[i];
            BruteFrame mono = cur;
            mono.ur.clear();
            build_queries_kf(C, sf, 8, log_sf, bounds[0], bounds[1], bounds[2], bounds[3], T_cur.data(), M, pos.data(), last_desc.data(), kf_ok.data(), max_d.data(),
                             min_d.data(), 10.0f, q, qd);
            std::vector<uint8_t> has(cur_has), blocked0(cur_has);
            dev.run(mono, q, qd, blocked0.data(), false, starve);
            CandidateSource src = dev.source(blocked0.data(), false);
            std::vector<int32_t> match(N);
            const int nm = resolve_kf(src, M, ang.data(), N, &cur.keys[0].angle, sizeof(orbfe_keypoint), has, 100, 1, match.data());
            wr(("h_kf" + tag + ".bin").c_str(), match); wr(("hn_kf" + tag + ".bin").c_str(), std::vector<int32_t>(1, nm));
        }
        // 4. SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, 100), nnratio 0.9
        {
            const std::vector<orbfe_keypoint> k1 = rd<orbfe_keypoint>("init_k1.bin");
            const std::vector<uint8_t> d1 = rd<uint8_t>("init_d1.bin");
            BruteFrame F2;
            F2.keys = rd<orbfe_keypoint>("init_k2.bin"); F2.desc = rd<uint8_t>("init_d2.bin");
            F2.build(bounds[0], bounds[1], bounds[2], bounds[3]);
            const int n1 = (int)k1.size(), n2 = (int)F2.keys.size();
            std::vector<float> prev(2 * (size_t)n1);
            for (int i = 0; i < n1; i++) { prev[2 * i] = k1[i].x; prev[2 * i + 1] = k1[i].y; }
            build_queries_initialization(n1, k1.data(), d1.data(), prev.data(), 100, q, qd);
            dev.run(F2, q, qd, nullptr, false, starve);
            CandidateSource src = dev.source(nullptr, false);
            std::vector<int32_t> m12(n1);
            const int nm = resolve_initialization(src, n1, n2, &k1[0].angle, sizeof(orbfe_keypoint), &F2.keys[0].angle, sizeof(orbfe_keypoint), &F2.keys[0].x, 0.9f, 1,
                                                  prev.data(), m12.data());
            wr(("h_init" + tag + ".bin").c_str(), m12); wr(("h_prev" + tag + ".bin").c_str(), prev); wr(("hn_init" + tag + ".bin").c_str(), std::vector<int32_t>(1, nm));
        }
        total_fallbacks += dev.fallbacks;
    }
    // ComputeThreeMaxima on a few histograms, including empty and tied ones
    {
        const int32_t h0[30] = {0}, h1[30] = {5, 5, 5}, h2[30] = {0, 0, 0, 50, 0, 0, 0, 20, 0, 6};
        int a, b, c;
        std::vector<int32_t> outv;
        three_maxima(h0, 30, &a, &b, &c); outv.push_back(a); outv.push_back(b); outv.push_back(c);
        three_maxima(h1, 30, &a, &b, &c); outv.push_back(a); outv.push_back(b); outv.push_back(c);
        three_maxima(h2, 30, &a, &b, &c); outv.push_back(a); outv.push_back(b); outv.push_back(c);
        wr("h_three.bin", outv);
    }
    std::printf("resolve harness ok, full-list replays: %d\n", total_fallbacks);
    return 0;
}
