// multi_device_selftest.cpp -- MultiDeviceFrontEnd (multi_device.h) against a single context, in one C++ process.
//   usage: multi_device_selftest <pairs_u8: [P][2][h][w]> <P> <w> <h> <nfeatures> <fx> <bf> <n_contexts> <out_prefix>
// Runs the same P pairs (a) through one context as one batch and (b) through n_contexts contexts with one feeder thread each
// (context i on device i % device_count: on a one-GPU box all on device 0), twenty steps, and requires identical results in
// frame order every time; writes pair 0 and pair P - 1 of the multi-context run for the pytest's oracle comparison.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "multi_device.h"

using namespace ORB_SLAM2;

template <typename T> static void dump(const std::string &path, const std::vector<T> &v)
{
    std::ofstream f(path, std::ios::binary);
    f.write((const char *)v.data(), (std::streamsize)(v.size() * sizeof(T)));
}
static bool same(const StereoPairResult &a, const StereoPairResult &b)
{
    return a.mvKeys.size() == b.mvKeys.size() && a.mvKeysRight.size() == b.mvKeysRight.size() &&
           !std::memcmp(a.mvKeys.data(), b.mvKeys.data(), a.mvKeys.size() * sizeof(orbfe_keypoint)) &&
           !std::memcmp(a.mvKeysRight.data(), b.mvKeysRight.data(), a.mvKeysRight.size() * sizeof(orbfe_keypoint)) && a.mDescriptors == b.mDescriptors &&
           a.mDescriptorsRight == b.mDescriptorsRight && a.mvuRight == b.mvuRight && a.mvDepth == b.mvDepth;
}

int main(int argc, char **argv)
{
    if (argc != 10) { std::cerr << "usage: multi_device_selftest pairs P w h nfeatures fx bf n_contexts out_prefix\n"; return 2; }
    const int P = std::atoi(argv[2]), w = std::atoi(argv[3]), h = std::atoi(argv[4]), nf = std::atoi(argv[5]), nctx = std::atoi(argv[8]);
    const float fx = (float)std::atof(argv[6]), bf = (float)std::atof(argv[7]);
    const std::string prefix = argv[9];
    std::vector<uint8_t> pairs((size_t)P * 2 * w * h);
    {
        std::ifstream f(argv[1], std::ios::binary);
        if (!f.read((char *)pairs.data(), (std::streamsize)pairs.size())) { std::cerr << "cannot read " << argv[1] << "\n"; return 2; }
    }
    try {
        orbfe_params p;
        std::memset(&p, 0, sizeof(p));
        p.nfeatures = nf; p.scale_factor = 1.2f; p.nlevels = 8; p.ini_th_fast = 20; p.min_th_fast = 7; p.patch_size = 31; p.half_patch_size = 15; p.edge_threshold = 19;
        p.fx = fx; p.fy = fx; p.cx = w * 0.5f; p.cy = h * 0.5f; p.bf = bf; p.width = w; p.height = h;
        std::vector<StereoPairResult> ref, got;
        {
            MultiDeviceFrontEnd one(p, 1, P);
            one.Process(pairs.data(), P, ref);
        }
        MultiDeviceFrontEnd many(p, nctx, P);
        int bad = 0;
        double best_ms = 1e30;
        for (int step = 0; step < 20; step++) {
            const auto t0 = std::chrono::steady_clock::now();
            many.Process(pairs.data(), P, got);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms < best_ms) best_ms = ms;
            for (int g = 0; g < P; g++) bad += !same(got[g], ref[g]);
        }
        std::printf("contexts %d (devices:", many.Contexts());
        for (int i = 0; i < many.Contexts(); i++) std::printf(" %d", many.DeviceOf(i));
        std::printf("), %d pairs per step, best step %.3f ms = %.0f pairs/s host-fed from pageable memory, mismatches %d\n", P, best_ms, P / best_ms * 1e3, bad);
        // a smaller batch than contexts, and an odd one: sharding must still cover every pair once
        std::vector<StereoPairResult> few;
        many.Process(pairs.data(), nctx > 1 ? nctx - 1 : 1, few);
        for (size_t g = 0; g < few.size(); g++) bad += !same(few[g], ref[g]);
        for (int g : {0, P - 1}) {
            const std::string t = prefix + "_p" + std::to_string(g);
            dump(t + ".kl", got[g].mvKeys); dump(t + ".kr", got[g].mvKeysRight); dump(t + ".dl", got[g].mDescriptors); dump(t + ".dr", got[g].mDescriptorsRight);
            dump(t + ".ur", got[g].mvuRight); dump(t + ".dp", got[g].mvDepth);
        }
        if (bad) { std::cerr << bad << " pair results differ from the single-context batch\n"; return 3; }
        std::printf("multi-device selftest ok\n");
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
