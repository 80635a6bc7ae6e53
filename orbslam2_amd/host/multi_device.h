// multi_device.h -- single-process, multi-device host of the front-end (round-2 verdict item 8).
//
// The reference is ONE process whose threads share everything (src/System.cc:98-113); `north_star` keeps the host code C++.  This
// is the C++ shape of BASELINE.json's config 4 ("64 frame pairs in flight, sharded across 8 GPUs"): N device contexts (context i on
// device i % orbfe_device_count(), so 8 contexts on an 8-GPU node, or several per GPU to keep more than one step chain in flight),
// one feeder thread each, frame pairs dealt round-robin exactly like orbslam2_amd/dist.py: shard_pairs (pair g -> context g % N),
// no exchange between contexts (pairs are independent: src/Frame.cc:61-117 touches no shared state), results gathered in frame
// order.  bench.py's one-process-per-GPU torchrun path stays the measuring harness the driver launches; this header is what a
// C++ application embeds.  Only the C ABI is used (no HIP headers needed to compile this).
#pragma once

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbfe.h"

namespace ORB_SLAM2
{

// what Frame::Frame(stereo) leaves behind for one pair (src/Frame.cc:61-117)
struct StereoPairResult {
    std::vector<orbfe_keypoint> mvKeys, mvKeysRight;
    std::vector<uint8_t> mDescriptors, mDescriptorsRight; // N x 32
    std::vector<float> mvuRight, mvDepth;
};

// pair g of a batch goes to context g % n_contexts (the same rule as orbslam2_amd/dist.py: shard_pairs)
inline std::vector<int> ShardPairs(int n_pairs, int rank, int world)
{
    std::vector<int> mine;
    for (int g = rank; g < n_pairs; g += world) mine.push_back(g);
    return mine;
}

class MultiDeviceFrontEnd
{
public:
    // `proto`: the extractor / camera / image-size parameters (device and max_images are set here); `n_contexts` contexts, each
    // sized for ceil(max_pairs / n_contexts) pairs per step
    MultiDeviceFrontEnd(const orbfe_params &proto, int n_contexts, int max_pairs) : mMaxPairs(max_pairs)
    {
        const int ndev = orbfe_device_count();
        if (ndev < 1) throw std::runtime_error("MultiDeviceFrontEnd: no HIP device (the library has no CPU path)");
        if (n_contexts < 1 || max_pairs < 1) throw std::invalid_argument("MultiDeviceFrontEnd: need at least one context and one pair");
        mImageBytes = (size_t)proto.width * proto.height;
        const int share = (max_pairs + n_contexts - 1) / n_contexts;
        mLanes.resize(n_contexts);
        try {
        for (int i = 0; i < n_contexts; i++) {
            Lane &L = mLanes[i];
            orbfe_params p = proto;
            p.device = i % ndev;
            p.max_images = 2 * share;
            if (orbfe_create(&p, &L.ctx) != ORBFE_OK) {
                const std::string msg = orbfe_last_error(nullptr);
                throw std::runtime_error("orbfe_create (context " + std::to_string(i) + ", device " + std::to_string(p.device) + "): " + msg);
            }
            L.device = p.device;
            L.cap = orbfe_keypoint_capacity(L.ctx);
            L.images.resize((size_t)2 * share * mImageBytes);
            orbfe_packed_layout lay; // the largest block this context returns (round 4: results come down packed, one copy per step)
            if (orbfe_get_packed_layout(L.ctx, 2 * share, ORBFE_PACK_STEREO, &lay) != ORBFE_OK) throw std::runtime_error("orbfe_get_packed_layout failed");
            L.block.resize(lay.bytes);
            L.thread = std::thread(&MultiDeviceFrontEnd::Feed, this, i);
        }
        } catch (...) { // a failed allocation or thread start after some feeders run: stop and join them (a joinable std::thread must not be destroyed)
            Shutdown();
            throw;
        }
    }
    ~MultiDeviceFrontEnd() { Shutdown(); }
    MultiDeviceFrontEnd(const MultiDeviceFrontEnd &) = delete;
    MultiDeviceFrontEnd &operator=(const MultiDeviceFrontEnd &) = delete;

    int Contexts() const { return (int)mLanes.size(); }
    int DeviceOf(int context) const { return mLanes[context].device; }

    // One step: n_pairs pairs ([pair][left | right][h][w] packed 8UC1 in host memory) -> results in frame order.  Every context
    // extracts its shard (upload, one stage chain, download) on its own feeder thread; the call returns when all are done.
    // One Process() at a time per object (the caller is the application's frame loop); `pairs` and `out` must outlive the call.
    void Process(const uint8_t *pairs, int n_pairs, std::vector<StereoPairResult> &out)
    {
        if (n_pairs < 1 || n_pairs > mMaxPairs) throw std::invalid_argument("MultiDeviceFrontEnd::Process: batch larger than the capacity fixed at construction");
        out.resize(n_pairs);
        const int N = (int)mLanes.size();
        {
            std::lock_guard<std::mutex> lk(mMu);
            mPairs = pairs; mPairsN = n_pairs; mOut = &out; mPending = 0;
            for (int i = 0; i < N; i++)
                if (i < n_pairs) { mLanes[i].go = true; mPending++; }
        }
        mCv.notify_all();
        std::unique_lock<std::mutex> lk(mMu);
        mDone.wait(lk, [&] { return mPending == 0; });
        for (int i = 0; i < N; i++)
            if (!mLanes[i].error.empty()) { const std::string e = mLanes[i].error; mLanes[i].error.clear(); throw std::runtime_error("context " + std::to_string(i) + ": " + e); }
    }

private:
    struct Lane {
        orbfe_context *ctx = nullptr;
        int device = 0, cap = 0;
        std::thread thread;
        bool go = false;
        std::string error;
        std::vector<uint8_t> images, block; // this context's pairs, packed L0 R0 L1 R1 ...; its packed result block (include/orbfe.h)
    };

    void Feed(int i)
    {
        Lane &L = mLanes[i];
        for (;;) {
            const uint8_t *pairs; int n; std::vector<StereoPairResult> *out;
            {
                std::unique_lock<std::mutex> lk(mMu);
                mCv.wait(lk, [&] { return L.go || mStop; });
                if (mStop) return;
                L.go = false;
                pairs = mPairs; n = mPairsN; out = mOut;
            }
            const std::vector<int> mine = ShardPairs(n, i, (int)mLanes.size());
            try {
                for (size_t j = 0; j < mine.size(); j++) // this context's pairs, packed L0 R0 L1 R1 ...
                    std::memcpy(&L.images[2 * j * mImageBytes], pairs + (size_t)2 * mine[j] * mImageBytes, 2 * mImageBytes);
                const int n_mine = (int)mine.size();
                orbfe_packed_layout lay;
                if (orbfe_get_packed_layout(L.ctx, 2 * n_mine, ORBFE_PACK_STEREO, &lay) != ORBFE_OK ||
                    orbfe_stereo_batch_packed(L.ctx, L.images.data(), n_mine, 0, L.block.data(), L.block.size()) != ORBFE_OK)
                    throw std::runtime_error(orbfe_last_error(L.ctx));
                const uint8_t *blk = L.block.data();
                const size_t cap = (size_t)lay.capacity;
                // expand on the host with the reference's own operations (pt = level coordinates * mvScaleFactor[octave],
                // src/ORBextractor.cc:909-915; size = scaledPatchSize, :838), gather in frame order
                auto keys = [&](int o, std::vector<orbfe_keypoint> &dst) {
                    const int n = ((const int32_t *)(blk + lay.counts_off))[o];
                    dst.resize((size_t)n);
                    int got = 0;
                    if (orbfe_expand_packed(L.ctx, blk, &lay, o, dst.data(), n, &got) != ORBFE_OK || got != n) throw std::runtime_error("orbfe_expand_packed failed");
                    return n;
                };
                for (int j = 0; j < n_mine; j++) {
                    StereoPairResult &r = (*out)[mine[j]];
                    const int l = 2 * j, rr = 2 * j + 1;
                    const int nl = keys(l, r.mvKeys), nr = keys(rr, r.mvKeysRight);
                    const uint8_t *dl = blk + lay.desc_off + (size_t)l * cap * 32, *dr = blk + lay.desc_off + (size_t)rr * cap * 32;
                    r.mDescriptors.assign(dl, dl + (size_t)nl * 32);
                    r.mDescriptorsRight.assign(dr, dr + (size_t)nr * 32);
                    const float *ur = (const float *)(blk + lay.u_right_off) + (size_t)j * cap, *dp = (const float *)(blk + lay.depth_off) + (size_t)j * cap;
                    r.mvuRight.assign(ur, ur + nl);
                    r.mvDepth.assign(dp, dp + nl);
                }
            } catch (const std::exception &e) {
                L.error = e.what();
            }
            {
                std::lock_guard<std::mutex> lk(mMu);
                mPending--;
            }
            mDone.notify_all();
        }
    }

    void Shutdown()
    {
        {
            std::lock_guard<std::mutex> lk(mMu);
            mStop = true;
        }
        mCv.notify_all();
        for (Lane &L : mLanes) {
            if (L.thread.joinable()) L.thread.join();
            if (L.ctx) { orbfe_destroy(L.ctx); L.ctx = nullptr; }
        }
    }

    std::vector<Lane> mLanes;
    int mMaxPairs;
    size_t mImageBytes = 0;
    std::mutex mMu;
    std::condition_variable mCv, mDone;
    bool mStop = false;
    const uint8_t *mPairs = nullptr;
    int mPairsN = 0, mPending = 0;
    std::vector<StereoPairResult> *mOut = nullptr;
};

} // namespace ORB_SLAM2
