// KeyFrameDatabase.h -- host-side mirror of ORB_SLAM2::KeyFrameDatabase (reference include/KeyFrameDatabase.h,
// src/KeyFrameDatabase.cc) over the C ABI (include/orbfe.h): the keyframes' BoW vectors live in HBM, add / erase / clear
// keep the reference's names, and DetectRelocalizationCandidates returns database indices instead of KeyFrame*.
// The caller keeps an index -> KeyFrame* table (add() returns the index) and the per-keyframe mRelocScore state.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/orbfe.h"

namespace ORB_SLAM2
{

class KeyFrameDatabase
{
public:
    explicit KeyFrameDatabase(orbfe_context *ctx) : mCtx(ctx) { Check(orbfe_kfdb_clear(ctx)); }

    // KeyFrameDatabase::add (src/KeyFrameDatabase.cc:38-44): the keyframe's fBow as ascending word ids + weights
    int add(const std::vector<uint32_t> &words, const std::vector<float> &weights)
    {
        int idx = -1;
        Check(orbfe_kfdb_add(mCtx, words.data(), weights.data(), (int)words.size(), &idx));
        mRelocScore.push_back(0.f); // the reference leaves KeyFrame::mRelocScore uninitialised; see DESIGN.md (Q10)
        return idx;
    }
    void erase(int kf) { Check(orbfe_kfdb_erase(mCtx, kf)); }              // :46-62
    void clear() { Check(orbfe_kfdb_clear(mCtx)); mRelocScore.clear(); }   // :64-70

    // DetectRelocalizationCandidates(Frame *F) (:196-307): F's fBow + every keyframe's GetBestCovisibilityKeyFrames(10)
    // list (CSR over database indices) -> candidate indices in the reference's order
    std::vector<int32_t> DetectRelocalizationCandidates(const std::vector<uint32_t> &words, const std::vector<float> &weights,
                                                        const std::vector<int32_t> &covisOff, const std::vector<int32_t> &covisIdx)
    {
        std::vector<int32_t> cand(mRelocScore.size() ? mRelocScore.size() : 1);
        int n = 0;
        Check(orbfe_detect_reloc_candidates(mCtx, words.data(), weights.data(), (int)words.size(), covisOff.data(), covisIdx.data(),
                                            mRelocScore.data(), cand.data(), (int)cand.size(), &n));
        cand.resize(n);
        return cand;
    }

    // DetectLoopCandidates(KeyFrame *pKF, float minScore) (:73-194): pKF's fBow, a flag per database keyframe for
    // pKF->GetConnectedKeyFrames(), and the covisibility lists -> candidate indices in the reference's order
    std::vector<int32_t> DetectLoopCandidates(const std::vector<uint32_t> &words, const std::vector<float> &weights,
                                              const std::vector<uint8_t> &connected, float minScore,
                                              const std::vector<int32_t> &covisOff, const std::vector<int32_t> &covisIdx)
    {
        std::vector<int32_t> cand(mRelocScore.size() ? mRelocScore.size() : 1);
        int n = 0;
        Check(orbfe_detect_loop_candidates(mCtx, words.data(), weights.data(), (int)words.size(), connected.empty() ? nullptr : connected.data(),
                                           minScore, covisOff.data(), covisIdx.data(), cand.data(), (int)cand.size(), &n));
        cand.resize(n);
        return cand;
    }

protected:
    void Check(int rc)
    {
        if (rc != ORBFE_OK) throw std::runtime_error(std::string("orbfe: ") + orbfe_last_error(mCtx));
    }
    orbfe_context *mCtx;
    std::vector<float> mRelocScore; // KeyFrame::mRelocScore of every keyframe (persistent across queries)
};

} // namespace ORB_SLAM2
