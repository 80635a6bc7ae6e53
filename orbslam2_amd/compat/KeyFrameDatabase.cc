// compat/KeyFrameDatabase.cc -- see KeyFrameDatabase.h.  Every method flattens (fBow -> ascending word ids + weights, KeyFrame* ->
// database index, GetBestCovisibilityKeyFrames(10) / GetConnectedKeyFrames() -> index lists), calls the orbfe_kfdb_* /
// orbfe_detect_* entry point that restates the reference method, and maps the indices back.  No scoring or selection logic here.
#include "KeyFrameDatabase.h"

#include <stdexcept>

#include "../../include/orbfe.h"
#include "compat_util.h"

namespace ORB_SLAM2
{
using namespace orbfe_compat;

namespace
{
template <class Bow> void flatten(const Bow &bow, std::vector<uint32_t> &words, std::vector<float> &weights)
{
    words.clear(); weights.clear();
    words.reserve(bow.size() + 1); weights.reserve(bow.size() + 1); // data() stays non-null for an empty vector
    for (typename Bow::const_iterator it = bow.begin(); it != bow.end(); ++it) { words.push_back(it->first); weights.push_back((float)it->second); }
}
} // namespace

KeyFrameDatabase::KeyFrameDatabase(fbow::Vocabulary *voc) : mpFBOWVoc(voc), mCtx(NULL) {}

namespace
{
// A device context holds ONE keyframe database (the reference has one per System, src/System.cc:88): a second KeyFrameDatabase
// object on the same context would clear and interleave the first one's entries, so ownership is explicit.
std::mutex g_owner_mu;
std::map<orbfe_context *, const KeyFrameDatabase *> g_owner;
} // namespace

KeyFrameDatabase::~KeyFrameDatabase()
{
    std::lock_guard<std::mutex> lk(g_owner_mu);
    std::map<orbfe_context *, const KeyFrameDatabase *>::iterator it = g_owner.find(mCtx);
    if (mCtx && it != g_owner.end() && it->second == this) g_owner.erase(it);
}

orbfe_context *KeyFrameDatabase::Context()
{
    if (!mCtx) {
        orbfe_context *ctx = ORBextractor::DefaultContext();
        if (!ctx) throw std::runtime_error("KeyFrameDatabase: no ORBextractor device context exists in this process yet");
        {
            std::lock_guard<std::mutex> lk(g_owner_mu);
            if (g_owner.count(ctx) && g_owner[ctx] != this)
                throw std::runtime_error("KeyFrameDatabase: this device context already serves another KeyFrameDatabase object (one database per context)");
            g_owner[ctx] = this;
        }
        mCtx = ctx;
        check(mCtx, orbfe_kfdb_clear(mCtx));
    }
    return mCtx;
}

// src/KeyFrameDatabase.cc:38-44
void KeyFrameDatabase::add(KeyFrame *pKF)
{
    std::unique_lock<std::mutex> lock(mMutex);
    orbfe_context *ctx = Context();
    std::vector<uint32_t> words;
    std::vector<float> weights;
    flatten(pKF->mFbowVec, words, weights);
    int idx = -1;
    check(ctx, orbfe_kfdb_add(ctx, words.data(), weights.data(), (int)words.size(), &idx));
    if (idx != (int)mvKeyFrames.size()) throw std::runtime_error("KeyFrameDatabase::add: the device database is shared with another KeyFrameDatabase object");
    mvKeyFrames.push_back(pKF);
    mvRelocScore.push_back(0.f);
    mIndexOf[pKF] = idx;
}

// :46-62
void KeyFrameDatabase::erase(KeyFrame *pKF)
{
    std::unique_lock<std::mutex> lock(mMutex);
    std::map<KeyFrame *, int>::iterator it = mIndexOf.find(pKF);
    if (it == mIndexOf.end()) return; // the reference's loops find nothing to erase
    orbfe_context *ctx = Context();
    check(ctx, orbfe_kfdb_erase(ctx, it->second));
    mvKeyFrames[it->second] = NULL;
    mIndexOf.erase(it);
}

// :64-70
void KeyFrameDatabase::clear()
{
    std::unique_lock<std::mutex> lock(mMutex);
    if (mCtx || ORBextractor::DefaultContext()) { orbfe_context *ctx = Context(); check(ctx, orbfe_kfdb_clear(ctx)); }
    mvKeyFrames.clear(); mIndexOf.clear(); mvRelocScore.clear();
}

// every entry's GetBestCovisibilityKeyFrames(10) (:270, :150) as index lists; neighbours outside the database can never carry the
// query's id (only database members are visited by the word loops, :201-218), so leaving them out changes nothing
void KeyFrameDatabase::Covisibility(std::vector<int32_t> &off, std::vector<int32_t> &idx)
{
    off.assign(1, 0); idx.clear();
    for (size_t k = 0; k < mvKeyFrames.size(); k++) {
        if (mvKeyFrames[k]) {
            const std::vector<KeyFrame *> vpNeighs = mvKeyFrames[k]->GetBestCovisibilityKeyFrames(10);
            for (size_t j = 0; j < vpNeighs.size(); j++) {
                std::map<KeyFrame *, int>::const_iterator it = mIndexOf.find(vpNeighs[j]);
                if (it != mIndexOf.end()) idx.push_back(it->second);
            }
        }
        off.push_back((int32_t)idx.size());
    }
    if (idx.empty()) idx.push_back(0);
}

// :196-307
std::vector<KeyFrame *> KeyFrameDatabase::DetectRelocalizationCandidates(Frame *F)
{
    std::unique_lock<std::mutex> lock(mMutex);
    if (mvKeyFrames.empty()) return std::vector<KeyFrame *>();
    orbfe_context *ctx = Context();
    std::vector<uint32_t> words;
    std::vector<float> weights;
    flatten(F->mFbowVec, words, weights);
    std::vector<int32_t> off, idx, cand(mvKeyFrames.size());
    Covisibility(off, idx);
    int n = 0;
    check(ctx, orbfe_detect_reloc_candidates(ctx, words.data(), weights.data(), (int)words.size(), off.data(), idx.data(), mvRelocScore.data(),
                                             cand.data(), (int)cand.size(), &n));
    for (size_t k = 0; k < mvKeyFrames.size(); k++)
        if (mvKeyFrames[k]) mvKeyFrames[k]->mRelocScore = mvRelocScore[k]; // public member of the reference's KeyFrame (:248)
    std::vector<KeyFrame *> vpRelocCandidates;
    for (int i = 0; i < n; i++) vpRelocCandidates.push_back(mvKeyFrames[cand[i]]);
    return vpRelocCandidates;
}

// :73-194
std::vector<KeyFrame *> KeyFrameDatabase::DetectLoopCandidates(KeyFrame *pKF, float minScore)
{
    const std::set<KeyFrame *> spConnectedKeyFrames = pKF->GetConnectedKeyFrames(); // :75, before the lock as in the reference
    std::unique_lock<std::mutex> lock(mMutex);
    if (mvKeyFrames.empty()) return std::vector<KeyFrame *>();
    orbfe_context *ctx = Context();
    std::vector<uint32_t> words;
    std::vector<float> weights;
    flatten(pKF->mFbowVec, words, weights);
    std::vector<uint8_t> connected(mvKeyFrames.size(), 0);
    for (size_t k = 0; k < mvKeyFrames.size(); k++) connected[k] = mvKeyFrames[k] && spConnectedKeyFrames.count(mvKeyFrames[k]);
    std::vector<int32_t> off, idx, cand(mvKeyFrames.size());
    Covisibility(off, idx);
    int n = 0;
    check(ctx, orbfe_detect_loop_candidates(ctx, words.data(), weights.data(), (int)words.size(), connected.data(), minScore, off.data(), idx.data(),
                                            cand.data(), (int)cand.size(), &n));
    std::vector<KeyFrame *> vpLoopCandidates;
    for (int i = 0; i < n; i++) vpLoopCandidates.push_back(mvKeyFrames[cand[i]]);
    return vpLoopCandidates;
}

} // namespace ORB_SLAM2
