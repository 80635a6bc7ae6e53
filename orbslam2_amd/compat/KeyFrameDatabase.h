// compat/KeyFrameDatabase.h -- ORB_SLAM2::KeyFrameDatabase with the reference's public interface (include/KeyFrameDatabase.h:42-62)
// over the C ABI: put orbslam2_amd/compat before the reference's include/ on the include path and compile
// compat/KeyFrameDatabase.cc instead of src/KeyFrameDatabase.cc.  src/Tracking.cc:1496 (DetectRelocalizationCandidates),
// src/LoopClosing.cc:131 (DetectLoopCandidates), src/LocalMapping.cc / src/KeyFrame.cc (add / erase) compile unchanged.
//
// The inverted file is replaced by the keyframes' BoW vectors resident in HBM (orbfe_kfdb_*); this class keeps the
// KeyFrame* <-> database index table and KeyFrame::mRelocScore's persistent state.  Not carried over: the
// boost::serialization hook of the reference's map save / load (map IO is outside the hot path; a reloaded map calls add()).
#pragma once

#include <list>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#if defined(__has_include)
#if __has_include("Thirdparty/fbow/include/fbow/fbow.h")
#include "Thirdparty/fbow/include/fbow/fbow.h" // as the reference's header does (include/KeyFrameDatabase.h:33)
#endif
#endif
#include "Frame.h"    // (the reference's KeyFrame.h includes KeyFrameDatabase.h itself: the forward declarations below cover that cycle)
#include "KeyFrame.h"

struct orbfe_context;

namespace ORB_SLAM2
{

class KeyFrame;
class Frame;

class KeyFrameDatabase
{
public:
    KeyFrameDatabase(fbow::Vocabulary *voc);

    void add(KeyFrame *pKF);
    void erase(KeyFrame *pKF);
    void clear();

    // Loop Detection
    std::vector<KeyFrame *> DetectLoopCandidates(KeyFrame *pKF, float minScore);
    // Relocalization
    std::vector<KeyFrame *> DetectRelocalizationCandidates(Frame *F);

    KeyFrameDatabase() : mpFBOWVoc(NULL), mCtx(NULL) {}
    ~KeyFrameDatabase(); // gives the device database back (the reference never destroys its one database; tests do)
    void SetFBOWvocabulary(fbow::Vocabulary *pfbowv) { mpFBOWVoc = pfbowv; }

protected:
    orbfe_context *Context();
    void Covisibility(std::vector<int32_t> &off, std::vector<int32_t> &idx);

    fbow::Vocabulary *mpFBOWVoc;
    orbfe_context *mCtx;                 // the process's default device context (ORBextractor::DefaultContext()), taken at first use
    std::vector<KeyFrame *> mvKeyFrames; // database index -> keyframe (NULL once erased; indices are not reused)
    std::map<KeyFrame *, int> mIndexOf;
    std::vector<float> mvRelocScore;     // KeyFrame::mRelocScore of every entry, persistent across queries as in the reference
    std::mutex mMutex;
};

} // namespace ORB_SLAM2
