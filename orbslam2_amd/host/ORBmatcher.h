// ORBmatcher.h -- host-side mirror of ORB_SLAM2::ORBmatcher (include/ORBmatcher.h:38-102).  See below.
#pragma once
#include "../../include/orbfe.h"
namespace ORB_SLAM2
{
class ORBmatcher
{
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
    static const int TH_LOW = 50;   // src/ORBmatcher.cc:35-37
    static const int TH_HIGH = 100;
    static const int HISTO_LENGTH = 30;
    // ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1643-1659) on two 32-byte rows.
    static int DescriptorDistance(const uint8_t *a, const uint8_t *b)
    {
        int dist = 0;
        for (int i = 0; i < 32; i++) dist += __builtin_popcount((unsigned)(a[i] ^ b[i]));
        return dist;
    }
protected:
    float mfNNratio;
    bool mbCheckOrientation;
};
} // namespace ORB_SLAM2
