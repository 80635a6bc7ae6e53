// SYNTAX-CHECK ONLY: the members of the reference's ORB_SLAM2::ORBextractor (include/ORBextractor.h:45-121) that pin.cpp touches, as
// declarations.  tests/test_reference_pins.py checks each of them against the reference's own header where that tree is present.
#pragma once
#include <vector>
#include <opencv2/core.hpp>
namespace ORB_SLAM2
{
class ORBextractor
{
public:
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int patchSize, int halfPatchSize, int edgeThreshold);
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints, cv::OutputArray descriptors);
    std::vector<float> GetScaleFactors();
    std::vector<float> GetInverseScaleFactors();
    std::vector<float> GetScaleSigmaSquares();
    std::vector<float> GetInverseScaleSigmaSquares();
    std::vector<cv::Mat> mvImagePyramid;
};
} // namespace ORB_SLAM2
