// TEST-ONLY: <opencv2/core/core.hpp> as orbslam2_amd/host/ORBextractor.h includes it under ORBFE_WITH_OPENCV -> the stand-in types
#pragma once
#include "../../cvstub.h"
