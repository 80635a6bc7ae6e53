// the extractor a Frame points to is the mirror over the C ABI (an integration replaces include/ORBextractor.h by it)
#pragma once
#include "../../orbslam2_amd/host/ORBextractor.h"
