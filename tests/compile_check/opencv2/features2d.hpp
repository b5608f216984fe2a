// NOT OpenCV (see ../README.md): cv::KeyPoint and cv::DMatch live in core.hpp of this subset.
#pragma once
#include <opencv2/core.hpp>
