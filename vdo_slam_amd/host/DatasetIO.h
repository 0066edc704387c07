// Dataset IO of the reference's example driver (example/vdo_slam.cc:105-122, 253-450) without OpenCV: the step in front of
// System::TrackRGBD.  Middlebury .flo (cv::optflow::readOpticalFlow), the text instance masks (LoadMask: one row of integers per
// image row - ~466k integers per KITTI frame, parsed here without stringstreams), and 8/16-bit PNG (the disparity maps are
// 16-bit grey PNGs; cv::imread(..., UNCHANGED) + convertTo(CV_32F)).  Host only.
#pragma once
#include <string>

#include "minicv.h"

namespace VDO_SLAM {

// CV_32FC2, rows x cols from the file header.  false: unreadable / not a .flo file.
bool ReadOpticalFlow(const std::string& path, cv::Mat& flow);
// mask: CV_32SC1, allocated by the caller (rows x cols like the image); entries are the file's integers (missing ones stay 0).
bool LoadMask(const std::string& path, cv::Mat& mask);
// Non-interlaced PNG, grey or RGB(A), 8 or 16 bit.  as_float: CV_32F with the sample values (grey only: the disparity path);
// otherwise CV_8UC1 / CV_8UC3 / CV_8UC4 in BGR(A) order like cv::imread (16-bit colour is not supported).
bool ReadPNG(const std::string& path, cv::Mat& img, bool as_float);

}  // namespace VDO_SLAM
