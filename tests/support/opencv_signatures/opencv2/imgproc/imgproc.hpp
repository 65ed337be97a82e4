// TEST INFRASTRUCTURE — declarations only (see ../core/core.hpp): the imgproc functions of this path with OpenCV 4.x's signatures.
#pragma once
#include "../core/core.hpp"
namespace cv {
enum InterpolationFlags { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_CUBIC = 2, INTER_AREA = 3 };
CV_EXPORTS_W void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sigmaX, double sigmaY = 0, int borderType = BORDER_DEFAULT);
CV_EXPORTS_W void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
CV_EXPORTS_W void copyMakeBorder(InputArray src, OutputArray dst, int top, int bottom, int left, int right, int borderType);
}  // namespace cv
