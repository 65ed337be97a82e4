#include "../../../sophus/se3.hpp"
