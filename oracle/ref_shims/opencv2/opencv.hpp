#pragma once
#include "core/core.hpp"
#include "imgproc/imgproc.hpp"
#include "features2d/features2d.hpp"
