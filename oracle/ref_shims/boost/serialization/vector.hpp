// TEST INFRASTRUCTURE: see serialization.hpp (the reference's include/KeyFrameDatabase.h:32-34 names this header; nothing of it is used)
#pragma once
#include "serialization.hpp"
