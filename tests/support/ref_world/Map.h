// TEST INFRASTRUCTURE: see ref_world.h
#include "ref_world.h"
