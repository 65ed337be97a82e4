#include "../../GeometricCamera.h"
