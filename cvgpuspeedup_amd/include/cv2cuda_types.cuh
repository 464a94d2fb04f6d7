// Source-compatibility forwarder: code written against the reference includes <cv2cuda_types.cuh>.
#pragma once
#include "cv2cuda_types.h"
