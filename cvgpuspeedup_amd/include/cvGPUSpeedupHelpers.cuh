// Source-compatibility forwarder: code written against the reference includes <cvGPUSpeedupHelpers.cuh>.
#pragma once
#include "cvGPUSpeedupHelpers.h"
