// Source-compatibility forwarder: code written against the reference includes <cvGPUSpeedup.cuh>.
#pragma once
#include "cvGPUSpeedup.h"
