// tests/hipemu/hip/hip_ext.h -- TEST INFRASTRUCTURE: hipExtLaunchKernelGGL lives in hip_runtime.h of the emulator
#pragma once
#include "hip_runtime.h"
