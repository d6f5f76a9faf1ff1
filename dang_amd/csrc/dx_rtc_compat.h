// dx_rtc_compat.h -- the two system headers of the device code, or their stand-ins when a kernel is specialised at run time:
// hiprtc (dangx_rtc.hip) compiles the same headers without a host tool chain, with the HIP device declarations built in.
#pragma once
#ifdef __HIPCC_RTC__
typedef unsigned long long uint64_t;
typedef long long int64_t;
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned long uintptr_t;
#ifndef INFINITY
#define INFINITY __builtin_huge_val()
#endif
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif
