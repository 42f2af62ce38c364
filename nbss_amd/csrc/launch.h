// Kernel-launch shim: the ONLY place that differs between the gfx950 build and the host
// emulator build used by the CPU tests (tests/hipemu, -DNBSS_EMU).
#pragma once
#include "common.h"

#ifdef NBSS_EMU
#define NBSS_LAUNCH(kern, grid, block, lds, stream, ...) \
    hipemu::launch((grid), (block), (lds), [=]() { kern(__VA_ARGS__); })
#define NBSS_CHECK_LAUNCH() 0
#define NBSS_SET_MAX_LDS(kern, bytes) 0
#else
#define NBSS_LAUNCH(kern, grid, block, lds, stream, ...) hipLaunchKernelGGL(kern, (grid), (block), (lds), (stream), __VA_ARGS__)
#define NBSS_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -3)
#define NBSS_SET_MAX_LDS(kern, bytes) \
    (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) == hipSuccess ? 0 : -4)
#endif

// error codes of the C ABI (include/nbss_hip.h)
#define NBSS_OK 0
#define NBSS_EINVAL (-1)
#define NBSS_EUNSUPPORTED (-2)
#define NBSS_ELAUNCH (-3)
