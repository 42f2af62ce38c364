// util.hip — small stream-ordered helpers used by the host-side sequencing code.
#include "launch.h"
#include <string.h>

int memset_async_impl(void* p, size_t bytes, hipStream_t st) {
#ifdef NBSS_EMU
    memset(p, 0, bytes);
    return 0;
#else
    return hipMemsetAsync(p, 0, bytes, st) == hipSuccess ? 0 : NBSS_ELAUNCH;
#endif
}
