// util.hip — small stream-ordered helpers used by the host-side sequencing code.
#include "launch.h"
#include "blocks.h"
#include <string.h>

int memset_async_impl(void* p, size_t bytes, hipStream_t st) {
#ifdef NBSS_EMU
    memset(p, 0, bytes);
    return 0;
#else
    return hipMemsetAsync(p, 0, bytes, st) == hipSuccess ? 0 : NBSS_ELAUNCH;
#endif
}

// G[seg offsets] += sum over workgroups of part[wg][e]   (one thread per element, coalesced over e)
__global__ void affine_reduce_kernel(const float* __restrict__ part, int nwg, int naff, AffSegs segs, float* __restrict__ G) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= naff) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = 0;
    for (; w + 4 <= nwg; w += 4) {
        s0 += part[(size_t)w * naff + e];
        s1 += part[(size_t)(w + 1) * naff + e];
        s2 += part[(size_t)(w + 2) * naff + e];
        s3 += part[(size_t)(w + 3) * naff + e];
    }
    for (; w < nwg; ++w) s0 += part[(size_t)w * naff + e];
    int r = e;
    for (int i = 0; i < segs.n; ++i) {
        if (r < segs.cnt[i]) {
            G[segs.off[i] + r] += (s0 + s1) + (s2 + s3);
            return;
        }
        r -= segs.cnt[i];
    }
}

int affine_reduce_launch(const float* part, int nwg, const AffSegs& segs, float* G, hipStream_t st) {
    int naff = 0;
    for (int i = 0; i < segs.n; ++i) naff += segs.cnt[i];
    NBSS_LAUNCH(affine_reduce_kernel, dim3((naff + 127) / 128), dim3(128), 0, st, part, nwg, naff, segs, G);
    return NBSS_CHECK_LAUNCH();
}
