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

// G[seg offsets] += sum over workgroups of part[wg][e].  Block (x, y) sums slice y of the workgroups for 128 elements
// (coalesced over e) and adds it with one atomicAdd per element; the first version walked all ~1000 workgroups in 2-5 blocks
// and cost 49 us per call (2 ms per training step).
#define AFF_SLICES 64
__global__ void affine_reduce_kernel(const float* __restrict__ part, int nwg, int naff, AffSegs segs, float* __restrict__ G) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= naff) return;
    const int w0 = (int)((long)nwg * blockIdx.y / gridDim.y), w1 = (int)((long)nwg * (blockIdx.y + 1) / gridDim.y);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = w0;
    for (; w + 4 <= w1; w += 4) {
        s0 += part[(size_t)w * naff + e];
        s1 += part[(size_t)(w + 1) * naff + e];
        s2 += part[(size_t)(w + 2) * naff + e];
        s3 += part[(size_t)(w + 3) * naff + e];
    }
    for (; w < w1; ++w) s0 += part[(size_t)w * naff + e];
    int r = e;
    for (int i = 0; i < segs.n; ++i) {
        if (r < segs.cnt[i]) {
            atomicAdd(G + segs.off[i] + r, (s0 + s1) + (s2 + s3));
            return;
        }
        r -= segs.cnt[i];
    }
}

int affine_reduce_launch(const float* part, int nwg, const AffSegs& segs, float* G, hipStream_t st) {
    int naff = 0;
    for (int i = 0; i < segs.n; ++i) naff += segs.cnt[i];
    NBSS_LAUNCH(affine_reduce_kernel, dim3((naff + 127) / 128, nwg < AFF_SLICES ? nwg : AFF_SLICES), dim3(128), 0, st, part, nwg, naff, segs, G);
    return NBSS_CHECK_LAUNCH();
}
