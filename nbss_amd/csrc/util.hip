// util.hip — small stream-ordered helpers used by the host-side sequencing code.
#include "launch.h"
#include "blocks.h"
#include <string.h>
#include <cstdlib>

// launch.h: the traversal direction of the walks' main-stream kernels
thread_local WalkFlip g_walk_flip = {0, 0u};
WalkFlipScope::WalkFlipScope(size_t stream_bytes) {
    static const int knob = [] { const char* v = getenv("NBSS_FLIP"); return v ? (v[0] == '0' ? 0 : 1) : -1; }();  // A/B knob: 0 off, 1 on at every size
    prev = g_walk_flip.on;
    g_walk_flip.on = knob >= 0 ? knob : (stream_bytes >= ((size_t)64 << 20) ? 1 : 0);
}

int memset_async_impl(void* p, size_t bytes, hipStream_t st) {
#ifdef NBSS_EMU
    memset(p, 0, bytes);
    return 0;
#else
    return hipMemsetAsync(p, 0, bytes, st) == hipSuccess ? 0 : NBSS_ELAUNCH;
#endif
}

// G[seg offsets] += sum over workgroups of part[wg][e], in a fixed order (bitwise repeatable; round 4's version ended in one atomicAdd per slice
// and element).  Two launches: (1) block (x, y) sums slice y of the workgroups' rows for 128 elements (coalesced over e, four rows in flight) and
// leaves the slice sum IN PLACE in the slice's first row — those entries were this block's alone to read; (2) one thread per element adds the
// slices' sums in slice order and adds the result to G with a plain read-modify-write (one owner per element; gradient launches of one parameter
// are stream-ordered).  The very first version walked all ~1000 workgroups in 2-5 blocks and cost 49 us per call (2 ms per training step).
#define AFF_SLICES 64
// first row of slice y: nwg y / nsl without a division (nsl is AFF_SLICES, or nwg itself when there are fewer rows than slices)
NBSS_DEV int aff_row0(int nwg, int nsl, int y) { return nsl == AFF_SLICES ? (int)(((unsigned)nwg * (unsigned)y) >> 6) : y; }
__global__ void affine_slices_kernel(float* __restrict__ part, int nwg, int naff) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= naff) return;
    const int w0 = aff_row0(nwg, gridDim.y, blockIdx.y), w1 = aff_row0(nwg, gridDim.y, blockIdx.y + 1);
    const float s = fold_strided<16>(part + e, (size_t)naff, w0, w1);
    part[(size_t)w0 * naff + e] = s;
}
__global__ void affine_final_kernel(const float* __restrict__ part, int nwg, int naff, int nsl, AffSegs segs, float* __restrict__ G) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= naff) return;
    float s16[16], v16[16];  // sixteen loads in flight; the order of the adds is fixed
#pragma unroll
    for (int k = 0; k < 16; ++k) s16[k] = 0.f;
    for (int y = 0; y < nsl; y += 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v16[k] = y + k < nsl ? part[(size_t)aff_row0(nwg, nsl, y + k) * naff + e] : 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s16[k] += v16[k];
    }
#pragma unroll
    for (int h = 8; h >= 1; h >>= 1) {
#pragma unroll
        for (int k = 0; k < h; ++k) s16[k] += s16[k + h];
    }
    const float s = s16[0];
    int r = e;
    for (int i = 0; i < segs.n; ++i) {
        if (r < segs.cnt[i]) {
            G[segs.off[i] + r] += s;
            return;
        }
        r -= segs.cnt[i];
    }
}

// (Tried, round 5: both passes in ONE launch — the block of an element range that draws the last ticket of a counter adds the slice sums.  The
//  device-scope release / acquire around the ticket is a write-back + invalidate of the XCD's whole L2 on gfx950 (buffer_wbl2 sc1 / buffer_inv sc1), per
//  block, beside the main stream's kernels: the step went 690 -> 623 utt/s at batch 32, 340 -> 228 at batch 2.  Two launches it stays.)
// (the partial rows are consumed: `part` is scratch that the next backward call of the sub-block overwrites)
int affine_reduce_launch(const float* part, int nwg, const AffSegs& segs, float* G, hipStream_t st) {
    int naff = 0;
    for (int i = 0; i < segs.n; ++i) naff += segs.cnt[i];
    const int nsl = nwg < AFF_SLICES ? nwg : AFF_SLICES;
    if (nsl < 1) return NBSS_OK;
    NBSS_FOLD_LAUNCH(affine_slices_kernel, dim3((naff + 127) / 128, nsl), dim3(128), 0, st, const_cast<float*>(part), nwg, naff);
    int e = NBSS_CHECK_LAUNCH();
    if (e) return e;
    NBSS_FOLD_LAUNCH(affine_final_kernel, dim3((naff + 127) / 128), dim3(128), 0, st, part, nwg, naff, nsl, segs, G);
    return NBSS_CHECK_LAUNCH();
}
