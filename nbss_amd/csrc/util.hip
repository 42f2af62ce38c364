// util.hip — small stream-ordered helpers used by the host-side sequencing code.
#include "launch.h"
#include "blocks.h"
#include "fold.h"
#include "foldk.h"
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
static_assert(AFF_SLICES == FK_AFF_SLICES, "foldk.h");
// (the bodies live in foldk.h: fold.hip's table kernel runs them too)
__global__ void affine_slices_kernel(float* __restrict__ part, int nwg, int naff) {
    fk_affine_slices(part, nwg, naff, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)blockIdx.y, (int)gridDim.y);
}
__global__ void affine_final_kernel(const float* __restrict__ part, int nwg, int naff, int nsl, AffSegs segs, float* __restrict__ G) {
    fk_affine_final(part, nwg, naff, nsl, segs, G, (int)(blockIdx.x * blockDim.x + threadIdx.x));
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
    if (g_fold) {  // inside a FoldScope (fold.h): the two passes join the scope's first and second stage
        g_fold->st = st;
        FoldItem it;
        it.kind = FK_AFF_SLICES;
        it.gx = (naff + 255) / 256; it.gy = nsl; it.nblk = it.gx * it.gy;
        it.u.af.part = const_cast<float*>(part); it.u.af.nwg = nwg; it.u.af.naff = naff; it.u.af.nsl = nsl; it.u.af.segs = segs; it.u.af.G = G;
        int e = g_fold->add(1, it);
        if (e) return e;
        it.kind = FK_AFF_FINAL;
        it.gy = 1; it.nblk = it.gx;
        return g_fold->add(2, it);
    }
    NBSS_FOLD_LAUNCH(affine_slices_kernel, dim3((naff + 127) / 128, nsl), dim3(128), 0, st, const_cast<float*>(part), nwg, naff);
    int e = NBSS_CHECK_LAUNCH();
    if (e) return e;
    NBSS_FOLD_LAUNCH(affine_final_kernel, dim3((naff + 127) / 128), dim3(128), 0, st, part, nwg, naff, nsl, segs, G);
    return NBSS_CHECK_LAUNCH();
}
