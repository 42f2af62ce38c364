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

// launches that only fold partial parameter gradients (timing knock-out, A/B flavour -DNBSS_KO_FOLDS: what would a step cost without them?)
#ifdef NBSS_KO_FOLDS
#define NBSS_FOLD_LAUNCH(kern, grid, block, lds, stream, ...) ((void)0)
#else
#define NBSS_FOLD_LAUNCH(kern, grid, block, lds, stream, ...) NBSS_LAUNCH(kern, grid, block, lds, stream, __VA_ARGS__)
#endif

// Traversal direction of the walks' main-stream kernels (round 6).  At batch 32 a stream tensor is 199 MB and the Infinity Cache 256 MB: when every kernel
// walks the utterances in the same order, what a kernel reads first is what its producer wrote first — long evicted.  Consecutive main-stream kernels of
// a walk therefore alternate the order (block -> work item mirrored: bid = grid - 1 - blockIdx): the consumer starts with the lines its producer touched
// last.  Measured on the full-band kernels alone (same call, in-order trace): full_fwd 188.0 -> 175.3 us, full_bwd 536.9 -> 523.9, the F-conv that reads
// full_fwd's output 138.3 -> 135.8.  Work items, partial-row indices and fold orders are those of the logical index: results do not depend on the direction
// (only the tail kernels' chunk -> workgroup assignment does; a kernel of a given layer always gets the same direction: gradients stay bitwise repeatable).
// The per-block C entry points run unflipped; NBSS_FLIP=0 switches the alternation off (A/B).
struct WalkFlip {
    int on;      // inside a walk with the alternation enabled
    unsigned k;  // main-stream kernels launched so far
};
extern thread_local WalkFlip g_walk_flip;
inline int walk_flip_next() { return g_walk_flip.on ? (int)(g_walk_flip.k++ & 1u) : 0; }
struct WalkFlipScope {  // capi.hip: the walks.  stream_bytes = one stream tensor of the batch: the alternation is switched on from 64 MB (below, producer and
    int prev;           // consumer share the 256 MB cache whatever the order: batch 8 measured 606 -> 598 utt/s with it, batch 16 694 -> 698, batch 32 718 -> 723)
    explicit WalkFlipScope(size_t stream_bytes);
    ~WalkFlipScope() { g_walk_flip.on = prev; }
};
NBSS_DEV int flip_bid(int flip) { return flip ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x; }

// error codes of the C ABI (include/nbss_hip.h)
#define NBSS_OK 0
#define NBSS_EINVAL (-1)
#define NBSS_EUNSUPPORTED (-2)
#define NBSS_ELAUNCH (-3)
