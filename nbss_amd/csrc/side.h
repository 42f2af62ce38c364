// side.h — the gradient stream of the whole-network backward walk (capi.hip: nbss_spatialnet_bwd_range).
//
// A backward sub-block is a few kernels on the critical path (data gradient: dy -> dx) followed by launches that only produce PARAMETER
// gradients (partial-row folds, the token-contraction weight-gradient problems, finalize kernels).  In one stream the second kind sits
// between one sub-block's dx and the next sub-block that needs it: 5.6 of 52 ms per step at batch 32, 1.6 of 8 ms at batch 2 — where it is ~25
// launch-latency-bound kernels per layer, and where the row kernels' second round leaves 254 of 256 CUs idle for half of their time.  The walk
// therefore hands every sub-block a Side: at the fork point the main stream records `ready`, the gradient stream waits for it, and everything
// launched on the returned stream afterwards overlaps with the next sub-blocks' data-gradient kernels.  The walk owns the other half of the
// contract (one workspace copy per sub-block kind, three rotating gradient buffers, `done` events before anything is overwritten, a join at
// the end).  sd == nullptr (the per-block C entry points, NBSS_SIDE_STREAM=0, the host emulator): everything stays in order on one stream.
#pragma once
#include "launch.h"
#include "../../include/nbss_hip.h"

#ifdef NBSS_EMU
struct Side { hipStream_t gs; };
inline hipStream_t side_fork(const Side*, hipStream_t st) { return st; }
#else
struct Side {
    hipStream_t gs;    // gradient stream
    hipEvent_t ready;  // fork event, recorded on the main stream
};
inline hipStream_t side_fork(const Side* sd, hipStream_t st) {
    if (!sd || sd->gs == st) return st;
    if (hipEventRecord(sd->ready, st) != hipSuccess || hipStreamWaitEvent(sd->gs, sd->ready, 0) != hipSuccess) return st;  // (in order: still correct)
    return sd->gs;
}
#endif

// Forward walk, bf16 row kernels (one workgroup per (b, f) sequence, one per CU): 129 B sequences on 256 CUs end in a round that is mostly empty
// (batch 32: 16 rounds + 32 sequences; batch 2: one round + 2 sequences).  The attention -> T-ConvFFN pair is per-sequence end to end, so the walk
// launches those last `n` sequences of both kernels on its second stream: they overlap with the main launches' full rounds instead of
// holding the whole chip for a round each.
struct SeqTail {
    int n;           // sequences of the tail launch (the last n of B F)
    hipStream_t ts;  // its stream
};

// gbwd.hip: geometry-generic backward (every geometry but SpatialNet-small)
int gb_fconv_bwd(const nbss_cfg& c, const float* P, float* G, int layer, int which, const void* x, const void* dy, void* dx, void* ws, hipStream_t st, const Side* sd);
int gb_full_bwd(const nbss_cfg& c, const float* P, float* G, int layer, const void* x, const void* dy, void* dx, void* ws, hipStream_t st, const Side* sd);
int gb_mhsa_bwd(const nbss_cfg& c, const float* P, float* G, int layer, const void* x, const void* dy, void* dx, void* ws, hipStream_t st, const Side* sd);
int gb_tconvffn_bwd(const nbss_cfg& c, const float* P, float* G, int layer, const void* x, const void* dy, void* dx, void* ws, hipStream_t st, const Side* sd);
int gb_decoder_bwd(const nbss_cfg& c, const float* P, float* G, const void* x, const float* dout, void* dx, void* ws, hipStream_t st);
