// fold.h — several folds of partial parameter gradients in ONE launch (round 6).  A backward sub-block used to end in a chain of small dependent launches
// on the gradient stream — per layer 25 of them (wgrad_reduce x 5, the two-launch affine fold x 4, tailw_finalize + tailw_affine x 2, the bf16 row folds,
// the squeeze finalize): at batch 2 a fifth of the step.  Inside a FoldScope the launch sites hand their fold to the scope's batch instead (descriptor =
// the kernel's arguments + its grid), and the scope's end launches one table kernel per dependency STAGE:
//   stage 1  first passes: wgrad_reduce, affine slices, tailw_finalize, bf16 row slices
//   stage 2  what reads a first pass's output: affine final, tailw_affine, the row folds' final pass
//   stage 3  what reads stage 2: the full-band block's squeeze finalize
// A sub-block then costs one launch per stage it uses (12 per layer instead of 25).  The bodies are the single kernels' (foldk.h): same order of every sum.
// Partial tiles / slice scratch of the batched folds must stay alive until the scope ends: they are bump-allocated from the scope's pool (the wgrad
// partial-tile region of the sub-block's workspace copy); a request that does not fit flushes the batch first.
#pragma once
#include "launch.h"
#include "blocks.h"
#include "wgrad.h"

enum FoldKind { FK_WGRAD_REDUCE = 0, FK_AFF_SLICES, FK_AFF_FINAL, FK_TAILW_FIN, FK_TAILW_AFF, FK_P16_SLICES, FK_TCONV_FINAL, FK_FCONV_FINAL, FK_FULL_SQ };

struct FoldItem {
    int kind;
    int nblk;    // blocks of 256 threads
    int gx, gy;  // logical grid of the single-kernel launch: block i -> (i % gx, i / gx)
    union U {
        struct { WgradArgs a; int xb, nt_major; } wr;                                                                  // grid (4 ntot, ybl)
        struct { float* part; int nwg, naff, nsl; AffSegs segs; float* G; } af;                                         // slices: grid (ceil(naff / 256), nsl); final: (ceil(naff / 256))
        struct { float* part; int xb, MTA, ntot; const float *W, *gamma, *beta; float *dW, *dbias, *dgamma, *dbeta; } tw;  // finalize: grid (4 ntot); affine: (1)
        struct { const void* part16; int nrows, p16, nsl; float* slices; float* G; long long off[4]; } p16;             // slices: grid (ceil(p16 / 8 / 256), nsl); finals: (ceil(p16 / 256))
        struct { const float *tmp, *Ws, *gamma, *beta; float *dWs, *dbs, *dgamma, *dbeta; } sq;                         // grid (1)
        U() {}
    } u;
    FoldItem() : kind(0), nblk(0), gx(1), gy(1) {}
};

#define FOLD_MAX_ITEMS 6
#define FOLD_STAGES 3
struct FoldTable {
    int n;
    int blk0[FOLD_MAX_ITEMS + 1];  // first block of item k
    FoldItem it[FOLD_MAX_ITEMS];
};

static_assert(sizeof(FoldTable) <= 4096, "the table travels as a by-value kernel argument");

struct FoldBatch {
    hipStream_t st;
    char* pool;
    size_t pool_bytes, used;
    FoldTable tab[FOLD_STAGES];
    void reset() {
        used = 0;
        for (int s = 0; s < FOLD_STAGES; ++s) { tab[s].n = 0; tab[s].blk0[0] = 0; }
    }
    int flush();                                 // launches the pending stages in order; the pool is free again
    int add(int stage, const FoldItem& it);      // stage 1..3
    void* alloc(size_t bytes, int* err);         // scratch that lives until the next flush (flushes first when the pool is full); nullptr: larger than the pool
};
extern thread_local FoldBatch* g_fold;  // the batch of the enclosing FoldScope, or nullptr: fold launches go out one by one

// RAII: sub-block backward code opens one around its gradient-stream section.  `pool` = the wgrad partial-tile region of the sub-block's workspace.
struct FoldScope {
    FoldBatch fb;
    FoldBatch* prev;
    bool open;
    FoldScope(hipStream_t st, void* pool, size_t pool_bytes, size_t ntokens);
    int end();  // flush + close (call before returning; the destructor flushes too but cannot report)
    ~FoldScope();
};
// The scopes open for SMALL grids only (up to FOLD_MAX_TOKENS tokens: batch <= 4 at 129 x 251).  Measured, same call, batched vs single launches: batch 1
// 222 -> 240 utt/s, batch 2 352 -> 372, batch 3 424 -> 435, batch 4 507 -> 516 — but batch 8 620 -> 614 and batch 16 708 -> 703 (batch 32: equal): from there
// on the folds are bytes, not launch latency, and the deferred second passes find their partial tiles spread over the pool instead of one cache-hot region.
// NBSS_FOLD_BATCH=0: never (every fold as its own launch), =1: at every size (A/B).
#define FOLD_MAX_TOKENS ((size_t)4 * 129 * 256)
bool fold_batch_enabled(size_t ntokens);
