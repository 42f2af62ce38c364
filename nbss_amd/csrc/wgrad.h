// Descriptor of one weight-gradient contraction (see wgrad.hip).
#pragma once
#include "launch.h"

struct WgradArgs {
    const void* A;   // dY  [Ntok][lda], columns [0, MA)           (stream dtype)
    int lda, MA;
    const void* B;   // X   [Ntok][ldb], columns [0, NB)           (stream dtype)
    int ldb, NB;
    int groups;      // block-diagonal: MA/groups x NB/groups per group
    int mvalid, nvalid;  // valid rows / cols per group (<= MA/groups, NB/groups); 0 = all.  dW is [groups*mvalid][nvalid][taps]
    int taps;        // 1 = dense
    int shift_stride;  // rows per tap step: 1 (T-conv) or T (F-conv)
    int shift_dim;     // 0: frame index t = n % T must stay in [0,T); 1: f = (n / T) % F in [0,F)
    const float* stats;  // optional [Ntok][2] (mean, rstd): X := LayerNorm(X) with gamma/beta
    const float* gamma;
    const float* beta;
    float* dW;       // [MA][NB/groups][taps], accumulated with atomicAdd
    float* dbias;    // optional [MA]: column sums of dY
    int Ntok, F, T;
    float* part = nullptr;  // optional scratch (layout.h: ws_wgpart_offset / WGPART_BYTES): per-workgroup partial tiles + a reduce pass replace
                            // the 256-deep same-address atomicAdd flush (31% of the kernel's time when measured)
    // optional group-major operand layouts [cols / gw][Ntok][gw] (gw = 0: plain [Ntok][ld]); transposing-read kernel only
    int a_gw = 0, a_gs = 0, b_gw = 0, b_gs = 0;  // group width (elements) and group stride (elements)
    int dbg = 0;  // diagnostic build only: NBSS_WG_DEBUG probe bits (1 no flush, 2 no MFMA, 4 no loads, 8 no LDS stash, 16 prologue only)
};

int wgrad_launch(const WgradArgs& a, int dtype, hipStream_t st);
