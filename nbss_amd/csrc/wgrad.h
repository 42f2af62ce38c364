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
};

int wgrad_launch(const WgradArgs& a, int dtype, hipStream_t st);
