// Descriptor of one tap-GEMM (gbwd.hip: every per-token linear map, grouped convolution and data gradient of the geometry-generic path):
//   Y[n][o] = sum_tap sum_i X[n + (tap - center) shift][i] W[g][tap][o][i]  (+ bias, activations, residual)
#pragma once
#include "launch.h"

struct TapGemm {
    const void* X;
    const void* W;      // prepared weights
    const float* bias;  // [groups][bgs] or null
    const void* R;      // optional residual, rows of ldr elements, columns as Y
    void* Y;
    int rows;
    int ldx, xcol, xgs;  // X row stride, first column, column stride between groups
    int ldy, ycol, ygs;
    int ldr;
    int groups, Mg, Kg, Mp, Kp, bgs;
    int taps, center, shift;  // tap reads row n + (tap - center) * shift ...
    int pos_div, pos_len;     // ... valid while 0 <= (n / pos_div) % pos_len + tap - center < pos_len
    int xact, yact;           // SiLU on the loaded X / on the result
    void* Y2;                 // optional second output, layout of Y: SiLU(result) next to the pre-activation (a and h of a forward step in one pass)
    const void* Dact;         // optional pre-activation tensor, layout of Y: the result is multiplied by SiLU'(Dact) (dh -> da of a backward step)
};

// gemm_g.hip: the dense (one tap, one group) bf16 problems with K % 32 == 0 on 128 x 192 workgroup tiles fed by an LDS-DMA ring
bool gl_gemm_takes(const TapGemm& p);
int gl_gemm_bf16(const TapGemm& p, hipStream_t st);
