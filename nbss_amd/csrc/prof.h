// Opt-in per-kernel timing with HIP events recorded on the launch stream (diagnostics for bench.py's
// roofline line; off by default, zero cost when off).
#pragma once
#include "launch.h"

enum ProfKernel {
    PK_ENC_F = 0, PK_FCONV_F, PK_FULL_F, PK_MHSA_F, PK_TCF_F, PK_DEC_F,
    PK_DEC_B, PK_TCF_B, PK_MHSA_B, PK_FCONV_B, PK_FULL_B, PK_WGRAD,
    PK_STFT, PK_ISTFT, PK_ISTFT_B, PK_LOSS, PK_ADAM, PK_PACK, PK_COUNT
};

void prof_begin(int id, hipStream_t st);
void prof_end(int id, hipStream_t st);

struct ProfScope {
    int id;
    hipStream_t st;
    ProfScope(int i, hipStream_t s) : id(i), st(s) { prof_begin(id, st); }
    ~ProfScope() { prof_end(id, st); }
};
