// Descriptor of the grouped T-convolution chain of the T-ConvFFN block on the geometry-generic path (tchain.hip).
#pragma once
#include "launch.h"

struct TChain {
    const void* a1;    // [N][FFN] pre-activation of the first linear map (stream dtype bf16)
    const void* dh5;   // [N][FFN] gradient w.r.t. h5 = SiLU(a5) (backward only)
    const void* wf[3];  // fragment-ordered weights of conv 1..3 (tc_wprep), forward
    const void* wd[3];  // ... and of their data gradients
    const float* cb[3];  // conv biases [FFN]
    const float* gn_w;
    const float* gn_b;
    void* h1;  // outputs, [N][FFN] each; h1 h2 h4 g5 g3 g2 may be null (operands of the weight gradients only)
    void* h2;
    void* h4;
    void* h5;
    void* g5;
    void* g3;
    void* g2;
    void* g1;
    float* dgn_w;  // accumulated
    float* dgn_b;
    float* part;   // [nseq][2 FFN] per-sequence rows of the two (affine_reduce folds them), or null: atomics
    int nseq, T, FFN, groups;
};

bool tc_chain_takes(int dtype, int CG, int KS, int T);
size_t tc_wfrag_elems(int groups, int CG, int KS);  // elements of ONE of the six fragment-ordered weight sets
// re-lay the three conv weights [FFN][CG][KS] (fp32) into wf[0..2] / wd[0..2] (one launch)
int tc_wprep(const float* const w[3], void* const wf[3], void* const wd[3], int groups, int CG, int KS, hipStream_t st);
// ... and one conv weight [groups CG][CG][KS] alone (fconv_g.hip)
int tc_wprep_one(const float* w, void* wf, void* wd, int groups, int CG, int KS, hipStream_t st);
int tc_chain_launch(const TChain& p, int CG, int KS, bool bwd, hipStream_t st);

// fconv_g.hip: the F-conv block's backward of the geometry-generic path in one kernel (bf16, 192 channels)
struct nbss_cfg;
bool fconv_g_takes(const nbss_cfg& c);
size_t fconv_g_wfrag_elems();
int fconv_g_bwd(const nbss_cfg& c, const float* P, float* G, int layer, int which, const void* x, const void* dy, void* dx, void* dv, float* stats, void* wf,
                void* wd, float* part, hipStream_t st);
