// capi.hip — the extern "C" surface declared in include/nbss_hip.h.  Thin argument checking
// and dtype dispatch only; kernels live in the sibling .hip files.
#include "launch.h"
#include "layout.h"
#include "side.h"
#include <stdlib.h>

int pack_params_impl(const nbss_cfg& c, const float* params, void* packed, hipStream_t stream);
int selftest_mma_impl(int dtype, int kperm, const float* A, const float* B, float* D, hipStream_t stream);
int encoder_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, const void* xin, void* y, hipStream_t st);
int decoder_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, const void* x, float* out, hipStream_t st);
int fconv_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, int which, const void* x, void* y, hipStream_t st);
int full_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st);
int mhsa_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, void* osave, hipStream_t st, const SeqTail* tl);
int mhsa_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, int layer, const void* x, const void* dy, const void* osave,
                  void* dx, void* ws, hipStream_t st, const Side* sd);
int tconvffn_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, void* tsave, hipStream_t st, const SeqTail* tl);
int gb_tconvffn_fwd(const nbss_cfg& c, const float* P, int layer, const void* x, void* y, void* ws, hipStream_t st);
size_t tconvffn_save_bytes(const nbss_cfg& c);

int tconvffn_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, int layer, const void* x, const void* dy, const void* tsave,
                      void* dx, void* ws, hipStream_t st, const Side* sd);

int fconv_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, int layer, int which, const void* x, const void* dy, void* dx,
                   void* ws, hipStream_t st, const Side* sd);
int full_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, int layer, const void* x, const void* dy, void* dx, void* ws,
                  hipStream_t st, const Side* sd);
int decoder_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, const void* x, const float* dout, void* dx, void* ws,
                     hipStream_t st);
int encoder_bwd_impl(const nbss_cfg& c, float* G, const void* xin, const void* dy, void* ws, hipStream_t st);

size_t stft_tables_bytes_impl(int nfft);
int stft_tables_impl(int nfft, int win_kind, float* tab, hipStream_t st);
int stft_norm_impl(int nfft, int dtype, int B, int C, int N, int ref, const float* tab, const float* x, void* X, float* xrmm, hipStream_t st);
int inorm_istft_impl(int nfft, int B, int S, int N, const float* tab, const float* out, const float* xrmm, float* ybuf, float* y, hipStream_t st);
int inorm_istft_bwd_impl(int nfft, int B, int S, int N, const float* tab, const float* dy, const float* xrmm, float* dout, hipStream_t st);
size_t pit_ws_floats(int B, int S);
int pit_sisdr_impl(int B, int S, int N, const float* p, const float* t, float* loss, int* perm, float* dp, float* ws, hipStream_t st);
int clip_adam_dev_impl(size_t n, float* p, float* g, float* m, float* v, float* scal, const float* hyper, float max_norm, float grad_scale, float beta1,
                       float beta2, float eps, float wd, int flags, hipStream_t st);
int adam_hyper_impl(int step, float lr, float beta1, float beta2, float* out);
int clip_adam_impl(size_t n, float* p, float* g, float* m, float* v, float* scal, float max_norm, float grad_scale, float lr, float beta1,
                   float beta2, float eps, float wd, int step, int flags, hipStream_t st);

size_t nb_ws_bytes_impl(int M, int K, int groups, int taps);
int nb_conv_t_impl(int dtype, long nseq, int Tn, int Cin, int ldx, int Cout, int groups, int taps, const void* x, const float* w, const float* bias, void* y,
                   const void* residual, int act_in, int act_out, void* ws, hipStream_t st);
int nb_layernorm_impl(int dtype, long rows, int C, const void* x, const float* gamma, const float* beta, void* y, float* stats, hipStream_t st);
int nb_gbn_impl(int dtype, int B, int F, int Tn, int C, const void* x, const float* gamma, const float* beta, float eps, int act, void* y, hipStream_t st);
int nb_attention_fwd_impl(int dtype, long nseq, int Tn, int H, int heads, const void* qkv, void* o, hipStream_t st);
int nb_attention_relpos_fwd_impl(int dtype, long nseq, int Tn, int H, int heads, const void* qkv, const void* pos, const float* ub, const float* vb, float scale, void* o,
                                 hipStream_t st, const uint32_t* mask, float keep);
size_t nb_relpos_bwd_ws_bytes_impl(long nseq, int Tn, int H, int heads);
int nb_attention_relpos_bwd_impl(int dtype, long nseq, int Tn, int H, int heads, const void* qkv, const void* pos, const float* ub, const float* vb, float scale,
                                 const uint32_t* mask, float keep, const void* dO, void* dqkv, float* dpos, float* du, float* dvb, void* ws, hipStream_t st);
int nb_group_norm_train_impl(int dtype, long nseq, int Tn, int C, int groups, const void* x, const float* gamma, const float* beta, int act, void* y, float* stats,
                             hipStream_t st);
int nb_group_norm_bwd_impl(int dtype, long nseq, int Tn, int C, int groups, const void* x, const float* stats, const float* gamma, const float* beta, void* dy_dx,
                           float* dgamma, float* dbeta, hipStream_t st);
int nb_group_norm_impl(int dtype, long nseq, int Tn, int C, int groups, const void* x, const float* gamma, const float* beta, int act, void* y, hipStream_t st);
size_t blstm_ws_bytes_impl(int HD, int dtype);
int blstm_fwd_impl(int dtype, long n, int Tn, int HD, int ldg, const void* gx, const float* whh0, const float* whh1, void* y, void* save, void* ws, hipStream_t st);
int blstm_bwd_impl(int dtype, long n, int Tn, int HD, const void* dy, const void* save, const float* whh0, const float* whh1, void* dg, void* ws, hipStream_t st);
size_t nb_bwd_ws_bytes_impl(int M, int K, int groups, int taps);
int nb_conv_t_train_impl(int dtype, long nseq, int Tn, int Cin, int ldx, int Cout, int groups, int taps, const void* x, const float* w, const float* bias, void* y,
                         void* y2, const void* residual, void* ws, hipStream_t st);
int nb_conv_t_bwd_impl(int dtype, long nseq, int Tn, int Cin, int ldx, int Cout, int groups, int taps, const void* x, const float* w, const void* dy, const void* dact,
                       void* dx, float* dw, float* dbias, void* ws, hipStream_t st);
int nb_layernorm_bwd_impl(int dtype, long rows, int C, const void* x, const float* stats, const float* gamma, const void* du, const void* dres, void* dx, float* dgamma,
                          float* dbeta, hipStream_t st);
int nb_gbn_bwd_impl(int dtype, int B, int F, int Tn, int C, const void* x, const float* gamma, const float* beta, float eps, int act, const void* dy, void* dx,
                    float* dgamma, float* dbeta, hipStream_t st);
size_t nb_attn_bwd_ws_bytes_impl(long N, int H, int heads, int dtype);
int nb_attention_bwd_impl(int dtype, long nseq, int Tn, int H, int heads, const void* qkv, const void* dO, void* dqkv, void* ws, hipStream_t st);

#define CHECK_CFG(cfg)                         \
    if (!(cfg)) return NBSS_EINVAL;            \
    {                                          \
        int _e = check_cfg(*(cfg));            \
        if (_e != NBSS_OK) return _e;          \
    }
/* backward (and a forward that saves state for it) keeps a whole sequence per workgroup: T <= NBSS_T_TRAIN_MAX.  The fused training kernels
   are specialised for the SpatialNet-small geometry; every other geometry check_cfg admits (SpatialNet-large) runs the generic backward (gbwd.hip) */
#define CHECK_CFG_TRAIN(cfg) \
    CHECK_CFG(cfg);          \
    if ((cfg)->T > NBSS_T_TRAIN_MAX) return NBSS_EUNSUPPORTED;
#define CHECK_LAYER(cfg, layer) \
    if ((layer) < 0 || (layer) >= (cfg)->L) return NBSS_EINVAL;

extern "C" {

int nbss_param_table(const nbss_cfg* cfg, int64_t* offsets, int64_t* numels, int max_entries) {
    CHECK_CFG(cfg);
    const nbss_cfg& c = *cfg;
    const int n = 2 + NUM_LAYER_PARAMS * c.L + 2;
    if (!offsets || !numels) return n;
    if (max_entries < n) return NBSS_EINVAL;
    int i = 0;
    offsets[i] = param_off_enc_w(c); numels[i++] = enc_w_numel(c);
    offsets[i] = param_off_enc_b(c); numels[i++] = c.H;
    for (int l = 0; l < c.L; ++l)
        for (int p = 0; p < NUM_LAYER_PARAMS; ++p) {
            offsets[i] = param_off(c, l, p);
            numels[i++] = layer_param_numel(c, p);
        }
    offsets[i] = param_off_dec_w(c); numels[i++] = (int64_t)c.C_out * c.H;
    offsets[i] = param_off_dec_b(c); numels[i++] = c.C_out;
    return n;
}

int64_t nbss_param_count(const nbss_cfg* cfg) {
    if (!cfg || check_cfg(*cfg) != NBSS_OK) return -1;
    return param_total(*cfg);
}

int64_t nbss_packed_bytes(const nbss_cfg* cfg) {
    if (!cfg || check_cfg(*cfg) != NBSS_OK) return -1;
    return pack_total(*cfg) * (cfg->dtype == NBSS_BF16 ? 2 : 4);
}

int nbss_pack_params(const nbss_cfg* cfg, const float* params, void* packed, void* stream) {
    CHECK_CFG(cfg);
    if (!params || !packed) return NBSS_EINVAL;
    return pack_params_impl(*cfg, params, packed, (hipStream_t)stream);
}

int nbss_encoder_fwd(const nbss_cfg* cfg, const float* params, const void* packed, const void* xin, void* y, void* stream) {
    CHECK_CFG(cfg);
    if (!params || !packed || !xin || !y) return NBSS_EINVAL;
    return encoder_fwd_impl(*cfg, params, packed, xin, y, (hipStream_t)stream);
}

int nbss_decoder_fwd(const nbss_cfg* cfg, const float* params, const void* packed, const void* x, float* out, void* stream) {
    CHECK_CFG(cfg);
    if (!params || !packed || !x || !out) return NBSS_EINVAL;
    return decoder_fwd_impl(*cfg, params, packed, x, out, (hipStream_t)stream);
}

int nbss_fconv_fwd(const nbss_cfg* cfg, const float* params, const void* packed, int layer, int which, const void* x, void* y, void* stream) {
    CHECK_CFG(cfg);
    CHECK_LAYER(cfg, layer);
    if (!params || !packed || !x || !y || x == y || (which != 0 && which != 1)) return NBSS_EINVAL;
    return fconv_fwd_impl(*cfg, params, packed, layer, which, x, y, (hipStream_t)stream);
}

int nbss_full_fwd(const nbss_cfg* cfg, const float* params, const void* packed, int layer, const void* x, void* y, void* stream) {
    CHECK_CFG(cfg);
    CHECK_LAYER(cfg, layer);
    if (!params || !packed || !x || !y || x == y) return NBSS_EINVAL;
    return full_fwd_impl(*cfg, params, packed, layer, x, y, (hipStream_t)stream);
}

int64_t nbss_mhsa_save_bytes(const nbss_cfg* cfg) {
    if (!cfg || check_cfg(*cfg) != NBSS_OK) return -1;
    return (int64_t)mhsa_save_bytes(*cfg);
}

int nbss_mhsa_fwd(const nbss_cfg* cfg, const float* params, const void* packed, int layer, const void* x, void* y, void* o_save, void* stream) {
    CHECK_CFG(cfg);
    CHECK_LAYER(cfg, layer);
    if (!params || !packed || !x || !y || x == y) return NBSS_EINVAL;
    return mhsa_fwd_impl(*cfg, params, packed, layer, x, y, o_save, (hipStream_t)stream, nullptr);
}

int nbss_mhsa_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, int layer, const void* x, const void* dy,
                  const void* o_save, void* dx, void* ws, void* stream) {
    CHECK_CFG_TRAIN(cfg);
    CHECK_LAYER(cfg, layer);
    if (!params || !grads || !packed || !x || !dy || !o_save || !dx || !ws) return NBSS_EINVAL;
    return mhsa_bwd_impl(*cfg, params, grads, packed, layer, x, dy, o_save, dx, ws, (hipStream_t)stream, nullptr);
}

int64_t nbss_tconvffn_save_bytes(const nbss_cfg* cfg) {
    if (!cfg || check_cfg(*cfg) != NBSS_OK) return -1;
    return (int64_t)tconvffn_save_bytes(*cfg);
}

int nbss_tconvffn_fwd(const nbss_cfg* cfg, const float* params, const void* packed, int layer, const void* x, void* y, void* t_save, void* stream) {
    CHECK_CFG(cfg);
    CHECK_LAYER(cfg, layer);
    if (!params || !packed || !x || !y || x == y) return NBSS_EINVAL;
    if (t_save && tconvffn_save_bytes(*cfg) == 0) return NBSS_EUNSUPPORTED;
    return tconvffn_fwd_impl(*cfg, params, packed, layer, x, y, t_save, (hipStream_t)stream, nullptr);
}

int64_t nbss_workspace_bytes(const nbss_cfg* cfg) {
    if (!cfg || check_cfg(*cfg) != NBSS_OK) return -1;
    return (int64_t)workspace_bytes(*cfg);
}

int nbss_tconvffn_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, int layer, const void* x, const void* dy,
                      const void* t_save, void* dx, void* ws, void* stream) {
    CHECK_CFG_TRAIN(cfg);
    CHECK_LAYER(cfg, layer);
    if (!params || !grads || !packed || !x || !dy || !dx || !ws) return NBSS_EINVAL;
    if (t_save && tconvffn_save_bytes(*cfg) == 0) return NBSS_EUNSUPPORTED;
    return tconvffn_bwd_impl(*cfg, params, grads, packed, layer, x, dy, t_save, dx, ws, (hipStream_t)stream, nullptr);
}

int nbss_fconv_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, int layer, int which, const void* x,
                   const void* dy, void* dx, void* ws, void* stream) {
    CHECK_CFG_TRAIN(cfg);
    CHECK_LAYER(cfg, layer);
    if (!params || !grads || !packed || !x || !dy || !dx || !ws || (which != 0 && which != 1)) return NBSS_EINVAL;
    return fconv_bwd_impl(*cfg, params, grads, packed, layer, which, x, dy, dx, ws, (hipStream_t)stream, nullptr);
}

int nbss_full_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, int layer, const void* x, const void* dy,
                  void* dx, void* ws, void* stream) {
    CHECK_CFG_TRAIN(cfg);
    CHECK_LAYER(cfg, layer);
    if (!params || !grads || !packed || !x || !dy || !dx || !ws) return NBSS_EINVAL;
    return full_bwd_impl(*cfg, params, grads, packed, layer, x, dy, dx, ws, (hipStream_t)stream, nullptr);
}

int nbss_decoder_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, const void* x, const float* dout, void* dx,
                     void* ws, void* stream) {
    CHECK_CFG_TRAIN(cfg);
    if (!params || !grads || !packed || !x || !dout || !dx || !ws) return NBSS_EINVAL;
    return decoder_bwd_impl(*cfg, params, grads, packed, x, dout, dx, ws, (hipStream_t)stream);
}

int nbss_encoder_bwd(const nbss_cfg* cfg, float* grads, const void* xin, const void* dy, void* stream) {
    CHECK_CFG_TRAIN(cfg);
    if (!grads || !xin || !dy) return NBSS_EINVAL;
    return encoder_bwd_impl(*cfg, grads, xin, dy, nullptr, (hipStream_t)stream);
}

// ---- whole network: native sequencing of the sub-block kernels (one C call per direction) --------
#define BWD_KINDS 5  // sub-blocks of a layer in backward order: T-ConvFFN, attention, F-conv 2, full, F-conv 1
static size_t stream_bytes(const nbss_cfg& c) {
    return ws_align((size_t)c.B * c.F * c.T * c.H * (c.dtype == NBSS_BF16 ? 2 : 4));
}

// saved state of the whole network: [5L+1 block inputs | L attention saves | L T-ConvFFN saves (bf16 stream; NBSS_TCF_RECOMPUTE flavour: none)]
static size_t tcf_save_bytes(const nbss_cfg& c) {
#ifdef NBSS_TCF_RECOMPUTE
    return 0;
#else
    return ws_align(tconvffn_save_bytes(c));
#endif
}
static size_t acts_tcf_offset(const nbss_cfg& c) { return (size_t)(5 * c.L + 1) * stream_bytes(c) + (size_t)c.L * mhsa_save_bytes(c); }

int64_t nbss_acts_bytes(const nbss_cfg* cfg) {
    if (!cfg || check_cfg(*cfg) != NBSS_OK) return -1;
    return (int64_t)(acts_tcf_offset(*cfg) + (size_t)cfg->L * tcf_save_bytes(*cfg));
}

int64_t nbss_train_ws_bytes(const nbss_cfg* cfg) {
    if (!cfg || check_cfg(*cfg) != NBSS_OK) return -1;
    // the backward walk: one workspace copy per sub-block kind + three rotating gradient buffers (see nbss_spatialnet_bwd_range);
    // the forward walk uses the first copy and the two buffers behind it
    return (int64_t)(BWD_KINDS * workspace_bytes(*cfg) + 3 * stream_bytes(*cfg));
}

// ---- the second stream of the walks (side.h): parameter-gradient launches in backward, the row kernels' tail launches in forward ------------
// Owned by the library, created at the first backward walk.  done[k]: recorded on the gradient stream behind sub-block kind k's parameter-gradient
// launches; the main stream waits for it before kind k's workspace copy (next layer) or a gradient buffer those launches read is written again.
#ifndef NBSS_EMU
struct SideState {
    int state = 0;  // 0 = not tried, 1 = ready, -1 = unavailable / switched off (NBSS_SIDE_STREAM=0): in order
    int device = -1;
    Side sd;
    hipStream_t gs_low;  // the gradient stream of small grids
    hipEvent_t done[BWD_KINDS], join;
    int ncu = 256;
};
static SideState g_side;
// `st` = the caller's stream: while it is being captured into a HIP graph the walks stay in order on it (a captured fork / join per sub-block
// turns the graph into ~100 branches, which ROCm 7.2 replays at ~45 us per kernel node: batch 2 went from 5.7 ms eager to 13.6 ms replayed;
// the linear chain replays at the ~1.5 us boundary cost)
static SideState* side_state(hipStream_t st = nullptr) {
    if (st) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return nullptr;
    }
    SideState& s = g_side;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (s.state == 1 && s.device != dev) s.state = 0;  // (one process per GPU: does not happen; a second device gets its own objects)
    if (s.state == 0) {
        s.state = -1;
        s.device = dev;
        const char* env = getenv("NBSS_SIDE_STREAM");
        if (env && env[0] == '0') return nullptr;
        // two gradient streams: default priority, and the LOWEST one for small grids (below 8 rounds of row-kernel workgroups) — measured on one
        // box, default vs lowest: batch 2 342 -> 346 utt/s, batch 8 528 -> 534, batch 32 628 -> 617 (there the delayed folds end up behind the join)
        int lo = 0, hi = 0;
        const bool prio = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi;
        bool ok = hipStreamCreateWithFlags(&s.sd.gs, hipStreamNonBlocking) == hipSuccess;
        ok = ok && (prio ? hipStreamCreateWithPriority(&s.gs_low, hipStreamNonBlocking, lo) : hipStreamCreateWithFlags(&s.gs_low, hipStreamNonBlocking)) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.sd.ready, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.join, hipEventDisableTiming) == hipSuccess;
        for (int k = 0; ok && k < BWD_KINDS; ++k) ok = hipEventCreateWithFlags(&s.done[k], hipEventDisableTiming) == hipSuccess;
        hipDeviceProp_t prop;
        if (ok && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) s.ncu = prop.multiProcessorCount;
        if (ok) s.state = 1;
    }
    return s.state == 1 ? &s : nullptr;
}
#endif

int nbss_spatialnet_fwd(const nbss_cfg* cfg, const float* params, const void* packed, const void* xin, void* acts, void* ws, float* out,
                        void* stream) {
    CHECK_CFG(cfg);
    WalkFlipScope flip_scope(stream_bytes(*cfg));
    if (!params || !packed || !xin || !out || (!acts && !ws)) return NBSS_EINVAL;
    const nbss_cfg& c = *cfg;
    if (acts && c.T > NBSS_T_TRAIN_MAX) return NBSS_EUNSUPPORTED;  // long sequences: inference only
    hipStream_t st = (hipStream_t)stream;
    const size_t sb = stream_bytes(c);
    // training: every block input is kept (acts = [5L+1 stream copies | L attention save buffers]);
    // inference: two ping-pong buffers at the tail of ws
    char* pp = acts ? nullptr : (char*)ws + workspace_bytes(c);
    int k = 0;
    auto buf = [&](int i) -> void* { return acts ? (void*)((char*)acts + (size_t)i * sb) : (void*)(pp + (size_t)(i & 1) * sb); };
    int e = encoder_fwd_impl(c, params, packed, xin, buf(0), st);
    if (e) return e;
#ifndef NBSS_TAIL_ROUNDS
#define NBSS_TAIL_ROUNDS 40  // tail launches for grids below this many rounds (rounds 4-5: 8; round 6, same call: batch 16 — 8 rounds + 16 sequences —
                             // 701 / 705 -> 715 / 716 utt/s with its tail launches, batch 32 739 / 731 -> 739 / 739)
#endif
    // tail launches of the bf16 row kernels (side.h: SeqTail): when the last round of sequences is at most half full
    SeqTail tl = {0, st};
#ifndef NBSS_EMU
    SideState* ss = side_state(st);
    const int ncu = ss ? ss->ncu : 0;
    if (ss) tl.ts = ss->sd.gs;
#else
    const int ncu = 4;  // (the emulator walks tiny grids through the same offset logic, in order)
#endif
    {
        const int nseq = c.B * c.F, rem = ncu > 0 ? nseq % ncu : 0;
        // measured (same box, NBSS_SEQ_TAIL=0 / 1): batch 2 (1 round + 2 sequences) 297 -> 302 utt/s, batch 8 (4 + 8) 512 -> 516, batch 32 (16 + 32)
        // 624 -> 623: the tail launch pays while the rounds are few
        static const bool off = [] { const char* v = getenv("NBSS_SEQ_TAIL"); return v && v[0] == '0'; }();  // A/B knob
        if (!off && c.dtype == NBSS_BF16 && c.H == 96 && c.T <= NBSS_T_TRAIN_MAX && nseq > ncu && nseq < NBSS_TAIL_ROUNDS * ncu && rem > 0 && 2 * rem <= ncu) tl.n = rem;
    }
    const SeqTail* tlp = tl.n > 0 ? &tl : nullptr;
    for (int l = 0; l < c.L; ++l) {
        // (inference beyond 256 frames: the head of ws, idle without a backward pass, is the attention's K | V scratch)
        void* osave = acts ? (void*)((char*)acts + (size_t)(5 * c.L + 1) * sb + (size_t)l * mhsa_save_bytes(c)) : (c.T > NBSS_T_TRAIN_MAX || c.H != 96) ? ws : nullptr;
        if ((e = fconv_fwd_impl(c, params, packed, l, 0, buf(k), buf(k + 1), st))) return e;
        if ((e = full_fwd_impl(c, params, packed, l, buf(k + 1), buf(k + 2), st))) return e;
        if ((e = fconv_fwd_impl(c, params, packed, l, 1, buf(k + 2), buf(k + 3), st))) return e;
#ifndef NBSS_EMU
        if (tlp && (hipEventRecord(ss->sd.ready, st) != hipSuccess || hipStreamWaitEvent(tl.ts, ss->sd.ready, 0) != hipSuccess)) return NBSS_ELAUNCH;
#endif
        if ((e = mhsa_fwd_impl(c, params, packed, l, buf(k + 3), buf(k + 4), osave, st, tlp))) return e;
        void* tsave = acts && tcf_save_bytes(c) ? (void*)((char*)acts + acts_tcf_offset(c) + (size_t)l * tcf_save_bytes(c)) : nullptr;
        // SpatialNet-large, bf16, T <= 256: the idle workspace carries the block's intermediates (gbwd.hip: gb_tconvffn_fwd)
        e = c.H != 96 && ws ? gb_tconvffn_fwd(c, params, l, buf(k + 4), buf(k + 5), ws, st) : NBSS_EUNSUPPORTED;
        if (e == NBSS_EUNSUPPORTED) e = tconvffn_fwd_impl(c, params, packed, l, buf(k + 4), buf(k + 5), tsave, st, tlp);
        if (e) return e;
#ifndef NBSS_EMU
        if (tlp && (hipEventRecord(ss->join, tl.ts) != hipSuccess || hipStreamWaitEvent(st, ss->join, 0) != hipSuccess)) return NBSS_ELAUNCH;
#endif
        k += 5;
    }
    return decoder_fwd_impl(c, params, packed, buf(k), out, st);
}

int nbss_spatialnet_bwd_range(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, const void* xin, const void* acts,
                              const float* dout, void* ws, int layer_hi, int layer_lo, void* stream) {
    CHECK_CFG_TRAIN(cfg);
    if (!params || !grads || !packed || !xin || !acts || !ws) return NBSS_EINVAL;
    const nbss_cfg& c = *cfg;
    if (layer_lo < 0 || layer_hi > c.L || layer_lo >= layer_hi) return NBSS_EINVAL;
    if (layer_hi == c.L && !dout) return NBSS_EINVAL;
    WalkFlipScope flip_scope(stream_bytes(*cfg));
    hipStream_t st = (hipStream_t)stream;
    const size_t sb = stream_bytes(c), wb = workspace_bytes(c);
    auto act = [&](int i) -> const void* { return (const char*)acts + (size_t)i * sb; };
    // ws = [BWD_KINDS workspace copies | 3 gradient buffers].  The gradient stream walks from buffer to buffer (sub-block j of the walk reads
    // buffer j % 3 and writes (j + 1) % 3; the decoder wrote buffer 0), so that a sub-block's upstream gradient — an operand of its out_proj / W2
    // weight-gradient problem on the gradient stream — is not overwritten by the NEXT sub-block (two buffers) but by the one after it
    char* gb = (char*)ws + BWD_KINDS * wb;
    auto gbuf = [&](int j) -> void* { return gb + (size_t)(j % 3) * sb; };
    auto wsk = [&](int kind) -> void* { return (char*)ws + (size_t)kind * wb; };
    int j = BWD_KINDS * (c.L - layer_hi);  // sub-blocks already walked
    int k = 5 * layer_hi;
    int e;
#ifndef NBSS_EMU
    SideState* ss = side_state(st);
    Side side;
    if (ss) {
        side = ss->sd;
#ifndef NBSS_GSLOW_ROUNDS
#define NBSS_GSLOW_ROUNDS 0  // the lowest-priority gradient stream for grids below this many rounds of row-kernel workgroups: none any more (rounds 4-5: 8 — then
                            // +1 % at batch 2 / 8; round 6, same call, 8 -> 0: batch 2 373.5 / 375.6 -> 378.3 / 378.0 utt/s, batch 8 630.6 / 629.8 -> 632.6 / 634.4;
                            // 8 -> 40: batch 16 and 32 unchanged)
#endif
        if (c.B * c.F < NBSS_GSLOW_ROUNDS * ss->ncu) side.gs = ss->gs_low;
    }
    const Side* sd = ss ? &side : nullptr;
    bool rec[BWD_KINDS] = {false, false, false, false, false};
    int reader[3] = {-1, -1, -1};  // kind whose gradient-stream launches read buffer b (as their dy)
    // before sub-block `kind` writes buffer `out`: its own workspace copy and that buffer must be free of gradient-stream readers
    // (a failed wait / record would be a silent race on a workspace copy or a rotating gradient buffer: it ends the walk with NBSS_ELAUNCH)
    bool ev_ok = true;
    auto before = [&](int kind, int out) {
        if (!ss) return;
        if (rec[kind]) ev_ok = ev_ok && hipStreamWaitEvent(st, ss->done[kind], 0) == hipSuccess;
        if (reader[out] >= 0 && reader[out] != kind) ev_ok = ev_ok && hipStreamWaitEvent(st, ss->done[reader[out]], 0) == hipSuccess;
        reader[out] = -1;
    };
    auto after = [&](int kind, int in) {
        if (!ss) return;
        ev_ok = ev_ok && hipEventRecord(ss->done[kind], side.gs) == hipSuccess;
        rec[kind] = true;
        if (in >= 0) reader[in] = kind;  // (only the T-ConvFFN's W2 and the attention's out_proj problems contract against the upstream gradient)
    };
#else
    const Side* sd = nullptr;
    const bool ev_ok = true;
    auto before = [&](int, int) {};
    auto after = [&](int, int) {};
#endif
    if (layer_hi == c.L && (e = decoder_bwd_impl(c, params, grads, packed, act(k), dout, gbuf(0), wsk(0), st))) return e;
    for (int l = layer_hi - 1; l >= layer_lo; --l) {
        const void* osave = (const char*)acts + (size_t)(5 * c.L + 1) * sb + (size_t)l * mhsa_save_bytes(c);
        const void* tsave = tcf_save_bytes(c) ? (const void*)((const char*)acts + acts_tcf_offset(c) + (size_t)l * tcf_save_bytes(c)) : nullptr;
        before(0, (j + 1) % 3);
        if ((e = tconvffn_bwd_impl(c, params, grads, packed, l, act(k - 1), gbuf(j), tsave, gbuf(j + 1), wsk(0), st, sd))) return e;
        after(0, j % 3);
        before(1, (j + 2) % 3);
        if ((e = mhsa_bwd_impl(c, params, grads, packed, l, act(k - 2), gbuf(j + 1), osave, gbuf(j + 2), wsk(1), st, sd))) return e;
        after(1, (j + 1) % 3);
        before(2, (j + 3) % 3);
        if ((e = fconv_bwd_impl(c, params, grads, packed, l, 1, act(k - 3), gbuf(j + 2), gbuf(j + 3), wsk(2), st, sd))) return e;
        after(2, -1);
        before(3, (j + 4) % 3);
        if ((e = full_bwd_impl(c, params, grads, packed, l, act(k - 4), gbuf(j + 3), gbuf(j + 4), wsk(3), st, sd))) return e;
        after(3, -1);
        before(4, (j + 5) % 3);
        if ((e = fconv_bwd_impl(c, params, grads, packed, l, 0, act(k - 5), gbuf(j + 4), gbuf(j + 5), wsk(4), st, sd))) return e;
        after(4, -1);
        if (!ev_ok) return NBSS_ELAUNCH;
        j += BWD_KINDS;
        k -= 5;
    }
#ifndef NBSS_EMU
    // join: the caller's next work on `st` (the gradient all-reduce of this range, the optimizer) sees every parameter gradient
    if (ss && (hipEventRecord(ss->join, side.gs) != hipSuccess || hipStreamWaitEvent(st, ss->join, 0) != hipSuccess)) return NBSS_ELAUNCH;
#endif
    return layer_lo == 0 ? encoder_bwd_impl(c, grads, xin, gbuf(j), wsk(0), st) : NBSS_OK;
}

int nbss_spatialnet_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, const void* xin, const void* acts,
                        const float* dout, void* ws, void* stream) {
    if (!cfg || !dout) return NBSS_EINVAL;
    return nbss_spatialnet_bwd_range(cfg, params, grads, packed, xin, acts, dout, ws, cfg->L, 0, stream);
}

int64_t nbss_stft_tables_bytes(int n_fft) {
    if (n_fft != 256 && n_fft != 512) return -1;
    return (int64_t)stft_tables_bytes_impl(n_fft);
}

int nbss_stft_tables(int n_fft, int window, float* tables, void* stream) {
    if (!tables || (window != 0 && window != 1)) return NBSS_EINVAL;
    return stft_tables_impl(n_fft, window, tables, (hipStream_t)stream);
}

int nbss_stft_norm_fwd(int n_fft, int dtype, int B, int C, int N, int ref_channel, const float* tables, const float* x, void* X, float* xrmm,
                       void* stream) {
    if (!tables || !x || !X || !xrmm || B <= 0 || C <= 0 || (dtype != NBSS_F32 && dtype != NBSS_BF16)) return NBSS_EINVAL;
    return stft_norm_impl(n_fft, dtype, B, C, N, ref_channel, tables, x, X, xrmm, (hipStream_t)stream);
}

int64_t nbss_istft_ws_bytes(int n_fft, int B, int S, int N) {
    if (n_fft <= 0 || B <= 0 || S <= 0 || N <= 0) return -1;
    return (int64_t)B * S * ((int64_t)(N / (n_fft / 2) + 2) * (n_fft / 2)) * (int64_t)sizeof(float);
}

int nbss_inorm_istft_fwd(int n_fft, int B, int S, int N, const float* tables, const float* out, const float* xrmm, float* ws, float* y,
                         void* stream) {
    if (!tables || !out || !xrmm || !ws || !y || B <= 0 || S <= 0 || N < n_fft) return NBSS_EINVAL;
    return inorm_istft_impl(n_fft, B, S, N, tables, out, xrmm, ws, y, (hipStream_t)stream);
}

int nbss_inorm_istft_bwd(int n_fft, int B, int S, int N, const float* tables, const float* dy, const float* xrmm, float* dout, void* stream) {
    if (!tables || !dy || !xrmm || !dout || B <= 0 || S <= 0 || N < n_fft) return NBSS_EINVAL;
    return inorm_istft_bwd_impl(n_fft, B, S, N, tables, dy, xrmm, dout, (hipStream_t)stream);
}

int64_t nbss_pit_ws_bytes(int B, int S) {
    if (B <= 0 || S <= 0) return -1;
    return (int64_t)pit_ws_floats(B, S) * (int64_t)sizeof(float);
}

int nbss_pit_neg_sisdr(int B, int S, int N, const float* preds, const float* target, float* loss, int32_t* perm, float* dpreds, float* ws,
                       void* stream) {
    if (!preds || !target || !loss || !perm || !ws || N <= 0) return NBSS_EINVAL;
    return pit_sisdr_impl(B, S, N, preds, target, loss, perm, dpreds, ws, (hipStream_t)stream);
}

int nbss_clip_adam_step(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* scratch, float max_norm,
                        float grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay, int step, int flags,
                        void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !scratch || n <= 0) return NBSS_EINVAL;
    return clip_adam_impl((size_t)n, params, grads, exp_avg, exp_avg_sq, scratch, max_norm, grad_scale, lr, beta1, beta2, eps, weight_decay, step,
                          flags, (hipStream_t)stream);
}

int nbss_clip_adam_step_dev(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* scratch, const float* hyper, float max_norm,
                            float grad_scale, float beta1, float beta2, float eps, float weight_decay, int flags, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !scratch || !hyper || n <= 0) return NBSS_EINVAL;
    return clip_adam_dev_impl((size_t)n, params, grads, exp_avg, exp_avg_sq, scratch, hyper, max_norm, grad_scale, beta1, beta2, eps, weight_decay, flags,
                              (hipStream_t)stream);
}

int nbss_adam_hyper(int step, float lr, float beta1, float beta2, float* hyper_host) { return adam_hyper_impl(step, lr, beta1, beta2, hyper_host); }

static bool nb_dtype_ok(int dtype) { return dtype == NBSS_F32 || dtype == NBSS_BF16; }
int64_t nbss_nb_ws_bytes(int Cout, int Cin, int groups, int taps) {
    if (Cout <= 0 || Cin <= 0 || groups <= 0 || taps <= 0 || Cout % groups || Cin % groups) return -1;
    return (int64_t)nb_ws_bytes_impl(Cout, Cin, groups, taps);
}
int nbss_nb_conv_t(int dtype, int64_t nseq, int T, int Cin, int ldx, int Cout, int groups, int taps, const void* x, const float* w, const float* bias, void* y,
                   const void* residual, int act_in, int act_out, void* ws, void* stream) {
    if (!nb_dtype_ok(dtype) || nseq <= 0 || T <= 0 || Cin <= 0 || Cout <= 0 || taps <= 0 || !(taps & 1) || !x || !w || !y || !ws || x == y) return NBSS_EINVAL;
    if (nseq * T >= ((int64_t)1 << 31)) return NBSS_EUNSUPPORTED;
    return nb_conv_t_impl(dtype, (long)nseq, T, Cin, ldx, Cout, groups, taps, x, w, bias, y, residual, act_in, act_out, ws, (hipStream_t)stream);
}
int nbss_nb_layernorm(int dtype, int64_t rows, int C, const void* x, const float* gamma, const float* beta, void* y, float* stats, void* stream) {
    if (!nb_dtype_ok(dtype) || rows <= 0 || C <= 0 || !x || !gamma || !beta || !y || !stats) return NBSS_EINVAL;
    return nb_layernorm_impl(dtype, (long)rows, C, x, gamma, beta, y, stats, (hipStream_t)stream);
}
int nbss_nb_group_batch_norm(int dtype, int B, int F, int T, int C, const void* x, const float* gamma, const float* beta, float eps, int act_out, void* y,
                             void* stream) {
    if (!nb_dtype_ok(dtype) || B <= 0 || F <= 0 || T <= 0 || C <= 0 || !x || !y || (!gamma) != (!beta)) return NBSS_EINVAL;
    return nb_gbn_impl(dtype, B, F, T, C, x, gamma, beta, eps, act_out, y, (hipStream_t)stream);
}
int nbss_nb_attention_fwd(int dtype, int64_t nseq, int T, int H, int heads, const void* qkv, void* o, void* stream) {
    if (!nb_dtype_ok(dtype) || nseq <= 0 || nseq >= 65536 * 32768LL || T <= 0 || H <= 0 || !qkv || !o) return NBSS_EINVAL;
    return nb_attention_fwd_impl(dtype, (long)nseq, T, H, heads, qkv, o, (hipStream_t)stream);
}
int nbss_nb_attention_relpos_fwd(int dtype, int64_t nseq, int T, int H, int heads, const void* qkv, const void* pos, const float* u_bias, const float* v_bias, float scale,
                                 void* o, void* stream) {
    if (!nb_dtype_ok(dtype) || nseq <= 0 || nseq >= 65536 * 32768LL || T <= 0 || H <= 0 || !qkv || !pos || !u_bias || !v_bias || !o) return NBSS_EINVAL;
    return nb_attention_relpos_fwd_impl(dtype, (long)nseq, T, H, heads, qkv, pos, u_bias, v_bias, scale, o, (hipStream_t)stream, nullptr, 1.0f);
}
int nbss_nb_group_norm(int dtype, int64_t nseq, int T, int C, int groups, const void* x, const float* gamma, const float* beta, int act_out, void* y, void* stream) {
    if (!nb_dtype_ok(dtype) || nseq <= 0 || nseq * (int64_t)(groups > 0 ? groups : 1) >= (1LL << 31) || T <= 0 || C <= 0 || !x || !y || !gamma || !beta) return NBSS_EINVAL;
    return nb_group_norm_impl(dtype, (long)nseq, T, C, groups, x, gamma, beta, act_out, y, (hipStream_t)stream);
}

int64_t nbss_nb_bwd_ws_bytes(int Cout, int Cin, int groups, int taps) {
    if (Cout <= 0 || Cin <= 0 || groups <= 0 || taps <= 0 || Cout % groups || Cin % groups) return -1;
    return (int64_t)nb_bwd_ws_bytes_impl(Cout, Cin, groups, taps);
}
int nbss_nb_conv_t_train(int dtype, int64_t nseq, int T, int Cin, int ldx, int Cout, int groups, int taps, const void* x, const float* w, const float* bias, void* y,
                         void* y_silu, const void* residual, void* ws, void* stream) {
    if (!nb_dtype_ok(dtype) || nseq <= 0 || T <= 0 || Cin <= 0 || Cout <= 0 || taps <= 0 || !(taps & 1) || !x || !w || !y || !ws || x == y) return NBSS_EINVAL;
    if (nseq * T >= ((int64_t)1 << 31)) return NBSS_EUNSUPPORTED;
    return nb_conv_t_train_impl(dtype, (long)nseq, T, Cin, ldx, Cout, groups, taps, x, w, bias, y, y_silu, residual, ws, (hipStream_t)stream);
}
int nbss_nb_conv_t_bwd(int dtype, int64_t nseq, int T, int Cin, int ldx, int Cout, int groups, int taps, const void* x, const float* w, const void* dy,
                       const void* x_pre, void* dx, float* dw, float* dbias, void* ws, void* stream) {
    if (!nb_dtype_ok(dtype) || nseq <= 0 || T <= 0 || Cin <= 0 || Cout <= 0 || taps <= 0 || !(taps & 1) || !x || !w || !dy || !ws || (!dx && !dw) || dx == dy)
        return NBSS_EINVAL;
    if (nseq * T >= ((int64_t)1 << 31)) return NBSS_EUNSUPPORTED;
    return nb_conv_t_bwd_impl(dtype, (long)nseq, T, Cin, ldx, Cout, groups, taps, x, w, dy, x_pre, dx, dw, dbias, ws, (hipStream_t)stream);
}
int nbss_nb_layernorm_bwd(int dtype, int64_t rows, int C, const void* x, const float* stats, const float* gamma, const void* dy, const void* dres, void* dx,
                          float* dgamma, float* dbeta, void* stream) {
    if (!nb_dtype_ok(dtype) || rows <= 0 || C <= 0 || !x || !stats || !gamma || !dy || !dres || !dx || !dgamma || !dbeta) return NBSS_EINVAL;
    return nb_layernorm_bwd_impl(dtype, (long)rows, C, x, stats, gamma, dy, dres, dx, dgamma, dbeta, (hipStream_t)stream);
}
int nbss_nb_group_batch_norm_bwd(int dtype, int B, int F, int T, int C, const void* x, const float* gamma, const float* beta, float eps, int act_out, const void* dy,
                                 void* dx, float* dgamma, float* dbeta, void* stream) {
    if (!nb_dtype_ok(dtype) || B <= 0 || F <= 0 || T <= 0 || C <= 0 || !x || !dy || !dx || (!gamma) != (!beta) || (!dgamma) != (!dbeta) || (gamma && !dgamma)) return NBSS_EINVAL;
    return nb_gbn_bwd_impl(dtype, B, F, T, C, x, gamma, beta, eps, act_out, dy, dx, dgamma, dbeta, (hipStream_t)stream);
}
int64_t nbss_nb_attention_bwd_ws_bytes(int dtype, int64_t nseq, int T, int H, int heads) {
    if (!nb_dtype_ok(dtype) || nseq <= 0 || T <= 0 || H <= 0 || heads <= 0) return -1;
    return (int64_t)nb_attn_bwd_ws_bytes_impl((long)nseq * T, H, heads, dtype);
}
int nbss_nb_attention_bwd(int dtype, int64_t nseq, int T, int H, int heads, const void* qkv, const void* d_o, void* dqkv, void* ws, void* stream) {
    if (!nb_dtype_ok(dtype) || nseq <= 0 || nseq >= 65536 * 32768LL || T <= 0 || H <= 0 || !qkv || !d_o || !dqkv || !ws) return NBSS_EINVAL;
    return nb_attention_bwd_impl(dtype, (long)nseq, T, H, heads, qkv, d_o, dqkv, ws, (hipStream_t)stream);
}
int nbss_nb_attention_relpos_train(int dtype, int64_t nseq, int T, int H, int heads, const void* qkv, const void* pos, const float* u_bias, const float* v_bias,
                                   float scale, const uint32_t* keep_bits, float keep_scale, void* o, void* stream) {
    if (!nb_dtype_ok(dtype) || nseq <= 0 || nseq >= 65536 * 32768LL || T <= 0 || H <= 0 || !qkv || !pos || !u_bias || !v_bias || !o) return NBSS_EINVAL;
    return nb_attention_relpos_fwd_impl(dtype, (long)nseq, T, H, heads, qkv, pos, u_bias, v_bias, scale, o, (hipStream_t)stream, keep_bits, keep_bits ? keep_scale : 1.0f);
}
int64_t nbss_nb_attention_relpos_bwd_ws_bytes(int64_t nseq, int T, int H, int heads) {
    if (nseq <= 0 || T <= 0 || H <= 0 || heads <= 0 || H % heads) return -1;
    return (int64_t)nb_relpos_bwd_ws_bytes_impl((long)nseq, T, H, heads);
}
int nbss_nb_attention_relpos_bwd(int dtype, int64_t nseq, int T, int H, int heads, const void* qkv, const void* pos, const float* u_bias, const float* v_bias, float scale,
                                 const uint32_t* keep_bits, float keep_scale, const void* d_o, void* dqkv, float* dpos, float* du_bias, float* dv_bias, void* ws,
                                 void* stream) {
    if (!nb_dtype_ok(dtype) || nseq <= 0 || nseq >= 65536 * 32768LL || T <= 0 || H <= 0 || !qkv || !pos || !u_bias || !v_bias || !d_o || !dqkv || !dpos || !du_bias ||
        !dv_bias || !ws)
        return NBSS_EINVAL;
    return nb_attention_relpos_bwd_impl(dtype, (long)nseq, T, H, heads, qkv, pos, u_bias, v_bias, scale, keep_bits, keep_bits ? keep_scale : 1.0f, d_o, dqkv, dpos, du_bias,
                                        dv_bias, ws, (hipStream_t)stream);
}
int nbss_nb_group_norm_train(int dtype, int64_t nseq, int T, int C, int groups, const void* x, const float* gamma, const float* beta, int act_out, void* y, float* stats,
                             void* stream) {
    if (!nb_dtype_ok(dtype) || nseq <= 0 || nseq * (int64_t)(groups > 0 ? groups : 1) >= (1LL << 31) || T <= 0 || C <= 0 || !x || !y || !gamma || !beta || !stats)
        return NBSS_EINVAL;
    return nb_group_norm_train_impl(dtype, (long)nseq, T, C, groups, x, gamma, beta, act_out, y, stats, (hipStream_t)stream);
}
int nbss_nb_group_norm_bwd(int dtype, int64_t nseq, int T, int C, int groups, const void* x, const float* stats, const float* gamma, const float* beta, void* dy_dx,
                           float* dgamma, float* dbeta, void* stream) {
    if (!nb_dtype_ok(dtype) || nseq <= 0 || nseq * (int64_t)(groups > 0 ? groups : 1) >= (1LL << 31) || T <= 0 || C <= 0 || !x || !stats || !gamma || !beta || !dy_dx ||
        !dgamma || !dbeta)
        return NBSS_EINVAL;
    return nb_group_norm_bwd_impl(dtype, (long)nseq, T, C, groups, x, stats, gamma, beta, dy_dx, dgamma, dbeta, (hipStream_t)stream);
}
int64_t nbss_nb_blstm_ws_bytes(int dtype, int hidden) {
    if (!nb_dtype_ok(dtype) || (hidden != 128 && hidden != 256)) return -1;
    return (int64_t)blstm_ws_bytes_impl(hidden, dtype);
}
int nbss_nb_blstm_fwd(int dtype, int64_t nseq, int T, int hidden, int ldg, const void* gx, const float* w_hh, const float* w_hh_reverse, void* y, void* save, void* ws,
                      void* stream) {
    if (!nb_dtype_ok(dtype) || nseq <= 0 || nseq >= ((int64_t)1 << 31) || T <= 0 || ldg < 8 * hidden || !gx || !w_hh || !w_hh_reverse || !y || !ws) return NBSS_EINVAL;
    return blstm_fwd_impl(dtype, (long)nseq, T, hidden, ldg, gx, w_hh, w_hh_reverse, y, save, ws, (hipStream_t)stream);
}
int nbss_nb_blstm_bwd(int dtype, int64_t nseq, int T, int hidden, const void* dy, const void* save, const float* w_hh, const float* w_hh_reverse, void* dg, void* ws,
                      void* stream) {
    if (!nb_dtype_ok(dtype) || nseq <= 0 || nseq >= ((int64_t)1 << 31) || T <= 0 || !dy || !save || !w_hh || !w_hh_reverse || !dg || !ws) return NBSS_EINVAL;
    return blstm_bwd_impl(dtype, (long)nseq, T, hidden, dy, save, w_hh, w_hh_reverse, dg, ws, (hipStream_t)stream);
}

int nbss_selftest_mma(int dtype, int kperm, const float* A, const float* B, float* D, void* stream) {
    if (!A || !B || !D || (dtype != NBSS_F32 && dtype != NBSS_BF16)) return NBSS_EINVAL;
    return selftest_mma_impl(dtype, kperm, A, B, D, (hipStream_t)stream);
}

const char* nbss_build_info(void) {
#ifdef NBSS_EMU
    return "nbss_amd host-emulator build (tests only)";
#else
    return "nbss_amd gfx950 build";
#endif
}

}  // extern "C"
