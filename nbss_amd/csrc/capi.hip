// capi.hip — the extern "C" surface declared in include/nbss_hip.h.  Thin argument checking
// and dtype dispatch only; kernels live in the sibling .hip files.
#include "launch.h"
#include "layout.h"

int pack_params_impl(const nbss_cfg& c, const float* params, void* packed, hipStream_t stream);
int selftest_mma_impl(int dtype, int kperm, const float* A, const float* B, float* D, hipStream_t stream);
int encoder_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, const void* xin, void* y, hipStream_t st);
int decoder_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, const void* x, float* out, hipStream_t st);
int fconv_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, int which, const void* x, void* y, hipStream_t st);
int full_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st);
int mhsa_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, void* osave, hipStream_t st);
int mhsa_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, int layer, const void* x, const void* dy, const void* osave,
                  void* dx, void* ws, hipStream_t st);
int tconvffn_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st);

int tconvffn_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, int layer, const void* x, const void* dy, void* dx,
                      void* ws, hipStream_t st);

int fconv_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, int layer, int which, const void* x, const void* dy, void* dx,
                   void* ws, hipStream_t st);
int full_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, int layer, const void* x, const void* dy, void* dx, void* ws,
                  hipStream_t st);
int decoder_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, const void* x, const float* dout, void* dx, void* ws,
                     hipStream_t st);
int encoder_bwd_impl(const nbss_cfg& c, float* G, const void* xin, const void* dy, hipStream_t st);

#define CHECK_CFG(cfg)                         \
    if (!(cfg)) return NBSS_EINVAL;            \
    {                                          \
        int _e = check_cfg(*(cfg));            \
        if (_e != NBSS_OK) return _e;          \
    }
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

int nbss_mhsa_fwd(const nbss_cfg* cfg, const float* params, const void* packed, int layer, const void* x, void* y, void* o_save, void* stream) {
    CHECK_CFG(cfg);
    CHECK_LAYER(cfg, layer);
    if (!params || !packed || !x || !y || x == y) return NBSS_EINVAL;
    return mhsa_fwd_impl(*cfg, params, packed, layer, x, y, o_save, (hipStream_t)stream);
}

int nbss_mhsa_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, int layer, const void* x, const void* dy,
                  const void* o_save, void* dx, void* ws, void* stream) {
    CHECK_CFG(cfg);
    CHECK_LAYER(cfg, layer);
    if (!params || !grads || !packed || !x || !dy || !o_save || !dx || !ws) return NBSS_EINVAL;
    return mhsa_bwd_impl(*cfg, params, grads, packed, layer, x, dy, o_save, dx, ws, (hipStream_t)stream);
}

int nbss_tconvffn_fwd(const nbss_cfg* cfg, const float* params, const void* packed, int layer, const void* x, void* y, void* stream) {
    CHECK_CFG(cfg);
    CHECK_LAYER(cfg, layer);
    if (!params || !packed || !x || !y || x == y) return NBSS_EINVAL;
    return tconvffn_fwd_impl(*cfg, params, packed, layer, x, y, (hipStream_t)stream);
}

int64_t nbss_workspace_bytes(const nbss_cfg* cfg) {
    if (!cfg || check_cfg(*cfg) != NBSS_OK) return -1;
    return (int64_t)workspace_bytes(*cfg);
}

int nbss_tconvffn_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, int layer, const void* x, const void* dy,
                      void* dx, void* ws, void* stream) {
    CHECK_CFG(cfg);
    CHECK_LAYER(cfg, layer);
    if (!params || !grads || !packed || !x || !dy || !dx || !ws) return NBSS_EINVAL;
    return tconvffn_bwd_impl(*cfg, params, grads, packed, layer, x, dy, dx, ws, (hipStream_t)stream);
}

int nbss_fconv_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, int layer, int which, const void* x,
                   const void* dy, void* dx, void* ws, void* stream) {
    CHECK_CFG(cfg);
    CHECK_LAYER(cfg, layer);
    if (!params || !grads || !packed || !x || !dy || !dx || !ws || (which != 0 && which != 1)) return NBSS_EINVAL;
    return fconv_bwd_impl(*cfg, params, grads, packed, layer, which, x, dy, dx, ws, (hipStream_t)stream);
}

int nbss_full_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, int layer, const void* x, const void* dy,
                  void* dx, void* ws, void* stream) {
    CHECK_CFG(cfg);
    CHECK_LAYER(cfg, layer);
    if (!params || !grads || !packed || !x || !dy || !dx || !ws) return NBSS_EINVAL;
    return full_bwd_impl(*cfg, params, grads, packed, layer, x, dy, dx, ws, (hipStream_t)stream);
}

int nbss_decoder_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, const void* x, const float* dout, void* dx,
                     void* ws, void* stream) {
    CHECK_CFG(cfg);
    if (!params || !grads || !packed || !x || !dout || !dx || !ws) return NBSS_EINVAL;
    return decoder_bwd_impl(*cfg, params, grads, packed, x, dout, dx, ws, (hipStream_t)stream);
}

int nbss_encoder_bwd(const nbss_cfg* cfg, float* grads, const void* xin, const void* dy, void* stream) {
    CHECK_CFG(cfg);
    if (!grads || !xin || !dy) return NBSS_EINVAL;
    return encoder_bwd_impl(*cfg, grads, xin, dy, (hipStream_t)stream);
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
