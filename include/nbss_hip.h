/* nbss_hip.h — C ABI of the MI355X-native SpatialNet hot path (libnbss_hip.so).
 *
 * The reference (Audio-WestlakeU/NBSS) is pure Python; its "plugin API" for this path is the
 * nn.Module contract `arch.forward([B,F,T,2C]) -> [B,F,T,2*Spk]` (SharedTrainer.py:117-122)
 * plus models/io/{stft,norm,loss}.py.  Every implicit ATen/cuDNN/cuBLAS/cuFFT kernel that
 * contract reaches is replaced by one entry point below; each comment cites the reference
 * site it stands in for.  Conventions:
 *   - plain pointers + sizes only, no torch types; all pointers are DEVICE pointers owned by
 *     the caller (no ownership transfer, no hidden allocation);
 *   - asynchronous on `stream` (a hipStream_t passed as void*); re-entrant, callable from any
 *     host thread (PyTorch's autograd thread calls the *_bwd entry points);
 *   - returns 0 on success, a negative NBSS_E* code otherwise; never throws or exits;
 *   - `dtype` = element type of the [B,F,T,H] residual stream and of the packed weights:
 *        NBSS_F32  fp32 stream, exact-f32 MFMA (v_mfma_f32_16x16x4_f32)
 *        NBSS_BF16 bf16 stream, bf16 MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulate/statistics
 *     master parameters, gradients, STFT/iSTFT, loss and optimizer state are always fp32.
 */
#ifndef NBSS_HIP_H
#define NBSS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBSS_F32 0
#define NBSS_BF16 1

#define NBSS_OK 0
#define NBSS_EINVAL (-1)       /* bad argument */
#define NBSS_EUNSUPPORTED (-2) /* shape/config this build has no kernel for */
#define NBSS_ELAUNCH (-3)      /* HIP launch error */

/* SpatialNet hyper-parameters: models/arch/SpatialNet.py:154-171 (ctor) + batch geometry.
 * Geometries this build has kernels for (anything else: NBSS_EUNSUPPORTED):
 *   small  H 96,  FFN 192, SQ 8,  4 heads   (configs/SpatialNet.yaml:16-24)          forward + backward (fused training kernels)
 *   large  H 192, FFN 384, SQ 16, 4 heads   (the "for large" comments of that file)  forward + backward (geometry-generic backward, csrc/gbwd.hip:
 *                                                                                     one tensor pass per operation, every intermediate in ws)
 *   both with conv groups (8, 8), kernel sizes (5, 3), encoder kernel 5; C_in % 4 == 0, C_in / C_out <= 16.
 *   F <= 272 (n_fft 256 -> 129, n_fft 512 -> 257); fp32-stream backward of the small geometry: F <= 160 (F-conv block).
 *   T <= 256 for a forward that saves state and for every backward; forward without `acts`: T <= 4096. */
typedef struct nbss_cfg {
    int32_t B, F, T;         /* batch, frequencies (129 | 257), frames (251 for 4 s) */
    int32_t C_in, C_out;     /* dim_input (2*channels), dim_output (2*speakers) */
    int32_t H, FFN, SQ;      /* dim_hidden, dim_ffn, dim_squeeze */
    int32_t L, heads;        /* num_layers, num_heads */
    int32_t enc_ks;          /* encoder_kernel_size */
    int32_t f_ks, t_ks;      /* kernel_size = (f, t) */
    int32_t f_groups, t_groups; /* conv_groups = (f, t) */
    int32_t full_share;      /* layers > full_share reuse layer full_share's LinearGroup */
    int32_t dtype;           /* NBSS_F32 | NBSS_BF16 */
} nbss_cfg;

/* ---- parameter geometry ------------------------------------------------------------------
 * Parameters (and their gradients) live in ONE flat fp32 buffer; tensors appear in the
 * reference's state_dict order (SURVEY.md §8(b) checkpoint contract):
 *   encoder.weight, encoder.bias,
 *   per layer l: fconv1.0.{weight,bias} fconv1.1.{weight,bias} fconv1.2.weight
 *                norm_full.{weight,bias} squeeze.0.{weight,bias} full.{weight,bias}
 *                unsqueeze.0.{weight,bias} fconv2.0.* fconv2.1.* fconv2.2.weight
 *                norm_mhsa.{weight,bias} mhsa.in_proj_{weight,bias} mhsa.out_proj.{weight,bias}
 *                tconvffn.{0,1,3,5,6,8,10}.{weight,bias}
 *   decoder.weight, decoder.bias
 * A shared `full` keeps the owner's offset (numel still reported, so offsets may repeat).
 * nbss_param_table fills offsets[i] / numels[i] (in floats) and returns the entry count
 * (2 + 38*L + 2), or a negative error.  Pass NULL arrays to query the count only. */
int nbss_param_table(const nbss_cfg* cfg, int64_t* offsets, int64_t* numels, int max_entries);
int64_t nbss_param_count(const nbss_cfg* cfg);   /* floats in the flat buffer */

/* Packed MFMA weight fragments (+transposes for backward), rebuilt from the fp32 master
 * copy after every optimizer step.  bytes needed / repack. */
int64_t nbss_packed_bytes(const nbss_cfg* cfg);
int nbss_pack_params(const nbss_cfg* cfg, const float* params, void* packed, void* stream);

/* ---- SpatialNet sub-blocks, forward --------------------------------------------------------
 * x/y: residual stream [B,F,T,H] of cfg->dtype.  y may not alias x (x is what backward
 * re-reads).  `layer` selects the parameter slice. */

/* encoder nn.Conv1d(C_in,H,5,'same') along T (SpatialNet.py:175,205).  xin [B,F,T,C_in]. */
int nbss_encoder_fwd(const nbss_cfg* cfg, const float* params, const void* packed, const void* xin, void* y, void* stream);
/* decoder nn.Linear(H,C_out) (SpatialNet.py:200,216).  out [B,F,T,C_out] fp32. */
int nbss_decoder_fwd(const nbss_cfg* cfg, const float* params, const void* packed, const void* x, float* out, void* stream);
/* x + _fconv(fconv1|fconv2): LN -> grouped Conv1d along F -> PReLU (SpatialNet.py:85,87,116-127). which = 0|1 */
int nbss_fconv_fwd(const nbss_cfg* cfg, const float* params, const void* packed, int layer, int which, const void* x, void* y, void* stream);
/* x + _full: LN -> squeeze+SiLU -> LinearGroup over F -> unsqueeze+SiLU (SpatialNet.py:86,129-146). */
int nbss_full_fwd(const nbss_cfg* cfg, const float* params, const void* packed, int layer, const void* x, void* y, void* stream);
/* x + _tsa: LN -> nn.MultiheadAttention over T per (b,f) (SpatialNet.py:88,93-100).
 * o_save (optional, nbss_mhsa_save_bytes(cfg) bytes): what backward needs besides the block input — the attention output
 * before out_proj ([B,F,T,H] of cfg->dtype) followed by the fp32 log2-sum-exp of every (token, head) score row. */
int64_t nbss_mhsa_save_bytes(const nbss_cfg* cfg);
int nbss_mhsa_fwd(const nbss_cfg* cfg, const float* params, const void* packed, int layer, const void* x, void* y, void* o_save, void* stream);
/* x + _tconvffn (SpatialNet.py:90,102-114,61-73).
 * t_save (optional, nbss_tconvffn_save_bytes(cfg) bytes; that is 0 — pass NULL — for the geometries / stream types whose backward
 * recomputes the chain instead): what a training-mode forward keeps for nbss_tconvffn_bwd — the bf16 outputs of the 1x1 conv and of
 * the three grouped T-convs (tconvffn.1 / .3 / .5 / .8 of SpatialNet.py:61-73, group-major), the LayerNorm statistics of every token
 * and the GroupNorm statistics of every (sequence, group). */
int64_t nbss_tconvffn_save_bytes(const nbss_cfg* cfg);
int nbss_tconvffn_fwd(const nbss_cfg* cfg, const float* params, const void* packed, int layer, const void* x, void* y, void* t_save, void* stream);

/* ---- SpatialNet sub-blocks, backward (autograd of the forward entry points) ------------------
 * x: the forward INPUT of the block (the only saved activation; everything else is recomputed
 * on chip), dy: gradient w.r.t. the block output, dx: gradient w.r.t. x (may not alias dy).
 * Parameter gradients are ACCUMULATED (+=) into `grads`, a flat fp32 buffer with the layout of
 * `params`; zero it at the start of a step.  `ws` is caller-provided scratch of at least
 * nbss_workspace_bytes(cfg) bytes (per-token LayerNorm statistics and the operands of the
 * weight-gradient contractions). */
int64_t nbss_workspace_bytes(const nbss_cfg* cfg);
/* t_save: the forward's saved state (see nbss_tconvffn_fwd), or NULL: the forward chain is recomputed from x */
int nbss_tconvffn_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, int layer, const void* x, const void* dy,
                      const void* t_save, void* dx, void* ws, void* stream);
int nbss_mhsa_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, int layer, const void* x, const void* dy,
                  const void* o_save, void* dx, void* ws, void* stream);
int nbss_fconv_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, int layer, int which, const void* x,
                   const void* dy, void* dx, void* ws, void* stream);
int nbss_full_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, int layer, const void* x, const void* dy,
                  void* dx, void* ws, void* stream);
/* decoder: x = decoder input stream, dout = fp32 gradient of the [B,F,T,C_out] output */
int nbss_decoder_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, const void* x, const float* dout, void* dx,
                     void* ws, void* stream);
/* encoder: weight/bias gradient only (the network input needs none); dy = gradient of the encoder output */
int nbss_encoder_bwd(const nbss_cfg* cfg, float* grads, const void* xin, const void* dy, void* stream);

/* ---- whole network (SpatialNet.forward, SpatialNet.py:202-220, and its autograd) ---------------
 * Native sequencing of the sub-block kernels: encoder, L x [fconv1, full, fconv2, mhsa, tconvffn],
 * decoder.  xin [B,F,T,C_in] of cfg->dtype, out/dout [B,F,T,C_out] fp32.
 * acts: nbss_acts_bytes() bytes holding the 5L+1 block inputs, the L attention outputs and the L T-ConvFFN
 * saves that backward re-reads; pass NULL for inference (then ws, >= nbss_workspace_bytes() + two [B,F,T,H] stream tensors, is used
 * as two ping-pong stream buffers behind one workspace).  Backward accumulates into `grads` and needs ws >= nbss_train_ws_bytes():
 * one workspace per sub-block kind of a layer + three gradient stream buffers — the walk launches everything that only produces
 * parameter gradients on a library-owned second stream (fork / done / join events; csrc/side.h), overlapping it with the next
 * sub-blocks' data-gradient kernels, and the copies keep those launches' operands alive.  When a backward call returns, the
 * caller's stream has been made to wait for that second stream: work enqueued on `stream` afterwards (the gradient all-reduce,
 * the optimizer) sees every gradient of the range.  NBSS_SIDE_STREAM=0 in the environment keeps everything on `stream`. */
int64_t nbss_acts_bytes(const nbss_cfg* cfg);
int64_t nbss_train_ws_bytes(const nbss_cfg* cfg);
int nbss_spatialnet_fwd(const nbss_cfg* cfg, const float* params, const void* packed, const void* xin, void* acts, void* ws, float* out,
                        void* stream);
int nbss_spatialnet_bwd(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, const void* xin, const void* acts,
                        const float* dout, void* ws, void* stream);
/* The same walk in pieces, for overlapping the data-parallel gradient exchange with backward (the role of DDP's reverse-order
 * buckets): layers [layer_lo, layer_hi), plus the decoder when layer_hi == L (dout needed only then) and the
 * encoder when layer_lo == 0.  Calls must cover L..0 in descending, adjacent ranges; after a call the gradients of the layers
 * it covered are final, except the LinearGroup shared through full_share, which is final after layer 0. */
/* Threading: the walks share ONE set of library-owned streams and events per process (csrc/capi.hip: SideState) — one walk at a time per
 * process (the one-process-per-GPU model of this library); concurrent walks from several host threads or engines are not supported.  Under
 * HIP-graph capture of `stream` the walks stay in order on it. */
int nbss_spatialnet_bwd_range(const nbss_cfg* cfg, const float* params, float* grads, const void* packed, const void* xin, const void* acts,
                              const float* dout, void* ws, int layer_hi, int layer_lo, void* stream);

/* ---- signal front / back end (always fp32 arithmetic) ------------------------------------------
 * n_fft in {256, 512}, hop = n_fft/2, win_len = n_fft; window 0 = periodic hann, 1 = sqrt-hann
 * (models/io/stft.py:23-35).  `tables` = windowed DFT matrices packed as MFMA fragments, built once
 * per (n_fft, window) into caller memory of nbss_stft_tables_bytes() bytes. */
int64_t nbss_stft_tables_bytes(int n_fft);
int nbss_stft_tables(int n_fft, int window, float* tables, void* stream);
/* STFT.stft (stft.py:49-66) + Norm('frequency', online=True).norm (norm.py:77-81,94) + the
 * [B,C,F,T] complex -> [B,F,T,2C] real re-layout of TrainModule.forward (SharedTrainer.py:113-117).
 * x [B,C,N] fp32 -> X [B,F,T,2C] of `dtype` (T = N/hop + 1), xrmm [B,F,T] fp32 = |X_ref| + 1e-6. */
int nbss_stft_norm_fwd(int n_fft, int dtype, int B, int C, int N, int ref_channel, const float* tables, const float* x, void* X, float* xrmm,
                       void* stream);
/* Norm.inorm (norm.py:97-108) + STFT.istft (stft.py:68-97): out [B,F,T,2S] fp32 -> y [B,S,N].
 * ws: nbss_istft_ws_bytes() bytes of scratch (overlap-add buffer). */
int64_t nbss_istft_ws_bytes(int n_fft, int B, int S, int N);
int nbss_inorm_istft_fwd(int n_fft, int B, int S, int N, const float* tables, const float* out, const float* xrmm, float* ws, float* y,
                         void* stream);
/* adjoint: dy [B,S,N] -> dout [B,F,T,2S] */
int nbss_inorm_istft_bwd(int n_fft, int B, int S, int N, const float* tables, const float* dy, const float* xrmm, float* dout, void* stream);

/* Loss(neg_si_sdr, pit=True).forward (models/io/loss.py:21-29,95-118; torchmetrics si_sdr + pit
 * 'permutation-wise'/'min').  preds/target [B,S,N] fp32.  loss: 1 float (mean over the batch),
 * perm [B,S] (prediction index paired with target s), dpreds (optional) = d loss / d preds.
 * ws: nbss_pit_ws_bytes() bytes. */
int64_t nbss_pit_ws_bytes(int B, int S);
int nbss_pit_neg_sisdr(int B, int S, int N, const float* preds, const float* target, float* loss, int32_t* perm, float* dpreds, float* ws,
                       void* stream);

/* clip_grad_norm_(max_norm, L2) + torch.optim.Adam(W) step on the flat fp32 buffers
 * (configs/SpatialNet.yaml:3-4,44; general_steps.py:243-271).  grads are first multiplied by
 * grad_scale (1/world_size after a SUM all-reduce).  scratch: >= 258 floats; scratch[0] returns the
 * (scaled) gradient norm.  step counts from 1.  flags: bit 0 clears grads for the next step; bit 1 selects
 * torch.optim.AdamW's decoupled weight decay (p *= 1 - lr*wd) instead of torch.optim.Adam's L2 term in the gradient. */
int nbss_clip_adam_step(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* scratch, float max_norm,
                        float grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay, int step, int flags,
                        void* stream);

/* The same update for HIP-graph replay (nbss_amd/engine.py: TrainStep.graph_step; the reference's per-step host work is Lightning's optimizer
 * loop, general_steps.py:243-271): the per-step scalars are read from the DEVICE buffer hyper[3] = {lr, 1 - beta1^step, sqrt(1 - beta2^step)},
 * which nbss_adam_hyper fills on the HOST (hyper_host: 3 floats of host memory, copied to the device by the caller before the replay) with the
 * very expressions nbss_clip_adam_step evaluates — a replayed step is bitwise the eager one. */
int nbss_clip_adam_step_dev(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* scratch, const float* hyper, float max_norm,
                            float grad_scale, float beta1, float beta2, float eps, float weight_decay, int flags, void* stream);
int nbss_adam_hyper(int step, float lr, float beta1, float beta2, float* hyper_host);

/* ---- OnlineSpatialNet: native streaming step (models/arch/OnlineSpatialNet.py:22-60,171-200,333-354; base/retention.py:194-253) -------
 * One call advances a chunk of C <= 32 frames of every (batch, frequency) sequence; all state lives in caller-owned fp32 device
 * buffers updated in place, so a whole step is a fixed launch sequence (capturable in a HIP graph).  fp32; geometry: dim_hidden 96,
 * dim_ffn 192, 8 conv groups of 24, 4 retention heads (key 24, value 48).  The cross-band blocks of a layer are per-frame operations:
 * nbss_fconv_fwd / nbss_full_fwd on the [B,F,C,H] chunk (cfg->T = C); the decoder: nbss_decoder_fwd.
 * encoder: causal Conv1d(C_in -> 96, k = 5): x [BF][C][C_in], state [BF][4][C_in] (the last four input frames), y [BF][C][96]. */
int nbss_online_encoder_step(int BF, int C, int C_in, const float* weight, const float* bias, const float* x, float* state, float* y, void* stream);
/* x += out_proj(SiLU(g) * RMSNorm_head(retention(q, k, v))) with LayerNorm(x) as input, recurrent form: kv [BF][4][24][48] and
 * scale [BF][4] (running decay sum, one copy per sequence) carry over; w*_t are the projection weights TRANSPOSED ([in][out]);
 * wk_t = NULL shares k with q ('ret(2,share_qk)'); decay [4] = the per-head gamma.  x [BF][C][96] in place. */
int nbss_online_ret_step(int BF, int C, const float* ln_w, const float* ln_b, const float* wq_t, const float* wk_t, const float* wv_t, const float* wg_t,
                         const float* wo_t, const float* decay, float* kv, float* scale, float* x, void* stream);
/* 'mhsa(N)': x += out_proj(causal windowed attention over the last `scope` frames) with LayerNorm(x) as input.  kring / vring [BF][ring][96]: the projected
 * keys / values of the last frames at slot (frame index mod ring), ring >= scope - 1 + C; pos[0] (device int32): frames seen before this chunk — every layer's
 * call of a step reads it, nbss_online_advance adds C once at the end of the step.  win_t [96][288] / wo_t [96][96]: in_proj / out_proj weights transposed. */
int nbss_online_mhsa_step(int BF, int C, int scope, int ring, const float* ln_w, const float* ln_b, const float* win_t, const float* bin, const float* wo_t,
                          const float* bo, float* kring, float* vring, const int32_t* pos, float* x, void* stream);
int nbss_online_advance(int32_t* pos, int C, void* stream);
/* x += causal T-ConvFFN(x): LayerNorm -> 1x1 -> SiLU -> causal gconv -> SiLU -> causal gconv -> GroupNorm of each FRAME over (24 channels
 * x all F frequencies) -> SiLU -> causal gconv -> SiLU -> 1x1.  w1_t [96][192], w2_t [192][96] transposed; conv weights [192][24][3];
 * s1 s2 s3 [BF][2][192] = the last two input frames of the three convs; a3 [BF][C][192] and gn_sums [B + BF][C][8][2] (the per-frame
 * GroupNorm sums, then every frequency's partials: folded in frequency order, no atomics — bitwise repeatable) are scratch. */
int nbss_online_tconvffn_step(int B, int F, int C, const float* ln_w, const float* ln_b, const float* w1_t, const float* b1, const float* c1w,
                              const float* c1b, const float* c2w, const float* c2b, const float* gn_w, const float* gn_b, const float* c3w, const float* c3b,
                              const float* w2_t, const float* b2, float* s1, float* s2, float* s3, float* a3, float* gn_sums, float* x, void* stream);

/* ---- narrow-band building blocks (models/arch/NBC2.py:152-238 in the reference: pre-norm self-attention over time + convolutional feed-forward with
 * GroupBatchNorm, per (batch, frequency) sequence) -------------------------------------------------------------------------------------------------
 * Geometry-generic kernels (csrc/gbwd.hip), one operation per call on caller-owned tensors of `dtype` (NBSS_F32 | NBSS_BF16) in the [nseq][T][C] layout
 * of the reference's [B*F, T, C] activations; weights / biases / affines are the fp32 parameters in their state_dict layout.  nbss_amd/nbc2.py sequences an
 * NBC2 forward from them.  act_in / act_out: 1 = SiLU applied to the input as it is read / to the result.
 * conv_t: y[n][t][o] = sum_tap sum_i x[n][t + tap - taps/2][i] w[o][i][tap] + bias[o] (+ residual[n][t][o]), grouped, zero padded ("same"); taps = 1 is a
 * per-token Linear (w [Cout][Cin]).  x rows are ldx elements apart (ldx >= Cin; groups = 1: ldx % 8 == 0 and the columns Cin..round_up(Cin, 8) must be
 * zero; groups > 1: ldx == Cin, Cin / groups % 8 == 0).  ws: nbss_nb_ws_bytes(Cout, Cin, groups, taps) bytes of scratch (the re-laid weights). */
int64_t nbss_nb_ws_bytes(int Cout, int Cin, int groups, int taps);
int nbss_nb_conv_t(int dtype, int64_t nseq, int T, int Cin, int ldx, int Cout, int groups, int taps, const void* x, const float* w, const float* bias, void* y,
                   const void* residual, int act_in, int act_out, void* ws, void* stream);
/* LayerNorm over C (eps 1e-5): y = xhat gamma + beta; stats [rows][2] fp32 scratch (mean, rstd) */
int nbss_nb_layernorm(int dtype, int64_t rows, int C, const void* x, const float* gamma, const float* beta, void* y, float* stats, void* stream);
/* GroupBatchNorm (NBC2.py:57-145, share_along_sequence_dim = false): statistics over the F sequences of an utterance x C features per frame, from the input
 * itself; x, y [B][F][T][C]; gamma / beta [C] or NULL */
int nbss_nb_group_batch_norm(int dtype, int B, int F, int T, int C, const void* x, const float* gamma, const float* beta, float eps, int act_out, void* y,
                             void* stream);
/* softmax(q k^T / sqrt(dh)) v per (sequence, head): qkv [nseq][T][3H] (q | k | v; head h at columns h dh), o [nseq][T][H]; T <= 256, dh in {24, 48} */
int nbss_nb_attention_fwd(int dtype, int64_t nseq, int T, int H, int heads, const void* qkv, void* o, void* stream);
/* the narrow-band conformer NBC (reference models/arch/NBC.py): Transformer-XL relative-position attention (NBC.py:106-143) —
 *   softmax(((q + u) k^T + (q + v) P[i - j]) * scale) v per (sequence, head); pos [2T - 1][H] = pos_proj of the sinusoid table for the offsets
 *   -(T - 1) .. T - 1 (stream dtype), u_bias / v_bias [heads][dh] fp32, scale = 1 / sqrt(d_model) in the reference; T <= 256, dh in {24, 48}
 * — and the GroupNorm(groups, C) + optional SiLU between the convolutions of its feed-forward (NBC.py:195-203; statistics over (C / groups) x T per
 * sequence, eps 1e-5): x, y [nseq][T][C]. */
int nbss_nb_attention_relpos_fwd(int dtype, int64_t nseq, int T, int H, int heads, const void* qkv, const void* pos, const float* u_bias, const float* v_bias,
                                 float scale, void* o, void* stream);
int nbss_nb_group_norm(int dtype, int64_t nseq, int T, int C, int groups, const void* x, const float* gamma, const float* beta, int act_out, void* y, void* stream);

/* ---- the same building blocks for TRAINING (autograd of the reference's torch.nn NBC2: NBC2.py:152-238; sequenced by nbss_amd/nbc2.py) ---------------
 * conv_t_train: conv_t without fused input / output activations, plus an optional second output y_silu = SiLU(y) (pre-activation and activation of a
 *   feed-forward step in one pass).
 * conv_t_bwd: gradients of y = conv_t(x): dx [nseq][T][ldx] = conv^T(dy) — multiplied by SiLU'(x_pre) when x_pre (the pre-activation x = SiLU(x_pre) was made
 *   from, same layout as dx) is given — and dw [Cout][Cin / groups][taps] += dy^T x, dbias [Cout] += colsum(dy) (fp32, ACCUMULATED; either of dx / dw may be
 *   NULL).  ws: nbss_nb_bwd_ws_bytes(Cout, Cin, groups, taps) bytes (the re-laid transposed weights + partial tiles of the weight gradient).
 * layernorm_bwd: dx = dres + LayerNorm'(dy) with the forward's stats; dgamma / dbeta accumulated.
 * group_batch_norm_bwd: gradient through y = act(GroupBatchNorm(x)) (statistics recomputed from x); dgamma / dbeta accumulated (NULL when not affine).
 * attention_bwd: dqkv [nseq][T][3H] from qkv and d_o = gradient w.r.t. the attention output; ws: nbss_nb_attention_bwd_ws_bytes() bytes. */
int64_t nbss_nb_bwd_ws_bytes(int Cout, int Cin, int groups, int taps);
int nbss_nb_conv_t_train(int dtype, int64_t nseq, int T, int Cin, int ldx, int Cout, int groups, int taps, const void* x, const float* w, const float* bias, void* y,
                         void* y_silu, const void* residual, void* ws, void* stream);
int nbss_nb_conv_t_bwd(int dtype, int64_t nseq, int T, int Cin, int ldx, int Cout, int groups, int taps, const void* x, const float* w, const void* dy,
                       const void* x_pre, void* dx, float* dw, float* dbias, void* ws, void* stream);
int nbss_nb_layernorm_bwd(int dtype, int64_t rows, int C, const void* x, const float* stats, const float* gamma, const void* dy, const void* dres, void* dx,
                          float* dgamma, float* dbeta, void* stream);
int nbss_nb_group_batch_norm_bwd(int dtype, int B, int F, int T, int C, const void* x, const float* gamma, const float* beta, float eps, int act_out, const void* dy,
                                 void* dx, float* dgamma, float* dbeta, void* stream);
int64_t nbss_nb_attention_bwd_ws_bytes(int dtype, int64_t nseq, int T, int H, int heads);
int nbss_nb_attention_bwd(int dtype, int64_t nseq, int T, int H, int heads, const void* qkv, const void* d_o, void* dqkv, void* ws, void* stream);

/* ---- ... and for training the narrow-band conformer NBC (autograd of the reference's NBC.py:73-237; sequenced by nbss_amd/nbc.py) --------------------------
 * attention_relpos_train: the relative-position attention with attention dropout (NBC.py:137): keep_bits [nseq][heads][T][ceil(T / 32)] uint32 — bit (j & 31) of
 *   word j >> 5 of row i = probability (i, j) kept — or NULL; kept probabilities are scaled by keep_scale = 1 / (1 - p).  The caller draws the bits (any RNG):
 *   forward and backward read the same tensor.
 * attention_relpos_bwd: dqkv [nseq][T][3H] from qkv, pos, the biases, the bits and d_o; dpos [2T - 1][H], du_bias / dv_bias [H] (fp32) ACCUMULATED (sums over
 *   all sequences, fixed order: no atomics).  ws: nbss_nb_attention_relpos_bwd_ws_bytes() bytes.
 * group_norm_train: group_norm that also keeps (mean, rstd) per (sequence, group) in stats [nseq * groups][2];  group_norm_bwd: gradient through
 *   y = SiLU(GroupNorm(x)) IN PLACE (dy_dx holds dy on entry, dx on return); dgamma / dbeta [C] accumulated. */
int nbss_nb_attention_relpos_train(int dtype, int64_t nseq, int T, int H, int heads, const void* qkv, const void* pos, const float* u_bias, const float* v_bias,
                                   float scale, const uint32_t* keep_bits, float keep_scale, void* o, void* stream);
int64_t nbss_nb_attention_relpos_bwd_ws_bytes(int64_t nseq, int T, int H, int heads);
int nbss_nb_attention_relpos_bwd(int dtype, int64_t nseq, int T, int H, int heads, const void* qkv, const void* pos, const float* u_bias, const float* v_bias, float scale,
                                 const uint32_t* keep_bits, float keep_scale, const void* d_o, void* dqkv, float* dpos, float* du_bias, float* dv_bias, void* ws,
                                 void* stream);
int nbss_nb_group_norm_train(int dtype, int64_t nseq, int T, int C, int groups, const void* x, const float* gamma, const float* beta, int act_out, void* y, float* stats,
                             void* stream);
int nbss_nb_group_norm_bwd(int dtype, int64_t nseq, int T, int C, int groups, const void* x, const float* stats, const float* gamma, const float* beta, void* dy_dx,
                           float* dgamma, float* dbeta, void* stream);

/* ---- the narrow-band BiLSTM (reference models/arch/blstm2_fc1.py:45-68: nn.LSTM(bidirectional) per frequency bin; sequenced by nbss_amd/blstm.py) ------------
 * One bidirectional layer's recurrences, all T steps in one persistent launch (a workgroup per tile of sequences and direction; gates GEMM on MFMA, h in LDS,
 * c in registers).  gx [nseq][T][ldg]: W_ih x + b_ih + b_hh of both directions (direction d at columns d * 4 hidden; gate order i | f | g | o), from
 * nbss_nb_conv_t.  w_hh / w_hh_reverse [4 hidden][hidden] fp32.  y [nseq][T][2 hidden] (direction d at columns d * hidden).  save (training; or NULL):
 * [2][nseq][T][5 hidden] i | f | g | o | c, stream dtype.  hidden in {128, 256}.  ws: nbss_nb_blstm_ws_bytes() bytes (the packed recurrent weights).
 * blstm_bwd: dg [nseq][T][8 hidden] = gradient w.r.t. the gate pre-activations of both directions from dy (gradient w.r.t. y) and save; the weight, bias and
 *   input gradients are dense contractions of dg (nbss_nb_conv_t_bwd). */
int64_t nbss_nb_blstm_ws_bytes(int dtype, int hidden);
int nbss_nb_blstm_fwd(int dtype, int64_t nseq, int T, int hidden, int ldg, const void* gx, const float* w_hh, const float* w_hh_reverse, void* y, void* save, void* ws,
                      void* stream);
int nbss_nb_blstm_bwd(int dtype, int64_t nseq, int T, int hidden, const void* dy, const void* save, const float* w_hh, const float* w_hh_reverse, void* dg, void* ws,
                      void* stream);

/* ---- diagnostics ---------------------------------------------------------------------------*/
/* D = A(16x32) * B(32x16) through the same MFMA fragment helpers the kernels use
 * (natural or permuted K order); used by the tests to pin the gfx950 fragment layouts. */
int nbss_selftest_mma(int dtype, int kperm, const float* A, const float* B, float* D, void* stream);
const char* nbss_build_info(void);
/* Opt-in per-kernel timing: HIP events recorded on the launch stream around each kernel whose bit is
 * set in `mask` (bit i = kernel id i, 0 .. nbss_profile_kernels()-1; 0 disables).  nbss_profile_read
 * waits for the recorded events, returns summed milliseconds / launch counts per id and clears them. */
int nbss_profile_enable(int64_t mask);
int nbss_profile_kernels(void);
const char* nbss_profile_name(int id);
int nbss_profile_read(double* total_ms, int64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* NBSS_HIP_H */
