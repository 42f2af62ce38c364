// Flat fp32 parameter layout (reference state_dict order) and the layout of
// the packed MFMA weight-fragment buffer.  Pure arithmetic on nbss_cfg so that host code and
// device code (pack.hip) agree by construction.
#pragma once
#include "../../include/nbss_hip.h"

#ifdef NBSS_EMU
#define NBSS_HD inline
#else
#define NBSS_HD __host__ __device__ inline
#endif

enum LayerParam {
    P_FC1_LN_W, P_FC1_LN_B, P_FC1_W, P_FC1_B, P_FC1_PRELU,
    P_FULL_LN_W, P_FULL_LN_B, P_SQ_W, P_SQ_B, P_FULL_W, P_FULL_B, P_USQ_W, P_USQ_B,
    P_FC2_LN_W, P_FC2_LN_B, P_FC2_W, P_FC2_B, P_FC2_PRELU,
    P_MH_LN_W, P_MH_LN_B, P_INP_W, P_INP_B, P_OUTP_W, P_OUTP_B,
    P_TF_LN_W, P_TF_LN_B, P_TF_W1, P_TF_B1, P_TF_C1W, P_TF_C1B, P_TF_C2W, P_TF_C2B,
    P_TF_GN_W, P_TF_GN_B, P_TF_C3W, P_TF_C3B, P_TF_W2, P_TF_B2,
    NUM_LAYER_PARAMS
};

NBSS_HD int64_t layer_param_numel(const nbss_cfg& c, int p) {
    const int64_t H = c.H, FFN = c.FFN, SQ = c.SQ, F = c.F;
    switch (p) {
        case P_FC1_LN_W: case P_FC1_LN_B: case P_FC2_LN_W: case P_FC2_LN_B:
        case P_FULL_LN_W: case P_FULL_LN_B: case P_MH_LN_W: case P_MH_LN_B:
        case P_TF_LN_W: case P_TF_LN_B:
        case P_FC1_B: case P_FC2_B: case P_FC1_PRELU: case P_FC2_PRELU:
        case P_USQ_B: case P_OUTP_B: case P_TF_B2:
            return H;
        case P_FC1_W: case P_FC2_W: return H * (H / c.f_groups) * c.f_ks;
        case P_SQ_W: return SQ * H;
        case P_SQ_B: return SQ;
        case P_FULL_W: return SQ * F * F;
        case P_FULL_B: return SQ * F;
        case P_USQ_W: return H * SQ;
        case P_INP_W: return 3 * H * H;
        case P_INP_B: return 3 * H;
        case P_OUTP_W: return H * H;
        case P_TF_W1: return FFN * H;
        case P_TF_B1: case P_TF_C1B: case P_TF_C2B: case P_TF_C3B: case P_TF_GN_W: case P_TF_GN_B: return FFN;
        case P_TF_C1W: case P_TF_C2W: case P_TF_C3W: return FFN * (FFN / c.t_groups) * c.t_ks;
        case P_TF_W2: return H * FFN;
    }
    return 0;
}

NBSS_HD int64_t enc_w_numel(const nbss_cfg& c) { return (int64_t)c.H * c.C_in * c.enc_ks; }
NBSS_HD int64_t param_off_enc_w(const nbss_cfg&) { return 0; }
NBSS_HD int64_t param_off_enc_b(const nbss_cfg& c) { return enc_w_numel(c); }

NBSS_HD int64_t layer_numel(const nbss_cfg& c, int l) {
    int64_t s = 0;
    for (int p = 0; p < NUM_LAYER_PARAMS; ++p) {
        if ((p == P_FULL_W || p == P_FULL_B) && l > c.full_share) continue;
        s += layer_param_numel(c, p);
    }
    return s;
}
NBSS_HD int64_t layer_base(const nbss_cfg& c, int l) {
    int64_t o = enc_w_numel(c) + c.H;
    for (int i = 0; i < l; ++i) o += layer_numel(c, i);
    return o;
}
NBSS_HD int64_t param_off(const nbss_cfg& c, int l, int p) {
    if ((p == P_FULL_W || p == P_FULL_B) && l > c.full_share) l = c.full_share;
    int64_t o = layer_base(c, l);
    for (int q = 0; q < p; ++q) {
        if ((q == P_FULL_W || q == P_FULL_B) && l > c.full_share) continue;
        o += layer_param_numel(c, q);
    }
    return o;
}
NBSS_HD int64_t param_off_dec_w(const nbss_cfg& c) { return layer_base(c, c.L); }
NBSS_HD int64_t param_off_dec_b(const nbss_cfg& c) { return param_off_dec_w(c) + (int64_t)c.C_out * c.H; }
NBSS_HD int64_t param_total(const nbss_cfg& c) { return param_off_dec_b(c) + c.C_out; }

// per-layer parameter pointers, resolved on the HOST and passed by value: the offset arithmetic above
// loops over layers and must never run inside a kernel (it cost +290 us per layer index when it did)
struct LayerPtrs {
    const float* p[NUM_LAYER_PARAMS];
};
inline LayerPtrs layer_ptrs(const nbss_cfg& c, const float* P, int layer) {
    LayerPtrs lp;
    for (int i = 0; i < NUM_LAYER_PARAMS; ++i) lp.p[i] = P + param_off(c, layer, i);
    return lp;
}

// ---- packed fragment buffer ------------------------------------------------------------------
// Every entry is [MT tiles][KS ksteps][64 lanes][8] elements of the stream dtype.
enum PackKind {
    // forward operands
    K_ENC = 0,   // encoder conv: rows = out ch, K = (tap, 4-ch piece)
    K_DEC,       // decoder linear, natural K
    K_FC1, K_FC2,  // f-conv, per group: rows = 12(+4 pad) out ch, K = (tap, 4-ch piece)
    K_SQ,        // squeeze 1x1, natural K
    K_FULL,      // LinearGroup, per squeeze channel: rows = k (out freq), K = h natural
    K_USQ,       // unsqueeze 1x1, K = SQ natural
    K_INP,       // in_proj: rows = [q|k|v] x heads x (dh + pad to 32), natural K
    K_OUTP,      // out_proj: K = heads x 32, permuted order inside each head
    K_TF_W1,     // 1x1 H->FFN, rows = groups x (cg + pad to 32), natural K
    K_TF_C1, K_TF_C2, K_TF_C3,  // t-conv, per group: rows = cg (+pad), K = (tap, 4-ch piece)
    K_TF_W2,     // 1x1 FFN->H, K = groups x 32 permuted
    // backward (data-gradient) operands: transposed weights
    K_DEC_T,     // dx = W^T dout: rows = H, K = C_out natural
    K_FC1_T, K_FC2_T,  // f-conv^T per group: rows = 12(+4) in ch, K = (flipped tap, 4-out-ch piece)
    K_SQ_T,      // du = Ws^T ds: rows = H, K = SQ natural
    K_FULL_T,    // per squeeze channel: rows = h (in freq), K = k natural
    K_USQ_T,     // dz = Wu^T dy: rows = SQ (+pad), K = H permuted
    K_INP_T,     // du = Win^T dqkv: rows = H, K = [q|k|v] x heads x 32 permuted
    K_OUTP_T,    // dO = Wo^T dy: rows = heads x (dh + pad to 32), K = H natural
    K_TF_W1_T,   // du = W1^T da1: rows = H, K = groups x 32 permuted
    K_TF_C1_T, K_TF_C2_T, K_TF_C3_T,  // t-conv^T per group: rows = cg in ch, K = (flipped tap, 4-out-ch piece)
    K_TF_W2_T,   // dh5 = W2^T dy: rows = groups x (cg + pad to 32), K = H natural
    K_TF_W1_TN,  // du = W1^T da1 from the emitted [N][FFN] operand: rows = H, K = FFN natural
    K_INP_TN,    // du = Win^T dqkv from the emitted [N][3H] operand: rows = H, K = 3H natural
    // ---- 32x32x16 fragments of the streaming T-ConvFFN kernels (tconvffn_s.hip): lane l = row l&31, K slots 8(l>>5)+j ----
    K_TS_W1,     // per group: rows = cg (+pad to 32), K = H natural, 16 per k-step
    K_TS_C1, K_TS_C2, K_TS_C3,  // t-conv per group: rows = cg out ch, K = (tap, 8-channel block) natural + bias slot (ts_conv_k)
    K_TS_W2,     // 1x1 FFN->H: rows = H (3 tiles of 32), K = FFN natural
    K_TS_C1_T, K_TS_C2_T, K_TS_C3_T,  // t-conv^T per group: rows = cg in ch, K = (flipped tap, 8-out-channel block), no bias
    K_TS_W2_T,   // per group: dh5_g = W2[:, g]^T dy: rows = cg, K = H natural
    NUM_PACK_KINDS
};

struct PackGeom {
    int MT, KS, NB;  // tiles, ksteps, number of independent blocks (groups) in this entry
};

NBSS_HD int cdiv(int a, int b) { return (a + b - 1) / b; }

NBSS_HD PackGeom pack_geom(const nbss_cfg& c, int kind) {
    PackGeom g = {0, 0, 1};
    const int dh = c.H / c.heads, cg = c.FFN / c.t_groups, fg = c.H / c.f_groups;
    switch (kind) {
        case K_ENC: g.MT = c.H / 16; g.KS = cdiv(c.enc_ks * (c.C_in / 4), 8); break;
        case K_DEC: g.MT = cdiv(c.C_out, 16); g.KS = c.H / 32; break;
        case K_FC1: case K_FC2: g.MT = cdiv(fg, 16); g.KS = cdiv(c.f_ks * (fg / 4), 8); g.NB = c.f_groups; break;
        case K_SQ: g.MT = cdiv(c.SQ, 16); g.KS = c.H / 32; break;
        case K_FULL: g.MT = cdiv(c.F, 16); g.KS = cdiv(c.F, 32); g.NB = c.SQ; break;
        case K_USQ: g.MT = c.H / 16; g.KS = cdiv(c.SQ, 32); break;
        case K_INP: g.MT = 3 * c.heads * cdiv(dh, 32) * 2; g.KS = c.H / 32; break;
        case K_OUTP: g.MT = c.H / 16; g.KS = c.heads * cdiv(dh, 32); break;
        case K_TF_W1: g.MT = c.t_groups * cdiv(cg, 32) * 2; g.KS = c.H / 32; break;
        case K_TF_C1: case K_TF_C2: case K_TF_C3: g.MT = cdiv(cg, 32) * 2; g.KS = cdiv(c.t_ks * (cg / 4), 8); g.NB = c.t_groups; break;
        case K_TF_W2: g.MT = c.H / 16; g.KS = c.t_groups * cdiv(cg, 32); break;
        case K_DEC_T: g.MT = c.H / 16; g.KS = cdiv(c.C_out, 32); break;
        case K_FC1_T: case K_FC2_T: g.MT = cdiv(fg, 16); g.KS = cdiv(c.f_ks * (fg / 4), 8); g.NB = c.f_groups; break;
        case K_SQ_T: g.MT = c.H / 16; g.KS = cdiv(c.SQ, 32); break;
        case K_FULL_T: g.MT = cdiv(c.F, 16); g.KS = cdiv(c.F, 32); g.NB = c.SQ; break;
        case K_USQ_T: g.MT = cdiv(c.SQ, 16); g.KS = c.H / 32; break;
        case K_INP_T: g.MT = c.H / 16; g.KS = 3 * c.heads * cdiv(dh, 32); break;
        case K_OUTP_T: g.MT = c.heads * cdiv(dh, 32) * 2; g.KS = c.H / 32; break;
        case K_TF_W1_T: g.MT = c.H / 16; g.KS = c.t_groups * cdiv(cg, 32); break;
        case K_TF_C1_T: case K_TF_C2_T: case K_TF_C3_T: g.MT = cdiv(cg, 32) * 2; g.KS = cdiv(c.t_ks * (cg / 4), 8); g.NB = c.t_groups; break;
        case K_TF_W2_T: g.MT = c.t_groups * cdiv(cg, 32) * 2; g.KS = c.H / 32; break;
        case K_TF_W1_TN: g.MT = c.H / 16; g.KS = c.FFN / 32; break;
        case K_INP_TN: g.MT = c.H / 16; g.KS = 3 * c.H / 32; break;
        case K_TS_W1: g.MT = 1; g.KS = c.H / 16 + 1; g.NB = c.t_groups; break;  // + one k-step whose first slot is the bias
        case K_TS_W2_T: g.MT = 1; g.KS = c.H / 16; g.NB = c.t_groups; break;
        case K_TS_C1: case K_TS_C2: case K_TS_C3: case K_TS_C1_T: case K_TS_C2_T: case K_TS_C3_T:
            g.MT = 1; g.KS = cdiv(c.t_ks * (cg / 8) + 1, 2); g.NB = c.t_groups; break;
        case K_TS_W2: g.MT = c.H / 32; g.KS = c.FFN / 16 + 1; break;  // + bias k-step
    }
    return g;
}
// K order of the streaming t-conv fragments (32x32x16, tconvffn_s.hip): natural (tap, channel) order in blocks of 8 channels,
// lane half h of k-step ks takes block 2 ks + h: blocks 0..8 = 3 taps x 3 channel blocks; block 9 (ks = 4, h = 1) carries the
// BIAS in its first slot (the B operand holds a constant 1 there).  Returns 0 padding, 1 weight (tap, ch), 2 bias slot.
NBSS_HD int ts_conv_k(int ks, int h, int j, int& tap, int& ch) {
    const int blk = 2 * ks + h;
    if (blk < 9) {
        tap = blk / 3;
        ch = (blk % 3) * 8 + j;
        return 1;
    }
    return (blk == 9 && j == 0) ? 2 : 0;
}
NBSS_HD int64_t pack_numel(const nbss_cfg& c, int kind) {
    PackGeom g = pack_geom(c, kind);
    return (int64_t)g.NB * g.MT * g.KS * 512;
}
NBSS_HD bool pack_is_global(int kind) { return kind == K_ENC || kind == K_DEC || kind == K_DEC_T; }
// offset (in elements) of entry `kind` for `layer` inside the packed buffer
NBSS_HD int64_t pack_layer_numel(const nbss_cfg& c) {
    int64_t s = 0;
    for (int k = 0; k < NUM_PACK_KINDS; ++k)
        if (!pack_is_global(k)) s += pack_numel(c, k);
    return s;
}
NBSS_HD int64_t pack_off(const nbss_cfg& c, int layer, int kind) {
    if (kind == K_ENC) return 0;
    if (kind == K_DEC) return pack_numel(c, K_ENC);
    if (kind == K_DEC_T) return pack_numel(c, K_ENC) + pack_numel(c, K_DEC);
    int64_t o = pack_numel(c, K_ENC) + pack_numel(c, K_DEC) + pack_numel(c, K_DEC_T) + (int64_t)layer * pack_layer_numel(c);
    for (int k = 0; k < kind; ++k)
        if (!pack_is_global(k)) o += pack_numel(c, k);
    return o;
}
NBSS_HD int64_t pack_total(const nbss_cfg& c) {
    return pack_numel(c, K_ENC) + pack_numel(c, K_DEC) + pack_numel(c, K_DEC_T) + (int64_t)c.L * pack_layer_numel(c);
}

// wgrad partial tiles: up to 512 workgroups x 112 tiles x (256 accumulators + 16 bias sums) floats
#define WGPART_BYTES ((size_t)512 * 112 * 272 * sizeof(float))
// backward workspace (caller-provided): per-token LN statistics + the widest set of wgrad operands
#ifndef NBSS_WS_PAD
#define NBSS_WS_PAD 0  // (A/B flavour: extra bytes behind every aligned region.  At batch 32 all regions start at multiples of 2 KB; 1 280 or 4 352 bytes of padding
                       //  changed nothing — 727 / 728 / 727 utt/s in one call: the batch-31 / 32 / 33 steps of 738 / 730 / 718 are round counts, not address aliasing)
#endif
NBSS_HD size_t ws_align(size_t b) { return ((b + 255) & ~(size_t)255) + NBSS_WS_PAD; }
// sequence lengths: backward keeps a whole sequence per workgroup (LDS), forward has chunked variants beyond that
#define NBSS_T_TRAIN_MAX 256
// frequencies: the cross-band kernels keep the whole F axis of a slab on chip: 17 tiles of 16 (n_fft 512 -> F = 257; fp32 backward: F <= 160)
#define NBSS_F_MAX 272
#define NBSS_T_MAX 4096
// fconv_bwd with the fused weight gradient (bf16, one frame per workgroup since round 5): per workgroup one fp32 row = the affine sums (3 H) + the conv
// bias sums (H) — affine_reduce folds those — and, behind ALL fp32 rows, one bf16 row = the whole conv weight gradient as this workgroup's frame sees it,
// [tap][group][input channel][outputs of the group] (round 6: 13 KB per workgroup instead of 24.6 KB of fp32; fconv.hip: fconv_part_final_kernel)
#define NBSS_FC_PROW(c) (4 * (c).H)
#define NBSS_FC_P16(c) ((c).H * ((c).H / (c).f_groups) * (c).f_ks)
#ifdef NBSS_FC_TT2  // (A/B flavour: round 4's two-frame slabs)
NBSS_HD size_t fc_part_bytes(const nbss_cfg& c) { return (size_t)c.B * ((c.T + 1) / 2) * (NBSS_FC_PROW(c) * sizeof(float) + NBSS_FC_P16(c) * 2); }
#else
NBSS_HD size_t fc_part_bytes(const nbss_cfg& c) { return (size_t)c.B * c.T * (NBSS_FC_PROW(c) * sizeof(float) + NBSS_FC_P16(c) * 2); }
#endif
// T-ConvFFN backward from saved pre-activations (tconvffn_s.hip: tconvffn_bwd_v_kernel; bf16 stream, small geometry): per sequence one fp32
// partial row (GroupNorm affine sums 2 FFN + the three conv bias sums 3 FFN + W2's bias sums H) and one bf16 row (the three conv weight
// gradients + the W2 weight gradient)
NBSS_HD size_t tc_part_bytes(const nbss_cfg& c) {
    return c.dtype == NBSS_BF16 && c.H == 96 && c.T <= 256
               ? (size_t)c.B * c.F * ((5 * c.FFN + c.H) * sizeof(float) + ((size_t)3 * c.FFN * (c.FFN / c.t_groups) * c.t_ks + (size_t)c.FFN * c.H) * 2)
               : 0;
}
// operand regions ([N][FFN] tensors of the stream dtype) behind the statistics: 8 for the fused backward kernels of the small geometry; the generic
// backward (gbwd.hip) keeps every intermediate of a block: the T-ConvFFN's 12 + LN output / gradient + re-laid weights
NBSS_HD size_t ws_nops(const nbss_cfg& c) { return c.H == 96 ? 8 : 16; }
// ... plus room for that path's re-laid weights (the LinearGroup pair at F = 272 is 10 MB in fp32), independent of the token count
NBSS_HD size_t ws_wprep_bytes(const nbss_cfg& c) { return c.H == 96 ? 0 : (size_t)16 << 20; }
// per-workgroup partial rows of the small (affine) parameter gradients: 576 floats per sequence / slab, plus the full-band block's fold scratch behind its
// rows (full.hip: D [SQ][H] | dbs | ones | zeros = 968 floats at an offset of up to nwg * SQ + 64) — on a single-token grid (nwg = 1) that scratch
// used to run into the wgrad partial tiles, whose writes zeroed four parameter gradients (round 5 review)
NBSS_HD size_t ws_part_bytes(const nbss_cfg& c) {
    const size_t nwg = (size_t)c.B * (c.F > c.T ? c.F : c.T);
    return ws_align((nwg * 576 + 64 + (size_t)c.SQ * c.H + c.SQ + 2 * c.H) * sizeof(float));
}
NBSS_HD size_t workspace_bytes(const nbss_cfg& c) {
    const size_t N = (size_t)c.B * c.F * c.T, esz = c.dtype == NBSS_BF16 ? 2 : 4;
    return ws_align(N * 2 * sizeof(float)) + ws_nops(c) * ws_align(N * c.FFN * esz) + ws_wprep_bytes(c) + ws_part_bytes(c) + ws_align(WGPART_BYTES) +
           ws_align(fc_part_bytes(c)) + ws_align(tc_part_bytes(c)) + 256;
}
// attention state saved by the forward pass for backward: O [N][H] (stream dtype) | log2-sum-exp [N][heads] fp32 | LayerNorm statistics [N][2] fp32
NBSS_HD size_t mhsa_lse_offset(const nbss_cfg& c) {
    return ws_align((size_t)c.B * c.F * c.T * c.H * (c.dtype == NBSS_BF16 ? 2 : 4));
}
// ... | LayerNorm (mean, rstd) of every token [N][2] fp32: the backward kernels re-apply the LayerNorm without recomputing its statistics
NBSS_HD size_t mhsa_stat_offset(const nbss_cfg& c) { return mhsa_lse_offset(c) + ws_align((size_t)c.B * c.F * c.T * c.heads * sizeof(float)); }
// (T > 256, forward only: the same buffer is the K | V scratch of the long-sequence attention path, two stream tensors)
NBSS_HD size_t mhsa_save_bytes(const nbss_cfg& c) {
    const size_t s = mhsa_stat_offset(c) + ws_align((size_t)c.B * c.F * c.T * 2 * sizeof(float));
    return (c.T > NBSS_T_TRAIN_MAX || c.H != 96) && s < 2 * mhsa_lse_offset(c) ? 2 * mhsa_lse_offset(c) : s;
}
// per-workgroup partial sums of the small (affine) parameter gradients live behind the wgrad operands
NBSS_HD size_t ws_part_offset(const nbss_cfg& c) {
    const size_t N = (size_t)c.B * c.F * c.T, esz = c.dtype == NBSS_BF16 ? 2 : 4;
    return ws_align(N * 2 * sizeof(float)) + ws_nops(c) * ws_align(N * c.FFN * esz) + ws_wprep_bytes(c);
}

// per-workgroup partial dW tiles of the wgrad kernels live behind those
NBSS_HD size_t ws_wgpart_offset(const nbss_cfg& c) { return ws_part_offset(c) + ws_part_bytes(c); }

// the fused f-conv weight-gradient partial rows live behind the wgrad partial tiles
NBSS_HD size_t ws_fcpart_offset(const nbss_cfg& c) { return ws_wgpart_offset(c) + ws_align(WGPART_BYTES); }
// ... and the T-ConvFFN partial rows behind those
NBSS_HD size_t ws_tcpart_offset(const nbss_cfg& c) { return ws_fcpart_offset(c) + ws_align(fc_part_bytes(c)); }

NBSS_HD int check_cfg(const nbss_cfg& c) {
    if (c.B <= 0 || c.F <= 0 || c.T <= 0 || c.L <= 0) return NBSS_EINVAL;
    if (c.dtype != NBSS_F32 && c.dtype != NBSS_BF16) return NBSS_EINVAL;
    // this build ships kernels for SpatialNet-small (configs/SpatialNet.yaml as shipped: training + inference) and for
    // SpatialNet-large (its "for large" comments: dim_hidden 192, dim_ffn 384, dim_squeeze 16; forward only — check_cfg_train)
    const bool small = c.H == 96 && c.FFN == 192 && c.SQ == 8 && c.heads == 4;
    const bool large = c.H == 192 && c.FFN == 384 && c.SQ == 16 && c.heads == 4;
    if (!small && !large) return NBSS_EUNSUPPORTED;
    if (c.f_groups != 8 || c.t_groups != 8 || c.f_ks != 5 || c.t_ks != 3 || c.enc_ks != 5) return NBSS_EUNSUPPORTED;
    if (c.C_in % 4 != 0 || c.C_in > 16 || c.C_out > 16 || c.C_out <= 0) return NBSS_EUNSUPPORTED;
    if (c.F > NBSS_F_MAX || c.T > NBSS_T_MAX) return NBSS_EUNSUPPORTED;
    if (c.full_share < 0 || c.full_share >= c.L) return NBSS_EINVAL;
    // operand offsets inside the widest tensor ([N][3H] dqkv) are 32-bit in the weight-gradient kernels
    if ((size_t)c.B * c.F * c.T * 3 * c.H >= ((size_t)1 << 31)) return NBSS_EUNSUPPORTED;
    return NBSS_OK;
}
