// pack.hip — re-layout of the fp32 master parameters into per-lane MFMA A-operand fragments.
// One thread per packed element; grid = (chunks, kind, layer).  Runs once per optimizer step
// (1.2 M parameters: a few microseconds), so every GEMM in the hot kernels fetches its weight
// fragment with a single 16-byte (bf16) load per lane and no index arithmetic.
#include "launch.h"
#include "layout.h"
#include "prof.h"

NBSS_DEV int perm_k(int g4, int j) { return j < 4 ? 4 * g4 + j : 16 + 4 * g4 + (j - 4); }


// flat-buffer offset of the weight tensor a pack kind reads (hoisted out of the per-element loop)
static int64_t pack_src_base(const nbss_cfg& c, int kind, int layer) {
    switch (kind) {
        case K_ENC: return param_off_enc_w(c);
        case K_DEC: case K_DEC_T: return param_off_dec_w(c);
        case K_FC1: case K_FC1_T: return param_off(c, layer, P_FC1_W);
        case K_FC2: case K_FC2_T: return param_off(c, layer, P_FC2_W);
        case K_SQ: case K_SQ_T: return param_off(c, layer, P_SQ_W);
        case K_FULL: case K_FULL_T: return param_off(c, layer, P_FULL_W);
        case K_USQ: case K_USQ_T: return param_off(c, layer, P_USQ_W);
        case K_INP: case K_INP_T: return param_off(c, layer, P_INP_W);
        case K_OUTP: case K_OUTP_T: return param_off(c, layer, P_OUTP_W);
        case K_TF_W1: case K_TF_W1_T: return param_off(c, layer, P_TF_W1);
        case K_TF_C1: case K_TF_C1_T: return param_off(c, layer, P_TF_C1W);
        case K_TF_C2: case K_TF_C2_T: return param_off(c, layer, P_TF_C2W);
        case K_TF_C3: case K_TF_C3_T: return param_off(c, layer, P_TF_C3W);
        case K_TF_W2: case K_TF_W2_T: return param_off(c, layer, P_TF_W2);
        case K_TF_W1_TN: return param_off(c, layer, P_TF_W1);
        case K_INP_TN: return param_off(c, layer, P_INP_W);
        case K_TS_W1: return param_off(c, layer, P_TF_W1);
        case K_TS_C1: case K_TS_C1_T: return param_off(c, layer, P_TF_C1W);
        case K_TS_C2: case K_TS_C2_T: return param_off(c, layer, P_TF_C2W);
        case K_TS_C3: case K_TS_C3_T: return param_off(c, layer, P_TF_C3W);
        case K_TS_W2: case K_TS_W2_T: return param_off(c, layer, P_TF_W2);
    }
    return 0;
}

// value of A[m-tile mt, row i=lane&15][kstep ks, lane group g4, slot j] for (kind, layer, block nb)
// flat-buffer offset of the bias a pack kind folds into a spare K slot (-1: none)
static int64_t pack_src2_base(const nbss_cfg& c, int kind, int layer) {
    switch (kind) {
        case K_TS_W1: return param_off(c, layer, P_TF_B1);
        case K_TS_C1: return param_off(c, layer, P_TF_C1B);
        case K_TS_C2: return param_off(c, layer, P_TF_C2B);
        case K_TS_C3: return param_off(c, layer, P_TF_C3B);
        case K_TS_W2: return param_off(c, layer, P_TF_B2);
    }
    return -1;
}

NBSS_DEV float pack_value(const nbss_cfg& c, const float* __restrict__ P /* the source weight tensor */, const float* __restrict__ P2 /* its bias */, int kind, int nb,
                          int mt, int ks, int lane, int j) {
    if (kind >= K_TS_W1) {  // 32x32x16 fragments: row = lane & 31, K slots 8 (lane >> 5) + j
        const int m = mt * 32 + (lane & 31), h = lane >> 5, knat = 16 * ks + 8 * h + j;
        const int cg = c.FFN / c.t_groups;
        int tap, ch;
        switch (kind) {
            case K_TS_W1:
                if (m >= cg) return 0.f;
                if (ks == c.H / 16) return (h == 0 && j == 0) ? P2[nb * cg + m] : 0.f;
                return P[(int64_t)(nb * cg + m) * c.H + knat];
            case K_TS_W2:
                if (ks == c.FFN / 16) return (h == 0 && j == 0) ? P2[m] : 0.f;
                return P[(int64_t)m * c.FFN + knat];
            case K_TS_W2_T: return m < cg ? P[(int64_t)knat * c.FFN + nb * cg + m] : 0.f;
            case K_TS_C1: case K_TS_C2: case K_TS_C3: {
                if (m >= cg) return 0.f;
                const int kk = ts_conv_k(ks, h, j, tap, ch);
                if (kk == 2) return P2[nb * cg + m];
                return kk == 1 ? P[((int64_t)(nb * cg + m) * cg + ch) * c.t_ks + tap] : 0.f;
            }
            case K_TS_C1_T: case K_TS_C2_T: case K_TS_C3_T:
                // dh[t][i] = sum_{tap',o} W[o][i][ks-1-tap'] da[t + tap' - 1][o]
                if (m >= cg || ts_conv_k(ks, h, j, tap, ch) != 1) return 0.f;
                return P[((int64_t)(nb * cg + ch) * cg + m) * c.t_ks + (c.t_ks - 1 - tap)];
        }
        return 0.f;
    }
    const int l15 = lane & 15, g4 = lane >> 4;
    const int m = mt * 16 + l15;
    const int knat = ks * 32 + 8 * g4 + j;
    const int H = c.H;
    switch (kind) {
        case K_ENC: {
            const int pc = c.C_in / 4, p = ks * 8 + g4 * 2 + (j >> 2);
            if (p >= c.enc_ks * pc) return 0.f;
            const int tap = p / pc, i = (p % pc) * 4 + (j & 3);
            return P[((int64_t)m * c.C_in + i) * c.enc_ks + tap];
        }
        case K_DEC:
            return m < c.C_out ? P[(int64_t)m * H + knat] : 0.f;
        case K_FC1: case K_FC2: {
            const int fg = H / c.f_groups, pc = fg / 4, p = ks * 8 + g4 * 2 + (j >> 2);
            if (m >= fg || p >= c.f_ks * pc) return 0.f;
            const int tap = p / pc, i = (p % pc) * 4 + (j & 3);
            return P[((int64_t)(nb * fg + m) * fg + i) * c.f_ks + tap];
        }
        case K_SQ:
            return m < c.SQ ? P[(int64_t)m * H + knat] : 0.f;
        case K_FULL:
            return (m < c.F && knat < c.F) ? P[((int64_t)nb * c.F + m) * c.F + knat] : 0.f;
        case K_USQ:
            return knat < c.SQ ? P[(int64_t)m * c.SQ + knat] : 0.f;
        case K_INP: {  // tiles per head: the head's dh rows padded to a multiple of 32 (dh = 24: 2 tiles, dh = 48: 4)
            const int dh = H / c.heads, tph = cdiv(dh, 32) * 2;
            const int which = mt / (c.heads * tph), rem = mt % (c.heads * tph), head = rem / tph, d = (rem % tph) * 16 + l15;
            if (d >= dh) return 0.f;
            return P[(int64_t)(which * H + head * dh + d) * H + knat];
        }
        case K_OUTP: {  // k-steps per head: cdiv(dh, 32), permuted order inside each
            const int dh = H / c.heads, kph = cdiv(dh, 32), d = (ks % kph) * 32 + perm_k(g4, j);
            if (d >= dh) return 0.f;
            return P[(int64_t)m * H + (ks / kph) * dh + d];
        }
        case K_TF_W1: {
            const int cg = c.FFN / c.t_groups, tpg = cdiv(cg, 32) * 2, grp = mt / tpg, ci = (mt % tpg) * 16 + l15;
            if (ci >= cg) return 0.f;
            return P[(int64_t)(grp * cg + ci) * H + knat];
        }
        case K_TF_C1: case K_TF_C2: case K_TF_C3: {
            const int cg = c.FFN / c.t_groups, pc = cg / 4, p = ks * 8 + g4 * 2 + (j >> 2);
            if (m >= cg || p >= c.t_ks * pc) return 0.f;
            const int tap = p / pc, i = (p % pc) * 4 + (j & 3);
            return P[((int64_t)(nb * cg + m) * cg + i) * c.t_ks + tap];
        }
        case K_TF_W2: {
            const int cg = c.FFN / c.t_groups, kpg = cdiv(cg, 32), d = (ks % kpg) * 32 + perm_k(g4, j);
            if (d >= cg) return 0.f;
            return P[(int64_t)m * c.FFN + (ks / kpg) * cg + d];
        }
        // ---- transposed (data-gradient) operands ----
        case K_DEC_T:
            return knat < c.C_out ? P[(int64_t)knat * H + m] : 0.f;
        case K_FC1_T: case K_FC2_T: {
            // du[f][i] = sum_{tap',o} W[o][i][ks-1-tap'] dv[f + tap' - half][o]
            const int fg = H / c.f_groups, pc = fg / 4, p = ks * 8 + g4 * 2 + (j >> 2);
            if (m >= fg || p >= c.f_ks * pc) return 0.f;
            const int tapp = p / pc, o = (p % pc) * 4 + (j & 3);
            return P[((int64_t)(nb * fg + o) * fg + m) * c.f_ks + (c.f_ks - 1 - tapp)];
        }
        case K_SQ_T:
            return knat < c.SQ ? P[(int64_t)knat * H + m] : 0.f;
        case K_FULL_T:
            return (m < c.F && knat < c.F) ? P[((int64_t)nb * c.F + knat) * c.F + m] : 0.f;
        case K_USQ_T: {
            const int ch = ks * 32 + perm_k(g4, j);
            return m < c.SQ ? P[(int64_t)ch * c.SQ + m] : 0.f;
        }
        case K_INP_T: {
            const int dh = H / c.heads, which = ks / c.heads, head = ks % c.heads, d = perm_k(g4, j);
            if (d >= dh) return 0.f;
            return P[(int64_t)(which * H + head * dh + d) * H + m];
        }
        case K_OUTP_T: {
            const int dh = H / c.heads, head = mt >> 1, d = (mt & 1) * 16 + l15;
            if (d >= dh) return 0.f;
            return P[(int64_t)knat * H + head * dh + d];
        }
        case K_TF_W1_T: {
            const int cg = c.FFN / c.t_groups, d = perm_k(g4, j);
            if (d >= cg) return 0.f;
            return P[(int64_t)(ks * cg + d) * H + m];
        }
        case K_TF_C1_T: case K_TF_C2_T: case K_TF_C3_T: {
            const int cg = c.FFN / c.t_groups, pc = cg / 4, p = ks * 8 + g4 * 2 + (j >> 2);
            if (m >= cg || p >= c.t_ks * pc) return 0.f;
            const int tapp = p / pc, o = (p % pc) * 4 + (j & 3);
            return P[((int64_t)(nb * cg + o) * cg + m) * c.t_ks + (c.t_ks - 1 - tapp)];
        }
        case K_TF_W1_TN:
            return P[(int64_t)knat * H + m];  // W1[k][m]
        case K_INP_TN:
            return P[(int64_t)knat * H + m];  // Win[k][m]
        case K_TF_W2_T: {
            const int cg = c.FFN / c.t_groups, grp = mt >> 1, ci = (mt & 1) * 16 + l15;
            if (ci >= cg) return 0.f;
            return P[(int64_t)knat * c.FFN + grp * cg + ci];
        }
    }
    return 0.f;
}

// per-kind source / destination offsets of one layer, resolved on the host (the offset arithmetic loops
// over layers and kinds: never inside a kernel)
struct PackTable {
    long long src[NUM_PACK_KINDS], src2[NUM_PACK_KINDS], dst[NUM_PACK_KINDS];
    int skip[NUM_PACK_KINDS];
};

#define PACK_L_MAX 32
// all layers in one launch (blockIdx.z = layer): the per-layer tables differ by constant strides — except the LinearGroup weight, which
// layers share from full_share on (its per-layer source offsets travel explicitly)
struct PackTableL {
    PackTable t0;  // first layer of the launch
    long long dst_stride;
    // source offset of layer l relative to the first: lbase[l] for the parameters in front of the LinearGroup weight in a layer's block,
    // lbase[l] + ladj[l] behind it (layers beyond full_share do not own one: their blocks are shorter), full_src[l] for the LinearGroup itself
    long long lbase[PACK_L_MAX], ladj[PACK_L_MAX], full_src[PACK_L_MAX];
    int cls[NUM_PACK_KINDS];
};

template <class T>
__global__ void pack_kernel(nbss_cfg c, PackTableL tb, const float* __restrict__ P, T* __restrict__ out) {
    const int kind = blockIdx.y, layer = blockIdx.z;
    const bool global = pack_is_global(kind);
    if (tb.t0.skip[kind] || (global && layer != 0)) return;
    const PackGeom g = pack_geom(c, kind);
    const int64_t n = (int64_t)g.NB * g.MT * g.KS * 512;
    T* dst = out + tb.t0.dst[kind] + (global ? 0 : layer * tb.dst_stride);
    const bool full = kind == K_FULL || kind == K_FULL_T;
    const long long rel = global ? 0 : tb.lbase[layer] + (tb.cls[kind] ? tb.ladj[layer] : 0);
    const float* W = P + (full ? tb.full_src[layer] : tb.t0.src[kind] + rel);
    const float* W2 = P + (tb.t0.src2[kind] >= 0 ? tb.t0.src2[kind] + rel : 0);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
        int64_t r = e >> 9;
        const int ks = (int)(r % g.KS);
        r /= g.KS;
        const int mt = (int)(r % g.MT), nb = (int)(r / g.MT);
        store1(dst + e, pack_value(c, W, W2, kind, nb, mt, ks, lane, j));
    }
}

int pack_params_impl(const nbss_cfg& c, const float* params, void* packed, hipStream_t stream) {
    const bool small = c.H == 96 && c.FFN == 192 && c.SQ == 8 && c.heads == 4;
    ProfScope ps(PK_PACK, stream);
    auto table = [&](int layer) {
        PackTable tb;
        for (int k = 0; k < NUM_PACK_KINDS; ++k) {
            // (geometries other than SpatialNet-small are forward only: their backward fragment kinds are never read)
            tb.skip[k] = (pack_is_global(k) && layer != 0) || (!small && k >= K_DEC_T);
            tb.src[k] = pack_src_base(c, k, layer);
            tb.src2[k] = pack_src2_base(c, k, layer);
            tb.dst[k] = pack_off(c, layer, k);
        }
        return tb;
    };
    // all layers in one launch (it was one launch per layer: 8 x 23 us of a 6 ms training step at batch 2).  The relative offsets are a
    // property of layout.h: checked, not assumed — anything else (or more than PACK_L_MAX layers) falls back to one launch per layer
    const long long dstride = pack_layer_numel(c);
    bool uniform = c.L <= PACK_L_MAX;
    const PackTable first = table(0);
    PackTableL tl;
    for (int k = 0; k < NUM_PACK_KINDS; ++k) tl.cls[k] = 0;
    for (int l = 0; l < c.L && uniform; ++l) {
        const PackTable t = table(l);
        tl.lbase[l] = t.src[K_FC1] - first.src[K_FC1];
        tl.ladj[l] = (t.src[K_TF_W2] - first.src[K_TF_W2]) - tl.lbase[l];
        tl.full_src[l] = t.src[K_FULL];
        for (int k = 0; k < NUM_PACK_KINDS; ++k) {
            if (pack_is_global(k) || first.skip[k] || k == K_FULL || k == K_FULL_T) continue;
            const long long rel = t.src[k] - first.src[k];
            if (rel == tl.lbase[l] && (!tl.cls[k] || tl.ladj[l] == 0)) {
            } else if (rel == tl.lbase[l] + tl.ladj[l]) {
                tl.cls[k] = 1;
            } else {
                uniform = false;
            }
            if (t.dst[k] != first.dst[k] + l * dstride || (t.src2[k] >= 0 && t.src2[k] - first.src2[k] != rel)) uniform = false;
        }
    }
    // (a kind classified "in front" on an early layer and "behind" later cannot happen: the class is a property of the parameter order; a
    //  second pass confirms every layer against the final classes)
    for (int l = 0; l < c.L && uniform; ++l) {
        const PackTable t = table(l);
        for (int k = 0; k < NUM_PACK_KINDS; ++k) {
            if (pack_is_global(k) || first.skip[k] || k == K_FULL || k == K_FULL_T) continue;
            if (t.src[k] - first.src[k] != tl.lbase[l] + (tl.cls[k] ? tl.ladj[l] : 0)) uniform = false;
        }
    }
    const int chunk = uniform ? c.L : 1;
    for (int l0 = 0; l0 < c.L; l0 += chunk) {
        const int nl = c.L - l0 < chunk ? c.L - l0 : chunk;
        if (!uniform) {  // one launch per layer: the layer's own table, no relative offsets
            tl.lbase[0] = 0; tl.ladj[0] = 0;
            tl.full_src[0] = pack_src_base(c, K_FULL, l0);
        }
        tl.t0 = table(l0);
        tl.dst_stride = dstride;
        dim3 grid(16, NUM_PACK_KINDS, nl), block(256);
        if (c.dtype == NBSS_BF16)
            NBSS_LAUNCH((pack_kernel<bf16_t>), grid, block, 0, stream, c, tl, params, (bf16_t*)packed);
        else
            NBSS_LAUNCH((pack_kernel<float>), grid, block, 0, stream, c, tl, params, (float*)packed);
        int e = NBSS_CHECK_LAUNCH();
        if (e) return e;
    }
    return NBSS_OK;
}

// ---- MFMA fragment self-test -------------------------------------------------------------
template <class T>
__global__ void selftest_mma_kernel(int kperm, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ D) {
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    Frag<T> a, b;
    for (int j = 0; j < 8; ++j) {
        const int k = kperm ? perm_k(g4, j) : 8 * g4 + j;
        frag_set(a, j, A[l15 * 32 + k]);  // A is 16x32 row-major
        frag_set(b, j, B[k * 16 + l15]);  // B is 32x16 row-major
    }
    f32x4 acc = mma(a, b, F32X4_ZERO);
    for (int r = 0; r < 4; ++r) D[(g4 * 4 + r) * 16 + l15] = acc[r];
}

int selftest_mma_impl(int dtype, int kperm, const float* A, const float* B, float* D, hipStream_t stream) {
    if (dtype == NBSS_BF16)
        NBSS_LAUNCH((selftest_mma_kernel<bf16_t>), dim3(1), dim3(64), 0, stream, kperm, A, B, D);
    else
        NBSS_LAUNCH((selftest_mma_kernel<float>), dim3(1), dim3(64), 0, stream, kperm, A, B, D);
    return NBSS_CHECK_LAUNCH();
}
