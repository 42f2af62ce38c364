// Register-level building blocks shared by the narrow-band / cross-band kernels (H = 96).
#pragma once
#include "common.h"

#define BK_H 96
#define BK_KS 3   // H / 32
#define BK_MT 6   // H / 16

// sum over the 16 lanes that share (lane >> 4): per-channel reduction over the 16 frames of a strip
NBSS_DEV float sum_l15_(float v) { return row_sum16(v); }

NBSS_DEV void load_ln_affine(const float* __restrict__ lnw, const float* __restrict__ lnb, float (&gam)[BK_KS][8], float (&bet)[BK_KS][8]) {
    const int g4 = lane_id() >> 4;
#pragma unroll
    for (int ks = 0; ks < BK_KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gam[ks][j] = lnw[ks * 32 + 8 * g4 + j];
            bet[ks][j] = lnb[ks * 32 + 8 * g4 + j];
        }
}

// LayerNorm of a 16-row strip held as natural-order B fragments: lane = row l&15, channels
// 32ks + 8(l>>4) + j.  (base/norm.py:11-27, eps 1e-5, fp32 statistics)
template <class T>
NBSS_DEV void ln_strip96(const T* __restrict__ xr, bool valid, const float (&gam)[BK_KS][8], const float (&bet)[BK_KS][8], Frag<T> (&u)[BK_KS]) {
    const int g4 = lane_id() >> 4;
    float v[BK_KS][8];
    float sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < BK_KS; ++ks) {
        if (valid) load8(xr + ks * 32 + 8 * g4, v[ks]);
        else
#pragma unroll
            for (int j = 0; j < 8; ++j) v[ks][j] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[ks][j];
    }
    const float mean = wave_sum16(sum) * (1.0f / BK_H);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < BK_KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = v[ks][j] - mean;
            q += d * d;
        }
    const float rstd = rsqrtf(wave_sum16(q) * (1.0f / BK_H) + 1e-5f);
#pragma unroll
    for (int ks = 0; ks < BK_KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) frag_set(u[ks], j, (v[ks][j] - mean) * rstd * gam[ks][j] + bet[ks][j]);
}

// LayerNorm backward + residual for one row per lane, entirely in registers.
//   du[mt][r] : gradient w.r.t. the LN output, C-tile layout (lane = row, channels 16mt + 4(l>>4) + r)
//   dx = dy + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = gamma * du
// Also emits (mean, rstd) of the row for the wgrad kernels and accumulates dgamma / dbeta partials.
template <class T>
NBSS_DEV void ln_bwd_row96(f32x4 (&du)[BK_MT], const T* __restrict__ xr, const T* __restrict__ dyr, T* __restrict__ dxr, float* __restrict__ stat,
                           bool valid, const float* __restrict__ lnw, float (&dlw)[BK_MT][4], float (&dlb)[BK_MT][4]) {
    const int g4 = lane_id() >> 4;
    float xv[BK_MT][4];
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < BK_MT; ++mt) {
        if (valid) load4(xr + 16 * mt + 4 * g4, xv[mt]);
        else xv[mt][0] = xv[mt][1] = xv[mt][2] = xv[mt][3] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += xv[mt][r];
    }
    const float mean = wave_sum16(sum) * (1.0f / BK_H);
    float q = 0.f;
#pragma unroll
    for (int mt = 0; mt < BK_MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xv[mt][r] -= mean;
            q += xv[mt][r] * xv[mt][r];
        }
    const float rstd = rsqrtf(wave_sum16(q) * (1.0f / BK_H) + 1e-5f);
    if (valid && g4 == 0 && stat) {
        stat[0] = mean;
        stat[1] = rstd;
    }
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < BK_MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = 16 * mt + 4 * g4 + r;
            xv[mt][r] *= rstd;  // xhat
            const float dv = valid ? du[mt][r] : 0.f;
            dlw[mt][r] += dv * xv[mt][r];
            dlb[mt][r] += dv;
            du[mt][r] = dv * lnw[ch];
            m1 += du[mt][r];
            m2 += du[mt][r] * xv[mt][r];
        }
    m1 = wave_sum16(m1) * (1.0f / BK_H);
    m2 = wave_sum16(m2) * (1.0f / BK_H);
    if (valid) {
#pragma unroll
        for (int mt = 0; mt < BK_MT; ++mt) {
            const int ch = 16 * mt + 4 * g4;
            float dv[4], o[4];
            load4(dyr + ch, dv);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = dv[r] + rstd * (du[mt][r] - m1 - xv[mt][r] * m2);
            store4(dxr + ch, o[0], o[1], o[2], o[3]);
        }
    }
}

// ---- raw C-layout row pieces (lane = row, 4 channels per 16-channel tile) for software-pipelined row loops --------------------------
// The cross-band row loops (full.hip) are bound by exposed HBM latency at one workgroup per CU: the next iteration's x / dy pieces are
// requested before the current iteration's math.  Addresses are clamped by the caller (always readable), validity is applied on use.
template <class T> struct RawC4;
template <> struct RawC4<bf16_t> { u32x2 v; };
template <> struct RawC4<float> { f32x4 v; };
NBSS_DEV void rawc_load(RawC4<bf16_t>& r, const bf16_t* p) { r.v = *reinterpret_cast<const u32x2*>(p); }
NBSS_DEV void rawc_load(RawC4<float>& r, const float* p) { r.v = *reinterpret_cast<const f32x4*>(p); }
NBSS_DEV void rawc_get(const RawC4<bf16_t>& r, float (&o)[4]) {
    o[0] = bf2f((bf16_t)(r.v[0] & 0xFFFF)); o[1] = bf2f((bf16_t)(r.v[0] >> 16));
    o[2] = bf2f((bf16_t)(r.v[1] & 0xFFFF)); o[3] = bf2f((bf16_t)(r.v[1] >> 16));
}
NBSS_DEV void rawc_get(const RawC4<float>& r, float (&o)[4]) { o[0] = r.v[0]; o[1] = r.v[1]; o[2] = r.v[2]; o[3] = r.v[3]; }
template <class T>
NBSS_DEV void rawc_load_row(RawC4<T> (&r)[BK_MT], const T* __restrict__ row) {
    const int g4 = lane_id() >> 4;
#pragma unroll
    for (int mt = 0; mt < BK_MT; ++mt) rawc_load(r[mt], row + 16 * mt + 4 * g4);
}

// ln_bwd_row96 with the x and dy pieces of the row already in registers (rawc_load_row)
template <class T>
NBSS_DEV void ln_bwd_row96_raw(f32x4 (&du)[BK_MT], const RawC4<T> (&xr)[BK_MT], const RawC4<T> (&dyr)[BK_MT], T* __restrict__ dxr, float* __restrict__ stat,
                               bool valid, const float* __restrict__ lnw, float (&dlw)[BK_MT][4], float (&dlb)[BK_MT][4]) {
    const int g4 = lane_id() >> 4;
    float xv[BK_MT][4];
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < BK_MT; ++mt) {
        rawc_get(xr[mt], xv[mt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xv[mt][r] = keep_if(valid, xv[mt][r]);
            sum += xv[mt][r];
        }
    }
    const float mean = wave_sum16(sum) * (1.0f / BK_H);
    float q = 0.f;
#pragma unroll
    for (int mt = 0; mt < BK_MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xv[mt][r] -= mean;
            q += xv[mt][r] * xv[mt][r];
        }
    const float rstd = rsqrtf(wave_sum16(q) * (1.0f / BK_H) + 1e-5f);
    if (valid && g4 == 0 && stat) {
        stat[0] = mean;
        stat[1] = rstd;
    }
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < BK_MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = 16 * mt + 4 * g4 + r;
            xv[mt][r] *= rstd;  // xhat
            const float dv = valid ? du[mt][r] : 0.f;
            dlw[mt][r] += dv * xv[mt][r];
            dlb[mt][r] += dv;
            du[mt][r] = dv * lnw[ch];
            m1 += du[mt][r];
            m2 += du[mt][r] * xv[mt][r];
        }
    m1 = wave_sum16(m1) * (1.0f / BK_H);
    m2 = wave_sum16(m2) * (1.0f / BK_H);
    if (valid) {
#pragma unroll
        for (int mt = 0; mt < BK_MT; ++mt) {
            const int ch = 16 * mt + 4 * g4;
            float dv[4], o[4];
            rawc_get(dyr[mt], dv);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = dv[r] + rstd * (du[mt][r] - m1 - xv[mt][r] * m2);
            store4(dxr + ch, o[0], o[1], o[2], o[3]);
        }
    }
}

// LayerNorm backward + residual only (no affine-gradient sums): for kernels that get dgamma / dbeta from the weight gradient of the first
// linear map instead (tailw.hip: dgamma[i] = sum_o W[o][i] D[o][i], dbeta[i] = sum_o W[o][i] db[o], D = da^T xhat)
template <class T>
NBSS_DEV void ln_bwd_row96_raw_na(f32x4 (&du)[BK_MT], const RawC4<T> (&xr)[BK_MT], const RawC4<T> (&dyr)[BK_MT], T* __restrict__ dxr, bool valid,
                                  const float* __restrict__ lnw) {
    const int g4 = lane_id() >> 4;
    float xv[BK_MT][4];
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < BK_MT; ++mt) {
        rawc_get(xr[mt], xv[mt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xv[mt][r] = keep_if(valid, xv[mt][r]);
            sum += xv[mt][r];
        }
    }
    const float mean = wave_sum16(sum) * (1.0f / BK_H);
    float q = 0.f;
#pragma unroll
    for (int mt = 0; mt < BK_MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xv[mt][r] -= mean;
            q += xv[mt][r] * xv[mt][r];
        }
    const float rstd = rsqrtf(wave_sum16(q) * (1.0f / BK_H) + 1e-5f);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < BK_MT; ++mt) {
        float gq[4];
        load4(lnw + 16 * mt + 4 * g4, gq);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xv[mt][r] *= rstd;  // xhat
            du[mt][r] = keep_if(valid, du[mt][r]) * gq[r];
            m1 += du[mt][r];
            m2 += du[mt][r] * xv[mt][r];
        }
    }
    m1 = wave_sum16(m1) * (1.0f / BK_H);
    m2 = wave_sum16(m2) * (1.0f / BK_H);
    if (valid) {
#pragma unroll
        for (int mt = 0; mt < BK_MT; ++mt) {
            const int ch = 16 * mt + 4 * g4;
            float dv[4], o[4];
            rawc_get(dyr[mt], dv);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = dv[r] + rstd * (du[mt][r] - m1 - xv[mt][r] * m2);
            store4(dxr + ch, o[0], o[1], o[2], o[3]);
        }
    }
}

// the same, also writing the row's (mean, rstd)
template <class T>
NBSS_DEV void ln_bwd_row96_raw_nas(f32x4 (&du)[BK_MT], const RawC4<T> (&xr)[BK_MT], const RawC4<T> (&dyr)[BK_MT], T* __restrict__ dxr, float* __restrict__ stat, bool valid,
                                  const float* __restrict__ lnw) {
    const int g4 = lane_id() >> 4;
    float xv[BK_MT][4];
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < BK_MT; ++mt) {
        rawc_get(xr[mt], xv[mt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xv[mt][r] = keep_if(valid, xv[mt][r]);
            sum += xv[mt][r];
        }
    }
    const float mean = wave_sum16(sum) * (1.0f / BK_H);
    float q = 0.f;
#pragma unroll
    for (int mt = 0; mt < BK_MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xv[mt][r] -= mean;
            q += xv[mt][r] * xv[mt][r];
        }
    const float rstd = rsqrtf(wave_sum16(q) * (1.0f / BK_H) + 1e-5f);
    if (valid && g4 == 0) {  // row statistics for the weight-gradient kernel's LayerNorm-on-the-fly
        stat[0] = mean;
        stat[1] = rstd;
    }
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < BK_MT; ++mt) {
        float gq[4];
        load4(lnw + 16 * mt + 4 * g4, gq);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xv[mt][r] *= rstd;  // xhat
            du[mt][r] = keep_if(valid, du[mt][r]) * gq[r];
            m1 += du[mt][r];
            m2 += du[mt][r] * xv[mt][r];
        }
    }
    m1 = wave_sum16(m1) * (1.0f / BK_H);
    m2 = wave_sum16(m2) * (1.0f / BK_H);
    if (valid) {
#pragma unroll
        for (int mt = 0; mt < BK_MT; ++mt) {
            const int ch = 16 * mt + 4 * g4;
            float dv[4], o[4];
            rawc_get(dyr[mt], dv);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = dv[r] + rstd * (du[mt][r] - m1 - xv[mt][r] * m2);
            store4(dxr + ch, o[0], o[1], o[2], o[3]);
        }
    }
}

// Small per-channel parameter gradients (LN / GN affine, PReLU slope) are summed per workgroup in LDS
// (ds_add_f32) and written as ONE partial row per workgroup; util.hip's affine_reduce folds the rows
// into the gradient buffer.  (Same-address global atomics from every wave serialise in the memory
// system and stalled the barriers of the first version: 24.7 ms -> see profiles/.)
struct AffSegs {
    int n;
    long long off[8];  // destination offsets in the flat gradient buffer
    int cnt[8];        // consecutive elements per segment; the partial row is the concatenation
};
int affine_reduce_launch(const float* part, int nwg, const AffSegs& segs, float* G, hipStream_t st);

// reduce the per-lane dgamma/dbeta partials over the 16 rows of the lane group and add them to the
// workgroup's LDS accumulators (gw/gb may also be global: then these are plain atomicAdd's)
NBSS_DEV void ln_affine_flush(float (&dlw)[BK_MT][4], float (&dlb)[BK_MT][4], float* __restrict__ gw, float* __restrict__ gb) {
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < BK_MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = sum_l15_(dlw[mt][r]), b = sum_l15_(dlb[mt][r]);
            if (l15 == 0) {
                atomicAdd(gw + 16 * mt + 4 * g4 + r, a);
                atomicAdd(gb + 16 * mt + 4 * g4 + r, b);
            }
        }
}
