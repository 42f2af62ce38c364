// tconvffn.hip — narrow-band T-ConvFFN module of SpatialNetLayer (SpatialNet.py:90,102-114,61-73):
//   y = x + W2 * SiLU(gconv3(SiLU(GN(gconv2(SiLU(gconv1(SiLU(W1 * LN(x) + b1))))))))
// with W1: 1x1 H->FFN, gconv: Conv1d(FFN,FFN,k=3,groups=8,'same') along T, GN = GroupNorm(8,FFN)
// whose statistics span (24 channels x all T frames) of one (b,f) sequence, W2: 1x1 FFN->H.
//
// One workgroup (8 waves) = one (b,f) sequence; each wave owns two 16-frame strips.  The whole
// chain between the two 1x1 convs is group-local (the 8 conv groups coincide with the 8 GN
// groups), so the kernel walks the groups one at a time: only a [T+2][24] ping-pong pair of
// the current group lives in LDS (25 KB), the H-wide input strip (as LN'ed B fragments) and the
// H-wide output accumulators stay in registers for the whole kernel, and the FFN-wide (2S)
// intermediates of the reference never exist in memory.
#include "launch.h"
#include "layout.h"

#define TF_H 96
#define TF_FFN 192
#define TF_G 8
#define TF_CG 24
#define TF_TP 256
#define TF_NSW 2     // strips per wave, 8 waves
#define TF_KS (TF_H / 32)
#define TF_CKS 3     // conv k-steps: 18 pieces of 4 channels -> 3 x 8

template <class T>
NBSS_DEV void ln_strip_tf(const T* __restrict__ xr, bool valid, const float (&gam)[TF_KS][8], const float (&bet)[TF_KS][8], Frag<T> (&u)[TF_KS]) {
    const int g4 = lane_id() >> 4;
    float v[TF_KS][8];
    float sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < TF_KS; ++ks) {
        if (valid) load8(xr + ks * 32 + 8 * g4, v[ks]);
        else
#pragma unroll
            for (int j = 0; j < 8; ++j) v[ks][j] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[ks][j];
    }
    const float mean = wave_sum16(sum) * (1.0f / TF_H);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < TF_KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = v[ks][j] - mean;
            q += d * d;
        }
    const float rstd = rsqrtf(wave_sum16(q) * (1.0f / TF_H) + 1e-5f);
#pragma unroll
    for (int ks = 0; ks < TF_KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) frag_set(u[ks], j, (v[ks][j] - mean) * rstd * gam[ks][j] + bet[ks][j]);
}

// B fragments of the k=3 grouped conv for one strip: piece p -> tap p/6, channels (p%6)*4..+3,
// LDS row (t + 1) + (tap - 1) = t + tap of a [TP+2][24] buffer.
template <class T>
NBSS_DEV void conv_bfrags(const T* __restrict__ hin, int t, Frag<T> (&bq)[TF_CKS]) {
    const int g4 = lane_id() >> 4;
#pragma unroll
    for (int ks = 0; ks < TF_CKS; ++ks) {
        const int p0 = ks * 8 + 2 * g4, p1 = p0 + 1;
        if (p0 < 18) frag_load_lo(bq[ks], hin + (size_t)(t + p0 / 6) * TF_CG + (p0 % 6) * 4);
        else frag_zero_lo(bq[ks]);
        if (p1 < 18) frag_load_hi(bq[ks], hin + (size_t)(t + p1 / 6) * TF_CG + (p1 % 6) * 4);
        else frag_zero_hi(bq[ks]);
    }
}

// one grouped conv for the wave's strips: out[si][half] (C tiles: lane = frame, rows = 4 channels)
template <class T>
NBSS_DEV void conv_group(const T* __restrict__ Wc, const T* __restrict__ hin, int w, f32x4 (&out)[TF_NSW][2]) {
    const int l15 = lane_id() & 15;
    Frag<T> bq[TF_NSW][TF_CKS];
#pragma unroll
    for (int si = 0; si < TF_NSW; ++si) conv_bfrags<T>(hin, (w * TF_NSW + si) * 16 + l15, bq[si]);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        Frag<T> a[TF_CKS];
#pragma unroll
        for (int ks = 0; ks < TF_CKS; ++ks) wfrag_load(a[ks], Wc, half, TF_CKS, ks);
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
            f32x4 acc = F32X4_ZERO;
#pragma unroll
            for (int ks = 0; ks < TF_CKS; ++ks) acc = mma(a[ks], bq[si][ks], acc);
            out[si][half] = acc;
        }
    }
}

// store a pair of C tiles (24 valid channels) of frame t into an LDS [TP+2][24] buffer, zero for t >= T
template <class T>
NBSS_DEV void store_rows(T* __restrict__ h, int t, bool valid, const f32x4& lo, const f32x4& hi) {
    const int g4 = lane_id() >> 4;
    T* r = h + (size_t)(t + 1) * TF_CG;
    if (valid) {
        store4(r + 4 * g4, lo[0], lo[1], lo[2], lo[3]);
        if (g4 < 2) store4(r + 16 + 4 * g4, hi[0], hi[1], hi[2], hi[3]);
    } else {
        store4(r + 4 * g4, 0.f, 0.f, 0.f, 0.f);
        if (g4 < 2) store4(r + 16 + 4 * g4, 0.f, 0.f, 0.f, 0.f);
    }
}

template <class T>
__global__ __launch_bounds__(512) void tconvffn_fwd_kernel(nbss_cfg c, const float* __restrict__ P, int layer, const T* __restrict__ W1,
                                                           const T* __restrict__ Wc1, const T* __restrict__ Wc2, const T* __restrict__ Wc3,
                                                           const T* __restrict__ W2, const T* __restrict__ x, T* __restrict__ y) {
    NBSS_LDS(smem);
    T* ha = reinterpret_cast<T*>(smem);                    // [TP+2][24]
    T* hb = ha + (TF_TP + 2) * TF_CG;                      // [TP+2][24]
    float* red = reinterpret_cast<float*>(hb + (TF_TP + 2) * TF_CG);  // [8 waves][2]
    const int T_ = c.T;
    const int bf = blockIdx.x;
    const int tid = threadIdx.x, lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const T* xb = x + (size_t)bf * T_ * TF_H;
    T* yb = y + (size_t)bf * T_ * TF_H;
    const float* lnw = P + param_off(c, layer, P_TF_LN_W);
    const float* lnb = P + param_off(c, layer, P_TF_LN_B);
    const float* b1 = P + param_off(c, layer, P_TF_B1);
    const float* cb1 = P + param_off(c, layer, P_TF_C1B);
    const float* cb2 = P + param_off(c, layer, P_TF_C2B);
    const float* cb3 = P + param_off(c, layer, P_TF_C3B);
    const float* gnw = P + param_off(c, layer, P_TF_GN_W);
    const float* gnb = P + param_off(c, layer, P_TF_GN_B);
    const float* b2 = P + param_off(c, layer, P_TF_B2);

    // halo rows (t = -1 and t = TP) are never written by the strips: zero them once
    if (tid < TF_CG) {
        store1(ha + tid, 0.f);
        store1(hb + tid, 0.f);
        store1(ha + (size_t)(TF_TP + 1) * TF_CG + tid, 0.f);
        store1(hb + (size_t)(TF_TP + 1) * TF_CG + tid, 0.f);
    }

    Frag<T> u[TF_NSW][TF_KS];
    {
        float gam[TF_KS][8], bet[TF_KS][8];
#pragma unroll
        for (int ks = 0; ks < TF_KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gam[ks][j] = lnw[ks * 32 + 8 * g4 + j];
                bet[ks][j] = lnb[ks * 32 + 8 * g4 + j];
            }
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
            const int t = (w * TF_NSW + si) * 16 + l15;
            ln_strip_tf<T>(xb + (size_t)t * TF_H, t < T_, gam, bet, u[si]);
        }
    }
    f32x4 yacc[TF_NSW][TF_H / 16];
#pragma unroll
    for (int si = 0; si < TF_NSW; ++si)
#pragma unroll
        for (int mt = 0; mt < TF_H / 16; ++mt) yacc[si][mt] = F32X4_ZERO;

    int tt[TF_NSW];
    bool tv[TF_NSW];
#pragma unroll
    for (int si = 0; si < TF_NSW; ++si) {
        tt[si] = (w * TF_NSW + si) * 16 + l15;
        tv[si] = tt[si] < T_;
    }
    const int d0 = 4 * g4, d1 = 16 + 4 * g4;
    const bool v1 = g4 < 2;  // second tile holds channels 16..23 only

    for (int gr = 0; gr < TF_G; ++gr) {
        const int cbase = gr * TF_CG;
        f32x4 ct[TF_NSW][2];
        // (a) h1 = SiLU(W1_g LN(x) + b1_g) -> ha
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            Frag<T> a[TF_KS];
#pragma unroll
            for (int ks = 0; ks < TF_KS; ++ks) wfrag_load(a[ks], W1, gr * 2 + half, TF_KS, ks);
#pragma unroll
            for (int si = 0; si < TF_NSW; ++si) {
                f32x4 acc = F32X4_ZERO;
#pragma unroll
                for (int ks = 0; ks < TF_KS; ++ks) acc = mma(a[ks], u[si][ks], acc);
                ct[si][half] = acc;
            }
        }
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ct[si][0][r] = silu_f(ct[si][0][r] + b1[cbase + d0 + r]);
                ct[si][1][r] = v1 ? silu_f(ct[si][1][r] + b1[cbase + d1 + r]) : 0.f;
            }
            store_rows<T>(ha, tt[si], tv[si], ct[si][0], ct[si][1]);
        }
        __syncthreads();
        // (b) h2 = SiLU(gconv1(h1)) -> hb
        conv_group<T>(Wc1 + (size_t)gr * 2 * TF_CKS * 512, ha, w, ct);
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ct[si][0][r] = silu_f(ct[si][0][r] + cb1[cbase + d0 + r]);
                ct[si][1][r] = v1 ? silu_f(ct[si][1][r] + cb1[cbase + d1 + r]) : 0.f;
            }
            store_rows<T>(hb, tt[si], tv[si], ct[si][0], ct[si][1]);
        }
        __syncthreads();
        // (c) h3 = gconv2(h2); GroupNorm over (24 ch x T) ; h4 = SiLU(GN(h3)) -> ha
        conv_group<T>(Wc2 + (size_t)gr * 2 * TF_CKS * 512, hb, w, ct);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ct[si][0][r] = round_to(ct[si][0][r] + cb2[cbase + d0 + r], x);
                ct[si][1][r] = v1 ? round_to(ct[si][1][r] + cb2[cbase + d1 + r], x) : 0.f;
                if (tv[si]) {
                    s1 += ct[si][0][r] + ct[si][1][r];
                    s2 += ct[si][0][r] * ct[si][0][r] + ct[si][1][r] * ct[si][1][r];
                }
            }
        }
        s1 = wave_sum64(s1);
        s2 = wave_sum64(s2);
        if (lane == 0) {
            red[2 * w] = s1;
            red[2 * w + 1] = s2;
        }
        __syncthreads();
        float ts1 = 0.f, ts2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ts1 += red[2 * i];
            ts2 += red[2 * i + 1];
        }
        const float cnt = (float)(TF_CG * T_);
        const float mean = ts1 / cnt;
        const float var = fmaxf(ts2 / cnt - mean * mean, 0.f);
        const float rstd = rsqrtf(var + 1e-5f);
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ct[si][0][r] = silu_f((ct[si][0][r] - mean) * rstd * gnw[cbase + d0 + r] + gnb[cbase + d0 + r]);
                ct[si][1][r] = v1 ? silu_f((ct[si][1][r] - mean) * rstd * gnw[cbase + d1 + r] + gnb[cbase + d1 + r]) : 0.f;
            }
            store_rows<T>(ha, tt[si], tv[si], ct[si][0], ct[si][1]);
        }
        __syncthreads();
        // (d) h5 = SiLU(gconv3(h4)) stays in registers and feeds y += W2[:, group] h5
        conv_group<T>(Wc3 + (size_t)gr * 2 * TF_CKS * 512, ha, w, ct);
        Frag<T> h5[TF_NSW];
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ct[si][0][r] = silu_f(ct[si][0][r] + cb3[cbase + d0 + r]);
                ct[si][1][r] = v1 ? silu_f(ct[si][1][r] + cb3[cbase + d1 + r]) : 0.f;
            }
            frag_from_c2(h5[si], ct[si][0], ct[si][1]);
        }
#pragma unroll
        for (int mt = 0; mt < TF_H / 16; ++mt) {
            Frag<T> a;
            wfrag_load(a, W2, mt, TF_G, gr);
#pragma unroll
            for (int si = 0; si < TF_NSW; ++si) yacc[si][mt] = mma(a, h5[si], yacc[si][mt]);
        }
        __syncthreads();  // ha / hb / red are rewritten by the next group
    }

#pragma unroll
    for (int si = 0; si < TF_NSW; ++si) {
        if (tv[si]) {
#pragma unroll
            for (int mt = 0; mt < TF_H / 16; ++mt) {
                const int ch = 16 * mt + 4 * g4;
                float xv[4];
                load4(xb + (size_t)tt[si] * TF_H + ch, xv);
                store4(yb + (size_t)tt[si] * TF_H + ch, xv[0] + round_to(yacc[si][mt][0] + b2[ch], x), xv[1] + round_to(yacc[si][mt][1] + b2[ch + 1], x),
                       xv[2] + round_to(yacc[si][mt][2] + b2[ch + 2], x), xv[3] + round_to(yacc[si][mt][3] + b2[ch + 3], x));
            }
        }
    }
}

template <class T>
static int tconvffn_fwd_t(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st) {
    if (c.T > TF_TP) return NBSS_EUNSUPPORTED;
    const size_t lds = (size_t)2 * (TF_TP + 2) * TF_CG * sizeof(T) + 16 * sizeof(float);
    const T* pk = (const T*)packed;
    dim3 grid(c.B * c.F), block(512);
    NBSS_LAUNCH((tconvffn_fwd_kernel<T>), grid, block, lds, st, c, P, layer, pk + pack_off(c, layer, K_TF_W1), pk + pack_off(c, layer, K_TF_C1),
                pk + pack_off(c, layer, K_TF_C2), pk + pack_off(c, layer, K_TF_C3), pk + pack_off(c, layer, K_TF_W2), (const T*)x, (T*)y);
    return NBSS_CHECK_LAUNCH();
}

int tconvffn_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st) {
    return c.dtype == NBSS_BF16 ? tconvffn_fwd_t<bf16_t>(c, P, packed, layer, x, y, st) : tconvffn_fwd_t<float>(c, P, packed, layer, x, y, st);
}
