// full.hip — cross-band full-band linear module of SpatialNetLayer:
//   y = x + SiLU(Wu * LinearGroup_F(SiLU(Ws * LayerNorm_H(x) + bs)) + bu)
// (SpatialNet.py:86,129-146; LinearGroup linear_group.py:7-34: einsum("...gh,gkh->...gk") with
//  one F x F matrix per squeeze channel, shared by all layers when full_share=0).
//
// One workgroup = one (b, 8 consecutive frames) slab.  Only the squeezed activations
// s[8][tt][F] and z[F][tt][8] live in LDS (38 KB bf16); the H-wide stream is streamed through
// registers twice (squeeze pass, unsqueeze+residual pass — the second read hits L2/MALL).
//   pass 1  rows (f,tt) as MFMA N dim: LN in registers -> s = SiLU(Ws u + bs)
//   pass 2  per squeeze channel c: z[:,tt] = Wf[c] (F x F) * s[c][tt][:]   (N = 8 frames)
//   pass 3  y = x + SiLU(Wu z + bu)
#include "launch.h"
#include "layout.h"

#define FL_H 96
#define FL_SQ 8
#define FL_TT 8
#define FL_KS (FL_H / 32)
#define FL_MT (FL_H / 16)

template <class T>
__global__ __launch_bounds__(256) void full_fwd_kernel(nbss_cfg c, const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                       const float* __restrict__ bs, const float* __restrict__ bfull,
                                                       const float* __restrict__ bu, const T* __restrict__ Wsq,
                                                       const T* __restrict__ Wfull, const T* __restrict__ Wusq,
                                                       const T* __restrict__ x, T* __restrict__ y) {
    NBSS_LDS(smem);
    const int F = c.F, T_ = c.T;
    const int mtf = cdiv(F, 16), ksf = cdiv(F, 32), FK = ksf * 32, FM = mtf * 16;
    T* s = reinterpret_cast<T*>(smem);             // [SQ][TT][FK]
    T* z = s + FL_SQ * FL_TT * FK;                 // [FM][TT][SQ]
    const int ntt = cdiv(T_, FL_TT);
    const int b = blockIdx.x / ntt, t0 = (blockIdx.x % ntt) * FL_TT;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id(), nw = nthr >> 6;
    const int ntile = cdiv(F, 2);  // n-tiles of (2 freqs x 8 frames)

    for (int i = tid; i < FL_SQ * FL_TT * FK + FM * FL_TT * FL_SQ; i += nthr) store1(s + i, 0.f);
    __syncthreads();

    // ---- pass 1: LN + squeeze + SiLU ---------------------------------------------------------
    {
        Frag<T> a[FL_KS];
#pragma unroll
        for (int ks = 0; ks < FL_KS; ++ks) wfrag_load(a[ks], Wsq, 0, FL_KS, ks);
        float gam[FL_KS][8], bet[FL_KS][8];
#pragma unroll
        for (int ks = 0; ks < FL_KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gam[ks][j] = lnw[ks * 32 + 8 * g4 + j];
                bet[ks][j] = lnb[ks * 32 + 8 * g4 + j];
            }
        for (int nt = w; nt < ntile; nt += nw) {
            const int f = 2 * nt + (l15 >> 3), tt = l15 & 7;
            const bool valid = f < F && t0 + tt < T_;
            const T* xr = x + (((size_t)b * F + f) * T_ + t0 + tt) * FL_H;
            float v[FL_KS][8];
            float sum = 0.f;
#pragma unroll
            for (int ks = 0; ks < FL_KS; ++ks) {
                if (valid) load8(xr + ks * 32 + 8 * g4, v[ks]);
                else
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[ks][j] = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += v[ks][j];
            }
            const float mean = wave_sum16(sum) * (1.0f / FL_H);
            float q = 0.f;
#pragma unroll
            for (int ks = 0; ks < FL_KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = v[ks][j] - mean;
                    q += d * d;
                }
            const float rstd = rsqrtf(wave_sum16(q) * (1.0f / FL_H) + 1e-5f);
            f32x4 acc = F32X4_ZERO;
#pragma unroll
            for (int ks = 0; ks < FL_KS; ++ks) {
                Frag<T> u;
#pragma unroll
                for (int j = 0; j < 8; ++j) frag_set(u, j, (v[ks][j] - mean) * rstd * gam[ks][j] + bet[ks][j]);
                acc = mma(a[ks], u, acc);
            }
            if (valid && g4 < 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = 4 * g4 + r;
                    store1(s + ((size_t)ch * FL_TT + tt) * FK + f, silu_f(acc[r] + bs[ch]));
                }
            }
        }
    }
    __syncthreads();

    // ---- pass 2: LinearGroup along F, one F x F matrix per squeeze channel -------------------
    for (int task = w; task < FL_SQ * mtf; task += nw) {
        const int ch = task / mtf, mt = task % mtf;
        f32x4 acc = F32X4_ZERO;
        for (int ks = 0; ks < ksf; ++ks) {
            Frag<T> a, bq;
            wfrag_load(a, Wfull + (size_t)ch * mtf * ksf * 512, mt, ksf, ks);
            if (l15 < FL_TT) frag_load(bq, s + ((size_t)ch * FL_TT + l15) * FK + ks * 32 + 8 * g4);
            else frag_zero(bq);
            acc = mma(a, bq, acc);
        }
        if (l15 < FL_TT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = mt * 16 + 4 * g4 + r;
                if (k < F) store1(z + ((size_t)k * FL_TT + l15) * FL_SQ + ch, acc[r] + bfull[ch * F + k]);
            }
        }
    }
    __syncthreads();

    // ---- pass 3: unsqueeze + SiLU + residual -----------------------------------------------------
    {
        Frag<T> a[FL_MT];
#pragma unroll
        for (int mt = 0; mt < FL_MT; ++mt) wfrag_load(a[mt], Wusq, mt, 1, 0);
        for (int nt = w; nt < ntile; nt += nw) {
            const int f = 2 * nt + (l15 >> 3), tt = l15 & 7;
            const bool valid = f < F && t0 + tt < T_;
            Frag<T> bq;
            if (g4 == 0 && f < F) frag_load(bq, z + ((size_t)f * FL_TT + tt) * FL_SQ);
            else frag_zero(bq);
            const size_t go = (((size_t)b * F + f) * T_ + t0 + tt) * FL_H;
#pragma unroll
            for (int mt = 0; mt < FL_MT; ++mt) {
                f32x4 acc = mma(a[mt], bq, F32X4_ZERO);
                if (valid) {
                    const int ch = 16 * mt + 4 * g4;
                    float xv[4];
                    load4(x + go + ch, xv);
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = xv[r] + round_to(silu_f(acc[r] + bu[ch + r]), x);
                    store4(y + go + ch, o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
}

template <class T>
static int full_fwd_t(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st) {
    const int mtf = cdiv(c.F, 16), ksf = cdiv(c.F, 32);
    const size_t lds = ((size_t)FL_SQ * FL_TT * ksf * 32 + (size_t)mtf * 16 * FL_TT * FL_SQ) * sizeof(T);
    const T* pk = (const T*)packed;
    int e = NBSS_SET_MAX_LDS((full_fwd_kernel<T>), lds);
    if (e) return e;
    dim3 grid(c.B * cdiv(c.T, FL_TT)), block(256);
    NBSS_LAUNCH((full_fwd_kernel<T>), grid, block, lds, st, c, P + param_off(c, layer, P_FULL_LN_W), P + param_off(c, layer, P_FULL_LN_B),
                P + param_off(c, layer, P_SQ_B), P + param_off(c, layer, P_FULL_B), P + param_off(c, layer, P_USQ_B),
                pk + pack_off(c, layer, K_SQ), pk + pack_off(c, layer, K_FULL), pk + pack_off(c, layer, K_USQ), (const T*)x, (T*)y);
    return NBSS_CHECK_LAUNCH();
}

int full_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st) {
    return c.dtype == NBSS_BF16 ? full_fwd_t<bf16_t>(c, P, packed, layer, x, y, st) : full_fwd_t<float>(c, P, packed, layer, x, y, st);
}
