// fconv.hip — cross-band frequency-convolutional module of SpatialNetLayer:
//   y = x + PReLU(Conv1d_F(LayerNorm_H(x)))       (SpatialNet.py:85,87,116-127; LN base/norm.py:11-27)
// Conv1d(H,H,k=5,groups=8,'same', zeros) runs ALONG F for every (b,t).
//
// Work decomposition: one workgroup = one (b, TT consecutive frames) slab with the whole F axis
// resident in LDS, so there is no halo re-read and no [B,T,H,F] permute copy (the reference
// makes two).  x is fetched as F chunks of TT*H contiguous elements, LayerNorm'ed in LDS, and the
// grouped conv is 8 independent GEMMs  out^T[12(16) x F] = W_g[12 x 60(64)] * im2col(u_g)[60 x F]
// on the matrix cores (form 2: weights = A, frequencies = N).  Results return through LDS so the
// residual add + store are full 16-byte coalesced accesses.
#include "launch.h"
#include "layout.h"

#define FC_H 96
#define FC_G 8
#define FC_CG 12   // channels per group
#define FC_KS 2    // 15 pieces of 4 channels -> 2 k-steps of 8 pieces
#define FC_MTF_MAX 10

template <class T> struct VecOf;
template <> struct VecOf<bf16_t> { static constexpr int N = 8; };
template <> struct VecOf<float> { static constexpr int N = 4; };

NBSS_DEV void vec_copy(bf16_t* d, const bf16_t* s) { *reinterpret_cast<u32x4*>(d) = *reinterpret_cast<const u32x4*>(s); }
NBSS_DEV void vec_copy(float* d, const float* s) { *reinterpret_cast<f32x4*>(d) = *reinterpret_cast<const f32x4*>(s); }
NBSS_DEV void vec_zero(bf16_t* d) { *reinterpret_cast<u32x4*>(d) = (u32x4){0, 0, 0, 0}; }
NBSS_DEV void vec_zero(float* d) { *reinterpret_cast<f32x4*>(d) = (f32x4){0, 0, 0, 0}; }

// LayerNorm over H=96 of one LDS-resident row, in place (fp32 statistics, eps 1e-5).
template <class T>
NBSS_DEV void ln_row_inplace(T* row, const float* __restrict__ gamma, const float* __restrict__ beta) {
    float s = 0.f;
    float v[FC_H];
#pragma unroll
    for (int i = 0; i < FC_H; i += 8) load8(row + i, v + i);
#pragma unroll
    for (int i = 0; i < FC_H; ++i) s += v[i];
    const float mean = s * (1.0f / FC_H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < FC_H; ++i) {
        const float d = v[i] - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(q * (1.0f / FC_H) + 1e-5f);
#pragma unroll
    for (int i = 0; i < FC_H; i += 4)
        store4(row + i, (v[i] - mean) * rstd * gamma[i] + beta[i], (v[i + 1] - mean) * rstd * gamma[i + 1] + beta[i + 1],
               (v[i + 2] - mean) * rstd * gamma[i + 2] + beta[i + 2], (v[i + 3] - mean) * rstd * gamma[i + 3] + beta[i + 3]);
}

template <class T, int TT>
__global__ __launch_bounds__(256) void fconv_fwd_kernel(nbss_cfg c, const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                        const float* __restrict__ cb, const float* __restrict__ slope,
                                                        const T* __restrict__ Wp, const T* __restrict__ x, T* __restrict__ y) {
    NBSS_LDS(smem);
    T* u = reinterpret_cast<T*>(smem);
    const int F = c.F, T_ = c.T;
    const int ntt = cdiv(T_, TT);
    const int b = blockIdx.x / ntt, t0 = (blockIdx.x % ntt) * TT;
    const int mtf = cdiv(F, 16), FP = mtf * 16 + 4;
    constexpr int ROW = TT * FC_H;           // elements per frequency row in LDS
    constexpr int VN = VecOf<T>::N;
    constexpr int VPR = ROW / VN;            // vectors per frequency row
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();

    // ---- phase 1: stage x (raw) into LDS rows f+2, zero halo / tail rows -------------------
    for (int i = tid; i < FP * VPR; i += nthr) {
        const int rr = i / VPR, off = (i % VPR) * VN, f = rr - 2, tt = off / FC_H;
        T* d = u + (size_t)rr * ROW + off;
        if (f >= 0 && f < F && t0 + tt < T_)
            vec_copy(d, x + (((size_t)b * F + f) * T_ + t0) * FC_H + off);
        else
            vec_zero(d);
    }
    __syncthreads();
    for (int r = tid; r < F * TT; r += nthr) {
        const int f = r / TT, tt = r % TT;
        if (t0 + tt < T_) ln_row_inplace(u + (size_t)(f + 2) * ROW + tt * FC_H, lnw, lnb);
    }
    __syncthreads();

    // ---- phase 2: grouped conv on the matrix cores -----------------------------------------
    f32x4 acc[2][TT][FC_MTF_MAX];
    Frag<T> a[2][FC_KS];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int ks = 0; ks < FC_KS; ++ks) wfrag_load(a[gi][ks], Wp, 2 * w + gi, FC_KS, ks);
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int ft = 0; ft < FC_MTF_MAX; ++ft) acc[gi][tt][ft] = F32X4_ZERO;
#pragma unroll
    for (int ft = 0; ft < FC_MTF_MAX; ++ft) {
        if (ft < mtf) {
            const int f = ft * 16 + l15;
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const int ch0 = (2 * w + gi) * FC_CG;
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
                    for (int ks = 0; ks < FC_KS; ++ks) {
                        Frag<T> bq;
                        const int p0 = ks * 8 + 2 * g4, p1 = p0 + 1;
                        // piece p -> tap p/3, channels (p%3)*4..+3 ; LDS row = f + tap
                        frag_load_lo(bq, u + (size_t)(f + p0 / 3) * ROW + tt * FC_H + ch0 + (p0 % 3) * 4);
                        if (p1 < 15) frag_load_hi(bq, u + (size_t)(f + p1 / 3) * ROW + tt * FC_H + ch0 + (p1 % 3) * 4);
                        else frag_zero_hi(bq);
                        acc[gi][tt][ft] = mma(a[gi][ks], bq, acc[gi][tt][ft]);
                    }
                }
            }
        }
    }
    __syncthreads();  // everyone is done reading u

    // ---- phase 3: bias + PReLU -> LDS [f][tt][H] ---------------------------------------------
    if (g4 < 3) {
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int ch = (2 * w + gi) * FC_CG + 4 * g4;
            float bb[4], sl[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { bb[r] = cb[ch + r]; sl[r] = slope[ch + r]; }
#pragma unroll
            for (int ft = 0; ft < FC_MTF_MAX; ++ft) {
                const int f = ft * 16 + l15;
                if (ft < mtf && f < F) {
#pragma unroll
                    for (int tt = 0; tt < TT; ++tt) {
                        float o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc[gi][tt][ft][r] + bb[r];
                            o[r] = v > 0.f ? v : sl[r] * v;
                        }
                        store4(u + (size_t)f * ROW + tt * FC_H + ch, o[0], o[1], o[2], o[3]);
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 4: residual add + coalesced store -------------------------------------------------
    for (int i = tid; i < F * VPR; i += nthr) {
        const int f = i / VPR, off = (i % VPR) * VN, tt = off / FC_H;
        if (t0 + tt >= T_) continue;
        const size_t go = (((size_t)b * F + f) * T_ + t0) * FC_H + off;
        float xv[8], yv[8];
        if (VN == 8) {
            load8(x + go, xv);
            load8(u + (size_t)f * ROW + off, yv);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = xv[j] + yv[j];
            store4(y + go, o[0], o[1], o[2], o[3]);
            store4(y + go + 4, o[4], o[5], o[6], o[7]);
        } else {
            load4(x + go, xv);
            load4(u + (size_t)f * ROW + off, yv);
            store4(y + go, xv[0] + yv[0], xv[1] + yv[1], xv[2] + yv[2], xv[3] + yv[3]);
        }
    }
}

template <class T, int TT>
static int fconv_fwd_t(const nbss_cfg& c, const float* P, const void* packed, int layer, int which, const void* x, void* y, hipStream_t st) {
    const int mtf = cdiv(c.F, 16);
    if (mtf > FC_MTF_MAX) return NBSS_EUNSUPPORTED;
    const size_t lds = (size_t)(mtf * 16 + 4) * TT * FC_H * sizeof(T);
    const float* lnw = P + param_off(c, layer, which ? P_FC2_LN_W : P_FC1_LN_W);
    const float* lnb = P + param_off(c, layer, which ? P_FC2_LN_B : P_FC1_LN_B);
    const float* cb = P + param_off(c, layer, which ? P_FC2_B : P_FC1_B);
    const float* sl = P + param_off(c, layer, which ? P_FC2_PRELU : P_FC1_PRELU);
    const T* Wp = (const T*)packed + pack_off(c, layer, which ? K_FC2 : K_FC1);
    int e = NBSS_SET_MAX_LDS((fconv_fwd_kernel<T, TT>), lds);
    if (e) return e;
    dim3 grid(c.B * cdiv(c.T, TT)), block(256);
    NBSS_LAUNCH((fconv_fwd_kernel<T, TT>), grid, block, lds, st, c, lnw, lnb, cb, sl, Wp, (const T*)x, (T*)y);
    return NBSS_CHECK_LAUNCH();
}

int fconv_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, int which, const void* x, void* y, hipStream_t st) {
    if (c.dtype == NBSS_BF16) return fconv_fwd_t<bf16_t, 2>(c, P, packed, layer, which, x, y, st);
    return fconv_fwd_t<float, 1>(c, P, packed, layer, which, x, y, st);
}
