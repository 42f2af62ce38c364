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
#include "prof.h"
#include "blocks.h"
#include "wgrad.h"

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
    lds_barrier();
    for (int r = tid; r < F * TT; r += nthr) {
        const int f = r / TT, tt = r % TT;
        if (t0 + tt < T_) ln_row_inplace(u + (size_t)(f + 2) * ROW + tt * FC_H, lnw, lnb);
    }
    lds_barrier();

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
    lds_barrier();  // everyone is done reading u

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
    lds_barrier();

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


// ---------------------------------------------------------------------------------------------
// Backward (data gradient): one workgroup = one (b,t) frame with the whole F axis in LDS.  Here a
// wave owns frequency tiles (16 rows of F) and walks all 8 groups, so a lane ends up holding all 96
// channels of "its" frequency (12g + 4(l>>4) + r, lanes 48..63 idle in the epilogues) and PReLU',
// the transposed conv and the LayerNorm backward run without leaving registers.  The conv weight
// gradient is contracted by wgrad.hip from dv (emitted here) and LN(x).
template <class T>
NBSS_DEV void fconv_bfrag(Frag<T>& bq, const T* __restrict__ u, int f, int ch0, int ks) {
    const int g4 = lane_id() >> 4;
    const int p0 = ks * 8 + 2 * g4, p1 = p0 + 1;
    frag_load_lo(bq, u + (size_t)(f + p0 / 3) * FC_H + ch0 + (p0 % 3) * 4);
    if (p1 < 15) frag_load_hi(bq, u + (size_t)(f + p1 / 3) * FC_H + ch0 + (p1 % 3) * 4);
    else frag_zero_hi(bq);
}

template <class T>
__global__ __launch_bounds__(256) void fconv_bwd_kernel(nbss_cfg c, const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                        const float* __restrict__ cb, const float* __restrict__ slope, float* __restrict__ part,
                                                        const T* __restrict__ Wp,
                                                        const T* __restrict__ WpT, const T* __restrict__ x, const T* __restrict__ dy,
                                                        T* __restrict__ dx, float* __restrict__ stats, T* __restrict__ dvout) {
    NBSS_LDS(smem);
    const int F = c.F, T_ = c.T;
    const int b = blockIdx.x / T_, t = blockIdx.x % T_;
    const int mtf = cdiv(F, 16), FP = mtf * 16 + 4;
    T* u = reinterpret_cast<T*>(smem);       // [FP][H]  LN(x), rows f+2
    T* dvb = u + (size_t)FP * FC_H;          // [FP][H]  dv, rows f+2
    float* aff = reinterpret_cast<float*>(dvb + (size_t)FP * FC_H);  // [3H] LN weight | LN bias | PReLU slope gradient sums
    T* wl = reinterpret_cast<T*>(aff + 3 * FC_H);  // conv and transposed-conv weight fragments (2 x 16), shared by the 4 waves
    constexpr bool STAGE_W = sizeof(T) == 2;  // the fp32 stream has no LDS room left: it keeps reading the packed buffer
    const T* wc = STAGE_W ? wl : Wp;
    const T* wct = STAGE_W ? wl + 16 * 512 : WpT;
    if (STAGE_W) {
        constexpr int VNW = 16 / sizeof(T);
        for (int v = threadIdx.x; v < 16 * 512 / VNW; v += blockDim.x) {
            *reinterpret_cast<u32x4*>(wl + (size_t)v * VNW) = *reinterpret_cast<const u32x4*>(Wp + (size_t)v * VNW);
            *reinterpret_cast<u32x4*>(wl + 16 * 512 + (size_t)v * VNW) = *reinterpret_cast<const u32x4*>(WpT + (size_t)v * VNW);
        }
    }
    for (int i = threadIdx.x; i < 3 * FC_H; i += blockDim.x) aff[i] = 0.f;
    constexpr int VN = VecOf<T>::N;
    constexpr int VPR = FC_H / VN;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id(), nw = nthr >> 6;
    const bool cvalid = g4 < 3;  // lanes 48..63 hold the 4 padding rows of every 12-channel group

    for (int i = tid; i < FP * VPR; i += nthr) {
        const int rr = i / VPR, off = (i % VPR) * VN, f = rr - 2;
        if (f >= 0 && f < F) vec_copy(u + (size_t)rr * FC_H + off, x + (((size_t)b * F + f) * T_ + t) * FC_H + off);
        else vec_zero(u + (size_t)rr * FC_H + off);
        vec_zero(dvb + (size_t)rr * FC_H + off);
    }
    lds_barrier();
    for (int f = tid; f < F; f += nthr) ln_row_inplace(u + (size_t)(f + 2) * FC_H, lnw, lnb);
    lds_barrier();

    float dsl[FC_G][4];
#pragma unroll
    for (int g = 0; g < FC_G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) dsl[g][r] = 0.f;

    // ---- conv forward recompute, PReLU', dv ----
    for (int ft = w; ft < mtf; ft += nw) {
        const int f = ft * 16 + l15;
        const bool fvalid = f < F;
        const size_t n = ((size_t)b * F + f) * T_ + t;
#pragma unroll
        for (int g = 0; g < FC_G; ++g) {
            f32x4 acc = F32X4_ZERO;
#pragma unroll
            for (int ks = 0; ks < FC_KS; ++ks) {
                Frag<T> a, bq;
                wfrag_load(a, wc, g, FC_KS, ks);
                fconv_bfrag<T>(bq, u, f, g * FC_CG, ks);
                acc = mma(a, bq, acc);
            }
            if (fvalid && cvalid) {
                const int ch = g * FC_CG + 4 * g4;
                float dyv[4], dv[4];
                load4(dy + n * FC_H + ch, dyv);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[r] + cb[ch + r];
                    dv[r] = v > 0.f ? dyv[r] : slope[ch + r] * dyv[r];
                    if (v <= 0.f) dsl[g][r] += dyv[r] * v;
                }
                store4(dvb + (size_t)(f + 2) * FC_H + ch, dv[0], dv[1], dv[2], dv[3]);
                store4(dvout + n * FC_H + ch, dv[0], dv[1], dv[2], dv[3]);
            }
        }
    }
    lds_barrier();

    // ---- transposed conv -> du, LayerNorm backward, residual ----
    float dlw[FC_G][4], dlb[FC_G][4];
#pragma unroll
    for (int g = 0; g < FC_G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) dlw[g][r] = dlb[g][r] = 0.f;
    for (int ft = w; ft < mtf; ft += nw) {
        const int f = ft * 16 + l15;
        const bool valid = f < F && cvalid;
        const size_t n = ((size_t)b * F + f) * T_ + t;
        f32x4 du[FC_G];
#pragma unroll
        for (int g = 0; g < FC_G; ++g) {
            f32x4 acc = F32X4_ZERO;
#pragma unroll
            for (int ks = 0; ks < FC_KS; ++ks) {
                Frag<T> a, bq;
                wfrag_load(a, wct, g, FC_KS, ks);
                fconv_bfrag<T>(bq, dvb, f, g * FC_CG, ks);
                acc = mma(a, bq, acc);
            }
            du[g] = acc;
        }
        float xv[FC_G][4];
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < FC_G; ++g) {
            if (valid) load4(x + n * FC_H + g * FC_CG + 4 * g4, xv[g]);
            else xv[g][0] = xv[g][1] = xv[g][2] = xv[g][3] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) sum += xv[g][r];
        }
        const float mean = wave_sum16(sum) * (1.0f / FC_H);
        float q = 0.f;
#pragma unroll
        for (int g = 0; g < FC_G; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xv[g][r] = valid ? xv[g][r] - mean : 0.f;
                q += xv[g][r] * xv[g][r];
            }
        const float rstd = rsqrtf(wave_sum16(q) * (1.0f / FC_H) + 1e-5f);
        if (f < F && g4 == 0) {
            stats[n * 2] = mean;
            stats[n * 2 + 1] = rstd;
        }
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int g = 0; g < FC_G; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xv[g][r] *= rstd;
                const float dv = valid ? du[g][r] : 0.f;
                dlw[g][r] += dv * xv[g][r];
                dlb[g][r] += dv;
                du[g][r] = valid ? dv * lnw[g * FC_CG + 4 * g4 + r] : 0.f;
                m1 += du[g][r];
                m2 += du[g][r] * xv[g][r];
            }
        m1 = wave_sum16(m1) * (1.0f / FC_H);
        m2 = wave_sum16(m2) * (1.0f / FC_H);
        if (valid) {
#pragma unroll
            for (int g = 0; g < FC_G; ++g) {
                const int ch = g * FC_CG + 4 * g4;
                float dyv[4], o[4];
                load4(dy + n * FC_H + ch, dyv);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = dyv[r] + rstd * (du[g][r] - m1 - xv[g][r] * m2);
                store4(dx + n * FC_H + ch, o[0], o[1], o[2], o[3]);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < FC_G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = sum_l15_(dlw[g][r]), bb = sum_l15_(dlb[g][r]), s2 = sum_l15_(dsl[g][r]);
            if (l15 == 0 && cvalid) {
                const int ch = g * FC_CG + 4 * g4 + r;
                atomicAdd(aff + ch, a);
                atomicAdd(aff + FC_H + ch, bb);
                atomicAdd(aff + 2 * FC_H + ch, s2);
            }
        }
    lds_barrier();
    for (int i = threadIdx.x; i < 3 * FC_H; i += blockDim.x) part[(size_t)blockIdx.x * 3 * FC_H + i] = aff[i];
}

template <class T>
static int fconv_bwd_t(const nbss_cfg& c, const float* P, float* part, const void* packed, int layer, int which, const void* x, const void* dy, void* dx,
                       float* stats, void* dv, hipStream_t st) {
    const int mtf = cdiv(c.F, 16);
    if (mtf > FC_MTF_MAX) return NBSS_EUNSUPPORTED;
    const size_t lds = (size_t)2 * (mtf * 16 + 4) * FC_H * sizeof(T) + 3 * FC_H * sizeof(float) + (sizeof(T) == 2 ? (size_t)32 * 512 * sizeof(T) : 0);
    const int lw = which ? P_FC2_LN_W : P_FC1_LN_W, lb = which ? P_FC2_LN_B : P_FC1_LN_B, sl = which ? P_FC2_PRELU : P_FC1_PRELU;
    const T* pk = (const T*)packed;
    int e = NBSS_SET_MAX_LDS((fconv_bwd_kernel<T>), lds);
    if (e) return e;
    dim3 grid(c.B * c.T), block(256);
    ProfScope ps(PK_FCONV_B, st);
    NBSS_LAUNCH((fconv_bwd_kernel<T>), grid, block, lds, st, c, P + param_off(c, layer, lw), P + param_off(c, layer, lb),
                P + param_off(c, layer, which ? P_FC2_B : P_FC1_B), P + param_off(c, layer, sl), part, pk + pack_off(c, layer, which ? K_FC2 : K_FC1),
                pk + pack_off(c, layer, which ? K_FC2_T : K_FC1_T), (const T*)x, (const T*)dy, (T*)dx, stats, (T*)dv);
    return NBSS_CHECK_LAUNCH();
}

int fconv_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, int layer, int which, const void* x, const void* dy, void* dx,
                   void* ws, hipStream_t st) {
    const size_t N = (size_t)c.B * c.F * c.T;
    float* stats = (float*)ws;
    void* dv = (char*)ws + ws_align(N * 2 * sizeof(float));
    float* part = (float*)((char*)ws + ws_part_offset(c));
    int e = c.dtype == NBSS_BF16 ? fconv_bwd_t<bf16_t>(c, P, part, packed, layer, which, x, dy, dx, stats, dv, st)
                                 : fconv_bwd_t<float>(c, P, part, packed, layer, which, x, dy, dx, stats, dv, st);
    if (e) return e;
    AffSegs sg;
    sg.n = 3;
    sg.off[0] = param_off(c, layer, which ? P_FC2_LN_W : P_FC1_LN_W); sg.cnt[0] = FC_H;
    sg.off[1] = param_off(c, layer, which ? P_FC2_LN_B : P_FC1_LN_B); sg.cnt[1] = FC_H;
    sg.off[2] = param_off(c, layer, which ? P_FC2_PRELU : P_FC1_PRELU); sg.cnt[2] = FC_H;
    if ((e = affine_reduce_launch(part, c.B * c.T, sg, G, st))) return e;
    // conv weight: dW[o][i][tap] = sum_n dv[n][o] LN(x)[n + (tap-2) T][i]   (shift along F = T rows), bias = colsum(dv)
    WgradArgs a;
    a.part = (float*)((char*)ws + ws_wgpart_offset(c));
    a.mvalid = 0; a.nvalid = 0;
    a.Ntok = (int)N; a.F = c.F; a.T = c.T; a.shift_stride = c.T; a.shift_dim = 1; a.groups = c.f_groups; a.taps = c.f_ks;
    a.A = dv; a.lda = FC_H; a.MA = FC_H; a.B = x; a.ldb = FC_H; a.NB = FC_H;
    a.stats = stats; a.gamma = P + param_off(c, layer, which ? P_FC2_LN_W : P_FC1_LN_W); a.beta = P + param_off(c, layer, which ? P_FC2_LN_B : P_FC1_LN_B);
    a.dW = G + param_off(c, layer, which ? P_FC2_W : P_FC1_W); a.dbias = G + param_off(c, layer, which ? P_FC2_B : P_FC1_B);
    return wgrad_launch(a, c.dtype, st);
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
    ProfScope ps(PK_FCONV_F, st);
    NBSS_LAUNCH((fconv_fwd_kernel<T, TT>), grid, block, lds, st, c, lnw, lnb, cb, sl, Wp, (const T*)x, (T*)y);
    return NBSS_CHECK_LAUNCH();
}

int fconv_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, int which, const void* x, void* y, hipStream_t st) {
    if (c.dtype == NBSS_BF16) return fconv_fwd_t<bf16_t, 2>(c, P, packed, layer, which, x, y, st);
    return fconv_fwd_t<float, 1>(c, P, packed, layer, which, x, y, st);
}
