// tconvffn_g.hip — forward T-ConvFFN (SpatialNet.py:90,102-114,61-73) for geometries other than SpatialNet-small
// (geom.h: SpatialNet-large, dim_hidden 192 / dim_ffn 384 -> 48 channels per conv / GroupNorm group).  Forward only: the large
// model is served for validate / test / predict; training kernels exist for the small geometry only.
//
//   y = x + W2 * SiLU(gconv3(SiLU(GN(gconv2(SiLU(gconv1(SiLU(W1 * LN(x) + b1))))))))
//
// Same decomposition as tconvffn.hip's group-serial kernel: one workgroup = one (b,f) sequence, 16 waves x one 16-frame strip,
// the 8 conv groups walked one at a time with [TP+2][CG] row buffers in LDS, the H-wide input strip (LN'ed B fragments) and the
// H-wide output accumulators in registers.  Any T: the sequence is walked in chunks of TP - 6 frames with a 3-frame halo
// (three k=3 convolutions), twice — pass 0 accumulates the sequence-wide GroupNorm sums, pass 1 computes.  The weight fragments
// of the per-lane MFMA chains are read straight from the packed buffer (L2): the LDS weight window of the small-geometry kernels
// does not fit 87 fragments per group.
#include "launch.h"
#include "layout.h"
#include "prof.h"
#include "geom.h"

#define TG_G 8

// NW = waves per workgroup = 16-frame strips per chunk buffer: 16 (bf16 stream, <= 128 VGPRs) or 8 (fp32 stream: 8-register fragments)
template <class T, class G, int NW>
__global__ __launch_bounds__(64 * NW) void tconvffn_fwd_g_kernel(nbss_cfg c, LayerPtrs lp, const T* __restrict__ W1, const T* __restrict__ Wc1,
                                                              const T* __restrict__ Wc2, const T* __restrict__ Wc3, const T* __restrict__ W2,
                                                              const T* __restrict__ x, T* __restrict__ y) {
    constexpr int H = G::H, FFN = G::FFN, CG = G::CG, KS = G::KS;
    constexpr int TPG = (CG + 31) / 32 * 2;   // packed 16-row tiles per group (rows padded to multiples of 32)
    constexpr int OT = (CG + 15) / 16;        // tiles that hold real channels
    constexpr int KSD = (CG + 31) / 32;       // k-steps of the W2 contraction per group
    constexpr int NPC = 3 * (CG / 4);         // im2col pieces of 4 channels: (tap, channel quad)
    constexpr int KSC = (NPC + 7) / 8;        // conv k-steps
    constexpr int TG_TP = 16 * NW, HALO = 3, CH = TG_TP - 2 * HALO;
    NBSS_LDS(smem);
    T* ha = reinterpret_cast<T*>(smem);                    // [TP+2][CG]
    T* hb = ha + (TG_TP + 2) * CG;                         // [TP+2][CG]
    float* red = reinterpret_cast<float*>(hb + (TG_TP + 2) * CG);  // [NW][2]
    float* prm = red + 2 * NW;                             // [7][FFN]: b1 cb1 cb2 cb3 gnw gnb b2
    float* gstat = prm + 7 * FFN;                          // [G][2] sequence-wide GroupNorm sums
    const int T_ = c.T, bf = blockIdx.x;
    const int tid = threadIdx.x, lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const T* xb = x + (size_t)bf * T_ * H;
    T* yb = y + (size_t)bf * T_ * H;
    const float *b1 = prm, *cb1 = prm + FFN, *cb2 = prm + 2 * FFN, *cb3 = prm + 3 * FFN, *gnw = prm + 4 * FFN, *gnb = prm + 5 * FFN, *b2 = prm + 6 * FFN;
    const int nck = cdiv(T_, CH);
    if (tid < 2 * TG_G) gstat[tid] = 0.f;
    for (int i = tid; i < FFN; i += blockDim.x) {
        prm[i] = lp.p[P_TF_B1][i]; prm[FFN + i] = lp.p[P_TF_C1B][i]; prm[2 * FFN + i] = lp.p[P_TF_C2B][i]; prm[3 * FFN + i] = lp.p[P_TF_C3B][i];
        prm[4 * FFN + i] = lp.p[P_TF_GN_W][i]; prm[5 * FFN + i] = lp.p[P_TF_GN_B][i];
        if (i < H) prm[6 * FFN + i] = lp.p[P_TF_B2][i];
    }
    for (int i = tid; i < CG; i += blockDim.x) {  // halo rows (buffer rows 0 and TP + 1) are never written by the strips
        store1(ha + i, 0.f);
        store1(hb + i, 0.f);
        store1(ha + (size_t)(TG_TP + 1) * CG + i, 0.f);
        store1(hb + (size_t)(TG_TP + 1) * CG + i, 0.f);
    }
    lds_barrier();

    // store the wave's strip (OT C tiles: lane = frame, rows = 4 channels) into a row buffer, zeros for frames outside the sequence
    auto store_rows = [&](T* h, int tt, bool valid, const f32x4 (&ct)[OT]) {
        T* r = h + (size_t)(tt + 1) * CG;
#pragma unroll
        for (int i = 0; i < OT; ++i)
            if (16 * i + 4 * g4 < CG) store4(r + 16 * i + 4 * g4, keep_if(valid, ct[i][0]), keep_if(valid, ct[i][1]), keep_if(valid, ct[i][2]), keep_if(valid, ct[i][3]));
    };
    // one grouped k=3 conv for the wave's strip: B fragments = (tap, channel-quad) pieces of rows tt-1, tt, tt+1 of the buffer
    auto conv_group = [&](const T* Wc, const T* hin, int gr, int tt, f32x4 (&ct)[OT]) {
        Frag<T> bq[KSC];
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) {
            const int p0 = ks * 8 + 2 * g4, p1 = p0 + 1;
            if (p0 < NPC) frag_load_lo(bq[ks], hin + (size_t)(tt + p0 / (CG / 4)) * CG + (p0 % (CG / 4)) * 4);
            else frag_zero_lo(bq[ks]);
            if (p1 < NPC) frag_load_hi(bq[ks], hin + (size_t)(tt + p1 / (CG / 4)) * CG + (p1 % (CG / 4)) * 4);
            else frag_zero_hi(bq[ks]);
        }
#pragma unroll
        for (int i = 0; i < OT; ++i) {
            f32x4 acc = F32X4_ZERO;
#pragma unroll
            for (int ks = 0; ks < KSC; ++ks) {
                Frag<T> a;
                wfrag_load(a, Wc + (size_t)gr * TPG * KSC * 512, i, KSC, ks);
                acc = mma(a, bq[ks], acc);
            }
            ct[i] = acc;
        }
    };

    for (int pass = 0; pass < 2; ++pass)
        for (int ck = 0; ck < nck; ++ck) {
            const int t0 = ck * CH - HALO;       // sequence frame of buffer row 0
            const int tt = w * 16 + l15, tg = t0 + tt;
            const bool tv = tg >= 0 && tg < T_;  // the frame exists (rows outside hold zeros: the convolutions' zero padding)
            const bool tin = tv && tt >= HALO && tt < HALO + CH;  // this chunk's to reduce / write
            Frag<T> u[KS];
            {  // LayerNorm of the strip as natural-order B fragments
                float v[KS][8];
                float sum = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (tv) load8(xb + (ptrdiff_t)tg * H + ks * 32 + 8 * g4, v[ks]);
                    else
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[ks][j] = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) sum += v[ks][j];
                }
                const float mean = wave_sum16(sum) * (1.0f / H);
                float q = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float d = v[ks][j] - mean;
                        q += d * d;
                    }
                const float rstd = rsqrtf(wave_sum16(q) * (1.0f / H) + 1e-5f);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    float gam[8], bet[8];
                    load8(lp.p[P_TF_LN_W] + ks * 32 + 8 * g4, gam);
                    load8(lp.p[P_TF_LN_B] + ks * 32 + 8 * g4, bet);
#pragma unroll
                    for (int j = 0; j < 8; ++j) frag_set(u[ks], j, (v[ks][j] - mean) * rstd * gam[j] + bet[j]);
                }
            }
            f32x4 yacc[H / 16];
#pragma unroll
            for (int mt = 0; mt < H / 16; ++mt) yacc[mt] = F32X4_ZERO;

            for (int gr = 0; gr < TG_G; ++gr) {
                const int cbase = gr * CG;
                f32x4 ct[OT];
                // (a) h1 = SiLU(W1_g LN(x) + b1_g) -> ha
#pragma unroll
                for (int i = 0; i < OT; ++i) {
                    f32x4 acc = F32X4_ZERO;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        Frag<T> a;
                        wfrag_load(a, W1, gr * TPG + i, KS, ks);
                        acc = mma(a, u[ks], acc);
                    }
                    const int d = 16 * i + 4 * g4, dc = d < CG ? d : 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ct[i][r] = silu_f(acc[r] + b1[cbase + dc + r]);
                }
                store_rows(ha, tt, tv, ct);
                lds_barrier();
                // (b) h2 = SiLU(gconv1(h1)) -> hb
                conv_group(Wc1, ha, gr, tt, ct);
#pragma unroll
                for (int i = 0; i < OT; ++i) {
                    const int d = 16 * i + 4 * g4, dc = d < CG ? d : 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ct[i][r] = silu_f(ct[i][r] + cb1[cbase + dc + r]);
                }
                store_rows(hb, tt, tv, ct);
                lds_barrier();
                // (c) h3 = gconv2(h2); GroupNorm over (CG channels x T frames); h4 = SiLU(GN(h3)) -> ha
                conv_group(Wc2, hb, gr, tt, ct);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < OT; ++i) {
                    const int d = 16 * i + 4 * g4;
                    const bool dv = d < CG;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ct[i][r] = dv ? round_to(ct[i][r] + cb2[cbase + (dv ? d : 0) + r], x) : 0.f;
                        if (tin) {
                            s1 += ct[i][r];
                            s2 += ct[i][r] * ct[i][r];
                        }
                    }
                }
                s1 = wave_sum64(s1);
                s2 = wave_sum64(s2);
                if (lane == 0) {
                    red[2 * w] = s1;
                    red[2 * w + 1] = s2;
                }
                lds_barrier();
                if (pass == 0) {
                    if (tid == 0) {
                        float ts1 = 0.f, ts2 = 0.f;
                        for (int i = 0; i < NW; ++i) {
                            ts1 += red[2 * i];
                            ts2 += red[2 * i + 1];
                        }
                        gstat[2 * gr] += ts1;
                        gstat[2 * gr + 1] += ts2;
                    }
                } else {
                    const float cnt = (float)CG * (float)T_;
                    const float mean = gstat[2 * gr] / cnt;
                    const float var = fmaxf(gstat[2 * gr + 1] / cnt - mean * mean, 0.f);
                    const float rstd = rsqrtf(var + 1e-5f);
#pragma unroll
                    for (int i = 0; i < OT; ++i) {
                        const int d = 16 * i + 4 * g4, dc = d < CG ? d : 0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) ct[i][r] = silu_f((ct[i][r] - mean) * rstd * gnw[cbase + dc + r] + gnb[cbase + dc + r]);
                    }
                    store_rows(ha, tt, tv, ct);
                    lds_barrier();
                    // (d) h5 = SiLU(gconv3(h4)) stays in registers and feeds y += W2[:, group] h5
                    conv_group(Wc3, ha, gr, tt, ct);
                    f32x4 c5[2 * KSD];
#pragma unroll
                    for (int i = 0; i < 2 * KSD; ++i) {
                        const int d = 16 * i + 4 * g4, dc = d < CG ? d : 0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) c5[i][r] = (i < OT && d < CG) ? silu_f(ct[i < OT ? i : 0][r] + cb3[cbase + dc + r]) : 0.f;
                    }
#pragma unroll
                    for (int p = 0; p < KSD; ++p) {
                        Frag<T> h5;
                        frag_from_c2(h5, c5[2 * p], c5[2 * p + 1]);
#pragma unroll
                        for (int mt = 0; mt < H / 16; ++mt) {
                            Frag<T> a;
                            wfrag_load(a, W2, mt, TG_G * KSD, gr * KSD + p);
                            yacc[mt] = mma(a, h5, yacc[mt]);
                        }
                    }
                }
                lds_barrier();  // ha / hb / red are rewritten by the next group
            }
            if (pass == 1 && tin) {
#pragma unroll
                for (int mt = 0; mt < H / 16; ++mt) {
                    const int ch = 16 * mt + 4 * g4;
                    float xv[4];
                    load4(xb + (size_t)tg * H + ch, xv);
                    store4(yb + (size_t)tg * H + ch, xv[0] + round_to(yacc[mt][0] + b2[ch], x), xv[1] + round_to(yacc[mt][1] + b2[ch + 1], x),
                           xv[2] + round_to(yacc[mt][2] + b2[ch + 2], x), xv[3] + round_to(yacc[mt][3] + b2[ch + 3], x));
                }
            }
        }
}

template <class T, class G, int NW>
static int tconvffn_fwd_g_t(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st) {
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    const size_t lds = (size_t)2 * (16 * NW + 2) * G::CG * sizeof(T) + (2 * NW + 7 * G::FFN + 2 * TG_G) * sizeof(float);
    if (lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    const T* pk = (const T*)packed;
    int e = NBSS_SET_MAX_LDS((tconvffn_fwd_g_kernel<T, G, NW>), lds);
    if (e) return e;
    ProfScope ps(PK_TCF_F, st);
    NBSS_LAUNCH((tconvffn_fwd_g_kernel<T, G, NW>), dim3(c.B * c.F), dim3(64 * NW), lds, st, c, lp, pk + pack_off(c, layer, K_TF_W1), pk + pack_off(c, layer, K_TF_C1),
                pk + pack_off(c, layer, K_TF_C2), pk + pack_off(c, layer, K_TF_C3), pk + pack_off(c, layer, K_TF_W2), (const T*)x, (T*)y);
    return NBSS_CHECK_LAUNCH();
}

int tconvffn_fwd_large_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st) {
    return c.dtype == NBSS_BF16 ? tconvffn_fwd_g_t<bf16_t, GeoL, 8>(c, P, packed, layer, x, y, st) : tconvffn_fwd_g_t<float, GeoL, 8>(c, P, packed, layer, x, y, st);
}
// the same kernel at the small geometry (tests: the generalised tiling against the kernels that ship for SpatialNet-small)
int tconvffn_fwd_generic_small_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st) {
    return c.dtype == NBSS_BF16 ? tconvffn_fwd_g_t<bf16_t, GeoS, 16>(c, P, packed, layer, x, y, st) : tconvffn_fwd_g_t<float, GeoS, 8>(c, P, packed, layer, x, y, st);
}
