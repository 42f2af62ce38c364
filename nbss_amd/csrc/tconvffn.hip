// tconvffn.hip — narrow-band T-ConvFFN module of SpatialNetLayer (SpatialNet.py:90,102-114,61-73):
//   y = x + W2 * SiLU(gconv3(SiLU(GN(gconv2(SiLU(gconv1(SiLU(W1 * LN(x) + b1))))))))
// with W1: 1x1 H->FFN, gconv: Conv1d(FFN,FFN,k=3,groups=8,'same') along T, GN = GroupNorm(8,FFN)
// whose statistics span (24 channels x all T frames) of one (b,f) sequence, W2: 1x1 FFN->H.
//
// One workgroup = one (b,f) sequence; a wave owns 16-frame strips (forward, bf16: 16 waves x 1 strip; backward and fp32:
// 8 waves x 2 strips).  The whole chain between the two 1x1 convs is group-local (the 8 conv groups coincide with the 8 GN
// groups), so the kernel walks the groups one at a time: only [T+2][24] row buffers of the current group (2 forward,
// 4 backward) and the group's weight fragments (double-buffered in forward) live in LDS, the H-wide input strip (as LN'ed
// B fragments) and the H-wide output accumulators stay in registers, and the FFN-wide (2S) intermediates of the reference
// never exist in memory.  Backward recomputes the forward chain per group and emits the eight wgrad operand tensors
// group-major ([G][N][24], non-temporal stores); every global read of a group is issued before that group's stores.
#include "launch.h"
#include "layout.h"
#include "prof.h"
#include "wgrad.h"
#include "blocks.h"
#include "fold.h"
#include "side.h"
#include <cstdlib>

#define TF_H 96
#define TF_FFN 192
#define TF_G 8
#define TF_CG 24
#define TF_TP 256
#define TF_NSW 2     // strips per wave of the backward kernel (8 waves); the forward kernel takes it as a template parameter
#define TF_KS (TF_H / 32)
#ifndef TF_FWD_WPS
#define TF_FWD_WPS 2
#endif
#define TF_CKS 3     // conv k-steps: 18 pieces of 4 channels -> 3 x 8
#define TF_AFF (2 * TF_FFN + 2 * TF_H)  // GN weight, GN bias, LN weight, LN bias partial sums per workgroup

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

// Weight staging for the group loops: every 16-byte load of the group's fragments (512 elements each) is in flight before the
// first LDS write (one copy loop per weight cost one exposed L2 latency each: 4 us per group when measured with phase timers).
template <class T>
struct StageSrcs {  // named members, not an array: a select between array elements becomes a per-lane scratch load
    const T *p0, *p1, *p2, *p3, *p4, *p5, *p6, *p7;
    template <int I>
    NBSS_DEV const T* get() const {
        if constexpr (I == 0) return p0;
        else if constexpr (I == 1) return p1;
        else if constexpr (I == 2) return p2;
        else if constexpr (I == 3) return p3;
        else if constexpr (I == 4) return p4;
        else if constexpr (I == 5) return p5;
        else if constexpr (I == 6) return p6;
        else return p7;
    }
};
template <class T, int NSRC, int I, int NTHR>
NBSS_DEV void stage_load1(u32x4& r, const StageSrcs<T>& srcs, size_t goff) {
    constexpr int VN = 16 / sizeof(T), VPS = 6 * 512 / VN;  // vectors per source (6 fragments)
    constexpr int s0 = (I * NTHR) / VPS < NSRC ? (I * NTHR) / VPS : NSRC - 1;  // round I touches at most 4 consecutive sources
    const int v = (int)threadIdx.x + I * NTHR;
    int off = v - s0 * VPS;
    const T* p = srcs.template get<s0>();
    if constexpr (s0 + 1 < NSRC) {
        const T* q = srcs.template get<s0 + 1>();
        if (off >= VPS) { p = q; off -= VPS; }
    }
    if constexpr (s0 + 2 < NSRC) {
        const T* q = srcs.template get<s0 + 2>();
        if (off >= VPS) { p = q; off -= VPS; }
    }
    if constexpr (s0 + 3 < NSRC) {
        const T* q = srcs.template get<s0 + 3>();
        if (off >= VPS) { p = q; off -= VPS; }
    }
    if (v < NSRC * VPS) r = *reinterpret_cast<const u32x4*>(p + goff + (size_t)off * VN);
}
template <class T, int NSRC, int I, int NV, int NTHR>
NBSS_DEV void stage_loads(u32x4 (&r)[NV], const StageSrcs<T>& srcs, size_t goff) {
    if constexpr (I < NV) {
        stage_load1<T, NSRC, I, NTHR>(r[I], srcs, goff);
        stage_loads<T, NSRC, I + 1, NV, NTHR>(r, srcs, goff);
    }
}
template <class T, int NSRC, int NTHR = 512>
struct StageRegs {
    static constexpr int VN = 16 / sizeof(T), VPS = 6 * 512 / VN, NV = (NSRC * VPS + NTHR - 1) / NTHR;
    u32x4 r[NV];
    NBSS_DEV void load(const StageSrcs<T>& srcs, size_t goff) { stage_loads<T, NSRC, 0, NV, NTHR>(r, srcs, goff); }
    NBSS_DEV void store(T* __restrict__ wl) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = (int)threadIdx.x + i * NTHR;
            if (v < NSRC * VPS) *reinterpret_cast<u32x4*>(wl + (size_t)v * VN) = r[i];
        }
    }
};
template <class T, int NSRC>
NBSS_DEV void stage_group(T* __restrict__ wl, const StageSrcs<T>& srcs, size_t goff) {
    StageRegs<T, NSRC> sr;
    sr.load(srcs, goff);
    sr.store(wl);
}
template <class T>
NBSS_DEV void lfrag(Frag<T>& f, const T* __restrict__ wl, int idx) { frag_load(f, wl + ((size_t)idx * 64 + lane_id()) * 8); }

// one grouped conv for the wave's strips: out[si][half] (C tiles: lane = frame, rows = 4 channels)
template <class T, int NSW>
NBSS_DEV void conv_group(const T* __restrict__ Wc, const T* __restrict__ hin, int w, f32x4 (&out)[NSW][2]) {
    const int l15 = lane_id() & 15;
    Frag<T> bq[NSW][TF_CKS];
#pragma unroll
    for (int si = 0; si < NSW; ++si) conv_bfrags<T>(hin, (w * NSW + si) * 16 + l15, bq[si]);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        Frag<T> a[TF_CKS];
#pragma unroll
        for (int ks = 0; ks < TF_CKS; ++ks) wfrag_load(a[ks], Wc, half, TF_CKS, ks);
#pragma unroll
        for (int si = 0; si < NSW; ++si) {
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
    store4(r + 4 * g4, keep_if(valid, lo[0]), keep_if(valid, lo[1]), keep_if(valid, lo[2]), keep_if(valid, lo[3]));
    if (g4 < 2) store4(r + 16 + 4 * g4, keep_if(valid, hi[0]), keep_if(valid, hi[1]), keep_if(valid, hi[2]), keep_if(valid, hi[3]));
}

// NSW = 16-frame strips per wave: 2 with 8 waves (fp32), 1 with 16 waves (bf16: 4 waves per SIMD to hide the LDS / MFMA chains)
//
// LONG (sequences beyond TP frames, forward only): the sequence is walked in chunks of TP - 6 frames with a 3-frame halo on
// either side (three k=3 convolutions), twice: pass 0 runs every chunk up to conv2 and accumulates the GroupNorm sums of
// the 8 groups in LDS, pass 1 recomputes the chain with the sequence-wide statistics and writes y.  Exact (no windowing
// approximation), any T, at twice the W1/conv1/conv2 work.
template <class T, int NSW, bool LONG>
__global__ __launch_bounds__(64 * 16 / NSW, NSW == 1 ? 4 : TF_FWD_WPS) void tconvffn_fwd_kernel(nbss_cfg c, LayerPtrs lp, const float* __restrict__ P, int layer, const T* __restrict__ W1,
                                                           const T* __restrict__ Wc1, const T* __restrict__ Wc2, const T* __restrict__ Wc3,
                                                           const T* __restrict__ W2, const T* __restrict__ x, T* __restrict__ y) {
    NBSS_LDS(smem);
    T* ha = reinterpret_cast<T*>(smem);                    // [TP+2][24]
    T* hb = ha + (TF_TP + 2) * TF_CG;                      // [TP+2][24]
    constexpr int NW = 16 / NSW, NTHR = 64 * NW;
    float* red = reinterpret_cast<float*>(hb + (TF_TP + 2) * TF_CG);  // [NW waves][2]
    T* wl0 = reinterpret_cast<T*>(red + 2 * NW);  // 2 x this group's weights: W1 | conv1 | conv2 | conv3 | W2, 6 fragments each
    float* prm = reinterpret_cast<float*>(wl0 + (sizeof(T) == 2 ? 2 : 1) * 30 * 512);  // [7][FFN]: b1 cb1 cb2 cb3 gnw gnb b2 (per-lane global reads of these sat in every dependency chain)
    const int T_ = c.T;
    const int bf = blockIdx.x;
    const int tid = threadIdx.x, lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const T* xb = x + (size_t)bf * T_ * TF_H;
    T* yb = y + (size_t)bf * T_ * TF_H;
    const float* lnw = lp.p[P_TF_LN_W];
    const float* lnb = lp.p[P_TF_LN_B];
    const float* b1_g = lp.p[P_TF_B1];
    const float* b1 = prm + 0 * TF_FFN;
    const float* cb1_g = lp.p[P_TF_C1B];
    const float* cb1 = prm + 1 * TF_FFN;
    const float* cb2_g = lp.p[P_TF_C2B];
    const float* cb2 = prm + 2 * TF_FFN;
    const float* cb3_g = lp.p[P_TF_C3B];
    const float* cb3 = prm + 3 * TF_FFN;
    const float* gnw_g = lp.p[P_TF_GN_W];
    const float* gnw = prm + 4 * TF_FFN;
    const float* gnb_g = lp.p[P_TF_GN_B];
    const float* gnb = prm + 5 * TF_FFN;
    const float* b2_g = lp.p[P_TF_B2];
    const float* b2 = prm + 6 * TF_FFN;
    float* gstat = prm + 7 * TF_FFN;  // LONG: [G][2] sequence-wide GroupNorm sums
    constexpr int HALO = LONG ? 3 : 0, CH = TF_TP - 2 * HALO;
    const int nck = LONG ? cdiv(T_, CH) : 1;
    if (LONG && tid < 2 * TF_G) gstat[tid] = 0.f;

    for (int i = tid; i < TF_FFN; i += blockDim.x) {
        prm[i] = b1_g[i]; prm[TF_FFN + i] = cb1_g[i]; prm[2 * TF_FFN + i] = cb2_g[i]; prm[3 * TF_FFN + i] = cb3_g[i];
        prm[4 * TF_FFN + i] = gnw_g[i]; prm[5 * TF_FFN + i] = gnb_g[i];
        if (i < TF_H) prm[6 * TF_FFN + i] = b2_g[i];
    }
    // halo rows (t = -1 and t = TP) are never written by the strips: zero them once
    if (tid < TF_CG) {
        store1(ha + tid, 0.f);
        store1(hb + tid, 0.f);
        store1(ha + (size_t)(TF_TP + 1) * TF_CG + tid, 0.f);
        store1(hb + (size_t)(TF_TP + 1) * TF_CG + tid, 0.f);
    }

    const int d0 = 4 * g4, d1 = 16 + 4 * g4;
    const bool v1 = g4 < 2;  // second tile holds channels 16..23 only
    for (int pass = LONG ? 0 : 1; pass < 2; ++pass)
    for (int ck = 0; ck < nck; ++ck) {
    const int t0 = ck * CH - HALO;  // sequence frame of buffer row 0
    int tt[NSW];   // buffer row of the strip's frame
    bool tv[NSW];  // the frame exists (rows outside the sequence hold zeros: the convolutions' zero padding)
    bool tin[NSW]; // the frame is this chunk's to reduce / write (LONG: the halo rows belong to the neighbours)
#pragma unroll
    for (int si = 0; si < NSW; ++si) {
        tt[si] = (w * NSW + si) * 16 + l15;
        const int tg = t0 + tt[si];
        tv[si] = tg >= 0 && tg < T_;
        tin[si] = tv[si] && (!LONG || (tt[si] >= HALO && tt[si] < HALO + CH));
    }
    Frag<T> u[NSW][TF_KS];
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
        for (int si = 0; si < NSW; ++si) {
            ln_strip_tf<T>(xb + (ptrdiff_t)(t0 + tt[si]) * TF_H, tv[si], gam, bet, u[si]);
        }
    }
    f32x4 yacc[NSW][TF_H / 16];
#pragma unroll
    for (int si = 0; si < NSW; ++si)
#pragma unroll
        for (int mt = 0; mt < TF_H / 16; ++mt) yacc[si][mt] = F32X4_ZERO;

    constexpr int FVN = 16 / sizeof(T), FVPF = 512 / FVN, FNV2 = (6 * FVPF + NTHR - 1) / NTHR;  // W2: 6 strided fragments
    StageRegs<T, 4, NTHR> fw;
    u32x4 fw2[FNV2];
    const StageSrcs<T> fsrc = {W1, Wc1, Wc2, Wc3, nullptr, nullptr, nullptr, nullptr};
    auto fwd_wloads = [&](int g) {
#pragma unroll
        for (int i = 0; i < FNV2; ++i) {  // W2: one fragment per output tile, strided by the group count
            const int v = tid + i * NTHR, mt = v / FVPF, off = v % FVPF;
            if (v < 6 * FVPF) fw2[i] = *reinterpret_cast<const u32x4*>(W2 + (size_t)(mt * TF_G + g) * 512 + (size_t)off * FVN);
        }
        fw.load(fsrc, (size_t)g * 6 * 512);
    };
    auto fwd_wstore = [&](T* dst) {
        fw.store(dst);
#pragma unroll
        for (int i = 0; i < FNV2; ++i) {
            const int v = tid + i * NTHR;
            if (v < 6 * FVPF) *reinterpret_cast<u32x4*>(dst + 24 * 512 + (size_t)v * FVN) = fw2[i];
        }
    };
    for (int gr = 0; gr < TF_G; ++gr) {
        const int cbase = gr * TF_CG;
        f32x4 ct[NSW][2];
        // the group's 30 weight fragments go through LDS once per workgroup (8 waves share them; the packed buffer is
        // regularly evicted from L2 by the activation traffic, and per-wave global fragment loads sat in every MFMA chain).
        // Two LDS buffers: group g+1's fragments are requested here and written to the other buffer at the end of group g.
        // (the fp32 stream has LDS room for one buffer only and stages at the top of every group)
        constexpr bool DB = sizeof(T) == 2;
        T* wl = wl0 + (size_t)(DB ? (gr & 1) : 0) * 30 * 512;
        if (!DB || gr == 0) {
            fwd_wloads(gr);
            fwd_wstore(wl);
            lds_barrier();
        }
        if (DB && gr + 1 < TF_G) fwd_wloads(gr + 1);
        // (a) h1 = SiLU(W1_g LN(x) + b1_g) -> ha
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            Frag<T> a[TF_KS];
#pragma unroll
            for (int ks = 0; ks < TF_KS; ++ks) lfrag<T>(a[ks], wl, half * 3 + ks);
#pragma unroll
            for (int si = 0; si < NSW; ++si) {
                f32x4 acc = F32X4_ZERO;
#pragma unroll
                for (int ks = 0; ks < TF_KS; ++ks) acc = mma(a[ks], u[si][ks], acc);
                ct[si][half] = acc;
            }
        }
#pragma unroll
        for (int si = 0; si < NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ct[si][0][r] = silu_f(ct[si][0][r] + b1[cbase + d0 + r]);
                ct[si][1][r] = keep_if(v1, silu_f(ct[si][1][r] + b1[cbase + d1 + r]));
            }
            store_rows<T>(ha, tt[si], tv[si], ct[si][0], ct[si][1]);
        }
        lds_barrier();
        // (b) h2 = SiLU(gconv1(h1)) -> hb
        conv_group<T>(wl + 6 * 512, ha, w, ct);
#pragma unroll
        for (int si = 0; si < NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ct[si][0][r] = silu_f(ct[si][0][r] + cb1[cbase + d0 + r]);
                ct[si][1][r] = keep_if(v1, silu_f(ct[si][1][r] + cb1[cbase + d1 + r]));
            }
            store_rows<T>(hb, tt[si], tv[si], ct[si][0], ct[si][1]);
        }
        lds_barrier();
        // (c) h3 = gconv2(h2); GroupNorm over (24 ch x T) ; h4 = SiLU(GN(h3)) -> ha
        conv_group<T>(wl + 12 * 512, hb, w, ct);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int si = 0; si < NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ct[si][0][r] = round_to(ct[si][0][r] + cb2[cbase + d0 + r], x);
                ct[si][1][r] = v1 ? round_to(ct[si][1][r] + cb2[cbase + d1 + r], x) : 0.f;
                if (tin[si]) {
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
        lds_barrier();
        float ts1 = 0.f, ts2 = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            ts1 += red[2 * i];
            ts2 += red[2 * i + 1];
        }
        if (LONG) {
            if (pass == 0) {
                if (tid == 0) {
                    gstat[2 * gr] += ts1;
                    gstat[2 * gr + 1] += ts2;
                }
            } else {
                ts1 = gstat[2 * gr];
                ts2 = gstat[2 * gr + 1];
            }
        }
        if (pass == 1) {
        const float cnt = (float)(TF_CG * T_);
        const float mean = ts1 / cnt;
        const float var = fmaxf(ts2 / cnt - mean * mean, 0.f);
        const float rstd = rsqrtf(var + 1e-5f);
#pragma unroll
        for (int si = 0; si < NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ct[si][0][r] = silu_f((ct[si][0][r] - mean) * rstd * gnw[cbase + d0 + r] + gnb[cbase + d0 + r]);
                ct[si][1][r] = keep_if(v1, silu_f((ct[si][1][r] - mean) * rstd * gnw[cbase + d1 + r] + gnb[cbase + d1 + r]));
            }
            store_rows<T>(ha, tt[si], tv[si], ct[si][0], ct[si][1]);
        }
        lds_barrier();
        // (d) h5 = SiLU(gconv3(h4)) stays in registers and feeds y += W2[:, group] h5
        conv_group<T>(wl + 18 * 512, ha, w, ct);
        Frag<T> h5[NSW];
#pragma unroll
        for (int si = 0; si < NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ct[si][0][r] = silu_f(ct[si][0][r] + cb3[cbase + d0 + r]);
                ct[si][1][r] = keep_if(v1, silu_f(ct[si][1][r] + cb3[cbase + d1 + r]));
            }
            frag_from_c2(h5[si], ct[si][0], ct[si][1]);
        }
#pragma unroll
        for (int mt = 0; mt < TF_H / 16; ++mt) {
            Frag<T> a;
            lfrag<T>(a, wl, 24 + mt);
#pragma unroll
            for (int si = 0; si < NSW; ++si) yacc[si][mt] = mma(a, h5[si], yacc[si][mt]);
        }
        }  // pass 1
        if (DB && gr + 1 < TF_G) fwd_wstore(wl0 + (size_t)((gr + 1) & 1) * 30 * 512);  // last read two barriers ago
        lds_barrier();  // ha / hb / red are rewritten by the next group; the other weight buffer is complete
    }

    if (pass == 1)
#pragma unroll
    for (int si = 0; si < NSW; ++si) {
        if (tin[si]) {
            const size_t tg = (size_t)(t0 + tt[si]);
#pragma unroll
            for (int mt = 0; mt < TF_H / 16; ++mt) {
                const int ch = 16 * mt + 4 * g4;
                float xv[4];
                load4(xb + tg * TF_H + ch, xv);
                store4(yb + tg * TF_H + ch, xv[0] + round_to(yacc[si][mt][0] + b2[ch], x), xv[1] + round_to(yacc[si][mt][1] + b2[ch + 1], x),
                       xv[2] + round_to(yacc[si][mt][2] + b2[ch + 2], x), xv[3] + round_to(yacc[si][mt][3] + b2[ch + 3], x));
            }
        }
    }
    }  // chunks x passes
}


// ---------------------------------------------------------------------------------------------
// Backward (data gradient).  Same decomposition as forward; the group chain is recomputed and then
// walked backwards.  Pre-activations stay in registers (they are only needed by the owning wave),
// activations / gradients that neighbouring frames need go through 4 LDS buffers.  Weight
// gradients are NOT formed here: the kernel emits the (activation, pre-activation-gradient) pairs of
// the five linear maps as [B,F,T,FFN] tensors and wgrad.hip contracts them over all tokens.
// LayerNorm / GroupNorm affine gradients are reduced in-kernel (shuffle + atomicAdd).
#define TF_OPS_GM(T) (sizeof(T) == 2)
template <class T>
struct TfOps {  // wgrad operands, each [B*F*T][FFN]
    T *h1, *h2, *h4, *h5, *da1, *da2, *da3, *da5;
};

template <class T>
NBSS_DEV void store_op(T* __restrict__ op, size_t n, bool valid, int gr, size_t ntok, const f32x4& lo, const f32x4& hi) {
    if (!valid) return;
    const int g4 = lane_id() >> 4;
    // bf16 stream: group-major operands [G][N][24] — the 16 frames of a strip write 768 contiguous bytes.  Token-major rows
    // ([N][FFN], 48-byte pieces of a 384-byte row per group) made every store a partial-line write: the kernel fetched 1 GB
    // from HBM per launch (FETCH_SIZE) for 0.2 GB of algorithmic reads.  The fp32 stream keeps [N][FFN] (generic wgrad kernel).
    T* r = TF_OPS_GM(T) ? op + ((size_t)gr * ntok + n) * TF_CG : op + n * TF_FFN + gr * TF_CG;
    store4_nt(r + 4 * g4, lo[0], lo[1], lo[2], lo[3]);
    if (g4 < 2) store4_nt(r + 16 + 4 * g4, hi[0], hi[1], hi[2], hi[3]);
}

// sum over the 16 lanes that share (lane>>4): per-channel reduction over the frames of a strip
NBSS_DEV float sum_l15(float v) { return row_sum16(v); }

template <class T>
__global__ __launch_bounds__(512) void tconvffn_bwd_kernel(nbss_cfg c, LayerPtrs lp, const float* __restrict__ P, float* __restrict__ part, int layer,
                                                           const T* __restrict__ W1, const T* __restrict__ Wc1, const T* __restrict__ Wc2,
                                                           const T* __restrict__ Wc3, const T* __restrict__ W1tn, const T* __restrict__ Wc1t,
                                                           const T* __restrict__ Wc2t, const T* __restrict__ Wc3t, const T* __restrict__ W2t,
                                                           const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                           float* __restrict__ stats, TfOps<T> ops) {
    NBSS_LDS(smem);
    const int T_ = c.T;
    constexpr int tp = TF_TP;  // buffers always hold 16 strips so that no strip ever indexes out of bounds
    T* buf0 = reinterpret_cast<T*>(smem);
    T* buf1 = buf0 + (size_t)(tp + 2) * TF_CG;
    T* buf2 = buf1 + (size_t)(tp + 2) * TF_CG;
    T* buf3 = buf2 + (size_t)(tp + 2) * TF_CG;
    float* red = reinterpret_cast<float*>(buf3 + (size_t)(tp + 2) * TF_CG);  // [8 waves][2]
    float* aff = red + 16;  // [576] per-workgroup sums: GN weight | GN bias | LN weight | LN bias
    float* lnp = aff + TF_AFF;  // [2H] LayerNorm gamma | beta
    float* prm = lnp + 2 * TF_H;  // [6][FFN]: b1 cb1 cb2 cb3 gnw gnb
    T* wl = reinterpret_cast<T*>(prm + 6 * TF_FFN);  // this group's weights: W1 c1 c2 c3 | W2^T c3^T c2^T c1^T W1^T, 6 fragments each
    PHASE_BEGIN(wl + (sizeof(T) == 2 ? 48 * 512 : 0));
    const int bf = blockIdx.x;
    const int tid = threadIdx.x, lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const size_t n0 = (size_t)bf * T_;
    const T* xb = x + n0 * TF_H;
    const T* dyb = dy + n0 * TF_H;
    T* dxb = dx + n0 * TF_H;
    const float* lnw = lp.p[P_TF_LN_W];
    const float* lnb = lp.p[P_TF_LN_B];
    const float* b1_g = lp.p[P_TF_B1];
    const float* b1 = prm + 0 * TF_FFN;
    const float* cb1_g = lp.p[P_TF_C1B];
    const float* cb1 = prm + 1 * TF_FFN;
    const float* cb2_g = lp.p[P_TF_C2B];
    const float* cb2 = prm + 2 * TF_FFN;
    const float* cb3_g = lp.p[P_TF_C3B];
    const float* cb3 = prm + 3 * TF_FFN;
    const float* gnw_g = lp.p[P_TF_GN_W];
    const float* gnw = prm + 4 * TF_FFN;
    const float* gnb_g = lp.p[P_TF_GN_B];
    const float* gnb = prm + 5 * TF_FFN;

    for (int i = tid; i < TF_AFF; i += blockDim.x) aff[i] = 0.f;
    if (tid < TF_CG) {
        T* bs[4] = {buf0, buf1, buf2, buf3};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            store1(bs[i] + tid, 0.f);
            store1(bs[i] + (size_t)(tp + 1) * TF_CG + tid, 0.f);
        }
    }

    int tt[TF_NSW], tc[TF_NSW];
    bool tv[TF_NSW];
#pragma unroll
    for (int si = 0; si < TF_NSW; ++si) {
        tt[si] = (w * TF_NSW + si) * 16 + l15;
        tv[si] = tt[si] < T_;
        tc[si] = tv[si] ? tt[si] : T_ - 1;
    }
    // strips beyond the padded length do nothing but must still hit every barrier
    const bool wact = (w * TF_NSW) * 16 < tp;

    // Registers are the scarce resource of this kernel (the first version kept LN(x), dy and the du accumulators live across
    // the group loop and spilled 416 B/lane: 78 % of the wave time was spent waiting on scratch reloads).  Now only the row
    // statistics persist; LN(x) and dy fragments are re-read per group (L2/L1 hits) and du is formed after the loop.
    for (int i = tid; i < 2 * TF_H; i += blockDim.x) lnp[i] = i < TF_H ? lnw[i] : lnb[i - TF_H];
    for (int i = tid; i < TF_FFN; i += blockDim.x) {
        prm[i] = b1_g[i]; prm[TF_FFN + i] = cb1_g[i]; prm[2 * TF_FFN + i] = cb2_g[i]; prm[3 * TF_FFN + i] = cb3_g[i];
        prm[4 * TF_FFN + i] = gnw_g[i]; prm[5 * TF_FFN + i] = gnb_g[i];
    }
    float smean[TF_NSW], srstd[TF_NSW];
#pragma unroll
    for (int si = 0; si < TF_NSW; ++si) {
        float v[TF_KS][8], sum = 0.f;
#pragma unroll
        for (int ks = 0; ks < TF_KS; ++ks) {
            load8(xb + (size_t)tc[si] * TF_H + ks * 32 + 8 * g4, v[ks]);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[ks][j];
        }
        smean[si] = wave_sum16(sum) * (1.0f / TF_H);
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < TF_KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) q += (v[ks][j] - smean[si]) * (v[ks][j] - smean[si]);
        srstd[si] = rsqrtf(wave_sum16(q) * (1.0f / TF_H) + 1e-5f);
    }

    const int d0 = 4 * g4, d1 = 16 + 4 * g4;
    const bool v1 = g4 < 2;
    const float cnt = (float)(TF_CG * T_);
    const size_t ntok = (size_t)c.B * c.F * T_;

    lds_barrier();  // lnp / prm / halo rows are in place
    PHASE(0);
    f32x4 pend[TF_NSW][2];  // da1 of the previous group, stored after the next group's loads are in flight
#pragma unroll
    for (int si = 0; si < TF_NSW; ++si) pend[si][0] = pend[si][1] = F32X4_ZERO;
    for (int gr = 0; gr < TF_G; ++gr) {
        const int cbase = gr * TF_CG;
        f32x4 a1[TF_NSW][2], a2[TF_NSW][2], a3h[TF_NSW][2], a5[TF_NSW][2], ct[TF_NSW][2];
        // this group's 54 weight fragments: global -> LDS once per workgroup (see the forward kernel); the fp32 stream has
        // no LDS room for them and keeps reading the packed buffer directly
        constexpr bool STAGE = sizeof(T) == 2;
        const T* srcs[8] = {W1, Wc1, Wc2, Wc3, W2t, Wc3t, Wc2t, Wc1t};
        const T* wsl[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wsl[i] = STAGE ? wl + (size_t)i * 6 * 512 : srcs[i] + (size_t)gr * 6 * 512;
        // Every global read of the group is issued here, ahead of the group's operand stores: loads and stores share vmcnt on
        // gfx9, so a load issued after a store cannot complete before that store is acknowledged (the three phases that read
        // x, dy and the weights behind stores were 43 % of the wave time).
        // (Issuing them one phase earlier still, before the last conv phase of the previous group, spilled 20 registers and
        // was slower: 6.7 vs 6.3 ms/step.)
        Frag<T> xr[TF_NSW][TF_KS], dr[TF_NSW][TF_KS];
        StageRegs<T, 8> wreg;
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si)
#pragma unroll
            for (int ks = 0; ks < TF_KS; ++ks) {
                // unconditional (clamped) loads: frames beyond T re-read the last frame; LN(x) of such frames is replaced by beta below
                // and every product of their dy is masked where it is used
                frag_load(xr[si][ks], xb + (size_t)tc[si] * TF_H + ks * 32 + 8 * g4);
                frag_load(dr[si][ks], dyb + (size_t)tc[si] * TF_H + ks * 32 + 8 * g4);
                if (!tv[si]) frag_zero(dr[si][ks]);
            }
        if (STAGE) {
            const StageSrcs<T> wsrc = {W1, Wc1, Wc2, Wc3, W2t, Wc3t, Wc2t, Wc1t};
            wreg.load(wsrc, (size_t)gr * 6 * 512);
        }
        if (gr > 0) {
#pragma unroll
            for (int si = 0; si < TF_NSW; ++si) store_op<T>(ops.da1, n0 + tt[si], tv[si], gr - 1, ntok, pend[si][0], pend[si][1]);
        }
        if (STAGE) {
            wreg.store(wl);
            lds_barrier();
        }
        PHASE(1);
        // ---------------- forward recompute ----------------
        f32x4 dh5[TF_NSW][2];  // W2[:, group]^T dy, formed now (dy is in registers) and used once a5 exists
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
            a1[si][0] = F32X4_ZERO;
            a1[si][1] = F32X4_ZERO;
            dh5[si][0] = F32X4_ZERO;
            dh5[si][1] = F32X4_ZERO;
#pragma unroll
            for (int ks = 0; ks < TF_KS; ++ks) {
                Frag<T> uf;
                float gm[8], bt[8];
                load8(lnp + ks * 32 + 8 * g4, gm);
                load8(lnp + TF_H + ks * 32 + 8 * g4, bt);
#pragma unroll
                for (int j = 0; j < 8; ++j) frag_set(uf, j, tv[si] ? (frag_get(xr[si][ks], j) - smean[si]) * srstd[si] * gm[j] + bt[j] : bt[j]);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    Frag<T> a;
                    lfrag<T>(a, wsl[0], half * 3 + ks);
                    a1[si][half] = mma(a, uf, a1[si][half]);
                    lfrag<T>(a, wsl[4], half * 3 + ks);
                    dh5[si][half] = mma(a, dr[si][ks], dh5[si][half]);
                }
            }
        }
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a1[si][0][r] += b1[cbase + d0 + r];
                a1[si][1][r] = v1 ? a1[si][1][r] + b1[cbase + d1 + r] : 0.f;
                ct[si][0][r] = silu_f(a1[si][0][r]);
                ct[si][1][r] = keep_if(v1, silu_f(a1[si][1][r]));
            }
            store_rows<T>(buf0, tt[si], tv[si], ct[si][0], ct[si][1]);
            store_op<T>(ops.h1, n0 + tt[si], tv[si], gr, ntok, ct[si][0], ct[si][1]);
        }
        PHASE(2);
        lds_barrier();
        PHASE(3);
        conv_group<T>(wsl[1], buf0, w, a2);
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a2[si][0][r] += cb1[cbase + d0 + r];
                a2[si][1][r] = v1 ? a2[si][1][r] + cb1[cbase + d1 + r] : 0.f;
                ct[si][0][r] = silu_f(a2[si][0][r]);
                ct[si][1][r] = keep_if(v1, silu_f(a2[si][1][r]));
            }
            store_rows<T>(buf1, tt[si], tv[si], ct[si][0], ct[si][1]);
            store_op<T>(ops.h2, n0 + tt[si], tv[si], gr, ntok, ct[si][0], ct[si][1]);
        }
        PHASE(4);
        lds_barrier();
        PHASE(5);
        conv_group<T>(wsl[2], buf1, w, a3h);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a3h[si][0][r] = round_to(a3h[si][0][r] + cb2[cbase + d0 + r], x);
                a3h[si][1][r] = v1 ? round_to(a3h[si][1][r] + cb2[cbase + d1 + r], x) : 0.f;
                if (tv[si]) {
                    s1 += a3h[si][0][r] + a3h[si][1][r];
                    s2 += a3h[si][0][r] * a3h[si][0][r] + a3h[si][1][r] * a3h[si][1][r];
                }
            }
        s1 = wave_sum64(s1);
        s2 = wave_sum64(s2);
        if (lane == 0) {
            red[2 * w] = s1;
            red[2 * w + 1] = s2;
        }
        PHASE(6);
        lds_barrier();
        PHASE(7);
        float ts1 = 0.f, ts2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ts1 += red[2 * i];
            ts2 += red[2 * i + 1];
        }
        const float mean = ts1 / cnt;
        const float rstd = rsqrtf(fmaxf(ts2 / cnt - mean * mean, 0.f) + 1e-5f);
        float gw0[4], gw1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gw0[r] = gnw[cbase + d0 + r];
            gw1[r] = v1 ? gnw[cbase + d1 + r] : 0.f;
        }
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a3h[si][0][r] = (a3h[si][0][r] - mean) * rstd;                   // \hat a3
                a3h[si][1][r] = v1 ? (a3h[si][1][r] - mean) * rstd : 0.f;
                ct[si][0][r] = silu_f(a3h[si][0][r] * gw0[r] + gnb[cbase + d0 + r]);
                ct[si][1][r] = keep_if(v1, silu_f(a3h[si][1][r] * gw1[r] + gnb[cbase + d1 + r]));
            }
            store_rows<T>(buf2, tt[si], tv[si], ct[si][0], ct[si][1]);
            store_op<T>(ops.h4, n0 + tt[si], tv[si], gr, ntok, ct[si][0], ct[si][1]);
        }
        PHASE(8);
        lds_barrier();
        PHASE(9);
        conv_group<T>(wsl[3], buf2, w, a5);
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a5[si][0][r] += cb3[cbase + d0 + r];
                a5[si][1][r] = v1 ? a5[si][1][r] + cb3[cbase + d1 + r] : 0.f;
                ct[si][0][r] = silu_f(a5[si][0][r]);
                ct[si][1][r] = keep_if(v1, silu_f(a5[si][1][r]));
            }
            store_op<T>(ops.h5, n0 + tt[si], tv[si], gr, ntok, ct[si][0], ct[si][1]);
        }
        PHASE(10);
        // ---------------- backward ----------------
        // dh5 = W2[:, group]^T dy ; da5 = dh5 * silu'(a5) -> buf3
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ct[si][0][r] = dh5[si][0][r] * dsilu_f(a5[si][0][r]);
                ct[si][1][r] = keep_if(v1, dh5[si][1][r] * dsilu_f(a5[si][1][r]));
            }
            store_rows<T>(buf3, tt[si], tv[si], ct[si][0], ct[si][1]);
            store_op<T>(ops.da5, n0 + tt[si], tv[si], gr, ntok, ct[si][0], ct[si][1]);
        }
        PHASE(11);
        lds_barrier();
        PHASE(12);
        // dh4 = conv3^T(da5) ; dn3 = dh4 * silu'(n3) ; GroupNorm backward -> da3 -> buf2
        conv_group<T>(wsl[5], buf3, w, ct);
        float sa = 0.f, sb = 0.f;
        float dgw[2][4], dgb[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) dgw[0][r] = dgw[1][r] = dgb[0][r] = dgb[1][r] = 0.f;
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float n30 = a3h[si][0][r] * gw0[r] + gnb[cbase + d0 + r];
                const float n31 = v1 ? a3h[si][1][r] * gw1[r] + gnb[cbase + d1 + r] : 0.f;
                ct[si][0][r] = keep_if(tv[si], ct[si][0][r] * dsilu_f(n30));            // dn3
                ct[si][1][r] = keep_if((tv[si] && v1), ct[si][1][r] * dsilu_f(n31));
                dgw[0][r] += ct[si][0][r] * a3h[si][0][r];
                dgw[1][r] += ct[si][1][r] * a3h[si][1][r];
                dgb[0][r] += ct[si][0][r];
                dgb[1][r] += ct[si][1][r];
                sa += gw0[r] * ct[si][0][r] + gw1[r] * ct[si][1][r];
                sb += gw0[r] * ct[si][0][r] * a3h[si][0][r] + gw1[r] * ct[si][1][r] * a3h[si][1][r];
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float w0 = sum_l15(dgw[0][r]), w1 = sum_l15(dgw[1][r]), q0 = sum_l15(dgb[0][r]), q1 = sum_l15(dgb[1][r]);
            if (l15 == 0 && wact) {
                atomicAdd(aff + cbase + d0 + r, w0);
                atomicAdd(aff + TF_FFN + cbase + d0 + r, q0);
                if (v1) {
                    atomicAdd(aff + cbase + d1 + r, w1);
                    atomicAdd(aff + TF_FFN + cbase + d1 + r, q1);
                }
            }
        }
        sa = wave_sum64(sa);
        sb = wave_sum64(sb);
        PHASE(13);
        lds_barrier();  // red is free again (everyone has read the forward statistics)
        if (lane == 0) {
            red[2 * w] = sa;
            red[2 * w + 1] = sb;
        }
        lds_barrier();
        PHASE(14);
        float tsa = 0.f, tsb = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            tsa += red[2 * i];
            tsb += red[2 * i + 1];
        }
        tsa /= cnt;
        tsb /= cnt;
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ct[si][0][r] = rstd * (gw0[r] * ct[si][0][r] - tsa - a3h[si][0][r] * tsb);
                ct[si][1][r] = v1 ? rstd * (gw1[r] * ct[si][1][r] - tsa - a3h[si][1][r] * tsb) : 0.f;
            }
            store_rows<T>(buf2, tt[si], tv[si], ct[si][0], ct[si][1]);
            store_op<T>(ops.da3, n0 + tt[si], tv[si], gr, ntok, ct[si][0], ct[si][1]);
        }
        PHASE(15);
        lds_barrier();
        PHASE(16);
        // dh2 = conv2^T(da3) ; da2 = dh2 * silu'(a2) -> buf1
        conv_group<T>(wsl[6], buf2, w, ct);
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ct[si][0][r] *= dsilu_f(a2[si][0][r]);
                ct[si][1][r] = keep_if(v1, ct[si][1][r] * dsilu_f(a2[si][1][r]));
            }
            store_rows<T>(buf1, tt[si], tv[si], ct[si][0], ct[si][1]);
            store_op<T>(ops.da2, n0 + tt[si], tv[si], gr, ntok, ct[si][0], ct[si][1]);
        }
        PHASE(17);
        lds_barrier();
        PHASE(18);
        // dh1 = conv1^T(da2) ; da1 = dh1 * silu'(a1) ; du += W1[group]^T da1
        conv_group<T>(wsl[7], buf1, w, ct);
#pragma unroll
        for (int si = 0; si < TF_NSW; ++si) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ct[si][0][r] *= dsilu_f(a1[si][0][r]);
                ct[si][1][r] = keep_if(v1, ct[si][1][r] * dsilu_f(a1[si][1][r]));
            }
            // the da1 store is deferred until the next group's loads have been issued (they would queue behind it in vmcnt)
            pend[si][0] = ct[si][0];
            pend[si][1] = ct[si][1];
        }
        PHASE(19);
        lds_barrier();
        PHASE(20);
    }

#pragma unroll
    for (int si = 0; si < TF_NSW; ++si) store_op<T>(ops.da1, n0 + tt[si], tv[si], TF_G - 1, ntok, pend[si][0], pend[si][1]);
    // du = W1^T da1 over all FFN channels, from the [N][FFN] operand this workgroup has just written (the weight-gradient
    // kernel reads the same buffer): a full barrier makes the stores of the other waves visible (never-read lines: no stale L1)
    __syncthreads();
    PHASE(21);
    f32x4 du[TF_NSW][TF_H / 16];
#pragma unroll
    for (int si = 0; si < TF_NSW; ++si) {
#pragma unroll
        for (int mt = 0; mt < TF_H / 16; ++mt) du[si][mt] = F32X4_ZERO;
        for (int k6 = 0; k6 < TF_FFN / 32; ++k6) {
            Frag<T> df;
            const int ch = k6 * 32 + 8 * g4;  // 8-channel pieces never straddle a 24-channel group
            frag_load(df, TF_OPS_GM(T) ? ops.da1 + ((size_t)(ch / TF_CG) * ntok + n0 + tc[si]) * TF_CG + ch % TF_CG
                                       : ops.da1 + (n0 + tc[si]) * TF_FFN + ch);  // (frames beyond T: clamped, their du is discarded)
#pragma unroll
            for (int mt = 0; mt < TF_H / 16; ++mt) {
                Frag<T> a;
                wfrag_load(a, W1tn, mt, TF_FFN / 32, k6);
                du[si][mt] = mma(a, df, du[si][mt]);
            }
        }
    }

    PHASE(22);
    // ---------------- LayerNorm backward + residual, in registers ----------------
    float dlw[TF_H / 16][4], dlb[TF_H / 16][4];
#pragma unroll
    for (int mt = 0; mt < TF_H / 16; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dlw[mt][r] = dlb[mt][r] = 0.f;
#pragma unroll
    for (int si = 0; si < TF_NSW; ++si) {
        float xv[TF_H / 16][4];
        float sum = 0.f;
#pragma unroll
        for (int mt = 0; mt < TF_H / 16; ++mt) {
            load4(xb + (size_t)tc[si] * TF_H + 16 * mt + 4 * g4, xv[mt]);
#pragma unroll
            for (int r = 0; r < 4; ++r) sum += xv[mt][r];
        }
        const float mean = wave_sum16(sum) * (1.0f / TF_H);
        float q = 0.f;
#pragma unroll
        for (int mt = 0; mt < TF_H / 16; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xv[mt][r] -= mean;
                q += xv[mt][r] * xv[mt][r];
            }
        const float rstd = rsqrtf(wave_sum16(q) * (1.0f / TF_H) + 1e-5f);
        if (tv[si] && g4 == 0) {
            stats[(n0 + tt[si]) * 2] = mean;
            stats[(n0 + tt[si]) * 2 + 1] = rstd;
        }
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int mt = 0; mt < TF_H / 16; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = 16 * mt + 4 * g4 + r;
                xv[mt][r] *= rstd;  // \hat x
                const float dv = tv[si] ? du[si][mt][r] : 0.f;
                dlw[mt][r] += dv * xv[mt][r];
                dlb[mt][r] += dv;
                du[si][mt][r] = dv * lnw[ch];
                m1 += du[si][mt][r];
                m2 += du[si][mt][r] * xv[mt][r];
            }
        m1 = wave_sum16(m1) * (1.0f / TF_H);
        m2 = wave_sum16(m2) * (1.0f / TF_H);
        if (tv[si]) {
#pragma unroll
            for (int mt = 0; mt < TF_H / 16; ++mt) {
                const int ch = 16 * mt + 4 * g4;
                float dv[4], o[4];
                load4(dyb + (size_t)tt[si] * TF_H + ch, dv);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = dv[r] + rstd * (du[si][mt][r] - m1 - xv[mt][r] * m2);
                store4(dxb + (size_t)tt[si] * TF_H + ch, o[0], o[1], o[2], o[3]);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < TF_H / 16; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = sum_l15(dlw[mt][r]), b = sum_l15(dlb[mt][r]);
            if (l15 == 0 && wact) {
                atomicAdd(aff + 2 * TF_FFN + 16 * mt + 4 * g4 + r, a);
                atomicAdd(aff + 2 * TF_FFN + TF_H + 16 * mt + 4 * g4 + r, b);
            }
        }
    PHASE(23);
    lds_barrier();
    for (int i = tid; i < TF_AFF; i += blockDim.x) part[(size_t)blockIdx.x * TF_AFF + i] = aff[i];
    PHASE(24);
    PHASE_END();
}
PHASE_READER(nbss_phase_read_tconvffn_bwd)

// ---------------------------------------------------------------------------------------------
// Tail of the backward pass as its own kernel — since tailw.hip fused it with the W1 weight gradient, built only into the A/B flavour
// (-DNBSS_NO_TAILW) that the fusion was measured against (bf16 stream, after tconvffn_s.hip's data-gradient kernel): du = W1^T da1 over all
// FFN channels from the group-major [G][N][24] operand, LayerNorm backward + residual in registers, row statistics for the weight-
// gradient kernel, and the LayerNorm affine partial sums of the workgroup (entries [2 FFN, 2 FFN + 2 H) of its `part` row; the
// GroupNorm entries are written by the data-gradient kernel).  One workgroup = one (b,f) sequence, 16 waves x one 16-frame strip
// (4 waves per SIMD, <= 128 VGPRs: the first version ran 8 waves x 2 strips at 256 VGPRs with 181 spilled registers).
template <class T>
__global__ __launch_bounds__(1024) void tconvffn_du_kernel(nbss_cfg c, LayerPtrs lp, float* __restrict__ part, const T* __restrict__ W1tn,
                                                           const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                           float* __restrict__ stats, const T* __restrict__ da1) {
    NBSS_LDS(smem);
    T* wl = reinterpret_cast<T*>(smem);  // the 36 W1^T fragments, once per workgroup
    float* aff = reinterpret_cast<float*>(wl + 36 * 512);  // [2H] LN weight | bias gradient sums of this workgroup
    float* lnl = aff + 2 * TF_H;                            // [H] LN weight (24 per-lane values: through LDS, the kernel sits at its 128-VGPR budget)
    const int T_ = c.T;
    const int bf = blockIdx.x;
    const int tid = threadIdx.x, lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const size_t n0 = (size_t)bf * T_, ntok = (size_t)c.B * c.F * T_;
    const T* xb = x + n0 * TF_H;
    const T* dyb = dy + n0 * TF_H;
    T* dxb = dx + n0 * TF_H;
    const float* lnw = lp.p[P_TF_LN_W];
    {
        constexpr int NV = 36 * 512 * (int)sizeof(T) / 16;
        for (int v = tid; v < NV; v += 1024) reinterpret_cast<u32x4*>(wl)[v] = reinterpret_cast<const u32x4*>(W1tn)[v];
    }
    for (int i = tid; i < 3 * TF_H; i += blockDim.x) aff[i] = i < 2 * TF_H ? 0.f : lnw[i - 2 * TF_H];
    // this wave's first strip: every global read issued before the barrier
    for (int s16 = w; s16 * 16 < T_ || s16 == w; s16 += 16) {
        const int tt = s16 * 16 + l15;
        const bool tv = tt < T_;
        const int tc = tv ? tt : T_ - 1;
        Frag<T> df[TF_FFN / 32];
        u32x2 xr[TF_H / 16], dr[TF_H / 16];
#pragma unroll
        for (int k6 = 0; k6 < TF_FFN / 32; ++k6) {  // 8-channel pieces never straddle a 24-channel group
            const int ch = k6 * 32 + 8 * g4;
            frag_load(df[k6], da1 + ((size_t)(ch / TF_CG) * ntok + n0 + tc) * TF_CG + ch % TF_CG);
        }
#pragma unroll
        for (int mt = 0; mt < TF_H / 16; ++mt) {
            xr[mt] = *reinterpret_cast<const u32x2*>(xb + (size_t)tc * TF_H + 16 * mt + 4 * g4);
            dr[mt] = *reinterpret_cast<const u32x2*>(dyb + (size_t)tc * TF_H + 16 * mt + 4 * g4);
        }
        if (s16 == w) lds_barrier();  // the fragments are in LDS, aff is zeroed (every wave passes here exactly once)
        f32x4 du[TF_H / 16];
#pragma unroll
        for (int mt = 0; mt < TF_H / 16; ++mt) du[mt] = F32X4_ZERO;
#pragma unroll
        for (int k6 = 0; k6 < TF_FFN / 32; ++k6)
#pragma unroll
            for (int mt = 0; mt < TF_H / 16; ++mt) {
                Frag<T> a;
                frag_load(a, wl + ((size_t)(mt * (TF_FFN / 32) + k6) * 64 + lane) * 8);
                du[mt] = mma(a, df[k6], du[mt]);
            }
        float xv[TF_H / 16][4];
        float sum = 0.f;
#pragma unroll
        for (int mt = 0; mt < TF_H / 16; ++mt) {
            xv[mt][0] = bf2f((bf16_t)(xr[mt][0] & 0xFFFF)); xv[mt][1] = bf2f((bf16_t)(xr[mt][0] >> 16));
            xv[mt][2] = bf2f((bf16_t)(xr[mt][1] & 0xFFFF)); xv[mt][3] = bf2f((bf16_t)(xr[mt][1] >> 16));
#pragma unroll
            for (int r = 0; r < 4; ++r) sum += xv[mt][r];
        }
        const float mean = wave_sum16(sum) * (1.0f / TF_H);
        float q = 0.f;
#pragma unroll
        for (int mt = 0; mt < TF_H / 16; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xv[mt][r] -= mean;
                q += xv[mt][r] * xv[mt][r];
            }
        const float rstd = rsqrtf(wave_sum16(q) * (1.0f / TF_H) + 1e-5f);
        if (tv && g4 == 0) {
            stats[(n0 + tt) * 2] = mean;
            stats[(n0 + tt) * 2 + 1] = rstd;
        }
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int mt = 0; mt < TF_H / 16; ++mt) {
            float gq4[4];
            load4(lnl + 16 * mt + 4 * g4, gq4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xv[mt][r] *= rstd;  // \hat x
                du[mt][r] = keep_if(tv, du[mt][r]);
                const float gq = du[mt][r] * gq4[r];
                m1 += gq;
                m2 += gq * xv[mt][r];
            }
        }
        m1 = wave_sum16(m1) * (1.0f / TF_H);
        m2 = wave_sum16(m2) * (1.0f / TF_H);
        if (tv) {
#pragma unroll
            for (int mt = 0; mt < TF_H / 16; ++mt) {
                const float dd[4] = {bf2f((bf16_t)(dr[mt][0] & 0xFFFF)), bf2f((bf16_t)(dr[mt][0] >> 16)), bf2f((bf16_t)(dr[mt][1] & 0xFFFF)),
                                     bf2f((bf16_t)(dr[mt][1] >> 16))};
                float o[4], gq4[4];
                load4(lnl + 16 * mt + 4 * g4, gq4);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = dd[r] + rstd * (du[mt][r] * gq4[r] - m1 - xv[mt][r] * m2);
                store4(dxb + (size_t)tt * TF_H + 16 * mt + 4 * g4, o[0], o[1], o[2], o[3]);
            }
        }
        // LayerNorm affine gradients of the strip: reduced over its 16 frames, added to the workgroup sums
#pragma unroll
        for (int mt = 0; mt < TF_H / 16; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = sum_l15(du[mt][r] * xv[mt][r]), b = sum_l15(du[mt][r]);
                if (l15 == 0) {
                    atomicAdd(aff + 16 * mt + 4 * g4 + r, a);
                    atomicAdd(aff + TF_H + 16 * mt + 4 * g4 + r, b);
                }
            }
    }
    lds_barrier();
    for (int i = tid; i < 2 * TF_H; i += blockDim.x) part[(size_t)blockIdx.x * TF_AFF + 2 * TF_FFN + i] = aff[i];
}

int tconvffn_bwd_s_launch(const nbss_cfg& c, const LayerPtrs& lp, float* part, const void* packed, int layer, const void* x, const void* dy,
                          void* const* opsv, float* stats, int pstride, hipStream_t st);
struct TailArgs;
int tailw_tconvffn(const nbss_cfg& c, const LayerPtrs& lp, const void* packed, int layer, const void* x, const void* dy, void* dx, float* stats,
                   const void* da1, float* wgpart, float* G, const float* P, hipStream_t st, const Side* sd, hipStream_t* gs);

// the tail (du, LayerNorm backward, dx) runs inside the W1 weight-gradient kernel (tailw.hip) unless built with -DNBSS_NO_TAILW (A/B flavour)
#ifdef NBSS_NO_TAILW
#define TF_FUSED_TAIL 0
#else
#define TF_FUSED_TAIL 1
#endif

// bf16 stream: data-gradient kernel of tconvffn_s.hip + the tail (same operand tensors, same `part` rows as the group-serial kernel)
static int tconvffn_bwd_bf16(const nbss_cfg& c, const float* P, float* G, float* part, const void* packed, int layer, const void* x, const void* dy, void* dx,
                             float* stats, void* const* opsv, float* wgpart, hipStream_t st, const Side* sd, hipStream_t* gs) {
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    ProfScope ps(PK_TCF_B, st);  // both kernels of the sub-block: ONE profiler interval per nbss_tconvffn_bwd call
    if (TF_FUSED_TAIL) {
        int e = tconvffn_bwd_s_launch(c, lp, part, packed, layer, x, dy, opsv, stats, 2 * TF_FFN, st);
        if (e) return e;
        return tailw_tconvffn(c, lp, packed, layer, x, dy, dx, stats, opsv[4], wgpart, G, P, st, sd, gs);
    }
    int e = tconvffn_bwd_s_launch(c, lp, part, packed, layer, x, dy, opsv, nullptr, TF_AFF, st);
    if (e) return e;
    const bf16_t* pk = (const bf16_t*)packed;
    NBSS_LAUNCH((tconvffn_du_kernel<bf16_t>), dim3(c.B * c.F), dim3(1024), 36 * 512 * sizeof(bf16_t) + 3 * TF_H * sizeof(float), st, c, lp, part, pk + pack_off(c, layer, K_TF_W1_TN),
                (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, stats, (const bf16_t*)opsv[4]);
    return NBSS_CHECK_LAUNCH();
}

template <class T>
static int tconvffn_bwd_t(const nbss_cfg& c, const float* P, float* part, const void* packed, int layer, const void* x, const void* dy, void* dx,
                          float* stats, void* const* opsv, hipStream_t st) {
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    if (c.T > TF_TP) return NBSS_EUNSUPPORTED;
    const size_t lds = (size_t)4 * (TF_TP + 2) * TF_CG * sizeof(T) + (16 + TF_AFF + 2 * TF_H + 6 * TF_FFN) * sizeof(float) + (sizeof(T) == 2 ? (size_t)48 * 512 * sizeof(T) : 0) + PHASE_LDS_BYTES;
    if (lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    const T* pk = (const T*)packed;
    TfOps<T> ops;
    ops.h1 = (T*)opsv[0]; ops.h2 = (T*)opsv[1]; ops.h4 = (T*)opsv[2]; ops.h5 = (T*)opsv[3];
    ops.da1 = (T*)opsv[4]; ops.da2 = (T*)opsv[5]; ops.da3 = (T*)opsv[6]; ops.da5 = (T*)opsv[7];
    int e = NBSS_SET_MAX_LDS((tconvffn_bwd_kernel<T>), lds);
    if (e) return e;
    dim3 grid(c.B * c.F), block(512);
    ProfScope ps(PK_TCF_B, st);
    NBSS_LAUNCH((tconvffn_bwd_kernel<T>), grid, block, lds, st, c, lp, P, part, layer, pk + pack_off(c, layer, K_TF_W1), pk + pack_off(c, layer, K_TF_C1),
                pk + pack_off(c, layer, K_TF_C2), pk + pack_off(c, layer, K_TF_C3), pk + pack_off(c, layer, K_TF_W1_TN),
                pk + pack_off(c, layer, K_TF_C1_T), pk + pack_off(c, layer, K_TF_C2_T), pk + pack_off(c, layer, K_TF_C3_T),
                pk + pack_off(c, layer, K_TF_W2_T), (const T*)x, (const T*)dy, (T*)dx, stats, ops);
    return NBSS_CHECK_LAUNCH();
}

int tconvffn_bwd_v_launch(const nbss_cfg& c, const LayerPtrs& lp, float* part, const void* packed, int layer, const void* dy, void* tsave, void* op_da1,
                          hipStream_t st);
float* tconvffn_save_ln_stats(const nbss_cfg& c, void* tsave);
int tconvffn_v_reduce16(const nbss_cfg& c, const void* part16, float* slices, float* G, const long long* offs, bool with_w2, hipStream_t st);
int tconvffn_bwd_q_launch(const nbss_cfg& c, const LayerPtrs& lp, float* part, const void* packed, int layer, const void* dy, void* tsave, void* op_h5, void* op_da1,
                          hipStream_t st);
#ifndef NBSS_TCF_BWD_DEFAULT_Q
#define NBSS_TCF_BWD_DEFAULT_Q 1
#endif
// which data-gradient kernel of tconvffn_s.hip: 1 = tconvffn_bwd_q (a sequence's group pairs: two workgroups per CU; h5 operand + wgrad.hip for W2),
// 0 = tconvffn_bwd_v (four groups per workgroup, the W2 weight gradient contracted inside).  NBSS_TCF_BWD=v|q overrides (A/B; read once per process).
// Same-box measurements (profiles/README.md, round 5): the sub-block (data-gradient kernel + tail kernel) 11.2 ms per step with v, 9.6 with q; the step
// 656-660 utt/s with v, 661-662 with q (in order 656.7 / 656.2): q's W2 weight gradient is a launch of its own again (147 us in order) — contracting it in
// the tail kernel instead (four images per chunk) made THAT kernel 160-250 us slower, in bwd_v it costs what it saves.
static bool tcf_use_q() {
    static const int v = [] {
        const char* e = getenv("NBSS_TCF_BWD");
        return e ? (e[0] == 'q' ? 1 : 0) : NBSS_TCF_BWD_DEFAULT_Q;
    }();
    return v != 0;
}

// bf16 stream, from the pre-activations a training-mode forward saved (tconvffn_s.hip): data gradient + the three T-conv weight gradients + the
// W2 weight gradient in one kernel, the tail + W1 weight gradient in tailw.hip, one fold of the per-sequence partial rows
static int tconvffn_bwd_saved(const nbss_cfg& c, const float* P, float* G, const void* packed, int layer, const void* x, const void* dy, void* tsave,
                              void* dx, void* ws, hipStream_t st, const Side* sd) {
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    const size_t N = (size_t)c.B * c.F * c.T;
    char* base = (char*)ws + ws_align(N * 2 * sizeof(float));
    void* op_h5 = base + (size_t)3 * ws_align(N * TF_FFN * 2);
    void* op_da1 = base + (size_t)4 * ws_align(N * TF_FFN * 2);
    float* part = (float*)((char*)ws + ws_tcpart_offset(c));
    float* wgpart = (float*)((char*)ws + ws_wgpart_offset(c));
    const bool q = tcf_use_q();
    int e;
    hipStream_t gs = st;  // parameter-gradient launches (side.h): everything behind the tail kernel
    FoldScope fs(st, wgpart, WGPART_BYTES, N);  // (fold.h: the sub-block's seven fold launches leave as two, one per stage, on the gradient stream)
    {
        ProfScope ps(PK_TCF_B, st);  // both kernels of the sub-block: ONE profiler interval per nbss_tconvffn_bwd call
        if ((e = q ? tconvffn_bwd_q_launch(c, lp, part, packed, layer, dy, tsave, op_h5, op_da1, st) : tconvffn_bwd_v_launch(c, lp, part, packed, layer, dy, tsave, op_da1, st)))
            return e;
        if ((e = tailw_tconvffn(c, lp, packed, layer, x, dy, dx, tconvffn_save_ln_stats(c, tsave), op_da1, wgpart, G, P, st, sd, &gs))) return e;
    }
    const int convW[3] = {P_TF_C1W, P_TF_C2W, P_TF_C3W}, convBias[3] = {P_TF_C1B, P_TF_C2B, P_TF_C3B};
    AffSegs sg;  // fp32 rows: GroupNorm affine sums + the three conv bias sums (+ W2's bias sums from the four-group kernel)
    sg.n = q ? 5 : 6;
    sg.off[0] = param_off(c, layer, P_TF_GN_W); sg.cnt[0] = TF_FFN;
    sg.off[1] = param_off(c, layer, P_TF_GN_B); sg.cnt[1] = TF_FFN;
    for (int k = 0; k < 3; ++k) { sg.off[2 + k] = param_off(c, layer, convBias[k]); sg.cnt[2 + k] = TF_FFN; }
    sg.off[5] = param_off(c, layer, P_TF_B2); sg.cnt[5] = TF_H;
    if ((e = affine_reduce_launch(part, c.B * c.F, sg, G, gs))) return e;
    const long long woffs[4] = {param_off(c, layer, convW[0]), param_off(c, layer, convW[1]), param_off(c, layer, convW[2]), param_off(c, layer, P_TF_W2)};
    // (the slice sums of the fold live in the wgrad partial-tile region, idle between this sub-block's wgrad launches: 64 x 59 904 floats = 15.3 MB)
    if ((e = tconvffn_v_reduce16(c, part + (size_t)c.B * c.F * (5 * TF_FFN + (q ? 0 : TF_H)), wgpart, G, woffs, !q, gs))) return e;
    if (!q) return fs.end();
    // W2: dW2[H][FFN] = dy^T h5 ; db2 = colsum(dy)
    WgradArgs a;
    a.part = wgpart;
    a.mvalid = 0; a.nvalid = 0;
    a.Ntok = (int)N; a.F = c.F; a.T = c.T; a.shift_stride = 1; a.shift_dim = 0;
    a.stats = nullptr; a.gamma = nullptr; a.beta = nullptr;
    a.A = dy; a.lda = TF_H; a.MA = TF_H; a.B = op_h5; a.ldb = TF_FFN; a.NB = TF_FFN; a.groups = 1; a.taps = 1;
    a.b_gw = TF_CG; a.b_gs = (int)(N * TF_CG);
    a.dW = G + param_off(c, layer, P_TF_W2); a.dbias = G + param_off(c, layer, P_TF_B2);
    if ((e = wgrad_launch(a, c.dtype, gs))) return e;
    return fs.end();
}

int tconvffn_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, int layer, const void* x, const void* dy, const void* tsave,
                      void* dx, void* ws, hipStream_t st, const Side* sd) {
    if (c.H != TF_H) return gb_tconvffn_bwd(c, P, G, layer, x, dy, dx, ws, st, sd);
    if (tsave && c.dtype == NBSS_BF16) return tconvffn_bwd_saved(c, P, G, packed, layer, x, dy, const_cast<void*>(tsave), dx, ws, st, sd);
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    // workspace: stats [N][2] f32 | h1 h2 h4 h5 da1 da2 da3 da5, each [N][FFN] of the stream dtype
    const size_t N = (size_t)c.B * c.F * c.T, esz = c.dtype == NBSS_BF16 ? 2 : 4;
    float* stats = (float*)ws;
    char* base = (char*)ws + ws_align(N * 2 * sizeof(float));
    void* ops[8];
    for (int i = 0; i < 8; ++i) ops[i] = base + (size_t)i * ws_align(N * TF_FFN * esz);
    float* part = (float*)((char*)ws + ws_part_offset(c));
    float* wgpart = (float*)((char*)ws + ws_wgpart_offset(c));
    const bool fused = c.dtype == NBSS_BF16 && TF_FUSED_TAIL;  // the tail kernel contracted dW1 / db1 and reduced the LayerNorm affine sums
    hipStream_t gs = st;  // parameter-gradient launches (side.h)
    int e = c.dtype == NBSS_BF16 ? tconvffn_bwd_bf16(c, P, G, part, packed, layer, x, dy, dx, stats, ops, wgpart, st, sd, &gs)
                                 : tconvffn_bwd_t<float>(c, P, part, packed, layer, x, dy, dx, stats, ops, st);
    if (e) return e;
    if (gs == st) gs = side_fork(sd, st);
    AffSegs sg;
    sg.n = fused ? 2 : 4;
    sg.off[0] = param_off(c, layer, P_TF_GN_W); sg.cnt[0] = TF_FFN;
    sg.off[1] = param_off(c, layer, P_TF_GN_B); sg.cnt[1] = TF_FFN;
    sg.off[2] = param_off(c, layer, P_TF_LN_W); sg.cnt[2] = TF_H;
    sg.off[3] = param_off(c, layer, P_TF_LN_B); sg.cnt[3] = TF_H;
    if ((e = affine_reduce_launch(part, c.B * c.F, sg, G, gs))) return e;
    WgradArgs a;
    a.part = wgpart;
    a.mvalid = 0; a.nvalid = 0;
    a.Ntok = (int)N; a.F = c.F; a.T = c.T; a.shift_stride = 1; a.shift_dim = 0;
    a.stats = nullptr; a.gamma = nullptr; a.beta = nullptr;
    // W2: dW2[H][FFN] = dy^T h5 ; db2 = colsum(dy)
    const bool gm = c.dtype == NBSS_BF16;  // TF_OPS_GM: operands are [G][N][24]
    const int ogw = gm ? TF_CG : 0, ogs = gm ? (int)(N * TF_CG) : 0;
    a.A = dy; a.lda = TF_H; a.MA = TF_H; a.B = ops[3]; a.ldb = TF_FFN; a.NB = TF_FFN; a.groups = 1; a.taps = 1;
    a.b_gw = ogw; a.b_gs = ogs;
    a.dW = G + param_off(c, layer, P_TF_W2); a.dbias = G + param_off(c, layer, P_TF_B2);
    if ((e = wgrad_launch(a, c.dtype, gs))) return e;
    // the three grouped k=3 convs
    const int convA[3] = {5, 6, 7}, convB[3] = {0, 1, 2};
    const int convW[3] = {P_TF_C1W, P_TF_C2W, P_TF_C3W}, convBias[3] = {P_TF_C1B, P_TF_C2B, P_TF_C3B};
    for (int k = 0; k < 3; ++k) {
        a.A = ops[convA[k]]; a.lda = TF_FFN; a.MA = TF_FFN; a.B = ops[convB[k]]; a.ldb = TF_FFN; a.NB = TF_FFN;
        a.groups = c.t_groups; a.taps = c.t_ks;
        a.a_gw = ogw; a.a_gs = ogs; a.b_gw = ogw; a.b_gs = ogs;
        a.dW = G + param_off(c, layer, convW[k]); a.dbias = G + param_off(c, layer, convBias[k]);
        if ((e = wgrad_launch(a, c.dtype, gs))) return e;
    }
    if (fused) return NBSS_OK;
    // W1: dW1[FFN][H] = da1^T LN(x) ; db1 = colsum(da1)
    a.A = ops[4]; a.lda = TF_FFN; a.MA = TF_FFN; a.B = x; a.ldb = TF_H; a.NB = TF_H; a.groups = 1; a.taps = 1;
    a.a_gw = ogw; a.a_gs = ogs; a.b_gw = 0; a.b_gs = 0;
    a.stats = stats; a.gamma = lp.p[P_TF_LN_W]; a.beta = lp.p[P_TF_LN_B];
    a.dW = G + param_off(c, layer, P_TF_W1); a.dbias = G + param_off(c, layer, P_TF_B1);
    return wgrad_launch(a, c.dtype, gs);
}

template <class T, int NSW, bool LONG>
static int tconvffn_fwd_t(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st) {
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    if (!LONG && c.T > TF_TP) return NBSS_EUNSUPPORTED;
    const size_t lds = (size_t)2 * (TF_TP + 2) * TF_CG * sizeof(T) + (32 + 7 * TF_FFN + 2 * TF_G) * sizeof(float) + (size_t)(sizeof(T) == 2 ? 2 : 1) * 30 * 512 * sizeof(T);
    const T* pk = (const T*)packed;
    dim3 grid(c.B * c.F), block(64 * 16 / NSW);
    ProfScope ps(PK_TCF_F, st);
    NBSS_LAUNCH((tconvffn_fwd_kernel<T, NSW, LONG>), grid, block, lds, st, c, lp, P, layer, pk + pack_off(c, layer, K_TF_W1), pk + pack_off(c, layer, K_TF_C1),
                pk + pack_off(c, layer, K_TF_C2), pk + pack_off(c, layer, K_TF_C3), pk + pack_off(c, layer, K_TF_W2), (const T*)x, (T*)y);
    return NBSS_CHECK_LAUNCH();
}

int tconvffn_fwd_s_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, void* tsave, hipStream_t st, const SeqTail* tl);

int tconvffn_fwd_large_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st);

// tsave (optional; bf16 stream, T <= 256, small geometry — tconvffn_save_bytes() > 0): the training-mode forward keeps its pre-activations there
int tconvffn_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, void* tsave, hipStream_t st, const SeqTail* tl) {
    if (c.H != TF_H) return tconvffn_fwd_large_impl(c, P, packed, layer, x, y, st);  // SpatialNet-large, forward only (tconvffn_g.hip)
    // bf16 stream: the streaming wave-per-group kernel (tconvffn_s.hip); fp32 stream: the group-serial kernel above
    // sequences beyond 256 frames (forward only): the chunked two-pass variant of the group-serial kernel
    if (c.T > TF_TP) return c.dtype == NBSS_BF16 ? tconvffn_fwd_t<bf16_t, 1, true>(c, P, packed, layer, x, y, st) : tconvffn_fwd_t<float, 2, true>(c, P, packed, layer, x, y, st);
    return c.dtype == NBSS_BF16 ? tconvffn_fwd_s_impl(c, P, packed, layer, x, y, tsave, st, tl) : tconvffn_fwd_t<float, 2, false>(c, P, packed, layer, x, y, st);
}
