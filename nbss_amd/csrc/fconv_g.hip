// fconv_g.hip — backward of the F-conv block y = x + PReLU(conv_F(LN(x))) (models/arch/SpatialNet.py:116-127,75-86) for the geometry-generic
// path (SpatialNet-large: 192 channels, 8 groups of 24, kernel 5 along F), bf16 stream, in ONE kernel per block.
//
// The block is local to a (batch item, frame) slab: LayerNorm is per token, the convolution runs along F.  One workgroup (8 waves) = one slab
// with the whole F axis in two LDS images ([F + 4 halo rows][192 + 8]): x -> LN(x) -> du in the first, dy -> dv in the second, each overwritten
// in place by its successor (a conv group only touches its own 24 columns, and the wave that owns the group is the one that overwrites them).
// Same plan as the small geometry's fconv_bwd_kernel (fconv.hip), whose constants are those of 96 channels / 12 per group:
//   rows   (phase 0): the slab of x and dy arrives by global -> LDS DMA; a wave owns 16-frequency tiles: statistics, LN(x) in place
//   groups (phase 1): wave g = conv group g for all tiles: conv recompute on the MFMA (weights = A, frequencies = N; (tap, channel) is ONE
//                     contraction axis of 120), PReLU', dv in place of dy, slope gradient
//   groups (phase 2): transposed conv of dv -> du in place of LN(x); dv rows out (operand of the conv weight gradient, wgrad.hip)
//   rows   (phase 3): LayerNorm backward + residual -> dx; affine gradients
// The unfused path (gbwd.hip) took LN forward + 2 tap-GEMMs + PReLU backward + LN backward = ~460 us per block at batch 4.
#include "launch.h"
#include "layout.h"
#include "prof.h"
#include "tchain.h"
#include "blocks.h"
#include <cstdlib>

#define FG_WAVES 8
#define FG_THREADS (64 * FG_WAVES)
#define FG_TAPS 5

template <int HH>
struct FgGeo {
    static constexpr int G = 8, CG = HH / G, NK = FG_TAPS * CG, NKS = (NK + 31) / 32, OT = (CG + 15) / 16, NP = NK / 8, PPR = CG / 8;
    static constexpr int LD = HH + 8;          // image row stride in elements (400 bytes: 16 consecutive rows start on distinct 16-byte bank slots)
    static constexpr int PR = LD / 8;          // 16-byte pieces per image row (the last one is the padding)
    static constexpr int KSR = HH / 32;        // k-steps of a row in the row phases
    static constexpr int WSET = NKS * OT * 512;
};

struct FconvG {
    const void* x;
    const void* dy;
    void* dx;
    void* dv;       // [N][H] out: gradient w.r.t. the conv output (operand of the weight gradient)
    float* stats;   // [N][2] out: LayerNorm (mean, rstd) of x
    const void* wf;  // fragment-ordered conv weights (tchain.hip: tc_wprep), forward / data gradient
    const void* wd;
    const float* lnw;
    const float* lnb;
    const float* cb;
    const float* slope;
    float* dlnw;  // accumulated (atomics)
    float* dlnb;
    float* dslope;
    float* part;  // [B T][3 HH] per-workgroup rows of those three (affine_reduce folds them), or null: atomics
    int B, F, T;
};

template <int HH>
__global__ __launch_bounds__(FG_THREADS, 1) void fconv_bwd_g_kernel(FconvG p) {
    using Geo = FgGeo<HH>;
    constexpr int CG = Geo::CG, NKS = Geo::NKS, OT = Geo::OT, LD = Geo::LD, PR = Geo::PR, KSR = Geo::KSR, PPR = Geo::PPR;
    typedef bf16_t T;
    NBSS_LDS(smem);
    const int F = p.F, T_ = p.T, mtf = cdiv(F, 16), FP = mtf * 16 + 4;
    T* U = reinterpret_cast<T*>(smem);              // [FP][LD]  x, then LN(x) (phases 0-1), then du (phases 2-3); image row f + 2
    T* D = U + (size_t)FP * LD;                      // [FP][LD]  dy (phases 0-1), then dv in place
    float* aff = reinterpret_cast<float*>(D + (size_t)FP * LD);  // [3 HH] LN weight | LN bias | PReLU slope gradient sums
    float* lnp = aff + 3 * HH;                       // [2 HH] gamma | beta
    float* rst = lnp + 2 * HH;                       // [16 mtf][2] row statistics
    const int tid = threadIdx.x, lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id_u();
    const int b = blockIdx.x / T_, t = blockIdx.x % T_;
    const T* xg = reinterpret_cast<const T*>(p.x);
    const T* dyg = reinterpret_cast<const T*>(p.dy);
    auto tok = [&](int f) { return ((size_t)b * F + f) * T_ + t; };

    // ---- phase 0: slab in by DMA (16-byte pieces, no register stop); zero rows; parameters ----
    {
        const int Q = F * PR;
        for (int i = w; i * 64 < Q; i += FG_WAVES) {
            const int q = i * 64 + lane, f = q / PR, cc = q - f * PR;
            if (q < Q && cc < PR - 1) {
                const size_t n = tok(f);
                dma16_to_lds(reinterpret_cast<char*>(U + 2 * LD) + (size_t)i * 1024, xg + n * HH + cc * 8);
                dma16_to_lds(reinterpret_cast<char*>(D + 2 * LD) + (size_t)i * 1024, dyg + n * HH + cc * 8);
            }
        }
        const u32x4 z = {0u, 0u, 0u, 0u};
        const int nz = (FP - F) * PR;  // halo rows 0, 1 and rows F + 2 .. FP - 1
        for (int i = tid; i < nz; i += FG_THREADS) {
            const int r = i / PR, cc = i % PR, row = r < 2 ? r : F + r;
            *reinterpret_cast<u32x4*>(U + (size_t)row * LD + cc * 8) = z;
            *reinterpret_cast<u32x4*>(D + (size_t)row * LD + cc * 8) = z;
        }
        for (int i = tid; i < 3 * HH; i += FG_THREADS) aff[i] = 0.f;
        for (int i = tid; i < 2 * HH; i += FG_THREADS) lnp[i] = i < HH ? p.lnw[i] : p.lnb[i - HH];
    }
    // weights of this wave's group, both directions (16 fragments)
    Frag<T> af[NKS][OT], at[NKS][OT];
    {
        const T* wf = reinterpret_cast<const T*>(p.wf) + (size_t)w * Geo::WSET + lane * 8;
        const T* wd = reinterpret_cast<const T*>(p.wd) + (size_t)w * Geo::WSET + lane * 8;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int i = 0; i < OT; ++i) {
                frag_load(af[ks][i], wf + (ks * OT + i) * 512);
                frag_load(at[ks][i], wd + (ks * OT + i) * 512);
            }
    }
    dma_wait_all();
    lds_barrier();
    // rows: statistics, LN(x) in place
    for (int ft = w; ft < mtf; ft += FG_WAVES) {
        const int f = ft * 16 + l15;
        const bool valid = f < F;
        T* ur = U + (size_t)(f + 2) * LD + 8 * g4;
        float xv[KSR][8];
        float sum = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSR; ++ks) {
            load8(ur + ks * 32, xv[ks]);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += xv[ks][j];
        }
        const float mean = wave_sum16(sum) * (1.0f / HH);
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = xv[ks][j] - mean;
                q += d * d;
            }
        const float rstd = rsqrtf(wave_sum16(q) * (1.0f / HH) + 1e-5f);
        if (g4 == 0) {
            rst[2 * f] = mean;
            rst[2 * f + 1] = rstd;
            if (valid) {
                p.stats[2 * tok(f)] = mean;
                p.stats[2 * tok(f) + 1] = rstd;
            }
        }
#pragma unroll
        for (int ks = 0; ks < KSR; ++ks) {
            float gm[8], bt[8], o[8];
            load8(lnp + ks * 32 + 8 * g4, gm);
            load8(lnp + HH + ks * 32 + 8 * g4, bt);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = valid ? (xv[ks][j] - mean) * rstd * gm[j] + bt[j] : 0.f;
            store8(ur + ks * 32, o);
        }
    }
    lds_barrier();

    // ---- phases 1 + 2: wave = conv group ----
    // lane's image offsets of its B-fragment piece per k-step: piece pk = 4 ks + g4 -> (tap, 8-channel piece); past the contraction: a valid piece (zero weights)
    int boff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        int pk = 4 * ks + g4;
        pk = pk < Geo::NP ? pk : Geo::NP - 1;
        boff[ks] = (pk / PPR) * LD + w * CG + (pk % PPR) * 8;  // image row f + tap (f + 2 + tap - 2)
    }
    auto conv = [&](const T* img, const Frag<T> (&wq)[NKS][OT], int ft, f32x4 (&acc)[OT]) {
        const T* rp = img + (size_t)(ft * 16 + l15) * LD;
#pragma unroll
        for (int i = 0; i < OT; ++i) acc[i] = F32X4_ZERO;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            Frag<T> bq;
            frag_load(bq, rp + boff[ks]);
#pragma unroll
            for (int i = 0; i < OT; ++i) acc[i] = mma(wq[ks][i], bq, acc[i]);
        }
    };
    // C layout: lane = frequency 16 ft + l15, channels w CG + 16 i + 4 g4 + r (valid while 16 i + 4 g4 < CG)
    float cbv[OT][4], slv[OT][4], dsl[OT][4];
#pragma unroll
    for (int i = 0; i < OT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cl = 16 * i + 4 * g4 + r;
            cbv[i][r] = cl < CG ? p.cb[w * CG + cl] : 0.f;
            slv[i][r] = cl < CG ? p.slope[w * CG + cl] : 0.f;
            dsl[i][r] = 0.f;
        }
    for (int ft = 0; ft < mtf; ++ft) {
        f32x4 acc[OT];
        conv(U, af, ft, acc);
        const int f = ft * 16 + l15;
#pragma unroll
        for (int i = 0; i < OT; ++i)
            if (16 * i + 4 * g4 < CG) {
                T* pd = D + (size_t)(f + 2) * LD + w * CG + 16 * i + 4 * g4;
                float dyv[4], dv[4];
                load4(pd, dyv);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = round_to(acc[i][r] + cbv[i][r], pd);  // (the pre-activation as the unfused path stores it)
                    dv[r] = v > 0.f ? dyv[r] : slv[i][r] * dyv[r];
                    if (v <= 0.f) dsl[i][r] += dyv[r] * v;
                }
                store4(pd, dv[0], dv[1], dv[2], dv[3]);
            }
    }
#pragma unroll
    for (int i = 0; i < OT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s2 = row_sum16(dsl[i][r]);
            if (l15 == 0 && 16 * i + 4 * g4 < CG) atomicAdd(aff + 2 * HH + w * CG + 16 * i + 4 * g4 + r, s2);
        }
    lds_barrier();  // dv complete (a transposed conv reads its group's columns of rows written by this wave only — but the row copy below reads all)
    for (int ft = 0; ft < mtf; ++ft) {
        f32x4 acc[OT];
        conv(D, at, ft, acc);
        const int f = ft * 16 + l15;
#pragma unroll
        for (int i = 0; i < OT; ++i)
            if (16 * i + 4 * g4 < CG) store4(U + (size_t)(f + 2) * LD + w * CG + 16 * i + 4 * g4, acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    }
    // dv rows -> global, 16-byte pieces of full rows
    {
        T* dvg = reinterpret_cast<T*>(p.dv);
        for (int i = tid; i < F * (PR - 1); i += FG_THREADS) {
            const int f = i / (PR - 1), cc = i % (PR - 1);
            *reinterpret_cast<u32x4*>(dvg + tok(f) * HH + cc * 8) = *reinterpret_cast<const u32x4*>(D + (size_t)(f + 2) * LD + cc * 8);
        }
    }
    lds_barrier();

    // ---- phase 3 (rows): LayerNorm backward + residual; x and dy again from global (L2) ----
    for (int ft = w; ft < mtf; ft += FG_WAVES) {
        const int f = ft * 16 + l15;
        const bool valid = f < F;
        const size_t n = tok(valid ? f : 0);
        const T* ur = U + (size_t)(f + 2) * LD + 8 * g4;
        const float mean = rst[2 * f], rstd = valid ? rst[2 * f + 1] : 0.f;
        float xh[KSR][8], dyv[KSR][8];
#pragma unroll
        for (int ks = 0; ks < KSR; ++ks) {
            load8(xg + n * HH + ks * 32 + 8 * g4, xh[ks]);
            load8(dyg + n * HH + ks * 32 + 8 * g4, dyv[ks]);
        }
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSR; ++ks) {
            float duv[8], gm[8];
            load8(ur + ks * 32, duv);
            load8(lnp + ks * 32 + 8 * g4, gm);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dv = valid ? duv[j] : 0.f;
                xh[ks][j] = (xh[ks][j] - mean) * rstd;
                const float a = row_sum16(dv * xh[ks][j]), bb = row_sum16(dv);
                if (l15 == 0) {
                    atomicAdd(aff + ks * 32 + 8 * g4 + j, a);
                    atomicAdd(aff + HH + ks * 32 + 8 * g4 + j, bb);
                }
                m1 += dv * gm[j];
                m2 += dv * gm[j] * xh[ks][j];
            }
        }
        m1 = wave_sum16(m1) * (1.0f / HH);
        m2 = wave_sum16(m2) * (1.0f / HH);
        if (valid) {
            T* dxr = reinterpret_cast<T*>(p.dx) + n * HH + 8 * g4;
#pragma unroll
            for (int ks = 0; ks < KSR; ++ks) {
                float duv[8], gm[8], o[8];
                load8(ur + ks * 32, duv);
                load8(lnp + ks * 32 + 8 * g4, gm);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = dyv[ks][j] + rstd * (duv[j] * gm[j] - m1 - xh[ks][j] * m2);
                store8(dxr + ks * 32, o);
            }
        }
    }
    lds_barrier();
    // (1 004 workgroups adding to the same 576 addresses: chains of same-address atomics that outlast the kernel's memory phase — rows + a reduce)
    for (int i = tid; i < 3 * HH; i += FG_THREADS) {
        if (p.part) p.part[(size_t)blockIdx.x * 3 * HH + i] = aff[i];
        else atomicAdd(i < HH ? p.dlnw + i : i < 2 * HH ? p.dlnb + (i - HH) : p.dslope + (i - 2 * HH), aff[i]);
    }
}

// host side -----------------------------------------------------------------------------------------------------------------------------
bool fconv_g_takes(const nbss_cfg& c) {
    static const bool off = [] {
        const char* e = getenv("NBSS_FCONVG_OFF");
        return e && e[0] == '1';
    }();
    const int mtf = cdiv(c.F, 16);
    const size_t lds = (size_t)2 * (mtf * 16 + 4) * FgGeo<192>::LD * 2 + (5 * 192 + 2 * 16 * mtf) * sizeof(float);
    return !off && c.dtype == NBSS_BF16 && c.H == 192 && c.f_groups == 8 && c.f_ks == FG_TAPS && lds <= 160 * 1024;
}
size_t fconv_g_wfrag_elems() { return tc_wfrag_elems(8, FgGeo<192>::CG, FG_TAPS); }
int fconv_g_bwd(const nbss_cfg& c, const float* P, float* G, int layer, int which, const void* x, const void* dy, void* dx, void* dv, float* stats, void* wf,
                void* wd, float* part, hipStream_t st) {
    const int pLW = which ? P_FC2_LN_W : P_FC1_LN_W, pLB = which ? P_FC2_LN_B : P_FC1_LN_B, pW = which ? P_FC2_W : P_FC1_W, pB = which ? P_FC2_B : P_FC1_B,
              pA = which ? P_FC2_PRELU : P_FC1_PRELU;
    int e = tc_wprep_one(P + param_off(c, layer, pW), wf, wd, 8, FgGeo<192>::CG, FG_TAPS, st);
    if (e) return e;
    FconvG p;
    p.x = x; p.dy = dy; p.dx = dx; p.dv = dv; p.stats = stats; p.wf = wf; p.wd = wd;
    p.lnw = P + param_off(c, layer, pLW); p.lnb = P + param_off(c, layer, pLB); p.cb = P + param_off(c, layer, pB); p.slope = P + param_off(c, layer, pA);
    p.dlnw = G + param_off(c, layer, pLW); p.dlnb = G + param_off(c, layer, pLB); p.dslope = G + param_off(c, layer, pA);
    p.B = c.B; p.F = c.F; p.T = c.T;
    p.part = part;
    const int mtf = cdiv(c.F, 16);
    const size_t lds = (size_t)2 * (mtf * 16 + 4) * FgGeo<192>::LD * 2 + (5 * 192 + 2 * 16 * mtf) * sizeof(float);
    if ((e = NBSS_SET_MAX_LDS(fconv_bwd_g_kernel<192>, lds))) return e;
    NBSS_LAUNCH(fconv_bwd_g_kernel<192>, dim3(c.B * c.T), dim3(FG_THREADS), lds, st, p);
    if ((e = NBSS_CHECK_LAUNCH()) || !part) return e;
    AffSegs sg;
    sg.n = 3;
    sg.off[0] = param_off(c, layer, pLW); sg.off[1] = param_off(c, layer, pLB); sg.off[2] = param_off(c, layer, pA);
    sg.cnt[0] = sg.cnt[1] = sg.cnt[2] = 192;
    return affine_reduce_launch(part, c.B * c.T, sg, G, st);
}
