// mhsa_bwd.hip — data gradient of the narrow-band self-attention module (forward: mhsa.hip).
//
// One workgroup (8 waves, two 16-frame strips each) = one (b,f) sequence, one head at a time.
// Recomputed per head: Q' = q * log2(e)/sqrt(dh), K, V (from LN(x)) and dO = Wo_h^T dy.  LDS holds
// Q', K, V, dO row-major [T][24] and (fp32 stream only) Q', K transposed [24][T]; nothing else is staged.
//   pass 1 (wave owns QUERY strips):  S^T = K Q'^T, P, D = rowsum(dO * O), dS^T = P (dP^T - D),
//                                     dQ^T = K^T dS^T              (O = saved forward attention output)
//   pass 2 (wave owns KEY strips):    S = Q' K^T, P = exp2(S - m)/l from the stored row statistics,
//                                     dS = P (dP - D),  dV^T += dO^T P,  dK^T += Q'^T dS
// Both passes keep their S / dS tiles in the C layout that is directly the next MFMA's B operand
// (permuted K order), so there is no register transpose and no atomics.  dQ/dK/dV tiles go (a) to
// global [N][3H] for the in_proj weight gradient (wgrad.hip) and (b) straight into
// du += Win^T dqkv, followed by the in-register LayerNorm backward.
#include "launch.h"
#include "layout.h"
#include "prof.h"
#include "blocks.h"
#include "wgrad.h"
#include "fold.h"
#include "side.h"
#include <cstdlib>

#define MB_H 96
#define MB_HEADS 4
#define MB_DH 24
#define MB_NT 16
#ifndef MB_NSW
#define MB_NSW 2   // strips per wave: 2 (8 waves per sequence) or 1 (16 waves, <= 128 VGPRs: four waves per SIMD)
#endif
#define MB_NTHR (64 * 16 / MB_NSW)
#define MB_KS 3

template <class T>
NBSS_DEV void row_pieces(Frag<T>& f, const T* __restrict__ row) {  // [.. 24] row -> permuted-K fragment (d = 4g+j | 16+4g+j)
    const int g4 = lane_id() >> 4;
    frag_load_lo(f, row + 4 * g4);
    if (g4 < 2) frag_load_hi(f, row + 16 + 4 * g4);
    else frag_zero_hi(f);
}

template <class T>
NBSS_DEV void store_row24(T* __restrict__ row, const f32x4& lo, const f32x4& hi) {
    const int g4 = lane_id() >> 4;
    store4(row + 4 * g4, lo[0], lo[1], lo[2], lo[3]);
    if (g4 < 2) store4(row + 16 + 4 * g4, hi[0], hi[1], hi[2], hi[3]);
}

template <class T>
NBSS_DEV void store_row24_nt(T* __restrict__ row, const f32x4& lo, const f32x4& hi) {  // streaming: the dqkv wgrad operand
    const int g4 = lane_id() >> 4;
    store4_nt(row + 4 * g4, lo[0], lo[1], lo[2], lo[3]);
    if (g4 < 2) store4_nt(row + 16 + 4 * g4, hi[0], hi[1], hi[2], hi[3]);
}

template <class T>
NBSS_DEV void store_col24(T* __restrict__ base, int tp, int t, const f32x4& lo, const f32x4& hi) {  // transposed [24][tp]
    const int g4 = lane_id() >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        store1(base + (size_t)(4 * g4 + r) * tp + t, lo[r]);
        if (g4 < 2) store1(base + (size_t)(16 + 4 * g4 + r) * tp + t, hi[r]);
    }
}

// A fragment of a transposed [24][tp] array: rows d = half*16 + l15, K = 32 frames of k-step ks (permuted order)
// bf16: the same operand straight from the ROW-MAJOR [tp][24] array through transposing LDS reads (no transposed copy,
// no 16-way bank conflicts of the [24][tp] image: its rows are 512 B apart)
NBSS_DEV void col_frag_tr(Frag<bf16_t>& f, const bf16_t* __restrict__ rowmajor, int half, int ks, bool hi_valid) {
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    const bf16_t* p = rowmajor + (size_t)(ks * 32 + 4 * g4 + (l15 >> 2)) * MB_DH + half * 16 + 4 * (l15 & 3);
    const u32x2 lo = lds_tr4_b16(p);
    u32x2 hi = {0u, 0u};
    if (hi_valid) hi = lds_tr4_b16(p + 16 * MB_DH);
    u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    f.v = __builtin_bit_cast(s16x8, v);
    // rows d >= 24 of the second half-tile pick up neighbouring data: finite, and every consumer discards those rows
}
NBSS_DEV void col_frag_tr(Frag<float>&, const float*, int, int, bool) {}

template <class T>
NBSS_DEV void col_frag(Frag<T>& f, const T* __restrict__ base, int tp, int half, int ks, bool hi_valid) {
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    const int d = half * 16 + l15;
    if (d < MB_DH) {
        const T* p = base + (size_t)d * tp + ks * 32 + 4 * g4;
        frag_load_lo(f, p);
        if (hi_valid) frag_load_hi(f, p + 16);
        else frag_zero_hi(f);
    } else {
        frag_zero(f);
    }
}

// fp32 stream: the same operand gathered from the ROW-MAJOR [tp][24] array with scalar reads (dO has no transposed copy: seven
// [tp][24] fp32 arrays would not fit 160 KB at tp = 256, and the YAML default `precision: 32` must train 4-s utterances)
template <class T>
NBSS_DEV void col_frag_rm(Frag<T>& f, const T* __restrict__ rowmajor, int half, int ks, bool hi_valid) {
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    const int d = half * 16 + l15;
    if (d < MB_DH) {
        const T* p = rowmajor + (size_t)(ks * 32 + 4 * g4) * MB_DH + d;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            frag_set(f, j, load1(p + (size_t)j * MB_DH));
            frag_set(f, 4 + j, hi_valid ? load1(p + (size_t)(16 + j) * MB_DH) : 0.f);
        }
    } else {
        frag_zero(f);
    }
}

// four consecutive stream elements as loaded (converted only on use)
template <class T> struct RawRow4;
template <> struct RawRow4<bf16_t> {
    u32x2 v;
    NBSS_DEV void load(const bf16_t* p) { v = *reinterpret_cast<const u32x2*>(p); }
    NBSS_DEV void get(float (&o)[4]) const {
        o[0] = bf2f((bf16_t)(v[0] & 0xFFFF)); o[1] = bf2f((bf16_t)(v[0] >> 16));
        o[2] = bf2f((bf16_t)(v[1] & 0xFFFF)); o[3] = bf2f((bf16_t)(v[1] >> 16));
    }
};
template <> struct RawRow4<float> {
    f32x4 v;
    NBSS_DEV void load(const float* p) { v = *reinterpret_cast<const f32x4*>(p); }
    NBSS_DEV void get(float (&o)[4]) const { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
};

// FULL: T in (240, 256] — all 16 strips of every wave exist, so the strip / tile-existence tests are compile-time true and the
// tile loops have constant trip counts (wave-uniform but dynamic branches kept the compiler from scheduling across them: the same
// effect cost wgrad 24 %, profiles/README.md row 27)
// XT (bf16): the tail — du = Win^T dqkv, LayerNorm backward, dx, LN-affine sums — runs in tailw.hip's fused tail / in_proj weight-gradient
// kernel instead; this kernel then ends with the dqkv operand and writes the LayerNorm row statistics for it.
template <class T, bool FULL, bool XT>
__global__ __launch_bounds__(MB_NTHR) void mhsa_bwd_kernel(nbss_cfg c, LayerPtrs lp, const float* __restrict__ P, float* __restrict__ part, int layer,
                                                       const T* __restrict__ Win, const T* __restrict__ WinT, const T* __restrict__ WoutT,
                                                       const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ osave,
                                                       const float* __restrict__ lse, T* __restrict__ dx, float* __restrict__ stats, T* __restrict__ dqkv) {
    NBSS_LDS(smem);
    const int T_ = c.T, nst = FULL ? MB_NT : cdiv(T_, 16), tp = nst * 16, nkp = FULL ? MB_NT / 2 : cdiv(nst, 2);
    T* Qr = reinterpret_cast<T*>(smem);
    T* Kr = Qr + (size_t)tp * MB_DH;
    T* Vr = Kr + (size_t)tp * MB_DH;
    T* dOr = Vr + (size_t)tp * MB_DH;
    constexpr bool TR = sizeof(T) == 2;  // bf16: transposing LDS reads replace the transposed copies
    T* Qt = dOr + (size_t)tp * MB_DH;
    T* Kt = Qt + (size_t)tp * MB_DH;
    float* m2s = reinterpret_cast<float*>(TR ? Qt : Kt + (size_t)tp * MB_DH);
    float* lis = m2s + tp;
    float* Dds = lis + tp;
    float* aff = Dds + tp;  // [2H] per-workgroup LN weight | bias gradient sums
    float* lnp = aff + 2 * MB_H;  // [2H] LayerNorm gamma | beta
    // bf16: every weight fragment of the module lives in LDS for the whole kernel (in_proj 72 + out_proj^T 24 fragments, 96 KB;
    // replaced by the 54 fragments of in_proj^T for the du phase) — per-wave global fragment reads sat behind the dqkv stores
    T* wl = reinterpret_cast<T*>(lnp + 2 * MB_H);
    constexpr int WL_FR = TR ? 96 : 0;
    PHASE_BEGIN(wl + (size_t)WL_FR * 512);
    const size_t ntok = (size_t)c.B * c.F * T_;
    const int bf = blockIdx.x;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const size_t n0 = (size_t)bf * T_;
    const T* xb = x + n0 * MB_H;
    const T* dyb = dy + n0 * MB_H;
    const T* ob = osave + n0 * MB_H;
    const float* bin = lp.p[P_INP_B];
    const float rs_dh = rsqrtf((float)MB_DH);
    const float qscale = 1.4426950408889634f * rs_dh;

    int tt[MB_NSW];
    bool tv[MB_NSW], sact[MB_NSW];
#pragma unroll
    for (int si = 0; si < MB_NSW; ++si) {
        tt[si] = (w * MB_NSW + si) * 16 + l15;
        tv[si] = tt[si] < T_;
        sact[si] = FULL || (w * MB_NSW + si) < nst;  // wave-uniform
    }
    // dqkv operand of the in_proj weight gradient.  bf16: group-major [12 (q|k|v x head)][N][24], a strip writes 768 contiguous
    // bytes (48-byte pieces of 576-byte token rows were partial-line writes); fp32: token-major [N][3H]
    auto dqkv_row = [&](int grp, size_t n) -> T* {
        return TR ? dqkv + ((size_t)grp * ntok + n) * MB_DH : dqkv + n * (3 * MB_H) + grp * MB_DH;
    };
    // bf16: LN(x) and dy fragments of the wave's strips stay in registers for all four heads (48 VGPRs) instead of being re-read
    // and re-normalised per head (PMC: the per-head re-reads of x, dy missed L2 — 3 GB fetched per launch for 0.6 GB of inputs)
    constexpr bool DYREG = sizeof(T) == 2;
    Frag<T> uf[MB_NSW][MB_KS], dr[MB_NSW][MB_KS];
    float smean[MB_NSW], srstd[MB_NSW];
    // saved attention output / log-sum-exp rows of the NEXT head, requested one head ahead (raw: no conversion, so nothing waits on them)
    RawRow4<T> onx0[MB_NSW], onx1[MB_NSW];
    float lsenx[MB_NSW];
    auto request_o1 = [&](int head, int si) {
        const int tc = tv[si] ? tt[si] : T_ - 1;  // clamped: the padding frames' values are replaced on use
        onx0[si].load(ob + (size_t)tc * MB_H + head * MB_DH + 4 * g4);
        onx1[si].load(ob + (size_t)tc * MB_H + head * MB_DH + 16 + 4 * (g4 & 1));
        lsenx[si] = lse[(n0 + tc) * MB_HEADS + head];
    };
    auto request_o = [&](int head) {
#pragma unroll
        for (int si = 0; si < MB_NSW; ++si) request_o1(head, si);
    };
    if constexpr (DYREG) {
        // Prologue of the bf16 kernel: EVERY global request of the workgroup's start is issued before anything waits — x / dy fragments,
        // the 96 weight fragments (12 16-byte pieces per thread, parked in registers), head 0's saved O rows.  (One workgroup per CU:
        // nothing else hides this latency; the serial form — weights through a load/store loop, then the statistics' reads, then the
        // fragment reads, then O — was 14 % + most of another 18 % of the kernel's wave time, profiles/r03a_phase_prof.txt.)
#pragma unroll
        for (int si = 0; si < MB_NSW; ++si) {
            const int tc = tv[si] ? tt[si] : T_ - 1;
#pragma unroll
            for (int ks = 0; ks < MB_KS; ++ks) {
                frag_load(uf[si][ks], xb + (size_t)tc * MB_H + ks * 32 + 8 * g4);
                frag_load(dr[si][ks], dyb + (size_t)tc * MB_H + ks * 32 + 8 * g4);
            }
        }
        constexpr int NWV = 96 * 64 / MB_NTHR;  // 16-byte pieces per thread
        u32x4 wreg[NWV];
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int v = threadIdx.x + i * MB_NTHR;
            wreg[i] = v < 72 * 64 ? reinterpret_cast<const u32x4*>(Win)[v] : reinterpret_cast<const u32x4*>(WoutT)[v - 72 * 64];
        }
        request_o(0);
        for (int i = threadIdx.x; i < 2 * MB_H; i += MB_NTHR) {
            aff[i] = 0.f;
            lnp[i] = i < MB_H ? lp.p[P_MH_LN_W][i] : lp.p[P_MH_LN_B][i - MB_H];
        }
#pragma unroll
        for (int i = 0; i < NWV; ++i) reinterpret_cast<u32x4*>(wl)[threadIdx.x + i * MB_NTHR] = wreg[i];
        // LayerNorm statistics from the fragments already in registers, then LN(x) in place
#pragma unroll
        for (int si = 0; si < MB_NSW; ++si) {
            float sum = 0.f;
#pragma unroll
            for (int ks = 0; ks < MB_KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += frag_get(uf[si][ks], j);
            smean[si] = wave_sum16(sum) * (1.0f / MB_H);
            float q = 0.f;
#pragma unroll
            for (int ks = 0; ks < MB_KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = frag_get(uf[si][ks], j) - smean[si];
                    q += d * d;
                }
            srstd[si] = rsqrtf(wave_sum16(q) * (1.0f / MB_H) + 1e-5f);
            if (XT && tv[si] && g4 == 0) {
                stats[(n0 + tt[si]) * 2] = smean[si];
                stats[(n0 + tt[si]) * 2 + 1] = srstd[si];
            }
        }
#pragma unroll
        for (int si = 0; si < MB_NSW; ++si)
#pragma unroll
            for (int ks = 0; ks < MB_KS; ++ks) {
                float gm[8], bt[8];
                load8(lp.p[P_MH_LN_W] + ks * 32 + 8 * g4, gm);
                load8(lp.p[P_MH_LN_B] + ks * 32 + 8 * g4, bt);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    frag_set(uf[si][ks], j, tv[si] ? (frag_get(uf[si][ks], j) - smean[si]) * srstd[si] * gm[j] + bt[j] : bt[j]);
                if (!tv[si]) frag_zero(dr[si][ks]);
            }
        lds_barrier();  // weights, lnp, aff are in LDS
        PHASE(0);
    } else {
        for (int i = threadIdx.x; i < 2 * MB_H; i += blockDim.x) aff[i] = 0.f;
        // Only the row statistics persist across the head loop: LN(x) and dy fragments are rebuilt per head and du is formed after
        // the loop from the emitted dqkv operand (keeping them live spilled 336 B/lane in the first version).
        for (int i = threadIdx.x; i < 2 * MB_H; i += blockDim.x) lnp[i] = i < MB_H ? lp.p[P_MH_LN_W][i] : lp.p[P_MH_LN_B][i - MB_H];
#pragma unroll
        for (int si = 0; si < MB_NSW; ++si) {
            float v[MB_KS][8], sum = 0.f;
#pragma unroll
            for (int ks = 0; ks < MB_KS; ++ks) {
                if (tv[si]) load8(xb + (size_t)tt[si] * MB_H + ks * 32 + 8 * g4, v[ks]);
                else
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[ks][j] = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += v[ks][j];
            }
            smean[si] = wave_sum16(sum) * (1.0f / MB_H);
            float q = 0.f;
#pragma unroll
            for (int ks = 0; ks < MB_KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) q += (v[ks][j] - smean[si]) * (v[ks][j] - smean[si]);
            srstd[si] = rsqrtf(wave_sum16(q) * (1.0f / MB_H) + 1e-5f);
            if (XT && tv[si] && g4 == 0) {
                stats[(n0 + tt[si]) * 2] = smean[si];
                stats[(n0 + tt[si]) * 2 + 1] = srstd[si];
            }
        }
        lds_barrier();  // lnp is read below
        PHASE(0);
    }
    for (int head = 0; head < MB_HEADS; ++head) {
        Frag<T> qf[MB_NSW], dof[MB_NSW];
        float Dv[MB_NSW];
        // ---------------- stage A: Q', K, V, dO of this head ----------------
        // all global reads of the head first (x, dy, saved O), ahead of this head's dqkv stores in the vmcnt queue
        float o0[MB_NSW][4], o1[MB_NSW][4], lsev[MB_NSW];
#pragma unroll
        for (int si = 0; si < MB_NSW; ++si) {
#pragma unroll
            for (int ks = 0; ks < MB_KS; ++ks) {
                if (DYREG) continue;
                if (tv[si]) {
                    frag_load(uf[si][ks], xb + (size_t)tt[si] * MB_H + ks * 32 + 8 * g4);
                    frag_load(dr[si][ks], dyb + (size_t)tt[si] * MB_H + ks * 32 + 8 * g4);
                } else {
                    frag_zero(uf[si][ks]);
                    frag_zero(dr[si][ks]);
                }
            }
            // this head's saved O / log2-sum-exp rows: bf16 — requested one head ago; fp32 — requested here (no registers to park them)
            if (!DYREG) request_o1(head, si);
            onx0[si].get(o0[si]);
            onx1[si].get(o1[si]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o0[si][r] = keep_if(tv[si], o0[si][r]);
                o1[si][r] = keep_if(tv[si] && g4 < 2, o1[si][r]);
            }
            lsev[si] = tv[si] ? lsenx[si] : 1e30f;  // padding frame: P = exp2(S - lse) = 0
        }
        if (DYREG) request_o(head + 1 < MB_HEADS ? head + 1 : head);  // (last head: a harmless re-read, keeps the loop body uniform)
#pragma unroll
        for (int si = 0; si < MB_NSW; ++si)
#pragma unroll
            for (int ks = 0; ks < MB_KS; ++ks) {  // LN(x) in place of x (rebuilt per head from the row statistics)
                if (DYREG) continue;
                float gm[8], bt[8];
                load8(lnp + ks * 32 + 8 * g4, gm);
                load8(lnp + MB_H + ks * 32 + 8 * g4, bt);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    frag_set(uf[si][ks], j, tv[si] ? (frag_get(uf[si][ks], j) - smean[si]) * srstd[si] * gm[j] + bt[j] : bt[j]);
            }
#pragma unroll
        for (int which = 0; which < 4; ++which) {  // 0 q, 1 k, 2 v, 3 dO
            f32x4 ct[MB_NSW][2];
#pragma unroll
            for (int si = 0; si < MB_NSW; ++si) ct[si][0] = ct[si][1] = F32X4_ZERO;
#pragma unroll
            for (int ks = 0; ks < MB_KS; ++ks) {
                Frag<T> a[2];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int fi = which < 3 ? ((which * MB_HEADS + head) * 2 + half) * MB_KS + ks : 72 + (head * 2 + half) * MB_KS + ks;
                    if (TR) frag_load(a[half], wl + ((size_t)fi * 64 + lane) * 8);
                    else if (which < 3) wfrag_load(a[half], Win, (which * MB_HEADS + head) * 2 + half, MB_KS, ks);
                    else wfrag_load(a[half], WoutT, head * 2 + half, MB_KS, ks);
                }
#pragma unroll
                for (int si = 0; si < MB_NSW; ++si) {
                    if (!sact[si]) continue;
                    ct[si][0] = mma(a[0], which < 3 ? uf[si][ks] : dr[si][ks], ct[si][0]);
                    ct[si][1] = mma(a[1], which < 3 ? uf[si][ks] : dr[si][ks], ct[si][1]);
                }
            }
            float b0[4] = {0.f, 0.f, 0.f, 0.f}, b1[4] = {0.f, 0.f, 0.f, 0.f};
            if (which < 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    b0[r] = bin[which * MB_H + head * MB_DH + 4 * g4 + r];
                    b1[r] = (16 + 4 * g4 + r < MB_DH) ? bin[which * MB_H + head * MB_DH + 16 + 4 * g4 + r] : 0.f;
                }
            }
#pragma unroll
            for (int si = 0; si < MB_NSW; ++si) {
                if (!sact[si]) continue;
                const int t = tt[si];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ct[si][0][r] += b0[r];
                    ct[si][1][r] += b1[r];
                    if (which == 0) {
                        ct[si][0][r] *= qscale;
                        ct[si][1][r] *= qscale;
                    }
                }
                if (which == 0) {
                    frag_from_c2(qf[si], ct[si][0], ct[si][1]);
                    store_row24<T>(Qr + (size_t)t * MB_DH, ct[si][0], ct[si][1]);
                    if (!TR) store_col24<T>(Qt, tp, t, ct[si][0], ct[si][1]);
                } else if (which == 1) {
                    store_row24<T>(Kr + (size_t)t * MB_DH, ct[si][0], ct[si][1]);
                    if (!TR) store_col24<T>(Kt, tp, t, ct[si][0], ct[si][1]);
                } else if (which == 2) {
                    store_row24<T>(Vr + (size_t)t * MB_DH, ct[si][0], ct[si][1]);
                } else {
                    frag_from_c2(dof[si], ct[si][0], ct[si][1]);
                    store_row24<T>(dOr + (size_t)t * MB_DH, ct[si][0], ct[si][1]);
                    // D = rowsum(dO * O) with the saved forward attention output
                    float dsum = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dsum += round_to(ct[si][0][r], x) * o0[si][r] + round_to(ct[si][1][r], x) * o1[si][r];
                    Dv[si] = wave_sum16(dsum);
                    if (g4 == 0) {
                        Dds[t] = Dv[si];
                        m2s[t] = lsev[si];
                    }
                }
            }
            PHASE(1 + which);
        }
        lds_barrier();
        PHASE(5);

        // ---------------- pass 1: query strips -> dQ ----------------
        // With the forward's log2-sum-exp there is no softmax sweep: P^T = exp2(S^T - lse) tile by tile, dS^T = P^T (dP^T - D),
        // dQ^T += K^T dS^T.  The key-side operands of a tile pair (K, V rows, K^T) are fetched once for both query strips.
        {
            f32x4 dq[MB_NSW][2];
#pragma unroll
            for (int si = 0; si < MB_NSW; ++si) dq[si][0] = dq[si][1] = F32X4_ZERO;
            for (int jp = 0; jp < nkp; ++jp) {
                Frag<T> ak[2], av[2], akt[2];
                const bool hi_valid = 2 * jp + 1 < nst;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int jj = 2 * jp + e;
                    if (jj < nst) {
                        row_pieces<T>(ak[e], Kr + (size_t)(jj * 16 + l15) * MB_DH);
                        row_pieces<T>(av[e], Vr + (size_t)(jj * 16 + l15) * MB_DH);
                    } else {
                        frag_zero(ak[e]);
                        frag_zero(av[e]);
                    }
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if (TR) col_frag_tr(akt[half], Kr, half, jp, hi_valid);
                    else col_frag<T>(akt[half], Kt, tp, half, jp, hi_valid);
                }
#pragma unroll
                for (int si = 0; si < MB_NSW; ++si) {
                    if (!sact[si]) continue;
                    f32x4 ds[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int jj = 2 * jp + e;
                        const f32x4 st = mma(ak[e], qf[si], F32X4_ZERO);   // rows = keys 16 jj + 4 g4 + r, column = query l15
                        const f32x4 dp = mma(av[e], dof[si], F32X4_ZERO);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool kv = jj != nst - 1 || jj * 16 + 4 * g4 + r < T_;  // (only the last key tile has padding keys; they hold finite values: select, do not multiply)
                            const float p = fast_exp2(st[r] - lsev[si]);
                            ds[e][r] = kv ? p * (dp[r] - Dv[si]) : 0.f;
                        }
                    }
                    Frag<T> dsf;
                    frag_from_c2(dsf, ds[0], ds[1]);
#pragma unroll
                    for (int half = 0; half < 2; ++half) dq[si][half] = mma(akt[half], dsf, dq[si][half]);
                }
            }
#pragma unroll
            for (int si = 0; si < MB_NSW; ++si) {
                if (!sact[si] || !tv[si]) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dq[si][0][r] *= rs_dh;
                    dq[si][1][r] *= rs_dh;
                }
                store_row24_nt<T>(dqkv_row(0 * MB_HEADS + head, n0 + tt[si]), dq[si][0], dq[si][1]);
            }
        }
        PHASE(6);
        lds_barrier();
        PHASE(7);

        // ---------------- pass 2: key strips -> dK, dV ----------------
        // The query-side operands of a tile pair (Q', dO rows, their transposes, the row statistics) are fetched once and used
        // for both key strips of the wave: 22 LDS instructions per pair instead of 80.
        {
            Frag<T> kf[MB_NSW], vf[MB_NSW];
            f32x4 dk[MB_NSW][2], dv[MB_NSW][2];
#pragma unroll
            for (int si = 0; si < MB_NSW; ++si) {
                row_pieces<T>(kf[si], Kr + (size_t)tt[si] * MB_DH);
                row_pieces<T>(vf[si], Vr + (size_t)tt[si] * MB_DH);
                dk[si][0] = dk[si][1] = dv[si][0] = dv[si][1] = F32X4_ZERO;
            }
            for (int jp = 0; jp < nkp; ++jp) {
                Frag<T> qa[2], doa[2], aq[2], ado[2];
                f32x4 mls[2], ddv[2];
                const bool hi_valid = 2 * jp + 1 < nst;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int jj = 2 * jp + e;
                    if (jj < nst) {
                        row_pieces<T>(qa[e], Qr + (size_t)(jj * 16 + l15) * MB_DH);
                        row_pieces<T>(doa[e], dOr + (size_t)(jj * 16 + l15) * MB_DH);
                        mls[e] = *reinterpret_cast<const f32x4*>(m2s + jj * 16 + 4 * g4);
                        ddv[e] = *reinterpret_cast<const f32x4*>(Dds + jj * 16 + 4 * g4);
                    } else {
                        frag_zero(qa[e]);
                        frag_zero(doa[e]);
                        mls[e] = (f32x4){1e30f, 1e30f, 1e30f, 1e30f};
                        ddv[e] = F32X4_ZERO;
                    }
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if (TR) {
                        col_frag_tr(ado[half], dOr, half, jp, hi_valid);
                        col_frag_tr(aq[half], Qr, half, jp, hi_valid);
                    } else {
                        col_frag_rm<T>(ado[half], dOr, half, jp, hi_valid);
                        col_frag<T>(aq[half], Qt, tp, half, jp, hi_valid);
                    }
                }
#pragma unroll
                for (int si = 0; si < MB_NSW; ++si) {
                    if (!sact[si]) continue;
                    f32x4 pt[2], dst[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f32x4 sq = mma(qa[e], kf[si], F32X4_ZERO);
                        const f32x4 dp = mma(doa[e], vf[si], F32X4_ZERO);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = fast_exp2(sq[r] - mls[e][r]);
                            pt[e][r] = p;
                            dst[e][r] = p * (dp[r] - ddv[e][r]);
                        }
                    }
                    Frag<T> pf, dsf;
                    frag_from_c2(pf, pt[0], pt[1]);
                    frag_from_c2(dsf, dst[0], dst[1]);
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        dv[si][half] = mma(ado[half], pf, dv[si][half]);
                        dk[si][half] = mma(aq[half], dsf, dk[si][half]);
                    }
                }
            }
#pragma unroll
            for (int si = 0; si < MB_NSW; ++si) {
                if (!sact[si] || !tv[si]) continue;  // (keys beyond T only ever produced their own, discarded, columns)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dk[si][0][r] *= 0.6931471805599453f;  // Q' carries log2(e)/sqrt(dh): dk = dS^T q / sqrt(dh) = dS^T Q' ln2
                    dk[si][1][r] *= 0.6931471805599453f;
                }
                store_row24_nt<T>(dqkv_row(1 * MB_HEADS + head, n0 + tt[si]), dk[si][0], dk[si][1]);
                store_row24_nt<T>(dqkv_row(2 * MB_HEADS + head, n0 + tt[si]), dv[si][0], dv[si][1]);
            }
        }
        PHASE(8);
        lds_barrier();
        PHASE(9);
    }

    if constexpr (XT) {
        PHASE_END();
        return;
    }
    // du = Win^T dqkv from the [N][3H] operand this workgroup has just written (full barrier: the other waves' stores are
    // complete, and these lines were never read before, so no stale L1 copies exist)
    __syncthreads();
    PHASE(10);
    if (TR) {  // in_proj^T fragments (54) replace the forward weights in LDS; everyone is past the head loop
        for (int v = threadIdx.x; v < 54 * 64; v += blockDim.x) reinterpret_cast<u32x4*>(wl)[v] = reinterpret_cast<const u32x4*>(WinT)[v];
        lds_barrier();
    }
    f32x4 du[MB_NSW][BK_MT];
#pragma unroll
    for (int si = 0; si < MB_NSW; ++si) {
#pragma unroll
        for (int mt = 0; mt < BK_MT; ++mt) du[si][mt] = F32X4_ZERO;
        Frag<T> df[3 * MB_H / 32];
#pragma unroll
        for (int k9 = 0; k9 < 3 * MB_H / 32; ++k9) {  // all nine loads in flight together
            const int ch = k9 * 32 + 8 * g4;  // 8-channel pieces never straddle a 24-channel group
            if (tv[si]) frag_load(df[k9], dqkv_row(ch / MB_DH, n0 + tt[si]) + ch % MB_DH);
            else frag_zero(df[k9]);
        }
#pragma unroll
        for (int k9 = 0; k9 < 3 * MB_H / 32; ++k9) {
#pragma unroll
            for (int mt = 0; mt < BK_MT; ++mt) {
                Frag<T> a;
                if (TR) frag_load(a, wl + ((size_t)(mt * (3 * MB_H / 32) + k9) * 64 + lane) * 8);
                else wfrag_load(a, WinT, mt, 3 * MB_H / 32, k9);
                du[si][mt] = mma(a, df[k9], du[si][mt]);
            }
        }
    }

    PHASE(11);
    // ---------------- LayerNorm backward + residual ----------------
    float dlw[BK_MT][4], dlb[BK_MT][4];
#pragma unroll
    for (int mt = 0; mt < BK_MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dlw[mt][r] = dlb[mt][r] = 0.f;
#pragma unroll
    for (int si = 0; si < MB_NSW; ++si) {
        const size_t n = n0 + tt[si];
        ln_bwd_row96<T>(du[si], x + n * MB_H, dy + n * MB_H, dx + n * MB_H, stats + n * 2, tv[si], lp.p[P_MH_LN_W], dlw, dlb);
    }
    ln_affine_flush(dlw, dlb, aff, aff + MB_H);
    lds_barrier();
    PHASE(12);
    for (int i = threadIdx.x; i < 2 * MB_H; i += blockDim.x) part[(size_t)blockIdx.x * 2 * MB_H + i] = aff[i];
    PHASE_END();
}
PHASE_READER(nbss_phase_read_mhsa_bwd)


// ------------------------------------------------------------------------------------------------------------------------------------
// bf16 stream, round 4: ONE sweep, one workgroup per (sequence, HEAD), two workgroups per CU.
//
// The two-sweep kernel above builds every S / dP tile twice (once per operand orientation: 14 MFMAs and 16 exponentials per 32 x 32
// block of scores) and is one 218-register workgroup per CU, parked in s_waitcnt / barriers for half of its wave time.  Here a wave
// owns 32 KEYS (dK, dV and its K, V strips never leave its registers) and walks the queries once:
//     S = Q' K^T - lse,  dP = dO V^T - D   (the subtractions are the accumulators' initial values)
//     P = exp2(S),  dS = P dP,  dV^T += dO^T P,  dK^T += Q'^T dS                     (8 MFMAs, 8 exponentials per 16 x 32 block)
// and hands dS — the only quantity the dQ product needs in the OTHER orientation — to a 64-query LDS image [key][query]; after a
// barrier the eight waves each contract one (16-query strip, dh half) of the chunk over all keys through transposing LDS reads:
//     dQ^T = K^T dS^T   (fixed summation order: no atomics, bitwise repeatable)
// 10 MFMAs per block instead of 14, half the exponentials, and the 72 KB / <= 128 registers of a workgroup let two share a CU.
// The four heads of a sequence are four workgroups on the same XCD (x / dy rows come from its L2); the saved attention output, the
// log2-sum-exp rows and this head's 24 weight fragments (staged in the LDS region that becomes the dS image) are read per head.
// Padding frames (T not a multiple of 16): K rows are zero and lse = 1e30, so they contribute nothing anywhere.
#ifndef MHB_KO
#define MHB_KO 0   // timing knock-outs (A/B flavours, results wrong): 1 = no dqkv stores, 2 = every wave reads the sequence's first x / dy strip
#endif
#define MH_QC 64   // queries per dS chunk
#define MH_RS 68   // dS image row stride (elements): rows 34 dwords apart — the 16 key rows of a b64 store hit 16 distinct bank pairs
#define MH_TP 256

NBSS_DEV void frag_pack_c2(Frag<bf16_t>& f, const f32x4& lo, const f32x4& hi) {  // four v_cvt_pk_bf16_f32
    const u32x4 v = {pack2bf(lo[0], lo[1]), pack2bf(lo[2], lo[3]), pack2bf(hi[0], hi[1]), pack2bf(hi[2], hi[3])};
    f.v = __builtin_bit_cast(s16x8, v);
}

template <bool FULL>
__global__ __launch_bounds__(512, 4) void mhsa_bwd_h_kernel(nbss_cfg c, LayerPtrs lp, int nseq, const bf16_t* __restrict__ Win, const bf16_t* __restrict__ WoutT,
                                                            const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, const bf16_t* __restrict__ osave,
                                                            const float* __restrict__ lse, const float* __restrict__ stats, bf16_t* __restrict__ dqkv, int flip) {
    typedef bf16_t T;
    NBSS_LDS(smem);
    // blocks b, b + 8, b + 16, b + 24 (same XCD, dispatched back to back) = the four heads of one sequence
#ifdef MHB_FLAT_MAP  // (A/B flavour: four consecutive blocks = the four heads of a sequence, on four XCDs)
    const int head = blockIdx.x & 3, bf0 = blockIdx.x >> 2;
#else
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, head = bi & 3, bf0 = (bi >> 2) * 8 + xcd;
#endif
    if (bf0 >= nseq) return;
    const int bf = flip ? nseq - 1 - bf0 : bf0;  // (launch.h: consecutive kernels of a walk traverse the utterances in opposite order)
    const int T_ = c.T, nst = FULL ? MB_NT : cdiv(T_, 16), nkp = FULL ? MB_NT / 2 : cdiv(nst, 2);
    T* Qr = reinterpret_cast<T*>(smem);
    T* Kr = Qr + MH_TP * MB_DH;
    T* dOr = Kr + MH_TP * MB_DH;
    float* nm2 = reinterpret_cast<float*>(dOr + MH_TP * MB_DH);  // -lse (log2 domain) per query; -1e30 for padding frames
    float* nDd = nm2 + MH_TP;                                    // -D = -rowsum(dO * O)
    float* lnp = nDd + MH_TP;                                    // LayerNorm gamma | beta [2 H], this head's q | k | v bias rows [3 dh]
    T* dqs = reinterpret_cast<T*>(lnp + 2 * MB_H + 3 * MB_DH);   // dQ rows of the current chunk [MH_QC][24] (leave as 16-byte pieces of one 3 KB run)
    T* dsb = dqs + MH_QC * MB_DH;                                // dS image of the current chunk [key][MH_RS]
    T* wl = dsb;                                                 // (until the images are built: this head's 24 weight fragments)
    PHASE_BEGIN(dsb + MH_TP * MH_RS);
    const size_t ntok = (size_t)c.B * c.F * T_;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id_u();
    const size_t n0 = (size_t)bf * T_;
    const T* xb = x + n0 * MB_H;
    const T* dyb = dy + n0 * MB_H;
    const T* ob = osave + n0 * MB_H;
    const float* bin = lp.p[P_INP_B];
    const float rs_dh = rsqrtf((float)MB_DH);
    const float qscale = 1.4426950408889634f * rs_dh;

    int tt[2];
    bool tv[2], sact[2];
#pragma unroll
    for (int si = 0; si < 2; ++si) {
        tt[si] = (w * 2 + si) * 16 + l15;
        tv[si] = tt[si] < T_;
        sact[si] = FULL || (w * 2 + si) < nst;  // wave-uniform
    }
    // wave-uniform sequence bases + 32-bit lane offsets (check_cfg: 3 H ntok < 2^31), so that no 64-bit lane addresses stay live
    T* dq_seq = dqkv + n0 * MB_DH;
    const int gstride = (int)ntok * MB_DH;
    auto dqkv_off = [&](int grp, int t) -> int { return grp * gstride + t * MB_DH; };
    const float* lse_seq = lse + n0 * MB_HEADS;
    const float* stats_seq = stats + n0 * 2;

    Frag<T> kf[2], vf[2];
    // ---------------- prologue: EVERY global request first, then LayerNorm, then the four projections of this head ----------------
    // (one memory round trip per workgroup: with two workgroups per CU nothing else hides a second one — the first version asked for
    // gamma / beta after the statistics, for dy / O after the first barrier and for the bias rows inside the projections: 65 % of its wave
    // time, profiles/r04b_phase_mhsa_bwd.txt)
    {
        Frag<T> uf[2][MB_KS], dr[2][MB_KS];
        RawRow4<T> orw0[2], orw1[2];
        float lsev[2];
        f32x2 ms[2];
        u32x4 wreg[3];  // 24 fragments = 1 536 16-byte pieces: in_proj (which, half, ks) of this head, then out_proj^T (half, ks)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int v = threadIdx.x + i * 512, lf = v >> 6, pl = v & 63;
            const T* src = lf < 18 ? Win + ((size_t)(((lf / 6) * MB_HEADS + head) * 6 + lf % 6) * 64 + pl) * 8
                                   : WoutT + ((size_t)(head * 6 + lf - 18) * 64 + pl) * 8;
            wreg[i] = *reinterpret_cast<const u32x4*>(src);
        }
        // LayerNorm gamma | beta (48 four-float pieces) and this head's q | k | v bias rows (18 pieces) go through LDS
        f32x4 ppc = F32X4_ZERO;
        {
            const int t = threadIdx.x;
            const float* src = t < 24 ? lp.p[P_MH_LN_W] + 4 * t : t < 48 ? lp.p[P_MH_LN_B] + 4 * (t - 24)
                                      : bin + ((t - 48) / 6) * MB_H + head * MB_DH + 4 * ((t - 48) % 6);
            if (t < 66) ppc = *reinterpret_cast<const f32x4*>(src);
        }
        // Global rows are read as 16-byte pieces of CONTIGUOUS runs (a strip of x / dy is 3 KB, a strip of this head's O columns 16 x 48 B) and
        // turned into MFMA fragments through a wave-private LDS round trip: the direct fragment form (lane = token l15, piece g4) makes the four
        // lanes of every quad touch four different rows — 64 cache-line lookups per wave instruction instead of 8, and the vector memory pipe,
        // not VALU or MFMA, was what this kernel waited for (TCP_TOTAL_CACHE_ACCESSES: 12 bytes per access, profiles/README.md round 4).
        u32x4 xc[2][MB_KS], dc[2][MB_KS], oc[2];
        const int tlast = T_ - 1;
#pragma unroll
        for (int si = 0; si < 2; ++si) {
            const int t0 = (w * 2 + si) * 16;
#pragma unroll
            for (int i = 0; i < MB_KS; ++i) {
#ifdef MH_STAGE_X  // piece p = 64 i + lane of the strip's 192: row p / 12, 16-byte column p % 12
                const int pc = i * 64 + lane, r = pc / 12, cc = pc - r * 12;
                const int tr = t0 + r < tlast ? t0 + r : tlast;  // clamped: the padding frames' values are replaced on use
                xc[si][i] = *reinterpret_cast<const u32x4*>(xb + (tr * MB_H + cc * 8));
                dc[si][i] = *reinterpret_cast<const u32x4*>(dyb + (tr * MB_H + cc * 8));
#else  // fragments straight from global memory (lane = token l15, piece g4)
                const int tc = (MHB_KO & 2) ? l15 : tv[si] ? tt[si] : tlast;
                frag_load(uf[si][i], xb + (tc * MB_H + i * 32 + 8 * g4));
                frag_load(dr[si][i], dyb + (tc * MB_H + i * 32 + 8 * g4));
#endif
            }
        }
#pragma unroll
        for (int si = 0; si < 2; ++si) {
            const int t0 = (w * 2 + si) * 16;
            const int lo = lane < 48 ? lane : 47, r = lo / 3, cc = lo - r * 3;  // 48 pieces: row l / 3, 16-byte column l % 3 of the head's 48 bytes
            const int tr = t0 + r < tlast ? t0 + r : tlast;
            oc[si] = *reinterpret_cast<const u32x4*>(ob + (tr * MB_H + head * MB_DH + cc * 8));
            const int tc = tv[si] ? tt[si] : tlast;
            lsev[si] = lse_seq[tc * MB_HEADS + head];
            ms[si] = *reinterpret_cast<const f32x2*>(stats_seq + tc * 2);
        }
        PHASE(8);
#pragma unroll
        for (int i = 0; i < 3; ++i) reinterpret_cast<u32x4*>(wl)[threadIdx.x + i * 512] = wreg[i];
        PHASE(9);
        if (threadIdx.x < 66) *reinterpret_cast<f32x4*>(lnp + 4 * threadIdx.x) = ppc;
        // wave-private staging: this wave's own rows of the Q' | K images (2 x 1.5 KB = one strip of x or dy) and of the dO image (O rows);
        // nobody else touches them before the barrier that follows the projections
        {
            T* sA = Qr + w * 32 * MB_DH;
            T* sB = Kr + w * 32 * MB_DH;
            T* sO = dOr + w * 32 * MB_DH;
            auto stg = [&](int pc) -> T* { return (pc < 96 ? sA : sB - 96 * 8) + pc * 8; };  // piece pc of a strip
            auto fr = [&](int ks) -> const T* { const int pc = l15 * 12 + ks * 4 + g4; return (pc < 96 ? sA : sB - 96 * 8) + pc * 8; };
#ifdef MH_STAGE_X
#pragma unroll
            for (int which = 0; which < 2; ++which)
#pragma unroll
                for (int si = 0; si < 2; ++si) {
#pragma unroll
                    for (int i = 0; i < MB_KS; ++i) *reinterpret_cast<u32x4*>(stg(i * 64 + lane)) = which ? dc[si][i] : xc[si][i];
                    wave_lds_sync();
#pragma unroll
                    for (int ks = 0; ks < MB_KS; ++ks) frag_load(which ? dr[si][ks] : uf[si][ks], fr(ks));
                    wave_lds_sync();
                }
#endif
#pragma unroll
            for (int si = 0; si < 2; ++si)
                if (lane < 48) *reinterpret_cast<u32x4*>(sO + si * 16 * MB_DH + lane * 8) = oc[si];
            wave_lds_sync();
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                orw0[si].load(sO + (si * 16 + l15) * MB_DH + 4 * g4);
                orw1[si].load(sO + (si * 16 + l15) * MB_DH + 16 + 4 * (g4 & 1));
            }
            wave_lds_sync();
        }
        // LayerNorm statistics: the forward pass left (mean, rstd) of every token in the save buffer (layout.h: mhsa_stat_offset) — recomputing
        // them was 4 of the prologue's 8.5 VALU operations per element, in a kernel whose VALU is busier than its MFMA pipe
        float mean[2], rsv[2];
#pragma unroll
        for (int si = 0; si < 2; ++si) {
            mean[si] = ms[si][0];
            rsv[si] = keep_if(tv[si], ms[si][1]);  // padding frames (a clamped, finite row was loaded): LN(x) = beta without a branch per element
        }
        PHASE(10);
        lds_barrier();  // the weight fragments and the parameter rows are in LDS
        PHASE(0);
#pragma unroll
        for (int si = 0; si < 2; ++si) {
            const short dmask = tv[si] ? (short)-1 : (short)0;
#pragma unroll
            for (int ks = 0; ks < MB_KS; ++ks) {
                float gm[8], bt[8];
                load8(lnp + ks * 32 + 8 * g4, gm);
                load8(lnp + MB_H + ks * 32 + 8 * g4, bt);
#pragma unroll
                for (int j = 0; j < 8; ++j) frag_set(uf[si][ks], j, (frag_get(uf[si][ks], j) - mean[si]) * rsv[si] * gm[j] + bt[j]);
                dr[si][ks].v &= dmask;  // dy = 0 for padding frames
            }
            sched_fence();
        }
#pragma unroll
        for (int which = 0; which < 4; ++which) {  // 0 q, 1 k, 2 v, 3 dO
            f32x4 ct[2][2];
#pragma unroll
            for (int si = 0; si < 2; ++si) ct[si][0] = ct[si][1] = F32X4_ZERO;
#pragma unroll
            for (int ks = 0; ks < MB_KS; ++ks) {
                Frag<T> a[2];
#pragma unroll
                for (int half = 0; half < 2; ++half) frag_load(a[half], wl + (((which * 6 + half * MB_KS + ks) * 64 + lane) * 8));
#pragma unroll
                for (int si = 0; si < 2; ++si) {
                    if (!sact[si]) continue;
                    ct[si][0] = mma(a[0], which < 3 ? uf[si][ks] : dr[si][ks], ct[si][0]);
                    ct[si][1] = mma(a[1], which < 3 ? uf[si][ks] : dr[si][ks], ct[si][1]);
                }
            }
            float b0[4] = {0.f, 0.f, 0.f, 0.f}, b1[4] = {0.f, 0.f, 0.f, 0.f};
            if (which < 3) {
                load4(lnp + 2 * MB_H + which * MB_DH + 4 * g4, b0);
                load4(lnp + 2 * MB_H + which * MB_DH + 16 + 4 * (g4 & 1), b1);
#pragma unroll
                for (int r = 0; r < 4; ++r) b1[r] = keep_if(g4 < 2, b1[r]);
            }
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                if (!sact[si]) continue;
                const int t = tt[si];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ct[si][0][r] += b0[r];
                    ct[si][1][r] += b1[r];
                    if (which == 0) {
                        ct[si][0][r] *= qscale;
                        ct[si][1][r] *= qscale;
                    }
                    if (which == 1) {  // padding frames: zero keys (their dS columns then vanish from dQ)
                        ct[si][0][r] = keep_if(tv[si], ct[si][0][r]);
                        ct[si][1][r] = keep_if(tv[si], ct[si][1][r]);
                    }
                }
                if (which == 0) {
                    store_row24<T>(Qr + t * MB_DH, ct[si][0], ct[si][1]);
                } else if (which == 1) {
                    frag_pack_c2(kf[si], ct[si][0], ct[si][1]);
                    store_row24<T>(Kr + t * MB_DH, ct[si][0], ct[si][1]);
                } else if (which == 2) {
                    frag_pack_c2(vf[si], ct[si][0], ct[si][1]);
                } else {
                    store_row24<T>(dOr + t * MB_DH, ct[si][0], ct[si][1]);
                    // D = rowsum(dO * O) with the saved forward attention output
                    float o0[4], o1[4];
                    orw0[si].get(o0);
                    orw1[si].get(o1);
                    float dsum = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dsum += round_to(ct[si][0][r], x) * keep_if(tv[si], o0[r]) + round_to(ct[si][1][r], x) * keep_if(tv[si] && g4 < 2, o1[r]);
                    const float Dv = wave_sum16(dsum);
                    if (g4 == 0) {
                        nDd[t] = -Dv;
                        nm2[t] = tv[si] ? -lsev[si] : -1e30f;  // padding frame: P = exp2(S - 1e30) = 0
                    }
                }
            }
        }
    }
    PHASE(1);
    lds_barrier();  // images complete; nobody reads the weight fragments any more (the dS image takes their place)
    PHASE(2);

    f32x4 dk[2][2], dv[2][2];
#pragma unroll
    for (int si = 0; si < 2; ++si) dk[si][0] = dk[si][1] = dv[si][0] = dv[si][1] = F32X4_ZERO;
    const int qs = w >> 1, qh = w & 1;  // dQ role: query strip of the chunk, dh half
    const int nch = FULL ? MH_TP / MH_QC : cdiv(nkp, 2);
    for (int ch = 0; ch < nch; ++ch) {
        // ---------------- key side: this wave's 32 keys x the chunk's 64 queries ----------------
#pragma unroll
        for (int jl = 0; jl < 2; ++jl) {
            const int jp = 2 * ch + jl;
            if (!FULL && jp >= nkp) break;
            Frag<T> qa[2], doa[2], aq[2], ado[2];
            f32x4 nml[2], ndd[2];
            const bool hi_valid = FULL || 2 * jp + 1 < nst;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int jj = 2 * jp + e;
                if (FULL || jj < nst) {
                    row_pieces<T>(qa[e], Qr + (jj * 16 + l15) * MB_DH);
                    row_pieces<T>(doa[e], dOr + (jj * 16 + l15) * MB_DH);
                    nml[e] = *reinterpret_cast<const f32x4*>(nm2 + jj * 16 + 4 * g4);
                    ndd[e] = *reinterpret_cast<const f32x4*>(nDd + jj * 16 + 4 * g4);
                } else {
                    frag_zero(qa[e]);
                    frag_zero(doa[e]);
                    nml[e] = (f32x4){-1e30f, -1e30f, -1e30f, -1e30f};
                    ndd[e] = F32X4_ZERO;
                }
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                col_frag_tr(ado[half], dOr, half, jp, hi_valid);
                col_frag_tr(aq[half], Qr, half, jp, hi_valid);
            }
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                if (!sact[si]) continue;
                f32x4 pt[2], dst[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const f32x4 sq = mma(qa[e], kf[si], nml[e]);   // rows = queries 16 jj + 4 g4 + r, column = key l15
                    const f32x4 dp = mma(doa[e], vf[si], ndd[e]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = fast_exp2(sq[r]);
                        pt[e][r] = p;
                        dst[e][r] = p * dp[r];
                    }
                }
                Frag<T> pf, dsf;
                frag_pack_c2(pf, pt[0], pt[1]);
                frag_pack_c2(dsf, dst[0], dst[1]);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    dv[si][half] = mma(ado[half], pf, dv[si][half]);
                    dk[si][half] = mma(aq[half], dsf, dk[si][half]);
                }
                // dS[key][query]: this lane's key row, queries 4 g4 + r of the pair's two tiles
                const u32x4 dsv = __builtin_bit_cast(u32x4, dsf.v);
                T* drow = dsb + (tt[si] * MH_RS + jl * 32 + 4 * g4);
                *reinterpret_cast<u32x2*>(drow) = (u32x2){dsv[0], dsv[1]};
                *reinterpret_cast<u32x2*>(drow + 16) = (u32x2){dsv[2], dsv[3]};
            }
        }
        PHASE(3);
        lds_barrier();  // the chunk's dS image is complete
        PHASE(4);
        // ---------------- query side: dQ^T (this wave: strip qs of the chunk, dh half qh) over all keys ----------------
        if (FULL || 4 * ch + qs < nst) {
            f32x4 acc[2] = {F32X4_ZERO, F32X4_ZERO};
#pragma unroll 4
            for (int kb = 0; kb < nkp; ++kb) {
                const bool hi_valid = FULL || 2 * kb + 1 < nst;
                Frag<T> akt, dst;
                col_frag_tr(akt, Kr, qh, kb, hi_valid);
                const T* p = dsb + ((kb * 32 + 4 * g4 + (l15 >> 2)) * MH_RS + qs * 16 + 4 * (l15 & 3));
                const u32x2 lo = lds_tr4_b16(p);
                u32x2 hi = {0u, 0u};
                if (hi_valid) hi = lds_tr4_b16(p + 16 * MH_RS);
                const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
                dst.v = __builtin_bit_cast(s16x8, v);
                acc[kb & 1] = mma(akt, dst, acc[kb & 1]);
            }
            if (qh == 0 || g4 < 2)
                store4(dqs + ((qs * 16 + l15) * MB_DH + qh * 16 + 4 * g4), (acc[0][0] + acc[1][0]) * rs_dh, (acc[0][1] + acc[1][1]) * rs_dh,
                       (acc[0][2] + acc[1][2]) * rs_dh, (acc[0][3] + acc[1][3]) * rs_dh);
        }
        PHASE(5);
        lds_barrier();  // dQ rows staged; the dS image may be overwritten
        PHASE(6);
        {   // the chunk's dQ rows are one 3 KB run of the group-major operand: 24 16-byte pieces per wave
            const int pc = w * 24 + lane, tq = ch * MH_QC + pc / 3;
            if (lane < 24 && tq < T_ && !(MHB_KO & 1 && nseq > 0)) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(dqs + pc * 8);
                store16_nt(dq_seq + (dqkv_off(head, ch * MH_QC) + pc * 8), v);
            }
        }
    }
    {   // dK, dV rows of this wave's 32 keys: through its own rows of the Q' / dO images (dead since the last key-side pass; the barrier after it
        // has been passed by everyone) and out as 16-byte pieces of one 1.5 KB run each
        T* sK = Qr + w * 32 * MB_DH;
        T* sV = dOr + w * 32 * MB_DH;
#pragma unroll
        for (int si = 0; si < 2; ++si) {
            if (!sact[si]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dk[si][0][r] *= 0.6931471805599453f;  // Q' carries log2(e)/sqrt(dh): dk = dS^T q / sqrt(dh) = dS^T Q' ln2
                dk[si][1][r] *= 0.6931471805599453f;
            }
            store_row24<T>(sK + (si * 16 + l15) * MB_DH, dk[si][0], dk[si][1]);
            store_row24<T>(sV + (si * 16 + l15) * MB_DH, dv[si][0], dv[si][1]);
        }
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 2; ++i) {  // 96 pieces per tensor: row pc / 3
            const int pc = i * 64 + lane, t = w * 32 + pc / 3;
            if (pc < 96 && t < T_ && !(MHB_KO & 1 && nseq > 0)) {  // (padding keys only ever produced their own, discarded, rows)
                const u32x4 vk = *reinterpret_cast<const u32x4*>(sK + pc * 8), vv = *reinterpret_cast<const u32x4*>(sV + pc * 8);
                store16_nt(dq_seq + (dqkv_off(1 * MB_HEADS + head, w * 32) + pc * 8), vk);
                store16_nt(dq_seq + (dqkv_off(2 * MB_HEADS + head, w * 32) + pc * 8), vv);
            }
        }
    }
    PHASE(7);
    PHASE_END();
}

int tailw_mhsa(const nbss_cfg& c, const LayerPtrs& lp, const void* packed, int layer, const void* x, const void* dy, void* dx, float* stats,
               const void* dqkv, float* wgpart, float* G, hipStream_t st, const Side* sd, hipStream_t* gs);

template <class T, bool FULL, bool XT>
static int mhsa_bwd_t(const nbss_cfg& c, const float* P, float* part, const void* packed, int layer, const void* x, const void* dy, const void* osave,
                      void* dx, float* stats, void* dqkv, hipStream_t st) {
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    const int tp = cdiv(c.T, 16) * 16;
    if (tp > 256) return NBSS_EUNSUPPORTED;
    const size_t lds = (size_t)(sizeof(T) == 2 ? 4 : 6) * tp * MB_DH * sizeof(T) + (size_t)(3 * tp + 4 * MB_H) * sizeof(float) + 64 + (sizeof(T) == 2 ? (size_t)96 * 512 * sizeof(T) : 0) + PHASE_LDS_BYTES;
    if (lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    const T* pk = (const T*)packed;
    int e = NBSS_SET_MAX_LDS((mhsa_bwd_kernel<T, FULL, XT>), lds);
    if (e) return e;
    dim3 grid(c.B * c.F), block(MB_NTHR);  // (timed by the caller's ProfScope, together with the fused tail kernel when there is one)
    NBSS_LAUNCH((mhsa_bwd_kernel<T, FULL, XT>), grid, block, lds, st, c, lp, P, part, layer, pk + pack_off(c, layer, K_INP), pk + pack_off(c, layer, K_INP_TN),
                pk + pack_off(c, layer, K_OUTP_T), (const T*)x, (const T*)dy, (const T*)osave, (const float*)((const char*)osave + mhsa_lse_offset(c)),
                (T*)dx, stats, (T*)dqkv);
    return NBSS_CHECK_LAUNCH();
}

template <bool FULL>
static int mhsa_bwd_h_t(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, const void* dy, const void* osave, const float* stats,
                        void* dqkv, hipStream_t st) {
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    if (c.T > MH_TP) return NBSS_EUNSUPPORTED;
    const size_t lds = (size_t)3 * MH_TP * MB_DH * sizeof(bf16_t) + (size_t)(2 * MH_TP + 2 * MB_H + 3 * MB_DH) * sizeof(float) + (size_t)(MH_QC * MB_DH + MH_TP * MH_RS) * sizeof(bf16_t) + PHASE_LDS_BYTES;  // 76 KB: two per CU
    const bf16_t* pk = (const bf16_t*)packed;
    int e = NBSS_SET_MAX_LDS((mhsa_bwd_h_kernel<FULL>), lds);
    if (e) return e;
    const int nseq = c.B * c.F;
    dim3 grid(cdiv(nseq, 8) * 8 * MB_HEADS), block(512);
    NBSS_LAUNCH((mhsa_bwd_h_kernel<FULL>), grid, block, lds, st, c, lp, nseq, pk + pack_off(c, layer, K_INP), pk + pack_off(c, layer, K_OUTP_T), (const bf16_t*)x,
                (const bf16_t*)dy, (const bf16_t*)osave, (const float*)((const char*)osave + mhsa_lse_offset(c)), stats, (bf16_t*)dqkv, walk_flip_next());
    return NBSS_CHECK_LAUNCH();
}

int mhsa_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, int layer, const void* x, const void* dy, const void* osave,
                  void* dx, void* ws, hipStream_t st, const Side* sd) {
    if (c.H != MB_H) return gb_mhsa_bwd(c, P, G, layer, x, dy, dx, ws, st, sd);
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    const size_t N = (size_t)c.B * c.F * c.T;
    float* stats = (float*)ws;
    void* dqkv = (char*)ws + ws_align(N * 2 * sizeof(float));
    float* part = (float*)((char*)ws + ws_part_offset(c));
    const bool full = cdiv(c.T, 16) == MB_NT;
#if defined(NBSS_NO_TAILW) || defined(NBSS_NO_TAILW_MHSA)
    const bool xt = false;
#else
    const bool xt = c.dtype == NBSS_BF16;  // tail + in_proj weight gradient in tailw.hip
#endif
#ifndef NBSS_MHSA_BWD_V1
    // single-sweep kernel: the LayerNorm row statistics are the forward pass's (save buffer), for it and for the tail kernel
    if (xt) stats = (float*)((char*)osave + mhsa_stat_offset(c));
#endif
    int e;
    hipStream_t gs = st;  // parameter-gradient launches (side.h)
    FoldScope fs(st, (char*)ws + ws_wgpart_offset(c), WGPART_BYTES, N);  // (fold.h: the sub-block's folds leave as one launch per stage, on the gradient stream)
    {
    ProfScope ps(PK_MHSA_B, st);  // ONE profiler interval per nbss_mhsa_bwd call: data-gradient kernel (+ fused tail / in_proj wgrad kernel)
    e = c.dtype != NBSS_BF16 ? mhsa_bwd_t<float, false, false>(c, P, part, packed, layer, x, dy, osave, dx, stats, dqkv, st)
#ifndef NBSS_MHSA_BWD_V1  // (A/B flavour: the two-sweep kernel of rounds 1-3)
            : xt ? (full ? mhsa_bwd_h_t<true>(c, P, packed, layer, x, dy, osave, stats, dqkv, st) : mhsa_bwd_h_t<false>(c, P, packed, layer, x, dy, osave, stats, dqkv, st))
#endif
            : xt ? (full ? mhsa_bwd_t<bf16_t, true, true>(c, P, part, packed, layer, x, dy, osave, dx, stats, dqkv, st)
                         : mhsa_bwd_t<bf16_t, false, true>(c, P, part, packed, layer, x, dy, osave, dx, stats, dqkv, st))
            : full ? mhsa_bwd_t<bf16_t, true, false>(c, P, part, packed, layer, x, dy, osave, dx, stats, dqkv, st)
                   : mhsa_bwd_t<bf16_t, false, false>(c, P, part, packed, layer, x, dy, osave, dx, stats, dqkv, st);
    if (e) return e;
    if (xt && (e = tailw_mhsa(c, lp, packed, layer, x, dy, dx, stats, dqkv, (float*)((char*)ws + ws_wgpart_offset(c)), G, st, sd, &gs))) return e;
    }
    if (!xt) gs = side_fork(sd, st);
    if (!xt) {
        AffSegs sg;
        sg.n = 2;
        sg.off[0] = param_off(c, layer, P_MH_LN_W); sg.cnt[0] = MB_H;
        sg.off[1] = param_off(c, layer, P_MH_LN_B); sg.cnt[1] = MB_H;
        if ((e = affine_reduce_launch(part, c.B * c.F, sg, G, gs))) return e;
    }
    WgradArgs a;
    a.part = (float*)((char*)ws + ws_wgpart_offset(c));
    a.mvalid = 0; a.nvalid = 0;
    a.Ntok = (int)N; a.F = c.F; a.T = c.T; a.shift_stride = 1; a.shift_dim = 0; a.groups = 1; a.taps = 1;
    // out_proj: dWo[H][H] = dy^T O ; dbo = colsum(dy)
    a.A = dy; a.lda = MB_H; a.MA = MB_H; a.B = osave; a.ldb = MB_H; a.NB = MB_H;
    a.stats = nullptr; a.gamma = nullptr; a.beta = nullptr;
    a.dW = G + param_off(c, layer, P_OUTP_W); a.dbias = G + param_off(c, layer, P_OUTP_B);
    if ((e = wgrad_launch(a, c.dtype, gs))) return e;
    if (xt) return fs.end();
    // in_proj: dWin[3H][H] = dqkv^T LN(x) ; dbin = colsum(dqkv)
    a.A = dqkv; a.lda = 3 * MB_H; a.MA = 3 * MB_H; a.B = x; a.ldb = MB_H; a.NB = MB_H;
    if (c.dtype == NBSS_BF16) { a.a_gw = MB_DH; a.a_gs = (int)(N * MB_DH); }  // group-major dqkv
    a.stats = stats; a.gamma = lp.p[P_MH_LN_W]; a.beta = lp.p[P_MH_LN_B];
    a.dW = G + param_off(c, layer, P_INP_W); a.dbias = G + param_off(c, layer, P_INP_B);
    if ((e = wgrad_launch(a, c.dtype, gs))) return e;
    return fs.end();
}
