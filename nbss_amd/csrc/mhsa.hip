// mhsa.hip — narrow-band multi-head self-attention module of SpatialNetLayer:
//   y = x + out_proj(softmax(q k^T / sqrt(dh)) v),  [q,k,v] = in_proj(LayerNorm_H(x))
// over the T frames of every (b,f) sequence (SpatialNet.py:88,93-100; nn.MultiheadAttention,
// 4 heads, dh = 24, no mask, no dropout, need_weights always False).
//
// One workgroup = one (b,f) sequence (T <= 256 frames, 48 KB of bf16 stream).  K and V of every head are LDS resident
// (bf16: both row-major, V fetched through transposing reads; fp32: V^T), the in_proj / out_proj fragments go through a
// 48-fragment LDS window; Q, the probabilities and the per-head outputs never leave registers.  In training the kernel also
// saves the attention output before out_proj and the log2-sum-exp of every score row (nbss_mhsa_save_bytes):
//   * all projections are "form 2" GEMMs (weights = MFMA A operand, frames = N), so a C tile
//     holds 4 consecutive channels of one frame per lane;
//   * S^T = K Q^T is computed (not S), which makes the C tiles of S^T directly the B operand of
//     O^T = V^T P^T and keeps the softmax row (all keys of one query) inside 4 lanes:
//     row max / sum = in-lane reduce + two xor-shuffles (16, 32);
//   * two stacked C tiles are re-used as the next B operand in the permuted K order
//     (common.h), which is how Q^T feeds S^T and O^T feeds the output projection.
// T = 251 is padded to 16 tiles of 16 keys; padded keys are masked to -inf.
#include "launch.h"
#include "layout.h"
#include "prof.h"
#include "geom.h"
#include "side.h"

#define MH_H 96
#define MH_HEADS 4
#define MH_DH 24
#define MH_TP 256
#define MH_NT 16       // key/query tiles of 16 frames
// strips per wave: template parameter NSW (2 = 8 waves x 2 strips; 1 = 16 waves x 1 strip)
#define MH_KS (MH_H / 32)
#ifndef MH_BF16_NSW
#define MH_BF16_NSW 2
#endif
#ifndef MH_BF16_HPP
#define MH_BF16_HPP 4  // heads per pass of the bf16 kernel (A/B: 2 = two passes of a head pair, K / V images of 49 KB)
#endif

// LN of a 16-frame strip held as natural-order B fragments (lane: frame l&15, channels 32ks+8g+j)
template <class T>
NBSS_DEV void ln_strip(const T* __restrict__ xr, bool valid, const float (&gam)[MH_KS][8], const float (&bet)[MH_KS][8], Frag<T> (&u)[MH_KS], float* __restrict__ stat = nullptr) {
    const int g4 = lane_id() >> 4;
    float v[MH_KS][8];
    float sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < MH_KS; ++ks) {
        if (valid) load8(xr + ks * 32 + 8 * g4, v[ks]);
        else
#pragma unroll
            for (int j = 0; j < 8; ++j) v[ks][j] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[ks][j];
    }
    const float mean = wave_sum16(sum) * (1.0f / MH_H);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < MH_KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = v[ks][j] - mean;
            q += d * d;
        }
    const float rstd = rsqrtf(wave_sum16(q) * (1.0f / MH_H) + 1e-5f);
    if (stat && valid && g4 == 0) {  // training-mode forward: the row statistics go to the save buffer (layout.h: mhsa_stat_offset)
        stat[0] = mean;
        stat[1] = rstd;
    }
#pragma unroll
    for (int ks = 0; ks < MH_KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) frag_set(u[ks], j, (v[ks][j] - mean) * rstd * gam[ks][j] + bet[ks][j]);
}

// A operand of O^T = V^T P^T from the row-major bf16 V image [TP][24]: rows d = half*16 + l15, K = the 32 frames of k-step ks
NBSS_DEV void v_frag_tr(Frag<bf16_t>& f, const bf16_t* __restrict__ vr, int half, int ks) {
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    frag_load_tr(f, vr + (size_t)(ks * 32 + 4 * g4 + (l15 >> 2)) * MH_DH + half * 16 + 4 * (l15 & 3), MH_DH);
}
NBSS_DEV void v_frag_tr(Frag<float>&, const float*, int, int) {}

// FULL: T in (240, 256]: every key tile exists, the tile tests fold at compile time (see mhsa_bwd.hip)
template <class T, int HPP, bool FULL, int NSW>
__global__ __launch_bounds__(64 * 16 / NSW, NSW == 4 ? 2 : 1) void mhsa_fwd_kernel(nbss_cfg c, const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                       const float* __restrict__ bin, const float* __restrict__ bout,
                                                       const T* __restrict__ Win, const T* __restrict__ Wout,
                                                       const T* __restrict__ x, T* __restrict__ y, T* __restrict__ osave, float* __restrict__ lse, int bf0, int flip) {
    NBSS_LDS(smem);
    T* Ks = reinterpret_cast<T*>(smem);              // [HPP][TP][DH]
    T* Vt = Ks + HPP * MH_TP * MH_DH;                // [HPP][DH][TP]
    // bf16 (all heads in one pass): weight fragments go through a 48-fragment LDS window — Q|K in_proj rows first, then
    // V in_proj | out_proj — instead of per-wave global fragment reads (the ISA had ~60 exposed vmcnt waits between MFMAs)
    constexpr bool WLDS = sizeof(T) == 2 && HPP == MH_HEADS;
    T* wl = Vt + HPP * MH_TP * MH_DH + 32;           // [48][512]
    float* prm = reinterpret_cast<float*>(wl + (WLDS ? 48 * 512 : 0));  // [3H in_proj bias | H out_proj bias | 2H LN gamma, beta]
    const int T_ = c.T, nst = FULL ? MH_NT : cdiv(T_, 16);
    const int bf = flip_bid(flip) + bf0;  // (bf0: first sequence of this launch — the walk's tail launch, side.h: SeqTail; flip: launch.h)
    const int tid = threadIdx.x, lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const T* xb = x + (size_t)bf * T_ * MH_H;
    T* yb = y + (size_t)bf * T_ * MH_H;
    const float qscale = 1.4426950408889634f * rsqrtf((float)MH_DH);
    float* lnstat = osave ? reinterpret_cast<float*>(reinterpret_cast<char*>(osave) + mhsa_stat_offset(c)) : nullptr;

    float gam[MH_KS][8], bet[MH_KS][8];
#pragma unroll
    for (int ks = 0; ks < MH_KS; ++ks) {
        load8(lnw + ks * 32 + 8 * g4, gam[ks]);
        load8(lnb + ks * 32 + 8 * g4, bet[ks]);
    }
    constexpr int NTHR = 64 * 16 / NSW, WPT = 3072 / NTHR;
    u32x4 wr[WPT];  // this thread's share of a 48-fragment weight window
    auto wwin_load = [&](const T* s0, const T* s1) {  // two runs of 24 fragments (1536 16-byte vectors) each
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int v = tid + i * NTHR;
            wr[i] = *reinterpret_cast<const u32x4*>((v < 1536 ? s0 : s1 - 1536 * 8) + (size_t)v * 8);
        }
    };
    auto wwin_store = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i) reinterpret_cast<u32x4*>(wl)[tid + i * NTHR] = wr[i];
    };
    if (WLDS) {
        wwin_load(Win, Win + 24 * 512);  // Q rows | K rows (24 fragments each: 4 heads x 2 halves x 3 k-steps)
        for (int i = tid; i < 4 * MH_H; i += blockDim.x) prm[i] = i < 3 * MH_H ? bin[i] : bout[i - 3 * MH_H];
    }

    for (int pass = 0; pass < MH_HEADS / HPP; ++pass) {
        // (all 16 strips are always projected, so every K / V^T entry is (re)written each pass and
        //  padded frames hold finite values: their keys are masked and their probabilities are 0)
        if (pass > 0) lds_barrier();

        // ---- stage A: LN, then Q (registers), K and V^T (LDS) for this pass's heads --------
        Frag<T> qf[NSW][HPP];
        {
            Frag<T> u[NSW][MH_KS];
#pragma unroll
            for (int si = 0; si < NSW; ++si) {
                const int t = (w * NSW + si) * 16 + l15;
                ln_strip<T>(xb + (size_t)t * MH_H, t < T_, gam, bet, u[si], lnstat ? lnstat + ((size_t)bf * T_ + t) * 2 : nullptr);
            }
            if (WLDS) {
                wwin_store();
                lds_barrier();
                wwin_load(Win + 48 * 512, Wout);  // V rows | out_proj: in flight during the Q and K projections
            }
#pragma unroll
            for (int which = 0; which < 3; ++which) {
                if (WLDS && which == 2) {
                    lds_barrier();  // everyone is done with the Q|K window
                    wwin_store();
                    lds_barrier();
                }
#pragma unroll
                for (int hh = 0; hh < HPP; ++hh) {
                    const int head = pass * HPP + hh;
                    f32x4 ct[NSW][2];
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        Frag<T> a[MH_KS];
#pragma unroll
                        for (int ks = 0; ks < MH_KS; ++ks) {
                            if (WLDS) frag_load(a[ks], wl + ((size_t)(((which & 1) * MH_HEADS + head) * 2 + half) * MH_KS + ks) * 512 + lane * 8);
                            else wfrag_load(a[ks], Win, (which * MH_HEADS + head) * 2 + half, MH_KS, ks);
                        }
#pragma unroll
                        for (int si = 0; si < NSW; ++si) {
                            f32x4 acc = F32X4_ZERO;
#pragma unroll
                            for (int ks = 0; ks < MH_KS; ++ks) acc = mma(a[ks], u[si][ks], acc);
                            ct[si][half] = acc;
                        }
                    }
                    // bias (rows d >= DH are padding: weight rows are zero, keep the bias zero too)
                    float b0[4], b1[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float* bsrc = WLDS ? prm : bin;
                        b0[r] = bsrc[which * MH_H + head * MH_DH + 4 * g4 + r];
                        b1[r] = (16 + 4 * g4 + r < MH_DH) ? bsrc[which * MH_H + head * MH_DH + 16 + 4 * g4 + r] : 0.f;
                    }
#pragma unroll
                    for (int si = 0; si < NSW; ++si) {
                        const int t = (w * NSW + si) * 16 + l15;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ct[si][0][r] += b0[r];
                            ct[si][1][r] += b1[r];
                        }
                        if (which == 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                ct[si][0][r] *= qscale;
                                ct[si][1][r] *= qscale;
                            }
                            frag_from_c2(qf[si][hh], ct[si][0], ct[si][1]);
                        } else if (which == 1) {
                            T* kr = Ks + ((size_t)hh * MH_TP + t) * MH_DH;
                            store4(kr + 4 * g4, ct[si][0][0], ct[si][0][1], ct[si][0][2], ct[si][0][3]);
                            if (g4 < 2) store4(kr + 16 + 4 * g4, ct[si][1][0], ct[si][1][1], ct[si][1][2], ct[si][1][3]);
                        } else if (sizeof(T) == 2) {
                            // bf16: V stays row-major like K; O^T = V^T P^T fetches its A operand through transposing LDS reads
                            T* vr = Vt + ((size_t)hh * MH_TP + t) * MH_DH;
                            store4(vr + 4 * g4, ct[si][0][0], ct[si][0][1], ct[si][0][2], ct[si][0][3]);
                            if (g4 < 2) store4(vr + 16 + 4 * g4, ct[si][1][0], ct[si][1][1], ct[si][1][2], ct[si][1][3]);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                store1(Vt + ((size_t)hh * MH_DH + 4 * g4 + r) * MH_TP + t, ct[si][0][r]);
                                if (g4 < 2) store1(Vt + ((size_t)hh * MH_DH + 16 + 4 * g4 + r) * MH_TP + t, ct[si][1][r]);
                            }
                        }
                    }
                }
            }
        }
        lds_barrier();

        // ---- stage B: attention per (strip, head); outputs stay in registers as B fragments ---
        // (Per pair the ISA is phased: 16 + 16 MFMAs, then 64 v_exp_f32 in a row.  Round 6 built the software-pipelined form — the O^T MFMAs of pair i-1
        //  issued between the exponentials of pair i, two per eight, pinned with an empty asm on the running sum + sched_barrier (without them the
        //  instruction selector emits all exponentials behind the last MFMA) — and measured it equal: 622 / 622 vs 616 / 628 us per launch, the step
        //  711-715 both ways.  With two waves per SIMD the other wave already fills the matrix pipe during this wave's softmax; s_setprio around the
        //  MFMA bursts changed nothing either: 620-631 vs 626-632 us.)
        Frag<T> of[NSW][HPP];
#pragma unroll
        for (int si = 0; si < NSW; ++si) {
#pragma unroll
            for (int hh = 0; hh < HPP; ++hh) {
                const T* kh = Ks + (size_t)hh * MH_TP * MH_DH;
                const T* vh = Vt + (size_t)hh * MH_DH * MH_TP;
                f32x4 sc[MH_NT];
                float mx = -1e30f;
#pragma unroll
                for (int j = 0; j < MH_NT; ++j) {
                    if (j < nst) {
                        Frag<T> a;
                        const T* kr = kh + (size_t)(j * 16 + l15) * MH_DH;
                        frag_load_lo(a, kr + 4 * g4);
                        if (g4 < 2) frag_load_hi(a, kr + 16 + 4 * g4);
                        else frag_zero_hi(a);
                        sc[j] = mma(a, qf[si][hh], F32X4_ZERO);
                        if (j == nst - 1) {  // only the last key tile can hold padded keys
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (j * 16 + 4 * g4 + r >= T_) sc[j][r] = -1e30f;
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[j][r]);
                    } else {
                        sc[j] = (f32x4){-1e30f, -1e30f, -1e30f, -1e30f};
                    }
                }
                mx = wave_max16(mx);
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < MH_NT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = (j < nst) ? fast_exp2(sc[j][r] - mx) : 0.f;
                        sc[j][r] = p;
                        sum += p;
                    }
                sum = wave_sum16(sum);
                const float inv = 1.0f / sum;
                if (lse && g4 == 0) {  // log2-sum-exp of the scaled score row: backward rebuilds P = exp2(S' - lse) from it
                    const int t = (w * NSW + si) * 16 + l15;
                    if (t < T_) lse[((size_t)bf * T_ + t) * MH_HEADS + pass * HPP + hh] = mx + log2f(sum);
                }
                f32x4 o0 = F32X4_ZERO, o1 = F32X4_ZERO;
#pragma unroll
                for (int ks = 0; ks < MH_NT / 2; ++ks) {
                    if (2 * ks < nst) {
                        Frag<T> pf, a0, a1;
                        frag_from_c2(pf, sc[2 * ks], sc[2 * ks + 1]);
                        if (sizeof(T) == 2) {
                            v_frag_tr(a0, vh, 0, ks);
                            v_frag_tr(a1, vh, 1, ks);  // rows d >= 24 pick up neighbouring (finite) data; zeroed below
                        } else {
                            const T* v0 = vh + (size_t)l15 * MH_TP + ks * 32 + 4 * g4;
                            frag_load_lo(a0, v0);
                            frag_load_hi(a0, v0 + 16);
                            if (l15 < MH_DH - 16) {
                                const T* v1 = vh + (size_t)(16 + l15) * MH_TP + ks * 32 + 4 * g4;
                                frag_load_lo(a1, v1);
                                frag_load_hi(a1, v1 + 16);
                            } else {
                                frag_zero(a1);
                            }
                        }
                        o0 = mma(a0, pf, o0);
                        o1 = mma(a1, pf, o1);
                    }
                }
                if (g4 >= 2) o1 = F32X4_ZERO;  // output rows d = 16 + 4 g4 + r >= dh are padding
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o0[r] *= inv;
                    o1[r] *= inv;
                }
                frag_from_c2(of[si][hh], o0, o1);
#ifdef MH_KO_SAVE  // (timing knock-out, A/B flavour: no saved attention output)
                if (false) {
#else
                if (osave) {  // attention output before out_proj: the only extra activation backward needs
#endif
                    const int t = (w * NSW + si) * 16 + l15;
                    if (t < T_) {
                        T* orow = osave + ((size_t)bf * T_ + t) * MH_H + (pass * HPP + hh) * MH_DH;
                        // (plain stores: the four heads' 48-byte pieces of a row merge in L2; non-temporal ones were 2x slower on some boxes)
                        store4(orow + 4 * g4, o0[0], o0[1], o0[2], o0[3]);
                        if (g4 < 2) store4(orow + 16 + 4 * g4, o1[0], o1[1], o1[2], o1[3]);
                    }
                }
            }
        }

        // ---- stage C: output projection (+bias, +residual) --------------------------------------
        if constexpr (NSW > 2) {  // four strips per wave: one strip at a time (the residual rows of all four would be 96 registers)
#pragma unroll 1
            for (int si = 0; si < NSW; ++si) {
                const int t = (w * NSW + si) * 16 + l15;
                float rv[MH_H / 16][4];
#pragma unroll
                for (int mt = 0; mt < MH_H / 16; ++mt) {
                    if (t < T_) load4((pass == 0 ? xb : yb) + (size_t)t * MH_H + 16 * mt + 4 * g4, rv[mt]);
                    else rv[mt][0] = rv[mt][1] = rv[mt][2] = rv[mt][3] = 0.f;
                }
                Frag<T> ofs[HPP];
#pragma unroll
                for (int hh = 0; hh < HPP; ++hh) {
                    ofs[hh] = of[0][hh];
#pragma unroll
                    for (int k = 1; k < NSW; ++k)
                        if (si == k) ofs[hh] = of[k][hh];
                }
#pragma unroll
                for (int mt = 0; mt < MH_H / 16; ++mt) {
                    const int ch = 16 * mt + 4 * g4;
                    f32x4 acc = F32X4_ZERO;
#pragma unroll
                    for (int hh = 0; hh < HPP; ++hh) {
                        Frag<T> a;
                        if (WLDS) frag_load(a, wl + ((size_t)24 + mt * MH_HEADS + hh) * 512 + lane * 8);
                        else wfrag_load(a, Wout, mt, MH_HEADS, pass * HPP + hh);
                        acc = mma(a, ofs[hh], acc);
                    }
                    float bo[4] = {0.f, 0.f, 0.f, 0.f};
                    if (pass == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) bo[r] = WLDS ? prm[3 * MH_H + ch + r] : bout[ch + r];
                    }
                    if (t < T_)
                        store4(yb + (size_t)t * MH_H + ch, rv[mt][0] + bo[0] + acc[0], rv[mt][1] + bo[1] + acc[1], rv[mt][2] + bo[2] + acc[2], rv[mt][3] + bo[3] + acc[3]);
                }
            }
        } else {
        // the residual rows are requested up front (one wait instead of one per output tile)
        float rv[NSW][MH_H / 16][4];
#pragma unroll
        for (int si = 0; si < NSW; ++si) {
            const int t = (w * NSW + si) * 16 + l15;
#pragma unroll
            for (int mt = 0; mt < MH_H / 16; ++mt) {
                if (t < T_) load4((pass == 0 ? xb : yb) + (size_t)t * MH_H + 16 * mt + 4 * g4, rv[si][mt]);
                else rv[si][mt][0] = rv[si][mt][1] = rv[si][mt][2] = rv[si][mt][3] = 0.f;
            }
        }
#pragma unroll
        for (int mt = 0; mt < MH_H / 16; ++mt) {
            Frag<T> a[HPP];
#pragma unroll
            for (int hh = 0; hh < HPP; ++hh) {
                if (WLDS) frag_load(a[hh], wl + ((size_t)24 + mt * MH_HEADS + hh) * 512 + lane * 8);
                else wfrag_load(a[hh], Wout, mt, MH_HEADS, pass * HPP + hh);
            }
            const int ch = 16 * mt + 4 * g4;
            float bo[4] = {0.f, 0.f, 0.f, 0.f};
            if (pass == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bo[r] = WLDS ? prm[3 * MH_H + ch + r] : bout[ch + r];
            }
#pragma unroll
            for (int si = 0; si < NSW; ++si) {
                const int t = (w * NSW + si) * 16 + l15;
                f32x4 acc = F32X4_ZERO;
#pragma unroll
                for (int hh = 0; hh < HPP; ++hh) acc = mma(a[hh], of[si][hh], acc);
                if (t < T_)
                    store4(yb + (size_t)t * MH_H + ch, rv[si][mt][0] + bo[0] + acc[0], rv[si][mt][1] + bo[1] + acc[1], rv[si][mt][2] + bo[2] + acc[2],
                           rv[si][mt][3] + bo[3] + acc[3]);
            }
        }
        }
    }
}

template <class T, int HPP, bool FULL, int NSW>
static int mhsa_fwd_t(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, void* osave, hipStream_t st, const SeqTail* tl) {
    if (c.T > MH_TP) return NBSS_EUNSUPPORTED;
    // +64: the last transposing read overreaches its row by 16 B; bf16: 48-fragment weight window + biases
    const size_t lds = (size_t)2 * HPP * MH_TP * MH_DH * sizeof(T) + 64 + (sizeof(T) == 2 && HPP == MH_HEADS ? (size_t)48 * 512 * sizeof(T) + 4 * MH_H * sizeof(float) : 0);
    const T* pk = (const T*)packed;
    int e = NBSS_SET_MAX_LDS((mhsa_fwd_kernel<T, HPP, FULL, NSW>), lds);
    if (e) return e;
    const int nseq = c.B * c.F, ntail = tl ? tl->n : 0;
    dim3 grid(nseq - ntail), block(64 * 16 / NSW);
    const int flip = walk_flip_next();  // (one direction for the main and the tail launch)
    ProfScope ps(PK_MHSA_F, st);
    float* lse = osave ? (float*)((char*)osave + mhsa_lse_offset(c)) : nullptr;
    NBSS_LAUNCH((mhsa_fwd_kernel<T, HPP, FULL, NSW>), grid, block, lds, st, c, P + param_off(c, layer, P_MH_LN_W), P + param_off(c, layer, P_MH_LN_B),
                P + param_off(c, layer, P_INP_B), P + param_off(c, layer, P_OUTP_B), pk + pack_off(c, layer, K_INP),
                pk + pack_off(c, layer, K_OUTP), (const T*)x, (T*)y, (T*)osave, lse, 0, flip);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    if (ntail > 0)  // the last, nearly empty round of sequences: on the walk's second stream, overlapping the next row kernel's full rounds
        NBSS_LAUNCH((mhsa_fwd_kernel<T, HPP, FULL, NSW>), dim3(ntail), block, lds, tl->ts, c, P + param_off(c, layer, P_MH_LN_W), P + param_off(c, layer, P_MH_LN_B),
                    P + param_off(c, layer, P_INP_B), P + param_off(c, layer, P_OUTP_B), pk + pack_off(c, layer, K_INP),
                    pk + pack_off(c, layer, K_OUTP), (const T*)x, (T*)y, (T*)osave, lse, nseq - ntail, flip);
    return NBSS_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------
// Sequences beyond 256 frames (forward only: validate / test / predict on full-length utterances).  K and V of a whole
// sequence no longer fit one workgroup's LDS, so the block is split in two launches:
//   mhsa_kv_kernel    : LN + K / V projections of every frame -> scratch [N][H] each (stream dtype, heads side by side)
//   mhsa_flash_kernel : one workgroup per 128 queries (8 waves x one 16-frame strip); K / V stream through LDS in blocks
//                       of 128 keys (the next block's global loads are in flight during the current block's math),
//                       online softmax per head, out_proj + bias + residual in the epilogue.
// Any T; no state is saved for backward (training keeps the single-workgroup kernel and its T <= 256 limit).
#define ML_QB 128  // queries per workgroup: 8 waves x one 16-frame strip
// G = geometry (geom.h): small dh = 24 (two 16-row tiles per head, the second half empty; one k-step for q k^T), large dh = 48 (three
// tiles, two k-steps).  in_proj fragments: K_INP tiles (which, head, i) with rows padded to multiples of 32 per head (layout.h).
template <class T, class G>
NBSS_DEV void ln_strip_g(const T* __restrict__ xr, bool valid, const float* __restrict__ lnw, const float* __restrict__ lnb, Frag<T> (&u)[G::KS]) {
    const int g4 = lane_id() >> 4;
    float v[G::KS][8];
    float sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
        if (valid) load8(xr + ks * 32 + 8 * g4, v[ks]);
        else
#pragma unroll
            for (int j = 0; j < 8; ++j) v[ks][j] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[ks][j];
    }
    const float mean = wave_sum16(sum) * (1.0f / G::H);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = v[ks][j] - mean;
            q += d * d;
        }
    const float rstd = rsqrtf(wave_sum16(q) * (1.0f / G::H) + 1e-5f);
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
        float gam[8], bet[8];
        load8(lnw + ks * 32 + 8 * g4, gam);
        load8(lnb + ks * 32 + 8 * g4, bet);
#pragma unroll
        for (int j = 0; j < 8; ++j) frag_set(u[ks], j, (v[ks][j] - mean) * rstd * gam[j] + bet[j]);
    }
}

template <class T, class G>
__global__ __launch_bounds__(512) void mhsa_kv_kernel(nbss_cfg c, const float* __restrict__ lnw, const float* __restrict__ lnb, const float* __restrict__ bin,
                                                      const T* __restrict__ Win, const T* __restrict__ x, T* __restrict__ Kg, T* __restrict__ Vg) {
    constexpr int H = G::H, DH = G::DH, HEADS = G::HEADS, KS = G::KS;
    constexpr int TPH = (DH + 31) / 32 * 2, OT = (DH + 15) / 16;  // packed tiles per head / tiles that hold real rows
    const int T_ = c.T, bf = blockIdx.x;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const int t = (int)blockIdx.y * ML_QB + w * 16 + l15;
    const bool tv = t < T_;
    const size_t n = (size_t)bf * T_ + (tv ? t : 0);
    Frag<T> u[KS];
    ln_strip_g<T, G>(x + n * H, tv, lnw, lnb, u);
#pragma unroll
    for (int which = 1; which < 3; ++which) {
        T* dst = (which == 1 ? Kg : Vg) + n * H;
#pragma unroll
        for (int head = 0; head < HEADS; ++head) {
#pragma unroll
            for (int i = 0; i < OT; ++i) {
                f32x4 acc = F32X4_ZERO;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    Frag<T> a;
                    wfrag_load(a, Win, (which * HEADS + head) * TPH + i, KS, ks);
                    acc = mma(a, u[ks], acc);
                }
                const int d = 16 * i + 4 * g4;
                if (tv && d < DH) {
                    const float* bs = bin + which * H + head * DH + d;
                    store4(dst + head * DH + d, acc[0] + bs[0], acc[1] + bs[1], acc[2] + bs[2], acc[3] + bs[3]);
                }
            }
        }
    }
}

// A operand of O^T = V^T P^T from a row-major bf16 V image [keys][DH]: rows d = tile*16 + l15, K = the 32 keys of k-step ks
template <int DH>
NBSS_DEV void v_frag_tr_g(Frag<bf16_t>& f, const bf16_t* __restrict__ vr, int tile, int ks) {
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    frag_load_tr(f, vr + (size_t)(ks * 32 + 4 * g4 + (l15 >> 2)) * DH + tile * 16 + 4 * (l15 & 3), DH);
}
template <int DH>
NBSS_DEV void v_frag_tr_g(Frag<float>&, const float*, int, int) {}

// KB = keys per LDS block (128; 64 for the fp32 stream at dh = 48 so that K and V^T fit)
template <class T, class G, int KB>
__global__ __launch_bounds__(512) void mhsa_flash_kernel(nbss_cfg c, const float* __restrict__ lnw, const float* __restrict__ lnb, const float* __restrict__ bin,
                                                         const float* __restrict__ bout, const T* __restrict__ Win, const T* __restrict__ Wout,
                                                         const T* __restrict__ x, const T* __restrict__ Kg, const T* __restrict__ Vg, T* __restrict__ y) {
    constexpr int H = G::H, DH = G::DH, HEADS = G::HEADS, KS = G::KS;
    constexpr int TPH = (DH + 31) / 32 * 2, OT = (DH + 15) / 16, KSD = (DH + 31) / 32;
    NBSS_LDS(smem);
    T* Ks = reinterpret_cast<T*>(smem);        // [heads][KB][DH]
    T* Vs = Ks + HEADS * KB * DH;              // bf16: [heads][KB][DH] (transposing reads); fp32: [heads][DH][KB]
    const int T_ = c.T, bf = blockIdx.x;
    const int tid = threadIdx.x, lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const int t = (int)blockIdx.y * ML_QB + w * 16 + l15;
    const bool tv = t < T_;
    const size_t n = (size_t)bf * T_ + (tv ? t : 0);
    const T* Kb = Kg + (size_t)bf * T_ * H;
    const T* Vb = Vg + (size_t)bf * T_ * H;
    const float qscale = 1.4426950408889634f * rsqrtf((float)DH);

    // ---- Q of this wave's strip, all heads, in registers (B fragments in the permuted K order of stacked C tiles) ----
    Frag<T> qf[HEADS][KSD];
    {
        Frag<T> u[KS];
        ln_strip_g<T, G>(x + n * H, tv, lnw, lnb, u);
#pragma unroll
        for (int head = 0; head < HEADS; ++head) {
            f32x4 ct[2 * KSD];
#pragma unroll
            for (int i = 0; i < 2 * KSD; ++i) {
                f32x4 acc = F32X4_ZERO;
                if (i < OT) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        Frag<T> a;
                        wfrag_load(a, Win, head * TPH + i, KS, ks);
                        acc = mma(a, u[ks], acc);
                    }
                }
                const int d = 16 * i + 4 * g4;
#pragma unroll
                for (int r = 0; r < 4; ++r) ct[i][r] = d < DH ? (acc[r] + bin[head * DH + (d < DH ? d : 0) + r]) * qscale : 0.f;
            }
#pragma unroll
            for (int p = 0; p < KSD; ++p) frag_from_c2(qf[head][p], ct[2 * p], ct[2 * p + 1]);
        }
    }

    // ---- key blocks ---------------------------------------------------------------------------------
    constexpr int VN = 16 / sizeof(T), RV = H / VN, NVB = KB * RV / 512;  // 16-byte vectors: per row, per thread
    static_assert(KB * RV % 512 == 0, "whole vectors per thread");
    u32x4 kreg[NVB], vreg[NVB];
    auto gload = [&](int kb) {
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int v = tid + i * 512, row = v / RV, e = (v % RV) * VN;
            const int tk = kb * KB + row, tc = tk < T_ ? tk : T_ - 1;  // clamped address, zero rows past the sequence
            const u32x4 kv = *reinterpret_cast<const u32x4*>(Kb + (size_t)tc * H + e);
            const u32x4 vv = *reinterpret_cast<const u32x4*>(Vb + (size_t)tc * H + e);
            const uint32_t keep = tk < T_ ? 0xffffffffu : 0u;
            kreg[i] = (u32x4){kv[0] & keep, kv[1] & keep, kv[2] & keep, kv[3] & keep};
            vreg[i] = (u32x4){vv[0] & keep, vv[1] & keep, vv[2] & keep, vv[3] & keep};
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int v = tid + i * 512, row = v / RV, e = (v % RV) * VN, hh = e / DH, d = e % DH;
            *reinterpret_cast<u32x4*>(Ks + ((size_t)hh * KB + row) * DH + d) = kreg[i];
            if (sizeof(T) == 2) {
                *reinterpret_cast<u32x4*>(Vs + ((size_t)hh * KB + row) * DH + d) = vreg[i];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) reinterpret_cast<uint32_t*>(Vs)[((size_t)hh * DH + d + q) * KB + row] = vreg[i][q];
            }
        }
    };

    float mrun[HEADS], lrun[HEADS];
    f32x4 o[HEADS][OT];
#pragma unroll
    for (int hh = 0; hh < HEADS; ++hh) {
        mrun[hh] = -1e30f;
        lrun[hh] = 0.f;
#pragma unroll
        for (int i = 0; i < OT; ++i) o[hh][i] = F32X4_ZERO;
    }
    const int nkb = cdiv(T_, KB);
    gload(0);
    for (int kb = 0; kb < nkb; ++kb) {
        lds_barrier();  // the previous block's readers are done
        sstore();
        lds_barrier();
        if (kb + 1 < nkb) gload(kb + 1);
        const bool lastb = kb == nkb - 1;
#pragma unroll
        for (int hh = 0; hh < HEADS; ++hh) {
            const T* kh = Ks + (size_t)hh * KB * DH;
            f32x4 sc[KB / 16];
            float mx = -1e30f;
#pragma unroll
            for (int j = 0; j < KB / 16; ++j) {
                const T* kr = kh + (size_t)(j * 16 + l15) * DH;
                f32x4 sacc = F32X4_ZERO;
#pragma unroll
                for (int p = 0; p < KSD; ++p) {
                    Frag<T> a;
                    if (32 * p + 4 * g4 < DH) frag_load_lo(a, kr + 32 * p + 4 * g4);
                    else frag_zero_lo(a);
                    if (32 * p + 16 + 4 * g4 < DH) frag_load_hi(a, kr + 32 * p + 16 + 4 * g4);
                    else frag_zero_hi(a);
                    sacc = mma(a, qf[hh][p], sacc);
                }
                sc[j] = sacc;
                if (lastb) {  // only the last block can hold keys past the sequence
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kb * KB + j * 16 + 4 * g4 + r >= T_) sc[j][r] = -1e30f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[j][r]);
            }
            mx = fmaxf(wave_max16(mx), mrun[hh]);
            const float alpha = fast_exp2(mrun[hh] - mx);
            mrun[hh] = mx;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < KB / 16; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = fast_exp2(sc[j][r] - mx);
                    sc[j][r] = pv;
                    sum += pv;
                }
            lrun[hh] = lrun[hh] * alpha + wave_sum16(sum);
#pragma unroll
            for (int i = 0; i < OT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[hh][i][r] *= alpha;
#pragma unroll
            for (int ks = 0; ks < KB / 32; ++ks) {
                Frag<T> pf;
                frag_from_c2(pf, sc[2 * ks], sc[2 * ks + 1]);
#pragma unroll
                for (int i = 0; i < OT; ++i) {
                    Frag<T> a;
                    if (sizeof(T) == 2) {
                        // (dh = 24: rows d >= 24 of the second tile pick up neighbouring data; those output rows are dropped below)
                        v_frag_tr_g<DH>(a, Vs + (size_t)hh * KB * DH, i, ks);
                    } else {
                        const T* vh = Vs + (size_t)hh * DH * KB;
                        if (16 * i + l15 < DH) {
                            const T* v0 = vh + (size_t)(16 * i + l15) * KB + ks * 32 + 4 * g4;
                            frag_load_lo(a, v0);
                            frag_load_hi(a, v0 + 16);
                        } else {
                            frag_zero(a);
                        }
                    }
                    o[hh][i] = mma(a, pf, o[hh][i]);
                }
            }
        }
    }

    // ---- normalise, out_proj + bias + residual -------------------------------------------------------
    Frag<T> of[HEADS][KSD];
#pragma unroll
    for (int hh = 0; hh < HEADS; ++hh) {
        const float inv = 1.0f / lrun[hh];
        f32x4 ct[2 * KSD];
#pragma unroll
        for (int i = 0; i < 2 * KSD; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = 0.f;
                if (i < OT) v = o[hh][i < OT ? i : 0][r] * inv;
                ct[i][r] = 16 * i + 4 * g4 < DH ? v : 0.f;  // rows d >= dh are padding
            }
#pragma unroll
        for (int p = 0; p < KSD; ++p) frag_from_c2(of[hh][p], ct[2 * p], ct[2 * p + 1]);
    }
    const T* xr = x + n * H;
    T* yr = y + n * H;
#pragma unroll
    for (int mt = 0; mt < H / 16; ++mt) {
        const int ch = 16 * mt + 4 * g4;
        f32x4 acc = F32X4_ZERO;
#pragma unroll
        for (int hh = 0; hh < HEADS; ++hh)
#pragma unroll
            for (int p = 0; p < KSD; ++p) {
                Frag<T> a;
                wfrag_load(a, Wout, mt, HEADS * KSD, hh * KSD + p);
                acc = mma(a, of[hh][p], acc);
            }
        if (tv) {
            float rv[4];
            load4(xr + ch, rv);
            store4(yr + ch, rv[0] + bout[ch] + acc[0], rv[1] + bout[ch + 1] + acc[1], rv[2] + bout[ch + 2] + acc[2], rv[3] + bout[ch + 3] + acc[3]);
        }
    }
}

template <class T, class G, int KB>
static int mhsa_fwd_long_t(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, void* scratch, hipStream_t st) {
    if (!scratch) return NBSS_EINVAL;  // K | V, nbss_mhsa_save_bytes() bytes
    const size_t N = (size_t)c.B * c.F * c.T;
    T* Kg = (T*)scratch;
    T* Vg = (T*)((char*)scratch + ws_align(N * G::H * sizeof(T)));
    const size_t lds = (size_t)2 * G::HEADS * KB * G::DH * sizeof(T) + 64;
    if (lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    const T* pk = (const T*)packed;
    int e = NBSS_SET_MAX_LDS((mhsa_flash_kernel<T, G, KB>), lds);
    if (e) return e;
    dim3 grid(c.B * c.F, cdiv(c.T, ML_QB)), block(512);
    ProfScope ps(PK_MHSA_F, st);
    const float* lnw = P + param_off(c, layer, P_MH_LN_W);
    const float* lnb = P + param_off(c, layer, P_MH_LN_B);
    const float* bin = P + param_off(c, layer, P_INP_B);
    NBSS_LAUNCH((mhsa_kv_kernel<T, G>), grid, block, 0, st, c, lnw, lnb, bin, pk + pack_off(c, layer, K_INP), (const T*)x, Kg, Vg);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    NBSS_LAUNCH((mhsa_flash_kernel<T, G, KB>), grid, block, lds, st, c, lnw, lnb, bin, P + param_off(c, layer, P_OUTP_B), pk + pack_off(c, layer, K_INP),
                pk + pack_off(c, layer, K_OUTP), (const T*)x, (const T*)Kg, (const T*)Vg, (T*)y);
    return NBSS_CHECK_LAUNCH();
}

int mhsa_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, void* osave, hipStream_t st, const SeqTail* tl) {
    // SpatialNet-large (forward only): always the two-launch path; `osave` is its K | V scratch
    if (c.H == GeoL::H) return c.dtype == NBSS_BF16 ? mhsa_fwd_long_t<bf16_t, GeoL, 128>(c, P, packed, layer, x, y, osave, st) : mhsa_fwd_long_t<float, GeoL, 64>(c, P, packed, layer, x, y, osave, st);
    // T > 256: `osave` is the K | V scratch of the two-launch long-sequence path (nothing is saved for backward)
    if (c.T > MH_TP) return c.dtype == NBSS_BF16 ? mhsa_fwd_long_t<bf16_t, GeoS, 128>(c, P, packed, layer, x, y, osave, st) : mhsa_fwd_long_t<float, GeoS, 128>(c, P, packed, layer, x, y, osave, st);
    if (c.dtype != NBSS_BF16) return mhsa_fwd_t<float, 2, false, 2>(c, P, packed, layer, x, y, osave, st, nullptr);
    return cdiv(c.T, 16) == MH_NT ? mhsa_fwd_t<bf16_t, MH_BF16_HPP, true, MH_BF16_NSW>(c, P, packed, layer, x, y, osave, st, tl) : mhsa_fwd_t<bf16_t, MH_BF16_HPP, false, MH_BF16_NSW>(c, P, packed, layer, x, y, osave, st, tl);
}
