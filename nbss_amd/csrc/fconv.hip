// fconv.hip — cross-band frequency-convolutional module of SpatialNetLayer:
//   y = x + PReLU(Conv1d_F(LayerNorm_H(x)))       (SpatialNet.py:85,87,116-127; LN base/norm.py:11-27)
// Conv1d(H,H,k=5,groups=8,'same', zeros) runs ALONG F for every (b,t).
//
// Work decomposition: one workgroup = one (b, TT consecutive frames) slab with the whole F axis
// resident in LDS, so there is no halo re-read and no [B,T,H,F] permute copy (the reference
// makes two).  x is fetched as F chunks of TT*H contiguous elements, LayerNorm'ed in LDS, and the
// grouped conv is 8 independent GEMMs  out^T[12(16) x F] = W_g[12 x 60(64)] * im2col(u_g)[60 x F]
// on the matrix cores (form 2: weights = A, frequencies = N).  Results return through LDS so the
// residual add + store are full 16-byte coalesced accesses.  bf16: 8 waves, one conv group per wave, two
// workgroups per CU; the backward kernel (below) alternates row phases and group phases on the same slab and (bf16, two-frame slabs)
// contracts the conv weight gradient between them, while both of its operands sit in the LDS images.
#include "launch.h"
#include "layout.h"
#include "prof.h"
#include "blocks.h"
#include "wgrad.h"
#include "fold.h"
#include "foldk.h"
#include "geom.h"
#include "side.h"
#include <cstdlib>

#define FC_H 96
#define FC_G 8
#define FC_CG 12   // channels per group
#define FC_KS 2    // 15 pieces of 4 channels -> 2 k-steps of 8 pieces
#define FC_MTF_MAX 10
#define FC_MTF_BIG 17  // F <= 272
#define FC_P16 (5 * FC_H * FC_CG)  // bf16 values per partial row of the fused conv weight gradient: [tap][group][input channel][12 outputs]

template <class T> struct VecOf;
template <> struct VecOf<bf16_t> { static constexpr int N = 8; };
template <> struct VecOf<float> { static constexpr int N = 4; };

NBSS_DEV void vec_copy(bf16_t* d, const bf16_t* s) { *reinterpret_cast<u32x4*>(d) = *reinterpret_cast<const u32x4*>(s); }
NBSS_DEV void vec_copy(float* d, const float* s) { *reinterpret_cast<f32x4*>(d) = *reinterpret_cast<const f32x4*>(s); }
NBSS_DEV void vec_zero(bf16_t* d) { *reinterpret_cast<u32x4*>(d) = (u32x4){0, 0, 0, 0}; }
NBSS_DEV void vec_zero(float* d) { *reinterpret_cast<f32x4*>(d) = (f32x4){0, 0, 0, 0}; }

// LayerNorm over H=96 of one LDS-resident row, in place (fp32 statistics, eps 1e-5).
template <class T, int HL>
NBSS_DEV void ln_row_inplace(T* row, const float* __restrict__ gamma, const float* __restrict__ beta) {
    float s = 0.f;
    float v[HL];
#pragma unroll
    for (int i = 0; i < HL; i += 8) load8(row + i, v + i);
#pragma unroll
    for (int i = 0; i < HL; ++i) s += v[i];
    const float mean = s * (1.0f / HL);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < HL; ++i) {
        const float d = v[i] - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(q * (1.0f / HL) + 1e-5f);
#pragma unroll
    for (int i = 0; i < HL; i += 4)
        store4(row + i, (v[i] - mean) * rstd * gamma[i] + beta[i], (v[i + 1] - mean) * rstd * gamma[i + 1] + beta[i + 1],
               (v[i + 2] - mean) * rstd * gamma[i + 2] + beta[i + 2], (v[i + 3] - mean) * rstd * gamma[i + 3] + beta[i + 3]);
}

// The same LayerNorm with TWO adjacent lanes per row (each HL / 2 channels; the partial sums are exchanged with a DPP quad swap): all 512
// threads of a two-frame slab are busy (258 rows) and a lane's serial chain is half as long.  `row` points at the lane's own half.
template <class T, int HL>
NBSS_DEV void ln_halfrow_inplace(T* row, const float* __restrict__ gamma, const float* __restrict__ beta, bool active) {
    constexpr int HP = HL / 2;
    float v[HP], s = 0.f;
#pragma unroll
    for (int i = 0; i < HP; i += 8) load8(row + i, v + i);
#pragma unroll
    for (int i = 0; i < HP; ++i) s += v[i];
#ifdef NBSS_EMU
    s += __shfl_xor(s, 1);
#else
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
#endif
    const float mean = s * (1.0f / HL);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < HP; ++i) {
        const float d = v[i] - mean;
        q += d * d;
    }
#ifdef NBSS_EMU
    q += __shfl_xor(q, 1);
#else
    q += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q), 0xB1, 0xF, 0xF, true));
#endif
    const float rstd = rsqrtf(q * (1.0f / HL) + 1e-5f);
    if (!active) return;  // (a lane beyond the last row read a clamped row so that whole waves take part in the exchange: no store)
#pragma unroll
    for (int i = 0; i < HP; i += 4)
        store4(row + i, (v[i] - mean) * rstd * gamma[i] + beta[i], (v[i + 1] - mean) * rstd * gamma[i + 1] + beta[i + 1],
               (v[i + 2] - mean) * rstd * gamma[i + 2] + beta[i + 2], (v[i + 3] - mean) * rstd * gamma[i + 3] + beta[i + 3]);
}

// GPW = conv groups per wave: 2 with 4 waves (fp32), 1 with 8 waves (bf16: two 8-wave workgroups per CU)
// MTF = frequency tiles the accumulators are sized for: 10 (F <= 160, the 8-kHz geometry) or 17 (F <= 272: 16 kHz, n_fft 512 -> 257 bins)
// HH = dim_hidden (geom.h): 96 (12 channels per conv group, one 16-row output tile) or 192 (24 channels, two tiles)
template <class T, int TT, int GPW, int MTF, int HH>
// (launch bounds: threads, waves per SIMD — four = two 8-wave workgroups per CU, at most 128 VGPRs; -DNBSS_FCONV_DMA: A/B flavour, rounds 4-5's prologue)
__global__ __launch_bounds__(64 * FC_G / GPW, GPW == 1 ? (MTF <= 10 && HH == 96 ? 4 : 2) : 1) void fconv_fwd_kernel(
                                                        nbss_cfg c, const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                        const float* __restrict__ cb, const float* __restrict__ slope,
                                                        const T* __restrict__ Wp, const T* __restrict__ x, T* __restrict__ y, int flip) {
    constexpr int FG = HH / FC_G;                 // channels per group
    constexpr int MTG = (FG + 15) / 16;           // output tiles per group
    constexpr int NP = 5 * (FG / 4);              // im2col pieces of 4 channels: (tap, channel quad)
    constexpr int KSG = (NP + 7) / 8;             // k-steps
    NBSS_LDS(smem);
    T* u = reinterpret_cast<T*>(smem);
    const int F = c.F, T_ = c.T;
    const int ntt = cdiv(T_, TT);
    const int bid = flip_bid(flip);  // (launch.h: consecutive kernels of a walk traverse the utterances in opposite order)
    const int b = bid / ntt, t0 = (bid % ntt) * TT;
    const int mtf = cdiv(F, 16), FP = mtf * 16 + 4;
    constexpr int HHP = HH + 8;              // padded (frequency, frame) row: 208 / 400 bytes — consecutive rows start on different LDS banks
    constexpr int ROW = TT * HHP;            // elements per frequency row in LDS
    constexpr int VN = VecOf<T>::N;
    constexpr int VPR = TT * HH / VN;        // payload vectors per frequency row
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();

    // ---- phase 1: stage x (raw) into LDS rows f+2, zero halo / tail rows -------------------
#ifdef NBSS_FCONV_DMA
    constexpr bool DMA = sizeof(T) == 2, REGX = false;
#else
    constexpr bool DMA = false, REGX = sizeof(T) == 2;
#endif
    // bf16 stream (round 6): the slab's rows come in through REGISTERS — every thread requests its 16-byte pieces up front (one round trip), copies
    // them into the image, and KEEPS them: they are the residual of phase 4.  With the LDS-DMA prologue (no register stop) the image was normalised in
    // place and the residual re-read from global memory — the kernel fetched 2.26 x the slab (PMC, round 5), on a kernel that runs at the HBM rate.
    // (six rounds of the workgroup's threads = 3 072 of the 3 096 pieces of a 129-bin slab; the few pieces beyond them take the old path: copied in a
    //  load -> store loop and re-read in phase 4 — registers for the largest slab the instance admits would be 8 - 13 rounds, over the 128-VGPR budget)
    constexpr int NTHR = 64 * FC_G / GPW, NXR = REGX ? 6 : 1;
    u32x4 xr[NXR];
    if constexpr (REGX) {
#pragma unroll
        for (int k = 0; k < NXR; ++k) {
            const int i = tid + k * NTHR, ic = i < F * VPR ? i : 0, f = ic / VPR, off = (ic % VPR) * VN, tt = off / HH;
            if (i < F * VPR && t0 + tt < T_) xr[k] = *reinterpret_cast<const u32x4*>(x + (((size_t)b * F + f) * T_ + t0 + tt) * HH + (off - tt * HH));
            else xr[k] = (u32x4){0u, 0u, 0u, 0u};  // (frames behind the end of the sequence: zero rows)
        }
        for (int i = tid; i < (FP - F) * VPR; i += nthr) {  // halo rows (f = -2, -1) and the rows behind the last frequency
            const int k = i / VPR, rr = k < 2 ? k : F + k, off = (i % VPR) * VN, tt = off / HH;
            vec_zero(u + (size_t)rr * ROW + tt * HHP + (off - tt * HH));
        }
#pragma unroll
        for (int k = 0; k < NXR; ++k) {
            const int i = tid + k * NTHR, f = i / VPR, off = (i % VPR) * VN, tt = off / HH;
            if (i < F * VPR) *reinterpret_cast<u32x4*>(u + (size_t)(f + 2) * ROW + tt * HHP + (off - tt * HH)) = xr[k];
        }
        for (int i = tid + NXR * NTHR; i < F * VPR; i += NTHR) {
            const int f = i / VPR, off = (i % VPR) * VN, tt = off / HH;
            T* d = u + (size_t)(f + 2) * ROW + tt * HHP + (off - tt * HH);
            if (t0 + tt < T_) vec_copy(d, x + (((size_t)b * F + f) * T_ + t0) * HH + off);
            else vec_zero(d);
        }
    } else if constexpr (DMA) {
        // bf16 stream: the slab comes in as one burst of global -> LDS copies (16-byte pieces, no register stop; the copy loop below is a chain
        // of load -> store round trips, seven per thread).  Piece q: row q / CPR (= f TT + tt), 16-byte column q % CPR (the last one is padding).
        constexpr int CPR = HHP / 8, DPR = HH / 8;
        const int wu = wave_id_u(), Q = F * TT * CPR, nw = nthr / 64;
        for (int i = wu; i * 64 < Q; i += nw) {
            const int q = i * 64 + lane, r = q / CPR, cc = q - r * CPR, f = r / TT, tt = r - f * TT;
            if (q < Q && cc < DPR && t0 + tt < T_)
                dma16_to_lds(reinterpret_cast<char*>(u + 2 * ROW) + (size_t)i * 1024, x + (((size_t)b * F + f) * T_ + t0 + tt) * HH + cc * 8);
        }
        for (int i = tid; i < (FP - F) * VPR; i += nthr) {  // halo rows (f = -2, -1) and the rows behind the last frequency
            const int k = i / VPR, rr = k < 2 ? k : F + k, off = (i % VPR) * VN, tt = off / HH;
            vec_zero(u + (size_t)rr * ROW + tt * HHP + (off - tt * HH));
        }
        if (t0 + TT > T_)  // a slab that ends the sequence: its missing frames are zero rows
            for (int i = tid; i < F * VPR; i += nthr) {
                const int f = i / VPR, off = (i % VPR) * VN, tt = off / HH;
                if (t0 + tt >= T_) vec_zero(u + (size_t)(f + 2) * ROW + tt * HHP + (off - tt * HH));
            }
        dma_wait_all();
    } else {
    for (int i = tid; i < FP * VPR; i += nthr) {
        const int rr = i / VPR, off = (i % VPR) * VN, f = rr - 2, tt = off / HH;
        T* d = u + (size_t)rr * ROW + tt * HHP + (off - tt * HH);
        if (f >= 0 && f < F && t0 + tt < T_)
            vec_copy(d, x + (((size_t)b * F + f) * T_ + t0) * HH + off);
        else
            vec_zero(d);
    }
    }
    lds_barrier();
    for (int base = 0; base < 2 * F * TT; base += nthr) {  // two adjacent lanes per (frequency, frame) row; whole waves take part
        const int r2 = base + tid, rc = (r2 < 2 * F * TT ? r2 : 2 * F * TT - 2 + (r2 & 1)) >> 1, hf = r2 & 1, f = rc / TT, tt = rc % TT;
        // (rows beyond T hold zeros: normalising them is harmless and keeps both lanes of a pair on the same path)
        ln_halfrow_inplace<T, HH>(u + (size_t)(f + 2) * ROW + tt * HHP + hf * (HH / 2), lnw + hf * (HH / 2), lnb + hf * (HH / 2), r2 < 2 * F * TT);
    }
    lds_barrier();

    // ---- phase 2: grouped conv on the matrix cores -----------------------------------------
    f32x4 acc[GPW][MTG][TT][MTF];
    Frag<T> a[GPW][MTG][KSG];
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi)
#pragma unroll
        for (int mt = 0; mt < MTG; ++mt)
#pragma unroll
            for (int ks = 0; ks < KSG; ++ks) wfrag_load(a[gi][mt][ks], Wp, (GPW * w + gi) * MTG + mt, KSG, ks);
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi)
#pragma unroll
        for (int mt = 0; mt < MTG; ++mt)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int ft = 0; ft < MTF; ++ft) acc[gi][mt][tt][ft] = F32X4_ZERO;
#pragma unroll
    for (int ft = 0; ft < MTF; ++ft) {
        if (ft < mtf) {
            const int f = ft * 16 + l15;
#pragma unroll
            for (int gi = 0; gi < GPW; ++gi) {
                const int ch0 = (GPW * w + gi) * FG;
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
                    for (int ks = 0; ks < KSG; ++ks) {
                        Frag<T> bq;
                        const int p0 = ks * 8 + 2 * g4, p1 = p0 + 1;
                        // piece p -> tap p / (FG/4), channels (p % (FG/4))*4..+3 ; LDS row = f + tap
                        if (p0 < NP) frag_load_lo(bq, u + (size_t)(f + p0 / (FG / 4)) * ROW + tt * HHP + ch0 + (p0 % (FG / 4)) * 4);
                        else frag_zero_lo(bq);
                        if (p1 < NP) frag_load_hi(bq, u + (size_t)(f + p1 / (FG / 4)) * ROW + tt * HHP + ch0 + (p1 % (FG / 4)) * 4);
                        else frag_zero_hi(bq);
#pragma unroll
                        for (int mt = 0; mt < MTG; ++mt) acc[gi][mt][tt][ft] = mma(a[gi][mt][ks], bq, acc[gi][mt][tt][ft]);
                    }
                }
            }
        }
    }
    lds_barrier();  // everyone is done reading u

    // ---- phase 3: bias + PReLU -> LDS [f][tt][H] ---------------------------------------------
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi)
#pragma unroll
        for (int mt = 0; mt < MTG; ++mt) {
            if (16 * mt + 4 * g4 < FG) {
                const int ch = (GPW * w + gi) * FG + 16 * mt + 4 * g4;
                float bb[4], sl[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { bb[r] = cb[ch + r]; sl[r] = slope[ch + r]; }
#pragma unroll
                for (int ft = 0; ft < MTF; ++ft) {
                    const int f = ft * 16 + l15;
                    if (ft < mtf && f < F) {
#pragma unroll
                        for (int tt = 0; tt < TT; ++tt) {
                            float o[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float v = acc[gi][mt][tt][ft][r] + bb[r];
                                o[r] = v > 0.f ? v : sl[r] * v;
                            }
                            store4(u + (size_t)f * ROW + tt * HHP + ch, o[0], o[1], o[2], o[3]);
                        }
                    }
                }
            }
        }
    lds_barrier();

    // ---- phase 4: residual add + coalesced store -------------------------------------------------
    if constexpr (REGX) {
#pragma unroll
        for (int k = 0; k < NXR; ++k) {
            const int i = tid + k * NTHR, f = i / VPR, off = (i % VPR) * VN, tt = off / HH;
            if (i >= F * VPR || t0 + tt >= T_) continue;
            const size_t go = (((size_t)b * F + f) * T_ + t0) * HH + off;
            float yv[8], o[8];
            load8(u + (size_t)f * ROW + tt * HHP + (off - tt * HH), yv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[2 * j] = bf2f((bf16_t)(xr[k][j] & 0xFFFF)) + yv[2 * j];
                o[2 * j + 1] = bf2f((bf16_t)(xr[k][j] >> 16)) + yv[2 * j + 1];
            }
            store8(reinterpret_cast<bf16_t*>(y) + go, o);
        }
        for (int i = tid + NXR * NTHR; i < F * VPR; i += NTHR) {
            const int f = i / VPR, off = (i % VPR) * VN, tt = off / HH;
            if (t0 + tt >= T_) continue;
            const size_t go = (((size_t)b * F + f) * T_ + t0) * HH + off;
            float xv[8], yv[8], o[8];
            load8(x + go, xv);
            load8(u + (size_t)f * ROW + tt * HHP + (off - tt * HH), yv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = xv[j] + yv[j];
            store8(reinterpret_cast<bf16_t*>(y) + go, o);
        }
    } else if constexpr (DMA) {  // (VN == 8) the residual rows of up to eight iterations are requested together: one round trip per batch instead of one per row
        constexpr int NB4 = 8;
        for (int i0 = tid; i0 < F * VPR; i0 += NB4 * nthr) {
            u32x4 xr4[NB4];
#pragma unroll
            for (int k = 0; k < NB4; ++k) {
                const int i = i0 + k * nthr, ic = i < F * VPR ? i : F * VPR - 1, f = ic / VPR, off = (ic % VPR) * VN, tt = off / HH;
                const int ttc = t0 + tt < T_ ? tt : 0;  // (clamped address: the value is not used)
                xr4[k] = *reinterpret_cast<const u32x4*>(x + (((size_t)b * F + f) * T_ + t0 + ttc) * HH + (off - tt * HH));
            }
#pragma unroll
            for (int k = 0; k < NB4; ++k) {
                const int i = i0 + k * nthr;
                if (i >= F * VPR) continue;
                const int f = i / VPR, off = (i % VPR) * VN, tt = off / HH;
                if (t0 + tt >= T_) continue;
                const size_t go = (((size_t)b * F + f) * T_ + t0) * HH + off;
                float yv[8], o[8];
                load8(u + (size_t)f * ROW + tt * HHP + (off - tt * HH), yv);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[2 * j] = bf2f((bf16_t)(xr4[k][j] & 0xFFFF)) + yv[2 * j];
                    o[2 * j + 1] = bf2f((bf16_t)(xr4[k][j] >> 16)) + yv[2 * j + 1];
                }
                store8(reinterpret_cast<bf16_t*>(y) + go, o);
            }
        }
    } else
    for (int i = tid; i < F * VPR; i += nthr) {
        const int f = i / VPR, off = (i % VPR) * VN, tt = off / HH;
        if (t0 + tt >= T_) continue;
        const size_t go = (((size_t)b * F + f) * T_ + t0) * HH + off;
        float xv[8], yv[8];
        if (VN == 8) {
            load8(x + go, xv);
            load8(u + (size_t)f * ROW + tt * HHP + (off - tt * HH), yv);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = xv[j] + yv[j];
            store4(y + go, o[0], o[1], o[2], o[3]);
            store4(y + go + 4, o[4], o[5], o[6], o[7]);
        } else {
            load4(x + go, xv);
            load4(u + (size_t)f * ROW + tt * HHP + (off - tt * HH), yv);
            store4(y + go, xv[0] + yv[0], xv[1] + yv[1], xv[2] + yv[2], xv[3] + yv[3]);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Backward (data gradient): one workgroup (8 waves) = one (b, TT frames) slab with the whole F axis in LDS.
// The work alternates between two decompositions so that every global access is a 16-byte piece of a full row:
//   row phases   (0, 3): a wave owns (frequency tile, frame) units in the B-fragment layout (lane = frequency row, 8 channels
//                        per k-step): LayerNorm forward / backward, residual, x and dy stay in registers in between
//   group phases (1, 2): wave g owns conv group g for ALL tiles, its 4 weight fragments (conv and transposed conv) live in
//                        registers, operands come from the LDS images with the +-2 row halo
// dv (for the conv weight gradient, wgrad.hip) is copied out of LDS as full rows.  The first version (one frame per 4-wave
// workgroup, weights through LDS, per-group 8-byte global accesses, a per-thread serial LayerNorm) ran at 0.38 TB/s.
#define FC_LD 104  // LDS row length: 208-byte rows put 16 consecutive rows on distinct 16-byte bank slots

template <class T>
NBSS_DEV void fconv_bfrag(Frag<T>& bq, const T* __restrict__ u, int rstride, int f, int ch0, int ks) {
    const int g4 = lane_id() >> 4;
    const int p0 = ks * 8 + 2 * g4, p1 = p0 + 1;  // piece p -> tap p/3, channels (p%3)*4..+3 ; image row = f + tap
    frag_load_lo(bq, u + (size_t)(f + p0 / 3) * rstride + ch0 + (p0 % 3) * 4);
    if (p1 < 15) frag_load_hi(bq, u + (size_t)(f + p1 / 3) * rstride + ch0 + (p1 % 3) * 4);
    else frag_zero_hi(bq);
}

// NW = waves per workgroup: 8 (= the conv groups), or 9 when the row phases have a multiple of 9 units (F = 129: 9 frequency tiles x 2
// frames = 18 units, which 8 waves take in 3 rounds with 6 of them idle in the last); the ninth wave sits out the group phases.
// WGF: the conv weight gradient is contracted here too (bf16, TT = 2): between phase 1 and phase 2 both of its operands — dv and LN(x) — sit
// in the two LDS images as row-major [frequency][channel] arrays, which is what a token-contraction needs: transposing reads give MFMA
// fragments with K = 32 frequencies, wave g accumulates dW[g][12 x 12] per tap (5 tiles) + the bias column sums, and the workgroup's partial
// goes out as one row (the whole [96][12][5] weight + [96] bias in dW's own memory order) behind its affine sums; affine_reduce folds the rows.
// The separate path wrote dv (S) and had wgrad.hip read dv + x (2 S) again: 0.6 GB per f-conv at batch 32 against 2 x 94 MB of partial rows.
template <class T, int TT, int NW, bool WGF>
__global__ __launch_bounds__(64 * NW, NW > 8 ? 3 : 2)  // (9 waves: one SIMD hosts three of them -> at most 168 VGPRs)
void fconv_bwd_kernel(nbss_cfg c, const float* __restrict__ lnw, const float* __restrict__ lnb, const float* __restrict__ cb,
                      const float* __restrict__ slope, float* __restrict__ part, const T* __restrict__ Wp, const T* __restrict__ WpT,
                      const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, float* __restrict__ stats, T* __restrict__ dvout, int flip) {
    NBSS_LDS(smem);
    const int F = c.F, T_ = c.T, ntt = cdiv(T_, TT);
    const int bid = flip_bid(flip);
    const int b = bid / ntt, t0 = (bid % ntt) * TT;
    const int mtf = cdiv(F, 16), FP = mtf * 16 + 4, ntile = mtf * TT;
    constexpr int ROW = TT * FC_LD;  // elements per frequency row of an image
    T* u = reinterpret_cast<T*>(smem);          // [FP][TT][LD]  LN(x) (phases 0-1), then du (phases 2-3); image row f+2
    T* dvb = u + (size_t)FP * ROW;               // [FP][TT][LD]  dy (phase 0), then dv in place
    float* aff = reinterpret_cast<float*>(dvb + (size_t)FP * ROW);  // [3H] LN weight | LN bias | PReLU slope gradient sums
    float* lnp = aff + 3 * FC_H;                 // [2H] gamma | beta
    float* affw = lnp + 2 * FC_H;                // [NW][2H] per-wave LayerNorm affine sums: each wave adds its units in its own order, the waves are
                                                 // added in wave order at the end (LDS float atomics made the partial row depend on the wave timing)
    PHASE_BEGIN(affw + NW * 2 * FC_H);
    const int tid = threadIdx.x, lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
#ifdef NBSS_FCONV_NO_DMA  // (A/B flavour)
    constexpr bool DMA = false;
#else
    constexpr bool DMA = sizeof(T) == 2;  // bf16 stream: the slab arrives through global -> LDS copies (phase 0)
#endif

    Frag<T> af[FC_KS], at[FC_KS];
#pragma unroll
    for (int ks = 0; ks < FC_KS; ++ks) {
        wfrag_load(af[ks], Wp, w < FC_G ? w : 0, FC_KS, ks);
        wfrag_load(at[ks], WpT, w < FC_G ? w : 0, FC_KS, ks);
    }
    const bool gwave = NW == FC_G || w < FC_G;  // this wave owns a conv group in the group phases
    for (int i = tid; i < 3 * FC_H; i += blockDim.x) aff[i] = 0.f;
    for (int i = tid; i < NW * 2 * FC_H; i += blockDim.x) affw[i] = 0.f;
    for (int i = tid; i < 2 * FC_H; i += blockDim.x) lnp[i] = i < FC_H ? lnw[i] : lnb[i - FC_H];
    constexpr int VZ = VecOf<T>::N;  // elements per 16-byte store (ROW is a multiple of 8)
    for (int i = tid; i < 4 * ROW / VZ; i += blockDim.x) {  // halo rows (f = -2, -1, 16 mtf, 16 mtf + 1) of both images
        const int e = i * VZ, hr = e / ROW, rr = hr < 2 ? hr : mtf * 16 + hr, off = e % ROW;
        vec_zero(u + (size_t)rr * ROW + off);
        vec_zero(dvb + (size_t)rr * ROW + off);
    }

    // ---- phase 0 (rows): x, dy -> registers; LayerNorm -> u; dy -> dvb; row statistics out ----
    // A wave keeps x, dy and the statistics of its FIRST unit in registers until phase 3; further units (F = 129 leaves one
    // frequency row for a ninth tile) are re-read from global there, so the register footprint is that of one unit.
    auto row_load = [&](int ti, Frag<T> (&xq)[BK_KS], Frag<T> (&dq)[BK_KS]) {
        const int ft = ti / TT, tt = ti % TT, f = ft * 16 + l15, t = t0 + tt;
        const bool valid = f < F && t < T_;
        const size_t n = ((size_t)b * F + f) * T_ + t;
#pragma unroll
        for (int ks = 0; ks < BK_KS; ++ks) {
            if (valid) {
                frag_load(xq[ks], x + n * FC_H + ks * 32 + 8 * g4);
                frag_load(dq[ks], dy + n * FC_H + ks * 32 + 8 * g4);
            } else {
                frag_zero(xq[ks]);
                frag_zero(dq[ks]);
            }
        }
    };
    auto row_stats = [&](const Frag<T> (&xq)[BK_KS], float& mean, float& rstd) {
        float sum = 0.f;
#pragma unroll
        for (int ks = 0; ks < BK_KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += frag_get(xq[ks], j);
        mean = wave_sum16(sum) * (1.0f / FC_H);
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < BK_KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = frag_get(xq[ks], j) - mean;
                q += d * d;
            }
        rstd = rsqrtf(wave_sum16(q) * (1.0f / FC_H) + 1e-5f);
    };
    auto row_fwd = [&](int ti, const Frag<T> (&xq)[BK_KS], const Frag<T> (&dq)[BK_KS], float mean, float rstd) {
        const int ft = ti / TT, tt = ti % TT, f = ft * 16 + l15, t = t0 + tt;
        const bool valid = f < F && t < T_;
        if (!WGF && valid && g4 == 0) {  // (row statistics: only the separate weight-gradient kernel needs them)
            const size_t n = ((size_t)b * F + f) * T_ + t;
            stats[n * 2] = mean;
            stats[n * 2 + 1] = rstd;
        }
        T* ur = u + (size_t)(f + 2) * ROW + tt * FC_LD;
        T* dr_ = dvb + (size_t)(f + 2) * ROW + tt * FC_LD;
#pragma unroll
        for (int ks = 0; ks < BK_KS; ++ks) {
            const int c0 = ks * 32 + 8 * g4;
            float gm[8], bt[8], o[8], d[8];
            load8(lnp + c0, gm);
            load8(lnp + FC_H + c0, bt);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = valid ? (frag_get(xq[ks], j) - mean) * rstd * gm[j] + bt[j] : 0.f;
                d[j] = frag_get(dq[ks], j);
            }
            store8(ur + c0, o);
            if (!DMA) store8(dr_ + c0, d);
        }
    };
    Frag<T> xr[BK_KS], dr[BK_KS];
    float rmean = 0.f, rrstd = 0.f;
    constexpr bool PF = false;
    Frag<T> xr2[BK_KS], dr2[BK_KS];
    if constexpr (DMA) {
        // bf16 stream: the whole slab of x and dy comes in as ONE burst of global -> LDS copies (16-byte pieces, no register stop), raw x into the
        // u image and dy into the dvb image — one memory round trip per workgroup instead of one per unit of a wave (a wave walks its two units
        // serially and the images leave room for one workgroup per CU).  Piece q of an image: row q / 13 (= f TT + tt), 16-byte column q % 13
        // (column 12 = the row padding, never written, never read).
        const int wu = wave_id_u(), Q = F * TT * 13;
        for (int i = wu; i * 64 < Q; i += NW) {
            const int q = i * 64 + lane, r = q / 13, cc = q - r * 13, f = r / TT, tt = r - f * TT;
            if (q < Q && cc < 12 && t0 + tt < T_) {
                const size_t n = ((size_t)b * F + f) * T_ + t0 + tt;
                dma16_to_lds(reinterpret_cast<char*>(u + 2 * ROW) + (size_t)i * 1024, x + n * FC_H + cc * 8);
                dma16_to_lds(reinterpret_cast<char*>(dvb + 2 * ROW) + (size_t)i * 1024, dy + n * FC_H + cc * 8);
            }
        }
        // rows the copies do not reach: the padding frequencies of the last tile, and the missing frame of a slab that ends the sequence
        for (int i = tid; i < (mtf * 16 - F) * ROW / VZ; i += blockDim.x) {
            vec_zero(u + (size_t)(F + 2) * ROW + i * VZ);
            vec_zero(dvb + (size_t)(F + 2) * ROW + i * VZ);
        }
        if (t0 + TT > T_)
            for (int i = tid; i < F * FC_LD / VZ; i += blockDim.x) {
                const int f = i / (FC_LD / VZ), e = (i % (FC_LD / VZ)) * VZ;
                for (int tt = T_ - t0; tt < TT; ++tt) {
                    vec_zero(u + (size_t)(f + 2) * ROW + tt * FC_LD + e);
                    vec_zero(dvb + (size_t)(f + 2) * ROW + tt * FC_LD + e);
                }
            }
        dma_wait_all();
        lds_barrier();  // images, lnp
        auto row_load_lds = [&](int ti, Frag<T> (&xq)[BK_KS], Frag<T> (&dq)[BK_KS]) {
            const int ft = ti / TT, tt = ti % TT, f = ft * 16 + l15;
#pragma unroll
            for (int ks = 0; ks < BK_KS; ++ks) {
                frag_load(xq[ks], u + (size_t)(f + 2) * ROW + tt * FC_LD + ks * 32 + 8 * g4);
                frag_load(dq[ks], dvb + (size_t)(f + 2) * ROW + tt * FC_LD + ks * 32 + 8 * g4);
            }
        };
        if (w < ntile) {
            row_load_lds(w, xr, dr);
            row_stats(xr, rmean, rrstd);
            row_fwd(w, xr, dr, rmean, rrstd);
        }
        for (int ti = w + NW; ti < ntile; ti += NW) {
            Frag<T> xq[BK_KS], dq[BK_KS];
            float mean, rstd;
            row_load_lds(ti, xq, dq);
            row_stats(xq, mean, rstd);
            row_fwd(ti, xq, dq, mean, rstd);
        }
    } else {
    if (w < ntile) row_load(w, xr, dr);
    // the wave's second unit is requested together with the first (bf16 stream: the row phases are bound by exposed HBM latency)
    // (measured: 6.69 -> 7.18 ms/step with the prefetch on — 168 VGPRs and twice the loads in flight ahead of the LN phase; kept off)
    if (PF && w + NW < ntile) row_load(w + NW, xr2, dr2);
    lds_barrier();  // lnp
    if (w < ntile) {
        row_stats(xr, rmean, rrstd);
        row_fwd(w, xr, dr, rmean, rrstd);
    }
    if (PF && w + NW < ntile) {
        float mean, rstd;
        row_stats(xr2, mean, rstd);
        row_fwd(w + NW, xr2, dr2, mean, rstd);
    }
    for (int ti = w + (PF ? 2 : 1) * NW; ti < ntile; ti += NW) {
        Frag<T> xq[BK_KS], dq[BK_KS];
        float mean, rstd;
        row_load(ti, xq, dq);
        row_stats(xq, mean, rstd);
        row_fwd(ti, xq, dq, mean, rstd);
    }
    }
    PHASE(0);
    lds_barrier();
    PHASE(1);

    // ---- phase 1 (groups): conv forward recompute, PReLU', dv in place of dy ----
    const int ch = w * FC_CG + 4 * g4;  // this lane's 4 channels of group w (g4 == 3: padding rows)
    const bool cvalid = g4 < 3 && gwave;
    float cbv[4] = {0.f, 0.f, 0.f, 0.f}, slv[4] = {0.f, 0.f, 0.f, 0.f}, dsl[4] = {0.f, 0.f, 0.f, 0.f};
    if (cvalid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { cbv[r] = cb[ch + r]; slv[r] = slope[ch + r]; }
    }
    for (int ti = 0; ti < (gwave ? ntile : 0); ++ti) {
        const int ft = ti / TT, tt = ti % TT, f = ft * 16 + l15;
        f32x4 acc = F32X4_ZERO;
#pragma unroll
        for (int ks = 0; ks < FC_KS; ++ks) {
            Frag<T> bq;
            fconv_bfrag<T>(bq, u + tt * FC_LD, ROW, f, w * FC_CG, ks);
            acc = mma(af[ks], bq, acc);
        }
        if (cvalid) {
            T* pd = dvb + (size_t)(f + 2) * ROW + tt * FC_LD + ch;
            float dyv[4], dv[4];
            load4(pd, dyv);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[r] + cbv[r];
                dv[r] = v > 0.f ? dyv[r] : slv[r] * dyv[r];
                if (v <= 0.f) dsl[r] += dyv[r] * v;
            }
            store4(pd, dv[0], dv[1], dv[2], dv[3]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float s2 = row_sum16(dsl[r]);
        if (l15 == 0 && cvalid) atomicAdd(aff + 2 * FC_H + ch + r, s2);
    }
    PHASE(2);
    lds_barrier();
    PHASE(3);

    constexpr int PROW = 4 * FC_H;  // floats per fp32 partial row: the three affine sums + the conv bias sums (layout.h NBSS_FC_PROW); the bf16 rows follow all of them
    if constexpr (WGF) {
        // ---- weight gradient of group w: dW[o][i][tap] = sum_{f,tt} dv[f][o] LN(x)[f + tap - 2][i], db[o] = sum dv[f][o] ----
        if (gwave) {
            f32x4 wacc[5], bsum = F32X4_ZERO;
#pragma unroll
            for (int tap = 0; tap < 5; ++tap) wacc[tap] = F32X4_ZERO;
            Frag<T> ones;
#pragma unroll
            for (int jq = 0; jq < 8; ++jq) frag_set(ones, jq, 1.0f);
            const int nk = cdiv(mtf * 16, 32);
            const bool odd = (mtf & 1) != 0;  // the last k-step has 16 real rows: the upper halves of the fragments are cleared
            const int roff = (4 * g4 + (l15 >> 2)) * ROW + w * FC_CG + 4 * (l15 & 3);
            for (int tt = 0; tt < TT; ++tt)
                for (int ks = 0; ks < nk; ++ks) {
                    const bool half = odd && ks == nk - 1;
                    Frag<T> fa;
                    frag_load_tr(fa, dvb + (size_t)(2 + 32 * ks) * ROW + tt * FC_LD + roff, ROW);
                    if (half) frag_zero_hi(fa);
                    bsum = mma(fa, ones, bsum);
#pragma unroll
                    for (int tap = 0; tap < 5; ++tap) {
                        Frag<T> fb;
                        frag_load_tr(fb, u + (size_t)(32 * ks + tap) * ROW + tt * FC_LD + roff, ROW);
                        if (half) frag_zero_hi(fb);
                        wacc[tap] = mma(fa, fb, wacc[tap]);
                    }
                }
            // C tile: lane = input channel i (l15), rows = output channels 4 g4 + r; valid 12 x 12.  The workgroup's partial of the weight gradient leaves in
            // bf16 as [tap][group][i][12 outputs] — a lane's four output channels are ONE 8-byte store and a wave's store instruction covers one contiguous
            // 288-byte run (round 6; rounds 2-5: fp32 in dW's own [o][i][tap] order = twenty scattered 4-byte stores per lane, 68 of the launch's 335 us
            // with the stores knocked out).  Only the per-slab partial is rounded (under the reference's autocast the weight gradient of a bf16 convolution
            // IS a bf16 tensor); the sum over the B T slabs is fp32 (fconv_part_final_kernel).  The bias sums stay fp32, in the row's head.
            bf16_t* prow16 = reinterpret_cast<bf16_t*>(part + (size_t)gridDim.x * PROW) + (size_t)bid * FC_P16;
            float* pb = part + (size_t)bid * PROW + 3 * FC_H;
#ifdef NBSS_FC_KO_PROW  // (timing knock-out, A/B flavour: no partial-row stores; the contraction stays: the compiler cannot see that `part` is never null)
            if (part == nullptr) {
#else
            if (l15 < FC_CG && g4 < 3) {
#endif
#pragma unroll
                for (int tap = 0; tap < 5; ++tap) {
                    const u32x2 v = {pack2bf(wacc[tap][0], wacc[tap][1]), pack2bf(wacc[tap][2], wacc[tap][3])};
                    *reinterpret_cast<u32x2*>(prow16 + ((size_t)(tap * FC_G + w) * FC_CG + l15) * FC_CG + 4 * g4) = v;
                }
                if (l15 == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pb[w * FC_CG + 4 * g4 + r] = bsum[r];
                }
            }
        }
    }

    // ---- phase 2 (groups): transposed conv -> du into the u image; dv rows -> global ----
    for (int ti = 0; ti < (gwave ? ntile : 0); ++ti) {
        const int ft = ti / TT, tt = ti % TT, f = ft * 16 + l15;
        f32x4 acc = F32X4_ZERO;
#pragma unroll
        for (int ks = 0; ks < FC_KS; ++ks) {
            Frag<T> bq;
            fconv_bfrag<T>(bq, dvb + tt * FC_LD, ROW, f, w * FC_CG, ks);
            acc = mma(at[ks], bq, acc);
        }
        if (cvalid) store4(u + (size_t)(f + 2) * ROW + tt * FC_LD + ch, acc[0], acc[1], acc[2], acc[3]);
    }
    if (!WGF) {
        constexpr int VN = VecOf<T>::N, VPR = FC_H / VN;
        for (int i = tid; i < F * TT * VPR; i += blockDim.x) {
            const int f = i / (TT * VPR), rem = i % (TT * VPR), tt = rem / VPR, off = (rem % VPR) * VN;
            if (t0 + tt < T_) vec_copy(dvout + (((size_t)b * F + f) * T_ + t0 + tt) * FC_H + off, dvb + (size_t)(f + 2) * ROW + tt * FC_LD + off);
        }
    }
    PHASE(4);
    lds_barrier();
    PHASE(5);

    // ---- phase 3 (rows): LayerNorm backward + residual from registers (x, dy) and the du image ----
    // two sweeps over the row (sums, then outputs) that re-read du from LDS instead of keeping 48 intermediates alive: the
    // kernel has to stay under 128 VGPRs for two workgroups per CU
    auto row_bwd = [&](int ti, const Frag<T> (&xq)[BK_KS], const Frag<T> (&dq)[BK_KS], float mean, float rstd0) {
        const int ft = ti / TT, tt = ti % TT, f = ft * 16 + l15, t = t0 + tt;
        const bool valid = f < F && t < T_;
        const size_t n = ((size_t)b * F + f) * T_ + t;
        const T* ur = u + (size_t)(f + 2) * ROW + tt * FC_LD;
        const float rstd = valid ? rstd0 : 0.f;
        float m1 = 0.f, m2 = 0.f;
        // LayerNorm affine sums of the unit: per channel the sum over the tile's 16 rows of du xhat (gamma) and du (beta) — 48 row sums.  They are taken
        // SIXTEEN at a time (row_reduce16x16: the lanes of a row end up with one total each and add it to the wave's own slot row, all lanes busy) in
        // three groups: xhat-products of k-steps 0-1 | xhat-products and plain sums of k-step 2 | plain sums of k-steps 0-1.  (Round 4: one 4-step DPP
        // reduction per value and 48 single-lane LDS atomics per unit.)
        static_assert(BK_KS == 3, "three k-steps of 32 channels");
        float* aw = affw + w * 2 * FC_H;
        float va[16];
        auto ks_vals = [&](int ks, float (&dvv)[8], float (&xhv)[8], bool stats) {
            const int c0 = ks * 32 + 8 * g4;
            float duv[8], gm[8];
            load8(ur + c0, duv);
            if (stats) load8(lnp + c0, gm);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dvv[j] = valid ? duv[j] : 0.f;
                xhv[j] = (frag_get(xq[ks], j) - mean) * rstd;
                if (stats) {
                    m1 += dvv[j] * gm[j];
                    m2 += dvv[j] * gm[j] * xhv[j];
                }
            }
        };
        {
            float dvv[8], xhv[8];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ks_vals(ks, dvv, xhv, true);
#pragma unroll
                for (int j = 0; j < 8; ++j) va[8 * ks + j] = dvv[j] * xhv[j];
            }
            const float r0 = row_reduce16x16(va);  // lane l15: k-step l15 >> 3, channel l15 & 7 of the lane group
            aw[(l15 >> 3) * 32 + 8 * g4 + (l15 & 7)] += r0;
            ks_vals(2, dvv, xhv, true);
#pragma unroll
            for (int j = 0; j < 8; ++j) va[j] = dvv[j] * xhv[j];
#pragma unroll
            for (int j = 0; j < 8; ++j) va[8 + j] = dvv[j];
            const float r1 = row_reduce16x16(va);
            aw[(l15 >> 3) * FC_H + 64 + 8 * g4 + (l15 & 7)] += r1;  // lanes 0-7: gamma sums of k-step 2, lanes 8-15: its beta sums
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ks_vals(ks, dvv, xhv, false);
#pragma unroll
                for (int j = 0; j < 8; ++j) va[8 * ks + j] = dvv[j];
            }
            const float r2 = row_reduce16x16(va);
            aw[FC_H + (l15 >> 3) * 32 + 8 * g4 + (l15 & 7)] += r2;
        }
        m1 = wave_sum16(m1) * (1.0f / FC_H);
        m2 = wave_sum16(m2) * (1.0f / FC_H);
        if (valid) {
#pragma unroll
            for (int ks = 0; ks < BK_KS; ++ks) {
                const int c0 = ks * 32 + 8 * g4;
                float duv[8], gm[8], o[8];
                load8(ur + c0, duv);
                load8(lnp + c0, gm);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (frag_get(xq[ks], j) - mean) * rstd;
                    o[j] = frag_get(dq[ks], j) + rstd * (duv[j] * gm[j] - m1 - xh * m2);
                }
                store8(dx + n * FC_H + c0, o);
            }
        }
    };
    if (PF && w + NW < ntile) row_load(w + NW, xr2, dr2);  // requested before the first unit's LayerNorm backward
    if (w < ntile) row_bwd(w, xr, dr, rmean, rrstd);
    if (PF && w + NW < ntile) {
        float mean, rstd;
        row_stats(xr2, mean, rstd);
        row_bwd(w + NW, xr2, dr2, mean, rstd);
    }
    for (int ti = w + (PF ? 2 : 1) * NW; ti < ntile; ti += NW) {
        Frag<T> xq[BK_KS], dq[BK_KS];
        float mean, rstd;
        row_load(ti, xq, dq);
        row_stats(xq, mean, rstd);
        row_bwd(ti, xq, dq, mean, rstd);
    }
    PHASE(6);
    lds_barrier();
    for (int i = threadIdx.x; i < 3 * FC_H; i += blockDim.x) {
        float v = aff[i];  // (PReLU slope sums: one contributor per channel)
        if (i < 2 * FC_H)
            for (int k = 0; k < NW; ++k) v += affw[k * 2 * FC_H + i];
        part[(size_t)bid * (WGF ? PROW : 3 * FC_H) + i] = v;
    }
    PHASE(7);
    PHASE_END();
}
PHASE_READER(nbss_phase_read_fconv_bwd)

template <class T, int TT, int NW, bool WGF>
static int fconv_bwd_t(const nbss_cfg& c, const float* P, float* part, const void* packed, int layer, int which, const void* x, const void* dy, void* dx,
                       float* stats, void* dv, hipStream_t st) {
    const int mtf = cdiv(c.F, 16);
    if (mtf > FC_MTF_BIG) return NBSS_EUNSUPPORTED;
    const size_t lds = (size_t)2 * (mtf * 16 + 4) * TT * FC_LD * sizeof(T) + (5 + 2 * NW) * FC_H * sizeof(float) + PHASE_LDS_BYTES;
    if (lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    const int lw = which ? P_FC2_LN_W : P_FC1_LN_W, lb = which ? P_FC2_LN_B : P_FC1_LN_B, sl = which ? P_FC2_PRELU : P_FC1_PRELU;
    const T* pk = (const T*)packed;
    int e = NBSS_SET_MAX_LDS((fconv_bwd_kernel<T, TT, NW, WGF>), lds);
    if (e) return e;
    dim3 grid(c.B * cdiv(c.T, TT)), block(64 * NW);
    ProfScope ps(PK_FCONV_B, st);
    NBSS_LAUNCH((fconv_bwd_kernel<T, TT, NW, WGF>), grid, block, lds, st, c, P + param_off(c, layer, lw), P + param_off(c, layer, lb),
                P + param_off(c, layer, which ? P_FC2_B : P_FC1_B), P + param_off(c, layer, sl), part, pk + pack_off(c, layer, which ? K_FC2 : K_FC1),
                pk + pack_off(c, layer, which ? K_FC2_T : K_FC1_T), (const T*)x, (const T*)dy, (T*)dx, stats, (T*)dv, walk_flip_next());
    return NBSS_CHECK_LAUNCH();
}

// frames per workgroup of the bf16 backward.  Round 5: ONE — the images are 62 KB, the 8-wave instance holds 122 VGPRs: two workgroups share a CU and one's
// prologue (the slab's memory round trip, a third of the wave time) overlaps the other's math: 403 -> 294 us per launch in order, the step 657 -> 688 utt/s
// (same box).  (Rounds 1-4 ran two frames per workgroup, one per CU: measured 2.2x better in round 1, before the LDS-DMA prologue and the fused weight gradient;
// -DNBSS_FC_TT2 builds that variant.)  Twice the partial rows for the fold (one [96][12][5] weight gradient per workgroup): 197 MB per launch at batch 32.
#ifdef NBSS_FC_TT2
#define FC_BWD_TT 2
#else
#define FC_BWD_TT 1
#endif

#define TV_RSL_MAX 64  // (tconvffn_s.hip: TV_RSL, the most slices part16_slices_launch writes)
// tconvffn_s.hip: fp32 slice sums of bf16 partial rows [nrows][p16] (fixed order; *nsl = slices written)
int part16_slices_launch(const void* part16, int nrows, float* slices, int p16, int* nsl, hipStream_t st, bool batch);
// dW[o][i][tap] += the slices' sums of the [tap][group][i][12 outputs] rows, in slice order (one owner per element: bitwise repeatable)
static_assert(FC_H == FK_H && FC_CG == FK_FCG && FC_G == FK_FG && FC_P16 == FK_FC_P16, "foldk.h");
__global__ __launch_bounds__(256) void fconv_part_final_kernel(const float* __restrict__ slices, int nsl, float* __restrict__ dW) {
    fk_fconv_final(slices, nsl, dW, (int)blockIdx.x);  // (the body lives in foldk.h: fold.hip's table kernel runs it too)
}

int fconv_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, int layer, int which, const void* x, const void* dy, void* dx,
                   void* ws, hipStream_t st, const Side* sd) {
    if (c.H != FC_H) return gb_fconv_bwd(c, P, G, layer, which, x, dy, dx, ws, st, sd);
    const size_t N = (size_t)c.B * c.F * c.T;
    float* stats = (float*)ws;
    void* dv = (char*)ws + ws_align(N * 2 * sizeof(float));
    float* part = (float*)((char*)ws + ws_part_offset(c));
#ifdef NBSS_FC_NW8
    const bool nine = false;
#else
    // nine waves when the row phases have a multiple of nine units AND the instance is alone on its CU (two-frame slabs); one-frame slabs: two 8-wave
    // workgroups per CU at <= 128 VGPRs (two of nine waves would not fit the register file)
    const bool nine = FC_BWD_TT == 2 && (cdiv(c.F, 16) * FC_BWD_TT) % 9 == 0;
#endif
    // (F > 160, the 16-kHz geometry: the two images of a 2-frame slab no longer fit the LDS -> one frame per workgroup; the fp32 stream
    //  images of even one frame are 229 KB at F = 257: fconv_bwd_t returns NBSS_EUNSUPPORTED there, fp32 TRAINING stops at F = 160)
    const bool big = c.F > 16 * FC_MTF_MAX;
#ifdef NBSS_FC_NOWGF
    const bool fused = false;
#else
    const bool fused = c.dtype == NBSS_BF16 && !big;  // the conv weight gradient is contracted inside the kernel (two-frame bf16 slabs)
#endif
    if (fused) part = (float*)((char*)ws + ws_fcpart_offset(c));
    int e = c.dtype != NBSS_BF16 ? fconv_bwd_t<float, 1, 8, false>(c, P, part, packed, layer, which, x, dy, dx, stats, dv, st)
            : big                ? fconv_bwd_t<bf16_t, 1, 8, false>(c, P, part, packed, layer, which, x, dy, dx, stats, dv, st)
            : fused && nine      ? fconv_bwd_t<bf16_t, FC_BWD_TT, 9, true>(c, P, part, packed, layer, which, x, dy, dx, stats, dv, st)
            : fused              ? fconv_bwd_t<bf16_t, FC_BWD_TT, 8, true>(c, P, part, packed, layer, which, x, dy, dx, stats, dv, st)
            : nine               ? fconv_bwd_t<bf16_t, FC_BWD_TT, 9, false>(c, P, part, packed, layer, which, x, dy, dx, stats, dv, st)
                                 : fconv_bwd_t<bf16_t, FC_BWD_TT, 8, false>(c, P, part, packed, layer, which, x, dy, dx, stats, dv, st);
    if (e) return e;
    const int nwg = c.dtype == NBSS_BF16 && !big ? c.B * cdiv(c.T, FC_BWD_TT) : c.B * c.T;
    AffSegs sg;
    sg.off[0] = param_off(c, layer, which ? P_FC2_LN_W : P_FC1_LN_W); sg.cnt[0] = FC_H;
    sg.off[1] = param_off(c, layer, which ? P_FC2_LN_B : P_FC1_LN_B); sg.cnt[1] = FC_H;
    sg.off[2] = param_off(c, layer, which ? P_FC2_PRELU : P_FC1_PRELU); sg.cnt[2] = FC_H;
    sg.n = fused ? 4 : 3;
    sg.off[3] = param_off(c, layer, which ? P_FC2_B : P_FC1_B); sg.cnt[3] = FC_H;  // (fused: the fp32 rows carry the conv bias sums behind the affine sums)
    const hipStream_t gs = side_fork(sd, st);  // the folds and the weight-gradient problem only produce parameter gradients (side.h)
    FoldScope fs(gs, (char*)ws + ws_wgpart_offset(c), WGPART_BYTES, N);  // (fold.h: the sub-block's folds leave as one launch per stage)
    if ((e = affine_reduce_launch(part, nwg, sg, G, gs))) return e;
    if (fused) {  // the bf16 rows of the conv weight gradient: slice sums in fp32 (the idle wgrad partial region is the scratch), then one owner per element
        float* slices = (float*)((char*)ws + ws_wgpart_offset(c));
        bool batch = false;
        if (g_fold) {
            int err;
            void* sl = g_fold->alloc((size_t)TV_RSL_MAX * FC_P16 * sizeof(float), &err);
            if (err) return err;
            if (sl) {
                slices = (float*)sl;
                batch = true;
            } else if ((err = g_fold->flush())) {  // (larger than the pool: on its own, behind everything pending)
                return err;
            }
        }
        int nsl = 0;
        if ((e = part16_slices_launch(part + (size_t)nwg * 4 * FC_H, nwg, slices, FC_P16, &nsl, gs, batch))) return e;
        if (batch) {
            FoldItem it;
            it.kind = FK_FCONV_FINAL;
            it.gx = (FC_P16 + 255) / 256; it.gy = 1; it.nblk = it.gx;
            it.u.p16.part16 = nullptr; it.u.p16.nrows = nwg; it.u.p16.p16 = FC_P16; it.u.p16.nsl = nsl; it.u.p16.slices = slices;
            it.u.p16.G = G + param_off(c, layer, which ? P_FC2_W : P_FC1_W);
            if ((e = g_fold->add(2, it))) return e;
            return fs.end();
        }
        NBSS_FOLD_LAUNCH(fconv_part_final_kernel, dim3((FC_P16 + 255) / 256), dim3(256), 0, gs, (const float*)slices, nsl, G + param_off(c, layer, which ? P_FC2_W : P_FC1_W));
        return NBSS_CHECK_LAUNCH();
    }
    // conv weight: dW[o][i][tap] = sum_n dv[n][o] LN(x)[n + (tap-2) T][i]   (shift along F = T rows), bias = colsum(dv)
    WgradArgs a;
    a.part = (float*)((char*)ws + ws_wgpart_offset(c));
    a.mvalid = 0; a.nvalid = 0;
    a.Ntok = (int)N; a.F = c.F; a.T = c.T; a.shift_stride = c.T; a.shift_dim = 1; a.groups = c.f_groups; a.taps = c.f_ks;
    a.A = dv; a.lda = FC_H; a.MA = FC_H; a.B = x; a.ldb = FC_H; a.NB = FC_H;
    a.stats = stats; a.gamma = P + param_off(c, layer, which ? P_FC2_LN_W : P_FC1_LN_W); a.beta = P + param_off(c, layer, which ? P_FC2_LN_B : P_FC1_LN_B);
    a.dW = G + param_off(c, layer, which ? P_FC2_W : P_FC1_W); a.dbias = G + param_off(c, layer, which ? P_FC2_B : P_FC1_B);
    if ((e = wgrad_launch(a, c.dtype, gs))) return e;
    return fs.end();
}

template <class T, int TT, int GPW, int MTF, int HH>
static int fconv_fwd_t(const nbss_cfg& c, const float* P, const void* packed, int layer, int which, const void* x, void* y, hipStream_t st) {
    const int mtf = cdiv(c.F, 16);
    if (mtf > MTF) return NBSS_EUNSUPPORTED;
    const size_t lds = (size_t)(mtf * 16 + 4) * TT * (HH + 8) * sizeof(T);  // (rows padded by 8 elements: see HHP in the kernel)
    if (lds > 160 * 1024) return NBSS_EUNSUPPORTED;  // (fp32 stream, dim_hidden 192, F = 257: 212 KB)
    const float* lnw = P + param_off(c, layer, which ? P_FC2_LN_W : P_FC1_LN_W);
    const float* lnb = P + param_off(c, layer, which ? P_FC2_LN_B : P_FC1_LN_B);
    const float* cb = P + param_off(c, layer, which ? P_FC2_B : P_FC1_B);
    const float* sl = P + param_off(c, layer, which ? P_FC2_PRELU : P_FC1_PRELU);
    const T* Wp = (const T*)packed + pack_off(c, layer, which ? K_FC2 : K_FC1);
    int e = NBSS_SET_MAX_LDS((fconv_fwd_kernel<T, TT, GPW, MTF, HH>), lds);
    if (e) return e;
    dim3 grid(c.B * cdiv(c.T, TT)), block(64 * FC_G / GPW);
    ProfScope ps(PK_FCONV_F, st);
    NBSS_LAUNCH((fconv_fwd_kernel<T, TT, GPW, MTF, HH>), grid, block, lds, st, c, lnw, lnb, cb, sl, Wp, (const T*)x, (T*)y, walk_flip_next());
    return NBSS_CHECK_LAUNCH();
}

int fconv_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, int which, const void* x, void* y, hipStream_t st) {
    const bool bigF = c.F > 16 * FC_MTF_MAX;  // 16-kHz geometry: one frame per workgroup, 17 frequency tiles
    if (c.H == GeoL::H) {  // SpatialNet-large (forward only): 24 channels per group, one frame per workgroup
        if (bigF) return c.dtype == NBSS_BF16 ? fconv_fwd_t<bf16_t, 1, 1, FC_MTF_BIG, GeoL::H>(c, P, packed, layer, which, x, y, st)
                                              : fconv_fwd_t<float, 1, 2, FC_MTF_BIG, GeoL::H>(c, P, packed, layer, which, x, y, st);
        return c.dtype == NBSS_BF16 ? fconv_fwd_t<bf16_t, 1, 1, FC_MTF_MAX, GeoL::H>(c, P, packed, layer, which, x, y, st)
                                    : fconv_fwd_t<float, 1, 2, FC_MTF_MAX, GeoL::H>(c, P, packed, layer, which, x, y, st);
    }
    if (bigF)
        return c.dtype == NBSS_BF16 ? fconv_fwd_t<bf16_t, 1, 1, FC_MTF_BIG, GeoS::H>(c, P, packed, layer, which, x, y, st)
                                    : fconv_fwd_t<float, 1, 2, FC_MTF_BIG, GeoS::H>(c, P, packed, layer, which, x, y, st);
    if (c.dtype == NBSS_BF16) return fconv_fwd_t<bf16_t, 2, 1, FC_MTF_MAX, GeoS::H>(c, P, packed, layer, which, x, y, st);
    return fconv_fwd_t<float, 1, 2, FC_MTF_MAX, GeoS::H>(c, P, packed, layer, which, x, y, st);
}
