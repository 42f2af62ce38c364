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
#include <stdlib.h>
#include "launch.h"
#include "layout.h"
#include "prof.h"
#include "blocks.h"
#include "wgrad.h"
#include "fold.h"
#include "foldk.h"
#include "geom.h"
#include "side.h"

#define FL_H 96
#define FL_SQ 8
#define FL_TT_MAX 8  // frames per slab: 8, or 4 / 2 when 8 would leave most of the 256 CUs without a workgroup (full_tt)
#define FL_KS (FL_H / 32)
#define FL_MT (FL_H / 16)
#define FL_KSF_MAX 5  // ceil(160 / 32): the 8-kHz geometry (F <= 160)
#define FL_KSF_BIG 9  // ceil(272 / 32): 16 kHz (n_fft 512 -> F = 257)
#define FL_THREADS 512  // 8 waves: one workgroup per CU (256 slabs), so the waves of a workgroup are all the latency hiding there is


// LinearGroup pass: task = (squeeze channel ch, 16-row tile mt of the F x F matrix), acc = W[ch] tile x s[ch][:, tt]; the body between BEGIN and
// END is the task's epilogue.  Narrow slabs (TT < 8: small grids, one workgroup per CU, the passes are a chain of exposed fragment loads —
// 70 of the backward kernel's 96 us per launch at batch 2) request the NEXT task's weight fragments before this task's MFMAs.
#define FL_TASK_LOOP_BEGIN(NTASK, WMAT)                                                                                  \
    constexpr bool TPF = TT < 8;                                                                                         \
    Frag<T> anext[KSFM];                                                                                                 \
    if (TPF && w < (NTASK)) {                                                                                            \
        _Pragma("unroll") for (int ks = 0; ks < KSFM; ++ks)                                                              \
            if (ks < ksf) wfrag_load(anext[ks], (WMAT) + (size_t)(w / mtf) * mtf * ksf * 512, w % mtf, ksf, ks);         \
    }                                                                                                                    \
    for (int task = w; task < (NTASK); task += nw) {                                                                     \
        const int ch = task / mtf, mt = task % mtf;                                                                      \
        f32x4 acc = F32X4_ZERO;                                                                                          \
        Frag<T> a[KSFM]; /* all k-step fragments of the tile requested together (at most KSFM) */                        \
        if (TPF) {                                                                                                       \
            _Pragma("unroll") for (int ks = 0; ks < KSFM; ++ks) a[ks] = anext[ks];                                       \
            const int nx = task + nw;                                                                                    \
            if (nx < (NTASK)) {                                                                                          \
                _Pragma("unroll") for (int ks = 0; ks < KSFM; ++ks)                                                      \
                    if (ks < ksf) wfrag_load(anext[ks], (WMAT) + (size_t)(nx / mtf) * mtf * ksf * 512, nx % mtf, ksf, ks); \
            }                                                                                                            \
        } else {                                                                                                         \
            _Pragma("unroll") for (int ks = 0; ks < KSFM; ++ks)                                                          \
                if (ks < ksf) wfrag_load(a[ks], (WMAT) + (size_t)ch * mtf * ksf * 512, mt, ksf, ks);                     \
        }                                                                                                                \
        _Pragma("unroll") for (int ks = 0; ks < KSFM; ++ks) {                                                            \
            if (ks < ksf) {                                                                                              \
                Frag<T> bq;                                                                                              \
                if (l15 < TT) frag_load(bq, s + ((size_t)ch * TT + l15) * FK + ks * 32 + 8 * g4);                        \
                else frag_zero(bq);                                                                                      \
                acc = mma(a[ks], bq, acc);                                                                               \
            }                                                                                                            \
        }
#define FL_TASK_LOOP_END }

// KSFM: LinearGroup k-steps the fragment arrays are sized for; HH / NSQ = dim_hidden / dim_squeeze (geom.h)
template <class T, int KSFM, int HH, int NSQ, int TT>
#ifdef NBSS_FULLF_NOCAP
__global__ __launch_bounds__(FL_THREADS)
#else
__global__ __launch_bounds__(FL_THREADS, HH == 96 && TT == 8 ? 4 : 1)  // small geometry, full grids: <= 128 VGPRs, two workgroups per CU
#endif
void full_fwd_kernel(nbss_cfg c, const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                       const float* __restrict__ bs, const float* __restrict__ bfull,
                                                       const float* __restrict__ bu, const T* __restrict__ Wsq,
                                                       const T* __restrict__ Wfull, const T* __restrict__ Wusq,
                                                       const T* __restrict__ x, T* __restrict__ y, int flip) {
    NBSS_LDS(smem);
    const int F = c.F, T_ = c.T;
    const int mtf = cdiv(F, 16), ksf = cdiv(F, 32), FK = ksf * 32, FM = mtf * 16;
    T* s = reinterpret_cast<T*>(smem);             // [SQ][TT][FK]
    T* z = s + NSQ * TT * FK;                 // [FM][TT][SQ]
    const int ntt = cdiv(T_, TT);
    const int bid = flip_bid(flip);  // (launch.h: consecutive kernels of a walk traverse the utterances in opposite order)
    const int b = bid / ntt, t0 = (bid % ntt) * TT;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id(), nw = nthr >> 6;
    const int ntile = cdiv(F, 16 / TT);  // n-tiles of (16 / TT freqs x TT frames)

    for (int i = tid; i < NSQ * TT * FK + FM * TT * NSQ; i += nthr) store1(s + i, 0.f);
    lds_barrier();

    // ---- pass 1: LN + squeeze + SiLU ---------------------------------------------------------
    {
        Frag<T> a[(HH / 32)];
#pragma unroll
        for (int ks = 0; ks < (HH / 32); ++ks) wfrag_load(a[ks], Wsq, 0, (HH / 32), ks);
        float gam[(HH / 32)][8], bet[(HH / 32)][8];
#pragma unroll
        for (int ks = 0; ks < (HH / 32); ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gam[ks][j] = lnw[ks * 32 + 8 * g4 + j];
                bet[ks][j] = lnb[ks * 32 + 8 * g4 + j];
            }
        // software pipeline (the same as full_bwd's row loops): the x pieces of the wave's NEXT tile are requested before this tile's math —
        // one workgroup of 8 waves per CU, and each wave walks 8 tiles load -> LayerNorm -> MFMA (SQ_WAIT_ANY was 77 % of the wave cycles)
        const int tclf = t0 + (l15 % TT) < T_ ? t0 + (l15 % TT) : T_ - 1;
        auto row_of_f = [&](int nt) -> const T* {
            const int f = (16 / TT) * nt + (l15 / TT);
            return x + (((size_t)b * F + (f < F ? f : F - 1)) * T_ + tclf) * HH;
        };
        Frag<T> xnext[(HH / 32)];
        if (w < ntile) {
#pragma unroll
            for (int ks = 0; ks < (HH / 32); ++ks) frag_load(xnext[ks], row_of_f(w) + ks * 32 + 8 * g4);
        }
        for (int nt = w; nt < ntile; nt += nw) {
            const int f = (16 / TT) * nt + (l15 / TT), tt = l15 % TT;
            const bool valid = f < F && t0 + tt < T_;
            Frag<T> xcur[(HH / 32)];
#pragma unroll
            for (int ks = 0; ks < (HH / 32); ++ks) xcur[ks] = xnext[ks];
            {
                const T* nx = row_of_f(nt + nw < ntile ? nt + nw : nt);
#pragma unroll
                for (int ks = 0; ks < (HH / 32); ++ks) frag_load(xnext[ks], nx + ks * 32 + 8 * g4);
            }
            float v[(HH / 32)][8];
            float sum = 0.f;
#pragma unroll
            for (int ks = 0; ks < (HH / 32); ++ks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[ks][j] = keep_if(valid, frag_get(xcur[ks], j));
                    sum += v[ks][j];
                }
            }
            const float mean = wave_sum16(sum) * (1.0f / HH);
            float q = 0.f;
#pragma unroll
            for (int ks = 0; ks < (HH / 32); ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = v[ks][j] - mean;
                    q += d * d;
                }
            const float rstd = rsqrtf(wave_sum16(q) * (1.0f / HH) + 1e-5f);
            f32x4 acc = F32X4_ZERO;
#pragma unroll
            for (int ks = 0; ks < (HH / 32); ++ks) {
                Frag<T> u;
#pragma unroll
                for (int j = 0; j < 8; ++j) frag_set(u, j, (v[ks][j] - mean) * rstd * gam[ks][j] + bet[ks][j]);
                acc = mma(a[ks], u, acc);
            }
            if (valid && 4 * g4 < NSQ) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = 4 * g4 + r;
                    store1(s + ((size_t)ch * TT + tt) * FK + f, silu_f(acc[r] + bs[ch]));
                }
            }
        }
    }
    lds_barrier();

    // ---- pass 2: LinearGroup along F, one F x F matrix per squeeze channel -------------------
    {
        FL_TASK_LOOP_BEGIN(NSQ * mtf, Wfull)
        if (l15 < TT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = mt * 16 + 4 * g4 + r;
                if (k < F) store1(z + ((size_t)k * TT + l15) * NSQ + ch, acc[r] + bfull[ch * F + k]);
            }
        }
        FL_TASK_LOOP_END
    }
    lds_barrier();

    // ---- pass 3: unsqueeze + SiLU + residual -----------------------------------------------------
    {
        Frag<T> a[(HH / 16)];
#pragma unroll
        for (int mt = 0; mt < (HH / 16); ++mt) wfrag_load(a[mt], Wusq, mt, 1, 0);
        // the residual rows of the NEXT tile are requested before this tile's math (same pipeline as pass 1)
        const int tcl3 = t0 + (l15 % TT) < T_ ? t0 + (l15 % TT) : T_ - 1;
        auto row_of_3 = [&](int nt) -> const T* {
            const int f = (16 / TT) * nt + (l15 / TT);
            return x + (((size_t)b * F + (f < F ? f : F - 1)) * T_ + tcl3) * HH;
        };
        RawC4<T> rnext[(HH / 16)];
        if (w < ntile) {
#pragma unroll
            for (int mt = 0; mt < (HH / 16); ++mt) rawc_load(rnext[mt], row_of_3(w) + 16 * mt + 4 * g4);
        }
        for (int nt = w; nt < ntile; nt += nw) {
            const int f = (16 / TT) * nt + (l15 / TT), tt = l15 % TT;
            const bool valid = f < F && t0 + tt < T_;
            RawC4<T> rcur[(HH / 16)];
#pragma unroll
            for (int mt = 0; mt < (HH / 16); ++mt) rcur[mt] = rnext[mt];
            {
                const T* nx = row_of_3(nt + nw < ntile ? nt + nw : nt);
#pragma unroll
                for (int mt = 0; mt < (HH / 16); ++mt) rawc_load(rnext[mt], nx + 16 * mt + 4 * g4);
            }
            Frag<T> bq;
            if (8 * g4 < NSQ && f < F) frag_load(bq, z + ((size_t)f * TT + tt) * NSQ + 8 * g4);
            else frag_zero(bq);
            const size_t go = (((size_t)b * F + f) * T_ + t0 + tt) * HH;
#pragma unroll
            for (int mt = 0; mt < (HH / 16); ++mt) {
                f32x4 acc = mma(a[mt], bq, F32X4_ZERO);
                if (valid) {
                    const int ch = 16 * mt + 4 * g4;
                    float xv[4];
                    rawc_get(rcur[mt], xv);
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = xv[r] + round_to(silu_f(acc[r] + bu[ch + r]), x);
                    store4(y + go + ch, o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward (data gradient).  Same slab decomposition; the squeezed tensors are recomputed in LDS
// and walked backwards:
//   p1  s_pre = Ws LN(x) + bs, s = SiLU(s_pre)                      -> LDS + global s (wgrad operand)
//   p2  z = Wf s + bf                                               -> LDS + global z
//   p3  y_pre = Wu z + bu ; dy_pre = dy SiLU'(y_pre) ; dz = Wu^T dy_pre   -> LDS dz, global dy_pre, dz
//   p4  ds = Wf^T dz ; ds_pre = ds SiLU'(s_pre)                     -> LDS, global ds_pre
//   p5  du = Ws^T ds_pre ; LayerNorm backward + residual in registers -> dx, (mean, rstd)
// The three weight gradients (squeeze, LinearGroup, unsqueeze) are contracted by wgrad.hip.
#define FL_FKP(F) (((F) + 3) & ~3)   // padded F stride of the global s / dz operands
#define FL_WFR 15  // weight fragments staged in LDS by the backward kernel: Wusq 6 | WusqT 3 | WsqT 6

template <class T, int KSFM, int TT>
// bf16 stream: <= 128 VGPRs (a few spilled registers) so that two workgroups share a CU: 4.72 -> 4.17 ms per step together with the LDS-resident
// weight fragments and the LayerNorm affine sums moved out of the row loop; the software prefetch of the row loops (round 2: worth 10 % at one
// workgroup per CU) costs more registers than it hides latency at two (4.57 with, 4.17 without)
__global__ __launch_bounds__(FL_THREADS, sizeof(T) == 2 && TT == 8 ? 4 : 2) void full_bwd_kernel(nbss_cfg c, LayerPtrs lp, const float* __restrict__ P, float* __restrict__ part, int layer,
                                                       const T* __restrict__ Wsq, const T* __restrict__ Wfull, const T* __restrict__ Wusq,
                                                       const T* __restrict__ WsqT, const T* __restrict__ WfullT, const T* __restrict__ WusqT,
                                                       const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                       float* __restrict__ stats, T* __restrict__ s_out, T* __restrict__ dz_out,
                                                       T* __restrict__ z_out, T* __restrict__ dyp_out, T* __restrict__ dsp_out, int flip) {
    NBSS_LDS(smem);
    const int F = c.F, T_ = c.T;
    const int mtf = cdiv(F, 16), ksf = cdiv(F, 32), FK = ksf * 32, FM = mtf * 16, FKP = FL_FKP(F);
    T* s = reinterpret_cast<T*>(smem);             // [SQ][TT][FK]   s, later dz
    T* sp = s + FL_SQ * TT * FK;                // [FM][TT][SQ]   s_pre
    T* z = sp + FM * TT * FL_SQ;                // [FM][TT][SQ]   z, later ds_pre
    float* aff = reinterpret_cast<float*>(z + FM * TT * FL_SQ);  // [SQ][17] squeeze bias gradient sums (fp32), one slot per (channel, frequency tile) task —
                                                                 // added in tile order at the end (waves adding to one slot with LDS atomics: order-dependent bits)
    // the unsqueeze / squeeze weight fragments of the two row loops live in LDS, not in 60 registers per lane: with the LayerNorm affine
    // sums gone as well (below) the kernel fits 128 VGPRs and TWO workgroups share a CU — its row loops are bound by exposed latency
    T* wl = reinterpret_cast<T*>(aff + FL_SQ * 17);                    // [FL_WFR][512]
    const int ntt = cdiv(T_, TT);
    const int bid = flip_bid(flip);  // (launch.h: consecutive kernels of a walk traverse the utterances in opposite order)
    const int b = bid / ntt, t0 = (bid % ntt) * TT;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id(), nw = nthr >> 6;
    const int ntile = cdiv(F, 16 / TT);
    const float* lnw = lp.p[P_FULL_LN_W];
    const float* lnb = lp.p[P_FULL_LN_B];
    const float* bs = lp.p[P_SQ_B];
    const float* bfull = lp.p[P_FULL_B];
    const float* bu = lp.p[P_USQ_B];

    for (int i = tid; i < FL_SQ * TT * FK + 2 * FM * TT * FL_SQ; i += nthr) store1(s + i, 0.f);
    for (int i = tid; i < FL_SQ * 17; i += nthr) aff[i] = 0.f;
    {
        constexpr int VPF = 512 * (int)sizeof(T) / 16;  // 16-byte pieces per fragment (64 bf16, 128 fp32)
        for (int v = tid; v < FL_WFR * VPF; v += nthr) {
            const int fr = v / VPF, pc = v % VPF;
            const T* src = fr < 6 ? Wusq + (size_t)fr * 512 : fr < 9 ? WusqT + (size_t)(fr - 6) * 512 : WsqT + (size_t)(fr - 9) * 512;
            reinterpret_cast<u32x4*>(wl)[v] = reinterpret_cast<const u32x4*>(src)[pc];
        }
    }
    PHASE_BEGIN(wl + FL_WFR * 512);
    lds_barrier();
    PHASE(0);

    // the squeezed image s[c][tt][f] (s, later dz) is copied to its [B*T][SQ][FKP] wgrad operand as whole frequency rows (8-byte pieces of
    // a contiguous 2 FKP-byte run) after the barrier that completes it: the first version stored every element from the MFMA epilogue,
    // 2-byte stores scattered over 64 rows per instruction
    auto copy_sq_image = [&](T* __restrict__ dst) {
        constexpr int VE = 8 / sizeof(T);  // elements per 8-byte piece
        const int vpr = FKP / VE;          // FKP is a multiple of 4
        for (int i = tid; i < FL_SQ * TT * vpr; i += nthr) {
            const int rowi = i / vpr, v = i % vpr, ch = rowi / TT, tt = rowi % TT;
            if (t0 + tt >= T_) continue;
            // (image columns F..FK are never written and hold the zero fill: the operand's padding columns F..FKP come out zero)
            const u32x2 val = *reinterpret_cast<const u32x2*>(s + ((size_t)ch * TT + tt) * FK + v * VE);
            *reinterpret_cast<u32x2*>(dst + (((size_t)b * T_ + t0 + tt) * FL_SQ + ch) * FKP + v * VE) = val;
        }
    };

    // ---- p1: LN + squeeze ----
    {
        Frag<T> a[FL_KS];
#pragma unroll
        for (int ks = 0; ks < FL_KS; ++ks) wfrag_load(a[ks], Wsq, 0, FL_KS, ks);
        float gam[BK_KS][8], bet[BK_KS][8];
        load_ln_affine(lnw, lnb, gam, bet);
        for (int nt = w; nt < ntile; nt += nw) {
            const int f = (16 / TT) * nt + (l15 / TT), tt = l15 % TT;
            const bool valid = f < F && t0 + tt < T_;
            Frag<T> u[BK_KS];
            ln_strip96<T>(x + (((size_t)b * F + f) * T_ + t0 + tt) * FL_H, valid, gam, bet, u);
            f32x4 acc = F32X4_ZERO;
#pragma unroll
            for (int ks = 0; ks < FL_KS; ++ks) acc = mma(a[ks], u[ks], acc);
            if (valid && g4 < 2) {
                float pre[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = 4 * g4 + r;
                    pre[r] = acc[r] + bs[ch];
                    const float sv = silu_f(pre[r]);
                    store1(s + ((size_t)ch * TT + tt) * FK + f, sv);  // (the global copy for wgrad leaves from this image: copy_sq_image)
                }
                store4(sp + ((size_t)f * TT + tt) * FL_SQ + 4 * g4, pre[0], pre[1], pre[2], pre[3]);
            }
        }
    }
    PHASE(1);
    lds_barrier();
    PHASE(2);
    copy_sq_image(s_out);

    // ---- p2: z = Wf s + bf ----
    {
        FL_TASK_LOOP_BEGIN(FL_SQ * mtf, Wfull)
        if (l15 < TT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = mt * 16 + 4 * g4 + r;
                if (k < F) store1(z + ((size_t)k * TT + l15) * FL_SQ + ch, acc[r] + bfull[ch * F + k]);
            }
        }
        FL_TASK_LOOP_END
    }
    PHASE(3);
    lds_barrier();
    PHASE(4);

    // ---- p3: recompute y_pre, dy_pre, dz = Wu^T dy_pre (dz overwrites s) ----
    {
        const T* wa = wl + (size_t)lane * 8;  // fragment fr of the window: wa + fr * 512
        // software pipeline: the dy pieces of the wave's NEXT row tile are requested before this tile's math (clamped addresses)
        const int tcl = t0 + (l15 % TT) < T_ ? t0 + (l15 % TT) : T_ - 1;
        auto row_of = [&](int nt) -> size_t {
            const int f = (16 / TT) * nt + (l15 / TT);
            return ((size_t)b * F + (f < F ? f : F - 1)) * T_ + tcl;
        };
        // (round 6, under the file's iterative-minreg schedule: 469 / 475 -> 460 / 466 us per launch with the prefetch in THIS pass — twelve registers;
        //  in p5, twenty-four registers and 55 instead of 28 spilled: 561 us)
        constexpr bool PF = true;
        RawC4<T> dnext[BK_MT];
        if (PF && w < ntile) rawc_load_row<T>(dnext, dy + row_of(w) * FL_H);
        for (int nt = w; nt < ntile; nt += nw) {
            const int f = (16 / TT) * nt + (l15 / TT), tt = l15 % TT;
            const bool valid = f < F && t0 + tt < T_;
            const size_t n = ((size_t)b * F + f) * T_ + t0 + tt;
            RawC4<T> dcur[BK_MT];
            if (PF) {
#pragma unroll
                for (int mt = 0; mt < BK_MT; ++mt) dcur[mt] = dnext[mt];
                rawc_load_row<T>(dnext, dy + row_of(nt + nw < ntile ? nt + nw : nt) * FL_H);
            } else {
                rawc_load_row<T>(dcur, dy + row_of(nt) * FL_H);
            }
            Frag<T> bq;
            if (g4 == 0 && f < F) {
                frag_load(bq, z + ((size_t)f * TT + tt) * FL_SQ);
                if (valid) {
                    float zv[8];
                    load8(z + ((size_t)f * TT + tt) * FL_SQ, zv);
                    store4(z_out + n * FL_SQ, zv[0], zv[1], zv[2], zv[3]);
                    store4(z_out + n * FL_SQ + 4, zv[4], zv[5], zv[6], zv[7]);
                }
            } else {
                frag_zero(bq);
            }
            f32x4 dyp[FL_MT];
#pragma unroll
            for (int mt = 0; mt < FL_MT; ++mt) {
                Frag<T> am;
                frag_load(am, wa + mt * 512);
                const f32x4 acc = mma(am, bq, F32X4_ZERO);
                const int ch = 16 * mt + 4 * g4;
                float dv[4];
                rawc_get(dcur[mt], dv);
#pragma unroll
                for (int r = 0; r < 4; ++r) dyp[mt][r] = valid ? dv[r] * dsilu_f(acc[r] + bu[ch + r]) : 0.f;
#ifndef FL_KO_OPS  // (timing knock-out, A/B flavour: no dy_pre operand)
                if (valid) store4(dyp_out + n * FL_H + ch, dyp[mt][0], dyp[mt][1], dyp[mt][2], dyp[mt][3]);
#endif
            }
            f32x4 dzt = F32X4_ZERO;
#pragma unroll
            for (int ks = 0; ks < FL_KS; ++ks) {
                Frag<T> df, atk;
                frag_from_c2(df, dyp[2 * ks], dyp[2 * ks + 1]);
                frag_load(atk, wa + (6 + ks) * 512);
                dzt = mma(atk, df, dzt);
            }
            if (valid && g4 < 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = 4 * g4 + r;
                    store1(s + ((size_t)ch * TT + tt) * FK + f, dzt[r]);
                }
            }
        }
    }
    PHASE(5);
    lds_barrier();
    PHASE(6);
    copy_sq_image(dz_out);

    // ---- p4: ds = Wf^T dz ; ds_pre = ds * SiLU'(s_pre)  (ds_pre overwrites z) ----
    {
        FL_TASK_LOOP_BEGIN(FL_SQ * mtf, WfullT)
        // the squeeze BIAS gradient is summed here from the fp32 values: as a column sum of the bf16 ds_pre operand over all ~10^6 tokens
        // (wgrad.hip) its rounding noise was the worst parameter-gradient error of the whole network (9.6e-2 on layers.0.squeeze.0.bias)
        float dbs = 0.f;
        if (l15 < TT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int h = mt * 16 + 4 * g4 + r;
                if (h < F) {
                    const float pre = load1(sp + ((size_t)h * TT + l15) * FL_SQ + ch);
                    const float v = acc[r] * dsilu_f(pre);
                    dbs += v;
                    store1(z + ((size_t)h * TT + l15) * FL_SQ + ch, v);
                }
            }
        }
        dbs = wave_sum64(dbs);
        if (lane == 0) aff[ch * 17 + mt] = dbs;
        FL_TASK_LOOP_END
    }
    PHASE(7);
    lds_barrier();
    PHASE(8);

    // ---- p5: du = Ws^T ds_pre, LayerNorm backward + residual ----
    {
        const T* wa = wl + (size_t)(9 * 512) + (size_t)lane * 8;  // the six Ws^T fragments
        // (no LayerNorm affine sums here: dgamma / dbeta follow from the squeeze weight gradient itself, D = ds_pre^T xhat —
        //  dgamma[i] = sum_o Ws[o][i] D[o][i], dbeta[i] = sum_o Ws[o][i] dbs[o] — in full_sq_finalize_kernel; 48 accumulators per lane gone)
        const int tcl = t0 + (l15 % TT) < T_ ? t0 + (l15 % TT) : T_ - 1;
        auto row_of = [&](int nt) -> size_t {
            const int f = (16 / TT) * nt + (l15 / TT);
            return ((size_t)b * F + (f < F ? f : F - 1)) * T_ + tcl;
        };
        constexpr bool PF = false;  // (see p3)
        RawC4<T> xnext[BK_MT], dnext[BK_MT];
        if (PF && w < ntile) {
            rawc_load_row<T>(xnext, x + row_of(w) * FL_H);
            rawc_load_row<T>(dnext, dy + row_of(w) * FL_H);
        }
        for (int nt = w; nt < ntile; nt += nw) {
            const int f = (16 / TT) * nt + (l15 / TT), tt = l15 % TT;
            const bool valid = f < F && t0 + tt < T_;
            const size_t n = ((size_t)b * F + f) * T_ + t0 + tt;
            RawC4<T> xcur[BK_MT], dcur[BK_MT];
            if (PF) {
#pragma unroll
                for (int mt = 0; mt < BK_MT; ++mt) { xcur[mt] = xnext[mt]; dcur[mt] = dnext[mt]; }
                const size_t nn = row_of(nt + nw < ntile ? nt + nw : nt);
                rawc_load_row<T>(xnext, x + nn * FL_H);
                rawc_load_row<T>(dnext, dy + nn * FL_H);
            } else {
                rawc_load_row<T>(xcur, x + row_of(nt) * FL_H);
                rawc_load_row<T>(dcur, dy + row_of(nt) * FL_H);
            }
            Frag<T> bq;
            if (g4 == 0 && valid) {
                frag_load(bq, z + ((size_t)f * TT + tt) * FL_SQ);
                float dv[8];
                load8(z + ((size_t)f * TT + tt) * FL_SQ, dv);
                store4(dsp_out + n * FL_SQ, dv[0], dv[1], dv[2], dv[3]);
                store4(dsp_out + n * FL_SQ + 4, dv[4], dv[5], dv[6], dv[7]);
            } else {
                frag_zero(bq);
            }
            f32x4 du[BK_MT];
#pragma unroll
            for (int mt = 0; mt < FL_MT; ++mt) {
                Frag<T> am;
                frag_load(am, wa + mt * 512);
                du[mt] = mma(am, bq, F32X4_ZERO);
            }
            ln_bwd_row96_raw_nas<T>(du, xcur, dcur, dx + n * FL_H, stats + n * 2, valid, lnw);
        }
    }
    PHASE(9);
    lds_barrier();
    PHASE(10);
    for (int i = tid; i < FL_SQ; i += nthr) {
        float v = 0.f;
        for (int mt = 0; mt < mtf; ++mt) v += aff[i * 17 + mt];
        part[(size_t)bid * FL_SQ + i] = v;
    }
    PHASE_END();
}
PHASE_READER(nbss_phase_read_full_bwd)

// tmp = D [SQ][H] | dbs [SQ] | ones [H] | zeros [H]: cleared / initialised before the backward kernel
__global__ void full_sq_prep_kernel(float* __restrict__ tmp) {
    for (int i = threadIdx.x; i < FL_SQ * FL_H + FL_SQ + 2 * FL_H; i += blockDim.x) tmp[i] = (i >= FL_SQ * FL_H + FL_SQ && i < FL_SQ * FL_H + FL_SQ + FL_H) ? 1.0f : 0.f;
}
// dWs[o][i] += D[o][i] gamma[i] + dbs[o] beta[i];  dbs[o] += dbs;  dgamma[i] += sum_o Ws[o][i] D[o][i];  dbeta[i] += sum_o Ws[o][i] dbs[o]
// (du = Ws^T ds_pre contracted with xhat resp. 1 over all tokens, reordered: the LayerNorm affine gradients cost no per-token work)
static_assert(FL_H == FK_H && FL_SQ == FK_SQ, "foldk.h");
__global__ void full_sq_finalize_kernel(const float* __restrict__ tmp, const float* __restrict__ Ws, const float* __restrict__ gamma, const float* __restrict__ beta,
                                        float* __restrict__ dWs, float* __restrict__ dbs, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    fk_full_sq_final(tmp, Ws, gamma, beta, dWs, dbs, dgamma, dbeta);  // (the body lives in foldk.h: fold.hip's table kernel runs it too)
}

// frames per slab: the largest of 8 / 4 / 2 that still gives (nearly) every CU a workgroup.  The LinearGroup passes cost the same per slab whatever its
// width (their MFMA N dimension is 16 frames wide either way), the row passes scale with it: at batch 2 (64 slabs of 8 frames on 256 CUs) the
// backward kernel took 158 us per launch, 4.8 x its share of the batch-32 launch.  "Nearly": batch 4 x 251 frames is 252 slabs of 4 frames — 468 -> 490
// utt/s against 504 slabs of 2 (round 5)
static int full_tt(const nbss_cfg& c) {
    static const int forced = [] { const char* e = getenv("NBSS_FULL_TT"); return e ? atoi(e) : 0; }();  // tuning / test knob: 8, 4 or 2
    if (forced == 8 || forced == 4 || forced == 2) return forced;
    for (int tt = FL_TT_MAX; tt > 2; tt >>= 1)
        if (c.B * cdiv(c.T, tt) >= 240) return tt;
    return 2;
}

// LDS of the backward kernel at slab width tt (PHASE_LDS_BYTES: the phase-timer build's counters)
static size_t full_bwd_lds(const nbss_cfg& c, int tt) {
    const size_t esz = c.dtype == NBSS_BF16 ? 2 : 4;
    const int mtf = cdiv(c.F, 16), ksf = cdiv(c.F, 32);
    return ((size_t)FL_SQ * tt * ksf * 32 + (size_t)2 * mtf * 16 * tt * FL_SQ + (size_t)FL_WFR * 512) * esz + FL_SQ * 17 * sizeof(float) + PHASE_LDS_BYTES;
}
// ... and the width the backward pass runs at: full_tt(), narrowed until the squeezed images fit (fp32 stream at F = 257: 213 KB at 8 frames, 136 KB at 4)
static int full_bwd_width(const nbss_cfg& c) {
    int tt = full_tt(c);
    while (tt > 2 && full_bwd_lds(c, tt) > 160 * 1024) tt >>= 1;
    return tt;
}

template <class T, int KSFM, int TT>
static int full_bwd_tt(const nbss_cfg& c, const float* P, float* part, const void* packed, int layer, const void* x, const void* dy, void* dx,
                       float* stats, void* const* o, hipStream_t st) {
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    const int mtf = cdiv(c.F, 16), ksf = cdiv(c.F, 32);
    const size_t lds = full_bwd_lds(c, TT);
    const T* pk = (const T*)packed;
    if (ksf > KSFM || lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    int e = NBSS_SET_MAX_LDS((full_bwd_kernel<T, KSFM, TT>), lds);
    if (e) return e;
    dim3 grid(c.B * cdiv(c.T, TT)), block(FL_THREADS);
    ProfScope ps(PK_FULL_B, st);
    NBSS_LAUNCH((full_bwd_kernel<T, KSFM, TT>), grid, block, lds, st, c, lp, P, part, layer, pk + pack_off(c, layer, K_SQ), pk + pack_off(c, layer, K_FULL),
                pk + pack_off(c, layer, K_USQ), pk + pack_off(c, layer, K_SQ_T), pk + pack_off(c, layer, K_FULL_T), pk + pack_off(c, layer, K_USQ_T),
                (const T*)x, (const T*)dy, (T*)dx, stats, (T*)o[0], (T*)o[1], (T*)o[2], (T*)o[3], (T*)o[4], walk_flip_next());
    return NBSS_CHECK_LAUNCH();
}

template <class T, int KSFM>
static int full_bwd_t(const nbss_cfg& c, const float* P, float* part, const void* packed, int layer, const void* x, const void* dy, void* dx,
                      float* stats, void* const* o, hipStream_t st) {
    switch (full_bwd_width(c)) {
        case 8: return full_bwd_tt<T, KSFM, 8>(c, P, part, packed, layer, x, dy, dx, stats, o, st);
        case 4: return full_bwd_tt<T, KSFM, 4>(c, P, part, packed, layer, x, dy, dx, stats, o, st);
        default: return full_bwd_tt<T, KSFM, 2>(c, P, part, packed, layer, x, dy, dx, stats, o, st);
    }
}

int memset_async_impl(void* p, size_t bytes, hipStream_t st);

int full_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, int layer, const void* x, const void* dy, void* dx, void* ws,
                  hipStream_t st, const Side* sd) {
    if (c.H != FL_H) return gb_full_bwd(c, P, G, layer, x, dy, dx, ws, st, sd);
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    // workspace: stats | s [B*T][SQ][FKP] | dz [B*T][SQ][FKP] | z [N][SQ] | dy_pre [N][H] | ds_pre [N][SQ]
    const size_t N = (size_t)c.B * c.F * c.T, BT = (size_t)c.B * c.T, esz = c.dtype == NBSS_BF16 ? 2 : 4;
    const int FKP = FL_FKP(c.F);
    float* stats = (float*)ws;
    char* p = (char*)ws + ws_align(N * 2 * sizeof(float));
    void* o[5];
    o[0] = p; p += ws_align(BT * FL_SQ * FKP * esz);
    o[1] = p; p += ws_align(BT * FL_SQ * FKP * esz);
    o[2] = p; p += ws_align(N * FL_SQ * esz);
    o[3] = p; p += ws_align(N * FL_H * esz);
    o[4] = p;
    // (the F..FKP padding columns of s / dz, read by the wgrad staging, are written as zeros by the kernel's row copies)
    int e;
    float* part = (float*)((char*)ws + ws_part_offset(c));
    // tmp (behind the per-workgroup partial rows): D [SQ][H] = ds_pre^T xhat | dbs [SQ] | ones [H] | zeros [H]
    const int nslab = c.B * cdiv(c.T, full_bwd_width(c));
    float* sqtmp = part + (size_t)nslab * FL_SQ + 64;
    if (c.F > 32 * FL_KSF_MAX)
        e = c.dtype == NBSS_BF16 ? full_bwd_t<bf16_t, FL_KSF_BIG>(c, P, part, packed, layer, x, dy, dx, stats, o, st)
                                 : full_bwd_t<float, FL_KSF_BIG>(c, P, part, packed, layer, x, dy, dx, stats, o, st);
    else
        e = c.dtype == NBSS_BF16 ? full_bwd_t<bf16_t, FL_KSF_MAX>(c, P, part, packed, layer, x, dy, dx, stats, o, st)
                                 : full_bwd_t<float, FL_KSF_MAX>(c, P, part, packed, layer, x, dy, dx, stats, o, st);
    if (e) return e;
    // everything below only produces parameter gradients: gradient stream (side.h)
    const hipStream_t gs = side_fork(sd, st);
    FoldScope fs(gs, (char*)ws + ws_wgpart_offset(c), WGPART_BYTES, N);  // (fold.h: the sub-block's seven fold launches leave as three, one per stage)
    NBSS_FOLD_LAUNCH(full_sq_prep_kernel, dim3(1), dim3(256), 0, gs, sqtmp);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    // squeeze bias gradient (fp32 sums of the kernel) -> tmp.dbs
    AffSegs sg;
    sg.n = 1;
    sg.off[0] = FL_SQ * FL_H; sg.cnt[0] = FL_SQ;
    if ((e = affine_reduce_launch(part, nslab, sg, sqtmp, gs))) return e;
    WgradArgs a;
    a.part = (float*)((char*)ws + ws_wgpart_offset(c));
    a.mvalid = 0; a.nvalid = 0;
    a.F = c.F; a.T = c.T; a.shift_stride = 1; a.shift_dim = 0; a.taps = 1;
    a.stats = nullptr; a.gamma = nullptr; a.beta = nullptr;
    // unsqueeze: dWu[H][SQ] = dy_pre^T z ; dbu = colsum(dy_pre)
    a.Ntok = (int)N; a.groups = 1;
    a.A = o[3]; a.lda = FL_H; a.MA = FL_H; a.B = o[2]; a.ldb = FL_SQ; a.NB = FL_SQ;
    a.dW = G + param_off(c, layer, P_USQ_W); a.dbias = G + param_off(c, layer, P_USQ_B);
    if ((e = wgrad_launch(a, c.dtype, gs))) return e;
    // LinearGroup: dWf[c][k][h] = sum_{b,t} dz[b,t,c,k] s[b,t,c,h] ; dbf = colsum(dz)   (rows = (b,t))
    a.Ntok = (int)BT; a.groups = FL_SQ; a.mvalid = c.F; a.nvalid = c.F;
    a.A = o[1]; a.lda = FL_SQ * FKP; a.MA = FL_SQ * FKP; a.B = o[0]; a.ldb = FL_SQ * FKP; a.NB = FL_SQ * FKP;
    a.dW = G + param_off(c, layer, P_FULL_W); a.dbias = G + param_off(c, layer, P_FULL_B);
    if ((e = wgrad_launch(a, c.dtype, gs))) return e;
    // squeeze: dWs[SQ][H] = ds_pre^T LN(x) ; dbs = colsum(ds_pre)
    a.Ntok = (int)N; a.groups = 1; a.mvalid = 0; a.nvalid = 0;
    a.A = o[4]; a.lda = FL_SQ; a.MA = FL_SQ; a.B = x; a.ldb = FL_H; a.NB = FL_H;
    a.stats = stats; a.gamma = lp.p[P_FULL_LN_W]; a.beta = lp.p[P_FULL_LN_B];
    // squeeze: D = ds_pre^T xhat (LayerNorm on the fly with gamma = 1, beta = 0) into tmp; the finalize kernel turns it into
    // dWs = D gamma + dbs (x) beta, dbs, and the LayerNorm affine gradients
    a.gamma = sqtmp + FL_SQ * FL_H + FL_SQ; a.beta = sqtmp + FL_SQ * FL_H + FL_SQ + FL_H;
    a.dW = sqtmp; a.dbias = nullptr;
    if ((e = wgrad_launch(a, c.dtype, gs))) return e;
    if (g_fold) {  // third stage: reads the squeeze problem's second pass (stage 1) and the bias fold (stage 2)
        FoldItem it;
        it.kind = FK_FULL_SQ;
        it.gx = 1; it.gy = 1; it.nblk = 1;
        it.u.sq.tmp = sqtmp; it.u.sq.Ws = lp.p[P_SQ_W]; it.u.sq.gamma = lp.p[P_FULL_LN_W]; it.u.sq.beta = lp.p[P_FULL_LN_B];
        it.u.sq.dWs = G + param_off(c, layer, P_SQ_W); it.u.sq.dbs = G + param_off(c, layer, P_SQ_B);
        it.u.sq.dgamma = G + param_off(c, layer, P_FULL_LN_W); it.u.sq.dbeta = G + param_off(c, layer, P_FULL_LN_B);
        if ((e = g_fold->add(3, it))) return e;
        return fs.end();
    }
    NBSS_FOLD_LAUNCH(full_sq_finalize_kernel, dim3(1), dim3(FL_H), 0, gs, (const float*)sqtmp, lp.p[P_SQ_W], lp.p[P_FULL_LN_W], lp.p[P_FULL_LN_B],
                G + param_off(c, layer, P_SQ_W), G + param_off(c, layer, P_SQ_B), G + param_off(c, layer, P_FULL_LN_W), G + param_off(c, layer, P_FULL_LN_B));
    return NBSS_CHECK_LAUNCH();
}

template <class T, int KSFM, class G, int TT>
static int full_fwd_tt(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st) {
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    const int mtf = cdiv(c.F, 16), ksf = cdiv(c.F, 32);
    const size_t lds = ((size_t)G::SQ * TT * ksf * 32 + (size_t)mtf * 16 * TT * G::SQ) * sizeof(T);
    const T* pk = (const T*)packed;
    if (ksf > KSFM || lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    int e = NBSS_SET_MAX_LDS((full_fwd_kernel<T, KSFM, G::H, G::SQ, TT>), lds);
    if (e) return e;
    dim3 grid(c.B * cdiv(c.T, TT)), block(FL_THREADS);
    ProfScope ps(PK_FULL_F, st);
    NBSS_LAUNCH((full_fwd_kernel<T, KSFM, G::H, G::SQ, TT>), grid, block, lds, st, c, lp.p[P_FULL_LN_W], lp.p[P_FULL_LN_B],
                lp.p[P_SQ_B], lp.p[P_FULL_B], lp.p[P_USQ_B],
                pk + pack_off(c, layer, K_SQ), pk + pack_off(c, layer, K_FULL), pk + pack_off(c, layer, K_USQ), (const T*)x, (T*)y, walk_flip_next());
    return NBSS_CHECK_LAUNCH();
}

template <class T, int KSFM, class G>
static int full_fwd_t(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st) {
    switch (full_tt(c)) {
        case 8: return full_fwd_tt<T, KSFM, G, 8>(c, P, packed, layer, x, y, st);
        case 4: return full_fwd_tt<T, KSFM, G, 4>(c, P, packed, layer, x, y, st);
        default: return full_fwd_tt<T, KSFM, G, 2>(c, P, packed, layer, x, y, st);
    }
}

template <class G>
static int full_fwd_g(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st) {
    if (c.F > 32 * FL_KSF_MAX)
        return c.dtype == NBSS_BF16 ? full_fwd_t<bf16_t, FL_KSF_BIG, G>(c, P, packed, layer, x, y, st) : full_fwd_t<float, FL_KSF_BIG, G>(c, P, packed, layer, x, y, st);
    return c.dtype == NBSS_BF16 ? full_fwd_t<bf16_t, FL_KSF_MAX, G>(c, P, packed, layer, x, y, st) : full_fwd_t<float, FL_KSF_MAX, G>(c, P, packed, layer, x, y, st);
}

int full_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, hipStream_t st) {
    return c.H == GeoL::H ? full_fwd_g<GeoL>(c, P, packed, layer, x, y, st) : full_fwd_g<GeoS>(c, P, packed, layer, x, y, st);
}
