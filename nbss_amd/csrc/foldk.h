// foldk.h — device bodies of the folds of the partial parameter gradients (round 6).  Every fold kernel of the backward pass (wgrad.hip, util.hip,
// tailw.hip, tconvffn_s.hip, fconv.hip, full.hip) is one of these bodies behind a thin __global__ wrapper in its own file; fold.hip runs SEVERAL of them
// in one launch from a descriptor table (fold.h: FoldBatch) — a sub-block's ~5 fold launches become one launch per dependency stage.  A body takes its
// block index as arguments instead of reading blockIdx, nothing else differs: the order of every sum is the one the single-kernel launches had, the
// parameter gradients stay bitwise what they were.  All bodies run in 256-thread blocks (fewer threads needed: the rest return).
#pragma once
#include "launch.h"
#include "layout.h"
#include "common.h"
#include "blocks.h"
#include "wgrad.h"

// (constants of the owning files, restated: each owner static_asserts its own against these)
#define FK_H 96          // dim_hidden of the small geometry (tailw.hip TW_H, tconvffn_s.hip TS_H, fconv.hip FC_H, full.hip FL_H)
#define FK_FFN 192       // tconvffn_s.hip TS_FFN
#define FK_TCG 24        // T-conv channels per group (TS_CG)
#define FK_FCG 12        // F-conv channels per group (FC_CG)
#define FK_FG 8          // F-conv groups (FC_G)
#define FK_SQ 8          // dim_squeeze (FL_SQ)
#define FK_RSL 4         // slices of a (tile, register) fold block (WG_RSL, TW_RSL)
#define FK_AFF_SLICES 64 // slices of the affine fold (AFF_SLICES)
#define FK_FC_P16 (5 * FK_H * FK_FCG)
#define FK_TCONVW (FK_FFN * FK_TCG * 3)

// ---- wgrad.hip: second pass of the two-stage flush (see wgrad_reduce_kernel) ----------------------------------------------------------------------
// bx in [0, 4 ntot), bz in [0, gz): the grid of the single-kernel launch; red: [4][64] + [4][16] floats of LDS
NBSS_DEV void fk_wgrad_reduce(const WgradArgs& a, int xb, int nt_major, int bx, int bz, int gz, float* red) {
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6, r = bx & 3, l15 = lane & 15, g4 = lane >> 4;
    const int nsl = xb < FK_RSL ? xb : FK_RSL;  // slices that have x-blocks
    const int mg = a.MA / a.groups, ng = a.NB / a.groups;
    const int mv = a.mvalid ? a.mvalid : mg, nv = a.nvalid ? a.nvalid : ng;
    const int mtiles = cdiv(mg, 16), nexp = a.taps * ng, ntiles = cdiv(nexp, 16);
    const int tpg = mtiles * ntiles;
    const bool per_group = gz > 1;
    const int ntot = (per_group ? 1 : a.groups) * tpg;
    const int tl = bx >> 2, y = bz;
    const int x0 = sl < nsl ? (int)((long)xb * sl / nsl) : 0, x1 = sl < nsl ? (int)((long)xb * (sl + 1) / nsl) : 0;
    const float* pt = a.part + ((size_t)y * xb * ntot + tl) * 256 + r * 64 + lane;
    const size_t xs = (size_t)ntot * 256;
    red[sl * 64 + lane] = fold_strided<16>(pt, xs, x0, x1);
    int g, mt, nt;
    if (nt_major) {  // wgrad_tr3_kernel: tl = nt * nfirst + g * mtiles + mt
        const int nfirst = ntot / ntiles, gm = tl % nfirst;
        nt = tl / nfirst; g = gm / mtiles; mt = gm % mtiles;
    } else {
        const int rem = tl % tpg;
        g = tl / tpg; mt = rem / ntiles; nt = rem % ntiles;
    }
    if (per_group) g += y;
    const bool bias = a.dbias && nt == 0 && r == 0;  // the tile's 16 bias sums: the r = 0 block
    if (bias && lane < 16) {
        const float* pbias = a.part + (size_t)gz * xb * ntot * 256 + ((size_t)y * xb * ntot + tl) * 16 + lane;
        red[FK_RSL * 64 + sl * 16 + lane] = fold_strided<16>(pbias, (size_t)ntot * 16, x0, x1);
    }
    __syncthreads();
    if (sl) return;
    const float sum = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
    const int q = nt * 16 + l15;
    if (q < nexp) {
        const int tap = q / ng, i = q % ng, m = mt * 16 + 4 * g4 + r;
        if (m < mv && i < nv) a.dW[((size_t)(g * mv + m) * nv + i) * a.taps + tap] += sum;
    }
    if (bias && lane < 16) {
        const float b = (red[FK_RSL * 64 + lane] + red[FK_RSL * 64 + 16 + lane]) + (red[FK_RSL * 64 + 32 + lane] + red[FK_RSL * 64 + 48 + lane]);
        const int m = mt * 16 + lane;
        if (m < mv) a.dbias[(size_t)g * mv + m] += b;
    }
}

// ---- util.hip: the affine fold (see affine_slices_kernel / affine_final_kernel) --------------------------------------------------------------------
// first row of slice y: nwg y / nsl without a division (nsl is FK_AFF_SLICES, or nwg itself when there are fewer rows than slices)
NBSS_DEV int fk_aff_row0(int nwg, int nsl, int y) { return nsl == FK_AFF_SLICES ? (int)(((unsigned)nwg * (unsigned)y) >> 6) : y; }
// e = element of the partial row; by in [0, gy = slices)
NBSS_DEV void fk_affine_slices(float* __restrict__ part, int nwg, int naff, int e, int by, int gy) {
    if (e >= naff) return;
    const int w0 = fk_aff_row0(nwg, gy, by), w1 = fk_aff_row0(nwg, gy, by + 1);
    const float s = fold_strided<16>(part + e, (size_t)naff, w0, w1);
    part[(size_t)w0 * naff + e] = s;
}
NBSS_DEV void fk_affine_final(const float* __restrict__ part, int nwg, int naff, int nsl, const AffSegs& segs, float* __restrict__ G, int e) {
    if (e >= naff) return;
    float s16[16], v16[16];  // sixteen loads in flight; the order of the adds is fixed
#pragma unroll
    for (int k = 0; k < 16; ++k) s16[k] = 0.f;
    for (int y = 0; y < nsl; y += 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v16[k] = y + k < nsl ? part[(size_t)fk_aff_row0(nwg, nsl, y + k) * naff + e] : 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s16[k] += v16[k];
    }
#pragma unroll
    for (int h = 8; h >= 1; h >>= 1) {
#pragma unroll
        for (int k = 0; k < h; ++k) s16[k] += s16[k + h];
    }
    const float s = s16[0];
    int r = e;
    for (int i = 0; i < segs.n; ++i) {
        if (r < segs.cnt[i]) {
            G[segs.off[i] + r] += s;
            return;
        }
        r -= segs.cnt[i];
    }
}

// ---- tailw.hip: second pass of the tail kernel's weight gradient (see tailw_finalize_kernel / tailw_affine_kernel) ----------------------------------
// bx in [0, 4 ntot); red: [2][4][64] floats of LDS
NBSS_DEV void fk_tailw_finalize(float* __restrict__ part, int xb, int MTA, int ntot, const float* __restrict__ W, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ dW, float* __restrict__ dbias, int bx, float* red) {
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6, r = bx & 3, l15 = lane & 15, g4 = lane >> 4;
    const int nsl = xb < FK_RSL ? xb : FK_RSL;
    const int tl = bx >> 2, nt = tl / MTA, mt = tl % MTA;
    const int x0 = sl < nsl ? (int)((long)xb * sl / nsl) : 0, x1 = sl < nsl ? (int)((long)xb * (sl + 1) / nsl) : 0;
    const float* pt = part + (size_t)tl * 256 + r * 64 + lane;
    const float* pb = part + (size_t)xb * ntot * 256 + (size_t)mt * 16 + 4 * g4 + r;  // bias sums of tile (nt = 0, mt): rows 4 g4 + r
    const size_t xs = (size_t)ntot * 256, bs = (size_t)ntot * 16;
    red[sl * 64 + lane] = fold_strided<16>(pt, xs, x0, x1);
    red[(FK_RSL + sl) * 64 + lane] = fold_strided<16>(pb, bs, x0, x1);
    __syncthreads();  // (also: every slice has read workgroup 0's values of this block, whose slot receives the column sums below)
    if (sl) return;
    const float D = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
    const float bsum = (red[256 + lane] + red[320 + lane]) + (red[384 + lane] + red[448 + lane]);
    const int o = mt * 16 + 4 * g4 + r, i = nt * 16 + l15;
    const float w = W[(size_t)o * FK_H + i];
    dW[(size_t)o * FK_H + i] += D * gamma[i] + bsum * beta[i];
    if (nt == 0 && l15 == 0) dbias[o] += bsum;
    // column sums over the block's four rows (the lane groups)
    float tg = w * D, tb = w * bsum;
    tg += __shfl_xor(tg, 16); tg += __shfl_xor(tg, 32);
    tb += __shfl_xor(tb, 16); tb += __shfl_xor(tb, 32);
    if (g4 == 0) {
        part[(size_t)tl * 256 + r * 64 + l15] = tg;
        part[(size_t)tl * 256 + r * 64 + 16 + l15] = tb;
    }
}
// one block; threads [0, 2 H)
NBSS_DEV void fk_tailw_affine(const float* __restrict__ part, int MTA, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    if (threadIdx.x >= 2 * FK_H) return;
    const int k = threadIdx.x / FK_H, i = threadIdx.x % FK_H, nt = i >> 4, j = i & 15;
    float s = 0.f;
    for (int mt = 0; mt < MTA; ++mt) {
        const float* p = part + (size_t)(nt * MTA + mt) * 256 + 16 * k + j;
        s += (p[0] + p[64]) + (p[128] + p[192]);
    }
    (k ? dbeta : dgamma)[i] += s;
}

// ---- tconvffn_s.hip / fconv.hip: folds of the bf16 partial rows (see tconv_part_reduce1_kernel, tconv_part_reduce2_kernel, fconv_part_final_kernel) ---
// bx: block of 256 groups of 8 elements; by in [0, gy = slices)
NBSS_DEV void fk_p16_slices(const bf16_t* __restrict__ part16, int nrows, float* __restrict__ slices, int P16, int bx, int by, int gy) {
    const int e8 = bx * 256 + threadIdx.x;  // group of 8 elements
    if (e8 >= P16 / 8) return;
    const int r0 = (int)((long)nrows * by / gy), r1 = (int)((long)nrows * (by + 1) / gy);
    const u32x4* p = reinterpret_cast<const u32x4*>(part16) + e8;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    auto add = [&](const u32x4& u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[2 * k] += __builtin_bit_cast(float, u[k] << 16);
            acc[2 * k + 1] += __builtin_bit_cast(float, u[k] & 0xFFFF0000u);
        }
    };
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
        const u32x4 a = p[(size_t)r * (P16 / 8)], b = p[(size_t)(r + 1) * (P16 / 8)], c = p[(size_t)(r + 2) * (P16 / 8)], d = p[(size_t)(r + 3) * (P16 / 8)];
        add(a); add(b); add(c); add(d);
    }
    for (; r < r1; ++r) add(p[(size_t)r * (P16 / 8)]);
    float* out = slices + (size_t)by * P16 + (size_t)e8 * 8;
    store4(out, acc[0], acc[1], acc[2], acc[3]);
    store4(out + 4, acc[4], acc[5], acc[6], acc[7]);
}
NBSS_DEV void fk_tconv_final(const float* __restrict__ slices, int nsl, float* __restrict__ G, long long off0, long long off1, long long off2, long long off3,
                             int P16, int bx) {
    const int e = bx * 256 + threadIdx.x;
    if (e >= P16) return;
    const float sum = fold_strided<16>(slices + e, (size_t)P16, 0, nsl);
    if (e >= 3 * FK_TCONVW) {  // dW2 partial: [FFN channel][H output] -> the parameter's [H][FFN]
        const int q = e - 3 * FK_TCONVW, ch = q / FK_H, o = q - ch * FK_H;
        G[off3 + (size_t)o * FK_FFN + ch] += sum;
        return;
    }
    const int k = e / FK_TCONVW, q = e - k * FK_TCONVW, grp = q / (3 * FK_TCG * FK_TCG), tap = (q / (FK_TCG * FK_TCG)) % 3, i = (q / FK_TCG) % FK_TCG, o = grp * FK_TCG + q % FK_TCG;
    float* g = G + (k == 0 ? off0 : k == 1 ? off1 : off2);
    g[((size_t)o * FK_TCG + i) * 3 + tap] += sum;  // (stream order: nothing else writes these gradients between the two launches)
}
NBSS_DEV void fk_fconv_final(const float* __restrict__ slices, int nsl, float* __restrict__ dW, int bx) {
    const int e = bx * 256 + threadIdx.x;
    if (e >= FK_FC_P16) return;
    const float sum = fold_strided<16>(slices + e, (size_t)FK_FC_P16, 0, nsl);
    const int ol = e % FK_FCG, i = (e / FK_FCG) % FK_FCG, g = (e / (FK_FCG * FK_FCG)) % FK_FG, tap = e / (FK_FCG * FK_FCG * FK_FG);
    dW[((size_t)(g * FK_FCG + ol) * FK_FCG + i) * 5 + tap] += sum;
}

// ---- full.hip: the squeeze-side finalize (see full_sq_finalize_kernel); one block, threads [0, H) ----------------------------------------------------
NBSS_DEV void fk_full_sq_final(const float* __restrict__ tmp, const float* __restrict__ Ws, const float* __restrict__ gamma, const float* __restrict__ beta,
                               float* __restrict__ dWs, float* __restrict__ dbs, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int i = threadIdx.x;
    if (i >= FK_H) return;
    float dg = 0.f, db = 0.f;
#pragma unroll
    for (int o = 0; o < FK_SQ; ++o) {
        const float D = tmp[o * FK_H + i], b = tmp[FK_SQ * FK_H + o], w = Ws[o * FK_H + i];
        dWs[o * FK_H + i] += D * gamma[i] + b * beta[i];
        dg += w * D;
        db += w * b;
    }
    dgamma[i] += dg;
    dbeta[i] += db;
    if (i < FK_SQ) dbs[i] += tmp[FK_SQ * FK_H + i];
}
