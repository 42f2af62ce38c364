// tailw.hip — the TAIL of a narrow-band block's backward pass fused with the weight gradient of its first linear map (bf16 stream).
//
// Both the T-ConvFFN (W1: H -> FFN) and the attention (in_proj: H -> 3H) start with  a = W LN(x) + b  and their data-gradient kernels end
// with the same three steps on the emitted pre-activation gradient `da` ([MA/24][N][24] group-major operand):
//     du = W^T da            LayerNorm backward + residual:  dx = dy + LN'(du)            dW = da^T LN(x),  db = colsum(da)
// Done as separate kernels (a tail kernel and a wgrad problem) `da` and `x` are read from HBM twice: 3 S·B of the 8 S·B the pair moves.
// Here one persistent kernel (256 workgroups x 8 waves, 64-token chunks, wgrad.hip's staging scheme: register prefetch of the next chunk,
// ROW-major LDS images, transposing reads for the token-contraction) reads them once:
//   * all 8 waves contract the chunk's dW tiles (A = da image, B = LN(x) image, K = the 64 tokens),
//   * waves 0-3 then take one 16-token tile each: du = W^T da with the B operand straight from the da image rows (natural K order) and the
//     W^T fragments resident in LDS, LayerNorm backward in registers (x, dy as 8-byte C-layout pieces, requested before the MFMA section),
//     dx stored, LayerNorm affine gradients accumulated per lane and flushed once per workgroup.
// Partial dW tiles go to wgrad_reduce_kernel (wgrad.hip), the affine partial rows to affine_reduce.
#include "launch.h"
#include "layout.h"
#include "prof.h"
#include "blocks.h"
#include "wgrad.h"

#define TW_KC 64
#define TW_H 96
#define TW_THREADS 512

struct TailArgs {
    const bf16_t* A;      // [MA/24][Ntok][24]  pre-activation gradient (group-major)
    const bf16_t* x;      // [Ntok][96]         block input
    const bf16_t* dy;     // [Ntok][96]         upstream gradient (residual path)
    bf16_t* dx;           // [Ntok][96]
    const float* stats;   // [Ntok][2]          LayerNorm (mean, rstd) of x, written by the data-gradient kernel
    const float* gamma;   // [96] LayerNorm weight / bias
    const float* beta;
    const bf16_t* WT;     // packed W^T fragments [6][MA/32][64][8] (K_TF_W1_TN / K_INP_TN: rows = H, K = MA natural)
    float* part;          // [grid][ntot][256] partial dW tiles | [grid][ntot][16] bias sums (WGPART region)
    float* affpart;       // [grid][2 * 96] LayerNorm affine partial sums
    int Ntok;
};

// Wave specialisation: waves 4-7 hold ALL dW accumulators (NTOT / 4 tiles each), waves 0-3 run the tail.  The two roles are two
// separate chunk loops (same sequence of barriers) so that their register sets — 27 accumulator tiles for in_proj on one side, du / x / dy /
// affine sums on the other — have disjoint live ranges; as one loop with `if (w < 4)` inside, both sets stayed live and 65-146 VGPRs spilled.
template <int MA, int NBUF>
__global__ __launch_bounds__(TW_THREADS, 2) void tailw_kernel(TailArgs a) {
    constexpr int LDA = MA + 16;                    // image row strides == 16 (mod 32) elements: wgrad.hip tr_ld()
    constexpr int LDX = 112;                        // 96 + 16
    constexpr int IMGA = TW_KC * LDA, IMG = IMGA + TW_KC * LDX;
    constexpr int MTA = MA / 16, NTB = TW_H / 16, NTOT = MTA * NTB;  // tile tl = nt * MTA + mt
    constexpr int NSW = (NTOT + 3) / 4;              // tile slots of a wgrad wave
    constexpr int BSW = (MTA + 3) / 4;               // slots that can hold nt == 0 tiles (bias sums)
    constexpr int KSW = MA / 32;                     // k-steps of du = W^T da
    constexpr int PA = MA / 8, NVA = TW_KC * PA, NVX = TW_KC * (TW_H / 8);
    constexpr int UA = (NVA + TW_THREADS - 1) / TW_THREADS, UX = (NVX + TW_THREADS - 1) / TW_THREADS;  // whole vector slots: da | x
    static_assert(LDA % 32 == 16, "image stride");
    NBSS_LDS(smem);
    bf16_t* base = reinterpret_cast<bf16_t*>(smem);
    bf16_t* wl = base + (size_t)NBUF * IMG;                       // W^T fragments
    float* lnp = reinterpret_cast<float*>(wl + 6 * KSW * 512);    // gamma | beta
    float* affl = lnp + 2 * TW_H;                                 // LN weight | bias gradient sums of this workgroup
    const int tid = threadIdx.x, lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id_u();
    for (int i = tid; i < NBUF * IMG / 2; i += TW_THREADS) reinterpret_cast<uint32_t*>(base)[i] = 0u;
    for (int i = tid; i < 6 * KSW * 64; i += TW_THREADS) reinterpret_cast<u32x4*>(wl)[i] = reinterpret_cast<const u32x4*>(a.WT)[i];
    for (int i = tid; i < 4 * TW_H; i += TW_THREADS) lnp[i] = i < TW_H ? a.gamma[i] : i < 2 * TW_H ? a.beta[i - TW_H] : 0.f;

    const int nchunks = cdiv(a.Ntok, TW_KC);
    // staging: slot u < UA is a da piece for every thread, the rest x pieces (a few lanes of the last slot of each kind idle)
    u32x4 preA[UA], preX[UX];
    float pmu[UX], prs[UX];
    auto prefetch = [&](int ch) {
        const long n0 = (long)ch * TW_KC;
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int v = tid + u * TW_THREADS, r = v / PA, col = (v % PA) * 8;
            preA[u] = (u32x4){0, 0, 0, 0};
            if (v < NVA && n0 + r < a.Ntok) preA[u] = *reinterpret_cast<const u32x4*>(a.A + ((size_t)(col / 24) * a.Ntok + n0 + r) * 24 + col % 24);
        }
#pragma unroll
        for (int u = 0; u < UX; ++u) {
            const int v = tid + u * TW_THREADS, r = v / (TW_H / 8), col = (v % (TW_H / 8)) * 8;
            preX[u] = (u32x4){0, 0, 0, 0};
            pmu[u] = 0.f; prs[u] = 0.f;
            if (v < NVX && n0 + r < a.Ntok) {
                preX[u] = *reinterpret_cast<const u32x4*>(a.x + (size_t)(n0 + r) * TW_H + col);
                pmu[u] = a.stats[2 * (n0 + r)];
                prs[u] = a.stats[2 * (n0 + r) + 1];
            }
        }
    };
    auto stash = [&](bf16_t* buf) {
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int v = tid + u * TW_THREADS, r = v / PA, col = (v % PA) * 8;
            if (v < NVA) *reinterpret_cast<u32x4*>(buf + r * LDA + col) = preA[u];
        }
#pragma unroll
        for (int u = 0; u < UX; ++u) {
            const int v = tid + u * TW_THREADS, r = v / (TW_H / 8), col = (v % (TW_H / 8)) * 8;
            if (v < NVX) {  // LayerNorm on the fly (rows past Ntok: rstd = 0, they stay 0)
                u32x4 xq = preX[u];
                float f[8], gm[8], bt[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { f[2 * i] = bf2f((bf16_t)(xq[i] & 0xFFFF)); f[2 * i + 1] = bf2f((bf16_t)(xq[i] >> 16)); }
                load8(lnp + col, gm);
                load8(lnp + TW_H + col, bt);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = prs[u] != 0.f ? (f[e] - pmu[u]) * prs[u] * gm[e] + bt[e] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) xq[i] = pack2bf(f[2 * i], f[2 * i + 1]);
                *reinterpret_cast<u32x4*>(buf + IMGA + r * LDX + col) = xq;
            }
        }
    };

    int ch = blockIdx.x;
    if (ch < nchunks) prefetch(ch);
    lds_barrier();  // zero fill, fragments, gamma / beta
    int b = 0;
    if (w >= 4) {
        // ================= weight-gradient waves: tile tl = 4 s + (w - 4) =================
        const int w4 = w - 4;
        f32x4 acc[NSW], bacc[BSW];
#pragma unroll
        for (int s = 0; s < NSW; ++s) acc[s] = F32X4_ZERO;
#pragma unroll
        for (int s = 0; s < BSW; ++s) bacc[s] = F32X4_ZERO;
        Frag<bf16_t> ones;
#pragma unroll
        for (int jq = 0; jq < 8; ++jq) frag_set(ones, jq, 1.0f);
        const int toff = (4 * g4 + (l15 >> 2)), tcol = 4 * (l15 & 3);
        const int la = toff * LDA + tcol, lb = IMGA + toff * LDX + tcol;
        auto off_a = [&](int s) { const int tl = 4 * s + w4; return la + (tl < NTOT ? tl % MTA : 0) * 16; };  // slots past the last tile
        auto off_b = [&](int s) { const int tl = 4 * s + w4; return lb + (tl < NTOT ? tl / MTA : 0) * 16; };  // re-contract tile 0: never flushed
        for (; ch < nchunks; ch += gridDim.x) {
            bf16_t* buf = base + (size_t)b * IMG;
            stash(buf);
            lds_barrier();
            if (ch + (int)gridDim.x < nchunks) prefetch(ch + gridDim.x);
            // software pipeline over the NSW x 2 (tile, k-half) steps: the operands of step i+1 are requested before step i's MFMA
            Frag<bf16_t> fa[2], fb[2];
            frag_load_tr(fa[0], buf + off_a(0), LDA);
            frag_load_tr(fb[0], buf + off_b(0), LDX);
#pragma unroll
            for (int i = 0; i < NSW * 2; ++i) {
                const int s = i / 2, cur = i & 1;
                if (i + 1 < NSW * 2) {
                    const int s1 = (i + 1) / 2, kh1 = (i + 1) % 2;
                    frag_load_tr(fa[cur ^ 1], buf + off_a(s1) + kh1 * 32 * LDA, LDA);
                    frag_load_tr(fb[cur ^ 1], buf + off_b(s1) + kh1 * 32 * LDX, LDX);
                }
                acc[s] = mma(fa[cur], fb[cur], acc[s]);
                if (s < BSW) bacc[s < BSW ? s : 0] = mma(fa[cur], ones, bacc[s < BSW ? s : 0]);
            }
            if (NBUF == 1) lds_barrier();
            else b ^= 1;
        }
        // partial tiles in fragment order + bias sums (wgrad_tr3_kernel's layout: wgrad_reduce_kernel folds them into dW / db)
        const size_t wg = blockIdx.x;
        float* pt = a.part + wg * NTOT * 256;
        float* pbias = a.part + (size_t)gridDim.x * NTOT * 256 + wg * NTOT * 16;
#pragma unroll
        for (int s = 0; s < NSW; ++s) {
            const int tl = 4 * s + w4;
            if (tl < NTOT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pt[((size_t)tl * 4 + r) * 64 + lane] = acc[s][r];
                if (s < BSW && tl < MTA && l15 == 0) {  // every column of bacc holds the sums: rows 4 g4 + r
#pragma unroll
                    for (int r = 0; r < 4; ++r) pbias[tl * 16 + 4 * g4 + r] = bacc[s < BSW ? s : 0][r];
                }
            }
        }
    } else {
        // ================= tail waves: 16 tokens of the chunk each =================
        float dlw[BK_MT][4], dlb[BK_MT][4];
#pragma unroll
        for (int mt = 0; mt < BK_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dlw[mt][r] = dlb[mt][r] = 0.f;
        for (; ch < nchunks; ch += gridDim.x) {
            bf16_t* buf = base + (size_t)b * IMG;
            stash(buf);
            lds_barrier();
            if (ch + (int)gridDim.x < nchunks) prefetch(ch + gridDim.x);
            const long nt0 = (long)ch * TW_KC + 16 * w + l15;
            const bool tv = nt0 < a.Ntok;
            const size_t nrow = (size_t)(tv ? nt0 : a.Ntok - 1);  // clamped address, validity applied on use
            RawC4<bf16_t> xr[BK_MT], dr[BK_MT];
            rawc_load_row<bf16_t>(xr, a.x + nrow * TW_H);   // (L2: the staging has just fetched these rows)
            rawc_load_row<bf16_t>(dr, a.dy + nrow * TW_H);
            f32x4 du[BK_MT];
#pragma unroll
            for (int mt = 0; mt < BK_MT; ++mt) du[mt] = F32X4_ZERO;
            const bf16_t* arow = buf + (size_t)(16 * w + l15) * LDA + 8 * g4;  // B operand: the token's da row, natural K order
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) {
                Frag<bf16_t> df;
                frag_load(df, arow + 32 * ks);
#pragma unroll
                for (int mt = 0; mt < BK_MT; ++mt) {
                    Frag<bf16_t> af;
                    frag_load(af, wl + ((size_t)(mt * KSW + ks) * 64 + lane) * 8);
                    du[mt] = mma(af, df, du[mt]);
                }
            }
            ln_bwd_row96_raw<bf16_t>(du, xr, dr, a.dx + nrow * TW_H, nullptr, tv, lnp, dlw, dlb);
            if (NBUF == 1) lds_barrier();
            else b ^= 1;
        }
        ln_affine_flush(dlw, dlb, affl, affl + TW_H);
    }
    lds_barrier();
    for (int i = tid; i < 2 * TW_H; i += TW_THREADS) a.affpart[(size_t)blockIdx.x * 2 * TW_H + i] = affl[i];
}

// One launch: partial tiles -> wgrad_reduce (dW, db), affine partial rows -> affine_reduce (LN weight / bias gradients).
// `wa` carries the reduce's view of the problem (MA, NB = 96, dW, dbias, part).
int wgrad_reduce_launch(const WgradArgs& a, int ntot, int xb, hipStream_t st);

template <int MA, int NBUF>
static int tailw_go(const TailArgs& t, int grid, hipStream_t st) {
    constexpr int LDA = MA + 16;
    const size_t lds = (size_t)NBUF * TW_KC * (LDA + 112) * sizeof(bf16_t) + (size_t)6 * (MA / 32) * 512 * sizeof(bf16_t) + 4 * TW_H * sizeof(float);
    if (lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    int e = NBSS_SET_MAX_LDS((tailw_kernel<MA, NBUF>), lds);
    if (e) return e;
    NBSS_LAUNCH((tailw_kernel<MA, NBUF>), dim3(grid), dim3(TW_THREADS), lds, st, t);
    return NBSS_CHECK_LAUNCH();
}

// MA = 192 (T-ConvFFN W1) or 288 (attention in_proj); dW / dbias / LN-affine gradients accumulate into G at the given offsets
int tailw_launch(int MA, const TailArgs& t0, float* wgpart, size_t wgpart_bytes, float* dW, float* dbias, float* G, long long off_lnw, long long off_lnb,
                 hipStream_t st) {
    TailArgs t = t0;
    const int nchunks = cdiv(t.Ntok, TW_KC);
    const int grid = nchunks < 256 ? nchunks : 256;
    const int ntot = (MA / 16) * (TW_H / 16);
    const size_t need = (size_t)grid * ntot * 272 * sizeof(float) + (size_t)grid * 2 * TW_H * sizeof(float);
    if (need > wgpart_bytes) return NBSS_EUNSUPPORTED;
    t.part = wgpart;
    t.affpart = wgpart + (size_t)grid * ntot * 272;
    // (MA = 288, the attention's in_proj, is the same template: its 108 tiles are 27 per accumulator wave and spill 78 registers; with 6
    //  accumulator + 2 tail waves the tail path spills instead (9 k-steps of W^T fragments) — not instantiated until one of them fits)
    int e = MA == 192 ? tailw_go<192, 2>(t, grid, st) : NBSS_EUNSUPPORTED;
    if (e) return e;
    WgradArgs a;
    a.A = nullptr; a.lda = MA; a.MA = MA; a.B = nullptr; a.ldb = TW_H; a.NB = TW_H; a.groups = 1; a.mvalid = 0; a.nvalid = 0; a.taps = 1;
    a.shift_stride = 1; a.shift_dim = 0; a.stats = nullptr; a.gamma = nullptr; a.beta = nullptr; a.dW = dW; a.dbias = dbias; a.Ntok = t.Ntok; a.F = 1; a.T = 1;
    a.part = wgpart;
    if ((e = wgrad_reduce_launch(a, ntot, grid, st))) return e;
    AffSegs sg;
    sg.n = 2;
    sg.off[0] = off_lnw; sg.cnt[0] = TW_H;
    sg.off[1] = off_lnb; sg.cnt[1] = TW_H;
    return affine_reduce_launch(t.affpart, grid, sg, G, st);
}

// T-ConvFFN: da1 (FFN = 192), W1^T fragments K_TF_W1_TN, LayerNorm P_TF_LN_*; dW1 / db1 / LN-affine gradients into G
int tailw_tconvffn(const nbss_cfg& c, const LayerPtrs& lp, const void* packed, int layer, const void* x, const void* dy, void* dx, float* stats,
                   const void* da1, float* wgpart, float* G, const float* P, hipStream_t st) {
    (void)P;
    TailArgs t;
    t.A = (const bf16_t*)da1; t.x = (const bf16_t*)x; t.dy = (const bf16_t*)dy; t.dx = (bf16_t*)dx; t.stats = stats;
    t.gamma = lp.p[P_TF_LN_W]; t.beta = lp.p[P_TF_LN_B];
    t.WT = (const bf16_t*)packed + pack_off(c, layer, K_TF_W1_TN);
    t.part = nullptr; t.affpart = nullptr;
    t.Ntok = c.B * c.F * c.T;
    return tailw_launch(192, t, wgpart, WGPART_BYTES, G + param_off(c, layer, P_TF_W1), G + param_off(c, layer, P_TF_B1), G,
                        param_off(c, layer, P_TF_LN_W), param_off(c, layer, P_TF_LN_B), st);
}
