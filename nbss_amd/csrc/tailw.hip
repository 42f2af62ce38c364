// tailw.hip — the TAIL of a narrow-band block's backward pass fused with the weight gradient of its first linear map (bf16 stream).
//
// Both the T-ConvFFN (W1: H -> FFN) and the attention (in_proj: H -> 3H) start with  a = W LN(x) + b  and their data-gradient kernels end
// with the same three steps on the emitted pre-activation gradient `da` ([MA/24][N][24] group-major operand):
//     du = W^T da            LayerNorm backward + residual:  dx = dy + LN'(du)            dW = da^T LN(x),  db = colsum(da),  dgamma, dbeta
// Done as separate kernels (a tail kernel and a wgrad problem) `da` and `x` are read from HBM twice: 3 S·B of the 8 S·B the pair moves.
// Here one persistent kernel (256 workgroups x 8 waves, 64-token chunks, wgrad.hip's staging scheme: register prefetch of the next chunk,
// ROW-major LDS images, transposing reads for the token-contraction) reads them once:
//   * the accumulator waves (4 for W1's 72 tiles, 6 for in_proj's 108: 18 tiles each) contract the chunk's dW tiles (A = da image, B = xhat
//     image, K = the 64 tokens),
//   * the tail waves (4 resp. 2) take the chunk's 16-token tiles: du = W^T da with the B operand straight from the da image rows (natural K
//     order) and the W^T fragments resident in LDS, LayerNorm backward in registers (x, dy as 8-byte C-layout pieces), dx stored.
//   The two roles are two chunk loops with the same barrier sequence, so that their register sets have disjoint live ranges.
// The LayerNorm affine gradients need no per-token work at all: the B image holds xhat = (x - mean) rstd (not LN(x)), so the contraction
// yields D = da^T xhat, and with db = colsum(da):   dW = D gamma + db (x) beta,   dgamma[i] = sum_o W[o][i] D[o][i],   dbeta[i] = sum_o W[o][i] db[o]
// (du = W^T da summed against xhat / 1 over the tokens, reordered) — tailw_finalize_kernel applies them while it folds the partial tiles.
// (48 per-lane accumulators for dgamma / dbeta were what kept the tail path above 256 registers.)
#include "launch.h"
#include "layout.h"
#include "prof.h"
#include "blocks.h"
#include "wgrad.h"
#include "fold.h"
#include "foldk.h"
#include "side.h"

#define TW_KC 64
#define TW_H 96
#define TW_THREADS 512
#ifndef TW288_NTW
#define TW288_NTW 4
#endif
// timing knock-outs (A/B flavours only, results wrong): bit 0 = no weight-gradient contraction, bit 1 = no tail tiles: what is left is the staging
#ifndef TW_TAIL_UNROLL
#define TW_TAIL_UNROLL 2
#endif
#ifndef TW_KO
#define TW_KO 0
#endif

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
    int Ntok;
    int flip;             // launch.h: walk the chunks from the last one down (the data-gradient kernel before this one ran the other way)
};

// Wave specialisation: waves 4-7 hold ALL dW accumulators (NTOT / 4 tiles each), waves 0-3 run the tail.  The two roles are two
// separate chunk loops (same sequence of barriers) so that their register sets — 27 accumulator tiles for in_proj on one side, du / x / dy /
// affine sums on the other — have disjoint live ranges; as one loop with `if (w < 4)` inside, both sets stayed live and 65-146 VGPRs spilled.
// NTW = tail waves: 4 (one 16-token tile of the chunk each, 4 accumulator waves) or 2 (two tiles each, 6 accumulator waves: in_proj's
// 108 tiles are 18 per wave then instead of 27, which did not fit the register file)
// PD = chunks requested ahead (register sets of the staging): 1; 2 was built for W1 (223 + 24 VGPRs, 6 of them spilled) and measured SLOWER in round 6,
// 281 -> 345 us per launch: the set that rotates into the stash set has to have arrived right behind the barrier, one more wait in the chunk's critical path
template <int MA, int NBUF, int NTW, int PD>
__global__ __launch_bounds__(TW_THREADS, 2) void tailw_kernel(TailArgs a) {
    constexpr int LDA = MA + 16;                    // image row strides == 16 (mod 32) elements: wgrad.hip tr_ld()
    constexpr int LDX = 112;                        // 96 + 16
    constexpr int IMGA = TW_KC * LDA, IMG = IMGA + TW_KC * LDX;
    constexpr int MTA = MA / 16, NTB = TW_H / 16, NTOT = MTA * NTB;  // tile tl = nt * MTA + mt
    constexpr int NWW = 8 - NTW;                     // accumulator waves
    constexpr int NSW = (NTOT + NWW - 1) / NWW;      // tile slots of an accumulator wave
    constexpr int BSW = (MTA + NWW - 1) / NWW;       // slots that can hold nt == 0 tiles (bias sums)
    constexpr int KSW = MA / 32;                     // k-steps of du = W^T da
    constexpr int PA = MA / 8, NVA = TW_KC * PA, NVX = TW_KC * (TW_H / 8);
    constexpr int UA = (NVA + TW_THREADS - 1) / TW_THREADS, UX = (NVX + TW_THREADS - 1) / TW_THREADS;  // whole vector slots: da | x
    static_assert(LDA % 32 == 16, "image stride");
    NBSS_LDS(smem);
    bf16_t* base = reinterpret_cast<bf16_t*>(smem);
    bf16_t* wl = base + (size_t)NBUF * IMG;                       // W^T fragments
    float* lnp = reinterpret_cast<float*>(wl + 6 * KSW * 512);    // LayerNorm gamma (du is scaled by it in the tail)
    const int tid = threadIdx.x, lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id_u();
    for (int i = tid; i < NBUF * IMG / 2; i += TW_THREADS) reinterpret_cast<uint32_t*>(base)[i] = 0u;
    for (int i = tid; i < 6 * KSW * 64; i += TW_THREADS) reinterpret_cast<u32x4*>(wl)[i] = reinterpret_cast<const u32x4*>(a.WT)[i];
    for (int i = tid; i < TW_H; i += TW_THREADS) lnp[i] = a.gamma[i];

    const int nchunks = cdiv(a.Ntok, TW_KC);
    // staging: slot u < UA is a da piece for every thread, the rest x pieces (a few lanes of the last slot of each kind idle)
    struct Pre {
        u32x4 A[UA], X[UX];
        float mu[UX], rs[UX];
    };
    Pre ps0, ps1;  // (ps1: PD == 2 only)
    auto phys = [&](int ch) -> int { return a.flip ? nchunks - 1 - ch : ch; };  // logical -> token chunk
    auto prefetch = [&](Pre& q, int ch) {
        const long n0 = (long)phys(ch) * TW_KC;
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int v = tid + u * TW_THREADS, r = v / PA, col = (v % PA) * 8;
            q.A[u] = (u32x4){0, 0, 0, 0};
            if (v < NVA && n0 + r < a.Ntok) q.A[u] = *reinterpret_cast<const u32x4*>(a.A + ((size_t)(col / 24) * a.Ntok + n0 + r) * 24 + col % 24);
        }
#pragma unroll
        for (int u = 0; u < UX; ++u) {
            const int v = tid + u * TW_THREADS, r = v / (TW_H / 8), col = (v % (TW_H / 8)) * 8;
            q.X[u] = (u32x4){0, 0, 0, 0};
            q.mu[u] = 0.f; q.rs[u] = 0.f;
            if (v < NVX && n0 + r < a.Ntok) {
                q.X[u] = *reinterpret_cast<const u32x4*>(a.x + (size_t)(n0 + r) * TW_H + col);
                q.mu[u] = a.stats[2 * (n0 + r)];
                q.rs[u] = a.stats[2 * (n0 + r) + 1];
            }
        }
    };
    auto stash = [&](const Pre& q, bf16_t* buf) {
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int v = tid + u * TW_THREADS, r = v / PA, col = (v % PA) * 8;
            if (v < NVA) *reinterpret_cast<u32x4*>(buf + r * LDA + col) = q.A[u];
        }
#pragma unroll
        for (int u = 0; u < UX; ++u) {
            const int v = tid + u * TW_THREADS, r = v / (TW_H / 8), col = (v % (TW_H / 8)) * 8;
            if (v < NVX) {  // xhat on the fly (rows past Ntok: rstd = 0, they stay 0)
                u32x4 xq = q.X[u];
                float f[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { f[2 * i] = bf2f((bf16_t)(xq[i] & 0xFFFF)); f[2 * i + 1] = bf2f((bf16_t)(xq[i] >> 16)); }
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (f[e] - q.mu[u]) * q.rs[u];
#pragma unroll
                for (int i = 0; i < 4; ++i) xq[i] = pack2bf(f[2 * i], f[2 * i + 1]);
                *reinterpret_cast<u32x4*>(buf + IMGA + r * LDX + col) = xq;
            }
        }
    };

    int ch = blockIdx.x;
    const int gstep = (int)gridDim.x;
    // behind the barrier of chunk `ch`: request the next chunk(s).  PD == 2: the set that arrived (requested a whole chunk ago) moves into the stash set,
    // its registers take the request for the chunk after the next — two chunks in flight through one loop body (two inlined bodies, one per set,
    // spilled 176 registers)
    auto advance = [&]() {
        if (PD == 2) {
            ps0 = ps1;
            if (ch + 2 * gstep < nchunks) prefetch(ps1, ch + 2 * gstep);
        } else if (ch + gstep < nchunks) {
            prefetch(ps0, ch + gstep);
        }
    };
    if (ch < nchunks) prefetch(ps0, ch);
    if (PD == 2 && ch + gstep < nchunks) prefetch(ps1, ch + gstep);
    lds_barrier();  // zero fill, fragments, gamma / beta
    int b = 0;
    if (w >= NTW) {
        // ================= weight-gradient waves: tile tl = NWW s + (w - NTW) =================
        const int w4 = w - NTW;
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
        auto off_a = [&](int s) { const int tl = NWW * s + w4; return la + (tl < NTOT ? tl % MTA : 0) * 16; };  // slots past the last tile
        auto off_b = [&](int s) { const int tl = NWW * s + w4; return lb + (tl < NTOT ? tl / MTA : 0) * 16; };  // re-contract tile 0: never flushed
        for (; ch < nchunks; ch += gstep) {
            bf16_t* buf = base + (size_t)b * IMG;
            stash(ps0, buf);
            lds_barrier();
            advance();
            // software pipeline over the NSW x 2 (tile, k-half) steps: the operands of step i+1 are requested before step i's MFMA
            Frag<bf16_t> fa[2], fb[2];
            frag_load_tr(fa[0], buf + off_a(0), LDA);
            frag_load_tr(fb[0], buf + off_b(0), LDX);
#pragma unroll
            for (int i = 0; i < ((TW_KO & 1) ? 1 : NSW * 2); ++i) {
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
            const int tl = NWW * s + w4;
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
        // The x / dy rows of a tile are requested ONE CHUNK AHEAD (round 6): as loads at the head of the tile's own math they were an exposed HBM round
        // trip per chunk for dy (36 - 54 MFMAs do not cover it) — with the tail tiles knocked out the kernel ran 424 -> 200 us (in_proj) and 283 ->
        // 153 us (W1), with the contraction waves knocked out 424 -> 408: the tail waves were the critical path of every chunk.
        constexpr int TPW = TW_KC / 16 / NTW;  // tiles of a chunk per tail wave
        auto tail_rows = [&](int ch, int tile, bool& tv, size_t& nrow) {
            const long nt0 = (long)phys(ch) * TW_KC + 16 * tile + l15;
            tv = nt0 < a.Ntok;
            nrow = (size_t)(tv ? nt0 : a.Ntok - 1);  // clamped address, validity applied on use
        };
        auto tail_load = [&](int ch, int tile, RawC4<bf16_t> (&xr)[BK_MT], RawC4<bf16_t> (&dr)[BK_MT]) {
            bool tv;
            size_t nrow;
            tail_rows(ch, tile, tv, nrow);
            rawc_load_row<bf16_t>(xr, a.x + nrow * TW_H);
            rawc_load_row<bf16_t>(dr, a.dy + nrow * TW_H);
        };
        auto tail_tile = [&](const bf16_t* buf, int ch, int tile, const RawC4<bf16_t> (&xr)[BK_MT], const RawC4<bf16_t> (&dr)[BK_MT]) {
            bool tv;
            size_t nrow;
            tail_rows(ch, tile, tv, nrow);
            f32x4 du[BK_MT];
#pragma unroll
            for (int mt = 0; mt < BK_MT; ++mt) du[mt] = F32X4_ZERO;
            const bf16_t* arow = buf + (size_t)(16 * tile + l15) * LDA + 8 * g4;  // B operand: the token's da row, natural K order
#pragma unroll(KSW > 6 ? 1 : TW_TAIL_UNROLL)  // (9 k-steps unrolled: the scheduler hoists all 54 W^T fragment reads, 216 registers)
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
            ln_bwd_row96_raw_na<bf16_t>(du, xr, dr, a.dx + nrow * TW_H, tv, lnp);
        };
        RawC4<bf16_t> xn[TPW][BK_MT], dn[TPW][BK_MT];
        if (ch < nchunks) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) tail_load(ch, w + i * NTW, xn[i], dn[i]);
        }
        for (; ch < nchunks; ch += gstep) {
            bf16_t* buf = base + (size_t)b * IMG;
            stash(ps0, buf);
            lds_barrier();
            const int nxt = ch + gstep;
            advance();
            RawC4<bf16_t> xc[TPW][BK_MT], dc[TPW][BK_MT];
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int mt = 0; mt < BK_MT; ++mt) { xc[i][mt] = xn[i][mt]; dc[i][mt] = dn[i][mt]; }
            if (nxt < nchunks) {
#pragma unroll
                for (int i = 0; i < TPW; ++i) tail_load(nxt, w + i * NTW, xn[i], dn[i]);
            }
#pragma unroll
            for (int i = 0; i < ((TW_KO & 2) ? 0 : TPW); ++i) tail_tile(buf, ch, w + i * NTW, xc[i], dc[i]);
            if (NBUF == 1) lds_barrier();
            else b ^= 1;
        }
    }
}

// Second pass: ONE block per (tile, accumulator register r) folds the workgroups' partial D values of its 64 elements (and the bias sums of their rows)
// in a fixed order — thread (slice, lane) sums its slice of the workgroups, the four slices meet in LDS, slice 0 adds them in slice order — and adds
//   dW[o][i] += D gamma[i] + db[o] beta[i],   db[o] (nt == 0 tiles)
// with plain read-modify-writes (one owner per element).  The LayerNorm affine gradients  dgamma[i] = sum_o W[o][i] D[o][i],  dbeta[i] = sum_o W[o][i] db[o]
// cross the blocks of a column: every block leaves its 16 + 16 column sums (over its four rows) in its own, now consumed, part of workgroup 0's partial
// tile, and tailw_affine_kernel adds the 4 MTA blocks of each column in (tile, r) order.  No float atomics anywhere: bitwise repeatable parameter
// gradients.  (256-thread blocks: see wgrad_reduce_kernel.)
// W = the fp32 master weight [MA][96].  Tile tl = nt * MTA + mt; block (tl, r), lane: row 16 mt + 4 (lane >> 4) + r, column 16 nt + (lane & 15).
#define TW_RSL 4
static_assert(TW_RSL == FK_RSL && TW_H == FK_H, "foldk.h");
// (the bodies live in foldk.h: fold.hip's table kernel runs them too)
__global__ __launch_bounds__(64 * TW_RSL) void tailw_finalize_kernel(float* __restrict__ part, int xb, int MTA, const float* __restrict__ W,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ dW,
                                                                     float* __restrict__ dbias) {
    NBSS_LDS(smem);
    fk_tailw_finalize(part, xb, MTA, (int)(gridDim.x >> 2), W, gamma, beta, dW, dbias, (int)blockIdx.x, reinterpret_cast<float*>(smem));
}
// dgamma[i] += the column sums tailw_finalize_kernel left, over the MTA tiles of column tile i / 16 and their four blocks in (tile, r) order; dbeta
// likewise (threads 96 ..)
__global__ __launch_bounds__(192) void tailw_affine_kernel(const float* __restrict__ part, int MTA, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    fk_tailw_affine(part, MTA, dgamma, dbeta);
}

template <int MA, int NBUF, int NTW, int PD>
static int tailw_go(const TailArgs& t, int grid, hipStream_t st) {
    constexpr int LDA = MA + 16;
    const size_t lds = (size_t)NBUF * TW_KC * (LDA + 112) * sizeof(bf16_t) + (size_t)6 * (MA / 32) * 512 * sizeof(bf16_t) + TW_H * sizeof(float);
    if (lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    int e = NBSS_SET_MAX_LDS((tailw_kernel<MA, NBUF, NTW, PD>), lds);
    if (e) return e;
    NBSS_LAUNCH((tailw_kernel<MA, NBUF, NTW, PD>), dim3(grid), dim3(TW_THREADS), lds, st, t);
    return NBSS_CHECK_LAUNCH();
}

// MA = 192 (T-ConvFFN W1) or 288 (attention in_proj).  W: fp32 master weight [MA][96]; the gradients accumulate into dW / dbias / dgamma / dbeta
// The finalize pass only produces parameter gradients: it goes to the gradient stream (side.h), which is handed back in *gs for the caller's own folds.
int tailw_launch(int MA, const TailArgs& t0, float* wgpart, size_t wgpart_bytes, const float* W, float* dW, float* dbias, float* dgamma, float* dbeta,
                 hipStream_t st, const Side* sd, hipStream_t* gs_out) {
    TailArgs t = t0;
    t.flip = walk_flip_next();
    const int nchunks = cdiv(t.Ntok, TW_KC);
    const int grid = nchunks < 256 ? nchunks : 256;
    const int ntot = (MA / 16) * (TW_H / 16);
    if ((size_t)grid * ntot * 272 * sizeof(float) > wgpart_bytes) return NBSS_EUNSUPPORTED;
    bool batched = false;
    if (g_fold) {  // inside a FoldScope (fold.h): the partial tiles come from the scope's pool, the two second passes join its first and second stage
        int err;
        void* p = g_fold->alloc((size_t)grid * ntot * 272 * sizeof(float), &err);
        if (err) return err;
        if (p) {
            wgpart = (float*)p;
            batched = true;
        } else if ((err = g_fold->flush())) {
            return err;
        }
    }
    t.part = wgpart;
#ifndef TW192_PD
#define TW192_PD 1  // (A/B: 2 = two chunks in flight for W1 — measured 281 -> 345 us per launch, see the kernel's note)
#endif
    int e = MA == 192 ? tailw_go<192, 2, 4, TW192_PD>(t, grid, st)
#ifdef NBSS_TW288_NBUF1
            : MA == 288 ? tailw_go<288, 1, 2, 1>(t, grid, st)
#else
            : MA == 288 ? tailw_go<288, 2, TW288_NTW, 1>(t, grid, st)  // both buffers + the 54 W^T fragments: 162 176 of 163 840 bytes
#endif
            : NBSS_EUNSUPPORTED;
    if (e) return e;
    const hipStream_t gs = side_fork(sd, st);
    if (gs_out) *gs_out = gs;
    if (batched) {
        g_fold->st = gs;
        FoldItem it;
        it.kind = FK_TAILW_FIN;
        it.gx = 4 * ntot; it.gy = 1; it.nblk = it.gx;
        it.u.tw.part = wgpart; it.u.tw.xb = grid; it.u.tw.MTA = MA / 16; it.u.tw.ntot = ntot; it.u.tw.W = W; it.u.tw.gamma = t.gamma; it.u.tw.beta = t.beta;
        it.u.tw.dW = dW; it.u.tw.dbias = dbias; it.u.tw.dgamma = dgamma; it.u.tw.dbeta = dbeta;
        if ((e = g_fold->add(1, it))) return e;
        it.kind = FK_TAILW_AFF;
        it.gx = 1; it.nblk = 1;
        return g_fold->add(2, it);
    }
    NBSS_FOLD_LAUNCH(tailw_finalize_kernel, dim3(4 * ntot), dim3(64 * TW_RSL), 2 * TW_RSL * 64 * sizeof(float), gs, wgpart, grid, MA / 16, W, t.gamma, t.beta, dW, dbias);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    NBSS_FOLD_LAUNCH(tailw_affine_kernel, dim3(1), dim3(2 * TW_H), 0, gs, (const float*)wgpart, MA / 16, dgamma, dbeta);
    return NBSS_CHECK_LAUNCH();
}

// T-ConvFFN: da1 (FFN = 192), W1^T fragments K_TF_W1_TN, LayerNorm P_TF_LN_*; dW1 / db1 / LN-affine gradients into G
int tailw_tconvffn(const nbss_cfg& c, const LayerPtrs& lp, const void* packed, int layer, const void* x, const void* dy, void* dx, float* stats,
                   const void* da1, float* wgpart, float* G, const float* P, hipStream_t st, const Side* sd, hipStream_t* gs) {
    (void)P;
    TailArgs t;
    t.A = (const bf16_t*)da1; t.x = (const bf16_t*)x; t.dy = (const bf16_t*)dy; t.dx = (bf16_t*)dx; t.stats = stats;
    t.gamma = lp.p[P_TF_LN_W]; t.beta = lp.p[P_TF_LN_B];
    t.WT = (const bf16_t*)packed + pack_off(c, layer, K_TF_W1_TN);
    t.part = nullptr;
    t.Ntok = c.B * c.F * c.T;
    return tailw_launch(192, t, wgpart, WGPART_BYTES, lp.p[P_TF_W1], G + param_off(c, layer, P_TF_W1), G + param_off(c, layer, P_TF_B1),
                        G + param_off(c, layer, P_TF_LN_W), G + param_off(c, layer, P_TF_LN_B), st, sd, gs);
}

// attention: dqkv (3H = 288), in_proj^T fragments K_INP_TN, LayerNorm P_MH_LN_*
int tailw_mhsa(const nbss_cfg& c, const LayerPtrs& lp, const void* packed, int layer, const void* x, const void* dy, void* dx, float* stats,
               const void* dqkv, float* wgpart, float* G, hipStream_t st, const Side* sd, hipStream_t* gs) {
    TailArgs t;
    t.A = (const bf16_t*)dqkv; t.x = (const bf16_t*)x; t.dy = (const bf16_t*)dy; t.dx = (bf16_t*)dx; t.stats = stats;
    t.gamma = lp.p[P_MH_LN_W]; t.beta = lp.p[P_MH_LN_B];
    t.WT = (const bf16_t*)packed + pack_off(c, layer, K_INP_TN);
    t.part = nullptr;
    t.Ntok = c.B * c.F * c.T;
    return tailw_launch(288, t, wgpart, WGPART_BYTES, lp.p[P_INP_W], G + param_off(c, layer, P_INP_W), G + param_off(c, layer, P_INP_B),
                        G + param_off(c, layer, P_MH_LN_W), G + param_off(c, layer, P_MH_LN_B), st, sd, gs);
}
