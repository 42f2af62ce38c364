// wgrad.hip — weight gradients of the linear maps on the SpatialNet hot path (bf16 stream: all but the three that are contracted where
// their operands already are — the f-convs inside fconv_bwd, W1 / in_proj inside tailw.hip's fused tail kernel; fp32 stream: all of them).
//
// All of them are the same contraction over tokens n = (b,f,t):
//     dW[o][i][tap] = sum_n dY[n][o] * X[n + (tap - taps/2) * shift][i]          (grouped)
// dense layers have taps = 1; the T-convs shift by one frame (valid while 0 <= t+d < T), the
// F-convs by T rows (valid while 0 <= f+d < F); LinearGroup is "8 groups of F x F" over rows
// (b,t).  X may be LayerNorm(x) applied on the fly from the per-token (mean, rstd) the data-gradient
// kernels emit, so the normalised tensor is never materialised.
//
// Two kernels:
//  * wgrad_tr3_kernel (bf16 stream, every hot problem): 256 persistent workgroups of 8 waves; a workgroup owns ALL output
//    tiles of the problem and contracts its share of 64-token chunks.  Operand rows are fetched as 16-byte pieces (register
//    prefetch of the next chunk, double-buffered LDS) into ROW-major LDS images; MFMA fragments whose K dimension is the token
//    axis come out of them through transposing reads (ds_read_b64_tr_b16), software-pipelined one tile ahead.  Chunks of a
//    tapped problem are aligned so that a tap is a row offset inside ONE X image.  Accumulators stay in registers across
//    all chunks; each workgroup stores its partial tiles and wgrad_reduce_kernel folds them into the fp32 gradient (the bias
//    gradients — column sums of dY — are an extra MFMA against a ones fragment).
//  * wgrad_kernel (fp32 stream, and shapes the first does not take: LinearGroup's F x F blocks, encoder/decoder): operands
//    staged TRANSPOSED in LDS ([column][32 tokens], 80-byte rows: conflict-free 16-byte fragment reads), atomicAdd flush.
// r01 history (profiles/README.md): per-group workgroups re-read every row 8x (27 ms/step at batch 8) -> 8-wave workgroups,
// hoisted divisions (18) -> transposing reads (16.6) -> double buffering + LN affine through LDS (12.9) -> two-stage flush
// (10.7) -> 64-token chunks / one X image (8.8) -> F-convs on the same kernel (8.1) -> pipelined reads (7.5).
#include "launch.h"
#include "layout.h"
#include "wgrad.h"
#include "fold.h"
#include "foldk.h"
#include <cstdlib>
#include "prof.h"

#define WG_KC 32
#define WG_LD 40     // LDS row length in elements (32 tokens + pad)
#define WG_TPW 14    // output tiles per wave
#define WG_WAVES 8
#define WG_THREADS (WG_WAVES * 64)

template <class T, int CW>  // CW = channels per staging block (4 or 8)
NBSS_DEV void load_cw(const T* p, float (&v)[8]) {
    if (CW == 8) load8(p, v);
    else load4(p, v);
}

// TPW = output tiles per wave: WG_TPW (LinearGroup: 81 tiles per group), or 4 for the small problems (encoder / decoder: 24 / 6 tiles) so
// that the kernel fits 128 VGPRs and several workgroups per CU overlap their load -> barrier -> MFMA -> barrier rounds (one 8-wave
// workgroup per CU spent every chunk's HBM latency exposed: 560 us for 220 MB of operands)
template <class T, int CW, int TPW>
__global__ __launch_bounds__(WG_THREADS, TPW <= 4 ? 4 : 2) void wgrad_kernel(WgradArgs a) {
    NBSS_LDS(smem);
    const int tid = threadIdx.x, lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const int mg = a.MA / a.groups, ng = a.NB / a.groups;
    const int mv = a.mvalid ? a.mvalid : mg, nv = a.nvalid ? a.nvalid : ng;
    const int mtiles = cdiv(mg, 16), nexp = a.taps * ng, ntiles_all = cdiv(nexp, 16);
    // blockIdx.y selects a group when the groups are too large to share a workgroup (LinearGroup); else all groups
    const bool per_group = gridDim.y > 1;
    // blockIdx.z selects a range of output-column tiles when even one group has more tiles than a workgroup holds (LinearGroup at
    // F = 257: 17 x 17); only dense per-group problems are launched that way (wgrad_launch_t)
    const int ntz = cdiv(ntiles_all, (int)gridDim.z), nt0 = (int)blockIdx.z * ntz;
    const int ntiles = ntiles_all - nt0 < ntz ? ntiles_all - nt0 : ntz;
    const int tpg = mtiles * ntiles;
    const int g_lo = per_group ? blockIdx.y : 0, ngrp = per_group ? 1 : a.groups;
    const int acols0 = g_lo * mg, ncolsA = ngrp * mg, bcols0 = g_lo * ng + nt0 * 16;
    const int ncolsB = gridDim.z > 1 ? (ng - nt0 * 16 < ntiles * 16 ? ng - nt0 * 16 : ntiles * 16) : ngrp * ng;
    const int mgp = mtiles * 16, ngp = ntiles * 16;
    const int rowsA = ngrp * mgp, rowsB = ngrp * ngp;
    T* At = reinterpret_cast<T*>(smem);            // [rowsA][WG_LD]   row = g*mgp + m
    T* Bt = At + (size_t)rowsA * WG_LD;            // [rowsB][WG_LD]   row = g*ngp + tap*ng + i
    for (int i = tid; i < (rowsA + rowsB) * WG_LD; i += WG_THREADS) store1(At + i, 0.f);

    f32x4 acc[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) acc[s] = F32X4_ZERO;
    float bsum = 0.f;
    const bool do_bias = a.dbias != nullptr && blockIdx.z == 0;
    const T* Ag = reinterpret_cast<const T*>(a.A);
    const T* Bg = reinterpret_cast<const T*>(a.B);
    const int pcA = ncolsA / CW, pcB = ncolsB / CW, center = a.taps / 2;
    const int qB = (WG_KC / 4) * pcB;
    const int nblkA = (WG_KC / 4) * pcA, nblk = nblkA + a.taps * qB;
    const int nchunks = cdiv(a.Ntok, WG_KC);
    const int ntot = ngrp * tpg;
    lds_barrier();

    // Block descriptors are chunk-independent: resolve the (token quad, columns, tap, LDS rows) of this thread's blocks ONCE
    // (runtime integer divisions are ~40 VALU instructions each and used to dominate the staging loop).
    constexpr int MAXB = 3;
    int bq[MAXB], bc0[MAXB], bd[MAXB];
    T* brow[MAXB];
    bool bA[MAXB], bok[MAXB];
#pragma unroll
    for (int u = 0; u < MAXB; ++u) {
        const int bi = tid + u * WG_THREADS;
        bok[u] = bi < nblk;
        bA[u] = bi < nblkA;
        int tap = 0, c0;
        if (bA[u]) {
            bq[u] = bi / pcA;
            c0 = (bi % pcA) * CW;
            brow[u] = At + (size_t)((c0 / mg) * mgp + c0 % mg) * WG_LD;
        } else {
            const int b2 = bok[u] ? bi - nblkA : 0, rem = b2 % qB;
            tap = b2 / qB;
            bq[u] = rem / pcB;
            c0 = (rem % pcB) * CW;
            brow[u] = Bt + (size_t)((c0 / ng) * ngp + tap * ng + c0 % ng) * WG_LD;
        }
        bc0[u] = c0;
        bd[u] = tap - center;
    }
    const bool shifted = a.taps > 1;

    for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const int n0 = ch * WG_KC;
        // ---- stage both operands, transposed (4 tokens x CW channels per thread-block) ----
#pragma unroll
        for (int u = 0; u < MAXB; ++u) {
            if (!bok[u]) continue;
            float v[4][8];
            const int nb = n0 + 4 * bq[u];
            if (bA[u]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (nb + r < a.Ntok) load_cw<T, CW>(Ag + (size_t)(nb + r) * a.lda + acols0 + bc0[u], v[r]);
                    else
#pragma unroll
                        for (int e = 0; e < CW; ++e) v[r][e] = 0.f;
                }
            } else {
                int pos = 0;
                if (shifted) pos = a.shift_dim == 0 ? nb % a.T : (nb / a.T) % a.F;  // one division per block and chunk
                const int lim = a.shift_dim == 0 ? a.T : a.F;
                const int tb = (shifted && a.shift_dim == 1) ? nb % a.T : 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = nb + r;
#pragma unroll
                    for (int e = 0; e < CW; ++e) v[r][e] = 0.f;
                    // index along the shifted axis of row r (frames wrap at T; the frequency changes every T rows)
                    int pr = pos;
                    if (shifted) {
                        if (a.shift_dim == 0) { pr = pos + r; while (pr >= a.T) pr -= a.T; }  // (T < 4: a quad of rows wraps more than once)
                        else if (tb + r >= a.T) pr = (n / a.T) % a.F;
                    }
                    if (n < a.Ntok && pr + bd[u] >= 0 && pr + bd[u] < lim) {
                        const size_t ns = (size_t)((long)n + (long)bd[u] * a.shift_stride);
                        load_cw<T, CW>(Bg + ns * a.ldb + bcols0 + bc0[u], v[r]);
                        if (a.stats) {
                            const float mu = a.stats[2 * ns], rs = a.stats[2 * ns + 1];
#pragma unroll
                            for (int e = 0; e < CW; ++e) {
                                const int col = bcols0 + bc0[u] + e;
                                v[r][e] = round_to((v[r][e] - mu) * rs * a.gamma[col] + a.beta[col], Bg);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < CW; ++e) store4(brow[u] + (size_t)e * WG_LD + 4 * bq[u], v[0][e], v[1][e], v[2][e], v[3][e]);
        }
        for (int bi = tid + MAXB * WG_THREADS; bi < nblk; bi += WG_THREADS) {  // (very wide operands only; none on the SpatialNet-small path)
            float v[4][8];
            const bool isA = bi < nblkA;
            int q, c0, tap = 0;
            if (isA) { q = bi / pcA; c0 = (bi % pcA) * CW; }
            else { const int b2 = bi - nblkA, rem = b2 % qB; tap = b2 / qB; q = rem / pcB; c0 = (rem % pcB) * CW; }
            const int d = tap - center;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 4 * q + r;
#pragma unroll
                for (int e = 0; e < CW; ++e) v[r][e] = 0.f;
                if (n >= a.Ntok) continue;
                if (isA) { load_cw<T, CW>(Ag + (size_t)n * a.lda + acols0 + c0, v[r]); continue; }
                const int pos = a.shift_dim == 0 ? n % a.T : (n / a.T) % a.F, lim = a.shift_dim == 0 ? a.T : a.F;
                if (pos + d < 0 || pos + d >= lim) continue;
                const size_t ns = (size_t)((long)n + (long)d * a.shift_stride);
                load_cw<T, CW>(Bg + ns * a.ldb + bcols0 + c0, v[r]);
                if (a.stats) {
                    const float mu = a.stats[2 * ns], rs = a.stats[2 * ns + 1];
#pragma unroll
                    for (int e = 0; e < CW; ++e) v[r][e] = round_to((v[r][e] - mu) * rs * a.gamma[bcols0 + c0 + e] + a.beta[bcols0 + c0 + e], Bg);
                }
            }
            T* row = isA ? At + (size_t)((c0 / mg) * mgp + c0 % mg) * WG_LD : Bt + (size_t)((c0 / ng) * ngp + tap * ng + c0 % ng) * WG_LD;
#pragma unroll
            for (int e = 0; e < CW; ++e) store4(row + (size_t)e * WG_LD + 4 * q, v[0][e], v[1][e], v[2][e], v[3][e]);
        }
        lds_barrier();
        // ---- MFMA: K = the 32 tokens of this chunk ----
#pragma unroll
        for (int s = 0; s < TPW; ++s) {
            const int tl = s * WG_WAVES + w;
            if (tl < ntot) {
                const int g = tl / tpg, rem = tl % tpg, mt = rem / ntiles, nt = rem % ntiles;
                Frag<T> fa, fb;
                frag_load(fa, At + (size_t)(g * mgp + mt * 16 + l15) * WG_LD + 8 * g4);
                frag_load(fb, Bt + (size_t)(g * ngp + nt * 16 + l15) * WG_LD + 8 * g4);
                acc[s] = mma(fa, fb, acc[s]);
            }
        }
        if (do_bias && tid < rowsA) {
            float v[8];
#pragma unroll
            for (int k8 = 0; k8 < WG_KC; k8 += 8) {
                load8(At + (size_t)tid * WG_LD + k8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum += v[e];
            }
        }
        lds_barrier();
    }

    // ---- flush ----
    if (a.part) {  // two-stage: partial tiles in fragment order (+ the bias sums of the nt = 0 tiles), folded by wgrad_reduce_kernel.
        // The atomicAdd flush below costs the LinearGroup problem (8 groups x 81 tiles x 48 x-blocks = 8 M same-address-heavy atomics) most
        // of its 96 us for 34 MB of operands.
        const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        float* pt = a.part + wg * ntot * 256;
        float* pbias = a.part + (size_t)gridDim.y * gridDim.x * ntot * 256 + wg * ntot * 16;
#pragma unroll
        for (int s = 0; s < TPW; ++s) {
            const int tl = s * WG_WAVES + w;
            if (tl < ntot) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pt[((size_t)tl * 4 + r) * 64 + lane] = acc[s][r];
            }
        }
        if (do_bias && tid < rowsA) {
            const int gl = tid / mgp, m = tid % mgp;
            pbias[(size_t)(gl * tpg + (m >> 4) * ntiles) * 16 + (m & 15)] = bsum;
        }
        return;
    }
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
        const int tl = s * WG_WAVES + w;
        if (tl < ntot) {
            const int g = g_lo + tl / tpg, rem = tl % tpg, mt = rem / ntiles, nt = rem % ntiles;
            const int q = (nt0 + nt) * 16 + l15;
            if (q < nexp) {
                const int tap = q / ng, i = q % ng;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mt * 16 + 4 * g4 + r;
                    if (m < mv && i < nv) atomicAdd(a.dW + ((size_t)(g * mv + m) * nv + i) * a.taps + tap, acc[s][r]);
                }
            }
        }
    }
    if (do_bias && tid < rowsA) {
        const int g = g_lo + tid / mgp, m = tid % mgp;
        if (m < mv) atomicAdd(a.dbias + (size_t)g * mv + m, bsum);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// bf16 fast path: ROW-MAJOR LDS images + transposing LDS reads (ds_read_b64_tr_b16).  Staging is a plain 16-byte copy
// (global row -> LDS row, one image per tap with out-of-range rows zeroed); an MFMA operand with K = tokens is two
// transposing reads of a 4-token x 16-channel block each.  No register transposes, no scalar LDS writes.
// Row stride = (cols + pad) elements with stride == 16 (mod 32), i.e. an odd multiple of 32 bytes: the 8 rows x 32 bytes a 32-lane
// service group touches then tile all 64 banks.
NBSS_HD int tr_ld(int cols) {
    int ld = cols + 16;
    while (ld % 32 != 16) ld += 8;
    return ld;
}

// Third generation: 64-token chunks and ONE shared X image for all taps of a T-conv.  Chunks of a tapped problem are aligned to
// (b,f) rows — tokens t0-h .. t0+63+h of one row, zero outside [0,T) — so tap d is just a row offset of d in the image and X is
// fetched and written to LDS once instead of once per tap (tr2: 6 vectors per thread per 32 tokens; here 6 per 64).  Tiles are
// ordered nt-major so the bias column sums only exist in the first slots of a wave.
// F-convs (taps along F = a stride of T tokens) use the same kernel with chunks of KC/2 frequencies x 2 adjacent frames of one
// batch item: image row 2 fo + tt holds token (f0 + fo, t0 + tt), a tap is a row offset of 2, and every global access is a
// 16-byte piece of a 384-byte (2-frame) run.
// Knock-out probes (tools/wgrad_probe.py): compiled only into the diagnostic build (python -m nbss_amd.build phase)
#ifdef NBSS_PHASE_PROF
#define WG_PROBE(bit) (a.dbg & (bit))
#define WG_PROBE_HOST(args, bit) ((args).dbg & (bit))
#else
#define WG_PROBE(bit) false
#define WG_PROBE_HOST(args, bit) false
#endif
#define W3_MAXV 7
#define W3_BS 3  // slots that can hold nt == 0 tiles (ngrp * mtiles <= 8 * W3_BS)

// NS = tile slots per wave (compile time, all executed: slots past the last tile contract a dummy tile that is never flushed).
// The MFMA section has no branches, so the LDS reads of the following tiles are scheduled ahead of each MFMA; with
// per-slot `if (tile exists)` tests every tile was read -> wait -> MFMA in sequence.
// NS > 14 (all 8 groups of a large T-conv in one workgroup: 216 tiles = 27 slots per wave, 32-token chunks): one workgroup per CU with up to 256
// registers, the per-slot read offsets recomputed from the (wave-uniform) tile index instead of kept in 2 NS registers.
template <int W3_KC, int NS>
__global__ __launch_bounds__(WG_THREADS, NS > 14 ? 1 : 2) void wgrad_tr3_kernel(WgradArgs a) {
    constexpr int NBUF = 2;
    NBSS_LDS(smem);
    typedef bf16_t T;
    const int tid = threadIdx.x, lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id_u();
    const int mg = a.MA / a.groups, ng = a.NB / a.groups;
    const int mv = a.mvalid ? a.mvalid : mg, nv = a.nvalid ? a.nvalid : ng;
    const int mtiles = cdiv(mg, 16), nexp = a.taps * ng, ntiles = cdiv(nexp, 16);
    const bool per_group = gridDim.y > 1;
    const int g_lo = per_group ? blockIdx.y : 0, ngrp = per_group ? 1 : a.groups;
    const int acols0 = g_lo * mg, ncolsA = ngrp * mg, bcols0 = g_lo * ng, ncolsB = ngrp * ng;
    const int lda = tr_ld(ncolsA), ldb = tr_ld(ncolsB);
    const bool fmode = a.taps > 1 && a.shift_dim == 1;
    const int rs = fmode ? 2 : 1;  // image rows per tap step
    const int h = a.taps / 2, rowsB = W3_KC + 2 * h * rs;
    const int imgA = W3_KC * lda, img = imgA + rowsB * ldb;  // elements per buffer
    T* base = reinterpret_cast<T*>(smem);
    for (int i = tid; i < NBUF * img / 2; i += WG_THREADS) reinterpret_cast<uint32_t*>(base)[i] = 0u;
    float* lnp = reinterpret_cast<float*>(base + (size_t)NBUF * img);  // [2 NB] LayerNorm gamma | beta of the X operand
    if (a.stats)
        for (int i = tid; i < 2 * a.NB; i += WG_THREADS) lnp[i] = i < a.NB ? a.gamma[i] : a.beta[i - a.NB];

    f32x4 acc[NS], bacc[W3_BS];  // bacc: column sums of the dY tiles (bias gradients) = dY^T * ones, on the otherwise idle MFMA pipe
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = F32X4_ZERO;
#pragma unroll
    for (int s = 0; s < W3_BS; ++s) bacc[s] = F32X4_ZERO;
    Frag<T> ones;
#pragma unroll
    for (int jq = 0; jq < 8; ++jq) frag_set(ones, jq, 1.0f);
    const bool do_bias = a.dbias != nullptr;
    const T* Ag = reinterpret_cast<const T*>(a.A);
    const T* Bg = reinterpret_cast<const T*>(a.B);
    const int pA = ncolsA / 8, pB = ncolsB / 8;
    const int nvA = W3_KC * pA, nvec = nvA + rowsB * pB;  // 16-byte vectors of the dY image / of both images
    const int nfirst = ngrp * mtiles, ntot = nfirst * ntiles;  // tile tl = nt * nfirst + (g * mtiles + mt)

    // per-slot offsets (elements, inside a buffer) of this lane's transposing reads
    constexpr bool GW = NS > 14;  // wave = group variant: wave w owns the 3 x 9 tiles of group w, tile (mt, nt) in accumulator nt * 3 + mt
    int oa[GW ? 3 : NS], ob[GW ? 9 : NS];
    const int trow = 4 * g4 + (l15 >> 2), tcol = 4 * (l15 & 3);
    if constexpr (GW) {
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) oa[mt] = trow * lda + w * mg + mt * 16 + tcol;
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) {
            const int q0 = nt * 16 + tcol;  // (9 x 16 = taps x ng exactly: no padding columns)
            ob[nt] = imgA + ((q0 / ng) * rs + trow) * ldb + w * ng + q0 % ng;
        }
    } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int tl = s * WG_WAVES + w;
            oa[s] = trow * lda + tcol; ob[s] = imgA + trow * ldb + tcol;  // dummy slot: tile 0's operands
            if (tl < ntot) {
                const int nt = tl / nfirst, gm = tl % nfirst, g = gm / mtiles, mt = gm % mtiles;
                oa[s] = trow * lda + g * mg + mt * 16 + tcol;
                int q0 = nt * 16 + tcol;
                if (q0 >= nexp) q0 = 0;  // padding columns of the last tile: any valid address, discarded at the flush
                ob[s] = imgA + ((q0 / ng) * rs + trow) * ldb + g * ng + q0 % ng;  // tap = q0 / ng: image row k + tap rs holds the token tap - h steps away
            }
        }
    }
    // Per-vector descriptors (chunk independent).  Slot u is an A slot for EVERY thread when u < UA and a B slot otherwise (a few
    // lanes of the last slot of each kind idle), so the source pointer of a slot is wave-uniform: the load is "scalar base +
    // 32-bit lane offset" with no per-lane 64-bit address arithmetic.  Offsets are made non-negative against the first image row.
    const int rsA = a.a_gw ? a.a_gw : a.lda, rsB = a.b_gw ? a.b_gw : a.ldb;  // global row strides
    const int UA = cdiv(nvA, WG_THREADS);
    int vkk[W3_MAXV], vtok[W3_MAXV], vdst[W3_MAXV];
    unsigned vgo[W3_MAXV];
    bool vt1[W3_MAXV], vok[W3_MAXV];
    const int tokB0 = fmode ? -h * a.T : -h;  // token offset of image row 0 of X
#pragma unroll
    for (int u = 0; u < W3_MAXV; ++u) {
        const bool isA = u < UA;
        const int v = tid + (isA ? u : u - UA) * WG_THREADS;
        vok[u] = isA ? v < nvA : v < nvec - nvA;
        const int pp = isA ? pA : pB, r = vok[u] ? v / pp : 0, c8 = vok[u] ? v % pp : 0;
        const int col = (isA ? acols0 : bcols0) + 8 * c8, gw = isA ? a.a_gw : a.b_gw, gs = isA ? a.a_gs : a.b_gs, rsg = isA ? rsA : rsB;
        const int hh = isA ? 0 : h;
        vkk[u] = fmode ? r / 2 - hh : r - hh;
        vt1[u] = fmode && (r & 1);
        vtok[u] = fmode ? (r / 2 - hh) * a.T + (r & 1) : r - hh;
        vgo[u] = (unsigned)((vtok[u] - (isA ? 0 : tokB0)) * rsg + (gw ? (col / gw) * gs + col % gw : col));
        vdst[u] = (isA ? 0 : imgA) + r * (isA ? lda : ldb) + 8 * c8;
    }
    const int cpr = cdiv(a.T, W3_KC);
    const int nfc = cdiv(a.F, W3_KC / 2), ntp = cdiv(a.T, 2);  // F mode: frequency chunks and frame pairs per batch item
    const int nchunks = fmode ? (a.Ntok / (a.F * a.T)) * ntp * nfc : a.taps > 1 ? (a.Ntok / a.T) * cpr : cdiv(a.Ntok, W3_KC);
    // chunk index as a mixed-radix counter (c0 < r0, c1 < r1, c2), advanced by gridDim.x without divisions
    const int r0 = fmode ? nfc : a.taps > 1 ? cpr : 1, r1 = fmode ? ntp : 1;
    const int G = gridDim.x, d0 = G % r0, d1 = (G / r0) % r1, d2 = G / r0 / r1;
    int c0 = (int)blockIdx.x % r0, c1 = ((int)blockIdx.x / r0) % r1, c2 = (int)blockIdx.x / r0 / r1;
    u32x4 pre[W3_MAXV];
    float pmu[W3_MAXV], prs[W3_MAXV];
    auto prefetch = [&]() {  // the chunk the counter points at; then advance the counter
        long nbase;
        int lo, span;  // chunk-relative rows (F mode: frequencies) lo <= k < lo + span exist
        bool t1ok = true;  // F mode: the second frame of the pair exists
        if (fmode) {
            const int t0 = 2 * c1, f0 = c0 * (W3_KC / 2);
            nbase = ((long)c2 * a.F + f0) * a.T + t0; lo = -f0; span = a.F; t1ok = t0 + 1 < a.T;
        } else if (a.taps > 1) {
            const int t0 = c0 * W3_KC;
            nbase = ((long)c2 * r1 + c1) * a.T + t0; lo = -t0; span = a.T;
        } else {
            nbase = ((long)c2 * r1 + c1) * W3_KC; lo = 0; span = a.Ntok - (int)nbase;
        }
        c0 += d0; c1 += d1; c2 += d2;
        if (c0 >= r0) { c0 -= r0; ++c1; }
        if (c1 >= r1) { c1 -= r1; ++c2; }
        const T* Ab = Ag + nbase * rsA;
        const T* Bb = Bg + (nbase + tokB0) * rsB;
        const float* sb = a.stats ? a.stats + 2 * nbase : nullptr;
#pragma unroll
        for (int u = 0; u < W3_MAXV; ++u) {
            pre[u] = (u32x4){0, 0, 0, 0};
            pmu[u] = 0.f; prs[u] = 0.f;
            if (!vok[u] || (unsigned)(vkk[u] - lo) >= (unsigned)span || (vt1[u] && !t1ok) || WG_PROBE(4)) continue;
            const T* sbase = u < UA ? Ab : Bb;  // wave-uniform
            pre[u] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(sbase) + (size_t)(vgo[u] * 2u));
            if (!GW && u >= UA && sb) { pmu[u] = sb[2 * vtok[u]]; prs[u] = sb[2 * vtok[u] + 1]; }
        }
    };
    auto stash = [&](T* buf) {
#pragma unroll
        for (int u = 0; u < W3_MAXV; ++u) {
            if (!vok[u] || WG_PROBE(8)) continue;
            u32x4 x = pre[u];
            if (!GW && u >= UA && a.stats) {  // LayerNorm on the fly (rows outside the valid range have rstd = 0 and stay 0)
                float f[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { f[2 * i] = bf2f((bf16_t)(x[i] & 0xFFFF)); f[2 * i + 1] = bf2f((bf16_t)(x[i] >> 16)); }
                float gm[8], bt[8];
                const int col = (int)vgo[u] - (vtok[u] - tokB0) * rsB;  // (LayerNorm operands are always plain row-major)
                load8(lnp + col, gm);
                load8(lnp + a.NB + col, bt);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = prs[u] != 0.f ? (f[e] - pmu[u]) * prs[u] * gm[e] + bt[e] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = pack2bf(f[2 * i], f[2 * i + 1]);
            }
            *reinterpret_cast<u32x4*>(buf + vdst[u]) = x;
        }
    };

    int ch = blockIdx.x;
    if (WG_PROBE(16)) return;  // probe: launch + prologue only
    if (ch < nchunks) prefetch();
    lds_barrier();  // zero fill done
    int b = 0;
    for (; ch < nchunks; ch += gridDim.x) {
        T* buf = base + (size_t)b * img;
        stash(buf);
        lds_barrier();
        if (ch + (int)gridDim.x < nchunks) prefetch();
        // software pipeline over the NS * KH (tile, k-half) steps: the operands of step i+1 are requested before step i's MFMA
        constexpr int KH = W3_KC / 32;
        Frag<T> fa[2], fb[2];
        if (WG_PROBE(2)) { b ^= 1; continue; }  // probe: no MFMA section
        if constexpr (GW) {  // (one 32-token k-step per chunk) 3 dY fragments x 9 X fragments: 12 transposing fragment reads per 27 MFMAs
            Frag<T> ga[3];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                frag_load_tr(ga[mt], buf + oa[mt], lda);
                bacc[mt] = mma(ga[mt], ones, bacc[mt]);
            }
            frag_load_tr(fb[0], buf + ob[0], ldb);
#pragma unroll
            for (int nt = 0; nt < 9; ++nt) {
                if (nt + 1 < 9) frag_load_tr(fb[(nt + 1) & 1], buf + ob[nt + 1], ldb);
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) acc[nt * 3 + mt] = mma(ga[mt], fb[nt & 1], acc[nt * 3 + mt]);
            }
        } else {
        frag_load_tr(fa[0], buf + oa[0], lda);
        frag_load_tr(fb[0], buf + ob[0], ldb);
#pragma unroll
        for (int i = 0; i < NS * KH; ++i) {
            const int s = i / KH, cur = i & 1;
            if (i + 1 < NS * KH) {
                const int s1 = (i + 1) / KH, kh1 = (i + 1) % KH;
                frag_load_tr(fa[cur ^ 1], buf + oa[s1] + kh1 * 32 * lda, lda);
                frag_load_tr(fb[cur ^ 1], buf + ob[s1] + kh1 * 32 * ldb, ldb);
            }
            acc[s] = mma(fa[cur], fb[cur], acc[s]);
            if (s < W3_BS) bacc[s < W3_BS ? s : 0] = mma(fa[cur], ones, bacc[s < W3_BS ? s : 0]);
        }
        }
        if (NBUF == 1) lds_barrier();
        else b ^= 1;
    }

    if (WG_PROBE(1)) return;  // probe: no flush
    if (a.part) {  // partial tiles in fragment order (coalesced 256-byte stores); wgrad_reduce_kernel folds them into dW
        const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        float* pt = a.part + wg * ntot * 256;
        float* pbias = a.part + (size_t)gridDim.y * gridDim.x * ntot * 256 + wg * ntot * 16;
        if constexpr (GW) {
#pragma unroll
            for (int nt = 0; nt < 9; ++nt)
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) {
                    const int tl = nt * nfirst + w * 3 + mt;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pt[((size_t)tl * 4 + r) * 64 + lane] = acc[nt * 3 + mt][r];
                }
            if (do_bias && l15 == 0) {
#pragma unroll
                for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pbias[(w * 3 + mt) * 16 + 4 * g4 + r] = bacc[mt][r];
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int tl = s * WG_WAVES + w;
            if (tl < ntot) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pt[((size_t)tl * 4 + r) * 64 + lane] = acc[s][r];
                if (s < W3_BS && do_bias && tl < nfirst && l15 == 0) {  // every column of bacc holds the sums: rows 4 g4 + r
#pragma unroll
                    for (int r = 0; r < 4; ++r) pbias[tl * 16 + 4 * g4 + r] = bacc[s < W3_BS ? s : 0][r];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int tl = s * WG_WAVES + w;
        if (tl < ntot) {
            const int nt = tl / nfirst, gm = tl % nfirst, g = g_lo + gm / mtiles, mt = gm % mtiles;
            const int q = nt * 16 + l15;
            if (q < nexp) {
                const int tap = q / ng, i = q % ng;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mt * 16 + 4 * g4 + r;
                    if (m < mv && i < nv) atomicAdd(a.dW + ((size_t)(g * mv + m) * nv + i) * a.taps + tap, acc[s][r]);
                }
            }
            if (s < W3_BS && do_bias && nt == 0 && l15 == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mt * 16 + 4 * g4 + r;
                    if (m < mv) atomicAdd(a.dbias + (size_t)g * mv + m, bacc[s < W3_BS ? s : 0][r]);
                }
            }
        }
    }
}


// Second pass of the two-stage flush: ONE block per (tile, accumulator register r, y) folds all x-blocks' partial values of its 64 elements in a fixed
// order — thread (slice, lane) sums its slice of the x-blocks with sixteen loads in flight (fold_strided), the four slices meet in LDS, slice 0 adds them in slice
// order and adds the result to dW with a plain read-modify-write (every dW element belongs to exactly one (tile, register, lane); gradient launches of
// one parameter are stream-ordered).  No float atomics: the parameter gradient is bitwise repeatable.  256-thread blocks on purpose: the fold runs on
// the gradient stream beside the main stream's one-workgroup-per-CU kernels, and a 1024-thread block (one block per whole tile, the first version of
// this fold) waits for a CU to drain (measured: step 663.9 -> 651.4 utt/s).  (Round 4's fold: 8 blocks per tile, each ending in an atomicAdd per element.)
#define WG_RSL 4
static_assert(WG_RSL == FK_RSL, "foldk.h");
// (the body lives in foldk.h: fold.hip's table kernel runs it too)
__global__ __launch_bounds__(64 * WG_RSL) void wgrad_reduce_kernel(WgradArgs a, int xb, int nt_major) {
    NBSS_LDS(smem);
    fk_wgrad_reduce(a, xb, nt_major, (int)blockIdx.x, (int)blockIdx.z, (int)gridDim.z, reinterpret_cast<float*>(smem));
}

// batched: the partial tiles came from the enclosing FoldScope's pool (fold_part) — the pass joins the scope's first stage
static int wgrad_reduce_go(const WgradArgs& a, int ntot, int xb, int ybl, int nt_major, hipStream_t st, bool batched = false) {
    if (batched && g_fold) {
        g_fold->st = st;
        FoldItem it;
        it.kind = FK_WGRAD_REDUCE;
        it.gx = 4 * ntot; it.gy = ybl; it.nblk = it.gx * it.gy;
        it.u.wr.a = a; it.u.wr.xb = xb; it.u.wr.nt_major = nt_major;
        return g_fold->add(1, it);
    }
    NBSS_FOLD_LAUNCH(wgrad_reduce_kernel, dim3(4 * ntot, 1, ybl), dim3(64 * WG_RSL), WG_RSL * 80 * sizeof(float), st, a, xb, nt_major);
    return NBSS_CHECK_LAUNCH();
}
// Inside a FoldScope (fold.h) the partial tiles of a launch come from the scope's pool — they have to outlive the launches that follow, whose partial
// tiles would otherwise land on the same bytes — and the second pass is batched.  A request the pool cannot hold goes out alone, after everything pending.
static int fold_part(WgradArgs& a, size_t need, bool* batched) {
    *batched = false;
    if (!g_fold || !a.part) return NBSS_OK;
    int err;
    void* p = g_fold->alloc(need, &err);
    if (err) return err;
    if (!p) return g_fold->flush();
    a.part = (float*)p;
    *batched = true;
    return NBSS_OK;
}

// second pass alone, for kernels that write partial tiles in wgrad_tr3_kernel's layout themselves (wgrad_g.hip)
int wgrad_reduce_launch(const WgradArgs& a, int ntot, int xb, hipStream_t st) { return wgrad_reduce_go(a, ntot, xb, 1, 1, st); }

template <class T>
static int wgrad_launch_t(const WgradArgs& a_in, hipStream_t st) {
    const WgradArgs& a = a_in;
    const int mg = a.MA / a.groups, ng = a.NB / a.groups;
    const int mtiles = cdiv(mg, 16), ntiles = cdiv(a.taps * ng, 16), tpg = mtiles * ntiles;
    const int cap = WG_WAVES * WG_TPW;
    // one group with more tiles than a workgroup holds (LinearGroup at F = 257: 17 x 17 > 112): ranges of output-column tiles on blockIdx.z
    int nz = 1;
    if (tpg > cap) {
        if (a.taps != 1 || mtiles > cap) return NBSS_EUNSUPPORTED;
        nz = cdiv(ntiles, cap / mtiles);
        nz = cdiv(ntiles, cdiv(ntiles, nz));  // no empty range
    }
    const int ntz = cdiv(ntiles, nz);
    // all groups in one workgroup when their tiles and LDS images fit, else one group per blockIdx.y
    size_t lds_all = (size_t)a.groups * (mtiles + ntiles) * 16 * WG_LD * sizeof(T);
    const bool all = nz == 1 && a.groups * tpg <= cap && lds_all <= 80 * 1024 && mtiles * 16 * a.groups <= WG_THREADS;
    const int ybl = all ? 1 : a.groups;
    const size_t lds = all ? lds_all : (size_t)(mtiles + ntz) * 16 * WG_LD * sizeof(T);
    if (lds > 160 * 1024 || mtiles * 16 > WG_THREADS) return NBSS_EUNSUPPORTED;
    // a staging block must not straddle a group: the per-group widths have to be multiples of the block width
    const bool cw8 = mg % 8 == 0 && ng % 8 == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0 && sizeof(T) == 2;
    const int nchunks = cdiv(a.Ntok, WG_KC);
    const int ncA = all ? a.MA : mg, ncB = all ? a.NB : ng;
    // transposing-read kernels: whole rows are copied in 16-byte pieces (staged widths % 8) and tiles are addressed in
    // 4-channel pieces (group widths % 4, checked by the caller)
    // T-convs whose groups do not fit the 112-tile workgroup together (SpatialNet-large: 8 groups x 27 tiles): as one group per blockIdx.y every
    // workgroup read 96-byte slices of the 768-byte rows and did 7 MFMAs per wave between two barriers (237 us per problem at batch 4, 0.85 TB/s).
    // All groups in one workgroup on full rows: 32-token chunks, 27 tile slots per wave, one workgroup per CU.
    if (sizeof(T) == 2 && !all && nz == 1 && a.taps > 1 && a.shift_dim == 0 && a.shift_stride == 1 && a.Ntok % a.T == 0 && a.groups * tpg <= WG_WAVES * 27 &&
        a.groups == WG_WAVES && mtiles == 3 && ntiles == 9 && a.taps * ng == 144 && a.MA % 8 == 0 && a.NB % 8 == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0 && !a.a_gw && !a.b_gw && !a.stats && a.part && !WG_PROBE_HOST(a, 1)) {
        const int kc = 32, h3 = a.taps / 2, rowsB = kc + 2 * h3;
        const size_t img = ((size_t)kc * tr_ld(a.MA) + (size_t)rowsB * tr_ld(a.NB)) * 2;
        const int nvec = (cdiv(kc * (a.MA / 8), WG_THREADS) + cdiv(rowsB * (a.NB / 8), WG_THREADS)) * WG_THREADS;
        const size_t lds3 = 2 * img + 2 * (size_t)a.NB * sizeof(float);
        const int ntot3 = a.groups * tpg;
        int xb = 256;
        const int nch = (a.Ntok / a.T) * cdiv(a.T, kc);
        if (xb > nch) xb = nch;
        if (nvec <= W3_MAXV * WG_THREADS && lds3 <= 158 * 1024 && (size_t)xb * ntot3 * 272 * sizeof(float) <= WGPART_BYTES) {
            ProfScope ps(PK_WGRAD, st);
            int e3;
            WgradArgs a27 = a;
            bool batched;
            if ((e3 = fold_part(a27, (size_t)xb * ntot3 * 272 * sizeof(float), &batched))) return e3;
            if ((e3 = NBSS_SET_MAX_LDS((wgrad_tr3_kernel<32, 27>), lds3))) return e3;
            NBSS_LAUNCH((wgrad_tr3_kernel<32, 27>), dim3(xb, 1), dim3(WG_THREADS), lds3, st, a27);
            if ((e3 = NBSS_CHECK_LAUNCH())) return e3;
            return wgrad_reduce_go(a27, ntot3, xb, 1, 1, st, batched);
        }
    }
    if (nz == 1 && sizeof(T) == 2 && ncA % 8 == 0 && ncB % 8 == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0) {
        // 64-token chunks, one X image for all taps: dense problems, T-convs and (96-row chunks of 48 frequencies x 2 frames) F-convs
        const bool fmode3 = a.taps > 1 && a.shift_dim == 1 && a.shift_stride == a.T && a.Ntok % (a.F * a.T) == 0;
        const bool tmode3 = a.taps > 1 && a.shift_dim == 0 && a.shift_stride == 1 && a.Ntok % a.T == 0;
        // skinny dense problems (squeeze / unsqueeze: 6 tiles, 13 KB per 64 tokens): 128-token chunks halve the barrier rounds per byte
#ifdef NBSS_WG_KC64
        const bool skinny = false;
#else
        const bool skinny = a.taps == 1 && cdiv((all ? a.groups : 1) * mtiles * ntiles, WG_WAVES) <= 2;
#endif
        const int kc3 = fmode3 ? 96 : skinny ? 128 : 64, h3 = a.taps / 2, rowsB3 = kc3 + 2 * h3 * (fmode3 ? 2 : 1), nfirst3 = (all ? a.groups : 1) * mtiles;
        const size_t img3 = ((size_t)kc3 * tr_ld(ncA) + (size_t)rowsB3 * tr_ld(ncB)) * 2;
        const int nvec3 = (cdiv(kc3 * (ncA / 8), WG_THREADS) + cdiv(rowsB3 * (ncB / 8), WG_THREADS)) * WG_THREADS;  // whole A slots + whole B slots
        const bool gm_ok = (!a.a_gw || a.a_gw % 8 == 0) && (!a.b_gw || (a.b_gw % 8 == 0 && !a.stats));
        if (gm_ok && (a.taps == 1 || tmode3 || fmode3) && nvec3 <= W3_MAXV * WG_THREADS && nfirst3 <= WG_WAVES * W3_BS &&
            2 * img3 + 2 * (size_t)a.NB * sizeof(float) <= 158 * 1024) {
            const int nch3 = fmode3 ? (a.Ntok / (a.F * a.T)) * cdiv(a.T, 2) * cdiv(a.F, kc3 / 2) : tmode3 ? (a.Ntok / a.T) * cdiv(a.T, kc3) : cdiv(a.Ntok, kc3);
            int xb = 256 / ybl;
            if (xb < 16) xb = 16;
            if (xb > nch3) xb = nch3;
            ProfScope ps(PK_WGRAD, st);
            const size_t lds3 = 2 * img3 + 2 * (size_t)a.NB * sizeof(float);
            const int ntot3 = nfirst3 * ntiles;
            WgradArgs a3 = a;
            if ((size_t)ybl * xb * ntot3 * 272 * sizeof(float) > WGPART_BYTES) a3.part = nullptr;
            int e3;
            bool batched;
            if ((e3 = fold_part(a3, (size_t)ybl * xb * ntot3 * 272 * sizeof(float), &batched))) return e3;
            const int need = cdiv(ntot3, WG_WAVES);  // tile slots per wave
#define W3_GO(KC, NS)                                                                             \
    do {                                                                                          \
        if ((e3 = NBSS_SET_MAX_LDS((wgrad_tr3_kernel<KC, NS>), lds3))) return e3;                 \
        NBSS_LAUNCH((wgrad_tr3_kernel<KC, NS>), dim3(xb, ybl), dim3(WG_THREADS), lds3, st, a3);   \
    } while (0)
            if (fmode3) {
                if (need <= 5) W3_GO(96, 5);
                else W3_GO(96, 14);
            } else if (skinny) W3_GO(128, 2);   // squeeze / unsqueeze (6 tiles for 8 waves): no dummy slots
            else if (need <= 2) W3_GO(64, 2);
            else if (need <= 5) W3_GO(64, 5);
            else if (need <= 10) W3_GO(64, 10);
            else W3_GO(64, 14);
#undef W3_GO
            if ((e3 = NBSS_CHECK_LAUNCH())) return e3;
            if (a3.part && !WG_PROBE_HOST(a3, 1)) {
                return wgrad_reduce_go(a3, ntot3, xb, ybl, 1, st, batched);
            }
            return NBSS_OK;
        }
    }
    if (a.a_gw || a.b_gw) return NBSS_EUNSUPPORTED;
#ifdef NBSS_WG_NOSMALL
    const bool small = false;
#else
    const bool small = (all ? a.groups : 1) * mtiles * ntz <= WG_WAVES * 4 && sizeof(T) == 2;
#endif
    int xbl = (small ? 1024 : 384) / (ybl * nz);  // resident workgroups; every x-block ends with one atomicAdd per output element
    if (xbl < 16) xbl = 16;
    if (xbl > nchunks) xbl = nchunks;
    dim3 grid(xbl, ybl, nz), block(WG_THREADS);
    ProfScope ps(PK_WGRAD, st);
    int e;
    WgradArgs ak = a_in;
    const int ntot_k = (all ? a.groups : 1) * tpg;
#ifdef NBSS_WG_ATOMIC_FLUSH
    ak.part = nullptr;
#endif
    if (nz > 1 || (size_t)ybl * xbl * ntot_k * 272 * sizeof(float) > WGPART_BYTES) ak.part = nullptr;  // (column-tile ranges / huge grids: atomic flush)
    bool batched;
    if ((e = fold_part(ak, (size_t)ybl * xbl * ntot_k * 272 * sizeof(float), &batched))) return e;
#define WK_GO(CW, TPW)                                                             \
    do {                                                                           \
        if ((e = NBSS_SET_MAX_LDS((wgrad_kernel<T, CW, TPW>), lds))) return e;     \
        NBSS_LAUNCH((wgrad_kernel<T, CW, TPW>), grid, block, lds, st, ak);         \
    } while (0)
    if (cw8 && small) WK_GO(8, 4);
    else if (cw8) WK_GO(8, WG_TPW);
    else if (small) WK_GO(4, 4);
    else WK_GO(4, WG_TPW);
#undef WK_GO
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    if (ak.part) {
        return wgrad_reduce_go(ak, ntot_k, xbl, ybl, 0, st, batched);
    }
    return NBSS_OK;
}

int wgrad_launch(const WgradArgs& a0, int dtype, hipStream_t st) {
    WgradArgs a = a0;
#ifdef NBSS_PHASE_PROF
    static const int dbg = getenv("NBSS_WG_DEBUG") ? atoi(getenv("NBSS_WG_DEBUG")) : 0;  // tools/wgrad_probe.py knock-out runs
    a.dbg = dbg;
#endif
    if (a.MA % a.groups || a.NB % a.groups) return NBSS_EINVAL;
    const int mg = a.MA / a.groups, ng = a.NB / a.groups;
    if (mg % 4 || ng % 4) return NBSS_EUNSUPPORTED;
    return dtype == NBSS_BF16 ? wgrad_launch_t<bf16_t>(a, st) : wgrad_launch_t<float>(a, st);
}
