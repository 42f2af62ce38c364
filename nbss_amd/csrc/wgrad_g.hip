// wgrad_g.hip — weight gradients of the dense per-token maps of the geometry-generic path (SpatialNet-large: in_proj 576 x 192, out_proj 192 x 192,
// the FFN maps 384 x 192 / 192 x 384; models/arch/SpatialNet.py:93-114), bf16 stream:
//
//   dW[m][k] += sum_n dY[n][m] X[n][k],   dbias[m] += sum_n dY[n][m]          (a contraction over the tokens)
//
// wgrad.hip's kernel gives every workgroup ALL output tiles of a problem (<= 112) with one (dY, X) fragment pair per MFMA, so these problems ran as
// 2 - 4 slices of dY columns that each re-read X (12 launches per layer, 147 TFLOP/s).  Here a workgroup (4 waves) owns one 192 x 96 output tile and a
// share of the 32-token chunks: a wave's 96 x 48 sub-tile takes 6 dY + 3 X transposing fragment reads per 18 MFMAs, both operand slabs go through
// registers (next chunk requested before this one's MFMAs) into double-buffered row-major LDS images ([32 tokens][columns + 16]: the token axis
// becomes the MFMA K dimension through ds_read_b64_tr_b16).
// Partial tiles go out in wgrad_reduce_kernel's layout (one x-block per token share), which folds them into dW / dbias.
#include "launch.h"
#include "layout.h"
#include "wgrad.h"
#include "prof.h"
#include <cstdlib>

#define WD_TM 192                  // dY columns of a tile
#define WD_TK 96                   // X columns of a tile
#define WD_LDA (WD_TM + 16)        // image row strides in elements: == 16 (mod 32), the transposing reads then tile the banks
#define WD_LDB (WD_TK + 16)
#define WD_KC 32                   // tokens per chunk
#define WD_ABYTES (WD_KC * WD_LDA * 2)
#define WD_BBYTES (WD_KC * WD_LDB * 2)
#define WD_STAGE (WD_ABYTES + WD_BBYTES)      // 20 480 bytes = 20 copy instructions of 1 KiB
#define WD_PER_WAVE (WD_STAGE / 1024 / 4)
#define WD_NST 2                   // LDS buffers

int wgrad_reduce_launch(const WgradArgs& a, int ntot, int xb, hipStream_t st);

__global__ __launch_bounds__(256, 2) void wgrad_dense_g_kernel(WgradArgs a, int ktiles, int split) {
    NBSS_LDS(smem);
    typedef bf16_t T;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id_u();
    const int wm = w & 1, wk = w >> 1;
    const int tile = blockIdx.x / split, x = blockIdx.x % split;  // output tile, token share
    const int tm = tile / ktiles, tk = tile % ktiles;
    const T* Ag = reinterpret_cast<const T*>(a.A) + tm * WD_TM;
    const T* Bg = reinterpret_cast<const T*>(a.B) + tk * WD_TK;
    const int nchunks = cdiv(a.Ntok, WD_KC);
    const int mine = nchunks > x ? (nchunks - x + split - 1) / split : 0;

    // copy piece of this lane in instruction q of a stage: the stage is one linear run of 16-byte pieces, rows of 26 (dY: 24 + 2 padding) then rows of 14 (X)
    int prow[WD_PER_WAVE];
    unsigned poff[WD_PER_WAVE];  // element offset inside the operand row block; ~0u: a padding piece
    bool pisa[WD_PER_WAVE];
#pragma unroll
    for (int q = 0; q < WD_PER_WAVE; ++q) {
        const int pc = (w * WD_PER_WAVE + q) * 64 + lane;  // piece index in the stage
        constexpr int PA = WD_LDA / 8, PB = WD_LDB / 8, NA = WD_KC * PA;
        const bool isa = pc < NA;
        const int r = isa ? pc / PA : (pc - NA) / PB, cc = isa ? pc % PA : (pc - NA) % PB;
        pisa[q] = isa;
        prow[q] = r;
        poff[q] = cc < (isa ? WD_TM : WD_TK) / 8 ? (unsigned)(r * (isa ? a.lda : a.ldb) + cc * 8) : ~0u;
    }
    // chunk c of this share -> registers (five 16-byte pieces per thread, every one requested before the first is used), then -> the LDS buffer the
    // MFMA section of the NEXT iteration reads.  (As LDS-DMA the copies would skip the registers, but the compiler orders every transposing read —
    // an intrinsic without a tracked memory operand — behind ALL pending LDS-DMA with vmcnt(0): the ring's prefetch was gone.)
    typedef u32x4 Pre[WD_PER_WAVE];
    Pre preA, preB;  // two chunks in flight: one iteration (~0.25 us of MFMA work) is shorter than a memory round trip
    auto prefetch = [&](int ci, Pre& pre) {
        const long n0 = ((long)x + (long)ci * split) * WD_KC;
        const int valid = a.Ntok - n0 < WD_KC ? (int)(a.Ntok - n0) : WD_KC;  // rows of the chunk that exist (the tensor's last chunk: the rest is zero)
#pragma unroll
        for (int q = 0; q < WD_PER_WAVE; ++q) {
            // (unconditional load of a clamped address + select: loads inside branches make the compiler wait with vmcnt(0), i.e. for BOTH chunks in flight)
            const bool ok = poff[q] != ~0u && prow[q] < valid;
            const u32x4 v = *reinterpret_cast<const u32x4*>((pisa[q] ? Ag + (size_t)n0 * a.lda : Bg + (size_t)n0 * a.ldb) + (ok ? poff[q] : 0u));
            pre[q] = ok ? v : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    auto stash = [&](int buf, const Pre& pre) {
        char* sb = smem + buf * WD_STAGE + w * WD_PER_WAVE * 1024 + lane * 16;
#pragma unroll
        for (int q = 0; q < WD_PER_WAVE; ++q) *reinterpret_cast<u32x4*>(sb + q * 1024) = pre[q];
    };
    f32x4 acc[6][3], bacc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        bacc[i] = F32X4_ZERO;
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = F32X4_ZERO;
    }
    Frag<T> ones;
#pragma unroll
    for (int jq = 0; jq < 8; ++jq) frag_set(ones, jq, 1.0f);
    const bool do_bias = a.dbias != nullptr && tk == 0 && wk == 0;
    const int trow = 4 * g4 + (l15 >> 2), tcol = 4 * (l15 & 3);
    const int oa = trow * WD_LDA + wm * 96 + tcol, ob = trow * WD_LDB + wk * 48 + tcol;
    auto step = [&](int s, Pre& pre) {
        const int buf = s & 1;
        stash(buf, pre);
        lds_barrier();  // (two buffers: the one written here was last read two iterations ago, before the previous barrier)
        if (s + 2 < mine) prefetch(s + 2, pre);
        const T* ia = reinterpret_cast<const T*>(smem + buf * WD_STAGE);
        const T* ib = reinterpret_cast<const T*>(smem + buf * WD_STAGE + WD_ABYTES);
        Frag<T> fa[6], fb[3];
#pragma unroll
        for (int i = 0; i < 6; ++i) frag_load_tr(fa[i], ia + oa + 16 * i, WD_LDA);
#pragma unroll
        for (int j = 0; j < 3; ++j) frag_load_tr(fb[j], ib + ob + 16 * j, WD_LDB);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = mma(fa[i], fb[j], acc[i][j]);
            if (do_bias) bacc[i] = mma(fa[i], ones, bacc[i]);
        }
    };
    if (mine > 0) prefetch(0, preA);
    if (mine > 1) prefetch(1, preB);
    for (int s = 0; s < mine; s += 2) {
        step(s, preA);
        if (s + 1 < mine) step(s + 1, preB);
    }
    // partial tiles in wgrad_reduce_kernel's layout: x-block x, tile tl = nt * mtiles + mt, [r][lane]
    const int mtiles = a.MA / 16, ntot = mtiles * (a.NB / 16);
    float* pt = a.part + (size_t)x * ntot * 256;
    float* pbias = a.part + (size_t)split * ntot * 256 + (size_t)x * ntot * 16;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int mt = tm * (WD_TM / 16) + wm * 6 + i;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int nt = tk * (WD_TK / 16) + wk * 3 + j, tl = nt * mtiles + mt;
#pragma unroll
            for (int r = 0; r < 4; ++r) pt[((size_t)tl * 4 + r) * 64 + lane] = acc[i][j][r];
        }
        if (do_bias && l15 == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pbias[mt * 16 + 4 * g4 + r] = bacc[i][r];
        }
    }
}

// dW[M][K] += dY^T X, dbias[M] += colsum(dY) over Ntok rows; returns NBSS_EUNSUPPORTED for shapes it does not take (the caller falls back to wgrad.hip)
int wgrad_dense_g(const void* A, int lda, int M, const void* B, int ldb, int K, float* dW, float* dbias, long Ntok, float* part, hipStream_t st) {
    // NBSS_WGRAD_TILE=0: the column-slice path of wgrad.hip (A/B knob; tests/test_wgrad_g.py compares the two)
    static const bool off = [] {
        const char* e = getenv("NBSS_WGRAD_TILE");
        return e && e[0] == '0';
    }();
    if (off || !part || M % WD_TM || K % WD_TK || lda % 8 || ldb % 8 || Ntok <= 0) return NBSS_EUNSUPPORTED;
    const int mt = M / WD_TM, kt = K / WD_TK, ntile = mt * kt, ntot = (M / 16) * (K / 16);
    const int nchunks = (int)((Ntok + WD_KC - 1) / WD_KC);
    int split = 512 / ntile;
    if (split > nchunks) split = nchunks;
    while (split > 1 && (size_t)split * ntot * 272 * sizeof(float) > WGPART_BYTES) --split;
    if ((size_t)split * ntot * 272 * sizeof(float) > WGPART_BYTES) return NBSS_EUNSUPPORTED;
    WgradArgs a;
    a.A = A; a.lda = lda; a.MA = M; a.B = B; a.ldb = ldb; a.NB = K;
    a.groups = 1; a.mvalid = 0; a.nvalid = 0; a.taps = 1; a.shift_stride = 1; a.shift_dim = 0;
    a.stats = nullptr; a.gamma = nullptr; a.beta = nullptr;
    a.dW = dW; a.dbias = dbias; a.Ntok = (int)Ntok; a.F = 1; a.T = 1; a.part = part;
    ProfScope ps(PK_WGRAD, st);
    const size_t lds = (size_t)WD_NST * WD_STAGE;
    int e = NBSS_SET_MAX_LDS(wgrad_dense_g_kernel, lds);
    if (e) return e;
    NBSS_LAUNCH(wgrad_dense_g_kernel, dim3(ntile * split), dim3(256), lds, st, a, kt, split);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    return wgrad_reduce_launch(a, ntot, split, st);
}
