// wgrad.hip — weight gradients of every linear map on the SpatialNet hot path.
//
// All of them are the same contraction over tokens n = (b,f,t):
//     dW[o][i][tap] = sum_n dY[n][o] * X[n + (tap - taps/2) * shift][i]          (grouped)
// dense layers have taps = 1; the T-convs shift by one frame (valid while 0 <= t+d < T), the
// F-convs by T rows (valid while 0 <= f+d < F); LinearGroup is "8 groups of F x F" over rows
// (b,t).  X may be LayerNorm(x) applied on the fly from the per-token (mean, rstd) the data-gradient
// kernels emit, so the normalised tensor is never materialised.
//
// One workgroup contracts its share of 32-token chunks: both operands are staged TRANSPOSED in LDS
// ([column][32 tokens], 80-byte rows: conflict-free 16-byte reads), so MFMA A/B fragments (K =
// tokens) are single ds_read_b128's, accumulates up to 8 output tiles per wave in registers across
// all its chunks and finally atomicAdd's them into the fp32 gradient buffer.  Column sums of dY
// (the bias gradients) ride along.
#include "launch.h"
#include "layout.h"
#include "wgrad.h"
#include "prof.h"

#define WG_KC 32
#define WG_LD 40     // LDS row length in elements (32 tokens + pad)
#define WG_TPW 8     // output tiles per wave
#define WG_WAVES 4

template <class T>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    NBSS_LDS(smem);
    const int tid = threadIdx.x, lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const int mg = a.MA / a.groups, ng = a.NB / a.groups;
    const int mv = a.mvalid ? a.mvalid : mg, nv = a.nvalid ? a.nvalid : ng;
    const int mtiles = cdiv(mg, 16), nexp = a.taps * ng, ntiles = cdiv(nexp, 16);
    const int tiles_per_group = mtiles * ntiles;
    const int TPB = WG_WAVES * WG_TPW;
    int grp, tile0, acols0, ncolsA, bcols0, ncolsB;
    if (a.groups > 1) {
        const int bpg = cdiv(tiles_per_group, TPB);
        grp = blockIdx.y / bpg;
        tile0 = (blockIdx.y % bpg) * TPB;
        acols0 = grp * mg; ncolsA = mg;
        bcols0 = grp * ng; ncolsB = ng;
    } else {
        grp = 0;
        tile0 = blockIdx.y * TPB;
        acols0 = 0; ncolsA = a.MA;
        bcols0 = 0; ncolsB = a.NB;
    }
    const int rowsA = mtiles * 16, rowsB = ntiles * 16;
    T* At = reinterpret_cast<T*>(smem);            // [rowsA][WG_LD]
    T* Bt = At + (size_t)rowsA * WG_LD;            // [rowsB][WG_LD]  (expanded columns: tap * ng + i)
    for (int i = tid; i < (rowsA + rowsB) * WG_LD; i += blockDim.x) store1(At + i, 0.f);
    __syncthreads();

    f32x4 acc[WG_TPW];
#pragma unroll
    for (int s = 0; s < WG_TPW; ++s) acc[s] = F32X4_ZERO;
    float bsum[2] = {0.f, 0.f};
    const bool do_bias = a.dbias != nullptr && tile0 == 0;
    const T* Ag = reinterpret_cast<const T*>(a.A);
    const T* Bg = reinterpret_cast<const T*>(a.B);
    const int pcA = ncolsA / 4, pcB = ncolsB / 4, center = a.taps / 2;
    const int nchunks = cdiv(a.Ntok, WG_KC);

    for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const int n0 = ch * WG_KC;
        // ---- stage dY chunk, transposed ----
        for (int i = tid; i < WG_KC * pcA; i += blockDim.x) {
            const int k = i / pcA, pc = i % pcA, n = n0 + k;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (n < a.Ntok) load4(Ag + (size_t)n * a.lda + acols0 + 4 * pc, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) store1(At + (size_t)(4 * pc + e) * WG_LD + k, v[e]);
        }
        // ---- stage X chunk for every tap, transposed (and LayerNorm'ed on the fly) ----
        for (int i = tid; i < a.taps * WG_KC * pcB; i += blockDim.x) {
            const int tap = i / (WG_KC * pcB), r = i % (WG_KC * pcB), k = r / pcB, pc = r % pcB, n = n0 + k;
            const int d = tap - center;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (n < a.Ntok) {
                const int pos = a.shift_dim == 0 ? n % a.T : (n / a.T) % a.F;
                const int lim = a.shift_dim == 0 ? a.T : a.F;
                if (pos + d >= 0 && pos + d < lim) {
                    const size_t ns = (size_t)((long)n + (long)d * a.shift_stride);
                    load4(Bg + ns * a.ldb + bcols0 + 4 * pc, v);
                    if (a.stats) {
                        const float mu = a.stats[2 * ns], rs = a.stats[2 * ns + 1];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int col = bcols0 + 4 * pc + e;
                            v[e] = round_to((v[e] - mu) * rs * a.gamma[col] + a.beta[col], Bg);
                        }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) store1(Bt + (size_t)(tap * ng + 4 * pc + e) * WG_LD + k, v[e]);
        }
        __syncthreads();
        // ---- MFMA: K = the 32 tokens of this chunk ----
#pragma unroll
        for (int s = 0; s < WG_TPW; ++s) {
            const int tl = tile0 + s * WG_WAVES + w;
            if (tl < tiles_per_group) {
                const int mt = tl / ntiles, nt = tl % ntiles;
                Frag<T> fa, fb;
                frag_load(fa, At + (size_t)(mt * 16 + l15) * WG_LD + 8 * g4);
                frag_load(fb, Bt + (size_t)(nt * 16 + l15) * WG_LD + 8 * g4);
                acc[s] = mma(fa, fb, acc[s]);
            }
        }
        if (do_bias) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int col = tid + q * 256;
                if (col < ncolsA) {
                    float v[8], sacc = 0.f;
#pragma unroll
                    for (int k8 = 0; k8 < WG_KC; k8 += 8) {
                        load8(At + (size_t)col * WG_LD + k8, v);
#pragma unroll
                        for (int e = 0; e < 8; ++e) sacc += v[e];
                    }
                    bsum[q] += sacc;
                }
            }
        }
        __syncthreads();
    }

    // ---- flush ----
#pragma unroll
    for (int s = 0; s < WG_TPW; ++s) {
        const int tl = tile0 + s * WG_WAVES + w;
        if (tl < tiles_per_group) {
            const int mt = tl / ntiles, nt = tl % ntiles;
            const int q = nt * 16 + l15;
            if (q < nexp) {
                const int tap = q / ng, i = q % ng;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mt * 16 + 4 * g4 + r;
                    if (m < mv && i < nv) atomicAdd(a.dW + ((size_t)(grp * mv + m) * nv + i) * a.taps + tap, acc[s][r]);
                }
            }
        }
    }
    if (do_bias) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int col = tid + q * 256;
            if (col < ncolsA && (col % mg) < mv) atomicAdd(a.dbias + (size_t)((acols0 + col) / mg) * mv + (col % mg), bsum[q]);
        }
    }
}

int wgrad_launch(const WgradArgs& a, int dtype, hipStream_t st) {
    if (a.MA % a.groups || a.NB % a.groups) return NBSS_EINVAL;
    const int mg = a.MA / a.groups, ng = a.NB / a.groups;
    if (mg % 4 || ng % 4 || mg > 512) return NBSS_EUNSUPPORTED;
    const int mtiles = cdiv(mg, 16), ntiles = cdiv(a.taps * ng, 16);
    const int tpb = WG_WAVES * WG_TPW;
    const int ybl = a.groups > 1 ? a.groups * cdiv(mtiles * ntiles, tpb) : cdiv(mtiles * ntiles, tpb);
    const int rowsA = (a.groups > 1 ? mtiles : cdiv(a.MA, 16)) * 16, rowsB = ntiles * 16;
    const size_t esz = dtype == NBSS_BF16 ? 2 : 4;
    const size_t lds = (size_t)(rowsA + rowsB) * WG_LD * esz;
    if (lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    const int nchunks = cdiv(a.Ntok, WG_KC);
    int xbl = 512 / ybl;
    if (xbl < 8) xbl = 8;
    if (xbl > nchunks) xbl = nchunks;
    dim3 grid(xbl, ybl), block(256);
    ProfScope ps(PK_WGRAD, st);
    int e;
    if (dtype == NBSS_BF16) {
        e = NBSS_SET_MAX_LDS((wgrad_kernel<bf16_t>), lds);
        if (e) return e;
        NBSS_LAUNCH((wgrad_kernel<bf16_t>), grid, block, lds, st, a);
    } else {
        e = NBSS_SET_MAX_LDS((wgrad_kernel<float>), lds);
        if (e) return e;
        NBSS_LAUNCH((wgrad_kernel<float>), grid, block, lds, st, a);
    }
    return NBSS_CHECK_LAUNCH();
}
