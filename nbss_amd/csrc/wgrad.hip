// wgrad.hip — weight gradients of every linear map on the SpatialNet hot path.
//
// All of them are the same contraction over tokens n = (b,f,t):
//     dW[o][i][tap] = sum_n dY[n][o] * X[n + (tap - taps/2) * shift][i]          (grouped)
// dense layers have taps = 1; the T-convs shift by one frame (valid while 0 <= t+d < T), the
// F-convs by T rows (valid while 0 <= f+d < F); LinearGroup is "8 groups of F x F" over rows
// (b,t).  X may be LayerNorm(x) applied on the fly from the per-token (mean, rstd) the data-gradient
// kernels emit, so the normalised tensor is never materialised.
//
// One workgroup contracts its share of 64-token chunks.  Both operands are staged TRANSPOSED in LDS
// ([column][64 tokens], 144-byte rows: conflict-free 16-byte fragment reads): a thread fetches a 4-token x
// 4-channel block with four 8-byte loads, transposes it in registers (free) and writes four 8-byte rows.
// The next chunk's global loads are issued before the current chunk's MFMAs (register prefetch).  Up to 8
// output tiles per wave accumulate in registers across all chunks and are flushed with atomicAdd into the
// fp32 gradient buffer; column sums of dY (the bias gradients) ride along.
#include "launch.h"
#include "layout.h"
#include "wgrad.h"
#include "prof.h"

#define WG_KC 64
#define WG_LD 72     // LDS row length in elements (64 tokens + pad)
#define WG_TPW 8     // output tiles per wave
#define WG_WAVES 4
#define WG_MAXBLK 2  // 4x4 blocks a thread prefetches per chunk (wider operands: the rest is fetched in place)

struct Blk {  // one 4-token x 4-channel block in flight
    float v[4][4];
};

template <class T>
__global__ __launch_bounds__(256, 4) void wgrad_kernel(WgradArgs a) {
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

    f32x4 acc[WG_TPW];
#pragma unroll
    for (int s = 0; s < WG_TPW; ++s) acc[s] = F32X4_ZERO;
    float bsum[2] = {0.f, 0.f};
    const bool do_bias = a.dbias != nullptr && tile0 == 0;
    const T* Ag = reinterpret_cast<const T*>(a.A);
    const T* Bg = reinterpret_cast<const T*>(a.B);
    const int pcA = ncolsA / 4, pcB = ncolsB / 4, center = a.taps / 2;
    const int qB = (WG_KC / 4) * pcB;
    const int nblkA = (WG_KC / 4) * pcA, nblk = nblkA + a.taps * qB;
    const int nchunks = cdiv(a.Ntok, WG_KC);

    // fetch block `bi` of the chunk starting at token n0 into registers (zero outside the valid range)
    auto fetch = [&](int bi, int n0, Blk& blk) {
        if (bi < nblkA) {
            const int q = bi / pcA, pc = bi % pcA;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 4 * q + r;
                if (n < a.Ntok) load4(Ag + (size_t)n * a.lda + acols0 + 4 * pc, blk.v[r]);
                else blk.v[r][0] = blk.v[r][1] = blk.v[r][2] = blk.v[r][3] = 0.f;
            }
        } else {
            const int b2 = bi - nblkA;
            const int tap = b2 / qB, rem = b2 % qB, q = rem / pcB, pc = rem % pcB;
            const int d = tap - center;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 4 * q + r;
                blk.v[r][0] = blk.v[r][1] = blk.v[r][2] = blk.v[r][3] = 0.f;
                if (n < a.Ntok) {
                    const int pos = a.shift_dim == 0 ? n % a.T : (n / a.T) % a.F;
                    const int lim = a.shift_dim == 0 ? a.T : a.F;
                    if (pos + d >= 0 && pos + d < lim) {
                        const size_t ns = (size_t)((long)n + (long)d * a.shift_stride);
                        load4(Bg + ns * a.ldb + bcols0 + 4 * pc, blk.v[r]);
                        if (a.stats) {
                            const float mu = a.stats[2 * ns], rs = a.stats[2 * ns + 1];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int col = bcols0 + 4 * pc + e;
                                blk.v[r][e] = round_to((blk.v[r][e] - mu) * rs * a.gamma[col] + a.beta[col], Bg);
                            }
                        }
                    }
                }
            }
        }
    };
    // write the (register-)transposed block: 4 channels x 4 consecutive tokens
    auto stash = [&](int bi, const Blk& blk) {
        T* dst;
        int q;
        if (bi < nblkA) {
            q = bi / pcA;
            dst = At + (size_t)(4 * (bi % pcA)) * WG_LD;
        } else {
            const int b2 = bi - nblkA;
            const int tap = b2 / qB, rem = b2 % qB;
            q = rem / pcB;
            dst = Bt + (size_t)(tap * ng + 4 * (rem % pcB)) * WG_LD;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) store4(dst + (size_t)e * WG_LD + 4 * q, blk.v[0][e], blk.v[1][e], blk.v[2][e], blk.v[3][e]);
    };

    __syncthreads();  // zero fill done
    for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        // latency is hidden by occupancy (4+ workgroups per CU), not by a register prefetch: keeping blocks in flight
        // across the barrier cost 3x the registers and 2.5x the time (profiles/r01_c_*)
        for (int bi = tid; bi < nblk; bi += 256) {
            Blk t;
            fetch(bi, ch * WG_KC, t);
            stash(bi, t);
        }
        __syncthreads();
        // ---- MFMA: K = the 64 tokens of this chunk (two k-steps) ----
#pragma unroll
        for (int s = 0; s < WG_TPW; ++s) {
            const int tl = tile0 + s * WG_WAVES + w;
            if (tl < tiles_per_group) {
                const int mt = tl / ntiles, nt = tl % ntiles;
#pragma unroll
                for (int ks = 0; ks < WG_KC / 32; ++ks) {
                    Frag<T> fa, fb;
                    frag_load(fa, At + (size_t)(mt * 16 + l15) * WG_LD + ks * 32 + 8 * g4);
                    frag_load(fb, Bt + (size_t)(nt * 16 + l15) * WG_LD + ks * 32 + 8 * g4);
                    acc[s] = mma(fa, fb, acc[s]);
                }
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
    // every x-block ends with one atomicAdd per output element: keep the per-address contention bounded
    int xbl = 768 / ybl;
    if (xbl > 192) xbl = 192;
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
