// gbwd.hip — the backward pass for geometries the fused training kernels are not specialised for (SpatialNet-large: dim_hidden 192,
// dim_ffn 384, dim_squeeze 16, head width 48; configs/SpatialNet.yaml "for large" comments, models/arch/SpatialNet.py:154-171).
//
// The fused backward kernels (fconv.hip, full.hip, mhsa_bwd.hip, tconvffn_s.hip) keep a whole slab / sequence of the small geometry in LDS;
// at twice the widths their images do not fit, so this path is built from geometry-generic pieces instead, one tensor
// pass each, every intermediate in the workspace (288 GB of HBM: a 4-utterance large step keeps ~40 [N][FFN] tensors per block alive
// for microseconds):
//   * tap_gemm      Y[n][o] = sum_tap sum_i X[n + (tap - c) S][i] W[g][tap][o][i] (+ bias, SiLU on load / on store): every per-token linear map,
//                   the grouped convolutions along F and along T, the LinearGroup (rows = (b, t), groups = squeeze channels) AND their data
//                   gradients (same form with the weights re-laid by wprep).  MFMA straight from global memory: weights = A, tokens = N.
//   * row kernels   LayerNorm forward / backward (+ affine gradients), SiLU / PReLU backward, GroupNorm statistics / apply / backward,
//                   the [N][SQ] <-> [B T][SQ][F] transposes of the full-band block
//   * attention     two kernels per (sequence, head): queries as the N dimension (O, row statistics, dQ) and keys as the N dimension (dK, dV)
//   * weight gradients: wgrad.hip's token-contraction kernels, which are generic in their dimensions already
// Semantics and parity: the same oracle functions as the fused path (oracle/spatialnet_ref.py), tests/test_large.py.
#include "launch.h"
#include "layout.h"
#include "prof.h"
#include "wgrad.h"
#include "side.h"
#include "tapgemm.h"
#include "tchain.h"
#include "blocks.h"

#define GB_THREADS 256

// ------------------------------------------------------------------------------------------------------------------------------------
// weights for tap_gemm: [groups][taps][Mp][Kp] of the stream dtype, zero padded (Mp % 16 == 0, Kp % 32 == 0)
enum { WP_LIN_FWD, WP_LIN_DGRAD, WP_CONV_FWD, WP_CONV_DGRAD, WP_LG_FWD, WP_LG_DGRAD };
struct WPrep {
    const float* src;
    void* dst;
    int mode, groups, taps, Mg, Kv, Mp, Kp;  // Mg x Kv valid per (group, tap)
};
template <class T>
NBSS_DEV void gb_wprep_body(const WPrep& p) {
    const long total = (long)p.groups * p.taps * p.Mp * p.Kp;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int k = (int)(e % p.Kp);
        long r = e / p.Kp;
        const int m = (int)(r % p.Mp);
        r /= p.Mp;
        const int tap = (int)(r % p.taps), g = (int)(r / p.taps);
        float v = 0.f;
        if (m < p.Mg && k < p.Kv) {
            switch (p.mode) {
                case WP_LIN_FWD: v = p.src[(long)m * p.Kv + k]; break;                                                   // W[o = m][i = k]
                case WP_LIN_DGRAD: v = p.src[(long)k * p.Mg + m]; break;                                                 // W[o = k][i = m]
                case WP_CONV_FWD: v = p.src[(((long)g * p.Mg + m) * p.Kv + k) * p.taps + tap]; break;                    // W[g Og + o][i][tap]
                case WP_CONV_DGRAD: v = p.src[(((long)g * p.Kv + k) * p.Mg + m) * p.taps + (p.taps - 1 - tap)]; break;   // W[g Og + o = k][i = m][flipped]
                case WP_LG_FWD: v = p.src[((long)g * p.Mg + m) * p.Mg + k]; break;                                       // Wf[g][k' = m][h = k]   (Mg = Kv = F)
                case WP_LG_DGRAD: v = p.src[((long)g * p.Mg + k) * p.Mg + m]; break;                                     // Wf[g][k' = k][h = m]
            }
        }
        store1(reinterpret_cast<T*>(p.dst) + e, v);
    }
}


template <class T>
__global__ void gb_wprep_kernel(WPrep p) { gb_wprep_body<T>(p); }
// several re-lays in one launch (blockIdx.y = descriptor): a block backward re-laid its 3 - 6 weights with one 5-us launch each, 1 060 per large step
#define GB_WPREP_MAX 6
struct WPrepMulti {
    WPrep d[GB_WPREP_MAX];
};
template <class T>
__global__ void gb_wprep_multi_kernel(WPrepMulti m) { gb_wprep_body<T>(m.d[blockIdx.y]); }

// epilogue of one row: 4 output tiles in C layout (lane: outputs 16 i + 4 g4 + r of its row)
template <class T>
NBSS_DEV void gb_tap_store(const TapGemm& p, const f32x4 (&acc)[4], long row, int g, int mc, int g4) {
    const size_t ro = (size_t)row * p.ldy + p.ycol + (size_t)g * p.ygs;
    T* yr = reinterpret_cast<T*>(p.Y) + ro;
    T* y2 = p.Y2 ? reinterpret_cast<T*>(p.Y2) + ro : nullptr;
    const T* da = p.Dact ? reinterpret_cast<const T*>(p.Dact) + ro : nullptr;
    const T* rr = p.R ? reinterpret_cast<const T*>(p.R) + (size_t)row * p.ldr + p.ycol + (size_t)g * p.ygs : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m0 = (mc * 4 + i) * 16 + 4 * g4;
        if (m0 >= p.Mg) continue;
        float o[4], o2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[i][r];
            const bool ok = m0 + r < p.Mg;
            if (p.bias && ok) v += p.bias[(size_t)g * p.bgs + m0 + r];
            if (p.yact) v = silu_f(v);
            if (da && ok) v *= dsilu_f(load1(da + m0 + r));
            if (rr && ok) v = load1(rr + m0 + r) + round_to(v, yr);
            o[r] = v;
            o2[r] = silu_f(round_to(v, yr));  // (the activation of the STORED pre-activation: what a separate pass over Y would compute)
        }
        if (m0 + 3 < p.Mg) {
            store4(yr + m0, o[0], o[1], o[2], o[3]);
            if (y2) store4(y2 + m0, o2[0], o2[1], o2[2], o2[3]);
        } else {
            for (int r = 0; r < 4 && m0 + r < p.Mg; ++r) {
                store1(yr + m0 + r, o[r]);
                if (y2) store1(y2 + m0 + r, o2[r]);
            }
        }
    }
}

// One wave = 16 rows (the MFMA N dimension) x up to 64 outputs (4 tiles of 16) of one group; operands come straight from global memory
// (B: 8 contiguous inputs of the lane's row; A: 8 contiguous prepared weights of the lane's output row — L2-resident, every wave reads
// the same few KB).  No LDS, no staging: this is the simple generic path, not the speed-of-light one.
template <class T>
__global__ __launch_bounds__(GB_THREADS) void gb_tap_gemm_kernel(TapGemm p) {
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const int mchunks = cdiv(p.Mp, 64);
    const int g = blockIdx.y / mchunks, mc = blockIdx.y % mchunks;
    const long row = ((long)blockIdx.x * (GB_THREADS / 64) + w) * 16 + l15;
    const bool rv = row < p.rows;
    const int pos = rv ? (int)((row / p.pos_div) % p.pos_len) : 0;
    const T* X = reinterpret_cast<const T*>(p.X);
    const T* Wg = reinterpret_cast<const T*>(p.W) + (size_t)g * p.taps * p.Mp * p.Kp;
    f32x4 acc[4] = {F32X4_ZERO, F32X4_ZERO, F32X4_ZERO, F32X4_ZERO};
    for (int tap = 0; tap < p.taps; ++tap) {
        const int d = tap - p.center;
        const bool valid = rv && pos + d >= 0 && pos + d < p.pos_len;
        const T* xr = X + (size_t)(valid ? row + (long)d * p.shift : 0) * p.ldx + p.xcol + (size_t)g * p.xgs;
        for (int k0 = 0; k0 < p.Kp; k0 += 32) {
            const int kk = k0 + 8 * g4;
            Frag<T> b;
            if (valid && kk < p.Kg) {
                frag_load(b, xr + kk);
                if (p.xact) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) frag_set(b, j, silu_f(frag_get(b, j)));
                }
            } else {
                frag_zero(b);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m0 = (mc * 4 + i) * 16;
                if (m0 < p.Mp) {
                    Frag<T> a;
                    frag_load(a, Wg + ((size_t)tap * p.Mp + m0 + l15) * p.Kp + kk);
                    acc[i] = mma(a, b, acc[i]);
                }
            }
        }
    }
    if (rv) gb_tap_store<T>(p, acc, row, g, mc, g4);
}

// The same contraction with the weights of the workgroup's (group, 64-output chunk) staged in LDS for all taps and GT_R row tiles per wave:
// without it every wave re-read its 64 x K weight block from L2 for 16 rows of work (25.8 % of the large train step).  Row stride of the image:
// Kp + 8 elements (a multiple of 16 bytes that is not a multiple of 128: the 16 rows of a fragment read spread over the banks).
#define GT_R 4
template <class T>
__global__ __launch_bounds__(GB_THREADS) void gb_tap_gemm_lds_kernel(TapGemm p) {
    NBSS_LDS(smem);
    T* Wl = reinterpret_cast<T*>(smem);  // [taps][64][Kp + 8]
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const int mchunks = cdiv(p.Mp, 64);
    const int g = blockIdx.y / mchunks, mc = blockIdx.y % mchunks;
    const int LDW = p.Kp + 8;
    const int mrows = p.Mp - mc * 64 < 64 ? p.Mp - mc * 64 : 64;  // rows of this chunk (a multiple of 16)
    {
        const T* Wg = reinterpret_cast<const T*>(p.W) + (size_t)g * p.taps * p.Mp * p.Kp;
        constexpr int VE = 16 / sizeof(T);  // elements per 16-byte piece
        const int vpr = p.Kp / VE, nv = p.taps * mrows * vpr;
        for (int v = threadIdx.x; v < nv; v += GB_THREADS) {
            const int col = (v % vpr) * VE, r = (v / vpr) % mrows, tap = v / (vpr * mrows);
            *reinterpret_cast<u32x4*>(Wl + ((size_t)tap * 64 + r) * LDW + col) =
                *reinterpret_cast<const u32x4*>(Wg + ((size_t)tap * p.Mp + mc * 64 + r) * p.Kp + col);
        }
    }
    __syncthreads();
    const T* X = reinterpret_cast<const T*>(p.X);
    for (int rt = 0; rt < GT_R; ++rt) {
        const long row = (((long)blockIdx.x * (GB_THREADS / 64) + w) * GT_R + rt) * 16 + l15;
        if ((row - l15) >= p.rows) break;  // (wave-uniform: the tile's first row)
        const bool rv = row < p.rows;
        const int pos = rv ? (int)((row / p.pos_div) % p.pos_len) : 0;
        f32x4 acc[4] = {F32X4_ZERO, F32X4_ZERO, F32X4_ZERO, F32X4_ZERO};
        for (int tap = 0; tap < p.taps; ++tap) {
            const int d = tap - p.center;
            const bool valid = rv && pos + d >= 0 && pos + d < p.pos_len;
            const T* xr = X + (size_t)(valid ? row + (long)d * p.shift : 0) * p.ldx + p.xcol + (size_t)g * p.xgs;
            const T* wt = Wl + (size_t)tap * 64 * LDW + (size_t)l15 * LDW;
#pragma unroll 2
            for (int k0 = 0; k0 < p.Kp; k0 += 32) {
                const int kk = k0 + 8 * g4;
                Frag<T> b;
                if (valid && kk < p.Kg) {
                    frag_load(b, xr + kk);
                    if (p.xact) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) frag_set(b, j, silu_f(frag_get(b, j)));
                    }
                } else {
                    frag_zero(b);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i * 16 < mrows) {
                        Frag<T> a;
                        frag_load(a, wt + (size_t)i * 16 * LDW + kk);
                        acc[i] = mma(a, b, acc[i]);
                    }
                }
            }
        }
        if (rv) gb_tap_store<T>(p, acc, row, g, mc, g4);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// row kernels: one wave per row, lanes over the channels (C <= 64 * GB_CPL)
#define GB_CPL 6  // channels per lane: 384 / 64

// LayerNorm over the last dim (eps 1e-5): u = xhat gamma + beta (optional), stats = (mean, rstd)
template <class T>
__global__ __launch_bounds__(GB_THREADS) void gb_ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               T* __restrict__ u, float* __restrict__ stats, long N, int C) {
    const int lane = lane_id();
    const long nw = (long)gridDim.x * (GB_THREADS / 64);
    for (long n = (long)blockIdx.x * (GB_THREADS / 64) + wave_id(); n < N; n += nw) {
        float v[GB_CPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < GB_CPL; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < C ? load1(x + n * C + c) : 0.f;
            s += v[i];
        }
        const float mean = wave_sum64(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < GB_CPL; ++i) {
            const int c = lane + 64 * i;
            const float d = c < C ? v[i] - mean : 0.f;
            q += d * d;
        }
        const float rstd = rsqrtf(wave_sum64(q) / C + 1e-5f);
        if (lane == 0) {
            stats[2 * n] = mean;
            stats[2 * n + 1] = rstd;
        }
        if (u) {
#pragma unroll
            for (int i = 0; i < GB_CPL; ++i) {
                const int c = lane + 64 * i;
                if (c < C) store1(u + n * C + c, (v[i] - mean) * rstd * gamma[c] + beta[c]);
            }
        }
    }
}

// dx = dy + rstd (g - mean(g) - xhat mean(g xhat)), g = du gamma;  dgamma += sum du xhat, dbeta += sum du   (per-lane sums, one atomicAdd per
// (workgroup, channel) at the end)
template <class T>
__global__ __launch_bounds__(GB_THREADS) void gb_ln_bwd_kernel(const T* __restrict__ du, const T* __restrict__ x, const float* __restrict__ stats,
                                                               const float* __restrict__ gamma, const T* __restrict__ dy, T* __restrict__ dx,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta, long N, int C) {
    NBSS_LDS(smem);
    float (*red)[64 * GB_CPL] = reinterpret_cast<float (*)[64 * GB_CPL]>(smem);  // [2][64 GB_CPL]
    const int lane = lane_id();
    for (int i = threadIdx.x; i < 2 * 64 * GB_CPL; i += GB_THREADS) (&red[0][0])[i] = 0.f;
    __syncthreads();
    float dg[GB_CPL], db[GB_CPL];
#pragma unroll
    for (int i = 0; i < GB_CPL; ++i) dg[i] = db[i] = 0.f;
    const long nw = (long)gridDim.x * (GB_THREADS / 64);
    for (long n = (long)blockIdx.x * (GB_THREADS / 64) + wave_id(); n < N; n += nw) {
        const float mean = stats[2 * n], rstd = stats[2 * n + 1];
        float xh[GB_CPL], g[GB_CPL];
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < GB_CPL; ++i) {
            const int c = lane + 64 * i;
            xh[i] = g[i] = 0.f;
            if (c < C) {
                xh[i] = (load1(x + n * C + c) - mean) * rstd;
                const float d = load1(du + n * C + c);
                dg[i] += d * xh[i];
                db[i] += d;
                g[i] = d * gamma[c];
                m1 += g[i];
                m2 += g[i] * xh[i];
            }
        }
        m1 = wave_sum64(m1) / C;
        m2 = wave_sum64(m2) / C;
#pragma unroll
        for (int i = 0; i < GB_CPL; ++i) {
            const int c = lane + 64 * i;
            if (c < C) store1(dx + n * C + c, load1(dy + n * C + c) + rstd * (g[i] - m1 - xh[i] * m2));
        }
    }
#pragma unroll
    for (int i = 0; i < GB_CPL; ++i) {
        atomicAdd(&red[0][lane + 64 * i], dg[i]);
        atomicAdd(&red[1][lane + 64 * i], db[i]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += GB_THREADS) {
        atomicAdd(dgamma + c, red[0][c]);
        atomicAdd(dbeta + c, red[1][c]);
    }
}


// ---- row kernels for widths that are multiples of 64 with fewer lanes per row: a whole wave per 192-wide row spent its time in six-step wave
// reductions (LayerNorm backward: 144 us per launch for a 50 MB tensor; with 16 lanes per row 82 us; 8 lanes and one load burst: below)
// ---- LayerNorm forward / backward with 8 lanes per row (8 rows per wave; lane = 16-byte pieces l7 + 8 k of the row): every load of an iteration is
// independent of its reductions and issued up front — x, du AND dy: the 16-lane version fetched dy after the row sums, a second memory round trip per
// 4 rows (82 us per launch for 200 MB of traffic at batch 4) — and the clamped (not branched) addresses keep them in one burst.
NBSS_DEV float row_sum8(float v) {  // sum over the 8 lanes of a row (lanes sharing l >> 3), result in every lane
#ifdef NBSS_EMU
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
#else
#define NBSS_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    NBSS_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    NBSS_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    NBSS_DPP_ADD(0x141);  // row_half_mirror
#undef NBSS_DPP_ADD
    return v;
#endif
}
template <class T, int NP>  // C = 64 NP
__global__ __launch_bounds__(GB_THREADS) void gb_ln_fwd8_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                T* __restrict__ u, float* __restrict__ stats, long N) {
    constexpr int C = 64 * NP;
    const int lane = lane_id(), l7 = lane & 7, g8 = lane >> 3;
    const long nw = (long)gridDim.x * (GB_THREADS / 64) * 8;
    for (long n0 = ((long)blockIdx.x * (GB_THREADS / 64) + wave_id()) * 8; n0 < N; n0 += nw) {  // (whole-wave loop: row_sum8 is a wave collective)
        const long n = n0 + g8;
        const bool v_ = n < N;
        const T* xr = x + (v_ ? n : N - 1) * C + 8 * l7;
        float v[NP][8];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) load8(xr + 64 * k, v[k]);
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[k][j];
        const float mean = row_sum8(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[k][j] - mean;
                q += d * d;
            }
        const float rstd = rsqrtf(row_sum8(q) * (1.0f / C) + 1e-5f);
        if (v_ && l7 == 0) {
            stats[2 * n] = mean;
            stats[2 * n + 1] = rstd;
        }
        if (u && v_) {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                float gm[8], bt[8], o[8];
                load8(gamma + 64 * k + 8 * l7, gm);
                load8(beta + 64 * k + 8 * l7, bt);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (v[k][j] - mean) * rstd * gm[j] + bt[j];
                store8(u + n * C + 64 * k + 8 * l7, o);
            }
        }
    }
}
template <class T, int NP>
__global__ __launch_bounds__(GB_THREADS) void gb_ln_bwd8_kernel(const T* __restrict__ du, const T* __restrict__ x, const float* __restrict__ stats,
                                                                const float* __restrict__ gamma, const T* __restrict__ dy, T* __restrict__ dx,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ part, long N) {
    constexpr int C = 64 * NP;
    NBSS_LDS(smem);
    float* red = reinterpret_cast<float*>(smem);  // [2][C]
    const int lane = lane_id(), l7 = lane & 7, g8 = lane >> 3;
    for (int i = threadIdx.x; i < 2 * C; i += GB_THREADS) red[i] = 0.f;
    __syncthreads();
    float dg[NP][8], db[NP][8];
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) dg[k][j] = db[k][j] = 0.f;
    const long nw = (long)gridDim.x * (GB_THREADS / 64) * 8;
    for (long n0 = ((long)blockIdx.x * (GB_THREADS / 64) + wave_id()) * 8; n0 < N; n0 += nw) {
        const long n = n0 + g8;
        const bool v_ = n < N;
        const long nc = v_ ? n : N - 1;
        const float mean = stats[2 * nc], rstd = v_ ? stats[2 * nc + 1] : 0.f;
        float xh[NP][8], g[NP][8], yv[NP][8];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            load8(x + nc * C + 64 * k + 8 * l7, xh[k]);
            load8(du + nc * C + 64 * k + 8 * l7, g[k]);
            load8(dy + nc * C + 64 * k + 8 * l7, yv[k]);
        }
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            float gm[8];
            load8(gamma + 64 * k + 8 * l7, gm);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dv = v_ ? g[k][j] : 0.f;
                xh[k][j] = (xh[k][j] - mean) * rstd;
                dg[k][j] += dv * xh[k][j];
                db[k][j] += dv;
                g[k][j] = dv * gm[j];
                m1 += g[k][j];
                m2 += g[k][j] * xh[k][j];
            }
        }
        m1 = row_sum8(m1) * (1.0f / C);
        m2 = row_sum8(m2) * (1.0f / C);
        if (v_) {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = yv[k][j] + rstd * (g[k][j] - m1 - xh[k][j] * m2);
                store8(dx + n * C + 64 * k + 8 * l7, o);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            atomicAdd(&red[64 * k + 8 * l7 + j], dg[k][j]);
            atomicAdd(&red[C + 64 * k + 8 * l7 + j], db[k][j]);
        }
    __syncthreads();
    // part: one row per workgroup, folded by affine_reduce (1 024 workgroups adding to the same 2 C addresses are 1 024-deep chains of same-address
    // atomics: ~50 us behind a kernel whose memory phase takes 30)
    for (int c = threadIdx.x; c < C; c += GB_THREADS) {
        if (part) {
            part[(size_t)blockIdx.x * 2 * C + c] = red[c];
            part[(size_t)blockIdx.x * 2 * C + C + c] = red[C + c];
        } else {
            atomicAdd(dgamma + c, red[c]);
            atomicAdd(dbeta + c, red[C + c]);
        }
    }
}
template <class T, int NQ>
__global__ __launch_bounds__(GB_THREADS) void gb_prelu_bwd4_kernel(const T* __restrict__ a, const T* __restrict__ dy, const float* __restrict__ alpha,
                                                                   T* __restrict__ da, float* __restrict__ dalpha, long N) {
    constexpr int C = 64 * NQ;
    NBSS_LDS(smem);
    float* red = reinterpret_cast<float*>(smem);  // [C]
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    for (int i = threadIdx.x; i < C; i += GB_THREADS) red[i] = 0.f;
    __syncthreads();
    float ds[NQ][4], al[NQ][4];
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ds[i][r] = 0.f;
            al[i][r] = alpha[64 * i + 4 * l15 + r];
        }
    const long nw = (long)gridDim.x * (GB_THREADS / 64) * 4;
    for (long n = ((long)blockIdx.x * (GB_THREADS / 64) + wave_id()) * 4 + g4; n < N; n += nw) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int c = 64 * i + 4 * l15;
            float av[4], dv[4], o[4];
            load4(a + n * C + c, av);
            load4(dy + n * C + c, dv);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[r] = av[r] > 0.f ? dv[r] : dv[r] * al[i][r];
                if (av[r] <= 0.f) ds[i][r] += dv[r] * av[r];
            }
            store4(da + n * C + c, o[0], o[1], o[2], o[3]);
        }
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&red[64 * i + 4 * l15 + r], ds[i][r]);
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += GB_THREADS) atomicAdd(dalpha + c, red[c]);
}

// gout = gin * SiLU'(a)   (dense tensors; gout may be gin)
template <class T>
__global__ void gb_silu_bwd_kernel(const T* __restrict__ a, const T* gin, T* gout, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) store1(gout + i, load1(gin + i) * dsilu_f(load1(a + i)));
}
// PReLU: y = x + (a > 0 ? a : alpha a) is the block output; da = dy (a > 0 ? 1 : alpha[c]), dalpha[c] += sum dy min(a, 0)
template <class T>
__global__ __launch_bounds__(GB_THREADS) void gb_prelu_bwd_kernel(const T* __restrict__ a, const T* __restrict__ dy, const float* __restrict__ alpha,
                                                                  T* __restrict__ da, float* __restrict__ dalpha, long N, int C) {
    NBSS_LDS(smem);
    float* red = reinterpret_cast<float*>(smem);  // [64 GB_CPL]
    const int lane = lane_id();
    for (int i = threadIdx.x; i < 64 * GB_CPL; i += GB_THREADS) red[i] = 0.f;
    __syncthreads();
    float ds[GB_CPL];
#pragma unroll
    for (int i = 0; i < GB_CPL; ++i) ds[i] = 0.f;
    const long nw = (long)gridDim.x * (GB_THREADS / 64);
    for (long n = (long)blockIdx.x * (GB_THREADS / 64) + wave_id(); n < N; n += nw) {
#pragma unroll
        for (int i = 0; i < GB_CPL; ++i) {
            const int c = lane + 64 * i;
            if (c < C) {
                const float av = load1(a + n * C + c), d = load1(dy + n * C + c);
                store1(da + n * C + c, av > 0.f ? d : d * alpha[c]);
                if (av <= 0.f) ds[i] += d * av;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < GB_CPL; ++i) atomicAdd(&red[lane + 64 * i], ds[i]);
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += GB_THREADS) atomicAdd(dalpha + c, red[c]);
}

// [N = (b, f, t)][SQ] -> [(b, t)][SQ][FK] (columns F..FK zero) and back
template <class T>
__global__ void gb_sq_to_f_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int F, int Tn, int SQ, int FK) {
    const long total = (long)B * Tn * SQ * FK;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int f = (int)(e % FK);
        long r = e / FK;
        const int g = (int)(r % SQ);
        r /= SQ;
        const int t = (int)(r % Tn), b = (int)(r / Tn);
        store1(dst + e, f < F ? load1(src + (((long)b * F + f) * Tn + t) * SQ + g) : 0.f);
    }
}
template <class T>
__global__ void gb_f_to_sq_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int F, int Tn, int SQ, int FK) {
    const long total = (long)B * F * Tn * SQ;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int g = (int)(e % SQ);
        long r = e / SQ;
        const int t = (int)(r % Tn);
        r /= Tn;
        const int f = (int)(r % F), b = (int)(r / F);
        store1(dst + e, load1(src + (((long)b * Tn + t) * SQ + g) * FK + f));
    }
}
// fp32 [N][Co] -> stream dtype [N][CP] (zero padded): the decoder's upstream gradient as a tap_gemm / wgrad operand
template <class T>
__global__ void gb_pad_cols_kernel(const float* __restrict__ src, T* __restrict__ dst, long N, int Co, int CP) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < N * CP; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % CP);
        store1(dst + e, c < Co ? src[(e / CP) * Co + c] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// GroupNorm(groups, FFN) over (T x CG) per sequence and group (eps 1e-5): statistics, h = SiLU(xhat gamma + beta), backward
// one workgroup per (sequence, group); thread = (frame lane, channel)
template <class T>
__global__ __launch_bounds__(GB_THREADS) void gb_gn_fwd_kernel(const T* __restrict__ a, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               T* __restrict__ h, float* __restrict__ stats, int Tn, int C, int CG, int act = 1) {
    NBSS_LDS(smem);
    float* red = reinterpret_cast<float*>(smem);  // [8]
    const int G = C / CG, seq = blockIdx.x / G, g = blockIdx.x % G;
    const int M = Tn * CG;
    const T* ab = a + (size_t)seq * Tn * C + g * CG;
    T* hb = h + (size_t)seq * Tn * C + g * CG;
    auto block_sum = [&](float v) -> float {
        v = wave_sum64(v);
        __syncthreads();
        if (lane_id() == 0) red[wave_id()] = v;
        __syncthreads();
        float s = 0.f;
        for (int i = 0; i < GB_THREADS / 64; ++i) s += red[i];
        return s;
    };
    float s = 0.f;
    for (int e = threadIdx.x; e < M; e += GB_THREADS) s += load1(ab + (size_t)(e / CG) * C + e % CG);
    const float mean = block_sum(s) / M;
    float q = 0.f;
    for (int e = threadIdx.x; e < M; e += GB_THREADS) {
        const float d = load1(ab + (size_t)(e / CG) * C + e % CG) - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(block_sum(q) / M + 1e-5f);
    if (stats && threadIdx.x == 0) {
        stats[2 * blockIdx.x] = mean;
        stats[2 * blockIdx.x + 1] = rstd;
    }
    for (int e = threadIdx.x; e < M; e += GB_THREADS) {
        const int c = e % CG;
        const size_t o = (size_t)(e / CG) * C + c;
        const float v = (load1(ab + o) - mean) * rstd * gamma[g * CG + c] + beta[g * CG + c];
        store1(hb + o, act ? silu_f(v) : v);
    }
}
// in: dh = gradient w.r.t. h = SiLU(a4), a4 = xhat gamma + beta; out (in place): gradient w.r.t. the GroupNorm input a; dgamma / dbeta accumulate
template <class T>
__global__ __launch_bounds__(GB_THREADS) void gb_gn_bwd_kernel(const T* __restrict__ a, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, T* __restrict__ dh, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int Tn, int C, int CG) {
    NBSS_LDS(smem);
    float* red = reinterpret_cast<float*>(smem);  // [8]
    float *cg_w = red + 8, *cg_b = cg_w + 64;    // [64] each: CG <= 64
    const int G = C / CG, seq = blockIdx.x / G, g = blockIdx.x % G;
    const int M = Tn * CG;
    const T* ab = a + (size_t)seq * Tn * C + g * CG;
    T* db = dh + (size_t)seq * Tn * C + g * CG;
    const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
    for (int i = threadIdx.x; i < 64; i += GB_THREADS) cg_w[i] = cg_b[i] = 0.f;
    __syncthreads();
    auto block_sum = [&](float v) -> float {
        v = wave_sum64(v);
        __syncthreads();
        if (lane_id() == 0) red[wave_id()] = v;
        __syncthreads();
        float s = 0.f;
        for (int i = 0; i < GB_THREADS / 64; ++i) s += red[i];
        return s;
    };
    // thread = (frame lane tl, channel ch): the channel's affine sums stay in registers over the frames (one LDS atomic per thread at the end;
    // as one LDS atomic per ELEMENT on 48 addresses the kernel took 531 us per launch)
    const int TPC = GB_THREADS / CG, tl = threadIdx.x / CG, ch = threadIdx.x % CG;
    const bool act = tl < TPC;
    const float gm = act ? gamma[g * CG + ch] : 0.f, bt = act ? beta[g * CG + ch] : 0.f;
    float s1 = 0.f, s2 = 0.f, dw = 0.f, dbv = 0.f;
    if (act) {
        for (int t = tl; t < Tn; t += TPC) {
            const size_t o = (size_t)t * C + ch;
            const float xh = (load1(ab + o) - mean) * rstd;
            const float d4 = load1(db + o) * dsilu_f(xh * gm + bt);
            dw += d4 * xh;
            dbv += d4;
            s1 += d4 * gm;
            s2 += d4 * gm * xh;
        }
        atomicAdd(&cg_w[ch], dw);
        atomicAdd(&cg_b[ch], dbv);
    }
    const float m1 = block_sum(s1) / M;
    const float m2 = block_sum(s2) / M;
    if (act) {
        for (int t = tl; t < Tn; t += TPC) {
            const size_t o = (size_t)t * C + ch;
            const float xh = (load1(ab + o) - mean) * rstd;
            const float d4 = load1(db + o) * dsilu_f(xh * gm + bt);
            store1(db + o, rstd * (d4 * gm - m1 - xh * m2));
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < CG; c += GB_THREADS) {
        atomicAdd(dgamma + g * CG + c, cg_w[c]);
        atomicAdd(dbeta + g * CG + c, cg_b[c]);
    }
}


// GroupBatchNorm of the narrow-band conformer (models/arch/NBC2.py:57-145 in the reference; share_along_sequence_dim = False): statistics over the
// F sequences of one utterance x the C features, per frame, always from the input itself (training AND evaluation); per-feature affine, optional SiLU.
// x [B][F][T][C]; one workgroup per (b, t).
template <class T>
__global__ __launch_bounds__(GB_THREADS) void gb_gbn_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y,
                                                            int F, int Tn, int C, float eps, int act) {
    NBSS_LDS(smem);
    float* red = reinterpret_cast<float*>(smem);  // [8]
    const int b = blockIdx.x / Tn, t = blockIdx.x % Tn;
    const size_t base = ((size_t)b * F * Tn + t) * C, fs = (size_t)Tn * C;  // element (f, c) at base + f fs + c
    const int M = F * C;
    auto block_sum = [&](float v) -> float {
        v = wave_sum64(v);
        __syncthreads();
        if (lane_id() == 0) red[wave_id()] = v;
        __syncthreads();
        float s = 0.f;
        for (int i = 0; i < GB_THREADS / 64; ++i) s += red[i];
        return s;
    };
    float s = 0.f;
    for (int e = threadIdx.x; e < M; e += GB_THREADS) s += load1(x + base + (size_t)(e / C) * fs + e % C);
    const float mean = block_sum(s) / M;
    float q = 0.f;
    for (int e = threadIdx.x; e < M; e += GB_THREADS) {
        const float d = load1(x + base + (size_t)(e / C) * fs + e % C) - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(block_sum(q) / M + eps);
    for (int e = threadIdx.x; e < M; e += GB_THREADS) {
        const int c = e % C;
        const size_t o = base + (size_t)(e / C) * fs + c;
        float v = (load1(x + o) - mean) * rstd;
        if (gamma) v = v * gamma[c] + beta[c];
        store1(y + o, act ? silu_f(v) : v);
    }
}

// Backward of the GroupBatchNorm above (+ its optional SiLU): one workgroup per (b, t), statistics recomputed from x.
//   a = xhat gamma + beta, y = act ? SiLU(a) : a;  d4 = dy (act ? SiLU'(a) : 1);  g = d4 gamma
//   dx = rstd (g - mean(g) - xhat mean(g xhat))  over the F x C elements of the frame;  dgamma[c] += sum_f d4 xhat, dbeta[c] += sum_f d4
// dy and dx may alias.  Per-channel sums: registers over the frequencies, one LDS atomic per thread, C global atomics per workgroup.
template <class T>
__global__ __launch_bounds__(GB_THREADS) void gb_gbn_bwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const T* dy, T* dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int F, int Tn, int C,
                                                                float eps, int act) {
    NBSS_LDS(smem);
    float* red = reinterpret_cast<float*>(smem);  // [8]
    float *cw = red + 8, *cb = cw + C;            // [C] each
    const int b = blockIdx.x / Tn, t = blockIdx.x % Tn;
    const size_t base = ((size_t)b * F * Tn + t) * C, fs = (size_t)Tn * C;
    const int M = F * C;
    for (int i = threadIdx.x; i < 2 * C; i += GB_THREADS) cw[i] = 0.f;
    auto block_sum = [&](float v) -> float {
        v = wave_sum64(v);
        __syncthreads();
        if (lane_id() == 0) red[wave_id()] = v;
        __syncthreads();
        float s = 0.f;
        for (int i = 0; i < GB_THREADS / 64; ++i) s += red[i];
        return s;
    };
    float s = 0.f;
    for (int e = threadIdx.x; e < M; e += GB_THREADS) s += load1(x + base + (size_t)(e / C) * fs + e % C);
    const float mean = block_sum(s) / M;
    float q = 0.f;
    for (int e = threadIdx.x; e < M; e += GB_THREADS) {
        const float d = load1(x + base + (size_t)(e / C) * fs + e % C) - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(block_sum(q) / M + eps);
    // thread = (frequency lane fl, channel c): C <= GB_THREADS is not required — channels are walked in rounds of GB_THREADS
    float s1 = 0.f, s2 = 0.f;
    for (int c0 = 0; c0 < C; c0 += GB_THREADS) {
        const int c = c0 + threadIdx.x;
        if (c < C) {
            const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
            float dw = 0.f, dbv = 0.f;
            for (int f = 0; f < F; ++f) {
                const size_t o = base + (size_t)f * fs + c;
                const float xh = (load1(x + o) - mean) * rstd;
                const float d4 = load1(dy + o) * (act ? dsilu_f(xh * gm + bt) : 1.f);
                dw += d4 * xh;
                dbv += d4;
                s1 += d4 * gm;
                s2 += d4 * gm * xh;
            }
            cw[c] = dw;  // (one thread per channel in this round: plain stores)
            cb[c] = dbv;
        }
    }
    const float m1 = block_sum(s1) / M;
    const float m2 = block_sum(s2) / M;
    for (int e = threadIdx.x; e < M; e += GB_THREADS) {
        const int c = e % C;
        const size_t o = base + (size_t)(e / C) * fs + c;
        const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        const float xh = (load1(x + o) - mean) * rstd;
        const float d4 = load1(dy + o) * (act ? dsilu_f(xh * gm + bt) : 1.f);
        store1(dx + o, rstd * (d4 * gm - m1 - xh * m2));
    }
    __syncthreads();
    if (dgamma)
        for (int c = threadIdx.x; c < C; c += GB_THREADS) {
            atomicAdd(dgamma + c, cw[c]);
            atomicAdd(dbeta + c, cb[c]);
        }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Attention backward for one (sequence, head), T <= 256, any head width DH % 8 == 0 (<= 64).  qkv [N][3H] (q | k | v, head h at columns h DH),
// scores = q k^T / sqrt(DH), softmax over the keys, O = P V.
//   kernel Q (queries are the MFMA N dimension; K, V of the head in LDS):  O, lse = log sum exp, D = rowsum(P dP), dQ
//   kernel K (keys are the N dimension; Q, dO of the head in LDS):         dK, dV
// Transposed operands (V^T, K^T, Q^T, dO^T: the token axis as K dimension) are gathered from the row-major LDS images element by element.
#define GA_TMAX 256
NBSS_DEV int ga_perm_k(int g4, int j) { return j < 4 ? 4 * g4 + j : 16 + 4 * g4 + (j - 4); }

// A fragment whose K dimension is the token axis (permuted order: two stacked C tiles), rows = channels 16 mt + l15, from a row-major
// [token][DH] LDS image: bf16 through two transposing reads (ds_read_b64_tr_b16), fp32 element by element
template <class T, int DH>
NBSS_DEV void ga_frag_t(Frag<T>& f, const T* img, int tok0, int mt) {
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    if constexpr (sizeof(T) == 2) {
        frag_load_tr(f, img + (size_t)(tok0 + 4 * g4 + (l15 >> 2)) * DH + 16 * mt + 4 * (l15 & 3), DH);
    } else {
        const int d = 16 * mt + l15;
#pragma unroll
        for (int j = 0; j < 8; ++j) frag_set(f, j, d < DH ? load1(img + (size_t)(tok0 + ga_perm_k(g4, j)) * DH + d) : 0.f);
    }
}
// rows of the head's [Tn][DH] slice of a [N][ld] tensor into a row-major image, zero rows up to TP: 16-byte pieces
template <class T, int DH>
NBSS_DEV void ga_stage(T* img, const T* src, int ld, int Tn, int TP) {
    constexpr int VE = 16 / sizeof(T), PR = DH / VE;
    for (int e = threadIdx.x; e < TP * PR; e += GB_THREADS) {
        const int t = e / PR, pc = e % PR;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (t < Tn) v = *reinterpret_cast<const u32x4*>(src + (size_t)t * ld + pc * VE);
        *reinterpret_cast<u32x4*>(img + (size_t)t * DH + pc * VE) = v;
    }
}

// BWD = false: the forward alone (O; dO / dqkv / lse / Dv are not touched) — the attention of the narrow-band building blocks (nbss_nb_attention_fwd)
template <class T, int DH, bool BWD>
__global__ __launch_bounds__(GB_THREADS) void gb_attn_q_kernel(const T* __restrict__ qkv, const T* __restrict__ dO, T* __restrict__ O, T* __restrict__ dqkv,
                                                               float* __restrict__ lse, float* __restrict__ Dv, int Tn, int H, int heads) {
    constexpr int KS = (DH + 31) / 32, MTD = (DH + 15) / 16, NTM = GA_TMAX / 16;
    NBSS_LDS(smem);
    const int NT = cdiv(Tn, 16), TP = 32 * cdiv(Tn, 32);
    T* Ks = reinterpret_cast<T*>(smem);  // [TP][DH]
    T* Vs = Ks + (size_t)TP * DH;        // [TP][DH]
    const int seq = blockIdx.x, head = blockIdx.y;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const size_t n0 = (size_t)seq * Tn;
    const int ld = 3 * H;
    ga_stage<T, DH>(Ks, qkv + n0 * ld + H + head * DH, ld, Tn, TP);
    ga_stage<T, DH>(Vs, qkv + n0 * ld + 2 * H + head * DH, ld, Tn, TP);
    __syncthreads();
    const float scale = rsqrtf((float)DH);
    for (int qt = w; qt < NT; qt += GB_THREADS / 64) {
        const int q = qt * 16 + l15;
        const bool qv = q < Tn;
        const size_t nq = n0 + (qv ? q : 0);
        Frag<T> qf[KS], dof[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = 32 * ks + 8 * g4;
            frag_zero(dof[ks]);
            if (qv && d0 < DH) {
                frag_load(qf[ks], qkv + nq * ld + head * DH + d0);
                if (BWD) frag_load(dof[ks], dO + nq * H + head * DH + d0);
            } else {
                frag_zero(qf[ks]);
            }
        }
        // S^T and dP^T tiles: rows = keys 16 jt + 4 g4 + r, column = the lane's query
        f32x4 st[NTM], dp[NTM];
        float mx = -3.0e38f;
#pragma unroll
        for (int jt = 0; jt < NTM; ++jt) {
            st[jt] = F32X4_ZERO;
            dp[jt] = F32X4_ZERO;
            if (jt < NT) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int d0 = 32 * ks + 8 * g4;
                    Frag<T> kf, vf;
                    if (d0 < DH) {
                        frag_load(kf, Ks + (size_t)(16 * jt + l15) * DH + d0);
                        frag_load(vf, Vs + (size_t)(16 * jt + l15) * DH + d0);
                    } else {
                        frag_zero(kf);
                        frag_zero(vf);
                    }
                    st[jt] = mma(kf, qf[ks], st[jt]);
                    if (BWD) dp[jt] = mma(vf, dof[ks], dp[jt]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool kv = 16 * jt + 4 * g4 + r < Tn;
                    st[jt][r] = kv ? st[jt][r] * scale : -3.0e38f;
                    mx = fmaxf(mx, st[jt][r]);
                }
            }
        }
        mx = wave_max16(mx);
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NTM; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool kv = jt < NT && 16 * jt + 4 * g4 + r < Tn;
                st[jt][r] = kv ? __expf(st[jt][r] - mx) : 0.f;
                sum += st[jt][r];
            }
        sum = wave_sum16(sum);
        const float inv = 1.0f / sum;
        float dsum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NTM; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st[jt][r] *= inv;  // P^T
                dsum += st[jt][r] * dp[jt][r];
            }
        dsum = wave_sum16(dsum);  // D = rowsum(P dP) = rowsum(dO O)
        if (BWD && qv && g4 == 0) {
            lse[nq * heads + head] = mx + __logf(sum);
            Dv[nq * heads + head] = dsum;
        }
        // O^T = V^T P^T and dQ^T = K^T dS^T: K dimension = keys in pairs of tiles (permuted order of two stacked C tiles)
        f32x4 oacc[MTD], qacc[MTD];
#pragma unroll
        for (int mt = 0; mt < MTD; ++mt) oacc[mt] = qacc[mt] = F32X4_ZERO;
#pragma unroll
        for (int kk = 0; kk < NTM / 2; ++kk) {
            if (2 * kk < NT) {
                f32x4 ds0, ds1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ds0[r] = st[2 * kk][r] * (dp[2 * kk][r] - dsum) * scale;
                    ds1[r] = st[2 * kk + 1][r] * (dp[2 * kk + 1][r] - dsum) * scale;
                }
                Frag<T> pf, dsf;
                frag_from_c2(pf, st[2 * kk], st[2 * kk + 1]);
                frag_from_c2(dsf, ds0, ds1);
#pragma unroll
                for (int mt = 0; mt < MTD; ++mt) {
                    Frag<T> vt, kt;
                    ga_frag_t<T, DH>(vt, Vs, 32 * kk, mt);
                    oacc[mt] = mma(vt, pf, oacc[mt]);
                    if (BWD) {
                        ga_frag_t<T, DH>(kt, Ks, 32 * kk, mt);
                        qacc[mt] = mma(kt, dsf, qacc[mt]);
                    }
                }
            }
        }
        if (qv) {
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                const int d = 16 * mt + 4 * g4;
                if (d < DH) {
                    store4(O + nq * H + head * DH + d, oacc[mt][0], oacc[mt][1], oacc[mt][2], oacc[mt][3]);
                    if (BWD) store4(dqkv + nq * ld + head * DH + d, qacc[mt][0], qacc[mt][1], qacc[mt][2], qacc[mt][3]);
                }
            }
        }
    }
}

template <class T, int DH>
__global__ __launch_bounds__(GB_THREADS) void gb_attn_k_kernel(const T* __restrict__ qkv, const T* __restrict__ dO, T* __restrict__ dqkv,
                                                               const float* __restrict__ lse, const float* __restrict__ Dv, int Tn, int H, int heads) {
    constexpr int KS = (DH + 31) / 32, MTD = (DH + 15) / 16;
    NBSS_LDS(smem);
    const int NT = cdiv(Tn, 16), TP = 32 * cdiv(Tn, 32);
    T* Qs = reinterpret_cast<T*>(smem);   // [TP][DH]
    T* dOs = Qs + (size_t)TP * DH;        // [TP][DH]
    float* ls = reinterpret_cast<float*>(dOs + (size_t)TP * DH);  // [TP] lse | [TP] D
    float* Ds = ls + TP;
    const int seq = blockIdx.x, head = blockIdx.y;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const size_t n0 = (size_t)seq * Tn;
    const int ld = 3 * H;
    ga_stage<T, DH>(Qs, qkv + n0 * ld + head * DH, ld, Tn, TP);
    ga_stage<T, DH>(dOs, dO + n0 * H + head * DH, H, Tn, TP);
    for (int t = threadIdx.x; t < TP; t += GB_THREADS) {
        ls[t] = t < Tn ? lse[(n0 + t) * heads + head] : 0.f;
        Ds[t] = t < Tn ? Dv[(n0 + t) * heads + head] : 0.f;
    }
    __syncthreads();
    const float scale = rsqrtf((float)DH);
    for (int kt = w; kt < NT; kt += GB_THREADS / 64) {
        const int key = kt * 16 + l15;
        const bool kv = key < Tn;
        const size_t nk = n0 + (kv ? key : 0);
        Frag<T> kf[KS], vf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = 32 * ks + 8 * g4;
            if (kv && d0 < DH) {
                frag_load(kf[ks], qkv + nk * ld + H + head * DH + d0);
                frag_load(vf[ks], qkv + nk * ld + 2 * H + head * DH + d0);
            } else {
                frag_zero(kf[ks]);
                frag_zero(vf[ks]);
            }
        }
        f32x4 kacc[MTD], vacc[MTD];
#pragma unroll
        for (int mt = 0; mt < MTD; ++mt) kacc[mt] = vacc[mt] = F32X4_ZERO;
        for (int kk = 0; 2 * kk < NT; ++kk) {
            // S and dP tiles of query tiles 2 kk, 2 kk + 1: rows = queries 16 it + 4 g4 + r, column = the lane's key
            f32x4 pt[2], dst[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int it = 2 * kk + h2;
                f32x4 s = F32X4_ZERO, dpv = F32X4_ZERO;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int d0 = 32 * ks + 8 * g4;
                    Frag<T> qf, dof;
                    if (d0 < DH) {
                        frag_load(qf, Qs + (size_t)(16 * it + l15) * DH + d0);
                        frag_load(dof, dOs + (size_t)(16 * it + l15) * DH + d0);
                    } else {
                        frag_zero(qf);
                        frag_zero(dof);
                    }
                    s = mma(qf, kf[ks], s);
                    dpv = mma(dof, vf[ks], dpv);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 16 * it + 4 * g4 + r;
                    const bool ok = kv && q < Tn;
                    const float p = ok ? __expf(s[r] * scale - ls[q]) : 0.f;
                    pt[h2][r] = p;
                    dst[h2][r] = p * (dpv[r] - Ds[q]) * scale;
                }
            }
            Frag<T> pf, dsf;
            frag_from_c2(pf, pt[0], pt[1]);
            frag_from_c2(dsf, dst[0], dst[1]);
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                Frag<T> dot, qt;
                ga_frag_t<T, DH>(dot, dOs, 32 * kk, mt);
                ga_frag_t<T, DH>(qt, Qs, 32 * kk, mt);
                vacc[mt] = mma(dot, pf, vacc[mt]);
                kacc[mt] = mma(qt, dsf, kacc[mt]);
            }
        }
        if (kv) {
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                const int d = 16 * mt + 4 * g4;
                if (d < DH) {
                    store4(dqkv + nk * ld + H + head * DH + d, kacc[mt][0], kacc[mt][1], kacc[mt][2], kacc[mt][3]);
                    store4(dqkv + nk * ld + 2 * H + head * DH + d, vacc[mt][0], vacc[mt][1], vacc[mt][2], vacc[mt][3]);
                }
            }
        }
    }
}

// Attention forward with Transformer-XL relative positions (the narrow-band conformer NBC: models/arch/NBC.py:106-143 in the reference):
//   score(i, j) = ((q_i + u) . k_j + (q_i + v) . P[i - j + T - 1]) * scale,   P = pos_proj(sinusoid table) [2T - 1][H], u / v per head
// One workgroup per (sequence, head): K, V and the head's P rows in LDS.  The position term of a (16 queries x 16 keys) tile needs the 31 offsets
// i - j; they are two MFMA tiles M[r'][i] = P[rb + r'] . (q_i + v) (queries stay the N dimension), written to 2 KB of wave-private LDS and read back
// along the diagonal r' = i - j + 15 (the reference materialises the whole [T][2T - 1] product and gathers).  T <= 256, DH in {24, 48}.
template <class T, int DH>
__global__ __launch_bounds__(GB_THREADS) void gb_attn_relpos_kernel(const T* __restrict__ qkv, const T* __restrict__ pos, const float* __restrict__ ub,
                                                                    const float* __restrict__ vb, T* __restrict__ O, float scale, int Tn, int H, int heads,
                                                                    const uint32_t* __restrict__ mask, float keep) {
    constexpr int KS = (DH + 31) / 32, MTD = (DH + 15) / 16, NTM = GA_TMAX / 16;
    NBSS_LDS(smem);
    const int NT = cdiv(Tn, 16), TP = 32 * cdiv(Tn, 32), NR = 2 * Tn - 1, RP = 32 * cdiv(NR + 32, 32);
    T* Ks = reinterpret_cast<T*>(smem);   // [TP][DH]
    T* Vs = Ks + (size_t)TP * DH;         // [TP][DH]
    T* Ps = Vs + (size_t)TP * DH;         // [RP][DH] rows 0 .. 2T - 2 = offsets -(T - 1) .. T - 1, zero rows behind
    float* Mb = reinterpret_cast<float*>(Ps + (size_t)RP * DH) + wave_id() * 32 * 16;  // [32 offsets][16 queries] per wave
    const int seq = blockIdx.x, head = blockIdx.y;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const size_t n0 = (size_t)seq * Tn;
    const int ld = 3 * H;
    ga_stage<T, DH>(Ks, qkv + n0 * ld + H + head * DH, ld, Tn, TP);
    ga_stage<T, DH>(Vs, qkv + n0 * ld + 2 * H + head * DH, ld, Tn, TP);
    ga_stage<T, DH>(Ps, pos + head * DH, H, NR, RP);
    __syncthreads();
    for (int qt = w; qt < NT; qt += GB_THREADS / 64) {
        const int q = qt * 16 + l15;
        const bool qv = q < Tn;
        const size_t nq = n0 + (qv ? q : 0);
        Frag<T> qc[KS], qp[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = 32 * ks + 8 * g4;
            frag_zero(qc[ks]);
            frag_zero(qp[ks]);
            if (qv && d0 < DH) {
                float qf[8];
                load8(qkv + nq * ld + head * DH + d0, qf);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    frag_set(qc[ks], j, qf[j] + ub[head * DH + d0 + j]);
                    frag_set(qp[ks], j, qf[j] + vb[head * DH + d0 + j]);
                }
            }
        }
        f32x4 st[NTM];
        float mx = -3.0e38f;
#pragma unroll
        for (int jt = 0; jt < NTM; ++jt) {
            st[jt] = F32X4_ZERO;
            if (jt < NT) {
                // offsets of this tile pair: r' = 0 .. 31 <-> P row rb + r', rb = (16 qt - 16 jt - 15) + (T - 1)
                const int rb = 16 * qt - 16 * jt - 15 + Tn - 1;
                f32x4 m0 = F32X4_ZERO, m1 = F32X4_ZERO;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int d0 = 32 * ks + 8 * g4;
                    Frag<T> kf, p0, p1;
                    frag_zero(kf); frag_zero(p0); frag_zero(p1);
                    if (d0 < DH) {
                        int r0 = rb + l15, r1 = rb + 16 + l15;  // (rows outside the table belong to masked keys: any valid row)
                        r0 = r0 < 0 ? 0 : r0 >= RP ? RP - 1 : r0;
                        r1 = r1 < 0 ? 0 : r1 >= RP ? RP - 1 : r1;
                        frag_load(kf, Ks + (size_t)(16 * jt + l15) * DH + d0);
                        frag_load(p0, Ps + (size_t)r0 * DH + d0);
                        frag_load(p1, Ps + (size_t)r1 * DH + d0);
                    }
                    st[jt] = mma(kf, qc[ks], st[jt]);
                    m0 = mma(p0, qp[ks], m0);
                    m1 = mma(p1, qp[ks], m1);
                }
                wave_lds_sync();  // (the previous tile's reads of Mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Mb[(4 * g4 + r) * 16 + l15] = m0[r];
                    Mb[(16 + 4 * g4 + r) * 16 + l15] = m1[r];
                }
                wave_lds_sync();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool kv = 16 * jt + 4 * g4 + r < Tn;
                    const float pt = Mb[(l15 - (4 * g4 + r) + 15) * 16 + l15];
                    st[jt][r] = kv ? (st[jt][r] + pt) * scale : -3.0e38f;
                    mx = fmaxf(mx, st[jt][r]);
                }
            }
        }
        mx = wave_max16(mx);
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NTM; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool kv = jt < NT && 16 * jt + 4 * g4 + r < Tn;
                st[jt][r] = kv ? __expf(st[jt][r] - mx) : 0.f;
                sum += st[jt][r];
            }
        sum = wave_sum16(sum);
        const float inv = 1.0f / sum;
        // training with attention dropout (NBC.py:137): keep-bits [nseq][heads][T][ceil(T / 32)], bit (j & 31) of word j >> 5 = key j of this query kept;
        // kept probabilities are scaled by keep = 1 / (1 - p)
        const uint32_t* mrow = mask ? mask + (((size_t)seq * heads + head) * Tn + (qv ? q : 0)) * ((Tn + 31) >> 5) : nullptr;
        f32x4 oacc[MTD];
#pragma unroll
        for (int mt = 0; mt < MTD; ++mt) oacc[mt] = F32X4_ZERO;
#pragma unroll
        for (int kk = 0; kk < NTM / 2; ++kk) {
            if (2 * kk < NT) {
                f32x4 a0, a1;
                const uint32_t mw = mrow ? mrow[kk] : 0xFFFFFFFFu;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a0[r] = ((mw >> (4 * g4 + r)) & 1u) ? st[2 * kk][r] * inv * keep : 0.f;
                    a1[r] = ((mw >> (16 + 4 * g4 + r)) & 1u) ? st[2 * kk + 1][r] * inv * keep : 0.f;
                }
                Frag<T> pf;
                frag_from_c2(pf, a0, a1);
#pragma unroll
                for (int mt = 0; mt < MTD; ++mt) {
                    Frag<T> vt;
                    ga_frag_t<T, DH>(vt, Vs, 32 * kk, mt);
                    oacc[mt] = mma(vt, pf, oacc[mt]);
                }
            }
        }
        if (qv) {
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                const int d = 16 * mt + 4 * g4;
                if (d < DH) store4(O + nq * H + head * DH + d, oacc[mt][0], oacc[mt][1], oacc[mt][2], oacc[mt][3]);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------------
// Backward of the relative-position attention (training of the narrow-band conformer NBC; reference models/arch/NBC.py:106-143 under autograd):
//   S_ij = ((q_i + u) . k_j + (q_i + v) . P[i - j + T - 1]) scale,  p = softmax_j S,  pd = dropout(p) (keep-bits, x keep),  o_i = sum_j pd_ij v_j
//   dpd_ij = dO_i . v_j     D_i = sum_j pd_ij dpd_ij     dS_ij = p_ij (m_ij keep dpd_ij - D_i) scale
//   dv_j = sum_i pd_ij dO_i            dk_j = sum_i dS_ij (q_i + u)            dq_i = sum_j dS_ij (k_j + P[i - j + T - 1])
//   du = sum_i sum_j dS_ij k_j         dvb = sum_i sum_j dS_ij P[i - j + T - 1]           dP[r] = sum_{i - j + T - 1 = r} dS_ij (q_i + v)
// Three kernels per (sequence, head), each recomputing the 16 x 16 score tiles it needs in the loop order that keeps ITS accumulators in registers
// (no atomics anywhere: every output has one owner, partial sums over the sequences are folded in a fixed order):
//   rel_bwd_q: a wave owns 16 queries (softmax over the keys in registers, as the forward kernel): lse, D, dq and the per-(sequence, head) sums du / dvb;
//   rel_bwd_k: a wave owns 16 keys: dk, dv (P rebuilt from lse);
//   rel_bwd_r: a wave owns 16 offsets i - j in [16 b, 16 b + 15]: the tiles of the two tile diagonals b and b + 1 that hold them -> dP rows.
// The position term of a score tile is the forward kernel's: two MFMA tiles M[r'][i] = P[rb + r'] . (q_i + v) through 2 KB of wave-private LDS, read
// along the diagonal r' = i - j + 15.  Its transpose in the gradients — dS entries regrouped by offset — goes the other way through a wave-private
// tile E: entries are scattered to (query, offset) resp. (offset, query) positions, read back as MFMA operands.
// P lives in LDS behind 32 zero rows (and ahead of at least 48): row rb + r' always exists, rows outside the table meet zero dS entries.
#define RB_PAD 32
NBSS_HD int rel_rpp(int Tn) { return 32 * cdiv(RB_PAD + 2 * Tn - 1 + 48, 32); }
template <class T, int DH>
NBSS_DEV void rel_stage_pos(T* Ps, const T* __restrict__ pos, int H, int head, int Tn) {
    const int RPP = rel_rpp(Tn);
    for (int i = threadIdx.x; i < RB_PAD * DH; i += GB_THREADS) store1(Ps + i, 0.f);
    ga_stage<T, DH>(Ps + RB_PAD * DH, pos + head * DH, H, 2 * Tn - 1, RPP - RB_PAD);
}
// rows t of the head's q slice + a per-head bias [DH] (q + u, q + v: rounded to the stream dtype like the forward's fragments), zero rows up to TP
template <class T, int DH>
NBSS_DEV void rel_stage_qb(T* img, const T* __restrict__ src, int ld, const float* __restrict__ bias, int Tn, int TP) {
    for (int e = threadIdx.x; e < TP * DH; e += GB_THREADS) {
        const int t = e / DH, d = e % DH;
        store1(img + e, t < Tn ? load1(src + (size_t)t * ld + d) + bias[d] : 0.f);
    }
}
// position term of the score tile (query tile it, key tile jt): M tiles of the offsets rbp .. rbp + 31 (rbp: padded P row of offset i - j = -15 of the tile
// pair) for the 16 queries whose (q + v) fragments are qp -> Mb [32][16] fp32 (wave-private)
template <class T, int DH>
NBSS_DEV void rel_pos_tiles(const T* Ps, int rbp, const Frag<T>* qp, float* Mb) {
    constexpr int KS = (DH + 31) / 32;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    f32x4 m0 = F32X4_ZERO, m1 = F32X4_ZERO;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int d0 = 32 * ks + 8 * g4;
        Frag<T> p0, p1;
        frag_zero(p0); frag_zero(p1);
        if (d0 < DH) {
            frag_load(p0, Ps + (size_t)(rbp + l15) * DH + d0);
            frag_load(p1, Ps + (size_t)(rbp + 16 + l15) * DH + d0);
        }
        m0 = mma(p0, qp[ks], m0);
        m1 = mma(p1, qp[ks], m1);
    }
    wave_lds_sync();  // (the previous tile's reads of Mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Mb[(4 * g4 + r) * 16 + l15] = m0[r];
        Mb[(16 + 4 * g4 + r) * 16 + l15] = m1[r];
    }
    wave_lds_sync();
}
NBSS_DEV bool rel_keep(const uint32_t* mrow, int key) { return !mrow || ((mrow[key >> 5] >> (key & 31)) & 1u); }

struct RelBwd {
    const void *qkv, *pos, *dO;
    const float *ub, *vb;
    const uint32_t* mask;
    void* dqkv;
    float *lse, *Dv, *duv_part, *dpos_part;
    float scale, keep;
    int Tn, H, heads;
};

template <class T, int DH>
__global__ __launch_bounds__(GB_THREADS) void rel_bwd_q_kernel(RelBwd a) {
    constexpr int KS = (DH + 31) / 32, MTD = (DH + 15) / 16, NTM = GA_TMAX / 16, NW = GB_THREADS / 64;
    NBSS_LDS(smem);
    const int Tn = a.Tn, H = a.H, heads = a.heads;
    const int NT = cdiv(Tn, 16), TP = 32 * cdiv(Tn, 32), RPP = rel_rpp(Tn), MW = (Tn + 31) >> 5;
    T* Ks = reinterpret_cast<T*>(smem);   // [TP][DH]
    T* Vs = Ks + (size_t)TP * DH;         // [TP][DH]
    T* Ps = Vs + (size_t)TP * DH;         // [RPP][DH]
    float* Mb = reinterpret_cast<float*>(Ps + (size_t)RPP * DH) + wave_id() * 32 * 16;       // [32 offsets][16 queries] per wave
    T* Eb = reinterpret_cast<T*>(reinterpret_cast<float*>(Ps + (size_t)RPP * DH) + NW * 32 * 16) + wave_id() * 16 * 32;  // [16 queries][32 offsets] per wave
    float* red = reinterpret_cast<float*>(reinterpret_cast<T*>(reinterpret_cast<float*>(Ps + (size_t)RPP * DH) + NW * 32 * 16) + NW * 16 * 32);  // [NW][2][64]
    const int seq = blockIdx.x, head = blockIdx.y;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const size_t n0 = (size_t)seq * Tn;
    const int ld = 3 * H;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* dO = reinterpret_cast<const T*>(a.dO);
    T* dqkv = reinterpret_cast<T*>(a.dqkv);
    ga_stage<T, DH>(Ks, qkv + n0 * ld + H + head * DH, ld, Tn, TP);
    ga_stage<T, DH>(Vs, qkv + n0 * ld + 2 * H + head * DH, ld, Tn, TP);
    rel_stage_pos<T, DH>(Ps, reinterpret_cast<const T*>(a.pos), H, head, Tn);
    __syncthreads();
    const float scale = a.scale, keep = a.keep;
    f32x4 usum[MTD], vsum[MTD];  // sums over this wave's queries of the content / position parts of dq: the (sequence, head) share of du / dvb
#pragma unroll
    for (int mt = 0; mt < MTD; ++mt) usum[mt] = vsum[mt] = F32X4_ZERO;
    for (int qt = w; qt < NT; qt += NW) {
        const int q = qt * 16 + l15;
        const bool qv = q < Tn;
        const size_t nq = n0 + (qv ? q : 0);
        const uint32_t* mrow = a.mask ? a.mask + (((size_t)seq * heads + head) * Tn + (qv ? q : 0)) * MW : nullptr;
        Frag<T> qc[KS], qp[KS], dof[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = 32 * ks + 8 * g4;
            frag_zero(qc[ks]); frag_zero(qp[ks]); frag_zero(dof[ks]);
            if (qv && d0 < DH) {
                float qf[8];
                load8(qkv + nq * ld + head * DH + d0, qf);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    frag_set(qc[ks], j, qf[j] + a.ub[head * DH + d0 + j]);
                    frag_set(qp[ks], j, qf[j] + a.vb[head * DH + d0 + j]);
                }
                frag_load(dof[ks], dO + nq * H + head * DH + d0);
            }
        }
        // S^T and dPd^T tiles: rows = keys 16 jt + 4 g4 + r, column = the lane's query
        f32x4 st[NTM], dp[NTM];
        float mx = -3.0e38f;
#pragma unroll
        for (int jt = 0; jt < NTM; ++jt) {
            st[jt] = F32X4_ZERO;
            dp[jt] = F32X4_ZERO;
            if (jt < NT) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int d0 = 32 * ks + 8 * g4;
                    Frag<T> kf, vf;
                    frag_zero(kf); frag_zero(vf);
                    if (d0 < DH) {
                        frag_load(kf, Ks + (size_t)(16 * jt + l15) * DH + d0);
                        frag_load(vf, Vs + (size_t)(16 * jt + l15) * DH + d0);
                    }
                    st[jt] = mma(kf, qc[ks], st[jt]);
                    dp[jt] = mma(vf, dof[ks], dp[jt]);
                }
                rel_pos_tiles<T, DH>(Ps, RB_PAD + 16 * qt - 16 * jt - 15 + Tn - 1, qp, Mb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool kv = 16 * jt + 4 * g4 + r < Tn;
                    const float pt = Mb[(l15 - (4 * g4 + r) + 15) * 16 + l15];
                    st[jt][r] = kv ? (st[jt][r] + pt) * scale : -3.0e38f;
                    mx = fmaxf(mx, st[jt][r]);
                }
            }
        }
        mx = wave_max16(mx);
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NTM; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool kv = jt < NT && 16 * jt + 4 * g4 + r < Tn;
                st[jt][r] = kv ? __expf(st[jt][r] - mx) : 0.f;
                sum += st[jt][r];
            }
        sum = wave_sum16(sum);
        const float inv = 1.0f / sum;
        float dsum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NTM; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st[jt][r] *= inv;  // p^T
                const bool kp = jt < NT && rel_keep(mrow, 16 * jt + 4 * g4 + r < Tn ? 16 * jt + 4 * g4 + r : 0);
                dp[jt][r] = kp ? dp[jt][r] * keep : 0.f;  // m keep dpd
                dsum += st[jt][r] * dp[jt][r];
            }
        dsum = wave_sum16(dsum);  // D = rowsum(pd dpd) = dO . o
        if (qv && g4 == 0) {
            a.lse[nq * heads + head] = mx + __logf(sum);
            a.Dv[nq * heads + head] = dsum;
        }
        f32x4 qc_acc[MTD], qp_acc[MTD];
#pragma unroll
        for (int mt = 0; mt < MTD; ++mt) qc_acc[mt] = qp_acc[mt] = F32X4_ZERO;
#pragma unroll
        for (int kk = 0; kk < NTM / 2; ++kk) {
            if (2 * kk < NT) {
                f32x4 ds[2];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ds[0][r] = st[2 * kk][r] * (dp[2 * kk][r] - dsum) * scale;
                    ds[1][r] = st[2 * kk + 1][r] * (dp[2 * kk + 1][r] - dsum) * scale;
                }
                Frag<T> dsf;
                frag_from_c2(dsf, ds[0], ds[1]);
#pragma unroll
                for (int mt = 0; mt < MTD; ++mt) {
                    Frag<T> kt;
                    ga_frag_t<T, DH>(kt, Ks, 32 * kk, mt);
                    qc_acc[mt] = mma(kt, dsf, qc_acc[mt]);
                }
                // position part, tile by tile: E[query][offset r' = i - j + 15] = dS -> dq += P[rb + r']^T E
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int jt = 2 * kk + h2;
                    if (jt < NT) {
                        wave_lds_sync();
                        for (int i = lane; i < 16 * 32; i += 64) store1(Eb + i, 0.f);
                        wave_lds_sync();
#pragma unroll
                        for (int r = 0; r < 4; ++r) store1(Eb + l15 * 32 + (l15 - (4 * g4 + r) + 15), ds[h2][r]);
                        wave_lds_sync();
                        Frag<T> ef;
                        frag_load_lo(ef, Eb + l15 * 32 + 4 * g4);
                        frag_load_hi(ef, Eb + l15 * 32 + 16 + 4 * g4);
                        const int rbp = RB_PAD + 16 * qt - 16 * jt - 15 + Tn - 1;
#pragma unroll
                        for (int mt = 0; mt < MTD; ++mt) {
                            Frag<T> pt;
                            ga_frag_t<T, DH>(pt, Ps, rbp, mt);
                            qp_acc[mt] = mma(pt, ef, qp_acc[mt]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MTD; ++mt) {
            const int d = 16 * mt + 4 * g4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                usum[mt][r] += qc_acc[mt][r];
                vsum[mt][r] += qp_acc[mt][r];
            }
            if (qv && d < DH)
                store4(dqkv + nq * ld + head * DH + d, qc_acc[mt][0] + qp_acc[mt][0], qc_acc[mt][1] + qp_acc[mt][1], qc_acc[mt][2] + qp_acc[mt][2],
                       qc_acc[mt][3] + qp_acc[mt][3]);
        }
    }
    // du / dvb share of this (sequence, head): rows d = 16 mt + 4 g4 + r summed over the lanes' queries, then over the waves in wave order
#pragma unroll
    for (int mt = 0; mt < MTD; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float us = row_sum16(usum[mt][r]), vs = row_sum16(vsum[mt][r]);
            if (l15 == 0) {
                red[(w * 2 + 0) * 64 + 16 * mt + 4 * g4 + r] = us;
                red[(w * 2 + 1) * 64 + 16 * mt + 4 * g4 + r] = vs;
            }
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * DH; i += GB_THREADS) {
        const int k = i / DH, d = i % DH;
        float v = 0.f;
        for (int ww = 0; ww < NW; ++ww) v += red[(ww * 2 + k) * 64 + d];
        a.duv_part[(((size_t)seq * heads + head) * 2 + k) * DH + d] = v;
    }
}

// shared by rel_bwd_k / rel_bwd_r: dS and pd of the score tile (queries 16 it + 4 g4 + r, key 16 jt + l15) from the LDS images
template <class T, int DH>
struct RelImg {
    const T *Qu, *Qv, *dOs, *Ps;  // LDS images (Qu / dOs: rel_bwd_k only)
    const T *gq, *gk, *gv, *gdo;  // rel_bwd_r: the head's q / k / v rows (stride 3 H) and dO rows (stride H) of this sequence in global memory (L2-resident:
    const float* ub;              //   five more [T][dh] images would not fit the LDS beside P in the fp32 stream); ub: the head's u bias
    int H;
    const float *ls, *Ds;
    const uint32_t* Mk;  // [Tn][MW] keep-bits of this (sequence, head) or nullptr
    float* Mb;
    int Tn, MW;
    float scale, keep;
};
// KEYS_IN_REGS (rel_bwd_k): kfr / vfr are the key tile's fragments (natural K = d order) and the query-side operands come from the LDS images; else
// (rel_bwd_r) everything but (q + v) is read from global memory
template <class T, int DH, bool KEYS_IN_REGS>
NBSS_DEV void rel_tile_ds(const RelImg<T, DH>& g, int it, int jt, const Frag<T>* kfr, const Frag<T>* vfr, f32x4& pd, f32x4& ds) {
    constexpr int KS = (DH + 31) / 32;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    Frag<T> qvf[KS];
    f32x4 s = F32X4_ZERO, dpv = F32X4_ZERO;
    const int qrow = 16 * it + l15, krow = 16 * jt + l15;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int d0 = 32 * ks + 8 * g4;
        Frag<T> quf, dof, kf, vf;
        frag_zero(quf); frag_zero(dof); frag_zero(qvf[ks]); frag_zero(kf); frag_zero(vf);
        if (d0 < DH) {
            frag_load(qvf[ks], g.Qv + (size_t)qrow * DH + d0);
            if (KEYS_IN_REGS) {
                frag_load(quf, g.Qu + (size_t)qrow * DH + d0);
                frag_load(dof, g.dOs + (size_t)qrow * DH + d0);
            } else {
                if (qrow < g.Tn) {
                    float qf[8];
                    load8(g.gq + (size_t)qrow * 3 * g.H + d0, qf);
#pragma unroll
                    for (int j = 0; j < 8; ++j) frag_set(quf, j, qf[j] + g.ub[d0 + j]);
                    frag_load(dof, g.gdo + (size_t)qrow * g.H + d0);
                }
                if (krow < g.Tn) {
                    frag_load(kf, g.gk + (size_t)krow * 3 * g.H + d0);
                    frag_load(vf, g.gv + (size_t)krow * 3 * g.H + d0);
                }
            }
        }
        s = mma(quf, KEYS_IN_REGS ? kfr[ks] : kf, s);
        dpv = mma(dof, KEYS_IN_REGS ? vfr[ks] : vf, dpv);
    }
    rel_pos_tiles<T, DH>(g.Ps, RB_PAD + 16 * it - 16 * jt - 15 + g.Tn - 1, qvf, g.Mb);
    const int key = 16 * jt + l15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int iloc = 4 * g4 + r, q = 16 * it + iloc;
        const bool ok = key < g.Tn && q < g.Tn;
        const float pt = g.Mb[(iloc - l15 + 15) * 16 + iloc];
        const float p = ok ? __expf((s[r] + pt) * g.scale - g.ls[q]) : 0.f;
        const bool kp = ok && (!g.Mk || ((g.Mk[(size_t)q * g.MW + (key >> 5)] >> (key & 31)) & 1u));
        const float dm = kp ? dpv[r] * g.keep : 0.f;
        pd[r] = kp ? p * g.keep : 0.f;
        ds[r] = p * (dm - g.Ds[q]) * g.scale;
    }
}
template <class T, int DH>
NBSS_DEV void rel_stage_common(const RelBwd& a, int seq, int head, T* Qu, T* Qv, T* dOs, T* Ps, float* ls, float* Ds, uint32_t* Mk) {
    const int Tn = a.Tn, H = a.H, heads = a.heads, TP = 32 * cdiv(Tn, 32) + 16, MW = (Tn + 31) >> 5;
    const size_t n0 = (size_t)seq * Tn;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    if (Qu) rel_stage_qb<T, DH>(Qu, qkv + n0 * 3 * H + head * DH, 3 * H, a.ub + head * DH, Tn, TP);
    rel_stage_qb<T, DH>(Qv, qkv + n0 * 3 * H + head * DH, 3 * H, a.vb + head * DH, Tn, TP);
    if (dOs) ga_stage<T, DH>(dOs, reinterpret_cast<const T*>(a.dO) + n0 * H + head * DH, H, Tn, TP);
    rel_stage_pos<T, DH>(Ps, reinterpret_cast<const T*>(a.pos), H, head, Tn);
    for (int t = threadIdx.x; t < TP; t += GB_THREADS) {
        ls[t] = t < Tn ? a.lse[(n0 + t) * heads + head] : 0.f;
        Ds[t] = t < Tn ? a.Dv[(n0 + t) * heads + head] : 0.f;
    }
    if (a.mask)
        for (int i = threadIdx.x; i < Tn * MW; i += GB_THREADS) Mk[i] = a.mask[((size_t)seq * heads + head) * Tn * MW + i];
}

template <class T, int DH>
__global__ __launch_bounds__(GB_THREADS) void rel_bwd_k_kernel(RelBwd a) {
    constexpr int KS = (DH + 31) / 32, MTD = (DH + 15) / 16, NW = GB_THREADS / 64;
    NBSS_LDS(smem);
    const int Tn = a.Tn, H = a.H;
    const int NT = cdiv(Tn, 16), TP = 32 * cdiv(Tn, 32) + 16, RPP = rel_rpp(Tn), MW = (Tn + 31) >> 5;
    T* Qu = reinterpret_cast<T*>(smem);
    T* Qv = Qu + (size_t)TP * DH;
    T* dOs = Qv + (size_t)TP * DH;
    T* Ps = dOs + (size_t)TP * DH;
    float* ls = reinterpret_cast<float*>(Ps + (size_t)RPP * DH);
    float* Ds = ls + TP;
    float* Mball = Ds + TP;
    uint32_t* Mk = reinterpret_cast<uint32_t*>(Mball + NW * 32 * 16);
    const int seq = blockIdx.x, head = blockIdx.y;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const size_t n0 = (size_t)seq * Tn;
    const int ld = 3 * H;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    T* dqkv = reinterpret_cast<T*>(a.dqkv);
    rel_stage_common<T, DH>(a, seq, head, Qu, Qv, dOs, Ps, ls, Ds, Mk);
    __syncthreads();
    RelImg<T, DH> g = {Qu, Qv, dOs, Ps, nullptr, nullptr, nullptr, nullptr, nullptr, H, ls, Ds, a.mask ? Mk : nullptr, Mball + w * 32 * 16, Tn, MW, a.scale, a.keep};
    for (int kt = w; kt < NT; kt += NW) {
        const int key = kt * 16 + l15;
        const bool kv = key < Tn;
        const size_t nk = n0 + (kv ? key : 0);
        Frag<T> kf[KS], vf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = 32 * ks + 8 * g4;
            frag_zero(kf[ks]); frag_zero(vf[ks]);
            if (kv && d0 < DH) {
                frag_load(kf[ks], qkv + nk * ld + H + head * DH + d0);
                frag_load(vf[ks], qkv + nk * ld + 2 * H + head * DH + d0);
            }
        }
        f32x4 kacc[MTD], vacc[MTD];
#pragma unroll
        for (int mt = 0; mt < MTD; ++mt) kacc[mt] = vacc[mt] = F32X4_ZERO;
        for (int kk = 0; 2 * kk < NT; ++kk) {
            f32x4 pd[2], ds[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) rel_tile_ds<T, DH, true>(g, 2 * kk + h2, kt, kf, vf, pd[h2], ds[h2]);  // (a query tile past NT: all entries masked)
            Frag<T> pf, dsf;
            frag_from_c2(pf, pd[0], pd[1]);
            frag_from_c2(dsf, ds[0], ds[1]);
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                Frag<T> dot, qt;
                ga_frag_t<T, DH>(dot, dOs, 32 * kk, mt);
                ga_frag_t<T, DH>(qt, Qu, 32 * kk, mt);
                vacc[mt] = mma(dot, pf, vacc[mt]);
                kacc[mt] = mma(qt, dsf, kacc[mt]);
            }
        }
        if (kv) {
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) {
                const int d = 16 * mt + 4 * g4;
                if (d < DH) {
                    store4(dqkv + nk * ld + H + head * DH + d, kacc[mt][0], kacc[mt][1], kacc[mt][2], kacc[mt][3]);
                    store4(dqkv + nk * ld + 2 * H + head * DH + d, vacc[mt][0], vacc[mt][1], vacc[mt][2], vacc[mt][3]);
                }
            }
        }
    }
}

// dP rows: wave = offset block b (offsets 16 b .. 16 b + 15, b = -NT .. NT - 1).  They live in the tiles of the diagonals it - jt = b (entries with
// i_loc >= j_loc: offset 16 b + i_loc - j_loc) and it - jt = b + 1 (entries with i_loc < j_loc: offset 16 (b + 1) + i_loc - j_loc).  Two tiles of a diagonal
// at a time: Et[offset][32 queries] (wave-private), dP^T[d][offset] += (q + v)^T[d][32 queries] Et^T — K = the 32 queries.
// Output: dpos_part[seq][head][32 NT offsets rows: offset + 16 NT][DH] (every row has one owner; folded over the sequences by rel_fold_kernel).
template <class T, int DH>
__global__ __launch_bounds__(GB_THREADS) void rel_bwd_r_kernel(RelBwd a) {
    constexpr int KS = (DH + 31) / 32, MTD = (DH + 15) / 16, NW = GB_THREADS / 64;
    NBSS_LDS(smem);
    const int Tn = a.Tn, heads = a.heads;
    const int NT = cdiv(Tn, 16), TP = 32 * cdiv(Tn, 32) + 16, RPP = rel_rpp(Tn), MW = (Tn + 31) >> 5;
    T* Qv = reinterpret_cast<T*>(smem);
    T* Ps = Qv + (size_t)TP * DH;
    float* ls = reinterpret_cast<float*>(Ps + (size_t)RPP * DH);
    float* Ds = ls + TP;
    float* Mball = Ds + TP;
    uint32_t* Mk = reinterpret_cast<uint32_t*>(Mball + NW * 32 * 16);
    T* Et = reinterpret_cast<T*>(Mk + (a.mask ? (Tn * MW + 3) & ~3 : 0)) + wave_id() * 16 * 32;  // (16-byte aligned: vector reads)
    const int seq = blockIdx.x, head = blockIdx.y, H = a.H;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const size_t n0 = (size_t)seq * Tn;
    const T* qkv = reinterpret_cast<const T*>(a.qkv) + n0 * 3 * H + head * DH;
    rel_stage_common<T, DH>(a, seq, head, (T*)nullptr, Qv, (T*)nullptr, Ps, ls, Ds, Mk);
    __syncthreads();
    RelImg<T, DH> g = {nullptr, Qv, nullptr, Ps, qkv, qkv + H, qkv + 2 * H, reinterpret_cast<const T*>(a.dO) + n0 * H + head * DH, a.ub + head * DH, H,
                       ls, Ds, a.mask ? Mk : nullptr, Mball + w * 32 * 16, Tn, MW, a.scale, a.keep};
    float* out = a.dpos_part + ((size_t)seq * heads + head) * 32 * NT * DH;
    for (int b = -NT + w; b < NT; b += NW) {
        f32x4 acc[MTD];
#pragma unroll
        for (int mt = 0; mt < MTD; ++mt) acc[mt] = F32X4_ZERO;
        for (int dg = 0; dg < 2; ++dg) {
            const int delta = b + dg;  // it - jt
            const int it_lo = delta > 0 ? delta : 0, it_hi = delta > 0 ? NT : NT + delta;  // tiles (it, it - delta) with both indices in [0, NT)
            for (int it = it_lo & ~1; it < it_hi; it += 2) {  // pairs (it, it + 1): the K = 32 queries 16 it .. 16 it + 31
                wave_lds_sync();
                for (int i = lane; i < 16 * 32; i += 64) store1(Et + i, 0.f);
                wave_lds_sync();
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int i2 = it + h2, jt = i2 - delta;
                    if (i2 >= it_lo && i2 < it_hi) {
                        f32x4 pd, ds;
                        rel_tile_ds<T, DH, false>(g, i2, jt, nullptr, nullptr, pd, ds);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int iloc = 4 * g4 + r;
                            const int off = dg == 0 ? iloc - l15 : 16 + iloc - l15;  // offset inside the block
                            if (off >= 0 && off < 16) store1(Et + off * 32 + 16 * h2 + iloc, ds[r]);
                        }
                    }
                }
                wave_lds_sync();
                Frag<T> ef;
                frag_load_lo(ef, Et + l15 * 32 + 4 * g4);
                frag_load_hi(ef, Et + l15 * 32 + 16 + 4 * g4);
#pragma unroll
                for (int mt = 0; mt < MTD; ++mt) {
                    Frag<T> qt;
                    ga_frag_t<T, DH>(qt, Qv, 16 * it, mt);
                    acc[mt] = mma(qt, ef, acc[mt]);
                }
            }
        }
        // acc[mt][r]: channel d = 16 mt + 4 g4 + r, offset 16 b + l15
#pragma unroll
        for (int mt = 0; mt < MTD; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 16 * mt + 4 * g4 + r;
                if (d < DH) out[(size_t)(16 * (b + NT) + l15) * DH + d] = acc[mt][r];
            }
    }
}

// fold over the sequences, in sequence order: dpos[r][head DH + d] += sum_seq dpos_part[seq][head][r - (Tn - 1) + 16 NT][d];  du / dvb [head][DH] likewise
__global__ void rel_fold_kernel(const float* __restrict__ dpos_part, const float* __restrict__ duv_part, long nseq, int Tn, int H, int heads, float* __restrict__ dpos,
                                float* __restrict__ du, float* __restrict__ dvb) {
    const int DH = H / heads, NT = cdiv(Tn, 16), NR = 2 * Tn - 1;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x, npos = (long)NR * H;
    if (e < npos) {
        const int r = (int)(e / H), c = (int)(e % H), head = c / DH, d = c % DH;
        const float* p = dpos_part + ((size_t)head * 32 * NT + (r - (Tn - 1) + 16 * NT)) * DH + d;
        const size_t ss = (size_t)heads * 32 * NT * DH;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        long q = 0;
        for (; q + 4 <= nseq; q += 4) {
            s0 += p[q * ss]; s1 += p[(q + 1) * ss]; s2 += p[(q + 2) * ss]; s3 += p[(q + 3) * ss];
        }
        for (; q < nseq; ++q) s0 += p[q * ss];
        dpos[e] += (s0 + s1) + (s2 + s3);
    } else if (e < npos + 2 * H) {
        const int i = (int)(e - npos), k = i / H, c = i % H, head = c / DH, d = c % DH;
        const float* p = duv_part + ((size_t)head * 2 + k) * DH + d;
        const size_t ss = (size_t)heads * 2 * DH;
        float s = 0.f;
        for (long q = 0; q < nseq; ++q) s += p[q * ss];
        (k ? dvb : du)[c] += s;
    }
}

// ====================================================================================================================================
// host side
// ====================================================================================================================================
static int gb_blocks(long n, int per_block) {
    const long b = (n + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : b > 4096 ? 4096 : b);
}
// bump allocator over the sub-block's workspace (behind the per-token statistics at its head; 256-byte aligned pieces)
struct GbArena {
    char* p;
    char* end;
    void* take(size_t bytes) {
        void* r = p;
        p += ws_align(bytes);
        return p <= end ? r : nullptr;
    }
};
static GbArena gb_arena(const nbss_cfg& c, void* ws) {
    const size_t N = (size_t)c.B * c.F * c.T;
    GbArena a;
    a.p = (char*)ws + ws_align(N * 2 * sizeof(float));
    a.end = (char*)ws + ws_part_offset(c);
    return a;
}

template <class T>
static int gb_wprep(const float* src, void* dst, int mode, int groups, int taps, int Mg, int Kv, int Mp, int Kp, hipStream_t st) {
    WPrep p = {src, dst, mode, groups, taps, Mg, Kv, Mp, Kp};
    NBSS_LAUNCH((gb_wprep_kernel<T>), dim3(gb_blocks((long)groups * taps * Mp * Kp, 256)), dim3(256), 0, st, p);
    return NBSS_CHECK_LAUNCH();
}
template <class T>
struct WPrepBatch {
    WPrepMulti m;
    int n = 0;
    long most = 0;
    void add(const float* src, void* dst, int mode, int groups, int taps, int Mg, int Kv, int Mp, int Kp) {
        m.d[n++] = {src, dst, mode, groups, taps, Mg, Kv, Mp, Kp};
        const long el = (long)groups * taps * Mp * Kp;
        most = el > most ? el : most;
    }
    int launch(hipStream_t st) {
        NBSS_LAUNCH((gb_wprep_multi_kernel<T>), dim3(gb_blocks(most, 256), n), dim3(256), 0, st, m);
        return NBSS_CHECK_LAUNCH();
    }
};
static int pad16(int v) { return (v + 15) & ~15; }
static int pad32(int v) { return (v + 31) & ~31; }
static int pad8(int v) { return (v + 7) & ~7; }

template <class T>
static int gb_gemm(const TapGemm& p, hipStream_t st) {
    if (p.Kg % 8 || p.ldx % 8 || p.xcol % 8 || p.xgs % 8 || p.ycol % 4 || p.ygs % 4 || p.ldy % 4) return NBSS_EUNSUPPORTED;
    if (sizeof(T) == 2 && gl_gemm_takes(p)) return gl_gemm_bf16(p, st);
    const size_t lds = (size_t)p.taps * 64 * (p.Kp + 8) * sizeof(T);
    if (lds <= 150 * 1024) {
        int e = NBSS_SET_MAX_LDS((gb_tap_gemm_lds_kernel<T>), lds);
        if (e) return e;
        dim3 grid(cdiv(p.rows, 64 * GT_R), p.groups * cdiv(p.Mp, 64));
        NBSS_LAUNCH((gb_tap_gemm_lds_kernel<T>), grid, dim3(GB_THREADS), lds, st, p);
        return NBSS_CHECK_LAUNCH();
    }
    dim3 grid(cdiv(p.rows, 64), p.groups * cdiv(p.Mp, 64));
    NBSS_LAUNCH((gb_tap_gemm_kernel<T>), grid, dim3(GB_THREADS), 0, st, p);
    return NBSS_CHECK_LAUNCH();
}
// dense per-token linear map (taps = 1, one group): Y[rows][M] = act(X[rows][K] Wp^T + bias) (+ R)
static TapGemm gb_lin(const void* X, int ldx, const void* Wp, const float* bias, void* Y, int ldy, long rows, int M, int K) {
    TapGemm p;
    p.X = X; p.W = Wp; p.bias = bias; p.R = nullptr; p.Y = Y;
    p.rows = (int)rows;
    p.ldx = ldx; p.xcol = 0; p.xgs = 0;
    p.ldy = ldy; p.ycol = 0; p.ygs = 0; p.ldr = 0;
    p.groups = 1; p.Mg = M; p.Kg = K; p.Mp = pad16(M); p.Kp = pad32(K); p.bgs = 0;
    p.taps = 1; p.center = 0; p.shift = 0; p.pos_div = 1; p.pos_len = 1 << 30;
    p.xact = 0; p.yact = 0; p.Y2 = nullptr; p.Dact = nullptr;
    return p;
}
// grouped convolution along one axis of the [B][F][T] token grid on [rows][C] tensors (C = groups * CG in and out)
static TapGemm gb_conv(const void* X, const void* Wp, const float* bias, void* Y, long rows, int C, int groups, int taps, int shift, int pos_div,
                       int pos_len) {
    const int CG = C / groups;
    TapGemm p = gb_lin(X, C, Wp, bias, Y, C, rows, CG, CG);
    p.groups = groups; p.xgs = CG; p.ygs = CG; p.bgs = CG;
    p.taps = taps; p.center = taps / 2; p.shift = shift; p.pos_div = pos_div; p.pos_len = pos_len;
    return p;
}

template <class T>
static int gb_ln_fwd(const void* x, const float* gamma, const float* beta, void* u, float* stats, long N, int C, hipStream_t st) {
    if (C > 64 * GB_CPL) return NBSS_EUNSUPPORTED;
    if (C == 192) NBSS_LAUNCH((gb_ln_fwd8_kernel<T, 3>), dim3(gb_blocks(N, 32 * 2)), dim3(GB_THREADS), 0, st, (const T*)x, gamma, beta, (T*)u, stats, N);
    else if (C == 384) NBSS_LAUNCH((gb_ln_fwd8_kernel<T, 6>), dim3(gb_blocks(N, 32 * 2)), dim3(GB_THREADS), 0, st, (const T*)x, gamma, beta, (T*)u, stats, N);
    else NBSS_LAUNCH((gb_ln_fwd_kernel<T>), dim3(gb_blocks(N, 4)), dim3(GB_THREADS), 0, st, (const T*)x, gamma, beta, (T*)u, stats, N, C);
    return NBSS_CHECK_LAUNCH();
}
template <class T>
static int gb_ln_bwd(const void* du, const void* x, const float* stats, const float* gamma, const void* dy, void* dx, float* dgamma, float* dbeta, long N,
                     int C, hipStream_t st, float* part = nullptr, size_t part_floats = 0) {
    if (C > 64 * GB_CPL) return NBSS_EUNSUPPORTED;
    const int blocks = gb_blocks(N, 4 * 16) < 1024 ? gb_blocks(N, 4 * 16) : 1024;  // >= 16 rows per wave: the affine sums end in C atomics per workgroup
    int blocks8 = gb_blocks(N, 32 * 4) < 1024 ? gb_blocks(N, 32 * 4) : 1024;  // (>= 4 rows per 8-lane group)
    if (part && (C == 192 || C == 384)) {
        if ((size_t)blocks8 * 2 * C > part_floats) blocks8 = (int)(part_floats / (2 * C));
        if (blocks8 < 64) part = nullptr, blocks8 = gb_blocks(N, 32 * 4) < 1024 ? gb_blocks(N, 32 * 4) : 1024;
    } else part = nullptr;
    if (C == 192)
        NBSS_LAUNCH((gb_ln_bwd8_kernel<T, 3>), dim3(blocks8), dim3(GB_THREADS), 2 * 192 * sizeof(float), st, (const T*)du, (const T*)x, stats, gamma, (const T*)dy, (T*)dx, dgamma, dbeta, part, N);
    else if (C == 384)
        NBSS_LAUNCH((gb_ln_bwd8_kernel<T, 6>), dim3(blocks8), dim3(GB_THREADS), 2 * 384 * sizeof(float), st, (const T*)du, (const T*)x, stats, gamma, (const T*)dy, (T*)dx, dgamma, dbeta, part, N);
    else
        NBSS_LAUNCH((gb_ln_bwd_kernel<T>), dim3(blocks), dim3(GB_THREADS), 2 * 64 * GB_CPL * sizeof(float), st, (const T*)du, (const T*)x, stats, gamma, (const T*)dy, (T*)dx, dgamma, dbeta, N, C);
    int e = NBSS_CHECK_LAUNCH();
    if (e || !part) return e;
    AffSegs sg;
    sg.n = 2;
    sg.off[0] = 0; sg.off[1] = dbeta - dgamma;  // (relative to dgamma)
    sg.cnt[0] = sg.cnt[1] = C;
    return affine_reduce_launch(part, blocks8, sg, dgamma, st);
}
// the workspace region of the per-workgroup affine rows (layout.h: ws_part_offset; B max(F, T) rows of 576 floats)
static float* gb_part(const nbss_cfg& c, void* ws) { return (float*)((char*)ws + ws_part_offset(c)); }
static size_t gb_part_floats(const nbss_cfg& c) { return (size_t)c.B * (c.F > c.T ? c.F : c.T) * 576; }
template <class T>
static int gb_silu_bwd(const void* a, const void* gin, void* gout, long n, hipStream_t st) {
    NBSS_LAUNCH((gb_silu_bwd_kernel<T>), dim3(gb_blocks(n, 1024)), dim3(256), 0, st, (const T*)a, (const T*)gin, (T*)gout, n);
    return NBSS_CHECK_LAUNCH();
}

int wgrad_dense_g(const void* A, int lda, int M, const void* B, int ldb, int K, float* dW, float* dbias, long Ntok, float* part, hipStream_t st);
static void gb_wgrad_base(WgradArgs& a, const nbss_cfg& c, void* ws, long Ntok) {
    a.part = (float*)((char*)ws + ws_wgpart_offset(c));
    a.mvalid = 0; a.nvalid = 0;
    a.Ntok = (int)Ntok; a.F = c.F; a.T = c.T; a.shift_stride = 1; a.shift_dim = 0; a.groups = 1; a.taps = 1;
    a.stats = nullptr; a.gamma = nullptr; a.beta = nullptr;
}
// dW[M][K] += A^T B over the rows, dbias += colsum(A).  M in slices whose tiles fit one workgroup of wgrad.hip's transposing-read kernel
// (<= 112 tiles of 16 x 16, <= 7 staging slots): as one 576 x 192 problem the in_proj gradient took the column-range fallback with its
// atomic flush (17.8 % of the large train step for the five dense problems of a layer)
static int gb_wgrad_dense(const nbss_cfg& c, void* ws, const void* A, int lda, int M, const void* B, int ldb, int K, float* dW, float* dbias, long Ntok,
                          hipStream_t st) {
    const size_t esz = c.dtype == NBSS_BF16 ? 2 : 4;
    if (c.dtype == NBSS_BF16) {  // 192 x 96 output tiles with fragment reuse (wgrad_g.hip) where the shape allows
        const int e = wgrad_dense_g(A, lda, M, B, ldb, K, dW, dbias, Ntok, (float*)((char*)ws + ws_wgpart_offset(c)), st);
        if (e != NBSS_EUNSUPPORTED) return e;
    }
    int mt = 112 / cdiv(K, 16);
    if (mt > 12) mt = 12;
    while (mt > 1 && cdiv(mt * 16, 64) + cdiv(K, 64) > 7) --mt;
    const int ms = mt * 16;
    for (int m0 = 0; m0 < M; m0 += ms) {
        const int mm = M - m0 < ms ? M - m0 : ms;
        WgradArgs a;
        gb_wgrad_base(a, c, ws, Ntok);
        a.A = (const char*)A + (size_t)m0 * esz; a.lda = lda; a.MA = mm;
        a.B = B; a.ldb = ldb; a.NB = K;
        a.dW = dW + (size_t)m0 * K; a.dbias = dbias ? dbias + m0 : nullptr;
        int e = wgrad_launch(a, c.dtype, st);
        if (e) return e;
    }
    return NBSS_OK;
}

// ---- F-conv block (SpatialNet.py:116-127): y = x + PReLU(conv_F(LN(x))) ----------------------------------------------------------------
template <class T>
static int gb_fconv_bwd_t(const nbss_cfg& c, const float* P, float* G, int layer, int which, const void* x, const void* dy, void* dx, void* ws, hipStream_t st,
                          const Side* sd) {
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    const long N = (long)c.B * c.F * c.T;
    const int H = c.H, CG = H / c.f_groups, Mp = pad16(CG), Kp = pad32(CG);
    const int pLW = which ? P_FC2_LN_W : P_FC1_LN_W, pLB = which ? P_FC2_LN_B : P_FC1_LN_B, pW = which ? P_FC2_W : P_FC1_W, pB = which ? P_FC2_B : P_FC1_B,
              pA = which ? P_FC2_PRELU : P_FC1_PRELU;
    float* stats = (float*)ws;
    GbArena ar = gb_arena(c, ws);
    if (sizeof(T) == 2 && fconv_g_takes(c)) {  // the whole block in one kernel per (b, t) slab (fconv_g.hip)
        void* dv = ar.take(N * H * sizeof(T));
        void* wfr = ar.take(fconv_g_wfrag_elems() * sizeof(T));
        void* wdr = ar.take(fconv_g_wfrag_elems() * sizeof(T));
        if (!wdr) return NBSS_EUNSUPPORTED;
        int e = fconv_g_bwd(c, P, G, layer, which, x, dy, dx, dv, stats, wfr, wdr, gb_part_floats(c) >= (size_t)c.B * c.T * 576 ? gb_part(c, ws) : nullptr, st);
        if (e) return e;
        // conv weight: dW[o][i][tap] = sum_n dv[n][o] LN(x)[n + (tap - 2) T][i] (LayerNorm applied on the fly from the statistics), bias = colsum(dv)
        const hipStream_t gs = side_fork(sd, st);
        WgradArgs wa;
        gb_wgrad_base(wa, c, ws, N);
        wa.shift_stride = c.T; wa.shift_dim = 1; wa.groups = c.f_groups; wa.taps = c.f_ks;
        wa.A = dv; wa.lda = H; wa.MA = H; wa.B = x; wa.ldb = H; wa.NB = H;
        wa.stats = stats; wa.gamma = lp.p[pLW]; wa.beta = lp.p[pLB];
        wa.dW = G + param_off(c, layer, pW); wa.dbias = G + param_off(c, layer, pB);
        return wgrad_launch(wa, c.dtype, gs);
    }
    void* u = ar.take(N * H * sizeof(T));
    void* a = ar.take(N * H * sizeof(T));
    void* da = ar.take(N * H * sizeof(T));
    void* du = ar.take(N * H * sizeof(T));
    void* wf = ar.take((size_t)c.f_groups * c.f_ks * Mp * Kp * sizeof(T));
    void* wd = ar.take((size_t)c.f_groups * c.f_ks * Mp * Kp * sizeof(T));
    if (!wd) return NBSS_EUNSUPPORTED;
    int e;
    if ((e = gb_wprep<T>(lp.p[pW], wf, WP_CONV_FWD, c.f_groups, c.f_ks, CG, CG, Mp, Kp, st))) return e;
    if ((e = gb_wprep<T>(lp.p[pW], wd, WP_CONV_DGRAD, c.f_groups, c.f_ks, CG, CG, Mp, Kp, st))) return e;
    if ((e = gb_ln_fwd<T>(x, lp.p[pLW], lp.p[pLB], u, stats, N, H, st))) return e;
    // a = conv along F (rows T apart, position f = (n / T) % F)
    if ((e = gb_gemm<T>(gb_conv(u, wf, lp.p[pB], a, N, H, c.f_groups, c.f_ks, c.T, c.T, c.F), st))) return e;
    if (H == 192)
        NBSS_LAUNCH((gb_prelu_bwd4_kernel<T, 3>), dim3(gb_blocks(N, 16 * 8) < 1024 ? gb_blocks(N, 16 * 8) : 1024), dim3(GB_THREADS), 192 * sizeof(float), st, (const T*)a,
                    (const T*)dy, lp.p[pA], (T*)da, G + param_off(c, layer, pA), N);
    else
        NBSS_LAUNCH((gb_prelu_bwd_kernel<T>), dim3(gb_blocks(N, 64) < 1024 ? gb_blocks(N, 64) : 1024), dim3(GB_THREADS), 64 * GB_CPL * sizeof(float), st, (const T*)a, (const T*)dy,
                    lp.p[pA], (T*)da, G + param_off(c, layer, pA), N, H);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    if ((e = gb_gemm<T>(gb_conv(da, wd, nullptr, du, N, H, c.f_groups, c.f_ks, c.T, c.T, c.F), st))) return e;
    if ((e = gb_ln_bwd<T>(du, x, stats, lp.p[pLW], dy, dx, G + param_off(c, layer, pLW), G + param_off(c, layer, pLB), N, H, st, gb_part(c, ws), gb_part_floats(c)))) return e;
    // conv weight: dW[o][i][tap] = sum_n da[n][o] u[n + (tap - 2) T][i], bias = colsum(da)
    const hipStream_t gs = side_fork(sd, st);
    WgradArgs wa;
    gb_wgrad_base(wa, c, ws, N);
    wa.shift_stride = c.T; wa.shift_dim = 1; wa.groups = c.f_groups; wa.taps = c.f_ks;
    wa.A = da; wa.lda = H; wa.MA = H; wa.B = u; wa.ldb = H; wa.NB = H;
    wa.dW = G + param_off(c, layer, pW); wa.dbias = G + param_off(c, layer, pB);
    return wgrad_launch(wa, c.dtype, gs);
}

// ---- full-band block (SpatialNet.py:129-146): y = x + SiLU(Wu LinearGroup_F(SiLU(Ws LN(x) + bs)) + bu) ---------------------------------
template <class T>
static int gb_full_bwd_t(const nbss_cfg& c, const float* P, float* G, int layer, const void* x, const void* dy, void* dx, void* ws, hipStream_t st, const Side* sd) {
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    const long N = (long)c.B * c.F * c.T, BT = (long)c.B * c.T;
    const int H = c.H, SQ = c.SQ, F = c.F, FK = pad8(F);
    float* stats = (float*)ws;
    GbArena ar = gb_arena(c, ws);
    void* u = ar.take(N * H * sizeof(T));
    void* sp = ar.take(N * SQ * sizeof(T));    // squeeze pre-activation, later ds_pre
    void* s = ar.take(N * SQ * sizeof(T));     // SiLU(sp), later ds
    void* sT = ar.take(BT * SQ * FK * sizeof(T));
    void* zT = ar.take(BT * SQ * FK * sizeof(T));   // later ds^T
    void* z = ar.take(N * SQ * sizeof(T));     // later dz
    void* dzT = ar.take(BT * SQ * FK * sizeof(T));
    void* yp = ar.take(N * H * sizeof(T));     // unsqueeze pre-activation
    void* dyp = ar.take(N * H * sizeof(T));    // later du
    const int Fp16 = pad16(F), Fp32 = pad32(FK);
    void* w_sq = ar.take((size_t)pad16(SQ) * pad32(H) * sizeof(T));
    void* w_sqT = ar.take((size_t)pad16(H) * pad32(SQ) * sizeof(T));
    void* w_us = ar.take((size_t)pad16(H) * pad32(SQ) * sizeof(T));
    void* w_usT = ar.take((size_t)pad16(SQ) * pad32(H) * sizeof(T));
    void* w_lg = ar.take((size_t)SQ * Fp16 * Fp32 * sizeof(T));
    void* w_lgT = ar.take((size_t)SQ * Fp16 * Fp32 * sizeof(T));
    if (!w_lgT) return NBSS_EUNSUPPORTED;
    int e;
    {
        WPrepBatch<T> wb;
        wb.add(lp.p[P_SQ_W], w_sq, WP_LIN_FWD, 1, 1, SQ, H, pad16(SQ), pad32(H));
        wb.add(lp.p[P_SQ_W], w_sqT, WP_LIN_DGRAD, 1, 1, H, SQ, pad16(H), pad32(SQ));
        wb.add(lp.p[P_USQ_W], w_us, WP_LIN_FWD, 1, 1, H, SQ, pad16(H), pad32(SQ));
        wb.add(lp.p[P_USQ_W], w_usT, WP_LIN_DGRAD, 1, 1, SQ, H, pad16(SQ), pad32(H));
        wb.add(lp.p[P_FULL_W], w_lg, WP_LG_FWD, SQ, 1, F, F, Fp16, Fp32);
        wb.add(lp.p[P_FULL_W], w_lgT, WP_LG_DGRAD, SQ, 1, F, F, Fp16, Fp32);
        if ((e = wb.launch(st))) return e;
    }
    // forward chain
    if ((e = gb_ln_fwd<T>(x, lp.p[P_FULL_LN_W], lp.p[P_FULL_LN_B], u, stats, N, H, st))) return e;
    {
        TapGemm p = gb_lin(u, H, w_sq, lp.p[P_SQ_B], sp, SQ, N, SQ, H);
        p.Y2 = s;  // s = SiLU(s_pre) in the same pass
        if ((e = gb_gemm<T>(p, st))) return e;
    }
    NBSS_LAUNCH((gb_sq_to_f_kernel<T>), dim3(gb_blocks(BT * SQ * FK, 1024)), dim3(256), 0, st, (const T*)s, (T*)sT, c.B, F, c.T, SQ, FK);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    auto lg = [&](const void* X, const void* Wp, const float* bias, void* Y) {
        TapGemm p = gb_lin(X, SQ * FK, Wp, bias, Y, SQ * FK, BT, F, FK);
        p.groups = SQ; p.xgs = FK; p.ygs = FK; p.bgs = F;
        p.Mp = Fp16; p.Kp = Fp32;
        return p;
    };
    if ((e = gb_gemm<T>(lg(sT, w_lg, lp.p[P_FULL_B], zT), st))) return e;
    NBSS_LAUNCH((gb_f_to_sq_kernel<T>), dim3(gb_blocks(N * SQ, 1024)), dim3(256), 0, st, (const T*)zT, (T*)z, c.B, F, c.T, SQ, FK);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    if ((e = gb_gemm<T>(gb_lin(z, SQ, w_us, lp.p[P_USQ_B], yp, H, N, H, SQ), st))) return e;
    // backward chain
    if ((e = gb_silu_bwd<T>(yp, dy, dyp, N * H, st))) return e;                               // dy_pre
    void* dz = ar.take(N * SQ * sizeof(T));
    void* ds = ar.take(N * SQ * sizeof(T));
    if (!ds) return NBSS_EUNSUPPORTED;
    if ((e = gb_gemm<T>(gb_lin(dyp, H, w_usT, nullptr, dz, SQ, N, SQ, H), st))) return e;      // dz = Wu^T dy_pre
    NBSS_LAUNCH((gb_sq_to_f_kernel<T>), dim3(gb_blocks(BT * SQ * FK, 1024)), dim3(256), 0, st, (const T*)dz, (T*)dzT, c.B, F, c.T, SQ, FK);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    void* dsT = ar.take(BT * SQ * FK * sizeof(T));
    if (!dsT) return NBSS_EUNSUPPORTED;
    if ((e = gb_gemm<T>(lg(dzT, w_lgT, nullptr, dsT), st))) return e;                         // ds^T = Wf^T dz^T
    NBSS_LAUNCH((gb_f_to_sq_kernel<T>), dim3(gb_blocks(N * SQ, 1024)), dim3(256), 0, st, (const T*)dsT, (T*)ds, c.B, F, c.T, SQ, FK);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    if ((e = gb_silu_bwd<T>(sp, ds, ds, N * SQ, st))) return e;                               // ds_pre (in ds)
    void* du = ar.take(N * H * sizeof(T));
    if (!du) return NBSS_EUNSUPPORTED;
    if ((e = gb_gemm<T>(gb_lin(ds, SQ, w_sqT, nullptr, du, H, N, H, SQ), st))) return e;
    if ((e = gb_ln_bwd<T>(du, x, stats, lp.p[P_FULL_LN_W], dy, dx, G + param_off(c, layer, P_FULL_LN_W), G + param_off(c, layer, P_FULL_LN_B), N, H, st, gb_part(c, ws), gb_part_floats(c)))) return e;
    // weight gradients
    const hipStream_t gs = side_fork(sd, st);
    if ((e = gb_wgrad_dense(c, ws, dyp, H, H, z, SQ, SQ, G + param_off(c, layer, P_USQ_W), G + param_off(c, layer, P_USQ_B), N, gs))) return e;
    WgradArgs wa;
    gb_wgrad_base(wa, c, ws, BT);
    wa.groups = SQ; wa.mvalid = F; wa.nvalid = F;
    wa.A = dzT; wa.lda = SQ * FK; wa.MA = SQ * FK; wa.B = sT; wa.ldb = SQ * FK; wa.NB = SQ * FK;
    wa.dW = G + param_off(c, layer, P_FULL_W); wa.dbias = G + param_off(c, layer, P_FULL_B);
    if ((e = wgrad_launch(wa, c.dtype, gs))) return e;
    return gb_wgrad_dense(c, ws, ds, SQ, SQ, u, H, H, G + param_off(c, layer, P_SQ_W), G + param_off(c, layer, P_SQ_B), N, gs);
}

// ---- attention block (SpatialNet.py:93-100): y = x + out_proj(MHSA(LN(x))) ---------------------------------------------------------------
template <class T, int DH>
static int gb_attn_launch(const nbss_cfg& c, const void* qkv, const void* dO, void* O, void* dqkv, float* lse, float* Dv, hipStream_t st) {
    const int TP = 32 * cdiv(c.T, 32);
    const size_t ldsq = (size_t)2 * TP * DH * sizeof(T) + 64, ldsk = ldsq + (size_t)2 * TP * sizeof(float);  // (+64: the transposing reads of a 24-wide head's second channel tile run 16 bytes past the last row)
    if (c.T > GA_TMAX || ldsk > 160 * 1024) return NBSS_EUNSUPPORTED;
    int e;
    if ((e = NBSS_SET_MAX_LDS((gb_attn_q_kernel<T, DH, true>), ldsq))) return e;
    if ((e = NBSS_SET_MAX_LDS((gb_attn_k_kernel<T, DH>), ldsk))) return e;
    dim3 grid(c.B * c.F, c.heads);
    NBSS_LAUNCH((gb_attn_q_kernel<T, DH, true>), grid, dim3(GB_THREADS), ldsq, st, (const T*)qkv, (const T*)dO, (T*)O, (T*)dqkv, lse, Dv, c.T, c.H, c.heads);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    NBSS_LAUNCH((gb_attn_k_kernel<T, DH>), grid, dim3(GB_THREADS), ldsk, st, (const T*)qkv, (const T*)dO, (T*)dqkv, (const float*)lse, (const float*)Dv, c.T, c.H, c.heads);
    return NBSS_CHECK_LAUNCH();
}

template <class T>
static int gb_mhsa_bwd_t(const nbss_cfg& c, const float* P, float* G, int layer, const void* x, const void* dy, void* dx, void* ws, hipStream_t st, const Side* sd) {
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    const long N = (long)c.B * c.F * c.T;
    const int H = c.H, DH = H / c.heads;
    float* stats = (float*)ws;
    GbArena ar = gb_arena(c, ws);
    void* u = ar.take(N * H * sizeof(T));
    void* qkv = ar.take(N * 3 * H * sizeof(T));
    void* dO = ar.take(N * H * sizeof(T));
    void* O = ar.take(N * H * sizeof(T));
    void* dqkv = ar.take(N * 3 * H * sizeof(T));
    void* du = ar.take(N * H * sizeof(T));
    float* lse = (float*)ar.take(N * c.heads * sizeof(float));
    float* Dv = (float*)ar.take(N * c.heads * sizeof(float));
    void* w_in = ar.take((size_t)pad16(3 * H) * pad32(H) * sizeof(T));
    void* w_inT = ar.take((size_t)pad16(H) * pad32(3 * H) * sizeof(T));
    void* w_outT = ar.take((size_t)pad16(H) * pad32(H) * sizeof(T));
    if (!w_outT) return NBSS_EUNSUPPORTED;
    int e;
    {
        WPrepBatch<T> wb;
        wb.add(lp.p[P_INP_W], w_in, WP_LIN_FWD, 1, 1, 3 * H, H, pad16(3 * H), pad32(H));
        wb.add(lp.p[P_INP_W], w_inT, WP_LIN_DGRAD, 1, 1, H, 3 * H, pad16(H), pad32(3 * H));
        wb.add(lp.p[P_OUTP_W], w_outT, WP_LIN_DGRAD, 1, 1, H, H, pad16(H), pad32(H));
        if ((e = wb.launch(st))) return e;
    }
    if ((e = gb_ln_fwd<T>(x, lp.p[P_MH_LN_W], lp.p[P_MH_LN_B], u, stats, N, H, st))) return e;
    if ((e = gb_gemm<T>(gb_lin(u, H, w_in, lp.p[P_INP_B], qkv, 3 * H, N, 3 * H, H), st))) return e;
    if ((e = gb_gemm<T>(gb_lin(dy, H, w_outT, nullptr, dO, H, N, H, H), st))) return e;  // dO = dy Wo
    if (DH == 48) e = gb_attn_launch<T, 48>(c, qkv, dO, O, dqkv, lse, Dv, st);
    else if (DH == 24) e = gb_attn_launch<T, 24>(c, qkv, dO, O, dqkv, lse, Dv, st);
    else e = NBSS_EUNSUPPORTED;
    if (e) return e;
    if ((e = gb_gemm<T>(gb_lin(dqkv, 3 * H, w_inT, nullptr, du, H, N, H, 3 * H), st))) return e;
    if ((e = gb_ln_bwd<T>(du, x, stats, lp.p[P_MH_LN_W], dy, dx, G + param_off(c, layer, P_MH_LN_W), G + param_off(c, layer, P_MH_LN_B), N, H, st, gb_part(c, ws), gb_part_floats(c)))) return e;
    const hipStream_t gs = side_fork(sd, st);
    if ((e = gb_wgrad_dense(c, ws, dy, H, H, O, H, H, G + param_off(c, layer, P_OUTP_W), G + param_off(c, layer, P_OUTP_B), N, gs))) return e;
    return gb_wgrad_dense(c, ws, dqkv, 3 * H, 3 * H, u, H, H, G + param_off(c, layer, P_INP_W), G + param_off(c, layer, P_INP_B), N, gs);
}

// ---- T-ConvFFN block (SpatialNet.py:102-114) -----------------------------------------------------------------------------------------------
template <class T>
static int gb_tconvffn_bwd_t(const nbss_cfg& c, const float* P, float* G, int layer, const void* x, const void* dy, void* dx, void* ws, hipStream_t st,
                             const Side* sd) {
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    const long N = (long)c.B * c.F * c.T;
    const int H = c.H, FFN = c.FFN, CG = FFN / c.t_groups, nseq = c.B * c.F;
    if (CG > 64) return NBSS_EUNSUPPORTED;
    float* stats = (float*)ws;
    GbArena ar = gb_arena(c, ws);
    void* u = ar.take(N * H * sizeof(T));
    void* du = ar.take(N * H * sizeof(T));
    void* t[12];  // a1 h1 a2 h2 a3 h4 a5 h5 | g5 g3 g2 g1
    for (int i = 0; i < 12; ++i) t[i] = ar.take(N * FFN * sizeof(T));
    void *a1 = t[0], *h1 = t[1], *a2 = t[2], *h2 = t[3], *a3 = t[4], *h4 = t[5], *a5 = t[6], *h5 = t[7], *g5 = t[8], *g3 = t[9], *g2 = t[10], *g1 = t[11];
    float* gstats = (float*)ar.take((size_t)nseq * c.t_groups * 2 * sizeof(float));
    const int Mp = pad16(CG), Kp = pad32(CG);
    void* w1 = ar.take((size_t)pad16(FFN) * pad32(H) * sizeof(T));
    void* w1T = ar.take((size_t)pad16(H) * pad32(FFN) * sizeof(T));
    void* w2T = ar.take((size_t)pad16(FFN) * pad32(H) * sizeof(T));
    void* cw[3], *cwT[3];
    for (int k = 0; k < 3; ++k) {
        cw[k] = ar.take((size_t)c.t_groups * c.t_ks * Mp * Kp * sizeof(T));
        cwT[k] = ar.take((size_t)c.t_groups * c.t_ks * Mp * Kp * sizeof(T));
    }
    if (!cwT[2]) return NBSS_EUNSUPPORTED;
    const int convW[3] = {P_TF_C1W, P_TF_C2W, P_TF_C3W}, convB[3] = {P_TF_C1B, P_TF_C2B, P_TF_C3B};
    int e;
    {
        WPrepBatch<T> wb;
        wb.add(lp.p[P_TF_W1], w1, WP_LIN_FWD, 1, 1, FFN, H, pad16(FFN), pad32(H));
        wb.add(lp.p[P_TF_W1], w1T, WP_LIN_DGRAD, 1, 1, H, FFN, pad16(H), pad32(FFN));
        wb.add(lp.p[P_TF_W2], w2T, WP_LIN_DGRAD, 1, 1, FFN, H, pad16(FFN), pad32(H));
        if ((e = wb.launch(st))) return e;
    }
    auto with = [](TapGemm p, void* y2, const void* dact) {  // second output SiLU(Y) / result times SiLU'(dact): the activation passes ride along
        p.Y2 = y2;
        p.Dact = dact;
        return p;
    };
    if ((e = gb_ln_fwd<T>(x, lp.p[P_TF_LN_W], lp.p[P_TF_LN_B], u, stats, N, H, st))) return e;
    if (tc_chain_takes(c.dtype, CG, c.t_ks, c.T)) {
        // the conv chain between the two dense maps in ONE kernel per layer (tchain.hip): a1 and dh5 in, every operand of the weight gradients out
        void* dh5 = a2;  // (the buffers of the pre-activations the chain keeps in registers)
        TChain tc;
        const float* wsrc[3];
        for (int k = 0; k < 3; ++k) {
            wsrc[k] = lp.p[convW[k]];
            tc.wf[k] = cw[k];
            tc.wd[k] = cwT[k];
            tc.cb[k] = lp.p[convB[k]];
        }
        if (tc_wfrag_elems(c.t_groups, CG, c.t_ks) > (size_t)c.t_groups * c.t_ks * Mp * Kp) return NBSS_EUNSUPPORTED;
        {
            void* wfv[3] = {cw[0], cw[1], cw[2]};
            void* wdv[3] = {cwT[0], cwT[1], cwT[2]};
            if ((e = tc_wprep(wsrc, wfv, wdv, c.t_groups, CG, c.t_ks, st))) return e;
        }
        if ((e = gb_gemm<T>(gb_lin(u, H, w1, lp.p[P_TF_B1], a1, FFN, N, FFN, H), st))) return e;
        if ((e = gb_gemm<T>(gb_lin(dy, H, w2T, nullptr, dh5, FFN, N, FFN, H), st))) return e;
        tc.a1 = a1; tc.dh5 = dh5;
        tc.gn_w = lp.p[P_TF_GN_W]; tc.gn_b = lp.p[P_TF_GN_B];
        tc.h1 = h1; tc.h2 = h2; tc.h4 = h4; tc.h5 = h5; tc.g5 = g5; tc.g3 = g3; tc.g2 = g2; tc.g1 = g1;
        tc.dgn_w = G + param_off(c, layer, P_TF_GN_W); tc.dgn_b = G + param_off(c, layer, P_TF_GN_B);
        tc.part = gb_part_floats(c) >= (size_t)nseq * 2 * FFN ? gb_part(c, ws) : nullptr;
        tc.nseq = nseq; tc.T = c.T; tc.FFN = FFN; tc.groups = c.t_groups;
        if ((e = tc_chain_launch(tc, CG, c.t_ks, true, st))) return e;
        if (tc.part) {
            AffSegs sg;
            sg.n = 2;
            sg.off[0] = param_off(c, layer, P_TF_GN_W); sg.off[1] = param_off(c, layer, P_TF_GN_B);
            sg.cnt[0] = sg.cnt[1] = FFN;
            if ((e = affine_reduce_launch(tc.part, nseq, sg, G, st))) return e;
        }
    } else {
        for (int k = 0; k < 3; ++k) {
            if ((e = gb_wprep<T>(lp.p[convW[k]], cw[k], WP_CONV_FWD, c.t_groups, c.t_ks, CG, CG, Mp, Kp, st))) return e;
            if ((e = gb_wprep<T>(lp.p[convW[k]], cwT[k], WP_CONV_DGRAD, c.t_groups, c.t_ks, CG, CG, Mp, Kp, st))) return e;
        }
        auto tconv = [&](const void* X, const void* Wp, const float* bias, void* Y) {  // along T: rows 1 apart, position t = n % T
            return gb_conv(X, Wp, bias, Y, N, FFN, c.t_groups, c.t_ks, 1, 1, c.T);
        };
        // forward chain, every pre-activation and activation kept
        if ((e = gb_gemm<T>(with(gb_lin(u, H, w1, lp.p[P_TF_B1], a1, FFN, N, FFN, H), h1, nullptr), st))) return e;
        if ((e = gb_gemm<T>(with(tconv(h1, cw[0], lp.p[convB[0]], a2), h2, nullptr), st))) return e;
        if ((e = gb_gemm<T>(tconv(h2, cw[1], lp.p[convB[1]], a3), st))) return e;
        NBSS_LAUNCH((gb_gn_fwd_kernel<T>), dim3(nseq * c.t_groups), dim3(GB_THREADS), 8 * sizeof(float), st, (const T*)a3, lp.p[P_TF_GN_W], lp.p[P_TF_GN_B], (T*)h4, gstats, c.T, FFN, CG, 1);
        if ((e = NBSS_CHECK_LAUNCH())) return e;
        if ((e = gb_gemm<T>(with(tconv(h4, cw[2], lp.p[convB[2]], a5), h5, nullptr), st))) return e;
        // backward chain: g5 = da5, g3 = da3 (through the GroupNorm), g2 = da2, g1 = da1
        if ((e = gb_gemm<T>(with(gb_lin(dy, H, w2T, nullptr, g5, FFN, N, FFN, H), nullptr, a5), st))) return e;
        if ((e = gb_gemm<T>(tconv(g5, cwT[2], nullptr, g3), st))) return e;
        NBSS_LAUNCH((gb_gn_bwd_kernel<T>), dim3(nseq * c.t_groups), dim3(GB_THREADS), (8 + 128) * sizeof(float), st, (const T*)a3, (const float*)gstats, lp.p[P_TF_GN_W], lp.p[P_TF_GN_B], (T*)g3,
                    G + param_off(c, layer, P_TF_GN_W), G + param_off(c, layer, P_TF_GN_B), c.T, FFN, CG);
        if ((e = NBSS_CHECK_LAUNCH())) return e;
        if ((e = gb_gemm<T>(with(tconv(g3, cwT[1], nullptr, g2), nullptr, a2), st))) return e;
        if ((e = gb_gemm<T>(with(tconv(g2, cwT[0], nullptr, g1), nullptr, a1), st))) return e;
    }
    if ((e = gb_gemm<T>(gb_lin(g1, FFN, w1T, nullptr, du, H, N, H, FFN), st))) return e;
    if ((e = gb_ln_bwd<T>(du, x, stats, lp.p[P_TF_LN_W], dy, dx, G + param_off(c, layer, P_TF_LN_W), G + param_off(c, layer, P_TF_LN_B), N, H, st, gb_part(c, ws), gb_part_floats(c)))) return e;
    // weight gradients (every operand above is still in place: nothing was overwritten)
    const hipStream_t gs = side_fork(sd, st);
    if ((e = gb_wgrad_dense(c, ws, dy, H, H, h5, FFN, FFN, G + param_off(c, layer, P_TF_W2), G + param_off(c, layer, P_TF_B2), N, gs))) return e;
    const void* cA[3] = {g2, g3, g5};
    const void* cB[3] = {h1, h2, h4};
    for (int k = 0; k < 3; ++k) {
        WgradArgs wa;
        gb_wgrad_base(wa, c, ws, N);
        wa.groups = c.t_groups; wa.taps = c.t_ks;
        wa.A = cA[k]; wa.lda = FFN; wa.MA = FFN; wa.B = cB[k]; wa.ldb = FFN; wa.NB = FFN;
        wa.dW = G + param_off(c, layer, convW[k]); wa.dbias = G + param_off(c, layer, convB[k]);
        if ((e = wgrad_launch(wa, c.dtype, gs))) return e;
    }
    return gb_wgrad_dense(c, ws, g1, FFN, FFN, u, H, H, G + param_off(c, layer, P_TF_W1), G + param_off(c, layer, P_TF_B1), N, gs);
}

// ---- T-ConvFFN forward on the same pieces (bf16 stream): LN -> dense map -> conv chain -> dense map + residual.  tconvffn_g.hip's one-kernel
// forward walks a sequence's 8 groups serially with every weight fragment from L2 (1.1 ms per layer at batch 4); these four launches take a third.
int gb_tconvffn_fwd(const nbss_cfg& c, const float* P, int layer, const void* x, void* y, void* ws, hipStream_t st) {
    typedef bf16_t T;
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    const long N = (long)c.B * c.F * c.T;
    const int H = c.H, FFN = c.FFN, CG = FFN / c.t_groups;
    if (!ws || !tc_chain_takes(c.dtype, CG, c.t_ks, c.T)) return NBSS_EUNSUPPORTED;
    ProfScope ps(PK_TCF_F, st);
    float* stats = (float*)ws;
    GbArena ar = gb_arena(c, ws);
    void* u = ar.take(N * H * sizeof(T));
    void* a1 = ar.take(N * FFN * sizeof(T));
    void* h5 = ar.take(N * FFN * sizeof(T));
    void* w1 = ar.take((size_t)pad16(FFN) * pad32(H) * sizeof(T));
    void* w2 = ar.take((size_t)pad16(H) * pad32(FFN) * sizeof(T));
    void* cw[3];
    for (int k = 0; k < 3; ++k) cw[k] = ar.take(tc_wfrag_elems(c.t_groups, CG, c.t_ks) * sizeof(T));
    if (!cw[2]) return NBSS_EUNSUPPORTED;
    int e;
    {
        WPrepBatch<T> wb;
        wb.add(lp.p[P_TF_W1], w1, WP_LIN_FWD, 1, 1, FFN, H, pad16(FFN), pad32(H));
        wb.add(lp.p[P_TF_W2], w2, WP_LIN_FWD, 1, 1, H, FFN, pad16(H), pad32(FFN));
        if ((e = wb.launch(st))) return e;
    }
    const float* wsrc[3] = {lp.p[P_TF_C1W], lp.p[P_TF_C2W], lp.p[P_TF_C3W]};
    void* none[3] = {nullptr, nullptr, nullptr};
    if ((e = tc_wprep(wsrc, cw, none, c.t_groups, CG, c.t_ks, st))) return e;
    if ((e = gb_ln_fwd<T>(x, lp.p[P_TF_LN_W], lp.p[P_TF_LN_B], u, stats, N, H, st))) return e;
    if ((e = gb_gemm<T>(gb_lin(u, H, w1, lp.p[P_TF_B1], a1, FFN, N, FFN, H), st))) return e;
    TChain tc = {};
    tc.a1 = a1;
    for (int k = 0; k < 3; ++k) tc.wf[k] = cw[k];
    tc.cb[0] = lp.p[P_TF_C1B]; tc.cb[1] = lp.p[P_TF_C2B]; tc.cb[2] = lp.p[P_TF_C3B];
    tc.gn_w = lp.p[P_TF_GN_W]; tc.gn_b = lp.p[P_TF_GN_B];
    tc.h5 = h5;
    tc.nseq = c.B * c.F; tc.T = c.T; tc.FFN = FFN; tc.groups = c.t_groups;
    if ((e = tc_chain_launch(tc, CG, c.t_ks, false, st))) return e;
    TapGemm p2 = gb_lin(h5, FFN, w2, lp.p[P_TF_B2], y, H, N, H, FFN);
    p2.R = x; p2.ldr = H;
    return gb_gemm<T>(p2, st);
}

// ---- decoder (SpatialNet.py:200,216): out = Wd x + bd; dx = Wd^T dout ---------------------------------------------------------------------
template <class T>
static int gb_decoder_bwd_t(const nbss_cfg& c, const float* P, float* G, const void* x, const float* dout, void* dx, void* ws, hipStream_t st) {
    const long N = (long)c.B * c.F * c.T;
    const int H = c.H, Co = c.C_out, CP = pad8(Co);
    GbArena ar = gb_arena(c, ws);
    void* dpad = ar.take(N * CP * sizeof(T));
    void* wT = ar.take((size_t)pad16(H) * pad32(CP) * sizeof(T));
    if (!wT) return NBSS_EUNSUPPORTED;
    int e;
    NBSS_LAUNCH((gb_pad_cols_kernel<T>), dim3(gb_blocks(N * CP, 1024)), dim3(256), 0, st, dout, (T*)dpad, N, Co, CP);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    // W^T as a tap_gemm weight: M = H inputs of the decoder, K = its outputs (valid Co, stored CP wide)
    if ((e = gb_wprep<T>(P + param_off_dec_w(c), wT, WP_LIN_DGRAD, 1, 1, H, Co, pad16(H), pad32(CP), st))) return e;
    TapGemm p = gb_lin(dpad, CP, wT, nullptr, dx, H, N, H, CP);
    if ((e = gb_gemm<T>(p, st))) return e;
    WgradArgs wa;
    gb_wgrad_base(wa, c, ws, N);
    wa.mvalid = Co;
    wa.A = dpad; wa.lda = CP; wa.MA = CP; wa.B = x; wa.ldb = H; wa.NB = H;
    wa.dW = G + param_off_dec_w(c); wa.dbias = G + param_off_dec_b(c);
    return wgrad_launch(wa, c.dtype, st);
}

// ---- entry points (capi.hip dispatches here for every geometry but SpatialNet-small) ----------------------------------------------------------
#define GB_DISPATCH(fn, ...) (c.dtype == NBSS_BF16 ? fn<bf16_t>(__VA_ARGS__) : fn<float>(__VA_ARGS__))
int gb_fconv_bwd(const nbss_cfg& c, const float* P, float* G, int layer, int which, const void* x, const void* dy, void* dx, void* ws, hipStream_t st, const Side* sd) {
    ProfScope ps(PK_FCONV_B, st);
    return GB_DISPATCH(gb_fconv_bwd_t, c, P, G, layer, which, x, dy, dx, ws, st, sd);
}
int gb_full_bwd(const nbss_cfg& c, const float* P, float* G, int layer, const void* x, const void* dy, void* dx, void* ws, hipStream_t st, const Side* sd) {
    ProfScope ps(PK_FULL_B, st);
    return GB_DISPATCH(gb_full_bwd_t, c, P, G, layer, x, dy, dx, ws, st, sd);
}
int gb_mhsa_bwd(const nbss_cfg& c, const float* P, float* G, int layer, const void* x, const void* dy, void* dx, void* ws, hipStream_t st, const Side* sd) {
    ProfScope ps(PK_MHSA_B, st);
    return GB_DISPATCH(gb_mhsa_bwd_t, c, P, G, layer, x, dy, dx, ws, st, sd);
}
int gb_tconvffn_bwd(const nbss_cfg& c, const float* P, float* G, int layer, const void* x, const void* dy, void* dx, void* ws, hipStream_t st, const Side* sd) {
    ProfScope ps(PK_TCF_B, st);
    return GB_DISPATCH(gb_tconvffn_bwd_t, c, P, G, layer, x, dy, dx, ws, st, sd);
}
int gb_decoder_bwd(const nbss_cfg& c, const float* P, float* G, const void* x, const float* dout, void* dx, void* ws, hipStream_t st) {
    return GB_DISPATCH(gb_decoder_bwd_t, c, P, G, x, dout, dx, ws, st);
}

// ---- narrow-band building blocks behind the C ABI (nbss_nb_*: include/nbss_hip.h) ----------------------------------------------------------
// The same generic kernels, one operation per call on caller-owned tensors: what a narrow-band network other than SpatialNet (NBC2: pre-norm
// attention over time + convolutional feed-forward with GroupBatchNorm) is sequenced from on the host side (nbss_amd/nbc2.py).
size_t nb_ws_bytes_impl(int M, int K, int groups, int taps) { return ws_align((size_t)groups * taps * pad16(M / groups) * pad32(pad8(K / groups)) * sizeof(float)); }

template <class T>
static int nb_conv_t(long nseq, int Tn, int Cin, int ldx, int Cout, int groups, int taps, const void* x, const float* w, const float* bias, void* y, const void* residual,
                     int act_in, int act_out, void* ws, hipStream_t st) {
    if (groups <= 0 || Cin % groups || Cout % groups || (groups > 1 && ldx != Cin)) return NBSS_EINVAL;
    const int Kv = Cin / groups, Kg = groups > 1 ? Kv : pad8(Kv), Mg = Cout / groups;
    if (Kg % 8 || ldx < (groups > 1 ? Cin : Kg)) return NBSS_EUNSUPPORTED;
    int e = gb_wprep<T>(w, ws, taps > 1 || groups > 1 ? WP_CONV_FWD : WP_LIN_FWD, groups, taps, Mg, Kv, pad16(Mg), pad32(Kg), st);
    if (e) return e;
    TapGemm p = gb_lin(x, ldx, ws, bias, y, Cout, nseq * Tn, Mg, Kg);
    p.groups = groups; p.xgs = groups > 1 ? Kv : 0; p.ygs = groups > 1 ? Mg : 0; p.bgs = Mg;
    p.taps = taps; p.center = taps / 2; p.shift = 1; p.pos_div = 1; p.pos_len = Tn;
    p.xact = act_in; p.yact = act_out;
    p.R = residual; p.ldr = Cout;
    return gb_gemm<T>(p, st);
}
int nb_conv_t_impl(int dtype, long nseq, int Tn, int Cin, int ldx, int Cout, int groups, int taps, const void* x, const float* w, const float* bias, void* y,
                   const void* residual, int act_in, int act_out, void* ws, hipStream_t st) {
    return dtype == NBSS_BF16 ? nb_conv_t<bf16_t>(nseq, Tn, Cin, ldx, Cout, groups, taps, x, w, bias, y, residual, act_in, act_out, ws, st)
                              : nb_conv_t<float>(nseq, Tn, Cin, ldx, Cout, groups, taps, x, w, bias, y, residual, act_in, act_out, ws, st);
}
int nb_layernorm_impl(int dtype, long rows, int C, const void* x, const float* gamma, const float* beta, void* y, float* stats, hipStream_t st) {
    return dtype == NBSS_BF16 ? gb_ln_fwd<bf16_t>(x, gamma, beta, y, stats, rows, C, st) : gb_ln_fwd<float>(x, gamma, beta, y, stats, rows, C, st);
}
int nb_gbn_impl(int dtype, int B, int F, int Tn, int C, const void* x, const float* gamma, const float* beta, float eps, int act, void* y, hipStream_t st) {
    if (dtype == NBSS_BF16)
        NBSS_LAUNCH((gb_gbn_kernel<bf16_t>), dim3(B * Tn), dim3(GB_THREADS), 8 * sizeof(float), st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, F, Tn, C, eps, act);
    else
        NBSS_LAUNCH((gb_gbn_kernel<float>), dim3(B * Tn), dim3(GB_THREADS), 8 * sizeof(float), st, (const float*)x, gamma, beta, (float*)y, F, Tn, C, eps, act);
    return NBSS_CHECK_LAUNCH();
}
template <class T, int DH>
static int nb_attn_fwd(long nseq, int Tn, int H, int heads, const void* qkv, void* o, hipStream_t st) {
    const int TP = 32 * cdiv(Tn, 32);
    const size_t lds = (size_t)2 * TP * DH * sizeof(T) + 64;
    if (Tn > GA_TMAX || lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    int e = NBSS_SET_MAX_LDS((gb_attn_q_kernel<T, DH, false>), lds);
    if (e) return e;
    NBSS_LAUNCH((gb_attn_q_kernel<T, DH, false>), dim3((unsigned)nseq, heads), dim3(GB_THREADS), lds, st, (const T*)qkv, (const T*)nullptr, (T*)o, (T*)nullptr, (float*)nullptr,
                (float*)nullptr, Tn, H, heads);
    return NBSS_CHECK_LAUNCH();
}
int nb_attention_fwd_impl(int dtype, long nseq, int Tn, int H, int heads, const void* qkv, void* o, hipStream_t st) {
    if (heads <= 0 || H % heads) return NBSS_EINVAL;
    const int dh = H / heads;
    if (dh == 48) return dtype == NBSS_BF16 ? nb_attn_fwd<bf16_t, 48>(nseq, Tn, H, heads, qkv, o, st) : nb_attn_fwd<float, 48>(nseq, Tn, H, heads, qkv, o, st);
    if (dh == 24) return dtype == NBSS_BF16 ? nb_attn_fwd<bf16_t, 24>(nseq, Tn, H, heads, qkv, o, st) : nb_attn_fwd<float, 24>(nseq, Tn, H, heads, qkv, o, st);
    return NBSS_EUNSUPPORTED;
}

template <class T, int DH>
static int nb_attn_relpos(long nseq, int Tn, int H, int heads, const void* qkv, const void* pos, const float* ub, const float* vb, float scale, void* o, hipStream_t st,
                          const uint32_t* mask, float keep) {
    const int TP = 32 * cdiv(Tn, 32), RP = 32 * cdiv(2 * Tn - 1 + 32, 32);
    const size_t lds = (size_t)(2 * TP + RP) * DH * sizeof(T) + (size_t)(GB_THREADS / 64) * 32 * 16 * sizeof(float) + 64;
    if (Tn > GA_TMAX || lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    int e = NBSS_SET_MAX_LDS((gb_attn_relpos_kernel<T, DH>), lds);
    if (e) return e;
    NBSS_LAUNCH((gb_attn_relpos_kernel<T, DH>), dim3((unsigned)nseq, heads), dim3(GB_THREADS), lds, st, (const T*)qkv, (const T*)pos, ub, vb, (T*)o, scale, Tn, H, heads, mask,
                keep);
    return NBSS_CHECK_LAUNCH();
}
int nb_attention_relpos_fwd_impl(int dtype, long nseq, int Tn, int H, int heads, const void* qkv, const void* pos, const float* ub, const float* vb, float scale, void* o,
                                 hipStream_t st, const uint32_t* mask, float keep) {
    if (heads <= 0 || H % heads) return NBSS_EINVAL;
    const int dh = H / heads;
    if (dh == 48)
        return dtype == NBSS_BF16 ? nb_attn_relpos<bf16_t, 48>(nseq, Tn, H, heads, qkv, pos, ub, vb, scale, o, st, mask, keep)
                                  : nb_attn_relpos<float, 48>(nseq, Tn, H, heads, qkv, pos, ub, vb, scale, o, st, mask, keep);
    if (dh == 24)
        return dtype == NBSS_BF16 ? nb_attn_relpos<bf16_t, 24>(nseq, Tn, H, heads, qkv, pos, ub, vb, scale, o, st, mask, keep)
                                  : nb_attn_relpos<float, 24>(nseq, Tn, H, heads, qkv, pos, ub, vb, scale, o, st, mask, keep);
    return NBSS_EUNSUPPORTED;
}
// backward of the relative-position attention: ws = lse [N][heads] | D [N][heads] | du/dvb shares [nseq][heads][2][dh] | dP shares [nseq][heads][32 NT][dh] (fp32)
size_t nb_relpos_bwd_ws_bytes_impl(long nseq, int Tn, int H, int heads) {
    const size_t N = (size_t)nseq * Tn, NT = cdiv(Tn, 16);
    return 2 * ws_align(N * heads * sizeof(float)) + ws_align((size_t)nseq * 2 * H * sizeof(float)) + ws_align((size_t)nseq * 32 * NT * H * sizeof(float));
}
template <class T, int DH>
static int nb_relpos_bwd(long nseq, const RelBwd& a0, float* dpos, float* du, float* dvb, void* ws, hipStream_t st) {
    RelBwd a = a0;
    const int Tn = a.Tn, H = a.H, heads = a.heads, NW = GB_THREADS / 64;
    const size_t N = (size_t)nseq * Tn, NT = cdiv(Tn, 16);
    char* w = (char*)ws;
    a.lse = (float*)w; w += ws_align(N * heads * sizeof(float));
    a.Dv = (float*)w; w += ws_align(N * heads * sizeof(float));
    a.duv_part = (float*)w; w += ws_align((size_t)nseq * 2 * H * sizeof(float));
    a.dpos_part = (float*)w;
    const size_t TPq = 32 * cdiv(Tn, 32), TP = TPq + 16, RPP = rel_rpp(Tn), MW = (Tn + 31) / 32, mk = a.mask ? (((size_t)Tn * MW + 3) & ~(size_t)3) * 4 : 0;
    const size_t lq = (2 * TPq + RPP) * DH * sizeof(T) + NW * 512 * sizeof(float) + NW * 512 * sizeof(T) + NW * 2 * 64 * sizeof(float);
    const size_t lk = (3 * TP + RPP) * DH * sizeof(T) + 2 * TP * sizeof(float) + NW * 512 * sizeof(float) + mk;
    const size_t lr = (TP + RPP) * DH * sizeof(T) + 2 * TP * sizeof(float) + NW * 512 * sizeof(float) + mk + NW * 512 * sizeof(T);
    if (Tn > GA_TMAX || lq > 160 * 1024 || lk > 160 * 1024 || lr > 160 * 1024) return NBSS_EUNSUPPORTED;
    int e;
    if ((e = NBSS_SET_MAX_LDS((rel_bwd_q_kernel<T, DH>), lq))) return e;
    if ((e = NBSS_SET_MAX_LDS((rel_bwd_k_kernel<T, DH>), lk))) return e;
    if ((e = NBSS_SET_MAX_LDS((rel_bwd_r_kernel<T, DH>), lr))) return e;
    const dim3 grid((unsigned)nseq, heads), block(GB_THREADS);
    NBSS_LAUNCH((rel_bwd_q_kernel<T, DH>), grid, block, lq, st, a);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    NBSS_LAUNCH((rel_bwd_k_kernel<T, DH>), grid, block, lk, st, a);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    NBSS_LAUNCH((rel_bwd_r_kernel<T, DH>), grid, block, lr, st, a);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    const long nel = (long)(2 * Tn - 1) * H + 2 * H;
    NBSS_LAUNCH(rel_fold_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, (const float*)a.dpos_part, (const float*)a.duv_part, nseq, Tn, H, heads, dpos, du, dvb);
    return NBSS_CHECK_LAUNCH();
}
int nb_attention_relpos_bwd_impl(int dtype, long nseq, int Tn, int H, int heads, const void* qkv, const void* pos, const float* ub, const float* vb, float scale,
                                 const uint32_t* mask, float keep, const void* dO, void* dqkv, float* dpos, float* du, float* dvb, void* ws, hipStream_t st) {
    if (heads <= 0 || H % heads) return NBSS_EINVAL;
    const int dh = H / heads;
    RelBwd a = {qkv, pos, dO, ub, vb, mask, dqkv, nullptr, nullptr, nullptr, nullptr, scale, keep, Tn, H, heads};
    if (dh == 48) return dtype == NBSS_BF16 ? nb_relpos_bwd<bf16_t, 48>(nseq, a, dpos, du, dvb, ws, st) : nb_relpos_bwd<float, 48>(nseq, a, dpos, du, dvb, ws, st);
    if (dh == 24) return dtype == NBSS_BF16 ? nb_relpos_bwd<bf16_t, 24>(nseq, a, dpos, du, dvb, ws, st) : nb_relpos_bwd<float, 24>(nseq, a, dpos, du, dvb, ws, st);
    return NBSS_EUNSUPPORTED;
}
// GroupNorm forward that keeps its (mean, rstd) per (sequence, group) for nb_group_norm_bwd_impl (dx in place of dy; dgamma / dbeta accumulated)
int nb_group_norm_train_impl(int dtype, long nseq, int Tn, int C, int groups, const void* x, const float* gamma, const float* beta, int act, void* y, float* stats,
                             hipStream_t st) {
    if (groups <= 0 || C % groups) return NBSS_EINVAL;
    const int CG = C / groups;
    if (dtype == NBSS_BF16)
        NBSS_LAUNCH((gb_gn_fwd_kernel<bf16_t>), dim3((unsigned)(nseq * groups)), dim3(GB_THREADS), 8 * sizeof(float), st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, stats, Tn, C, CG, act);
    else
        NBSS_LAUNCH((gb_gn_fwd_kernel<float>), dim3((unsigned)(nseq * groups)), dim3(GB_THREADS), 8 * sizeof(float), st, (const float*)x, gamma, beta, (float*)y, stats, Tn, C, CG, act);
    return NBSS_CHECK_LAUNCH();
}
int nb_group_norm_bwd_impl(int dtype, long nseq, int Tn, int C, int groups, const void* x, const float* stats, const float* gamma, const float* beta, void* dy_dx,
                           float* dgamma, float* dbeta, hipStream_t st) {
    if (groups <= 0 || C % groups) return NBSS_EINVAL;
    const int CG = C / groups;
    if (CG > 64) return NBSS_EUNSUPPORTED;
    const size_t lds = (8 + 128) * sizeof(float);
    if (dtype == NBSS_BF16)
        NBSS_LAUNCH((gb_gn_bwd_kernel<bf16_t>), dim3((unsigned)(nseq * groups)), dim3(GB_THREADS), lds, st, (const bf16_t*)x, stats, gamma, beta, (bf16_t*)dy_dx, dgamma, dbeta, Tn, C, CG);
    else
        NBSS_LAUNCH((gb_gn_bwd_kernel<float>), dim3((unsigned)(nseq * groups)), dim3(GB_THREADS), lds, st, (const float*)x, stats, gamma, beta, (float*)dy_dx, dgamma, dbeta, Tn, C, CG);
    return NBSS_CHECK_LAUNCH();
}
// GroupNorm(groups, C) over (C / groups x T) per sequence (eps 1e-5), optional SiLU: x, y [nseq][T][C]
int nb_group_norm_impl(int dtype, long nseq, int Tn, int C, int groups, const void* x, const float* gamma, const float* beta, int act, void* y, hipStream_t st) {
    if (groups <= 0 || C % groups) return NBSS_EINVAL;
    const int CG = C / groups;
    if (dtype == NBSS_BF16)
        NBSS_LAUNCH((gb_gn_fwd_kernel<bf16_t>), dim3((unsigned)(nseq * groups)), dim3(GB_THREADS), 8 * sizeof(float), st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, (float*)nullptr, Tn, C, CG, act);
    else
        NBSS_LAUNCH((gb_gn_fwd_kernel<float>), dim3((unsigned)(nseq * groups)), dim3(GB_THREADS), 8 * sizeof(float), st, (const float*)x, gamma, beta, (float*)y, (float*)nullptr, Tn, C, CG, act);
    return NBSS_CHECK_LAUNCH();
}

// ---- training-mode building blocks (nbss_nb_*_train / _bwd: include/nbss_hip.h) ------------------------------------------------------------
// ws layout of the backward calls: [re-laid weights: nb_ws_bytes_impl()] [WGPART_BYTES of weight-gradient partial tiles]
size_t nb_bwd_ws_bytes_impl(int M, int K, int groups, int taps) {
    const size_t a = nb_ws_bytes_impl(M, K, groups, taps), b = nb_ws_bytes_impl(K, M, groups, taps);
    return (a > b ? a : b) + ws_align(WGPART_BYTES);
}
template <class T>
static int nb_conv_t_train(long nseq, int Tn, int Cin, int ldx, int Cout, int groups, int taps, const void* x, const float* w, const float* bias, void* y, void* y2,
                           const void* residual, void* ws, hipStream_t st) {
    if (groups <= 0 || Cin % groups || Cout % groups || (groups > 1 && ldx != Cin)) return NBSS_EINVAL;
    const int Kv = Cin / groups, Kg = groups > 1 ? Kv : pad8(Kv), Mg = Cout / groups;
    if (Kg % 8 || ldx < (groups > 1 ? Cin : Kg)) return NBSS_EUNSUPPORTED;
    int e = gb_wprep<T>(w, ws, taps > 1 || groups > 1 ? WP_CONV_FWD : WP_LIN_FWD, groups, taps, Mg, Kv, pad16(Mg), pad32(Kg), st);
    if (e) return e;
    TapGemm p = gb_lin(x, ldx, ws, bias, y, Cout, nseq * Tn, Mg, Kg);
    p.groups = groups; p.xgs = groups > 1 ? Kv : 0; p.ygs = groups > 1 ? Mg : 0; p.bgs = Mg;
    p.taps = taps; p.center = taps / 2; p.shift = 1; p.pos_div = 1; p.pos_len = Tn;
    p.R = residual; p.ldr = Cout;
    p.Y2 = y2;
    return gb_gemm<T>(p, st);
}
int nb_conv_t_train_impl(int dtype, long nseq, int Tn, int Cin, int ldx, int Cout, int groups, int taps, const void* x, const float* w, const float* bias, void* y,
                         void* y2, const void* residual, void* ws, hipStream_t st) {
    return dtype == NBSS_BF16 ? nb_conv_t_train<bf16_t>(nseq, Tn, Cin, ldx, Cout, groups, taps, x, w, bias, y, y2, residual, ws, st)
                              : nb_conv_t_train<float>(nseq, Tn, Cin, ldx, Cout, groups, taps, x, w, bias, y, y2, residual, ws, st);
}
// data gradient (dx [N][Cin] = conv^T(dy), optionally times SiLU'(dact)) and weight / bias gradient (dw [Cout][Cin / groups][taps] += dy^T x) of
// y = conv(x): x [N][ldx] is the tensor the forward call read (valid columns Cin), dy [N][Cout]
template <class T>
static int nb_conv_t_bwd(int dtype, long nseq, int Tn, int Cin, int ldx, int Cout, int groups, int taps, const void* x, const float* w, const void* dy, const void* dact,
                         void* dx, float* dw, float* dbias, void* ws, hipStream_t st) {
    if (groups <= 0 || Cin % groups || Cout % groups || (groups > 1 && ldx != Cin)) return NBSS_EINVAL;
    const long N = nseq * Tn;
    const int Kv = Cin / groups, Mg = Cout / groups;
    int e;
    if (dx) {
        // the transposed map: outputs = the forward's inputs (Cin, written ldx wide: padding columns get zero weight rows), K = Cout
        if (Mg % 8 || (groups == 1 && ldx % 4)) return NBSS_EUNSUPPORTED;
        const int Mo = groups > 1 ? Kv : ldx;  // rows of the re-laid weight per group: valid Kv, the rest zero
        if ((e = gb_wprep<T>(w, ws, taps > 1 || groups > 1 ? WP_CONV_DGRAD : WP_LIN_DGRAD, groups, taps, Kv, Mg, pad16(Mo), pad32(Mg), st))) return e;
        TapGemm p = gb_lin(dy, Cout, ws, nullptr, dx, ldx, N, Mo, Mg);
        p.groups = groups; p.xgs = groups > 1 ? Mg : 0; p.ygs = groups > 1 ? Kv : 0; p.bgs = 0;
        p.taps = taps; p.center = taps / 2; p.shift = 1; p.pos_div = 1; p.pos_len = Tn;
        p.Dact = dact;
        if ((e = gb_gemm<T>(p, st))) return e;
    }
    if (dw) {
        if (Kv % 4 || Mg % 4) return NBSS_EUNSUPPORTED;
        float* part = (float*)((char*)ws + nb_bwd_ws_bytes_impl(Cout, Cin, groups, taps) - ws_align(WGPART_BYTES));
        const size_t esz = sizeof(T);
        // dense problems in row slices that fit one workgroup of the transposing-read kernel (gb_wgrad_dense's rule); grouped convs as one problem
        int mt = groups > 1 ? Cout / 16 + 1 : 112 / cdiv(Cin, 16);
        if (groups == 1) {
            if (mt > 12) mt = 12;
            while (mt > 1 && cdiv(mt * 16, 64) + cdiv(Cin, 64) > 7) --mt;
            if (mt < 1) mt = 1;
        }
        const int ms = groups > 1 ? Cout : mt * 16;
        for (int m0 = 0; m0 < Cout; m0 += ms) {
            const int mm = Cout - m0 < ms ? Cout - m0 : ms;
            WgradArgs a;
            a.part = part;
            a.mvalid = 0; a.nvalid = 0;
            a.Ntok = (int)N; a.F = (int)nseq; a.T = Tn; a.shift_stride = 1; a.shift_dim = 0;
            a.groups = groups; a.taps = taps;
            a.stats = nullptr; a.gamma = nullptr; a.beta = nullptr;
            a.A = (const char*)dy + (size_t)m0 * esz; a.lda = Cout; a.MA = mm;
            a.B = x; a.ldb = ldx; a.NB = Cin;
            a.dW = dw + (size_t)m0 * Kv * taps; a.dbias = dbias ? dbias + m0 : nullptr;
            if ((e = wgrad_launch(a, dtype, st))) return e;
        }
    }
    return NBSS_OK;
}
int nb_conv_t_bwd_impl(int dtype, long nseq, int Tn, int Cin, int ldx, int Cout, int groups, int taps, const void* x, const float* w, const void* dy, const void* dact,
                       void* dx, float* dw, float* dbias, void* ws, hipStream_t st) {
    return dtype == NBSS_BF16 ? nb_conv_t_bwd<bf16_t>(dtype, nseq, Tn, Cin, ldx, Cout, groups, taps, x, w, dy, dact, dx, dw, dbias, ws, st)
                              : nb_conv_t_bwd<float>(dtype, nseq, Tn, Cin, ldx, Cout, groups, taps, x, w, dy, dact, dx, dw, dbias, ws, st);
}
int nb_layernorm_bwd_impl(int dtype, long rows, int C, const void* x, const float* stats, const float* gamma, const void* du, const void* dres, void* dx, float* dgamma,
                          float* dbeta, hipStream_t st) {
    return dtype == NBSS_BF16 ? gb_ln_bwd<bf16_t>(du, x, stats, gamma, dres, dx, dgamma, dbeta, rows, C, st)
                              : gb_ln_bwd<float>(du, x, stats, gamma, dres, dx, dgamma, dbeta, rows, C, st);
}
int nb_gbn_bwd_impl(int dtype, int B, int F, int Tn, int C, const void* x, const float* gamma, const float* beta, float eps, int act, const void* dy, void* dx,
                    float* dgamma, float* dbeta, hipStream_t st) {
    const size_t lds = (8 + 2 * (size_t)C) * sizeof(float);
    if (dtype == NBSS_BF16)
        NBSS_LAUNCH((gb_gbn_bwd_kernel<bf16_t>), dim3(B * Tn), dim3(GB_THREADS), lds, st, (const bf16_t*)x, gamma, beta, (const bf16_t*)dy, (bf16_t*)dx, dgamma, dbeta, F, Tn,
                    C, eps, act);
    else
        NBSS_LAUNCH((gb_gbn_bwd_kernel<float>), dim3(B * Tn), dim3(GB_THREADS), lds, st, (const float*)x, gamma, beta, (const float*)dy, (float*)dx, dgamma, dbeta, F, Tn, C,
                    eps, act);
    return NBSS_CHECK_LAUNCH();
}
// attention backward from the packed projections: qkv [N][3H] (q | k | v), dO [N][H] -> dqkv [N][3H].  ws: O [N][H] (recomputed) | lse, D [N][heads] fp32
size_t nb_attn_bwd_ws_bytes_impl(long N, int H, int heads, int dtype) {
    return ws_align((size_t)N * H * (dtype == NBSS_BF16 ? 2 : 4)) + 2 * ws_align((size_t)N * heads * sizeof(float));
}
int nb_attention_bwd_impl(int dtype, long nseq, int Tn, int H, int heads, const void* qkv, const void* dO, void* dqkv, void* ws, hipStream_t st) {
    if (heads <= 0 || H % heads) return NBSS_EINVAL;
    nbss_cfg c = {};
    c.B = 1; c.F = (int)nseq; c.T = Tn; c.H = H; c.heads = heads; c.dtype = dtype;
    const long N = nseq * Tn;
    void* O = ws;
    float* lse = (float*)((char*)ws + ws_align((size_t)N * H * (dtype == NBSS_BF16 ? 2 : 4)));
    float* Dv = (float*)((char*)lse + ws_align((size_t)N * heads * sizeof(float)));
    const int dh = H / heads;
    if (dh == 48) return dtype == NBSS_BF16 ? gb_attn_launch<bf16_t, 48>(c, qkv, dO, O, dqkv, lse, Dv, st) : gb_attn_launch<float, 48>(c, qkv, dO, O, dqkv, lse, Dv, st);
    if (dh == 24) return dtype == NBSS_BF16 ? gb_attn_launch<bf16_t, 24>(c, qkv, dO, O, dqkv, lse, Dv, st) : gb_attn_launch<float, 24>(c, qkv, dO, O, dqkv, lse, Dv, st);
    return NBSS_EUNSUPPORTED;
}
