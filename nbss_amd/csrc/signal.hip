// signal.hip — fp32 signal front/back end of the training step (always fp32, like the reference
// forces under autocast: models/io/stft.py:59-61,79-81):
//   stft_norm      x[B,C,N] -> STFT (torch.stft center=True, reflect pad, periodic hann: stft.py:49-66)
//                  -> per-T-F-bin magnitude normalisation by the reference channel
//                     (Norm('frequency', online=True): XrMM = |Xr| + 1e-6, norm.py:77-81,94)
//                  -> [B,F,T,2C] real layout of TrainModule.forward (SharedTrainer.py:116-117)
//   inorm_istft    out[B,F,T,2Spk] * XrMM (norm.py:107-108) -> iSTFT per (b,spk) (stft.py:68-97):
//                  irfft, window, overlap-add, / window envelope, trim the centre padding
//   ..._bwd        adjoint of the above (linear in `out`)
// Both transforms are DFT-as-GEMM on the exact-f32 matrix cores (v_mfma_f32_16x16x4_f32) with the
// windowed DFT matrices pre-packed as A fragments ("tables", built once by nbss_stft_tables);
// frames are the MFMA N dimension, so a lane holds (re,im) pairs of one frame.
#include "launch.h"
#include "layout.h"
#include "prof.h"

#define SG_PI 3.14159265358979323846

struct StftGeom {
    int nfft, hop, F, MTq, KSm, MTm, KSq;
};
NBSS_HD StftGeom stft_geom(int nfft) {
    StftGeom g;
    g.nfft = nfft; g.hop = nfft / 2; g.F = nfft / 2 + 1;
    g.MTq = cdiv(2 * g.F, 16); g.KSm = nfft / 32; g.MTm = nfft / 16; g.KSq = cdiv(2 * g.F, 32);
    return g;
}
NBSS_HD size_t stft_tables_floats(int nfft) {
    StftGeom g = stft_geom(nfft);
    return (size_t)nfft + (size_t)2 * g.MTq * g.KSm * 512 + (size_t)g.MTm * g.KSq * 512;
}

NBSS_DEV float window_val(int nfft, int kind, int m) {
    const double h = 0.5 - 0.5 * cos(2.0 * SG_PI * m / nfft);  // periodic hann (torch.hann_window)
    return (float)(kind == 0 ? h : sqrt(h));
}

// tables: [window nfft][D pack: STFT  rows q=(f,ri), K = m][ET pack: iSTFT adjoint rows q, K = m][E pack: iSTFT rows m, K = q]
__global__ void stft_tables_kernel(int nfft, int kind, float* __restrict__ tab) {
    const StftGeom g = stft_geom(nfft);
    float* win = tab;
    float* Dp = tab + nfft;
    float* ETp = Dp + (size_t)g.MTq * g.KSm * 512;
    float* Ep = ETp + (size_t)g.MTq * g.KSm * 512;
    const size_t nD = (size_t)g.MTq * g.KSm * 512, nE = (size_t)g.MTm * g.KSq * 512;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nfft + 2 * nD + nE; e += (size_t)gridDim.x * blockDim.x) {
        if (e < (size_t)nfft) { win[e] = window_val(nfft, kind, (int)e); continue; }
        size_t i = e - nfft;
        const int which = i < nD ? 0 : (i < 2 * nD ? 1 : 2);
        if (which == 1) i -= nD;
        if (which == 2) i -= 2 * nD;
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63), l15 = lane & 15, g4 = lane >> 4;
        size_t r = i >> 9;
        const int KS = which == 2 ? g.KSq : g.KSm;
        const int ks = (int)(r % KS), mt = (int)(r / KS);
        const int row = mt * 16 + l15, k = ks * 32 + 8 * g4 + j;
        const int q = which == 2 ? k : row, m = which == 2 ? row : k;
        float v = 0.f;
        if (q < 2 * g.F && m < nfft) {
            const int f = q >> 1, ri = q & 1;
            const int ph = (int)(((long)f * m) % nfft);
            const double ang = 2.0 * SG_PI * ph / nfft;
            const double w = window_val(nfft, kind, m);
            if (which == 0) {
                v = (float)(w * (ri == 0 ? cos(ang) : -sin(ang)));
            } else {
                // irfft: x[m] = (1/N) sum_f c_f (Re Y_f cos - Im Y_f sin), c_0 = c_{N/2} = 1 (their imaginary parts are ignored)
                const bool edge = f == 0 || f == nfft / 2;
                const double cf = edge ? 1.0 : 2.0;
                v = (ri == 1 && edge) ? 0.f : (float)(w * cf / nfft * (ri == 0 ? cos(ang) : -sin(ang)));
            }
        }
        (which == 0 ? Dp : (which == 1 ? ETp : Ep))[i] = v;
    }
}

NBSS_DEV int reflect_idx(int n, int N) {
    if (n < 0) n = -n;
    if (n >= N) n = 2 * (N - 1) - n;
    return n;
}

// one wave = 16 frames of one batch item, all channels; output [B,F,T,2C] normalised + XrMM [B,F,T]
template <class T, int NFFT>
__global__ __launch_bounds__(256) void stft_norm_kernel(int B, int C, int N, int Tn, int ref, const float* __restrict__ tab,
                                                        const float* __restrict__ x, T* __restrict__ X, float* __restrict__ xrmm) {
    constexpr int HOP = NFFT / 2, F = NFFT / 2 + 1, KSm = NFFT / 32, MTq = (2 * F + 15) / 16;
    const float* Dp = tab + NFFT;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    const int nst = cdiv(Tn, 16), ntask = B * nst * MTq;
    const int wpb = blockDim.x >> 6;
    for (int task = blockIdx.x * wpb + wave_id(); task < ntask; task += gridDim.x * wpb) {
        const int mt = task % MTq, st = (task / MTq) % nst, b = task / (MTq * nst);
        const int t = st * 16 + l15;
        Frag<float> a[KSm];
#pragma unroll
        for (int ks = 0; ks < KSm; ++ks) wfrag_load(a[ks], Dp, mt, KSm, ks);
        float mm[2] = {1.f, 1.f};
        for (int cc = 0; cc < C; ++cc) {
            const int c = (ref + cc) % C;  // reference channel first: its magnitude normalises the others
            const float* xc = x + ((size_t)b * C + c) * N;
            f32x4 acc = F32X4_ZERO;
#pragma unroll
            for (int ks = 0; ks < KSm; ++ks) {
                Frag<float> bq;
                const int p0 = t * HOP + ks * 32 + 8 * g4 - NFFT / 2;  // first sample of this lane's 8 (centre padding removed)
                if (t < Tn) {
                    if (p0 >= 0 && p0 + 8 <= N) {
                        load8(xc + p0, bq.v);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) bq.v[j] = xc[reflect_idx(p0 + j, N)];
                    }
                } else {
                    frag_zero(bq);
                }
                acc = mma(a[ks], bq, acc);
            }
            // lane: frame t, rows q = 16mt + 4g4 + {0,1,2,3} = (f0,re) (f0,im) (f0+1,re) (f0+1,im)
            const int f0 = (16 * mt + 4 * g4) >> 1;
            if (cc == 0) {
                mm[0] = sqrtf(acc[0] * acc[0] + acc[1] * acc[1]) + 1e-6f;
                mm[1] = sqrtf(acc[2] * acc[2] + acc[3] * acc[3]) + 1e-6f;
                if (t < Tn) {
                    if (f0 < F) xrmm[((size_t)b * F + f0) * Tn + t] = mm[0];
                    if (f0 + 1 < F) xrmm[((size_t)b * F + f0 + 1) * Tn + t] = mm[1];
                }
            }
            if (t < Tn) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int f = f0 + e;
                    if (f < F) {
                        T* o = X + (((size_t)b * F + f) * Tn + t) * (2 * C) + 2 * c;
                        store1(o, acc[2 * e] / mm[e]);
                        store1(o + 1, acc[2 * e + 1] / mm[e]);
                    }
                }
            }
        }
    }
}

NBSS_DEV float ola_env(const float* __restrict__ win, int nfft, int hop, int Tn, int p) {
    // sum_t w^2[p - t hop] over the frames that cover padded sample p (hop = nfft/2: at most two)
    float e = 0.f;
    const int t1 = p / hop;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int t = t1 - d, m = p - t * hop;
        if (t >= 0 && t < Tn && m >= 0 && m < nfft) e += win[m] * win[m];
    }
    return e;
}

// out[B,F,T,2Spk] (fp32) * XrMM -> frames -> overlap-add into ybuf[B,Spk,(T+1)*hop] (zeroed by the caller)
template <int NFFT>
__global__ __launch_bounds__(256) void inorm_istft_kernel(int B, int S, int Tn, const float* __restrict__ tab, const float* __restrict__ out,
                                                          const float* __restrict__ xrmm, float* __restrict__ ybuf, int np) {
    // np: the output-row tiles of a task are dealt to np waves (small batches: B S T/16 tasks alone are 64 waves at batch 2 on 1024 SIMDs)
    constexpr int HOP = NFFT / 2, F = NFFT / 2 + 1, MTm = NFFT / 16, KSq = (2 * F + 31) / 32;
    const StftGeom g = stft_geom(NFFT);
    const float* Ep = tab + NFFT + (size_t)2 * g.MTq * g.KSm * 512;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    const int nst = cdiv(Tn, 16), ntask = B * S * nst * np;
    const int LP = (Tn + 1) * HOP;
    const int wpb = blockDim.x >> 6;
    for (int task0 = blockIdx.x * wpb + wave_id(); task0 < ntask; task0 += gridDim.x * wpb) {
        const int part = task0 % np, task = task0 / np;
        const int st = task % nst, s = (task / nst) % S, b = task / (nst * S);
        const int t = st * 16 + l15;
        Frag<float> bq[KSq];
#pragma unroll
        for (int ks = 0; ks < KSq; ++ks) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = (ks * 32 + 8 * g4) / 2 + e;
                float re = 0.f, im = 0.f;
                if (t < Tn && f < F) {
                    const size_t n = ((size_t)b * F + f) * Tn + t;
                    const float mmv = xrmm[n];
                    const f32x2 v = *reinterpret_cast<const f32x2*>(out + n * (2 * S) + 2 * s);
                    re = v[0] * mmv;
                    im = v[1] * mmv;
                }
                bq[ks].v[2 * e] = re;
                bq[ks].v[2 * e + 1] = im;
            }
        }
        float* yb = ybuf + ((size_t)b * S + s) * LP;
        for (int mt = part; mt < MTm; mt += np) {
            f32x4 acc = F32X4_ZERO;
#pragma unroll
            for (int ks = 0; ks < KSq; ++ks) {
                Frag<float> a;
                wfrag_load(a, Ep, mt, KSq, ks);
                acc = mma(a, bq[ks], acc);
            }
            if (t < Tn) {
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(yb + t * HOP + 16 * mt + 4 * g4 + r, acc[r]);
            }
        }
    }
}

// y[b,s,n] = ybuf[b,s,n + nfft/2] / envelope
__global__ void istft_finalize_kernel(int BS, int N, int Tn, int nfft, const float* __restrict__ tab, const float* __restrict__ ybuf,
                                      float* __restrict__ y) {
    const int hop = nfft / 2, LP = (Tn + 1) * hop;
    const size_t total = (size_t)BS * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const size_t bs = i / N;
        const int p = n + nfft / 2;
        float v = 0.f;
        if (p < LP) {
            const float e = ola_env(tab, nfft, hop, Tn, p);
            v = e > 1e-11f ? ybuf[bs * LP + p] / e : 0.f;
        }
        y[i] = v;
    }
}

// adjoint: dy[B,Spk,N] -> dout[B,F,T,2Spk] = XrMM * E^T (window-framed dy / envelope)
template <int NFFT>
__global__ __launch_bounds__(256) void inorm_istft_bwd_kernel(int B, int S, int N, int Tn, const float* __restrict__ tab,
                                                              const float* __restrict__ dy, const float* __restrict__ xrmm,
                                                              float* __restrict__ dout, int np) {
    constexpr int HOP = NFFT / 2, F = NFFT / 2 + 1, KSm = NFFT / 32, MTq = (2 * F + 15) / 16;
    const float* ETp = tab + NFFT + (size_t)MTq * KSm * 512;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    const int nst = cdiv(Tn, 16), ntask = B * S * nst * np;
    const int LP = (Tn + 1) * HOP;
    const int wpb = blockDim.x >> 6;
    for (int task0 = blockIdx.x * wpb + wave_id(); task0 < ntask; task0 += gridDim.x * wpb) {
        const int part = task0 % np, task = task0 / np;
        const int st = task % nst, s = (task / nst) % S, b = task / (nst * S);
        const int t = st * 16 + l15;
        const float* dyb = dy + ((size_t)b * S + s) * N;
        Frag<float> bq[KSm];
#pragma unroll
        for (int ks = 0; ks < KSm; ++ks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = t * HOP + ks * 32 + 8 * g4 + j, n = p - NFFT / 2;
                float v = 0.f;
                if (t < Tn && n >= 0 && n < N && p < LP) {
                    const float e = ola_env(tab, NFFT, HOP, Tn, p);
                    v = e > 1e-11f ? dyb[n] / e : 0.f;
                }
                bq[ks].v[j] = v;
            }
        }
        for (int mt = part; mt < MTq; mt += np) {
            f32x4 acc = F32X4_ZERO;
#pragma unroll
            for (int ks = 0; ks < KSm; ++ks) {
                Frag<float> a;
                wfrag_load(a, ETp, mt, KSm, ks);
                acc = mma(a, bq[ks], acc);
            }
            if (t < Tn) {
                const int f0 = (16 * mt + 4 * g4) >> 1;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int f = f0 + e;
                    if (f < F) {
                        const size_t n = ((size_t)b * F + f) * Tn + t;
                        const float mmv = xrmm[n];
                        f32x2 v = {acc[2 * e] * mmv, acc[2 * e + 1] * mmv};
                        *reinterpret_cast<f32x2*>(dout + n * (2 * S) + 2 * s) = v;
                    }
                }
            }
        }
    }
}

int memset_async_impl(void* p, size_t bytes, hipStream_t st);

int stft_tables_impl(int nfft, int win_kind, float* tab, hipStream_t st) {
    if (nfft != 256 && nfft != 512) return NBSS_EUNSUPPORTED;
    NBSS_LAUNCH(stft_tables_kernel, dim3(64), dim3(256), 0, st, nfft, win_kind, tab);
    return NBSS_CHECK_LAUNCH();
}

static int grid_for(int ntask) { return cdiv(ntask, 4) < 4096 ? cdiv(ntask, 4) : 4096; }

int stft_norm_impl(int nfft, int dtype, int B, int C, int N, int ref, const float* tab, const float* x, void* X, float* xrmm, hipStream_t st) {
    if (ref < 0 || ref >= C || N < nfft) return NBSS_EINVAL;
    const int Tn = N / (nfft / 2) + 1, MTq = cdiv(nfft + 2, 16);
    ProfScope ps(PK_STFT, st);
    dim3 grid(grid_for(B * cdiv(Tn, 16) * MTq)), block(256);
    if (nfft == 256) {
        if (dtype == NBSS_BF16) NBSS_LAUNCH((stft_norm_kernel<bf16_t, 256>), grid, block, 0, st, B, C, N, Tn, ref, tab, x, (bf16_t*)X, xrmm);
        else NBSS_LAUNCH((stft_norm_kernel<float, 256>), grid, block, 0, st, B, C, N, Tn, ref, tab, x, (float*)X, xrmm);
    } else if (nfft == 512) {
        if (dtype == NBSS_BF16) NBSS_LAUNCH((stft_norm_kernel<bf16_t, 512>), grid, block, 0, st, B, C, N, Tn, ref, tab, x, (bf16_t*)X, xrmm);
        else NBSS_LAUNCH((stft_norm_kernel<float, 512>), grid, block, 0, st, B, C, N, Tn, ref, tab, x, (float*)X, xrmm);
    } else {
        return NBSS_EUNSUPPORTED;
    }
    return NBSS_CHECK_LAUNCH();
}

// waves a (b, speaker, 16-frame tile) task of the iSTFT kernels is dealt to: 1 / 2 / 4 / 8, until the launch has a wave per SIMD
static int istft_parts(int ntask) {
    int np = 1;
    while (np < 8 && ntask * np < 1024) np *= 2;
    return np;
}

int inorm_istft_impl(int nfft, int B, int S, int N, const float* tab, const float* out, const float* xrmm, float* ybuf, float* y, hipStream_t st) {
    const int Tn = N / (nfft / 2) + 1;
    const size_t LP = (size_t)(Tn + 1) * (nfft / 2);
    ProfScope ps(PK_ISTFT, st);
    int e = memset_async_impl(ybuf, (size_t)B * S * LP * sizeof(float), st);
    if (e) return e;
    const int np = istft_parts(B * S * cdiv(Tn, 16));
    dim3 grid(grid_for(B * S * cdiv(Tn, 16) * np)), block(256);
    if (nfft == 256) NBSS_LAUNCH((inorm_istft_kernel<256>), grid, block, 0, st, B, S, Tn, tab, out, xrmm, ybuf, np);
    else if (nfft == 512) NBSS_LAUNCH((inorm_istft_kernel<512>), grid, block, 0, st, B, S, Tn, tab, out, xrmm, ybuf, np);
    else return NBSS_EUNSUPPORTED;
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    const size_t total = (size_t)B * S * N;
    NBSS_LAUNCH(istft_finalize_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, st, B * S, N, Tn, nfft,
                tab, ybuf, y);
    return NBSS_CHECK_LAUNCH();
}

int inorm_istft_bwd_impl(int nfft, int B, int S, int N, const float* tab, const float* dy, const float* xrmm, float* dout, hipStream_t st) {
    const int Tn = N / (nfft / 2) + 1;
    ProfScope ps(PK_ISTFT_B, st);
    const int np = istft_parts(B * S * cdiv(Tn, 16));
    dim3 grid(grid_for(B * S * cdiv(Tn, 16) * np)), block(256);
    if (nfft == 256) NBSS_LAUNCH((inorm_istft_bwd_kernel<256>), grid, block, 0, st, B, S, N, Tn, tab, dy, xrmm, dout, np);
    else if (nfft == 512) NBSS_LAUNCH((inorm_istft_bwd_kernel<512>), grid, block, 0, st, B, S, N, Tn, tab, dy, xrmm, dout, np);
    else return NBSS_EUNSUPPORTED;
    return NBSS_CHECK_LAUNCH();
}

size_t stft_tables_bytes_impl(int nfft) { return stft_tables_floats(nfft) * sizeof(float); }
