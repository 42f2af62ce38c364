// encdec.hip — SpatialNet encoder (nn.Conv1d(C_in,H,5,'same') along T, SpatialNet.py:175,205)
// and decoder (nn.Linear(H,C_out), SpatialNet.py:200,216), forward and backward.
//
// One wave owns a strip of 16 consecutive frames of one (b,f) sequence; the weights are the
// MFMA A operand (form 2), so a lane ends up with 4 consecutive channels of one frame and the
// [B,F,T,H] stream is written with 8/16-byte stores.
#include "launch.h"
#include "layout.h"
#include "wgrad.h"
#include "side.h"
#include "prof.h"
#include "geom.h"

#define ENC_H 96  // backward (training) kernels: SpatialNet-small

template <class T, int H>  // H = dim_hidden (geom.h)
__global__ __launch_bounds__(256) void encoder_fwd_kernel(nbss_cfg c, const float* __restrict__ P, const T* __restrict__ Wp,
                                                          const T* __restrict__ xin, T* __restrict__ y) {
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    const int T_ = c.T, Cin = c.C_in, pc = Cin / 4, np = c.enc_ks * pc, half = c.enc_ks / 2;
    const int nst = cdiv(T_, 16);
    const int nstrips = c.B * c.F * nst;
    constexpr int MT = H / 16;
    const int KS = cdiv(np, 8);  // <= 3 for C_in <= 16 (checked by host)
    const float* bias = P + param_off_enc_b(c);
    const int wpb = blockDim.x >> 6;
    for (int s = blockIdx.x * wpb + wave_id(); s < nstrips; s += gridDim.x * wpb) {
        const int bf = s / nst, t0 = (s % nst) * 16;
        const T* xb = xin + (size_t)bf * T_ * Cin;
        f32x4 acc[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i] = F32X4_ZERO;
        for (int ks = 0; ks < KS; ++ks) {
            Frag<T> b;
            frag_zero(b);
            const int p0 = ks * 8 + g4 * 2, p1 = p0 + 1;
            if (p0 < np) {
                const int t = t0 + l15 + p0 / pc - half;
                if (t >= 0 && t < T_) frag_load_lo(b, xb + (size_t)t * Cin + (p0 % pc) * 4);
            }
            if (p1 < np) {
                const int t = t0 + l15 + p1 / pc - half;
                if (t >= 0 && t < T_) frag_load_hi(b, xb + (size_t)t * Cin + (p1 % pc) * 4);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                Frag<T> a;
                wfrag_load(a, Wp, i, KS, ks);
                acc[i] = mma(a, b, acc[i]);
            }
        }
        const int t = t0 + l15;
        if (t < T_) {
            T* yr = y + ((size_t)bf * T_ + t) * H;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int ch = 16 * i + 4 * g4;
                store4(yr + ch, acc[i][0] + bias[ch], acc[i][1] + bias[ch + 1], acc[i][2] + bias[ch + 2], acc[i][3] + bias[ch + 3]);
            }
        }
    }
}

template <class T, int H>
__global__ __launch_bounds__(256) void decoder_fwd_kernel(nbss_cfg c, const float* __restrict__ bias, const T* __restrict__ Wp,
                                                          const T* __restrict__ x, float* __restrict__ out) {
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    const int T_ = c.T, Co = c.C_out;
    const int nst = cdiv(T_, 16);
    const int nstrips = c.B * c.F * nst;
    constexpr int KS = H / 32;
    Frag<T> a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wfrag_load(a[ks], Wp, 0, KS, ks);
    const int wpb = blockDim.x >> 6;
    for (int s = blockIdx.x * wpb + wave_id(); s < nstrips; s += gridDim.x * wpb) {
        const int bf = s / nst, t = (s % nst) * 16 + l15;
        f32x4 acc = F32X4_ZERO;
        const T* xr = x + ((size_t)bf * T_ + t) * H;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            Frag<T> b;
            if (t < T_) frag_load(b, xr + ks * 32 + 8 * g4);
            else frag_zero(b);
            acc = mma(a[ks], b, acc);
        }
        if (t < T_) {
            float* o = out + ((size_t)bf * T_ + t) * Co;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = 4 * g4 + r;
                if (ch < Co) o[ch] = acc[r] + bias[ch];
            }
        }
    }
}

template <class T, int H>
static int encoder_fwd_t(const nbss_cfg& c, const float* P, const void* packed, const void* xin, void* y, hipStream_t st) {
    const int nstrips = c.B * c.F * cdiv(c.T, 16);
    dim3 grid(cdiv(nstrips, 4) < 2048 ? cdiv(nstrips, 4) : 2048), block(256);
    ProfScope ps(PK_ENC_F, st);
    NBSS_LAUNCH((encoder_fwd_kernel<T, H>), grid, block, 0, st, c, P, (const T*)packed + pack_off(c, 0, K_ENC), (const T*)xin, (T*)y);
    return NBSS_CHECK_LAUNCH();
}
template <class T, int H>
static int decoder_fwd_t(const nbss_cfg& c, const float* P, const void* packed, const void* x, float* out, hipStream_t st) {
    const int nstrips = c.B * c.F * cdiv(c.T, 16);
    dim3 grid(cdiv(nstrips, 4) < 2048 ? cdiv(nstrips, 4) : 2048), block(256);
    ProfScope ps(PK_DEC_F, st);
    NBSS_LAUNCH((decoder_fwd_kernel<T, H>), grid, block, 0, st, c, P + param_off_dec_b(c), (const T*)packed + pack_off(c, 0, K_DEC), (const T*)x, out);
    return NBSS_CHECK_LAUNCH();
}

int encoder_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, const void* xin, void* y, hipStream_t st) {
    if (c.H == GeoL::H) return c.dtype == NBSS_BF16 ? encoder_fwd_t<bf16_t, GeoL::H>(c, P, packed, xin, y, st) : encoder_fwd_t<float, GeoL::H>(c, P, packed, xin, y, st);
    return c.dtype == NBSS_BF16 ? encoder_fwd_t<bf16_t, GeoS::H>(c, P, packed, xin, y, st) : encoder_fwd_t<float, GeoS::H>(c, P, packed, xin, y, st);
}
int decoder_fwd_impl(const nbss_cfg& c, const float* P, const void* packed, const void* x, float* out, hipStream_t st) {
    if (c.H == GeoL::H) return c.dtype == NBSS_BF16 ? decoder_fwd_t<bf16_t, GeoL::H>(c, P, packed, x, out, st) : decoder_fwd_t<float, GeoL::H>(c, P, packed, x, out, st);
    return c.dtype == NBSS_BF16 ? decoder_fwd_t<bf16_t, GeoS::H>(c, P, packed, x, out, st) : decoder_fwd_t<float, GeoS::H>(c, P, packed, x, out, st);
}

// ---- backward -------------------------------------------------------------------------------
// decoder: dx = W^T dout (form 2, K = C_out) and a stream-dtype copy of dout for the weight gradient
template <class T>
__global__ __launch_bounds__(256) void decoder_bwd_kernel(nbss_cfg c, const T* __restrict__ WpT, const float* __restrict__ dout,
                                                          T* __restrict__ dx, T* __restrict__ dout_t, int CP) {
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    const int T_ = c.T, Co = c.C_out;
    const int nst = cdiv(T_, 16);
    const int nstrips = c.B * c.F * nst;
    constexpr int MT = ENC_H / 16;
    Frag<T> a[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) wfrag_load(a[i], WpT, i, 1, 0);
    const int wpb = blockDim.x >> 6;
    for (int s = blockIdx.x * wpb + wave_id(); s < nstrips; s += gridDim.x * wpb) {
        const int bf = s / nst, t = (s % nst) * 16 + l15;
        const size_t n = (size_t)bf * T_ + t;
        Frag<T> b;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * g4 + j;
            const float v = (t < T_ && k < Co) ? dout[n * Co + k] : 0.f;
            frag_set(b, j, v);
            if (t < T_ && k < CP) store1(dout_t + n * CP + k, v);
        }
        if (t < T_) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const f32x4 acc = mma(a[i], b, F32X4_ZERO);
                store4(dx + n * ENC_H + 16 * i + 4 * g4, acc[0], acc[1], acc[2], acc[3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i) (void)mma(a[i], b, F32X4_ZERO);
        }
    }
}

int decoder_bwd_impl(const nbss_cfg& c, const float* P, float* G, const void* packed, const void* x, const float* dout, void* dx, void* ws,
                     hipStream_t st) {
    if (c.H != ENC_H) return gb_decoder_bwd(c, P, G, x, dout, dx, ws, st);
    const size_t N = (size_t)c.B * c.F * c.T;
    // the stream-dtype copy of dout for the weight gradient: rows padded to 8 channels in bf16 (zeros; mvalid keeps them out of dW) so that the problem
    // takes wgrad.hip's transposing-read kernel (16-byte row pieces) — with 4-channel rows it fell to the generic kernel: 215 us per step at batch 32
    // for a 4 x 96 weight (round 6)
    const int CP = c.dtype == NBSS_BF16 ? (c.C_out + 7) & ~7 : (c.C_out + 3) & ~3;
    const int nstrips = c.B * c.F * cdiv(c.T, 16);
    dim3 grid(cdiv(nstrips, 4) < 2048 ? cdiv(nstrips, 4) : 2048), block(256);
    prof_begin(PK_DEC_B, st);
    if (c.dtype == NBSS_BF16)
        NBSS_LAUNCH((decoder_bwd_kernel<bf16_t>), grid, block, 0, st, c, (const bf16_t*)packed + pack_off(c, 0, K_DEC_T), dout, (bf16_t*)dx, (bf16_t*)ws, CP);
    else
        NBSS_LAUNCH((decoder_bwd_kernel<float>), grid, block, 0, st, c, (const float*)packed + pack_off(c, 0, K_DEC_T), dout, (float*)dx, (float*)ws, CP);
    prof_end(PK_DEC_B, st);
    int e = NBSS_CHECK_LAUNCH();
    if (e) return e;
    WgradArgs a;
    a.mvalid = c.C_out; a.nvalid = 0;
    a.Ntok = (int)N; a.F = c.F; a.T = c.T; a.shift_stride = 1; a.shift_dim = 0; a.groups = 1; a.taps = 1;
    a.stats = nullptr; a.gamma = nullptr; a.beta = nullptr;
    a.A = ws; a.lda = CP; a.MA = CP; a.B = x; a.ldb = ENC_H; a.NB = ENC_H;
    a.dW = G + param_off_dec_w(c); a.dbias = G + param_off_dec_b(c);
    a.part = (float*)((char*)ws + ws_wgpart_offset(c));  // two-stage flush (1024 x-blocks of same-address atomics otherwise)
    return wgrad_launch(a, c.dtype, st);
}

// encoder: the network input needs no gradient; only dW[o][i][tap] = sum_n dy[n][o] xin[n+tap-2][i] and db
// ws: the backward workspace when the caller has one (nbss_spatialnet_bwd*): partial tiles + reduce instead of the atomicAdd flush
// bf16 rows of C_in channels -> rows of CP (zero padding): 16-byte pieces for the transposing-read weight-gradient kernel
__global__ __launch_bounds__(256) void pad_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, size_t n, int C, int CP) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * (size_t)CP; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / CP;
        const int k = (int)(i - r * CP);
        dst[i] = k < C ? src[r * C + k] : (bf16_t)0;
    }
}

int encoder_bwd_impl(const nbss_cfg& c, float* G, const void* xin, const void* dy, void* ws, hipStream_t st) {
    WgradArgs a;
    if (ws) a.part = (float*)((char*)ws + ws_wgpart_offset(c));
    a.mvalid = 0; a.nvalid = 0;
    if (ws && c.dtype == NBSS_BF16 && c.H == ENC_H && c.C_in % 8 != 0) {
        // the network input has 2 C = 12 channels (24-byte rows): a zero-padded copy with 16 (the operand region of the workspace is idle here) takes the
        // weight gradient to the transposing-read kernel, nvalid keeps the padding out of dW (round 6: 215 -> ~70 us per step at batch 32)
        const size_t N = (size_t)c.B * c.F * c.T;
        const int CP = (c.C_in + 7) & ~7;
        bf16_t* xp = (bf16_t*)((char*)ws + ws_align(N * 2 * sizeof(float)));
        NBSS_LAUNCH(pad_rows_kernel, dim3(2048), dim3(256), 0, st, (const bf16_t*)xin, xp, N, c.C_in, CP);
        int e = NBSS_CHECK_LAUNCH();
        if (e) return e;
        a.Ntok = (int)N; a.F = c.F; a.T = c.T; a.shift_stride = 1; a.shift_dim = 0; a.groups = 1; a.taps = c.enc_ks;
        a.stats = nullptr; a.gamma = nullptr; a.beta = nullptr;
        a.A = dy; a.lda = c.H; a.MA = c.H; a.B = xp; a.ldb = CP; a.NB = CP; a.nvalid = c.C_in;
        a.dW = G + param_off_enc_w(c); a.dbias = G + param_off_enc_b(c);
        return wgrad_launch(a, c.dtype, st);
    }
    a.Ntok = c.B * c.F * c.T; a.F = c.F; a.T = c.T; a.shift_stride = 1; a.shift_dim = 0; a.groups = 1; a.taps = c.enc_ks;
    a.stats = nullptr; a.gamma = nullptr; a.beta = nullptr;
    a.A = dy; a.lda = c.H; a.MA = c.H; a.B = xin; a.ldb = c.C_in; a.NB = c.C_in;
    a.dW = G + param_off_enc_w(c); a.dbias = G + param_off_enc_b(c);
    return wgrad_launch(a, c.dtype, st);
}
