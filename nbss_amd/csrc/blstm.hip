// blstm.hip — the recurrences of the narrow-band BiLSTM (reference models/arch/blstm2_fc1.py:45-68: two bidirectional nn.LSTM layers per frequency bin,
// every (batch item, frequency) an independent sequence), forward and backward through time.
//
// One LSTM direction:  G_t = W_ih x_t + b_ih + b_hh + W_hh h_{t-1}  (gate rows i | f | g | o, torch's order),  c_t = f c_{t-1} + i g,  h_t = o tanh(c_t).
// The input part  Gx = W_ih x + b  of ALL frames is one dense map (nbss_nb_conv_t, taps = 1); the weight gradients and the input gradient are dense
// contractions over all (sequence, frame) pairs (nbss_nb_conv_t_bwd).  What is sequential — T steps, each needing the previous step's h — runs here as ONE
// persistent workgroup per (tile of 16 NTN sequences, direction) for all T steps:
//   * the recurrent product is an MFMA GEMM per step: gate rows = M (wave w owns the unit tiles ut = w, w + 8, ... and for each ALL FOUR gate rows of its
//     16 units, so a lane ends a step holding i, f, g, o of the same (unit, sequence) pairs: the cell update needs no exchange), sequences = N, K = the HD
//     hidden units; W_hh streams from L2 as packed A fragments every step (512 KB per step at HD = 256 in bf16: it does not fit the LDS), h_{t-1} is the B
//     operand straight from an LDS image [sequence][HD] (natural K order, 16-byte reads), double-buffered: one workgroup barrier per step;
//   * c stays in registers for the whole sequence; the accumulators start from the step's Gx values (no separate add);
//   * training keeps i, f, g, o, c per (sequence, frame, unit) for the backward kernel, which walks the steps in reverse: dh = dy + W_hh^T dG_{t+1}
//     (the same MFMA scheme with W_hh^T fragments, K = the 4 HD gate rows, dG of the step as the B operand from LDS), gate gradients in registers,
//     dG out for the dense contractions.
// fp32 stream: the same code on the exact-f32 MFMA (Frag<float>).  No atomics; results do not depend on the grid.
#include "launch.h"
#include "layout.h"

#define BL_THREADS 512
#define BL_WAVES 8

NBSS_DEV float bl_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
NBSS_DEV float bl_tanh(float x) { return 2.0f / (1.0f + __expf(-2.0f * x)) - 1.0f; }

// A-operand fragments of a row-major matrix W [M][K] (tiles of 16 rows, k-steps of 32): out[(tile * KS + ks) * 64 + lane][8] = W[16 tile + l15][32 ks + 8 g4 + j];
// TRANSPOSE: the fragments of W^T (W is [K][M]: out = W[32 ks + 8 g4 + j][16 tile + l15])
template <class T, bool TRANSPOSE>
__global__ void bl_pack_kernel(const float* __restrict__ W, T* __restrict__ out, int M, int K) {
    const int KS = K / 32;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)(M / 16) * KS * 64 * 8) return;
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), ks = (int)((e >> 9) % KS), tile = (int)((e >> 9) / KS);
    const int m = 16 * tile + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + j;
    store1(out + e, TRANSPOSE ? W[(size_t)k * M + m] : W[(size_t)m * K + k]);
}

struct BlArgs {
    const void* gx;     // [n][T][ldg]     Gx of both directions (direction d at columns d * 4 HD), stream dtype
    const void* whh;    // packed W_hh fragments of both directions: [2][(4 HD / 16) * (HD / 32) * 512]
    void* y;            // [n][T][2 HD]    h of both directions side by side (direction d at columns d * HD)
    void* save;         // [2][n][T][5 HD] i | f | g | o | c per direction (training) or nullptr
    long n;
    int T, ldg;
};

// NTN = 16-sequence tiles per workgroup
template <class T, int HD, int NTN>
__global__ __launch_bounds__(BL_THREADS) void blstm_fwd_kernel(BlArgs a) {
    constexpr int UT = HD / 16 / BL_WAVES, KS = HD / 32, HLD = HD + 8, NS = 16 * NTN;
    static_assert(HD % (16 * BL_WAVES) == 0, "unit tiles per wave");
    NBSS_LDS(smem);
    T* hb = reinterpret_cast<T*>(smem);  // [2][NS][HLD]
    const int dir = blockIdx.y, Tn = a.T;
    const long s0 = (long)blockIdx.x * NS;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const T* gx = reinterpret_cast<const T*>(a.gx) + (size_t)dir * 4 * HD;
    const T* wp = reinterpret_cast<const T*>(a.whh) + (size_t)dir * (4 * HD / 16) * KS * 512;
    T* y = reinterpret_cast<T*>(a.y) + (size_t)dir * HD;
    T* sv = a.save ? reinterpret_cast<T*>(a.save) + (size_t)dir * a.n * Tn * 5 * HD : nullptr;
    for (int i = threadIdx.x; i < 2 * NS * HLD; i += BL_THREADS) store1(hb + i, 0.f);
    float c[UT][NTN][4];
#pragma unroll
    for (int u = 0; u < UT; ++u)
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) c[u][nt][r] = 0.f;
    __syncthreads();
    for (int step = 0; step < Tn; ++step) {
        const int t = dir ? Tn - 1 - step : step;
        const T* hprev = hb + (size_t)(step & 1) * NS * HLD;
        T* hnext = hb + (size_t)((step + 1) & 1) * NS * HLD;
        f32x4 acc[UT][4][NTN];
        // the step's Gx values are the accumulators' initial values: lane = (sequence s0 + 16 nt + l15, units 16 ut + 4 g4 + 0..3)
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                const long sq = s0 + 16 * nt + l15;
                const int unit = 16 * (w + BL_WAVES * u) + 4 * g4;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4] = {0.f, 0.f, 0.f, 0.f};
                    if (sq < a.n) load4(gx + ((size_t)sq * Tn + t) * a.ldg + g * HD + unit, v);
                    acc[u][g][nt] = (f32x4){v[0], v[1], v[2], v[3]};
                }
            }
#pragma unroll 2
        for (int ks = 0; ks < KS; ++ks) {
            Frag<T> bq[NTN];
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) frag_load(bq[nt], hprev + (size_t)(16 * nt + l15) * HLD + 32 * ks + 8 * g4);
#pragma unroll
            for (int u = 0; u < UT; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    Frag<T> af;
                    frag_load(af, wp + ((size_t)((g * (HD / 16) + w + BL_WAVES * u) * KS + ks) * 64 + lane) * 8);
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt) acc[u][g][nt] = mma(af, bq[nt], acc[u][g][nt]);
                }
        }
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                const long sq = s0 + 16 * nt + l15;
                const int unit = 16 * (w + BL_WAVES * u) + 4 * g4;
                float iv[4], fv[4], gv[4], ov[4], hv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    iv[r] = bl_sigmoid(acc[u][0][nt][r]);
                    fv[r] = bl_sigmoid(acc[u][1][nt][r]);
                    gv[r] = bl_tanh(acc[u][2][nt][r]);
                    ov[r] = bl_sigmoid(acc[u][3][nt][r]);
                    c[u][nt][r] = fv[r] * c[u][nt][r] + iv[r] * gv[r];
                    hv[r] = ov[r] * bl_tanh(c[u][nt][r]);
                }
                store4(hnext + (size_t)(16 * nt + l15) * HLD + unit, hv[0], hv[1], hv[2], hv[3]);
                if (sq < a.n) {
                    store4(y + ((size_t)sq * Tn + t) * 2 * HD + unit, hv[0], hv[1], hv[2], hv[3]);
                    if (sv) {
                        T* p = sv + ((size_t)sq * Tn + t) * 5 * HD + unit;
                        store4(p, iv[0], iv[1], iv[2], iv[3]);
                        store4(p + HD, fv[0], fv[1], fv[2], fv[3]);
                        store4(p + 2 * HD, gv[0], gv[1], gv[2], gv[3]);
                        store4(p + 3 * HD, ov[0], ov[1], ov[2], ov[3]);
                        store4(p + 4 * HD, c[u][nt][0], c[u][nt][1], c[u][nt][2], c[u][nt][3]);
                    }
                }
            }
        __syncthreads();  // h_t is complete (and every wave is done reading h_{t-1}: its buffer is the next step's target)
    }
}

struct BlBwdArgs {
    const void* dy;     // [n][T][2 HD]    gradient w.r.t. the layer output (direction d at columns d * HD)
    const void* save;   // [2][n][T][5 HD]
    const void* whhT;   // packed W_hh^T fragments of both directions: [2][(HD / 16) * (4 HD / 32) * 512]
    void* dg;           // [n][T][8 HD]    gate pre-activation gradients of both directions (direction d at columns d * 4 HD)
    long n;
    int T;
};

template <class T, int HD, int NTN>
__global__ __launch_bounds__(BL_THREADS) void blstm_bwd_kernel(BlBwdArgs a) {
    constexpr int UT = HD / 16 / BL_WAVES, KS = 4 * HD / 32, GLD = 4 * HD + 8, NS = 16 * NTN;
    NBSS_LDS(smem);
    T* gb = reinterpret_cast<T*>(smem);  // [NS][GLD]  dG of the current step (the B operand of dh_rec = W_hh^T dG)
    const int dir = blockIdx.y, Tn = a.T;
    const long s0 = (long)blockIdx.x * NS;
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id();
    const T* dy = reinterpret_cast<const T*>(a.dy) + (size_t)dir * HD;
    const T* sv = reinterpret_cast<const T*>(a.save) + (size_t)dir * a.n * Tn * 5 * HD;
    const T* wp = reinterpret_cast<const T*>(a.whhT) + (size_t)dir * (HD / 16) * KS * 512;
    T* dg = reinterpret_cast<T*>(a.dg) + (size_t)dir * 4 * HD;
    float dc[UT][NTN][4];
    f32x4 dhr[UT][NTN];  // W_hh^T dG of the step processed before (= the later frame in the direction's order)
#pragma unroll
    for (int u = 0; u < UT; ++u)
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            dhr[u][nt] = F32X4_ZERO;
#pragma unroll
            for (int r = 0; r < 4; ++r) dc[u][nt][r] = 0.f;
        }
    for (int step = Tn - 1; step >= 0; --step) {
        const int t = dir ? Tn - 1 - step : step, tp = dir ? t + 1 : t - 1;  // tp: the frame of c_{t-1} in the direction's order (none at step 0)
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                const long sq = s0 + 16 * nt + l15;
                const int unit = 16 * (w + BL_WAVES * u) + 4 * g4;
                float di[4] = {0.f, 0.f, 0.f, 0.f}, df[4] = {0.f, 0.f, 0.f, 0.f}, dgg[4] = {0.f, 0.f, 0.f, 0.f}, dov[4] = {0.f, 0.f, 0.f, 0.f};
                if (sq < a.n) {
                    float iv[4], fv[4], gv[4], ov[4], cv[4], cp[4] = {0.f, 0.f, 0.f, 0.f}, dyv[4];
                    const T* p = sv + ((size_t)sq * Tn + t) * 5 * HD + unit;
                    load4(p, iv); load4(p + HD, fv); load4(p + 2 * HD, gv); load4(p + 3 * HD, ov); load4(p + 4 * HD, cv);
                    if (step > 0) load4(sv + ((size_t)sq * Tn + tp) * 5 * HD + 4 * HD + unit, cp);
                    load4(dy + ((size_t)sq * Tn + t) * 2 * HD + unit, dyv);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dh = dyv[r] + dhr[u][nt][r];
                        const float th = bl_tanh(cv[r]);
                        const float dct = dh * ov[r] * (1.0f - th * th) + dc[u][nt][r];
                        dov[r] = dh * th * ov[r] * (1.0f - ov[r]);
                        di[r] = dct * gv[r] * iv[r] * (1.0f - iv[r]);
                        dgg[r] = dct * iv[r] * (1.0f - gv[r] * gv[r]);
                        df[r] = dct * cp[r] * fv[r] * (1.0f - fv[r]);
                        dc[u][nt][r] = dct * fv[r];
                    }
                    T* q = dg + ((size_t)sq * Tn + t) * 8 * HD + unit;
                    store4(q, di[0], di[1], di[2], di[3]);
                    store4(q + HD, df[0], df[1], df[2], df[3]);
                    store4(q + 2 * HD, dgg[0], dgg[1], dgg[2], dgg[3]);
                    store4(q + 3 * HD, dov[0], dov[1], dov[2], dov[3]);
                }
                T* gr = gb + (size_t)(16 * nt + l15) * GLD + unit;
                store4(gr, di[0], di[1], di[2], di[3]);
                store4(gr + HD, df[0], df[1], df[2], df[3]);
                store4(gr + 2 * HD, dgg[0], dgg[1], dgg[2], dgg[3]);
                store4(gr + 3 * HD, dov[0], dov[1], dov[2], dov[3]);
            }
        __syncthreads();  // dG of the step is complete
        if (step > 0) {
#pragma unroll
            for (int u = 0; u < UT; ++u)
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) dhr[u][nt] = F32X4_ZERO;
#pragma unroll 2
            for (int ks = 0; ks < KS; ++ks) {
                Frag<T> bq[NTN];
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) frag_load(bq[nt], gb + (size_t)(16 * nt + l15) * GLD + 32 * ks + 8 * g4);
#pragma unroll
                for (int u = 0; u < UT; ++u) {
                    Frag<T> af;
                    frag_load(af, wp + ((size_t)((w + BL_WAVES * u) * KS + ks) * 64 + lane) * 8);
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt) dhr[u][nt] = mma(af, bq[nt], dhr[u][nt]);
                }
            }
        }
        __syncthreads();  // every wave is done reading dG before the next step overwrites it
    }
}

// ws: packed W_hh (forward) resp. W_hh^T (backward) fragments of both directions, stream dtype
size_t blstm_ws_bytes_impl(int HD, int dtype) { return (size_t)2 * (4 * HD / 16) * (HD / 32) * 512 * (dtype == NBSS_BF16 ? 2 : 4); }

template <class T, int HD, int NTN>
static int bl_fwd_go(const BlArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * 16 * NTN * (HD + 8) * sizeof(T);
    if (lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    int e = NBSS_SET_MAX_LDS((blstm_fwd_kernel<T, HD, NTN>), lds);
    if (e) return e;
    NBSS_LAUNCH((blstm_fwd_kernel<T, HD, NTN>), dim3((unsigned)cdiv(a.n, 16 * NTN), 2), dim3(BL_THREADS), lds, st, a);
    return NBSS_CHECK_LAUNCH();
}
template <class T, int HD, int NTN>
static int bl_bwd_go(const BlBwdArgs& a, hipStream_t st) {
    const size_t lds = (size_t)16 * NTN * (4 * HD + 8) * sizeof(T);
    if (lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    int e = NBSS_SET_MAX_LDS((blstm_bwd_kernel<T, HD, NTN>), lds);
    if (e) return e;
    NBSS_LAUNCH((blstm_bwd_kernel<T, HD, NTN>), dim3((unsigned)cdiv(a.n, 16 * NTN), 2), dim3(BL_THREADS), lds, st, a);
    return NBSS_CHECK_LAUNCH();
}
// sequences per workgroup: as few as keep the grid within one round of the chip (the recurrence is latency-bound: T dependent steps), at most 64
static int bl_ntn(long n, int cap) {
    int ntn = 1;
    while (ntn < cap && 2 * cdiv(n, 16 * ntn) > 256) ntn *= 2;
    return ntn;
}

template <class T, int HD>
static int blstm_fwd_t(long n, int Tn, int ldg, const void* gx, const float* whh0, const float* whh1, void* y, void* save, void* ws, hipStream_t st) {
    constexpr int KS = HD / 32;
    const long per = (long)(4 * HD / 16) * KS * 512;
    T* wp = reinterpret_cast<T*>(ws);
    NBSS_LAUNCH((bl_pack_kernel<T, false>), dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, whh0, wp, 4 * HD, HD);
    NBSS_LAUNCH((bl_pack_kernel<T, false>), dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, whh1, wp + per, 4 * HD, HD);
    int e = NBSS_CHECK_LAUNCH();
    if (e) return e;
    BlArgs a = {gx, ws, y, save, n, Tn, ldg};
    const int ntn = bl_ntn(n, sizeof(T) == 2 || HD <= 128 ? 4 : 2);
    return ntn == 1 ? bl_fwd_go<T, HD, 1>(a, st) : ntn == 2 ? bl_fwd_go<T, HD, 2>(a, st) : bl_fwd_go<T, HD, 4>(a, st);
}
template <class T, int HD>
static int blstm_bwd_t(long n, int Tn, const void* dy, const void* save, const float* whh0, const float* whh1, void* dg, void* ws, hipStream_t st) {
    constexpr int KS = 4 * HD / 32;
    const long per = (long)(HD / 16) * KS * 512;
    T* wp = reinterpret_cast<T*>(ws);
    NBSS_LAUNCH((bl_pack_kernel<T, true>), dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, whh0, wp, HD, 4 * HD);
    NBSS_LAUNCH((bl_pack_kernel<T, true>), dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, whh1, wp + per, HD, 4 * HD);
    int e = NBSS_CHECK_LAUNCH();
    if (e) return e;
    BlBwdArgs a = {dy, save, ws, dg, n, Tn};
    // (dG of a step in LDS: 16 NTN x 4 HD values — 64 sequences fit in bf16 at HD = 256, 32 in fp32)
    const int cap = (size_t)64 * (4 * HD + 8) * sizeof(T) <= 160 * 1024 ? 4 : (size_t)32 * (4 * HD + 8) * sizeof(T) <= 160 * 1024 ? 2 : 1;
    const int ntn = bl_ntn(n, cap);
    return ntn == 1 ? bl_bwd_go<T, HD, 1>(a, st) : ntn == 2 ? bl_bwd_go<T, HD, 2>(a, st) : bl_bwd_go<T, HD, 4>(a, st);
}

int blstm_fwd_impl(int dtype, long n, int Tn, int HD, int ldg, const void* gx, const float* whh0, const float* whh1, void* y, void* save, void* ws, hipStream_t st) {
    if (dtype == NBSS_BF16) {
        if (HD == 256) return blstm_fwd_t<bf16_t, 256>(n, Tn, ldg, gx, whh0, whh1, y, save, ws, st);
        if (HD == 128) return blstm_fwd_t<bf16_t, 128>(n, Tn, ldg, gx, whh0, whh1, y, save, ws, st);
    } else {
        if (HD == 256) return blstm_fwd_t<float, 256>(n, Tn, ldg, gx, whh0, whh1, y, save, ws, st);
        if (HD == 128) return blstm_fwd_t<float, 128>(n, Tn, ldg, gx, whh0, whh1, y, save, ws, st);
    }
    return NBSS_EUNSUPPORTED;
}
int blstm_bwd_impl(int dtype, long n, int Tn, int HD, const void* dy, const void* save, const float* whh0, const float* whh1, void* dg, void* ws, hipStream_t st) {
    if (dtype == NBSS_BF16) {
        if (HD == 256) return blstm_bwd_t<bf16_t, 256>(n, Tn, dy, save, whh0, whh1, dg, ws, st);
        if (HD == 128) return blstm_bwd_t<bf16_t, 128>(n, Tn, dy, save, whh0, whh1, dg, ws, st);
    } else {
        if (HD == 256) return blstm_bwd_t<float, 256>(n, Tn, dy, save, whh0, whh1, dg, ws, st);
        if (HD == 128) return blstm_bwd_t<float, 128>(n, Tn, dy, save, whh0, whh1, dg, ws, st);
    }
    return NBSS_EUNSUPPORTED;
}
