// online.hip — native streaming step of the OnlineSpatialNet (reference: models/arch/OnlineSpatialNet.py:22-60 causal convolutions with
// state, :171-200 the per-frame inference loop, :333-354 forward(inference=True); models/arch/base/retention.py:194-253 recurrent retention).
//
// The reference evaluates a streaming model frame by frame in Python (one module call per frame and layer).  Here one CHUNK of C frames
// of one (batch, frequency) sequence is one workgroup, and a layer's narrow-band half is three launches:
//   online_ret_kernel      LayerNorm -> q (= k: shared), v, g projections of all C frames -> the recurrent multi-scale retention
//                          S_t = carry_t S_{t-1} + k_t^T v_t / sqrt(scale_t), o_t = q_t S_t with the [24 x 48] state column of a
//                          (head, value channel) pair in the registers of its thread -> per-head RMS norm -> SiLU gate -> out_proj -> +res
//   online_tconv_a_kernel  LayerNorm -> 1x1 (96 -> 192) + SiLU -> causal grouped conv + SiLU -> causal grouped conv (pre-GroupNorm a3),
//                          conv states = the last two input frames of each conv; per-(frame, group) partial sums of a3, a3^2 of this
//                          frequency are added to the cross-frequency GroupNorm statistics
//   online_tconv_b_kernel  GroupNorm of each FRAME over (24 channels x all frequencies) -> SiLU -> causal conv + SiLU -> 1x1 (192 -> 96) -> +res
// plus online_encoder_kernel (causal Conv1d k = 5 with a 4-frame state).  The cross-band blocks of a layer are per-frame operations and run
// through the existing nbss_fconv_fwd / nbss_full_fwd kernels on the [B, F, C, H] chunk; the decoder through nbss_decoder_fwd.  All state
// lives in caller-owned device buffers that a step updates in place, so a whole step is a fixed launch sequence: captured once into a HIP
// graph and replayed per chunk (nbss_amd/online.py).  fp32 throughout (a chunk is a few thousand tokens: the step is launch- and
// latency-bound, not arithmetic-bound); geometry: dim_hidden 96, dim_ffn 192, 8 conv groups, 4 retention heads of 24 / 48 (value factor 2).
#include "launch.h"
#include "layout.h"

#define ON_H 96
#define ON_FFN 192
#define ON_CG 24
#define ON_G 8
#define ON_HEADS 4
#define ON_DK 24
#define ON_DV 48
#define ON_CMAX 32  // frames per chunk

NBSS_DEV float on_silu(float x) { return x / (1.0f + __expf(-x)); }

// LayerNorm over H = 96 of the C frames of one sequence: wave w takes frames w, w + nw, ...; u[c][96] in LDS
NBSS_DEV void on_layernorm(const float* __restrict__ xrow, int C, const float* __restrict__ lw, const float* __restrict__ lb, float* u) {
    const int lane = lane_id(), w = wave_id(), nw = (int)(blockDim.x >> 6);
    for (int c = w; c < C; c += nw) {
        const float a = xrow[c * ON_H + lane], b = lane < 32 ? xrow[c * ON_H + 64 + lane] : 0.f;
        const float mean = wave_sum64(a + b) * (1.0f / ON_H);
        const float da = a - mean, db = lane < 32 ? b - mean : 0.f;
        const float rstd = rsqrtf(wave_sum64(da * da + db * db) * (1.0f / ON_H) + 1e-5f);
        u[c * ON_H + lane] = da * rstd * lw[lane] + lb[lane];
        if (lane < 32) u[c * ON_H + 64 + lane] = db * rstd * lw[64 + lane] + lb[64 + lane];
    }
}

// causal Conv1d(C_in -> 96, k = 5) with a 4-frame state: grid = B*F sequences, block = 96 (one output channel per thread)
__global__ __launch_bounds__(ON_H) void online_encoder_kernel(int C, int C_in, const float* __restrict__ w, const float* __restrict__ b,
                                                             const float* __restrict__ x, float* __restrict__ state, float* __restrict__ y) {
    NBSS_LDS(smem);
    float* xin = reinterpret_cast<float*>(smem);  // [4 + C][C_in]: frames -4 .. C-1
    const int bf = blockIdx.x, h = threadIdx.x;
    for (int i = h; i < 4 * C_in; i += ON_H) xin[i] = state[(size_t)bf * 4 * C_in + i];
    for (int i = h; i < C * C_in; i += ON_H) xin[4 * C_in + i] = x[(size_t)bf * C * C_in + i];
    __syncthreads();
    for (int c = 0; c < C; ++c) {
        float acc = b[h];
        for (int i = 0; i < C_in; ++i)
#pragma unroll
            for (int k = 0; k < 5; ++k) acc += w[(h * C_in + i) * 5 + k] * xin[(c + k) * C_in + i];
        y[((size_t)bf * C + c) * ON_H + h] = acc;
    }
    for (int i = h; i < 4 * C_in; i += ON_H) state[(size_t)bf * 4 * C_in + i] = xin[C * C_in + i];  // the last four input frames
}

// grid = B*F, block = 192.  wq_t [96][96], wk_t [96][96] or nullptr (shared with q), wv_t / wg_t [96][192], wo_t [192][96]: TRANSPOSED
// projection weights ([in][out]: adjacent threads read adjacent addresses).  kv [B*F][4][24][48], scale [B*F][4] (a copy per sequence).
__global__ __launch_bounds__(ON_FFN) void online_ret_kernel(int C, const float* __restrict__ lw, const float* __restrict__ lb, const float* __restrict__ wq_t,
                                                           const float* __restrict__ wk_t, const float* __restrict__ wv_t, const float* __restrict__ wg_t,
                                                           const float* __restrict__ wo_t, const float* __restrict__ decay, float kscale,
                                                           float* __restrict__ kv, float* __restrict__ scale, float* __restrict__ x) {
    NBSS_LDS(smem);
    float* u = reinterpret_cast<float*>(smem);  // [C][96]
    float* q = u + C * ON_H;                    // [C][96]
    float* k = q + C * ON_H;                    // [C][96] (aliases q when shared)
    float* v = k + C * ON_H;                    // [C][192]
    float* g = v + C * ON_FFN;                  // [C][192]  the gate, then the gated normalised output
    float* red = g + C * ON_FFN;                // [192]
    const int bf = blockIdx.x, j = threadIdx.x;
    float* xr = x + (size_t)bf * C * ON_H;
    on_layernorm(xr, C, lw, lb, u);
    __syncthreads();
    // projections of all frames: the weight column of this thread is read once per 8 frames
    for (int c0 = 0; c0 < C; c0 += 8) {
        float aq[8], ak[8], av[8], ag[8];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) aq[cc] = ak[cc] = av[cc] = ag[cc] = 0.f;
        for (int i = 0; i < ON_H; ++i) {
            const float wq = j < ON_H ? wq_t[i * ON_H + j] : 0.f, wk = (wk_t && j < ON_H) ? wk_t[i * ON_H + j] : 0.f;
            const float wv = wv_t[i * ON_FFN + j], wg = wg_t[i * ON_FFN + j];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const float uu = u[(c0 + cc < C ? c0 + cc : C - 1) * ON_H + i];
                aq[cc] += wq * uu;
                ak[cc] += wk * uu;
                av[cc] += wv * uu;
                ag[cc] += wg * uu;
            }
        }
#pragma unroll
        for (int cc = 0; cc < 8; ++cc)
            if (c0 + cc < C) {
                if (j < ON_H) {
                    q[(c0 + cc) * ON_H + j] = aq[cc];
                    k[(c0 + cc) * ON_H + j] = wk_t ? ak[cc] * kscale : aq[cc];
                }
                v[(c0 + cc) * ON_FFN + j] = av[cc];
                g[(c0 + cc) * ON_FFN + j] = ag[cc];
            }
    }
    __syncthreads();
    // recurrence: thread = (head, value channel); its [24] state column stays in registers over the chunk
    const int hd = j / ON_DV, dv = j - hd * ON_DV;
    float col[ON_DK];
    float* kvp = kv + (((size_t)bf * ON_HEADS + hd) * ON_DK) * ON_DV + dv;
#pragma unroll
    for (int d = 0; d < ON_DK; ++d) col[d] = kvp[d * ON_DV];
    float sc = scale[(size_t)bf * ON_HEADS + hd];
    const float gam = decay[hd];
    for (int c = 0; c < C; ++c) {
        const float sc2 = sc * gam + 1.0f, carry = sqrtf(sc) * gam / sqrtf(sc2), inv = 1.0f / sqrtf(sc2);
        const float vv = v[c * ON_FFN + j] * inv;
        float o = 0.f;
#pragma unroll
        for (int d = 0; d < ON_DK; ++d) {
            col[d] = col[d] * carry + k[c * ON_H + hd * ON_DK + d] * vv;
            o += q[c * ON_H + hd * ON_DK + d] * col[d];
        }
        sc = sc2;
        red[j] = o * o;
        __syncthreads();
        float ss = 0.f;
        for (int d = 0; d < ON_DV; ++d) ss += red[hd * ON_DV + d];
        __syncthreads();
        g[c * ON_FFN + j] = on_silu(g[c * ON_FFN + j]) * o * rsqrtf(ss * (1.0f / ON_DV) + 1e-6f);
    }
#pragma unroll
    for (int d = 0; d < ON_DK; ++d) kvp[d * ON_DV] = col[d];
    if (dv == 0) scale[(size_t)bf * ON_HEADS + hd] = sc;
    __syncthreads();
    if (j < ON_H) {
        for (int c0 = 0; c0 < C; c0 += 8) {
            float acc[8];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) acc[cc] = 0.f;
            for (int i = 0; i < ON_FFN; ++i) {
                const float wo = wo_t[i * ON_H + j];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) acc[cc] += wo * g[(c0 + cc < C ? c0 + cc : C - 1) * ON_FFN + i];
            }
#pragma unroll
            for (int cc = 0; cc < 8; ++cc)
                if (c0 + cc < C) xr[(c0 + cc) * ON_H + j] += acc[cc];
        }
    }
}

// causal grouped conv (k = 3, 24 channels per group) of output channel j over the padded image in[(C + 2)][192] (rows 0, 1 = the state)
NBSS_DEV float on_conv3(const float* __restrict__ wrow, float bias, const float* in, int c, int g) {
    float acc = bias;
#pragma unroll
    for (int i = 0; i < ON_CG; ++i)
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) acc += wrow[i * 3 + kk] * in[(c + kk) * ON_FFN + g * ON_CG + i];
    return acc;
}

// first half of the T-ConvFFN (up to the GroupNorm input).  w1_t [96][192]; conv weights [192][24][3] (the module's own layout);
// s1, s2 [B*F][2][192]: the last two input frames of conv1 / conv2; a3 [B*F][C][192]; gn_part [B*F][C][8][2]: per-frequency partial sums /
// sums of squares of the GroupNorm input (no atomics: an inference path has to be bitwise repeatable).
__global__ __launch_bounds__(ON_FFN) void online_tconv_a_kernel(int F, int C, const float* __restrict__ lw, const float* __restrict__ lb,
                                                               const float* __restrict__ w1_t, const float* __restrict__ b1,
                                                               const float* __restrict__ c1w, const float* __restrict__ c1b,
                                                               const float* __restrict__ c2w, const float* __restrict__ c2b, float* __restrict__ s1,
                                                               float* __restrict__ s2, const float* __restrict__ x, float* __restrict__ a3,
                                                               float* __restrict__ gn_part) {
    NBSS_LDS(smem);
    float* u = reinterpret_cast<float*>(smem);   // [C][96]
    float* h1 = u + C * ON_H;                    // [C + 2][192]
    float* h2 = h1 + (C + 2) * ON_FFN;           // [C + 2][192]
    float* red = h2 + (C + 2) * ON_FFN;          // [2][192]
    const int bf = blockIdx.x, b = bf / F, j = threadIdx.x, grp = j / ON_CG;
    on_layernorm(x + (size_t)bf * C * ON_H, C, lw, lb, u);
    h1[j] = s1[(size_t)bf * 2 * ON_FFN + j];
    h1[ON_FFN + j] = s1[(size_t)bf * 2 * ON_FFN + ON_FFN + j];
    h2[j] = s2[(size_t)bf * 2 * ON_FFN + j];
    h2[ON_FFN + j] = s2[(size_t)bf * 2 * ON_FFN + ON_FFN + j];
    __syncthreads();
    for (int c0 = 0; c0 < C; c0 += 8) {
        float acc[8];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) acc[cc] = b1[j];
        for (int i = 0; i < ON_H; ++i) {
            const float w = w1_t[i * ON_FFN + j];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) acc[cc] += w * u[(c0 + cc < C ? c0 + cc : C - 1) * ON_H + i];
        }
#pragma unroll
        for (int cc = 0; cc < 8; ++cc)
            if (c0 + cc < C) h1[(2 + c0 + cc) * ON_FFN + j] = on_silu(acc[cc]);
    }
    __syncthreads();
    s1[(size_t)bf * 2 * ON_FFN + j] = h1[C * ON_FFN + j];
    s1[(size_t)bf * 2 * ON_FFN + ON_FFN + j] = h1[(C + 1) * ON_FFN + j];
    for (int c = 0; c < C; ++c) h2[(2 + c) * ON_FFN + j] = on_silu(on_conv3(c1w + (size_t)j * ON_CG * 3, c1b[j], h1, c, grp));
    __syncthreads();
    s2[(size_t)bf * 2 * ON_FFN + j] = h2[C * ON_FFN + j];
    s2[(size_t)bf * 2 * ON_FFN + ON_FFN + j] = h2[(C + 1) * ON_FFN + j];
    for (int c = 0; c < C; ++c) {
        const float a = on_conv3(c2w + (size_t)j * ON_CG * 3, c2b[j], h2, c, grp);
        a3[((size_t)bf * C + c) * ON_FFN + j] = a;
        red[j] = a;
        red[ON_FFN + j] = a * a;
        __syncthreads();
        if (j < 2 * ON_G) {  // threads 0..7: sums, 8..15: sums of squares, one conv group each
            const int gg = j & 7, kind = j >> 3;
            float s = 0.f;
            for (int i = 0; i < ON_CG; ++i) s += red[kind * ON_FFN + gg * ON_CG + i];
            gn_part[(((size_t)bf * C + c) * ON_G + gg) * 2 + kind] = s;  // this frequency's partial: folded in a fixed order by online_gn_fold_kernel
        }
        __syncthreads();
    }
}

// GroupNorm statistics of a frame span all F frequencies of a batch item: the per-frequency partials are summed in frequency order (one thread
// per (batch, frame, group, kind)), so the result does not depend on the order in which the first kernel's workgroups finished
__global__ __launch_bounds__(64) void online_gn_fold_kernel(int F, int n_per_b, const float* __restrict__ gn_part, float* __restrict__ gn_sums) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;  // i over C * 8 * 2
    if (i >= n_per_b) return;
    float s = 0.f;
    for (int f = 0; f < F; ++f) s += gn_part[((size_t)b * F + f) * n_per_b + i];
    gn_sums[(size_t)b * n_per_b + i] = s;
}

// second half: GroupNorm of each frame over (24 channels x F frequencies), SiLU, causal conv3 + SiLU, 1x1 (192 -> 96), residual
__global__ __launch_bounds__(ON_FFN) void online_tconv_b_kernel(int F, int C, const float* __restrict__ gw, const float* __restrict__ gb,
                                                               const float* __restrict__ c3w, const float* __restrict__ c3b,
                                                               const float* __restrict__ w2_t, const float* __restrict__ b2, float* __restrict__ s3,
                                                               const float* __restrict__ a3, const float* __restrict__ gn_sums, float* __restrict__ x) {
    NBSS_LDS(smem);
    float* h4 = reinterpret_cast<float*>(smem);  // [C + 2][192]
    float* h5 = h4 + (C + 2) * ON_FFN;           // [C][192]
    const int bf = blockIdx.x, b = bf / F, j = threadIdx.x, grp = j / ON_CG;
    h4[j] = s3[(size_t)bf * 2 * ON_FFN + j];
    h4[ON_FFN + j] = s3[(size_t)bf * 2 * ON_FFN + ON_FFN + j];
    const float cnt = (float)(ON_CG * F), gwj = gw[j], gbj = gb[j];
    for (int c = 0; c < C; ++c) {
        const float* sm = gn_sums + (((size_t)b * C + c) * ON_G + grp) * 2;
        const float mean = sm[0] / cnt;
        const float rstd = rsqrtf(fmaxf(sm[1] / cnt - mean * mean, 0.f) + 1e-5f);
        h4[(2 + c) * ON_FFN + j] = on_silu((a3[((size_t)bf * C + c) * ON_FFN + j] - mean) * rstd * gwj + gbj);
    }
    __syncthreads();
    s3[(size_t)bf * 2 * ON_FFN + j] = h4[C * ON_FFN + j];
    s3[(size_t)bf * 2 * ON_FFN + ON_FFN + j] = h4[(C + 1) * ON_FFN + j];
    for (int c = 0; c < C; ++c) h5[c * ON_FFN + j] = on_silu(on_conv3(c3w + (size_t)j * ON_CG * 3, c3b[j], h4, c, grp));
    __syncthreads();
    if (j < ON_H) {
        float* xr = x + (size_t)bf * C * ON_H;
        for (int c0 = 0; c0 < C; c0 += 8) {
            float acc[8];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) acc[cc] = b2[j];
            for (int i = 0; i < ON_FFN; ++i) {
                const float w = w2_t[i * ON_H + j];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) acc[cc] += w * h5[(c0 + cc < C ? c0 + cc : C - 1) * ON_FFN + i];
            }
#pragma unroll
            for (int cc = 0; cc < 8; ++cc)
                if (c0 + cc < C) xr[(c0 + cc) * ON_H + j] += acc[cc];
        }
    }
}


// Causal windowed multi-head self-attention over a K/V ring ('mhsa(N)', OnlineSpatialNet.py:356-385 + nn.MultiheadAttention): frame i of the chunk
// attends to the last `scope` frames up to and including itself.  The ring holds the projected keys / values of the last R >= scope - 1 + C frames
// of the sequence at slot (frame index mod R); `pos` (device-side, one int per launch grid: frames seen so far) advances by C per call, so the
// step stays a fixed launch sequence.  grid = B*F, block = 256.  win_t [96][288] = in_proj_weight^T, wo_t [96][96] = out_proj.weight^T.
#define ON_AT 256
__global__ __launch_bounds__(ON_AT) void online_mhsa_kernel(int C, int scope, int R, const float* __restrict__ lw, const float* __restrict__ lb,
                                                           const float* __restrict__ win_t, const float* __restrict__ bin, const float* __restrict__ wo_t,
                                                           const float* __restrict__ bo, float* __restrict__ kring, float* __restrict__ vring,
                                                           const int* __restrict__ pos, float* __restrict__ x) {
    NBSS_LDS(smem);
    float* u = reinterpret_cast<float*>(smem);  // [C][96]   LN(x), later the attention output of the chunk
    float* q = u + C * ON_H;                    // [C][96]
    float* sc = q + C * ON_H;                   // [4][R]    scores / probabilities of one frame
    float* red = sc + ON_HEADS * R;             // [2][4]    per-head max / sum
    const int bf = blockIdx.x, j = threadIdx.x, t0 = pos[0];
    float* xr = x + (size_t)bf * C * ON_H;
    float* kr = kring + (size_t)bf * R * ON_H;
    float* vr = vring + (size_t)bf * R * ON_H;
    on_layernorm(xr, C, lw, lb, u);
    __syncthreads();
    // in_proj: 288 outputs over 256 threads (thread j and, for j < 32, j + 256); q is scaled by 1 / sqrt(dh); k, v go to their ring slots
    for (int o = j; o < 3 * ON_H; o += ON_AT) {
        for (int c0 = 0; c0 < C; c0 += 8) {
            float acc[8];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) acc[cc] = bin[o];
            for (int i = 0; i < ON_H; ++i) {
                const float w = win_t[i * 3 * ON_H + o];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) acc[cc] += w * u[(c0 + cc < C ? c0 + cc : C - 1) * ON_H + i];
            }
#pragma unroll
            for (int cc = 0; cc < 8; ++cc)
                if (c0 + cc < C) {
                    const int c = c0 + cc, slot = (t0 + c) % R;
                    if (o < ON_H) q[c * ON_H + o] = acc[cc] * 0.20412414523193154f;  // 1 / sqrt(24)
                    else if (o < 2 * ON_H) kr[(size_t)slot * ON_H + o - ON_H] = acc[cc];
                    else vr[(size_t)slot * ON_H + o - 2 * ON_H] = acc[cc];
                }
        }
    }
    __syncthreads();  // (the ring rows written above are read below by other threads of this workgroup; these lines were not read earlier in this launch)
    for (int c = 0; c < C; ++c) {
        const int tcur = t0 + c, nk = tcur + 1 < scope ? tcur + 1 : scope;  // keys: frames tcur - nk + 1 .. tcur
        for (int e = j; e < ON_HEADS * nk; e += ON_AT) {
            const int hd = e / nk, kk = e - hd * nk, slot = (tcur - nk + 1 + kk) % R;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < ON_DK; ++d) s += q[c * ON_H + hd * ON_DK + d] * kr[(size_t)slot * ON_H + hd * ON_DK + d];
            sc[hd * R + kk] = s;
        }
        __syncthreads();
        if (j < ON_HEADS) {
            float mx = -1e30f;
            for (int kk = 0; kk < nk; ++kk) mx = fmaxf(mx, sc[j * R + kk]);
            red[j] = mx;
        }
        __syncthreads();
        for (int e = j; e < ON_HEADS * nk; e += ON_AT) {
            const int hd = e / nk, kk = e - hd * nk;
            sc[hd * R + kk] = __expf(sc[hd * R + kk] - red[hd]);
        }
        __syncthreads();
        if (j < ON_HEADS) {
            float sum = 0.f;
            for (int kk = 0; kk < nk; ++kk) sum += sc[j * R + kk];
            red[ON_HEADS + j] = 1.0f / sum;
        }
        __syncthreads();
        if (j < ON_H) {  // output channel j = (head, d): sum over the keys
            const int hd = j / ON_DK;
            float o = 0.f;
            for (int kk = 0; kk < nk; ++kk) o += sc[hd * R + kk] * vr[(size_t)((tcur - nk + 1 + kk) % R) * ON_H + j];
            u[c * ON_H + j] = o * red[ON_HEADS + hd];
        }
        __syncthreads();
    }
    if (j < ON_H) {
        for (int c0 = 0; c0 < C; c0 += 8) {
            float acc[8];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) acc[cc] = bo[j];
            for (int i = 0; i < ON_H; ++i) {
                const float w = wo_t[i * ON_H + j];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) acc[cc] += w * u[(c0 + cc < C ? c0 + cc : C - 1) * ON_H + i];
            }
#pragma unroll
            for (int cc = 0; cc < 8; ++cc)
                if (c0 + cc < C) xr[(c0 + cc) * ON_H + j] += acc[cc];
        }
    }
}
// advances the frame counter of a stream by C (after every layer's attention of the step has read it)
__global__ void online_pos_advance_kernel(int* pos, int C) {
    if (threadIdx.x == 0 && blockIdx.x == 0) pos[0] += C;
}

int memset_async_impl(void* p, size_t bytes, hipStream_t st);

static bool on_ok(int BF, int C) { return BF > 0 && C > 0 && C <= ON_CMAX; }

extern "C" {

int nbss_online_encoder_step(int BF, int C, int C_in, const float* weight, const float* bias, const float* x, float* state, float* y, void* stream) {
    if (!on_ok(BF, C) || C_in <= 0 || C_in > 32 || !weight || !bias || !x || !state || !y) return NBSS_EINVAL;
    NBSS_LAUNCH(online_encoder_kernel, dim3(BF), dim3(ON_H), (size_t)(4 + C) * C_in * sizeof(float), (hipStream_t)stream, C, C_in, weight, bias, x, state, y);
    return NBSS_CHECK_LAUNCH();
}

int nbss_online_ret_step(int BF, int C, const float* ln_w, const float* ln_b, const float* wq_t, const float* wk_t, const float* wv_t, const float* wg_t,
                         const float* wo_t, const float* decay, float* kv, float* scale, float* x, void* stream) {
    if (!on_ok(BF, C) || !ln_w || !ln_b || !wq_t || !wv_t || !wg_t || !wo_t || !decay || !kv || !scale || !x) return NBSS_EINVAL;
    const size_t lds = ((size_t)C * (3 * ON_H + 2 * ON_FFN) + ON_FFN) * sizeof(float);
    int e = NBSS_SET_MAX_LDS(online_ret_kernel, lds);
    if (e) return e;
    NBSS_LAUNCH(online_ret_kernel, dim3(BF), dim3(ON_FFN), lds, (hipStream_t)stream, C, ln_w, ln_b, wq_t, wk_t, wv_t, wg_t, wo_t, decay,
                1.0f / sqrtf((float)ON_DK), kv, scale, x);
    return NBSS_CHECK_LAUNCH();
}

int nbss_online_tconvffn_step(int B, int F, int C, const float* ln_w, const float* ln_b, const float* w1_t, const float* b1, const float* c1w,
                              const float* c1b, const float* c2w, const float* c2b, const float* gn_w, const float* gn_b, const float* c3w, const float* c3b,
                              const float* w2_t, const float* b2, float* s1, float* s2, float* s3, float* a3, float* gn_sums, float* x, void* stream) {
    if (!on_ok(B * F, C) || B <= 0 || !ln_w || !ln_b || !w1_t || !b1 || !c1w || !c1b || !c2w || !c2b || !gn_w || !gn_b || !c3w || !c3b || !w2_t || !b2 ||
        !s1 || !s2 || !s3 || !a3 || !gn_sums || !x)
        return NBSS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int e;
    float* gn_part = gn_sums + (size_t)B * C * ON_G * 2;  // [B*F][C][8][2] behind the sums
    const size_t lds_a = ((size_t)C * ON_H + 2 * (size_t)(C + 2) * ON_FFN + 2 * ON_FFN) * sizeof(float);
    if ((e = NBSS_SET_MAX_LDS(online_tconv_a_kernel, lds_a))) return e;
    NBSS_LAUNCH(online_tconv_a_kernel, dim3(B * F), dim3(ON_FFN), lds_a, st, F, C, ln_w, ln_b, w1_t, b1, c1w, c1b, c2w, c2b, s1, s2, (const float*)x, a3, gn_part);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    NBSS_LAUNCH(online_gn_fold_kernel, dim3((C * ON_G * 2 + 63) / 64, B), dim3(64), 0, st, F, C * ON_G * 2, (const float*)gn_part, gn_sums);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    const size_t lds_b = ((size_t)(C + 2) * ON_FFN + (size_t)C * ON_FFN) * sizeof(float);
    if ((e = NBSS_SET_MAX_LDS(online_tconv_b_kernel, lds_b))) return e;
    NBSS_LAUNCH(online_tconv_b_kernel, dim3(B * F), dim3(ON_FFN), lds_b, st, F, C, gn_w, gn_b, c3w, c3b, w2_t, b2, s3, (const float*)a3, (const float*)gn_sums, x);
    return NBSS_CHECK_LAUNCH();
}

int nbss_online_mhsa_step(int BF, int C, int scope, int ring, const float* ln_w, const float* ln_b, const float* win_t, const float* bin, const float* wo_t,
                          const float* bo, float* kring, float* vring, const int32_t* pos, float* x, void* stream) {
    if (!on_ok(BF, C) || scope <= 0 || ring < scope - 1 + C || !ln_w || !ln_b || !win_t || !bin || !wo_t || !bo || !kring || !vring || !pos || !x) return NBSS_EINVAL;
    const size_t lds = ((size_t)2 * C * ON_H + (size_t)ON_HEADS * ring + 2 * ON_HEADS) * sizeof(float);
    if (lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    int e = NBSS_SET_MAX_LDS(online_mhsa_kernel, lds);
    if (e) return e;
    NBSS_LAUNCH(online_mhsa_kernel, dim3(BF), dim3(ON_AT), lds, (hipStream_t)stream, C, scope, ring, ln_w, ln_b, win_t, bin, wo_t, bo, kring, vring, (const int*)pos, x);
    return NBSS_CHECK_LAUNCH();
}

int nbss_online_advance(int32_t* pos, int C, void* stream) {
    if (!pos || C <= 0) return NBSS_EINVAL;
    NBSS_LAUNCH(online_pos_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int*)pos, C);
    return NBSS_CHECK_LAUNCH();
}

}  // extern "C"
