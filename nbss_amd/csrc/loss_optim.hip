// loss_optim.hip — uPIT negative SI-SDR loss (forward + gradient) and the fused clip + Adam step.
//
// Loss (models/io/loss.py:21-29,95-118 + torchmetrics si_sdr / pit, restated in oracle/loss_ref.py):
//   alpha = (<p,t> + eps) / (<t,t> + eps);  si_sdr = 10 log10((|alpha t|^2 + eps) / (|alpha t - p|^2 + eps)),
//   eps = finfo(float32).eps, zero_mean = False;  loss_b = min over speaker permutations pi of
//   -mean_s si_sdr(p_{pi(s)}, t_s);  loss = mean_b loss_b;  the gradient flows through the argmin only.
// Three launches: (1) all pairwise dot products as deterministic two-stage block reductions,
// (2) one thread per batch item enumerates the permutations and derives, for the winning one, the two
// scalars of d loss / d p_i = ct_i * t_{sel(i)} + cp_i * p_i, (3) an elementwise kernel applies them.
//
// Optimizer (general_steps.py:243-271: torch.optim.Adam; Trainer gradient_clip_val=5, 'norm'):
// global L2 norm (two-stage reduction), clip coefficient min(1, max_norm / (norm + 1e-6)), Adam with
// bias correction; all 1.19 M parameters live in one flat fp32 buffer, so the whole update is one
// elementwise launch which also re-zeroes the gradient buffer for the next step.
#include "launch.h"
#include "layout.h"
#include "prof.h"

#define LS_MAXS 4
#define LS_CHUNKS 64

NBSS_DEV float block_sum_256(float v, float* red) {  // red: >= 4 floats of LDS
    v = wave_sum64(v);
    __syncthreads();
    if (lane_id() == 0) red[wave_id()] = v;
    __syncthreads();
    float s = 0.f;
    for (unsigned i = 0; i < (blockDim.x >> 6); ++i) s += red[i];
    return s;
}

// part[b][chunk][S*S + 2S]: <p_i,t_j> (i*S+j), <p_i,p_i>, <t_j,t_j>
__global__ __launch_bounds__(256) void sisdr_dots_kernel(int S, int N, const float* __restrict__ p, const float* __restrict__ t,
                                                         float* __restrict__ part) {
    NBSS_LDS(smem);
    float* red = reinterpret_cast<float*>(smem);
    const int b = blockIdx.y, chunk = blockIdx.x, nq = S * S + 2 * S;
    const int per = cdiv(N, (int)gridDim.x), n0 = chunk * per, n1 = n0 + per < N ? n0 + per : N;
    float acc[LS_MAXS * LS_MAXS + 2 * LS_MAXS];
    for (int q = 0; q < nq; ++q) acc[q] = 0.f;
    for (int n = n0 + threadIdx.x; n < n1; n += blockDim.x) {
        float pv[LS_MAXS], tv[LS_MAXS];
        for (int s = 0; s < S; ++s) {
            pv[s] = p[((size_t)b * S + s) * N + n];
            tv[s] = t[((size_t)b * S + s) * N + n];
        }
        for (int i = 0; i < S; ++i) {
            for (int j = 0; j < S; ++j) acc[i * S + j] += pv[i] * tv[j];
            acc[S * S + i] += pv[i] * pv[i];
            acc[S * S + S + i] += tv[i] * tv[i];
        }
    }
    for (int q = 0; q < nq; ++q) {
        const float s = block_sum_256(acc[q], red);
        if (threadIdx.x == 0) part[((size_t)b * gridDim.x + chunk) * nq + q] = s;
    }
}

NBSS_DEV float sisdr_from_dots(float pt, float pp, float tt, float eps) {
    const float alpha = (pt + eps) / (tt + eps);
    const float num = alpha * alpha * tt + eps;
    const float den = alpha * alpha * tt - 2.f * alpha * pt + pp + eps;
    return 10.f * (__logf(num / den) * 0.4342944819032518f);
}

// one thread per batch item: PIT over S! permutations; coef[b][i] = {ct, cp, sel}; loss_b; perm[b][s]
__global__ void pit_finalize_kernel(int B, int S, int nchunks, const float* __restrict__ part, float* __restrict__ loss_b,
                                    int* __restrict__ perm_out, float* __restrict__ coef, float* __restrict__ loss_mean) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const float eps = 1.1920928955078125e-07f;
    const int nq = S * S + 2 * S;
    if (b < B) {
        float d[LS_MAXS * LS_MAXS + 2 * LS_MAXS];
        for (int q = 0; q < nq; ++q) {
            float s = 0.f;
            for (int c = 0; c < nchunks; ++c) s += part[((size_t)b * nchunks + c) * nq + q];
            d[q] = s;
        }
        float sd[LS_MAXS][LS_MAXS];
        for (int i = 0; i < S; ++i)
            for (int j = 0; j < S; ++j) sd[i][j] = sisdr_from_dots(d[i * S + j], d[S * S + i], d[S * S + S + j], eps);
        // enumerate permutations in itertools.permutations order (what torchmetrics uses): perm[s] = prediction paired with target s
        int best[LS_MAXS], cur[LS_MAXS];
        float bestv = 3.0e38f;
        int nperm = 1;
        for (int s = 2; s <= S; ++s) nperm *= s;
        for (int pi = 0; pi < nperm; ++pi) {
            int avail[LS_MAXS], k = pi, fact = nperm;
            for (int s = 0; s < S; ++s) avail[s] = s;
            for (int s = 0; s < S; ++s) {
                fact /= (S - s);
                const int idx = k / fact;
                k %= fact;
                cur[s] = avail[idx];
                for (int e = idx; e < S - 1 - s; ++e) avail[e] = avail[e + 1];
            }
            float v = 0.f;
            for (int s = 0; s < S; ++s) v += sd[cur[s]][s];
            v = -v / S;
            if (v < bestv) {
                bestv = v;
                for (int s = 0; s < S; ++s) best[s] = cur[s];
            }
        }
        loss_b[b] = bestv;
        const float gscale = -(10.f * 0.4342944819032518f) / (float)(B * S);  // d loss / d si_sdr(pair) * d(10 log10)/d ln
        for (int s = 0; s < S; ++s) {
            const int i = best[s];
            perm_out[b * S + s] = i;
            const float pt = d[i * S + s], pp = d[S * S + i], tt = d[S * S + S + s];
            const float alpha = (pt + eps) / (tt + eps);
            const float num = alpha * alpha * tt + eps;
            const float den = alpha * alpha * tt - 2.f * alpha * pt + pp + eps;
            const float da = 1.f / (tt + eps);  // d alpha / d p = t * da
            const float a1 = 2.f * alpha * tt * da / num;
            const float d1 = ((2.f * alpha * tt - 2.f * pt) * da - 2.f * alpha) / den;
            const float d2 = 2.f / den;
            coef[(b * S + i) * 3 + 0] = gscale * (a1 - d1);
            coef[(b * S + i) * 3 + 1] = -gscale * d2;
            coef[(b * S + i) * 3 + 2] = (float)s;
        }
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // B is small: serial, deterministic mean (this kernel is launched with one block)
        float s = 0.f;
        for (int i = 0; i < B; ++i) s += loss_b[i];
        *loss_mean = s / B;
    }
}

__global__ void sisdr_grad_kernel(int BS, int S, int N, const float* __restrict__ p, const float* __restrict__ t, const float* __restrict__ coef,
                                  float* __restrict__ dp) {
    const size_t total = (size_t)BS * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t bi = i / N;
        const int n = (int)(i % N);
        const int b = (int)(bi / S);
        const float ct = coef[bi * 3], cp = coef[bi * 3 + 1];
        const int sel = (int)coef[bi * 3 + 2];
        dp[i] = ct * t[((size_t)b * S + sel) * N + n] + cp * p[i];
    }
}

// ws: part[B][LS_CHUNKS][nq] | loss_b[B] | coef[B*S*3]   (floats)
size_t pit_ws_floats(int B, int S) { return (size_t)B * LS_CHUNKS * (S * S + 2 * S) + B + (size_t)B * S * 3; }

int pit_sisdr_impl(int B, int S, int N, const float* p, const float* t, float* loss, int* perm, float* dp, float* ws, hipStream_t st) {
    if (S < 1 || S > LS_MAXS || B < 1 || B > 1024) return NBSS_EUNSUPPORTED;
    const int nq = S * S + 2 * S;
    float* part = ws;
    float* loss_b = part + (size_t)B * LS_CHUNKS * nq;
    float* coef = loss_b + B;
    ProfScope ps(PK_LOSS, st);
    NBSS_LAUNCH(sisdr_dots_kernel, dim3(LS_CHUNKS, B), dim3(256), 64, st, S, N, p, t, part);
    int e = NBSS_CHECK_LAUNCH();
    if (e) return e;
    NBSS_LAUNCH(pit_finalize_kernel, dim3(1), dim3(1024), 0, st, B, S, LS_CHUNKS, part, loss_b, perm, coef, loss);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    if (dp) {
        const size_t total = (size_t)B * S * N;
        NBSS_LAUNCH(sisdr_grad_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, st, B * S, S, N, p, t, coef,
                    dp);
        e = NBSS_CHECK_LAUNCH();
    }
    return e;
}

// ---------------- clip + Adam ----------------
#define OP_BLOCKS 256

__global__ __launch_bounds__(256) void sumsq_kernel(size_t n, const float* __restrict__ g, float* __restrict__ part) {
    NBSS_LDS(smem);
    float* red = reinterpret_cast<float*>(smem);
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += g[i] * g[i];
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// scal[0] = total grad norm, scal[1] = clip coefficient
__global__ void clip_coef_kernel(int nparts, const float* __restrict__ part, float max_norm, float gscale, float* __restrict__ scal) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < nparts; ++i) s += part[i];
        const float norm = sqrtf(s) * gscale;
        float coef = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.f;
        if (coef > 1.f) coef = 1.f;
        scal[0] = norm;
        scal[1] = coef * gscale;
    }
}

__global__ void adam_kernel(size_t n, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ scal, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt,
                            int flags) {
    const float coef = scal[1];
    const bool zero_grad = flags & 1, decoupled = flags & 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float gi = g[i] * coef;
        float pi = p[i];
        if (decoupled) pi -= lr * wd * pi;       // torch.optim.AdamW: p *= 1 - lr * wd before the Adam update
        else if (wd != 0.f) gi += wd * pi;       // torch.optim.Adam: L2 term added to the gradient
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
        if (zero_grad) g[i] = 0.f;
    }
}

// the same update with the per-step scalars read from device memory: hyper = [lr, 1 - beta1^step, sqrt(1 - beta2^step)] (adam_hyper_impl) — a launch
// without per-step arguments, replayable from a HIP graph (engine.py: TrainStep.graph_step)
__global__ void adam_dev_kernel(size_t n, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                const float* __restrict__ scal, const float* __restrict__ hyper, float beta1, float beta2, float eps, float wd, int flags) {
    const float coef = scal[1], lr = hyper[0], bc1 = hyper[1], bc2_sqrt = hyper[2];
    const bool zero_grad = flags & 1, decoupled = flags & 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float gi = g[i] * coef;
        float pi = p[i];
        if (decoupled) pi -= lr * wd * pi;
        else if (wd != 0.f) gi += wd * pi;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
        if (zero_grad) g[i] = 0.f;
    }
}

// host side of the pair: the three floats exactly as clip_adam_impl computes them (same powf / sqrtf), so that a replayed step is bitwise the eager one
int adam_hyper_impl(int step, float lr, float beta1, float beta2, float* out) {
    if (step < 1 || !out) return NBSS_EINVAL;
    out[0] = lr;
    out[1] = 1.f - powf(beta1, (float)step);
    out[2] = sqrtf(1.f - powf(beta2, (float)step));
    return NBSS_OK;
}

int clip_adam_dev_impl(size_t n, float* p, float* g, float* m, float* v, float* scal, const float* hyper, float max_norm, float grad_scale, float beta1,
                       float beta2, float eps, float wd, int flags, hipStream_t st) {
    float* part = scal + 2;
    ProfScope ps(PK_ADAM, st);
    NBSS_LAUNCH(sumsq_kernel, dim3(OP_BLOCKS), dim3(256), 64, st, n, (const float*)g, part);
    int e = NBSS_CHECK_LAUNCH();
    if (e) return e;
    NBSS_LAUNCH(clip_coef_kernel, dim3(1), dim3(64), 0, st, OP_BLOCKS, (const float*)part, max_norm, grad_scale, scal);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    NBSS_LAUNCH(adam_dev_kernel, dim3(1024), dim3(256), 0, st, n, p, g, m, v, (const float*)scal, hyper, beta1, beta2, eps, wd, flags);
    return NBSS_CHECK_LAUNCH();
}

int clip_adam_impl(size_t n, float* p, float* g, float* m, float* v, float* scal /* >= 2 + OP_BLOCKS floats */, float max_norm, float grad_scale,
                   float lr, float beta1, float beta2, float eps, float wd, int step, int flags, hipStream_t st) {
    if (step < 1) return NBSS_EINVAL;
    float* part = scal + 2;
    ProfScope ps(PK_ADAM, st);
    NBSS_LAUNCH(sumsq_kernel, dim3(OP_BLOCKS), dim3(256), 64, st, n, (const float*)g, part);
    int e = NBSS_CHECK_LAUNCH();
    if (e) return e;
    NBSS_LAUNCH(clip_coef_kernel, dim3(1), dim3(64), 0, st, OP_BLOCKS, (const float*)part, max_norm, grad_scale, scal);
    if ((e = NBSS_CHECK_LAUNCH())) return e;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
    NBSS_LAUNCH(adam_kernel, dim3(1024), dim3(256), 0, st, n, p, g, m, v, (const float*)scal, lr, beta1, beta2, eps, wd, bc1, bc2s, flags);
    return NBSS_CHECK_LAUNCH();
}
