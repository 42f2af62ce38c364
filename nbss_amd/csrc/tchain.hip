// tchain.hip — the grouped T-convolution chain of the T-ConvFFN block (models/arch/SpatialNet.py:102-114,61-73) for the geometry-generic
// path (SpatialNet-large: 8 groups of 48 channels, kernel 3), forward recomputation AND backward in one kernel:
//
//   h1 = SiLU(a1) -> a2 = conv1(h1) -> h2 = SiLU(a2) -> a3 = conv2(h2) -> h4 = SiLU(GroupNorm(a3)) -> a5 = conv3(h4) -> h5 = SiLU(a5)
//   g5 = dh5 SiLU'(a5) -> dh4 = conv3^T(g5) -> g3 = GroupNorm'(dh4 SiLU'(.)) -> dh2 = conv2^T(g3) -> g2 = dh2 SiLU'(a2) -> g1 = conv1^T(g2) SiLU'(a1)
//
// Everything between the two dense maps of the block is local to one (sequence, conv group): the convolutions are grouped, GroupNorm's groups
// ARE the conv groups, its statistics run over the sequence.  One workgroup (4 waves) = one (sequence, group): the [T + halo][48] row image
// of the current conv input lives in LDS (overwritten in place by the next one between two barriers), a wave owns 64 frames x 48 channels
// (12 accumulator tiles; weights = MFMA A operand, fragment-ordered in LDS; tokens = N), the pre-activations the backward
// needs again stay in registers (a3) or in LDS (a2) as packed bf16.  69 KB of LDS: two workgroups per CU.
// A tap is a row offset into the image: (tap, channel) is ONE contraction axis of 3 x 48 = 144 (4.5 k-steps instead of 3 x 2).
// The unfused path (gbwd.hip) ran six tap-GEMM launches + GroupNorm forward / backward per layer through ~40 [N][FFN] tensor passes.
#include "tchain.h"
#include "layout.h"
#include <cstdlib>

#define TC_THREADS 256
#define TC_T 256   // frames per workgroup (T <= 256: one sequence)
#define TC_HALO 2  // zero rows above and below (kernel sizes 3 and 5)

template <int CG, int KS>
struct TcGeo {
    static constexpr int NK = KS * CG, NKS = (NK + 31) / 32, OT = CG / 16, NP = NK / 8, PPR = CG / 8;
    static constexpr int RS = CG + 8;                       // image row stride in elements
    static constexpr int IMG = (TC_T + 2 * TC_HALO) * RS;   // elements of one image
    static constexpr int WSET = NKS * OT * 64 * 8;          // elements of one group's fragment-ordered weights
};

// fragment-ordered conv weights [which 0..5][group][k-step][out tile][lane][8]: A[m = 16 ot + (l & 15)][k = 32 ks + 8 (l >> 4) + j], k = tap CG + i
//   forward:        W[g CG + m][i][tap]
//   data gradient:  W[g CG + i][m][KS - 1 - tap]   (the transposed, tap-flipped kernel: the same conv form with zero padding)
struct TcWPrep {
    const float* src[3];
    bf16_t* dst[6];  // [direction][conv]
    int groups, CG, KS, NKS, OT;
    int nconv;  // 3 (the T-conv chain) or 1
};
__global__ void tc_wprep_kernel(TcWPrep p) {
    const int conv = blockIdx.y % p.nconv, dgrad = blockIdx.y / p.nconv, which = 3 * dgrad + conv;
    const long per_g = (long)p.NKS * p.OT * 512, total = per_g * p.groups;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int j = (int)(e & 7), l = (int)((e >> 3) & 63);
        long r = e >> 9;
        const int ot = (int)(r % p.OT);
        r /= p.OT;
        const int ks = (int)(r % p.NKS), g = (int)(r / p.NKS);
        const int m = 16 * ot + (l & 15), k = 32 * ks + 8 * (l >> 4) + j, tap = k / p.CG, i = k % p.CG;
        float v = 0.f;
        if (m < p.CG && tap < p.KS)
            v = dgrad ? p.src[conv][(((long)g * p.CG + i) * p.CG + m) * p.KS + (p.KS - 1 - tap)] : p.src[conv][(((long)g * p.CG + m) * p.CG + i) * p.KS + tap];
        p.dst[which][e] = f2bf(v);
    }
}

NBSS_DEV float bf_lo(uint32_t v) { return bf2f((bf16_t)(v & 0xFFFF)); }
NBSS_DEV float bf_hi(uint32_t v) { return bf2f((bf16_t)(v >> 16)); }

template <int CG, int KS, bool BWD>
__global__ __launch_bounds__(TC_THREADS, 2) void tc_chain_kernel(TChain p) {
    using G = TcGeo<CG, KS>;
    constexpr int NKS = G::NKS, OT = G::OT, RS = G::RS, PPR = G::PPR;
    constexpr int WPC = G::WSET / 8, NW = (WPC + TC_THREADS - 1) / TC_THREADS;  // 16-byte pieces of a weight set, per thread
    NBSS_LDS(smem);
    bf16_t* img = reinterpret_cast<bf16_t*>(smem);           // [T + 2 halo][RS]: the current conv's input, overwritten in place by its successor's
    bf16_t* wl = img + G::IMG;                               // [WSET] weights of the current conv in fragment order
    float* red = reinterpret_cast<float*>(wl + G::WSET);     // [4 reductions][8]
    float* cgs = red + 32;                                   // [2][CG] per-channel GroupNorm affine sums
    uint32_t* park = reinterpret_cast<uint32_t*>(cgs + 2 * CG);  // [8 OT][256 threads] a2 between its forward and its backward use (the compiler spilled it
                                                             // to scratch, whose reloads then waited — one in-order vmcnt — for the stores around them)
    const int tid = threadIdx.x, lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id_u();
    // the groups of a sequence run on one XCD (they read neighbouring 96-byte slices of the same rows)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int seq = (idx / p.groups) * 8 + xcd, g = idx % p.groups;
    if (seq >= p.nseq) return;
    const int T = p.T, FFN = p.FFN;
    const size_t base = (size_t)seq * T * FFN + (size_t)g * CG;  // element (t = 0, channel 0 of the group)
    const float invM = 1.f / (float)(T * CG);

    // weights: requested from L2 into registers before a conv starts, stashed into the LDS buffer after it — behind the barrier that ends the
    // conv's reads — (one copy per workgroup, latency under the conv's MFMAs; as per-wave register fragments for the whole conv: 77 spills)
    u32x4 wq[NW];
    auto w_fetch = [&](const void* wbase) {
        const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(wbase) + (size_t)g * G::WSET);
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int e = tid + k * TC_THREADS;
            wq[k] = src[e < WPC ? e : WPC - 1];
        }
    };
    auto w_stash = [&]() {
        u32x4* dst = reinterpret_cast<u32x4*>(wl);
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int e = tid + k * TC_THREADS;
            if (e < WPC) dst[e] = wq[k];
        }
    };
    // lane's offsets of its B-fragment piece per k-step: piece pk = 4 ks + g4 -> (tap, 8-channel piece); past the contraction: any valid piece (zero weights)
    int boff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        int pk = 4 * ks + g4;
        pk = pk < G::NP ? pk : G::NP - 1;
        boff[ks] = ((pk / PPR) + TC_HALO - KS / 2) * RS + (pk % PPR) * 8;
    }
    f32x4 acc[4][OT];
    auto conv = [&]() {
        const bf16_t* rowp = img + (size_t)(64 * w + l15) * RS;
        const bf16_t* wp = wl + lane * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < OT; ++i) acc[j][i] = F32X4_ZERO;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            Frag<bf16_t> a[OT], b[4];
#pragma unroll
            for (int i = 0; i < OT; ++i) frag_load(a[i], wp + (ks * OT + i) * 512);
#pragma unroll
            for (int j = 0; j < 4; ++j) frag_load(b[j], rowp + (size_t)16 * j * RS + boff[ks]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < OT; ++i) acc[j][i] = mma(a[i], b[j], acc[j][i]);
        }
    };
    // C layout: tile (j, i) of this lane = frame 64 w + 16 j + l15, channels 16 i + 4 g4 + r
    auto frame = [&](int j) { return 64 * w + 16 * j + l15; };
    auto put_img = [&](int j, int i, const float (&v)[4]) {
        u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *reinterpret_cast<u32x2*>(img + (size_t)(frame(j) + TC_HALO) * RS + 16 * i + 4 * g4) = pk;
    };
    // global element offset of tile (j, i) relative to the (uniform) slice base: 32-bit, so that the nine tensors share the offset registers
    auto goff = [&](int j, int i) { return (unsigned)(frame(j) * FFN + 16 * i + 4 * g4); };
    auto put_glb = [&](void* dst, int j, int i, const float (&v)[4]) {
        if (dst && frame(j) < T) store4(reinterpret_cast<bf16_t*>(dst) + base + goff(j, i), v[0], v[1], v[2], v[3]);
    };
    // the image -> a [N][FFN] tensor as 16-byte pieces of full 96-byte row slices (between the barrier that completes the image and the one that
    // ends the conv reading it); from the C layout the same tensor is 12 stores of 8 bytes per lane in 32-byte runs
    auto img_out = [&](void* dst) {
        if (!dst) return;
        bf16_t* d = reinterpret_cast<bf16_t*>(dst) + base;
#pragma unroll 2
        for (int q = 0; q < TC_T * PPR / TC_THREADS; ++q) {
            const int e = tid + q * TC_THREADS, t = e / PPR, pc = e % PPR;
            const u32x4 v = *reinterpret_cast<const u32x4*>(img + (size_t)(t + TC_HALO) * RS + pc * 8);
            if (t < T) *reinterpret_cast<u32x4*>(d + (unsigned)(t * FFN + pc * 8)) = v;
        }
    };
    // the lane's 4 OT channels of a per-channel parameter, requested BEFORE the phase's stores (a load between two stores waits for the first)
    auto chan = [&](const float* prm, float (&o)[OT][4]) {
#pragma unroll
        for (int i = 0; i < OT; ++i) load4(prm + (size_t)g * CG + 16 * i + 4 * g4, o[i]);
    };
    auto tile_in = [&](const void* src, u32x2 (&o)[4][OT]) {  // a [N][FFN] tensor's values of the lane's 12 tiles (clamped rows)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int fr = frame(j) < T ? frame(j) : T - 1;
#pragma unroll
            for (int i = 0; i < OT; ++i) o[j][i] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(src) + base + (unsigned)(fr * FFN + 16 * i + 4 * g4));
        }
    };
    int nred = 0;
    auto wg_sum2 = [&](float& a, float& b) {  // sums over the workgroup; every reduction has its own slots (no second barrier)
        a = wave_sum64(a);
        b = wave_sum64(b);
        float* slot = red + 8 * nred;
        if (lane == 0) {
            slot[w] = a;
            slot[4 + w] = b;
        }
        lds_barrier();
        a = slot[0] + slot[1] + slot[2] + slot[3];
        b = slot[4] + slot[5] + slot[6] + slot[7];
        ++nred;
    };

    w_fetch(p.wf[0]);
    // halo rows (never written again), the affine sums
    for (int e = tid; e < 2 * TC_HALO * RS / 8; e += TC_THREADS) {
        const int row = e / (RS / 8), col = (e % (RS / 8)) * 8;
        const u32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(img + (size_t)(row < TC_HALO ? row : TC_T + row) * RS + col) = z;
    }
    if (tid < 2 * CG) cgs[tid] = 0.f;
    // h1 = SiLU(a1) into the image: 16-byte pieces, coalesced over the 96 contiguous bytes of a row's group slice
    {
        const bf16_t* a1 = reinterpret_cast<const bf16_t*>(p.a1) + base;
        constexpr int NQ = TC_T * PPR / TC_THREADS;
        float v[NQ][8];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = tid + q * TC_THREADS, t = e / PPR, pc = e % PPR;
            load8(a1 + (unsigned)((t < T ? t : T - 1) * FFN + pc * 8), v[q]);  // (clamped, not branched: six requests in flight, not six round trips)
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = tid + q * TC_THREADS, t = e / PPR, pc = e % PPR;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[q][k] = t < T ? silu_fast(v[q][k]) : 0.f;
            store8(img + (size_t)(t + TC_HALO) * RS + pc * 8, v[q]);
            if (p.h1 && t < T) store8(reinterpret_cast<bf16_t*>(p.h1) + base + (unsigned)(t * FFN + pc * 8), v[q]);
        }
    }
    w_stash();
    lds_barrier();

    // ---- forward ----
    uint32_t A2[4][OT][2], A3[4][OT][2];  // pre-activations as stored by the unfused path (bf16), packed (a2 only until it is parked)
    float cbv[OT][4];
    w_fetch(p.wf[1]);
    chan(p.cb[0], cbv);
    conv();
    lds_barrier();
    w_stash();
#pragma unroll
    for (int i = 0; i < OT; ++i) {
        const float (&bs)[4] = cbv[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = frame(j) < T;
            float h[4];
            A2[j][i][0] = pack2bf(acc[j][i][0] + bs[0], acc[j][i][1] + bs[1]);
            A2[j][i][1] = pack2bf(acc[j][i][2] + bs[2], acc[j][i][3] + bs[3]);
            park[(2 * (j * OT + i)) * TC_THREADS + tid] = A2[j][i][0];
            park[(2 * (j * OT + i) + 1) * TC_THREADS + tid] = A2[j][i][1];
            h[0] = keep_if(ok, silu_fast(bf_lo(A2[j][i][0])));
            h[1] = keep_if(ok, silu_fast(bf_hi(A2[j][i][0])));
            h[2] = keep_if(ok, silu_fast(bf_lo(A2[j][i][1])));
            h[3] = keep_if(ok, silu_fast(bf_hi(A2[j][i][1])));
            put_img(j, i, h);
            sched_fence();
        }
    }
    lds_barrier();
    img_out(p.h2);
    float gmv[OT][4], btv[OT][4];
    w_fetch(p.wf[2]);
    chan(p.cb[1], cbv);
    chan(p.gn_w, gmv);
    chan(p.gn_b, btv);
    conv();
    lds_barrier();
    w_stash();
    float mean, rstd;
    {
        float s = 0.f, dummy = 0.f;
#pragma unroll
        for (int i = 0; i < OT; ++i) {
            const float (&bs)[4] = cbv[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                A3[j][i][0] = pack2bf(acc[j][i][0] + bs[0], acc[j][i][1] + bs[1]);
                A3[j][i][1] = pack2bf(acc[j][i][2] + bs[2], acc[j][i][3] + bs[3]);
                if (frame(j) < T) s += (bf_lo(A3[j][i][0]) + bf_hi(A3[j][i][0])) + (bf_lo(A3[j][i][1]) + bf_hi(A3[j][i][1]));
                sched_fence();
            }
        }
        wg_sum2(s, dummy);
        mean = s * invM;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < OT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (frame(j) < T) {
                    const float d0 = bf_lo(A3[j][i][0]) - mean, d1 = bf_hi(A3[j][i][0]) - mean, d2 = bf_lo(A3[j][i][1]) - mean, d3 = bf_hi(A3[j][i][1]) - mean;
                    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
        wg_sum2(q, dummy);
        rstd = rsqrtf(q * invM + 1e-5f);
    }
#pragma unroll
    for (int i = 0; i < OT; ++i) {
        const float (&gm)[4] = gmv[i];
        const float (&bt)[4] = btv[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = frame(j) < T;
            const float a[4] = {bf_lo(A3[j][i][0]), bf_hi(A3[j][i][0]), bf_lo(A3[j][i][1]), bf_hi(A3[j][i][1])};
            float h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = keep_if(ok, silu_fast((a[r] - mean) * rstd * gm[r] + bt[r]));
            put_img(j, i, h);
            sched_fence();
        }
    }
    lds_barrier();
    img_out(p.h4);
    u32x2 tin[4][OT];
    if (BWD) w_fetch(p.wd[2]);
    chan(p.cb[2], cbv);
    if (BWD) tile_in(p.dh5, tin);
    conv();
    if (BWD) {
        lds_barrier();
        w_stash();
    }
#pragma unroll
    for (int i = 0; i < OT; ++i) {
        const float (&bs)[4] = cbv[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = frame(j) < T;
            const uint32_t a0 = pack2bf(acc[j][i][0] + bs[0], acc[j][i][1] + bs[1]), a1 = pack2bf(acc[j][i][2] + bs[2], acc[j][i][3] + bs[3]);
            const float a[4] = {bf_lo(a0), bf_hi(a0), bf_lo(a1), bf_hi(a1)};
            float h[4], dh[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) silu_pair(a[r], h[r], dh[r]);
            put_glb(p.h5, j, i, h);
            if (BWD) {
                const float d[4] = {bf_lo(tin[j][i][0]), bf_hi(tin[j][i][0]), bf_lo(tin[j][i][1]), bf_hi(tin[j][i][1])};
                float gq[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) gq[r] = keep_if(ok, d[r] * dh[r]);
                put_img(j, i, gq);
                sched_fence();
            }
        }
    }
    if (!BWD) return;
    lds_barrier();
    img_out(p.g5);

    // ---- backward ----
    w_fetch(p.wd[1]);
    chan(p.gn_w, gmv);
    chan(p.gn_b, btv);
    conv();  // dh4
    lds_barrier();
    w_stash();
    {
        mean = opaque(mean);  // (xhat is recomputed here: the forward's values must not be kept alive across three conv phases)
        rstd = opaque(rstd);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < OT; ++i) {
            const float (&gm)[4] = gmv[i];
            const float (&bt)[4] = btv[i];
            float dwc[4] = {0.f, 0.f, 0.f, 0.f}, dbc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = frame(j) < T;
                const float a[4] = {bf_lo(A3[j][i][0]), bf_hi(A3[j][i][0]), bf_lo(A3[j][i][1]), bf_hi(A3[j][i][1])};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xh = (a[r] - mean) * rstd;
                    const float d4 = keep_if(ok, round_to(acc[j][i][r], img) * dsilu_fast(xh * gm[r] + bt[r]));
                    dwc[r] += d4 * xh;
                    dbc[r] += d4;
                    s1 += d4 * gm[r];
                    s2 += d4 * gm[r] * xh;
                    acc[j][i][r] = d4 * gm[r];
                }
                sched_fence();
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // affine gradients: sum over the 16 frames of the lane row, then one LDS atomic per (wave, channel)
                const float sw = row_sum16(dwc[r]), sb = row_sum16(dbc[r]);
                if (l15 == 0) {
                    atomicAdd(&cgs[16 * i + 4 * g4 + r], sw);
                    atomicAdd(&cgs[CG + 16 * i + 4 * g4 + r], sb);
                }
            }
            sched_fence();
        }
        wg_sum2(s1, s2);
        mean = opaque(mean);
        rstd = opaque(rstd);
        const float m1 = s1 * invM, m2 = s2 * invM;
#pragma unroll
        for (int i = 0; i < OT; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = frame(j) < T;
                const float a[4] = {bf_lo(A3[j][i][0]), bf_hi(A3[j][i][0]), bf_lo(A3[j][i][1]), bf_hi(A3[j][i][1])};
                float gq[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) gq[r] = keep_if(ok, rstd * (acc[j][i][r] - m1 - (a[r] - mean) * rstd * m2));
                put_img(j, i, gq);
                sched_fence();
            }
        }
    }
    lds_barrier();
    img_out(p.g3);
    if (p.part) {  // (4 128 workgroups on 768 addresses: rows + a reduce instead of same-address atomic chains)
        if (tid < 2 * CG) p.part[(size_t)seq * 2 * FFN + (tid < CG ? 0 : FFN - CG) + (size_t)g * CG + tid] = cgs[tid];
    } else if (tid < CG) atomicAdd(p.dgn_w + (size_t)g * CG + tid, cgs[tid]);
    else if (tid < 2 * CG) atomicAdd(p.dgn_b + (size_t)g * CG + tid - CG, cgs[tid]);
    w_fetch(p.wd[0]);
    conv();  // dh2
    lds_barrier();
    w_stash();
#pragma unroll
    for (int i = 0; i < OT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = frame(j) < T;
            const uint32_t q0 = park[(2 * (j * OT + i)) * TC_THREADS + tid], q1 = park[(2 * (j * OT + i) + 1) * TC_THREADS + tid];
            const float a[4] = {bf_lo(q0), bf_hi(q0), bf_lo(q1), bf_hi(q1)};
            float gq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) gq[r] = keep_if(ok, acc[j][i][r] * dsilu_fast(a[r]));
            put_img(j, i, gq);
            sched_fence();
        }
    lds_barrier();
    img_out(p.g2);
    tile_in(p.a1, tin);
    conv();  // dh1
#pragma unroll
    for (int i = 0; i < OT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (frame(j) < T) {
                const float a[4] = {bf_lo(tin[j][i][0]), bf_hi(tin[j][i][0]), bf_lo(tin[j][i][1]), bf_hi(tin[j][i][1])};
                float gq[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) gq[r] = acc[j][i][r] * dsilu_fast(a[r]);
                put_glb(p.g1, j, i, gq);
            sched_fence();
            }
        }
}

bool tc_chain_takes(int dtype, int CG, int KS, int T) {
    static const bool off = [] {
        const char* e = getenv("NBSS_TCHAIN_OFF");
        return e && e[0] == '1';
    }();
    return !off && dtype == NBSS_BF16 && CG == 48 && KS == 3 && T <= TC_T;
}
size_t tc_wfrag_elems(int groups, int CG, int KS) { return (size_t)groups * ((KS * CG + 31) / 32) * ((CG + 15) / 16) * 512; }

int tc_wprep(const float* const w[3], void* const wf[3], void* const wd[3], int groups, int CG, int KS, hipStream_t st) {
    TcWPrep p;
    for (int k = 0; k < 3; ++k) {
        p.src[k] = w[k];
        p.dst[k] = reinterpret_cast<bf16_t*>(wf[k]);
        p.dst[3 + k] = reinterpret_cast<bf16_t*>(wd[k]);
    }
    p.groups = groups; p.CG = CG; p.KS = KS; p.NKS = (KS * CG + 31) / 32; p.OT = (CG + 15) / 16; p.nconv = 3;
    const long total = (long)tc_wfrag_elems(groups, CG, KS);
    NBSS_LAUNCH(tc_wprep_kernel, dim3((unsigned)((total + 255) / 256), wd[0] ? 6 : 3), dim3(256), 0, st, p);
    return NBSS_CHECK_LAUNCH();
}
int tc_wprep_one(const float* w, void* wf, void* wd, int groups, int CG, int KS, hipStream_t st) {
    TcWPrep p = {};
    p.src[0] = w;
    p.dst[0] = reinterpret_cast<bf16_t*>(wf);
    p.dst[3] = reinterpret_cast<bf16_t*>(wd);
    p.groups = groups; p.CG = CG; p.KS = KS; p.NKS = (KS * CG + 31) / 32; p.OT = (CG + 15) / 16; p.nconv = 1;
    const long total = (long)tc_wfrag_elems(groups, CG, KS);
    NBSS_LAUNCH(tc_wprep_kernel, dim3((unsigned)((total + 255) / 256), wd ? 2 : 1), dim3(256), 0, st, p);
    return NBSS_CHECK_LAUNCH();
}

template <int CG, int KS>
static int tc_launch_t(const TChain& p, bool bwd, hipStream_t st) {
    using G = TcGeo<CG, KS>;
    const size_t lds = ((size_t)G::IMG + G::WSET) * sizeof(bf16_t) + (32 + 2 * CG) * sizeof(float) + (size_t)8 * (CG / 16) * TC_THREADS * sizeof(uint32_t);
    const int grid = 8 * p.groups * cdiv(p.nseq, 8);
    int e;
    if (bwd) {
        if ((e = NBSS_SET_MAX_LDS((tc_chain_kernel<CG, KS, true>), lds))) return e;
        NBSS_LAUNCH((tc_chain_kernel<CG, KS, true>), dim3(grid), dim3(TC_THREADS), lds, st, p);
    } else {
        if ((e = NBSS_SET_MAX_LDS((tc_chain_kernel<CG, KS, false>), lds))) return e;
        NBSS_LAUNCH((tc_chain_kernel<CG, KS, false>), dim3(grid), dim3(TC_THREADS), lds, st, p);
    }
    return NBSS_CHECK_LAUNCH();
}
int tc_chain_launch(const TChain& p, int CG, int KS, bool bwd, hipStream_t st) {
    if (CG == 48 && KS == 3) return tc_launch_t<48, 3>(p, bwd, st);
    return NBSS_EUNSUPPORTED;
}
