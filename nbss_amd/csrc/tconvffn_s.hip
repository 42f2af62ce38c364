// tconvffn_s.hip — T-ConvFFN kernels for the bf16 stream (SpatialNet.py:90,102-114,61-73):
//   y = x + W2 * SiLU(gconv3(SiLU(GN(gconv2(SiLU(gconv1(SiLU(W1 * LN(x) + b1))))))))
//
// One workgroup (8 waves) = one (b,f) sequence.  The FFN-wide activations of the WHOLE sequence live in ONE bf16 LDS image
// [T + 4][192] that every stage updates in place; two kinds of phases alternate:
//   * strip phases (wave w owns frames 32w..32w+31, all channels): LayerNorm + W1 at the start, W2 + residual at the end;
//   * group phases (wave g owns the 24 channels of conv group g = GroupNorm group g, ALL frames): the three k=3 grouped convs
//     and GroupNorm.  A group's chain touches only its own 48-byte column slice of the image, so the group phases need NO
//     workgroup barrier: each conv stage walks the sequence in blocks of strips (independent MFMA chains -> ILP), reads its
//     three taps as row-shifted 16-byte LDS reads and writes its output ONE ROW LOWER than its input — the rows a later block
//     still needs are never overwritten.  The group's conv weights stay in registers for the whole kernel, GroupNorm statistics
//     are a wave-local reduction.
// Products are v_mfma_f32_32x32x16_bf16: weights = A (24 of 32 rows), 32 frames = N; a D tile gives a lane 12 valid channels
// (rows (r&3) + 8(r>>2) + 4(lane>>5), r < 12) of frame lane&31, i.e. three 8-byte pieces of an image row.  Biases ride in spare
// K slots of the packed weights (layout.h: ts_conv_k) against a constant-1 slot of the B operand.
// The fp32 stream keeps the group-serial kernels of tconvffn.hip.
#include "side.h"
#include "launch.h"
#include "layout.h"
#include "prof.h"
#include "fold.h"
#include "foldk.h"

#define TS_H 96
#define TS_FFN 192
#define TS_G 8
#define TS_CG 24
#define TS_RS 200   // image row stride in elements (400 B: 16 consecutive frames hit 16 distinct 16-byte bank groups)
#define TS_PAD 4    // rows: 3 for the three downward shifts of the conv stages + 1 zero row above the last frame
#define TS_ONE 0x3F80u  // bf16 1.0 in the low half of a dword
#define TS_WL_FR 56  // fragments in the LDS weight window (W1: 8 groups x 7; W2: 3 x 13 = 39)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef Frag<bf16_t> FragH;

NBSS_DEV f32x16 mma32(const FragH& a, const FragH& b, f32x16 c) {
#ifdef NBSS_EMU
    return hipemu::mfma_32x32x16_bf16(a.v, b.v, c);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
#endif
}
NBSS_DEV f32x16 f32x16_zero() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

NBSS_DEV float bf_lo(uint32_t d) { return __builtin_bit_cast(float, d << 16); }
NBSS_DEV float bf_hi(uint32_t d) { return __builtin_bit_cast(float, d & 0xFFFF0000u); }

// the 12 valid values of a D tile packed pairwise: d[2q], d[2q+1] = channels 8q + 4h + 0..3 (one 8-byte piece of an image row)
struct P6 {
    uint32_t d[6];
};
NBSS_DEV void p6_store(bf16_t* r, const P6& p) {  // r = &img[row][24 g + 4 h]
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        u32x2 v = {p.d[2 * q], p.d[2 * q + 1]};
        *reinterpret_cast<u32x2*>(r + 8 * q) = v;
    }
}
NBSS_DEV void p6_load(const bf16_t* r, P6& p) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(r + 8 * q);
        p.d[2 * q] = v[0];
        p.d[2 * q + 1] = v[1];
    }
}
// Masking is ALWAYS a bitwise AND with an all-ones / zero lane mask: `valid ? f(x) : 0` around a transcendental compiles to
// exec-mask branches, which cut a stage into small basic blocks that the scheduler cannot move loads or MFMAs across.
NBSS_DEV uint32_t lane_mask(bool valid) { return valid ? 0xFFFFFFFFu : 0u; }
NBSS_DEV void silu_pack(const f32x16& a, uint32_t vm, P6& out) {
#pragma unroll
    for (int i = 0; i < 6; ++i) out.d[i] = pack2bf(silu_f(a[2 * i]), silu_f(a[2 * i + 1])) & vm;
}

NBSS_DEV void p6_gstore(bf16_t* __restrict__ g, const P6& p, bool ok, bool nt) {  // g = &op[group][token][4 h]
#ifdef NBSS_K1_NOSTORE  // knock-out probe (flavour build): how much of the kernel is the operand stores?
    return;
#endif
    if (!ok) return;
#ifdef NBSS_TS_NT0  // (A/B flavour: plain stores)
    nt = false;
#endif
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        u32x2 v = {p.d[2 * q], p.d[2 * q + 1]};
#ifndef NBSS_EMU
        if (nt) __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(g + 8 * q));
        else
#endif
            *reinterpret_cast<u32x2*>(g + 8 * q) = v;
    }
}
// whole rows of a group's [tokens][24] bf16 slice: LDS rows (stride RS elements) -> the group-major global tensor, 16 bytes per lane, consecutive lanes =
// consecutive addresses (three pieces per token)
template <int NPMAX, int RS>
NBSS_DEV void rows_gstore_t(bf16_t* __restrict__ gdst, const bf16_t* lsrc, int nrows, int nvalid) {
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < (NPMAX + 63) / 64; ++k) {
        const int i = lane + 64 * k, tok = i / 3, part = i - 3 * tok;
        if (i < 3 * nrows && tok < nvalid) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(lsrc + (size_t)tok * RS + 8 * part);
#ifndef NBSS_EMU
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(gdst + (size_t)tok * TS_CG + 8 * part));
#else
            *reinterpret_cast<u32x4*>(gdst + (size_t)tok * TS_CG + 8 * part) = v;
#endif
        }
    }
}
NBSS_DEV FragH frag_const_one() {  // K slot 0 = 1.0, the rest 0: the B side of a bias slot
    u32x4 v = {TS_ONE, 0u, 0u, 0u};
    FragH f;
    f.v = __builtin_bit_cast(s16x8, v);
    return f;
}

// per-lane vector of a per-channel parameter in D-register order (rows >= 24 of the tile are padding)
NBSS_DEV void chan_vec12(const float* __restrict__ p, int h, float (&v)[12]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) load4(p + 8 * q + 4 * h, &v[4 * q]);
}

template <int N>
NBSS_DEV void load_wfrags(FragH (&w)[N], const bf16_t* __restrict__ base, int blk, int lane) {
#pragma unroll
    for (int ks = 0; ks < N; ++ks) frag_load(w[ks], base + ((size_t)(blk * N + ks) * 64 + lane) * 8);
}

struct TsLane {
    int lane, n, h;
    NBSS_DEV TsLane() {
        lane = lane_id();
        n = lane & 31;
        h = lane >> 5;
    }
};

// B fragments of a k=3 grouped conv for one strip: k-step ks, lane half h = block 2ks + h of (tap, 8 channels); block 9 = bias slot.
// `row0` = &img[in_off + 32 s + n][24 g]  (the frame's own row in the stage's input numbering)
NBSS_DEV void conv_bfrags(const TsLane& L, const bf16_t* row0, FragH (&b)[5]) {
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int b0 = 2 * ks, b1 = ks < 4 ? 2 * ks + 1 : 8;  // (k-step 4: both halves read block 8; h = 1 is overridden below)
        const bf16_t* p = row0 + (L.h ? ((b1 / 3 - 1) * TS_RS + (b1 % 3) * 8) : ((b0 / 3 - 1) * TS_RS + (b0 % 3) * 8));
        frag_load(b[ks], p);
    }
    const FragH one = frag_const_one();
    if (L.h) b[4].v = one.v;
}

NBSS_DEV f32x16 conv_mma(const FragH (&w)[5], const FragH (&b)[5]) {
    f32x16 acc = mma32(w[0], b[0], f32x16_zero());
#pragma unroll
    for (int ks = 1; ks < 5; ++ks) acc = mma32(w[ks], b[ks], acc);
    return acc;
}

struct TsFwdW {  // packed fragment bases of one layer
    const bf16_t *W1, *C1, *C2, *C3, *W2;
};

#ifndef TS_SB
#define TS_SB 2  // strips per block of a group phase (independent MFMA chains in flight)
#endif

// What a training-mode forward keeps for the backward pass (tconvffn_bwd_v_kernel below): the pre-activations a1 (W1 output), a2, a3
// (conv1 / conv2 outputs; a3 = GroupNorm input) as group-major [G][N][24] bf16 tensors — what the reference's autocast graph holds as
// bf16 conv outputs —, the LayerNorm (mean, rstd) of every token and the GroupNorm (mean, rstd) of every (sequence, group).  With them
// the backward pass evaluates each SiLU / SiLU' pair from ONE sigmoid and recomputes one convolution only: a5 = conv3(h4), whose input
// it has in LDS anyway (saving a5 as well cost 2 S.B of stores here and 2 S.B of loads there for 5 MFMAs per strip).
struct TsSave {
    bf16_t *a1, *a2, *a3;
    float *ln, *gn;  // [N][2], [B*F][G][2]
};

// SAVE: training-mode forward (sv is filled); inference launches the SAVE = false instance (no stores, no extra packing)
template <bool SAVE>
__global__ __launch_bounds__(512) void tconvffn_fwd_s_kernel(nbss_cfg c, LayerPtrs lp, TsFwdW W, const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                             TsSave sv, int bf0, int flip) {
    NBSS_LDS(smem);
    const int T_ = c.T, NS = (T_ + 31) >> 5, NT = NS * 32;
    bf16_t* img = reinterpret_cast<bf16_t*>(smem);  // [NT + TS_PAD][TS_RS]
    bf16_t* wl = img + (size_t)(NT + TS_PAD) * TS_RS;  // [56 | 39 fragments][512]: W1 during strip phase 0, then W2
    PHASE_BEGIN(wl + (size_t)TS_WL_FR * 512);
    const TsLane L;
    const int w = wave_id_u(), tid = threadIdx.x;
    // the group's conv weights for the group phases: requested first, resident for the whole kernel
    FragH wc1[5], wc2[5], wc3[5];
    load_wfrags<5>(wc1, W.C1, w, L.lane);
    load_wfrags<5>(wc2, W.C2, w, L.lane);
    load_wfrags<5>(wc3, W.C3, w, L.lane);
    // One workgroup per sequence.  (A persistent grid of 256 workgroups looping over sequences was measured SLOWER, 675 vs 524 us
    // per launch at batch 31: all CUs then march through the HBM-latency and the VALU-bound phases in lock step.)
    const int bf = flip_bid(flip) + bf0;  // (bf0: first sequence of this launch — side.h: SeqTail; flip: launch.h)
    const size_t n0 = (size_t)bf * T_, ntok = (size_t)c.B * c.F * T_;
    {
    const bf16_t* xb = x + (size_t)bf * T_ * TS_H;
    bf16_t* yb = y + (size_t)bf * T_ * TS_H;

    // ---- strip phase 0: h1 = SiLU(W1 LN(x) + b1) for the wave's 32 frames, all 8 groups -> image rows 3 + t -------------------
    // (rows 0..2 are the "frame -1" rows of the three conv stages; row NT + 3 is "frame NT" of the first one)
    // The 56 W1 fragments go global -> LDS ONCE per workgroup: eight waves fetching them separately is 448 KB per sequence through
    // a 64 B/clk L1 (7k cycles); the same holds for the 39 W2 fragments of the last phase.
    {
        constexpr int NV = TS_G * 7 * 64;  // 16-byte vectors
        u32x4 wr[NV / 512];
#pragma unroll
        for (int i = 0; i < NV / 512; ++i) wr[i] = reinterpret_cast<const u32x4*>(W.W1)[tid + i * 512];
#pragma unroll
        for (int i = 0; i < NV / 512; ++i) reinterpret_cast<u32x4*>(wl)[tid + i * 512] = wr[i];
    }
    for (int i = tid; i < 3 * TS_RS / 2; i += blockDim.x) reinterpret_cast<uint32_t*>(img)[i] = 0u;
    for (int i = tid; i < TS_RS / 2; i += blockDim.x) reinterpret_cast<uint32_t*>(img + (size_t)(NT + 3) * TS_RS)[i] = 0u;
    u32x4 raw[6];
    {
        const int t = 32 * w + L.n, tc = t < T_ ? t : T_ - 1;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) raw[ks] = *reinterpret_cast<const u32x4*>(xb + (size_t)tc * TS_H + 16 * ks + 8 * L.h);
    }
    lds_barrier();  // W1 fragments are in LDS
    if (w < NS) {
        const int t = 32 * w + L.n;
        const bool tv = t < T_;
        // LN(x) of the frame as natural-order B fragments: lane half h holds channels 16 ks + 8 h + j
        float v[6][8];
        float sum = 0.f;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[ks][2 * j] = bf_lo(raw[ks][j]);
                v[ks][2 * j + 1] = bf_hi(raw[ks][j]);
                sum += v[ks][2 * j] + v[ks][2 * j + 1];
            }
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / TS_H);
        float sq = 0.f;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[ks][j] -= mean;
                sq += v[ks][j] * v[ks][j];
            }
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq * (1.0f / TS_H) + 1e-5f);
        if (SAVE && tv && L.h == 0) {
            sv.ln[(n0 + t) * 2] = mean;
            sv.ln[(n0 + t) * 2 + 1] = rstd;
        }
        FragH u[7];
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            float gam[8], bet[8], o[8];
            load8(lp.p[P_TF_LN_W] + 16 * ks + 8 * L.h, gam);
            load8(lp.p[P_TF_LN_B] + 16 * ks + 8 * L.h, bet);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = v[ks][j] * rstd * gam[j] + bet[j];
            frag_from(u[ks], o);
        }
        {
            const FragH one = frag_const_one();
            FragH zero;
            frag_zero(zero);
            u[6].v = L.h ? zero.v : one.v;  // bias k-step of K_TS_W1: slot (h = 0, j = 0)
        }
        bf16_t* orow = img + (size_t)(3 + t) * TS_RS + 4 * L.h;
        const uint32_t vm = lane_mask(tv);
        P6 hprev;
#pragma unroll
        for (int i = 0; i < 6; ++i) hprev.d[i] = 0u;
#pragma unroll 2
        for (int g = 0; g < TS_G; ++g) {
            FragH w1[7];
            load_wfrags<7>(w1, wl, g, L.lane);
            f32x16 a1 = mma32(w1[0], u[0], f32x16_zero());
#pragma unroll
            for (int ks = 1; ks < 7; ++ks) a1 = mma32(w1[ks], u[ks], a1);
            P6 h1;
            silu_pack(a1, vm, h1);
#ifdef NBSS_TS_LANE_SAVE  // (A/B flavour: rounds 3-5's lane-wise saves)
            p6_store(orow + g * TS_CG, h1);
            if (SAVE) {
                P6 pa;
#pragma unroll
                for (int i = 0; i < 6; ++i) pa.d[i] = pack2bf(a1[2 * i], a1[2 * i + 1]);
                p6_gstore(sv.a1 + ((size_t)g * ntok + n0 + t) * TS_CG + 4 * L.h, pa, tv, true);
            }
#else
            if (SAVE) {
                // The saved pre-activation leaves as WHOLE ROWS (round 6): staged in the image where h1 of the same group goes — the wave's own 32 rows,
                // nobody else touches them before the barrier —, read back one group later as 16-byte pieces in address order and only then replaced
                // by h1.  (Lane-wise: three 8-byte stores per lane, each instruction covering 16 of every 48 bytes of a 1.5 KB span.  Measured in one call: the
                // kernel 708 -> 689 us, the step +0.4 % at batch 32 and +1.5 % at batch 8; with the three saves knocked out it runs 548 us — the rest is their bytes.)
                P6 pa;
#pragma unroll
                for (int i = 0; i < 6; ++i) pa.d[i] = pack2bf(a1[2 * i], a1[2 * i + 1]);
                p6_store(orow + g * TS_CG, pa);
                if (g > 0) {
                    wave_lds_sync();
                    rows_gstore_t<96, TS_RS>(sv.a1 + ((size_t)(g - 1) * ntok + n0 + 32 * w) * TS_CG, img + (size_t)(3 + 32 * w) * TS_RS + (g - 1) * TS_CG, 32, T_ - 32 * w);
                    wave_lds_sync();
                    p6_store(orow + (g - 1) * TS_CG, hprev);
                }
                hprev = h1;
            } else {
                p6_store(orow + g * TS_CG, h1);
            }
#endif
        }
#ifndef NBSS_TS_LANE_SAVE
        if (SAVE) {
            wave_lds_sync();
            rows_gstore_t<96, TS_RS>(sv.a1 + ((size_t)(TS_G - 1) * ntok + n0 + 32 * w) * TS_CG, img + (size_t)(3 + 32 * w) * TS_RS + (TS_G - 1) * TS_CG, 32, T_ - 32 * w);
            wave_lds_sync();
            p6_store(orow + (TS_G - 1) * TS_CG, hprev);
        }
#endif
    }
    PHASE(0);
    lds_barrier();
    PHASE(1);
    // W2 fragments: requested now, parked in registers during conv1, written over W1 after it (visible after the last barrier)
    constexpr int NV2 = 3 * (TS_FFN / 16 + 1) * 64;
    u32x4 w2r[(NV2 + 511) / 512];
#pragma unroll
    for (int i = 0; i < (NV2 + 511) / 512; ++i) {
        const int v = tid + i * 512;
        w2r[i] = reinterpret_cast<const u32x4*>(W.W2)[v < NV2 ? v : NV2 - 1];
    }

    // ---- group phases: wave g owns channels 24g..24g+23 of every row ------------------------------------------------------------
    const int g = w, cbase = g * TS_CG;
    bf16_t* col = img + cbase;  // &img[0][24 g]
    const size_t gsave0 = ((size_t)g * ntok + n0) * TS_CG, gsave = gsave0 + 4 * L.h;  // token 0 of this group in a saved [G][N][24] tensor (+ this lane's piece)
    bf16_t* scr = wl + (size_t)NV2 * 8 + (size_t)w * (32 * TS_CG);  // [32][24] per wave, behind the 39 W2 fragments
    // conv1: h1 (rows 3 + t) -> h2 = SiLU(.) (rows 2 + t)
#pragma unroll 1
    for (int s0 = 0; s0 < NS; s0 += TS_SB) {
        FragH b[TS_SB][5];
#pragma unroll
        for (int k = 0; k < TS_SB; ++k)
            if (s0 + k < NS) conv_bfrags(L, col + (size_t)(3 + 32 * (s0 + k) + L.n) * TS_RS, b[k]);
#pragma unroll
        for (int k = 0; k < TS_SB; ++k)
            if (s0 + k < NS) {
                const f32x16 a2 = conv_mma(wc1, b[k]);
                P6 h2;
                silu_pack(a2, lane_mask(32 * (s0 + k) + L.n < T_), h2);
                p6_store(col + (size_t)(2 + 32 * (s0 + k) + L.n) * TS_RS + 4 * L.h, h2);
                if (SAVE) {
                    P6 pa;
#pragma unroll
                    for (int i = 0; i < 6; ++i) pa.d[i] = pack2bf(a2[2 * i], a2[2 * i + 1]);
#ifdef NBSS_TS_LANE_SAVE
                    p6_gstore(sv.a2 + gsave + (size_t)(32 * (s0 + k) + L.n) * TS_CG, pa, 32 * (s0 + k) + L.n < T_, true);
#else
                    // whole rows through the wave's scratch strip (the window's 17 KB behind the W2 fragments: W1 is dead since the barrier)
                    p6_store(scr + L.n * TS_CG + 4 * L.h, pa);
                    wave_lds_sync();
                    rows_gstore_t<96, TS_CG>(sv.a2 + gsave0 + (size_t)32 * (s0 + k) * TS_CG, scr, 32, T_ - 32 * (s0 + k));
                    wave_lds_sync();
#endif
                }
            }
    }
    if (L.lane < 6) *reinterpret_cast<u32x2*>(col + (size_t)(2 + NT) * TS_RS + 4 * L.lane) = (u32x2){0u, 0u};  // "frame NT" of the next stage
#pragma unroll
    for (int i = 0; i < (NV2 + 511) / 512; ++i) {
        const int v = tid + i * 512;
        if (v < NV2) reinterpret_cast<u32x4*>(wl)[v] = w2r[i];
    }
    wave_lds_sync();
    PHASE(2);
    // conv2: h2 (rows 2 + t) -> a3 (bf16, rows 1 + t); GroupNorm statistics over the valid frames
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 1
    for (int s0 = 0; s0 < NS; s0 += TS_SB) {
        FragH b[TS_SB][5];
#pragma unroll
        for (int k = 0; k < TS_SB; ++k)
            if (s0 + k < NS) conv_bfrags(L, col + (size_t)(2 + 32 * (s0 + k) + L.n) * TS_RS, b[k]);
#pragma unroll
        for (int k = 0; k < TS_SB; ++k)
            if (s0 + k < NS) {
                const f32x16 a3 = conv_mma(wc2, b[k]);
                const uint32_t vm = lane_mask(32 * (s0 + k) + L.n < T_);
                P6 a3p;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    a3p.d[i] = pack2bf(a3[2 * i], a3[2 * i + 1]) & vm;
                    const float v0 = bf_lo(a3p.d[i]), v1 = bf_hi(a3p.d[i]);  // statistics of the bf16 values GroupNorm sees (0 beyond T)
                    s1 += v0 + v1;
                    s2 += v0 * v0 + v1 * v1;
                }
                p6_store(col + (size_t)(1 + 32 * (s0 + k) + L.n) * TS_RS + 4 * L.h, a3p);
#ifdef NBSS_TS_LANE_SAVE
                if (SAVE) p6_gstore(sv.a3 + gsave + (size_t)(32 * (s0 + k) + L.n) * TS_CG, a3p, 32 * (s0 + k) + L.n < T_, true);
#endif
            }
    }
    if (L.lane < 6) *reinterpret_cast<u32x2*>(col + (size_t)(1 + NT) * TS_RS + 4 * L.lane) = (u32x2){0u, 0u};
#ifndef NBSS_TS_LANE_SAVE
    if (SAVE) {  // a3 sits in the image (rows 1 + t of the group's column) until GroupNorm rewrites it: the whole column slice leaves in address order
        wave_lds_sync();
        rows_gstore_t<3 * 256, TS_RS>(sv.a3 + gsave0, col + (size_t)TS_RS, NT, T_);
        wave_lds_sync();
    }
#endif
    s1 = wave_sum64(s1);
    s2 = wave_sum64(s2);
    const float cnt = (float)(TS_CG * T_);
    const float mean = s1 / cnt;
    const float rstd = rsqrtf(fmaxf(s2 / cnt - mean * mean, 0.f) + 1e-5f);
    if (SAVE && L.lane == 0) {
        sv.gn[((size_t)bf * TS_G + g) * 2] = mean;
        sv.gn[((size_t)bf * TS_G + g) * 2 + 1] = rstd;
    }
    PHASE(3);
    // GroupNorm + SiLU in place (each lane rewrites the values it wrote: no cross-lane hazard)
    {
        float gsc[12], gsh[12];
        chan_vec12(lp.p[P_TF_GN_W] + cbase, L.h, gsc);
        chan_vec12(lp.p[P_TF_GN_B] + cbase, L.h, gsh);
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            gsc[r] *= rstd;
            gsh[r] -= mean * gsc[r];
        }
#pragma unroll 2
        for (int s = 0; s < NS; ++s) {
            bf16_t* r = col + (size_t)(1 + 32 * s + L.n) * TS_RS + 4 * L.h;
            P6 a3p, h4;
            p6_load(r, a3p);
            const uint32_t vm = lane_mask(32 * s + L.n < T_);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float n0 = bf_lo(a3p.d[i]) * gsc[2 * i] + gsh[2 * i], n1 = bf_hi(a3p.d[i]) * gsc[2 * i + 1] + gsh[2 * i + 1];
                h4.d[i] = pack2bf(silu_f(n0), silu_f(n1)) & vm;
            }
            p6_store(r, h4);
        }
    }
    wave_lds_sync();
    PHASE(4);
    // conv3: h4 (rows 1 + t) -> h5 = SiLU(.) (rows t)
#pragma unroll 1
    for (int s0 = 0; s0 < NS; s0 += TS_SB) {
        FragH b[TS_SB][5];
#pragma unroll
        for (int k = 0; k < TS_SB; ++k)
            if (s0 + k < NS) conv_bfrags(L, col + (size_t)(1 + 32 * (s0 + k) + L.n) * TS_RS, b[k]);
#pragma unroll
        for (int k = 0; k < TS_SB; ++k)
            if (s0 + k < NS) {
                const f32x16 a5 = conv_mma(wc3, b[k]);
                P6 h5;
                silu_pack(a5, lane_mask(32 * (s0 + k) + L.n < T_), h5);
                p6_store(col + (size_t)(32 * (s0 + k) + L.n) * TS_RS + 4 * L.h, h5);
            }
    }
    PHASE(5);
    // ---- strip phase 1: y = x + W2 h5 + b2 for the wave's 32 frames.  The branch output goes back into the wave's own image rows
    // (bf16) and leaves as full-row 16-byte pieces (12 per frame, lane-contiguous) together with the residual rows, which are
    // requested before the barrier ---------------------------------------------------------------------------------------------
    const int s = w;
    constexpr int K2 = TS_FFN / 16 + 1;
    u32x4 xres[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int idx = L.lane + 64 * k, tr = 32 * s + idx / 12, trc = tr < T_ ? tr : T_ - 1;
        xres[k] = *reinterpret_cast<const u32x4*>(xb + (size_t)trc * TS_H + 8 * (idx % 12));
    }
    lds_barrier();
    PHASE(6);
    if (s < NS) {
    const int t = 32 * s + L.n;
    f32x16 acc[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) acc[mt] = f32x16_zero();
    bf16_t* hrow = img + (size_t)t * TS_RS;
#pragma unroll
    for (int ks = 0; ks < K2; ++ks) {
        FragH b;
        if (ks < K2 - 1) frag_load(b, hrow + 16 * ks + 8 * L.h);
        else {
            const FragH one = frag_const_one();
            FragH zero;
            frag_zero(zero);
            b.v = L.h ? zero.v : one.v;
        }
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            FragH a;
            frag_load(a, wl + ((size_t)(mt * K2 + ks) * 64 + L.lane) * 8);
            acc[mt] = mma32(a, b, acc[mt]);
        }
    }
    PHASE(7);
    wave_lds_sync();  // every lane has read its h5 row before the rows are overwritten
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x2 v = {pack2bf(acc[mt][4 * q], acc[mt][4 * q + 1]), pack2bf(acc[mt][4 * q + 2], acc[mt][4 * q + 3])};
            *reinterpret_cast<u32x2*>(hrow + 32 * mt + 8 * q + 4 * L.h) = v;
        }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int idx = L.lane + 64 * k, tr = 32 * s + idx / 12, pc = idx % 12;
        const u32x4 br = *reinterpret_cast<const u32x4*>(img + (size_t)tr * TS_RS + 8 * pc);
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack2bf(bf_lo(xres[k][j]) + bf_lo(br[j]), bf_hi(xres[k][j]) + bf_hi(br[j]));
        if (tr < T_) *reinterpret_cast<u32x4*>(yb + (size_t)tr * TS_H + 8 * pc) = o;
    }
    }
    PHASE(8);
    }
    PHASE_END();
}
PHASE_READER(nbss_phase_read_tconvffn_fwd)

size_t tconvffn_s_fwd_lds(int T) {
    const size_t NT = (size_t)((T + 31) / 32) * 32;
    return (NT + TS_PAD) * TS_RS * sizeof(bf16_t) + (size_t)TS_WL_FR * 512 * sizeof(bf16_t) + PHASE_LDS_BYTES;
}

// layout of one layer's saved state (tconvffn_save_bytes(c) bytes): a1 | a2 | a3 ([G][N][24] bf16 each) | LayerNorm stats [N][2] f32 |
// GroupNorm stats [B*F][G][2] f32
size_t tconvffn_save_bytes(const nbss_cfg& c) {
    if (c.dtype != NBSS_BF16 || c.H != TS_H || c.T > 256) return 0;
    const size_t N = (size_t)c.B * c.F * c.T;
    return 3 * ws_align(N * TS_FFN * sizeof(bf16_t)) + ws_align(N * 2 * sizeof(float)) + ws_align((size_t)c.B * c.F * TS_G * 2 * sizeof(float));
}
TsSave ts_save_ptrs(const nbss_cfg& c, void* tsave) {
    const size_t N = (size_t)c.B * c.F * c.T, tb = ws_align(N * TS_FFN * sizeof(bf16_t));
    char* b = (char*)tsave;
    TsSave sv;
    sv.a1 = (bf16_t*)b; sv.a2 = (bf16_t*)(b + tb); sv.a3 = (bf16_t*)(b + 2 * tb);
    sv.ln = (float*)(b + 3 * tb);
    sv.gn = (float*)(b + 3 * tb + ws_align(N * 2 * sizeof(float)));
    return sv;
}

// bf16 stream only; returns NBSS_EUNSUPPORTED when the sequence does not fit the LDS image.  tsave != nullptr: training-mode forward
int tconvffn_fwd_s_impl(const nbss_cfg& c, const float* P, const void* packed, int layer, const void* x, void* y, void* tsave, hipStream_t st, const SeqTail* tl) {
    if (c.dtype != NBSS_BF16) return NBSS_EUNSUPPORTED;
    const size_t lds = tconvffn_s_fwd_lds(c.T);
    if (lds > 160 * 1024 || c.T > 256) return NBSS_EUNSUPPORTED;
    const LayerPtrs lp = layer_ptrs(c, P, layer);
    const bf16_t* pk = (const bf16_t*)packed;
    TsFwdW W = {pk + pack_off(c, layer, K_TS_W1), pk + pack_off(c, layer, K_TS_C1), pk + pack_off(c, layer, K_TS_C2), pk + pack_off(c, layer, K_TS_C3),
                pk + pack_off(c, layer, K_TS_W2)};
    const int nseq = c.B * c.F, ntail = tl ? tl->n : 0;  // (tail launch: see mhsa_fwd_t)
    dim3 grid(nseq - ntail), block(512);
    const int flip = walk_flip_next();  // (one direction for the main and the tail launch)
    ProfScope ps(PK_TCF_F, st);
    int e;
    if (tsave) {
        if ((e = NBSS_SET_MAX_LDS(tconvffn_fwd_s_kernel<true>, lds))) return e;
        NBSS_LAUNCH(tconvffn_fwd_s_kernel<true>, grid, block, lds, st, c, lp, W, (const bf16_t*)x, (bf16_t*)y, ts_save_ptrs(c, tsave), 0, flip);
        if (ntail > 0 && !(e = NBSS_CHECK_LAUNCH()))
            NBSS_LAUNCH(tconvffn_fwd_s_kernel<true>, dim3(ntail), block, lds, tl->ts, c, lp, W, (const bf16_t*)x, (bf16_t*)y, ts_save_ptrs(c, tsave), nseq - ntail, flip);
    } else {
        TsSave none = {nullptr, nullptr, nullptr, nullptr, nullptr};
        if ((e = NBSS_SET_MAX_LDS(tconvffn_fwd_s_kernel<false>, lds))) return e;
        NBSS_LAUNCH(tconvffn_fwd_s_kernel<false>, grid, block, lds, st, c, lp, W, (const bf16_t*)x, (bf16_t*)y, none, 0, flip);
        if (ntail > 0 && !(e = NBSS_CHECK_LAUNCH()))
            NBSS_LAUNCH(tconvffn_fwd_s_kernel<false>, dim3(ntail), block, lds, tl->ts, c, lp, W, (const bf16_t*)x, (bf16_t*)y, none, nseq - ntail, flip);
    }
    if (e) return e;
    return NBSS_CHECK_LAUNCH();
}

// =====================================================================================================================================
// Backward (data gradient) for the bf16 stream.  Same phases as the forward kernel, but the backward chain needs more than the spatial
// image at a time (dSiLU(a2) and the GroupNorm-normalised a3), so the work of one sequence is split over TWO workgroups of 4 conv
// groups each; a workgroup holds three half-width [T][96] bf16 images: S (the spatial chain, updated in place), A = dSiLU(a2), B = a3hat.
//   * 8 waves = 4 groups x 2 halves of the sequence.  The lower half shifts its image rows DOWN by one per forward conv and walks the
//     strips upwards, the upper half shifts UP and walks downwards (mirrored in the backward convs): the two rows either wave reads
//     across the middle are never overwritten inside a stage, so the pair needs no intra-stage synchronisation, only the barrier
//     between stages.
//   * SiLU and its derivative are evaluated together where the pre-activation exists (one sigmoid): a2 -> (h2, dSiLU) with the derivative
//     parked in image A; a5 -> (h5, dSiLU) consumed on the spot; a1 in the strip phase, its derivative parked in the da1 operand buffer
//     (global, L2-hot) until the last stage turns it into da1; likewise dh5 = W2^T dy waits in the da5 operand buffer.
//   * GroupNorm statistics: the two waves of a group exchange partial sums through LDS at the stage barriers (forward statistics are
//     recomputed, nothing needs to be saved by the forward pass).
// Emits the same eight group-major [G][N][24] operand tensors as the group-serial kernel (h1 h2 h4 h5 | da1 da2 da3 da5) for wgrad.hip;
// du = W1^T da1, the LayerNorm backward and dx are tconvffn.hip's tail kernel.
#define TB_RS 104   // half-width image row stride in elements (208 B)
#define TB_PAD 9    // spatial image: 4 rows below frame 0 + 5 rows above the last frame
#define TB_SB 2     // strips per block

struct TsBwdW {
    const bf16_t *W1, *W2T, *C1, *C2, *C3, *C1T, *C2T, *C3T;
};
struct TsOps {
    bf16_t *h1, *h2, *h4, *h5, *da1, *da2, *da3, *da5;
};

NBSS_DEV void silu_dsilu(float a, float& h, float& d) {  // one sigmoid for SiLU and its derivative s (1 + a (1 - s)) = s + h (1 - s)
    const float s = 1.0f / (1.0f + __expf(-a));
    h = a * s;
    d = s + h * (1.0f - s);
}
NBSS_DEV float dsilu_only(float a) {
    const float s = 1.0f / (1.0f + __expf(-a));
    return s * (1.0f + a * (1.0f - s));
}
NBSS_DEV void p6_unpack(const P6& p, float (&v)[12]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        v[2 * i] = bf_lo(p.d[i]);
        v[2 * i + 1] = bf_hi(p.d[i]);
    }
}
NBSS_DEV void p6_pack(const float (&v)[12], uint32_t vm, P6& p) {
#pragma unroll
    for (int i = 0; i < 6; ++i) p.d[i] = pack2bf(v[2 * i], v[2 * i + 1]) & vm;
}
NBSS_DEV void p6_gload(const bf16_t* __restrict__ g, P6& p) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(g + 8 * q);
        p.d[2 * q] = v[0];
        p.d[2 * q + 1] = v[1];
    }
}
// A run of consecutive frames x the 24 channels of one group leaves an LDS image for its group-major operand tensor as 16-byte pieces of
// ONE contiguous run (48 bytes per frame), read back by the wave that has just written the rows (same wave: LDS executes in order).
// The lane-wise form (p6_gstore: three 8-byte stores per lane, each instruction covering 16 of every 48 bytes of a 1.5 KB span) cost a
// quarter of the kernel: knocked out, tconvffn_bwd went from 12.8 to 10.3 ms per step.
template <int NPMAX>
NBSS_DEV void rows_gstore(bf16_t* __restrict__ gdst, const bf16_t* lsrc, int nrows, int nvalid) { rows_gstore_t<NPMAX, TB_RS>(gdst, lsrc, nrows, nvalid); }

// sum over the 32 lanes of each wave half (lanes sharing lane >> 5)
NBSS_DEV float half_sum32(float v) {
    v = row_sum16(v);
    v += __shfl_xor(v, 16);
    return v;
}

// B fragments of a k=3 grouped conv from three row pointers (frames t-1, t, t+1 at the group's column)
NBSS_DEV void conv_bfrags3(const TsLane& L, const bf16_t* r0, const bf16_t* r1, const bf16_t* r2, FragH (&b)[5]) {
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int b0 = 2 * ks, b1 = ks < 4 ? 2 * ks + 1 : 8;
        const bf16_t* p0 = (b0 / 3 == 0 ? r0 : b0 / 3 == 1 ? r1 : r2) + (b0 % 3) * 8;
        const bf16_t* p1 = (b1 / 3 == 0 ? r0 : b1 / 3 == 1 ? r1 : r2) + (b1 % 3) * 8;
        frag_load(b[ks], L.h ? p1 : p0);
    }
    const FragH one = frag_const_one();
    if (L.h) b[4].v = one.v;  // bias slot of the forward convs; the transposed convs carry zeros there
}

__global__ __launch_bounds__(512) void tconvffn_bwd_s_kernel(nbss_cfg c, LayerPtrs lp, TsBwdW W, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                             float* __restrict__ part, TsOps ops, float* __restrict__ stats, int pstride) {
    NBSS_LDS(smem);
    const int T_ = c.T, NS = (T_ + 31) >> 5, NT = NS * 32, NSL = NS >> 1, TS = 32 * NSL;
    bf16_t* S = reinterpret_cast<bf16_t*>(smem);             // [NT + TB_PAD][TB_RS]
    bf16_t* A = S + (size_t)(NT + TB_PAD) * TB_RS;          // [NT][TB_RS]  dSiLU(a2)   (first: the weight window of the strip phase)
    // (short sequences: the 52-fragment weight window of the strip phase is larger than the image it aliases)
    const size_t a_el = (size_t)NT * TB_RS > (size_t)52 * 512 ? (size_t)NT * TB_RS : (size_t)52 * 512;
    bf16_t* Bi = A + a_el;                                  // [NT][TB_RS]  a3hat   (first: dh5 = W2^T dy of the strip phase)
    float* red = reinterpret_cast<float*>(Bi + (size_t)NT * TB_RS);  // [4 groups][2 halves][2]
    float* gnp = red + 16;                                   // [2 halves][2 kinds][96] GroupNorm affine partial sums
    bf16_t* mbox = reinterpret_cast<bf16_t*>(gnp + 4 * 96);  // [8 waves][24]: the frame across the middle, private copy (backward convs)
    bf16_t* wl = A;
    PHASE_BEGIN(mbox + 8 * TS_CG);
    const TsLane L;
    const int w = wave_id_u(), tid = threadIdx.x;
    const int row = blockIdx.x >> 1, gh = blockIdx.x & 1;
    const size_t n0 = (size_t)row * T_, ntok = (size_t)c.B * c.F * T_;
    const bf16_t* xb = x + n0 * TS_H;
    const bf16_t* dyb = dy + n0 * TS_H;

    // ---- strip phase: h1 (-> S, operand), dSiLU(a1) (-> da1 buffer), dh5 = W2^T dy (-> da5 buffer) for the 4 groups of this workgroup
    {
        constexpr int NV = 52 * 64;  // 28 W1 + 24 W2^T fragments
        u32x4 wr[(NV + 511) / 512];
#pragma unroll
        for (int i = 0; i < (NV + 511) / 512; ++i) {
            const int v = tid + i * 512;
            const u32x4* src = v < 28 * 64 ? reinterpret_cast<const u32x4*>(W.W1 + (size_t)gh * 28 * 512) + v
                                           : reinterpret_cast<const u32x4*>(W.W2T + (size_t)gh * 24 * 512) + (v < NV ? v - 28 * 64 : 0);
            wr[i] = *src;
        }
#pragma unroll
        for (int i = 0; i < (NV + 511) / 512; ++i) {
            const int v = tid + i * 512;
            if (v < NV) reinterpret_cast<u32x4*>(wl)[v] = wr[i];
        }
    }
    for (int i = tid; i < 4 * TB_RS / 2; i += blockDim.x) reinterpret_cast<uint32_t*>(S)[i] = 0u;
    for (int i = tid; i < 5 * TB_RS / 2; i += blockDim.x) reinterpret_cast<uint32_t*>(S + (size_t)(NT + 4) * TB_RS)[i] = 0u;
    P6 d1keep[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 6; ++i) d1keep[k].d[i] = 0u;
    u32x4 rawx[6], rawd[6];
    {
        const int t = 32 * w + L.n, tc = t < T_ ? t : T_ - 1;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            rawx[ks] = *reinterpret_cast<const u32x4*>(xb + (size_t)tc * TS_H + 16 * ks + 8 * L.h);
            rawd[ks] = *reinterpret_cast<const u32x4*>(dyb + (size_t)tc * TS_H + 16 * ks + 8 * L.h);
        }
    }
    PHASE(0);
    lds_barrier();
    PHASE(1);
    if (w < NS) {
        const int t = 32 * w + L.n;
        const bool tv = t < T_;
        const uint32_t vm = lane_mask(tv);
        float v[6][8];
        float sum = 0.f;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[ks][2 * j] = bf_lo(rawx[ks][j]);
                v[ks][2 * j + 1] = bf_hi(rawx[ks][j]);
                sum += v[ks][2 * j] + v[ks][2 * j + 1];
            }
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / TS_H);
        float sq = 0.f;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[ks][j] -= mean;
                sq += v[ks][j] * v[ks][j];
            }
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq * (1.0f / TS_H) + 1e-5f);
        if (stats && gh == 0 && tv && L.h == 0) {  // LayerNorm row statistics for the fused tail / weight-gradient kernel (tailw.hip)
            stats[(n0 + t) * 2] = mean;
            stats[(n0 + t) * 2 + 1] = rstd;
        }
        FragH u[7], dq[6];
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            float gam[8], bet[8], o[8];
            load8(lp.p[P_TF_LN_W] + 16 * ks + 8 * L.h, gam);
            load8(lp.p[P_TF_LN_B] + 16 * ks + 8 * L.h, bet);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = v[ks][j] * rstd * gam[j] + bet[j];
            frag_from(u[ks], o);
            dq[ks].v = __builtin_bit_cast(s16x8, rawd[ks]);
        }
        {
            const FragH one = frag_const_one();
            FragH zero;
            frag_zero(zero);
            u[6].v = L.h ? zero.v : one.v;
        }
        bf16_t* srow = S + (size_t)(4 + t) * TB_RS + 4 * L.h;
        bf16_t* brow = Bi + (size_t)t * TB_RS + 4 * L.h;
#pragma unroll
        for (int gl = 0; gl < 4; ++gl) {
            FragH w1[7], w2t[6];
            load_wfrags<7>(w1, wl, gl, L.lane);
            load_wfrags<6>(w2t, wl + 28 * 512, gl, L.lane);
            f32x16 a1 = mma32(w1[0], u[0], f32x16_zero());
#pragma unroll
            for (int ks = 1; ks < 7; ++ks) a1 = mma32(w1[ks], u[ks], a1);
            f32x16 d5 = mma32(w2t[0], dq[0], f32x16_zero());
#pragma unroll
            for (int ks = 1; ks < 6; ++ks) d5 = mma32(w2t[ks], dq[ks], d5);
            float hv[12], dv[12], d5v[12];
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                silu_dsilu(a1[r], hv[r], dv[r]);
                d5v[r] = d5[r];
            }
            P6 ph, pd, p5;
            p6_pack(hv, vm, ph);
            p6_pack(dv, vm, pd);
            p6_pack(d5v, vm, p5);
            p6_store(srow + gl * TS_CG, ph);
            p6_store(brow + gl * TS_CG, p5);  // dh5: picked up by the group waves right after the barrier (image B is free until F2b)
            d1keep[gl] = pd;                  // dSiLU(a1) stays in this wave's registers until the last strip phase
        }
        wave_lds_sync();
#pragma unroll
        for (int gl = 0; gl < 4; ++gl)
            rows_gstore<96>(ops.h1 + ((size_t)(4 * gh + gl) * ntok + n0 + 32 * w) * TS_CG, S + (size_t)(4 + 32 * w) * TB_RS + gl * TS_CG, 32, T_ - 32 * w);
    }
    PHASE(2);
    lds_barrier();  // S and the dh5 image are complete; the weight window is dead

    // ---- group phases: wave = (group gl, half th) -------------------------------------------------------------------------------
    const int gl = w >> 1, th = w & 1, g = 4 * gh + gl;
    const int s_beg = th ? NSL : 0, s_end = th ? NS : NSL, nblk = (s_end - s_beg + TB_SB - 1) / TB_SB;
    bf16_t* Sc = S + gl * TS_CG;
    bf16_t* Ac = A + gl * TS_CG + 4 * L.h;
    bf16_t* Bc = Bi + gl * TS_CG + 4 * L.h;
    const size_t gbase = ((size_t)g * ntok + n0) * TS_CG + 4 * L.h;
    // the wave's own run of frames [32 s_beg, 32 s_end) of an operand tensor / of the spatial image at stage bases (bl, bu)
    const int run0 = 32 * s_beg, nrun = 32 * (s_end - s_beg);
    auto run_gstore = [&](bf16_t* __restrict__ op, int bl, int bu) {
        wave_lds_sync();
        rows_gstore<128 * 3>(op + ((size_t)g * ntok + n0 + run0) * TS_CG, Sc + (size_t)((th ? bu : bl) + run0) * TB_RS, nrun, T_ - run0);
    };
    auto rowp = [&](int tt, int bl, int bu) -> const bf16_t* { return Sc + (size_t)((tt < TS ? bl : bu) + tt) * TB_RS; };
    auto orow = [&](int tt, int bl, int bu) -> bf16_t* { return Sc + (size_t)((th ? bu : bl) + tt) * TB_RS + 4 * L.h; };
    // Backward convs: the halves shift TOWARDS each other, so the one row a wave reads across the middle (frame TS for the lower half,
    // TS - 1 for the upper) is overwritten by the other wave's second output row of the same stage.  Each wave therefore copies that
    // row into its private mailbox right after the stage barrier; a second barrier separates the copies from the stage's writes.
    bf16_t* mb = mbox + w * TS_CG;
    auto fetch_cross = [&](int bl, int bu) {
        const bf16_t* src = Sc + (size_t)(th ? bl + TS - 1 : bu + TS) * TB_RS;
        if (L.lane < 6) *reinterpret_cast<u32x2*>(mb + 4 * L.lane) = *reinterpret_cast<const u32x2*>(src + 4 * L.lane);
        lds_barrier();
    };
    auto rowx = [&](int tt, int bl, int bu) -> const bf16_t* {
        const bool lower = tt < TS;
        const bf16_t* r = Sc + (size_t)((lower ? bl : bu) + tt) * TB_RS;
        return lower != (th == 0) ? mb : r;
    };
    float gw[12], gb[12];
    chan_vec12(lp.p[P_TF_GN_W] + g * TS_CG, L.h, gw);
    chan_vec12(lp.p[P_TF_GN_B] + g * TS_CG, L.h, gb);
    const float cnt = (float)(TS_CG * T_);

    FragH wa[5], wb[5], wc[5];
    load_wfrags<5>(wa, W.C1, g, L.lane);
    load_wfrags<5>(wb, W.C2, g, L.lane);
    load_wfrags<5>(wc, W.C3, g, L.lane);
    // dh5 of the wave's strips: out of image B into registers now (the image becomes a3hat in F2b), consumed in F3.  (Parking such rows in
    // the operand buffers instead costs a vmcnt(0) drain of the strip phase's stores: 16 % of the wave time when measured.)
    constexpr int TB_MAXS = 4;  // strips per wave (T <= 256)
    // (named variables + per-dword selects in F3: `cond ? park[i] : park[j]` on the struct becomes a select of ADDRESSES, which put the
    //  array into scratch memory — and scratch loads queue behind the operand stores like any other vector-memory access)
    P6 park0, park1, park2, park3;
    static_assert(TB_MAXS == 4, "park0..3");
#define TB_PARK_LOAD(k, dst)                                                   \
    if (s_beg + (k) < s_end) p6_load(Bc + (size_t)(32 * (s_beg + (k)) + L.n) * TB_RS, dst); \
    else                                                                       \
        for (int i_ = 0; i_ < 6; ++i_) dst.d[i_] = 0u;
    TB_PARK_LOAD(0, park0)
    TB_PARK_LOAD(1, park1)
    TB_PARK_LOAD(2, park2)
    TB_PARK_LOAD(3, park3)
#undef TB_PARK_LOAD

#define TB_BLOCKS(fwd_dir)                                                   \
    for (int bi_ = 0; bi_ < nblk; ++bi_)                                     \
        for (int s0 = s_beg + TB_SB * (((fwd_dir) == (th == 0)) ? bi_ : nblk - 1 - bi_), once_ = 1; once_; once_ = 0)

    PHASE(3);
    // F1: conv1: h1 (bases 4,4) -> h2 = SiLU(a2) (bases 3,5), dSiLU(a2) -> A
#pragma unroll 1
    TB_BLOCKS(true) {
        FragH b[TB_SB][5];
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                conv_bfrags3(L, rowp(t - 1, 4, 4), rowp(t, 4, 4), rowp(t + 1, 4, 4), b[k]);
            }
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                const bool tv = t < T_;
                const uint32_t vm = lane_mask(tv);
                const f32x16 a2 = conv_mma(wa, b[k]);
                float hv[12], dv[12];
#pragma unroll
                for (int r = 0; r < 12; ++r) silu_dsilu(a2[r], hv[r], dv[r]);
                P6 ph, pd;
                p6_pack(hv, vm, ph);
                p6_pack(dv, vm, pd);
                p6_store(orow(t, 3, 5), ph);
                p6_store(Ac + (size_t)t * TB_RS, pd);
            }
    }
    run_gstore(ops.h2, 3, 5);
    load_wfrags<5>(wa, W.C1T, g, L.lane);  // conv1 is done: its transposed weights arrive long before B1 needs them
    PHASE(4);
    lds_barrier();
    PHASE(5);
    // F2a: conv2: h2 (3,5) -> a3 (bf16, bases 2,6) + partial GroupNorm sums
    {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 1
        TB_BLOCKS(true) {
            FragH b[TB_SB][5];
#pragma unroll
            for (int k = 0; k < TB_SB; ++k)
                if (s0 + k < s_end) {
                    const int t = 32 * (s0 + k) + L.n;
                    conv_bfrags3(L, rowp(t - 1, 3, 5), rowp(t, 3, 5), rowp(t + 1, 3, 5), b[k]);
                }
#pragma unroll
            for (int k = 0; k < TB_SB; ++k)
                if (s0 + k < s_end) {
                    const int t = 32 * (s0 + k) + L.n;
                    const uint32_t vm = lane_mask(t < T_);
                    const f32x16 a3 = conv_mma(wb, b[k]);
                    P6 pa;
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        pa.d[i] = pack2bf(a3[2 * i], a3[2 * i + 1]) & vm;
                        const float v0 = bf_lo(pa.d[i]), v1 = bf_hi(pa.d[i]);
                        s1 += v0 + v1;
                        s2 += v0 * v0 + v1 * v1;
                    }
                    p6_store(orow(t, 2, 6), pa);
                }
        }
        s1 = wave_sum64(s1);
        s2 = wave_sum64(s2);
        if (L.lane == 0) {
            red[(gl * 2 + th) * 2] = s1;
            red[(gl * 2 + th) * 2 + 1] = s2;
        }
    }
    load_wfrags<5>(wb, W.C2T, g, L.lane);
    PHASE(6);
    lds_barrier();
    PHASE(7);
    const float gmean = (red[gl * 4] + red[gl * 4 + 2]) / cnt;
    const float grstd = rsqrtf(fmaxf((red[gl * 4 + 1] + red[gl * 4 + 3]) / cnt - gmean * gmean, 0.f) + 1e-5f);
    // F2b (in place, own values): a3hat -> B, h4 = SiLU(a3hat * gw + gb) -> S (2,6)
#pragma unroll 2
    for (int s = s_beg; s < s_end; ++s) {
        const int t = 32 * s + L.n;
        const bool tv = t < T_;
        const uint32_t vm = lane_mask(tv);
        bf16_t* r = orow(t, 2, 6);
        P6 pa, pn, ph;
        p6_load(r, pa);
        float a3[12], hv[12];
        p6_unpack(pa, a3);
#pragma unroll
        for (int q = 0; q < 12; ++q) a3[q] = (a3[q] - gmean) * grstd;
        p6_pack(a3, vm, pn);
        p6_unpack(pn, a3);  // the bf16 values the backward pass will see
#pragma unroll
        for (int q = 0; q < 12; ++q) hv[q] = silu_f(a3[q] * gw[q] + gb[q]);
        p6_pack(hv, vm, ph);
        p6_store(Bc + (size_t)t * TB_RS, pn);
        p6_store(r, ph);
    }
    run_gstore(ops.h4, 2, 6);
    PHASE(8);
    lds_barrier();
    PHASE(9);
    // F3: conv3: h4 (2,6) -> a5; h5 = SiLU(a5) (operand only); da5 = dh5 * dSiLU(a5) -> S (1,7)
#pragma unroll 1
    TB_BLOCKS(true) {
        FragH b[TB_SB][5];
        P6 p5[TB_SB];
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                conv_bfrags3(L, rowp(t - 1, 2, 6), rowp(t, 2, 6), rowp(t + 1, 2, 6), b[k]);
                const int pi = s0 + k - s_beg;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const uint32_t lo = pi == 0 ? park0.d[i] : park1.d[i], hi = pi == 2 ? park2.d[i] : park3.d[i];
                    p5[k].d[i] = pi < 2 ? lo : hi;
                }
            }
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                const bool tv = t < T_;
                const uint32_t vm = lane_mask(tv);
                const f32x16 a5 = conv_mma(wc, b[k]);
                float hv[12], dv[12], d5[12];
                p6_unpack(p5[k], d5);
#pragma unroll
                for (int r = 0; r < 12; ++r) {
                    silu_dsilu(a5[r], hv[r], dv[r]);
                    dv[r] *= d5[r];
                }
                P6 ph, pd;
                p6_pack(hv, vm, ph);
                p6_pack(dv, vm, pd);
                p6_store(orow(t, 1, 7), pd);
                p6_gstore(ops.h5 + gbase + (size_t)t * TS_CG, ph, tv, true);  // (h5 never enters an image: lane-wise)
            }
    }
    run_gstore(ops.da5, 1, 7);
    load_wfrags<5>(wc, W.C3T, g, L.lane);  // (behind F3's operand stores; a separate register set requested before F3 measured the same)
    PHASE(10);
    lds_barrier();
    PHASE(11);
    // B3: conv3^T: da5 (1,7) -> dh4; dn3 = dh4 * dSiLU(n3) -> S (2,6); GroupNorm backward sums and affine gradients
    fetch_cross(1, 7);
    {
        float sa = 0.f, sb = 0.f, dgw[12], dgb[12];
#pragma unroll
        for (int r = 0; r < 12; ++r) dgw[r] = dgb[r] = 0.f;
#pragma unroll 1
        TB_BLOCKS(false) {
            FragH b[TB_SB][5];
#pragma unroll
            for (int k = 0; k < TB_SB; ++k)
                if (s0 + k < s_end) {
                    const int t = 32 * (s0 + k) + L.n;
                    conv_bfrags3(L, rowx(t - 1, 1, 7), rowp(t, 1, 7), rowx(t + 1, 1, 7), b[k]);
                }
#pragma unroll
            for (int k = 0; k < TB_SB; ++k)
                if (s0 + k < s_end) {
                    const int t = 32 * (s0 + k) + L.n;
                    const uint32_t vm = lane_mask(t < T_);
                    const f32x16 dh4 = conv_mma(wc, b[k]);
                    P6 pn, pd;
                    p6_load(Bc + (size_t)t * TB_RS, pn);
                    float ah[12], dn[12];
                    p6_unpack(pn, ah);
#pragma unroll
                    for (int r = 0; r < 12; ++r) dn[r] = dh4[r] * dsilu_only(ah[r] * gw[r] + gb[r]);
                    p6_pack(dn, vm, pd);
                    p6_unpack(pd, dn);  // masked, bf16 (what the next stage reads back)
#pragma unroll
                    for (int r = 0; r < 12; ++r) {
                        dgw[r] += dn[r] * ah[r];
                        dgb[r] += dn[r];
                        sa += gw[r] * dn[r];
                        sb += gw[r] * dn[r] * ah[r];
                    }
                    p6_store(orow(t, 2, 6), pd);
                }
        }
        sa = wave_sum64(sa);
        sb = wave_sum64(sb);
        if (L.lane == 0) {
            red[(gl * 2 + th) * 2] = sa;
            red[(gl * 2 + th) * 2 + 1] = sb;
        }
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            const float a = half_sum32(dgw[r]), bq = half_sum32(dgb[r]);
            if (L.n == 0) {
                const int ch = gl * TS_CG + (r & 3) + 8 * (r >> 2) + 4 * L.h;
                gnp[(th * 2 + 0) * 96 + ch] = a;
                gnp[(th * 2 + 1) * 96 + ch] = bq;
            }
        }
        // the next stage's "frame -1" / "frame NT" row held this stage's input: clear it (only this wave read it)
        if (L.lane < 6) *reinterpret_cast<u32x2*>(Sc + (size_t)(th ? NT + 6 : 1) * TB_RS + 4 * L.lane) = (u32x2){0u, 0u};
    }
    PHASE(12);
    lds_barrier();
    PHASE(13);
    // B3b (in place, own values): da3 = rstd (gw dn3 - mean(gw dn3) - a3hat mean(gw dn3 a3hat)) -> S (2,6)
    {
        const float msa = (red[gl * 4] + red[gl * 4 + 2]) / cnt, msb = (red[gl * 4 + 1] + red[gl * 4 + 3]) / cnt;
#pragma unroll 2
        for (int s = s_beg; s < s_end; ++s) {
            const int t = 32 * s + L.n;
            const bool tv = t < T_;
            const uint32_t vm = lane_mask(tv);
            bf16_t* r = orow(t, 2, 6);
            P6 pd, pn, po;
            p6_load(r, pd);
            p6_load(Bc + (size_t)t * TB_RS, pn);
            float dn[12], ah[12];
            p6_unpack(pd, dn);
            p6_unpack(pn, ah);
#pragma unroll
            for (int q = 0; q < 12; ++q) dn[q] = grstd * (gw[q] * dn[q] - msa - ah[q] * msb);
            p6_pack(dn, vm, po);
            p6_store(r, po);
        }
        run_gstore(ops.da3, 2, 6);
    }
    PHASE(14);
    lds_barrier();
    PHASE(15);
    // B2: conv2^T: da3 (2,6) -> dh2; da2 = dh2 * dSiLU(a2) (image A) -> S (3,5)
    fetch_cross(2, 6);
#pragma unroll 1
    TB_BLOCKS(false) {
        FragH b[TB_SB][5];
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                conv_bfrags3(L, rowx(t - 1, 2, 6), rowp(t, 2, 6), rowx(t + 1, 2, 6), b[k]);
            }
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                const bool tv = t < T_;
                const uint32_t vm = lane_mask(tv);
                const f32x16 dh2 = conv_mma(wb, b[k]);
                P6 pd2, po;
                p6_load(Ac + (size_t)t * TB_RS, pd2);
                float d2[12], o[12];
                p6_unpack(pd2, d2);
#pragma unroll
                for (int r = 0; r < 12; ++r) o[r] = dh2[r] * d2[r];
                p6_pack(o, vm, po);
                p6_store(orow(t, 3, 5), po);
            }
    }
    run_gstore(ops.da2, 3, 5);
    if (L.lane < 6) *reinterpret_cast<u32x2*>(Sc + (size_t)(th ? NT + 5 : 2) * TB_RS + 4 * L.lane) = (u32x2){0u, 0u};
    PHASE(16);
    lds_barrier();
    PHASE(17);
    // B1: conv1^T: da2 (3,5) -> dh1 -> S (4,4): the halves meet again, strip-contiguous
    fetch_cross(3, 5);
#pragma unroll 1
    TB_BLOCKS(false) {
        FragH b[TB_SB][5];
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                conv_bfrags3(L, rowx(t - 1, 3, 5), rowp(t, 3, 5), rowx(t + 1, 3, 5), b[k]);
            }
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                const f32x16 dh1 = conv_mma(wa, b[k]);
                float o[12];
#pragma unroll
                for (int r = 0; r < 12; ++r) o[r] = dh1[r];
                P6 po;
                p6_pack(o, 0xFFFFFFFFu, po);
                p6_store(orow(t, 4, 4), po);
            }
    }
    PHASE(18);
    lds_barrier();
    // last strip phase: da1 = dh1 * dSiLU(a1) for the wave's 32 frames (the derivative never left this wave's registers) -> operand
    if (w < NS) {
        const int t = 32 * w + L.n;
        const bool tv = t < T_;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            P6 ph, po;
            p6_load(S + (size_t)(4 + t) * TB_RS + q * TS_CG + 4 * L.h, ph);
            float dh[12], d1[12];
            p6_unpack(ph, dh);
            p6_unpack(d1keep[q], d1);
#pragma unroll
            for (int r = 0; r < 12; ++r) dh[r] *= d1[r];
            p6_pack(dh, lane_mask(tv), po);
            p6_store(S + (size_t)(4 + t) * TB_RS + q * TS_CG + 4 * L.h, po);  // in place (this lane's own piece), then out as whole rows
        }
        wave_lds_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q)
            rows_gstore<96>(ops.da1 + ((size_t)(4 * gh + q) * ntok + n0 + 32 * w) * TS_CG, S + (size_t)(4 + 32 * w) * TB_RS + q * TS_CG, 32, T_ - 32 * w);
    }
#undef TB_BLOCKS
    // GroupNorm affine partial sums of this workgroup's 96 channels -> its `part` row (entries [0, 2 FFN); the tail kernel writes the rest)
    for (int i = tid; i < 2 * 96; i += blockDim.x) {
        const int kind = i / 96, ch = i % 96;
        part[(size_t)row * pstride + kind * TS_FFN + gh * 96 + ch] = gnp[(0 * 2 + kind) * 96 + ch] + gnp[(1 * 2 + kind) * 96 + ch];
    }
    PHASE_END();
}
PHASE_READER(nbss_phase_read_tconvffn_bwd_s)

// stats != nullptr: the kernel also writes the LayerNorm (mean, rstd) rows; pstride = floats per `part` row (GroupNorm sums in [0, 2 FFN))
int tconvffn_bwd_s_launch(const nbss_cfg& c, const LayerPtrs& lp, float* part, const void* packed, int layer, const void* x, const void* dy,
                          void* const* opsv, float* stats, int pstride, hipStream_t st) {
    if (c.dtype != NBSS_BF16 || c.T > 256) return NBSS_EUNSUPPORTED;
    const size_t NT = (size_t)((c.T + 31) / 32) * 32;
    const size_t a_el = NT * TB_RS > (size_t)52 * 512 ? NT * TB_RS : (size_t)52 * 512;
    const size_t lds = ((NT + TB_PAD) * TB_RS + a_el + NT * TB_RS) * sizeof(bf16_t) + (16 + 4 * 96) * sizeof(float) + 8 * TS_CG * sizeof(bf16_t) + PHASE_LDS_BYTES;
    if (lds > 160 * 1024) return NBSS_EUNSUPPORTED;
    const bf16_t* pk = (const bf16_t*)packed;
    TsBwdW W = {pk + pack_off(c, layer, K_TS_W1),  pk + pack_off(c, layer, K_TS_W2_T), pk + pack_off(c, layer, K_TS_C1),  pk + pack_off(c, layer, K_TS_C2),
                pk + pack_off(c, layer, K_TS_C3),  pk + pack_off(c, layer, K_TS_C1_T), pk + pack_off(c, layer, K_TS_C2_T), pk + pack_off(c, layer, K_TS_C3_T)};
    TsOps ops = {(bf16_t*)opsv[0], (bf16_t*)opsv[1], (bf16_t*)opsv[2], (bf16_t*)opsv[3], (bf16_t*)opsv[4], (bf16_t*)opsv[5], (bf16_t*)opsv[6], (bf16_t*)opsv[7]};
    int e = NBSS_SET_MAX_LDS(tconvffn_bwd_s_kernel, lds);
    if (e) return e;
    // (timed by the caller's ProfScope together with the tail kernel: one "launch" of the sub-block in bench.py's roofline line)
    NBSS_LAUNCH(tconvffn_bwd_s_kernel, dim3(2 * c.B * c.F), dim3(512), lds, st, c, lp, W, (const bf16_t*)x, (const bf16_t*)dy, part, ops, stats, pstride);
    return NBSS_CHECK_LAUNCH();
}

// =====================================================================================================================================
// Backward (data gradient + the three T-conv weight gradients) from the pre-activations a training-mode forward SAVED (TsSave above).
// Round-2's kernel above recomputes the forward chain (LayerNorm, W1, three convolutions, GroupNorm statistics: two thirds of its wave
// time, profiles/r03a_phase_prof.txt) and then writes six of its activations / gradients out as operands of three wgrad launches.  Here
//   * nothing is recomputed: each stage loads ONE saved pre-activation tensor (group-major, the owning wave's own strips), evaluates
//     SiLU and SiLU' from one sigmoid, puts the activation into a second LDS image H and parks the derivative in registers;
//   * the gradient chain lives in the in-place image S exactly as before (dh5 = W2^T dy in the strip phase, then conv3^T, GroupNorm
//     backward, conv2^T, conv1^T with the two T-halves shifting towards each other);
//   * between two gradient stages both operands of a conv weight gradient are resident as row-major [frame][channel] images —
//     da_{k+1} in S, h_k in H — so wave (group, m) contracts dW[g] = sum_t da[t]^T h[t + tap - 1] over the WHOLE sequence with
//     transposing LDS reads (K = 32 frames per MFMA; m = which 16 of the group's 24 output channels; the 72 (tap, input-channel)
//     columns are five 16-wide tiles whose 4-column pieces carry their own tap shift in the read address), bias gradients through a
//     constant-one operand, and the workgroup's partial leaves in dW's own memory order inside the sequence's `part` row, which
//     affine_reduce folds (the fconv_bwd pattern, profiles/README.md row 45).
// Gone per layer: 12 S.B of operand stores, 12 S.B of operand reads, three wgrad_tr3 launches; new: 8 S.B of saved pre-activations written
// by the forward and read here, 170 KB of partial row per sequence.  Still emitted: da1 (tail + W1 weight gradient, tailw.hip).
//   * (round 5) the W2 weight gradient dW2[o][c] = sum_t dy[t][o] h5[t][c] is contracted here as well: the strip phase leaves its dy strip
//     in a third LDS image D ([frame][96], 200-byte rows: the three images fill 158 of the 160 KB), h5 — rebuilt from a5 in stage 1b — is
//     parked in registers until the conv3 contraction has released H, written there, and after the next barrier waves 0-5 contract one
//     16-channel tile column of the workgroup's 96 x 96 block each (wave 6: db2 = colsum(dy)); the partial leaves inside the sequence's
//     bf16 row as [channel][output] and is folded with the conv weight gradients.  Gone per layer: the h5 operand (2 S.B of lane-wise
//     stores, 2 S.B of reads), the dy re-read and the wgrad_tr3<64,10> launch with its fold.
struct TvIn {
    const bf16_t *a1, *a2, *a3;
    const float* gn;
};
struct TvW {
    const bf16_t *W2T, *C1T, *C2T, *C3T, *C3;
};
#define TV_CONVW (TS_FFN * TS_CG * 3)                 // one conv weight [192][24][3]
#define TV_PSTRIDE (2 * TS_FFN + 3 * TS_FFN + TS_H)    // floats per fp32 `part` row: GN w | GN b | conv1 b | conv2 b | conv3 b | W2 b
#define TV_P16 (3 * TV_CONVW + TS_FFN * TS_H)          // bf16 per `part16` row: the three conv weight gradients, each as [group][tap][in][24 out], then dW2 as [FFN channel][H output]
#define TQ_RS 56                                       // image row stride of the group-pair kernel (112 B: 48 channels + 8; conflict-free 16-byte row reads)
#define TQ_PSTRIDE (5 * TS_FFN)                        // its fp32 `part` row: GN w | GN b | conv1 b | conv2 b | conv3 b
#define TQ_P16 (3 * TV_CONVW)                          // its bf16 row: the three conv weight gradients
#define TV_DRS 100                                     // dy image row stride in elements (200 B; with 208 B the three images miss the 160 KB by 176 bytes)

// dW tile accumulation of one conv group over the whole sequence.  Sg = &S[0][24 gl], Hg = &H[0][24 gl]; bases (bl, bu) as in the stage that
// wrote S; H holds token t at row t + 1 (rows 0 and NT + 1 are zero).
template <int RS>
NBSS_DEV void tv_contract_t(const bf16_t* Sg, const bf16_t* Hg, int bl, int bu, int NS, int NSL, int mt, f32x4 (&acc)[5], f32x4& bsum) {
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    // which frame of the 32-frame k-step a lane's transposing read starts at.  Any map works as long as BOTH operands use it (the contraction sums over
    // the frames).  The natural one — 4 g4 + (l15 >> 2): a 16-lane group reads four CONSECUTIVE rows — puts those rows 52 dwords apart (208-byte image
    // rows): bank offsets {0, 20, 8, 28}, and the 8-dword pieces of rows 0 and 3 overlap (SQ_LDS_BANK_CONFLICT 37 % of the kernel's LDS cycles,
    // round 4).  With every second row per group the offsets are {0, 8, 16, 24}: conflict-free.
#ifdef NBSS_TV_ROWS_NATURAL
    const int rowoff = 4 * g4 + (l15 >> 2);
#else
    const int rowoff = 8 * (g4 >> 1) + (g4 & 1) + 2 * (l15 >> 2);
#endif
    int boff[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        int pc = 4 * j + (l15 & 3);
        pc = pc < 18 ? pc : 17;  // (the last tile's two dummy pieces re-read a valid one; their columns are never flushed)
        const int tap = pc / 6, ch4 = pc - 6 * tap;
        boff[j] = (tap + rowoff) * RS + 4 * ch4;
    }
    const int aoff = rowoff * RS + 16 * mt + 4 * (l15 & 3);
    Frag<bf16_t> ones;
#pragma unroll
    for (int jq = 0; jq < 8; ++jq) frag_set(ones, jq, 1.0f);
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[j] = F32X4_ZERO;
    bsum = F32X4_ZERO;
    for (int ks = 0; ks < NS; ++ks) {
        const int base = (ks < NSL ? bl : bu) + 32 * ks;
        Frag<bf16_t> fa, fb[5];
        frag_load_tr(fa, Sg + (size_t)base * RS + aoff, RS);
#pragma unroll
        for (int j = 0; j < 5; ++j) frag_load_tr(fb[j], Hg + (size_t)(32 * ks) * RS + boff[j], RS);
        bsum = mma(fa, ones, bsum);
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[j] = mma(fa, fb[j], acc[j]);
    }
}
NBSS_DEV void tv_contract(const bf16_t* Sg, const bf16_t* Hg, int bl, int bu, int NS, int NSL, int mt, f32x4 (&acc)[5], f32x4& bsum) {
    tv_contract_t<TB_RS>(Sg, Hg, bl, bu, NS, NSL, mt, acc, bsum);
}
// The per-sequence partial of a conv weight gradient leaves in bf16, as [group][tap][input channel][24 outputs of the group] (a lane's four output channels are one
// 8-byte store; round 6: the rows of one store instruction are 48 bytes apart — one 768-byte region — instead of 384 bytes apart in [tap][in][192 out]); tconv_part_reduce_kernel sums the rows in fp32 and writes the parameter's own [out][in][tap] order.  (Under the reference's
// autocast the weight gradient of a bf16 convolution IS a bf16 tensor before it is cast up for the fp32 parameter; here only the per-sequence
// partial sums are rounded, the sum over the 4 128 sequences is fp32.  Half the partial-row traffic of the fp32 rows: 0.35 instead of 0.7 GB each way.)
// w16 = the row's block of this conv; brow = the fp32 row's bias block of this conv.
NBSS_DEV void tv_flush(bf16_t* __restrict__ w16, float* __restrict__ brow, int g, int mt, const f32x4 (&acc)[5], const f32x4& bsum) {
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4;
    const int oc0 = 16 * mt + 4 * g4;
#ifdef TV_KO_FLUSH  // (timing knock-out, A/B flavour: the contraction stays — the compiler cannot see that the row pointer is never null — its stores go)
    if (w16 != nullptr) return;
#endif
    if (oc0 < TS_CG) {
        const int o0 = g * TS_CG + oc0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int pc = 4 * j + (l15 >> 2);
            if (pc < 18) {
                const int tap = pc / 6, i = (pc - 6 * tap) * 4 + (l15 & 3);
                const u32x2 v = {pack2bf(acc[j][0], acc[j][1]), pack2bf(acc[j][2], acc[j][3])};
                *reinterpret_cast<u32x2*>(w16 + (((size_t)g * 3 + tap) * TS_CG + i) * TS_CG + oc0) = v;
            }
        }
        if (l15 == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) brow[o0 + r] = bsum[r];
        }
    }
}

// Fold of the bf16 partial rows, two stages without atomics: (1) block (x, y) sums slice y of the rows for 256 x 8 consecutive elements (16-byte loads,
// four rows in flight per thread) into slices[y][e] (fp32); (2) one thread per element sums the slices and adds the result to the parameter's own
// [out][in][tap] order in G.  (The first version — 4-byte loads, two rows in flight, 1.3 M atomicAdds — took 130 us for 0.34 GB.)
#define TV_RSL 64
static_assert(TS_H == FK_H && TS_FFN == FK_FFN && TS_CG == FK_TCG && TV_CONVW == FK_TCONVW, "foldk.h");
// (the bodies live in foldk.h: fold.hip's table kernel runs them too)
__global__ __launch_bounds__(256) void tconv_part_reduce1_kernel(const bf16_t* __restrict__ part16, int nrows, float* __restrict__ slices, int P16) {
    fk_p16_slices(part16, nrows, slices, P16, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}
__global__ __launch_bounds__(256) void tconv_part_reduce2_kernel(const float* __restrict__ slices, int nsl, float* __restrict__ G, long long off0, long long off1, long long off2,
                                                                 long long off3, int P16) {
    fk_tconv_final(slices, nsl, G, off0, off1, off2, off3, P16, (int)blockIdx.x);
}

__global__ __launch_bounds__(512) void tconvffn_bwd_v_kernel(nbss_cfg c, LayerPtrs lp, TvW W, TvIn sv, const bf16_t* __restrict__ dy, float* __restrict__ part,
                                                             bf16_t* __restrict__ part16, bf16_t* __restrict__ op_da1) {
    NBSS_LDS(smem);
    const int T_ = c.T, NS = (T_ + 31) >> 5, NT = NS * 32, NSL = NS >> 1, TS = 32 * NSL;
    bf16_t* S = reinterpret_cast<bf16_t*>(smem);         // [NT + TB_PAD][TB_RS]  the gradient chain, in place
    bf16_t* H = S + (size_t)(NT + TB_PAD) * TB_RS;      // [NT + 2][TB_RS]       the activation of the current stage (first: the W2^T window)
    const size_t h_el = (size_t)(NT + 2) * TB_RS > (size_t)24 * 512 ? (size_t)(NT + 2) * TB_RS : (size_t)24 * 512;
    float* red = reinterpret_cast<float*>(H + h_el);     // [4 groups][2 halves][2]
    float* gnp = red + 16;                               // [2 halves][2 kinds][96] GroupNorm affine partial sums
    bf16_t* mbox = reinterpret_cast<bf16_t*>(gnp + 4 * 96);  // [8 waves][24]
    bf16_t* D = mbox + 8 * TS_CG;                            // [NT][TV_DRS]          dy of the sequence (the W2 weight gradient's other operand)
    bf16_t* wl = H;
    PHASE_BEGIN(D + (size_t)NT * TV_DRS);
    const TsLane L;
    const int w = wave_id_u(), tid = threadIdx.x;
    const int row = blockIdx.x >> 1, gh = blockIdx.x & 1;
    const size_t n0 = (size_t)row * T_, ntok = (size_t)c.B * c.F * T_;
    const bf16_t* dyb = dy + n0 * TS_H;

    // group-phase roles (wave = (group gl, half th)) — needed already here: the FIRST group stage's input (a3 of the wave's own strips)
    // is requested together with the strip phase's inputs
    const int gl = w >> 1, th = w & 1, g = 4 * gh + gl;
    const int s_beg = th ? NSL : 0, s_end = th ? NS : NSL, nblk = (s_end - s_beg + TB_SB - 1) / TB_SB;
    const size_t gsv = ((size_t)g * ntok + n0) * TS_CG + 4 * L.h;  // this lane's piece of token 0 in a saved [G][N][24] tensor
    P6 pn0, pn1, pn2, pn3, pd0, pd1, pd2, pd3, ra0, ra1, ra2, ra3;
#define TV_LOADA(k, src, dst)                                                        \
    {                                                                                \
        const int t_ = 32 * (s_beg + (k)) + L.n, tc_ = t_ < T_ ? t_ : T_ - 1;        \
        if (s_beg + (k) < s_end) p6_gload((src) + gsv + (size_t)tc_ * TS_CG, dst);   \
        else                                                                         \
            for (int i_ = 0; i_ < 6; ++i_) dst.d[i_] = 0u;                           \
    }
#define TV_LOADA4(src) TV_LOADA(0, src, ra0) TV_LOADA(1, src, ra1) TV_LOADA(2, src, ra2) TV_LOADA(3, src, ra3)
    // ---- strip phase: dh5 = W2^T dy -> S at bases (1,7) (it becomes da5 in place once a5 has been rebuilt from h4, stage 1b) --------------------
    {
        u32x4 wr[3];  // 24 W2^T fragments of this workgroup's four groups
#pragma unroll
        for (int i = 0; i < 3; ++i) wr[i] = reinterpret_cast<const u32x4*>(W.W2T + (size_t)gh * 24 * 512)[tid + i * 512];
        u32x4 rawd[6];
        const int t = 32 * w + L.n, tc = t < T_ ? t : T_ - 1;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) rawd[ks] = *reinterpret_cast<const u32x4*>(dyb + (size_t)tc * TS_H + 16 * ks + 8 * L.h);
        TV_LOADA4(sv.a3)
        for (int i = tid; i < 4 * TB_RS / 2; i += 512) reinterpret_cast<uint32_t*>(S)[i] = 0u;
        for (int i = tid; i < 5 * TB_RS / 2; i += 512) reinterpret_cast<uint32_t*>(S + (size_t)(NT + 4) * TB_RS)[i] = 0u;
#pragma unroll
        for (int i = 0; i < 3; ++i) reinterpret_cast<u32x4*>(wl)[tid + i * 512] = wr[i];
        PHASE(0);
        lds_barrier();
        PHASE(1);
        if (w < NS) {
            const uint32_t vm = lane_mask(t < T_);
            FragH dq[6];
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) dq[ks].v = __builtin_bit_cast(s16x8, rawd[ks]);
            {  // the strip's dy rows -> D (frames past T: zero rows)
                bf16_t* drow = D + (size_t)t * TV_DRS + 8 * L.h;
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) {
                    *reinterpret_cast<u32x2*>(drow + 16 * ks) = (u32x2){rawd[ks][0] & vm, rawd[ks][1] & vm};
                    *reinterpret_cast<u32x2*>(drow + 16 * ks + 4) = (u32x2){rawd[ks][2] & vm, rawd[ks][3] & vm};
                }
            }
            bf16_t* srow = S + (size_t)((w < NSL ? 1 : 7) + t) * TB_RS + 4 * L.h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                FragH w2t[6];
                load_wfrags<6>(w2t, wl, q, L.lane);
                f32x16 d5 = mma32(w2t[0], dq[0], f32x16_zero());
#pragma unroll
                for (int ks = 1; ks < 6; ++ks) d5 = mma32(w2t[ks], dq[ks], d5);
                float dv[12];
#pragma unroll
                for (int r = 0; r < 12; ++r) dv[r] = d5[r];
                P6 pd;
                p6_pack(dv, vm, pd);
                p6_store(srow + q * TS_CG, pd);
            }
        }
    }
    PHASE(2);

    // ---- group phases: wave = (group gl, half th) ------------------------------------------------------------------------------------------------
    bf16_t* Sc = S + gl * TS_CG;
    bf16_t* Hc = H + gl * TS_CG;
    const int run0 = 32 * s_beg, nrun = 32 * (s_end - s_beg);
    auto rowp = [&](int tt, int bl, int bu) -> const bf16_t* { return Sc + (size_t)((tt < TS ? bl : bu) + tt) * TB_RS; };
    auto orow = [&](int tt, int bl, int bu) -> bf16_t* { return Sc + (size_t)((th ? bu : bl) + tt) * TB_RS + 4 * L.h; };
    bf16_t* mb = mbox + w * TS_CG;
    auto fetch_cross = [&](int bl, int bu) {  // (see tconvffn_bwd_s_kernel: the one row a wave reads across the middle is copied before the stage writes)
        const bf16_t* src = Sc + (size_t)(th ? bl + TS - 1 : bu + TS) * TB_RS;
        if (L.lane < 6) *reinterpret_cast<u32x2*>(mb + 4 * L.lane) = *reinterpret_cast<const u32x2*>(src + 4 * L.lane);
        lds_barrier();
    };
    auto rowx = [&](int tt, int bl, int bu) -> const bf16_t* {
        const bool lower = tt < TS;
        const bf16_t* r = Sc + (size_t)((lower ? bl : bu) + tt) * TB_RS;
        return lower != (th == 0) ? mb : r;
    };
    float gw[12], gb[12];
    chan_vec12(lp.p[P_TF_GN_W] + g * TS_CG, L.h, gw);
    chan_vec12(lp.p[P_TF_GN_B] + g * TS_CG, L.h, gb);
    const float cnt = (float)(TS_CG * T_);
    const float gmean = sv.gn[((size_t)row * TS_G + g) * 2], grstd = sv.gn[((size_t)row * TS_G + g) * 2 + 1];
    float* prow = part + (size_t)row * TV_PSTRIDE + 2 * TS_FFN;   // conv bias blocks of the fp32 row
    bf16_t* prow16 = part16 + (size_t)row * TV_P16;

    FragH wt[5];
    load_wfrags<5>(wt, W.C3T, g, L.lane);
    // per-strip parks pn* / pd* / ra*: named registers + per-dword selects (indexed structs become scratch arrays, see the kernel above)
#define TV_PICK(dst, k_, hib_, q0, q1, q2, q3)                                                            \
    for (int i_ = 0; i_ < 6; ++i_) {                                                                      \
        const uint32_t lo_ = (k_) == 0 ? q0.d[i_] : q1.d[i_], hi_ = (k_) == 0 ? q2.d[i_] : q3.d[i_];      \
        dst.d[i_] = (hib_) ? hi_ : lo_;                                                                   \
    }
#define TB_BLOCKS(fwd_dir)                                                   \
    for (int bi_ = 0; bi_ < nblk; ++bi_)                                     \
        for (int s0 = s_beg + TB_SB * (((fwd_dir) == (th == 0)) ? bi_ : nblk - 1 - bi_), once_ = 1; once_; once_ = 0)

    // stage 1: a3 -> a3hat (parked), h4 = SiLU(a3hat gw + gb) -> H, SiLU' parked
    lds_barrier();  // S = da5 is complete; the weight window (aliasing H) is dead
    PHASE(3);
    if (L.lane < 6) *reinterpret_cast<u32x2*>(Hc + (size_t)(th ? NT + 1 : 0) * TB_RS + 4 * L.lane) = (u32x2){0u, 0u};  // H's halo rows
#define TV_STAGE1(k, RA, PN, PD)                                                          \
    if (s_beg + (k) < s_end) {                                                            \
        const int t = 32 * (s_beg + (k)) + L.n;                                           \
        const uint32_t vm = lane_mask(t < T_);                                            \
        float a3[12], hv[12], dv[12];                                                     \
        p6_unpack(RA, a3);                                                                \
        for (int q = 0; q < 12; ++q) a3[q] = (a3[q] - gmean) * grstd;                     \
        p6_pack(a3, vm, PN);                                                              \
        p6_unpack(PN, a3);                                                                \
        for (int q = 0; q < 12; ++q) silu_dsilu(a3[q] * gw[q] + gb[q], hv[q], dv[q]);     \
        P6 ph;                                                                            \
        p6_pack(hv, vm, ph);                                                              \
        p6_pack(dv, vm, PD);                                                              \
        p6_store(Hc + (size_t)(1 + t) * TB_RS + 4 * L.h, ph);                             \
    } else {                                                                              \
        for (int i_ = 0; i_ < 6; ++i_) PN.d[i_] = PD.d[i_] = 0u;                          \
    }
    TV_STAGE1(0, ra0, pn0, pd0)
    TV_STAGE1(1, ra1, pn1, pd1)
    TV_STAGE1(2, ra2, pn2, pd2)
    TV_STAGE1(3, ra3, pn3, pd3)
#undef TV_STAGE1
    PHASE(4);
    lds_barrier();
    // stage 1b: a5 = conv3(h4) rebuilt from H (own strips; neighbours' rows are complete), (h5, SiLU'(a5)) from one sigmoid: h5 -> parked in ra*
    // (H still holds h4 for the conv3 contraction), da5 = dh5 * SiLU'(a5) in place in S (this lane's own piece)
    {
        FragH wf[5];
        load_wfrags<5>(wf, W.C3, g, L.lane);
#pragma unroll 1
        for (int s0 = s_beg; s0 < s_end; s0 += TB_SB) {
            FragH b[TB_SB][5];
            const bool hib = s0 - s_beg >= 2;
#pragma unroll
            for (int k = 0; k < TB_SB; ++k)
                if (s0 + k < s_end) {
                    const bf16_t* r1 = Hc + (size_t)(1 + 32 * (s0 + k) + L.n) * TB_RS;
                    conv_bfrags3(L, r1 - TB_RS, r1, r1 + TB_RS, b[k]);
                }
#pragma unroll
            for (int k = 0; k < TB_SB; ++k)
                if (s0 + k < s_end) {
                    const int t = 32 * (s0 + k) + L.n;
                    const bool tv = t < T_;
                    const uint32_t vm = lane_mask(tv);
                    const f32x16 a5 = conv_mma(wf, b[k]);
                    bf16_t* r = orow(t, 1, 7);
                    P6 p5, ph, pd;
                    p6_load(r, p5);
                    float hv[12], dv[12], d5[12];
                    p6_unpack(p5, d5);
#pragma unroll
                    for (int q = 0; q < 12; ++q) {
                        silu_dsilu(a5[q], hv[q], dv[q]);
                        dv[q] *= d5[q];
                    }
                    p6_pack(hv, vm, ph);
                    p6_pack(dv, vm, pd);
                    p6_store(r, pd);
#pragma unroll
                    for (int i_ = 0; i_ < 6; ++i_) {  // park h5 of strip 2 hib + k (named registers + selects, as TV_PICK)
                        if (k == 0) {
                            ra0.d[i_] = hib ? ra0.d[i_] : ph.d[i_];
                            ra2.d[i_] = hib ? ph.d[i_] : ra2.d[i_];
                        } else {
                            ra1.d[i_] = hib ? ra1.d[i_] : ph.d[i_];
                            ra3.d[i_] = hib ? ph.d[i_] : ra3.d[i_];
                        }
                    }
                }
        }
    }
    lds_barrier();
    PHASE(5);
    // stage 2: conv3 weight gradient: da5 (1,7) x h4
    {
        f32x4 acc[5], bsum;
        tv_contract(Sc, Hc, 1, 7, NS, NSL, th, acc, bsum);
        tv_flush(prow16 + 2 * TV_CONVW, prow + 2 * TS_FFN, g, th, acc, bsum);
    }
    PHASE(6);
    // B3: conv3^T: da5 (1,7) -> dh4; dn3 = dh4 * SiLU'(n3) -> S (2,6); GroupNorm backward sums and affine gradients
    fetch_cross(1, 7);
    // (every wave is past its conv3 contraction: H is free) h5 -> H for the W2 contraction behind B3's barrier; then the a2 request (needed in B3b)
#define TV_H5STORE(k, RA)                                                                                                      \
    if (s_beg + (k) < s_end) p6_store(Hc + (size_t)(1 + 32 * (s_beg + (k)) + L.n) * TB_RS + 4 * L.h, RA);
    TV_H5STORE(0, ra0) TV_H5STORE(1, ra1) TV_H5STORE(2, ra2) TV_H5STORE(3, ra3)
#undef TV_H5STORE
    TV_LOADA4(sv.a2)
    PHASE(7);
    {
        float sa = 0.f, sb = 0.f, dgw[12], dgb[12];
#pragma unroll
        for (int r = 0; r < 12; ++r) dgw[r] = dgb[r] = 0.f;
#pragma unroll 1
        TB_BLOCKS(false) {
            FragH b[TB_SB][5];
            const bool hib = s0 - s_beg >= 2;
#pragma unroll
            for (int k = 0; k < TB_SB; ++k)
                if (s0 + k < s_end) {
                    const int t = 32 * (s0 + k) + L.n;
                    conv_bfrags3(L, rowx(t - 1, 1, 7), rowp(t, 1, 7), rowx(t + 1, 1, 7), b[k]);
                }
#pragma unroll
            for (int k = 0; k < TB_SB; ++k)
                if (s0 + k < s_end) {
                    const int t = 32 * (s0 + k) + L.n;
                    const uint32_t vm = lane_mask(t < T_);
                    const f32x16 dh4 = conv_mma(wt, b[k]);
                    P6 pn, pdv, pd;
                    TV_PICK(pn, k, hib, pn0, pn1, pn2, pn3)
                    TV_PICK(pdv, k, hib, pd0, pd1, pd2, pd3)
                    float ah[12], d4[12], dn[12];
                    p6_unpack(pn, ah);
                    p6_unpack(pdv, d4);
#pragma unroll
                    for (int r = 0; r < 12; ++r) dn[r] = dh4[r] * d4[r];
                    p6_pack(dn, vm, pd);
                    p6_unpack(pd, dn);  // masked, bf16 (what the next stage reads back)
#pragma unroll
                    for (int r = 0; r < 12; ++r) {
                        dgw[r] += dn[r] * ah[r];
                        dgb[r] += dn[r];
                        sa += gw[r] * dn[r];
                        sb += gw[r] * dn[r] * ah[r];
                    }
                    p6_store(orow(t, 2, 6), pd);
                }
        }
        sa = wave_sum64(sa);
        sb = wave_sum64(sb);
        if (L.lane == 0) {
            red[(gl * 2 + th) * 2] = sa;
            red[(gl * 2 + th) * 2 + 1] = sb;
        }
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            const float a = half_sum32(dgw[r]), bq = half_sum32(dgb[r]);
            if (L.n == 0) {
                const int ch = gl * TS_CG + (r & 3) + 8 * (r >> 2) + 4 * L.h;
                gnp[(th * 2 + 0) * 96 + ch] = a;
                gnp[(th * 2 + 1) * 96 + ch] = bq;
            }
        }
        if (L.lane < 6) *reinterpret_cast<u32x2*>(Sc + (size_t)(th ? NT + 6 : 1) * TB_RS + 4 * L.lane) = (u32x2){0u, 0u};
    }
    load_wfrags<5>(wt, W.C2T, g, L.lane);
    PHASE(8);
    lds_barrier();
    // W2 weight gradient of the workgroup's 96 channels: dW2[o][c] = sum_t dy[t][o] h5[t][c]  (D x H, K = the frames).  Wave w < 6: channel tile w, all six
    // output tiles (one h5 + six dy transposing fragment reads per six MFMAs and 32 frames); wave 6 of the sequence's first workgroup: db2 = colsum(dy)
    {
        const int l15 = L.lane & 15, g4 = L.lane >> 4;
        const int rowoff = 4 * g4 + (l15 >> 2), c4 = 4 * (l15 & 3);
        if (w < 6) {
            f32x4 acc[6];
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) acc[mt] = F32X4_ZERO;
#pragma unroll 2
            for (int ks = 0; ks < NS; ++ks) {
                Frag<bf16_t> fb, fa[6];
                frag_load_tr(fb, H + (size_t)(1 + 32 * ks + rowoff) * TB_RS + 16 * w + c4, TB_RS);
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) frag_load_tr(fa[mt], D + (size_t)(32 * ks + rowoff) * TV_DRS + 16 * mt + c4, TV_DRS);
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) acc[mt] = mma(fa[mt], fb, acc[mt]);
            }
            // lane: outputs 16 mt + 4 g4 + r of channel 96 gh + 16 w + l15 -> the sequence's bf16 row, [channel][output]: one 8-byte store per tile
            bf16_t* dst = prow16 + 3 * TV_CONVW + (size_t)(96 * gh + 16 * w + l15) * TS_H + 4 * g4;
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) *reinterpret_cast<u32x2*>(dst + 16 * mt) = (u32x2){pack2bf(acc[mt][0], acc[mt][1]), pack2bf(acc[mt][2], acc[mt][3])};
        } else if (w == 6 && gh == 0) {
            Frag<bf16_t> ones;
#pragma unroll
            for (int jq = 0; jq < 8; ++jq) frag_set(ones, jq, 1.0f);
            f32x4 bs[6];
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) bs[mt] = F32X4_ZERO;
            for (int ks = 0; ks < NS; ++ks) {
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) {
                    Frag<bf16_t> fa;
                    frag_load_tr(fa, D + (size_t)(32 * ks + rowoff) * TV_DRS + 16 * mt + c4, TV_DRS);
                    bs[mt] = mma(fa, ones, bs[mt]);
                }
            }
            if (l15 == 0) {  // every column holds the sums: rows 4 g4 + r
                float* brow2 = part + (size_t)row * TV_PSTRIDE + 5 * TS_FFN + 4 * g4;
#pragma unroll
                for (int mt = 0; mt < 6; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) brow2[16 * mt + r] = bs[mt][r];
            }
        }
    }
    PHASE(19);
    lds_barrier();  // H is rewritten (h2) by the next stage
    PHASE(9);
    // B3b (in place, own values): da3 = rstd (gw dn3 - mean(gw dn3) - a3hat mean(gw dn3 a3hat)) -> S (2,6);
    // and the next activation: (h2, SiLU'(a2)) from the saved a2: h2 -> H (every wave is past its conv3 contraction), SiLU' parked
    {
        const float msa = (red[gl * 4] + red[gl * 4 + 2]) / cnt, msb = (red[gl * 4 + 1] + red[gl * 4 + 3]) / cnt;
#define TV_STAGE3B(k, RA, PN, PD)                                                         \
    if (s_beg + (k) < s_end) {                                                            \
        const int t = 32 * (s_beg + (k)) + L.n;                                           \
        const uint32_t vm = lane_mask(t < T_);                                            \
        bf16_t* r = orow(t, 2, 6);                                                        \
        P6 pdn, po, ph;                                                                   \
        p6_load(r, pdn);                                                                  \
        float dn[12], ah[12], a2[12], hv[12], dv[12];                                     \
        p6_unpack(pdn, dn);                                                               \
        p6_unpack(PN, ah);                                                                \
        for (int q = 0; q < 12; ++q) dn[q] = grstd * (gw[q] * dn[q] - msa - ah[q] * msb); \
        p6_pack(dn, vm, po);                                                              \
        p6_store(r, po);                                                                  \
        p6_unpack(RA, a2);                                                                \
        for (int q = 0; q < 12; ++q) silu_dsilu(a2[q], hv[q], dv[q]);                     \
        p6_pack(hv, vm, ph);                                                              \
        p6_pack(dv, vm, PD);                                                              \
        p6_store(Hc + (size_t)(1 + t) * TB_RS + 4 * L.h, ph);                             \
    }
        TV_STAGE3B(0, ra0, pn0, pd0)
        TV_STAGE3B(1, ra1, pn1, pd1)
        TV_STAGE3B(2, ra2, pn2, pd2)
        TV_STAGE3B(3, ra3, pn3, pd3)
#undef TV_STAGE3B
        TV_LOADA4(sv.a1)  // needed at the end of B2: requested ahead of the conv2 contraction's stores
    }
    PHASE(10);
    lds_barrier();
    PHASE(11);
    // conv2 weight gradient: da3 (2,6) x h2
    {
        f32x4 acc[5], bsum;
        tv_contract(Sc, Hc, 2, 6, NS, NSL, th, acc, bsum);
        tv_flush(prow16 + 1 * TV_CONVW, prow + 1 * TS_FFN, g, th, acc, bsum);
    }
    PHASE(12);
    // B2: conv2^T: da3 (2,6) -> dh2; da2 = dh2 * SiLU'(a2) (parked) -> S (3,5)
    fetch_cross(2, 6);
    PHASE(13);
#pragma unroll 1
    TB_BLOCKS(false) {
        FragH b[TB_SB][5];
        const bool hib = s0 - s_beg >= 2;
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                conv_bfrags3(L, rowx(t - 1, 2, 6), rowp(t, 2, 6), rowx(t + 1, 2, 6), b[k]);
            }
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                const uint32_t vm = lane_mask(t < T_);
                const f32x16 dh2 = conv_mma(wt, b[k]);
                P6 pdv, po;
                TV_PICK(pdv, k, hib, pd0, pd1, pd2, pd3)
                float d2[12], o[12];
                p6_unpack(pdv, d2);
#pragma unroll
                for (int r = 0; r < 12; ++r) o[r] = dh2[r] * d2[r];
                p6_pack(o, vm, po);
                p6_store(orow(t, 3, 5), po);
            }
    }
    if (L.lane < 6) *reinterpret_cast<u32x2*>(Sc + (size_t)(th ? NT + 5 : 2) * TB_RS + 4 * L.lane) = (u32x2){0u, 0u};
    load_wfrags<5>(wt, W.C1T, g, L.lane);
    // (h1, SiLU'(a1)) from the saved a1: h1 -> H (every wave passed fetch_cross' barrier, i.e. its conv2 contraction), SiLU' parked in pn*
#define TV_STAGE5(k, RA, PN)                                                              \
    if (s_beg + (k) < s_end) {                                                            \
        const int t = 32 * (s_beg + (k)) + L.n;                                           \
        const uint32_t vm = lane_mask(t < T_);                                            \
        float a1[12], hv[12], dv[12];                                                     \
        P6 ph;                                                                            \
        p6_unpack(RA, a1);                                                                \
        for (int q = 0; q < 12; ++q) silu_dsilu(a1[q], hv[q], dv[q]);                     \
        p6_pack(hv, vm, ph);                                                              \
        p6_pack(dv, vm, PN);                                                              \
        p6_store(Hc + (size_t)(1 + t) * TB_RS + 4 * L.h, ph);                             \
    }
    TV_STAGE5(0, ra0, pn0)
    TV_STAGE5(1, ra1, pn1)
    TV_STAGE5(2, ra2, pn2)
    TV_STAGE5(3, ra3, pn3)
#undef TV_STAGE5
    PHASE(14);
    lds_barrier();
    PHASE(15);
    // conv1 weight gradient: da2 (3,5) x h1
    {
        f32x4 acc[5], bsum;
        tv_contract(Sc, Hc, 3, 5, NS, NSL, th, acc, bsum);
        tv_flush(prow16, prow, g, th, acc, bsum);
    }
    PHASE(16);
    // B1: conv1^T: da2 (3,5) -> dh1; da1 = dh1 * SiLU'(a1) (parked) -> S (4,4) -> operand (whole rows, this wave's own run)
    fetch_cross(3, 5);
    PHASE(17);
#pragma unroll 1
    TB_BLOCKS(false) {
        FragH b[TB_SB][5];
        const bool hib = s0 - s_beg >= 2;
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                conv_bfrags3(L, rowx(t - 1, 3, 5), rowp(t, 3, 5), rowx(t + 1, 3, 5), b[k]);
            }
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                const uint32_t vm = lane_mask(t < T_);
                const f32x16 dh1 = conv_mma(wt, b[k]);
                P6 pdv, po;
                TV_PICK(pdv, k, hib, pn0, pn1, pn2, pn3)
                float d1[12], o[12];
                p6_unpack(pdv, d1);
#pragma unroll
                for (int r = 0; r < 12; ++r) o[r] = dh1[r] * d1[r];
                p6_pack(o, vm, po);
                p6_store(orow(t, 4, 4), po);
            }
    }
    wave_lds_sync();
    rows_gstore<128 * 3>(op_da1 + ((size_t)g * ntok + n0 + run0) * TS_CG, Sc + (size_t)(4 + run0) * TB_RS, nrun, T_ - run0);
    PHASE(18);
#undef TB_BLOCKS
#undef TV_PICK
#undef TV_LOADA4
#undef TV_LOADA
    // GroupNorm affine partial sums of this workgroup's 96 channels -> the sequence's `part` row (written in B3, two barriers ago)
    for (int i = tid; i < 2 * 96; i += 512) {
        const int kind = i / 96, ch = i % 96;
        part[(size_t)row * TV_PSTRIDE + kind * TS_FFN + gh * 96 + ch] = gnp[(0 * 2 + kind) * 96 + ch] + gnp[(1 * 2 + kind) * 96 + ch];
    }
    PHASE_END();
}

// The same backward with HALF the workgroup (round 5): one workgroup = one sequence x TWO conv groups (4 waves = (group, half of the frames)), the images
// 112-byte rows, 58 KB of LDS — two workgroups share a CU, so one's cold start (the prologue's memory round trip, 10 % of the big kernel's wave time) and
// barrier waits overlap the other's math.  The strip phase takes two strips per wave; dy is read by four workgroups per sequence (L2).  No room for the
// dy image beside two resident workgroups: this variant emits the h5 operand and the W2 weight gradient is wgrad.hip's, as in round 4.
__global__ __launch_bounds__(256, 2) void tconvffn_bwd_q_kernel(nbss_cfg c, LayerPtrs lp, TvW W, TvIn sv, const bf16_t* __restrict__ dy, float* __restrict__ part,
                                                                bf16_t* __restrict__ part16, bf16_t* __restrict__ op_h5, bf16_t* __restrict__ op_da1, int flip) {
    NBSS_LDS(smem);
    const int T_ = c.T, NS = (T_ + 31) >> 5, NT = NS * 32, NSL = NS >> 1, TS = 32 * NSL;
    bf16_t* S = reinterpret_cast<bf16_t*>(smem);         // [NT + TB_PAD][TQ_RS]  the gradient chain, in place
    bf16_t* H = S + (size_t)(NT + TB_PAD) * TQ_RS;      // [NT + 2][TQ_RS]       the activation of the current stage (first: the W2^T window)
    const size_t h_el = (size_t)(NT + 2) * TQ_RS > (size_t)12 * 512 ? (size_t)(NT + 2) * TQ_RS : (size_t)12 * 512;
    float* red = reinterpret_cast<float*>(H + h_el);     // [2 groups][2 halves][2]
    float* gnp = red + 8;                                // [2 halves][2 kinds][48] GroupNorm affine partial sums
    bf16_t* mbox = reinterpret_cast<bf16_t*>(gnp + 4 * 48);  // [4 waves][24]
    bf16_t* wl = H;
    PHASE_BEGIN(mbox + 4 * TS_CG);
    const TsLane L;
    const int w = wave_id_u(), tid = threadIdx.x;
    // the sequence, its group pair (conv groups 2 gq, 2 gq + 1): four consecutive blocks = four XCDs.  Round 6, both measured and not kept:
    //  * the four pairs of a sequence on ONE XCD (blocks b, b + 8, b + 16, b + 24, as mhsa_bwd_h maps its heads) so that the four readers of a sequence's
    //    dy share an L2: SLOWER, the kernel 0.950 -> 0.973 of the previous build's time in the same call;
    //  * a persistent grid (512 workgroups walking the sequences with a static stride, the conv weight-gradient tiles of four sequences accumulated in
    //    20.7 KB of LDS, 1 152 partial rows instead of 4 128: the partial-row stores are 128 of this kernel's 1 028 us, knocked out): inside a sequence
    //    loop the body no longer fits 256 registers — ~100 VGPRs go to scratch with every lane / wave / kernel-argument value made opaque per iteration
    //    (217 without) — and the launch ran 1 890 us, at one or at thirty-two sequences per workgroup alike.
    const int row0 = blockIdx.x >> 2, gq = blockIdx.x & 3, row = flip ? (int)(gridDim.x >> 2) - 1 - row0 : row0;  // (flip: launch.h)
    const size_t n0 = (size_t)row * T_, ntok = (size_t)c.B * c.F * T_;
    const bf16_t* dyb = dy + n0 * TS_H;

    // group-phase roles (wave = (group gl, half th)) — needed already here: the FIRST group stage's input (a3 of the wave's own strips)
    // is requested together with the strip phase's inputs
    const int gl = w >> 1, th = w & 1, g = 2 * gq + gl;
    const int s_beg = th ? NSL : 0, s_end = th ? NS : NSL, nblk = (s_end - s_beg + TB_SB - 1) / TB_SB;
    const size_t gsv = ((size_t)g * ntok + n0) * TS_CG + 4 * L.h;  // this lane's piece of token 0 in a saved [G][N][24] tensor
    P6 pn0, pn1, pn2, pn3, pd0, pd1, pd2, pd3, ra0, ra1, ra2, ra3;
#define TV_LOADA(k, src, dst)                                                        \
    {                                                                                \
        const int t_ = 32 * (s_beg + (k)) + L.n, tc_ = t_ < T_ ? t_ : T_ - 1;        \
        if (s_beg + (k) < s_end) p6_gload((src) + gsv + (size_t)tc_ * TS_CG, dst);   \
        else                                                                         \
            for (int i_ = 0; i_ < 6; ++i_) dst.d[i_] = 0u;                           \
    }
#define TV_LOADA4(src) TV_LOADA(0, src, ra0) TV_LOADA(1, src, ra1) TV_LOADA(2, src, ra2) TV_LOADA(3, src, ra3)
    // ---- strip phase: dh5 = W2^T dy -> S at bases (1,7) (it becomes da5 in place once a5 has been rebuilt from h4, stage 1b) --------------------
    {
        u32x4 wr[3];  // 12 W2^T fragments of this workgroup's two groups
#pragma unroll
        for (int i = 0; i < 3; ++i) wr[i] = reinterpret_cast<const u32x4*>(W.W2T + (size_t)gq * 12 * 512)[tid + i * 256];
        // four waves, up to eight strips: wave w takes strips w and w + 4 (both requested before the first wait)
        u32x4 rawd[2][6];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int t = 32 * (w + 4 * k2) + L.n, tc = t < T_ ? t : T_ - 1;
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) rawd[k2][ks] = *reinterpret_cast<const u32x4*>(dyb + (size_t)tc * TS_H + 16 * ks + 8 * L.h);
        }
        TV_LOADA4(sv.a3)
        for (int i = tid; i < 4 * TQ_RS / 2; i += 256) reinterpret_cast<uint32_t*>(S)[i] = 0u;
        for (int i = tid; i < 5 * TQ_RS / 2; i += 256) reinterpret_cast<uint32_t*>(S + (size_t)(NT + 4) * TQ_RS)[i] = 0u;
#pragma unroll
        for (int i = 0; i < 3; ++i) reinterpret_cast<u32x4*>(wl)[tid + i * 256] = wr[i];
        PHASE(0);
        lds_barrier();
        PHASE(1);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int sw = w + 4 * k2, t = 32 * sw + L.n;
            if (sw < NS) {
                const uint32_t vm = lane_mask(t < T_);
                FragH dq[6];
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) dq[ks].v = __builtin_bit_cast(s16x8, rawd[k2][ks]);
                bf16_t* srow = S + (size_t)((sw < NSL ? 1 : 7) + t) * TQ_RS + 4 * L.h;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    FragH w2t[6];
                    load_wfrags<6>(w2t, wl, q, L.lane);
                    f32x16 d5 = mma32(w2t[0], dq[0], f32x16_zero());
#pragma unroll
                    for (int ks = 1; ks < 6; ++ks) d5 = mma32(w2t[ks], dq[ks], d5);
                    float dv[12];
#pragma unroll
                    for (int r = 0; r < 12; ++r) dv[r] = d5[r];
                    P6 pd;
                    p6_pack(dv, vm, pd);
                    p6_store(srow + q * TS_CG, pd);
                }
            }
        }
    }
    PHASE(2);

    // ---- group phases: wave = (group gl, half th) ------------------------------------------------------------------------------------------------
    bf16_t* Sc = S + gl * TS_CG;
    bf16_t* Hc = H + gl * TS_CG;
    const int run0 = 32 * s_beg, nrun = 32 * (s_end - s_beg);
    auto rowp = [&](int tt, int bl, int bu) -> const bf16_t* { return Sc + (size_t)((tt < TS ? bl : bu) + tt) * TQ_RS; };
    auto orow = [&](int tt, int bl, int bu) -> bf16_t* { return Sc + (size_t)((th ? bu : bl) + tt) * TQ_RS + 4 * L.h; };
    bf16_t* mb = mbox + w * TS_CG;
    auto fetch_cross = [&](int bl, int bu) {  // (see tconvffn_bwd_s_kernel: the one row a wave reads across the middle is copied before the stage writes)
        const bf16_t* src = Sc + (size_t)(th ? bl + TS - 1 : bu + TS) * TQ_RS;
        if (L.lane < 6) *reinterpret_cast<u32x2*>(mb + 4 * L.lane) = *reinterpret_cast<const u32x2*>(src + 4 * L.lane);
        lds_barrier();
    };
    auto rowx = [&](int tt, int bl, int bu) -> const bf16_t* {
        const bool lower = tt < TS;
        const bf16_t* r = Sc + (size_t)((lower ? bl : bu) + tt) * TQ_RS;
        return lower != (th == 0) ? mb : r;
    };
    float gw[12], gb[12];
    chan_vec12(lp.p[P_TF_GN_W] + g * TS_CG, L.h, gw);
    chan_vec12(lp.p[P_TF_GN_B] + g * TS_CG, L.h, gb);
    const float cnt = (float)(TS_CG * T_);
    const float gmean = sv.gn[((size_t)row * TS_G + g) * 2], grstd = sv.gn[((size_t)row * TS_G + g) * 2 + 1];
    float* prow = part + (size_t)row * TQ_PSTRIDE + 2 * TS_FFN;   // conv bias blocks of the fp32 row
    bf16_t* prow16 = part16 + (size_t)row * TQ_P16;

    FragH wt[5];
    load_wfrags<5>(wt, W.C3T, g, L.lane);
    // per-strip parks pn* / pd* / ra*: named registers + per-dword selects (indexed structs become scratch arrays, see the kernel above)
#define TV_PICK(dst, k_, hib_, q0, q1, q2, q3)                                                            \
    for (int i_ = 0; i_ < 6; ++i_) {                                                                      \
        const uint32_t lo_ = (k_) == 0 ? q0.d[i_] : q1.d[i_], hi_ = (k_) == 0 ? q2.d[i_] : q3.d[i_];      \
        dst.d[i_] = (hib_) ? hi_ : lo_;                                                                   \
    }
#define TB_BLOCKS(fwd_dir)                                                   \
    for (int bi_ = 0; bi_ < nblk; ++bi_)                                     \
        for (int s0 = s_beg + TB_SB * (((fwd_dir) == (th == 0)) ? bi_ : nblk - 1 - bi_), once_ = 1; once_; once_ = 0)

    // stage 1: a3 -> a3hat (parked), h4 = SiLU(a3hat gw + gb) -> H, SiLU' parked
    lds_barrier();  // S = da5 is complete; the weight window (aliasing H) is dead
    PHASE(3);
    if (L.lane < 6) *reinterpret_cast<u32x2*>(Hc + (size_t)(th ? NT + 1 : 0) * TQ_RS + 4 * L.lane) = (u32x2){0u, 0u};  // H's halo rows
#define TV_STAGE1(k, RA, PN, PD)                                                          \
    if (s_beg + (k) < s_end) {                                                            \
        const int t = 32 * (s_beg + (k)) + L.n;                                           \
        const uint32_t vm = lane_mask(t < T_);                                            \
        float a3[12], hv[12], dv[12];                                                     \
        p6_unpack(RA, a3);                                                                \
        for (int q = 0; q < 12; ++q) a3[q] = (a3[q] - gmean) * grstd;                     \
        p6_pack(a3, vm, PN);                                                              \
        p6_unpack(PN, a3);                                                                \
        for (int q = 0; q < 12; ++q) silu_dsilu(a3[q] * gw[q] + gb[q], hv[q], dv[q]);     \
        P6 ph;                                                                            \
        p6_pack(hv, vm, ph);                                                              \
        p6_pack(dv, vm, PD);                                                              \
        p6_store(Hc + (size_t)(1 + t) * TQ_RS + 4 * L.h, ph);                             \
    } else {                                                                              \
        for (int i_ = 0; i_ < 6; ++i_) PN.d[i_] = PD.d[i_] = 0u;                          \
    }
    TV_STAGE1(0, ra0, pn0, pd0)
    TV_STAGE1(1, ra1, pn1, pd1)
    TV_STAGE1(2, ra2, pn2, pd2)
    TV_STAGE1(3, ra3, pn3, pd3)
#undef TV_STAGE1
    TV_LOADA4(sv.a2)  // needed in B3b: requested now, AHEAD of the contraction's partial-row stores (loads and stores share vmcnt)
    PHASE(4);
    lds_barrier();
    // stage 1b: a5 = conv3(h4) rebuilt from H (own strips; neighbours' rows are complete), (h5, SiLU'(a5)) from one sigmoid: h5 -> operand,
    // da5 = dh5 * SiLU'(a5) in place in S (this lane's own piece)
    {
        FragH wf[5];
        load_wfrags<5>(wf, W.C3, g, L.lane);
#pragma unroll 1
        for (int s0 = s_beg; s0 < s_end; s0 += TB_SB) {
            FragH b[TB_SB][5];
#pragma unroll
            for (int k = 0; k < TB_SB; ++k)
                if (s0 + k < s_end) {
                    const bf16_t* r1 = Hc + (size_t)(1 + 32 * (s0 + k) + L.n) * TQ_RS;
                    conv_bfrags3(L, r1 - TQ_RS, r1, r1 + TQ_RS, b[k]);
                }
#pragma unroll
            for (int k = 0; k < TB_SB; ++k)
                if (s0 + k < s_end) {
                    const int t = 32 * (s0 + k) + L.n;
                    const bool tv = t < T_;
                    const uint32_t vm = lane_mask(tv);
                    const f32x16 a5 = conv_mma(wf, b[k]);
                    bf16_t* r = orow(t, 1, 7);
                    P6 p5, ph, pd;
                    p6_load(r, p5);
                    float hv[12], dv[12], d5[12];
                    p6_unpack(p5, d5);
#pragma unroll
                    for (int q = 0; q < 12; ++q) {
                        silu_dsilu(a5[q], hv[q], dv[q]);
                        dv[q] *= d5[q];
                    }
                    p6_pack(hv, vm, ph);
                    p6_pack(dv, vm, pd);
                    p6_store(r, pd);
                    // (lane-wise on purpose.  Round 6: with these stores knocked out the kernel runs 1 113 -> 995 us, but as whole rows through a per-wave
                    //  staging strip — 16-byte pieces in address order, the form the forward kernel's saves and the da1 operand use — it was no faster:
                    //  1 072 -> 1 089 us in the same call.  The cost is the operand's bytes, not the form of its stores.)
                    p6_gstore(op_h5 + gsv + (size_t)t * TS_CG, ph, tv, true);
                }
        }
    }
    lds_barrier();
    PHASE(5);
    // stage 2: conv3 weight gradient: da5 (1,7) x h4
    {
        f32x4 acc[5], bsum;
        tv_contract_t<TQ_RS>(Sc, Hc, 1, 7, NS, NSL, th, acc, bsum);
        tv_flush(prow16 + 2 * TV_CONVW, prow + 2 * TS_FFN, g, th, acc, bsum);
    }
    PHASE(6);
    // B3: conv3^T: da5 (1,7) -> dh4; dn3 = dh4 * SiLU'(n3) -> S (2,6); GroupNorm backward sums and affine gradients
    fetch_cross(1, 7);
    PHASE(7);
    {
        float sa = 0.f, sb = 0.f, dgw[12], dgb[12];
#pragma unroll
        for (int r = 0; r < 12; ++r) dgw[r] = dgb[r] = 0.f;
#pragma unroll 1
        TB_BLOCKS(false) {
            FragH b[TB_SB][5];
            const bool hib = s0 - s_beg >= 2;
#pragma unroll
            for (int k = 0; k < TB_SB; ++k)
                if (s0 + k < s_end) {
                    const int t = 32 * (s0 + k) + L.n;
                    conv_bfrags3(L, rowx(t - 1, 1, 7), rowp(t, 1, 7), rowx(t + 1, 1, 7), b[k]);
                }
#pragma unroll
            for (int k = 0; k < TB_SB; ++k)
                if (s0 + k < s_end) {
                    const int t = 32 * (s0 + k) + L.n;
                    const uint32_t vm = lane_mask(t < T_);
                    const f32x16 dh4 = conv_mma(wt, b[k]);
                    P6 pn, pdv, pd;
                    TV_PICK(pn, k, hib, pn0, pn1, pn2, pn3)
                    TV_PICK(pdv, k, hib, pd0, pd1, pd2, pd3)
                    float ah[12], d4[12], dn[12];
                    p6_unpack(pn, ah);
                    p6_unpack(pdv, d4);
#pragma unroll
                    for (int r = 0; r < 12; ++r) dn[r] = dh4[r] * d4[r];
                    p6_pack(dn, vm, pd);
                    p6_unpack(pd, dn);  // masked, bf16 (what the next stage reads back)
#pragma unroll
                    for (int r = 0; r < 12; ++r) {
                        dgw[r] += dn[r] * ah[r];
                        dgb[r] += dn[r];
                        sa += gw[r] * dn[r];
                        sb += gw[r] * dn[r] * ah[r];
                    }
                    p6_store(orow(t, 2, 6), pd);
                }
        }
        sa = wave_sum64(sa);
        sb = wave_sum64(sb);
        if (L.lane == 0) {
            red[(gl * 2 + th) * 2] = sa;
            red[(gl * 2 + th) * 2 + 1] = sb;
        }
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            const float a = half_sum32(dgw[r]), bq = half_sum32(dgb[r]);
            if (L.n == 0) {
                const int ch = gl * TS_CG + (r & 3) + 8 * (r >> 2) + 4 * L.h;
                gnp[(th * 2 + 0) * 48 + ch] = a;
                gnp[(th * 2 + 1) * 48 + ch] = bq;
            }
        }
        if (L.lane < 6) *reinterpret_cast<u32x2*>(Sc + (size_t)(th ? NT + 6 : 1) * TQ_RS + 4 * L.lane) = (u32x2){0u, 0u};
    }
    load_wfrags<5>(wt, W.C2T, g, L.lane);
    PHASE(8);
    lds_barrier();
    PHASE(9);
    // B3b (in place, own values): da3 = rstd (gw dn3 - mean(gw dn3) - a3hat mean(gw dn3 a3hat)) -> S (2,6);
    // and the next activation: (h2, SiLU'(a2)) from the saved a2: h2 -> H (every wave is past its conv3 contraction), SiLU' parked
    {
        const float msa = (red[gl * 4] + red[gl * 4 + 2]) / cnt, msb = (red[gl * 4 + 1] + red[gl * 4 + 3]) / cnt;
#define TV_STAGE3B(k, RA, PN, PD)                                                         \
    if (s_beg + (k) < s_end) {                                                            \
        const int t = 32 * (s_beg + (k)) + L.n;                                           \
        const uint32_t vm = lane_mask(t < T_);                                            \
        bf16_t* r = orow(t, 2, 6);                                                        \
        P6 pdn, po, ph;                                                                   \
        p6_load(r, pdn);                                                                  \
        float dn[12], ah[12], a2[12], hv[12], dv[12];                                     \
        p6_unpack(pdn, dn);                                                               \
        p6_unpack(PN, ah);                                                                \
        for (int q = 0; q < 12; ++q) dn[q] = grstd * (gw[q] * dn[q] - msa - ah[q] * msb); \
        p6_pack(dn, vm, po);                                                              \
        p6_store(r, po);                                                                  \
        p6_unpack(RA, a2);                                                                \
        for (int q = 0; q < 12; ++q) silu_dsilu(a2[q], hv[q], dv[q]);                     \
        p6_pack(hv, vm, ph);                                                              \
        p6_pack(dv, vm, PD);                                                              \
        p6_store(Hc + (size_t)(1 + t) * TQ_RS + 4 * L.h, ph);                             \
    }
        TV_STAGE3B(0, ra0, pn0, pd0)
        TV_STAGE3B(1, ra1, pn1, pd1)
        TV_STAGE3B(2, ra2, pn2, pd2)
        TV_STAGE3B(3, ra3, pn3, pd3)
#undef TV_STAGE3B
        TV_LOADA4(sv.a1)  // needed at the end of B2: requested ahead of the conv2 contraction's stores
    }
    PHASE(10);
    lds_barrier();
    PHASE(11);
    // conv2 weight gradient: da3 (2,6) x h2
    {
        f32x4 acc[5], bsum;
        tv_contract_t<TQ_RS>(Sc, Hc, 2, 6, NS, NSL, th, acc, bsum);
        tv_flush(prow16 + 1 * TV_CONVW, prow + 1 * TS_FFN, g, th, acc, bsum);
    }
    PHASE(12);
    // B2: conv2^T: da3 (2,6) -> dh2; da2 = dh2 * SiLU'(a2) (parked) -> S (3,5)
    fetch_cross(2, 6);
    PHASE(13);
#pragma unroll 1
    TB_BLOCKS(false) {
        FragH b[TB_SB][5];
        const bool hib = s0 - s_beg >= 2;
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                conv_bfrags3(L, rowx(t - 1, 2, 6), rowp(t, 2, 6), rowx(t + 1, 2, 6), b[k]);
            }
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                const uint32_t vm = lane_mask(t < T_);
                const f32x16 dh2 = conv_mma(wt, b[k]);
                P6 pdv, po;
                TV_PICK(pdv, k, hib, pd0, pd1, pd2, pd3)
                float d2[12], o[12];
                p6_unpack(pdv, d2);
#pragma unroll
                for (int r = 0; r < 12; ++r) o[r] = dh2[r] * d2[r];
                p6_pack(o, vm, po);
                p6_store(orow(t, 3, 5), po);
            }
    }
    if (L.lane < 6) *reinterpret_cast<u32x2*>(Sc + (size_t)(th ? NT + 5 : 2) * TQ_RS + 4 * L.lane) = (u32x2){0u, 0u};
    load_wfrags<5>(wt, W.C1T, g, L.lane);
    // (h1, SiLU'(a1)) from the saved a1: h1 -> H (every wave passed fetch_cross' barrier, i.e. its conv2 contraction), SiLU' parked in pn*
#define TV_STAGE5(k, RA, PN)                                                              \
    if (s_beg + (k) < s_end) {                                                            \
        const int t = 32 * (s_beg + (k)) + L.n;                                           \
        const uint32_t vm = lane_mask(t < T_);                                            \
        float a1[12], hv[12], dv[12];                                                     \
        P6 ph;                                                                            \
        p6_unpack(RA, a1);                                                                \
        for (int q = 0; q < 12; ++q) silu_dsilu(a1[q], hv[q], dv[q]);                     \
        p6_pack(hv, vm, ph);                                                              \
        p6_pack(dv, vm, PN);                                                              \
        p6_store(Hc + (size_t)(1 + t) * TQ_RS + 4 * L.h, ph);                             \
    }
    TV_STAGE5(0, ra0, pn0)
    TV_STAGE5(1, ra1, pn1)
    TV_STAGE5(2, ra2, pn2)
    TV_STAGE5(3, ra3, pn3)
#undef TV_STAGE5
    PHASE(14);
    lds_barrier();
    PHASE(15);
    // conv1 weight gradient: da2 (3,5) x h1
    {
        f32x4 acc[5], bsum;
        tv_contract_t<TQ_RS>(Sc, Hc, 3, 5, NS, NSL, th, acc, bsum);
        tv_flush(prow16, prow, g, th, acc, bsum);
    }
    PHASE(16);
    // B1: conv1^T: da2 (3,5) -> dh1; da1 = dh1 * SiLU'(a1) (parked) -> S (4,4) -> operand (whole rows, this wave's own run)
    fetch_cross(3, 5);
    PHASE(17);
#pragma unroll 1
    TB_BLOCKS(false) {
        FragH b[TB_SB][5];
        const bool hib = s0 - s_beg >= 2;
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                conv_bfrags3(L, rowx(t - 1, 3, 5), rowp(t, 3, 5), rowx(t + 1, 3, 5), b[k]);
            }
#pragma unroll
        for (int k = 0; k < TB_SB; ++k)
            if (s0 + k < s_end) {
                const int t = 32 * (s0 + k) + L.n;
                const uint32_t vm = lane_mask(t < T_);
                const f32x16 dh1 = conv_mma(wt, b[k]);
                P6 pdv, po;
                TV_PICK(pdv, k, hib, pn0, pn1, pn2, pn3)
                float d1[12], o[12];
                p6_unpack(pdv, d1);
#pragma unroll
                for (int r = 0; r < 12; ++r) o[r] = dh1[r] * d1[r];
                p6_pack(o, vm, po);
                p6_store(orow(t, 4, 4), po);
            }
    }
    wave_lds_sync();
    rows_gstore_t<128 * 3, TQ_RS>(op_da1 + ((size_t)g * ntok + n0 + run0) * TS_CG, Sc + (size_t)(4 + run0) * TQ_RS, nrun, T_ - run0);
    PHASE(18);
#undef TB_BLOCKS
#undef TV_PICK
#undef TV_LOADA4
#undef TV_LOADA
    // GroupNorm affine partial sums of this workgroup's 96 channels -> the sequence's `part` row (written in B3, two barriers ago)
    for (int i = tid; i < 2 * 48; i += 256) {
        const int kind = i / 48, ch = i % 48;
        part[(size_t)row * TQ_PSTRIDE + kind * TS_FFN + gq * 48 + ch] = gnp[(0 * 2 + kind) * 48 + ch] + gnp[(1 * 2 + kind) * 48 + ch];
    }
    PHASE_END();
}
PHASE_READER(nbss_phase_read_tconvffn_bwd_v)

size_t tconvffn_v_part_bytes(const nbss_cfg& c) { return (size_t)c.B * c.F * (TV_PSTRIDE * sizeof(float) + TV_P16 * sizeof(bf16_t)); }
// fold of the bf16 weight-gradient partial rows into G (fp32); offs = flat-gradient offsets of the three conv weights and of W2; `slices`: TV_RSL x TV_P16 floats of scratch
// with_w2: the rows carry the dW2 block behind the three conv blocks (tconvffn_bwd_v_kernel) or not (tconvffn_bwd_q_kernel)
// first stage alone, for other kernels' bf16 partial rows (fconv.hip)
int part16_slices_launch(const void* part16, int nrows, float* slices, int p16, int* nsl_out, hipStream_t st, bool batch) {
    const int nsl = nrows < TV_RSL ? nrows : TV_RSL;
    *nsl_out = nsl;
    if (g_fold && batch) {  // inside a FoldScope (fold.h; `slices` is the caller's allocation from the scope's pool): first stage
        g_fold->st = st;
        FoldItem it;
        it.kind = FK_P16_SLICES;
        it.gx = (p16 / 8 + 255) / 256; it.gy = nsl; it.nblk = it.gx * it.gy;
        it.u.p16.part16 = part16; it.u.p16.nrows = nrows; it.u.p16.p16 = p16; it.u.p16.nsl = nsl; it.u.p16.slices = slices; it.u.p16.G = nullptr;
        return g_fold->add(1, it);
    }
    NBSS_FOLD_LAUNCH(tconv_part_reduce1_kernel, dim3((p16 / 8 + 255) / 256, nsl), dim3(256), 0, st, (const bf16_t*)part16, nrows, slices, p16);
    return NBSS_CHECK_LAUNCH();
}
int tconvffn_v_reduce16(const nbss_cfg& c, const void* part16, float* slices, float* G, const long long* offs, bool with_w2, hipStream_t st) {
    const int nrows = c.B * c.F, nsl = nrows < TV_RSL ? nrows : TV_RSL, p16 = with_w2 ? TV_P16 : TQ_P16;
    if (g_fold) {  // inside a FoldScope (fold.h): the slice sums come from the scope's pool, the two passes join its first and second stage
        int err;
        void* sl = g_fold->alloc((size_t)nsl * p16 * sizeof(float), &err);
        if (err) return err;
        if (sl) {
            g_fold->st = st;
            FoldItem it;
            it.kind = FK_P16_SLICES;
            it.gx = (p16 / 8 + 255) / 256; it.gy = nsl; it.nblk = it.gx * it.gy;
            it.u.p16.part16 = part16; it.u.p16.nrows = nrows; it.u.p16.p16 = p16; it.u.p16.nsl = nsl; it.u.p16.slices = (float*)sl; it.u.p16.G = G;
            for (int k = 0; k < 4; ++k) it.u.p16.off[k] = offs[k];
            if ((err = g_fold->add(1, it))) return err;
            it.kind = FK_TCONV_FINAL;
            it.gx = (p16 + 255) / 256; it.gy = 1; it.nblk = it.gx;
            return g_fold->add(2, it);
        }
        if ((err = g_fold->flush())) return err;
    }
    NBSS_FOLD_LAUNCH(tconv_part_reduce1_kernel, dim3((p16 / 8 + 255) / 256, nsl), dim3(256), 0, st, (const bf16_t*)part16, nrows, slices, p16);
    int e = NBSS_CHECK_LAUNCH();
    if (e) return e;
    NBSS_FOLD_LAUNCH(tconv_part_reduce2_kernel, dim3((p16 + 255) / 256), dim3(256), 0, st, (const float*)slices, nsl, G, offs[0], offs[1], offs[2], offs[3], p16);
    return NBSS_CHECK_LAUNCH();
}
size_t tconvffn_v_slices_bytes() { return (size_t)TV_RSL * TV_P16 * sizeof(float); }
// data-gradient + T-conv weight-gradient kernel from saved pre-activations; `part`: [B*F][TV_PSTRIDE] floats, then [B*F][TV_P16] bf16
int tconvffn_bwd_v_launch(const nbss_cfg& c, const LayerPtrs& lp, float* part, const void* packed, int layer, const void* dy, void* tsave, void* op_da1,
                          hipStream_t st) {
    bf16_t* part16 = reinterpret_cast<bf16_t*>(part + (size_t)c.B * c.F * TV_PSTRIDE);
    if (c.dtype != NBSS_BF16 || c.T > 256 || !tsave) return NBSS_EUNSUPPORTED;
    const size_t NT = (size_t)((c.T + 31) / 32) * 32;
    const size_t h_el = (NT + 2) * TB_RS > (size_t)24 * 512 ? (NT + 2) * TB_RS : (size_t)24 * 512;
    const size_t lds = ((NT + TB_PAD) * TB_RS + h_el + NT * TV_DRS) * sizeof(bf16_t) + (16 + 4 * 96) * sizeof(float) + 8 * TS_CG * sizeof(bf16_t) + PHASE_LDS_BYTES;
    if (lds > 160 * 1024) return NBSS_EUNSUPPORTED;  // (T = 256: 161 968 of 163 840 bytes; the diagnostic build's timer slots fit up to T = 224)
    const bf16_t* pk = (const bf16_t*)packed;
    TvW W = {pk + pack_off(c, layer, K_TS_W2_T), pk + pack_off(c, layer, K_TS_C1_T), pk + pack_off(c, layer, K_TS_C2_T), pk + pack_off(c, layer, K_TS_C3_T),
             pk + pack_off(c, layer, K_TS_C3)};
    const TsSave s = ts_save_ptrs(c, tsave);
    TvIn in = {s.a1, s.a2, s.a3, s.gn};
    int e = NBSS_SET_MAX_LDS(tconvffn_bwd_v_kernel, lds);
    if (e) return e;
    NBSS_LAUNCH(tconvffn_bwd_v_kernel, dim3(2 * c.B * c.F), dim3(512), lds, st, c, lp, W, in, (const bf16_t*)dy, part, part16, (bf16_t*)op_da1);
    return NBSS_CHECK_LAUNCH();
}
int tconvffn_bwd_q_launch(const nbss_cfg& c, const LayerPtrs& lp, float* part, const void* packed, int layer, const void* dy, void* tsave, void* op_h5, void* op_da1,
                          hipStream_t st) {
    bf16_t* part16 = reinterpret_cast<bf16_t*>(part + (size_t)c.B * c.F * TQ_PSTRIDE);
    if (c.dtype != NBSS_BF16 || c.T > 256 || !tsave) return NBSS_EUNSUPPORTED;
    const size_t NT = (size_t)((c.T + 31) / 32) * 32;
    const size_t h_el = (NT + 2) * TQ_RS > (size_t)12 * 512 ? (NT + 2) * TQ_RS : (size_t)12 * 512;
    const size_t lds = ((NT + TB_PAD) * TQ_RS + h_el) * sizeof(bf16_t) + (8 + 4 * 48) * sizeof(float) + 4 * TS_CG * sizeof(bf16_t) + PHASE_LDS_BYTES;
    const bf16_t* pk = (const bf16_t*)packed;
    TvW W = {pk + pack_off(c, layer, K_TS_W2_T), pk + pack_off(c, layer, K_TS_C1_T), pk + pack_off(c, layer, K_TS_C2_T), pk + pack_off(c, layer, K_TS_C3_T),
             pk + pack_off(c, layer, K_TS_C3)};
    const TsSave s = ts_save_ptrs(c, tsave);
    TvIn in = {s.a1, s.a2, s.a3, s.gn};
    int e = NBSS_SET_MAX_LDS(tconvffn_bwd_q_kernel, lds);
    if (e) return e;
    NBSS_LAUNCH(tconvffn_bwd_q_kernel, dim3(4 * c.B * c.F), dim3(256), lds, st, c, lp, W, in, (const bf16_t*)dy, part, part16, (bf16_t*)op_h5, (bf16_t*)op_da1,
                walk_flip_next());
    return NBSS_CHECK_LAUNCH();
}
float* tconvffn_save_ln_stats(const nbss_cfg& c, void* tsave) { return ts_save_ptrs(c, tsave).ln; }
