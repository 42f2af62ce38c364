// gemm_g.hip — dense per-token linear maps of the geometry-generic path (SpatialNet-large: in_proj 192 -> 576, out_proj, the two FFN maps
// 192 <-> 384 and their data gradients; models/arch/SpatialNet.py:93-114 at the widths of configs/SpatialNet.yaml "for large").
//
//   Y[n][o] = epilogue(sum_i X[n][i] W[o][i])      bf16 stream, fp32 accumulation, weights = MFMA A operand, tokens = N dimension
//
// Workgroup = 4 waves on a 128-row x 192-output tile (wave: 64 rows x 96 outputs = 24 accumulator tiles), persistent over its share of
// the tiles.  Both operands reach the LDS by DMA (global_load_lds_dwordx4, no register stop) in K-slabs of 32 through a 3-stage ring; a
// DMA instruction writes one FRAGMENT BLOCK: the 16 rows x 32 k of one MFMA operand in reader-lane order (lane l's 16-byte piece at 16 l),
// so the fragment read is one conflict-free ds_read_b128 at base + 16 lane and the global side fetches 64 contiguous bytes per row.
// Two workgroups share a CU (76 KB of LDS each): one computes while the other waits for its slab or stores a tile.
// gb_tap_gemm_lds_kernel (gbwd.hip) fed the MFMA from global memory one 16-row tile at a time: 102 TF/s over the large train step.
#include "tapgemm.h"
#include "layout.h"
#include <cstdlib>

#define GL_ROWS 128
#define GL_OUTS 192
#define GL_FB 1024                                 // bytes of a fragment block
#define GL_AFB (GL_OUTS / 16)                      // 12 weight blocks ...
#define GL_NFB (GL_AFB + GL_ROWS / 16)             // ... + 8 token blocks per stage
#define GL_PER_WAVE (GL_NFB / 4)                   // DMA instructions per wave and stage
#define GL_STAGE (GL_NFB * GL_FB)
#define GL_NST 3
#define GL_SLD 104                                 // row stride (elements) of a wave's store-staging tile: 16 rows x 96 outputs, 208-byte rows

// vmcnt(n): the wave's n most recent vector-memory operations may still be in flight
template <int N>
NBSS_DEV void dma_wait_but() {
#ifndef NBSS_EMU
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
#endif
}

struct GlTile {
    int rt, mc;
};

__global__ __launch_bounds__(256, 2) void gl_gemm_kernel(TapGemm p, int nrt, int nmc) {
    NBSS_LDS(smem);
    const int lane = lane_id(), l15 = lane & 15, g4 = lane >> 4, w = wave_id_u();
    const int wr = w & 1, wm = w >> 1;
    const bf16_t* X = reinterpret_cast<const bf16_t*>(p.X) + p.xcol;
    const bf16_t* W = reinterpret_cast<const bf16_t*>(p.W);
    const int nk = p.Kp / 32;
    // tiles of this workgroup: XCD x = blockIdx % 8 owns the row tiles rt % 8 == x (every output chunk of a row tile goes through the same L2)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int nu = nmc * ((nrt - xcd + 7) / 8);  // tiles of this XCD
    const int mine = nu > slot ? (nu - slot + slots - 1) / slots : 0;
    auto tile_at = [&](int i) {
        const int u = slot + i * slots;
        GlTile t = {(u / nmc) * 8 + xcd, u % nmc};
        return t;
    };
    // the copies of the next stage: the lane's five source rows are set up once per tile (the tile index -> rows divisions and the 64-bit row
    // addresses took ~50 instructions per copy when redone per stage: 250 per k-step against 24 MFMAs), a stage adds the K offset
    const bf16_t* src[GL_PER_WAVE];
    int i_tile = 0, i_k = 0, i_slot = 0;  // issue side: tile, K-slab, ring slot
    auto issue = [&]() {
        if (i_k == 0) {
            const GlTile t = tile_at(i_tile);
#pragma unroll
            for (int q = 0; q < GL_PER_WAVE; ++q) {
                const int f = w * GL_PER_WAVE + q;  // wave-uniform
                if (f < GL_AFB) {
                    int m = t.mc * GL_OUTS + f * 16 + l15;
                    m = m < p.Mp ? m : p.Mp - 1;
                    src[q] = W + (size_t)m * p.Kp + 8 * g4;
                } else {
                    long r = (long)t.rt * GL_ROWS + (f - GL_AFB) * 16 + l15;
                    r = r < p.rows ? r : p.rows - 1;
                    src[q] = X + (size_t)r * p.ldx + 8 * g4;
                }
            }
        }
        char* sb = smem + i_slot * GL_STAGE + w * GL_PER_WAVE * GL_FB;
#pragma unroll
        for (int q = 0; q < GL_PER_WAVE; ++q) dma16_to_lds(sb + q * GL_FB, src[q] + i_k * 32);
        if (++i_k == nk) {
            i_k = 0;
            ++i_tile;
        }
        i_slot = i_slot == GL_NST - 1 ? 0 : i_slot + 1;
    };
    const int S = mine * nk;
    if (S == 0) return;
    // bias of every output in LDS: a global load in the tile epilogue would wait (vmcnt is one in-order counter) for the stores of the previous row
    // tile ahead of it — 24 serialised store round trips per tile, 16 us per tile against 1 us of MFMA work in the first version
    float* bl = reinterpret_cast<float*>(smem + GL_NST * GL_STAGE);
    for (int i = threadIdx.x; i < p.Mp; i += 256) bl[i] = p.bias && i < p.Mg ? p.bias[i] : 0.f;
    issue();
    if (S > 1) issue();
    f32x4 acc[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[i][t] = F32X4_ZERO;
    const bf16_t* ext = reinterpret_cast<const bf16_t*>(p.Dact ? p.Dact : p.R);  // per-element operand of the epilogue (at most one of the two)
    const int lde = p.Dact ? p.ldy : p.ldr;
    int c_k = 0, c_tile = 0, c_slot = 0;  // consumer side
    GlTile tl = tile_at(0);
    for (int s = 0; s < S; ++s) {
        if (s + 1 < S) dma_wait_but<GL_PER_WAVE>();
        else dma_wait_all();
        lds_barrier();
        const bool last = c_k == nk - 1;
        const int mbase = tl.mc * GL_OUTS + wm * 96 + 4 * g4;
        // the tile's per-element operand is requested BEFORE the next slab's copies (in-order counter: its wait then leaves the copies in flight)
        u32x2 ex[4][6];
        if (last && ext) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                long row = (long)tl.rt * GL_ROWS + wr * 64 + t * 16 + l15;
                row = row < p.rows ? row : p.rows - 1;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    int m0 = mbase + 16 * i;
                    m0 = m0 < p.Mg ? m0 : 0;
                    ex[t][i] = *reinterpret_cast<const u32x2*>(ext + (size_t)row * lde + p.ycol + m0);
                }
            }
        }
        if (s + 2 < S) issue();
        const char* sb = smem + c_slot * GL_STAGE + lane * 16;
        c_slot = c_slot == GL_NST - 1 ? 0 : c_slot + 1;
        Frag<bf16_t> a[6], b[4];
#pragma unroll
        for (int i = 0; i < 6; ++i) frag_load(a[i], reinterpret_cast<const bf16_t*>(sb + (wm * 6 + i) * GL_FB));
#pragma unroll
        for (int t = 0; t < 4; ++t) frag_load(b[t], reinterpret_cast<const bf16_t*>(sb + (GL_AFB + wr * 4 + t) * GL_FB));
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][t] = mma(a[i], b[t], acc[i][t]);
        if (!last) {
            ++c_k;
            continue;
        }
        // epilogue of the tile (C layout: lane = outputs 16 i + 4 g4 + r of its row): every value first, then the stores back to back — a load or a
        // conditional block between two stores makes the compiler wait for the earlier store (vmcnt(0) per block)
        u32x2 outp[4][6];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16_t* yr = reinterpret_cast<const bf16_t*>(p.Y);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                int m0 = mbase + 16 * i;
                m0 = m0 < p.Mg ? m0 : 0;
                float o[4] = {acc[i][t][0], acc[i][t][1], acc[i][t][2], acc[i][t][3]};
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bl + m0);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] += bv[r];
                if (p.yact) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = silu_f(o[r]);
                }
                if (ext) {
                    const float ev[4] = {bf2f((bf16_t)(ex[t][i][0] & 0xFFFF)), bf2f((bf16_t)(ex[t][i][0] >> 16)), bf2f((bf16_t)(ex[t][i][1] & 0xFFFF)),
                                         bf2f((bf16_t)(ex[t][i][1] >> 16))};
                    if (p.Dact) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] *= dsilu_f(ev[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = ev[r] + round_to(o[r], yr);
                    }
                }
                outp[t][i] = (u32x2){pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                acc[i][t] = F32X4_ZERO;
            }
        }
        sched_fence();
        // stores: a row tile (16 rows x 96 outputs) goes through the wave's own 3 KB of LDS and leaves as 16-byte pieces, 12 adjacent lanes per
        // 192-byte row run (from the C layout directly: 8 bytes per lane in 32-byte runs, 24 store instructions per tile instead of 12)
        bf16_t* stg = reinterpret_cast<bf16_t*>(smem + GL_NST * GL_STAGE + ((p.Mp * 4 + 15) & ~15)) + w * 16 * GL_SLD;
        const int mw = tl.mc * GL_OUTS + wm * 96;  // first output of the wave's tile
        for (int pass = 0; pass < (p.Y2 ? 2 : 1); ++pass) {
            bf16_t* Yb = reinterpret_cast<bf16_t*>(pass ? p.Y2 : p.Y) + p.ycol;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    u32x2 v = outp[t][i];
                    if (pass) {  // SiLU of the STORED value (a and h of a forward step in one pass)
                        v[0] = pack2bf(silu_f(bf2f((bf16_t)(v[0] & 0xFFFF))), silu_f(bf2f((bf16_t)(v[0] >> 16))));
                        v[1] = pack2bf(silu_f(bf2f((bf16_t)(v[1] & 0xFFFF))), silu_f(bf2f((bf16_t)(v[1] >> 16))));
                    }
                    *reinterpret_cast<u32x2*>(stg + l15 * GL_SLD + 16 * i + 4 * g4) = v;
                }
                wave_lds_sync();
                const long row0 = (long)tl.rt * GL_ROWS + wr * 64 + t * 16;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int pc = lane + 64 * k, r = pc / 12, c8 = (pc % 12) * 8;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(stg + r * GL_SLD + c8);
                    if (row0 + r < p.rows && mw + c8 < p.Mg) *reinterpret_cast<u32x4*>(Yb + (size_t)(row0 + r) * p.ldy + mw + c8) = v;
                }
                wave_lds_sync();
            }
        }
        c_k = 0;
        tl = tile_at(++c_tile);
    }
}

static bool gl_disabled() {
    static const bool off = [] {
        const char* e = getenv("NBSS_GEMM_V1");
        return e && e[0] == '1';
    }();
    return off;
}
bool gl_gemm_takes(const TapGemm& p) {
    return !gl_disabled() && p.taps == 1 && p.groups == 1 && !p.xact && p.Kg == p.Kp && p.Kg >= 64 && p.Mg >= 64 && p.Mg % 8 == 0 && p.ldx % 8 == 0 && p.xcol % 8 == 0 &&
           p.ldy % 8 == 0 && p.ycol % 8 == 0 && (!p.R || p.ldr % 4 == 0) && !(p.R && p.Dact) && p.Mp <= 1024;
}
int gl_gemm_bf16(const TapGemm& p, hipStream_t st) {
    const int nrt = cdiv(p.rows, GL_ROWS), nmc = cdiv(p.Mg, GL_OUTS);
    const size_t lds = (size_t)GL_NST * GL_STAGE + (((size_t)p.Mp * sizeof(float) + 15) & ~(size_t)15) + (size_t)4 * 16 * GL_SLD * sizeof(bf16_t);
    int e = NBSS_SET_MAX_LDS(gl_gemm_kernel, lds);
    if (e) return e;
    const long ntiles = (long)nrt * nmc;
    int grid = 512;  // two workgroups on each of the 256 CUs; a multiple of 8 (the XCD mapping above)
    if (ntiles < grid) grid = (int)((ntiles + 7) / 8 * 8);
    NBSS_LAUNCH(gl_gemm_kernel, dim3(grid), dim3(256), lds, st, p, nrt, nmc);
    return NBSS_CHECK_LAUNCH();
}
