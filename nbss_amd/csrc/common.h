// Device-side building blocks shared by every nbss_amd kernel (gfx950 / CDNA4).
//
// Everything on the SpatialNet hot path is written against ONE fragment scheme for the
// 16x16x32 MFMA ("form 2": weights are the A operand, tokens are the N dimension):
//
//   A frag  lane l : A[m = l&15][k = kmap(l>>4, j)],  j = 0..7
//   B frag  lane l : B[k = kmap(l>>4, j)][n = l&15]
//   C/D     lane l : D[m = (l>>4)*4 + r][n = l&15],   r = 0..3
//
// with two K orders inside a 32-wide K block:
//   natural ("N"):  kmap(g, j) = 8g + j                      (16-byte contiguous per lane)
//   permuted ("P"): kmap(g, j) = j < 4 ? 4g + j : 16 + 4g + (j-4)
// The P order is exactly what two stacked C tiles (rows 0..15 and 16..31) look like when
// they are re-used as the B operand of the next product, so chains of per-token linear
// maps never leave registers.  Weights are pre-packed per lane (pack.hip) in whichever
// order the consuming kernel needs.
//
// The stream dtype T is bf16 (bf16-mixed training, v_mfma_f32_16x16x32_bf16) or float
// (fp32 training / parity runs, 8x v_mfma_f32_16x16x4_f32 per 32-wide K block: lane group
// g supplies k = kmap(g, j) at sub-step j, identical data placement).
#pragma once
#ifdef NBSS_EMU
#include "hipemu.h"
#define NBSS_LDS(name) char* name = hipemu::g_block->lds
#else
#include <hip/hip_runtime.h>
#define NBSS_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]
#endif
#include <stdint.h>

typedef unsigned short bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define NBSS_DEV __device__ __forceinline__

NBSS_DEV float bf2f(bf16_t h) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)h) << 16;
    return c.f;
}
#ifdef NBSS_EMU
NBSS_DEV bf16_t f2bf(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    u += 0x7FFFu + ((u >> 16) & 1u);  // round to nearest even
    return (bf16_t)(u >> 16);
}
NBSS_DEV uint32_t pack2bf(float a, float b) { return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16); }
#else
// gfx950: v_cvt_pk_bf16_f32 (round to nearest even), one instruction per pair instead of ~10 integer ops
NBSS_DEV bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
NBSS_DEV uint32_t pack2bf(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
#endif

// `cond ? f(x) : 0` with an expensive f compiles to an exec-mask BRANCH around f (one basic block per use: the scheduler cannot move
// loads or MFMAs across them; the narrow-band backward kernels had one branch per ~20 instructions).  Arguments of a call are
// evaluated unconditionally, so keep_if(cond, f(x)) is a v_cndmask.
NBSS_DEV float keep_if(bool c, float v) { return c ? v : 0.f; }

// 2^x as ONE v_exp_f32.  exp2f() expands to a denormal-safe sequence (v_cmp, 2 x v_cndmask, v_add, v_exp, v_ldexp: six VALU
// instructions per softmax element); the attention kernels only exponentiate differences <= 0 whose results may flush to zero.
NBSS_DEV float fast_exp2(float x) {
#ifdef NBSS_EMU
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);
#endif
}

NBSS_DEV float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// d/dx silu(x) = s + x*s*(1-s), s = sigmoid(x)
NBSS_DEV float dsilu_f(float x) {
    float s = 1.0f / (1.0f + __expf(-x));
    return s * (1.0f + x * (1.0f - s));
}

// SiLU and its derivative with the sigmoid written as ONE v_exp_f32 + ONE v_rcp_f32, and silu_pair() sharing it between value and derivative.
// (The library is built with -ffast-math, under which silu_f / dsilu_f above compile to the same v_exp + v_rcp mix — checked on the ISA of
// tchain.hip: 530 v_rcp, 528 v_exp either way — so these only make the instruction choice independent of the build flags.)
NBSS_DEV float fast_rcp(float x) {
#ifdef NBSS_EMU
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
NBSS_DEV float sigmoid_fast(float x) { return fast_rcp(1.0f + fast_exp2(-1.44269504f * x)); }
NBSS_DEV float silu_fast(float x) { return x * sigmoid_fast(x); }
NBSS_DEV float dsilu_fast(float x) {
    const float s = sigmoid_fast(x);
    return s * (1.0f + x * (1.0f - s));
}
NBSS_DEV void silu_pair(float x, float& y, float& dy) {
    const float s = sigmoid_fast(x);
    y = x * s;
    dy = s * (1.0f + x * (1.0f - s));
}

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic (lgkmcnt) but NOT for its global loads /
// stores (vmcnt) — __syncthreads() drains both, which exposed the full global-store latency at each of the ~80 barriers
// of the narrow-band backward kernels.  Use __syncthreads() wherever global memory written by another wave is read.
NBSS_DEV void lds_barrier() {
#ifdef NBSS_EMU
    __syncthreads();
#else
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), vmcnt/expcnt untouched
    __builtin_amdgcn_s_barrier();
#endif
}

// the same value, but unknown to the optimiser: expressions built from it are not recognised as common with earlier ones (a value recomputed
// in a late phase from registers that are live anyway, instead of 48 results kept alive — spilled — since an early phase)
NBSS_DEV float opaque(float x) {
#ifndef NBSS_EMU
    asm volatile("" : "+v"(x));
#endif
    return x;
}

// nothing is scheduled across this point (keeps independent register-hungry sections — e.g. the two strips of a LayerNorm — from being
// interleaved by the scheduler, which doubles their live temporaries)
NBSS_DEV void sched_fence() {
#ifndef NBSS_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// LDS written by one lane and read by another lane of the SAME wave: the hardware executes a wave's LDS instructions in order;
// the compiler must not reorder across this point and the emulator's fibers must meet here
NBSS_DEV void wave_lds_sync() {
#ifdef NBSS_EMU
    hipemu::wave_sync();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// Asynchronous global -> LDS copy of 16 bytes per lane without a register stop (global_load_lds_dwordx4): lane l's piece lands at
// `lds_wave_base + 16 l` (the destination is wave-uniform base + lane x 16, not a scatter; inactive lanes write nothing), `g` is per lane.
// In flight like any global load (vmcnt): dma_wait_all() — then a barrier before other waves read the image.
NBSS_DEV void dma16_to_lds(void* lds_wave_base, const void* g) {
#ifdef NBSS_EMU
    uint32_t* d = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lds_wave_base) + 16 * (threadIdx.x & 63));
    const uint32_t* s = reinterpret_cast<const uint32_t*>(g);
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}
NBSS_DEV void dma_wait_all() {
#ifndef NBSS_EMU
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); lgkmcnt / expcnt untouched
#endif
}

NBSS_DEV int lane_id() { return (int)(threadIdx.x & 63); }
NBSS_DEV int wave_id() { return (int)(threadIdx.x >> 6); }
// same value, but known to the compiler as wave-uniform: index arithmetic derived from it stays in SGPRs
NBSS_DEV int wave_id_u() {
#ifdef NBSS_EMU
    return (int)(threadIdx.x >> 6);
#else
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#endif
}

NBSS_DEV float wave_sum16(float v) {  // sum over the 4 lane groups (lanes l, l^16, l^32, l^48)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
NBSS_DEV float wave_max16(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}
// sum over the 16 lanes of a row (lanes sharing l>>4), result in every lane.  DPP row operations (quad swaps, half-mirror, mirror)
// run in the VALU; __shfl_xor compiles to ds_bpermute_b32, which queues behind the workgroup's real LDS traffic.
NBSS_DEV float row_sum16(float v) {
#ifdef NBSS_EMU
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
#else
#define NBSS_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    NBSS_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    NBSS_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    NBSS_DPP_ADD(0x141);  // row_half_mirror
    NBSS_DPP_ADD(0x140);  // row_mirror
#undef NBSS_DPP_ADD
    return v;
#endif
}
// One lane pairing inside a row of 16 lanes as a VALU move (DPP): STEP 0 = row_mirror (l <-> 15 - l), 1 = row_half_mirror (l <-> 7 - l inside each 8),
// 2 = quad_perm [2,3,0,1] (l ^ 2), 3 = quad_perm [1,0,3,2] (l ^ 1)
template <int STEP>
NBSS_DEV float row_pair(float v) {
#ifdef NBSS_EMU
    const int l = (int)(threadIdx.x & 63);
    const int p = STEP == 0 ? (l & ~15) | (15 - (l & 15)) : STEP == 1 ? (l & ~7) | (7 - (l & 7)) : STEP == 2 ? l ^ 2 : l ^ 1;
    return __shfl(v, p);
#else
    constexpr int ctrl = STEP == 0 ? 0x140 : STEP == 1 ? 0x141 : STEP == 2 ? 0x4E : 0xB1;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true));
#endif
}
// Transpose-reduce: SIXTEEN per-lane values summed over the 16 lanes of a row at once — each step pairs the lanes, one of a pair keeps the first half of
// the values and sends the second, its partner the other way round, so the value count halves per step: 8 + 4 + 2 + 1 adds (and two selects each)
// instead of 16 x 4 DPP adds; lane l15 returns the row total of v[l15] (a fixed summation order).
NBSS_DEV float row_reduce16x16(const float (&v)[16]) {
    const int l = (int)(threadIdx.x & 63);
    const bool c0 = l & 8, c1 = l & 4, c2 = l & 2, c3 = l & 1;
    float a[8], b[4], c[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (c0 ? v[8 + i] : v[i]) + row_pair<0>(c0 ? v[i] : v[8 + i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = (c1 ? a[4 + i] : a[i]) + row_pair<1>(c1 ? a[i] : a[4 + i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) c[i] = (c2 ? b[2 + i] : b[i]) + row_pair<2>(c2 ? b[i] : b[2 + i]);
    return (c3 ? c[1] : c[0]) + row_pair<3>(c3 ? c[0] : c[1]);
}
// Strided fold in a fixed order with N loads in flight: sum of p[x * stride] over x0 <= x < x1 — accumulator k takes x = x0 + k (mod N), the accumulators
// meet in a pairwise tree.  The folds of the partial parameter gradients are chains of dependent HBM / L2 round trips (64 rows per thread at four in
// flight: 16 round trips, 9 - 17 us per launch and ~100 launches per step — a quarter of a batch-2 step): N = 16 quarters the chain.
template <int N>
NBSS_DEV float fold_strided(const float* p, size_t stride, int x0, int x1) {
    float s[N], v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) s[k] = 0.f;
    for (int x = x0; x < x1; x += N) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = x + k < x1 ? p[(size_t)(x + k) * stride] : 0.f;
#pragma unroll
        for (int k = 0; k < N; ++k) s[k] += v[k];
    }
#pragma unroll
    for (int h = N / 2; h >= 1; h >>= 1) {
#pragma unroll
        for (int k = 0; k < h; ++k) s[k] += s[k + h];
    }
    return s[0];
}
NBSS_DEV float wave_sum64(float v) {
    v = row_sum16(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// ---- element access helpers, generic over the stream dtype ------------------------------
NBSS_DEV void load4(const float* p, float o[4]) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}
NBSS_DEV void load4(const bf16_t* p, float o[4]) {
    u32x2 v = *reinterpret_cast<const u32x2*>(p);
    o[0] = bf2f((bf16_t)(v[0] & 0xFFFF)); o[1] = bf2f((bf16_t)(v[0] >> 16));
    o[2] = bf2f((bf16_t)(v[1] & 0xFFFF)); o[3] = bf2f((bf16_t)(v[1] >> 16));
}
NBSS_DEV void load8(const float* p, float o[8]) { load4(p, o); load4(p + 4, o + 4); }
NBSS_DEV void load8(const bf16_t* p, float o[8]) {
    u32x4 v = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[2 * i] = bf2f((bf16_t)(v[i] & 0xFFFF));
        o[2 * i + 1] = bf2f((bf16_t)(v[i] >> 16));
    }
}
NBSS_DEV void store4(float* p, float a, float b, float c, float d) {
    f32x4 v = {a, b, c, d};
    *reinterpret_cast<f32x4*>(p) = v;
}
NBSS_DEV void store4(bf16_t* p, float a, float b, float c, float d) {
    u32x2 v = {pack2bf(a, b), pack2bf(c, d)};
    *reinterpret_cast<u32x2*>(p) = v;
}
// Streaming (non-temporal) variants for write-once tensors that nobody on this workgroup's XCD reads again soon (the wgrad
// operands): they must not evict the x / dy rows the same kernel re-reads for every conv group / head from L2.
NBSS_DEV void store4_nt(float* p, float a, float b, float c, float d) {
    f32x4 v = {a, b, c, d};
#ifdef NBSS_EMU
    *reinterpret_cast<f32x4*>(p) = v;
#else
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#endif
}
NBSS_DEV void store4_nt(bf16_t* p, float a, float b, float c, float d) {
    u32x2 v = {pack2bf(a, b), pack2bf(c, d)};
#ifdef NBSS_EMU
    *reinterpret_cast<u32x2*>(p) = v;
#else
    __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p));
#endif
}
NBSS_DEV void store16_nt(bf16_t* p, const u32x4& v) {  // one 16-byte streaming store
#if defined(NBSS_EMU) || defined(NBSS_NO_NT)  // (NBSS_NO_NT: A/B flavour, plain stores)
    *reinterpret_cast<u32x4*>(p) = v;
#else
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
#endif
}
NBSS_DEV void store1(float* p, float a) { *p = a; }
NBSS_DEV void store1(bf16_t* p, float a) { *p = f2bf(a); }
NBSS_DEV float load1(const float* p) { return *p; }
NBSS_DEV float load1(const bf16_t* p) { return bf2f(*p); }
// rounding a value to the stream precision (identity for float)
NBSS_DEV float round_to(float v, const float*) { return v; }
NBSS_DEV float round_to(float v, const bf16_t*) { return bf2f(f2bf(v)); }

NBSS_DEV void store8(float* p, const float (&o)[8]) {
    store4(p, o[0], o[1], o[2], o[3]);
    store4(p + 4, o[4], o[5], o[6], o[7]);
}
NBSS_DEV void store8(bf16_t* p, const float (&o)[8]) {  // one 16-byte store
    u32x4 v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
    *reinterpret_cast<u32x4*>(p) = v;
}

// ---- MFMA fragments ---------------------------------------------------------------------
template <class T> struct Frag;
template <> struct Frag<bf16_t> { s16x8 v; };
template <> struct Frag<float> { float v[8]; };

NBSS_DEV void frag_zero(Frag<bf16_t>& f) { f.v = (s16x8){0, 0, 0, 0, 0, 0, 0, 0}; }
NBSS_DEV void frag_zero(Frag<float>& f) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = 0.f;
}
// 8 contiguous elements (natural K order)
NBSS_DEV void frag_load(Frag<bf16_t>& f, const bf16_t* p) { f.v = *reinterpret_cast<const s16x8*>(p); }
NBSS_DEV void frag_load(Frag<float>& f, const float* p) { load8(p, f.v); }
// two 4-element pieces (permuted K order, or conv taps)
NBSS_DEV void frag_load_lo(Frag<bf16_t>& f, const bf16_t* p) {
    s16x4 a = *reinterpret_cast<const s16x4*>(p);
    f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
}
NBSS_DEV void frag_load_hi(Frag<bf16_t>& f, const bf16_t* p) {
    s16x4 a = *reinterpret_cast<const s16x4*>(p);
    f.v[4] = a[0]; f.v[5] = a[1]; f.v[6] = a[2]; f.v[7] = a[3];
}
NBSS_DEV void frag_load_lo(Frag<float>& f, const float* p) { load4(p, f.v); }
NBSS_DEV void frag_load_hi(Frag<float>& f, const float* p) { load4(p, f.v + 4); }
NBSS_DEV void frag_zero_lo(Frag<bf16_t>& f) { f.v[0] = 0; f.v[1] = 0; f.v[2] = 0; f.v[3] = 0; }
NBSS_DEV void frag_zero_hi(Frag<bf16_t>& f) { f.v[4] = 0; f.v[5] = 0; f.v[6] = 0; f.v[7] = 0; }
NBSS_DEV void frag_zero_lo(Frag<float>& f) { f.v[0] = 0; f.v[1] = 0; f.v[2] = 0; f.v[3] = 0; }
NBSS_DEV void frag_zero_hi(Frag<float>& f) { f.v[4] = 0; f.v[5] = 0; f.v[6] = 0; f.v[7] = 0; }
NBSS_DEV void frag_set(Frag<bf16_t>& f, int j, float x) { f.v[j] = (short)f2bf(x); }
NBSS_DEV void frag_set(Frag<float>& f, int j, float x) { f.v[j] = x; }
NBSS_DEV float frag_get(const Frag<bf16_t>& f, int j) { return bf2f((bf16_t)f.v[j]); }
NBSS_DEV float frag_get(const Frag<float>& f, int j) { return f.v[j]; }
// fragment from 8 floats
template <class T>
NBSS_DEV void frag_from(Frag<T>& f, const float x[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) frag_set(f, j, x[j]);
}
// two stacked C tiles -> permuted-order B (or A) fragment
template <class T>
NBSS_DEV void frag_from_c2(Frag<T>& f, const f32x4& lo, const f32x4& hi) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        frag_set(f, j, lo[j]);
        frag_set(f, 4 + j, hi[j]);
    }
}

NBSS_DEV f32x4 mma(const Frag<bf16_t>& a, const Frag<bf16_t>& b, f32x4 c) {
#ifdef NBSS_EMU
    return hipemu::mfma_16x16x32_bf16(a.v, b.v, c);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
#endif
}
NBSS_DEV f32x4 mma(const Frag<float>& a, const Frag<float>& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#ifdef NBSS_EMU
        c = hipemu::mfma_16x16x4_f32(a.v[j], b.v[j], c);
#else
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
#endif
    }
    return c;
}

// Transposing LDS read (bf16 only): the 16 lanes of a group read a 4-row x 16-column block of a row-major image — lane p
// passes the address of row (p>>2), columns 4(p&3)..+3 — and lane l receives column l of the 4 rows.  Two of these turn a
// row-major [token][channel] tile into an MFMA operand whose K dimension is the token axis (permuted K order).
NBSS_DEV u32x2 lds_tr4_b16(const bf16_t* p) {
#ifdef NBSS_EMU
    uint64_t r = hipemu::ds_read_tr16_b64(p);
    u32x2 v = {(uint32_t)(r & 0xFFFFFFFFu), (uint32_t)(r >> 32)};
    return v;
#else
    typedef short v4s_t __attribute__((ext_vector_type(4)));
    v4s_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)p);
    return __builtin_bit_cast(u32x2, v);
#endif
}
// fragment (K = 32 tokens, permuted order) of 16 channels from a row-major LDS image; `p` = &img[4(l>>4) + ((l&15)>>2)][c0 + 4(l&3)]
NBSS_DEV void frag_load_tr(Frag<bf16_t>& f, const bf16_t* p, int row_stride) {
    const u32x2 lo = lds_tr4_b16(p), hi = lds_tr4_b16(p + 16 * (size_t)row_stride);
    u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    f.v = __builtin_bit_cast(s16x8, v);
}

#define F32X4_ZERO ((f32x4){0.f, 0.f, 0.f, 0.f})

// packed weight fragments live in global memory as [tile][kstep][lane][8] of T
template <class T>
NBSS_DEV void wfrag_load(Frag<T>& f, const T* base, int tile, int ksteps, int ks) {
    frag_load(f, base + ((size_t)(tile * ksteps + ks) * 64 + lane_id()) * 8);
}
