// hipemu — a tiny fiber-based CPU executor for the HIP kernels in nbss_amd/csrc.
//
// TEST INFRASTRUCTURE ONLY.  There is no GPU in the build container, so the kernel
// sources are additionally compiled for the host (clang++ -x c++ -DNBSS_EMU) and run
// here, one fiber per GPU thread, 64-lane waves, with faithful emulation of
//   * __syncthreads()            (block barrier)
//   * wave collectives           (__shfl*, MFMA 16x16x32 bf16 / 16x16x4 f32 fragment layouts)
//   * dynamic LDS                (one private buffer per block)
//   * atomicAdd on global memory
// This checks index math, LDS hazards (fiber order can be reversed / randomised with
// HIPEMU_ORDER=rev|rand to expose missing barriers) and the MFMA fragment bookkeeping.
// Nothing in the product (nbss_amd/, models/, bench.py) may load the emulator library.
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace hipemu {

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct Wave {
    int arrived = 0;
    int alive = 0;
    unsigned long gen = 0;
    alignas(16) unsigned char scratch[64][64];
};

struct Fiber {
    void* sp = nullptr;  // saved stack pointer (hand-rolled x86-64 context switch: no signal-mask syscalls)
    char* stack = nullptr;
    dim3 tid;
    int linear = 0, lane = 0, wave = 0;
    bool done = false;
    int wait = 0;  // 0 runnable, 1 block barrier, 2 wave collective
    unsigned long wait_gen = 0;
};

struct Block {
    dim3 bid, bdim, gdim;
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    int bar_arrived = 0;
    unsigned long bar_gen = 0;
    int alive = 0;
    char* lds = nullptr;
    size_t lds_bytes = 0;
    void* sched_sp = nullptr;
    void (*entry)(void*) = nullptr;
    void* entry_arg = nullptr;
};

extern thread_local Block* g_block;
extern thread_local Fiber* g_fiber;

void yield_to_sched();
void block_barrier();
void wave_sync();
void run_grid(dim3 grid, dim3 block, size_t lds_bytes, void (*entry)(void*), void* arg);

inline int lane_id() { return g_fiber->lane; }
inline Wave& cur_wave() { return g_block->waves[g_fiber->wave]; }

// ---- wave collectives ------------------------------------------------------------------
template <class T>
inline T shfl_idx(T v, int src) {
    static_assert(sizeof(T) <= 64, "");
    Wave& w = cur_wave();
    std::memcpy(w.scratch[g_fiber->lane], &v, sizeof(T));
    wave_sync();
    T r;
    std::memcpy(&r, w.scratch[src & 63], sizeof(T));
    wave_sync();
    return r;
}
template <class T>
inline T shfl_xor(T v, int mask) { return shfl_idx(v, g_fiber->lane ^ mask); }
template <class T>
inline T shfl_down(T v, int d) {
    int s = g_fiber->lane + d;
    return shfl_idx(v, s > 63 ? g_fiber->lane : s);
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

inline float bf16_bits_to_f32(unsigned short h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// v_mfma_f32_16x16x32_bf16:  A lane l holds A[l&15][(l>>4)*8 + j], B lane l holds
// B[(l>>4)*8 + j][l&15], C/D lane l reg r holds D[(l>>4)*4 + r][l&15].
inline f32x4_t mfma_16x16x32_bf16(s16x8_t a, s16x8_t b, f32x4_t c) {
    Wave& w = cur_wave();
    int l = g_fiber->lane;
    std::memcpy(w.scratch[l], &a, 16);
    std::memcpy(w.scratch[l] + 16, &b, 16);
    wave_sync();
    int col = l & 15, rg = l >> 4;
    f32x4_t d = c;
    for (int r = 0; r < 4; ++r) {
        int row = rg * 4 + r;
        float acc = d[r];
        for (int g = 0; g < 4; ++g) {
            unsigned short av[8], bv[8];
            std::memcpy(av, w.scratch[g * 16 + row], 16);
            std::memcpy(bv, w.scratch[g * 16 + col] + 16, 16);
            for (int j = 0; j < 8; ++j) acc = std::fmaf(bf16_bits_to_f32(av[j]), bf16_bits_to_f32(bv[j]), acc);
        }
        d[r] = acc;
    }
    wave_sync();
    return d;
}

// v_mfma_f32_16x16x4_f32: A lane l holds A[l&15][l>>4], B lane l holds B[l>>4][l&15].
inline f32x4_t mfma_16x16x4_f32(float a, float b, f32x4_t c) {
    Wave& w = cur_wave();
    int l = g_fiber->lane;
    std::memcpy(w.scratch[l], &a, 4);
    std::memcpy(w.scratch[l] + 4, &b, 4);
    wave_sync();
    int col = l & 15, rg = l >> 4;
    f32x4_t d = c;
    for (int r = 0; r < 4; ++r) {
        int row = rg * 4 + r;
        float acc = d[r];
        for (int g = 0; g < 4; ++g) {
            float av, bv;
            std::memcpy(&av, w.scratch[g * 16 + row], 4);
            std::memcpy(&bv, w.scratch[g * 16 + col] + 4, 4);
            acc = std::fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    wave_sync();
    return d;
}


// v_mfma_f32_32x32x16_bf16:  A lane l holds A[l&31][(l>>5)*8 + j], B lane l holds B[(l>>5)*8 + j][l&31],
// C/D lane l reg r holds D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
typedef float f32x16_t __attribute__((ext_vector_type(16)));
inline f32x16_t mfma_32x32x16_bf16(s16x8_t a, s16x8_t b, f32x16_t c) {
    Wave& w = cur_wave();
    int l = g_fiber->lane;
    std::memcpy(w.scratch[l], &a, 16);
    std::memcpy(w.scratch[l] + 16, &b, 16);
    wave_sync();
    int col = l & 31, hf = l >> 5;
    f32x16_t d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
        float acc = d[r];
        for (int g = 0; g < 2; ++g) {
            unsigned short av[8], bv[8];
            std::memcpy(av, w.scratch[g * 32 + row], 16);
            std::memcpy(bv, w.scratch[g * 32 + col] + 16, 16);
            for (int j = 0; j < 8; ++j) acc = std::fmaf(bf16_bits_to_f32(av[j]), bf16_bits_to_f32(bv[j]), acc);
        }
        d[r] = acc;
    }
    wave_sync();
    return d;
}

// ds_read_b64_tr_b16 (gfx950), semantics pinned on hardware by tools/probe/tr_read.cpp: every lane loads 4 x b16 from ITS
// address; within each 16-lane group  out[l][j] = in[lane 4j + (l>>2)][l & 3]  — a 4-row x 16-column block whose lane p holds
// row p>>2, columns 4(p&3)..+3 comes back as lane = column, elements = the 4 rows.
inline uint64_t ds_read_tr16_b64(const void* p) {
    Wave& w = cur_wave();
    int l = g_fiber->lane;
    std::memcpy(w.scratch[l], p, 8);
    wave_sync();
    int g = l >> 4, li = l & 15;
    unsigned short o[4];
    for (int j = 0; j < 4; ++j) {
        unsigned short in4[4];
        std::memcpy(in4, w.scratch[g * 16 + 4 * j + (li >> 2)], 8);
        o[j] = in4[li & 3];
    }
    wave_sync();
    uint64_t r;
    std::memcpy(&r, o, 8);
    return r;
}

inline float atomic_add_f32(float* p, float v) {
    auto* ap = reinterpret_cast<std::atomic<uint32_t>*>(p);
    uint32_t old = ap->load(std::memory_order_relaxed);
    for (;;) {
        float f;
        std::memcpy(&f, &old, 4);
        float nf = f + v;
        uint32_t nu;
        std::memcpy(&nu, &nf, 4);
        if (ap->compare_exchange_weak(old, nu)) return f;
    }
}

// ---- launch ---------------------------------------------------------------------------
template <class F>
struct Thunk {
    F f;
    static void call(void* p) { static_cast<Thunk*>(p)->f(); }
};

template <class F>
inline void launch(dim3 grid, dim3 block, size_t lds_bytes, F&& body) {
    Thunk<F> t{static_cast<F&&>(body)};
    run_grid(grid, block, lds_bytes, &Thunk<F>::call, &t);
}

}  // namespace hipemu

// ---- the HIP surface the kernels use ----------------------------------------------------
using hipemu::dim3;
typedef void* hipStream_t;
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define threadIdx (hipemu::g_fiber->tid)
#define blockIdx (hipemu::g_block->bid)
#define blockDim (hipemu::g_block->bdim)
#define gridDim (hipemu::g_block->gdim)
inline void __syncthreads() { hipemu::block_barrier(); }
template <class T>
inline T __shfl_xor(T v, int m) { return hipemu::shfl_xor(v, m); }
template <class T>
inline T __shfl(T v, int s) { return hipemu::shfl_idx(v, s); }
template <class T>
inline T __shfl_down(T v, int d) { return hipemu::shfl_down(v, d); }
inline float atomicAdd(float* p, float v) { return hipemu::atomic_add_f32(p, v); }
inline float __expf(float x) { return std::exp(x); }
inline float __logf(float x) { return std::log(x); }
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
