// Opt-in per-kernel timing with HIP events recorded on the launch stream (diagnostics for bench.py's
// roofline line; off by default, zero cost when off).
#pragma once
#include "launch.h"

enum ProfKernel {
    PK_ENC_F = 0, PK_FCONV_F, PK_FULL_F, PK_MHSA_F, PK_TCF_F, PK_DEC_F,
    PK_DEC_B, PK_TCF_B, PK_MHSA_B, PK_FCONV_B, PK_FULL_B, PK_WGRAD,
    PK_STFT, PK_ISTFT, PK_ISTFT_B, PK_LOSS, PK_ADAM, PK_PACK, PK_COUNT
};

void prof_begin(int id, hipStream_t st);
void prof_end(int id, hipStream_t st);

struct ProfScope {
    int id;
    hipStream_t st;
    ProfScope(int i, hipStream_t s) : id(i), st(s) { prof_begin(id, st); }
    ~ProfScope() { prof_end(id, st); }
};

// ---- in-kernel phase timing (diagnostic build only: `python -m nbss_amd.build phase` -> lib/libnbss_hip_phase.so) ----
// Every wave accumulates the shader-clock time it spends between PHASE marks in an LDS slot and adds the totals to a
// per-translation-unit device array at the end; tools/phase_prof.py prints each phase's share of the wave time.
#ifdef NBSS_PHASE_PROF
#define PH_N 32
static __device__ unsigned long long nbss_phase_acc[PH_N];
struct PhaseTimer {
    unsigned* slot;
    unsigned long long t;
    __device__ PhaseTimer(void* lds) {
        slot = reinterpret_cast<unsigned*>(lds) + (threadIdx.x >> 6) * PH_N;
        if ((threadIdx.x & 63) < PH_N) slot[threadIdx.x & 63] = 0u;
        t = clock64();
    }
    __device__ void mark(int i) {
        const unsigned long long n = clock64();
        if ((threadIdx.x & 63) == 0) slot[i] += (unsigned)(n - t);
        t = clock64();
    }
    __device__ void flush() {
        if ((threadIdx.x & 63) < PH_N) atomicAdd(&nbss_phase_acc[threadIdx.x & 63], (unsigned long long)slot[threadIdx.x & 63]);
    }
};
#define PHASE_LDS_BYTES (16 * PH_N * sizeof(unsigned))
#define PHASE_BEGIN(lds) PhaseTimer _pt(lds)
#define PHASE(i) _pt.mark(i)
#define PHASE_END() _pt.flush()
#define PHASE_READER(name)                                                                              \
    extern "C" int name(unsigned long long* out) {                                                      \
        unsigned long long z[PH_N] = {0};                                                               \
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nbss_phase_acc), sizeof(z)) != hipSuccess) return -1;   \
        return hipMemcpyToSymbol(HIP_SYMBOL(nbss_phase_acc), z, sizeof(z)) == hipSuccess ? PH_N : -1;   \
    }
#else
#define PHASE_LDS_BYTES 0
#define PHASE_BEGIN(lds)
#define PHASE(i)
#define PHASE_END()
#define PHASE_READER(name)
#endif
