// Micro-benchmarks of the primitives the narrow-band kernels are made of (gfx950): cycles per wave-instruction at a given
// occupancy.  Build: hipcc --offload-arch=gfx950 -O3 -ffast-math tools/probe/ubench.hip -o gpurun_out/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(1024) void k_valu(float* out, long long* cyc, int iters) {
    float x[12];
    for (int i = 0; i < 12; ++i) x[i] = 0.01f * (threadIdx.x + i);
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if (MODE == 0) x[i] = x[i] / (1.0f + __expf(-x[i])) + 0.5f;          // silu
            if (MODE == 1) x[i] = __builtin_amdgcn_exp2f(x[i] * -0.3f);            // mul + exp
            if (MODE == 2) x[i] = __builtin_amdgcn_rcpf(x[i] + 1.5f);              // add + rcp
            if (MODE == 3) x[i] = x[i] * 0.99f + 0.5f;                             // fma
            if (MODE == 4) { float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x[i] * -1.4427f)); x[i] = s * (1.0f + x[i] * (1.0f - s)) + 0.3f; }  // dsilu
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 12; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int CHAINS, int LEN>
__global__ __launch_bounds__(1024) void k_mfma(float* out, long long* cyc, int iters) {
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3F80 + threadIdx.x % 7); b[i] = (short)(0x3C00 + i); }
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < LEN; ++k)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <class K>
void run(const char* name, K kern, int threads, int iters, double per_iter_ops) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256 * 16);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; int nw = 256 * threads / 64;
    for (int i = 0; i < nw; ++i) avg += h[i];
    avg /= nw;
    printf("%-28s waves/CU %2d: %8.0f ticks/wave, %6.2f ticks per op-instance per wave, wall %.1f us (%.2f GHz if tick=cycle)\n", name, threads / 64, avg,
           avg / (iters * per_iter_ops), ms * 1e3, avg / (ms * 1e3) / 1e3);
    hipFree(out); hipFree(cyc);
}

int main() {
    const int it = 2000;
    for (int thr : {256, 512, 1024}) {
        run("silu (12 indep)", k_valu<0>, thr, it, 12);
        run("mul+exp2", k_valu<1>, thr, it, 12);
        run("add+rcp", k_valu<2>, thr, it, 12);
        run("fma", k_valu<3>, thr, it, 12);
        run("sigmoid+dsilu", k_valu<4>, thr, it, 12);
    }
    for (int thr : {256, 512, 1024}) {
        run("mfma32 1 chain x5", k_mfma<1, 5>, thr, 500, 5);
        run("mfma32 2 chains x5", k_mfma<2, 5>, thr, 500, 10);
        run("mfma32 4 chains x5", k_mfma<4, 5>, thr, 500, 20);
    }
    return 0;
}
