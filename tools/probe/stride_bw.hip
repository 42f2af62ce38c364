// Does the DRAM-page locality of the cross-band slabs matter?  One workgroup copies a slab of NCH chunks of CH bytes that lie STR bytes apart
// (the cross-band kernels' pattern in the [B,F,T,H] stream: 129 chunks of TT*192 B, 48 KB apart) — all loads of a thread issued before its
// stores, 16 bytes per lane.   hipcc --offload-arch=gfx950 -O3 -o tools/probe/stride_bw tools/probe/stride_bw.hip && tools/probe/stride_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(512) void slab_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, int nch, int ch16, long str16, long slab16) {
    // slab s: chunks c = 0..nch-1 at src + s * slab16 + c * str16, each ch16 16-byte pieces
    const long base = (long)blockIdx.x * slab16;
    const int per = nch * ch16;
    uint4 r[8];
    for (int i0 = threadIdx.x; i0 < per; i0 += 512 * 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * 512;
            if (i < per) r[k] = src[base + (long)(i / ch16) * str16 + i % ch16];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * 512;
            if (i < per) dst[base + (long)(i / ch16) * str16 + i % ch16] = r[k];
        }
    }
}

int main() {
    const long N = 32L * 129 * 251 * 192;  // one bf16 stream tensor at batch 32
    uint4 *a, *b;
    hipMalloc(&a, 2 * N);  // (generous: the synthetic slab bases of the cases below overrun the tensor itself)
    hipMalloc(&b, 2 * N);
    hipMemset(a, 1, N);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct Case { const char* name; int nch, ch; long str, slab; int grid; } cases[] = {
        // name, chunks per slab, chunk bytes, chunk stride bytes, slab offset bytes, slabs
        {"[B,F,T,H] 2-frame slab: 129 x 384 B, 48 KB apart", 129, 384, 251L * 192, 384, 32 * 125},
        {"[B,F,T,H] 8-frame slab: 129 x 1536 B, 48 KB apart", 129, 1536, 251L * 192, 1536, 32 * 31},
        {"tiled [B,T/8,F,8,H] 2-frame slab: 129 x 384 B, 1536 B apart", 129, 384, 1536, 129L * 1536, 32 * 31},
        {"tiled 8-frame slab: 198 KB contiguous", 129, 1536, 1536, 129L * 1536, 32 * 31},
        {"narrow-band sequence in the tiled layout: 32 x 1536 B, 198 KB apart", 32, 1536, 129L * 1536, 1536, 129 * 31},
        {"narrow-band sequence today: 48 KB contiguous", 32, 1536, 1536, 32L * 1536, 32 * 129},
    };
    for (auto& c : cases) {
        // (the 2-frame tiled case covers a quarter of each tile: it is launched 4x denser in t by shifting the slab base — same bytes per slab)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 5; ++it)
                hipLaunchKernelGGL(slab_copy, dim3(c.grid), dim3(512), 0, 0, a, b, c.nch, c.ch / 16, c.str / 16, c.slab / 16);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 2.0 * c.grid * c.nch * c.ch * 5;
        printf("%-72s %6.2f TB/s (read + write)  %7.1f us per launch\n", c.name, bytes / (ms * 1e-3) / 1e12, ms / 5 * 1e3);
    }
    return 0;
}
