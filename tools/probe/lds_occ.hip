// How much dynamic LDS may a 256-thread workgroup take and still share a CU with a second one?  (hipOccupancyMaxActiveBlocksPerMultiprocessor over sizes)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void k(float* p) {
    extern __shared__ float s[];
    s[threadIdx.x] = p[threadIdx.x];
    __syncthreads();
    p[threadIdx.x] = s[255 - threadIdx.x];
}
int main() {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int last = -1;
    for (int b = 70 * 1024; b <= 82 * 1024; b += 128) {
        int n = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, b);
        if (n != last) printf("lds %d B -> %d blocks per CU\n", b, n);
        last = n;
    }
    return 0;
}
