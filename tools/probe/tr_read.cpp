// Probe the semantics of ds_read_b64_tr_b16 on gfx950: LDS holds lds[i] = i (16-bit); each lane supplies an element offset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const int* pat, short* out) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    int l = threadIdx.x;
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + pat[l]));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    int* dp; short* dout;
    hipMalloc(&dp, 64 * 4); hipMalloc(&dout, 256 * 2);
    const char* names[] = {"all-zero", "lane*4", "row-major16: (l&15)*16+(l>>4)*4", "(l>>4)*64+(l&15)*4", "lane*16 (row l of a 16-wide matrix)", "(l&15)*32 + (l>>4)*4 (stride 32 rows)"};
    for (int p = 0; p < 6; ++p) {
        std::vector<int> pat(64);
        for (int l = 0; l < 64; ++l) {
            switch (p) {
                case 0: pat[l] = 0; break;
                case 1: pat[l] = l * 4; break;
                case 2: pat[l] = (l & 15) * 16 + (l >> 4) * 4; break;
                case 3: pat[l] = (l >> 4) * 64 + (l & 15) * 4; break;
                case 4: pat[l] = l * 16; break;
                case 5: pat[l] = (l & 15) * 32 + (l >> 4) * 4; break;
            }
        }
        hipMemcpy(dp, pat.data(), 256, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dp, dout);
        std::vector<short> o(256);
        hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
        printf("pattern %d: %s\n", p, names[p]);
        for (int l = 0; l < 64; ++l) {
            printf("  l%02d addr %4d -> %4d %4d %4d %4d", l, pat[l], o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
            if (l % 2 == 1) printf("\n");
        }
    }
    return 0;
}
