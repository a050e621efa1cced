// Diagnostic: which hardware ids distinguish the two workgroups that share a CU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned* out) {
    __shared__ float pad[18000];  // ~72 KB: two blocks per CU, like the GEMM
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x % 64 == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID, 32 bits
        unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20); // HW_REG_XCC_ID
        out[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = hw;
        out[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = xcc;
    }
    // keep the block alive a little so that all 512 are co-resident
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(127);
    if (pad[(threadIdx.x * 7) % 256] < 0) out[0] = 0;
}
int main() {
    unsigned* d; hipMalloc(&d, 512 * 4 * 2 * 4);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d);
    std::vector<unsigned> h(512 * 4 * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;  // key = xcc|se|sh|cu -> blocks
    for (int b = 0; b < 512; ++b) {
        unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 15;
        unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 15);
        cu[key].push_back(b);
    }
    printf("distinct CUs: %zu\n", cu.size());
    int shown = 0;
    for (auto& kv : cu) {
        if (shown++ >= 12) break;
        printf("cu %05x:", kv.first);
        for (int b : kv.second) {
            unsigned hw = h[b * 8];
            printf("  blk %3d wave_id %u simd %u tg_id %u |w1 wave_id %u simd %u", b, hw & 15, (hw >> 4) & 3, (hw >> 16) & 15,
                   h[b * 8 + 2] & 15, (h[b * 8 + 2] >> 4) & 3);
        }
        printf("\n");
    }
    return 0;
}
