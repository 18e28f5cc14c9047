// Checks, on the device, the lane exchanges the linear layer's last step relies on: every lane must receive the value
// of lane ^ K (K = 1, 2, 4, 8 by DPP; K = 16 through v_permlane16_swap, which trades the odd 16-lane rows of one
// register for the even rows of another).  Build: hipcc --offload-arch=gfx950 -O2 tools/lane_exchange_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ int dpp_xor1(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ int dpp_xor2(int v) { return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true); }
__device__ __forceinline__ int dpp_xor4(int v) {
    int t = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);      // row_shl:4 into banks 0 and 2
    return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);       // row_shr:4 into banks 1 and 3
}
__device__ __forceinline__ int dpp_xor8(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false); }   // row_ror:8
__global__ void k(int *o) {
    const int l = threadIdx.x;
    o[l] = dpp_xor1(l); o[64 + l] = dpp_xor2(l); o[128 + l] = dpp_xor4(l); o[192 + l] = dpp_xor8(l);
    auto r = __builtin_amdgcn_permlane16_swap((unsigned)(1000 + l), (unsigned)(2000 + l), false, false);
    o[256 + l] = (int)r[0]; o[320 + l] = (int)r[1];
}
int main() {
    int *d, h[384];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 2;
    k<<<1, 64>>>(d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        bad += h[l] != (l ^ 1); bad += h[64 + l] != (l ^ 2); bad += h[128 + l] != (l ^ 4); bad += h[192 + l] != (l ^ 8);
        // first register keeps its even rows and receives the second register's even rows in its odd rows;
        // second register receives the first one's odd rows in its even rows and keeps its odd rows
        const int odd = (l >> 4) & 1;
        bad += h[256 + l] != (odd ? 2000 + (l ^ 16) : 1000 + l);
        bad += h[320 + l] != (odd ? 2000 + l : 1000 + (l ^ 16));
    }
    printf("lane exchanges: %d mismatches\n", bad);
    if (bad) { for (int r = 0; r < 6; ++r) { for (int l = 0; l < 64; ++l) printf("%d ", h[64 * r + l]); printf("\n"); } }
    return bad != 0;
}
