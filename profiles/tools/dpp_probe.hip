#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) { int x = threadIdx.x * 3; o[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x130, 0xf, 0xf, false); }
int main() { int* d; hipMalloc(&d, 256); k<<<1, 64>>>(d); int h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); printf("wave_shl:1 lane0=%d lane1=%d lane62=%d lane63=%d (expect 3 6 189 -1)\n", h[0], h[1], h[62], h[63]); return 0; }
