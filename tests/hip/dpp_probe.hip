// Hardware probe: semantics of the gfx950 cross-lane moves the kernels rely on.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(int *out) {
  const int lane = threadIdx.x;
  const int v = 100 + lane;
  out[lane] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);        // wave_shr:1
  out[64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
  out[128 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  out[192 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x101, 0xf, 0xf, false);  // row_shl:1
}
int main() {
  int *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int shr = l == 0 ? -1 : 100 + l - 1, shl = l == 63 ? -1 : 100 + l + 1;
    if (h[l] != shr) { ++bad; printf("wave_shr lane %d got %d want %d\n", l, h[l], shr); }
    if (h[64 + l] != shl) { ++bad; printf("wave_shl lane %d got %d want %d\n", l, h[64 + l], shl); }
  }
  printf("row_shr:1 lanes 15..17: %d %d %d ; row_shl:1 lanes 14..16: %d %d %d\n", h[128+15], h[128+16], h[128+17], h[192+14], h[192+15], h[192+16]);
  printf(bad ? "DPP_PROBE_FAIL\n" : "DPP_PROBE_OK\n");
  return bad != 0;
}
