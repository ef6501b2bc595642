// Prints the lane mapping of the DPP controls the x-blur relies on (gfx950, wave64).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, bool BC>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  const int src = 100 + lane;
  out[lane] = __builtin_amdgcn_update_dpp(-1, src, CTRL, 0xf, 0xf, BC);
}
template <int CTRL, bool BC>
void run(const char* name, int* d) {
  hipLaunchKernelGGL((k<CTRL, BC>), dim3(1), dim3(64), 0, 0, d);
  int h[64];
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-22s:", name);
  for (int i = 0; i < 34; ++i) printf(" %d", h[i]);
  printf("\n");
}
int main() {
  int* d; (void)hipMalloc(&d, 256);
  run<0x111, true>("row_shr:1 bc=1", d);
  run<0x113, true>("row_shr:3 bc=1", d);
  run<0x113, false>("row_shr:3 bc=0", d);
  run<0x101, true>("row_shl:1 bc=1", d);
  run<0x103, false>("row_shl:3 bc=0", d);
  run<0x121, true>("row_ror:1", d);
  run<0x138, true>("wave_shr:1 bc=1", d);
  run<0x130, true>("wave_shl:1 bc=1", d);
  return 0;
}
