// Micro-benchmark: VALU issue cost of v_fma_f32 vs v_pk_fma_f32 on gfx950 (wave64), and of ds_bpermute.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_bench valu_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((vector_size(8)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  float acc[16];
  v2f accp[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (float)(threadIdx.x + i);
#pragma unroll
  for (int i = 0; i < 8; ++i) accp[i] = v2f{(float)threadIdx.x, (float)i};
  const v2f aa = v2f{a, a}, bb = v2f{b, b};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(accp[i]) : "v"(aa), "v"(bb));
    } else if (MODE == 2) {  // 16 scalar fma + 16 packed (8 pk)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = acc[i] * a + b;
    } else if (MODE == 3) {  // ds_bpermute chain-free
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __shfl(acc[i], (threadIdx.x + 1) & 63, 64);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += accp[i][0] + accp[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* d, int wg_per_cu, int flop_per_iter_per_lane) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 100, 1.0001f, 0.5f);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double lanes = (double)grid * 256;
  const double tf = lanes * iters * flop_per_iter_per_lane / (ms * 1e-3) / 1e12;
  // cycles per wave-instruction per SIMD at 2.4 GHz: waves per SIMD = wg_per_cu (4 waves per WG, 4 SIMDs)
  const double instr_per_simd = (double)wg_per_cu * iters * (MODE == 1 ? 8 : 16);
  printf("%-14s wg/cu=%d  %.3f ms  %.1f TFLOP/s  %.2f cyc/instr/SIMD@2.4GHz\n", name, wg_per_cu, ms, tf,
         ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", d, w, 32);
    run<1>("v_pk_fma_f32", d, w, 32);
    run<3>("ds_bpermute", d, w, 0);
  }
  return 0;
}
