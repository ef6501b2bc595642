// Micro-benchmark: issue cost per wave64 instruction of the ops in the z kernels' ray bookkeeping on gfx950:
// v_add_f64, v_cvt_f64_f32, v_cvt_f32_f64, v_pk_add_f32, v_pk_mul_f32, v_rcp_f32, v_med3_f32, v_cndmask_b32.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_bench2 valu_bench2.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((vector_size(8)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  float f[8];
  double d[8];
  v2f p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f[i] = (float)(threadIdx.x + i) * 0.001f + 0.5f;
    d[i] = (double)f[i];
    p[i] = v2f{f[i], f[i] + 1.f};
  }
  const v2f aa = v2f{a, b};
  const double da = (double)a;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da));
      else if (MODE == 1) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
      else if (MODE == 2) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
      else if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(aa));
      else if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(aa));
      else if (MODE == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
      else if (MODE == 6) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(a), "v"(b));
      else if (MODE == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(a));
      else if (MODE == 8) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(da));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += f[i] + (float)d[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* d, int wg_per_cu) {
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
  const double instr_per_simd = (double)wg_per_cu * iters * 8;   // one wave of each WG per SIMD
  printf("%-16s wg/cu=%d  %.3f ms  %.2f cyc/instr/SIMD@2.4GHz\n", name, wg_per_cu, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int w : {1, 4}) {
    run<7>("v_add_f32", d, w);
    run<3>("v_pk_add_f32", d, w);
    run<4>("v_pk_mul_f32", d, w);
    run<0>("v_add_f64", d, w);
    run<8>("v_fma_f64", d, w);
    run<1>("v_cvt_f64_f32", d, w);
    run<2>("v_cvt_f32_f64", d, w);
    run<5>("v_rcp_f32", d, w);
    run<6>("v_med3_f32", d, w);
  }
  return 0;
}
