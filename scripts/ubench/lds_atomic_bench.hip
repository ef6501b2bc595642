// Throughput of LDS atomics on gfx950: ds_add_f32 vs ds_add_u32 vs ds_add_u64, conflict-free and random addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE, int PATTERN>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  __shared__ unsigned long long tile64[4096];
  float* tf = reinterpret_cast<float*>(tile64);
  unsigned* tu = reinterpret_cast<unsigned*>(tile64);
  for (int i = threadIdx.x; i < 4096; i += 256) tile64[i] = 0;
  __syncthreads();
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int it = 0; it < iters; ++it) {
    unsigned a;
    if (PATTERN == 0) a = (threadIdx.x + it * 256) & 4095;                 // distinct, conflict-free
    else { x = x * 1664525u + 1013904223u; a = (x >> 10) & 4095; }        // random cell
    if (MODE == 0) atomicAdd(&tf[a], 1.0f);
    else if (MODE == 1) atomicAdd(&tu[a], 1u);
    else if (MODE == 2) atomicAdd(&tile64[a], 1ull);
    else if (MODE == 3) tf[a] = tf[a] + 1.0f;                               // plain read-modify-write (racy), for scale
  }
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < 4096; i += 256) s += (float)tile64[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int PATTERN>
void run(const char* name, float* d) {
  const int iters = 2000, grid = 256 * 4;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(grid), dim3(256), 0, 0, d, 10);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(grid), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)grid * 256 * iters;
  printf("%-28s %.3f ms  %.1f G atomics/s  %.2f lanes/clk/CU @2.4GHz\n", name, ms, ops / (ms * 1e-3) / 1e9,
         ops / (ms * 1e-3) / 256 / 2.4e9);
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 4 * 256 * 4);
  run<0, 0>("ds_add_f32 distinct", d);
  run<0, 1>("ds_add_f32 random", d);
  run<1, 0>("ds_add_u32 distinct", d);
  run<1, 1>("ds_add_u32 random", d);
  run<2, 0>("ds_add_u64 distinct", d);
  run<2, 1>("ds_add_u64 random", d);
  run<3, 0>("plain rmw distinct", d);
  run<3, 1>("plain rmw random", d);
  return 0;
}
