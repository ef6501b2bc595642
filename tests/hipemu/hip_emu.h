// Minimal HIP execution-model emulator for the CPU-only test tier.
//
// TEST INFRASTRUCTURE.  It compiles the SAME kernel source
// (differentiable-point-clouds_amd/csrc/dpc_kernels.hip) with g++ so that the
// indexing / tiling / reduction logic of every kernel can be checked against
// the oracle in this GPU-less container before GPU minutes are spent.  It is
// not a fallback: the product loader only ever opens libdpc_hip.so; this
// library is built into tests/hipemu/ and opened only by tests (-m "not gpu").
//
// Model (round 5: fibers; rounds 1-4 ran every lane as an OS thread and spent six times the kernels' own time in futex
// calls): the threads of a block are user-space fibers (ucontext) of ONE OS thread, switched at the block's synchronisation
// points only -- __syncthreads and the per-wavefront rendezvous of shuffles / ballots / DPP moves -- in lane order, so a run
// is deterministic within a block; `__shared__` becomes a function-local `static thread_local`, i.e. one copy per OS
// thread = per block in flight; a launch's blocks are dealt round-robin to a small pool of OS threads (DPC_EMU_THREADS,
// default min(8, cores)) and run one after another on each, so atomics on global memory meet real concurrency between
// blocks, as on the GPU.  Wave shuffles go through a per-block exchange buffer and rendezvous per WAVE: they may sit in
// wave-uniform (not necessarily block-uniform) control flow.  Wavefront width is 64, as on gfx950.
#pragma once
#include <sched.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <functional>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int4 { int x, y, z, w; };
static inline int4 make_int4(int x, int y, int z, int w) { int4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local
#define __launch_bounds__(...)

namespace hipemu {
struct Ctx {
  dim3 tid, bid, bdim, gdim;
  unsigned flat;  // flat thread index in block
};
// the block this OS thread is running (hip_emu.cpp)
struct Block {
  unsigned nthreads;
  unsigned char* dyn_smem;
  unsigned int* xchg;      // per-thread 32-bit exchange words for shuffles
  int vote;
};
unsigned concurrency();                // OS threads of the block pool = work-groups that run concurrently (block i on thread i % n)
extern thread_local Ctx* t_ctxp;       // the fiber that is running on this OS thread
extern thread_local Block* t_block;
inline Ctx& cur() { return *t_ctxp; }
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
void sync();        // all fibers of the block
void wave_sync();   // the (up to) 64 fibers of the caller's wavefront
}  // namespace hipemu

#define threadIdx (hipemu::cur().tid)
#define blockIdx (hipemu::cur().bid)
#define blockDim (hipemu::cur().bdim)
#define gridDim (hipemu::cur().gdim)

static inline void __syncthreads() { hipemu::sync(); }

static inline int __syncthreads_or(int pred) {
  hipemu::sync();
  if (hipemu::cur().flat == 0) __atomic_store_n(&hipemu::t_block->vote, 0, __ATOMIC_RELAXED);
  hipemu::sync();
  if (pred) __atomic_store_n(&hipemu::t_block->vote, 1, __ATOMIC_RELAXED);
  hipemu::sync();
  const int r = __atomic_load_n(&hipemu::t_block->vote, __ATOMIC_RELAXED);
  hipemu::sync();
  return r;
}

static inline float atomicAdd(float* p, float v) {
  uint32_t* ip = reinterpret_cast<uint32_t*>(p);
  uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED);
  for (;;) {
    float f;
    memcpy(&f, &old, 4);
    f += v;
    uint32_t nw;
    memcpy(&nw, &f, 4);
    if (__atomic_compare_exchange_n(ip, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
      float r;
      memcpy(&r, &old, 4);
      return r;
    }
  }
}

static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

// wave64 shuffles / votes through the exchange buffer; they rendezvous per WAVE, so
// they may sit in wave-uniform (not necessarily block-uniform) control flow, as on the GPU
static inline float __shfl_down(float v, unsigned delta, int width = 64) {
  (void)width;
  unsigned f = hipemu::cur().flat;
  memcpy(&hipemu::t_block->xchg[f], &v, 4);
  hipemu::wave_sync();
  unsigned lane = f & 63u;
  unsigned nthreads = hipemu::cur().bdim.x * hipemu::cur().bdim.y * hipemu::cur().bdim.z;
  float r = v;
  if (lane + delta < 64u && f + delta < nthreads) memcpy(&r, &hipemu::t_block->xchg[f + delta], 4);
  hipemu::wave_sync();
  return r;
}
static inline unsigned long long __ballot(int pred) {
  unsigned f = hipemu::cur().flat;
  hipemu::t_block->xchg[f] = pred ? 1u : 0u;
  hipemu::wave_sync();
  unsigned nthreads = hipemu::cur().bdim.x * hipemu::cur().bdim.y * hipemu::cur().bdim.z;
  unsigned base = f & ~63u;
  unsigned long long r = 0;
  for (unsigned i = 0; i < 64u && base + i < nthreads; ++i)
    if (hipemu::t_block->xchg[base + i]) r |= 1ull << i;
  hipemu::wave_sync();
  return r;
}
// wave-level vote (block-uniform control flow only, like the shuffles)
static inline int __any(int pred) {
  unsigned f = hipemu::cur().flat;
  hipemu::t_block->xchg[f] = pred ? 1u : 0u;
  hipemu::wave_sync();
  unsigned nthreads = hipemu::cur().bdim.x * hipemu::cur().bdim.y * hipemu::cur().bdim.z;
  unsigned base = f & ~63u;
  int r = 0;
  for (unsigned i = base; i < base + 64u && i < nthreads; ++i) r |= (int)hipemu::t_block->xchg[i];
  hipemu::wave_sync();
  return r;
}
static inline float __shfl(float v, int src_lane, int width = 64) {
  (void)width;
  unsigned f = hipemu::cur().flat;
  memcpy(&hipemu::t_block->xchg[f], &v, 4);
  hipemu::wave_sync();
  unsigned nthreads = hipemu::cur().bdim.x * hipemu::cur().bdim.y * hipemu::cur().bdim.z;
  unsigned src = (f & ~63u) | ((unsigned)src_lane & 63u);
  float r = v;
  if (src < nthreads) memcpy(&r, &hipemu::t_block->xchg[src], 4);
  hipemu::wave_sync();
  return r;
}
static inline float __shfl_xor(float v, int mask, int width = 64) {
  (void)width;
  unsigned f = hipemu::cur().flat;
  memcpy(&hipemu::t_block->xchg[f], &v, 4);
  hipemu::wave_sync();
  unsigned nthreads = hipemu::cur().bdim.x * hipemu::cur().bdim.y * hipemu::cur().bdim.z;
  unsigned src = (f & ~63u) | ((f & 63u) ^ (unsigned)mask);
  float r = v;
  if (src < nthreads) memcpy(&r, &hipemu::t_block->xchg[src], 4);
  hipemu::wave_sync();
  return r;
}

// DPP row shifts and row broadcasts (v_mov_b32 dpp row_shl:n / row_shr:n / row_bcast:15 / row_bcast:31, the controls the
// kernels use): rows of 16 lanes;
// row_shr:n -- lane i reads lane i-n, row_shl:n -- lane i reads lane i+n; a lane without a source inside
// its row gets 0 (bound_ctrl) or keeps `old`.  Lane mapping verified on MI355X (scripts/ubench/dpp_check.hip).
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  (void)bank_mask;
  unsigned f = hipemu::cur().flat;
  memcpy(&hipemu::t_block->xchg[f], &src, 4);
  hipemu::wave_sync();
  const int lane = (int)(f & 63u), r = lane & 15, row = lane >> 4;
  int from = -1;
  if (ctrl >= 0x101 && ctrl <= 0x10f) from = (r + (ctrl - 0x100) < 16) ? lane + (ctrl - 0x100) : -1;
  else if (ctrl >= 0x111 && ctrl <= 0x11f) from = (r - (ctrl - 0x110) >= 0) ? lane - (ctrl - 0x110) : -1;
  else if (ctrl == 0x142) from = row >= 1 ? 16 * row - 1 : -1;     // row_bcast:15: lane 15 of the previous row
  else if (ctrl == 0x143) from = row >= 2 ? 31 : -1;               // row_bcast:31: lane 31 to rows 2 and 3
  else abort();
  unsigned nthreads = hipemu::cur().bdim.x * hipemu::cur().bdim.y * hipemu::cur().bdim.z;
  int res = bound_ctrl ? 0 : old;
  if (!((row_mask >> row) & 1)) from = -1, res = old;              // rows outside row_mask keep `old`
  if (from >= 0 && (f & ~63u) + (unsigned)from < nthreads) memcpy(&res, &hipemu::t_block->xchg[(f & ~63u) + (unsigned)from], 4);
  hipemu::wave_sync();
  return res;
}
static inline int __float_as_int(float v) { int r; memcpy(&r, &v, 4); return r; }
static inline float __int_as_float(int v) { float r; memcpy(&r, &v, 4); return r; }
static inline unsigned __float_as_uint(float v) { unsigned r; memcpy(&r, &v, 4); return r; }

// csrc/k_asm_gfx950.inc, restated
typedef float dpc_v2f __attribute__((vector_size(8)));
template <int SEL>
static inline void dpc_pk_fma_tap(dpc_v2f& acc, dpc_v2f pair, dpc_v2f v) {
  const dpc_v2f tt = dpc_v2f{pair[SEL], pair[SEL]};
  acc += tt * v;
}

template <int SEL>
static inline void dpc_pk_mul_tap(dpc_v2f& acc, dpc_v2f pair, dpc_v2f v) {
  const dpc_v2f tt = dpc_v2f{pair[SEL], pair[SEL]};
  acc = tt * v;
}

static inline void dpc_consume(float) {}
static inline float __fdividef(float a, float b) { return a / b; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // only used on wave-uniform values
static inline void __builtin_amdgcn_s_sleep(int) { sched_yield(); }          // (a spin on a counter another work-group's OS thread advances)
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  memset(p, v, n);
  return hipSuccess;
}
static inline hipError_t hipGetLastError() { return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...)               \
  do {                                                                          \
    (void)(stream);                                                             \
    hipemu::launch((grid), (block), (smem), [&]() { kernel(__VA_ARGS__); });    \
  } while (0)

// HIP's own spelling of `extern __shared__ type var[];`
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::t_block->dyn_smem);

// AMDGCN builtins used by the kernels
static inline float __builtin_amdgcn_fmed3f(float v, float lo, float hi) { return fmaxf(fminf(v, hi), lo); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }

// HIP events (the library's optional per-kernel timing): no-ops, elapsed time 0
typedef int hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = 0; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
