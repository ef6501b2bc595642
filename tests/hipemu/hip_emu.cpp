// See hip_emu.h.  Block-sequential, thread-parallel launcher.
#include "hip_emu.h"

namespace hipemu {
thread_local Ctx t_ctx;
pthread_barrier_t* g_barrier = nullptr;
unsigned char* g_dyn_smem = nullptr;
unsigned int* g_xchg = nullptr;
pthread_barrier_t* g_wave_barriers = nullptr;
int g_vote = 0;

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
  const unsigned nthreads = block.x * block.y * block.z;
  pthread_barrier_t barrier;
  pthread_barrier_init(&barrier, nullptr, nthreads);
  g_barrier = &barrier;
  std::vector<unsigned char> dyn(smem + 64);
  g_dyn_smem = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(dyn.data()) + 63) & ~uintptr_t(63));
  std::vector<unsigned int> xchg(nthreads);
  g_xchg = xchg.data();
  const unsigned nwaves = (nthreads + 63) / 64;
  std::vector<pthread_barrier_t> wbar(nwaves);
  for (unsigned w = 0; w < nwaves; ++w) {
    const unsigned cnt = (w + 1) * 64 <= nthreads ? 64 : nthreads - w * 64;
    pthread_barrier_init(&wbar[w], nullptr, cnt);
  }
  g_wave_barriers = wbar.data();
  auto worker = [&](unsigned flat) {
    Ctx& c = t_ctx;
    c.flat = flat;
    c.bdim = block;
    c.gdim = grid;
    c.tid = dim3(flat % block.x, (flat / block.x) % block.y, flat / (block.x * block.y));
    for (unsigned bz = 0; bz < grid.z; ++bz)
      for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
          c.bid = dim3(bx, by, bz);
          body();
          pthread_barrier_wait(&barrier);  // block boundary: LDS statics are reused
        }
  };
  std::vector<std::thread> th;
  th.reserve(nthreads);
  for (unsigned i = 0; i < nthreads; ++i) th.emplace_back(worker, i);
  for (auto& t : th) t.join();
  pthread_barrier_destroy(&barrier);
  for (auto& wb : wbar) pthread_barrier_destroy(&wb);
  g_wave_barriers = nullptr;
  g_barrier = nullptr;
  g_dyn_smem = nullptr;
  g_xchg = nullptr;
}
}  // namespace hipemu
