// See hip_emu.h.  Launcher of the CPU emulation tier: a block's threads are fibers of one OS thread, a launch's blocks are
// dealt to a small pool of OS threads.  TEST INFRASTRUCTURE.
#include "hip_emu.h"

#include <sys/mman.h>
#include <ucontext.h>

#include <condition_variable>
#include <mutex>

namespace hipemu {
thread_local Ctx* t_ctxp = nullptr;
thread_local Block* t_block = nullptr;

namespace {
constexpr size_t kStack = 256 << 10;   // per fiber; mapped lazily (MAP_NORESERVE), only touched pages exist

struct Fiber {
  ucontext_t uc;
  Ctx ctx;
  bool done;
};

// Everything one OS thread needs to run blocks of `nthreads` fibers; kept between launches (stacks, contexts).
struct Runner {
  std::vector<Fiber> fib;
  char* stacks = nullptr;
  size_t stack_bytes = 0;
  ucontext_t sched;
  unsigned cur = 0, nthreads = 0;
  // barriers: a counter and a generation each; the last arriver advances the generation and runs on, the others yield
  unsigned block_arrived = 0, block_gen = 0;
  std::vector<unsigned> wave_arrived, wave_gen, wave_size;
  Block blk{};
  std::vector<unsigned char> dyn;
  std::vector<unsigned> xchg;
  // the job
  const std::function<void()>* body = nullptr;
  dim3 grid, bdim;
  unsigned first = 0, stride = 1;     // this runner's blocks: first, first + stride, ...
  unsigned remaining = 0;

  ~Runner() {
    if (stacks) munmap(stacks, stack_bytes);
  }
  void yield() { swapcontext(&fib[cur].uc, &sched); }
  void sync_block() {
    const unsigned gen = block_gen;
    if (++block_arrived == nthreads) {
      block_arrived = 0;
      ++block_gen;
      return;
    }
    while (block_gen == gen) yield();
  }
  void sync_wave() {
    const unsigned w = fib[cur].ctx.flat >> 6;
    const unsigned gen = wave_gen[w];
    if (++wave_arrived[w] == wave_size[w]) {
      wave_arrived[w] = 0;
      ++wave_gen[w];
      return;
    }
    while (wave_gen[w] == gen) yield();
  }
  void fiber_main(unsigned i) {
    Ctx& c = fib[i].ctx;
    const unsigned long long nblocks = (unsigned long long)grid.x * grid.y * grid.z;
    for (unsigned long long b = first; b < nblocks; b += stride) {
      c.bid = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((unsigned long long)grid.x * grid.y)));
      (*body)();
      sync_block();   // block boundary: LDS statics and the exchange buffer are reused by the next block
    }
    fib[i].done = true;
    --remaining;
  }
  void run(dim3 g, dim3 b, size_t smem, const std::function<void()>* fn, unsigned first_, unsigned stride_);
};

thread_local Runner* t_runner = nullptr;

void fiber_entry(unsigned lo, unsigned hi) {
  (void)hi;
  t_runner->fiber_main(lo);
}

void Runner::run(dim3 g, dim3 b, size_t smem, const std::function<void()>* fn, unsigned first_, unsigned stride_) {
  grid = g;
  bdim = b;
  body = fn;
  first = first_;
  stride = stride_;
  nthreads = b.x * b.y * b.z;
  if ((unsigned long long)g.x * g.y * g.z <= first) return;
  if (fib.size() < nthreads) fib.resize(nthreads);
  if (stack_bytes < (size_t)nthreads * kStack) {
    if (stacks) munmap(stacks, stack_bytes);
    stack_bytes = (size_t)nthreads * kStack;
    stacks = static_cast<char*>(mmap(nullptr, stack_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (stacks == MAP_FAILED) abort();
  }
  const unsigned nwaves = (nthreads + 63) / 64;
  wave_arrived.assign(nwaves, 0);
  wave_gen.assign(nwaves, 0);
  wave_size.resize(nwaves);
  for (unsigned w = 0; w < nwaves; ++w) wave_size[w] = (w + 1) * 64 <= nthreads ? 64 : nthreads - w * 64;
  block_arrived = 0;
  block_gen = 0;
  dyn.resize(smem + 64);
  xchg.resize(nthreads);
  blk.nthreads = nthreads;
  blk.dyn_smem = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(dyn.data()) + 63) & ~uintptr_t(63));
  blk.xchg = xchg.data();
  blk.vote = 0;
  t_block = &blk;
  t_runner = this;
  for (unsigned i = 0; i < nthreads; ++i) {
    Fiber& f = fib[i];
    f.done = false;
    f.ctx.flat = i;
    f.ctx.bdim = b;
    f.ctx.gdim = g;
    f.ctx.tid = dim3(i % b.x, (i / b.x) % b.y, i / (b.x * b.y));
    getcontext(&f.uc);
    f.uc.uc_stack.ss_sp = stacks + (size_t)i * kStack;
    f.uc.uc_stack.ss_size = kStack;
    f.uc.uc_link = &sched;
    makecontext(&f.uc, reinterpret_cast<void (*)()>(fiber_entry), 2, i, 0u);
  }
  remaining = nthreads;
  // lane order, round and round: a fiber runs until it has to wait for its block / wavefront, the last arriver of a
  // rendezvous runs straight on -- one switch per fiber and rendezvous
  while (remaining) {
    for (unsigned i = 0; i < nthreads; ++i) {
      if (fib[i].done) continue;
      cur = i;
      t_ctxp = &fib[i].ctx;
      swapcontext(&sched, &fib[i].uc);
    }
  }
  t_ctxp = nullptr;
  t_block = nullptr;
}

// ---- the pool: workers sleep on a condition variable between launches -------------------------------------------------
struct Pool {
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::vector<std::thread> workers;
  unsigned long long epoch = 0;
  unsigned pending = 0, nworkers = 0, active = 0;
  dim3 grid, block;
  size_t smem = 0;
  const std::function<void()>* body = nullptr;
  bool quit = false;

  explicit Pool(unsigned n) : nworkers(n) {
    for (unsigned w = 0; w < n; ++w) workers.emplace_back([this, w] { loop(w); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cv_job.notify_all();
    for (auto& t : workers) t.join();
  }
  void loop(unsigned w) {
    Runner r;
    unsigned long long seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu);
      cv_job.wait(lk, [&] { return quit || epoch != seen; });
      if (quit) return;
      seen = epoch;
      const dim3 g = grid, b = block;
      const size_t s = smem;
      const std::function<void()>* fn = body;
      const unsigned act = active;
      lk.unlock();
      if (w < act) r.run(g, b, s, fn, w, act);
      lk.lock();
      if (--pending == 0) cv_done.notify_one();
    }
  }
  void launch(dim3 g, dim3 b, size_t s, const std::function<void()>& fn) {
    const unsigned long long nblocks = (unsigned long long)g.x * g.y * g.z;
    std::unique_lock<std::mutex> lk(mu);
    grid = g;
    block = b;
    smem = s;
    body = &fn;
    active = (unsigned)(nblocks < nworkers ? nblocks : nworkers);
    pending = nworkers;
    ++epoch;
    cv_job.notify_all();
    cv_done.wait(lk, [&] { return pending == 0; });
  }
};

Pool& pool() {
  static Pool p([] {
    const char* e = getenv("DPC_EMU_THREADS");
    unsigned n = e ? (unsigned)atoi(e) : 0;
    if (n == 0) {
      n = std::thread::hardware_concurrency();
      if (n > 8) n = 8;
    }
    return n ? n : 1u;
  }());
  return p;
}
}  // namespace

unsigned concurrency() { return pool().nworkers; }
void sync() { t_runner->sync_block(); }
void wave_sync() { t_runner->sync_wave(); }

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
  if ((unsigned long long)grid.x * grid.y * grid.z == 0 || block.x * block.y * block.z == 0) return;
  pool().launch(grid, block, smem, body);
}
}  // namespace hipemu
