// TEST INFRASTRUCTURE ONLY -- SIMT emulator runtime (see include/hip/hip_runtime.h).
// One OS worker thread runs one workgroup at a time; each GPU thread of the workgroup is a
// fiber on that OS thread.  Barriers are cooperative: the last arriver releases the rest.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

extern "C" void emu_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_ctx_switch, .-emu_ctx_switch
)");

namespace emu {

thread_local Ctx* cur = nullptr;

namespace {

constexpr int kMaxThreads = 1024;
constexpr size_t kStackBytes = 96 * 1024;
constexpr int kSlotBytes = 256;

enum State { RUNNABLE = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

struct Fiber {
  void* sp;
  int state;
  Ctx ctx;
};

struct BlockRunner {
  char* stacks = nullptr;
  Fiber fibers[kMaxThreads];
  unsigned char* slots = nullptr;  // [waves][64][kSlotBytes]
  void* sched_sp = nullptr;
  Fiber* running = nullptr;
  const std::function<void()>* body = nullptr;
  int nthreads = 0, nwaves = 0;
  int live = 0, block_arrived = 0;
  int wave_live[kMaxThreads / 64];
  int wave_arrived[kMaxThreads / 64];

  BlockRunner() {
    stacks = (char*)mmap(nullptr, kStackBytes * kMaxThreads, PROT_READ | PROT_WRITE,
                         MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (stacks == MAP_FAILED) {
      fprintf(stderr, "emu: stack mmap failed\n");
      abort();
    }
    slots = (unsigned char*)aligned_alloc(64, (size_t)(kMaxThreads / 64) * 64 * kSlotBytes);
  }
  ~BlockRunner() {
    munmap(stacks, kStackBytes * kMaxThreads);
    free(slots);
  }
};

thread_local BlockRunner* tl_runner = nullptr;

void release_block(BlockRunner* r) {
  for (int i = 0; i < r->nthreads; ++i)
    if (r->fibers[i].state == WAIT_BLOCK) r->fibers[i].state = RUNNABLE;
  r->block_arrived = 0;
}
void release_wave(BlockRunner* r, int w) {
  int lo = w * 64, hi = lo + 64 < r->nthreads ? lo + 64 : r->nthreads;
  for (int i = lo; i < hi; ++i)
    if (r->fibers[i].state == WAIT_WAVE) r->fibers[i].state = RUNNABLE;
  r->wave_arrived[w] = 0;
}

void fiber_entry() {
  BlockRunner* r = tl_runner;
  Fiber* f = r->running;
  (*r->body)();
  f->state = DONE;
  r->live--;
  int w = f->ctx.wave;
  r->wave_live[w]--;
  if (r->live > 0 && r->block_arrived == r->live) release_block(r);
  if (r->wave_live[w] > 0 && r->wave_arrived[w] == r->wave_live[w]) release_wave(r, w);
  void* dummy;
  emu_ctx_switch(&dummy, r->sched_sp);
  abort();  // never resumed
}

void yield_to_scheduler() {
  BlockRunner* r = tl_runner;
  Fiber* f = r->running;
  emu_ctx_switch(&f->sp, r->sched_sp);
}

void run_block(BlockRunner* r, dim3 grid, dim3 block, unsigned bx, unsigned by, unsigned bz,
               const std::function<void()>& body) {
  int n = (int)(block.x * block.y * block.z);
  if (n > kMaxThreads) {
    fprintf(stderr, "emu: block too large (%d)\n", n);
    abort();
  }
  r->body = &body;
  r->nthreads = n;
  r->nwaves = (n + 63) / 64;
  r->live = n;
  r->block_arrived = 0;
  for (int w = 0; w < r->nwaves; ++w) {
    r->wave_arrived[w] = 0;
    int lo = w * 64, hi = lo + 64 < n ? lo + 64 : n;
    r->wave_live[w] = hi - lo;
  }
  for (int i = 0; i < n; ++i) {
    Fiber& f = r->fibers[i];
    f.state = RUNNABLE;
    f.ctx.tid.x = i % block.x;
    f.ctx.tid.y = (i / block.x) % block.y;
    f.ctx.tid.z = i / (block.x * block.y);
    f.ctx.bid = uint3{bx, by, bz};
    f.ctx.bdim = block;
    f.ctx.gdim = grid;
    f.ctx.linear = i;
    f.ctx.lane = i & 63;
    f.ctx.wave = i >> 6;
    char* top = r->stacks + (size_t)(i + 1) * kStackBytes;
    void** sp = (void**)top;
    *--sp = nullptr;               // fake return address of fiber_entry
    *--sp = (void*)&fiber_entry;   // 'ret' target of the first switch
    for (int k = 0; k < 6; ++k) *--sp = nullptr;  // r15 r14 r13 r12 rbx rbp
    f.sp = (void*)sp;
  }
  while (r->live > 0) {
    bool progressed = false;
    for (int i = 0; i < n; ++i) {
      Fiber& f = r->fibers[i];
      if (f.state != RUNNABLE) continue;
      progressed = true;
      r->running = &f;
      cur = &f.ctx;
      emu_ctx_switch(&r->sched_sp, f.sp);
    }
    if (!progressed) {
      fprintf(stderr, "emu: deadlock in block (%u,%u,%u): live=%d block_arrived=%d\n", bx, by, bz, r->live,
              r->block_arrived);
      abort();
    }
  }
  cur = nullptr;
}

// ---- worker pool --------------------------------------------------------------------------
struct Pool {
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  const std::function<void()>* body = nullptr;
  dim3 grid, block;
  std::atomic<long> next{0};
  long total = 0;
  int generation = 0;
  int active = 0;
  bool stop = false;

  void worker() {
    BlockRunner* r = new BlockRunner();
    tl_runner = r;
    int seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_work.wait(lk, [&] { return stop || generation != seen; });
        if (stop) break;
        seen = generation;
      }
      drain(r);
      {
        std::unique_lock<std::mutex> lk(mu);
        if (--active == 0) cv_done.notify_all();
      }
    }
    delete r;
  }
  void drain(BlockRunner* r) {
    for (;;) {
      long b = next.fetch_add(1);
      if (b >= total) break;
      unsigned bx = (unsigned)(b % grid.x);
      unsigned by = (unsigned)((b / grid.x) % grid.y);
      unsigned bz = (unsigned)(b / ((long)grid.x * grid.y));
      run_block(r, grid, block, bx, by, bz, *body);
    }
  }
  int nworkers() {
    const char* e = getenv("MAPNET_EMU_THREADS");
    int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return n;
  }
  void run(dim3 g, dim3 b, const std::function<void()>& f) {
    long tot = (long)g.x * g.y * g.z;
    if (tot == 0) return;
    int nw = nworkers();
    if (nw == 1 || tot == 1) {
      // run inline on the calling thread
      static thread_local BlockRunner* inline_runner = nullptr;
      if (!inline_runner) inline_runner = new BlockRunner();
      BlockRunner* saved = tl_runner;
      tl_runner = inline_runner;
      for (long i = 0; i < tot; ++i) {
        unsigned bx = (unsigned)(i % g.x);
        unsigned by = (unsigned)((i / g.x) % g.y);
        unsigned bz = (unsigned)(i / ((long)g.x * g.y));
        run_block(inline_runner, g, b, bx, by, bz, f);
      }
      tl_runner = saved;
      return;
    }
    std::unique_lock<std::mutex> lk(mu);
    if (threads.empty())
      for (int i = 0; i < nw; ++i) threads.emplace_back([this] { worker(); });
    body = &f;
    grid = g;
    block = b;
    total = tot;
    next.store(0);
    active = (int)threads.size();
    generation++;
    cv_work.notify_all();
    cv_done.wait(lk, [&] { return active == 0; });
  }
};

Pool& pool() {
  static Pool* p = new Pool();  // leaked on purpose: workers live for the process
  return *p;
}
std::mutex launch_mu;

}  // namespace

void block_sync() {
  BlockRunner* r = tl_runner;
  Fiber* f = r->running;
  f->state = WAIT_BLOCK;
  if (++r->block_arrived == r->live) {
    release_block(r);
    return;
  }
  yield_to_scheduler();
}

void wave_sync() {
  BlockRunner* r = tl_runner;
  Fiber* f = r->running;
  int w = f->ctx.wave;
  f->state = WAIT_WAVE;
  if (++r->wave_arrived[w] == r->wave_live[w]) {
    release_wave(r, w);
    return;
  }
  yield_to_scheduler();
}

unsigned char* wave_slot(int lane) {
  BlockRunner* r = tl_runner;
  return r->slots + ((size_t)r->running->ctx.wave * 64 + lane) * kSlotBytes;
}

int wave_live_lanes() {
  BlockRunner* r = tl_runner;
  int w = r->running->ctx.wave;
  int lo = w * 64, hi = lo + 64 < r->nthreads ? lo + 64 : r->nthreads;
  return hi - lo;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  std::lock_guard<std::mutex> g(launch_mu);
  pool().run(grid, block, body);
}

}  // namespace emu

struct emu_event {
  std::chrono::steady_clock::time_point t;
};
hipError_t hipEventCreate(hipEvent_t* e) {
  *e = new emu_event();
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  e->t = std::chrono::steady_clock::now();
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
