// TEST INFRASTRUCTURE (tests/emu): runtime of the host emulation declared in shim/hip/hip_runtime.h -- fibers, the
// workgroup scheduler, wave-level exchanges with the gfx950 lane layouts, and a malloc-backed HIP memory / stream API.
// Nothing in hybridq_amd loads this; see the header for what the model covers and what it does not.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <map>

#include "shim/hip/hip_runtime.h"
#undef asm
#undef volatile

namespace hq_emu {

Lane* cur = nullptr;
dim3 cur_grid, cur_block;

// ---- fibers: a minimal x86-64 context switch (callee-saved registers + stack pointer) ------------------------------
extern "C" void hq_emu_switch(void** save_sp, void* new_sp);
__asm__(
    ".text\n.globl hq_emu_switch\n.type hq_emu_switch,@function\nhq_emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size hq_emu_switch, .-hq_emu_switch\n");

constexpr size_t kStack = 512 << 10;
constexpr int kMaxThreads = 1024;

struct Wave {
  int nactive = 0, count = 0;
  unsigned gen = 0;
  uint64_t xa[2][64], xb[2][64];
  bool here[2][64];
};
struct Fiber {
  void* sp = nullptr;
  Lane lane;
  bool done = false;
  int wave = 0, lane_id = 0;
};
struct Block {
  std::vector<Fiber> f;
  std::vector<Wave> w;
  int nactive = 0, count = 0;
  unsigned gen = 0;
  unsigned long progress = 0;
};

static unsigned char* g_stacks = nullptr;
static unsigned char* g_lds = nullptr;
static void* g_sched_sp = nullptr;
static Block* g_blk = nullptr;
static Fiber* g_fib = nullptr;
static const std::function<void()>* g_body = nullptr;
constexpr size_t kLdsBytes = 1 << 20;

static void init_once() {
  if (g_stacks) return;
  g_stacks = (unsigned char*)mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (g_stacks == MAP_FAILED) { perror("hq_emu: stacks"); abort(); }
  // LDS in the low 4 GiB, aligned to 1 MiB: a kernel's 32-bit LDS addresses are host addresses
  for (uintptr_t hint = 0x20000000u; hint < 0xF0000000u && !g_lds; hint += 0x10000000u) {
    void* p = mmap((void*)hint, kLdsBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED_NOREPLACE, -1, 0);
    if (p == (void*)hint) g_lds = (unsigned char*)p;
    else if (p != MAP_FAILED) munmap(p, kLdsBytes);
  }
  if (!g_lds) { fprintf(stderr, "hq_emu: no room for the LDS window below 4 GiB\n"); abort(); }
}

unsigned char* dyn_lds() { return g_lds; }

static void yield() {
  Fiber* me = g_fib;
  hq_emu_switch(&me->sp, g_sched_sp);
  g_fib = me;
  cur = &me->lane;
}

static void fiber_exit() {
  Fiber* me = g_fib;
  Block& b = *g_blk;
  Wave& w = b.w[me->wave];
  me->done = true;
  ++b.progress;
  // a lane that has left no longer takes part in rendezvous (hardware: exited waves leave the barrier count)
  if (--w.nactive > 0 && w.count == w.nactive) { w.count = 0; ++w.gen; }
  if (--b.nactive > 0 && b.count == b.nactive) { b.count = 0; ++b.gen; }
  void* dummy;
  hq_emu_switch(&dummy, g_sched_sp);
  abort();  // never resumed
}

static void fiber_main() {
  (*g_body)();
  fiber_exit();
}

static void wave_sync() {
  Block& b = *g_blk;
  Wave& w = b.w[g_fib->wave];
  const unsigned gen = w.gen;
  if (++w.count == w.nactive) {
    w.count = 0;
    ++w.gen;
    ++b.progress;
  } else {
    while (w.gen == gen) yield();
  }
}

void block_barrier() {
  Block& b = *g_blk;
  const unsigned gen = b.gen;
  if (++b.count == b.nactive) {
    b.count = 0;
    ++b.gen;
    ++b.progress;
  } else {
    while (b.gen == gen) yield();
  }
}

int readfirstlane(int v) {
  Fiber* me = g_fib;
  Wave& w = g_blk->w[me->wave];
  const unsigned par = me->lane.nops++ & 1;
  w.xa[par][me->lane_id] = (uint64_t)(uint32_t)v;
  w.here[par][me->lane_id] = true;
  wave_sync();
  int first = 0;
  while (first < 64 && !w.here[par][first]) ++first;
  const int out = (int)(uint32_t)w.xa[par][first];
  wave_sync();  // everybody has read before the flags are cleared
  w.here[par][me->lane_id] = false;
  return out;
}

uint64_t shfl_bits(uint64_t bits, int arg, int mode) {
  Fiber* me = g_fib;
  Wave& w = g_blk->w[me->wave];
  const unsigned par = me->lane.nops++ & 1;
  w.xa[par][me->lane_id] = bits;
  wave_sync();
  int src = mode == 0 ? me->lane_id + arg : (mode == 1 ? (me->lane_id ^ arg) : arg);
  if (src < 0 || src > 63) src = me->lane_id;  // out of range: own value (HIP semantics)
  return w.xa[par][src];
}

void mfma_f32(float a, float b, const float* c, float* d) {
  Fiber* me = g_fib;
  Wave& w = g_blk->w[me->wave];
  const unsigned par = me->lane.nops++ & 1;
  const int l = me->lane_id;
  memcpy(&w.xa[par][l], &a, 4);
  memcpy(&w.xb[par][l], &b, 4);
  wave_sync();
  const int j = l & 15;
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * (l >> 4) + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, &w.xa[par][i + 16 * k], 4);
      memcpy(&bv, &w.xb[par][j + 16 * k], 4);
      acc = __builtin_fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
}

void mfma_f64(double a, double b, const double* c, double* d) {
  Fiber* me = g_fib;
  Wave& w = g_blk->w[me->wave];
  const unsigned par = me->lane.nops++ & 1;
  const int l = me->lane_id;
  memcpy(&w.xa[par][l], &a, 8);
  memcpy(&w.xb[par][l], &b, 8);
  wave_sync();
  const int j = l & 15;
  for (int r = 0; r < 4; ++r) {
    const int i = (l >> 4) + 4 * r;
    double acc = c[r];
    for (int k = 0; k < 4; ++k) {
      double av, bv;
      memcpy(&av, &w.xa[par][i + 16 * k], 8);
      memcpy(&bv, &w.xb[par][j + 16 * k], 8);
      acc = __builtin_fma(av, bv, acc);
    }
    d[r] = acc;
  }
}

static int order_mode() {
  static int m = -1;
  if (m < 0) {
    const char* e = getenv("HQ_EMU_ORDER");
    m = !e ? 0 : (!strcmp(e, "reverse") ? 1 : (!strcmp(e, "random") ? 2 : 0));
  }
  return m;
}

}  // namespace hq_emu
struct hq_emu_graph { std::vector<std::function<void()>> ops; };
struct hq_emu_graph_exec { std::vector<std::function<void()>> ops; };
struct hq_emu_stream { hq_emu_graph* capture = nullptr; };
namespace hq_emu {

static void run_grid(dim3 grid, dim3 block, size_t lds, const std::function<void()>& body);

void launch(dim3 grid, dim3 block, size_t lds, hipStream_t stream, std::function<void()> body) {
  if (stream && stream->capture) {
    stream->capture->ops.push_back([=]() { run_grid(grid, block, lds, body); });
    return;
  }
  run_grid(grid, block, lds, body);
}

static void run_grid(dim3 grid, dim3 block, size_t lds, const std::function<void()>& body) {
  init_once();
  const unsigned nt = block.x * block.y * block.z;
  if (nt == 0 || nt > (unsigned)kMaxThreads || lds > kLdsBytes) { fprintf(stderr, "hq_emu: launch shape not supported (%u threads, %zu B LDS)\n", nt, lds); abort(); }
  if (g_blk) { fprintf(stderr, "hq_emu: nested launch\n"); abort(); }
  cur_grid = grid;
  cur_block = block;
  const int nw = (int)((nt + 63) / 64);
  static uint64_t rng = 0x9E3779B97F4A7C15ull;
  std::vector<int> worder(nw);
  Block blk;
  blk.f.resize(nt);
  g_body = &body;
  for (unsigned bz = 0; bz < grid.z; ++bz)
  for (unsigned by = 0; by < grid.y; ++by)
  for (unsigned bx = 0; bx < grid.x; ++bx) {
    blk.w.assign(nw, Wave());
    for (auto& w : blk.w) memset(w.here, 0, sizeof(w.here));
    blk.nactive = (int)nt;
    blk.count = 0;
    blk.progress = 0;
    for (unsigned t = 0; t < nt; ++t) {
      Fiber& f = blk.f[t];
      f.done = false;
      f.wave = (int)(t / 64);
      f.lane_id = (int)(t % 64);
      f.lane.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
      f.lane.bid = dim3(bx, by, bz);
      f.lane.nops = 0;
      blk.w[f.wave].nactive++;
      // initial frame: six callee-saved slots, then the entry address `ret` jumps to; at entry rsp % 16 == 8
      uintptr_t top = (uintptr_t)(g_stacks + kStack * (t + 1));
      top &= ~(uintptr_t)15;
      void** sp = (void**)(top - 8);
      *--sp = (void*)&fiber_main;
      for (int i = 0; i < 6; ++i) *--sp = nullptr;
      f.sp = sp;
    }
    g_blk = &blk;
    for (int w = 0; w < nw; ++w) worder[w] = order_mode() == 1 ? nw - 1 - w : w;
    // One "wave round" runs each lane of a wave up to its next rendezvous.  forward: one round per wave in turn (the
    // waves advance together).  reverse: the waves in descending order, each as far as it can go -- up to a workgroup
    // barrier or its end -- before the next one starts: the largest drift between waves the program allows, which is what
    // exposes a missing barrier (a fair schedule hides it: every wave reads long after every wave has written).
    // random: a random wave for a random number of rounds.
    auto wave_round = [&](int w) {
      const unsigned long before = blk.progress;
      for (unsigned l = 0; l < 64; ++l) {
        const unsigned t = (unsigned)w * 64 + l;
        if (t >= nt || blk.f[t].done) continue;
        g_fib = &blk.f[t];
        cur = &g_fib->lane;
        hq_emu_switch(&g_sched_sp, g_fib->sp);
      }
      return blk.progress != before;
    };
    unsigned long seen = ~0ul;
    while (blk.nactive > 0) {
      if (blk.progress == seen) { fprintf(stderr, "hq_emu: deadlock in workgroup (%u,%u,%u): %d threads wait at a rendezvous nobody else reaches\n", bx, by, bz, blk.nactive); abort(); }
      seen = blk.progress;
      if (order_mode() == 2) {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        const int w = (int)((rng >> 33) % (unsigned)nw);
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        int burst = 1 + (int)((rng >> 33) % 300u);
        while (burst-- > 0 && wave_round(w)) {}
        if (blk.progress == seen)  // that wave is stuck at a barrier: give everybody a turn so that the check above is fair
          for (int wi = 0; wi < nw; ++wi) wave_round(wi);
      } else if (order_mode() == 1) {
        for (int wi = 0; wi < nw; ++wi)
          while (wave_round(worder[wi])) {}
      } else {
        for (int wi = 0; wi < nw; ++wi) wave_round(worder[wi]);
      }
    }
  }
  g_blk = nullptr;
  g_fib = nullptr;
  cur = nullptr;
  g_body = nullptr;
}

// ---- memory: malloc-backed "device" allocations, tracked so that pointer queries can tell them from host memory ------
static std::map<uintptr_t, size_t>& allocs() { static std::map<uintptr_t, size_t> m; return m; }
static bool find_alloc(const void* p, uintptr_t* base, size_t* size) {
  auto& m = allocs();
  auto it = m.upper_bound((uintptr_t)p);
  if (it == m.begin()) return false;
  --it;
  if ((uintptr_t)p >= it->first + it->second) return false;
  if (base) *base = it->first;
  if (size) *size = it->second;
  return true;
}
}  // namespace hq_emu

using namespace hq_emu;
struct hq_emu_event { std::chrono::steady_clock::time_point t; };
static hipError_t g_last = hipSuccess;
static hipError_t ret(hipError_t e) { if (e != hipSuccess) g_last = e; return e; }

const char* hipGetErrorString(hipError_t e) {
  switch (e) {
    case hipSuccess: return "no error";
    case hipErrorInvalidValue: return "invalid argument";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorNotSupported: return "operation not supported (host emulation)";
    case hipErrorNotReady: return "not ready";
  }
  return "unknown error";
}
hipError_t hipGetLastError() { hipError_t e = g_last; g_last = hipSuccess; return e; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
// Device allocations are POSIX shared-memory objects mapped MAP_SHARED, so that hipIpcGetMemHandle / hipIpcOpenMemHandle can
// hand them to ANOTHER PROCESS of the test (the peer-to-peer exchange stores into the other ranks' planes).  Under
// AddressSanitizer they are plain heap blocks instead (bounds-checked; no IPC there).
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HQ_EMU_HEAP_ALLOC 1
#endif
#endif
namespace {
struct Shm { std::string name; size_t len; };
std::map<uintptr_t, Shm>& shm_blocks() { static std::map<uintptr_t, Shm> m; return m; }
std::map<uintptr_t, size_t>& ipc_views() { static std::map<uintptr_t, size_t> m; return m; }
void unlink_all() {
  for (auto& kv : shm_blocks()) shm_unlink(kv.second.name.c_str());
}
}  // namespace
hipError_t hipMalloc(void** p, size_t n) {
  const size_t want = n ? n : 1;
#ifdef HQ_EMU_HEAP_ALLOC
  void* q = nullptr;
  if (posix_memalign(&q, 4096, want)) return ret(hipErrorOutOfMemory);
#else
  static unsigned long counter = 0;
  static bool hooked = false;
  if (!hooked) { atexit(unlink_all); hooked = true; }
  const size_t len = (want + 4095) & ~(size_t)4095;
  char name[64];
  snprintf(name, sizeof(name), "/hq_emu_%d_%lu", (int)getpid(), counter++);
  const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return ret(hipErrorOutOfMemory);
  if (ftruncate(fd, (off_t)len)) { close(fd); shm_unlink(name); return ret(hipErrorOutOfMemory); }
  void* q = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (q == MAP_FAILED) { shm_unlink(name); return ret(hipErrorOutOfMemory); }
  shm_blocks()[(uintptr_t)q] = Shm{name, len};
#endif
  allocs()[(uintptr_t)q] = want;
  *p = q;
  return hipSuccess;
}
hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  if (!allocs().erase((uintptr_t)p)) return ret(hipErrorInvalidValue);
#ifdef HQ_EMU_HEAP_ALLOC
  free(p);
#else
  auto it = shm_blocks().find((uintptr_t)p);
  if (it != shm_blocks().end()) {
    munmap(p, it->second.len);
    shm_unlink(it->second.name.c_str());
    shm_blocks().erase(it);
  }
#endif
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return posix_memalign(p, 4096, n ? n : 1) ? ret(hipErrorOutOfMemory) : hipSuccess; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = *t = (size_t)16 << 30; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemcpyFromSymbol(void* d, const void* s, size_t n) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
  // HQ_EMU_HOST_IS_DEVICE=1: every pointer counts as device memory (CPU torch tensors stand for device tensors when the
  // -m gpu tests run against the emulation; tests/emu/fake_cuda.py)
  static const bool all_device = getenv("HQ_EMU_HOST_IS_DEVICE") && atoi(getenv("HQ_EMU_HOST_IS_DEVICE")) != 0;
  if (!all_device && !find_alloc(p, nullptr, nullptr)) return ret(hipErrorInvalidValue);
  a->type = hipMemoryTypeDevice;
  a->device = 0;
  a->devicePointer = const_cast<void*>(p);
  a->hostPointer = nullptr;
  return hipSuccess;
}
hipError_t hipMemGetAddressRange(hipDeviceptr_t* base, size_t* size, hipDeviceptr_t p) {
  uintptr_t b;
  if (!find_alloc(p, &b, size)) return ret(hipErrorInvalidValue);
  *base = (void*)b;
  return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t* s) { *s = new hq_emu_stream(); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
// graphs: kernel launches on a capturing stream are recorded (arguments by value) and replayed by hipGraphLaunch
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) {
  if (!s || s->capture) return ret(hipErrorInvalidValue);
  s->capture = new hq_emu_graph();
  return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g) {
  if (!s || !s->capture) return ret(hipErrorInvalidValue);
  *g = s->capture;
  s->capture = nullptr;
  return hipSuccess;
}
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus* st) {
  *st = (s && s->capture) ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
  return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) {
  *e = new hq_emu_graph_exec{g->ops};
  return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) {
  for (auto& op : e->ops) op();
  return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hq_emu_event{std::chrono::steady_clock::now()}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
// virtual-memory management: address ranges are PROT_NONE reservations, physical granules are memfd files, a mapping is
// a MAP_SHARED | MAP_FIXED view of a granule -- so remapping the same granules elsewhere keeps their contents, an
// unmapped range faults, and the library's bookkeeping (hq_state.hip) is exercised for real.  IPC is not emulated.
namespace {
constexpr size_t kGranule = 1 << 16;  // "minimum granularity" of the emulated device
struct Granule { int fd; size_t size; };
std::map<uint64_t, Granule>& granules() { static std::map<uint64_t, Granule> m; return m; }
std::map<uintptr_t, size_t>& reservations() { static std::map<uintptr_t, size_t> m; return m; }
uint64_t g_next_handle = 1;
bool inside_reservation(void* p, size_t n) {
  auto& m = reservations();
  auto it = m.upper_bound((uintptr_t)p);
  if (it == m.begin()) return false;
  --it;
  return (uintptr_t)p + n <= it->first + it->second;
}
}  // namespace
hipError_t hipMemGetAllocationGranularity(size_t* g, const hipMemAllocationProp*, hipMemAllocationGranularity_flags) { *g = kGranule; return hipSuccess; }
hipError_t hipMemAddressReserve(void** p, size_t n, size_t, void*, unsigned long long) {
  void* q = mmap(nullptr, n, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (q == MAP_FAILED) return ret(hipErrorOutOfMemory);
  reservations()[(uintptr_t)q] = n;
  *p = q;
  return hipSuccess;
}
hipError_t hipMemAddressFree(void* p, size_t n) {
  auto it = reservations().find((uintptr_t)p);
  if (it == reservations().end() || it->second != n) return ret(hipErrorInvalidValue);
  munmap(p, n);
  reservations().erase(it);
  return hipSuccess;
}
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t* h, size_t n, const hipMemAllocationProp*, unsigned long long) {
  if (n == 0 || n % kGranule) return ret(hipErrorInvalidValue);
  const int fd = memfd_create("hq_emu_granule", 0);
  if (fd < 0 || ftruncate(fd, (off_t)n)) { if (fd >= 0) close(fd); return ret(hipErrorOutOfMemory); }
  *h = g_next_handle++;
  granules()[*h] = Granule{fd, n};
  return hipSuccess;
}
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t h) {
  auto it = granules().find(h);
  if (it == granules().end()) return ret(hipErrorInvalidValue);
  close(it->second.fd);  // views that are still mapped keep the pages alive, as on the device
  granules().erase(it);
  return hipSuccess;
}
hipError_t hipMemMap(void* p, size_t n, size_t off, hipMemGenericAllocationHandle_t h, unsigned long long) {
  auto it = granules().find(h);
  if (it == granules().end() || off + n > it->second.size || !inside_reservation(p, n)) return ret(hipErrorInvalidValue);
  // no access until hipMemSetAccess, as on the device
  if (mmap(p, n, PROT_NONE, MAP_SHARED | MAP_FIXED, it->second.fd, (off_t)off) == MAP_FAILED) return ret(hipErrorInvalidValue);
  allocs()[(uintptr_t)p] = n;  // device memory from now on
  return hipSuccess;
}
hipError_t hipMemUnmap(void* p, size_t n) {
  if (!inside_reservation(p, n)) return ret(hipErrorInvalidValue);
  if (mmap(p, n, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED, -1, 0) == MAP_FAILED) return ret(hipErrorInvalidValue);
  auto& m = allocs();
  for (auto it = m.lower_bound((uintptr_t)p); it != m.end() && it->first < (uintptr_t)p + n;) it = m.erase(it);
  return hipSuccess;
}
hipError_t hipMemSetAccess(void* p, size_t n, const hipMemAccessDesc*, size_t) {
  if (!inside_reservation(p, n)) return ret(hipErrorInvalidValue);
  return mprotect(p, n, PROT_READ | PROT_WRITE) ? ret(hipErrorInvalidValue) : hipSuccess;
}
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) {
#ifdef HQ_EMU_HEAP_ALLOC
  (void)h; (void)p;
  return ret(hipErrorNotSupported);
#else
  auto it = shm_blocks().find((uintptr_t)p);  // the BASE of an allocation, as on the device
  if (it == shm_blocks().end()) return ret(hipErrorInvalidValue);
  memset(h->reserved, 0, sizeof(h->reserved));
  snprintf(h->reserved, sizeof(h->reserved), "%s", it->second.name.c_str());
  return hipSuccess;
#endif
}
hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) {
  char name[65];
  memcpy(name, h.reserved, 64);
  name[64] = 0;
  const int fd = shm_open(name, O_RDWR, 0600);
  if (fd < 0) return ret(hipErrorInvalidValue);
  struct stat st;
  if (fstat(fd, &st)) { close(fd); return ret(hipErrorInvalidValue); }
  void* q = mmap(nullptr, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (q == MAP_FAILED) return ret(hipErrorOutOfMemory);
  ipc_views()[(uintptr_t)q] = (size_t)st.st_size;
  allocs()[(uintptr_t)q] = (size_t)st.st_size;  // a peer's memory is device memory here too
  *p = q;
  return hipSuccess;
}
hipError_t hipIpcCloseMemHandle(void* p) {
  auto it = ipc_views().find((uintptr_t)p);
  if (it == ipc_views().end()) return ret(hipErrorInvalidValue);
  munmap(p, it->second);
  allocs().erase((uintptr_t)p);
  ipc_views().erase(it);
  return hipSuccess;
}
