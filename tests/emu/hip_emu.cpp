// TEST INFRASTRUCTURE (tests/emu): runtime of the host emulation declared in shim/hip/hip_runtime.h -- fibers, the
// workgroup scheduler, wave-level exchanges with the gfx950 lane layouts, and a malloc-backed HIP memory / stream API.
// Nothing in hybridq_amd loads this; see the header for what the model covers and what it does not.
#include <dlfcn.h>
#include <sys/mman.h>
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

void launch(dim3 grid, dim3 block, size_t lds, const std::function<void()>& body) {
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
    unsigned long seen = ~0ul;
    while (blk.nactive > 0) {
      if (blk.progress == seen) { fprintf(stderr, "hq_emu: deadlock in workgroup (%u,%u,%u): %d threads wait at a rendezvous nobody else reaches\n", bx, by, bz, blk.nactive); abort(); }
      seen = blk.progress;
      if (order_mode() == 2)
        for (int w = nw - 1; w > 0; --w) {
          rng = rng * 6364136223846793005ull + 1442695040888963407ull;
          std::swap(worder[w], worder[(rng >> 33) % (unsigned)(w + 1)]);
        }
      for (int wi = 0; wi < nw; ++wi)
        for (unsigned l = 0; l < 64; ++l) {
          const unsigned t = (unsigned)worder[wi] * 64 + l;
          if (t >= nt || blk.f[t].done) continue;
          g_fib = &blk.f[t];
          cur = &g_fib->lane;
          hq_emu_switch(&g_sched_sp, g_fib->sp);
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
struct hq_emu_stream { int id; };
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
hipError_t hipMalloc(void** p, size_t n) {
  void* q = nullptr;
  if (posix_memalign(&q, 4096, n ? n : 1)) return ret(hipErrorOutOfMemory);
  allocs()[(uintptr_t)q] = n ? n : 1;
  *p = q;
  return hipSuccess;
}
hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  if (!allocs().erase((uintptr_t)p)) return ret(hipErrorInvalidValue);
  free(p);
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
  if (!find_alloc(p, nullptr, nullptr)) return ret(hipErrorInvalidValue);
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
hipError_t hipStreamCreate(hipStream_t* s) { *s = new hq_emu_stream{1}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return ret(hipErrorNotSupported); }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return ret(hipErrorNotSupported); }
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return ret(hipErrorNotSupported); }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return ret(hipErrorNotSupported); }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
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
// virtual-memory management and IPC: not emulated (the tuned placement and the peer-to-peer transport are about physical
// memory and other processes' devices); callers take their documented failure paths
hipError_t hipMemGetAllocationGranularity(size_t*, const hipMemAllocationProp*, hipMemAllocationGranularity_flags) { return ret(hipErrorNotSupported); }
hipError_t hipMemAddressReserve(void**, size_t, size_t, void*, unsigned long long) { return ret(hipErrorNotSupported); }
hipError_t hipMemAddressFree(void*, size_t) { return ret(hipErrorNotSupported); }
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t*, size_t, const hipMemAllocationProp*, unsigned long long) { return ret(hipErrorNotSupported); }
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t) { return ret(hipErrorNotSupported); }
hipError_t hipMemMap(void*, size_t, size_t, hipMemGenericAllocationHandle_t, unsigned long long) { return ret(hipErrorNotSupported); }
hipError_t hipMemUnmap(void*, size_t) { return ret(hipErrorNotSupported); }
hipError_t hipMemSetAccess(void*, size_t, const hipMemAccessDesc*, size_t) { return ret(hipErrorNotSupported); }
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t*, void*) { return ret(hipErrorNotSupported); }
hipError_t hipIpcOpenMemHandle(void**, hipIpcMemHandle_t, unsigned) { return ret(hipErrorNotSupported); }
hipError_t hipIpcCloseMemHandle(void*) { return ret(hipErrorNotSupported); }
