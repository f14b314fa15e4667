// hq_core.hip -- process-wide context of libhq_hip.so (stream, kernel selection, upload arena, scratch buffers),
// compiled programs (hq_program_*) and the small control entry points of include/hq_hip.h.
#include "hq_common.h"

namespace hq {

Context& ctx() {
  static Context c;
  return c;
}

int fail(const std::string& msg) {
  ctx().last_error = msg;
  // A failed runtime call leaves its code in the thread's "last error" until somebody reads it, and the launch checks
  // (hipGetLastError() after every kernel launch) would report it as THEIR failure -- after the caller has already taken
  // its fallback (tuned placement -> plain memory, IPC export refused -> another transport).  Whoever reports a failure
  // through here has consumed the runtime's error with it.  (Round 4, found twice by running against the host emulation:
  // hq_alloc_state without VMM, hq_ipc_export on memory IPC cannot export.)
  (void)hipGetLastError();
  return 1;
}

void read_env(Context& c) {
  if (c.env_read) return;
  c.env_read = true;
  if (const char* e = getenv("HQ_LOG2_PACK_SIZE")) {
    int v = atoi(e);
    if (v >= 1 && v <= 5) c.log2_pack = (unsigned)v;
  }
  if (const char* e = getenv("HQ_APPLY_MODE")) {
    std::string s(e);
    if (s == "direct") c.mode = Mode::Direct;
    else if (s == "mfma") c.mode = Mode::Mfma;
    else if (s == "generic") c.mode = Mode::Generic;
    else if (s == "naive") c.mode = Mode::Naive;
    else if (s == "tile") c.mode = Mode::Tile;
    else if (s == "gemm") c.mode = Mode::Gemm;
  }
  if (const char* e = getenv("HQ_NONTEMPORAL")) c.nontemporal = atoi(e) < 0 ? -1 : (atoi(e) != 0);
}

// The library keeps its upload arena and scratch buffers on ONE device: the design is one
// process per GPU (hybridq_amd.dist).  Using a second device from the same process is refused
// loudly instead of silently reading another device's memory.
int check_device(Context& c) {
  int dev = -1;
  HQ_HIP_CHECK(hipGetDevice(&dev));
  if (c.device < 0) c.device = dev;
  if (dev != c.device)
    return fail("libhq_hip is bound to device " + std::to_string(c.device) + " but the current device is " +
                std::to_string(dev) + ": use one process per GPU");
  return 0;
}

int get_scratch(Context& c, int slot, size_t bytes, void** out) {
  if (check_device(c)) return 1;
  if (c.scratch_size[slot] < bytes) {
    if (c.scratch[slot]) {
      HQ_HIP_CHECK(hipStreamSynchronize(c.stream));
      HQ_HIP_CHECK(hipFree(c.scratch[slot]));
      c.scratch[slot] = nullptr;
      c.scratch_size[slot] = 0;
    }
    HQ_HIP_CHECK(hipMalloc(&c.scratch[slot], bytes));
    c.scratch_size[slot] = bytes;
  }
  *out = c.scratch[slot];
  return 0;
}

// Copy `bytes` of host data into the arena; returns the device address in *dev.  The
// host copy is taken immediately (the caller's buffer may be a temporary), the H2D
// transfer is asynchronous on the stream.
int arena_upload(Context& c, const void* host, size_t bytes, void** dev) {
  const size_t kArena = 64u << 20;
  if (check_device(c)) return 1;
  if (!c.arena_host) {
    HQ_HIP_CHECK(hipHostMalloc((void**)&c.arena_host, kArena, hipHostMallocDefault));
    HQ_HIP_CHECK(hipMalloc((void**)&c.arena_dev, kArena));
    c.arena_size = kArena;
  }
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (c.rec) {  // recording: the table goes into the program's own buffer
    const size_t off = c.rec->host.size();
    if (off + need > c.rec->cap) return fail("program table buffer exhausted (HQ_PROGRAM_MB)");
    c.rec->host.resize(off + need);
    memcpy(c.rec->host.data() + off, host, bytes);
    *dev = c.rec->dev + off;
    return 0;
  }
  if (need > c.arena_size) return fail("matrix too large for the upload arena");
  if (c.arena_used + need > c.arena_size) {
    HQ_HIP_CHECK(hipStreamSynchronize(c.stream));  // wrap: wait for in-flight users
    c.arena_used = 0;
  }
  memcpy(c.arena_host + c.arena_used, host, bytes);
  if (bytes >= (64u << 10)) {
    // Large tables (k >= 6 operand tables, up to 8 MiB): copied by a kernel that reads the pinned
    // host arena directly, on the SAME stream.  hipMemcpyAsync would go through an SDMA queue and
    // the cross-queue dependency costs sporadic ~75 ms host-side stalls on this platform.
    const size_t n16 = (bytes + 15) / 16;
    const unsigned grid = (unsigned)std::min<size_t>((n16 + kBlock - 1) / kBlock, 512);
    hipLaunchKernelGGL(upload_kernel, dim3(grid), dim3(kBlock), 0, c.stream,
                       reinterpret_cast<uint4*>(c.arena_dev + c.arena_used),
                       reinterpret_cast<const uint4*>(c.arena_host + c.arena_used), n16);
    HQ_HIP_CHECK(hipGetLastError());
  } else {
    HQ_HIP_CHECK(hipMemcpyAsync(c.arena_dev + c.arena_used, c.arena_host + c.arena_used, bytes,
                                hipMemcpyHostToDevice, c.stream));
  }
  *dev = c.arena_dev + c.arena_used;
  c.arena_used += need;
  return 0;
}

// true if `p` can be dereferenced by a kernel
bool is_device_pointer(const void* p) {
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // unregistered host memory
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged ||
         attr.type == hipMemoryTypeUnified;
}

int check_positions(const unsigned* pos, unsigned n, unsigned k) {
  if (k > n || n > 62) return 1;
  uint64_t seen = 0;
  for (unsigned i = 0; i < k; ++i) {
    if (pos[i] >= n) return 1;
    if (seen & (1ull << pos[i])) return 1;
    seen |= 1ull << pos[i];
  }
  return 0;
}

}  // namespace hq

extern "C" {

unsigned int get_log2_pack_size(void) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::read_env(c);
  return c.log2_pack;
}

int hq_program_begin(void) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::read_env(c);
  if (c.rec) return hq::fail("hq_program_begin: already recording");
  if (hq::check_device(c)) return 1;
  hq::Program* p = new hq::Program();
  size_t mb = 64;
  if (const char* e = getenv("HQ_PROGRAM_MB")) mb = (size_t)std::max(1, atoi(e));
  p->cap = mb << 20;
  if (const char* e = getenv("HQ_PROGRAM_GRAPH")) p->use_graph = atoi(e) != 0;
  hipError_t err = hipMalloc((void**)&p->dev, p->cap);
  if (err != hipSuccess) {
    delete p;
    return hq::fail(std::string("hq_program_begin: hipMalloc: ") + hipGetErrorString(err));
  }
  c.rec = p;
  return 0;
}

int hq_program_end(void** handle) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  if (!c.rec) return hq::fail("hq_program_end: not recording");
  hq::Program* p = c.rec;
  c.rec = nullptr;
  if (!handle) {  // nobody could run or free it
    (void)hipFree(p->dev);
    delete p;
    return hq::fail("hq_program_end: null handle pointer");
  }
  if (!p->host.empty()) {
    hipError_t err = hipMemcpy(p->dev, p->host.data(), p->host.size(), hipMemcpyHostToDevice);
    if (err != hipSuccess) {
      (void)hipFree(p->dev);
      delete p;
      return hq::fail(std::string("hq_program_end: hipMemcpy: ") + hipGetErrorString(err));
    }
  }
  p->host.clear();
  p->host.shrink_to_fit();
  p->finalized = true;
  *handle = p;
  return 0;
}

int hq_program_size(void* handle) {
  return handle ? (int)reinterpret_cast<hq::Program*>(handle)->ops.size() : -1;
}

int hq_program_run(void* handle) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::Program* p = reinterpret_cast<hq::Program*>(handle);
  if (!p || !p->finalized) return hq::fail("hq_program_run: invalid program");
  if (c.rec) return hq::fail("hq_program_run: cannot run while recording");
  if (!p->use_graph || p->ops.size() < 2) {
    for (auto& op : p->ops) op(c.stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hq::fail(std::string("hq_program_run: ") + hipGetErrorString(e));
    return 0;
  }
  // hipGraph replay.  The legacy default stream cannot be captured, so a null library stream
  // is replaced by a private BLOCKING stream: legacy-stream semantics order it with the work
  // around it on the default stream.
  hipStream_t s = c.stream;
  if (s == nullptr) {
    if (!p->graph_stream) {
      hipError_t e = hipStreamCreate(&p->graph_stream);
      if (e != hipSuccess) return hq::fail(std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    s = p->graph_stream;
  }
  if (!p->exec) {
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return hq::fail(std::string("hipStreamBeginCapture: ") + hipGetErrorString(e));
    for (auto& op : p->ops) op(s);
    e = hipStreamEndCapture(s, &graph);
    if (e != hipSuccess) return hq::fail(std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    e = hipGraphInstantiate(&p->exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return hq::fail(std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
  }
  hipError_t e = hipGraphLaunch(p->exec, s);
  if (e != hipSuccess) return hq::fail(std::string("hipGraphLaunch: ") + hipGetErrorString(e));
  return 0;
}

int hq_program_free(void* handle) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::Program* p = reinterpret_cast<hq::Program*>(handle);
  if (!p) return 0;
  if (c.rec == p) c.rec = nullptr;
  (void)hipStreamSynchronize(c.stream);
  if (p->graph_stream) {
    (void)hipStreamSynchronize(p->graph_stream);
    (void)hipStreamDestroy(p->graph_stream);
  }
  if (p->exec) (void)hipGraphExecDestroy(p->exec);
  if (p->dev) (void)hipFree(p->dev);
  delete p;
  return 0;
}

int hq_set_stream(void* hip_stream) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  if (s == c.stream) return 0;
  if (c.rec) return hq::fail("hq_set_stream: cannot change the stream while recording a program");
  // Work already enqueued on the old stream may still be reading the upload arena / scratch
  // buffers (they are recycled in issue order): the new stream waits for it on the DEVICE, the
  // host does not block.
  if (c.device >= 0) {
    hipEvent_t ev = nullptr;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ev, c.stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(s, ev, 0);
    if (ev) (void)hipEventDestroy(ev);
    if (e != hipSuccess) return hq::fail(std::string("hq_set_stream: ") + hipGetErrorString(e));
  }
  c.stream = s;
  return 0;
}

int hq_sync(void) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hipError_t e = hipStreamSynchronize(c.stream);
  if (e != hipSuccess) return hq::fail(std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
  return 0;
}

int hq_set_log2_pack_size(unsigned int v) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::read_env(c);
  if (v < 1 || v > 5) return hq::fail("log2_pack_size must be in 1..5");
  c.log2_pack = v;
  return 0;
}

const char* hq_last_error(void) { return hq::ctx().last_error.c_str(); }

int hq_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int hq_set_apply_mode(const char* name) {
  hq::Context& c = hq::ctx();
  std::lock_guard<std::mutex> lock(c.mu);
  hq::read_env(c);
  std::string s(name ? name : "");
  if (s == "auto") c.mode = hq::Mode::Auto;
  else if (s == "direct") c.mode = hq::Mode::Direct;
  else if (s == "mfma") c.mode = hq::Mode::Mfma;
  else if (s == "generic") c.mode = hq::Mode::Generic;
  else if (s == "naive") c.mode = hq::Mode::Naive;
  else if (s == "tile") c.mode = hq::Mode::Tile;
  else if (s == "gemm") c.mode = hq::Mode::Gemm;
  else if (s == "nt=1") c.nontemporal = 1;
  else if (s == "nt=0") c.nontemporal = 0;
  else if (s == "nt=auto") c.nontemporal = -1;
  else if (s == "dummy=comp") c.dummy_policy = 0;
  else if (s == "dummy=low") c.dummy_policy = 1;
  else if (s == "dummy=high") c.dummy_policy = 2;
  else if (s == "dummy=auto") c.dummy_policy = -1;
  else return hq::fail("unknown apply mode: " + s);
  return 0;
}

const char* hq_last_kernel(void) { return hq::ctx().last_kernel; }

const char* hq_last_kernel_desc(void) { return hq::ctx().last_desc.c_str(); }

}  // extern "C"

// Diagnostics (not part of the reference boundary): raw hipPointerGetAttributes result.
extern "C" int hq_pointer_info(const void* p, int* type, int* device, int* err) {
  hipPointerAttribute_t attr;
  memset(&attr, 0, sizeof(attr));
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (err) *err = (int)e;
  if (e != hipSuccess) (void)hipGetLastError();
  if (type) *type = (int)attr.type;
  if (device) *device = attr.device;
  return e == hipSuccess ? 0 : 1;
}
