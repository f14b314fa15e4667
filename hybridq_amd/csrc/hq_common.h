// hq_common.h -- what the translation units of libhq_hip.so share: the process-wide context (stream, kernel
// selection, upload arena, scratch buffers, the program being recorded), error reporting and the launch macros.
// The C ABI (include/hq_hip.h) is implemented by
//   hq_core.hip   context, streams, kernel selection, compiled programs
//   hq_apply.hip  apply_U_* and hq_apply_blocked_*     (reference: include/python_U.cpp:33-112, 131-143)
//   hq_swap.hip   swap_*, hq_permute_bits_*, to_complex (reference: include/python_swap.cpp:31-99, python_U.cpp:114-153)
//   hq_shard.hip  hq_shard_* / hq_exchange_* / hq_ipc_* (no reference counterpart)
//   hq_state.hip  state memory, initial states, reductions, Measure / Projection device side
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/hq_hip.h"
#include "hq_kernels_common.h"

namespace hq {

enum class Mode { Auto, Direct, Mfma, Generic, Naive, Tile, Gemm };

// A recorded sequence of launches ("compiled circuit"): every matrix / operand table it needs
// lives in its own device buffer, so replaying it is pure kernel launches -- from a plain loop
// or, after the first run, from one hipGraph launch.
struct Program {
  std::vector<std::function<void(hipStream_t)>> ops;
  std::vector<unsigned char> host;  // staged tables, uploaded once by hq_program_end
  unsigned char* dev = nullptr;
  size_t cap = 0;
  bool finalized = false;
  bool use_graph = true;
  hipGraphExec_t exec = nullptr;
  hipStream_t graph_stream = nullptr;  // capture needs a non-default stream
};

struct Context {
  std::mutex mu;
  Program* rec = nullptr;  // non-null while hq_program_begin .. hq_program_end records
  hipStream_t stream = nullptr;
  unsigned log2_pack = 1;
  Mode mode = Mode::Auto;
  int nontemporal = -1;  // -1 auto, 0 never, 1 always
  int dummy_policy = -1;  // mfma identity dummies: 0 = free vector components first, 1 = lowest free bits >= 2, 2 = free bits >= 6, -1 = auto (= 2)
  std::string last_error = "";
  const char* last_kernel = "none";
  std::string last_desc = "none";  // full instantiation name of the last apply_U kernel
  // arena for matrices that do not fit kernel arguments (generic / naive kernels)
  unsigned char* arena_host = nullptr;  // pinned
  unsigned char* arena_dev = nullptr;
  size_t arena_size = 0, arena_used = 0;
  // scratch (host staging of planes, naive / swap_gather temporaries, norm2)
  void* scratch[3] = {nullptr, nullptr, nullptr};
  size_t scratch_size[3] = {0, 0, 0};
  bool attr_set = false;
  bool env_read = false;
  int device = -1;  // device that owns the arena / scratch buffers (one device per process)
};

Context& ctx();
int fail(const std::string& msg);
// integer switch from the environment (experiment knobs: HQ_*), `dflt` when unset
inline int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
void read_env(Context& c);
// The library keeps its upload arena and scratch buffers on ONE device (one process per GPU); a second
// device in the same process is refused loudly.
int check_device(Context& c);
int get_scratch(Context& c, int slot, size_t bytes, void** out);
// Copy `bytes` of host data into the arena (or the program being recorded); device address in *dev.
int arena_upload(Context& c, const void* host, size_t bytes, void** dev);
// true if `p` can be dereferenced by a kernel
bool is_device_pointer(const void* p);
int check_positions(const unsigned* pos, unsigned n, unsigned k);
// apply_U on device planes with the context lock HELD (the state allocator probes placements with it)
int apply_device_f32(Context& c, float* re, float* im, const float* U, const unsigned* pos, unsigned n, unsigned k);
int apply_device_f64(Context& c, double* re, double* im, const double* U, const unsigned* pos, unsigned n, unsigned k);

}  // namespace hq

// A failed runtime call leaves its code in the thread's "last error"; it is cleared here so that the launch checks
// (HQ_HIP_CHECK(hipGetLastError()) after a kernel launch) of LATER calls do not report it again -- callers do carry on
// after a failure (the tuned placement falling back to hipMalloc / torch memory, a transport falling back).  Found by
// running the -m gpu tests against the host emulation (tests/emu), where the VMM calls fail by construction.
#define HQ_HIP_CHECK(expr)                                                             \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      (void)hipGetLastError();                                                         \
      return hq::fail(std::string(#expr) + ": " + hipGetErrorString(_e));              \
    }                                                                                  \
  } while (0)

// Launch now, or append to the program being recorded (arguments are captured by value).
#define HQ_LAUNCH(c_, kern, grid, block, lds, ...)                                              \
  do {                                                                                          \
    if ((c_).rec) {                                                                             \
      (c_).rec->ops.emplace_back(                                                               \
          [=](hipStream_t s_) { hipLaunchKernelGGL(kern, grid, block, lds, s_, __VA_ARGS__); });  \
    } else {                                                                                    \
      hipLaunchKernelGGL(kern, grid, block, lds, (c_).stream, __VA_ARGS__);                     \
    }                                                                                           \
  } while (0)

#define HQ_NOT_RECORDABLE(c_, what)                                                             \
  do {                                                                                          \
    if ((c_).rec) return hq::fail(std::string(what) + " cannot be recorded into a program");    \
  } while (0)
