// Shared host-side plumbing of libtg_hip.so: status/error strings, launch checking, per-class profiling.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tg_kernels.h"

namespace tg {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

// kernel classes for tg_prof_* (keep in sync with kClassNames in runtime.cpp)
enum ProfClass {
  PC_IGEMM = 0,      // MFMA implicit-GEMM (conv fwd / dgrad / deconv / dense)
  PC_WGRAD,          // MFMA filter-gradient
  PC_PREP,           // weight-norm reparam, filter re-layout, slab reduce
  PC_NORM,           // mean-only BN / BN statistics + apply
  PC_ELEMWISE,       // activations, dropout, noise, concat, pooling
  PC_LOSS,           // loss heads
  PC_OPTIM,          // Adam / EMA
  PC_COUNT
};

struct ProfScope {
  ProfScope(int cls, double flops, double bytes, hipStream_t s, const char* desc = nullptr);
  ~ProfScope();
  int idx;
  hipStream_t stream;
};

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

#define TG_CHECK_LAUNCH(name)                                   \
  do {                                                          \
    hipError_t e__ = hipGetLastError();                         \
    if (e__ != hipSuccess) return tg::hip_fail(e__, name);      \
  } while (0)

#define TG_REQUIRE(cond, ...)                                   \
  do {                                                          \
    if (!(cond)) { tg::set_error(__VA_ARGS__); return TG_ERR_INVALID; } \
  } while (0)

}  // namespace tg
