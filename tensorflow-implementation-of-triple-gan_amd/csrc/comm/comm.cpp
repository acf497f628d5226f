// libtg_comm.so — RCCL behind include/tg_comm.h.  Host code only (no kernels): every entry point is one RCCL call on the caller's
// stream.  Built separately from libtg_hip.so (csrc/Makefile target libtg_comm.so) and loaded on demand by tg/comm.py.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../../include/tg_comm.h"

static_assert(TG_COMM_ID_BYTES == sizeof(ncclUniqueId), "TG_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

namespace {
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

struct Comm {
  ncclComm_t nccl;
  int nranks, rank, device;
};

#define NCCL_TRY(expr)                                                                   \
  do {                                                                                   \
    ncclResult_t r_ = (expr);                                                            \
    if (r_ != ncclSuccess) return fail(-2, "%s: %s", #expr, ncclGetErrorString(r_));     \
  } while (0)
}  // namespace

extern "C" {

const char* tg_comm_last_error_string(void) { return g_err; }

int tg_comm_unique_id(void* id) {
  if (!id) return fail(-1, "tg_comm_unique_id: null id");
  ncclUniqueId u;
  NCCL_TRY(ncclGetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return 0;
}

int tg_comm_init_rank(void** comm, int nranks, const void* id, int rank, int device) {
  if (!comm || !id) return fail(-1, "tg_comm_init_rank: null argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(-1, "tg_comm_init_rank: rank %d of %d", rank, nranks);
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(-2, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  Comm* c = new Comm{nullptr, nranks, rank, device};
  ncclResult_t r = ncclCommInitRank(&c->nccl, nranks, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(-2, "ncclCommInitRank: %s", ncclGetErrorString(r));
  }
  *comm = c;
  return 0;
}

int tg_comm_count(void* comm, int* nranks, int* rank) {
  if (!comm) return fail(-1, "tg_comm_count: null communicator");
  Comm* c = static_cast<Comm*>(comm);
  if (nranks) *nranks = c->nranks;
  if (rank) *rank = c->rank;
  return 0;
}

int tg_allreduce_sum_f32(void* buf, int64_t count, void* comm, void* stream) {
  if (!comm || (!buf && count)) return fail(-1, "tg_allreduce_sum_f32: null argument");
  if (count < 0) return fail(-1, "tg_allreduce_sum_f32: count %lld", (long long)count);
  if (count == 0) return 0;
  NCCL_TRY(ncclAllReduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, static_cast<Comm*>(comm)->nccl, (hipStream_t)stream));
  return 0;
}

int tg_allreduce_max_f64(void* buf, int64_t count, void* comm, void* stream) {
  if (!comm || (!buf && count)) return fail(-1, "tg_allreduce_max_f64: null argument");
  if (count < 0) return fail(-1, "tg_allreduce_max_f64: count %lld", (long long)count);
  if (count == 0) return 0;
  NCCL_TRY(ncclAllReduce(buf, buf, (size_t)count, ncclFloat64, ncclMax, static_cast<Comm*>(comm)->nccl, (hipStream_t)stream));
  return 0;
}

int tg_broadcast_f32(void* buf, int64_t count, int root, void* comm, void* stream) {
  if (!comm || (!buf && count)) return fail(-1, "tg_broadcast_f32: null argument");
  Comm* c = static_cast<Comm*>(comm);
  if (count < 0 || root < 0 || root >= c->nranks) return fail(-1, "tg_broadcast_f32: count %lld root %d", (long long)count, root);
  if (count == 0) return 0;
  NCCL_TRY(ncclBroadcast(buf, buf, (size_t)count, ncclFloat32, root, c->nccl, (hipStream_t)stream));
  return 0;
}

int tg_comm_destroy(void* comm) {
  if (!comm) return 0;
  Comm* c = static_cast<Comm*>(comm);
  ncclResult_t r = ncclCommDestroy(c->nccl);
  delete c;
  if (r != ncclSuccess) return fail(-2, "ncclCommDestroy: %s", ncclGetErrorString(r));
  return 0;
}

}  // extern "C"
