// Launch plans (include/tg_plan.h): a recorded list of entry-point launches and stream-ordering events, re-issued from one C loop.
#include <cstring>
#include <deque>
#include <string>
#include <vector>
#include "tg_common.h"
#include "../../include/tg_plan.h"

namespace {

typedef int (*Thunk)(const tg_plan_word*, void*);
struct ThunkEntry { const char* name; Thunk fn; int n_args; const char* kinds; };

#include "plan_thunks.inc"

const ThunkEntry* find_thunk(const char* name) {
  for (const ThunkEntry& t : kThunks)
    if (std::strcmp(t.name, name) == 0) return &t;
  return nullptr;
}

enum OpKind { OP_LAUNCH = 0, OP_RECORD = 1, OP_WAIT = 2 };

struct Op {
  int kind;
  int slot;
  const ThunkEntry* thunk;     // OP_LAUNCH
  uint32_t arg0;               // first argument word in Plan::words
  hipEvent_t event;            // OP_RECORD / OP_WAIT
};

struct Plan {
  std::vector<Op> ops;
  std::vector<tg_plan_word> words;
  std::deque<std::vector<uint64_t>> held;      // 8-byte words: host data copied by tg_plan_hold (vector storage is 16-byte aligned by the allocator)
  int64_t n_launch = 0;
  int max_slot = -1;
};

Plan* as_plan(void* p) { return static_cast<Plan*>(p); }

}  // namespace

using namespace tg;

extern "C" {

int tg_plan_create(void** plan_out) {
  TG_REQUIRE(plan_out != nullptr, "tg_plan_create: null output");
  *plan_out = new (std::nothrow) Plan();
  TG_REQUIRE(*plan_out != nullptr, "tg_plan_create: out of host memory");
  return TG_OK;
}

int tg_plan_destroy(void* plan) {
  delete as_plan(plan);
  return TG_OK;
}

int tg_plan_hold(void* plan, const void* host_data, int64_t bytes, void** held_out) {
  TG_REQUIRE(plan && host_data && held_out && bytes > 0, "tg_plan_hold: bad argument");
  Plan* P = as_plan(plan);
  P->held.emplace_back((size_t)(bytes + 15) / 16 * 2);
  std::memcpy(P->held.back().data(), host_data, (size_t)bytes);
  *held_out = P->held.back().data();
  return TG_OK;
}

int tg_plan_add_launch(void* plan, const char* entry, const tg_plan_word* args, int n_args, int stream_slot) {
  TG_REQUIRE(plan && entry, "tg_plan_add_launch: bad argument");
  const ThunkEntry* t = find_thunk(entry);
  TG_REQUIRE(t != nullptr, "tg_plan_add_launch: %s is not a launch entry point of tg_kernels.h", entry);
  TG_REQUIRE(n_args == t->n_args, "tg_plan_add_launch: %s takes %d arguments before the stream, got %d", entry, t->n_args, n_args);
  TG_REQUIRE(stream_slot >= 0 && stream_slot < 16, "tg_plan_add_launch: stream slot %d out of range", stream_slot);
  TG_REQUIRE(n_args == 0 || args != nullptr, "tg_plan_add_launch: null argument array");
  Plan* P = as_plan(plan);
  Op op{OP_LAUNCH, stream_slot, t, (uint32_t)P->words.size(), nullptr};
  P->words.insert(P->words.end(), args, args + n_args);
  P->ops.push_back(op);
  P->n_launch++;
  if (stream_slot > P->max_slot) P->max_slot = stream_slot;
  return TG_OK;
}

static int add_event_op(void* plan, int kind, void* event, int slot, const char* what) {
  TG_REQUIRE(plan && event, "%s: bad argument", what);
  TG_REQUIRE(slot >= 0 && slot < 16, "%s: stream slot %d out of range", what, slot);
  Plan* P = as_plan(plan);
  P->ops.push_back(Op{kind, slot, nullptr, 0, reinterpret_cast<hipEvent_t>(event)});
  if (slot > P->max_slot) P->max_slot = slot;
  return TG_OK;
}

int tg_plan_add_event_record(void* plan, void* event, int stream_slot) { return add_event_op(plan, OP_RECORD, event, stream_slot, "tg_plan_add_event_record"); }
int tg_plan_add_stream_wait(void* plan, int stream_slot, void* event) { return add_event_op(plan, OP_WAIT, event, stream_slot, "tg_plan_add_stream_wait"); }

int64_t tg_plan_length(const void* plan) { return plan ? (int64_t)static_cast<const Plan*>(plan)->ops.size() : 0; }
int64_t tg_plan_launches(const void* plan) { return plan ? static_cast<const Plan*>(plan)->n_launch : 0; }

const char* tg_plan_signature(const char* entry) {
  const ThunkEntry* t = entry ? find_thunk(entry) : nullptr;
  return t ? t->kinds : nullptr;
}

int tg_plan_replay(void* plan, void* const* streams, int n_streams) {
  TG_REQUIRE(plan && streams, "tg_plan_replay: bad argument");
  Plan* P = as_plan(plan);
  TG_REQUIRE(n_streams > P->max_slot, "tg_plan_replay: the plan uses stream slot %d, %d stream(s) given", P->max_slot, n_streams);
  const tg_plan_word* words = P->words.data();
  const size_t n = P->ops.size();
  for (size_t k = 0; k < n; ++k) {
    const Op& op = P->ops[k];
    if (op.kind == OP_LAUNCH) {
      int rc = op.thunk->fn(words + op.arg0, streams[op.slot]);
      if (rc != TG_OK) {
        std::string inner = tg_last_error_string();
        set_error("tg_plan_replay: operation %zu (%s) failed: %s", k, op.thunk->name, inner.c_str());
        return rc;
      }
    } else if (op.kind == OP_RECORD) {
      hipError_t e = hipEventRecord(op.event, as_stream(streams[op.slot]));
      if (e != hipSuccess) { hip_fail(e, "hipEventRecord"); std::string inner = tg_last_error_string(); set_error("tg_plan_replay: operation %zu: %s", k, inner.c_str()); return TG_ERR_HIP; }
    } else {
      hipError_t e = hipStreamWaitEvent(as_stream(streams[op.slot]), op.event, 0);
      if (e != hipSuccess) { hip_fail(e, "hipStreamWaitEvent"); std::string inner = tg_last_error_string(); set_error("tg_plan_replay: operation %zu: %s", k, inner.c_str()); return TG_ERR_HIP; }
    }
  }
  return TG_OK;
}

}  // extern "C"
