/*
 * tg_plan.h — C ABI of the launch-plan recorder / replayer of libtg_hip.so.
 *
 * What it replaces: the reference hands a whole solver run to the TensorFlow runtime in ONE call
 * (Training/Train_goodGAN.py:266-276: `sess.run([d_solver, d_loss])`, `sess.run([g_solver, g_loss])`,
 * `sess.run([c_solver, c_loss])` on one feed) and TF's executor walks the cached sub-graph natively.  Here a solver
 * run is ~100 kernel launches through include/tg_kernels.h whose arguments are fixed after the first iterations
 * (call-site buffers, device-resident hyper-parameters, counter-based RNG): a PLAN is the recorded list of those
 * launches (entry point + argument words + stream slot) and of the events that order two streams, and
 * tg_plan_replay re-issues it from one C loop — ordinary eager launches, so the second-stream overlap of the
 * filter gradients survives (a captured hipGraph with cross-stream edges replays slower on ROCm 7.2), without a
 * trip through the host language per launch.
 *
 * Contract (same as tg_kernels.h): plain pointers and sizes; the caller owns every DEVICE buffer a recorded launch
 * names and must keep it alive and in place while the plan is; HOST data a launch reads at issue time (descriptors,
 * segment tables, job arrays) is copied into the plan by tg_plan_hold and recorded by that address; replay makes no
 * allocation and no synchronisation; 0 or a negative tg_status, message in tg_last_error_string().
 * Not thread-safe per plan (one host thread drives one GPU).
 */
#ifndef TG_PLAN_H
#define TG_PLAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* one argument of a recorded launch: pointers in .p, every integer type in .i, float in .f */
typedef union tg_plan_word {
  void* p;
  int64_t i;
  float f;
} tg_plan_word;

int tg_plan_create(void** plan_out);
int tg_plan_destroy(void* plan);

/* copy `bytes` of host data into storage owned by the plan (16-byte aligned, stable until tg_plan_destroy);
 * *held_out is the address to record in place of `host_data`. */
int tg_plan_hold(void* plan, const void* host_data, int64_t bytes, void** held_out);

/* append a launch of entry point `entry` (a name declared in tg_kernels.h whose last parameter is `void* stream`):
 * args[0..n_args) are its parameters in order WITHOUT the stream; the launch goes to streams[stream_slot] of the replay.
 * TG_ERR_INVALID: unknown entry point, wrong argument count, slot out of range. */
int tg_plan_add_launch(void* plan, const char* entry, const tg_plan_word* args, int n_args, int stream_slot);

/* append hipEventRecord(event, streams[stream_slot]) / hipStreamWaitEvent(streams[stream_slot], event);
 * `event` is a hipEvent_t owned by the caller. */
int tg_plan_add_event_record(void* plan, void* event, int stream_slot);
int tg_plan_add_stream_wait(void* plan, int stream_slot, void* event);

/* recorded operations (launches + event operations) / launches only */
int64_t tg_plan_length(const void* plan);
int64_t tg_plan_launches(const void* plan);

/* "p" / "i" / "f" per parameter of a launch entry point (without the stream), or NULL if `entry` is not one */
const char* tg_plan_signature(const char* entry);

/* re-issue everything in recorded order; streams[slot] are hipStream_t.  Stops at the first failing operation
 * (its index and entry point are in the error string). */
int tg_plan_replay(void* plan, void* const* streams, int n_streams);

#ifdef __cplusplus
}
#endif
#endif
