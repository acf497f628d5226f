/* tg_comm.h — C ABI of libtg_comm.so: the data-parallel exchange of the Triple-GAN step (SURVEY.md §8b "comm", §8e) on RCCL.
 *
 * The reference trains in one process (Training/Train_goodGAN.py:266-276: d_solver, g_solver, c_solver on one device); with
 * R replicas each solver's flat gradient buffer is summed over the replicas between `compute_gradients` and `apply_gradients`
 * (Training/train_base.py:64-68, :91-97) — that sum is tg_allreduce_sum_f32.  libtg_comm.so is separate from libtg_hip.so so
 * that a single-GPU user never loads RCCL; the package's default exchange runs through torch.distributed's RCCL process group
 * (tg/dist.py) and TG_DIST_BACKEND=rccl-direct selects this library instead.
 *
 * Contract: as tg_kernels.h — plain pointers, caller-owned device buffers, asynchronous on `stream`, status codes (0 or < 0),
 * tg_comm_last_error_string() thread-local.  Collectives are in place and legal inside hipStream capture (RCCL records them into
 * the graph).  One communicator per process, bound to the device that is current when tg_comm_init_rank is called.
 */
#ifndef TG_COMM_H
#define TG_COMM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_COMM_ID_BYTES 128

const char* tg_comm_last_error_string(void);
/* host: fill id[TG_COMM_ID_BYTES] on ONE rank; the caller hands the bytes to the other ranks (file, socket, TCPStore ...). */
int tg_comm_unique_id(void* id);
/* collective over all ranks: join the communicator `id` as `rank` of `nranks` on HIP device `device`. */
int tg_comm_init_rank(void** comm, int nranks, const void* id, int rank, int device);
int tg_comm_count(void* comm, int* nranks, int* rank);
/* buf[i] <- sum over ranks of buf[i]   (gradient exchange; fp32, in place) */
int tg_allreduce_sum_f32(void* buf, int64_t count, void* comm, void* stream);
/* buf[i] <- max over ranks of buf[i]   (fp64; the benchmark's max-over-ranks time) */
int tg_allreduce_max_f64(void* buf, int64_t count, void* comm, void* stream);
/* buf <- root's buf   (identical initial weights on every replica) */
int tg_broadcast_f32(void* buf, int64_t count, int root, void* comm, void* stream);
int tg_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif
