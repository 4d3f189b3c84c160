// RCCL communicator management for the row-partitioned CG (one process per GPU, xGMI).
// The communicator is created from a unique id that rank 0 generates and the host broadcasts
// (torch.distributed in the Python wrapper); all data-path collectives are issued from
// operator.hip on the caller's stream.
#include <rccl/rccl.h>
#include <string.h>
#include "mgp_common.h"
#include "mgp_internal.h"

extern "C" int mgp_dist_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

extern "C" int mgp_dist_unique_id(void* id_out) {
  if (!id_out) return MGP_ERR_ARG;
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return 1000 + (int)r;
  memcpy(id_out, &id, sizeof(id));
  return MGP_OK;
}

extern "C" int mgp_dist_init(int rank, int world, const void* id_bytes, void** comm_out) {
  if (!id_bytes || !comm_out || world < 1 || rank < 0 || rank >= world) return MGP_ERR_ARG;
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  ncclComm_t comm;
  ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);
  if (r != ncclSuccess) return 1000 + (int)r;
  *comm_out = comm;
  return MGP_OK;
}

// What RCCL itself says about a communicator: its size, this process's rank in it and the HIP device it is bound to
// (ncclCommCount / ncclCommUserRank / ncclCommCuDevice).  bench.py gathers these from every rank into the N > 1 line
// (`rccl_ranks`), so that a scale record shows that the collectives ran over the communicator it claims.
extern "C" int mgp_dist_comm_info(void* comm, int32_t* count, int32_t* user_rank, int32_t* device) {
  if (!comm || !count || !user_rank || !device) return MGP_ERR_ARG;
  int c = 0, r = 0, d = 0;
  ncclResult_t e = ncclCommCount(static_cast<ncclComm_t>(comm), &c);
  if (e == ncclSuccess) e = ncclCommUserRank(static_cast<ncclComm_t>(comm), &r);
  if (e == ncclSuccess) e = ncclCommCuDevice(static_cast<ncclComm_t>(comm), &d);
  if (e != ncclSuccess) return 1000 + (int)e;
  *count = c; *user_rank = r; *device = d;
  return MGP_OK;
}

extern "C" int mgp_dist_destroy(void* comm) {
  if (!comm) return MGP_ERR_ARG;
  ncclResult_t r = ncclCommDestroy(static_cast<ncclComm_t>(comm));
  return r == ncclSuccess ? MGP_OK : 1000 + (int)r;
}

// in-place all-gather of a replicated buffer: rank p contributes floats [p*count, (p+1)*count)
extern "C" int mgp_dist_allgather(void* comm, int rank, int world, float* buf, int64_t count_per_rank, void* stream) {
  if (!comm || !buf || count_per_rank <= 0) return MGP_ERR_ARG;
  MgpDist d{comm, rank, world, count_per_rank, (int64_t)rank * count_per_rank};
  return mgp_dist_allgather_f32(&d, buf, count_per_rank, stream);
}

// Y = A X with the rows of A partitioned (op_local->L = rows of this rank, vectors global length)
extern "C" int mgp_operator_apply_part(const mgp_operator_t* op_local, void* comm, int rank, int world, const float* X,
                                       int C, float* Y, void* work, size_t work_bytes, void* stream) {
  if (!op_local || !comm) return MGP_ERR_ARG;
  MgpDist d{comm, rank, world, op_local->L.n, (int64_t)rank * op_local->L.n};
  return mgp_operator_apply_dist(op_local, &d, X, nullptr, C, Y, nullptr, nullptr, 0, nullptr, nullptr, work, work_bytes,
                                 stream);
}
