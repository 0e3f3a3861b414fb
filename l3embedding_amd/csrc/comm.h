// comm.h -- RCCL communicator behind libl3hip.so's l3_comm_* entry points (one rank = one process = one GPU).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

namespace l3 {

struct Comm;   // ncclComm_t + the HIP stream its collectives run on

constexpr int COMM_ID_BYTES = 128;   // == NCCL_UNIQUE_ID_BYTES

// rank 0: a fresh ncclUniqueId (the caller ships the bytes to the other ranks)
int comm_unique_id(void* id128, std::string* err);
// ncclCommInitRank on `device`; collective over all `world` ranks
int comm_create(const void* id128, int world, int rank, int device, Comm** out, std::string* err);
void comm_destroy(Comm* c);
int comm_world(const Comm* c);
int comm_rank(const Comm* c);
hipStream_t comm_stream(const Comm* c);
// in-place all-reduce on the communicator's stream (asynchronous); op: 0 = sum, 1 = max
int comm_allreduce_f32(Comm* c, float* buf, size_t n, int op, std::string* err);
int comm_allreduce_f64(Comm* c, double* buf, size_t n, int op, std::string* err);
// recv[r * n_per_rank ...] = rank r's send[0 .. n_per_rank) on every rank (asynchronous, communicator stream)
int comm_allgather_f32(Comm* c, const float* send, float* recv, size_t n_per_rank, std::string* err);
int comm_version();                  // ncclGetVersion code of the bound library (0 before the first use)
const char* comm_library_path();     // which librccl was bound ("" before the first use)

}  // namespace l3
