// RCCL behind the C ABI (SURVEY 8 rows e and f2): librccl.so is dlopen-ed on the first ira_comm_* call, so a
// single-GPU process never loads it and libira_hip.so has no link-time dependency on it.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace ira_comm {

struct Comm;                                    // one RCCL communicator bound to one model handle's device and stream

const char* last_error();                       // of the calling thread
bool get_unique_id(void* id128);                // ncclGetUniqueId
Comm* init_rank(const void* id128, int rank, int world);       // ncclCommInitRank on the CURRENT device
void destroy(Comm* c);
void abort(Comm* c);                            // ncclCommAbort: local collectives in flight end with an error; the Comm stays allocated but unusable
bool usable(const Comm* c);                     // false after abort()
// wait for `st` (collectives included) with a deadline; on time-out or an asynchronous RCCL error: abort(c), false, last_error() says why
bool sync(Comm* c, hipStream_t st, long timeout_ms);
int rank(const Comm* c);
int world(const Comm* c);
// every call enqueues on `st` and returns at once (no host synchronisation)
bool all_gather(Comm* c, const void* send, void* recv, size_t bytes_per_rank, hipStream_t st);        // in place allowed (send == recv + rank * bytes)
// halo rows with the two neighbours of a band: one grouped send/recv pair per existing neighbour (peer < 0: none)
bool exchange_neighbours(Comm* c, int up, const void* send_up, void* recv_up, int down, const void* send_down, void* recv_down, size_t bytes,
                         hipStream_t st);

}  // namespace ira_comm
