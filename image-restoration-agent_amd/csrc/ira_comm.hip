// RCCL (over xGMI) for the two multi-GPU paths of the library: the all-gather of restored tiles (demo.py:145-153 spread over
// ranks) and the halo-row / Gram-record exchange of the band-parallel whole frame (8f-2).  Host code only.  See ira_comm.h.
#include "ira_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

namespace ira_comm {

namespace {

thread_local std::string t_err;

struct Api {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;                 // optional: a library without it only loses the abort path
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr; // optional
    std::string why;
};

Api* api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        // IRA_RCCL_LIB names the one library to try (a site build of RCCL; also how the tests provoke the failure path)
        const char* forced = getenv("IRA_RCCL_LIB");
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        std::string first_err;
        auto try_open = [&](const char* n) {
            dlerror();
            a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (!a.lib && first_err.empty()) {
                const char* e = dlerror();          // ONE call: a second dlerror() returns NULL and invalidates the first string
                first_err = e ? e : "?";
            }
        };
        if (forced && *forced) try_open(forced);
        else for (const char* n : names) { try_open(n); if (a.lib) break; }
        if (!a.lib) { a.why = std::string("cannot load librccl.so: ") + first_err; return; }
#define IRA_SYM(field, name)                                                                          \
    a.field = (decltype(a.field))dlsym(a.lib, name);                                                   \
    if (!a.field) { a.why = std::string("librccl.so lacks ") + name; return; }
        IRA_SYM(GetUniqueId, "ncclGetUniqueId") IRA_SYM(CommInitRank, "ncclCommInitRank") IRA_SYM(CommDestroy, "ncclCommDestroy")
        IRA_SYM(AllGather, "ncclAllGather") IRA_SYM(Send, "ncclSend") IRA_SYM(Recv, "ncclRecv") IRA_SYM(GroupStart, "ncclGroupStart")
        IRA_SYM(GroupEnd, "ncclGroupEnd") IRA_SYM(GetErrorString, "ncclGetErrorString")
#undef IRA_SYM
        a.CommAbort = (decltype(a.CommAbort))dlsym(a.lib, "ncclCommAbort");
        a.CommGetAsyncError = (decltype(a.CommGetAsyncError))dlsym(a.lib, "ncclCommGetAsyncError");
    });
    if (!a.why.empty()) { t_err = a.why; return nullptr; }
    return &a;
}

bool ok(Api* a, ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return true;
    t_err = std::string(what) + ": " + a->GetErrorString(r);
    return false;
}

}  // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

const char* last_error() { return t_err.c_str(); }

bool get_unique_id(void* id128) {
    Api* a = api();
    if (!a) return false;
    static_assert(sizeof(ncclUniqueId) == 128, "ira.h promises a 128-byte id");
    ncclUniqueId id;
    if (!ok(a, a->GetUniqueId(&id), "ncclGetUniqueId")) return false;
    memcpy(id128, &id, sizeof id);
    return true;
}

Comm* init_rank(const void* id128, int rank, int world) {
    Api* a = api();
    if (!a) return nullptr;
    if (world < 1 || rank < 0 || rank >= world) { t_err = "ira_comm: need 0 <= rank < world"; return nullptr; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    Comm* c = new Comm();
    c->rank = rank; c->world = world;
    if (!ok(a, a->CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank")) { delete c; return nullptr; }
    return c;
}

void destroy(Comm* c) {
    if (!c) return;
    if (Api* a = api()) if (c->comm) a->CommDestroy(c->comm);
    delete c;
}

void abort(Comm* c) {
    if (!c || !c->comm) return;
    Api* a = api();
    if (a && a->CommAbort) a->CommAbort(c->comm);           // frees the communicator: its kernels see the abort flag and leave
    else if (a) a->CommDestroy(c->comm);
    c->comm = nullptr;
}

bool usable(const Comm* c) { return c && c->comm; }

// Host-side wait for everything enqueued on `st`, the collectives included.  A peer that never enters a collective (it failed an
// argument check, ran out of memory, died) would otherwise leave this rank's RCCL kernel spinning for ever and hipStreamSynchronize
// with it: poll the stream and the communicator's asynchronous error state instead, and after `timeout_ms` (or on an asynchronous
// error) abort the communicator, which makes the local kernel exit, and report.  The communicator is unusable afterwards.
bool sync(Comm* c, hipStream_t st, long timeout_ms) {
    Api* a = api();
    if (!a || !usable(c)) { if (a) t_err = "communicator was aborted"; return false; }
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    for (;;) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return true;
        if (q != hipErrorNotReady) { t_err = std::string("stream: ") + hipGetErrorString(q); abort(c); return false; }
        if (a->CommGetAsyncError) {
            ncclResult_t ar = ncclSuccess;
            if (a->CommGetAsyncError(c->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
                t_err = std::string("asynchronous RCCL error: ") + a->GetErrorString(ar);
                abort(c);
                (void)hipStreamSynchronize(st);
                return false;
            }
        }
        const long ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (timeout_ms > 0 && ms > timeout_ms) {
            t_err = "collective did not complete within " + std::to_string(timeout_ms) + " ms (a peer rank failed or never entered it); communicator aborted";
            abort(c);
            (void)hipStreamSynchronize(st);
            return false;
        }
        if (++spins < 2000) std::this_thread::yield();     // the first ~ms: a frame normally completes here
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

int rank(const Comm* c) { return c->rank; }
int world(const Comm* c) { return c->world; }

bool all_gather(Comm* c, const void* send, void* recv, size_t bytes_per_rank, hipStream_t st) {
    Api* a = api();
    if (!a) return false;
    if (!usable(c)) { t_err = "communicator was aborted"; return false; }
    return ok(a, a->AllGather(send, recv, bytes_per_rank, ncclUint8, c->comm, st), "ncclAllGather");
}

bool exchange_neighbours(Comm* c, int up, const void* send_up, void* recv_up, int down, const void* send_down, void* recv_down, size_t bytes,
                         hipStream_t st) {
    Api* a = api();
    if (!a) return false;
    if (up < 0 && down < 0) return true;
    if (!usable(c)) { t_err = "communicator was aborted"; return false; }
    if (!ok(a, a->GroupStart(), "ncclGroupStart")) return false;
    bool good = true;
    if (up >= 0) {
        good = good && ok(a, a->Send(send_up, bytes, ncclUint8, up, c->comm, st), "ncclSend(up)");
        good = good && ok(a, a->Recv(recv_up, bytes, ncclUint8, up, c->comm, st), "ncclRecv(up)");
    }
    if (down >= 0) {
        good = good && ok(a, a->Send(send_down, bytes, ncclUint8, down, c->comm, st), "ncclSend(down)");
        good = good && ok(a, a->Recv(recv_down, bytes, ncclUint8, down, c->comm, st), "ncclRecv(down)");
    }
    const bool ended = ok(a, a->GroupEnd(), "ncclGroupEnd");
    return good && ended;
}

}  // namespace ira_comm
