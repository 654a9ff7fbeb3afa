// comm.hpp -- transport of the row-block partitioned solver.  One process per GPU in production
// (RcclComm: RCCL over xGMI, bootstrapped from a 128-byte unique id any side channel can carry,
// e.g. a torch.distributed broadcast); ThreadComm runs G virtual ranks inside one process on one
// GPU for tests; SelfComm is the single-GPU no-op.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "dist.hpp"

namespace sparsh {

// device-side image of a HaloPlan
struct DevPlan {
    int nloc = 0, nhalo = 0, nsend = 0;
    int *send_idx = nullptr;   // device, nsend
    double *sendbuf = nullptr; // device, nsend
    std::vector<HaloSeg> recv, send;
    bool need_pack = false;  // some send segment is not a contiguous run of the vector
    bool empty() const { return nhalo == 0 && nsend == 0; }
};

// device-side image of a DeepPlan (dist.hpp): ghost layers travel through staging buffers, one contiguous
// segment per peer each way; the caller scatters recvbuf to the ghost positions afterwards
struct DevDeepPlan {
    int depth = 0, nrecv = 0, nsend = 0;
    std::vector<HaloSeg> recv, send;   // off / cnt into recvbuf / sendbuf
    int *send_idx = nullptr;           // device, nsend: own local indices to pack
    int *recv_pos = nullptr;           // device, nrecv: local index of every staged entry
    double *sendbuf = nullptr, *recvbuf = nullptr;
    bool empty() const { return nrecv == 0 && nsend == 0; }
};

class Comm {
public:
    virtual ~Comm() = default;
    int rank = 0, size = 1;
    // vec holds nloc own entries followed by room for nhalo received ones.  Packs the non-contiguous
    // send segments into sendbuf (contiguous ones, start >= 0, go out of vec in place), then moves
    // the data.  Everything is enqueued on st; on return nothing has necessarily happened yet.
    // When the work queued on st up to here has completed, this rank's halo has landed AND every
    // peer has finished reading this rank's boundary entries (send + receive semantics).
    virtual bool exchange(const DevPlan &p, double *vec, hipStream_t st) = 0;
    // deep-halo exchange: packs src[send_idx] into sendbuf and moves every peer's segment into this rank's recvbuf
    // (same completion semantics as exchange()).  The caller unpacks recvbuf.
    virtual bool exchange_staged(const DevDeepPlan &p, const double *src, hipStream_t st) = 0;
    // in-place sum over ranks of n doubles in device memory
    virtual bool allreduce_sum(double *dev, int n, hipStream_t st) = 0;
    // every rank contributes full[lo(rank) .. hi(rank)) of `part`; afterwards all ranks hold all of it
    virtual bool allgather(double *full, const Partition &part, hipStream_t st) = 0;
    virtual bool barrier(hipStream_t st) = 0;
    // bytes of device memory from rank `root` to the same buffer on every rank (setup: the hierarchy image)
    virtual bool bcast(void *dev, size_t bytes, int root, hipStream_t st) = 0;
    std::string error;
};

std::unique_ptr<Comm> make_self_comm();

// ---- RCCL ----
bool rccl_unique_id(char out[128], std::string &err);
std::unique_ptr<Comm> make_rccl_comm(const char id[128], int rank, int nranks, std::string &err);

// ---- in-process thread group (tests) ----
struct ThreadGroup;
ThreadGroup *thread_group_create(int nranks);
void thread_group_destroy(ThreadGroup *g);
void thread_group_fail_after(ThreadGroup *g, int n);  // test hook: exchanges from call n on fail on every rank
void thread_group_set_delay(ThreadGroup *g, double microseconds);  // test hook: every transport call costs this much more (a slow link)
std::unique_ptr<Comm> make_thread_comm(ThreadGroup *g, int rank);

}  // namespace sparsh
