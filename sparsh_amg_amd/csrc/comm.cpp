// comm.cpp -- Comm backends (see comm.hpp).
#include "comm.hpp"

#include <condition_variable>
#include <cstring>
#include <mutex>

#include <rccl/rccl.h>

namespace sparsh {

// ------------------------------------------------------------------ single rank
namespace {
class SelfComm : public Comm {
public:
    bool exchange(const DevPlan &, double *, hipStream_t) override { return true; }
    bool allreduce_sum(double *, int, hipStream_t) override { return true; }
    bool allgather(double *, const Partition &, hipStream_t) override { return true; }
    bool barrier(hipStream_t) override { return true; }
};
}  // namespace

std::unique_ptr<Comm> make_self_comm() { return std::unique_ptr<Comm>(new SelfComm()); }

// ------------------------------------------------------------------ RCCL
namespace {
class RcclComm : public Comm {
public:
    ncclComm_t comm = nullptr;
    ~RcclComm() override
    {
        if (comm) ncclCommDestroy(comm);
    }
    bool ok(ncclResult_t r, const char *what)
    {
        if (r == ncclSuccess) return true;
        error = std::string(what) + ": " + ncclGetErrorString(r);
        return false;
    }
    // neighbour halo exchange: one grouped send/recv per peer pair, all on the caller's stream
    bool exchange(const DevPlan &p, double *vec, hipStream_t st) override
    {
        if (p.recv.empty() && p.send.empty()) return true;
        if (!ok(ncclGroupStart(), "ncclGroupStart")) return false;
        for (const HaloSeg &s : p.send) {
            const double *src = s.start >= 0 ? vec + s.start : p.sendbuf + s.off;
            if (!ok(ncclSend(src, (size_t)s.cnt, ncclDouble, s.peer, comm, st), "ncclSend")) return false;
        }
        for (const HaloSeg &r : p.recv)
            if (!ok(ncclRecv(vec + p.nloc + r.off, (size_t)r.cnt, ncclDouble, r.peer, comm, st), "ncclRecv")) return false;
        return ok(ncclGroupEnd(), "ncclGroupEnd");
    }
    bool allreduce_sum(double *dev, int n, hipStream_t st) override
    {
        return ok(ncclAllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum, comm, st), "ncclAllReduce");
    }
    bool allgather(double *full, const Partition &part, hipStream_t st) override
    {
        // ranges differ in length: one broadcast per owner, grouped
        if (!ok(ncclGroupStart(), "ncclGroupStart")) return false;
        for (int r = 0; r < size; ++r) {
            const int lo = part.lo(r), cnt = part.hi(r) - lo;
            if (cnt <= 0) continue;
            if (!ok(ncclBroadcast(full + lo, full + lo, (size_t)cnt, ncclDouble, r, comm, st), "ncclBroadcast")) return false;
        }
        return ok(ncclGroupEnd(), "ncclGroupEnd");
    }
    bool barrier(hipStream_t st) override
    {
        if (!scratch && hipMalloc(reinterpret_cast<void **>(&scratch), sizeof(double)) != hipSuccess) return false;
        (void)hipMemsetAsync(scratch, 0, sizeof(double), st);
        if (!allreduce_sum(scratch, 1, st)) return false;
        return hipStreamSynchronize(st) == hipSuccess;
    }
    double *scratch = nullptr;
};
}  // namespace

bool rccl_unique_id(char out[128], std::string &err)
{
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) {
        err = std::string("ncclGetUniqueId: ") + ncclGetErrorString(r);
        return false;
    }
    std::memcpy(out, &id, 128);
    return true;
}

std::unique_ptr<Comm> make_rccl_comm(const char id128[128], int rank, int nranks, std::string &err)
{
    std::unique_ptr<RcclComm> c(new RcclComm());
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r);
        return nullptr;
    }
    c->rank = rank;
    c->size = nranks;
    return c;
}

// ------------------------------------------------------------------ thread group (tests)
struct ThreadGroup {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    std::vector<const DevPlan *> plans;
    std::vector<double *> vecs;
    std::vector<double *> ptrs;
    std::vector<std::vector<double>> host;
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        const long gen = generation;
        if (++arrived == n) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen; });
        }
    }
};

ThreadGroup *thread_group_create(int nranks)
{
    ThreadGroup *g = new ThreadGroup();
    g->n = nranks;
    g->plans.assign((size_t)nranks, nullptr);
    g->vecs.assign((size_t)nranks, nullptr);
    g->ptrs.assign((size_t)nranks, nullptr);
    g->host.resize((size_t)nranks);
    return g;
}

void thread_group_destroy(ThreadGroup *g) { delete g; }

namespace {
class ThreadComm : public Comm {
public:
    ThreadGroup *g = nullptr;
    bool exchange(const DevPlan &p, double *vec, hipStream_t st) override
    {
        // publish my packed send buffer, then pull my halo segments out of the peers' buffers
        (void)hipStreamSynchronize(st);
        g->plans[rank] = &p;
        g->vecs[rank] = vec;
        g->wait();
        for (const HaloSeg &r : p.recv) {
            const DevPlan *q = g->plans[r.peer];
            const HaloSeg *src = nullptr;
            for (const HaloSeg &s : q->send)
                if (s.peer == rank) src = &s;
            if (!src || src->cnt != r.cnt) {
                error = "thread comm: send/recv plans disagree";
                g->wait();
                return false;
            }
            const double *from = src->start >= 0 ? g->vecs[r.peer] + src->start : q->sendbuf + src->off;
            (void)hipMemcpyAsync(vec + p.nloc + r.off, from, (size_t)r.cnt * sizeof(double), hipMemcpyDeviceToDevice, st);
        }
        (void)hipStreamSynchronize(st);
        g->wait();
        return true;
    }
    bool allreduce_sum(double *dev, int n, hipStream_t st) override
    {
        std::vector<double> &mine = g->host[rank];
        mine.resize((size_t)n);
        (void)hipMemcpyAsync(mine.data(), dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        g->wait();
        std::vector<double> sum((size_t)n, 0.0);
        for (int r = 0; r < size; ++r)
            for (int k = 0; k < n; ++k) sum[k] += g->host[r][k];  // rank order: same result on every rank
        g->wait();
        (void)hipMemcpyAsync(dev, sum.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, st);
        (void)hipStreamSynchronize(st);
        return true;
    }
    bool allgather(double *full, const Partition &part, hipStream_t st) override
    {
        (void)hipStreamSynchronize(st);
        g->ptrs[rank] = full;
        g->wait();
        for (int r = 0; r < size; ++r) {
            if (r == rank) continue;
            const int lo = part.lo(r), cnt = part.hi(r) - lo;
            if (cnt > 0) (void)hipMemcpyAsync(full + lo, g->ptrs[r] + lo, (size_t)cnt * sizeof(double), hipMemcpyDeviceToDevice, st);
        }
        (void)hipStreamSynchronize(st);
        g->wait();
        return true;
    }
    bool barrier(hipStream_t st) override
    {
        (void)hipStreamSynchronize(st);
        g->wait();
        return true;
    }
};
}  // namespace

std::unique_ptr<Comm> make_thread_comm(ThreadGroup *g, int rank)
{
    std::unique_ptr<ThreadComm> c(new ThreadComm());
    c->g = g;
    c->rank = rank;
    c->size = g->n;
    return c;
}

}  // namespace sparsh
