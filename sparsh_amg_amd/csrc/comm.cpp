// comm.cpp -- Comm backends (see comm.hpp).
#include "comm.hpp"

#include <condition_variable>
#include <cstring>
#include <mutex>

#include <rccl/rccl.h>

#include "kernels.hpp"

namespace sparsh {

// ------------------------------------------------------------------ single rank
namespace {
class SelfComm : public Comm {
public:
    bool exchange(const DevPlan &, double *, hipStream_t) override { return true; }
    bool exchange_staged(const DevDeepPlan &, const double *, hipStream_t) override { return true; }
    bool allreduce_sum(double *, int, hipStream_t) override { return true; }
    bool allgather(double *, const Partition &, hipStream_t) override { return true; }
    bool barrier(hipStream_t) override { return true; }
    bool bcast(void *, size_t, int, hipStream_t) override { return true; }
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
        if (p.nsend > 0 && p.need_pack) launch_pack(p.nsend, p.send_idx, vec, p.sendbuf, st);
        if (!ok(ncclGroupStart(), "ncclGroupStart")) return false;
        bool good = true;  // a started group is always ended, also after a failed call inside it
        for (const HaloSeg &s : p.send) {
            const double *src = s.start >= 0 ? vec + s.start : p.sendbuf + s.off;
            if (good) good = ok(ncclSend(src, (size_t)s.cnt, ncclDouble, s.peer, comm, st), "ncclSend");
        }
        for (const HaloSeg &r : p.recv)
            if (good) good = ok(ncclRecv(vec + p.nloc + r.off, (size_t)r.cnt, ncclDouble, r.peer, comm, st), "ncclRecv");
        const ncclResult_t e = ncclGroupEnd();
        return good && ok(e, "ncclGroupEnd");
    }
    bool exchange_staged(const DevDeepPlan &p, const double *src, hipStream_t st) override
    {
        if (p.recv.empty() && p.send.empty()) return true;
        if (p.nsend > 0) launch_pack(p.nsend, p.send_idx, src, p.sendbuf, st);
        if (!ok(ncclGroupStart(), "ncclGroupStart")) return false;
        bool good = true;
        for (const HaloSeg &s : p.send)
            if (good) good = ok(ncclSend(p.sendbuf + s.off, (size_t)s.cnt, ncclDouble, s.peer, comm, st), "ncclSend");
        for (const HaloSeg &r : p.recv)
            if (good) good = ok(ncclRecv(p.recvbuf + r.off, (size_t)r.cnt, ncclDouble, r.peer, comm, st), "ncclRecv");
        const ncclResult_t e = ncclGroupEnd();
        return good && ok(e, "ncclGroupEnd");
    }
    bool allreduce_sum(double *dev, int n, hipStream_t st) override
    {
        return ok(ncclAllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum, comm, st), "ncclAllReduce");
    }
    bool allgather(double *full, const Partition &part, hipStream_t st) override
    {
        // ranges differ in length, and on a full xGMI mesh every pair has its own link: one grouped
        // set of point-to-point transfers (my range to every peer, every peer's range from it), in place
        // -- the same primitives, in the same kind of group, as the halo exchange
        const int mylo = part.lo(rank), mycnt = part.hi(rank) - mylo;
        if (!ok(ncclGroupStart(), "ncclGroupStart")) return false;
        bool good = true;
        for (int q = 0; q < size; ++q) {
            if (q == rank) continue;
            const int lo = part.lo(q), cnt = part.hi(q) - lo;
            if (good && mycnt > 0) good = ok(ncclSend(full + mylo, (size_t)mycnt, ncclDouble, q, comm, st), "ncclSend");
            if (good && cnt > 0) good = ok(ncclRecv(full + lo, (size_t)cnt, ncclDouble, q, comm, st), "ncclRecv");
        }
        const ncclResult_t e = ncclGroupEnd();
        return good && ok(e, "ncclGroupEnd");
    }
    bool barrier(hipStream_t st) override
    {
        if (!scratch && hipMalloc(reinterpret_cast<void **>(&scratch), sizeof(double)) != hipSuccess) return false;
        (void)hipMemsetAsync(scratch, 0, sizeof(double), st);
        if (!allreduce_sum(scratch, 1, st)) return false;
        return hipStreamSynchronize(st) == hipSuccess;
    }
    bool bcast(void *dev, size_t bytes, int root, hipStream_t st) override
    {
        return ok(ncclBroadcast(dev, dev, bytes, ncclChar, root, comm, st), "ncclBroadcast");
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
    std::vector<const DevDeepPlan *> dplans;
    std::vector<double *> vecs;
    std::vector<hipEvent_t> pub, done;  // per rank: "my boundary data is ready" / "my pulls are finished"
    std::vector<double *> ptrs;
    std::vector<std::vector<double>> host;
    int fail_after = -1;  // test hook: every rank's exchange() number fail_after (0-based) and later ones fail
    double delay_us = 0.0;  // test hook: every transport call first occupies the caller's stream this long (a slow link)
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
    g->dplans.assign((size_t)nranks, nullptr);
    g->vecs.assign((size_t)nranks, nullptr);
    g->ptrs.assign((size_t)nranks, nullptr);
    g->pub.assign((size_t)nranks, nullptr);
    g->done.assign((size_t)nranks, nullptr);
    g->host.resize((size_t)nranks);
    return g;
}

void thread_group_fail_after(ThreadGroup *g, int n)
{
    if (g) g->fail_after = n;
}

void thread_group_set_delay(ThreadGroup *g, double microseconds)
{
    if (g) g->delay_us = microseconds;
}

void thread_group_destroy(ThreadGroup *g)
{
    if (!g) return;
    for (hipEvent_t e : g->pub)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : g->done)
        if (e) (void)hipEventDestroy(e);
    delete g;
}

namespace {
class ThreadComm : public Comm {
public:
    ThreadGroup *g = nullptr;
    // Stream-asynchronous pull exchange: the host threads only rendezvous to trade pointers and
    // event handles; all data movement and all ordering is expressed with stream events, exactly as
    // the engine has to express it for RCCL.  A missing dependency in the engine (e.g. a kernel that
    // does not wait for the halo) therefore shows up as wrong data in the virtual-rank tests.
    int calls = 0;
    bool exchange(const DevPlan &p, double *vec, hipStream_t st) override
    {
        // injected transport failure (tests): all ranks count the same call sequence, so all of them
        // fail at the same call and nobody is left waiting at a rendezvous
        if (g->fail_after >= 0 && calls++ >= g->fail_after) {
            error = "injected transport failure (test hook)";
            return false;
        }
        if (!g->pub[rank]) {
            (void)hipEventCreateWithFlags(&g->pub[rank], hipEventDisableTiming);
            (void)hipEventCreateWithFlags(&g->done[rank], hipEventDisableTiming);
        }
        launch_spin(g->delay_us, st);
        // peers may still be pulling the previous exchange's data out of my sendbuf
        g->wait();
        if (primed)
            for (int r = 0; r < size; ++r)
                if (r != rank && g->done[r]) (void)hipStreamWaitEvent(st, g->done[r], 0);
        if (p.nsend > 0 && p.need_pack) launch_pack(p.nsend, p.send_idx, vec, p.sendbuf, st);
        (void)hipEventRecord(g->pub[rank], st);
        g->plans[rank] = &p;
        g->vecs[rank] = vec;
        g->wait();  // every rank's data-ready event is enqueued and its pointers are published
        bool okp = true;
        for (const HaloSeg &r : p.recv) {
            const DevPlan *q = g->plans[r.peer];
            const HaloSeg *src = nullptr;
            for (const HaloSeg &s : q->send)
                if (s.peer == rank) src = &s;
            if (!src || src->cnt != r.cnt) {
                error = "thread comm: send/recv plans disagree";
                okp = false;
                continue;
            }
            (void)hipStreamWaitEvent(st, g->pub[r.peer], 0);
            const double *from = src->start >= 0 ? g->vecs[r.peer] + src->start : q->sendbuf + src->off;
            (void)hipMemcpyAsync(vec + p.nloc + r.off, from, (size_t)r.cnt * sizeof(double), hipMemcpyDeviceToDevice, st);
        }
        (void)hipEventRecord(g->done[rank], st);
        g->wait();  // every rank's pulls are enqueued
        // completion of this exchange on st must also mean: peers have finished reading my data
        for (const HaloSeg &s : p.send)
            if (g->done[s.peer]) (void)hipStreamWaitEvent(st, g->done[s.peer], 0);
        primed = true;
        return okp;
    }
    bool primed = false;
    // same stream-asynchronous pull protocol for the staged (deep-halo) exchange: peers read my sendbuf
    bool exchange_staged(const DevDeepPlan &p, const double *src, hipStream_t st) override
    {
        if (g->fail_after >= 0 && calls++ >= g->fail_after) {
            error = "injected transport failure (test hook)";
            return false;
        }
        if (!g->pub[rank]) {
            (void)hipEventCreateWithFlags(&g->pub[rank], hipEventDisableTiming);
            (void)hipEventCreateWithFlags(&g->done[rank], hipEventDisableTiming);
        }
        launch_spin(g->delay_us, st);
        g->wait();  // peers may still be pulling the previous exchange's data out of my sendbuf
        if (primed)
            for (int r = 0; r < size; ++r)
                if (r != rank && g->done[r]) (void)hipStreamWaitEvent(st, g->done[r], 0);
        if (p.nsend > 0) launch_pack(p.nsend, p.send_idx, src, p.sendbuf, st);
        (void)hipEventRecord(g->pub[rank], st);
        g->dplans[rank] = &p;
        g->wait();
        bool okp = true;
        for (const HaloSeg &r : p.recv) {
            const DevDeepPlan *q = g->dplans[r.peer];
            const HaloSeg *sseg = nullptr;
            if (q)
                for (const HaloSeg &s : q->send)
                    if (s.peer == rank) sseg = &s;
            if (!sseg || sseg->cnt != r.cnt) {
                error = "thread comm: staged send/recv plans disagree";
                okp = false;
                continue;
            }
            (void)hipStreamWaitEvent(st, g->pub[r.peer], 0);
            (void)hipMemcpyAsync(p.recvbuf + r.off, q->sendbuf + sseg->off, (size_t)r.cnt * sizeof(double), hipMemcpyDeviceToDevice, st);
        }
        (void)hipEventRecord(g->done[rank], st);
        g->wait();
        for (const HaloSeg &s : p.send)
            if (g->done[s.peer]) (void)hipStreamWaitEvent(st, g->done[s.peer], 0);
        primed = true;
        return okp;
    }
    bool allreduce_sum(double *dev, int n, hipStream_t st) override
    {
        launch_spin(g->delay_us, st);
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
        launch_spin(g->delay_us, st);
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
    bool bcast(void *dev, size_t bytes, int root, hipStream_t st) override
    {
        (void)hipStreamSynchronize(st);  // the root's buffer is complete before anyone reads it
        g->ptrs[rank] = static_cast<double *>(dev);
        g->wait();
        if (rank != root && bytes > 0) (void)hipMemcpyAsync(dev, g->ptrs[root], bytes, hipMemcpyDeviceToDevice, st);
        (void)hipStreamSynchronize(st);
        g->wait();  // the root may reuse its buffer only after every pull has finished
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
