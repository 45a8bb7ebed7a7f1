// tests/fake_rccl/fake_nccl.cpp -- TEST-ONLY fake of ncclSend / ncclRecv / ncclGroup* over named pipes (see rccl/rccl.h here).
// Receives posted inside a group are carried out at ncclGroupEnd in posting order, sends immediately: enough for a gather.
#include "rccl/rccl.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct fakeComm { int rank, world; char tag[32]; int rfd[64], wfd[64]; };   // one pipe per ordered pair, opened on first use, kept open
namespace {
struct Pending { void *buf; size_t n; int peer; fakeComm *c; };
std::vector<Pending> g_pending;
int g_group = 0;
std::string fifo(const fakeComm *c, int src, int dst)
{
    const char *d = getenv("FAKE_NCCL_DIR");
    const std::string p = std::string(d ? d : "/tmp") + "/fake_nccl_" + c->tag + "_" + std::to_string(src) + "_to_" + std::to_string(dst);
    mkfifo(p.c_str(), 0600);                                  // EEXIST is fine: either end may create it
    return p;
}
#define DBG(...) do { if (getenv("FAKE_NCCL_DEBUG")) { fprintf(stderr, __VA_ARGS__); fflush(stderr); } } while (0)
ncclResult_t do_recv(const Pending &p)
{
    DBG("rank %d: recv %zu from %d via %s\n", p.c->rank, p.n, p.peer, fifo(p.c, p.peer, p.c->rank).c_str());
    int &fd = p.c->rfd[p.peer];
    if (fd < 0) fd = open(fifo(p.c, p.peer, p.c->rank).c_str(), O_RDONLY);
    if (fd < 0) return ncclSystemError;
    size_t got = 0;
    while (got < p.n) { const ssize_t r = read(fd, (char *)p.buf + got, p.n - got); if (r <= 0) return ncclSystemError; got += (size_t)r; }
    return ncclSuccess;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "%ld_%d", (long)getpid(), rand());
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks > 64) return ncclSystemError;
    fakeComm *c = new fakeComm{rank, nranks, {0}, {0}, {0}};
    for (int i = 0; i < 64; i++) c->rfd[i] = c->wfd[i] = -1;
    strncpy(c->tag, id.internal, sizeof(c->tag) - 1);         // ranks that were handed different ids never meet: the test would hang and time out
    *comm = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    for (int i = 0; i < 64; i++) { if (comm->rfd[i] >= 0) close(comm->rfd[i]); if (comm->wfd[i] >= 0) close(comm->wfd[i]); }
    delete comm;
    return ncclSuccess;
}
ncclResult_t ncclGroupStart(void) { g_group++; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void)
{
    if (--g_group > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    for (const Pending &p : g_pending) if (do_recv(p) != ncclSuccess) rc = ncclSystemError;
    g_pending.clear();
    return rc;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t, int peer, ncclComm_t comm, void *)
{
    DBG("rank %d: send %zu to %d via %s\n", comm->rank, count, peer, fifo(comm, comm->rank, peer).c_str());
    int &fd = comm->wfd[peer];
    if (fd < 0) fd = open(fifo(comm, comm->rank, peer).c_str(), O_WRONLY);
    if (fd < 0) return ncclSystemError;
    size_t put = 0;
    while (put < count) { const ssize_t w = write(fd, (const char *)buf + put, count - put); if (w <= 0) return ncclSystemError; put += (size_t)w; }
    return ncclSuccess;
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t, int peer, ncclComm_t comm, void *)
{
    const Pending p{buf, count, peer, comm};
    if (g_group > 0) { g_pending.push_back(p); return ncclSuccess; }
    return do_recv(p);
}
}
