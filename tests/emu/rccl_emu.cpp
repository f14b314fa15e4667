// TEST INFRASTRUCTURE: a stand-in for librccl with the eight entry points hq_shard.hip binds at run time (HQ_RCCL_LIBRARY
// names it), so that the library's RCCL transport -- communicator from a unique id, grouped ncclSend / ncclRecv of one chunk
// per peer, the pack / transfer / self-copy sequence around them -- runs between the PROCESSES of a multi-rank test on a box
// without a GPU.  Transport: a full mesh of unix-domain stream sockets (abstract names derived from the unique id); a group
// is progressed with non-blocking sends and receives on all of them at once, as the grouped calls of the real library are.
// Streams are ignored: the emulated device runs everything synchronously.  Nothing outside tests/ refers to this file.
#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include <deque>
#include <vector>

extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
}

struct Op { bool send; char* buf; size_t left; int peer; };
struct ncclComm {
  int world = 1, rank = 0;
  std::vector<int> fd;  // per peer (-1 for self)
  int listener = -1;
};
static thread_local int g_depth = 0;
static thread_local std::vector<Op> g_ops;
static thread_local ncclComm* g_comm = nullptr;

static size_t dtype_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}
static void make_addr(sockaddr_un* a, socklen_t* len, const char* tag, int rank) {
  memset(a, 0, sizeof(*a));
  a->sun_family = AF_UNIX;
  const int n = snprintf(a->sun_path + 1, sizeof(a->sun_path) - 1, "hq_emu_nccl_%s_%d", tag, rank);  // abstract namespace
  *len = (socklen_t)(offsetof(sockaddr_un, sun_path) + 1 + n);
}
static double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static ncclResult_t progress(ncclComm* c, std::vector<Op>& ops) {
  // per peer, sends and receives each in the order they were posted; everything non-blocking
  std::vector<std::deque<Op*>> sq(c->world), rq(c->world);
  std::vector<Op*> self_s, self_r;
  for (auto& o : ops) {
    if (o.peer < 0 || o.peer >= c->world) return ncclInvalidArgument;
    if (o.peer == c->rank) (o.send ? self_s : self_r).push_back(&o);
    else (o.send ? sq : rq)[o.peer].push_back(&o);
  }
  if (self_s.size() != self_r.size()) return ncclInvalidArgument;
  for (size_t i = 0; i < self_s.size(); ++i) {
    if (self_s[i]->left != self_r[i]->left) return ncclInvalidArgument;
    memmove(self_r[i]->buf, self_s[i]->buf, self_s[i]->left);
  }
  const double deadline = now_s() + 120.0;
  for (;;) {
    std::vector<pollfd> pf;
    std::vector<int> who;
    for (int p = 0; p < c->world; ++p) {
      if (p == c->rank) continue;
      short ev = 0;
      if (!sq[p].empty()) ev |= POLLOUT;
      if (!rq[p].empty()) ev |= POLLIN;
      if (ev) { pf.push_back(pollfd{c->fd[p], ev, 0}); who.push_back(p); }
    }
    if (pf.empty()) return ncclSuccess;
    if (now_s() > deadline) return ncclSystemError;
    if (poll(pf.data(), pf.size(), 1000) < 0 && errno != EINTR) return ncclSystemError;
    for (size_t i = 0; i < pf.size(); ++i) {
      const int p = who[i];
      if ((pf[i].revents & POLLOUT) && !sq[p].empty()) {
        Op* o = sq[p].front();
        const ssize_t n = send(c->fd[p], o->buf, o->left, MSG_DONTWAIT | MSG_NOSIGNAL);
        if (n > 0) { o->buf += n; o->left -= (size_t)n; if (!o->left) sq[p].pop_front(); }
        else if (n < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) return ncclSystemError;
      }
      if ((pf[i].revents & (POLLIN | POLLHUP)) && !rq[p].empty()) {
        Op* o = rq[p].front();
        const ssize_t n = recv(c->fd[p], o->buf, o->left, MSG_DONTWAIT);
        if (n > 0) { o->buf += n; o->left -= (size_t)n; if (!o->left) rq[p].pop_front(); }
        else if (n == 0) return ncclSystemError;  // the peer went away
        else if (errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) return ncclSystemError;
      }
    }
  }
}

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  unsigned char r[8];
  const int fd = open("/dev/urandom", O_RDONLY);
  if (fd < 0 || read(fd, r, sizeof(r)) != (ssize_t)sizeof(r)) { if (fd >= 0) close(fd); return ncclSystemError; }
  close(fd);
  char* p = id->internal;
  for (unsigned i = 0; i < sizeof(r); ++i) p += sprintf(p, "%02x", r[i]);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
  if (!out || world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
  char tag[32];
  memcpy(tag, id.internal, 16);
  tag[16] = 0;
  ncclComm* c = new ncclComm();
  c->world = world;
  c->rank = rank;
  c->fd.assign(world, -1);
  sockaddr_un a;
  socklen_t alen;
  c->listener = socket(AF_UNIX, SOCK_STREAM, 0);
  make_addr(&a, &alen, tag, rank);
  if (c->listener < 0 || bind(c->listener, (sockaddr*)&a, alen) || listen(c->listener, world)) { delete c; return ncclSystemError; }
  const double deadline = now_s() + 60.0;
  for (int p = 0; p < rank; ++p) {  // connect to every lower rank (it may not be listening yet)
    for (;;) {
      const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
      make_addr(&a, &alen, tag, p);
      if (connect(fd, (sockaddr*)&a, alen) == 0) {
        const int32_t me = rank;
        if (send(fd, &me, sizeof(me), MSG_NOSIGNAL) != (ssize_t)sizeof(me)) { close(fd); delete c; return ncclSystemError; }
        c->fd[p] = fd;
        break;
      }
      close(fd);
      if (now_s() > deadline) { delete c; return ncclSystemError; }
      usleep(2000);
    }
  }
  for (int k = rank + 1; k < world; ++k) {  // accept every higher rank
    pollfd pf{c->listener, POLLIN, 0};
    while (poll(&pf, 1, 1000) <= 0)
      if (now_s() > deadline) { delete c; return ncclSystemError; }
    const int fd = accept(c->listener, nullptr, nullptr);
    int32_t who = -1;
    if (fd < 0 || recv(fd, &who, sizeof(who), MSG_WAITALL) != (ssize_t)sizeof(who) || who <= rank || who >= world || c->fd[who] >= 0) {
      if (fd >= 0) close(fd);
      delete c;
      return ncclSystemError;
    }
    c->fd[who] = fd;
  }
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  for (int fd : c->fd)
    if (fd >= 0) close(fd);
  if (c->listener >= 0) close(c->listener);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* count) {
  if (!c || !count) return ncclInvalidArgument;
  *count = c->world;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return ncclInvalidArgument;
  if (--g_depth > 0) return ncclSuccess;
  ncclResult_t r = ncclSuccess;
  if (!g_ops.empty()) r = progress(g_comm, g_ops);
  g_ops.clear();
  g_comm = nullptr;
  return r;
}

static ncclResult_t post(bool send, void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c) {
  if (!c || (!buf && count)) return ncclInvalidArgument;
  if (g_comm && g_comm != c) return ncclInvalidArgument;  // one communicator per group is all the library uses
  g_comm = c;
  g_ops.push_back(Op{send, (char*)buf, count * dtype_size(t), peer});
  if (g_depth == 0) {  // outside a group: a group of one
    ncclResult_t r = progress(c, g_ops);
    g_ops.clear();
    g_comm = nullptr;
    return r;
  }
  return ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, void*) { return post(true, const_cast<void*>(buf), count, t, peer, c); }
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, void*) { return post(false, buf, count, t, peer, c); }

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "unhandled system error (emulated RCCL: a peer went away or did not arrive)";
    case ncclInvalidArgument: return "invalid argument";
    default: return "internal error";
  }
}

}  // extern "C"
