// xf_group.cc — the process group of a multi-GPU run: one process per GPU, the exchange steps
// of the sharded table behind the C ABI.
//
// Replaces ps-lite's van / postoffice as xflow uses them (src/model/main.cc:22-47:
// ps::Start / IsWorker / MyRank / Finalize; the launch environment of scripts/local.sh:3-14:
// DMLC_NUM_WORKER, DMLC_PS_ROOT_URI, DMLC_PS_ROOT_PORT) and the transport under
// KVWorker::Push/Pull (src/model/lr/lr_worker.cc:170,175): keys / weights / gradients travel
// to and from the shard that owns them as ONE all-to-all-v per direction.
//
// Two layers:
//   * bootstrap: a TCP star around rank 0 (the process that owns MASTER_ADDR:MASTER_PORT —
//     what ps-lite's scheduler is to its nodes).  Ranks are given (RANK / DMLC_RANK) or, as in
//     ps-lite, handed out in arrival order.  Small host-side collectives (barrier, allgather of
//     counts / flags, the ncclUniqueId broadcast) run over it.
//   * data path: RCCL over xGMI — grouped ncclSend / ncclRecv per peer on the caller's stream
//     (xGMI is a full mesh: each peer pair has its own link, an all-to-all is one message per
//     link).  librccl is opened at run time (dlopen), so the library has no link-time
//     dependency on it and shares the copy a host program (e.g. torch) may already have mapped.
//     XF_TRANSPORT_HOST stages the same exchange through the bootstrap sockets instead: for
//     tests that run several ranks on ONE GPU (RCCL refuses two ranks per device) or none.
#include <arpa/inet.h>
#include <dlfcn.h>
#include <errno.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "xf_common.h"

namespace xf {
int device_copy(void *dst, const void *src, size_t bytes, hipStream_t s);  // xf_table.hip
}

namespace {

// ---- the slice of rccl.h this file uses (the library is dlopen'ed)
typedef struct { char internal[128]; } ncclUniqueId_t;
typedef void *ncclComm_h;
struct Rccl {
  void *so = nullptr;
  int (*GetUniqueId)(ncclUniqueId_t *) = nullptr;
  int (*CommInitRank)(ncclComm_h *, int, ncclUniqueId_t, int) = nullptr;
  int (*CommDestroy)(ncclComm_h) = nullptr;
  int (*CommAbort)(ncclComm_h) = nullptr;  // (optional symbol)
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, ncclComm_h, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, ncclComm_h, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
constexpr int kNcclChar = 0;  // ncclInt8 / ncclChar
constexpr int32_t kHelloMagic = 0x78664731;  // "xfG1": first word of a rank's hello and of its answer

int load_rccl(Rccl &r) {
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.so) break;
  }
  if (!r.so) return xf::set_error(XF_EHIP, "cannot open librccl.so.1: %s", dlerror());
#define XF_SYM(field, name)                                             \
  *(void **)(&r.field) = dlsym(r.so, name);                             \
  if (!r.field) return xf::set_error(XF_EHIP, "librccl has no %s", name)
  XF_SYM(GetUniqueId, "ncclGetUniqueId");
  XF_SYM(CommInitRank, "ncclCommInitRank");
  XF_SYM(CommDestroy, "ncclCommDestroy");
  XF_SYM(GroupStart, "ncclGroupStart");
  XF_SYM(GroupEnd, "ncclGroupEnd");
  XF_SYM(Send, "ncclSend");
  XF_SYM(Recv, "ncclRecv");
  XF_SYM(GetErrorString, "ncclGetErrorString");
#undef XF_SYM
  *(void **)(&r.CommAbort) = dlsym(r.so, "ncclCommAbort");
  return XF_OK;
}

// ---- sockets
int send_all(int fd, const void *p, size_t n) {
  const char *c = (const char *)p;
  while (n) {
    const ssize_t k = send(fd, c, n, MSG_NOSIGNAL);
    if (k < 0) {
      if (errno == EINTR) continue;
      return xf::set_error(XF_EIO, "group: send failed: %s", strerror(errno));
    }
    c += k;
    n -= (size_t)k;
  }
  return XF_OK;
}

int recv_all(int fd, void *p, size_t n) {
  char *c = (char *)p;
  while (n) {
    const ssize_t k = recv(fd, c, n, 0);
    if (k == 0) return xf::set_error(XF_EIO, "group: a peer closed its connection");
    if (k < 0) {
      if (errno == EINTR) continue;
      return xf::set_error(XF_EIO, "group: recv failed: %s", strerror(errno));
    }
    c += k;
    n -= (size_t)k;
  }
  return XF_OK;
}

void tune_socket(int fd) {
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
}

const char *env_first(std::initializer_list<const char *> names) {
  for (const char *n : names) {
    const char *v = getenv(n);
    if (v && *v) return v;
  }
  return nullptr;
}

}  // namespace

struct xf_group {
  int rank = 0, world = 1, transport = XF_TRANSPORT_RCCL;
  int listen_fd = -1;
  std::vector<int> peer;  // rank 0: socket of every other rank; others: peer[0] = rank 0
  Rccl rccl;
  // two communicators ("channels"): RCCL serialises the work of ONE communicator and must see
  // every rank enqueue on it in the same order, so work that two streams of a rank issue
  // independently (the sharded trainer's stale1 schedule: Push(t) on a side stream under
  // Pull(t+1)) gets a communicator per stream
  ncclComm_h comm[XF_GROUP_CHANNELS] = {nullptr, nullptr};
  std::vector<char> hs, hr;  // host staging of the host transport
  bool aborted = false;      // xf_group_abort: no device exchange any more
};

namespace {

// rank 0 gathers `bytes` from everybody, everybody gets all of it back
int allgather_host(xf_group *g, const void *in, size_t bytes, void *out) {
  char *o = (char *)out;
  if (g->world == 1) {
    memcpy(o, in, bytes);
    return XF_OK;
  }
  if (g->rank == 0) {
    memcpy(o, in, bytes);
    for (int r = 1; r < g->world; ++r) XF_TRY(recv_all(g->peer[r], o + (size_t)r * bytes, bytes));
    for (int r = 1; r < g->world; ++r) XF_TRY(send_all(g->peer[r], o, bytes * g->world));
  } else {
    XF_TRY(send_all(g->peer[0], in, bytes));
    XF_TRY(recv_all(g->peer[0], o, bytes * g->world));
  }
  return XF_OK;
}

}  // namespace

// rank < 0 / world <= 0 / addr == NULL / port <= 0: taken from the environment —
//   world: WORLD_SIZE, XF_WORLD, DMLC_NUM_WORKER      rank: RANK, XF_RANK, DMLC_RANK (absent:
//   addr:  MASTER_ADDR, DMLC_PS_ROOT_URI (127.0.0.1)         handed out in arrival order,
//   port:  MASTER_PORT, DMLC_PS_ROOT_PORT (29512)            the process that binds the port
//                                                            first being rank 0)
// every rank's (ok, message) for one bring-up stage; returns XF_OK when all ranks are fine,
// otherwise the first failing rank's message on every rank
static int agree(xf_group *g, int my_rc, const std::string &my_msg, std::string *why) {
  struct Slot {
    int32_t rc;
    char msg[252];
  } mine{};
  mine.rc = my_rc;
  snprintf(mine.msg, sizeof(mine.msg), "%s", my_msg.c_str());
  std::vector<Slot> all(g->world);
  if (allgather_host(g, &mine, sizeof(mine), all.data()) != XF_OK) {
    *why = "the bootstrap connection failed";
    return XF_EIO;
  }
  for (int r = 0; r < g->world; ++r)
    if (all[r].rc != XF_OK) {
      all[r].msg[sizeof(all[r].msg) - 1] = 0;
      *why = "rank " + std::to_string(r) + ": " + all[r].msg;
      return all[r].rc;
    }
  return XF_OK;
}

static int alltoallv_rccl(xf_group *g, int ch, const char *sb, const std::vector<size_t> &so,
                          char *rb, const std::vector<size_t> &ro, hipStream_t s);
static void destroy_comms(xf_group *g) {
  for (ncclComm_h &c : g->comm) {
    if (c && g->rccl.CommDestroy) g->rccl.CommDestroy(c);
    c = nullptr;
  }
}

static int rccl_bring_up(xf_group *g, std::string *why) {
  // 1. the library
  int rc = load_rccl(g->rccl);
  int ndev = 0;
  if (rc == XF_OK && (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0))
    rc = xf::set_error(XF_ENOGPU, "no HIP device");
  XF_TRY(agree(g, rc, rc ? xf_last_error() : "", why));
  // 2. the communicators, one per channel
  rc = XF_OK;
  std::string msg;
  for (int ch = 0; ch < XF_GROUP_CHANNELS; ++ch) {
    std::vector<ncclUniqueId_t> ids(g->world);
    ncclUniqueId_t mine{};
    if (g->rank == 0 && rc == XF_OK) {
      const int n = g->rccl.GetUniqueId(&mine);
      if (n) {
        rc = XF_EHIP;
        msg = std::string("ncclGetUniqueId: ") + g->rccl.GetErrorString(n);
      }
    }
    XF_TRY(agree(g, rc, msg, why));
    if (allgather_host(g, &mine, sizeof(mine), ids.data()) != XF_OK) {
      *why = "the bootstrap connection failed";
      return XF_EIO;
    }
    const int n = g->rccl.CommInitRank(&g->comm[ch], g->world, ids[0], g->rank);
    if (n) {
      rc = XF_EHIP;
      msg = std::string("ncclCommInitRank: ") + g->rccl.GetErrorString(n);
      g->comm[ch] = nullptr;
    }
    XF_TRY(agree(g, rc, msg, why));
  }
  // 3. one exchange on the device: every rank sends its number to every peer
  const int W = g->world;
  std::vector<int32_t> h(W * 4, g->rank), back(W * 4, -1);
  std::vector<size_t> off(W + 1);
  for (int p = 0; p <= W; ++p) off[p] = (size_t)p * 16;
  void *ds = nullptr, *dr = nullptr;
  auto hip_ok = [&](hipError_t e, const char *what) {
    if (e != hipSuccess && rc == XF_OK) {
      rc = XF_EHIP;
      msg = std::string(what) + ": " + hipGetErrorString(e);
    }
    return e == hipSuccess;
  };
  if (hip_ok(hipMalloc(&ds, off[W]), "hipMalloc") && hip_ok(hipMalloc(&dr, off[W]), "hipMalloc") &&
      hip_ok(hipMemcpy(ds, h.data(), off[W], hipMemcpyHostToDevice), "hipMemcpy") &&
      hip_ok(hipMemset(dr, 0xff, off[W]), "hipMemset")) {
    for (int ch = 0; ch < XF_GROUP_CHANNELS && rc == XF_OK; ++ch) {
      if (!hip_ok(hipMemset(dr, 0xff, off[W]), "hipMemset")) break;
      if (alltoallv_rccl(g, ch, (const char *)ds, off, (char *)dr, off, nullptr) != XF_OK) {
        rc = XF_EHIP;
        msg = xf_last_error();
      } else if (hip_ok(hipStreamSynchronize(nullptr), "the first all-to-all-v") &&
                 hip_ok(hipMemcpy(back.data(), dr, off[W], hipMemcpyDeviceToHost), "hipMemcpy")) {
        for (int p = 0; p < W && rc == XF_OK; ++p)
          for (int q = 0; q < 4; ++q)
            if (back[p * 4 + q] != p) {
              rc = XF_EHIP;
              msg = "the first all-to-all-v (channel " + std::to_string(ch) + ") delivered " +
                    std::to_string(back[p * 4 + q]) + " from rank " + std::to_string(p);
              break;
            }
      }
    }
  }
  if (ds) (void)hipFree(ds);
  if (dr) (void)hipFree(dr);
  return agree(g, rc, msg, why);
}

extern "C" int xf_group_create(xf_group **out, int rank, int world, const char *addr, int port,
                               int transport, int device) {
  XF_REQUIRE(out, "xf_group_create: null argument");
  XF_REQUIRE(transport == XF_TRANSPORT_RCCL || transport == XF_TRANSPORT_HOST ||
                 transport == XF_TRANSPORT_AUTO,
             "xf_group_create: transport %d", transport);
  if (world <= 0) {
    const char *v = env_first({"WORLD_SIZE", "XF_WORLD", "DMLC_NUM_WORKER"});
    world = v ? atoi(v) : 1;
  }
  if (rank < 0) {
    const char *v = env_first({"RANK", "XF_RANK", "DMLC_RANK"});
    rank = v ? atoi(v) : -1;
  }
  std::string host = addr ? addr : "";
  if (host.empty()) {
    const char *v = env_first({"MASTER_ADDR", "DMLC_PS_ROOT_URI"});
    host = v ? v : "127.0.0.1";
  }
  if (port <= 0) {
    const char *v = env_first({"MASTER_PORT", "DMLC_PS_ROOT_PORT"});
    port = v ? atoi(v) : 29512;
  }
  XF_REQUIRE(world >= 1 && rank < world, "xf_group_create: rank %d of %d", rank, world);
  xf_group *g = new xf_group;
  g->world = world;
  g->transport = transport;
  struct Guard {
    xf_group *g;
    ~Guard() {
      if (g) xf_group_destroy(g);
    }
  } guard{g};
  if (world == 1) {
    g->rank = 0;
  } else {
    sockaddr_in sa{};
    sa.sin_family = AF_INET;
    sa.sin_port = htons((uint16_t)port);
    if (inet_pton(AF_INET, host.c_str(), &sa.sin_addr) != 1) {
      addrinfo hints{}, *res = nullptr;
      hints.ai_family = AF_INET;
      hints.ai_socktype = SOCK_STREAM;
      if (getaddrinfo(host.c_str(), nullptr, &hints, &res) != 0 || !res)
        return xf::set_error(XF_EIO, "xf_group_create: cannot resolve %s", host.c_str());
      sa.sin_addr = ((sockaddr_in *)res->ai_addr)->sin_addr;
      freeaddrinfo(res);
    }
    bool root = rank == 0;
    if (rank <= 0) {  // rank 0, or "whoever binds first"
      const int fd = socket(AF_INET, SOCK_STREAM, 0);
      int one = 1;
      setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
      // A loopback rendezvous address stays on this host, and an explicit numeric address is
      // taken at its word; a NAME is not: it may resolve to a loopback alias on this host
      // (127.0.1.1 in /etc/hosts) or to another interface than the peers reach, so an EXPLICIT
      // rank 0 then listens on every interface (the hello's magic word turns strays away).
      // Ranks handed out on arrival (rank < 0) always bind the resolved address: on a node that
      // does not own it the bind fails, and that failure is what elects ONE root across nodes —
      // a wildcard bind would succeed on every node and give each its own.
      sockaddr_in any = sa;
      const bool numeric = inet_pton(AF_INET, host.c_str(), &any.sin_addr) == 1;
      any.sin_addr = sa.sin_addr;
      if (!numeric && rank == 0) any.sin_addr.s_addr = htonl(INADDR_ANY);
      if (bind(fd, (sockaddr *)&any, sizeof(any)) == 0 && listen(fd, world + 8) == 0) {
        g->listen_fd = fd;
        root = true;
      } else {
        close(fd);
        if (rank == 0)
          return xf::set_error(XF_EIO, "xf_group_create: rank 0 cannot listen on port %d: %s",
                               port, strerror(errno));
      }
    }
    if (root) {
      g->rank = 0;
      g->peer.assign(world, -1);
      int next_auto = 1;
      std::vector<int> pending;  // connections that asked for "any rank"
      for (int have = 1; have < world; ++have) {
        pollfd pf{g->listen_fd, POLLIN, 0};
        if (poll(&pf, 1, 120000) <= 0)
          return xf::set_error(XF_EIO, "xf_group_create: %d of %d ranks arrived within 120 s",
                               have, world);
        const int fd = accept(g->listen_fd, nullptr, nullptr);
        if (fd < 0) return xf::set_error(XF_EIO, "xf_group_create: accept: %s", strerror(errno));
        tune_socket(fd);
        // the hello: a magic word and the rank asked for, within 10 s — a stray connection
        // (a port scanner, another service's client) is dropped instead of holding the group up
        timeval tv{10, 0};
        setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
        int32_t hello[2] = {0, -1};
        if (recv_all(fd, hello, 8) != XF_OK || hello[0] != kHelloMagic) {
          close(fd);
          --have;
          continue;
        }
        timeval none{0, 0};
        setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &none, sizeof(none));
        const int32_t want = hello[1];
        if (want < 0) {
          pending.push_back(fd);
        } else {
          if (want == 0 || want >= world || g->peer[want] != -1) {
            close(fd);
            for (int p : pending) close(p);
            return xf::set_error(XF_EINVAL, "xf_group_create: rank %d announced twice or out of "
                                 "range", want);
          }
          g->peer[want] = fd;
        }
      }
      for (int fd : pending) {  // arrival order fills the ranks nobody claimed
        while (next_auto < world && g->peer[next_auto] != -1) ++next_auto;
        g->peer[next_auto] = fd;
      }
      for (int r = 1; r < world; ++r) {
        const int32_t id[3] = {r, world, kHelloMagic};
        XF_TRY(send_all(g->peer[r], id, 12));
      }
    } else {
      int fd = -1;
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {  // rank 0 may not be up yet
        fd = socket(AF_INET, SOCK_STREAM, 0);
        if (connect(fd, (sockaddr *)&sa, sizeof(sa)) == 0) break;
        close(fd);
        fd = -1;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
          return xf::set_error(XF_EIO, "xf_group_create: cannot reach rank 0 at %s:%d: %s",
                               host.c_str(), port, strerror(errno));
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
      }
      tune_socket(fd);
      g->peer.assign(1, fd);
      const int32_t hello[2] = {kHelloMagic, rank};
      XF_TRY(send_all(fd, hello, 8));
      int32_t id[3] = {0, 0, 0};
      XF_TRY(recv_all(fd, id, 12));
      if (id[2] != kHelloMagic)
        return xf::set_error(XF_EIO, "xf_group_create: %s:%d answers, but it is not an xf_group "
                             "rank 0 (is the port taken by another service?)", host.c_str(), port);
      if (id[1] != world)
        return xf::set_error(XF_EINVAL, "xf_group_create: rank 0 runs a world of %d, this "
                             "process one of %d", id[1], world);
      g->rank = id[0];
    }
  }
  // this process's GPU: a given device, the current one (-1), or by rank (-2: LOCAL_RANK, else
  // rank modulo the visible devices) — ranks may only be known now
  if (device >= 0 || device == -2) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) {
      int d = device;
      if (device == -2) {
        const char *lr = env_first({"LOCAL_RANK"});
        d = lr ? atoi(lr) : g->rank % ndev;
      }
      XF_REQUIRE(d >= 0 && d < ndev, "xf_group_create: device %d of %d", d, ndev);
      XF_HIP(hipSetDevice(d));
    } else if (transport == XF_TRANSPORT_RCCL) {
      return xf::set_error(XF_ENOGPU, "xf_group_create: no HIP device for the RCCL transport");
    }
  }
  if (transport != XF_TRANSPORT_HOST) {
    // RCCL comes up in three stages (library, communicator, one all-to-all-v on the device);
    // after each one the ranks agree over the bootstrap, so that a failure on one rank is an
    // error (or, XF_TRANSPORT_AUTO, the host transport) on all of them instead of a hang
    std::string why;
    int stage_rc = rccl_bring_up(g, &why);
    if (stage_rc == XF_OK) {
      g->transport = XF_TRANSPORT_RCCL;
    } else if (transport == XF_TRANSPORT_AUTO && stage_rc != XF_EIO) {
      destroy_comms(g);
      g->transport = XF_TRANSPORT_HOST;
      if (g->rank == 0)
        fprintf(stderr, "xf_group: RCCL is not usable (%s): the exchange is staged through the "
                "bootstrap sockets\n", why.c_str());
    } else {
      return xf::set_error(stage_rc, "xf_group_create: %s", why.c_str());
    }
  }
  guard.g = nullptr;
  *out = g;
  return XF_OK;
}

// Give up on the communicators: work of theirs that is stuck on a stream (a peer has died) is
// made to end.  The group keeps its sockets; device exchanges fail from here on.
extern "C" int xf_group_abort(xf_group *g) {
  XF_REQUIRE(g, "xf_group_abort: null group");
  for (ncclComm_h &c : g->comm) {
    if (c && g->rccl.CommAbort) g->rccl.CommAbort(c);
    c = nullptr;  // (without ncclCommAbort the communicator is leaked, never destroyed: a
                  // destroy would wait for the stuck work)
  }
  g->aborted = true;
  return XF_OK;
}

extern "C" int xf_group_destroy(xf_group *g) {
  if (!g) return XF_OK;
  destroy_comms(g);
  for (int fd : g->peer)
    if (fd >= 0) close(fd);
  if (g->listen_fd >= 0) close(g->listen_fd);
  delete g;
  return XF_OK;
}

extern "C" int xf_group_info(const xf_group *g, int *rank, int *world, int *transport) {
  XF_REQUIRE(g, "xf_group_info: null group");
  if (rank) *rank = g->rank;
  if (world) *world = g->world;
  if (transport) *transport = g->transport;
  return XF_OK;
}

extern "C" int xf_group_allgather_host(xf_group *g, const void *in, size_t bytes, void *out) {
  XF_REQUIRE(g && (bytes == 0 || (in && out)), "xf_group_allgather_host: null argument");
  if (bytes == 0) return XF_OK;
  return allgather_host(g, in, bytes, out);
}

extern "C" int xf_group_barrier(xf_group *g) {
  XF_REQUIRE(g, "xf_group_barrier: null group");
  char a = 0;
  std::vector<char> all(g->world);
  return allgather_host(g, &a, 1, all.data());
}

// rank 0 receives every rank's variable-size blob (sizes[r] bytes each); others get nothing
extern "C" int xf_group_gatherv_host(xf_group *g, const void *in, size_t bytes, void *out,
                                     const uint64_t *sizes) {
  XF_REQUIRE(g, "xf_group_gatherv_host: null group");
  if (g->rank != 0) return bytes ? send_all(g->peer[0], in, bytes) : XF_OK;
  XF_REQUIRE(out && sizes, "xf_group_gatherv_host: rank 0 needs the output buffer and sizes");
  char *o = (char *)out;
  if (bytes) memcpy(o, in, bytes);
  size_t off = (size_t)sizes[0];
  for (int r = 1; r < g->world; ++r) {
    if (sizes[r]) XF_TRY(recv_all(g->peer[r], o + off, (size_t)sizes[r]));
    off += (size_t)sizes[r];
  }
  return XF_OK;
}

// one grouped ncclSend/ncclRecv per peer with bytes to move; byte offsets so/ro per rank
static int alltoallv_rccl(xf_group *g, int ch, const char *sb, const std::vector<size_t> &so,
                          char *rb, const std::vector<size_t> &ro, hipStream_t s) {
  const int W = g->world;
  const size_t self = so[g->rank + 1] - so[g->rank];
  if (self)  // the slice that stays: a copy kernel on the stream, no collective
    XF_TRY(xf::device_copy(rb + ro[g->rank], sb + so[g->rank], self, s));
  if (W == 1) return XF_OK;
  int rc = g->rccl.GroupStart();
  for (int p = 0; p < W && !rc; ++p) {
    if (p == g->rank) continue;
    if (so[p + 1] > so[p])
      rc = g->rccl.Send(sb + so[p], so[p + 1] - so[p], kNcclChar, p, g->comm[ch], s);
    if (!rc && ro[p + 1] > ro[p])
      rc = g->rccl.Recv(rb + ro[p], ro[p + 1] - ro[p], kNcclChar, p, g->comm[ch], s);
  }
  const int rc2 = g->rccl.GroupEnd();
  if (rc || rc2)
    return xf::set_error(XF_EHIP, "xf_group_alltoallv: %s", g->rccl.GetErrorString(rc ? rc : rc2));
  return XF_OK;
}

// Both channels at once: on two streams of this rank, one per communicator, a grouped
// ncclSend / ncclRecv of `bytes` bytes with EVERY rank — this one included, so that a group of
// one drives RCCL too — then both streams are waited for and what arrived is checked.
// Collective.  RCCL transport only (the host transport has nothing to overlap).
extern "C" int xf_group_selftest(xf_group *g, size_t bytes) {
  XF_REQUIRE(g && bytes >= 4 && bytes % 4 == 0, "xf_group_selftest: bad argument");
  if (g->transport != XF_TRANSPORT_RCCL) return XF_OK;
  const int W = g->world;
  const size_t n = bytes / 4;
  hipStream_t st[XF_GROUP_CHANNELS] = {};
  uint32_t *ds[XF_GROUP_CHANNELS] = {}, *dr[XF_GROUP_CHANNELS] = {};
  std::vector<uint32_t> h(n * W), back(n * W);
  int rc = XF_OK;
  auto fail = [&](const char *what, const char *why) {
    if (rc == XF_OK) rc = xf::set_error(XF_EHIP, "xf_group_selftest: %s: %s", what, why);
  };
  for (int ch = 0; ch < XF_GROUP_CHANNELS && rc == XF_OK; ++ch) {
    for (int p = 0; p < W; ++p)
      for (size_t i = 0; i < n; ++i) h[p * n + i] = (uint32_t)(g->rank * 1000003 + ch * 7 + p);
    hipError_t e = hipStreamCreateWithFlags(&st[ch], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&ds[ch], bytes * W);
    if (e == hipSuccess) e = hipMalloc((void **)&dr[ch], bytes * W);
    if (e == hipSuccess) e = hipMemcpy(ds[ch], h.data(), bytes * W, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(dr[ch], 0xff, bytes * W);
    if (e != hipSuccess) fail("set-up", hipGetErrorString(e));
  }
  // enqueue on both communicators before waiting for either
  for (int ch = 0; ch < XF_GROUP_CHANNELS && rc == XF_OK; ++ch) {
    int r = g->rccl.GroupStart();
    for (int p = 0; p < W && !r; ++p) {
      r = g->rccl.Send(ds[ch] + (size_t)p * n, bytes, kNcclChar, p, g->comm[ch], st[ch]);
      if (!r) r = g->rccl.Recv(dr[ch] + (size_t)p * n, bytes, kNcclChar, p, g->comm[ch], st[ch]);
    }
    const int r2 = g->rccl.GroupEnd();
    if (r || r2) fail("send/recv", g->rccl.GetErrorString(r ? r : r2));
  }
  for (int ch = 0; ch < XF_GROUP_CHANNELS; ++ch) {
    if (rc == XF_OK) {
      hipError_t e = hipStreamSynchronize(st[ch]);
      if (e == hipSuccess) e = hipMemcpy(back.data(), dr[ch], bytes * W, hipMemcpyDeviceToHost);
      if (e != hipSuccess) fail("wait", hipGetErrorString(e));
      for (int p = 0; p < W && rc == XF_OK; ++p)
        for (size_t i = 0; i < n; ++i)
          if (back[p * n + i] != (uint32_t)(p * 1000003 + ch * 7 + g->rank)) {
            fail("data", "a slice arrived with the wrong contents");
            break;
          }
    }
    if (ds[ch]) (void)hipFree(ds[ch]);
    if (dr[ch]) (void)hipFree(dr[ch]);
    if (st[ch]) (void)hipStreamDestroy(st[ch]);
  }
  return rc;
}

// The exchange step of the sharded table: rank p's slice send[off_p .. off_p + send_counts[p])
// goes to rank p, recv is filled in source-rank order (counts in elements of elem_bytes).
// RCCL: asynchronous on `stream`, the buffers are device memory.  Host transport: blocking;
// device buffers are staged through the host (host_buffers != 0: they are host memory).
extern "C" int xf_group_alltoallv(xf_group *g, const void *send, const uint64_t *send_counts,
                                  void *recv, const uint64_t *recv_counts, size_t elem_bytes,
                                  int host_buffers, void *stream) {
  return xf_group_alltoallv_ch(g, 0, send, send_counts, recv, recv_counts, elem_bytes,
                               host_buffers, stream);
}

extern "C" int xf_group_alltoallv_ch(xf_group *g, int channel, const void *send,
                                     const uint64_t *send_counts, void *recv,
                                     const uint64_t *recv_counts, size_t elem_bytes,
                                     int host_buffers, void *stream) {
  XF_REQUIRE(g && send_counts && recv_counts && elem_bytes, "xf_group_alltoallv: null argument");
  XF_REQUIRE(channel >= 0 && channel < XF_GROUP_CHANNELS, "xf_group_alltoallv: channel %d",
             channel);
  const int W = g->world;
  std::vector<size_t> so(W + 1, 0), ro(W + 1, 0);
  for (int p = 0; p < W; ++p) {
    so[p + 1] = so[p] + (size_t)send_counts[p] * elem_bytes;
    ro[p + 1] = ro[p] + (size_t)recv_counts[p] * elem_bytes;
  }
  XF_REQUIRE((so[W] == 0 || send) && (ro[W] == 0 || recv), "xf_group_alltoallv: null buffer");
  XF_REQUIRE(send_counts[g->rank] == recv_counts[g->rank],
             "xf_group_alltoallv: a rank's slice for itself differs between send and recv");
  hipStream_t s = (hipStream_t)stream;
  const char *sb = (const char *)send;
  char *rb = (char *)recv;
  if (g->aborted)
    return xf::set_error(XF_EIO, "xf_group_alltoallv: the group's communicators were aborted");
  if (g->transport == XF_TRANSPORT_RCCL) {
    XF_REQUIRE(!host_buffers, "xf_group_alltoallv: the RCCL transport moves device memory");
    return alltoallv_rccl(g, channel, sb, so, rb, ro, s);
  }
  // ---- host transport: through rank 0
  const void *hsend = send;
  void *hrecv = recv;
  if (!host_buffers) {
    XF_HIP(hipStreamSynchronize(s));
    g->hs.resize(so[W]);
    g->hr.resize(ro[W]);
    if (so[W]) XF_HIP(hipMemcpy(g->hs.data(), send, so[W], hipMemcpyDeviceToHost));
    hsend = g->hs.data();
    hrecv = g->hr.data();
  }
  const char *hsb = (const char *)hsend;
  char *hrb = (char *)hrecv;
  if (W == 1) {
    if (so[1]) memcpy(hrb, hsb, so[1]);
  } else if (g->rank == 0) {
    // everybody's send buffer (with its byte offsets) comes in, every rank's recv buffer goes out
    std::vector<std::vector<char>> bufs(W);
    std::vector<std::vector<uint64_t>> offs(W, std::vector<uint64_t>(W + 1));
    bufs[0].assign(hsb, hsb + so[W]);
    for (int p = 0; p <= W; ++p) offs[0][p] = so[p];
    for (int r = 1; r < W; ++r) {
      XF_TRY(recv_all(g->peer[r], offs[r].data(), (W + 1) * 8));
      for (int p = 0; p < W; ++p)
        if (offs[r][p] > offs[r][p + 1] || offs[r][0] != 0 || offs[r][W] > (1ull << 40))
          return xf::set_error(XF_EINVAL, "xf_group_alltoallv: rank %d sent slice offsets that "
                               "do not ascend", r);
      bufs[r].resize((size_t)offs[r][W]);
      if (offs[r][W]) XF_TRY(recv_all(g->peer[r], bufs[r].data(), bufs[r].size()));
    }
    for (int dst = 0; dst < W; ++dst) {
      std::vector<char> outb;
      for (int src = 0; src < W; ++src)
        outb.insert(outb.end(), bufs[src].begin() + offs[src][dst],
                    bufs[src].begin() + offs[src][dst + 1]);
      if (dst == 0) {
        if (outb.size() != ro[W])
          return xf::set_error(XF_EINVAL, "xf_group_alltoallv: rank 0 expected %zu bytes, the "
                               "ranks sent it %zu", ro[W], outb.size());
        if (!outb.empty()) memcpy(hrb, outb.data(), outb.size());
      } else {
        const uint64_t n = outb.size();
        XF_TRY(send_all(g->peer[dst], &n, 8));
        if (n) XF_TRY(send_all(g->peer[dst], outb.data(), outb.size()));
      }
    }
  } else {
    std::vector<uint64_t> off(W + 1);
    for (int p = 0; p <= W; ++p) off[p] = so[p];
    XF_TRY(send_all(g->peer[0], off.data(), (W + 1) * 8));
    if (so[W]) XF_TRY(send_all(g->peer[0], hsb, so[W]));
    uint64_t n = 0;
    XF_TRY(recv_all(g->peer[0], &n, 8));
    if (n != ro[W])
      return xf::set_error(XF_EINVAL, "xf_group_alltoallv: rank %d expected %zu bytes, got %llu",
                           g->rank, ro[W], (unsigned long long)n);
    if (n) XF_TRY(recv_all(g->peer[0], hrb, (size_t)n));
  }
  if (!host_buffers && ro[W]) XF_HIP(hipMemcpy(recv, g->hr.data(), ro[W], hipMemcpyHostToDevice));
  return XF_OK;
}
