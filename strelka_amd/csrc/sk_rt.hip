// sk_rt.hip -- the broker: one server process per GPU holding the device's context, caller processes as its clients (sk_rt.h).
//
//   client (a caller process, $STRELKA_AMD_BROKER=1)                      server (`sk_broker`, one per device, started by the first client)
//   ------------------------------------------------                      ------------------------------------------------------------
//   skrt::launch / memcpyAsync / ...  --> records in a ring (shared) -->  a thread per client: executes the records in order on that
//   skrt::streamSynchronize           --> SYNC record, futex wait   <--  client's own stream; hipStreamSynchronize, then done_seq + wake
//   skrt::hostMalloc                  --> memfd, same address both sides, page-locked by the server (hipHostRegister)
//
// Rendezvous: an abstract unix socket named by user, library build and device; the first client that finds nobody listening starts
// the server (setsid + exec of `sk_broker` beside this library) under a file lock; the server leaves when it has had no client for
// $STRELKA_AMD_BROKER_IDLE_S (20) seconds.  A client that dies is noticed by its server thread (the socket closes), which waits for
// the client's stream and frees what it held.
// The "host" backend ($STRELKA_AMD_BROKER_BACKEND=host: device memory = the server's heap, a launch = a call of a host function) exists
// for the no-GPU test tier: it exercises everything here but the HIP calls.

#include "sk_rt.h"

#include <atomic>
#include <cerrno>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <fcntl.h>
#include <linux/futex.h>
#include <poll.h>
#include <signal.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <unistd.h>

extern "C" int sk_version(void);

namespace skrt
{
namespace
{
bool env_remote()
{
    const char* e = std::getenv("STRELKA_AMD_BROKER");
    return e && e[0] && std::strcmp(e, "0") != 0;
}
}
bool g_remote = env_remote();
// (sk_broker_enable: a caller that wants the broker without exporting the variable -- the adapter, by default -- before sk_init)
void set_remote(const bool on) { g_remote = on; }

namespace
{
// ---------------------------------------------------------------------------------------------------------------------
// what both sides share

constexpr uint32_t MAGIC = 0x534b4252u; // "SKBR"
constexpr uint32_t PROTO = 4;
constexpr uint32_t RING_BYTES = 1u << 20;
constexpr uint64_t SLOT_BASE = 0x600000000000ull; // client i's page-locked segments live at SLOT_BASE + i * SLOT_BYTES in BOTH processes
constexpr uint64_t SLOT_BYTES = 1ull << 36;       // 64 GB of address space per client
constexpr int MAX_SLOTS = 1024;
constexpr uintptr_t STREAM_HANDLE = 0x534b0001u;  // what a client's hipStream_t is

enum : uint32_t { OP_PAD = 0, OP_LAUNCH, OP_MEMCPY, OP_MEMSET, OP_SYNC, OP_MALLOC, OP_FREE, OP_FUNC_ATTR, OP_HOST_MAP, OP_HOST_UNMAP, OP_BYE };
enum : uint32_t { MSG_HELLO = 1, MSG_SLOT, MSG_SLOT_OK, MSG_SLOT_RETRY, MSG_MAP, MSG_REFUSED, MSG_SEGS, MSG_SEG };

struct Rec
{
    uint32_t op, bytes; // bytes: the whole record, a multiple of 16
    uint32_t seq;       // for the ops the client waits on
    uint32_t pad;
};
struct RecLaunch
{
    Rec r;
    uint64_t fn_off;
    uint32_t grid[3], block[3];
    uint32_t lds, n_args;
    // uint32_t sizes[n_args] (padded to 16), then each argument's bytes at the next multiple of 16
};
struct RecCopy
{
    Rec r;
    uint64_t dst, src, bytes;
    uint32_t kind, value;
};
struct RecMem
{
    Rec r;
    uint64_t a, b;
    uint32_t c, d;
};

struct Ctl
{
    uint32_t magic, proto;
    alignas(64) std::atomic<uint64_t> head; // bytes the client has published
    alignas(64) std::atomic<uint64_t> tail; // bytes the server has executed
    alignas(64) std::atomic<uint32_t> server_idle; // futex: 1 while the server thread sleeps on an empty ring
    alignas(64) std::atomic<uint32_t> done_seq;    // futex: the last op the client may wait on that is complete
    std::atomic<uint32_t> client_waiting;
    std::atomic<int32_t> status; // the first error since the client last asked (a hipError_t)
    uint64_t result;             // of the last MALLOC
    uint64_t gpu_flag;           // address (the same in both processes) of the word a one-lane kernel stores a SYNC's sequence number to
    double t_first, t_reached, t_done; // ($STRELKA_AMD_BROKER_TIMING) of the last batch: first record seen, SYNC reached, stream waited for (CLOCK_MONOTONIC)
    char error_text[200];
    alignas(64) char ring[RING_BYTES];
};

struct Msg
{
    uint32_t magic, type;
    uint64_t a, b, c;
    char text[96];
};

inline size_t up16(const size_t n) { return (n + 15) & ~size_t(15); }

long futex(std::atomic<uint32_t>* addr, int op, uint32_t val, const timespec* ts)
{
    return syscall(SYS_futex, reinterpret_cast<uint32_t*>(addr), op, val, ts, nullptr, 0);
}
void futex_wake(std::atomic<uint32_t>* addr) { (void)futex(addr, FUTEX_WAKE, INT_MAX, nullptr); }
void futex_wait(std::atomic<uint32_t>* addr, uint32_t expected, long ms)
{
    timespec ts{ ms / 1000, (ms % 1000) * 1000000L };
    (void)futex(addr, FUTEX_WAIT, expected, &ts);
}

int send_msg(const int sock, const Msg& m, const int fd)
{
    iovec iov{ const_cast<Msg*>(&m), sizeof(Msg) };
    msghdr h{};
    h.msg_iov = &iov;
    h.msg_iovlen = 1;
    alignas(cmsghdr) char cbuf[CMSG_SPACE(sizeof(int))];
    if (fd >= 0) {
        std::memset(cbuf, 0, sizeof(cbuf));
        h.msg_control = cbuf;
        h.msg_controllen = sizeof(cbuf);
        cmsghdr* c = CMSG_FIRSTHDR(&h);
        c->cmsg_level = SOL_SOCKET;
        c->cmsg_type = SCM_RIGHTS;
        c->cmsg_len = CMSG_LEN(sizeof(int));
        std::memcpy(CMSG_DATA(c), &fd, sizeof(int));
    }
    for (;;) {
        const ssize_t n = sendmsg(sock, &h, MSG_NOSIGNAL);
        if (n == ssize_t(sizeof(Msg))) return 0;
        if (n < 0 && errno == EINTR) continue;
        return 1;
    }
}
// fd_out: the descriptor that came with the message, or -1
int recv_msg(const int sock, Msg& m, int* fd_out, const int timeout_ms)
{
    if (fd_out) *fd_out = -1;
    if (timeout_ms >= 0) {
        pollfd p{ sock, POLLIN, 0 };
        int r;
        do r = poll(&p, 1, timeout_ms);
        while (r < 0 && errno == EINTR);
        if (r <= 0) return 1;
    }
    iovec iov{ &m, sizeof(Msg) };
    msghdr h{};
    h.msg_iov = &iov;
    h.msg_iovlen = 1;
    alignas(cmsghdr) char cbuf[CMSG_SPACE(sizeof(int))];
    h.msg_control = cbuf;
    h.msg_controllen = sizeof(cbuf);
    ssize_t n;
    do n = recvmsg(sock, &h, MSG_CMSG_CLOEXEC);
    while (n < 0 && errno == EINTR);
    if (n != ssize_t(sizeof(Msg)) || m.magic != MAGIC) return 1;
    for (cmsghdr* c = CMSG_FIRSTHDR(&h); c; c = CMSG_NXTHDR(&h, c))
        if (c->cmsg_level == SOL_SOCKET && c->cmsg_type == SCM_RIGHTS && fd_out) std::memcpy(fd_out, CMSG_DATA(c), sizeof(int));
    return 0;
}
bool peer_gone(const int sock)
{
    char c;
    const ssize_t n = recv(sock, &c, 1, MSG_PEEK | MSG_DONTWAIT);
    if (n == 0) return true;
    return n < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR;
}

// this library as a loaded file: where it starts (kernels are named by offsets from here), and a tag of the build
struct LibInfo
{
    uintptr_t base = 0;
    std::string path, dir;
    uint64_t size = 0, mtime = 0;
    uint32_t tag = 0;
};
const LibInfo& lib_info()
{
    static LibInfo li = [] {
        LibInfo x;
        Dl_info d;
        if (dladdr(reinterpret_cast<const void*>(&sk_version), &d) && d.dli_fname) {
            x.base = reinterpret_cast<uintptr_t>(d.dli_fbase);
            char rp[PATH_MAX];
            x.path = realpath(d.dli_fname, rp) ? rp : d.dli_fname;
            const size_t s = x.path.rfind('/');
            x.dir = s == std::string::npos ? "." : x.path.substr(0, s);
            struct stat st;
            if (stat(x.path.c_str(), &st) == 0) {
                x.size = uint64_t(st.st_size);
                x.mtime = uint64_t(st.st_mtim.tv_sec) * 1000000000ull + uint64_t(st.st_mtim.tv_nsec);
            }
        }
        uint64_t hsh = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t n) {
            for (size_t i = 0; i < n; ++i) hsh = (hsh ^ static_cast<const unsigned char*>(p)[i]) * 1099511628211ull;
        };
        mix(x.path.data(), x.path.size());
        mix(&x.size, 8);
        mix(&x.mtime, 8);
        x.tag = uint32_t(hsh ^ (hsh >> 32));
        return x;
    }();
    return li;
}
std::string socket_name(const int device)
{
    if (const char* e = std::getenv("STRELKA_AMD_BROKER_SOCKET")) return std::string(e) + "." + std::to_string(device);
    char b[128];
    std::snprintf(b, sizeof(b), "strelka_amd_broker.%u.%08x.%d", unsigned(getuid()), lib_info().tag, device);
    return b;
}
socklen_t abstract_addr(sockaddr_un& a, const std::string& name)
{
    std::memset(&a, 0, sizeof(a));
    a.sun_family = AF_UNIX;
    const size_t n = std::min(name.size(), sizeof(a.sun_path) - 2);
    std::memcpy(a.sun_path + 1, name.data(), n); // (leading NUL: the abstract namespace, nothing to unlink)
    return socklen_t(offsetof(sockaddr_un, sun_path) + 1 + n);
}

// =====================================================================================================================
// client

struct Seg
{
    char* va = nullptr;
    size_t bytes = 0;
    int fd = -1; // (server: the segment's memfd, kept so that the next client of the slot can map the same pages)
};
struct Pending
{
    void* dst;
    const char* src;
    size_t bytes;
};
unsigned env_us(const char* name, const unsigned def)
{
    const char* e = std::getenv(name);
    return (e && e[0]) ? unsigned(std::strtoul(e, nullptr, 10)) : def;
}

struct Client
{
    // How a client waits: it sleeps on a futex the server thread wakes -- at once by default: with a caller process per core the
    // server thread needs the very core the client would poll on (profiles/r06_v5: every polling variant is slower at 16 callers on 16
    // cores).  For a box with idle cores: $STRELKA_AMD_BROKER_CLIENT_SPIN_US polls first, and with $STRELKA_AMD_BROKER_GPU_FLAG=1 the poll
    // also ends on a word that a one-lane kernel at the end of the client's stream stores into page-locked memory (the device says
    // "done" itself, no wake-up of a server thread on the way).
    unsigned spin_us = env_us("STRELKA_AMD_BROKER_CLIENT_SPIN_US", 0);
    bool use_gpu_flag = env_us("STRELKA_AMD_BROKER_GPU_FLAG", 0) != 0;
    bool eager_wake = env_us("STRELKA_AMD_BROKER_EAGER_WAKE", 0) != 0;
    volatile uint32_t* gpu_flag = nullptr;
    int sock = -1;
    int device = -1, device_count = 0;
    bool host_backend = false;
    Ctl* ctl = nullptr;
    uint64_t head = 0;
    uint32_t seq = 0;
    char* slot = nullptr;
    size_t slot_used = 0;
    std::vector<Seg> segs;
    Seg staging;
    size_t staging_used = 0;
    std::vector<Pending> pending;
    hipError_t last = hipSuccess;
    hipError_t text_for = hipSuccess; // the error `text` describes
    std::string text;
    bool dead = false;
};
Client g_cl;
// The C-ABI is called by one thread per process and so is this layer; the lock is for the caller that is not: two threads
// writing one ring would interleave records.  (Held across a wait: a client has one stream, its calls are serial by nature.)
std::recursive_mutex g_cl_mu;
#define SKRT_CLIENT_LOCK() std::lock_guard<std::recursive_mutex> skrt_client_lock_(g_cl_mu)

// $STRELKA_AMD_BROKER_TIMING: where a client's time inside this layer goes, on stderr when the process ends
struct ClientTiming
{
    bool on = std::getenv("STRELKA_AMD_BROKER_TIMING") != nullptr;
    double wait_other_s = 0, wait_max = 0, wait_long_s = 0;
    uint64_t waits_other = 0, waits_long = 0;
    double wait_s = 0, connect_s = 0, stage_copy_s = 0, leg_wake = 0, leg_submit = 0, leg_device = 0, leg_back = 0;
    uint64_t waits = 0, launches = 0, copies = 0, staged_bytes = 0, futex_sleeps = 0, flag_hits = 0, ring_full_spins = 0;
    double in_launch_s = 0, in_copy_s = 0, in_alloc_s = 0, in_status_s = 0;
    uint64_t allocs = 0, host_frees = 0, status_reads = 0;
    ~ClientTiming()
    {
        if (on && (waits || launches))
            std::fprintf(stderr, "strelka_amd broker client: connect=%.4f wait=%.4f waits=%llu futex_sleeps=%llu flag_hits=%llu launches=%llu copies=%llu staged_bytes=%llu stage_copy=%.4f in_launch=%.4f in_copy=%.4f in_alloc=%.4f allocs=%llu host_frees=%llu in_status=%.4f status_reads=%llu ring_full_spins=%llu leg_server_wake=%.4f leg_submit=%.4f leg_device=%.4f leg_client_wake=%.4f wait_not_sync=%.4f waits_not_sync=%llu wait_over_5ms=%.4f waits_over_5ms=%llu wait_max=%.4f\n", connect_s,
                         wait_s, (unsigned long long)waits, (unsigned long long)futex_sleeps, (unsigned long long)flag_hits, (unsigned long long)launches, (unsigned long long)copies,
                         (unsigned long long)staged_bytes, stage_copy_s, in_launch_s, in_copy_s, in_alloc_s, (unsigned long long)allocs, (unsigned long long)host_frees, in_status_s, (unsigned long long)status_reads,
                         (unsigned long long)ring_full_spins, leg_wake, leg_submit, leg_device, leg_back, wait_other_s, (unsigned long long)waits_other, wait_long_s,
                         (unsigned long long)waits_long, wait_max);
    }
} g_tm;
inline double tm_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

hipError_t cl_fail(const std::string& what, const hipError_t e = hipErrorUnknown)
{
    g_cl.text = what;
    g_cl.text_for = e;
    g_cl.last = e;
    return e;
}

void wake_server(Ctl* c)
{
    if (c->server_idle.load(std::memory_order_seq_cst)) {
        c->server_idle.store(0, std::memory_order_seq_cst);
        futex_wake(&c->server_idle);
    }
}

// room for a record of `bytes` (a multiple of 16) in one piece
char* ring_reserve(const uint32_t bytes)
{
    Client& cl = g_cl;
    Ctl* c = cl.ctl;
    unsigned spins = 0;
    for (;;) {
        const uint64_t tail = c->tail.load(std::memory_order_acquire);
        const uint32_t pos = uint32_t(cl.head % RING_BYTES);
        const uint32_t contiguous = RING_BYTES - pos;
        const uint32_t skip = contiguous < bytes ? contiguous : 0;
        if (cl.head + skip + bytes - tail <= RING_BYTES) {
            if (skip) {
                Rec* pad = reinterpret_cast<Rec*>(c->ring + pos);
                pad->op = OP_PAD;
                pad->bytes = skip;
                cl.head += skip;
            }
            return c->ring + (cl.head % RING_BYTES);
        }
        c->head.store(cl.head, std::memory_order_seq_cst);
        wake_server(c);
        ++g_tm.ring_full_spins;
        if (++spins > 200) {
            if (peer_gone(cl.sock)) {
                cl.dead = true;
                return nullptr;
            }
            usleep(50);
        }
    }
}
// Publishes a record.  The server thread is woken when the client is about to WAIT (`wake`), not when work merely exists: caller
// processes fill the box's cores, so a server thread woken early runs on the very core its client is still writing records from --
// the two take turns on it and the batch's submission takes ten times as long (profiles/r06_v6: "ring empty inside a batch" 0.55 s
// of a client's 1.1 s).  Woken at the wait, the server thread gets the core the client has just left.  $STRELKA_AMD_BROKER_EAGER_WAKE=1
// (a box with idle cores): wake at the first record, the device then starts while the client is still writing.
void ring_commit(const uint32_t bytes, const bool wake = false)
{
    Client& cl = g_cl;
    cl.head += bytes;
    cl.ctl->head.store(cl.head, std::memory_order_seq_cst);
    if (wake || cl.eager_wake || cl.head - cl.ctl->tail.load(std::memory_order_relaxed) > RING_BYTES / 2) wake_server(cl.ctl);
}

hipError_t take_status()
{
    Client& cl = g_cl;
    const int32_t s = cl.ctl->status.load(std::memory_order_acquire);
    if (s == 0) return hipSuccess;
    cl.text.assign(cl.ctl->error_text, strnlen(cl.ctl->error_text, sizeof(cl.ctl->error_text)));
    cl.ctl->error_text[0] = 0;
    cl.ctl->status.store(0, std::memory_order_release);
    cl.text_for = hipError_t(s);
    return hipError_t(s);
}

hipError_t wait_seq(const uint32_t seq, const bool sync = false)
{
    Client& cl = g_cl;
    Ctl* c = cl.ctl;
    unsigned spins = 0;
    int idle_rounds = 0;
    const double t_begin = g_tm.on ? tm_now() : 0.0;
    struct Lap
    {
        double t0;
        bool sync;
        ~Lap()
        {
            if (!g_tm.on) return;
            const double dt = tm_now() - t0;
            g_tm.wait_s += dt, ++g_tm.waits;
            if (!sync) g_tm.wait_other_s += dt, ++g_tm.waits_other;
            if (dt > g_tm.wait_max) g_tm.wait_max = dt;
            if (dt > 0.005) g_tm.wait_long_s += dt, ++g_tm.waits_long;
        }
    } lap{ t_begin, sync };
    const bool flag_counts = sync && cl.gpu_flag != nullptr;
    double spin_until = 0;
    for (;;) {
        const uint32_t d = c->done_seq.load(std::memory_order_acquire);
        if (int32_t(d - seq) >= 0) break;
        if (flag_counts && int32_t(*cl.gpu_flag - seq) >= 0) {
            std::atomic_thread_fence(std::memory_order_acquire);
            ++g_tm.flag_hits;
            break;
        }
        if (++spins < 64) {
            __builtin_ia32_pause();
            continue;
        }
        if (cl.spin_us) {
            const double now = tm_now();
            if (spin_until == 0) spin_until = now + 1e-6 * cl.spin_us;
            if (now < spin_until) {
                for (int i = 0; i < 32; ++i) __builtin_ia32_pause();
                continue;
            }
        }
        c->client_waiting.store(1, std::memory_order_seq_cst);
        ++g_tm.futex_sleeps;
        if (int32_t(c->done_seq.load(std::memory_order_seq_cst) - seq) < 0) futex_wait(&c->done_seq, d, 500);
        c->client_waiting.store(0, std::memory_order_relaxed);
        if (int32_t(c->done_seq.load(std::memory_order_acquire) - seq) >= 0) break;
        if (++idle_rounds % 2 == 0 && peer_gone(cl.sock)) {
            cl.dead = true;
            return cl_fail("strelka_amd: the broker process is gone");
        }
    }
    return take_status();
}

bool in_shared(const void* p, const size_t bytes)
{
    const char* q = static_cast<const char*>(p);
    for (const Seg& s : g_cl.segs)
        if (q >= s.va && q + bytes <= s.va + s.bytes) return true;
    return false;
}

template <typename R> R* new_rec(const uint32_t op, const uint32_t bytes)
{
    if (g_cl.dead || !g_cl.ctl) return nullptr;
    char* p = ring_reserve(bytes);
    if (!p) return nullptr;
    R* r = reinterpret_cast<R*>(p);
    std::memset(static_cast<void*>(r), 0, sizeof(R));
    r->r.op = op;
    r->r.bytes = bytes;
    return r;
}

hipError_t cl_sync()
{
    Client& cl = g_cl;
    RecMem* r = new_rec<RecMem>(OP_SYNC, uint32_t(up16(sizeof(RecMem))));
    if (!r) return cl_fail("strelka_amd: no broker connection");
    const uint32_t seq = r->r.seq = ++cl.seq;
    const double t_commit = g_tm.on ? tm_now() : 0.0;
    ring_commit(r->r.bytes, true);
    const hipError_t e = wait_seq(seq, true);
    if (g_tm.on && !cl.host_backend) {
        // the four legs of a wait: the server thread's wake-up, its submissions, the device, this process's wake-up
        const double t_back = tm_now(), tf = cl.ctl->t_first, tr = cl.ctl->t_reached, td = cl.ctl->t_done;
        if (tf > 0 && td >= tr && tr >= tf) {
            g_tm.leg_wake += std::max(0.0, tf - std::min(tf, t_commit)) ;
            g_tm.leg_submit += tr - std::max(tf, t_commit);
            g_tm.leg_device += td - tr;
            g_tm.leg_back += std::max(0.0, t_back - td);
        }
    }
    // copies to the caller's pageable arrays are complete now
    if (!cl.pending.empty()) {
        const double t0 = g_tm.on ? tm_now() : 0.0;
        for (const Pending& p : cl.pending) {
            std::memcpy(p.dst, p.src, p.bytes);
            g_tm.staged_bytes += p.bytes;
        }
        if (g_tm.on) g_tm.stage_copy_s += tm_now() - t0;
        cl.pending.clear();
    }
    cl.staging_used = 0;
    return e;
}

hipError_t cl_host_map(Seg* out, const size_t want)
{
    Client& cl = g_cl;
    const size_t bytes = (want + 65535) & ~size_t(65535);
    if (cl.slot_used + bytes > SLOT_BYTES) return cl_fail("strelka_amd: a client's page-locked address range is used up", hipErrorOutOfMemory);
    char* va = cl.slot + cl.slot_used;
    const int fd = memfd_create("strelka_amd_pinned", MFD_CLOEXEC);
    if (fd < 0 || ftruncate(fd, off_t(bytes)) != 0) {
        if (fd >= 0) close(fd);
        return cl_fail("strelka_amd: memfd for page-locked memory failed", hipErrorOutOfMemory);
    }
    void* p = mmap(va, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0); // (over this client's own reservation)
    if (p != va) {
        close(fd);
        return cl_fail("strelka_amd: mapping page-locked memory failed", hipErrorOutOfMemory);
    }
    Msg m{ MAGIC, MSG_MAP, uint64_t(reinterpret_cast<uintptr_t>(va)), uint64_t(bytes), 0, {} };
    const int sent = send_msg(cl.sock, m, fd);
    close(fd);
    if (sent) return cl_fail("strelka_amd: the broker connection is closed");
    RecMem* r = new_rec<RecMem>(OP_HOST_MAP, uint32_t(up16(sizeof(RecMem))));
    if (!r) return cl_fail("strelka_amd: no broker connection");
    r->a = uint64_t(reinterpret_cast<uintptr_t>(va));
    r->b = bytes;
    const uint32_t seq = r->r.seq = ++cl.seq;
    ring_commit(r->r.bytes, true);
    const hipError_t e = wait_seq(seq);
    if (e != hipSuccess) {
        (void)mmap(va, bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED, -1, 0);
        return e;
    }
    cl.slot_used += bytes;
    out->va = va;
    out->bytes = bytes;
    cl.segs.push_back(*out);
    return hipSuccess;
}

hipError_t cl_host_unmap(const Seg s)
{
    Client& cl = g_cl;
    for (size_t i = 0; i < cl.segs.size(); ++i)
        if (cl.segs[i].va == s.va) {
            cl.segs.erase(cl.segs.begin() + long(i));
            break;
        }
    RecMem* r = new_rec<RecMem>(OP_HOST_UNMAP, uint32_t(up16(sizeof(RecMem))));
    if (!r) return cl_fail("strelka_amd: no broker connection");
    r->a = uint64_t(reinterpret_cast<uintptr_t>(s.va));
    r->b = s.bytes;
    const uint32_t seq = r->r.seq = ++cl.seq;
    ring_commit(r->r.bytes, true);
    const hipError_t e = wait_seq(seq); // (the server has waited for the stream before it let go of the pages)
    (void)mmap(s.va, s.bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED, -1, 0);
    return e;
}

// `bytes` of the staging segment; waits for the stream (which empties the segment) or grows it when there is no room
hipError_t staging_take(const size_t bytes, char** out)
{
    Client& cl = g_cl;
    const size_t need = (bytes + 255) & ~size_t(255);
    if (cl.staging_used + need > cl.staging.bytes) {
        if (cl.staging_used || !cl.pending.empty()) {
            const hipError_t e = cl_sync();
            if (e != hipSuccess) return e;
        }
        if (need > cl.staging.bytes) {
            if (cl.staging.va) {
                const hipError_t e = cl_host_unmap(cl.staging);
                cl.staging = Seg();
                if (e != hipSuccess) return e;
            }
            const hipError_t e = cl_host_map(&cl.staging, std::max<size_t>(need + need / 2, size_t(8) << 20));
            if (e != hipSuccess) return e;
        }
    }
    *out = cl.staging.va + cl.staging_used;
    cl.staging_used += need;
    return hipSuccess;
}

int connect_once(const std::string& name)
{
    const int s = socket(AF_UNIX, SOCK_SEQPACKET | SOCK_CLOEXEC, 0);
    if (s < 0) return -1;
    sockaddr_un a;
    const socklen_t len = abstract_addr(a, name);
    if (connect(s, reinterpret_cast<sockaddr*>(&a), len) == 0) return s;
    close(s);
    return -1;
}

// start `sk_broker` for the device, detached from this process (own session, no descriptor of ours, output to a log file)
int spawn_server(const int device, const std::string& name, std::string* why)
{
    const std::string exe = lib_info().dir + "/sk_broker";
    if (access(exe.c_str(), X_OK) != 0) {
        *why = "strelka_amd: " + exe + " is missing (python -m strelka_amd.build)";
        return 1;
    }
    const pid_t mid = fork();
    if (mid < 0) {
        *why = "strelka_amd: fork failed";
        return 1;
    }
    if (mid == 0) {
        if (setsid() < 0) _exit(1);
        const pid_t srv = fork();
        if (srv != 0) _exit(srv < 0 ? 1 : 0);
        const char* log = std::getenv("STRELKA_AMD_BROKER_LOG");
        char def[128];
        std::snprintf(def, sizeof(def), "/tmp/strelka_amd_broker.%u.%d.log", unsigned(getuid()), device);
        const int in = open("/dev/null", O_RDONLY);
        int out = open(log ? log : def, O_WRONLY | O_CREAT | O_APPEND, 0600);
        if (out < 0) out = open("/dev/null", O_WRONLY);
        if (in >= 0) dup2(in, 0);
        if (out >= 0) {
            dup2(out, 1);
            dup2(out, 2);
        }
        long maxfd = sysconf(_SC_OPEN_MAX);
        if (maxfd < 0 || maxfd > 65536) maxfd = 65536;
        for (int fd = 3; fd < int(maxfd); ++fd) close(fd);
        const std::string dev = std::to_string(device);
        const char* argv[] = { exe.c_str(), "--device", dev.c_str(), "--socket", name.c_str(), nullptr };
        execv(exe.c_str(), const_cast<char* const*>(argv));
        _exit(127);
    }
    int st = 0;
    while (waitpid(mid, &st, 0) < 0 && errno == EINTR) {}
    return 0;
}

int connect_or_spawn(const int device, std::string* why)
{
    const std::string name = socket_name(device);
    int s = connect_once(name);
    if (s >= 0) return s;
    if (const char* e = std::getenv("STRELKA_AMD_BROKER_NO_SPAWN"))
        if (e[0] && e[0] != '0') {
            *why = "strelka_amd: no broker is listening on @" + name + " ($STRELKA_AMD_BROKER_NO_SPAWN)";
            return -1;
        }
    char lockp[160];
    std::snprintf(lockp, sizeof(lockp), "/tmp/strelka_amd_broker.%u.%08x.%d.lock", unsigned(getuid()), lib_info().tag, device);
    const int lock = open(lockp, O_RDWR | O_CREAT | O_CLOEXEC, 0600);
    if (lock >= 0) (void)flock(lock, LOCK_EX);
    s = connect_once(name);
    if (s < 0) {
        if (spawn_server(device, name, why) == 0) {
            const auto t0 = std::chrono::steady_clock::now();
            while (s < 0 && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(120)) {
                usleep(20000);
                s = connect_once(name);
            }
            if (s < 0) *why = "strelka_amd: the broker for device " + std::to_string(device) + " did not come up (see its log under /tmp)";
        }
    }
    if (lock >= 0) {
        (void)flock(lock, LOCK_UN);
        close(lock);
    }
    return s;
}

struct SubAllocator
{
    struct Slab
    {
        char* base;
        size_t size, used;
    };
    struct Piece
    {
        char* p;
        size_t size;
    };
    size_t slab_bytes;
    std::vector<Slab> slabs;
    std::vector<Piece> live, spare;
    explicit SubAllocator(const size_t slab) : slab_bytes(slab) {}
    static size_t round(const size_t n) { return (n + 4095) & ~size_t(4095); }
    bool direct(const size_t need) const { return need >= slab_bytes / 2; }
    /// a piece from the spare list or from a slab with room; nullptr: the caller adds a slab (add_slab) and asks again
    char* take(const size_t need)
    {
        for (size_t i = 0; i < spare.size(); ++i)
            if (spare[i].size >= need && spare[i].size <= need + need / 2 + 65536) {
                const Piece pc = spare[i];
                spare.erase(spare.begin() + long(i));
                live.push_back(pc);
                return pc.p;
            }
        for (Slab& sl : slabs)
            if (sl.used + need <= sl.size) {
                char* p = sl.base + sl.used;
                sl.used += need;
                live.push_back(Piece{ p, need });
                return p;
            }
        return nullptr;
    }
    void add_slab(void* base, const size_t size) { slabs.push_back(Slab{ static_cast<char*>(base), size, 0 }); }
    bool give_back(void* p)
    {
        for (size_t i = 0; i < live.size(); ++i)
            if (live[i].p == p) {
                spare.push_back(live[i]);
                live.erase(live.begin() + long(i));
                return true;
            }
        return false;
    }
};
SubAllocator g_dev_alloc(size_t(env_us("STRELKA_AMD_BROKER_DEVICE_SLAB_MB", 256)) << 20);
SubAllocator g_pin_alloc(size_t(env_us("STRELKA_AMD_BROKER_PINNED_SLAB_MB", 32)) << 20);

void reset_suballocators() // (the connection is gone: so is everything its slabs were carved from)
{
    for (SubAllocator* a : { &g_dev_alloc, &g_pin_alloc }) {
        a->slabs.clear();
        a->live.clear();
        a->spare.clear();
    }
}

void cl_close()
{
    Client& cl = g_cl;
    const Client knobs = Client();
    reset_suballocators();
    if (cl.sock >= 0) close(cl.sock); // (the server thread sees the socket close and frees what this client held)
    if (cl.ctl) munmap(cl.ctl, sizeof(Ctl));
    if (cl.slot) munmap(cl.slot, SLOT_BYTES);
    cl = knobs;
}
} // namespace

// ---------------------------------------------------------------------------------------------------------------------
// client entry points (sk_rt.h, sk_context.hip)

static int r_connect_once(int device, std::string* why, bool* refused);

int r_connect(const int device, std::string* why)
{
    Client& cl = g_cl;
    if (cl.sock >= 0 && cl.device == device && !cl.dead) return 0;
    if (cl.sock >= 0) cl_close();
    const double t_begin = tm_now();
    struct Lap
    {
        double t0;
        ~Lap() { g_tm.connect_s += tm_now() - t0; }
    } lap{ t_begin };
    // (a server that was just leaving -- idle for its 20 s -- may accept the connection and be gone before the handshake: the next
    // attempt finds nobody listening and starts a new one)
    for (int attempt = 0; attempt < 4; ++attempt) {
        bool refused = false;
        why->clear();
        if (r_connect_once(device, why, &refused) == 0) return 0;
        if (refused) return 1;
        usleep(50000);
    }
    return 1;
}

static int r_connect_once(const int device, std::string* why, bool* refused)
{
    Client& cl = g_cl;
    const int s = connect_or_spawn(device, why);
    if (s < 0) {
        if (why->empty()) *why = "strelka_amd: cannot reach the broker";
        return 1;
    }
    const int fd = memfd_create("strelka_amd_ring", MFD_CLOEXEC);
    if (fd < 0 || ftruncate(fd, off_t(sizeof(Ctl))) != 0) {
        *why = "strelka_amd: memfd for the command ring failed";
        close(s);
        return 1;
    }
    void* p = mmap(nullptr, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (p == MAP_FAILED) {
        *why = "strelka_amd: mapping the command ring failed";
        close(fd);
        close(s);
        return 1;
    }
    Ctl* c = new (p) Ctl();
    c->magic = MAGIC;
    c->proto = PROTO;
    Msg m{ MAGIC, MSG_HELLO, lib_info().size, lib_info().mtime, (uint64_t(PROTO) << 32) | uint64_t(uint32_t(getpid())), {} };
    int rc = send_msg(s, m, fd);
    close(fd);
    char* slot = nullptr;
    std::vector<Seg> recycled;
    size_t recycled_used = 0;
    for (int attempt = 0; !rc && attempt < 64; ++attempt) {
        Msg r;
        if (recv_msg(s, r, nullptr, 120000)) {
            rc = 1;
            break;
        }
        if (r.type == MSG_REFUSED) {
            *why = std::string("strelka_amd: the broker refused this client: ") + std::string(r.text, strnlen(r.text, sizeof(r.text)));
            rc = 2;
            *refused = true;
            break;
        }
        if (r.type != MSG_SLOT) {
            rc = 1;
            break;
        }
        void* want = reinterpret_cast<void*>(uintptr_t(r.a));
        void* got = mmap(want, SLOT_BYTES, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED_NOREPLACE, -1, 0);
        if (got == want) {
            slot = static_cast<char*>(got);
            cl.device_count = int(r.b);
            cl.host_backend = r.c != 0;
            Msg ok{ MAGIC, MSG_SLOT_OK, 0, 0, 0, {} };
            rc = send_msg(s, ok, -1);
            // the slot's page-locked segments as the client before this one left them: still mapped and page-locked in the server, so
            // this client maps the same pages at the same addresses and has its slabs without a registration (hipHostRegister costs
            // ~4 ms per 32 MB and serialises in the server: profiles/r06_v17, 17 allocation round trips per caller = 0.1-0.5 s)
            Msg sg;
            if (!rc && (recv_msg(s, sg, nullptr, 120000) || sg.type != MSG_SEGS)) rc = 1;
            if (!rc) {
                recycled_used = size_t(sg.b);
                for (uint64_t i = 0; i < sg.a && !rc; ++i) {
                    Msg one;
                    int sfd = -1;
                    if (recv_msg(s, one, &sfd, 120000) || one.type != MSG_SEG || sfd < 0) {
                        rc = 1;
                    } else {
                        void* va = reinterpret_cast<void*>(uintptr_t(one.a));
                        void* mp = mmap(va, size_t(one.b), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, sfd, 0);
                        if (mp != va) rc = 1;
                        else recycled.push_back(Seg{ static_cast<char*>(va), size_t(one.b), -1 });
                    }
                    if (sfd >= 0) close(sfd);
                }
            }
            break;
        }
        if (got != MAP_FAILED) munmap(got, SLOT_BYTES);
        Msg again{ MAGIC, MSG_SLOT_RETRY, 0, 0, 0, {} };
        rc = send_msg(s, again, -1);
    }
    if (rc || !slot) {
        if (why->empty()) *why = "strelka_amd: the handshake with the broker failed";
        munmap(p, sizeof(Ctl));
        if (slot) munmap(slot, SLOT_BYTES);
        close(s);
        return 1;
    }
    cl.sock = s;
    cl.device = device;
    cl.ctl = c;
    cl.slot = slot;
    cl.slot_used = recycled_used;
    cl.head = 0;
    cl.seq = 0;
    for (const Seg& sg : recycled) {
        cl.segs.push_back(sg);
        if (!cl.staging.va && sg.bytes >= (size_t(8) << 20) && sg.bytes != g_pin_alloc.slab_bytes) cl.staging = sg; // (the staging segment of the client before)
        else g_pin_alloc.add_slab(sg.va, sg.bytes);
    }
    if (cl.use_gpu_flag && !cl.host_backend) { // a page-locked word of this client's for the device to say "done" in
        Seg fs;
        if (cl_host_map(&fs, 4096) == hipSuccess) {
            cl.gpu_flag = reinterpret_cast<volatile uint32_t*>(fs.va);
            *cl.gpu_flag = 0;
            c->gpu_flag = uint64_t(reinterpret_cast<uintptr_t>(fs.va));
        }
    }
    return 0;
}

// An entry point that is about to fill its input block and then submit calls this first: the server thread's wake-up (tens of
// microseconds from a sleep) then runs beside the packing instead of after it.
void r_wake_hint()
{
    if (g_cl.eager_wake && g_cl.ctl && !g_cl.dead) wake_server(g_cl.ctl);
}
// Work was submitted and the caller goes on with host work of its own: the server thread starts on the records now.  ($STRELKA_AMD_BROKER_
// LAZY_KICK=1: not before the client waits -- on a box whose cores are all taken by callers the server thread otherwise runs beside its
// client instead of in its place.)
void r_kick()
{
    static const bool lazy = env_us("STRELKA_AMD_BROKER_LAZY_KICK", 0) != 0;
    if (!lazy && g_cl.ctl && !g_cl.dead) wake_server(g_cl.ctl);
}
void r_disconnect() { cl_close(); }

namespace
{
// A client that ends normally says so: its server thread then returns its stream and its device memory to the server's pools at
// once, for the caller processes that start next (without the word the thread notices the closed socket at its next 50 ms look).
struct ByeAtExit
{
    ~ByeAtExit()
    {
        if (!g_remote || !g_cl.ctl || g_cl.dead) return;
        if (RecMem* r = new_rec<RecMem>(OP_BYE, uint32_t(up16(sizeof(RecMem))))) ring_commit(r->r.bytes, true);
        g_cl.dead = true; // (whatever a later static destructor asks for fails at once instead of waiting for a thread that is gone)
    }
} g_bye_at_exit;
}
int r_device_count(std::string* why)
{
    if (g_cl.sock < 0 && r_connect(0, why)) return 0;
    return g_cl.device_count;
}
bool r_host_backend() { return g_cl.host_backend; }

hipStream_t r_stream() { return reinterpret_cast<hipStream_t>(STREAM_HANDLE); }
// the text that belongs to error `e`: of a failure reported by this layer or by the broker; "" for an error of anything else (the device
// library's scans, say, called in a client by mistake) -- the runtime's own string is used then
const char* r_error_text(const hipError_t e) { return (e != hipSuccess && e == g_cl.text_for) ? g_cl.text.c_str() : ""; }

static inline bool own_stream(hipStream_t st) { return st == r_stream(); }
static hipError_t foreign_stream()
{
    return cl_fail("strelka_amd: a broker client has one stream, its own -- the *_dev entry points on a caller's stream need a process with a GPU context "
                   "(unset STRELKA_AMD_BROKER)",
                   hipErrorNotSupported);
}

// ---- allocation.  Every request that reaches the server is a round trip, and the server's own call (hipMalloc; for page-locked memory
// mmap + hipHostRegister) runs under locks that all of its client threads share: sixteen caller processes starting together spent
// ~0.45 s each waiting for their ~60 grow-only buffers (profiles/r06_v8: 7.5 of 13 s of waits were not waits for the device).  A client
// therefore asks for memory by the SLAB -- device memory 256 MB, page-locked memory 32 MB at a time, from the server's pool of
// departed clients' blocks when there is one -- and hands pieces out itself; a freed piece goes on the client's own list (its buffers
// only grow: a freed piece is followed by a larger request).  Requests of half a slab or more go to the server as they are.
namespace
{

hipError_t server_malloc(void** p, const size_t bytes)
{
    *p = nullptr;
    RecMem* r = new_rec<RecMem>(OP_MALLOC, uint32_t(up16(sizeof(RecMem))));
    if (!r) return cl_fail("strelka_amd: no broker connection");
    r->a = bytes;
    const uint32_t seq = r->r.seq = ++g_cl.seq;
    ring_commit(r->r.bytes, true);
    const hipError_t e = wait_seq(seq);
    if (e != hipSuccess) return e;
    *p = reinterpret_cast<void*>(uintptr_t(g_cl.ctl->result));
    return *p ? hipSuccess : cl_fail("strelka_amd: the broker could not allocate device memory", hipErrorOutOfMemory);
}
}

hipError_t r_malloc(void** p, const size_t bytes)
{
    SKRT_CLIENT_LOCK();
    struct Lap
    {
        double t0 = g_tm.on ? tm_now() : 0.0;
        ~Lap()
        {
            if (g_tm.on) g_tm.in_alloc_s += tm_now() - t0, ++g_tm.allocs;
        }
    } lap;
    *p = nullptr;
    const size_t need = SubAllocator::round(bytes ? bytes : 1);
    if (g_dev_alloc.direct(need)) return server_malloc(p, need);
    char* q = g_dev_alloc.take(need);
    if (!q) {
        void* slab = nullptr;
        const hipError_t e = server_malloc(&slab, g_dev_alloc.slab_bytes);
        if (e != hipSuccess) return e;
        g_dev_alloc.add_slab(slab, g_dev_alloc.slab_bytes);
        q = g_dev_alloc.take(need);
    }
    *p = q;
    return q ? hipSuccess : cl_fail("strelka_amd: device slab allocation failed", hipErrorOutOfMemory);
}
hipError_t r_free(void* p)
{
    SKRT_CLIENT_LOCK();
    struct Lap
    {
        double t0 = g_tm.on ? tm_now() : 0.0;
        ~Lap()
        {
            if (g_tm.on) g_tm.in_alloc_s += tm_now() - t0, ++g_tm.allocs;
        }
    } lap;
    if (!p) return hipSuccess;
    if (g_dev_alloc.give_back(p)) return hipSuccess; // (a piece of a slab: kept for this client's next request; the stream's order protects it)
    RecMem* r = new_rec<RecMem>(OP_FREE, uint32_t(up16(sizeof(RecMem))));
    if (!r) return cl_fail("strelka_amd: no broker connection");
    r->a = uint64_t(reinterpret_cast<uintptr_t>(p));
    ring_commit(r->r.bytes); // (in order behind everything that uses the memory; nothing to wait for)
    return hipSuccess;
}
hipError_t r_host_malloc(void** p, const size_t bytes)
{
    SKRT_CLIENT_LOCK();
    struct Lap
    {
        double t0 = g_tm.on ? tm_now() : 0.0;
        ~Lap()
        {
            if (g_tm.on) g_tm.in_alloc_s += tm_now() - t0, ++g_tm.allocs;
        }
    } lap;
    *p = nullptr;
    const size_t need = SubAllocator::round(bytes ? bytes : 1);
    Seg s;
    if (g_pin_alloc.direct(need)) {
        if (char* big = g_pin_alloc.take(need)) { // (a large segment the slot's last client left)
            *p = big;
            return hipSuccess;
        }
        const hipError_t e = cl_host_map(&s, need);
        if (e == hipSuccess) *p = s.va;
        return e;
    }
    char* q = g_pin_alloc.take(need);
    if (!q) {
        const hipError_t e = cl_host_map(&s, g_pin_alloc.slab_bytes);
        if (e != hipSuccess) return e;
        g_pin_alloc.add_slab(s.va, s.bytes);
        q = g_pin_alloc.take(need);
    }
    *p = q;
    return q ? hipSuccess : cl_fail("strelka_amd: page-locked slab allocation failed", hipErrorOutOfMemory);
}
hipError_t r_host_free(void* p)
{
    SKRT_CLIENT_LOCK();
    struct Lap
    {
        double t0 = g_tm.on ? tm_now() : 0.0;
        ~Lap()
        {
            if (g_tm.on) g_tm.in_alloc_s += tm_now() - t0, ++g_tm.allocs;
        }
    } lap;
    if (!p) return hipSuccess;
    // A piece that is handed out again may still be read or written by kernels of the stream the old owner queued: the runtime's own
    // hipHostFree waits for the device, so does this one (frees of page-locked memory happen when a buffer grows: a handful per process).
    ++g_tm.host_frees;
    if (g_pin_alloc.give_back(p)) return cl_sync();
    for (const Seg& s : g_cl.segs)
        if (s.va == p) return cl_host_unmap(s);
    return cl_fail("strelka_amd: hostFree of a pointer that is not page-locked memory of this client", hipErrorInvalidValue);
}
hipError_t r_memcpy_async(void* dst, const void* src, const size_t bytes, const hipMemcpyKind kind, hipStream_t st)
{
    SKRT_CLIENT_LOCK();
    struct Lap
    {
        double t0 = g_tm.on ? tm_now() : 0.0;
        ~Lap()
        {
            if (g_tm.on) g_tm.in_copy_s += tm_now() - t0;
        }
    } lap;
    if (!own_stream(st)) return foreign_stream();
    if (bytes == 0) return hipSuccess;
    Client& cl = g_cl;
    uint64_t d = uint64_t(reinterpret_cast<uintptr_t>(dst)), s = uint64_t(reinterpret_cast<uintptr_t>(src));
    if (kind == hipMemcpyHostToDevice && !in_shared(src, bytes)) {
        char* stg;
        const hipError_t e = staging_take(bytes, &stg);
        if (e != hipSuccess) return e;
        const double t0 = g_tm.on ? tm_now() : 0.0;
        std::memcpy(stg, src, bytes);
        if (g_tm.on) g_tm.stage_copy_s += tm_now() - t0, g_tm.staged_bytes += bytes;
        s = uint64_t(reinterpret_cast<uintptr_t>(stg));
    } else if (kind == hipMemcpyDeviceToHost && !in_shared(dst, bytes)) {
        char* stg;
        const hipError_t e = staging_take(bytes, &stg);
        if (e != hipSuccess) return e;
        cl.pending.push_back(Pending{ dst, stg, bytes });
        d = uint64_t(reinterpret_cast<uintptr_t>(stg));
    } else if (kind != hipMemcpyHostToDevice && kind != hipMemcpyDeviceToHost && kind != hipMemcpyDeviceToDevice) {
        return cl_fail("strelka_amd: a broker client names the direction of every copy", hipErrorInvalidValue);
    }
    RecCopy* r = new_rec<RecCopy>(OP_MEMCPY, uint32_t(up16(sizeof(RecCopy))));
    if (!r) return cl_fail("strelka_amd: no broker connection");
    r->dst = d;
    r->src = s;
    r->bytes = bytes;
    r->kind = uint32_t(kind);
    ring_commit(r->r.bytes);
    ++g_tm.copies;
    return hipSuccess;
}
hipError_t r_memset_async(void* dst, const int value, const size_t bytes, hipStream_t st)
{
    SKRT_CLIENT_LOCK();
    if (!own_stream(st)) return foreign_stream();
    if (bytes == 0) return hipSuccess;
    RecCopy* r = new_rec<RecCopy>(OP_MEMSET, uint32_t(up16(sizeof(RecCopy))));
    if (!r) return cl_fail("strelka_amd: no broker connection");
    r->dst = uint64_t(reinterpret_cast<uintptr_t>(dst));
    r->bytes = bytes;
    r->value = uint32_t(value);
    ring_commit(r->r.bytes);
    return hipSuccess;
}
hipError_t r_stream_synchronize(hipStream_t st)
{
    SKRT_CLIENT_LOCK();
    if (!own_stream(st)) return foreign_stream();
    return cl_sync();
}
hipError_t r_get_last_error()
{
    SKRT_CLIENT_LOCK();
    struct Lap
    {
        double t0 = g_tm.on ? tm_now() : 0.0;
        ~Lap()
        {
            if (g_tm.on) g_tm.in_status_s += tm_now() - t0, ++g_tm.status_reads;
        }
    } lap;
    Client& cl = g_cl;
    if (cl.last != hipSuccess) { // an error of a call made here (a refused stream, a lost connection): reported once, like the runtime's
        const hipError_t e = cl.last;
        cl.last = hipSuccess;
        return e;
    }
    return cl.ctl ? take_status() : hipSuccess;
}
hipError_t r_func_set_attribute(const void* fn, const hipFuncAttribute attr, const int value)
{
    SKRT_CLIENT_LOCK();
    RecMem* r = new_rec<RecMem>(OP_FUNC_ATTR, uint32_t(up16(sizeof(RecMem))));
    if (!r) return cl_fail("strelka_amd: no broker connection");
    r->a = uint64_t(reinterpret_cast<uintptr_t>(fn) - lib_info().base);
    r->c = uint32_t(attr);
    r->d = uint32_t(value);
    const uint32_t seq = r->r.seq = ++g_cl.seq;
    ring_commit(r->r.bytes, true);
    return wait_seq(seq);
}
void r_launch(const void* fn, const dim3 grid, const dim3 block, const size_t lds_bytes, hipStream_t st, void** args, const uint32_t* sizes, const int n_args)
{
    SKRT_CLIENT_LOCK();
    struct Lap
    {
        double t0 = g_tm.on ? tm_now() : 0.0;
        ~Lap()
        {
            if (g_tm.on) g_tm.in_launch_s += tm_now() - t0;
        }
    } lap;
    if (!own_stream(st)) {
        (void)foreign_stream();
        return;
    }
    size_t bytes = up16(sizeof(RecLaunch)) + up16(4 * size_t(n_args));
    for (int i = 0; i < n_args; ++i) bytes += up16(sizes[i]);
    if (bytes > RING_BYTES / 4) {
        (void)cl_fail("strelka_amd: a kernel's parameter block is too large for the broker's ring", hipErrorInvalidValue);
        return;
    }
    RecLaunch* r = new_rec<RecLaunch>(OP_LAUNCH, uint32_t(bytes));
    if (!r) {
        (void)cl_fail("strelka_amd: no broker connection");
        return;
    }
    r->fn_off = uint64_t(reinterpret_cast<uintptr_t>(fn) - lib_info().base);
    r->grid[0] = grid.x, r->grid[1] = grid.y, r->grid[2] = grid.z;
    r->block[0] = block.x, r->block[1] = block.y, r->block[2] = block.z;
    r->lds = uint32_t(lds_bytes);
    r->n_args = uint32_t(n_args);
    char* p = reinterpret_cast<char*>(r) + up16(sizeof(RecLaunch));
    std::memcpy(p, sizes, 4 * size_t(n_args));
    p += up16(4 * size_t(n_args));
    for (int i = 0; i < n_args; ++i) {
        std::memcpy(p, args[i], sizes[i]);
        p += up16(sizes[i]);
    }
    ring_commit(uint32_t(bytes));
    ++g_tm.launches;
}

// =====================================================================================================================
// server

namespace
{
struct Server
{
    int device = 0;
    bool host_backend = false;
    int device_count = 1;
    std::mutex mu;
    std::vector<bool> slot_taken = std::vector<bool>(MAX_SLOTS, false);
    // slots of departed clients whose page-locked segments are still mapped and registered here, for the next client
    struct Recycled
    {
        int slot;
        std::vector<Seg> segs;
        size_t used; // bytes of the slot's address range handed out so far
    };
    std::vector<Recycled> recycled;
    uint64_t recycled_given = 0;
    std::atomic<int> clients{ 0 };
    std::atomic<int64_t> served{ 0 };
    std::chrono::steady_clock::time_point last_client = std::chrono::steady_clock::now();
};
Server g_srv;

// Device memory of clients that have left, kept for the clients to come: a workflow's caller processes come and go by the hundred
// (a genome is ~260 segments), each asking for the same buffers -- from the pool a block costs a map look-up instead of the driver's
// allocation (and hipFree, which waits for every stream of the device, is not called while clients run).  $STRELKA_AMD_BROKER_POOL_GB (64).
struct BlockPool
{
    std::mutex mu;
    std::multimap<size_t, void*> blocks;
    size_t bytes = 0, cap = size_t(env_us("STRELKA_AMD_BROKER_POOL_GB", 64)) << 30;
    uint64_t hits = 0, misses = 0;
    /// a block of at least `need` bytes and at most half as much again; *size: what it really holds
    void* take(const size_t need, size_t* size)
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = blocks.lower_bound(need);
        if (it == blocks.end() || it->first > need + need / 2 + 65536) {
            ++misses;
            return nullptr;
        }
        void* p = it->second;
        *size = it->first;
        bytes -= it->first;
        blocks.erase(it);
        ++hits;
        return p;
    }
    bool give(void* p, const size_t size)
    {
        std::lock_guard<std::mutex> g(mu);
        if (bytes + size > cap) return false;
        blocks.emplace(size, p);
        bytes += size;
        return true;
    }
};
BlockPool g_pool;

// ... and their streams: creating one costs the runtime ~10 ms alone and ~100-200 ms when sixteen clients connect at once
// (profiles/r06_v14_init_laps.txt: a warm server's clients waited 120 ms each for "their first allocation" -- for the server thread's
// hipStreamCreate, in fact).  A departed client's stream has been waited for and is as good as new; a few are made ahead at start-up.
struct StreamPool
{
    std::mutex mu;
    std::vector<hipStream_t> idle;
    hipStream_t take()
    {
        {
            std::lock_guard<std::mutex> g(mu);
            if (!idle.empty()) {
                hipStream_t st = idle.back();
                idle.pop_back();
                return st;
            }
        }
        hipStream_t st = nullptr;
        (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        return st;
    }
    void give(hipStream_t st)
    {
        std::lock_guard<std::mutex> g(mu);
        idle.push_back(st);
    }
};
StreamPool g_streams;

struct Block
{
    void* p;
    size_t size, pool_size; // what the client asked for; what the block really holds (larger when it came from the pool)
};

struct Conn
{
    int sock = -1;
    Ctl* ctl = nullptr;
    int slot = -1;
    char* slot_va = nullptr;
    hipStream_t stream = nullptr;
    std::vector<Block> allocs, freed; // in use; freed by the client (reused by its own later allocations: same stream, in order)
    double alloc_s = 0;
    std::vector<Seg> segs;
    uint64_t n_launch = 0, n_sync = 0, n_sleeps = 0;
    unsigned pid = 0;
    bool bye = false; // the client has said it is leaving
    size_t slot_used = 0; // the end of the highest segment mapped in the slot (recycling)
    // ($STRELKA_AMD_BROKER_VERBOSE) from the first record after a wait to the SYNC record; inside hipStreamSynchronize
    bool timing = std::getenv("STRELKA_AMD_BROKER_VERBOSE") != nullptr;
    bool batch_open = false;
    double batch_begin = 0, submit_s = 0, sync_s = 0;
    double batch_first = 0; // when the server thread saw the first record since the last SYNC
    double in_launch_s = 0, in_copy_s = 0, in_fill_s = 0, idle_in_batch_s = 0; // inside the runtime's calls; with a batch open and the ring empty
    uint64_t n_copy = 0, n_fill = 0;
};

void srv_error(Conn& c, const int32_t code, const char* what, const char* detail)
{
    // (one writer -- this client's server thread; the client takes the status with acquire: the text is complete before the code shows)
    if (c.ctl->status.load(std::memory_order_acquire) != 0) return; // the first error since the client last asked stays
    std::snprintf(c.ctl->error_text, sizeof(c.ctl->error_text), "%s: %s (broker)", what, detail ? detail : "");
    c.ctl->status.store(code ? code : int32_t(hipErrorUnknown), std::memory_order_release);
}
#define SRV_HIP(c, expr)                                                                  \
    do {                                                                                  \
        const hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) srv_error(c, int32_t(e_), #expr, hipGetErrorString(e_));    \
    } while (0)

void srv_done(Conn& c, const uint32_t seq)
{
    c.ctl->done_seq.store(seq, std::memory_order_seq_cst);
    if (c.ctl->client_waiting.load(std::memory_order_seq_cst)) futex_wake(&c.ctl->done_seq);
}

bool in_segs(const Conn& c, const uint64_t p, const uint64_t bytes)
{
    for (const Seg& s : c.segs) {
        const uint64_t b = uint64_t(reinterpret_cast<uintptr_t>(s.va));
        if (p >= b && p + bytes <= b + s.bytes) return true;
    }
    return false;
}

// A client's copies between its page-locked segments and device memory as a KERNEL on the client's stream: the copy is then one more
// packet of that stream's queue, in order, instead of a transfer on a DMA queue that all clients of this process share and that the
// runtime ties to the compute queue with signals on both sides (the caller processes' window copies queued behind each other's:
// profiles/r06_v2_sharing.txt, pileup / feed ABI seconds).  $STRELKA_AMD_BROKER_COPY=dma: hipMemcpyAsync instead.
__global__ __launch_bounds__(256) void broker_copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const uint64_t n16)
{
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += uint64_t(gridDim.x) * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void broker_copy1_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const uint64_t n)
{
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) dst[i] = src[i];
}
__global__ void broker_flag_kernel(uint32_t* flag, const uint32_t seq)
{
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
bool copy_by_kernel()
{
    static const bool k = [] {
        const char* e = std::getenv("STRELKA_AMD_BROKER_COPY");
        return !(e && std::strcmp(e, "dma") == 0);
    }();
    return k;
}
void srv_copy(Conn& c, const uint64_t dst, const uint64_t src, const uint64_t bytes, const hipMemcpyKind kind)
{
    if (!copy_by_kernel() || kind == hipMemcpyDeviceToDevice) {
        SRV_HIP(c, hipMemcpyAsync(reinterpret_cast<void*>(uintptr_t(dst)), reinterpret_cast<const void*>(uintptr_t(src)), bytes, kind, c.stream));
        return;
    }
    uint64_t d = dst, s = src, n = bytes;
    if (((d ^ s) & 15) == 0) { // the same phase: bytes up to a 16-byte boundary, 16-byte pieces, the tail
        const uint64_t head = std::min<uint64_t>(n, (16 - (d & 15)) & 15);
        if (head) hipLaunchKernelGGL(broker_copy1_kernel, dim3(1), dim3(256), 0, c.stream, reinterpret_cast<const uint8_t*>(uintptr_t(s)), reinterpret_cast<uint8_t*>(uintptr_t(d)), head);
        d += head, s += head, n -= head;
        const uint64_t n16 = n / 16;
        if (n16) hipLaunchKernelGGL(broker_copy16_kernel, dim3(unsigned(std::min<uint64_t>((n16 + 255) / 256, 1024))), dim3(256), 0, c.stream,
                                    reinterpret_cast<const uint4*>(uintptr_t(s)), reinterpret_cast<uint4*>(uintptr_t(d)), n16);
        d += n16 * 16, s += n16 * 16, n -= n16 * 16;
    }
    if (n) hipLaunchKernelGGL(broker_copy1_kernel, dim3(unsigned(std::min<uint64_t>((n + 255) / 256, 1024))), dim3(256), 0, c.stream,
                              reinterpret_cast<const uint8_t*>(uintptr_t(s)), reinterpret_cast<uint8_t*>(uintptr_t(d)), n);
    SRV_HIP(c, hipGetLastError());
}

void srv_execute(Conn& c, const Rec* rec)
{
    const bool host = g_srv.host_backend;
    switch (rec->op) {
    case OP_PAD: break;
    case OP_LAUNCH: {
        const RecLaunch* r = reinterpret_cast<const RecLaunch*>(rec);
        const char* p = reinterpret_cast<const char*>(r) + up16(sizeof(RecLaunch));
        const uint32_t* sizes = reinterpret_cast<const uint32_t*>(p);
        p += up16(4 * size_t(r->n_args));
        void* argv[64];
        if (r->n_args > 64) {
            srv_error(c, int32_t(hipErrorInvalidValue), "launch", "more than 64 kernel parameters");
            break;
        }
        for (uint32_t i = 0; i < r->n_args; ++i) {
            argv[i] = const_cast<char*>(p);
            p += up16(sizes[i]);
        }
        if (r->fn_off >= lib_info().size) { // (a kernel is a host stub of THIS library: an offset inside the file both sides loaded)
            srv_error(c, int32_t(hipErrorInvalidValue), "launch", "kernel offset outside libstrelka_amd.so");
            break;
        }
        const void* fn = reinterpret_cast<const void*>(lib_info().base + uintptr_t(r->fn_off));
        ++c.n_launch;
        const double t0 = c.timing ? tm_now() : 0.0;
        if (host) reinterpret_cast<void (*)(void**)>(const_cast<void*>(fn))(argv);
        else SRV_HIP(c, hipLaunchKernel(fn, dim3(r->grid[0], r->grid[1], r->grid[2]), dim3(r->block[0], r->block[1], r->block[2]), argv, r->lds, c.stream));
        if (c.timing) c.in_launch_s += tm_now() - t0;
        break;
    }
    case OP_MEMCPY: {
        const RecCopy* r = reinterpret_cast<const RecCopy*>(rec);
        // host sides of a copy lie in this client's page-locked segments (the client staged everything else)
        if ((r->kind == hipMemcpyHostToDevice && !in_segs(c, r->src, r->bytes)) || (r->kind == hipMemcpyDeviceToHost && !in_segs(c, r->dst, r->bytes))) {
            srv_error(c, int32_t(hipErrorInvalidValue), "copy", "host side outside the client's page-locked segments");
            break;
        }
        if (host) std::memcpy(reinterpret_cast<void*>(uintptr_t(r->dst)), reinterpret_cast<const void*>(uintptr_t(r->src)), r->bytes);
        else {
            const double t0 = c.timing ? tm_now() : 0.0;
            srv_copy(c, r->dst, r->src, r->bytes, hipMemcpyKind(r->kind));
            if (c.timing) c.in_copy_s += tm_now() - t0, ++c.n_copy;
        }
        break;
    }
    case OP_MEMSET: {
        const RecCopy* r = reinterpret_cast<const RecCopy*>(rec);
        if (host) std::memset(reinterpret_cast<void*>(uintptr_t(r->dst)), int(r->value), r->bytes);
        else {
            const double t0 = c.timing ? tm_now() : 0.0;
            SRV_HIP(c, hipMemsetAsync(reinterpret_cast<void*>(uintptr_t(r->dst)), int(r->value), r->bytes, c.stream));
            if (c.timing) c.in_fill_s += tm_now() - t0, ++c.n_fill;
        }
        break;
    }
    case OP_SYNC: {
        ++c.n_sync;
        if (!host) {
            const uint64_t f = c.ctl->gpu_flag;
            if (f && in_segs(c, f, 4)) hipLaunchKernelGGL(broker_flag_kernel, dim3(1), dim3(1), 0, c.stream, reinterpret_cast<uint32_t*>(uintptr_t(f)), rec->seq);
            const double t0 = tm_now();
            if (c.timing) c.submit_s += t0 - c.batch_begin;
            SRV_HIP(c, hipStreamSynchronize(c.stream));
            const double t1 = tm_now();
            if (c.timing) c.sync_s += t1 - t0;
            c.ctl->t_first = c.batch_first;
            c.ctl->t_reached = t0;
            c.ctl->t_done = t1;
        }
        srv_done(c, rec->seq);
        c.batch_open = false;
        c.batch_first = 0;
        break;
    }
    case OP_MALLOC: {
        const RecMem* r = reinterpret_cast<const RecMem*>(rec);
        const double t0 = c.timing ? tm_now() : 0.0;
        const size_t need = r->a ? r->a : 1;
        Block b{ nullptr, need, need };
        for (size_t i = 0; i < c.freed.size() && !b.p; ++i)
            if (c.freed[i].pool_size >= need && c.freed[i].pool_size <= need + need / 2 + 65536) {
                b = c.freed[i];
                b.size = need;
                c.freed.erase(c.freed.begin() + long(i));
            }
        if (!b.p && !host) b.p = g_pool.take(need, &b.pool_size); // a block a departed client left
        if (!b.p) {
            if (host) b.p = std::malloc(need);
            else SRV_HIP(c, hipMalloc(&b.p, need));
        }
        if (b.p) c.allocs.push_back(b);
        c.ctl->result = uint64_t(reinterpret_cast<uintptr_t>(b.p));
        if (c.timing) c.alloc_s += tm_now() - t0;
        srv_done(c, rec->seq);
        break;
    }
    case OP_FREE: {
        // hipFree waits for the whole device -- every other client's stream too.  A caller's buffers only grow (a free is followed by a
        // larger allocation): what a client frees stays its own (a later allocation of its may take it: same stream, in order) until
        // it leaves, when everything goes to the pool.
        const RecMem* r = reinterpret_cast<const RecMem*>(rec);
        for (size_t i = 0; i < c.allocs.size(); ++i)
            if (uint64_t(reinterpret_cast<uintptr_t>(c.allocs[i].p)) == r->a) {
                c.freed.push_back(c.allocs[i]);
                c.allocs.erase(c.allocs.begin() + long(i));
                break;
            }
        break;
    }
    case OP_FUNC_ATTR: {
        const RecMem* r = reinterpret_cast<const RecMem*>(rec);
        if (!host) SRV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(lib_info().base + uintptr_t(r->a)), hipFuncAttribute(r->c), int(r->d)));
        srv_done(c, rec->seq);
        break;
    }
    case OP_HOST_MAP: {
        const RecMem* r = reinterpret_cast<const RecMem*>(rec);
        Msg m;
        int fd = -1;
        const uint64_t lo = uint64_t(reinterpret_cast<uintptr_t>(c.slot_va));
        if (recv_msg(c.sock, m, &fd, 10000) || m.type != MSG_MAP || fd < 0 || m.a != r->a || m.b != r->b || r->a < lo || r->a + r->b > lo + SLOT_BYTES) {
            srv_error(c, int32_t(hipErrorInvalidValue), "page-locked segment", "the descriptor did not arrive with the request");
        } else {
            void* want = reinterpret_cast<void*>(uintptr_t(r->a));
            void* p = mmap(want, r->b, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0);
            if (p != want) srv_error(c, int32_t(hipErrorOutOfMemory), "page-locked segment", "mmap failed");
            else {
                hipError_t e = hipSuccess;
                if (!host) e = hipHostRegister(p, r->b, hipHostRegisterDefault);
                if (e != hipSuccess) {
                    srv_error(c, int32_t(e), "hipHostRegister", hipGetErrorString(e));
                    (void)mmap(want, r->b, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED, -1, 0);
                } else {
                    c.segs.push_back(Seg{ static_cast<char*>(p), size_t(r->b), fd });
                    fd = -1; // (kept: the slot's next client maps the same pages)
                    c.slot_used = std::max(c.slot_used, size_t(r->a + r->b - lo));
                }
            }
        }
        if (fd >= 0) close(fd);
        srv_done(c, rec->seq);
        break;
    }
    case OP_HOST_UNMAP: {
        const RecMem* r = reinterpret_cast<const RecMem*>(rec);
        for (size_t i = 0; i < c.segs.size(); ++i)
            if (uint64_t(reinterpret_cast<uintptr_t>(c.segs[i].va)) == r->a) {
                if (!host) {
                    SRV_HIP(c, hipStreamSynchronize(c.stream));
                    SRV_HIP(c, hipHostUnregister(c.segs[i].va));
                }
                (void)mmap(c.segs[i].va, c.segs[i].bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED, -1, 0);
                if (c.segs[i].fd >= 0) close(c.segs[i].fd);
                c.segs.erase(c.segs.begin() + long(i));
                break;
            }
        srv_done(c, rec->seq);
        break;
    }
    case OP_BYE: c.bye = true; break;
    default: srv_error(c, int32_t(hipErrorInvalidValue), "record", "unknown op"); break;
    }
}

void serve_client(const int sock)
{
    Conn c;
    c.sock = sock;
    Msg hello;
    int fd = -1;
    auto refuse = [&](const char* why) {
        Msg m{ MAGIC, MSG_REFUSED, 0, 0, 0, {} };
        std::snprintf(m.text, sizeof(m.text), "%s", why);
        (void)send_msg(sock, m, -1);
        if (fd >= 0) close(fd);
        close(sock);
    };
    {
        // (the rendezvous name is per user; a client of another user has no business here whatever name it knows)
        ucred cred{};
        socklen_t cl = sizeof(cred);
        if (getsockopt(sock, SOL_SOCKET, SO_PEERCRED, &cred, &cl) != 0 || cred.uid != getuid()) return refuse("another user's process");
    }
    if (recv_msg(sock, hello, &fd, 10000) || hello.type != MSG_HELLO || fd < 0) return refuse("bad hello");
    if (uint32_t(hello.c >> 32) != PROTO || hello.a != lib_info().size || hello.b != lib_info().mtime)
        return refuse("the client has loaded another build of libstrelka_amd.so than this broker");
    c.pid = unsigned(hello.c & 0xffffffffu);
    void* p = mmap(nullptr, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    fd = -1;
    if (p == MAP_FAILED) return refuse("cannot map the ring");
    c.ctl = static_cast<Ctl*>(p);
    if (c.ctl->magic != MAGIC || c.ctl->proto != PROTO) {
        munmap(p, sizeof(Ctl));
        return refuse("bad ring");
    }
    // an address range for the client's page-locked segments that is free in both processes; first choice: the slot of a client that
    // has left, with its segments still mapped and page-locked here
    std::vector<int> tried;
    Server::Recycled inherited{ -1, {}, 0 };
    for (;;) {
        int slot = -1;
        bool is_recycled = false;
        {
            std::lock_guard<std::mutex> g(g_srv.mu);
            if (inherited.slot < 0 && !g_srv.recycled.empty()) {
                inherited = g_srv.recycled.back();
                g_srv.recycled.pop_back();
                slot = inherited.slot; // (slot_taken stays true: it was never released)
                is_recycled = true;
            }
            for (int i = 0; i < MAX_SLOTS && slot < 0; ++i) {
                bool was_tried = false;
                for (int t : tried) was_tried |= (t == i);
                if (!g_srv.slot_taken[size_t(i)] && !was_tried) slot = i;
            }
            if (slot >= 0) g_srv.slot_taken[size_t(slot)] = true;
        }
        if (slot < 0) {
            munmap(p, sizeof(Ctl));
            return refuse("no free address range");
        }
        tried.push_back(slot);
        void* want = reinterpret_cast<void*>(uintptr_t(SLOT_BASE + uint64_t(slot) * SLOT_BYTES));
        bool ok = true;
        if (!is_recycled) { // (a recycled slot's range is still reserved here, its segments mapped inside)
            void* got = mmap(want, SLOT_BYTES, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED_NOREPLACE, -1, 0);
            ok = (got == want);
            if (!ok && got != MAP_FAILED) munmap(got, SLOT_BYTES);
        }
        if (ok) {
            Msg m{ MAGIC, MSG_SLOT, uint64_t(reinterpret_cast<uintptr_t>(want)), uint64_t(g_srv.device_count), uint64_t(g_srv.host_backend ? 1 : 0), {} };
            Msg r;
            if (send_msg(sock, m, -1) || recv_msg(sock, r, nullptr, 10000) || (r.type != MSG_SLOT_OK && r.type != MSG_SLOT_RETRY)) {
                if (is_recycled)
                    for (const Seg& sgm : inherited.segs) {
                        if (!g_srv.host_backend) (void)hipHostUnregister(sgm.va);
                        if (sgm.fd >= 0) close(sgm.fd);
                    }
                munmap(want, SLOT_BYTES);
                std::lock_guard<std::mutex> g(g_srv.mu);
                g_srv.slot_taken[size_t(slot)] = false;
                munmap(p, sizeof(Ctl));
                close(sock);
                return;
            }
            if (r.type == MSG_SLOT_OK) {
                c.slot = slot;
                c.slot_va = static_cast<char*>(want);
                // the segments that come with the slot
                const bool give = is_recycled;
                Msg sg{ MAGIC, MSG_SEGS, uint64_t(give ? inherited.segs.size() : 0), uint64_t(give ? inherited.used : 0), 0, {} };
                bool sent = (send_msg(sock, sg, -1) == 0);
                for (size_t i = 0; sent && give && i < inherited.segs.size(); ++i) {
                    Msg one{ MAGIC, MSG_SEG, uint64_t(reinterpret_cast<uintptr_t>(inherited.segs[i].va)), uint64_t(inherited.segs[i].bytes), 0, {} };
                    sent = (send_msg(sock, one, inherited.segs[i].fd) == 0);
                }
                if (give) {
                    c.segs = inherited.segs;
                    c.slot_used = inherited.used;
                    std::lock_guard<std::mutex> g(g_srv.mu);
                    ++g_srv.recycled_given;
                }
                (void)sent; // (a client that went away mid-handshake is noticed by the loop below)
                break;
            }
            // the client cannot have this range: a recycled slot is let go of for good
            if (is_recycled) {
                for (const Seg& sgm : inherited.segs) {
                    if (!g_srv.host_backend) (void)hipHostUnregister(sgm.va);
                    if (sgm.fd >= 0) close(sgm.fd);
                }
                inherited.segs.clear();
            }
            munmap(want, SLOT_BYTES);
        }
        std::lock_guard<std::mutex> g(g_srv.mu);
        g_srv.slot_taken[size_t(slot)] = false; // (free for other clients; not tried again for this one)
    }
    if (!g_srv.host_backend) {
        (void)hipSetDevice(g_srv.device);
        c.stream = g_streams.take();
        if (!c.stream) srv_error(c, int32_t(hipErrorUnknown), "hipStreamCreateWithFlags", "no stream for this client");
    }
    g_srv.clients.fetch_add(1);
    g_srv.served.fetch_add(1);

    Ctl* ctl = c.ctl;
    uint64_t tail = 0;
    for (;;) {
        uint64_t head = ctl->head.load(std::memory_order_acquire);
        if (head == tail) {
            const double idle_t0 = (c.timing && c.batch_open) ? tm_now() : 0.0;
            struct IdleLap
            {
                Conn& c;
                double t0;
                ~IdleLap()
                {
                    if (t0 != 0.0) c.idle_in_batch_s += tm_now() - t0;
                }
            } idle_lap{ c, idle_t0 };
            bool got = false;
            static const unsigned spin_us = env_us("STRELKA_AMD_BROKER_SERVER_SPIN_US", 2);
            const double until = tm_now() + 1e-6 * spin_us;
            do {
                for (int i = 0; i < 64 && !got; ++i) {
                    __builtin_ia32_pause();
                    got = (ctl->head.load(std::memory_order_acquire) != tail);
                }
            } while (!got && tm_now() < until);
            if (got) continue;
            ++c.n_sleeps;
            ctl->server_idle.store(1, std::memory_order_seq_cst);
            if (ctl->head.load(std::memory_order_seq_cst) != tail) {
                ctl->server_idle.store(0, std::memory_order_seq_cst);
                continue;
            }
            futex_wait(&ctl->server_idle, 1, 50);
            ctl->server_idle.store(0, std::memory_order_seq_cst);
            if (ctl->head.load(std::memory_order_acquire) == tail && peer_gone(sock)) break;
            continue;
        }
        while (tail != head) {
            const Rec* rec = reinterpret_cast<const Rec*>(ctl->ring + (tail % RING_BYTES));
            const uint32_t bytes = rec->bytes;
            if (bytes < 16 || (bytes & 15) || bytes > RING_BYTES || tail + bytes > head) { // (a client that writes nonsense is dropped)
                std::fprintf(stderr, "[sk_broker] client pid %u: malformed record, dropping the client\n", c.pid);
                tail = head;
                shutdown(sock, SHUT_RDWR);
                break;
            }
            if (c.batch_first == 0) c.batch_first = tm_now();
            if (c.timing && !c.batch_open) {
                c.batch_open = true;
                c.batch_begin = c.batch_first;
            }
            srv_execute(c, rec);
            tail += bytes;
            ctl->tail.store(tail, std::memory_order_release);
            if (c.bye) break;
        }
        if (c.bye) break;
    }
    // the client is gone: wait for its work, give back what it held
    if (!g_srv.host_backend) {
        (void)hipStreamSynchronize(c.stream);
        for (const std::vector<Block>* v : { &c.allocs, &c.freed })
            for (const Block& b : *v)
                if (!g_pool.give(b.p, b.pool_size)) (void)hipFree(b.p);
        if (c.stream) g_streams.give(c.stream);
    } else {
        for (const std::vector<Block>* v : { &c.allocs, &c.freed })
            for (const Block& b : *v) std::free(b.p);
    }
    // the slot: kept whole -- range reserved, segments mapped and page-locked -- for the next client, unless enough are kept already
    bool kept = false;
    if (!c.segs.empty()) {
        size_t bytes = 0;
        for (const Seg& sg : c.segs) bytes += sg.bytes;
        std::lock_guard<std::mutex> g(g_srv.mu);
        if (g_srv.recycled.size() < 64 && bytes <= (size_t(1) << 30)) {
            g_srv.recycled.push_back(Server::Recycled{ c.slot, c.segs, c.slot_used });
            kept = true;
        }
    }
    if (!kept) {
        for (const Seg& sg : c.segs) {
            if (!g_srv.host_backend) (void)hipHostUnregister(sg.va);
            if (sg.fd >= 0) close(sg.fd);
        }
        munmap(c.slot_va, SLOT_BYTES); // (the segments inside go with it)
    }
    munmap(ctl, sizeof(Ctl));
    close(sock);
    {
        std::lock_guard<std::mutex> g(g_srv.mu);
        if (!kept) g_srv.slot_taken[size_t(c.slot)] = false;
        g_srv.last_client = std::chrono::steady_clock::now();
    }
    if (std::getenv("STRELKA_AMD_BROKER_VERBOSE"))
        std::fprintf(stderr, "[sk_broker] client pid %u left: %llu launches, %llu waits, %llu sleeps, submit %.4f s, in hipStreamSynchronize %.4f s, in hipLaunchKernel %.4f s, "
                             "%llu copies %.4f s, %llu fills %.4f s, ring empty inside a batch %.4f s, allocations %.4f s\n", c.pid,
                     (unsigned long long)c.n_launch, (unsigned long long)c.n_sync, (unsigned long long)c.n_sleeps, c.submit_s, c.sync_s, c.in_launch_s,
                     (unsigned long long)c.n_copy, c.in_copy_s, (unsigned long long)c.n_fill, c.in_fill_s, c.idle_in_batch_s, c.alloc_s);
    g_srv.clients.fetch_sub(1);
}
} // namespace
} // namespace skrt

// ---------------------------------------------------------------------------------------------------------------------
// the server's main loop (called by `sk_broker`, broker/sk_broker_main.cpp)

extern "C" int sk_broker_serve(const int device, const char* socket_name_arg, const int idle_seconds)
{
    using namespace skrt;
    signal(SIGPIPE, SIG_IGN);
    const char* be = std::getenv("STRELKA_AMD_BROKER_BACKEND");
    g_srv.host_backend = be && std::strcmp(be, "host") == 0;
    g_srv.device = device;
    const std::string name = socket_name_arg && socket_name_arg[0] ? socket_name_arg : socket_name(device);
    const int ls = socket(AF_UNIX, SOCK_SEQPACKET | SOCK_CLOEXEC, 0);
    if (ls < 0) return 1;
    sockaddr_un a;
    const socklen_t len = abstract_addr(a, name);
    if (bind(ls, reinterpret_cast<sockaddr*>(&a), len) != 0) {
        std::fprintf(stderr, "[sk_broker] @%s is taken: another broker serves this device\n", name.c_str());
        return 3;
    }
    if (!g_srv.host_backend) {
        // Few hardware queues, shared by the clients' streams: the device runs the queues of one process side by side up to a handful;
        // beyond that its scheduler multiplexes them (profiles/r06_v1_broker_probe.txt: 16 caller threads, 4 queues: 43 600 jobs/s; 16
        // queues: 11 300; 16 caller PROCESSES: 6 000)
        (void)setenv("GPU_MAX_HW_QUEUES", std::getenv("STRELKA_AMD_BROKER_QUEUES") ? std::getenv("STRELKA_AMD_BROKER_QUEUES") : "4", 1);
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || device < 0 || device >= n) {
            std::fprintf(stderr, "[sk_broker] device %d of %d: %s\n", device, n, hipGetErrorString(e));
            return 2;
        }
        g_srv.device_count = n;
        if (hipSetDevice(device) != hipSuccess) return 2;
        if (std::getenv("STRELKA_AMD_SPIN_WAIT") == nullptr) (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
            std::fprintf(stderr, "[sk_broker] built for gfx950 only\n");
            return 2;
        }
        void* warm = nullptr; // the context, before the first client waits for it
        if (hipMalloc(&warm, 256) == hipSuccess) (void)hipFree(warm);
        std::thread([] { // streams for the first wave of clients, made while the first of them is still shaking hands
            (void)hipSetDevice(g_srv.device);
            for (int i = 0; i < int(env_us("STRELKA_AMD_BROKER_STREAMS_AHEAD", 16)); ++i) {
                hipStream_t st = nullptr;
                if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) g_streams.give(st);
            }
        }).detach();
    } else if (const char* dc = std::getenv("STRELKA_AMD_BROKER_HOST_DEVICES")) g_srv.device_count = std::max(1, std::atoi(dc));
    if (listen(ls, 256) != 0) return 1;
    std::fprintf(stderr, "[sk_broker] pid %d serving device %d on @%s (%s backend)\n", int(getpid()), device, name.c_str(), g_srv.host_backend ? "host" : "hip");
    g_srv.last_client = std::chrono::steady_clock::now();
    for (;;) {
        pollfd p{ ls, POLLIN, 0 };
        const int r = poll(&p, 1, 500);
        if (r > 0 && (p.revents & POLLIN)) {
            const int s = accept4(ls, nullptr, nullptr, SOCK_CLOEXEC);
            if (s >= 0) std::thread(serve_client, s).detach();
            std::lock_guard<std::mutex> g(g_srv.mu);
            g_srv.last_client = std::chrono::steady_clock::now();
            continue;
        }
        if (g_srv.clients.load() == 0) {
            std::lock_guard<std::mutex> g(g_srv.mu);
            if (std::chrono::steady_clock::now() - g_srv.last_client > std::chrono::seconds(idle_seconds)) break;
        }
    }
    std::fprintf(stderr, "[sk_broker] pid %d: no client for %d s after %lld served, leaving (device blocks from the pool / from the driver: %llu / %llu; clients that inherited a slot's page-locked segments: %llu)\n", int(getpid()),
                 idle_seconds, (long long)g_srv.served.load(), (unsigned long long)g_pool.hits, (unsigned long long)g_pool.misses, (unsigned long long)g_srv.recycled_given);
    close(ls);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// self-test of the client / server path (tests/test_broker.py): device memory, a copy from and to pageable memory (staged), a copy to
// page-locked memory (direct), a launch -- a kernel on a GPU, a host function under the broker's no-GPU backend -- and the wait

namespace
{
__global__ void selftest_kernel(const uint32_t* in, uint32_t* out, const int n, const uint32_t mul)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * mul + 1u;
}
void selftest_host_fn(void** args) // (the same parameter block, as the host backend's "launch" hands it over)
{
    const uint32_t* in = *static_cast<const uint32_t**>(args[0]);
    uint32_t* out = *static_cast<uint32_t**>(args[1]);
    const int n = *static_cast<int*>(args[2]);
    const uint32_t mul = *static_cast<uint32_t*>(args[3]);
    for (int i = 0; i < n; ++i) out[i] = in[i] * mul + 1u;
}
}

extern "C" int sk_broker_selftest(const int n, const uint32_t mul)
{
    using namespace skrt;
    if (n <= 0) return 1;
    hipStream_t st = remote() ? r_stream() : nullptr;
    uint32_t *d_in = nullptr, *d_out = nullptr;
    void* pinned = nullptr;
    std::vector<uint32_t> src(static_cast<size_t>(n)), back(static_cast<size_t>(n), 0u);
    for (int i = 0; i < n; ++i) src[size_t(i)] = uint32_t(i) * 2654435761u;
    const size_t bytes = 4 * size_t(n);
    int rc = 1;
    do {
        if (malloc_(&d_in, bytes) != hipSuccess || malloc_(&d_out, bytes) != hipSuccess || hostMalloc(&pinned, bytes) != hipSuccess) break;
        if (memsetAsync(d_out, 0, bytes, st) != hipSuccess) break;
        if (memcpyAsync(d_in, src.data(), bytes, hipMemcpyHostToDevice, st) != hipSuccess) break;
        if (remote() && r_host_backend()) {
            const uint32_t* a0 = d_in;
            uint32_t* a1 = d_out;
            int a2 = n;
            uint32_t a3 = mul;
            void* argv[4] = { &a0, &a1, &a2, &a3 };
            const uint32_t sizes[4] = { 8, 8, 4, 4 };
            r_launch(reinterpret_cast<const void*>(&selftest_host_fn), dim3(1), dim3(1), 0, st, argv, sizes, 4);
        } else {
            SK_LAUNCH(selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_in, d_out, n, mul);
        }
        if (getLastError() != hipSuccess) break;
        if (memcpyAsync(back.data(), d_out, bytes, hipMemcpyDeviceToHost, st) != hipSuccess) break;
        if (memcpyAsync(pinned, d_out, bytes, hipMemcpyDeviceToHost, st) != hipSuccess) break;
        if (streamSynchronize(st) != hipSuccess) break;
        rc = 0;
        const uint32_t* pn = static_cast<const uint32_t*>(pinned);
        for (int i = 0; i < n; ++i)
            if (back[size_t(i)] != src[size_t(i)] * mul + 1u || pn[i] != back[size_t(i)]) rc = 2;
    } while (false);
    if (pinned) (void)hostFree(pinned);
    if (d_in) (void)free_(d_in);
    if (d_out) (void)free_(d_out);
    return rc;
}
