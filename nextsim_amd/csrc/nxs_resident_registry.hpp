// nxs_resident_registry.hpp -- who may run a grid of WAITING workgroups on a device (host-only, plain C++17 + POSIX; no HIP).
//
// k_substep_resident needs every workgroup of its grid on a CU at once (its workgroups spin for each other), k_substep_pair<HALO> its band patches.  Two such
// grids whose sum does not fit keep each other's missing workgroups from ever starting (both time out).  And a co-tenant's ORDINARY kernels can do the same where
// their blocks wait too: k_smooth_halo's blocks spin for a neighbour rank's flags, so do k_halo_pull's and the boundary patches of k_substep_fused<HALO> -- if the
// rank they wait for is this grid's rank and its last workgroups cannot start because those spinning blocks hold the CUs, nobody moves until the 10 s bound
// (the lost steps of round 3's shared-device rehearsals: two ranks of one process, 2 x 245 resident workgroups = 96 % of the device).
// So a handle CLAIMS its workgroup slots before it builds such a grid, every several-rank handle REGISTERS the largest grid of waiting blocks its ordinary kernels
// launch, and a claim is refused up front -- the step then runs one kernel per sub-step, deterministically -- unless
//
//     this claim + every other claim on the device + the waiting ordinary grids of every OTHER handle on the device   <=   1.0
//
// all as fractions of the device (a grid's workgroups / the workgroups of that kernel build the device holds at once: occupancy query x CUs).  That is the worst
// case made explicit: every block that may spin is on a CU and spinning, and the claimed grids still fit.  Round 4 had a measured constant here (0.70: "0.68 ran
// clean 32 of 32, 0.96 lost 1 in 12"); the rule reproduces both observations from the grids themselves -- the rehearsal's two ranks of 44 k nodes launch 172 blocks
// of k_smooth_halo, 8.4 % of the 2 048 the device holds: 0.48 + 0.48 + 0.084 > 1 is refused, 0.34 + 0.34 + 0.084 admitted -- and needs no headroom where the
// co-tenants' kernels cannot wait (a single-rank handle registers none).  The rule is evaluated when a claim is made: hosts register (nxs_dyn_set_halo) on every
// rank before any rank builds its grid (the first step / option "prepare"), which the start barrier of a several-rank run gives anyway.
// NXS_RESIDENT_SHARED_LIMIT (percent) still caps the sum where claimants share a device, for experiments.
//
// Handles of OTHER PROCESSES on the device count: the table lives in POSIX shared memory named after the user and the device (its PCI bus id), guarded by flock() on
// the segment (released by the kernel when a process dies).  Assumptions, stated because the table's content is trusted:
//   * the segment is PER USER (mode 0600, the uid in its name): ranks sharing a GPU are one user's MPI job; another user's processes on the same device are not seen
//     (and cannot plant entries).  Entries are still validated on read (counts in range), a bad one is ignored and cleared;
//   * liveness by pid + start time is judged only for entries of THIS PID namespace (/proc/self/ns/pid): a process of another container that shares /dev/shm is
//     invisible in /proc here and would be swept although alive.  Foreign-namespace entries expire instead: every entry carries the time its owner last touched it
//     (CLOCK_BOOTTIME; claim, registration, and every ~10 s of stepping), one that is 300 s old is dropped;
//   * a version mismatch (another build of the library on the device) does not wipe live entries: this process then keeps a table of its own, as without shared memory;
//   * flock() is per open file description: a fork()ed child shares its parent's -- so a child opens one of its own the first time it takes the lock (own_description).
// tests/native/registry_host.cpp drives this file from two processes (tests/test_resident_registry.py): a second process' grid is refused up front, a dead
// process' claim is reclaimed, the arithmetic of claims and waiting grids.
#ifndef NXS_RESIDENT_REGISTRY_HPP
#define NXS_RESIDENT_REGISTRY_HPP

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <map>
#include <mutex>
#include <signal.h>
#include <string>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

namespace nxs_reg {

constexpr uint32_t MAGIC = 0x4e585352u;  // 'NXSR'
constexpr uint32_t VERSION = 2;
constexpr int MAX_ENTRIES = 256;
constexpr int MAX_COUNT = 1 << 22;       // no grid and no device capacity is larger: an entry beyond it was not written by this code
constexpr uint64_t STALE_SECONDS = 300;  // a foreign-namespace entry nobody touched for this long is dropped
enum { KIND_NONE = 0, KIND_RESIDENT = 1 /* k_substep_resident* */, KIND_PAIR = 2 /* the band patches of k_substep_pair<HALO> */ };

struct Entry {
    int32_t in_use, pid;
    uint64_t start;     // the process' start time (/proc/<pid>/stat field 22): a recycled pid is not the same process
    uint64_t handle;    // the handle's address in its process (pid + handle name an entry)
    uint64_t seq;       // registration order on the device
    int32_t wg, slots;  // the claim: workgroups that must be resident together / workgroups of that kernel build the device holds at once; wg == 0: no claim
    int32_t multi_rank, kind;   // KIND_*: which grid of the handle holds the claim
    int32_t ord_blocks, ord_slots;   // the largest grid of blocks that may WAIT inside this handle's ordinary kernels / blocks of that kernel the device holds; 0: none
    uint64_t pidns;     // inode of the owner's /proc/self/ns/pid
    uint64_t stamp;     // CLOCK_BOOTTIME seconds of the owner's last touch
};
struct Table {
    uint32_t magic, version;
    uint64_t next_seq;
    Entry e[MAX_ENTRIES];
};

inline uint64_t boot_seconds() {
    struct timespec ts;
    return clock_gettime(CLOCK_BOOTTIME, &ts) == 0 ? (uint64_t)ts.tv_sec : 0;
}
inline uint64_t my_pid_namespace() {
    static const uint64_t ns = [] { struct stat st; return stat("/proc/self/ns/pid", &st) == 0 ? (uint64_t)st.st_ino : (uint64_t)0; }();
    return ns;
}
inline bool entry_valid(const Entry &e) {
    return e.wg >= 0 && e.slots >= 0 && e.wg <= MAX_COUNT && e.slots <= MAX_COUNT && (e.wg == 0 || e.slots > 0) && e.ord_blocks >= 0 && e.ord_slots >= 0 &&
           e.ord_blocks <= MAX_COUNT && e.ord_slots <= MAX_COUNT && (e.ord_blocks == 0 || e.ord_slots > 0) && e.kind >= KIND_NONE && e.kind <= KIND_PAIR && e.pid > 0;
}
inline double fraction(int n, int cap) { const double f = (double)n / (double)(cap > 0 ? cap : 1); return f > 1. ? 1. : f; }

inline uint64_t process_start_time(int pid) {
    char path[64];
    snprintf(path, sizeof path, "/proc/%d/stat", pid);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    char buf[1024];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    const char *p = strrchr(buf, ')');  // the command name may hold spaces and parentheses
    if (!p) return 0;
    int field = 2;
    unsigned long long v = 0;
    for (p += 1; *p && field < 22; ++p)
        if (*p == ' ') { ++field; if (field == 22) { v = strtoull(p + 1, nullptr, 10); break; } }
    return (uint64_t)v;
}

inline double shared_limit() {   // an experimenter's cap on the sum where claimants share a device (percent); 1.0 = the rule alone
    static const double lim = [] {
        const char *s = getenv("NXS_RESIDENT_SHARED_LIMIT");
        if (s && *s) { const double v = atof(s) / 100.; if (v > 0. && v <= 1.) return v; }
        return 1.0;
    }();
    return lim;
}

// One device's table as this process sees it.
class DeviceTable {
public:
    explicit DeviceTable(const std::string &key) {
        const char *prefix = getenv("NXS_RESIDENT_SHM_PREFIX");  // tests: a name of their own
        name_ = std::string("/") + (prefix && *prefix ? prefix : "nxs_resident_") + "u" + std::to_string((unsigned long)getuid()) + "_" + key;
        for (char &c : name_) if (&c != &name_[0] && !((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_')) c = '_';
        fd_ = shm_open(name_.c_str(), O_RDWR | O_CREAT, 0600);
        if (fd_ >= 0) {
            if (flock(fd_, LOCK_EX) == 0) {
                struct stat st;
                bool ok = fstat(fd_, &st) == 0 && st.st_uid == getuid();
                if (ok && (size_t)st.st_size < sizeof(Table)) ok = ftruncate(fd_, sizeof(Table)) == 0;
                void *p = ok ? mmap(nullptr, sizeof(Table), PROT_READ | PROT_WRITE, MAP_SHARED, fd_, 0) : MAP_FAILED;
                if (p != MAP_FAILED) {
                    Table *t = static_cast<Table *>(p);
                    if (t->magic == MAGIC && t->version == VERSION) tab_ = t;
                    else if (t->magic == MAGIC && other_version_in_use(t)) {   // another build of the library holds claims here: leave them alone
                        munmap(p, sizeof(Table));
                        if (getenv("NXS_DEBUG_PATCHES")) fprintf(stderr, "[nxs] registry %s: written by another version of the library and in use; this process keeps a table of its own\n", name_.c_str());
                    } else {
                        std::memset(t, 0, sizeof(Table)); t->magic = MAGIC; t->version = VERSION; t->next_seq = 1;
                        tab_ = t;
                    }
                }
                flock(fd_, LOCK_UN);
            }
            if (!tab_) { close(fd_); fd_ = -1; }
        }
        if (!tab_) {  // no shared memory here: this process only
            tab_ = &local_;
            std::memset(tab_, 0, sizeof(Table));
            tab_->magic = MAGIC; tab_->version = VERSION; tab_->next_seq = 1;
        }
    }
    ~DeviceTable() {
        if (tab_ && tab_ != &local_) munmap(tab_, sizeof(Table));
        if (fd_ >= 0) close(fd_);
    }
    DeviceTable(const DeviceTable &) = delete;
    DeviceTable &operator=(const DeviceTable &) = delete;
    bool shared() const { return fd_ >= 0; }
    int lock_fd() const { return fd_; }   // (tests)
    const std::string &name() const { return name_; }

    // a handle exists on the device (nxs_dyn_create); multi_rank is brought up to date by claim()
    void add(uint64_t handle) {
        Guard g(this);
        sweep();
        if (find(handle)) return;
        for (Entry &e : tab_->e)
            if (!e.in_use) {
                e = Entry{};
                e.in_use = 1; e.pid = (int32_t)getpid(); e.start = my_start(); e.handle = handle; e.seq = tab_->next_seq++;
                e.pidns = my_pid_namespace(); e.stamp = boot_seconds();
                return;
            }
        // table full: the handle stays unregistered and every claim of it is refused (claim() finds no entry)
    }
    void remove(uint64_t handle) {
        Guard g(this);
        if (Entry *e = find(handle)) *e = Entry{};
    }
    // gives back the claim of one grid kind (KIND_NONE: whatever is held): a handle's resident loop going away must not take the pair patches' claim with it
    void release(uint64_t handle, int kind = KIND_NONE) {
        Guard g(this);
        if (Entry *e = find(handle))
            if (kind == KIND_NONE || e->kind == kind) { e->wg = 0; e->slots = 0; e->kind = KIND_NONE; }
    }
    // the largest grid of blocks that may wait inside this handle's ORDINARY kernels (k_smooth_halo, k_halo_pull, the boundary patches of k_substep_fused<HALO>):
    // blocks, and the blocks of that kernel the device holds at once.  0, 0: none (a single-rank handle).  Replaces what was registered before.
    void set_ordinary(uint64_t handle, int blocks, int slots) {
        Guard g(this);
        if (Entry *e = find(handle)) { e->ord_blocks = blocks > 0 ? blocks : 0; e->ord_slots = blocks > 0 ? (slots > 0 ? slots : 1) : 0; e->stamp = boot_seconds(); }
    }
    // "still here" (no lock: one aligned 64-bit store into the handle's own entry); callers rate-limit it
    void touch(uint64_t handle) {
        for (Entry &e : tab_->e)
            if (e.in_use && e.pid == (int32_t)getpid() && e.handle == handle) { __atomic_store_n(&e.stamp, boot_seconds(), __ATOMIC_RELAXED); return; }
    }
    // May this handle run a grid of `wg` workgroups that wait for each other (the device holds `slots` of that build at once)?  Its previous claim is replaced.
    // why (optional) receives the reason of a refusal.
    bool claim(uint64_t handle, int wg, int slots, bool multi_rank, std::string *why = nullptr, int kind = KIND_RESIDENT) {
        Guard g(this);
        sweep();
        Entry *me = find(handle);
        if (!me) { if (why) *why = "the device's registry of resident grids is full"; return false; }
        me->wg = 0; me->slots = 0; me->kind = KIND_NONE; me->multi_rank = multi_rank ? 1 : 0; me->stamp = boot_seconds();
        if (wg < 0 || slots <= 0 || wg > MAX_COUNT || slots > MAX_COUNT) { if (why) *why = "a claim outside the sane range"; return false; }
        const double mine = (double)wg / (double)slots;   // (not capped: more workgroups than the device holds never fit)
        double claims = 0., waiting = 0.;
        int claimants = 1, handles = 0, foreign = 0;
        for (const Entry &e : tab_->e) {
            if (!e.in_use || !entry_valid(e)) continue;
            ++handles;
            if (&e == me) continue;
            waiting += fraction(e.ord_blocks, e.ord_slots);   // every block of a co-tenant that may spin, spinning
            if (e.wg <= 0) continue;
            claims += fraction(e.wg, e.slots);
            ++claimants;
            if (e.pid != (int32_t)getpid()) ++foreign;
        }
        const double used = mine + claims + waiting;
        const double limit = claimants > 1 ? shared_limit() : 1.0;
        if (used > limit + 1e-9) {
            if (why) {
                char buf[320];
                snprintf(buf, sizeof buf, "%d workgroups of %d slots = %.1f %% of the device, beside %.1f %% claimed by %d other grid(s) (%d of another process) and %.1f %% that the "
                                          "ordinary kernels of the %d other handle(s) may hold while they wait for a neighbour rank: %.1f %% > %.0f %%", wg, slots, 100. * mine,
                         100. * claims, claimants - 1, foreign, 100. * waiting, handles - 1, 100. * used, 100. * limit);
                *why = buf;
            }
            return false;
        }
        me->wg = wg; me->slots = slots; me->kind = kind;
        return true;
    }
    // the sum of all claims on the device, as a fraction (diagnostics, tests); waiting(): the registered waiting grids
    double claimed() {
        Guard g(this);
        sweep();
        double used = 0.;
        for (const Entry &e : tab_->e) if (e.in_use && entry_valid(e) && e.wg > 0) used += fraction(e.wg, e.slots);
        return used;
    }
    double waiting() {
        Guard g(this);
        sweep();
        double used = 0.;
        for (const Entry &e : tab_->e) if (e.in_use && entry_valid(e)) used += fraction(e.ord_blocks, e.ord_slots);
        return used;
    }
    int handles() {
        Guard g(this);
        sweep();
        int n = 0;
        for (const Entry &e : tab_->e) n += e.in_use ? 1 : 0;
        return n;
    }
    // tests: writes an entry as another process / namespace would have left it
    void plant(const Entry &e) {
        Guard g(this);
        for (Entry &x : tab_->e) if (!x.in_use) { x = e; x.in_use = 1; x.seq = tab_->next_seq++; return; }
    }

private:
    struct Guard {  // this process' threads by the mutex, other processes by flock (per open file description: it does not separate threads)
        DeviceTable *t;
        explicit Guard(DeviceTable *tt) : t(tt) { t->mu_.lock(); t->own_description(); if (t->fd_ >= 0) while (flock(t->fd_, LOCK_EX) != 0 && errno == EINTR) {} }
        ~Guard() { if (t->fd_ >= 0) flock(t->fd_, LOCK_UN); t->mu_.unlock(); }
    };
    // flock() is per open file description and a fork()ed child shares its parent's: the first time a child comes here it opens a description of its own (the mapping
    // is shared memory and stays) and forgets the parent's start time
    void own_description() {
        const int me = (int)getpid();
        if (me == open_pid_) return;
        open_pid_ = me; start_ = 0;
        if (fd_ >= 0) { const int nfd = shm_open(name_.c_str(), O_RDWR, 0600); if (nfd >= 0) { close(fd_); fd_ = nfd; } }
    }
    uint64_t my_start() { if (!start_) start_ = process_start_time((int)getpid()); return start_; }
    Entry *find(uint64_t handle) {
        for (Entry &e : tab_->e) if (e.in_use && e.pid == (int32_t)getpid() && e.pidns == my_pid_namespace() && e.handle == handle && e.start == my_start()) return &e;
        return nullptr;
    }
    static bool other_version_in_use(const Table *t) {   // (another layout: only the head of an entry -- in_use, pid, start -- is the same in every version)
        const size_t stride_v1 = 48;
        const char *base = reinterpret_cast<const char *>(t) + 16;
        for (int i = 0; i < MAX_ENTRIES; ++i) {
            int32_t in_use, pid; uint64_t start;
            std::memcpy(&in_use, base + i * stride_v1, 4); std::memcpy(&pid, base + i * stride_v1 + 4, 4); std::memcpy(&start, base + i * stride_v1 + 8, 8);
            if (in_use == 1 && pid > 0 && process_start_time(pid) == start && start != 0) return true;
        }
        return false;
    }
    // Entries whose owner is gone.  Same PID namespace: pid not alive, or alive with another start time.  Another namespace (a container sharing /dev/shm): its
    // pids mean nothing here -- such an entry goes when its owner has not touched it for STALE_SECONDS.  Entries with counts out of range are not this code's.
    void sweep() {
        const uint64_t now = boot_seconds(), ns = my_pid_namespace();
        for (Entry &e : tab_->e) {
            if (!e.in_use) continue;
            if (!entry_valid(e)) { e = Entry{}; continue; }
            if (e.pid == (int32_t)getpid() && e.pidns == ns) { if (e.start != my_start()) e = Entry{}; continue; }
            if (e.pidns != ns || ns == 0) { if (now > e.stamp + STALE_SECONDS) e = Entry{}; continue; }
            const bool dead = (kill(e.pid, 0) != 0 && errno == ESRCH) || process_start_time(e.pid) != e.start;
            if (dead) e = Entry{};
        }
    }
    std::string name_;
    int open_pid_ = (int)getpid();
    int fd_ = -1;
    Table *tab_ = nullptr;
    Table local_{};
    std::mutex mu_;
    uint64_t start_ = 0;
};

// the tables of this process, one per device key, made on first use and kept until exit
inline DeviceTable &table_for(const std::string &key) {
    static std::mutex mu;
    static std::map<std::string, DeviceTable *> tabs;
    std::lock_guard<std::mutex> lk(mu);
    auto it = tabs.find(key);
    if (it == tabs.end()) it = tabs.emplace(key, new DeviceTable(key)).first;
    return *it->second;
}

}  // namespace nxs_reg

#endif  // NXS_RESIDENT_REGISTRY_HPP
