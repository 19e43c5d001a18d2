// nxs_resident_registry.hpp -- who may run a RESIDENT sub-step grid on a device (host-only, plain C++17 + POSIX; no HIP).
//
// k_substep_resident needs every workgroup of its grid on a CU at once, and its workgroups spin for each other: two such grids whose
// sum does not fit keep each other's missing workgroups from ever starting (both time out), and a grid that fills the device leaves no
// room for the ordinary kernels of a co-tenant whose own waves may be waiting for THIS grid's rank (prep, the smoother with the exchange
// inside, k_halo_pull) -- the lost steps of round 3's shared-device rehearsals.  So a handle CLAIMS its workgroup
// slots before it builds the resident loop, and the claim is refused up front -- the step then runs one kernel per sub-step, deterministically --
// unless everything claimed on the device stays within
//     1.0                      the only claimant, and either a single-rank handle or the only handle on the device
//     shared_limit (0.70)      otherwise: several claimants, or a several-rank claimant with other handles on the device (0.68 is the occupancy
//                              at which 32 of 32 shared-device rehearsal passes ran clean in round 3; 0.96 lost 1 in 12)
// Handles of OTHER PROCESSES on the device count: the table lives in POSIX shared memory named after the device (its PCI bus id), guarded by
// flock() on the segment (released by the kernel when a process dies), entries of dead processes are dropped by pid + start time.  Two MPI
// ranks per GPU -- a plausible deployment, and what the 8-rank tests are -- therefore see each other.  Where shared memory is not available
// the table is per process, as in round 3.
// tests/native/registry_host.cpp drives this file from two processes (tests/test_resident_registry.py): a second process' grid is refused up
// front, a dead process' claim is reclaimed.
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
#include <unistd.h>

namespace nxs_reg {

constexpr uint32_t MAGIC = 0x4e585352u;  // 'NXSR'
constexpr int MAX_ENTRIES = 256;

struct Entry {
    int32_t in_use, pid;
    uint64_t start;     // the process' start time (/proc/<pid>/stat field 22): a recycled pid is not the same process
    uint64_t handle;    // the handle's address in its process (pid + handle name an entry)
    uint64_t seq;       // registration order on the device
    int32_t wg, slots;  // the resident claim: workgroups of the grid / workgroups of that kernel build the device holds at once; wg == 0: no claim
    int32_t multi_rank, reserved;
};
struct Table {
    uint32_t magic, version;
    uint64_t next_seq;
    Entry e[MAX_ENTRIES];
};

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

inline double shared_limit() {
    static const double lim = [] {
        const char *s = getenv("NXS_RESIDENT_SHARED_LIMIT");  // percent; experiments only (rehearsals that knowingly fill a shared device)
        if (s && *s) { const double v = atof(s) / 100.; if (v > 0. && v <= 1.) return v; }
        return 0.70;
    }();
    return lim;
}

// One device's table as this process sees it.
class DeviceTable {
public:
    explicit DeviceTable(const std::string &key) {
        const char *prefix = getenv("NXS_RESIDENT_SHM_PREFIX");  // tests: a name of their own
        name_ = std::string("/") + (prefix && *prefix ? prefix : "nxs_resident_") + key;
        for (char &c : name_) if (&c != &name_[0] && !((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_')) c = '_';
        fd_ = shm_open(name_.c_str(), O_RDWR | O_CREAT, 0666);
        if (fd_ >= 0) {
            (void)fchmod(fd_, 0666);  // (another user's process on the same device must be able to see the claims)
            if (flock(fd_, LOCK_EX) == 0) {
                struct stat st;
                bool ok = fstat(fd_, &st) == 0;
                if (ok && (size_t)st.st_size < sizeof(Table)) ok = ftruncate(fd_, sizeof(Table)) == 0;
                void *p = ok ? mmap(nullptr, sizeof(Table), PROT_READ | PROT_WRITE, MAP_SHARED, fd_, 0) : MAP_FAILED;
                if (p != MAP_FAILED) {
                    tab_ = static_cast<Table *>(p);
                    if (tab_->magic != MAGIC || tab_->version != 1) { std::memset(tab_, 0, sizeof(Table)); tab_->magic = MAGIC; tab_->version = 1; tab_->next_seq = 1; }
                }
                flock(fd_, LOCK_UN);
            }
            if (!tab_) { close(fd_); fd_ = -1; }
        }
        if (!tab_) {  // no shared memory here: this process only
            tab_ = &local_;
            std::memset(tab_, 0, sizeof(Table));
            tab_->magic = MAGIC; tab_->version = 1; tab_->next_seq = 1;
        }
    }
    ~DeviceTable() {
        if (tab_ && tab_ != &local_) munmap(tab_, sizeof(Table));
        if (fd_ >= 0) close(fd_);
    }
    DeviceTable(const DeviceTable &) = delete;
    DeviceTable &operator=(const DeviceTable &) = delete;
    bool shared() const { return fd_ >= 0; }
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
                return;
            }
        // table full: the handle stays unregistered and every claim of it is refused (claim() finds no entry)
    }
    void remove(uint64_t handle) {
        Guard g(this);
        if (Entry *e = find(handle)) *e = Entry{};
    }
    void release(uint64_t handle) {
        Guard g(this);
        if (Entry *e = find(handle)) { e->wg = 0; e->slots = 0; }
    }
    // May this handle run a resident grid of `wg` workgroups (the device holds `slots` of that build at once)?  Its previous claim is replaced.
    // why (optional) receives the reason of a refusal.
    bool claim(uint64_t handle, int wg, int slots, bool multi_rank, std::string *why = nullptr) {
        Guard g(this);
        sweep();
        Entry *me = find(handle);
        if (!me) { if (why) *why = "the device's registry of resident grids is full"; return false; }
        me->wg = 0; me->slots = 0; me->multi_rank = multi_rank ? 1 : 0;
        double used = (double)wg / (double)(slots > 0 ? slots : 1);  // fractions of the device: the builds differ in how many of their workgroups a CU holds
        int claimants = 1, handles = 0, foreign = 0;
        for (const Entry &e : tab_->e) {
            if (!e.in_use) continue;
            ++handles;
            if (&e == me || e.wg <= 0) continue;
            used += (double)e.wg / (double)(e.slots > 0 ? e.slots : 1);
            ++claimants;
            if (e.pid != (int32_t)getpid()) ++foreign;
        }
        const bool alone = claimants == 1 && (!multi_rank || handles == 1);
        const double limit = alone ? 1.0 : shared_limit();
        if (used > limit + 1e-9) {
            if (why) {
                char buf[256];
                snprintf(buf, sizeof buf, "%d workgroups of %d slots would bring the device's resident grids to %.0f %% (%d claimant(s), %d of another process, %d handle(s) on the device): "
                                          "the limit is %.0f %%%s", wg, slots, 100. * used, claimants, foreign, handles, 100. * limit,
                         alone ? "" : " where a device is shared (headroom for the co-tenants' ordinary kernels)");
                *why = buf;
            }
            return false;
        }
        me->wg = wg; me->slots = slots;
        return true;
    }
    // the sum of all claims on the device, as a fraction (diagnostics, tests)
    double claimed() {
        Guard g(this);
        sweep();
        double used = 0.;
        for (const Entry &e : tab_->e) if (e.in_use && e.wg > 0) used += (double)e.wg / (double)(e.slots > 0 ? e.slots : 1);
        return used;
    }
    int handles() {
        Guard g(this);
        sweep();
        int n = 0;
        for (const Entry &e : tab_->e) n += e.in_use ? 1 : 0;
        return n;
    }

private:
    struct Guard {  // this process' threads by the mutex, other processes by flock (per open file description: it does not separate threads)
        DeviceTable *t;
        explicit Guard(DeviceTable *tt) : t(tt) { t->mu_.lock(); if (t->fd_ >= 0) while (flock(t->fd_, LOCK_EX) != 0 && errno == EINTR) {} }
        ~Guard() { if (t->fd_ >= 0) flock(t->fd_, LOCK_UN); t->mu_.unlock(); }
    };
    uint64_t my_start() { if (!start_) start_ = process_start_time((int)getpid()); return start_; }
    Entry *find(uint64_t handle) {
        for (Entry &e : tab_->e) if (e.in_use && e.pid == (int32_t)getpid() && e.handle == handle && e.start == my_start()) return &e;
        return nullptr;
    }
    void sweep() {  // entries of processes that are gone (killed before they could let go): pid not alive, or alive with another start time
        for (Entry &e : tab_->e) {
            if (!e.in_use || e.pid == (int32_t)getpid()) { if (e.in_use && e.pid == (int32_t)getpid() && e.start != my_start()) e = Entry{}; continue; }
            const bool dead = (kill(e.pid, 0) != 0 && errno == ESRCH) || process_start_time(e.pid) != e.start;
            if (dead) e = Entry{};
        }
    }
    std::string name_;
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
