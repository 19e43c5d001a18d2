// nxs_dyn.hip -- libnxsdyn.so: neXtSIM's per-time-step dynamics (explicitSolve + update) on MI355X.
//
// gfx950 only.  fp64 throughout; memory-bound (no MFMA).  One handle = one GPU = one HIP stream.
// Compiled with -ffp-contract=off: the reference is built without FMA contraction
// (model/Makefile:5-8, -O3 and no -march), so every a*b+c below is two roundings, like there.
//
// Kernel inventory ("v1": one kernel per reference loop; FE.cpp = model/finiteelement.cpp):
//   k_prep_elements   FE.cpp:10235-10308  geometry, slab mass, basal C_bu, per-step element constants
//   k_prep_nodes      FE.cpp:10309-10416  node-centric gather of the element->node scatters + prep nodes
//   k_sigma_bbm       FE.cpp:4137-4260    updateSigmaDamage; also emits the 6 corner forces of K4
//   k_sigma_vp        FE.cpp:10649-10726  EVP / mEVP stress
//   k_solve_move      FE.cpp:10445-10553  grad_terms gather + nodal solve + mesh move (owned nodes)
//   k_halo_pack/unpack FE.cpp:13963-13996 updateGhosts (unpack also moves the ghost nodes)
//   k_smooth          FE.cpp:10580-10608  one Jacobi sweep of the open-water smoother
//   k_ow_tail         FE.cpp:10613-10640  D_tau_w, open-water mesh move
//   k_update          FE.cpp:3946-4131    update()
//   k_free_drift      FE.cpp:10140-10176
//   k_regrid_*/k_check_* FE.cpp:8298-8309, 14536-14655 reductions
//
// Determinism: the reference scatters element contributions into nodes in ascending element
// order.  Here every node GATHERS over its element fan sorted ascending, which performs the same
// floating-point additions in the same order: no atomics, bit-reproducible, and identical to the
// serial loop.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unistd.h>
#include <vector>

#include "nxs_dyn.h"
#include "nxs_guard.hpp"
#include "nxs_patchcut.hpp"
#include "nxs_resident_registry.hpp"

#include "nxs_dyn_kernels.inl"

// ================================================================================================
// host side

namespace {

thread_local std::string g_create_error;

struct Rccl {  // RCCL entry points, resolved at comm_init (no link-time dependency)
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    void *CommInitRank = nullptr;  // ncclCommInitRank(comm*, nranks, ncclUniqueId by value (128 B), rank)
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

}  // namespace

struct NcclId { char internal[128]; };
typedef int (*nccl_comm_init_rank_t)(void **, int, NcclId, int);

using nxs_cut::HostPatches;
using nxs_cut::HostPatches2;
static_assert(NXS_CUT_BLOCK == BLOCK && NXS_CUT_T256_MAXP == NXS_T256_MAXP && NXS_CUT_RES_NBR == NXS_RES_NBR && NXS_CUT_RES_MAXNB == NXS_RES_MAXNB,
              "nxs_patchcut.hpp and nxs_dyn_kernels.inl disagree about a limit");

struct nxs_dyn_handle {
    int device = 0;
    std::string reg_key;    // the device's name in the registry of resident grids (its PCI bus id): nxs_resident_registry.hpp
    hipStream_t stream = nullptr;
    nxs_dyn_params params{};
    DevParams dp{};
    DevParams *d_dp = nullptr;   // device copy (the fused sub-step kernels read their parameters from memory)
    bool dp_dirty = true;
    bool have_mesh = false, have_state = false, have_forcing = false;
    DevMesh dm{};
    DevState ds{};
    DevWork dw{};
    DevPatches dpch{};
    int fused = 3;          // 3 (default): v3 (two sub-steps per launch) on single-rank meshes that live in the caches, else v2;
                            // 2: v3 wherever it is possible; 1: v2 fused sub-step kernel; 0: v1 two-kernel sub-step
    int pair_nodes = 0;     // v3: own nodes per patch; 0 = auto
    int pair_depth = 0;     // v3: sub-steps per launch, 2..NXS_MAX_DEPTH; 0 = auto
    int pair_depth_built = 0;
    int depth_now = 1;      // sub-steps per launch of the current step (choose_depth)
    bool pair_failed = false;   // the D-ring patches could not be built for this mesh: v2 instead
    DevPatches2 dpch2{};
    size_t pair_lds = 0;
    int pair_threads = 512;
    bool pair_ready = false;
    std::vector<void *> pair_allocs;
    int patch_nodes = 0;    // own nodes per patch; 0 = auto
    int um_ring = 0;        // fused path: apply the mesh move every um_ring sub-steps from a ring of VT buffers
                            // (1 = every sub-step; 0 = auto: once per step on meshes that stream from HBM, 1 on cache-resident ones)
    VTRing ring{};
    std::vector<void *> ring_allocs;
    int nt_mask = -1;       // non-temporal access classes of the fused kernel (1 sigma/damage, 2 UM/UT, 4 element constants); -1 = automatic:
                            // 3 where a sub-step streams more than the Infinity Cache holds (>= 1 M local triangles), 0 below
    size_t fused_lds = 0;
    std::vector<int> h_t[3];               // kept for rebuilding patches when patch_nodes changes
    std::vector<int> h_n2n, h_n2n_cnt;     // NodalConnectivity rows [W2][Nn] + counts (for the blocked smoother's tables)
    std::vector<int> h_n2e;                // NodalElementConnectivity rows [W1][Nn], -1 = pad (for k_prep_fused's rows in patch slots)
    int pair_own_max = 0;                  // most own nodes of a multi-sub-step patch
    size_t prep_lds = 0;                   // LDS of k_prep_fused for the current patches; 0: the two separate prep kernels run
    int band_nodes = -1;                   // option "band_patch_nodes": several ranks, resident loop: the sent nodes in patches of their own of this size; -1 = 48, 0 = off
    int pair_regs = -1;                    // option "pair_regs": two sub-steps per launch with the stresses between them in registers (k_substep_pair): -1 = on
                                           // single-rank meshes of more than 65 k nodes (an even number of sub-steps), 0 = never, 1 = wherever depth 2 runs
    bool pair_kernel = false;              // the multi-sub-step patches were cut for k_substep_pair
    int pair_hilbert = 0;                  // option "pair_hilbert": 1 = the two-ring patches of a single rank are cut along a Hilbert curve even where the caller's numbering has locality
    int pair_move = -1;                    // option "pair_move": k_substep_pair on a single rank applies the mesh move of its two sub-steps itself (no ring of velocity slots, no
                                           // k_move_ring): -1 = automatic, 0 = never (the move deferred to one flush per step), 1 = wherever that kernel runs on one rank
    bool move_now = false, last_move_in_pair = false;                 // ... decided for the step being built (run_substeps), read by launch_multi
    int pair_flow = -1;                    // option "pair_flow": the pairs of sub-steps of a step as ONE data-flow launch (k_substep_flow): -1 = wherever k_substep_pair runs on a
                                           // single rank with 512 threads, 0 = never (one launch per pair), 1 = the same as -1
    bool flow_ready = false, flow_failed = false;
    PairFlow flow{};                       // its dependency lists, queues and counters (they go with the patches)
    size_t flow_words = 0;                 // ... the words zeroed before every launch
    int flow_grid = 0;
    int pair_T = 512;                      // option "pair_threads": threads of a k_substep_pair workgroup on a single rank (512: two per CU; 256: four per CU, smaller patches)
    PairHalo pairh{};                      // several ranks: the patches' duties in the exchange inside k_substep_pair<HALO>, the ticket words
    bool pair_claim = false;               // ... and their claim on the device's workgroup slots (nxs_resident_registry.hpp)
    int pair_hint = 0;                     // the patch size the planner kept for the previous mesh (tried first after a regrid)
    int prep_fused = -1;                   // option "prep_fused": -1 where it pays (single rank, records only, >= 250 k triangles), 0 never, 1 wherever it can run
    size_t smooth_lds = 0;
    // node-ring patches for the smoother alone (single rank, meshes on the one-sub-step-per-launch kernels): D sweeps per launch
    DevPatches2 dsm{};
    std::vector<void *> sm_allocs;
    bool sm_ready = false, sm_failed = false;
    int sm_depth = 0;  // option smooth_depth: sweeps per launch of the smoother on its own node-ring patches (0 = automatic)
    size_t sm_lds = 0;
    std::vector<unsigned char> h_ghost;
    std::vector<double> h_x0, h_y0;
    std::vector<void *> patch_allocs;
    std::vector<void *> mesh_allocs, state_allocs;
    // Device memory that a rebuild INSIDE a step lets go of is parked here and freed by the next call that is outside a step AND outside the time loop (set_mesh,
    // set_params, set_option, destroy -- NOT put_state / set_forcing, which a host with its thermodynamics on the CPU calls between any two steps): a hipFree
    // synchronises the whole device, and where ranks share one (tests, rehearsals, two MPI ranks per GPU) a neighbour rank's kernel may already be spinning for this
    // rank's next launch -- the free would wait for it, the launch for the free.
    std::vector<void *> retired;
    bool in_step = false;
    // halo
    bool have_halo = false;
    int rank = 0, nranks = 1;
    std::vector<int> send_procs, send_offsets, recv_procs, recv_offsets;   // as nxs_dyn_set_halo keeps them: the caller's neighbours, then the directions it added (empty segments)
    int ns_caller = 0, nr_caller = 0;      // neighbours the caller's own lists named (the low-level nxs_dyn_ipc_connect takes tables for those)
    int one_directional = 0;               // test door "halo_one_directional": 1 = set_halo takes the lists as given and ipc_connect does not refuse (round 4's defect, for the delay tests)
    unsigned ipc_delay_opt = 0;            // test door "ipc_delay": rank << 16 | point << 8 | units (include/nxs_dyn.h)
    int ord_blocks = 0, ord_slots = 0;     // the largest grid of blocks that may WAIT inside this handle's ordinary kernels, as registered on the device (nxs_resident_registry.hpp)
    long long reg_touched = 0;             // when the registry entry was last touched (seconds)
    int *d_send_index = nullptr, *d_send_seg = nullptr, *d_send_off = nullptr;
    int *d_recv_index = nullptr, *d_recv_seg = nullptr, *d_recv_off = nullptr;
    double *d_send_buf = nullptr, *d_recv_buf = nullptr;
    std::vector<void *> halo_allocs;
    Rccl rccl;
    void *comm = nullptr;
    // device-direct transport (peer-mapped mailboxes)
    bool ipc_ready = false;
    IpcDev ipc{};
    void *ipc_block = nullptr;             // my mailbox allocation (exported)
    size_t ipc_block_bytes = 0;
    bool ipc_uncached = false;             // the mailbox is MTYPE_UC memory (what the fence-less in-kernel exchange relies on)
    int ipc_pad = 0, ipc_cap = 0;          // option "ipc_pad" / the number of received nodes the exported mailbox has room for
    std::vector<void *> ipc_peer_base;     // opened peer mailboxes (to close)
    std::vector<void *> ipc_local_peers;   // mailboxes of other handles of THIS process this handle stores through (counted in g_mailboxes)
    std::vector<void *> ipc_allocs;
    int *d_recv_procs = nullptr;
    // halo exchange fused into the sub-step kernel (device-direct transport + fused path)
    double *f_snap[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // wind0, wind1, ocean0, ocean1, ssh0, ssh1 (forcing pair)
    bool have_pair = false;
    std::vector<void *> forcing_allocs;
    // v4: the whole sub-step loop in one resident launch (option "fused" = 4; see k_substep_resident)
    DevResident res{};
    std::vector<void *> res_allocs;  // its tables, exchange buffer and ghost ring (a pool of their own: rebuilt whenever an option of the loop changes)
    bool res_ready = false, res_failed = false;
    int res_wide = 0;     // option resident_wide
    bool res_pow4 = true; // the build for BBM's default exponent (two squarings instead of pow)
    bool res_big = false; // k_substep_resident_big: one large patch per CU (the patches hold more than one element per thread)
    bool cut_big = false; // the mesh was cut for that kernel (not what the one-launch-per-sub-step kernel wants: re-cut if the resident loop is refused)
    bool no_big_cut = false;
    int res_substeps = 0; // the number of sub-steps the tables (the ghosts' ring) were sized for
    int res_wpe = 4;      // waves per SIMD of the resident kernel build in use (2 on several ranks where one workgroup per CU covers the partition)
    int res_overlap = -1; // option resident_overlap: interior elements of the next sub-step computed while the exchange is awaited; -1 = where it pays
                          // (the large patches of k_substep_resident_big: one workgroup per CU, nothing else fills its wait), 0 never, 1 wherever built
    bool res_ovl = false; // what the tables of the resident loop were built for
    size_t res_lds = 0;
    double *d_vt3 = nullptr;
    double *d_icediag = nullptr;           // [Ne][NXS_ICE_DIAG_FIELDS] rows of nxs_dyn_ice_diagnostics (state pool: goes with the mesh)
    double *d_icediag_soa = nullptr;       // [NXS_ICE_DIAG_FIELDS][Ne] the same per field, made when the host asks for its vectors
    double *smooth_second = nullptr;       // the ring slot that equals M_VT after the sub-step loop (the smoother's second buffer), or NULL
    int sig_loc = 0;                       // where M_sigma / M_damage are current: 0 = the state arrays, 1 = the records in S4a (left there by
                                           // the fused sub-step loop; k_update works on them, the arrays follow on demand: ensure_arrays)
    int trace_branches = 0;                // option "trace_branches": the per-loop kernels keep the branch trace of updateSigmaDamage (dw.trace)
    int shape_mem = -1;                    // option "shape_mem": the several-sub-steps kernel reads M_shape_coeff from per-step records (1, and -1 = automatic)
                                           // or rebuilds it from the staged coordinates like the one-sub-step kernel (0)
    double *d_srec = nullptr;              // the records (allocated at set_mesh)
    int work_arrays = 0;                   // option "work_arrays": the prep kernels also fill the one-array-per-quantity work vectors
    int pin_host = 0;                      // option "pin_host": page-lock the caller's state / forcing vectors on first use
    std::map<const void *, size_t> pinned; // what this handle has registered with hipHostRegister
    int halo_fused = 1;                    // option "halo_fused"
    int res_no_release = 0;                // option "resident_release" = 0 (kept here: h->hf is rebuilt with the tables)
    int smooth_persist = -1;               // option "smooth_persist": the 50 sweeps with the exchange inside as ONE launch of persistent workgroups (k_smooth_persist): -1 / 1 = on, 0 = 50 launches of k_smooth_halo
    bool hf_ready = false;
    HaloFused *d_hf = nullptr;  // device copy of hf with the mailbox addresses filled in (what k_substep_fused<.., HALO> reads)
    bool d_hf_dirty = true;
    HaloFused hf{};
    std::vector<void *> hf_allocs;
    std::vector<int> h_send_index, h_recv_index;   // host copies of the halo lists
    std::vector<char> h_sent;                      // [No] != 0: an own node this rank sends (the cut for the resident loop puts those in the small boundary patches)
    std::shared_ptr<HostPatches> hp;  // host copy of the patches (re-uploaded boundary-first for the fused halo)
    nxs_dyn_halo_fn halo_fn = nullptr;  // host-staged exchange through the caller's communicator
    void *halo_ctx = nullptr;
    double *h_send = nullptr, *h_recv = nullptr;  // pinned staging buffers
    // reductions
    RegridPartial *d_partials = nullptr, *d_regrid = nullptr;
    int *d_crash = nullptr;
    int n_partials = 0;
    // graph of the sub-step loop
    int use_graph = 1;
    hipGraphExec_t substep_graph = nullptr, tail_graph = nullptr;
    bool graph_valid = false, tail_graph_valid = false;
    // timing: a ring of event sets so that steps can be enqueued back to back; a set is harvested
    // (its elapsed times added to the sums) when it is about to be reused or when timing is read
    static constexpr int NSETS = 8;
    hipEvent_t ev[NSETS][5] = {};
    hipEvent_t ev_flush[NSETS][2] = {};   // around the last k_move_ring of a step (launched behind the sub-step graph so that it can be timed on its own)
    bool flush_timed[NSETS] = {};
    double sum_flush_ms = 0.;
    bool set_pending[NSETS] = {};
    int set_next = 0;
    double sum_ms[4] = {0, 0, 0, 0};
    int sum_steps = 0;
    std::vector<float> step_ms;  // device time of every step since "timing_reset" (at most 4096 kept): nxs_dyn_get_step_times
    hipEvent_t *cur = nullptr;  // event set of the step being enqueued (nullptr: untimed)
    // sums over the patch tables (filled where the tables are uploaded) and what the last step launched: nxs_dyn_get_traffic_model
    struct PatchSums { double nP = 0, M = 0, E = 0, O = 0, W = 0; } sums1;                       // DevPatches: staged nodes, elements, own nodes, written elements
    struct PatchSums2 { double nP = 0, W = 0, E1_second_round = 0; std::vector<double> N, E; } sums2;   // DevPatches2: nodes per level N_0..N_D, elements per level E_1..E_D; elements of E_1 beyond the first 512 of their patch
    int last_kernel = 0, last_ring_count = 0, last_prep = 0;                                     // NXS_KERNEL_* / slots of the last k_move_ring / NXS_PREP_*
    bool last_deferred = false, last_halo_in_kernel = false;
    nxs_dyn_timing timing{};
    int timing_enabled = 1;
    std::string err;
};

namespace {

// the kernel family of this step: option "fused", except that the branch trace lives in the per-loop kernels
inline int eff_fused(const nxs_dyn_handle *h) { return h->trace_branches ? 0 : h->fused; }
int build_halo_fused(nxs_dyn_handle *h);  // (defined with the launch logic below)
int build_resident(nxs_dyn_handle *h);
void resident_registry_release(const nxs_dyn_handle *h, int kind);
bool resident_registry_claim(const nxs_dyn_handle *h, int workgroups, int slots, std::string *why, int kind);
void release_resident(nxs_dyn_handle *h);
void release_graph(nxs_dyn_handle *h);
bool multi_rank(const nxs_dyn_handle *h);
void register_waiting_grid(nxs_dyn_handle *h, int blocks, int slots, bool reset = false);
void register_smoother_grid(nxs_dyn_handle *h);

int fail(nxs_dyn_handle *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

// handler of every extern "C" function-try-block of this file (nxs_guard.hpp): a std::bad_alloc from a table that cannot grow, or anything
// else the host code throws, becomes a status code and nxs_dyn_last_error's text -- never an exception across the ABI
int dyn_caught(nxs_dyn_handle *h, const char *entry) noexcept {
    return nxs_guard::caught(entry, [h](int code, const char *text) { (void)fail(h, code, "%s", text); });
}

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t _e = (call);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return fail(h, NXS_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

template <typename T>
int dev_alloc(nxs_dyn_handle *h, std::vector<void *> &pool, T **out, size_t count) {
    void *p = nullptr;
    HIPCHK(h, hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    pool.push_back(p);
    *out = static_cast<T *>(p);
    return NXS_OK;
}

template <typename T>
int dev_upload(nxs_dyn_handle *h, std::vector<void *> &pool, const T **out, const std::vector<T> &v) {
    T *p = nullptr;
    int rc = dev_alloc(h, pool, &p, v.size());
    if (rc) return rc;
    if (!v.empty()) HIPCHK(h, hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // v may be a temporary of the caller
    *out = p;
    return NXS_OK;
}

void free_pool(std::vector<void *> &pool) {
    for (void *p : pool) (void)hipFree(p);
    pool.clear();
}
// the same for a pool of a live handle: parked while a step is being enqueued (see nxs_dyn_handle::retired)
void drop_pool(nxs_dyn_handle *h, std::vector<void *> &pool) {
    if (!h->in_step) { free_pool(pool); return; }
    h->retired.insert(h->retired.end(), pool.begin(), pool.end());
    pool.clear();
}
void flush_retired(nxs_dyn_handle *h) {
    if (!h->retired.empty() && !h->in_step) { (void)hipSetDevice(h->device); free_pool(h->retired); }
}
struct StepScope {  // marks "a step is being enqueued" for the length of an entry point
    nxs_dyn_handle *h; bool was;
    explicit StepScope(nxs_dyn_handle *hh) : h(hh), was(hh->in_step) { h->in_step = true; }
    ~StepScope() { h->in_step = was; }
};

inline int nblocks(int n) { return n > 0 ? (n + BLOCK - 1) / BLOCK : 1; }

void derive_params(nxs_dyn_handle *h) {
    const nxs_dyn_params &p = h->params;
    DevParams &d = h->dp;
    d.dtime_step = p.dtime_step;
    d.substeps = p.substeps;
    d.dte = p.dtime_step / (double)p.substeps;  // FE.cpp:10185
    d.dynamics_type = p.dynamics_type;
    d.basal_stress_type = p.basal_stress_type;
    d.young_cat = p.ice_cat_type == NXS_ICECAT_YOUNG_ICE;
    d.newice_type = p.newice_type;
    d.equal_ridging = p.equal_ridging;
    d.use_young_myi = p.use_young_ice_in_myi_reset;
    d.young = p.young; d.nu0 = p.nu0; d.tan_phi = p.tan_phi; d.compr_strength = p.compr_strength;
    d.compaction_param = p.compaction_param;
    d.utrs = p.undamaged_time_relaxation_sigma;
    d.ers_m1 = p.exponent_relaxation_sigma - 1.;  // FE.cpp:4186
    d.ers_int = (d.ers_m1 >= 0. && d.ers_m1 <= 16. && d.ers_m1 == std::floor(d.ers_m1)) ? (int)d.ers_m1 : -1;
    d.compression_factor = p.compression_factor;
    d.ecf = p.exponent_compression_factor;
    d.min_h = p.min_h; d.min_c = p.min_c;
    d.min_m = NXS_RHOI * p.min_h;  // FE.cpp:10191
    d.qdw = p.quad_drag_coef_water; d.ldw = p.lin_drag_coef_water;
    d.qda = p.quad_drag_coef_air; d.lda = p.lin_drag_coef_air;
    d.cos_ota = std::cos(p.ocean_turning_angle_rad);  // FE.cpp:10187-10188
    d.sin_ota = std::sin(p.ocean_turning_angle_rad);
    d.k1 = p.basal_k1; d.k2 = p.basal_k2; d.Cb = p.basal_Cb; d.u0 = p.basal_u_0;
    d.evp_e = p.evp_e; d.evp_Pstar = p.evp_Pstar; d.evp_C = p.evp_C; d.evp_dmin = p.evp_dmin;
    d.mevp_beta = p.mevp_beta;
    if (p.dynamics_type == NXS_DYN_EVP) {  // FE.cpp:10705-10713
        const double T = p.dtime_step / 3.;
        d.ralpha1 = 0.5 * d.dte / T;
        d.ralpha2 = 0.5 * d.dte / T * p.evp_e * p.evp_e;
    } else {  // FE.cpp:10724
        d.ralpha1 = 1. / p.mevp_alpha;
        d.ralpha2 = 1. / p.mevp_alpha;
    }
    d.sqrt_nu_rhoi = std::sqrt(2. * (1. + p.nu0) * NXS_RHOI);  // FE.cpp:4140
    // initFETensors, FE.cpp:1491-1507
    for (double &x : d.D) x = 0.;
    const double Dunit_factor = 1. / (1. - p.nu0 * p.nu0);
    d.D[0] = Dunit_factor * 1.;
    d.D[1] = Dunit_factor * p.nu0;
    d.D[3] = Dunit_factor * p.nu0;
    d.D[4] = Dunit_factor * 1.;
    d.D[8] = Dunit_factor * (1. - p.nu0) / 2.;
    h->graph_valid = false;       // kernel arguments are baked into the graphs
    h->tail_graph_valid = false;
    h->dp_dirty = true;
}

int check_params(nxs_dyn_handle *h, const nxs_dyn_params *p) {
    if (!p) return fail(h, NXS_ERR_INVALID, "params is NULL");
    if (!(p->dtime_step > 0.) || p->substeps < 1) return fail(h, NXS_ERR_INVALID, "dtime_step/substeps invalid");
    if (p->dynamics_type < NXS_DYN_BBM || p->dynamics_type > NXS_DYN_MEVP)
        return fail(h, NXS_ERR_INVALID, "unknown dynamics_type %d (FE.cpp:1352-1358 allows bbm|no_motion|free_drift|evp|mevp)", p->dynamics_type);
    if (p->basal_stress_type != NXS_BASAL_NONE && p->basal_stress_type != NXS_BASAL_LEMIEUX)
        return fail(h, NXS_ERR_INVALID, "unknown basal_stress_type %d", p->basal_stress_type);
    return NXS_OK;
}

int harvest(nxs_dyn_handle *h, int k) {
    if (!h->set_pending[k]) return NXS_OK;
    HIPCHK(h, hipEventSynchronize(h->ev[k][4]));
    float whole = 0.f;
    for (int i = 0; i < 4; ++i) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev[k][i], h->ev[k][i + 1]));
        h->sum_ms[i] += ms;
        whole += ms;
    }
    if (h->flush_timed[k]) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev_flush[k][0], h->ev_flush[k][1]));
        h->sum_flush_ms += ms;
        h->flush_timed[k] = false;
    }
    if (h->step_ms.size() < 4096) h->step_ms.push_back(whole);
    h->sum_steps++;
    h->set_pending[k] = false;
    return NXS_OK;
}

// option "pair_flow": 1 = on, 0 = off, -1 = automatic
bool flow_wanted(const nxs_dyn_handle *h) { return h->pair_flow == 1; }
// option "pair_move" = -1: wherever k_substep_pair runs on a single rank (2 km: 5.27 -> 5.17 ms per step -- the launch grows by 2.8 us, the 0.27 ms flush goes)
bool pair_move_default(const nxs_dyn_handle *) { return true; }
// the data-flow build of k_substep_pair for this handle's parameters: one place for the occupancy query and the launch
const void *flow_kernel(const nxs_dyn_handle *h) {
    return h->dp.ers_int == 4 ? (const void *)k_substep_flow<512, true, 3> : (const void *)k_substep_flow<512, false, 3>;
}

#include "nxs_dyn_patches.inl"

// What nxs_dyn_ipc_export publishes (NXS_IPC_BLOB_BYTES bytes): the hipIpc handle of the mailbox, and enough about it for the
// neighbour to check the tables it was given before any kernel stores through them.  A neighbour that lives in the SAME process
// (a host that drives several GPUs from one process, one thread and one handle per GPU) cannot open its own process's hipIpc
// handle; it takes the device pointer itself.
struct IpcBlob {
    hipIpcMemHandle_t mem;      // 64 bytes
    unsigned long long magic;   // 'NXSIPC01'
    long long pid;
    unsigned long long ptr;     // the mailbox in the exporting process
    int device, uncached;
    int tr, nr;                 // received nodes in total, receive neighbours
    unsigned long long token;   // drawn once per process: equal pids in two containers that share a GPU are not "the same process"
};
static_assert(sizeof(IpcBlob) <= NXS_IPC_BLOB_BYTES, "blob too small");
constexpr unsigned long long IPC_MAGIC = 0x4e58534950433031ull;

// Layout of a mailbox allocation (doubles): [2 halves of 2*tr] [flags: nr] [sflags: nr] [sstatic: nr] [pad] [NXS_SMOOTH_SWEEPS slots of 2*tr]
struct IpcLayout { size_t flags, sflags, sstatic, slots, total; };
IpcLayout ipc_layout(size_t tr, int nr) {
    const size_t n = (size_t)std::max(nr, 1);
    IpcLayout l;
    l.flags = 4 * tr; l.sflags = l.flags + n; l.sstatic = l.sflags + n;
    l.slots = (l.sstatic + n + 1) & ~(size_t)1;  // 16-byte aligned
    l.total = l.slots + (size_t)NXS_SMOOTH_SWEEPS * 2 * tr + 16;
    return l;
}

// Mailboxes of THIS process.  A neighbour handle of the same process stores through the exporter's raw device pointer (hipIpc cannot open
// its own process's handle), and its tables and captured graphs keep that pointer: the allocation must outlive them.  Every exported
// mailbox is registered here; a same-process connect counts itself in, and a mailbox whose owner lets go of it (a second
// nxs_dyn_ipc_export, nxs_dyn_set_halo, nxs_dyn_destroy) while peers are still connected is freed by the LAST peer that disconnects.
struct MailboxRegistry {
    struct Entry { int device; bool owner_alive; int peers; };
    std::mutex mu;
    std::map<void *, Entry> boxes;
};
MailboxRegistry g_mailboxes;
unsigned long long process_token() {
    static const unsigned long long tok = [] {
        unsigned long long t = 0;
        if (FILE *f = fopen("/dev/urandom", "rb")) { if (fread(&t, sizeof t, 1, f) != 1) t = 0; fclose(f); }
        if (t == 0) t = 0x9E3779B97F4A7C15ull * (unsigned long long)getpid() ^ (unsigned long long)(uintptr_t)&g_mailboxes;
        return t;
    }();
    return tok;
}
void mailbox_drop_locked(std::map<void *, MailboxRegistry::Entry>::iterator it) {  // (g_mailboxes.mu held)
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (it->second.device != cur) (void)hipSetDevice(it->second.device);
    (void)hipFree(it->first);
    if (it->second.device != cur) (void)hipSetDevice(cur);
    g_mailboxes.boxes.erase(it);
}

// test door "ipc_delay": the word the kernels see -- point << 8 | units on the named rank, 0 on every other
unsigned ipc_delay_word(const nxs_dyn_handle *h) {
    const unsigned v = h->ipc_delay_opt;
    return (v != 0u && (int)(v >> 16) == h->rank && (v & 0xffu) != 0u) ? (v & 0xffffu) : 0u;
}

void ipc_disconnect(nxs_dyn_handle *h) {  // the peer mappings and the tables of one nxs_dyn_ipc_connect
    for (void *p : h->ipc_peer_base) if (p) (void)hipIpcCloseMemHandle(p);
    h->ipc_peer_base.clear();
    {
        std::lock_guard<std::mutex> lk(g_mailboxes.mu);
        for (void *p : h->ipc_local_peers) {
            auto it = g_mailboxes.boxes.find(p);
            if (it == g_mailboxes.boxes.end()) continue;
            if (--it->second.peers <= 0 && !it->second.owner_alive) mailbox_drop_locked(it);  // its owner is gone: the last peer frees it
        }
    }
    h->ipc_local_peers.clear();
    free_pool(h->ipc_allocs);
    h->ipc_ready = false;
    h->ipc = IpcDev{};
}
void ipc_release(nxs_dyn_handle *h) {
    ipc_disconnect(h);
    if (h->ipc_block) {
        std::lock_guard<std::mutex> lk(g_mailboxes.mu);
        auto it = g_mailboxes.boxes.find(h->ipc_block);
        if (it == g_mailboxes.boxes.end()) (void)hipFree(h->ipc_block);
        else if (it->second.peers > 0) it->second.owner_alive = false;  // handles of this process still store through it: freed by the last of them
        else mailbox_drop_locked(it);
        h->ipc_block = nullptr;
    }
    h->ipc_uncached = false;
}
// option "pin_host": the caller's vectors (FiniteElement's M_VT, M_conc, ... live as long as the mesh) are page-locked the first
// time they are seen, so that the per-step copies of a host-side thermodynamics run at PCIe speed and overlap; a vector that was
// reallocated simply registers anew, stale registrations are dropped at set_mesh / destroy.  Failure to register is not an error.
void pin_host_buffer(nxs_dyn_handle *h, const void *p, size_t bytes) {
    if (!h->pin_host || !p || bytes == 0) return;
    auto it = h->pinned.find(p);
    if (it != h->pinned.end() && it->second >= bytes) return;
    if (it != h->pinned.end()) { (void)hipHostUnregister(const_cast<void *>(p)); h->pinned.erase(it); }
    if (hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault) == hipSuccess) h->pinned[p] = bytes;
    else (void)hipGetLastError();
}
void unpin_all(nxs_dyn_handle *h) {
    for (auto &kv : h->pinned) (void)hipHostUnregister(const_cast<void *>(kv.first));
    (void)hipGetLastError();
    h->pinned.clear();
}

// M_sigma / M_damage back into their arrays when the sub-step loop left them as records
void ensure_arrays(nxs_dyn_handle *h) {
    if (h->sig_loc == 0 || !h->have_mesh) return;
    hipLaunchKernelGGL(k_unpack_state, dim3(nblocks(h->dm.Ne)), dim3(BLOCK), 0, h->stream, h->dm, h->ds, h->dp.dynamics_type == NXS_DYN_BBM ? 1 : 0, (const double *)h->ds.S4a);
    h->sig_loc = 0;
}

void release_graph(nxs_dyn_handle *h) {
    if (h->substep_graph) { (void)hipGraphExecDestroy(h->substep_graph); h->substep_graph = nullptr; }
    if (h->tail_graph) { (void)hipGraphExecDestroy(h->tail_graph); h->tail_graph = nullptr; }
    h->graph_valid = false;
    h->tail_graph_valid = false;
}

// the resident loop's tables and its claim on the device's workgroup slots
void release_resident(nxs_dyn_handle *h) {
    drop_pool(h, h->res_allocs);
    h->res = DevResident{};
    h->d_vt3 = nullptr;
    h->res_ready = false; h->res_failed = false;
    resident_registry_release(h, nxs_reg::KIND_RESIDENT);
}

// A resident launch that gave up (k_substep_resident's bounded waits) leaves a mixture of the step's start and its end behind (no patch writes after
// it has seen the error, the ones that had finished before have written); whoever hands state to the host next says so.  Call with the stream synchronised.
int resident_error(nxs_dyn_handle *h) {
    if (h->flow_ready && h->flow.error) {   // the data-flow launch of k_substep_pair's patches: the same contract
        int ferr = 0;
        HIPCHK(h, hipMemcpyAsync(&ferr, h->flow.error, sizeof ferr, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (ferr) {
            HIPCHK(h, hipMemsetAsync(h->flow.error, 0, sizeof(int), h->stream));
            h->flow_failed = true; h->flow_ready = false; release_graph(h);
            return fail(h, NXS_ERR_HIP, "the data-flow sub-step launch gave up (code %d): a patch waited 10 s for the patches around it (is the device shared with a process that "
                                        "holds it?); the step is lost and M_UM, M_UT, sigma and damage are undefined: put the state again before going on; later steps run one "
                                        "launch per pair of sub-steps", ferr);
        }
    }
    if (!h->res_ready) return NXS_OK;
    int err = 0;
    HIPCHK(h, hipMemcpyAsync(&err, h->res.error, sizeof err, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!err) return NXS_OK;
    HIPCHK(h, hipMemsetAsync(h->res.error, 0, sizeof(int), h->stream));
    h->res_failed = true; h->res_ready = false; release_graph(h);
    resident_registry_release(h, nxs_reg::KIND_RESIDENT);
    const char *what = err == 5 ? "a patch waited 12 s for a neighbouring patch of its own rank: the workgroups of the grid were not all resident (is the device shared with another process?)"
                     : err == 6 ? "the boundary patches' publishing order stalled for 10 s (an earlier sub-step was never published: a patch of this rank is missing)"
                     : err == 7 ? "a neighbour rank's flag did not arrive within 10 s (that rank started its step late, stopped, or its launch failed)"
                                : "unknown wait";
    return fail(h, NXS_ERR_HIP, "the resident sub-step launch gave up (code %d): %s; the step is lost and M_UM, M_UT, sigma and damage are undefined (patches that had finished "
                                "before the time-out have written their result, the others have not): put the state again before going on; later steps run one kernel per sub-step", err, what);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" {

int nxs_dyn_abi_version(void) { return NXS_DYN_ABI_VERSION; }

const char *nxs_dyn_last_error(const nxs_dyn_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int nxs_dyn_default_params(nxs_dyn_params *p) try {  // model/options.cpp:43,80,109-111,314-376,397,545-547
    if (!p) return NXS_ERR_INVALID;
    std::memset(p, 0, sizeof *p);
    p->dtime_step = 200.; p->substeps = 120;
    p->dynamics_type = NXS_DYN_BBM; p->basal_stress_type = NXS_BASAL_LEMIEUX; p->ice_cat_type = NXS_ICECAT_YOUNG_ICE;
    p->newice_type = 4; p->equal_ridging = 0; p->use_young_ice_in_myi_reset = 1;
    p->young = 5.9605e+08; p->nu0 = 1. / 3.; p->tan_phi = 0.7; p->compr_strength = 1e10; p->compaction_param = -20.;
    p->undamaged_time_relaxation_sigma = 1e7; p->exponent_relaxation_sigma = 5.;
    p->compression_factor = 10e3; p->exponent_compression_factor = 1.5;
    p->min_h = 0.05; p->min_c = 0.01;
    p->quad_drag_coef_water = 0.0055; p->lin_drag_coef_water = 0.; p->quad_drag_coef_air = 0.0049; p->lin_drag_coef_air = 0.;
    p->ocean_turning_angle_rad = (NXS_PI / 180.) * 25.;
    p->basal_k1 = 10.; p->basal_k2 = 15.; p->basal_Cb = 20.; p->basal_u_0 = 5e-5;
    p->evp_e = 2.; p->evp_Pstar = 27.5e3; p->evp_C = 20.; p->evp_dmin = 1e-9;
    p->mevp_alpha = 500.; p->mevp_beta = 500.;
    p->regrid_angle = 10.;
    return NXS_OK;
} catch (...) { return dyn_caught(nullptr, "nxs_dyn_default_params"); }

int nxs_dyn_physical_constants(double *out, int32_t count) try {  // model/constants.hpp:56-87, OppositeAngle.h:4, finiteelement.hpp:549
    if (!out || count < 0) return NXS_ERR_INVALID;
    const double c[NXS_CONST_COUNT] = {NXS_RHOI, NXS_RHOW, NXS_RHOS, NXS_RHOA, NXS_GRAVITY, NXS_OMEGA, NXS_PI, NXS_DAYS_IN_SEC};
    for (int i = 0; i < count && i < NXS_CONST_COUNT; ++i) out[i] = c[i];
    return NXS_OK;
} catch (...) { return dyn_caught(nullptr, "nxs_dyn_physical_constants"); }

int nxs_dyn_selftest_quotients(int32_t device, int64_t n, uint64_t seed, int32_t mode, int64_t *mismatches) try {
    if (!mismatches || n < 0 || mode < 0 || mode > 2) return fail(nullptr, NXS_ERR_INVALID, "selftest_quotients: n >= 0, mode 0, 1 or 2, a place for the count");
    *mismatches = -1;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, NXS_ERR_HIP, "selftest_quotients: no device %d", (int)device); }
    unsigned long long *d = nullptr, host = 0ull;
    if (hipMalloc((void **)&d, sizeof *d) != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, NXS_ERR_HIP, "selftest_quotients: hipMalloc"); }
    hipError_t e = hipMemset(d, 0, sizeof *d);
    const long long chunk = 1ll << 28;   // (grid sizes stay well inside 2^31 blocks)
    for (long long done = 0; done < (long long)n && e == hipSuccess; done += chunk) {
        const long long m = std::min(chunk, (long long)n - done);
        hipLaunchKernelGGL(k_selftest_quotients, dim3((unsigned)((m + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, 0, m, (unsigned long long)seed + 0x632be59bd9b4e019ull * (unsigned long long)done, (int)mode, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(&host, d, sizeof host, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(nullptr, NXS_ERR_HIP, "selftest_quotients: %s", hipGetErrorString(e));
    *mismatches = (int64_t)host;
    return NXS_OK;
} catch (...) { return dyn_caught(nullptr, "nxs_dyn_selftest_quotients"); }

int nxs_dyn_create(const nxs_dyn_params *p, int device, nxs_dyn_handle **out) try {
    if (!out) return fail(nullptr, NXS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, NXS_ERR_NO_DEVICE, "no HIP device visible (%s): libnxsdyn has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, NXS_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    nxs_dyn_handle *h = new nxs_dyn_handle();
    int rc = check_params(h, p);
    if (rc) { g_create_error = h->err; delete h; return rc; }
    h->device = device;
    h->params = *p;
#define CREATE_CHK(call)                                                                       \
    do {                                                                                        \
        hipError_t _e = (call);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            fail(nullptr, NXS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e));           \
            (void)nxs_dyn_destroy(h); /* the stream, events and buffers made so far */           \
            return NXS_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)
    CREATE_CHK(hipSetDevice(device));
    CREATE_CHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (auto &set : h->ev) for (auto &ev : set) CREATE_CHK(hipEventCreate(&ev));
    for (auto &set : h->ev_flush) for (auto &ev : set) CREATE_CHK(hipEventCreate(&ev));
    CREATE_CHK(hipMalloc((void **)&h->d_regrid, sizeof(RegridPartial)));
    CREATE_CHK(hipMalloc((void **)&h->d_crash, sizeof(int)));
#undef CREATE_CHK
    {   // the handle is now a tenant of the device, for every process that looks (nxs_resident_registry.hpp)
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess || !bus[0]) { (void)hipGetLastError(); snprintf(bus, sizeof bus, "device%d", device); }
        h->reg_key = bus;
        nxs_reg::table_for(h->reg_key).add((uint64_t)(uintptr_t)h);
    }
    derive_params(h);
    *out = h;
    return NXS_OK;
} catch (...) { return dyn_caught(nullptr, "nxs_dyn_create"); }

int nxs_dyn_destroy(nxs_dyn_handle *h) try {
    if (!h) return NXS_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    release_graph(h);
    h->in_step = false;
    flush_retired(h);
    if (h->comm && h->rccl.CommDestroy) h->rccl.CommDestroy(h->comm);
    ipc_release(h);
    free_pool(h->mesh_allocs);
    free_pool(h->state_allocs);
    free_pool(h->halo_allocs);
    free_pool(h->patch_allocs);
    release_resident(h);
    if (!h->reg_key.empty()) nxs_reg::table_for(h->reg_key).remove((uint64_t)(uintptr_t)h);
    free_pool(h->pair_allocs);
    h->pair_ready = false;
    h->pair_failed = false;
    free_pool(h->sm_allocs);
    h->sm_ready = false;
    h->sm_failed = false;
    free_pool(h->ring_allocs);
    free_pool(h->hf_allocs);
    h->hf_ready = false;
    free_pool(h->forcing_allocs);
    for (auto &q : h->f_snap) q = nullptr;
    h->have_pair = false;
    unpin_all(h);
    if (h->h_send) (void)hipHostFree(h->h_send);
    if (h->h_recv) (void)hipHostFree(h->h_recv);
    if (h->d_partials) (void)hipFree(h->d_partials);
    if (h->d_regrid) (void)hipFree(h->d_regrid);
    if (h->d_crash) (void)hipFree(h->d_crash);
    if (h->d_dp) (void)hipFree(h->d_dp);
    for (auto &set : h->ev) for (auto &ev : set) if (ev) (void)hipEventDestroy(ev);
    for (auto &set : h->ev_flush) for (auto &ev : set) if (ev) (void)hipEventDestroy(ev);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_destroy"); }

int nxs_dyn_set_params(nxs_dyn_handle *h, const nxs_dyn_params *p) try {
    if (!h) return NXS_ERR_INVALID;
    int rc = check_params(h, p);
    if (rc) return rc;
    flush_retired(h);
    if (h->sig_loc) {  // the records' damage slot belongs to the OLD dynamics type
        HIPCHK(h, hipSetDevice(h->device));
        ensure_arrays(h);
    }
    h->params = *p;
    derive_params(h);
    h->res_failed = false;  // (what the resident loop was refused for may have been the old parameters; its tables are re-checked at the next step)
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_set_params"); }

int nxs_dyn_set_option(nxs_dyn_handle *h, const char *key, int64_t value) try {
    if (!h || !key) return NXS_ERR_INVALID;
    if (std::strcmp(key, "timing_reset") && std::strcmp(key, "timing")) flush_retired(h);   // (not from the two a caller may use between steps of a timed run)
    if (!std::strcmp(key, "graph")) { h->use_graph = value != 0; release_graph(h); return NXS_OK; }
    if (!std::strcmp(key, "timing")) { h->timing_enabled = value != 0; return NXS_OK; }
    if (!std::strcmp(key, "um_ring")) {
        if (value < 0 || value > NXS_MAX_RING - 1) return fail(h, NXS_ERR_INVALID, "um_ring must be in [0,%d]", NXS_MAX_RING - 1);
        h->um_ring = (int)value; release_graph(h);
        return NXS_OK;
    }
    if (!std::strcmp(key, "nt_mask")) { h->nt_mask = (int)value; release_graph(h); return NXS_OK; }  // -1 = automatic
    if (!std::strcmp(key, "ipc_delay")) {   // test door (include/nxs_dyn.h, NXS_DELAY_*): the kernels read it from IpcDev, by value in captured graphs and from the device copy of HaloFused
        if (value < 0 || value > 0x7fffffffll || ((value >> 8) & 0xff) >= NXS_DELAY_POINTS) return fail(h, NXS_ERR_INVALID, "ipc_delay: rank << 16 | point << 8 | units, point < %d", NXS_DELAY_POINTS);
        HIPCHK(h, hipSetDevice(h->device));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->ipc_delay_opt = (unsigned)value;
        h->ipc.delay = ipc_delay_word(h);
        h->d_hf_dirty = true; release_graph(h);
        return NXS_OK;
    }
    if (!std::strcmp(key, "resident_release")) { h->hf.no_release = value == 0 ? 1 : 0; h->res_no_release = h->hf.no_release; h->d_hf_dirty = true; release_graph(h); return NXS_OK; }
    if (!std::strcmp(key, "smooth_persist")) { h->smooth_persist = (int)value; release_graph(h); HIPCHK(h, hipSetDevice(h->device)); register_smoother_grid(h); return NXS_OK; }
    if (!std::strcmp(key, "halo_one_directional")) { h->one_directional = value != 0; return NXS_OK; }   // test door: before nxs_dyn_set_halo
    if (!std::strcmp(key, "fused")) {
        if (value < 0 || value > 4) return fail(h, NXS_ERR_INVALID, "fused must be 0, 1, 2, 3 or 4");
        const bool was_resident = h->fused == 4, now_resident = value == 4;
        h->fused = (int)value; h->res_failed = false; h->no_big_cut = false; release_graph(h);
        if (h->pair_claim) { resident_registry_release(h, nxs_reg::KIND_PAIR); h->pair_claim = false; h->pair_ready = false; }   // (the several-rank pair patches: cut again when they are wanted)
        // the resident loop has a cut of its own (one round of workgroups; one LARGE patch per CU for partitions of 200 k - 400 k triangles, which is not what
        // the one-launch-per-sub-step kernel wants): asking for it or giving it up on a live mesh cuts the mesh again -- any cut gives the same bits
        if (h->have_mesh && was_resident != now_resident) {
            HIPCHK(h, hipSetDevice(h->device));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            return upload_patches(h);
        }
        return NXS_OK;
    }
    if (!std::strcmp(key, "resident_dryrun")) {  // builds the tables of the resident loop now (mesh and, on several ranks, halo lists set; no
        // transport needed) and says whether this partition can run it: checks a rank's partition without its neighbours
        if (!h->have_mesh || (multi_rank(h) && !h->have_halo)) return fail(h, NXS_ERR_STATE, "resident_dryrun needs set_mesh (and set_halo)");
        HIPCHK(h, hipSetDevice(h->device));
        h->res_ready = false; h->res_failed = false;
        int rc = NXS_OK;
        if (multi_rank(h) && !h->hf_ready && (rc = build_halo_fused(h))) return rc;
        if ((rc = build_resident(h))) return rc;
        if (!h->res_ready) return fail(h, NXS_ERR_INVALID, "the resident sub-step loop cannot run this partition (NXS_DEBUG_PATCHES=1 says why)");
        return NXS_OK;
    }
    if (!std::strcmp(key, "prepare")) {
        // Everything the first step would otherwise build lazily and that FREES device memory while doing so (the patch arrays re-uploaded boundary
        // first for the exchange inside the kernels, the resident loop's tables): a hipFree synchronises the whole device, and where several ranks
        // of one process share a device (tests, rehearsals) a rank that is already spinning for its neighbour's first exchange would keep that
        // neighbour's hipFree -- and with it the exchange -- from ever happening.  Hosts call it after set_halo / the transport set-up, before a barrier.
        if (!h->have_mesh) return fail(h, NXS_ERR_STATE, "prepare needs set_mesh");
        HIPCHK(h, hipSetDevice(h->device));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const bool mr = multi_rank(h);
        const bool device_halo = mr && h->have_halo && h->ipc_ready && !h->halo_fn;
        int rc = NXS_OK;
        if (device_halo && h->halo_fused && !h->hf_ready && (rc = build_halo_fused(h))) return rc;
        if (h->fused == 4 && (!mr || (device_halo && h->halo_fused)) && !h->res_ready && !h->res_failed && h->dp.dynamics_type != NXS_DYN_MEVP) {
            if ((rc = build_resident(h))) return rc;
            if (h->res_failed && h->cut_big && !h->no_big_cut) {
                release_graph(h);
                h->no_big_cut = true;
                if ((rc = upload_patches(h))) return rc;
                if (device_halo && h->halo_fused && !h->hf_ready && (rc = build_halo_fused(h))) return rc;
                h->res_failed = true;   // (after the rebuilds: they give a fresh cut a fresh chance, this cut was made because the resident loop cannot run)
            }
        }
        return NXS_OK;
    }
    if (!std::strcmp(key, "resident_wide")) {
        h->res_wide = value != 0; h->res_ready = false; h->res_failed = false; release_graph(h); return NXS_OK;
    }
    if (!std::strcmp(key, "resident_overlap")) {
        h->res_overlap = value < 0 ? -1 : (value != 0); h->res_ready = false; h->res_failed = false; release_graph(h); return NXS_OK;
    }
    if (!std::strcmp(key, "smooth_depth")) {
        if (value != 0 && value != 5 && value != 10 && value != 25) return fail(h, NXS_ERR_INVALID, "smooth_depth must be 0 (auto), 5, 10 or 25");
        h->sm_depth = (int)value; h->sm_ready = false; h->sm_failed = false; h->tail_graph_valid = false; return NXS_OK;
    }
    if (!std::strcmp(key, "substeps_per_launch")) {
        if (value != 0 && (value < 2 || value > NXS_MAX_DEPTH)) return fail(h, NXS_ERR_INVALID, "substeps_per_launch must be 0 (auto) or in [2,%d]", NXS_MAX_DEPTH);
        h->pair_depth = (int)value; h->pair_failed = false; release_graph(h); return NXS_OK;
    }
    if (!std::strcmp(key, "pair_nodes")) {
        if (value != 0 && (value < 16 || value > 1024)) return fail(h, NXS_ERR_INVALID, "pair_nodes must be 0 (auto) or in [16,1024]");
        h->pair_nodes = (int)value; h->pair_ready = false; h->pair_failed = false; release_graph(h); return NXS_OK;
    }
    if (!std::strcmp(key, "work_arrays")) { h->work_arrays = value != 0; release_graph(h); return NXS_OK; }
    if (!std::strcmp(key, "band_patch_nodes")) {
        if (value > 0 && (value < 16 || value > 512)) return fail(h, NXS_ERR_INVALID, "band_patch_nodes must be -1 (automatic), 0 (off) or in [16,512]");
        h->band_nodes = value < 0 ? -1 : (int)value;
        if (h->have_mesh) { HIPCHK(h, hipSetDevice(h->device)); HIPCHK(h, hipStreamSynchronize(h->stream)); release_graph(h); return upload_patches(h); }
        return NXS_OK;
    }
    if (!std::strcmp(key, "pair_threads")) {
        if (value != 256 && value != 512) return fail(h, NXS_ERR_INVALID, "pair_threads must be 256 or 512");
        h->pair_T = (int)value; h->pair_ready = false; h->pair_failed = false; h->pair_hint = 0; release_graph(h); return NXS_OK;
    }
    if (!std::strcmp(key, "pair_hilbert")) { h->pair_hilbert = value != 0; h->pair_ready = false; h->pair_failed = false; h->pair_hint = 0; release_graph(h); return NXS_OK; }
    if (!std::strcmp(key, "pair_move")) { h->pair_move = value < 0 ? -1 : (value != 0); release_graph(h); return NXS_OK; }
    if (!std::strcmp(key, "pair_flow")) { h->pair_flow = value < 0 ? -1 : (value != 0); h->pair_ready = false; h->pair_failed = false; h->flow_failed = false; release_graph(h); return NXS_OK; }
    if (!std::strcmp(key, "pair_regs")) { h->pair_regs = value < 0 ? -1 : (value != 0); h->pair_ready = false; h->pair_failed = false; release_graph(h); return NXS_OK; }
    if (!std::strcmp(key, "prep_fused")) { h->prep_fused = value < 0 ? -1 : (value != 0); release_graph(h); return NXS_OK; }
    if (!std::strcmp(key, "shape_mem")) {
        if (value < -1 || value > 1) return fail(h, NXS_ERR_INVALID, "shape_mem must be -1 (auto), 0 or 1");
        h->shape_mem = (int)value; release_graph(h);
        return NXS_OK;
    }
    if (!std::strcmp(key, "trace_branches")) {  // 1 = start (or restart) the trace: zeroed records; the step then runs the per-loop kernels
        if (value && !h->have_mesh) return fail(h, NXS_ERR_STATE, "trace_branches before set_mesh");
        h->trace_branches = value != 0;
        release_graph(h);
        if (h->trace_branches) {
            HIPCHK(h, hipSetDevice(h->device));
            if (!h->dw.trace) { int rc = dev_alloc(h, h->state_allocs, &h->dw.trace, 4 * (size_t)h->dm.Ne); if (rc) return rc; }
            HIPCHK(h, hipMemsetAsync(h->dw.trace, 0, 4 * (size_t)h->dm.Ne * sizeof(unsigned long long), h->stream));
        }
        return NXS_OK;
    }
    if (!std::strcmp(key, "ipc_pad")) {   // before nxs_dyn_ipc_export: room for at least this many received nodes (dynamics.ipc_loopback: a rank's send segments stored into its own mailbox)
        if (value < 0 || value > (1 << 28)) return fail(h, NXS_ERR_INVALID, "ipc_pad out of range");
        h->ipc_pad = (int)value; return NXS_OK;
    }
    if (!std::strcmp(key, "pin_host")) { h->pin_host = value != 0; if (!h->pin_host) unpin_all(h); return NXS_OK; }
    if (!std::strcmp(key, "halo_fused")) { h->halo_fused = value != 0; release_graph(h); return NXS_OK; }
    if (!std::strcmp(key, "patch_nodes")) {
        if (value != 0 && (value < 64 || value > 1024)) return fail(h, NXS_ERR_INVALID, "patch_nodes must be 0 (auto) or in [64,1024]");
        h->patch_nodes = (int)value;
        release_graph(h);
        if (h->have_mesh) { HIPCHK(h, hipSetDevice(h->device)); HIPCHK(h, hipStreamSynchronize(h->stream)); return upload_patches(h); }
        return NXS_OK;
    }
    if (!std::strcmp(key, "timing_reset")) {  // drop what was accumulated so far (e.g. after warm-up)
        for (int k = 0; k < nxs_dyn_handle::NSETS; ++k) { int rc = harvest(h, k); if (rc) return rc; }
        for (double &x : h->sum_ms) x = 0.;
        h->sum_steps = 0;
        h->sum_flush_ms = 0.;
        h->step_ms.clear();
        return NXS_OK;
    }
    return fail(h, NXS_ERR_INVALID, "unknown option '%s'", key);
} catch (...) { return dyn_caught(h, "nxs_dyn_set_option"); }

// ------------------------------------------------------------------------------------------------
int nxs_dyn_set_mesh(nxs_dyn_handle *h, const nxs_dyn_mesh *m) try {
    if (!h || !m) return NXS_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    const int Nn = m->num_nodes, Ne = m->num_elements, No = m->local_ndof, Neo = m->local_nelements;
    if (Nn <= 0 || Ne <= 0 || No < 0 || No > Nn || Neo < 0 || Neo > Ne)
        return fail(h, NXS_ERR_INVALID, "mesh sizes invalid: Nn=%d Ne=%d No=%d Neo=%d", Nn, Ne, No, Neo);
    if (Ne >= (1 << 28)) return fail(h, NXS_ERR_INVALID, "too many elements for the packed fan entries");
    if (!m->indices || !m->ghost_nodes || !m->coord_x || !m->coord_y || !m->lat || !m->mask_dirichlet)
        return fail(h, NXS_ERR_INVALID, "mesh has NULL arrays");
    if (m->num_neumann_flags < 0 || (m->num_neumann_flags > 0 && !m->neumann_flags))
        return fail(h, NXS_ERR_INVALID, "neumann_flags invalid");
    for (int64_t i = 0; i < 3ll * Ne; ++i)
        if (m->indices[i] < 1 || m->indices[i] > Nn)
            return fail(h, NXS_ERR_INVALID, "indices[%lld]=%d outside [1,%d] (1-based local ids expected)", (long long)i, m->indices[i], Nn);
    for (int i = 0; i < m->num_neumann_flags; ++i) {
        if (m->neumann_flags[i] < 0 || m->neumann_flags[i] >= Nn) return fail(h, NXS_ERR_INVALID, "neumann_flags[%d] out of range", i);
        if (i > 0 && m->neumann_flags[i] <= m->neumann_flags[i - 1]) return fail(h, NXS_ERR_INVALID, "neumann_flags must be sorted and unique (FE.cpp:251-252)");
    }

    if (h->stream) HIPCHK(h, hipStreamSynchronize(h->stream));
    release_graph(h);
    flush_retired(h);
    free_pool(h->mesh_allocs);
    free_pool(h->state_allocs);
    free_pool(h->halo_allocs);
    free_pool(h->patch_allocs);
    release_resident(h);
    if (h->pair_claim) { resident_registry_release(h, nxs_reg::KIND_PAIR); h->pair_claim = false; }
    free_pool(h->pair_allocs);
    h->pair_ready = false;
    h->pair_failed = false;
    free_pool(h->sm_allocs);
    h->sm_ready = false;
    h->sm_failed = false;
    free_pool(h->ring_allocs);
    free_pool(h->hf_allocs);
    h->hf_ready = false;
    free_pool(h->forcing_allocs);
    for (auto &q : h->f_snap) q = nullptr;
    h->have_pair = false;
    unpin_all(h);
    h->ring = VTRing{};
    h->have_mesh = h->have_state = h->have_forcing = h->have_halo = false;
    h->no_big_cut = false;
    h->sig_loc = 0;
    h->trace_branches = 0;
    h->rank = 0; h->nranks = 1;
    h->send_procs.clear(); h->recv_procs.clear(); h->send_offsets.assign(1, 0); h->recv_offsets.assign(1, 0);

    DevMesh &d = h->dm;
    d = DevMesh{};
    d.Nn = Nn; d.Ne = Ne; d.No = No; d.Neo = Neo;
    int rc;
    // triangles, SoA, 0-based
    std::vector<int> *t = h->h_t;
    for (int k = 0; k < 3; ++k) { t[k].resize(Ne); for (int e = 0; e < Ne; ++e) t[k][e] = m->indices[3 * e + k] - 1; }
    h->h_ghost.assign(m->ghost_nodes, m->ghost_nodes + 3 * (size_t)Ne);
    h->h_x0.assign(m->coord_x, m->coord_x + Nn);
    h->h_y0.assign(m->coord_y, m->coord_y + Nn);
    if ((rc = dev_upload(h, h->mesh_allocs, &d.t0, t[0]))) return rc;
    if ((rc = dev_upload(h, h->mesh_allocs, &d.t1, t[1]))) return rc;
    if ((rc = dev_upload(h, h->mesh_allocs, &d.t2, t[2]))) return rc;
    // node flags
    std::vector<unsigned char> nf(Nn, 0);
    for (int n = 0; n < Nn; ++n) if (m->mask_dirichlet[n]) nf[n] |= NF_DIRICHLET;
    for (int n = 0; n < Nn; ++n) if (std::signbit(m->lat[n])) nf[n] |= NF_LAT_NEG;
    for (int i = 0; i < m->num_neumann_flags; ++i) nf[m->neumann_flags[i]] |= NF_NEUMANN;
    if ((rc = dev_upload(h, h->mesh_allocs, &d.nflags, nf))) return rc;
    // element flags
    std::vector<unsigned char> ef(Ne, 0);
    for (int e = 0; e < Ne; ++e) {
        for (int k = 0; k < 3; ++k) {
            if (m->ghost_nodes[3 * e + k]) ef[e] |= (1 << k);
            if (nf[t[k][e]] & NF_NEUMANN) ef[e] |= EF_ON_NEUMANN;
        }
    }
    if ((rc = dev_upload(h, h->mesh_allocs, &d.eflags, ef))) return rc;
    // coordinates
    {
        std::vector<double> tmp(m->coord_x, m->coord_x + Nn);
        if ((rc = dev_upload(h, h->mesh_allocs, &d.x0, tmp))) return rc;
        tmp.assign(m->coord_y, m->coord_y + Nn);
        if ((rc = dev_upload(h, h->mesh_allocs, &d.y0, tmp))) return rc;
        tmp.assign(m->lat, m->lat + Nn);
        if ((rc = dev_upload(h, h->mesh_allocs, &d.lat, tmp))) return rc;
    }
    // ascending element fan of every node (ELL, slot-major)
    {
        std::vector<int> deg(Nn, 0);
        for (int e = 0; e < Ne; ++e) for (int k = 0; k < 3; ++k) deg[t[k][e]]++;
        int W = 0;
        for (int n = 0; n < Nn; ++n) W = std::max(W, deg[n]);
        std::vector<int> fan((size_t)W * Nn, -1), fill(Nn, 0);
        for (int e = 0; e < Ne; ++e)  // ascending e => ascending rows
            for (int k = 0; k < 3; ++k) {
                const int n = t[k][e];
                fan[(size_t)(fill[n]++) * Nn + n] = (e << 3) | (m->ghost_nodes[3 * e + k] ? 4 : 0) | k;
            }
        d.W = W;
        if ((rc = dev_upload(h, h->mesh_allocs, &d.fan, fan))) return rc;
    }
    // bamg tables (given, or built with identical ordering)
    {
        std::vector<double> nec_own, nc_own;
        const double *nec = m->nodal_element_connectivity, *nc = m->nodal_connectivity;
        int w1 = m->nec_width, w2 = m->nc_width;
        if (!nec || !nc) {
            int bw1 = 0, bw2 = 0;
            if (nxs_mesh_connectivity(m->indices, Nn, Ne, &bw1, nullptr, &bw2, nullptr)) return fail(h, NXS_ERR_INVALID, "connectivity build failed");
            nec_own.resize((size_t)bw1 * Nn); nc_own.resize((size_t)bw2 * Nn);
            nxs_mesh_connectivity(m->indices, Nn, Ne, &bw1, nec_own.data(), &bw2, nc_own.data());
            if (!nec) { nec = nec_own.data(); w1 = bw1; }
            if (!nc) { nc = nc_own.data(); w2 = bw2; }
        }
        if (w1 <= 0 || w2 <= 1) return fail(h, NXS_ERR_INVALID, "connectivity widths invalid (%d, %d)", w1, w2);
        std::vector<int> n2e((size_t)w1 * Nn, -1);
        for (int n = 0; n < Nn; ++n)
            for (int j = 0; j < w1; ++j) {
                const double v = nec[(size_t)n * w1 + j];
                if (std::isnan(v)) continue;  // Q2: NaN pad -> skipped
                const int e = (int)(v - 1);
                if (e < 0) continue;
                if (e >= Ne) return fail(h, NXS_ERR_INVALID, "NodalElementConnectivity[%d][%d]=%g beyond %d elements", n, j, v, Ne);
                n2e[(size_t)j * Nn + n] = e;
            }
        std::vector<int> n2n((size_t)(w2 - 1) * Nn, 0), cnt(Nn, 0);
        for (int n = 0; n < Nn; ++n) {
            const int c = (int)nc[(size_t)n * w2 + (w2 - 1)];
            if (c < 0 || c > w2 - 1) return fail(h, NXS_ERR_INVALID, "NodalConnectivity count of node %d invalid (%d)", n, c);
            cnt[n] = c;
            for (int j = 0; j < c; ++j) {
                const int nb = (int)(nc[(size_t)n * w2 + j] - 1);
                if (nb < 0 || nb >= Nn) return fail(h, NXS_ERR_INVALID, "NodalConnectivity[%d][%d] out of range", n, j);
                n2n[(size_t)j * Nn + n] = nb;
            }
        }
        d.W1 = w1; d.W2 = w2 - 1;
        if ((rc = dev_upload(h, h->mesh_allocs, &d.n2e, n2e))) return rc;
        if ((rc = dev_upload(h, h->mesh_allocs, &d.n2n, n2n))) return rc;
        if ((rc = dev_upload(h, h->mesh_allocs, &d.n2n_cnt, cnt))) return rc;
        h->h_n2n = std::move(n2n); h->h_n2n_cnt = std::move(cnt); h->h_n2e = std::move(n2e);
    }

    // state + work arrays
    DevState &s = h->ds;
    DevWork &w = h->dw;
    s = DevState{}; w = DevWork{};
    h->d_icediag = nullptr; h->d_icediag_soa = nullptr;
    auto &P = h->state_allocs;
    const size_t n2 = 2 * (size_t)Nn, ne = Ne;
#define A(ptr, cnt) if ((rc = dev_alloc(h, P, &(ptr), (cnt)))) return rc
    A(s.VT, n2); A(s.VT2, n2); A(s.UM, n2); A(s.UT, n2);
    A(s.conc, ne); A(s.thick, ne); A(s.snow, ne); A(s.damage, ne); A(s.ridge, ne);
    A(s.s0, ne); A(s.s1, ne); A(s.s2, ne);
    A(s.S4a, 4 * ne); A(s.S4b, 4 * ne);
    A(s.cyoung, ne); A(s.hyoung, ne); A(s.hsyoung, ne); A(s.cmyi, ne); A(s.tmyi, ne);
    A(s.cohesion, ne); A(s.theal, ne); A(s.drag_ui, ne); A(s.drag_ui_young, ne);
    A(s.wind, n2); A(s.ocean, n2); A(s.ssh, (size_t)Nn); A(s.depth, ne);
    A(w.delta_x, ne); A(w.surface, ne); A(w.shape, 6 * ne); A(w.emass, ne); A(w.ecbu, ne); A(w.prec, 8 * ne); A(w.dragsurf, ne);
    A(w.expC, ne); A(w.pmax, ne); A(w.heal, ne); A(w.dxs, ne); A(w.volume, ne); A(w.eskip, ne); A(w.dxi, ne); A(w.open_blk, (size_t)nblocks(Nn)); A(w.shape_range, 1); A(w.erec, 6 * ne); A(w.nrec, 10 * (size_t)Nn);
    A(w.force, 6 * ne);
    A(h->d_srec, 6 * ne);
    w.srec = nullptr;  // (set per step: only the several-sub-steps kernel reads the records)
    A(w.rlmass, (size_t)Nn); A(w.node_mass, (size_t)Nn); A(w.C_bu, (size_t)Nn); A(w.grad_ssh, n2);
    A(w.fcor, (size_t)Nn); A(w.VTM, n2); A(w.xy, n2); A(w.D_tau_a, n2); A(w.D_tau_w, n2); A(w.D_del, ne);
#undef A
    HIPCHK(h, hipMemsetAsync(w.surface, 0, ne * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(w.delta_x, 0, ne * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(w.D_tau_a, 0, n2 * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(w.D_tau_w, 0, n2 * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(w.D_del, 0, ne * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(w.shape_range, 0, sizeof(int), h->stream));
    HIPCHK(h, hipMemsetAsync(w.open_blk, 0, (size_t)nblocks(Nn), h->stream));

    if (h->d_partials) { (void)hipFree(h->d_partials); h->d_partials = nullptr; }
    h->n_partials = std::min(nblocks(Ne), 1024);
    HIPCHK(h, hipMalloc((void **)&h->d_partials, sizeof(RegridPartial) * h->n_partials));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if ((rc = upload_patches(h))) return rc;
    h->have_mesh = true;
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_set_mesh"); }

// ------------------------------------------------------------------------------------------------
int nxs_dyn_set_halo(nxs_dyn_handle *h, const nxs_dyn_halo *halo) try {
    if (!h || !halo) return NXS_ERR_INVALID;
    if (!h->have_mesh) return fail(h, NXS_ERR_STATE, "set_halo before set_mesh");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    release_graph(h);
    free_pool(h->halo_allocs);
    ipc_release(h);
    h->have_halo = false;
    h->h_sent.clear();
    const int Nn = h->dm.Nn, No = h->dm.No;
    if (halo->nranks < 1 || halo->rank < 0 || halo->rank >= halo->nranks) return fail(h, NXS_ERR_INVALID, "rank/nranks invalid");
    const int ns = halo->num_send_procs, nr = halo->num_recv_procs;
    if (ns < 0 || nr < 0) return fail(h, NXS_ERR_INVALID, "negative neighbour count");
    if ((ns > 0 && (!halo->send_procs || !halo->send_offsets)) || (nr > 0 && (!halo->recv_procs || !halo->recv_offsets)))
        return fail(h, NXS_ERR_INVALID, "halo lists are NULL");
    for (int side = 0; side < 2; ++side) {  // offsets: start at 0, never decrease (the segment loops and every buffer size rely on it)
        const int32_t *off = side ? halo->recv_offsets : halo->send_offsets;
        const int n = side ? nr : ns;
        if (n == 0) continue;
        if (off[0] != 0) return fail(h, NXS_ERR_INVALID, "%s_offsets[0] = %d, expected 0", side ? "recv" : "send", off[0]);
        for (int k = 0; k < n; ++k)
            if (off[k + 1] < off[k]) return fail(h, NXS_ERR_INVALID, "%s_offsets decrease at %d (%d -> %d)", side ? "recv" : "send", k, off[k], off[k + 1]);
        if (off[n] > 0 && !(side ? halo->recv_index : halo->send_index)) return fail(h, NXS_ERR_INVALID, "%s_index is NULL", side ? "recv" : "send");
    }
    h->rank = halo->rank; h->nranks = halo->nranks;
    if (h->pair_claim) { resident_registry_release(h, nxs_reg::KIND_PAIR); h->pair_claim = false; }
    h->pair_ready = false; h->pair_failed = false;   // (the several-rank pair patches depend on the send lists)
    h->send_procs.assign(halo->send_procs, halo->send_procs + ns);
    h->recv_procs.assign(halo->recv_procs, halo->recv_procs + nr);
    if (ns > 0) h->send_offsets.assign(halo->send_offsets, halo->send_offsets + ns + 1); else h->send_offsets.assign(1, 0);
    if (nr > 0) h->recv_offsets.assign(halo->recv_offsets, halo->recv_offsets + nr + 1); else h->recv_offsets.assign(1, 0);
    h->ns_caller = ns; h->nr_caller = nr;
    const int ts = h->send_offsets[ns], tr = h->recv_offsets[nr];
    {   // NEIGHBOURS IN BOTH DIRECTIONS (include/nxs_dyn.h): a direction that carries no node is added as an empty segment BEHIND the caller's neighbours
        std::vector<int> sp(h->send_procs), so(h->send_offsets), rp(h->recv_procs), ro(h->recv_offsets);
        const std::string bad = nxs_cut::pad_halo_directions(sp, so, rp, ro);   // (also refuses a rank named twice)
        if (!bad.empty()) return fail(h, NXS_ERR_INVALID, "%s", bad.c_str());
        if (!h->one_directional) { h->send_procs = sp; h->send_offsets = so; h->recv_procs = rp; h->recv_offsets = ro; }
    }
    std::vector<int> sidx(halo->send_index, halo->send_index + ts), ridx(halo->recv_index, halo->recv_index + tr);
    h->h_send_index = sidx; h->h_recv_index = ridx;
    h->hf_ready = false;
    std::vector<int> sseg(ts), rseg(tr);
    for (int k = 0; k < ns; ++k) {
        if (h->send_procs[k] < 0 || h->send_procs[k] >= halo->nranks || h->send_procs[k] == halo->rank) return fail(h, NXS_ERR_INVALID, "send_procs[%d] invalid", k);
        for (int j = h->send_offsets[k]; j < h->send_offsets[k + 1]; ++j) {
            if (sidx[j] < 0 || sidx[j] >= No) return fail(h, NXS_ERR_INVALID, "send_index[%d]=%d is not an owned node", j, sidx[j]);
            sseg[j] = k;
        }
    }
    std::vector<char> seen(Nn, 0);
    for (int k = 0; k < nr; ++k) {
        if (h->recv_procs[k] < 0 || h->recv_procs[k] >= halo->nranks || h->recv_procs[k] == halo->rank) return fail(h, NXS_ERR_INVALID, "recv_procs[%d] invalid", k);
        for (int j = h->recv_offsets[k]; j < h->recv_offsets[k + 1]; ++j) {
            if (ridx[j] < No || ridx[j] >= Nn) return fail(h, NXS_ERR_INVALID, "recv_index[%d]=%d is not a ghost node", j, ridx[j]);
            if (seen[ridx[j]]) return fail(h, NXS_ERR_INVALID, "ghost node %d received twice", ridx[j]);
            seen[ridx[j]] = 1;
            rseg[j] = k;
        }
    }
    if (tr != Nn - No) return fail(h, NXS_ERR_INVALID, "recv lists cover %d of %d ghost nodes", tr, Nn - No);
    int rc;
    const int *cp;
    if ((rc = dev_upload(h, h->halo_allocs, &cp, sidx))) return rc; h->d_send_index = const_cast<int *>(cp);
    if ((rc = dev_upload(h, h->halo_allocs, &cp, sseg))) return rc; h->d_send_seg = const_cast<int *>(cp);
    if ((rc = dev_upload(h, h->halo_allocs, &cp, h->send_offsets))) return rc; h->d_send_off = const_cast<int *>(cp);
    if ((rc = dev_upload(h, h->halo_allocs, &cp, ridx))) return rc; h->d_recv_index = const_cast<int *>(cp);
    if ((rc = dev_upload(h, h->halo_allocs, &cp, rseg))) return rc; h->d_recv_seg = const_cast<int *>(cp);
    if ((rc = dev_upload(h, h->halo_allocs, &cp, h->recv_offsets))) return rc; h->d_recv_off = const_cast<int *>(cp);
    if ((rc = dev_upload(h, h->halo_allocs, &cp, h->recv_procs))) return rc; h->d_recv_procs = const_cast<int *>(cp);
    if ((rc = dev_alloc(h, h->halo_allocs, &h->d_send_buf, 2 * (size_t)ts))) return rc;
    if ((rc = dev_alloc(h, h->halo_allocs, &h->d_recv_buf, 2 * (size_t)tr))) return rc;
    if (h->h_send) { (void)hipHostFree(h->h_send); h->h_send = nullptr; }
    if (h->h_recv) { (void)hipHostFree(h->h_recv); h->h_recv = nullptr; }
    HIPCHK(h, hipHostMalloc((void **)&h->h_send, std::max<size_t>(2 * (size_t)ts, 1) * sizeof(double), hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc((void **)&h->h_recv, std::max<size_t>(2 * (size_t)tr, 1) * sizeof(double), hipHostMallocDefault));
    h->have_halo = true;
    h->h_sent.assign((size_t)std::max(No, 1), 0);
    for (int n : sidx) h->h_sent[n] = 1;
    if (h->fused == 4 && halo->nranks > 1 && h->band_nodes != 0) {   // cut for the resident loop before the lists were known: again, with every sent node in the band
        if ((rc = upload_patches(h))) return rc;
    }
    register_smoother_grid(h);
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_set_halo"); }

static int load_rccl(nxs_dyn_handle *h, Rccl &r) {
    if (r.lib) return NXS_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) return fail(h, NXS_ERR_COMM, "cannot dlopen librccl: %s", dlerror());
#define SYM(field, name)                                                       \
    *(void **)(&r.field) = dlsym(r.lib, name);                                 \
    if (!r.field) return fail(h, NXS_ERR_COMM, "librccl lacks %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return NXS_OK;
}

int nxs_dyn_comm_unique_id(void *id128) try {
    if (!id128) return NXS_ERR_INVALID;
    Rccl r;
    int rc = load_rccl(nullptr, r);
    if (rc) return rc;
    int e = r.GetUniqueId(id128);
    return e == 0 ? NXS_OK : fail(nullptr, NXS_ERR_COMM, "ncclGetUniqueId: %s", r.GetErrorString(e));
} catch (...) { return dyn_caught(nullptr, "nxs_dyn_comm_unique_id"); }

int nxs_dyn_comm_init(nxs_dyn_handle *h, const void *id128, int rank, int nranks) try {
    if (!h || !id128) return NXS_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = load_rccl(h, h->rccl);
    if (rc) return rc;
    NcclId id;
    std::memcpy(id.internal, id128, 128);
    nccl_comm_init_rank_t init = (nccl_comm_init_rank_t)h->rccl.CommInitRank;
    int e = init(&h->comm, nranks, id, rank);
    if (e != 0) return fail(h, NXS_ERR_COMM, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, h->rccl.GetErrorString(e));
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_comm_init"); }

// One exchange of coded payloads through the RCCL communicator (collective when nranks > 1): a grouped ncclSend/ncclRecv of the
// rank to ITSELF, and -- when halo lists are set -- the grouped send/recv of updateGhosts with every segment carrying
// (sending rank, position).  *errors = number of wrong values received.  With nranks == 1 this runs the dlopen'ed entry points,
// the by-value ncclUniqueId, the stream use and the error mapping on one GPU.
int nxs_dyn_comm_selftest(nxs_dyn_handle *h, int32_t *errors) try {
    if (!h || !errors) return NXS_ERR_INVALID;
    if (!h->comm) return fail(h, NXS_ERR_STATE, "comm_selftest before comm_init");
    HIPCHK(h, hipSetDevice(h->device));
    const int ncclDouble = 8, N = 256;
    const int ns = h->have_halo ? (int)h->send_procs.size() : 0, nr = h->have_halo ? (int)h->recv_procs.size() : 0;
    const int ts = ns ? h->send_offsets[ns] : 0, tr = nr ? h->recv_offsets[nr] : 0;
    std::vector<double> hs(N), hr(N, -1.), hsend(2 * (size_t)ts), hrecv(2 * (size_t)tr, -1.);
    for (int i = 0; i < N; ++i) hs[i] = 1e6 * h->rank + i + 0.25;
    for (int k = 0; k < ns; ++k)
        for (int i = 2 * h->send_offsets[k]; i < 2 * h->send_offsets[k + 1]; ++i) hsend[i] = 1e6 * h->rank + (i - 2 * h->send_offsets[k]) + 0.5;
    double *ds = nullptr, *dr = nullptr;
    std::vector<void *> pool;
    int rc;
    if ((rc = dev_alloc(h, pool, &ds, N)) || (rc = dev_alloc(h, pool, &dr, N))) { free_pool(pool); return rc; }
    auto done = [&](int code) { free_pool(pool); return code; };
#define ST_CHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) return done(fail(h, NXS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e))); } while (0)
    ST_CHK(hipMemcpyAsync(ds, hs.data(), N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    ST_CHK(hipMemsetAsync(dr, 0xff, N * sizeof(double), h->stream));
    if (ts > 0) ST_CHK(hipMemcpyAsync(h->d_send_buf, hsend.data(), hsend.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (tr > 0) ST_CHK(hipMemsetAsync(h->d_recv_buf, 0xff, hrecv.size() * sizeof(double), h->stream));
    int e = h->rccl.GroupStart();
    if (e == 0) e = h->rccl.Send(ds, (size_t)N, ncclDouble, h->rank, h->comm, h->stream);
    if (e == 0) e = h->rccl.Recv(dr, (size_t)N, ncclDouble, h->rank, h->comm, h->stream);
    for (int k = 0; k < ns && e == 0; ++k)
        if (h->send_offsets[k + 1] > h->send_offsets[k])
            e = h->rccl.Send(h->d_send_buf + 2 * (size_t)h->send_offsets[k], 2 * (size_t)(h->send_offsets[k + 1] - h->send_offsets[k]), ncclDouble, h->send_procs[k], h->comm, h->stream);
    for (int k = 0; k < nr && e == 0; ++k)
        if (h->recv_offsets[k + 1] > h->recv_offsets[k])
            e = h->rccl.Recv(h->d_recv_buf + 2 * (size_t)h->recv_offsets[k], 2 * (size_t)(h->recv_offsets[k + 1] - h->recv_offsets[k]), ncclDouble, h->recv_procs[k], h->comm, h->stream);
    const int e2 = h->rccl.GroupEnd();
    if (e == 0) e = e2;
    if (e != 0) return done(fail(h, NXS_ERR_COMM, "comm_selftest send/recv: %s", h->rccl.GetErrorString(e)));
    ST_CHK(hipMemcpyAsync(hr.data(), dr, N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (tr > 0) ST_CHK(hipMemcpyAsync(hrecv.data(), h->d_recv_buf, hrecv.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    ST_CHK(hipStreamSynchronize(h->stream));
#undef ST_CHK
    int bad = 0;
    for (int i = 0; i < N; ++i) bad += hr[i] != hs[i];
    for (int k = 0; k < nr; ++k)
        for (int i = 2 * h->recv_offsets[k]; i < 2 * h->recv_offsets[k + 1]; ++i) bad += hrecv[i] != 1e6 * h->recv_procs[k] + (i - 2 * h->recv_offsets[k]) + 0.5;
    *errors = bad;
    return done(NXS_OK);
} catch (...) { return dyn_caught(h, "nxs_dyn_comm_selftest"); }

// Device-direct transport, step 1: allocate my mailbox and export it.  blob receives NXS_IPC_BLOB_BYTES.
int nxs_dyn_ipc_export(nxs_dyn_handle *h, void *blob) try {
    if (!h || !blob) return NXS_ERR_INVALID;
    if (!h->have_halo) return fail(h, NXS_ERR_STATE, "ipc_export before set_halo");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    release_graph(h);
    ipc_release(h);
    const int nr = (int)h->recv_procs.size();
    // (option "ipc_pad": a mailbox for at least that many received nodes -- the looped-back profiling set-up stores its own send segments into it)
    const size_t tr = std::max((size_t)h->recv_offsets[nr], (size_t)std::max(h->ipc_pad, 0));
    h->ipc_cap = (int)tr;
    const size_t bytes = ipc_layout(tr, nr).total * sizeof(double);
    // uncached (MTYPE_UC) device memory: neither my L2 nor a neighbour's can hold a stale copy of a mailbox line or a flag.
    // The kernels rely on that: a receiver takes no acquire after its flag wait and a sender releases once per launch.  With
    // ordinary (cached) device memory a receiver's L2 could serve a stale line, because peer stores do not pass through the home
    // GPU's L2 -- so when the runtime refuses an uncached allocation the transport is refused too, and the caller stays on RCCL
    // or its own communicator.
    {
        const hipError_t ue = hipExtMallocWithFlags(&h->ipc_block, bytes, hipDeviceMallocUncached);
        if (ue != hipSuccess) {
            (void)hipGetLastError();
            h->ipc_block = nullptr;
            return fail(h, NXS_ERR_HIP, "device-direct halo transport unavailable: uncached device memory refused (%s)", hipGetErrorString(ue));
        }
    }
    h->ipc_uncached = true;
    {
        std::lock_guard<std::mutex> lk(g_mailboxes.mu);
        g_mailboxes.boxes[h->ipc_block] = MailboxRegistry::Entry{h->device, true, 0};
    }
    if (getenv("NXS_DEBUG_PATCHES")) fprintf(stderr, "[nxs] rank %d mailbox: %zu bytes, uncached (MTYPE_UC)\n", h->rank, bytes);
    // (everything on the handle's own stream: another handle of this process may be capturing a graph on its thread right now,
    // and legacy-stream operations are refused while any blocking capture is open)
    HIPCHK(h, hipMemsetAsync(h->ipc_block, 0, bytes, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->ipc_block_bytes = bytes;
    IpcBlob b;
    std::memset(&b, 0, sizeof b);
    HIPCHK(h, hipIpcGetMemHandle(&b.mem, h->ipc_block));
    b.magic = IPC_MAGIC;
    b.pid = (long long)getpid();
    b.ptr = (unsigned long long)(uintptr_t)h->ipc_block;
    b.device = h->device; b.uncached = 1;
    b.tr = (int)tr; b.nr = nr;
    b.token = process_token();
    std::memset(blob, 0, NXS_IPC_BLOB_BYTES);
    std::memcpy(blob, &b, sizeof b);
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_ipc_export"); }

// Step 2: map the neighbours' mailboxes.  For send neighbour k (order of the handle's send_procs: the caller's neighbours, then the directions nxs_dyn_set_halo added):
// blobs + k*NXS_IPC_BLOB_BYTES is its exported blob, peer_recv_offset[k] the offset (in nodes) of MY
// segment inside its receive lists, peer_recv_total[k] its total number of received nodes and
// peer_flag_slot[k] my position in its recv_procs.
static int ipc_connect_impl(nxs_dyn_handle *h, const void *blobs, const int32_t *peer_recv_offset, const int32_t *peer_recv_total, const int32_t *peer_flag_slot) {
    const int ns = (int)h->send_procs.size(), nr = (int)h->recv_procs.size();
    HIPCHK(h, hipStreamSynchronize(h->stream));
    release_graph(h);
    ipc_disconnect(h);  // a second connect replaces the first: its peer mappings and tables go
    std::vector<double *> seg(ns), sseg(ns);
    std::vector<long long> stride(ns);
    std::vector<unsigned long long *> flag(ns), sflag(ns), sstat(ns);
    for (int k = 0; k < ns; ++k) {
        // the tables are checked against what the neighbour itself published: a kernel that trusted a wrong offset would store
        // outside the neighbour's mailbox, in another process's memory
        IpcBlob b;
        std::memcpy(&b, (const char *)blobs + (size_t)k * NXS_IPC_BLOB_BYTES, sizeof b);
        const int q = h->send_procs[k], nseg = h->send_offsets[k + 1] - h->send_offsets[k];
        if (b.magic != IPC_MAGIC) return fail(h, NXS_ERR_INVALID, "ipc_connect: blob of neighbour %d was not made by nxs_dyn_ipc_export", q);
        if (!b.uncached) return fail(h, NXS_ERR_COMM, "ipc_connect: neighbour %d's mailbox is not uncached memory", q);
        if (peer_recv_total[k] != b.tr) return fail(h, NXS_ERR_INVALID, "ipc_connect: neighbour %d receives %d nodes, the table says %d", q, b.tr, peer_recv_total[k]);
        if (peer_flag_slot[k] < 0 || peer_flag_slot[k] >= b.nr) return fail(h, NXS_ERR_INVALID, "ipc_connect: flag slot %d outside neighbour %d's %d receive neighbours", peer_flag_slot[k], q, b.nr);
        if (peer_recv_offset[k] < 0 || (long long)peer_recv_offset[k] + nseg > b.tr)
            return fail(h, NXS_ERR_INVALID, "ipc_connect: my segment [%d, %d) does not fit neighbour %d's %d received nodes", peer_recv_offset[k], peer_recv_offset[k] + nseg, q, b.tr);
        void *base = nullptr;
        if (b.pid == (long long)getpid() && b.token == process_token()) {  // a handle of this process: its pointer is valid here, on another device after peer access
            base = (void *)(uintptr_t)b.ptr;
            {
                std::lock_guard<std::mutex> lk(g_mailboxes.mu);
                auto it = g_mailboxes.boxes.find(base);
                if (it == g_mailboxes.boxes.end() || !it->second.owner_alive)
                    return fail(h, NXS_ERR_STATE, "ipc_connect: neighbour %d's mailbox no longer exists (exported again, or its handle was destroyed, since the blob was made)", q);
                it->second.peers++;
            }
            h->ipc_local_peers.push_back(base);
            if (b.device != h->device) {
                const hipError_t pe = hipDeviceEnablePeerAccess(b.device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return fail(h, NXS_ERR_COMM, "hipDeviceEnablePeerAccess(%d -> %d): %s", h->device, b.device, hipGetErrorString(pe));
                (void)hipGetLastError();
            }
        } else {
            const hipError_t e = hipIpcOpenMemHandle(&base, b.mem, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) return fail(h, NXS_ERR_COMM, "hipIpcOpenMemHandle(neighbour %d): %s", q, hipGetErrorString(e));
            h->ipc_peer_base.push_back(base);
        }
        double *mb = static_cast<double *>(base);
        seg[k] = mb + 2 * (size_t)peer_recv_offset[k];
        stride[k] = 2ll * peer_recv_total[k];
        const IpcLayout pl = ipc_layout((size_t)b.tr, b.nr);
        flag[k] = reinterpret_cast<unsigned long long *>(mb + pl.flags) + peer_flag_slot[k];
        sflag[k] = reinterpret_cast<unsigned long long *>(mb + pl.sflags) + peer_flag_slot[k];
        sstat[k] = reinterpret_cast<unsigned long long *>(mb + pl.sstatic) + peer_flag_slot[k];
        sseg[k] = mb + pl.slots + 2 * (size_t)peer_recv_offset[k];
    }
    IpcDev &d = h->ipc;
    const size_t tr = (size_t)h->ipc_cap;   // the capacity the mailbox was exported with (the received nodes, or option "ipc_pad" if that is more)
    const IpcLayout ml = ipc_layout(tr, nr);
    d.mailbox = static_cast<double *>(h->ipc_block);
    d.flags = reinterpret_cast<unsigned long long *>(d.mailbox + ml.flags);
    d.sflags = reinterpret_cast<unsigned long long *>(d.mailbox + ml.sflags);
    d.sstatic = reinterpret_cast<unsigned long long *>(d.mailbox + ml.sstatic);
    d.smb = d.mailbox + ml.slots;
    d.tr = (int)tr; d.ns = ns; d.nr = nr;
    d.delay = ipc_delay_word(h);
    int rc;
    unsigned long long *ctr = nullptr;
    if ((rc = dev_alloc(h, h->ipc_allocs, &ctr, 8))) return rc;
    HIPCHK(h, hipMemsetAsync(ctr, 0, 8 * sizeof(unsigned long long), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    d.seq_push = ctr; d.seq_pull = ctr + 1;
    d.done_push = reinterpret_cast<unsigned int *>(ctr + 2); d.done_pull = reinterpret_cast<unsigned int *>(ctr + 3);
    d.error = reinterpret_cast<int *>(ctr + 4);
    double *const *dseg; const long long *dstr; unsigned long long *const *dfl;
    { const double **tmp; std::vector<const double *> v(seg.begin(), seg.end()); if ((rc = dev_upload(h, h->ipc_allocs, (const double *const **)&tmp, v))) return rc; dseg = (double *const *)tmp; }
    if ((rc = dev_upload(h, h->ipc_allocs, &dstr, stride))) return rc;
    { const unsigned long long **tmp; std::vector<const unsigned long long *> v(flag.begin(), flag.end()); if ((rc = dev_upload(h, h->ipc_allocs, (const unsigned long long *const **)&tmp, v))) return rc; dfl = (unsigned long long *const *)tmp; }
    d.peer_seg = dseg; d.peer_parity_stride = dstr; d.peer_flag = dfl;
    {   // the smoother's mailbox: peer addresses, my epoch, the static words
        const double **tmp; std::vector<const double *> v(sseg.begin(), sseg.end());
        if ((rc = dev_upload(h, h->ipc_allocs, (const double *const **)&tmp, v))) return rc;
        d.peer_smb = (double *const *)tmp;
        const unsigned long long **t2; std::vector<const unsigned long long *> v2(sflag.begin(), sflag.end());
        if ((rc = dev_upload(h, h->ipc_allocs, (const unsigned long long *const **)&t2, v2))) return rc;
        d.peer_sflag = (unsigned long long *const *)t2;
        std::vector<const unsigned long long *> v3(sstat.begin(), sstat.end());
        if ((rc = dev_upload(h, h->ipc_allocs, (const unsigned long long *const **)&t2, v3))) return rc;
        d.peer_sstatic = (unsigned long long *const *)t2;
        d.epoch = ctr + 5;  // (zeroed above)
        const int *ip;
        std::vector<int> ones((size_t)std::max(ns, 1), 1), zeros((size_t)std::max(nr, 1), 0);
        if ((rc = dev_upload(h, h->ipc_allocs, &ip, ones))) return rc; d.my_static = const_cast<int *>(ip);
        if ((rc = dev_upload(h, h->ipc_allocs, &ip, zeros))) return rc; d.peer_static = const_cast<int *>(ip);
    }
    h->ipc_ready = true;
    h->d_hf_dirty = true;
    release_graph(h);
    return NXS_OK;
}

// The low-level form: tables for the CALLER's send neighbours, in the order of nxs_dyn_halo.send_procs.
int nxs_dyn_ipc_connect(nxs_dyn_handle *h, const void *blobs, const int32_t *peer_recv_offset, const int32_t *peer_recv_total,
                        const int32_t *peer_flag_slot) try {
    if (!h) return NXS_ERR_INVALID;
    if (!h->ipc_block) return fail(h, NXS_ERR_STATE, "ipc_connect before ipc_export");
    HIPCHK(h, hipSetDevice(h->device));
    const int ns = (int)h->send_procs.size(), nr = (int)h->recv_procs.size();
    if (ns > 0 && (!blobs || !peer_recv_offset || !peer_recv_total || !peer_flag_slot)) return fail(h, NXS_ERR_INVALID, "ipc_connect: NULL tables");
    if (!h->one_directional && (ns != h->ns_caller || nr != h->nr_caller)) {
        // Two buffers per link are safe only with a hand-shake: "a neighbour cannot start exchange x + 2 before it has received my exchange x + 1, which I send only
        // after my pull of exchange x" (k_halo_push) holds when every rank I send to also sends to me.  A rank that sends to q without receiving from q could run two
        // exchanges ahead of q and overwrite the half q still reads -- seen once in round 4 as a wrong payload in the self-test of a 4-rank mosaic whose rank 0 sends to
        // rank 3 and receives nothing from it (and made deterministic since: tests/test_gpu_protocol_delays.py).  nxs_dyn_set_halo has added the missing direction as
        // an empty segment; the caller's tables do not know it, the neighbours' records do.
        const int odd = ns != h->ns_caller ? h->send_procs[h->ns_caller] : h->recv_procs[h->nr_caller];
        return fail(h, NXS_ERR_INVALID, "ipc_connect: rank %d exchanges with rank %d in one direction only; the device-direct mailboxes need every link in both directions, and "
                                        "nxs_dyn_set_halo has added the missing one as an empty segment -- which these tables cannot describe: connect with "
                                        "nxs_dyn_ipc_export_record / nxs_dyn_ipc_connect_records", h->rank, odd);
    }
    return ipc_connect_impl(h, blobs, peer_recv_offset, peer_recv_total, peer_flag_slot);
} catch (...) { return dyn_caught(h, "nxs_dyn_ipc_connect"); }

// The record form (include/nxs_dyn.h): a rank's record = its blob + its receive lists as this library holds them (the added directions included), so that every
// neighbour finds its segment, the totals and its flag slot itself.
namespace {
constexpr int IPC_RECORD_MAGIC = 0x4e585231;   // 'NXR1'
struct IpcRecordHead { int magic, rank, nr, reserved; };
inline int ipc_record_size(int nr) { return NXS_IPC_BLOB_BYTES + (int)sizeof(IpcRecordHead) + 4 * nr + 4 * (nr + 1); }
}  // namespace

int nxs_dyn_ipc_record_bytes(nxs_dyn_handle *h, int32_t *bytes) try {
    if (!h || !bytes) return NXS_ERR_INVALID;
    if (!h->have_halo) return fail(h, NXS_ERR_STATE, "ipc_record_bytes before set_halo");
    *bytes = ipc_record_size((int)h->recv_procs.size());
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_ipc_record_bytes"); }

int nxs_dyn_ipc_export_record(nxs_dyn_handle *h, void *record, int32_t capacity) try {
    if (!h || !record) return NXS_ERR_INVALID;
    if (!h->have_halo) return fail(h, NXS_ERR_STATE, "ipc_export_record before set_halo");
    const int nr = (int)h->recv_procs.size(), need = ipc_record_size(nr);
    if (capacity < need) return fail(h, NXS_ERR_INVALID, "ipc_export_record: the record needs %d bytes, %d given", need, capacity);
    std::memset(record, 0, (size_t)capacity);
    const int rc = nxs_dyn_ipc_export(h, record);
    if (rc) return rc;
    char *p = static_cast<char *>(record) + NXS_IPC_BLOB_BYTES;
    const IpcRecordHead hd{IPC_RECORD_MAGIC, h->rank, nr, 0};
    std::memcpy(p, &hd, sizeof hd); p += sizeof hd;
    if (nr > 0) std::memcpy(p, h->recv_procs.data(), 4 * (size_t)nr);
    p += 4 * (size_t)nr;
    std::memcpy(p, h->recv_offsets.data(), 4 * (size_t)(nr + 1));
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_ipc_export_record"); }

int nxs_dyn_ipc_connect_records(nxs_dyn_handle *h, const void *records, int64_t stride, int32_t nranks) try {
    if (!h) return NXS_ERR_INVALID;
    if (!h->ipc_block) return fail(h, NXS_ERR_STATE, "ipc_connect_records before ipc_export_record");
    HIPCHK(h, hipSetDevice(h->device));
    const int ns = (int)h->send_procs.size();
    if (nranks != h->nranks) return fail(h, NXS_ERR_INVALID, "ipc_connect_records: %d records, the halo lists were set for %d ranks", nranks, h->nranks);
    if (!records || stride < ipc_record_size(0)) return fail(h, NXS_ERR_INVALID, "ipc_connect_records: NULL records or a stride below %d bytes", ipc_record_size(0));
    std::vector<char> blobs((size_t)std::max(ns, 1) * NXS_IPC_BLOB_BYTES);
    std::vector<int32_t> off(std::max(ns, 1)), tot(std::max(ns, 1)), slot(std::max(ns, 1));
    for (int k = 0; k < ns; ++k) {
        const int q = h->send_procs[k], nseg = h->send_offsets[k + 1] - h->send_offsets[k];
        const char *rec = static_cast<const char *>(records) + (size_t)q * (size_t)stride;
        IpcRecordHead hd;
        std::memcpy(&hd, rec + NXS_IPC_BLOB_BYTES, sizeof hd);
        if (hd.magic != IPC_RECORD_MAGIC || hd.rank != q) return fail(h, NXS_ERR_INVALID, "ipc_connect_records: record %d was not made by nxs_dyn_ipc_export_record on rank %d", q, q);
        if (hd.nr < 0 || (int64_t)ipc_record_size(hd.nr) > stride) return fail(h, NXS_ERR_INVALID, "ipc_connect_records: record %d names %d receive neighbours, more than the stride holds", q, hd.nr);
        std::vector<int32_t> lists(2 * (size_t)hd.nr + 1);   // (copied out: a caller's byte buffer owes no alignment)
        std::memcpy(lists.data(), rec + NXS_IPC_BLOB_BYTES + sizeof hd, lists.size() * sizeof(int32_t));
        const int32_t *rp = lists.data(), *ro = rp + hd.nr;
        int me = -1;
        for (int j = 0; j < hd.nr; ++j) if (rp[j] == h->rank) { me = j; break; }
        if (me < 0) return fail(h, NXS_ERR_INVALID, "ipc_connect_records: rank %d does not list rank %d among its receive neighbours (its halo lists and mine disagree)", q, h->rank);
        if (ro[me + 1] - ro[me] != nseg)
            return fail(h, NXS_ERR_INVALID, "ipc_connect_records: I send %d nodes to rank %d, which expects %d from me (its halo lists and mine disagree)", nseg, q, ro[me + 1] - ro[me]);
        std::memcpy(blobs.data() + (size_t)k * NXS_IPC_BLOB_BYTES, rec, NXS_IPC_BLOB_BYTES);
        off[k] = ro[me]; tot[k] = ro[hd.nr]; slot[k] = me;
    }
    return ipc_connect_impl(h, blobs.data(), off.data(), tot.data(), slot.data());
} catch (...) { return dyn_caught(h, "nxs_dyn_ipc_connect_records"); }

// Profiling aid (include/nxs_dyn.h): the mailboxes of this handle connected to THEMSELVES.
int nxs_dyn_ipc_loopback(nxs_dyn_handle *h) try {
    if (!h) return NXS_ERR_INVALID;
    if (!h->have_halo) return fail(h, NXS_ERR_STATE, "ipc_loopback before set_halo");
    const int ns = (int)h->send_procs.size(), nr = (int)h->recv_procs.size();
    if (ns == 0 || nr == 0 || ns < nr) return fail(h, NXS_ERR_INVALID, "ipc_loopback: this partition's halo lists cannot be looped back (%d send, %d receive neighbours)", ns, nr);
    int tr = h->recv_offsets[nr];
    for (int k = 0; k < ns; ++k) tr = std::max(tr, h->send_offsets[k + 1] - h->send_offsets[k]);   // every send segment is stored at offset 0 of the mailbox: room for the longest
    h->ipc_pad = tr;
    char blob[NXS_IPC_BLOB_BYTES];
    int rc = nxs_dyn_ipc_export(h, blob);
    if (rc) return rc;
    std::vector<char> blobs((size_t)ns * NXS_IPC_BLOB_BYTES);
    std::vector<int32_t> off(ns, 0), tot(ns, tr), slot(ns);
    for (int k = 0; k < ns; ++k) { std::memcpy(blobs.data() + (size_t)k * NXS_IPC_BLOB_BYTES, blob, NXS_IPC_BLOB_BYTES); slot[k] = k % nr; }
    return ipc_connect_impl(h, blobs.data(), off.data(), tot.data(), slot.data());
} catch (...) { return dyn_caught(h, "nxs_dyn_ipc_loopback"); }

// Step 3 (collective): `rounds` exchanges of a synthetic pattern through the mailboxes; *errors gets
// 0 when every value arrived intact and in time on this rank.
int nxs_dyn_ipc_selftest(nxs_dyn_handle *h, int rounds, int32_t *errors) try {
    if (!h || !errors) return NXS_ERR_INVALID;
    if (!h->ipc_ready) return fail(h, NXS_ERR_STATE, "ipc_selftest before ipc_connect");
    HIPCHK(h, hipSetDevice(h->device));
    const int ns = (int)h->send_procs.size(), nr = (int)h->recv_procs.size();
    const int ts = h->send_offsets[ns], tr = h->recv_offsets[nr];
    // Odd rounds publish the way the in-kernel exchange of k_substep_fused / k_smooth_halo does -- every wave drains its stores,
    // ONE release per launch, no acquire at the receiver -- even rounds the way k_halo_push does (a release per block): whichever
    // variant the step uses later has then been through every link of this rank with checked payloads.
    for (int it = 0; it < rounds; ++it) {
        hipLaunchKernelGGL(k_halo_push, dim3(nblocks(ts)), dim3(BLOCK), 0, h->stream, (const double *)nullptr, h->dm.Nn, ts,
                           h->d_send_index, h->d_send_seg, h->d_send_off, h->ipc, h->rank, 1 + (it & 1));
        hipLaunchKernelGGL(k_halo_pull, dim3(nblocks(tr)), dim3(BLOCK), 0, h->stream, (double *)nullptr, h->dm, h->ds, tr,
                           h->d_recv_index, h->d_recv_seg, h->d_recv_off, h->ipc, 0., 1, h->d_recv_procs, 0);
    }
    int err = 0;
    HIPCHK(h, hipMemcpyAsync(&err, h->ipc.error, sizeof err, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *errors = err;
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_ipc_selftest"); }

int nxs_dyn_set_halo_exchange_fn(nxs_dyn_handle *h, nxs_dyn_halo_fn fn, void *ctx) try {
    if (!h) return NXS_ERR_INVALID;
    h->halo_fn = fn;
    h->halo_ctx = ctx;
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_set_halo_exchange_fn"); }

// ------------------------------------------------------------------------------------------------
int nxs_dyn_put_state(nxs_dyn_handle *h, const nxs_dyn_state *s) try {
    if (!h || !s) return NXS_ERR_INVALID;
    if (!h->have_mesh) return fail(h, NXS_ERR_STATE, "put_state before set_mesh");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n2 = 2 * (size_t)h->dm.Nn * sizeof(double), ne = (size_t)h->dm.Ne * sizeof(double);
    DevState &d = h->ds;
    struct { double *dst; const double *src; size_t bytes; const char *name; } cp[] = {
        {d.VT, s->VT, n2, "VT"}, {d.UM, s->UM, n2, "UM"}, {d.UT, s->UT, n2, "UT"},
        {d.conc, s->conc, ne, "conc"}, {d.thick, s->thick, ne, "thick"}, {d.snow, s->snow_thick, ne, "snow_thick"},
        {d.damage, s->damage, ne, "damage"}, {d.ridge, s->ridge_ratio, ne, "ridge_ratio"},
        {d.s0, s->sigma[0], ne, "sigma[0]"}, {d.s1, s->sigma[1], ne, "sigma[1]"}, {d.s2, s->sigma[2], ne, "sigma[2]"},
        {d.cyoung, s->conc_young, ne, "conc_young"}, {d.hyoung, s->h_young, ne, "h_young"}, {d.hsyoung, s->hs_young, ne, "hs_young"},
        {d.cmyi, s->conc_myi, ne, "conc_myi"}, {d.tmyi, s->thick_myi, ne, "thick_myi"},
        {d.cohesion, s->cohesion, ne, "cohesion"}, {d.theal, s->time_relaxation_damage, ne, "time_relaxation_damage"},
        {d.drag_ui, s->drag_ui, ne, "drag_ui"}, {d.drag_ui_young, s->drag_ui_young, ne, "drag_ui_young"},
    };
    // the first put after set_mesh must bring every array; later ones may leave members NULL = "the device copy is current"
    // (a host whose thermodynamics only touched concentration and thickness uploads only those)
    if (!h->have_state)
        for (auto &c : cp) if (!c.src) return fail(h, NXS_ERR_INVALID, "put_state: %s is NULL", c.name);
    if (s->damage || s->sigma[0] || s->sigma[1] || s->sigma[2]) ensure_arrays(h);  // the members not given must not be lost
    for (auto &c : cp) if (c.src) pin_host_buffer(h, c.src, c.bytes);
    for (auto &c : cp) if (c.src) HIPCHK(h, hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_state = true;
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_put_state"); }

int nxs_dyn_get_state(nxs_dyn_handle *h, nxs_dyn_state *s) try {
    if (!h || !s) return NXS_ERR_INVALID;
    if (!h->have_state) return fail(h, NXS_ERR_STATE, "get_state before put_state");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n2 = 2 * (size_t)h->dm.Nn * sizeof(double), ne = (size_t)h->dm.Ne * sizeof(double);
    DevState &d = h->ds;
    struct { double *dst; const double *src; size_t bytes; } cp[] = {
        {s->VT, d.VT, n2}, {s->UM, d.UM, n2}, {s->UT, d.UT, n2},
        {s->conc, d.conc, ne}, {s->thick, d.thick, ne}, {s->snow_thick, d.snow, ne},
        {s->damage, d.damage, ne}, {s->ridge_ratio, d.ridge, ne},
        {s->sigma[0], d.s0, ne}, {s->sigma[1], d.s1, ne}, {s->sigma[2], d.s2, ne},
        {s->conc_young, d.cyoung, ne}, {s->h_young, d.hyoung, ne}, {s->hs_young, d.hsyoung, ne},
        {s->conc_myi, d.cmyi, ne}, {s->thick_myi, d.tmyi, ne},
    };
    if (h->res_ready || h->flow_ready) { HIPCHK(h, hipStreamSynchronize(h->stream)); int rc = resident_error(h); if (rc) return rc; }  // never a half-made step without an error
    if (s->damage || s->sigma[0] || s->sigma[1] || s->sigma[2]) ensure_arrays(h);
    for (auto &c : cp) if (c.dst) pin_host_buffer(h, c.dst, c.bytes);
    for (auto &c : cp) if (c.dst) HIPCHK(h, hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_get_state"); }

int nxs_dyn_set_forcing(nxs_dyn_handle *h, const nxs_dyn_forcing *f) try {
    if (!h || !f) return NXS_ERR_INVALID;
    if (!h->have_mesh) return fail(h, NXS_ERR_STATE, "set_forcing before set_mesh");
    if (!f->wind || !f->ocean || !f->ssh || !f->element_depth) return fail(h, NXS_ERR_INVALID, "forcing has NULL arrays");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t Nn = h->dm.Nn, Ne = h->dm.Ne;
    pin_host_buffer(h, f->wind, 2 * Nn * sizeof(double)); pin_host_buffer(h, f->ocean, 2 * Nn * sizeof(double));
    pin_host_buffer(h, f->ssh, Nn * sizeof(double)); pin_host_buffer(h, f->element_depth, Ne * sizeof(double));
    HIPCHK(h, hipMemcpyAsync(h->ds.wind, f->wind, 2 * Nn * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->ds.ocean, f->ocean, 2 * Nn * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->ds.ssh, f->ssh, Nn * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->ds.depth, f->element_depth, Ne * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_forcing = true;
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_set_forcing"); }

int nxs_dyn_set_forcing_pair(nxs_dyn_handle *h, const nxs_dyn_forcing *f0, const nxs_dyn_forcing *f1) try {
    if (!h || !f0 || !f1) return NXS_ERR_INVALID;
    if (!h->have_mesh) return fail(h, NXS_ERR_STATE, "set_forcing_pair before set_mesh");
    if (!f0->wind || !f0->ocean || !f0->ssh || !f0->element_depth || !f1->wind || !f1->ocean || !f1->ssh) return fail(h, NXS_ERR_INVALID, "forcing has NULL arrays");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t Nn = h->dm.Nn, Ne = h->dm.Ne;
    const size_t len[6] = {2 * Nn, 2 * Nn, 2 * Nn, 2 * Nn, Nn, Nn};
    const double *src[6] = {f0->wind, f1->wind, f0->ocean, f1->ocean, f0->ssh, f1->ssh};
    if (!h->f_snap[0])
        for (int k = 0; k < 6; ++k) { int rc = dev_alloc(h, h->forcing_allocs, &h->f_snap[k], len[k]); if (rc) return rc; }
    for (int k = 0; k < 6; ++k) HIPCHK(h, hipMemcpyAsync(h->f_snap[k], src[k], len[k] * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->ds.depth, f0->element_depth, Ne * sizeof(double), hipMemcpyHostToDevice, h->stream));  // a "constant" dataset
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_pair = true;
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_set_forcing_pair"); }

int nxs_dyn_set_forcing_time(nxs_dyn_handle *h, double fcoeff0, double fcoeff1, const double factor[3], const double bias[3]) try {
    if (!h) return NXS_ERR_INVALID;
    if (!h->have_pair) return fail(h, NXS_ERR_STATE, "set_forcing_time before set_forcing_pair");
    HIPCHK(h, hipSetDevice(h->device));
    ForcingBlend b{h->f_snap[0], h->f_snap[1], h->f_snap[2], h->f_snap[3], h->f_snap[4], h->f_snap[5], fcoeff0, fcoeff1, {1., 1., 1.}, {0., 0., 0.}};
    for (int k = 0; k < 3; ++k) { if (factor) b.factor[k] = factor[k]; if (bias) b.bias[k] = bias[k]; }
    hipLaunchKernelGGL(k_blend_forcing, dim3(nblocks(2 * h->dm.Nn)), dim3(BLOCK), 0, h->stream, h->dm.Nn, b, h->ds.wind, h->ds.ocean, h->ds.ssh);
    h->have_forcing = true;
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_set_forcing_time"); }

int nxs_dyn_get_diag(nxs_dyn_handle *h, nxs_dyn_diag *dg) try {
    if (!h || !dg) return NXS_ERR_INVALID;
    if (!h->have_mesh) return fail(h, NXS_ERR_STATE, "get_diag before set_mesh");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n2 = 2 * (size_t)h->dm.Nn * sizeof(double), ne = (size_t)h->dm.Ne * sizeof(double);
    if (dg->surface) HIPCHK(h, hipMemcpyAsync(dg->surface, h->dw.surface, ne, hipMemcpyDeviceToHost, h->stream));
    if (dg->delta_x) HIPCHK(h, hipMemcpyAsync(dg->delta_x, h->dw.delta_x, ne, hipMemcpyDeviceToHost, h->stream));
    if (dg->D_tau_a) HIPCHK(h, hipMemcpyAsync(dg->D_tau_a, h->dw.D_tau_a, n2, hipMemcpyDeviceToHost, h->stream));
    if (dg->D_tau_w) HIPCHK(h, hipMemcpyAsync(dg->D_tau_w, h->dw.D_tau_w, n2, hipMemcpyDeviceToHost, h->stream));
    if (dg->D_del_ci_ridge_myi) HIPCHK(h, hipMemcpyAsync(dg->D_del_ci_ridge_myi, h->dw.D_del, ne, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_get_diag"); }

int nxs_dyn_ice_diagnostics(nxs_dyn_handle *h, nxs_dyn_ice_diag *dg, const double **device_rows) try {
    if (!h) return NXS_ERR_INVALID;
    if (!h->have_mesh || !h->have_state) return fail(h, NXS_ERR_STATE, "ice_diagnostics needs set_mesh and put_state");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t Ne = (size_t)h->dm.Ne;
    if (h->res_ready || h->flow_ready) { HIPCHK(h, hipStreamSynchronize(h->stream)); int rc = resident_error(h); if (rc) return rc; }
    if (!h->d_icediag) { int rc = dev_alloc(h, h->state_allocs, &h->d_icediag, NXS_ICE_DIAG_FIELDS * Ne); if (rc) return rc; }
    hipLaunchKernelGGL(k_ice_diagnostics, dim3(nblocks(h->dm.Ne)), dim3(BLOCK), 0, h->stream, h->dm, h->ds, h->dp.young_cat ? 1 : 0,
                       h->sig_loc ? (const double *)h->ds.S4a : (const double *)nullptr, h->d_icediag);
    HIPCHK(h, hipGetLastError());
    if (dg) {   // the host's vectors from a field-major copy of the rows: six contiguous transfers (a strided copy of 8-byte rows moves one row per descriptor)
        double *dst[NXS_ICE_DIAG_FIELDS] = {dg->D_conc, dg->D_thick, dg->D_snow_thick, dg->D_sigma0, dg->D_sigma1, dg->D_divergence};
        bool any = false;
        for (double *q : dst) any = any || q;
        if (any) {
            if (!h->d_icediag_soa) { int rc = dev_alloc(h, h->state_allocs, &h->d_icediag_soa, NXS_ICE_DIAG_FIELDS * Ne); if (rc) return rc; }
            hipLaunchKernelGGL(k_icediag_soa, dim3(nblocks(h->dm.Ne)), dim3(BLOCK), 0, h->stream, h->dm.Ne, (const double *)h->d_icediag, h->d_icediag_soa);
            for (int k = 0; k < NXS_ICE_DIAG_FIELDS; ++k)
                if (dst[k]) HIPCHK(h, hipMemcpyAsync(dst[k], h->d_icediag_soa + (size_t)k * Ne, Ne * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        }
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (device_rows) *device_rows = h->d_icediag;
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_ice_diagnostics"); }

// test door: the branch trace of updateSigmaDamage (option "trace_branches"), 4 words per element
int nxs_dyn_get_branch_trace(nxs_dyn_handle *h, uint64_t *out, int64_t num_words) try {
    if (!h || !out) return NXS_ERR_INVALID;
    if (!h->have_mesh || !h->dw.trace) return fail(h, NXS_ERR_STATE, "get_branch_trace: set option trace_branches = 1 first");
    if (num_words != 4 * (int64_t)h->dm.Ne) return fail(h, NXS_ERR_INVALID, "get_branch_trace: %lld words expected", 4ll * h->dm.Ne);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(out, h->dw.trace, (size_t)num_words * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_get_branch_trace"); }

// debug / test door: copy a named work array to the host (n doubles)
int nxs_dyn_debug_array(nxs_dyn_handle *h, const char *name, double *out, int64_t n) try {
    if (!h || !name || !out) return NXS_ERR_INVALID;
    if (!h->have_mesh) return fail(h, NXS_ERR_STATE, "debug_array before set_mesh");
    HIPCHK(h, hipSetDevice(h->device));
    const int64_t Nn = h->dm.Nn, Ne = h->dm.Ne;
    struct { const char *nm; const double *p; int64_t len; } tab[] = {
        {"rlmass", h->dw.rlmass, Nn}, {"node_mass", h->dw.node_mass, Nn}, {"C_bu", h->dw.C_bu, Nn},
        {"grad_ssh", h->dw.grad_ssh, 2 * Nn}, {"fcor", h->dw.fcor, Nn}, {"VTM", h->dw.VTM, 2 * Nn},
        {"shape", h->dw.shape, 6 * Ne}, {"emass", h->dw.emass, Ne}, {"ecbu", h->dw.ecbu, Ne},
        {"force", h->dw.force, 6 * Ne}, {"volume", h->dw.volume, Ne}, {"expC", h->dw.expC, Ne},
    };
    for (auto &t : tab)
        if (!std::strcmp(t.nm, name)) {
            if (eff_fused(h) != 0 && !h->work_arrays && std::strcmp(name, "VTM") && std::strcmp(name, "node_mass"))
                return fail(h, NXS_ERR_STATE, "debug_array %s: the fused path fills records only; set option work_arrays = 1 before the step", name);
            if (n != t.len) return fail(h, NXS_ERR_INVALID, "debug_array %s has %lld entries, caller asked %lld", name, (long long)t.len, (long long)n);
            HIPCHK(h, hipMemcpyAsync(out, t.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            return NXS_OK;
        }
    {   // what the fused path does fill: the records of the sub-step kernels and the prep kernels' other outputs (explicit_solve leaves them as the
        // prep kernels wrote them; step() goes on to update(), which renews M_surface)
        struct { const char *nm; const double *p; int64_t len; } rec[] = {
            {"erec", h->dw.erec, 6 * Ne}, {"nrec", h->dw.nrec, 10 * Nn}, {"xy", h->dw.xy, 2 * Nn}, {"delta_x", h->dw.delta_x, Ne},
            {"surface", h->dw.surface, Ne}, {"tau_a", h->dw.D_tau_a, 2 * Nn},
        };
        for (auto &t : rec)
            if (!std::strcmp(t.nm, name)) {
                if (n != t.len) return fail(h, NXS_ERR_INVALID, "debug_array %s has %lld entries, caller asked %lld", name, (long long)t.len, (long long)n);
                HIPCHK(h, hipMemcpyAsync(out, t.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                return NXS_OK;
            }
    }
    if (!std::strcmp(name, "shape_range")) {   // [1] the per-step range flag of the shared-reciprocal shape coefficients (raised by the prep kernels, lowered by k_update)
        if (n != 1) return fail(h, NXS_ERR_INVALID, "debug_array shape_range has 1 entry");
        int v = 0;
        HIPCHK(h, hipMemcpyAsync(&v, h->dw.shape_range, sizeof v, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        out[0] = (double)v;
        return NXS_OK;
    }
#ifdef NXS_PHASE_TIMING
    if (!std::strcmp(name, "phase_times")) {  // [8192][8] timestamps (100 MHz) of the last fused launch, as doubles
        std::vector<long long> t(8 * 8192);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_phase_t), t.size() * sizeof(long long)));
        for (int64_t i = 0; i < n && i < (int64_t)t.size(); ++i) out[i] = (double)t[i];
        return NXS_OK;
    }
    if (!std::strcmp(name, "phase_times_prep")) {  // the same of the last k_prep_fused
        std::vector<long long> t(8 * 8192);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_phase_p), t.size() * sizeof(long long)));
        for (int64_t i = 0; i < n && i < (int64_t)t.size(); ++i) out[i] = (double)t[i];
        return NXS_OK;
    }
#endif
    return fail(h, NXS_ERR_INVALID, "unknown debug array '%s'", name);
} catch (...) { return dyn_caught(h, "nxs_dyn_debug_array"); }

// ------------------------------------------------------------------------------------------------
// the launches

namespace {

#define LAUNCH(h, kern, n, ...)                                                              \
    do {                                                                                      \
        hipLaunchKernelGGL(kern, dim3(nblocks(n)), dim3(BLOCK), 0, (h)->stream, __VA_ARGS__); \
    } while (0)

int halo_exchange(nxs_dyn_handle *h, double *vec, double move_dt) {
    // updateGhosts (FE.cpp:13963-13996): pack -> grouped send/recv -> unpack
    const int ns = (int)h->send_procs.size(), nr = (int)h->recv_procs.size();
    const int ts = h->send_offsets[ns], tr = h->recv_offsets[nr];
    if (h->ipc_ready && !h->halo_fn) {
        // device-direct: pack + peer stores + flags, then wait + unpack (+ ghost-node move); no host work
        hipLaunchKernelGGL(k_halo_push, dim3(nblocks(ts)), dim3(BLOCK), 0, h->stream, (const double *)vec, h->dm.Nn, ts,
                           h->d_send_index, h->d_send_seg, h->d_send_off, h->ipc, h->rank, 0);
        hipLaunchKernelGGL(k_halo_pull, dim3(nblocks(tr)), dim3(BLOCK), 0, h->stream, vec, h->dm, h->ds, tr, h->d_recv_index,
                           h->d_recv_seg, h->d_recv_off, h->ipc, move_dt, 0, h->d_recv_procs, 0);
        return NXS_OK;
    }
    if (!h->comm && !h->halo_fn) return fail(h, NXS_ERR_STATE, "halo exchange needs nxs_dyn_comm_init, nxs_dyn_ipc_connect or nxs_dyn_set_halo_exchange_fn");
    if (ts > 0) LAUNCH(h, k_halo_pack, ts, vec, h->dm.Nn, ts, h->d_send_index, h->d_send_seg, h->d_send_off, h->d_send_buf);
    if (h->halo_fn) {
        // host-staged: exactly the reference's M_comm.send / M_comm.recv of packed std::vector<double>
        if (ts > 0) HIPCHK(h, hipMemcpyAsync(h->h_send, h->d_send_buf, 2 * (size_t)ts * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const int rc = h->halo_fn(h->halo_ctx, h->h_send, h->h_recv);
        if (rc != 0) return fail(h, NXS_ERR_COMM, "halo exchange callback returned %d", rc);
        if (tr > 0) {
            HIPCHK(h, hipMemcpyAsync(h->d_recv_buf, h->h_recv, 2 * (size_t)tr * sizeof(double), hipMemcpyHostToDevice, h->stream));
            LAUNCH(h, k_halo_unpack, tr, vec, h->dm, h->ds, tr, h->d_recv_index, h->d_recv_seg, h->d_recv_off, h->d_recv_buf, move_dt);
        }
        return NXS_OK;
    }
    const int ncclDouble = 8;  // ncclFloat64
    int e = h->rccl.GroupStart();
    for (int k = 0; k < ns && e == 0; ++k)   // (a segment without nodes -- a direction set_halo added -- is no message: both ends know it is empty)
        if (h->send_offsets[k + 1] > h->send_offsets[k])
            e = h->rccl.Send(h->d_send_buf + 2 * (size_t)h->send_offsets[k], 2 * (size_t)(h->send_offsets[k + 1] - h->send_offsets[k]),
                             ncclDouble, h->send_procs[k], h->comm, h->stream);
    for (int k = 0; k < nr && e == 0; ++k)
        if (h->recv_offsets[k + 1] > h->recv_offsets[k])
            e = h->rccl.Recv(h->d_recv_buf + 2 * (size_t)h->recv_offsets[k], 2 * (size_t)(h->recv_offsets[k + 1] - h->recv_offsets[k]),
                             ncclDouble, h->recv_procs[k], h->comm, h->stream);
    int e2 = h->rccl.GroupEnd();
    if (e == 0) e = e2;
    if (e != 0) return fail(h, NXS_ERR_COMM, "halo send/recv: %s", h->rccl.GetErrorString(e));
    if (tr > 0) LAUNCH(h, k_halo_unpack, tr, vec, h->dm, h->ds, tr, h->d_recv_index, h->d_recv_seg, h->d_recv_off, h->d_recv_buf, move_dt);
    return NXS_OK;
}

bool multi_rank(const nxs_dyn_handle *h) { return h->nranks > 1; }

PingPong pingpong(const nxs_dyn_handle *h, int parity) {
    const DevState &s = h->ds;
    PingPong b;
    if (parity == 0) { b.VTc = s.VT; b.Sc = s.S4a; b.VTn = s.VT2; b.Sn = s.S4b; }
    else { b.VTc = s.VT2; b.Sc = s.S4b; b.VTn = s.VT; b.Sn = s.S4a; }
    return b;
}

// sub-step `sidx` of the fused path: sigma/damage ping-pong by parity; velocities move through the ring
// (ring of 2 == ping-pong between VT and VT2 when the deferred mesh move is off)
// halo != 0: the sub-step also performs updateGhosts (HaloFused); from_mailbox = ghosts come from exchange x-1
void launch_fused(nxs_dyn_handle *h, int sidx, double move_dt, int halo = 0, int from_mailbox = 0) {
    PingPong b = pingpong(h, sidx & 1);
    const int R = h->ring.R;
    b.VTc = h->ring.slot[sidx % R];
    b.VTn = h->ring.slot[(sidx + 1) % R];
    const dim3 grid(h->dpch.nP);
    const bool big = h->dpch.Pmax > NXS_T256_MAXP || h->dpch.Emax > 3 * 256, pow4 = h->dp.ers_int == 4;
    // streaming hints keep the state from displacing the reusable arrays -- a gain only when a sub-step's ~210 B/triangle do not fit
    // the 256 MiB Infinity Cache anyway: 1.46 M triangles 7.08 -> 6.96 ms/step with them, 730 k 3.75 -> 3.98, 367 k 2.20 -> 2.33, 182 k 1.27 -> 1.31
    const int nt_mask = h->nt_mask >= 0 ? h->nt_mask : (h->dm.Ne >= 1000000 ? 3 : 0);
    // parameters from memory (PMEM) where one round of resident workgroups covers the partition, by value where several rounds stream
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
    const bool pmem = h->dpch.nP <= 5 * cus / 2;  // 511 patches (182 k triangles): 1.375 -> 1.344 ms/step; 752 patches (263 k): 1.83 -> 1.91
#define FUSED_K(TT, PP, NN, HH, MM, HFP, NB, FM) hipLaunchKernelGGL((k_substep_fused<TT, PP, NN, HH, MM>), grid, dim3(TT), h->fused_lds, h->stream, h->dm, h->dpch, h->ds, h->dw, h->dp, (const DevParams *)h->d_dp, b, move_dt, HFP, NB, FM)
    if (halo) {
#define FUSED_H(TT, PP, NN) do { if (pmem) FUSED_K(TT, PP, NN, true, true, (const HaloFused *)h->d_hf, h->hf.n_boundary, from_mailbox); else FUSED_K(TT, PP, NN, true, false, (const HaloFused *)h->d_hf, h->hf.n_boundary, from_mailbox); } while (0)
        if (big) { if (pow4) { if (nt_mask) FUSED_H(512, true, 3); else FUSED_H(512, true, 0); } else { FUSED_H(512, false, 0); } }
        else { if (pow4) { if (nt_mask) FUSED_H(256, true, 3); else FUSED_H(256, true, 0); } else { FUSED_H(256, false, 0); } }
#undef FUSED_H
        return;
    }
#define FUSED(TT, PP, NN) do { if (pmem) FUSED_K(TT, PP, NN, false, true, (const HaloFused *)nullptr, 0, 0); else FUSED_K(TT, PP, NN, false, false, (const HaloFused *)nullptr, 0, 0); } while (0)
#define FUSED_NT(TT, PP) switch (nt_mask) { case 0: FUSED(TT, PP, 0); break; case 1: FUSED(TT, PP, 1); break; case 3: FUSED(TT, PP, 3); break; case 4: FUSED(TT, PP, 4); break; case 5: FUSED(TT, PP, 5); break; default: FUSED(TT, PP, 7); break; }
    if (big) { if (pow4) { FUSED_NT(512, true); } else { FUSED(512, false, 0); } }
    else { if (pow4) { FUSED_NT(256, true); } else { FUSED(256, false, 0); } }
#undef FUSED_NT
#undef FUSED
#undef FUSED_K
}

// sub-steps sidx .. sidx+D-1 in one launch (k_substep_multi): sigma/damage ping-pong per LAUNCH, velocities through the ring
void launch_multi(nxs_dyn_handle *h, int sidx, int D, bool halo = false) {
    PingPong b = pingpong(h, (sidx / D) & 1);
    const int R = h->ring.R;
    b.VTc = h->ring.slot[sidx % R];
    b.VTn = nullptr;
    VTOut vo{};
    for (int k = 0; k < D; ++k) vo.slot[k] = h->ring.slot[(sidx + 1 + k) % R];
    const dim3 grid(h->dpch2.nP);
    const bool pow4 = h->dp.ers_int == 4;
    if (h->pair_kernel) {  // (D == 2: upload_patches2 / upload_pair_patches_mr cut the patches for it)
        if (halo) {        // several ranks: both exchanges inside the launch; the first launch of a step finds its ghosts in the velocity buffer
            PairHalo ph = h->pairh;
            ph.from_mailbox = sidx > 0 ? 1 : 0;
            if (pow4) hipLaunchKernelGGL((k_substep_pair<512, true, 3, true>), grid, dim3(512), h->pair_lds, h->stream, h->dm, h->dpch2, h->ds, h->dw, h->dp, b, vo, (const HaloFused *)h->d_hf, ph);
            else hipLaunchKernelGGL((k_substep_pair<512, false, 3, true>), grid, dim3(512), h->pair_lds, h->stream, h->dm, h->dpch2, h->ds, h->dw, h->dp, b, vo, (const HaloFused *)h->d_hf, ph);
            return;
        }
        if (h->pair_threads == 256) {
            if (pow4) hipLaunchKernelGGL((k_substep_pair<256, true, 3>), grid, dim3(256), h->pair_lds, h->stream, h->dm, h->dpch2, h->ds, h->dw, h->dp, b, vo, (const HaloFused *)nullptr, PairHalo{});
            else hipLaunchKernelGGL((k_substep_pair<256, false, 3>), grid, dim3(256), h->pair_lds, h->stream, h->dm, h->dpch2, h->ds, h->dw, h->dp, b, vo, (const HaloFused *)nullptr, PairHalo{});
            return;
        }
        if (h->move_now) {   // the mesh move of the two sub-steps inside the launch
            // (the first velocity slot is not written; the LAST launch of a step whose final velocity lands in M_VT itself writes it a second time there instead: the
            // smoother wants two equal buffers and would otherwise copy one)
            vo.slot[0] = (sidx + D == h->dp.substeps && (h->dp.substeps % R) == 0) ? h->ring.slot[(sidx + 1) % R] : nullptr;
            if (pow4) hipLaunchKernelGGL((k_substep_pair<512, true, 3, false, true>), grid, dim3(512), h->pair_lds, h->stream, h->dm, h->dpch2, h->ds, h->dw, h->dp, b, vo, (const HaloFused *)nullptr, PairHalo{});
            else hipLaunchKernelGGL((k_substep_pair<512, false, 3, false, true>), grid, dim3(512), h->pair_lds, h->stream, h->dm, h->dpch2, h->ds, h->dw, h->dp, b, vo, (const HaloFused *)nullptr, PairHalo{});
            return;
        }
        if (pow4) hipLaunchKernelGGL((k_substep_pair<512, true, 3>), grid, dim3(512), h->pair_lds, h->stream, h->dm, h->dpch2, h->ds, h->dw, h->dp, b, vo, (const HaloFused *)nullptr, PairHalo{});
        else hipLaunchKernelGGL((k_substep_pair<512, false, 3>), grid, dim3(512), h->pair_lds, h->stream, h->dm, h->dpch2, h->ds, h->dw, h->dp, b, vo, (const HaloFused *)nullptr, PairHalo{});
        return;
    }
#define MULTI(TT, PP, NN) hipLaunchKernelGGL((k_substep_multi<TT, PP, NN>), grid, dim3(TT), h->pair_lds, h->stream, h->dm, h->dpch2, h->ds, h->dw, (const DevParams *)h->d_dp, b, vo)
// no non-temporal hints: this kernel runs where the mesh lives in the caches (58 k triangles: 0.768 ms/step with them, 0.750 without; 111 k: 0.927 / 0.90)
#define MULTI_T(TT) do { if (pow4) MULTI(TT, true, 0); else MULTI(TT, false, 0); } while (0)
    if (h->pair_threads == 768) MULTI_T(768); else if (h->pair_threads == 512) MULTI_T(512); else MULTI_T(256);
#undef MULTI_T
#undef MULTI
}

// (re)build the ring of velocity buffers of the fused path: slot 0 is M_VT itself, slot 1 is VT2
int setup_ring(nxs_dyn_handle *h, int K) {
    const int R = K + 1;
    if (h->ring.R == R) return NXS_OK;
    // Buffers are only ever ADDED (they go with the mesh): a hipFree synchronises the whole device, and on a device that another handle of this
    // process shares (the several-ranks-per-process tests and rehearsals) that other rank's kernels may be spinning for THIS rank's next launch --
    // freeing here, inside a step, deadlocked such runs until the 10 s guard fired whenever a run switched from the long ring to the short one
    // (round 3: found in the kernel statistics of a two-rank run, one k_halo_pull of 10 s).
    h->ring.slot[0] = h->ds.VT;
    h->ring.slot[1] = h->ds.VT2;
    for (int i = 2; i < R; ++i) {
        if (i - 2 < (int)h->ring_allocs.size()) { h->ring.slot[i] = static_cast<double *>(h->ring_allocs[i - 2]); continue; }
        int rc = dev_alloc(h, h->ring_allocs, &h->ring.slot[i], 2 * (size_t)h->dm.Nn);
        if (rc) return rc;
    }
    for (int i = R; i < NXS_MAX_RING; ++i) h->ring.slot[i] = nullptr;
    h->ring.R = R;
    return NXS_OK;
}

// The resident sub-step kernel of this handle: ONE place decides the instantiation, for the occupancy query and for the launch alike.
// (WPE = 2: the several-rank build with all the registers it wants, BBM's default exponent only.)
const void *resident_kernel(const nxs_dyn_handle *h, bool mr, bool ovl) {
    const bool p4 = h->res_pow4;
    if (h->res_big) {  // one large patch per CU, four elements and two own nodes per thread
        if (mr && ovl) return p4 ? (const void *)k_substep_resident_big<true, true, true> : (const void *)k_substep_resident_big<false, true, true>;
        if (mr) return p4 ? (const void *)k_substep_resident_big<true, true, false> : (const void *)k_substep_resident_big<false, true, false>;
        if (ovl) return p4 ? (const void *)k_substep_resident_big<true, false, true> : (const void *)k_substep_resident_big<false, false, true>;
        return p4 ? (const void *)k_substep_resident_big<true, false, false> : (const void *)k_substep_resident_big<false, false, false>;
    }
    if (mr && h->res_wpe == 2 && p4) return ovl ? (const void *)k_substep_resident<512, true, true, true, 2> : (const void *)k_substep_resident<512, true, true, false, 2>;
    if (mr && ovl) return p4 ? (const void *)k_substep_resident<512, true, true, true> : (const void *)k_substep_resident<512, false, true, true>;
    if (mr) return p4 ? (const void *)k_substep_resident<512, true, true> : (const void *)k_substep_resident<512, false, true>;
    return p4 ? (const void *)k_substep_resident<512, true, false> : (const void *)k_substep_resident<512, false, false>;
}

// The resident launch needs every workgroup of its grid on a CU at once, and its workgroups spin: a handle claims its workgroup slots in the device's
// registry before it builds the loop and gives them back when its tables go; a claim that does not fit beside what is claimed already -- by handles
// of this process or of any other process on the device, with headroom for the co-tenants' ordinary kernels where the device is shared -- is refused
// up front and the step runs one kernel per sub-step (nxs_resident_registry.hpp has the rule and the evidence behind it).
// (kind: nxs_reg::KIND_RESIDENT / KIND_PAIR -- one claim per handle, held by ONE of its two grids of waiting workgroups; a release names the grid that lets go,
// so that the resident loop's tables going away do not take the pair patches' claim with them)
void resident_registry_release(const nxs_dyn_handle *h, int kind) {
    if (!h->reg_key.empty()) nxs_reg::table_for(h->reg_key).release((uint64_t)(uintptr_t)h, kind);
}
bool resident_registry_claim(const nxs_dyn_handle *h, int workgroups, int slots, std::string *why, int kind) {
    if (h->reg_key.empty()) { if (why) *why = "the handle is not registered on its device"; return false; }
    return nxs_reg::table_for(h->reg_key).claim((uint64_t)(uintptr_t)h, workgroups, slots, multi_rank(h), why, kind);
}

// The blocks of this handle's ORDINARY kernels that may spin for a neighbour rank (k_smooth_halo, k_halo_pull, the boundary patches of k_substep_fused<HALO>): the
// largest such grid, as a fraction of the device, is registered -- other handles' claims leave it room (nxs_resident_registry.hpp).  reset: start from nothing.
void register_waiting_grid(nxs_dyn_handle *h, int blocks, int slots, bool reset) {
    if (h->reg_key.empty()) return;
    if (reset) { h->ord_blocks = 0; h->ord_slots = 0; }
    if (blocks > 0 && slots > 0 && (h->ord_blocks == 0 || (double)blocks / slots > (double)h->ord_blocks / h->ord_slots)) { h->ord_blocks = blocks; h->ord_slots = slots; }
    nxs_reg::table_for(h->reg_key).set_ordinary((uint64_t)(uintptr_t)h, h->ord_blocks, h->ord_slots);
}

// ... the smoother's: k_smooth_persist runs at most 128 persistent workgroups that wait for each other and for the neighbour ranks; with option smooth_persist 0 every
// block of k_smooth_halo (one per BLOCK own nodes) may spin for a neighbour's sweep; k_halo_pull's blocks (one per BLOCK ghosts) spin for its flags.  Called where the
// halo lists are set and where the option changes.
void register_smoother_grid(nxs_dyn_handle *h) {
    if (!h->have_halo) return;
    const int No = h->dm.No, tr = h->recv_offsets[h->recv_procs.size()];
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_smooth_halo, BLOCK, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
    const int sweeps = h->smooth_persist != 0 ? std::min(nblocks(No), 128) : nblocks(No);
    register_waiting_grid(h, h->nranks > 1 ? std::max(sweeps, nblocks(tr)) : 0, per_cu * device_cus(h), true);
    if (h->hf_ready) register_waiting_grid(h, h->hf.n_boundary, 2 * device_cus(h));   // (the boundary patches of k_substep_fused<HALO>, as build_halo_fused registers them)
}

// tables of the halo exchange fused into the sub-step kernel (see HaloFused)
int build_halo_fused(nxs_dyn_handle *h) {
    drop_pool(h, h->hf_allocs);
    h->hf = HaloFused{};
    h->hf_ready = false;
    const int Nn = h->dm.Nn, No = h->dm.No, nP = h->dpch.nP;
    if (!h->hp || h->hp->nP != nP) return fail(h, NXS_ERR_STATE, "fused halo tables: patches / halo lists missing");
    const nxs_cut::HaloLists hl{&h->send_offsets, &h->recv_offsets, &h->h_send_index, &h->h_recv_index, (int)h->send_procs.size(), (int)h->recv_procs.size()};
    nxs_cut::HaloFusedPlan plan;
    const std::string why = nxs_cut::plan_halo_fused(Nn, No, hl, *h->hp, plan);  // (host only: nxs_patchcut.hpp)
    if (!why.empty()) return fail(h, NXS_ERR_STATE, "%s", why.c_str());
    if (plan.reordered) {  // the patch arrays again, boundary patches first: the grid starts with them and "boundary" is blk < n_boundary
        HIPCHK(h, hipStreamSynchronize(h->stream));
        release_graph(h);
        int rcu = upload_host_patches(h, *h->hp);
        if (rcu) return rcu;
    }
    HaloFused &f = h->hf;
    int rc;
    if ((rc = dev_upload(h, h->hf_allocs, &f.send_ptr, plan.sptr))) return rc;
    if ((rc = dev_upload(h, h->hf_allocs, &f.send_k, plan.sk))) return rc;
    if ((rc = dev_upload(h, h->hf_allocs, &f.send_pos, plan.spos))) return rc;
    if ((rc = dev_upload(h, h->hf_allocs, &f.ghost_off, plan.goff))) return rc;
    if ((rc = dev_upload(h, h->hf_allocs, &f.ghost_srl, plan.gsrl))) return rc;
    if ((rc = dev_upload(h, h->hf_allocs, &f.ghost_k, plan.gk))) return rc;
    unsigned int *ctr = nullptr;
    if ((rc = dev_alloc(h, h->hf_allocs, &ctr, 32 * 19))) return rc;   // [0] and [32 (g + 1)]: two-level tickets; [32 * 17]: the generation word of k_smooth_persist's barrier
    HIPCHK(h, hipMemsetAsync(ctr, 0, 32 * 19 * sizeof(unsigned int), h->stream));
    f.done_all = ctr;
    if ((rc = dev_upload(h, h->hf_allocs, &f.send_block_rank, plan.send_block_rank))) return rc;  // k_smooth_halo: which blocks store into a mailbox
    f.n_send_blocks = plan.n_send_blocks;
    f.no_release = h->res_no_release;
    f.send_off = h->d_send_off;
    f.n_boundary = plan.n_boundary;
    register_waiting_grid(h, plan.n_boundary, 2 * device_cus(h));   // (two 512-thread workgroups of the fused kernel per CU)
    f.No = No;
    {
        unsigned long long *raw = nullptr;  // device copy of the struct itself (filled in by run_substeps once the mailboxes are connected)
        if ((rc = dev_alloc(h, h->hf_allocs, &raw, (sizeof(HaloFused) + 7) / 8))) return rc;
        h->d_hf = reinterpret_cast<HaloFused *>(raw);
        h->d_hf_dirty = true;
    }
    h->hf_ready = true;
    if (getenv("NXS_DEBUG_PATCHES")) {
        fprintf(stderr, "[nxs] rank %d fused halo: %d of %d patches on the boundary, %d sent nodes, %d ghosts\n", h->rank, plan.n_boundary, nP, plan.sptr[No], Nn - No);
        std::string sizes;   // own nodes / elements / of them sent, of the boundary patches (they lead the arrays now)
        for (int q = 0; q < plan.n_boundary && q < 80; ++q) {
            int sent = 0, ghosts = 0;
            const int *nd = h->hp->pnodes.data() + (size_t)q * h->hp->Mmax;
            for (int i = 0; i < h->hp->node_cnt[q]; ++i) { if (nd[i] >= No) ++ghosts; else if (i < h->hp->own_cnt[q] && plan.sptr[nd[i] + 1] > plan.sptr[nd[i]]) ++sent; }
            char b[64]; snprintf(b, sizeof b, " %d/%d/s%d/g%d", h->hp->own_cnt[q], h->hp->elem_cnt[q], sent, ghosts); sizes += b;
        }
        fprintf(stderr, "[nxs] rank %d boundary patches (own nodes / elements / sent / ghosts staged):%s\n", h->rank, sizes.c_str());
    }
    return NXS_OK;
}

// Tables of the resident sub-step kernel (nxs_cut::plan_resident), the counters, the exchange buffers.  NXS_OK with res_ready == false
// means "not possible here" (the caller then runs one kernel per sub-step).  Everything lives in a pool of its own that is given back
// before it is rebuilt (options fused / resident_wide / resident_overlap / resident_dryrun, a change of parameters, a timed-out launch).
int build_resident(nxs_dyn_handle *h) {
    h->res_ready = false;
    drop_pool(h, h->res_allocs);
    h->res = DevResident{};
    h->d_vt3 = nullptr;
    if (!h->hp || h->hp->nP != h->dpch.nP) return NXS_OK;
    const HostPatches &hp = *h->hp;
    const int nP = hp.nP, No = h->dm.No, Nn = h->dm.Nn, S = h->dp.substeps;
    const bool mr = multi_rank(h);
    const bool big = nxs_cut::resident_is_big(hp);
    const bool ovl = big ? h->res_overlap != 0 : (mr && h->res_overlap == 1);
    h->res_ovl = ovl;
    auto refuse = [&](const char *why) {
        if (getenv("NXS_DEBUG_PATCHES")) fprintf(stderr, "[nxs] rank %d resident kernel not possible: %s\n", h->rank, why);
        h->res_failed = true;
        return NXS_OK;
    };
    if (S > NXS_RES_MAXS) return refuse("more sub-steps than the kernel keeps counters for");
    nxs_cut::ResidentPlan plan;
    nxs_cut::plan_resident(hp, Nn, No, mr, (int)h->send_procs.size(), ovl, plan);
    if (!plan.ok) return refuse(plan.why.c_str());
    h->res_big = big;
    h->res_lds = big ? nxs_cut::resident_big_lds_of(hp, mr) : nxs_cut::resident_lds_of(hp, mr);
    if (h->res_lds > 160 * 1024) return refuse("a patch needs more LDS than a CU has");
    // every workgroup must be resident at once
    int per_cu = 0;
    const int cus = device_cus(h);
    const bool p4 = h->dp.ers_int == 4;
    // one workgroup per CU is enough and the caller says the device is this handle's alone (option resident_wide): the several-rank build with all
    // the registers it wants -- one such workgroup fills a CU, so ranks that share a device (the tests) would no longer fit side by side
    h->res_wpe = (mr && p4 && h->res_wide && nP <= cus && !big) ? 2 : 4;
    h->res_pow4 = p4;
    const void *kern = resident_kernel(h, mr, ovl);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 512, h->res_lds);
    if (e != hipSuccess || (long long)per_cu * cus < nP) {
        (void)hipGetLastError();
        char why[160];
        snprintf(why, sizeof why, "%d patches, %d x %d workgroups fit (%zu B of LDS each)", nP, per_cu, cus, h->res_lds);
        return refuse(why);
    }
    // the device's other resident grids (other handles of this process: the ranks a host drives from one process, the tests): all of them
    // together must fit, or the spinning workgroups of one keep the other's from ever starting
    {
        std::string why;
        if (!resident_registry_claim(h, nP, per_cu * cus, &why, nxs_reg::KIND_RESIDENT)) return refuse(why.c_str());
        if (h->pair_claim) { h->pair_claim = false; h->pair_ready = false; }   // (a handle holds ONE claim: the pair patches' went with it and are cut -- and claimed -- again if they are wanted)
    }
    int rc;
    DevResident &r = h->res;
    if (ovl) {
        if (getenv("NXS_DEBUG_PATCHES")) fprintf(stderr, "[nxs] rank %d resident kernel: %.1f %% of the patch elements computed under the exchange\n", h->rank, 100. * plan.early_fraction);
        if ((rc = dev_upload(h, h->res_allocs, &r.pelem, plan.rpelem))) return rc;
        if ((rc = dev_upload(h, h->res_allocs, &r.ptri, plan.rptri))) return rc;
        if ((rc = dev_upload(h, h->res_allocs, &r.pfan, plan.rpfan))) return rc;
        if ((rc = dev_upload(h, h->res_allocs, &r.ecut, plan.ecut))) return rc;
    }
    if ((rc = dev_upload(h, h->res_allocs, &r.pnbr, plan.nbr))) return rc;
    if ((rc = dev_upload(h, h->res_allocs, &r.pnbr_cnt, plan.cnt))) return rc;
    r.rank = h->rank;
    if (mr) {  // the ghosts' ring: what arrived after every sub-step but the last (sized for THIS number of sub-steps: a change of parameters rebuilds the tables)
        r.NG = Nn - No;
        if ((rc = dev_alloc(h, h->res_allocs, &r.gring, std::max<size_t>((size_t)std::max(S - 1, 1) * 2 * (size_t)r.NG, 1)))) return rc;
    }
    h->res_substeps = S;
    if ((rc = dev_alloc(h, h->res_allocs, &r.flag, 32 * (size_t)nP + NXS_RES_MAXS + 32))) return rc;  // counters behind the flags: one memset per launch
    r.cnt = r.flag + 32 * (size_t)nP;
    r.raised = r.cnt + NXS_RES_MAXS;
    if ((rc = dev_alloc(h, h->res_allocs, &r.error, 1))) return rc;
    HIPCHK(h, hipMemsetAsync(r.error, 0, sizeof(int), h->stream));
    if ((rc = dev_alloc(h, h->res_allocs, &h->d_vt3, 2 * (size_t)Nn))) return rc;
    r.X0 = h->ds.VT2; r.X1 = h->d_vt3;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (getenv("NXS_DEBUG_PATCHES")) fprintf(stderr, "[nxs] rank %d resident kernel: %d patches (%d x %d fit), %zu B of LDS each, up to %d neighbour patches, %d ghost nodes\n", h->rank, nP, per_cu, cus, h->res_lds, plan.max_nbr, mr ? Nn - No : 0);
    h->res_ready = true;
    return NXS_OK;
}

void launch_substep(nxs_dyn_handle *h, double move_dt) {
    if (h->dp.dynamics_type == NXS_DYN_BBM) {
        if (h->trace_branches) {
            if (h->dp.ers_int == 4) LAUNCH(h, (k_sigma_bbm<true, true>), h->dm.Ne, h->dm, h->ds, h->dw, h->dp);
            else LAUNCH(h, (k_sigma_bbm<false, true>), h->dm.Ne, h->dm, h->ds, h->dw, h->dp);
        }
        else if (h->dp.ers_int == 4) LAUNCH(h, k_sigma_bbm<true>, h->dm.Ne, h->dm, h->ds, h->dw, h->dp);
        else LAUNCH(h, k_sigma_bbm<false>, h->dm.Ne, h->dm, h->ds, h->dw, h->dp);
    }
    else
        LAUNCH(h, k_sigma_vp, h->dm.Ne, h->dm, h->ds, h->dw, h->dp);
    LAUNCH(h, k_solve_move, h->dm.No, h->dm, h->ds, h->dw, h->dp, move_dt);
}

// sub-steps per launch of this step (1 = the v2 / v1 kernels); builds the D-ring patches when they are needed
int choose_depth(nxs_dyn_handle *h) {
    const int S = h->dp.substeps;
    const double move_dt = (h->dp.dynamics_type == NXS_DYN_MEVP) ? 0. : h->dp.dte;
    int D = 1;
    // Automatic (fused == 3): only where ONE round of one patch per CU covers the mesh (<= 256 own nodes per patch: 65 k nodes, 130 k
    // triangles on 256 CUs) -- 111 k triangles: 1.47 (v2) -> 0.97 ms/step; 182 k triangles, two patches per CU: 1.65 -> 1.90-2.31.
    // ... and on meshes that STREAM from HBM (fused == 3, >= 400 k triangles, an even number of sub-steps) two sub-steps per launch with the
    // stresses between them in registers and two workgroups per CU (k_substep_pair): 2 km 6.3 -> 5.7 ms of sub-steps.
    const bool single = !multi_rank(h) && move_dt != 0. && S >= 2 && !h->pair_failed;
    // several ranks: two sub-steps per launch with BOTH exchanges inside it (k_substep_pair<HALO>: device-direct mailboxes, the exchange inside the kernels) --
    // automatically where the partition streams from HBM (more than 65 k nodes: a rank of two of the 2 km mesh; smaller partitions run the resident loop where
    // they have a device to themselves), with option pair_regs = 1 at any size
    const bool mr_pair = multi_rank(h) && h->have_halo && h->ipc_ready && !h->halo_fn && h->halo_fused && eff_fused(h) == 3 && move_dt != 0. && S >= 2 && S % 2 == 0 &&
                         !h->pair_failed && (h->pair_regs == 1 || (h->pair_regs < 0 && (long long)h->dm.Nn > 256ll * 256)) && (h->pair_depth == 0 || h->pair_depth == 2);
    if (mr_pair) {
        if (!h->hf_ready && build_halo_fused(h) != NXS_OK) { h->pair_failed = true; h->depth_now = 1; return 1; }
        if ((!h->pair_ready || h->pair_depth_built != 2 || !h->pair_kernel) && upload_pair_patches_mr(h) != NXS_OK) h->pair_failed = true;
        h->depth_now = (h->pair_ready && !h->pair_failed) ? 2 : 1;
        return h->depth_now;
    }
    // (automatic: every mesh too large for one k_substep_multi patch per CU, i.e. above 65 k nodes)
    const bool streaming_pair = single && eff_fused(h) == 3 && h->pair_regs != 0 && (long long)h->dm.Nn > 256ll * 256 && S % 2 == 0 && (h->pair_depth == 0 || h->pair_depth == 2);
    if (streaming_pair) {
        D = 2;
        if ((!h->pair_ready || h->pair_depth_built != 2 || !h->pair_kernel) && upload_patches2(h, 2, false, true) != NXS_OK) {
            h->pair_failed = true;  // (a numbering without any locality, huge fans): one sub-step per launch
            D = 1;
        }
    } else if ((eff_fused(h) == 2 || (eff_fused(h) == 3 && (long long)h->dm.Nn <= 256ll * 256)) && single) {
        D = std::min(h->pair_depth > 0 ? h->pair_depth : 4, std::min(S, NXS_MAX_DEPTH));
        while (D > 1 && S % D != 0) --D;
        const bool want_pair = D == 2 && h->pair_regs == 1;   // (forced depth 2 with option pair_regs = 1: k_substep_pair at any size)
        if (D >= 2 && (!h->pair_ready || h->pair_depth_built != D || h->pair_kernel != want_pair) && upload_patches2(h, D, eff_fused(h) == 3, want_pair) != NXS_OK) {
            h->pair_failed = true;  // no patch size fits (a numbering without any locality, huge fans): one sub-step per launch
            D = 1;
        }
    }
    h->depth_now = D;
    return D;
}

int run_substeps(nxs_dyn_handle *h) {
    const int S = h->dp.substeps;
    const double move_dt = (h->dp.dynamics_type == NXS_DYN_MEVP) ? 0. : h->dp.dte;
    const bool fused = eff_fused(h) != 0;
    const int bbm = h->dp.dynamics_type == NXS_DYN_BBM;
    const bool mr = multi_rank(h);
    // deferred mesh move (fused path, not mEVP whose single move comes after the loop)
    const bool device_halo = mr && h->ipc_ready && !h->halo_fn;  // no host work inside the loop: graph-capturable
    // auto ring: one flush per step (up to 120 sub-steps) on meshes that stream from HBM -- the flush reads every slot once
    // whatever its period, so a longer ring only saves UM/UT passes (2 km: 7.60 -> 7.49 ms/step from 16 to 120, 1.4 GB of
    // slots); also whenever the halo exchange runs inside the sub-step kernel
    const int want_ring = h->um_ring > 0 ? h->um_ring : ((h->dm.Ne >= 400000 || (device_halo && h->halo_fused)) ? 120 : 1);
    // v3: D sub-steps per launch -- single rank, the deferred mesh move (ring of >= D+1 buffers).  It trades redundant arithmetic
    // on the halo rings for less HBM traffic and fewer launches: a gain where the sub-step is latency-bound (10 km: 1.23 -> 0.97
    // ms/step), a loss as soon as a CU hosts more than one patch (182 k triangles: 1.65 -> 1.90) and where the v2 kernel already
    // runs at 5.5 TB/s with its VALUs half busy (2 km, D = 2: 7.4 -> 8.0).
    // D sub-steps per launch: the requested depth, else (auto) 4 (10 km: D = 2 / 3 / 4 / 5 / 6 / 8: 1.11 / 1.01 / 0.97 / 0.96 / 0.98 / 1.08 ms/step; the
    // rings grow the arithmetic by x2.0 per sub-step at D = 4) -- lowered until it divides the number of sub-steps
    const int D = h->depth_now;  // decided by choose_depth() before the prep kernels (they fill the records the multi kernel reads)
    const bool pair = D >= 2;
    int K = (fused && move_dt != 0.) ? std::max(1, std::min(want_ring, S)) : 1;
    // k_substep_pair on a single rank can apply the mesh move of its two sub-steps itself (M_UM / M_UT in and out once per launch): no ring beyond the three
    // buffers a launch reads and writes, no k_move_ring
    const bool move_in_pair = pair && !mr && D == 2 && h->pair_kernel && h->pair_threads == 512 && move_dt != 0. && !h->trace_branches && h->um_ring <= 0 &&
                              (h->pair_move == 1 || (h->pair_move < 0 && pair_move_default(h))) && !flow_wanted(h);
    h->move_now = move_in_pair;
    if (pair) {  // the ring is flushed between launches; by default once per step (a flush per launch costs 30 small launches at 10 km: 66 us of 0.88 ms)
        if (h->um_ring <= 0) K = std::min(S, NXS_MAX_RING - 1);
        K = std::max(D, K - K % D);
        if (move_in_pair) K = D;
    }
    const bool deferred = K > 1;
    if (fused) { int rc = setup_ring(h, K); if (rc) return rc; }
    const int R = h->ring.R;
    // the exchange inside the sub-step kernel: needs the deferred mesh move (ghost nodes are moved from the ring)
    // or no move at all (mEVP)
    const bool halo_in_kernel = device_halo && fused && h->halo_fused && (deferred || move_dt == 0.);
    if (halo_in_kernel && !h->hf_ready) { int rc = build_halo_fused(h); if (rc) return rc; }
    // v4: resident sub-step loop (opt-in).  Several ranks: only with the exchange inside the kernels (device-direct mailboxes).
    const bool res_wanted = h->fused == 4 && !h->trace_branches && !pair && move_dt != 0. && (!mr || (device_halo && h->halo_fused));
    if (res_wanted && mr && !h->hf_ready) { int rc = build_halo_fused(h); if (rc) return rc; }  // (re-uploads the patches boundary-first)
    if (h->res_ready && (h->res_substeps != S || h->res_pow4 != (h->dp.ers_int == 4))) {  // parameters changed since the tables were built:
        h->res_ready = false;                                                           // another kernel build (its residency unchecked), a ring of another length
        release_graph(h);
    }
    if (res_wanted && !h->res_ready && !h->res_failed) {  // (outside any capture)
        int rcr = build_resident(h);
        if (rcr) return rcr;
        if (h->res_failed && h->cut_big && !h->no_big_cut) {
            // the mesh was cut into one large patch per CU for k_substep_resident_big and that launch is not possible after all (the device's
            // workgroup slots are taken, ...): the one-launch-per-sub-step kernel wants its own cut (two smaller workgroups per CU, whole rounds)
            HIPCHK(h, hipStreamSynchronize(h->stream));
            release_graph(h);
            h->no_big_cut = true;
            int rcu = upload_patches(h);
            if (rcu) return rcu;
            if (mr && !h->hf_ready) { int rc = build_halo_fused(h); if (rc) return rc; }
            h->res_failed = true;   // (after the rebuilds, which reset it: this cut exists because the resident loop cannot run)
        }
    }
    const bool resident = res_wanted && h->res_ready && !h->res_failed;
    const bool pair_halo = pair && mr;   // (choose_depth built the tables: device-direct mailboxes, the exchange inside the kernels)
    // ... single rank: all the pairs of a step in one data-flow launch (needs the whole step in the ring: one flush, behind the launch)
    const bool flow = pair && !mr && D == 2 && h->pair_kernel && h->flow_ready && !h->flow_failed && flow_wanted(h) && !h->trace_branches && deferred && K == S && S % 2 == 0;
    if ((halo_in_kernel || (resident && mr) || pair_halo) && h->d_hf_dirty) {  // (outside any stream capture)
        HaloFused tmp = h->hf;
        tmp.ipc = h->ipc;
        HIPCHK(h, hipMemcpyAsync(h->d_hf, &tmp, sizeof tmp, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));  // tmp leaves scope
        h->d_hf_dirty = false;
    }
    const bool records_end_odd = resident ? false : (pair ? ((S / D) & 1) : (S & 1));
    // with the deferred mesh move the last flush of the step reads the newest velocity anyway and puts it back into M_VT itself; the
    // ring slot it came from then equals M_VT and serves the smoother as its second buffer (no copy before the sweeps)
    double *const vt_back = (deferred && !resident && !move_in_pair && (S % R) != 0) ? h->ds.VT : nullptr;
    h->smooth_second = vt_back ? h->ring.slot[S % R] : nullptr;
    if (move_in_pair)   // the final velocity: in a ring slot that k_pingpong_copy_back copies into M_VT, or in M_VT itself with a second copy in the last launch's free slot
        h->smooth_second = (S % R) ? h->ring.slot[S % R] : h->ring.slot[(S - D + 1) % R];
    auto pull_latest = [&](double *vec) {
        const int tr = h->recv_offsets[h->recv_procs.size()];
        hipLaunchKernelGGL(k_halo_pull, dim3(nblocks(tr)), dim3(BLOCK), 0, h->stream, vec, h->dm, h->ds, tr, h->d_recv_index,
                           h->d_recv_seg, h->d_recv_off, h->ipc, 0., 0, h->d_recv_procs, 1);
    };
    // the LAST flush of the step is launched behind the graph (one plain launch per step) so that its own events can bracket it
    const int final_count = (deferred && !resident && !move_in_pair) ? S - K * ((S - 1) / K) : 0;
    auto flush = [&](int s, int pending) {
        if (move_in_pair) return;                    // (the launches have moved the mesh)
        if (s == S - 1 && final_count > 0) return;   // (launched below)
        LAUNCH(h, k_move_ring, h->dm.Nn, h->dm, h->ds, h->ring, (s + 1 - (pending - 1)) % R, pending, move_dt, (double *)nullptr);
    };
    auto final_flush = [&]() -> int {
        if (final_count <= 0) return NXS_OK;
        const int k = h->cur ? (int)((h->cur - &h->ev[0][0]) / 5) : -1;
        if (k >= 0) HIPCHK(h, hipEventRecord(h->ev_flush[k][0], h->stream));
        LAUNCH(h, k_move_ring, h->dm.Nn, h->dm, h->ds, h->ring, (S - final_count + 1) % R, final_count, move_dt, vt_back);
        if (k >= 0) { HIPCHK(h, hipEventRecord(h->ev_flush[k][1], h->stream)); h->flush_timed[k] = true; }
        return NXS_OK;
    };
    auto loop = [&]() -> int {
        if (resident) {  // the whole loop in one launch; the element state is read from and written back to S4a (each record by its one writer)
            HIPCHK(h, hipMemsetAsync(h->res.flag, 0, (32 * (size_t)h->dpch.nP + NXS_RES_MAXS + 32) * sizeof(unsigned int), h->stream));
            {
                const DevParams *pdev = h->d_dp;
                const double *Sc = h->ds.S4a;
                double *Sn = h->ds.S4a;
                double mdt = move_dt;
                const HaloFused *hfp = mr ? h->d_hf : nullptr;
                int nb = mr ? h->hf.n_boundary : 0;
                void *args[] = {&h->dm, &h->dpch, &h->ds, &h->dw, &pdev, &h->res, &Sc, &Sn, &mdt, &hfp, &nb};
                HIPCHK(h, hipLaunchKernel(resident_kernel(h, mr, h->res_ovl), dim3(h->dpch.nP), dim3(512), args, h->res_lds, h->stream));
            }
            if (mr) {
                // the ghosts' mesh moves of all sub-steps but the last, from the ring the launch filled ...
                if (move_dt != 0. && h->res.NG > 0 && S > 1)
                    hipLaunchKernelGGL(k_ghost_ring_move, dim3(nblocks(h->res.NG)), dim3(BLOCK), 0, h->stream, h->dm, h->ds, (const double *)h->res.gring, h->res.NG, S - 1, move_dt, (const int *)h->res.error);
                // ... and the exchange of the last sub-step: the ghosts land in M_VT and make their last move
                const int tr = h->recv_offsets[h->recv_procs.size()];
                hipLaunchKernelGGL(k_halo_pull, dim3(nblocks(tr)), dim3(BLOCK), 0, h->stream, h->ds.VT, h->dm, h->ds, tr, h->d_recv_index,
                                   h->d_recv_seg, h->d_recv_off, h->ipc, move_dt, 0, h->d_recv_procs, 1);
            }
            return NXS_OK;
        }
        if (flow) {   // every pair of sub-steps of the step in ONE data-flow launch over the patches of k_substep_pair; the ring is flushed behind it (final_flush)
            HIPCHK(h, hipMemsetAsync(h->flow.queue, 0, h->flow_words * sizeof(unsigned int), h->stream));
            PairFlow f = h->flow;
            f.K = S / 2;
            f.S[0] = h->ds.S4a; f.S[1] = h->ds.S4b;
            const DevParams *pdev = h->d_dp;   // (explicit_solve keeps the device copy current)
            void *args[] = {&h->dm, &h->dpch2, &h->ds, &h->dw, &pdev, &h->ring, &f};
            HIPCHK(h, hipLaunchKernel(flow_kernel(h), dim3(h->flow_grid), dim3(512), args, h->pair_lds, h->stream));
            if (records_end_odd) LAUNCH(h, k_unpack_state, h->dm.Ne, h->dm, h->ds, bbm, (const double *)h->ds.S4b);
            return NXS_OK;
        }
        int pending = 0;  // sub-steps whose velocity still has to be applied to UM/UT
        for (int s = 0; s < S; ++s) {
            if (pair) {
                launch_multi(h, s, D, pair_halo);
                s += D - 1;
                pending += D;
                if (pending == K || s == S - 1) {
                    if (pair_halo) pull_latest(h->ring.slot[(s + 1) % R]);  // the newest ghosts, for the move / the end of the step
                    flush(s, pending);
                    pending = 0;
                }
                continue;
            }
            if (halo_in_kernel) {
                launch_fused(h, s, 0., 1, s > 0);
                const bool flush_now = deferred && (pending + 1 == K || s == S - 1);
                if (flush_now || s == S - 1) pull_latest(h->ring.slot[(s + 1) % R]);  // the newest ghosts, for the move / the end of the step
                if (flush_now) {
                    ++pending;
                    flush(s, pending);
                    pending = 0;
                } else if (deferred) ++pending;
                continue;
            }
            if (fused) launch_fused(h, s, deferred ? 0. : move_dt); else launch_substep(h, move_dt);
            if (mr) {
                // owned nodes were written to the buffer the next sub-step reads; ghosts must land there too
                double *vec = fused ? h->ring.slot[(s + 1) % R] : h->ds.VT;
                int rc = halo_exchange(h, vec, deferred ? 0. : move_dt);
                if (rc) return rc;
            }
            if (deferred && (++pending == K || s == S - 1)) {
                flush(s, pending);
                pending = 0;
            }
        }
        if (fused) {  // bring the result back to the primary buffers
            const double *vt_src = (S % R) ? h->ring.slot[S % R] : nullptr;
            if (vt_src && !vt_back) LAUNCH(h, k_pingpong_copy_back, 2 * h->dm.Nn, h->dm, h->ds, vt_src);  // (else the last ring flush has done it)
            // the element state stays in its records when the loop ends in the first buffer (an even number of launches): update()
            // works on them and the arrays follow when somebody asks (ensure_arrays); from the second buffer it is unpacked here
            if (records_end_odd) LAUNCH(h, k_unpack_state, h->dm.Ne, h->dm, h->ds, bbm, (const double *)h->ds.S4b);
        }
        return NXS_OK;
    };
    // (outside the graph: whether the records are current depends on what the caller did since the last step)
    if (fused && h->sig_loc == 0) LAUNCH(h, k_pack_state, h->dm.Ne, h->dm, h->ds, bbm, h->ds.S4a);
    if (!fused) ensure_arrays(h);
    if (fused) h->sig_loc = records_end_odd ? 0 : 1;
    h->timing.substep_launches = (resident || flow) ? 1 : pair ? S / D : halo_in_kernel ? S + (S + K - 1) / K : S * ((fused ? 1 : 2) + (mr ? 2 : 0));
    h->last_kernel = resident ? (h->res_big ? NXS_KERNEL_RESIDENT_BIG : NXS_KERNEL_RESIDENT) : flow ? NXS_KERNEL_PAIR_FLOW : pair ? (h->pair_kernel ? NXS_KERNEL_PAIR : NXS_KERNEL_MULTI) : fused ? NXS_KERNEL_FUSED : NXS_KERNEL_PER_LOOP;
    h->last_deferred = deferred; h->last_halo_in_kernel = halo_in_kernel || (resident && mr) || pair_halo;
    h->last_ring_count = (deferred && !resident && !move_in_pair) ? K : 0;
    h->last_move_in_pair = move_in_pair;
    if (!h->use_graph || (mr && !device_halo)) { int lrc = loop(); return lrc ? lrc : final_flush(); }
    if (!h->graph_valid) {
        release_graph(h);
        hipGraph_t g = nullptr;
        HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        int lrc = loop();
        if (lrc) { hipGraph_t dead = nullptr; (void)hipStreamEndCapture(h->stream, &dead); if (dead) (void)hipGraphDestroy(dead); return lrc; }
        HIPCHK(h, hipStreamEndCapture(h->stream, &g));
        hipError_t e = hipGraphInstantiate(&h->substep_graph, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) return fail(h, NXS_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
        h->graph_valid = true;
    }
    HIPCHK(h, hipGraphLaunch(h->substep_graph, h->stream));
    return final_flush();
}

int ready(nxs_dyn_handle *h) {
    if (!h) return NXS_ERR_INVALID;
    if (!h->have_mesh || !h->have_state || !h->have_forcing)
        return fail(h, NXS_ERR_STATE, "step needs set_mesh, put_state and set_forcing first");
    if (multi_rank(h) && !h->have_halo) return fail(h, NXS_ERR_STATE, "nranks>1 needs set_halo");
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return fail(h, NXS_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    return NXS_OK;
}

int explicit_solve(nxs_dyn_handle *h) {
    const bool timed = h->cur != nullptr;
    // FE.cpp:10182-10643
    const DevMesh &m = h->dm;
    if (timed) HIPCHK(h, hipEventRecord(h->cur[0], h->stream));
    {
        // (the per-step shape-coefficient records are read by k_substep_multi only: k_substep_pair rebuilds the coefficients from the staged coordinates)
        double *want = (choose_depth(h) >= 2 && !h->pair_kernel && h->shape_mem != 0) ? h->d_srec : nullptr;
        if (want != h->dw.srec) { h->dw.srec = want; release_graph(h); }  // (kernel arguments are baked into the graphs)
    }
    if (h->dp_dirty) {  // (outside any stream capture)
        if (!h->d_dp) HIPCHK(h, hipMalloc((void **)&h->d_dp, sizeof(DevParams)));
        HIPCHK(h, hipMemcpyAsync(h->d_dp, &h->dp, sizeof(DevParams), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));  // h->dp may change right after
        h->dp_dirty = false;
    }
    // the fused kernels read records only: the per-quantity work vectors (v1 kernels, debug door) are filled on request
    // (automatic: meshes that stream from HBM; on cache-resident ones the two small kernels are as fast: 10 km 24.7 vs 27.4 us)
    // (automatic: from 250 k triangles on a single rank; on a rank of several from 500 k -- measured on rank 0's partitions of the 2 km mesh, looped back: 730 k
    // 0.108 -> 0.086 ms, 366 k in the resident loop's large patches 0.046 -> 0.053, 184 k 0.033 -> 0.043: gpurun_out/r5_prepmr_ab.log)
    if (eff_fused(h) != 0 && !h->work_arrays && (h->prep_fused == 1 || (h->prep_fused < 0 && m.Ne >= (multi_rank(h) ? 500000 : 250000))) && h->prep_lds > 0 && h->dpch.prow && h->dpch.nP > 0) {
        // one launch over the sub-step kernel's patches: the elements' values reach their nodes through LDS (k_prep_fused)
        // (the open-water flags of the node blocks are lowered by the step before -- k_update's first threads, as the range flag -- and at allocation: no memset per step;
        // a step that ended without update() leaves them raised, which only costs the smoother some blocks it could have skipped)
        hipLaunchKernelGGL(k_prep_fused, dim3(h->dpch.nP), dim3(512), h->prep_lds, h->stream, m, h->dpch, h->ds, h->dw, h->dp);
        if (m.Nn > m.No) LAUNCH(h, k_prep_ghost_nodes, m.Nn - m.No, m, h->ds, h->dw, h->dp);   // several ranks: the ghost nodes' share of the nodal loops
        HIPCHK(h, hipGetLastError());
        h->last_prep = NXS_PREP_FUSED;
    } else if (eff_fused(h) != 0 && !h->work_arrays) {
        LAUNCH(h, k_prep_elements<true>, m.Ne, m, h->ds, h->dw, h->dp);
        LAUNCH(h, k_prep_nodes<true>, m.Nn, m, h->ds, h->dw, h->dp);
        h->last_prep = NXS_PREP_LEAN;
    } else {
        h->last_prep = NXS_PREP_FULL;
        LAUNCH(h, k_prep_elements<false>, m.Ne, m, h->ds, h->dw, h->dp);
        LAUNCH(h, k_prep_nodes<false>, m.Nn, m, h->ds, h->dw, h->dp);
    }
    if (timed) HIPCHK(h, hipEventRecord(h->cur[1], h->stream));
    int rc = run_substeps(h);
    if (rc) return rc;
    if (!multi_rank(h) && !h->sm_ready && !h->sm_failed && eff_fused(h) != 0) {
        // automatic: ten sweeps per launch (five launches; 10 km: smoother 43 -> 25 us per step) where ten rings fit the LDS, else five
        if (h->sm_depth > 0) { if (build_smooth_patches(h, h->sm_depth) != NXS_OK) h->sm_failed = true; }  // the smoother then runs sweep by sweep
        else if (build_smooth_patches(h, 10) != NXS_OK && build_smooth_patches(h, 5) != NXS_OK) h->sm_failed = true;
        h->tail_graph_valid = false;
    }
    if (h->dp.dynamics_type == NXS_DYN_MEVP)  // FE.cpp:10559-10573
        LAUNCH(h, k_move, m.Nn, m, h->ds, 0, m.Nn, h->dp.dtime_step);
    if (timed) HIPCHK(h, hipEventRecord(h->cur[2], h->stream));
    // Q9: 50 sweeps, hard-coded (FE.cpp:10580); + open-water mesh move.  52+ small launches: replayed from
    // a second hipGraph whenever no host work is needed inside (single rank, or device-direct halo).
    auto smooth_and_tail = [&]() -> int {
        double *a = h->ds.VT, *b = h->smooth_second ? h->smooth_second : h->ds.VT2;
        if (!h->smooth_second) LAUNCH(h, k_copy_vt, 2 * m.Nn, 2 * m.Nn, a, b);  // both buffers equal: a sweep writes the ice-free nodes only
        // single rank: D sweeps per launch on patches with D rings of nodes (k_smooth_multi) -- those of k_substep_multi where that
        // kernel runs, else node-ring patches built for the smoother alone
        const bool v3_patches = h->pair_ready && !h->pair_failed && h->dpch2.pnbr && eff_fused(h) >= 2 && h->depth_now >= 2 && !h->sm_ready;  // (only where the smoother's own patches could not be built)
        if (!multi_rank(h) && (v3_patches || h->sm_ready)) {
            const DevPatches2 &pp = v3_patches ? h->dpch2 : h->dsm;
            const size_t lds = v3_patches ? h->smooth_lds : h->sm_lds;
            const int D = pp.D, L = (50 + D - 1) / D;
            if (L & 1) std::swap(a, b);  // the buffers are equal now; end in ds.VT after L swaps
            for (int nit = 0; nit < 50; nit += D) {
                const int ks = std::min(D, 50 - nit);
                if (pp.NSmax > 256) hipLaunchKernelGGL((k_smooth_multi<512>), dim3(pp.nP), dim3(512), lds, h->stream, m, pp, h->dw, (const double *)a, b, ks);
                else hipLaunchKernelGGL((k_smooth_multi<256>), dim3(pp.nP), dim3(256), lds, h->stream, m, pp, h->dw, (const double *)a, b, ks);
                std::swap(a, b);
            }
            LAUNCH(h, k_ow_tail, m.Nn, m, h->ds, h->dw, h->dp);
            return NXS_OK;
        }
        const bool halo_in_kernel = multi_rank(h) && h->ipc_ready && !h->halo_fn && h->halo_fused && h->hf_ready && m.No > 0;
        if (halo_in_kernel) {  // which of my directions carry values the sweeps cannot change (they are sent once)
            const int ts = h->send_offsets[h->send_procs.size()];
            if (ts > 0) LAUNCH(h, k_smooth_static, ts, ts, h->d_send_index, h->d_send_seg, m, h->dw, h->ipc.my_static);
        }
        static_assert(NXS_SMOOTH_SWEEPS == 50, "Q9: FE.cpp:10580 hard-codes 50 sweeps");
        if (halo_in_kernel && h->smooth_persist != 0) {   // all 50 sweeps in ONE launch of persistent workgroups (k_smooth_persist); the result is back in `a` (50 is even)
            HaloFused hf = h->hf;
            hf.ipc = h->ipc;
            const int nblk = nblocks(m.No), G = std::min(nblk, 128);
            hipLaunchKernelGGL(k_smooth_persist, dim3(G), dim3(BLOCK), 0, h->stream, m, h->dw, a, b, hf, nblk);
            const int tr = h->recv_offsets[h->recv_procs.size()];
            hipLaunchKernelGGL(k_smooth_pull, dim3(nblocks(tr)), dim3(BLOCK), 0, h->stream, a, m.Nn, tr, h->d_recv_index, h->d_recv_seg, h->d_recv_off, h->ipc);
            LAUNCH(h, k_ow_tail, m.Nn, m, h->ds, h->dw, h->dp);
            return NXS_OK;
        }
        for (int nit = 0; nit < 50; ++nit) {
            if (halo_in_kernel) {  // updateGhosts inside the sweep; the ghosts land in the array once, after the last sweep
                HaloFused hf = h->hf;
                hf.ipc = h->ipc;
                LAUNCH(h, k_smooth_halo, m.No, m, h->dw, (const double *)a, b, hf, nit);
                if (nit == 49) {
                    const int tr = h->recv_offsets[h->recv_procs.size()];
                    hipLaunchKernelGGL(k_smooth_pull, dim3(nblocks(tr)), dim3(BLOCK), 0, h->stream, b, m.Nn, tr, h->d_recv_index,
                                       h->d_recv_seg, h->d_recv_off, h->ipc);
                }
            } else {
                LAUNCH(h, k_smooth, m.No, m, h->dw, a, b);
                if (multi_rank(h)) { int r2 = halo_exchange(h, b, 0.); if (r2) return r2; }
            }
            std::swap(a, b);
        }
        // 50 is even: the result is back in ds.VT
        LAUNCH(h, k_ow_tail, m.Nn, m, h->ds, h->dw, h->dp);
        return NXS_OK;
    };
    const bool tail_capturable = h->use_graph && (!multi_rank(h) || (h->ipc_ready && !h->halo_fn));
    if (!tail_capturable) {
        rc = smooth_and_tail();
        if (rc) return rc;
    } else {
        if (!h->tail_graph_valid) {
            if (h->tail_graph) { (void)hipGraphExecDestroy(h->tail_graph); h->tail_graph = nullptr; }
            hipGraph_t g = nullptr;
            HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            rc = smooth_and_tail();
            if (rc) { hipGraph_t dead = nullptr; (void)hipStreamEndCapture(h->stream, &dead); if (dead) (void)hipGraphDestroy(dead); return rc; }
            HIPCHK(h, hipStreamEndCapture(h->stream, &g));
            hipError_t e = hipGraphInstantiate(&h->tail_graph, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e != hipSuccess) return fail(h, NXS_ERR_HIP, "hipGraphInstantiate(tail): %s", hipGetErrorString(e));
            h->tail_graph_valid = true;
        }
        HIPCHK(h, hipGraphLaunch(h->tail_graph, h->stream));
    }
    if (timed) HIPCHK(h, hipEventRecord(h->cur[3], h->stream));
    // a refused launch (a launch configuration the device rejects, e.g. more dynamic LDS than a CU has) is reported here,
    // at the step that issued it, not as a generic error at the next synchronize
    HIPCHK(h, hipGetLastError());
    return NXS_OK;
}

}  // namespace

int nxs_dyn_explicit_solve(nxs_dyn_handle *h) try {
    int rc = ready(h);
    if (rc) return rc;
    h->cur = nullptr;
    StepScope scope(h);
    return explicit_solve(h);
} catch (...) { return dyn_caught(h, "nxs_dyn_explicit_solve"); }

int nxs_dyn_update(nxs_dyn_handle *h) try {
    int rc = ready(h);
    if (rc) return rc;
    if (h->sig_loc) LAUNCH(h, k_update<true>, h->dm.Ne, h->dm, h->ds, h->dw, h->dp);
    else LAUNCH(h, k_update<false>, h->dm.Ne, h->dm, h->ds, h->dw, h->dp);
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_update"); }

int nxs_dyn_step(nxs_dyn_handle *h) try {  // FE.cpp:8197-8214
    int rc = ready(h);
    if (rc) return rc;
    const int type = h->dp.dynamics_type;
    if (type == NXS_DYN_FREE_DRIFT) {
        LAUNCH(h, k_free_drift, h->dm.Nn, h->dm, h->ds, h->dp);
        return NXS_OK;
    }
    if (type == NXS_DYN_NO_MOTION) return NXS_OK;
    int k = -1;
    h->cur = nullptr;
    StepScope scope(h);
    if (!h->reg_key.empty()) {   // "still here" for the device's registry, at most every ten seconds (nxs_resident_registry.hpp: entries of other PID namespaces expire)
        const long long now = (long long)nxs_reg::boot_seconds();
        if (now - h->reg_touched >= 10) { nxs_reg::table_for(h->reg_key).touch((uint64_t)(uintptr_t)h); h->reg_touched = now; }
    }
    if (h->timing_enabled) {
        k = h->set_next;
        if ((rc = harvest(h, k))) return rc;
        h->cur = h->ev[k];
    }
    rc = explicit_solve(h);
    if (rc) { h->cur = nullptr; return rc; }
    if (h->sig_loc) LAUNCH(h, k_update<true>, h->dm.Ne, h->dm, h->ds, h->dw, h->dp);
    else LAUNCH(h, k_update<false>, h->dm.Ne, h->dm, h->ds, h->dw, h->dp);
    HIPCHK(h, hipGetLastError());
    if (k >= 0) {
        HIPCHK(h, hipEventRecord(h->cur[4], h->stream));
        h->set_pending[k] = true;
        h->set_next = (k + 1) % nxs_dyn_handle::NSETS;
        h->cur = nullptr;
    }
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_step"); }

int nxs_dyn_synchronize(nxs_dyn_handle *h) try {
    if (!h) return NXS_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    { int rc = resident_error(h); if (rc) return rc; }
    if (h->ipc_ready) {
        int err = 0;
        HIPCHK(h, hipMemcpyAsync(&err, h->ipc.error, sizeof err, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (err) return fail(h, NXS_ERR_COMM, "device-direct halo exchange failed (%s)", err == 2 ? "self-test mismatch" : "a neighbour's flag did not arrive within 10 s");
    }
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_synchronize"); }

int nxs_dyn_get_timing(nxs_dyn_handle *h, nxs_dyn_timing *t) try {
    if (!h || !t) return NXS_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    for (int k = 0; k < nxs_dyn_handle::NSETS; ++k) {
        int rc = harvest(h, k);
        if (rc) return rc;
    }
    const double n = h->sum_steps > 0 ? (double)h->sum_steps : 1.;
    h->timing.prep_ms = h->sum_ms[0] / n; h->timing.substeps_ms = h->sum_ms[1] / n;
    h->timing.smoother_ms = h->sum_ms[2] / n; h->timing.update_ms = h->sum_ms[3] / n;
    h->timing.total_ms = (h->sum_ms[0] + h->sum_ms[1] + h->sum_ms[2] + h->sum_ms[3]) / n;
    h->timing.ring_flush_ms = h->sum_flush_ms / n;
    h->timing.steps_averaged = h->sum_steps;
    *t = h->timing;
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_get_timing"); }

int nxs_dyn_get_step_times(nxs_dyn_handle *h, double *ms, int32_t capacity, int32_t *count) try {
    if (!h || !count || capacity < 0 || (capacity > 0 && !ms)) return NXS_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    for (int k = 0; k < nxs_dyn_handle::NSETS; ++k) { int rc = harvest(h, k); if (rc) return rc; }
    const int n = (int)std::min<size_t>(h->step_ms.size(), (size_t)capacity);
    for (int i = 0; i < n; ++i) ms[i] = h->step_ms[i];
    *count = (int32_t)h->step_ms.size();
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_get_step_times"); }

// The bytes the last step's kernels had to move (see include/nxs_dyn.h).  Every term below names the array it prices and is the width of the
// access in the kernel text (nxs_dyn_kernels.inl); `fanw` = the fan entries a node loads up front (eight; longer fans are rare).
int nxs_dyn_get_traffic_model(nxs_dyn_handle *h, nxs_dyn_traffic *t) try {
    if (!h || !t) return NXS_ERR_INVALID;
    if (!h->have_mesh) return fail(h, NXS_ERR_STATE, "get_traffic_model before set_mesh");
    std::memset(t, 0, sizeof *t);
    const double Ne = h->dm.Ne, Nn = h->dm.Nn, S = h->dp.substeps;
    const bool young = h->dp.young_cat != 0, bbm = h->dp.dynamics_type == NXS_DYN_BBM;
    t->substep_kernel = h->last_kernel;
    t->halo_in_kernel = h->last_halo_in_kernel ? 1 : 0;
    t->prep_kernel = h->last_prep;
    const double node_in = 1. /*nflags*/ + 80. /*nrec*/;
    const auto &s1 = h->sums1;
    const auto &s2 = h->sums2;
    switch (h->last_kernel) {
    case NXS_KERNEL_PAIR: case NXS_KERNEL_PAIR_FLOW: if (s2.N.size() == 3 && s2.E.size() == 2) {
        const double fanw = 16.;   // (four words of ready-made LDS indices per solved node: DevPatches2::pfan8)
        const double N0 = s2.N[0], N1 = s2.N[1], N2 = s2.N[2], E2 = s2.E[1];
        t->substeps_per_launch = 2;
        t->substep_scheme_bytes = s2.nP * 20. /*ncnt, ecnt*/ + N2 * (4. /*pnodes*/ + 16. /*VT*/ + 16. /*xy*/) + E2 * (8. /*pet*/ + 32. /*S in*/ + 48. /*erec*/)
                                  + N1 * (node_in + fanw) + s2.W * 32. /*S out*/ + N0 * 2. * 16. /*two velocity slots*/;
        t->substep_reread_bytes = s2.E1_second_round * 48. /*the constants of sub-step 1's second round, read again 3-6 us after the first time (the first round's stay in registers)*/
                                  + ((h->last_halo_in_kernel || h->last_kernel == NXS_KERNEL_PAIR_FLOW) ? N0 * (node_in + fanw) : 0.) /*(the single-rank launch keeps the own nodes' inputs in registers between its two solves)*/;
        t->substep_unique_bytes = Ne * (8. + 32. + 48. + 32.) + Nn * (4. + 16. + 16. + node_in + fanw + 32.);
        if (h->last_move_in_pair) {   // M_UM, M_UT in and out, no first velocity slot
            t->substep_scheme_bytes += N0 * (64. - 16.);
            t->substep_unique_bytes += Nn * (64. - 16.);
        }
        if (h->last_kernel == NXS_KERNEL_PAIR_FLOW) {   // ONE launch runs every pair of sub-steps of the step over the same tables
            const double pairs = std::floor(S / 2.);
            t->substeps_per_launch = (int)S;
            t->substep_scheme_bytes *= pairs; t->substep_reread_bytes *= pairs; t->substep_unique_bytes *= pairs;
        }
    } break;
    case NXS_KERNEL_MULTI: if (!s2.N.empty() && s2.E.size() + 1 == s2.N.size()) {
        const int D = (int)s2.E.size();
        const double fanw = 2. * std::min(h->dpch2.Wp, 8), shape = h->dw.srec ? 48. : 0., xy = h->dw.srec ? 0. : 16.;
        t->substeps_per_launch = D;
        t->substep_scheme_bytes = s2.nP * 4. * (2 * D + 1) + s2.N[D] * (4. + 16. + xy) + s2.E[D - 1] * (12. + 32. + 48. + shape) + s2.N[D - 1] * (node_in + fanw)
                                  + s2.W * 32. + s2.N[0] * D * 16.;
        for (int k = 1; k < D; ++k) t->substep_reread_bytes += s2.E[D - 1 - k] * (48. + shape) + s2.N[D - 1 - k] * (node_in + fanw);
        t->substep_unique_bytes = Ne * (12. + 32. + 48. + shape + 32.) + Nn * (4. + 16. + xy + node_in + fanw + D * 16.);
    } break;
    case NXS_KERNEL_FUSED: {
        const double fanw = 2. * std::min(h->dpch.Wp, 8), move = h->last_deferred || h->dp.dynamics_type == NXS_DYN_MEVP ? 0. : 64. /*M_UM, M_UT read and written*/;
        t->substeps_per_launch = 1;
        t->substep_scheme_bytes = s1.nP * 12. + s1.M * (4. + 16. + 16.) + s1.E * (8. /*pet*/ + 32. + 48.) + s1.O * (node_in + fanw + 16. /*VT out*/ + move) + s1.W * 32.;
        t->substep_unique_bytes = Ne * (8. + 32. + 48. + 32.) + Nn * (4. + 16. + 16.) + (double)h->dm.No * (node_in + fanw + 16. + move);
    } break;
    case NXS_KERNEL_RESIDENT: case NXS_KERNEL_RESIDENT_BIG: {
        // once per step: indices, state in and out, element constants, nodal inputs, coordinates, M_UM / M_UT; per sub-step: the own nodes' velocity to the
        // exchange buffer and the halo nodes' velocity back (the big kernel also re-reads the element constants and nodal inputs every sub-step, from L2)
        const double fanw = 2. * std::min(h->dpch.Wp, 8);
        const bool big = h->last_kernel == NXS_KERNEL_RESIDENT_BIG;
        t->substeps_per_launch = (int)S;
        t->substep_scheme_bytes = s1.M * (4. + 16. + 16.) + s1.E * (12. + 32. + 48.) + s1.O * (node_in + fanw + 64.) + s1.W * 32. + S * (s1.O * 16. + (s1.M - s1.O) * 16.);
        t->substep_reread_bytes = big ? (S - 1.) * (s1.E * 48. + s1.O * 80.) : 0.;
        t->substep_unique_bytes = Ne * (12. + 32. + 48. + 32.) + Nn * (4. + 16. + 16.) + (double)h->dm.No * (node_in + fanw + 64. + S * 16.);
    } break;
    case NXS_KERNEL_PER_LOOP:
        t->substeps_per_launch = 1;   // two launches: k_sigma_* (13 B indices + flags, state in and out, constants, shape coefficients, corner forces out) + k_solve_move
        t->substep_scheme_bytes = t->substep_unique_bytes = Ne * (13. + (bbm ? 64. : 48.) + 48. + 48. + 48.) + Nn * 32. + (double)h->dm.No * (4. * h->dm.W + 48. * 3. + 90. + 16. + 64.);
        break;
    default: break;
    }
    t->survey_model_bytes = (172. * Ne + 217. * Nn) * std::max(t->substeps_per_launch, 1);
    t->move_ring_slots = h->last_ring_count;
    if (h->last_ring_count > 0) t->move_ring_bytes = Nn * (1. /*nflags*/ + 32. /*UM, UT in*/ + 32. /*out*/ + 16. * h->last_ring_count /*the slots*/ + 16. /*M_VT back, last flush*/);
    {   // prep: per element 9 state fields (5 without the young category) + cohesion and healing time by the writer, out: delta_x, surface, 48-byte record;
        // per node x0, y0, UM, ssh in; VT, wind, ocean, lat, flags in; xy, tau_a, node_mass, VTM, 80-byte record out
        const double efields = (young ? 9. : 5.) * 8., ewrite = 16. + (bbm ? 16. : 8.) + 48. + (h->dw.srec ? 48. : 0.);
        const double nstage = 4. + 40., nin = 1. + 16. + 16. + 8. + 16., nout = 16. + 16. + 8. + 16. + 80.;
        if (h->last_prep == NXS_PREP_FUSED) {
            const double rows = 2. * (std::min(h->dpch.Wp, 8) + std::min(h->dpch.W1, 10));
            t->prep_scheme_bytes = s1.M * nstage + s1.E * (12. + efields) + s1.W * ewrite + s1.O * (rows + nin + nout);
            t->prep_unique_bytes = Nn * (nstage + rows + nin + nout) + Ne * (12. + efields + ewrite);
        } else if (h->last_prep == NXS_PREP_LEAN || h->last_prep == NXS_PREP_FULL) {
            // two kernels: the elements gather their corners' x0, y0, UM, ssh (40 B per node, once in the unique count, three times in the scheme) and leave a
            // 64-byte record + drag x area per element that every corner node gathers again
            const double full = h->last_prep == NXS_PREP_FULL ? 104. + 48. : 0.;
            t->prep_scheme_bytes = Ne * (13. + 3. * 40. + efields + ewrite + 64. + 8. + full) + Nn * (4. * (h->dm.W + h->dm.W1) + nin + nout + 32.) + Ne * 3. * (64. + 8.);
            t->prep_unique_bytes = Ne * (13. + efields + ewrite + 64. + 8. + full + 64. + 8.) + Nn * (40. + 4. * (h->dm.W + h->dm.W1) + nin + nout + 32.);
        }
    }
    // update(): flags, corners, area in and out, six fields (+ three young) in and out, the stress record in and out, D_del out; the corners' x0, y0, UM gathered
    t->update_bytes = Ne * (1. + 12. + 16. + (young ? 9. : 6.) * 16. + 64. + 8.) + Nn * 32.;
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_get_traffic_model"); }

int nxs_dyn_step_host(nxs_dyn_handle *h, nxs_dyn_state *s, const nxs_dyn_forcing *f) try {
    int rc;
    if (!h || !s || !f) return NXS_ERR_INVALID;
    // The literal drop-in moves the whole state and forcing over PCIe every step (2 km: 275 MB up, 187 MB down at 50-54 GB/s = 8.9 ms beside 4.7 ms of kernels).
    // With page-locked host vectors (option "pin_host") the copies are asynchronous: put, forcing, step and get become ONE queue without a synchronisation between
    // them, and without the young-ice category the young-ice trio -- which no kernel reads or writes then -- is not brought down again (the step did not change it).
    // Round 5 also tried, on a second stream: the arrays only update() reads going up beside the sub-steps, M_VT / M_UM / M_UT coming down beside update(), the arrays
    // no kernel reads going up beside the downloads.  None of it pays (13.5-14.2 ms against 13.0-13.6: `profiles/r05_experiments/r5_copies*.log`): the two directions of
    // the link slow each other down (downloads 54 -> 35 GB/s beside an upload), and the kernels run 11 % slower inside this call whatever the copies do (sub-steps 4.85 ms
    // against 4.35 back to back: the device has idled through 5 ms of uploads) -- the floor of this API on this link is the sum of its parts.
    const bool moving = h->have_mesh && (h->dp.dynamics_type == NXS_DYN_BBM || h->dp.dynamics_type == NXS_DYN_EVP || h->dp.dynamics_type == NXS_DYN_MEVP);
    const bool complete = s->VT && s->UM && s->UT && s->conc && s->thick && s->snow_thick && s->damage && s->ridge_ratio && s->sigma[0] && s->sigma[1] && s->sigma[2] &&
                          s->conc_young && s->h_young && s->hs_young && s->conc_myi && s->thick_myi && s->cohesion && s->time_relaxation_damage && s->drag_ui && s->drag_ui_young &&
                          f->wind && f->ocean && f->ssh && f->element_depth;
    if (!(h->pin_host && moving && complete)) {
        if ((rc = nxs_dyn_put_state(h, s))) return rc;
        if ((rc = nxs_dyn_set_forcing(h, f))) return rc;
        if ((rc = nxs_dyn_step(h))) return rc;
        return nxs_dyn_get_state(h, s);
    }
    HIPCHK(h, hipSetDevice(h->device));
    const size_t Nn = h->dm.Nn, Ne = h->dm.Ne, n2 = 2 * Nn * sizeof(double), n1 = Nn * sizeof(double), ne = Ne * sizeof(double);
    DevState &d = h->ds;
    struct Cp { double *dev; double *host; size_t bytes; };
    const auto H = [](const double *p) { return const_cast<double *>(p); };
    const Cp up[] = {{d.VT, s->VT, n2}, {d.UM, s->UM, n2}, {d.wind, H(f->wind), n2}, {d.ocean, H(f->ocean), n2}, {d.ssh, H(f->ssh), n1}, {d.depth, H(f->element_depth), ne},
                     {d.conc, s->conc, ne}, {d.thick, s->thick, ne}, {d.snow, s->snow_thick, ne}, {d.cohesion, H(s->cohesion), ne}, {d.theal, H(s->time_relaxation_damage), ne},
                     {d.drag_ui, H(s->drag_ui), ne}, {d.cyoung, s->conc_young, ne}, {d.hyoung, s->h_young, ne}, {d.hsyoung, s->hs_young, ne}, {d.drag_ui_young, H(s->drag_ui_young), ne},
                     {d.UT, s->UT, n2}, {d.s0, s->sigma[0], ne}, {d.s1, s->sigma[1], ne}, {d.s2, s->sigma[2], ne}, {d.damage, s->damage, ne},
                     {d.ridge, s->ridge_ratio, ne}, {d.cmyi, s->conc_myi, ne}, {d.tmyi, s->thick_myi, ne}};
    for (const Cp &c : up) pin_host_buffer(h, c.host, c.bytes);
    h->sig_loc = 0;   // M_sigma and M_damage arrive as arrays (all four: nothing of the records is kept)
    for (const Cp &c : up) HIPCHK(h, hipMemcpyAsync(c.dev, c.host, c.bytes, hipMemcpyHostToDevice, h->stream));
    h->have_state = true; h->have_forcing = true;
    if ((rc = nxs_dyn_step(h))) { (void)hipStreamSynchronize(h->stream); return rc; }
    if (h->res_ready || h->flow_ready) { HIPCHK(h, hipStreamSynchronize(h->stream)); if ((rc = resident_error(h))) return rc; }
    ensure_arrays(h);
    const Cp down[] = {{d.VT, s->VT, n2}, {d.UM, s->UM, n2}, {d.UT, s->UT, n2}, {d.conc, s->conc, ne}, {d.thick, s->thick, ne}, {d.snow, s->snow_thick, ne}, {d.damage, s->damage, ne},
                       {d.ridge, s->ridge_ratio, ne}, {d.s0, s->sigma[0], ne}, {d.s1, s->sigma[1], ne}, {d.s2, s->sigma[2], ne}, {d.cmyi, s->conc_myi, ne}, {d.tmyi, s->thick_myi, ne}};
    for (const Cp &c : down) HIPCHK(h, hipMemcpyAsync(c.host, c.dev, c.bytes, hipMemcpyDeviceToHost, h->stream));
    if (h->dp.young_cat)   // (without the category update() leaves the trio alone: the host's copy is the current one)
        for (const Cp &c : {Cp{d.cyoung, s->conc_young, ne}, Cp{d.hyoung, s->h_young, ne}, Cp{d.hsyoung, s->hs_young, ne}}) HIPCHK(h, hipMemcpyAsync(c.host, c.dev, c.bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_step_host"); }

int nxs_dyn_check_regridding(nxs_dyn_handle *h, double *min_angle, int32_t *flip, int32_t *regrid_local) try {
    if (!h) return NXS_ERR_INVALID;
    if (!h->have_mesh || !h->have_state) return fail(h, NXS_ERR_STATE, "check_regridding needs set_mesh and put_state");
    HIPCHK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(k_regrid_partials, dim3(h->n_partials), dim3(BLOCK), 0, h->stream, h->dm, h->ds, h->d_partials);
    hipLaunchKernelGGL(k_regrid_final, dim3(1), dim3(64), 0, h->stream, h->d_partials, h->n_partials, h->d_regrid);
    RegridPartial r;
    HIPCHK(h, hipMemcpyAsync(&r, h->d_regrid, sizeof r, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int fl = (r.min_jac <= 0.) && (r.max_jac >= 0.);  // FE.cpp:1838
    if (min_angle) *min_angle = r.min_angle;
    if (flip) *flip = fl;
    if (regrid_local) *regrid_local = (r.min_angle < h->params.regrid_angle) || fl;  // FE.cpp:8303-8305
    { int rc = resident_error(h); if (rc) return rc; }
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_check_regridding"); }

int nxs_dyn_check_fields_fast(nxs_dyn_handle *h, int32_t *crash_local) try {
    if (!h || !crash_local) return NXS_ERR_INVALID;
    if (!h->have_mesh || !h->have_state) return fail(h, NXS_ERR_STATE, "check_fields_fast needs set_mesh and put_state");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemsetAsync(h->d_crash, 0, sizeof(int), h->stream));
    LAUNCH(h, k_check_fields, std::max(h->dm.Ne, h->dm.Nn), h->dm, h->ds, h->dp, h->d_crash, (h->sig_loc && h->dp.dynamics_type == NXS_DYN_BBM) ? 1 : 0);
    int c = 0;
    HIPCHK(h, hipMemcpyAsync(&c, h->d_crash, sizeof c, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *crash_local = c;
    { int rc = resident_error(h); if (rc) return rc; }
    return NXS_OK;
} catch (...) { return dyn_caught(h, "nxs_dyn_check_fields_fast"); }

}  // extern "C"
