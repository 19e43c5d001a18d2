/*
 * nxs_dyn.h -- C ABI of libnxsdyn.so: the MI355X (gfx950) implementation of neXtSIM's
 * per-time-step sea-ice dynamics hot path.
 *
 * The reference (nansencenter/nextsim) has no plugin / FFI interface for this path: it is two
 * member functions of class FiniteElement that work on ~40 member vectors
 * (model/finiteelement.hpp:386-838).  This header *creates* the boundary: one entry point per
 * reference call site, taking exactly the arrays that call site reads and writes, as plain
 * pointers + sizes (no C++ types, no torch types).  "FE.cpp" = model/finiteelement.cpp.
 *
 *   reference call site                         replaced by
 *   ------------------------------------------  ------------------------------------------
 *   FiniteElement::initOptAndParam/init         nxs_dyn_create          (FE.cpp:1066-1209, 6993-6999)
 *   distributedMeshProcessing (per (re)mesh)    nxs_dyn_set_mesh        (FE.cpp:50-143, 150-271)
 *   initUpdateGhosts                            nxs_dyn_set_halo        (FE.cpp:14003-14088)
 *   ExternalData::getVector() snapshots         nxs_dyn_set_forcing     (model/externaldata.cpp:441-459)
 *   member vectors M_VT, M_conc, ...            nxs_dyn_put_state / nxs_dyn_get_state
 *   step(): UM_P=M_UM; explicitSolve(); update  nxs_dyn_step            (FE.cpp:8197-8214)
 *   explicitSolve()                             nxs_dyn_explicit_solve  (FE.cpp:10182-10643)
 *   update(UM_P)                                nxs_dyn_update          (FE.cpp:3919-4132)
 *   updateFreeDriftVelocity()                   (inside nxs_dyn_step)   (FE.cpp:10140-10176)
 *   checkRegridding()                           nxs_dyn_check_regridding(FE.cpp:8298-8309)
 *   checkFieldsFast()                           nxs_dyn_check_fields_fast (FE.cpp:14536-14655)
 *   M_surface, D_tau_a, D_tau_w, ...            nxs_dyn_get_diag
 *   BamgConvertMeshx connectivity tables        nxs_mesh_connectivity   (contrib/bamg/src/Mesh.cpp:495-865)
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (never throws across the ABI; the reference
 *     throws std::runtime_error and aborts, FE.cpp:14653).  nxs_dyn_last_error() gives the text.
 *   - all reals are fp64, all indices int32.  Nodal vectors are [u(0..Nn-1) | v(0..Nn-1)]
 *     (FE.cpp:10152-10153).  Element indices are 1-based local node ids (core/include/entities.hpp:151).
 *   - caller owns every host buffer; the library owns the device mirrors.
 *   - a handle is driven by one host thread and one HIP stream; handles are independent.
 *   - there is NO CPU fallback: without a HIP device nxs_dyn_create fails with NXS_ERR_NO_DEVICE.
 */
#ifndef NXS_DYN_H
#define NXS_DYN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NXS_DYN_ABI_VERSION 2

/* error codes */
#define NXS_OK 0
#define NXS_ERR_INVALID (-1)   /* bad argument / inconsistent sizes */
#define NXS_ERR_NO_DEVICE (-2) /* no HIP device: the product path never falls back to the CPU */
#define NXS_ERR_HIP (-3)       /* a HIP runtime call failed */
#define NXS_ERR_STATE (-4)     /* call order (e.g. step before set_mesh) */
#define NXS_ERR_COMM (-5)      /* RCCL failure */
#define NXS_ERR_NOMEM (-6)     /* host memory exhausted (a std::bad_alloc inside the library, caught at the boundary) */
#define NXS_ERR_INTERNAL (-7)  /* any other C++ exception inside the library, caught at the boundary; the text names it */

/* setup::DynamicsType, model/enums.hpp:142-149 */
enum { NXS_DYN_BBM = 0, NXS_DYN_NO_MOTION = 1, NXS_DYN_FREE_DRIFT = 2, NXS_DYN_EVP = 3, NXS_DYN_MEVP = 4 };
/* setup::BasalStressType, model/enums.hpp:82-86 */
enum { NXS_BASAL_NONE = 0, NXS_BASAL_LEMIEUX = 1 };
/* setup::IceCategoryType, model/enums.hpp:88-93 */
enum { NXS_ICECAT_CLASSIC = 0, NXS_ICECAT_YOUNG_ICE = 1 };

/* Everything explicitSolve()/update() read from vm[] or from members set in initOptAndParam/init.
 * Defaults: model/options.cpp:43,80,314-376.  compr_strength is the value AFTER the scale_coef
 * multiplication of FE.cpp:6996-6999 (cohesion arrives per element in nxs_dyn_state). */
typedef struct nxs_dyn_params {
    double dtime_step;                      /* simul.timestep [s] (FE.cpp: dtime_step) */
    int32_t substeps;                       /* dynamics.substeps (options.cpp:363) */
    int32_t dynamics_type;                  /* NXS_DYN_* */
    int32_t basal_stress_type;              /* NXS_BASAL_* */
    int32_t ice_cat_type;                   /* NXS_ICECAT_* (thermo.newice_type==4 -> YOUNG_ICE, FE.cpp:1206-1209) */
    int32_t newice_type;                    /* thermo.newice_type (FE.cpp:3943) */
    int32_t equal_ridging;                  /* age.equal_ridging (FE.cpp:3942) */
    int32_t use_young_ice_in_myi_reset;     /* age.include_young_ice (FE.cpp:3944) */
    int32_t reserved0;
    double young;                           /* dynamics.young */
    double nu0;                             /* dynamics.nu0 */
    double tan_phi;                         /* dynamics.tan_phi */
    double compr_strength;                  /* dynamics.compr_strength * scale_coef */
    double compaction_param;                /* dynamics.compaction_param */
    double undamaged_time_relaxation_sigma; /* dynamics.undamaged_time_relaxation_sigma */
    double exponent_relaxation_sigma;       /* dynamics.exponent_relaxation_sigma */
    double compression_factor;              /* dynamics.compression_factor */
    double exponent_compression_factor;     /* dynamics.exponent_compression_factor */
    double min_h;                           /* dynamics.min_h */
    double min_c;                           /* dynamics.min_c (used by update() only, FE.cpp:4063) */
    double quad_drag_coef_water;            /* dynamics.quad_drag_coef_water */
    double lin_drag_coef_water;             /* dynamics.lin_drag_coef_water (free drift only) */
    double quad_drag_coef_air;              /* <atm>_quad_drag_coef_air (free drift only; BBM uses state.drag_ui) */
    double lin_drag_coef_air;               /* dynamics.lin_drag_coef_air (free drift only) */
    double ocean_turning_angle_rad;         /* FE.cpp:1167-1172 */
    double basal_k1, basal_k2, basal_Cb, basal_u_0; /* dynamics.Lemieux_basal_* */
    double evp_e, evp_Pstar, evp_C, evp_dmin;       /* dynamics.evp.* */
    double mevp_alpha, mevp_beta;                   /* dynamics.mevp.* */
    double regrid_angle;                    /* numerics.regrid_angle [deg] */
} nxs_dyn_params;

/* The per-rank mesh as FiniteElement holds it after distributedMeshProcessing().
 * Ordering contract (core/src/gmshmesh.cpp:1165-1169, 1379-1417): nodes [0,local_ndof) are owned,
 * then ghosts; elements [0,local_nelements) are owned, then ghost elements.  Every owned node has
 * its complete element fan locally. */
typedef struct nxs_dyn_mesh {
    int32_t num_nodes;              /* M_num_nodes  (owned + ghost) */
    int32_t num_elements;           /* M_num_elements (owned + ghost) */
    int32_t local_ndof;             /* M_local_ndof (owned nodes) */
    int32_t local_nelements;        /* M_local_nelements (owned elements) */
    const int32_t *indices;         /* [3*Ne] M_elements[e].indices[k], 1-based local node ids */
    const uint8_t *ghost_nodes;     /* [3*Ne] M_elements[e].ghostNodes[k] (gmshmesh.cpp:1289-1301) */
    const double *coord_x;          /* [Nn] M_mesh.coordX()  (undisplaced) */
    const double *coord_y;          /* [Nn] M_mesh.coordY() */
    const double *lat;              /* [Nn] M_mesh.lat() in degrees (gmshmesh.cpp:1800-1824) */
    const uint8_t *mask_dirichlet;  /* [Nn] M_mask_dirichlet (false on ghosts, FE.cpp:228-234) */
    int32_t num_neumann_flags;      /* M_neumann_flags.size() */
    int32_t reserved0;
    const int32_t *neumann_flags;   /* sorted, 0-based local node ids incl. ghosts (FE.cpp:236-252) */
    /* bamgmesh tables exactly as BamgConvertMeshx leaves them (doubles, 1-based, NaN / 0 padded).
     * Either may be NULL: the library then builds an identical table with nxs_mesh_connectivity(). */
    const double *nodal_element_connectivity;  /* bamgmesh->NodalElementConnectivity [Nn*nec_width] */
    const double *nodal_connectivity;          /* bamgmesh->NodalConnectivity [Nn*nc_width], last col = count */
    int32_t nec_width;                         /* NodalElementConnectivitySize[1] */
    int32_t nc_width;                          /* NodalConnectivitySize[1] */
} nxs_dyn_mesh;

/* Halo lists of initUpdateGhosts() (FE.hpp:615-618), flattened CSR-style.
 * send_index = M_extract_local_index[q][], recv_index = M_local_ghosts_local_index[q][]; send_procs = M_recipients_proc_id,
 * recv_procs = M_local_ghosts_proc_id -- VERBATIM, as FE.cpp:14003-14088 leaves them.  On a ragged partition (Gmsh / METIS) a rank may send a node to a rank it
 * receives nothing from; the device-direct mailboxes need every link in both directions (below), so nxs_dyn_set_halo itself adds the missing direction as an
 * empty segment -- APPENDED behind the caller's neighbours, so that the caller's neighbour numbers k and its offsets (the layout of nxs_dyn_halo_fn's buffers)
 * stay what they were.  Both ranks of such a link do the same without talking to each other: one has the link in its send list, the other in its receive list. */
typedef struct nxs_dyn_halo {
    int32_t rank, nranks;
    int32_t num_send_procs;        /* M_recipients_proc_id.size() */
    int32_t num_recv_procs;        /* M_local_ghosts_proc_id.size() */
    const int32_t *send_procs;     /* [num_send_procs] */
    const int32_t *send_offsets;   /* [num_send_procs+1] into send_index */
    const int32_t *send_index;     /* 0-based local node ids */
    const int32_t *recv_procs;     /* [num_recv_procs] */
    const int32_t *recv_offsets;   /* [num_recv_procs+1] into recv_index */
    const int32_t *recv_index;     /* 0-based local (ghost) node ids */
} nxs_dyn_halo;

/* Prognostic state touched by the path (FE.hpp:712-746).  put: host -> device, get: device -> host.
 * A NULL member is skipped on get; on put every non-const member must be non-NULL. */
typedef struct nxs_dyn_state {
    double *VT, *UM, *UT;                 /* [2*Nn] M_VT, M_UM, M_UT */
    double *conc, *thick, *snow_thick;    /* [Ne] M_conc, M_thick, M_snow_thick */
    double *damage, *ridge_ratio;         /* [Ne] M_damage, M_ridge_ratio */
    double *sigma[3];                     /* [Ne] M_sigma[0..2] = s11, s22, s12 */
    double *conc_young, *h_young, *hs_young; /* [Ne] young-ice category */
    double *conc_myi, *thick_myi;         /* [Ne] multi-year ice */
    /* inputs only (written by thermo / calcCohesion in the reference) */
    const double *cohesion;               /* [Ne] M_Cohesion (FE.cpp:3909-3914) */
    const double *time_relaxation_damage; /* [Ne] M_time_relaxation_damage [s] */
    const double *drag_ui;                /* [Ne] M_drag_ui */
    const double *drag_ui_young;          /* [Ne] M_drag_ui_young */
} nxs_dyn_state;

/* Flat per-step forcing snapshot (ExternalData::getVector semantics). */
typedef struct nxs_dyn_forcing {
    const double *wind;          /* [2*Nn] M_wind */
    const double *ocean;         /* [2*Nn] M_ocean */
    const double *ssh;           /* [Nn]   M_ssh */
    const double *element_depth; /* [Ne]   M_element_depth */
} nxs_dyn_forcing;

/* Side outputs other parts of the model consume (moorings, coupler, exporter). NULL = skip. */
typedef struct nxs_dyn_diag {
    double *surface;            /* [Ne] M_surface */
    double *delta_x;            /* [Ne] M_delta_x */
    double *D_tau_a;            /* [2*Nn] */
    double *D_tau_w;            /* [2*Nn] */
    double *D_del_ci_ridge_myi; /* [Ne] */
} nxs_dyn_diag;

/* updateIceDiagnostics() (FE.cpp:7860-7905), the element diagnostics checkOutputs() / exportResults() feed to the Moorings and the Exporter:
 * totals over the ice categories, the principal stresses, the divergence of the velocity on the displaced mesh.  NULL = skip.
 * (D_tsurf mixes thermodynamic variables -- M_tice, M_tsurf_young, M_sst -- that never cross this boundary: it stays with the host;
 * D_dmean / D_dmax are 0 without OASIS, FE.cpp:7903-7904.) */
typedef struct nxs_dyn_ice_diag {
    double *D_conc;         /* [Ne] M_conc (+ M_conc_young with the young-ice category) */
    double *D_thick;        /* [Ne] M_thick (+ M_h_young) */
    double *D_snow_thick;   /* [Ne] M_snow_thick (+ M_hs_young) */
    double *D_sigma0;       /* [Ne] (sigma11 + sigma22) / 2 */
    double *D_sigma1;       /* [Ne] hypot((sigma11 - sigma22) / 2, sigma12) */
    double *D_divergence;   /* [Ne] sum_j dxN_j u_j + dyN_j v_j with shapeCoeff on x0 + M_UM */
} nxs_dyn_ice_diag;
#define NXS_ICE_DIAG_FIELDS 6   /* order of the interleaved device rows: D_conc, D_thick, D_snow_thick, D_sigma0, D_sigma1, D_divergence */

/* Per-phase device time of nxs_dyn_step, averaged over the steps since the last "timing_reset"
 * option (HIP events on the handle's stream; steps stay asynchronous), named after the reference's
 * Timer rows (FE.cpp:8197-8221, 10217-10642). Milliseconds per step. */
typedef struct nxs_dyn_timing {
    double prep_ms;        /* "prep elements" + "prep nodes" */
    double substeps_ms;    /* "sub-time stepping" (all sub-steps, halo included) */
    double smoother_ms;    /* "OW smoother" (+ open-water mesh move) */
    double update_ms;      /* "update" */
    double total_ms;       /* "dynamics" */
    int32_t substep_launches; /* kernel launches inside substeps_ms */
    int32_t steps_averaged;   /* number of steps the averages cover */
    double ring_flush_ms;     /* (ABI 2) the part of substeps_ms spent in the step's last k_move_ring -- the deferred M_UM / M_UT += dt * M_VT of FE.cpp:10543-10550
                               * for all the sub-steps the velocity ring holds; 0 where the mesh move is inside the sub-step kernel */
} nxs_dyn_timing;

typedef struct nxs_dyn_handle nxs_dyn_handle;

#if defined(__GNUC__)
#define NXS_API __attribute__((visibility("default")))
#else
#define NXS_API
#endif

NXS_API int nxs_dyn_abi_version(void);
NXS_API const char *nxs_dyn_last_error(const nxs_dyn_handle *h); /* h may be NULL: last create() error */

NXS_API int nxs_dyn_default_params(nxs_dyn_params *p); /* model/options.cpp defaults, bbm */
/* The physical constants compiled into the kernels, in the order NXS_CONST_* names them: physical::rhoi, rhow, rhos, rhoa, gravity, omega
 * (model/constants.hpp:56-87), PI (contrib/bamg/include/OppositeAngle.h:4), days_in_sec (model/finiteelement.hpp:549).  Host only; lets a
 * caller (and tests/test_reference_constants.py, against values printed by a translation unit that includes the reference's headers) check
 * that library and model agree before the first step. */
enum { NXS_CONST_RHOI = 0, NXS_CONST_RHOW, NXS_CONST_RHOS, NXS_CONST_RHOA, NXS_CONST_GRAVITY, NXS_CONST_OMEGA, NXS_CONST_PI, NXS_CONST_DAYS_IN_SEC, NXS_CONST_COUNT };
NXS_API int nxs_dyn_physical_constants(double *out, int32_t count);
/* Self-test of one arithmetic short-cut of the sub-step kernels (no reference call site).  The six shape coefficients of a triangle (FE.cpp:1951-1964) are six
 * quotients by ONE divisor, the Jacobian; where every operand of a step lies in a range the prep kernels check once per step (|coordinate| zero or in
 * [1e-100, 1e100], |Jacobian| in [1e-100, 1e100]) the kernels refine the divisor's reciprocal once and finish every quotient with the three operations the
 * compiler's own division sequence ends in -- the same instructions on the same operands, hence the same bits as six divisions; outside that range they divide.
 * This entry point computes n pseudo-random sextuples both ways on `device` and returns in *mismatches the number of quotients whose bits differ (0 expected):
 * mode 0 = triangles of the size and position meshes have, mode 1 = operands spread over the whole admitted range, zeros among the numerators.
 * mode 2 tests the second short-cut: the strain-rate and stress-increment sums of updateSigmaDamage (FE.cpp:4167-4176, 4204-4210) are formed without the terms
 * that are products with a LITERAL zero of M_B0T / M_Dunit (adding +-0 to a sum that cannot be -0 returns it unchanged); n random operand sets, zeros of both signs
 * among the velocities and stresses, computed with and without those terms: *mismatches = values whose bits differ (0 expected). */
NXS_API int nxs_dyn_selftest_quotients(int32_t device, int64_t n, uint64_t seed, int32_t mode, int64_t *mismatches);
NXS_API int nxs_dyn_create(const nxs_dyn_params *p, int device, nxs_dyn_handle **out);
NXS_API int nxs_dyn_destroy(nxs_dyn_handle *h);
NXS_API int nxs_dyn_set_params(nxs_dyn_handle *h, const nxs_dyn_params *p);

NXS_API int nxs_dyn_set_mesh(nxs_dyn_handle *h, const nxs_dyn_mesh *m);
NXS_API int nxs_dyn_set_halo(nxs_dyn_handle *h, const nxs_dyn_halo *halo);
/* RCCL communicator for the halo exchange: every rank passes the same 128-byte ncclUniqueId
 * (nxs_dyn_comm_unique_id on rank 0, broadcast by the host launcher). */
NXS_API int nxs_dyn_comm_unique_id(void *id128);
NXS_API int nxs_dyn_comm_init(nxs_dyn_handle *h, const void *id128, int rank, int nranks);
/* One exchange of coded payloads through that communicator (collective): a grouped ncclSend/ncclRecv of the rank to itself and,
 * when halo lists are set, updateGhosts' grouped send/recv with checked contents.  *errors = wrong values received. */
NXS_API int nxs_dyn_comm_selftest(nxs_dyn_handle *h, int32_t *errors);

/* Device-direct transport: updateGhosts through peer-mapped mailboxes (stores over xGMI, flags, no RCCL
 * launch, replayable from a hipGraph).  Collective setup driven by the host launcher:
 *   1. every rank: nxs_dyn_ipc_export(h, blob)                 -> NXS_IPC_BLOB_BYTES bytes to publish
 *   2. the launcher all-gathers the blobs and each rank's receive lists
 *   3. every rank: nxs_dyn_ipc_connect(h, blobs_of_my_send_neighbours, ...)
 *   4. every rank: nxs_dyn_ipc_selftest(h, rounds, &errors)    -> use it only if errors == 0 everywhere
 * Takes precedence over RCCL once connected; the host-staged callback below overrides both. */
/* The mailbox must be uncached device memory (the in-kernel exchange takes no acquire after its flag wait): ipc_export fails when
 * the runtime refuses it, and the caller stays on RCCL or its own communicator.  Neighbour handles may live in other processes
 * (hipIpc) or in this one (a host that drives several GPUs from one process); ipc_connect checks the tables it is given against
 * what each neighbour published, and a second connect replaces the first.  The self-test pushes checked payloads through every
 * link with both publishing protocols the step can use (one release per block / one per launch).
 * NEIGHBOURS IN BOTH DIRECTIONS: a mailbox has two buffers per link, which is safe because "a neighbour cannot start exchange x + 2 before it has received my
 * exchange x + 1, which I send only after my pull of exchange x" -- a hand-shake that needs every rank I send to to send to me as well.  A ragged partition can
 * send a node to a rank it receives nothing from: nxs_dyn_set_halo adds that direction itself as an empty segment (its flag is still raised and waited for: that
 * is the hand-shake), on both ranks, without communication.  The LOW-LEVEL nxs_dyn_ipc_connect below takes tables for the caller's own send neighbours only, so on
 * such a partition it returns NXS_ERR_INVALID and names the rank (round 4: found as one wrong payload in the self-test of a 4-rank mosaic): use the record form,
 * which finds the added direction in the neighbours' records.  RCCL and the host-staged transport do not need the pairing; they skip segments without nodes. */
#define NXS_IPC_BLOB_BYTES 128
NXS_API int nxs_dyn_ipc_export(nxs_dyn_handle *h, void *blob);
NXS_API int nxs_dyn_ipc_connect(nxs_dyn_handle *h, const void *blobs, const int32_t *peer_recv_offset,
                                const int32_t *peer_recv_total, const int32_t *peer_flag_slot);
NXS_API int nxs_dyn_ipc_selftest(nxs_dyn_handle *h, int rounds, int32_t *errors);
/* The same set-up WITHOUT bookkeeping on the caller's side (round 5) -- what INTEGRATION.md section 3 shows, and the only form that works on a partition with
 * one-directional neighbours (the direction nxs_dyn_set_halo added is not in any list the caller holds):
 *   1. every rank: nxs_dyn_ipc_record_bytes(h, &n)         the size of its record: the blob of nxs_dyn_ipc_export + its receive lists as the library holds them
 *   2. the launcher: stride = MPI_Allreduce(MAX) of n
 *   3. every rank: nxs_dyn_ipc_export_record(h, rec, stride)   (exports the mailbox, like nxs_dyn_ipc_export; the tail of rec is zeroed)
 *   4. the launcher: MPI_Allgather of the records, `stride` bytes each, in rank order
 *   5. every rank: nxs_dyn_ipc_connect_records(h, records, stride, nranks)   finds its segment, the totals and its flag slot in every neighbour's record itself
 *   6. every rank: nxs_dyn_ipc_selftest
 * connect_records returns NXS_ERR_INVALID when the records disagree with this rank's lists (a neighbour that does not list this rank, a segment of another length). */
NXS_API int nxs_dyn_ipc_record_bytes(nxs_dyn_handle *h, int32_t *bytes);
NXS_API int nxs_dyn_ipc_export_record(nxs_dyn_handle *h, void *record, int32_t capacity);
NXS_API int nxs_dyn_ipc_connect_records(nxs_dyn_handle *h, const void *records, int64_t stride, int32_t nranks);
/* Profiling aid (no reference call site): the mailboxes of this handle connected to THEMSELVES, so that a rank's partition can be stepped alone on a device with the
 * exchange inside its kernels -- every flag a kernel waits for is raised by the rank's own launches (rocprofv3's counter collection serialises the kernels of a
 * device: two ranks whose kernels wait for each other cannot be profiled together, a looped-back rank can; bench.py's roofline.traffic at N > 1 and its
 * aux_partition_floor).  The ghosts receive meaningless velocities -- results are NOT the model's --, the launches walk the same tables and move the same bytes.
 * NXS_ERR_INVALID when the lists do not allow it (no neighbour). */
NXS_API int nxs_dyn_ipc_loopback(nxs_dyn_handle *h);

/* Test door "ipc_delay" (option of nxs_dyn_set_option; compiled in, off by default, no reference call site): ONE named rank sleeps at ONE named point of the exchange
 * protocols -- value = rank << 16 | point << 8 | units, a unit = 10 us, units 1..255; 0 = off.  The blocking send / recv of the reference (FE.cpp:13981-13985) cannot
 * reorder; the flag protocols of the device-direct transport can, in windows a few microseconds wide that a bitwise test only sees when the race is lost.  A delay at
 * the right point makes the race lose every time: tests/test_gpu_protocol_delays.py walks (variant x point x delayed rank) on ragged 3- and 4-rank partitions and
 * requires the bits of the undelayed separate kernels -- and, with the test door "halo_one_directional" (= 1 BEFORE nxs_dyn_set_halo: the lists are taken as given, no
 * direction is added, nxs_dyn_ipc_connect does not refuse), shows round 4's defect fail deterministically. */
enum { NXS_DELAY_NONE = 0,
       NXS_DELAY_PULL_READ = 1,          /* k_halo_pull: flags seen, before the mailbox half is read */
       NXS_DELAY_PUSH_STORE = 2,         /* k_halo_push: before the stores into the neighbours' mailboxes */
       NXS_DELAY_PUSH_FLAG = 3,          /* k_halo_push: stores drained, before the flags are raised */
       NXS_DELAY_STAGE_READ = 4,         /* k_substep_fused / k_substep_pair<HALO> / k_substep_resident*: flags seen, before the ghosts are staged from the mailbox half */
       NXS_DELAY_SEND_STORE = 5,         /* the same kernels: before the (first) store of the sent nodes */
       NXS_DELAY_PAIR_SECOND_STORE = 6,  /* k_substep_pair<HALO>: before the second sub-step's store */
       NXS_DELAY_PUBLISH_FLAG = 7,       /* the same kernels: stores drained, before the (first) flag is raised */
       NXS_DELAY_PAIR_MID_READ = 8,      /* k_substep_pair<HALO>: flags x + 1 seen, before the ghosts of N_1 are read */
       NXS_DELAY_SMOOTH_READ = 9,        /* k_smooth_halo: flags seen, before the ghosts' slot is read */
       NXS_DELAY_SMOOTH_STORE = 10,      /* k_smooth_halo: before a sweep's stores */
       NXS_DELAY_SMOOTH_FLAG = 11,       /* k_smooth_halo: before a sweep's publication */
       NXS_DELAY_SMOOTH_PULL_READ = 12,  /* k_smooth_pull: flags seen, before the last slot is read */
       NXS_DELAY_PAIR_SECOND_FLAG = 13,  /* k_substep_pair<HALO>: before the second publication */
       NXS_DELAY_POINTS = 14 };

/* Alternative transport: host-staged exchange through the CALLER's communicator -- the literal
 * M_comm.send / M_comm.recv of FE.cpp:13981-13985.  send holds, per send neighbour k, 2*n_k doubles
 * [u-block | v-block] at offset 2*send_offsets[k]; recv is laid out the same way from recv_offsets.
 * fn must fill recv and return 0.  Takes precedence over RCCL when set; NULL unsets it. */
typedef int (*nxs_dyn_halo_fn)(void *ctx, const double *send, double *recv);
NXS_API int nxs_dyn_set_halo_exchange_fn(nxs_dyn_handle *h, nxs_dyn_halo_fn fn, void *ctx);

/* Host <-> device copies of the prognostic arrays.  The first put after nxs_dyn_set_mesh must bring every member; afterwards a
 * NULL member means "the device copy is current" (put) / "not wanted" (get), so a host whose thermodynamics only touched
 * concentration, thickness and snow moves only those across PCIe instead of the whole state every step. */
NXS_API int nxs_dyn_put_state(nxs_dyn_handle *h, const nxs_dyn_state *s);
NXS_API int nxs_dyn_get_state(nxs_dyn_handle *h, nxs_dyn_state *s);
NXS_API int nxs_dyn_set_forcing(nxs_dyn_handle *h, const nxs_dyn_forcing *f);
/* Forcing that is interpolated linearly in time (ExternalData::get, model/externaldata.cpp:360-401) without a host->device
 * copy per step: the two snapshots of a forcing interval (dataset->variables[].interpolated_data[0] and [1] of M_wind,
 * M_ocean, M_ssh; element_depth of f0, a constant dataset) become resident once per interval, and every step only passes
 *   fcoeff[0] = |t - ftime_range[1]| / fdt,  fcoeff[1] = |t - ftime_range[0]| / fdt,
 * M_factor (spin-up ramp, Q10) and M_bias_correction of {wind, ocean, ssh} (NULL = 1 and 0); the device evaluates
 *   M_factor*(fcoeff[0]*d0[i] + fcoeff[1]*d1[i]) + M_bias_correction      -- the reference's expression, same bits. */
NXS_API int nxs_dyn_set_forcing_pair(nxs_dyn_handle *h, const nxs_dyn_forcing *f0, const nxs_dyn_forcing *f1);
NXS_API int nxs_dyn_set_forcing_time(nxs_dyn_handle *h, double fcoeff0, double fcoeff1, const double factor[3], const double bias[3]);
NXS_API int nxs_dyn_get_diag(nxs_dyn_handle *h, nxs_dyn_diag *d);
/* updateIceDiagnostics() on the device-resident state.  d (may be NULL): host arrays to fill.  device_rows (may be NULL): receives a DEVICE
 * pointer to the same diagnostics as [Ne][NXS_ICE_DIAG_FIELDS] interleaved rows (library-owned, valid until the next call on this handle
 * or nxs_dyn_set_mesh) -- the layout nxs_interp_mesh_to_grid_device samples, so a Moorings record needs no round trip of the state. */
NXS_API int nxs_dyn_ice_diagnostics(nxs_dyn_handle *h, nxs_dyn_ice_diag *d, const double **device_rows);

/* One dynamics step on the device-resident state: FE.cpp:8197-8214.  Asynchronous on the
 * handle's stream; nxs_dyn_synchronize() waits for it. */
NXS_API int nxs_dyn_step(nxs_dyn_handle *h);
NXS_API int nxs_dyn_explicit_solve(nxs_dyn_handle *h);
NXS_API int nxs_dyn_update(nxs_dyn_handle *h);
NXS_API int nxs_dyn_synchronize(nxs_dyn_handle *h);
/* Literal drop-in for the three lines of step(): put_state + set_forcing + step + get_state. */
NXS_API int nxs_dyn_step_host(nxs_dyn_handle *h, nxs_dyn_state *s, const nxs_dyn_forcing *f);

/* checkRegridding(): local minimum angle [deg] and flip test; the cross-rank reduction
 * (FE.cpp:8306, 1812) is left to the caller's communicator. */
NXS_API int nxs_dyn_check_regridding(nxs_dyn_handle *h, double *min_angle, int32_t *flip, int32_t *regrid_local);
/* checkFieldsFast(): crash_local != 0 when a field is out of range / NaN (FE.cpp:14541-14629). */
NXS_API int nxs_dyn_check_fields_fast(nxs_dyn_handle *h, int32_t *crash_local);

NXS_API int nxs_dyn_get_timing(nxs_dyn_handle *h, nxs_dyn_timing *t);
/* Device time [ms] of every single nxs_dyn_step since the last "timing_reset" (HIP events on the handle's stream, steps stay asynchronous; at most 4096
 * are kept): ms[0 .. min(*count, capacity)) are filled, *count = steps recorded.  SURVEY 8d quotes the metric on the MEDIAN step. */
NXS_API int nxs_dyn_get_step_times(nxs_dyn_handle *h, double *ms, int32_t capacity, int32_t *count);

/* The bytes the kernels of the LAST step had to move, per launch, computed on the host from the tables the launches really walk (the patch
 * lists of nxs_dyn_set_mesh) -- the roofline model bench.py prices the event-timed launches with (SURVEY 8d asks for algorithmic bytes per
 * launch; with temporal blocking its per-sub-step figure is no longer a lower bound, this one is):
 *   *_scheme_bytes  every list a workgroup reads, once, plus what it writes, summed over the workgroups of one launch: the halo rings of the
 *                   blocking scheme count (several workgroups must read them), a workgroup's SECOND read of a record does not
 *   *_reread_bytes  those second reads (the caches may or may not serve them: the hardware counters say)
 *   *_unique_bytes  every array entry the launch touches, once: the floor of ANY kernel that advances this many sub-steps per launch
 * so unique <= scheme <= scheme + reread, and counted HBM traffic (rocprofv3 --pmc) lands between unique and scheme + reread (below scheme where the L2 serves
 * rings that neighbouring workgroups share). */
enum { NXS_KERNEL_NONE = 0, NXS_KERNEL_PER_LOOP = 1 /* k_sigma_* + k_solve_move */, NXS_KERNEL_FUSED = 2 /* k_substep_fused */,
       NXS_KERNEL_MULTI = 3 /* k_substep_multi */, NXS_KERNEL_PAIR = 4 /* k_substep_pair */, NXS_KERNEL_RESIDENT = 5 /* k_substep_resident */,
       NXS_KERNEL_RESIDENT_BIG = 6 /* k_substep_resident_big */, NXS_KERNEL_PAIR_FLOW = 7 /* k_substep_flow: k_substep_pair's patches, one data-flow launch per step */ };
enum { NXS_PREP_NONE = 0, NXS_PREP_FULL = 1 /* k_prep_elements + k_prep_nodes with the work arrays */, NXS_PREP_LEAN = 2 /* the same, records only */,
       NXS_PREP_FUSED = 3 /* k_prep_fused */ };
typedef struct nxs_dyn_traffic {
    int32_t substep_kernel;          /* NXS_KERNEL_*: the kernel the sub-step loop of the last step ran on */
    int32_t substeps_per_launch;     /* sub-steps one launch of it advances (the resident kernels: all of them) */
    int32_t halo_in_kernel;          /* != 0: the launch also performs updateGhosts through the device-direct mailboxes */
    int32_t prep_kernel;             /* NXS_PREP_* */
    double substep_scheme_bytes, substep_reread_bytes, substep_unique_bytes;   /* per launch of the sub-step kernel */
    double survey_model_bytes;       /* SURVEY 8d's 172 B per element + 217 B per node, x substeps_per_launch (the "algorithmic equivalent") */
    int32_t move_ring_slots;         /* velocity slots one k_move_ring launch applies (0: the mesh move is inside the sub-step kernel) */
    int32_t reserved0;
    double move_ring_bytes;          /* per k_move_ring launch */
    double prep_scheme_bytes, prep_unique_bytes;   /* per step: the prep kernel(s) */
    double update_bytes;             /* per step: k_update */
} nxs_dyn_traffic;
NXS_API int nxs_dyn_get_traffic_model(nxs_dyn_handle *h, nxs_dyn_traffic *t);
/* Options (none changes a bit of the results; the tests assert that):
 *   "prepare"      1 = build NOW what the first step would build lazily (tables of the exchange inside the kernels, of the resident loop): hosts that
 *                  run several ranks of one process on ONE device call it before their start barrier (building frees device memory, which waits for the
 *                  whole device -- including a neighbour rank's kernel that is already waiting for this rank); harmless anywhere else
 *   "graph"        1 = sub-step loop replayed from a hipGraph (default); 0 = plain launches
 *   "timing"       1 = record the per-phase events (default); "timing_reset": zero the averages
 *   "fused"        3 = automatic (default): on a single rank several sub-steps per launch -- four on meshes small enough for one patch per CU
 *                  (<= 256 nodes each, patches with that many rings of halo), two with the stresses in registers on larger ones (see
 *                  "pair_regs") --, one patch kernel per sub-step otherwise (several ranks, mEVP, a sub-step count the depth does not divide);
 *                  2 = several sub-steps per launch wherever possible (single rank, not mEVP, a depth that divides the
 *                  number of sub-steps); 1 = one patch kernel per sub-step; 0 = one kernel per reference loop;
 *                  4 = the whole sub-step loop in ONE resident launch whose workgroups wait for their neighbouring patches only
 *                  (one rank, or several with the device-direct mailboxes and "halo_fused" 1: the exchange between ranks then happens
 *                  inside that launch too; not mEVP; every workgroup resident at once -- checked, also against the other resident grids this
 *                  process runs on the device, else as 1): for a device the handle has to itself.  Partitions of up to ~200 k triangles run two
 *                  workgroups per CU with one element per thread (a rank of eight of a 1.5 M-triangle mesh: 0.85 instead of 1.3 ms per step),
 *                  partitions of 200 k - 400 k ONE workgroup per CU with four elements and two nodes per thread (a rank of four: 1.3 instead of
 *                  2.2 ms).  The option decides how the mesh is cut: setting or clearing it on a live mesh cuts the mesh again (same bits)
 *   "resident_dryrun"  (an action, not a setting) builds the tables of the resident loop for the mesh and halo lists set so far -- no transport,
 *                  no neighbours needed -- and fails with NXS_ERR_INVALID when this partition cannot run it (a patch with more elements than
 *                  threads, more than one round of workgroups, LDS, > 24 neighbouring patches): a partition can be checked on its own
 *   "resident_wide"  with "fused" 4 on several ranks, a device that is this handle's alone and a partition that one workgroup per CU covers:
 *                  1 = the build of the resident kernel compiled for two waves per SIMD (no register limit to speak of: 15 % faster there);
 *                  default 0, because one such workgroup fills a CU and ranks sharing a device would no longer be resident side by side
 *   "resident_overlap"  with "fused" 4: 1 = the interior elements of every patch (no corner is a halo node) run one exchange ahead -- their next
 *                  update is computed while the exchange of the sub-step is awaited; 0 = never; -1 (default) = where it is known to pay: the
 *                  large patches of a 200 k - 400 k partition (one workgroup per CU, nothing else fills its wait: 4 % faster), not the
 *                  one-element-per-thread patches (inside one GPU the wait is filled by the other workgroup of the CU; between GPUs it is a
 *                  round trip over xGMI: bench.py times both and keeps one).  The same bits either way
 *   "band_patch_nodes"  several ranks with "fused" 4 (one element per thread): the nodes this rank sends -- its own nodes along the partition boundary --
 *                  are cut into small patches of their own: a boundary patch pays the exchange between ranks in every sub-step of the resident
 *                  loop, so it gets a shorter compute phase (two ranks of 87 k triangles rehearsed on one GPU: 0.985 -> 0.89-0.91 ms of
 *                  sub-steps).  16..512 nodes; 0 = off; -1 (default) = 48.  The cut does not change a bit of the results
 *   "prep_fused"   prep elements + prep nodes (FE.cpp:10235-10416) as ONE launch over the sub-step kernel's node patches, the elements'
 *                  values reaching their nodes through LDS (k_prep_fused: 2 km mesh 203 -> 113 us per step, the same bits); on a rank of several over the
 *                  patches of its own nodes, the ghost nodes' share of the nodal loops in a small pass of its own (k_prep_ghost_nodes): -1 (default) =
 *                  on meshes of 250 k triangles and more (a rank of several: 500 k), 0 = never, 1 = wherever its tables exist
 *   "smooth_depth" sweeps of the open-water smoother per launch on its own node-ring patches (single rank): 5, 10 or 25; 0 = automatic
 *                  (10 where ten rings of neighbours fit the LDS, else 5; sweep by sweep where neither fits)
 *   "substeps_per_launch"  depth of that temporal blocking, 2..8; 0 = automatic (4, lowered until it divides the count)
 *   "patch_nodes"  own nodes per patch of the fused kernel, 64..1024; 0 = automatic (whole rounds of resident workgroups)
 *   "pair_nodes"   the same for the several-sub-steps kernels, 16..1024; 0 = automatic
 *   "pair_regs"    not mEVP, an even number of sub-steps: TWO sub-steps per launch with the stresses between them in registers and two
 *                  workgroups per CU (k_substep_pair: stress, damage, element constants and nodal inputs cross HBM once per two sub-steps; 2 km
 *                  mesh 6.62 -> 5.5 ms per step, the same bits).  Several ranks (device-direct mailboxes, "halo_fused" 1): both updateGhosts of a launch
 *                  happen inside it -- the patches along the partition boundary store their first velocities into the neighbours' mailboxes, wait for
 *                  the neighbours' and go on, every other patch runs the single-rank body.  -1 (default) = on meshes / partitions of more than 65 k nodes
 *                  (smaller single-rank meshes run four sub-steps per launch, one patch per CU), 0 = never, 1 = wherever it can run
 *   "pair_hilbert" single rank: 1 = the two-ring patches are cut along a Hilbert curve even where the caller's numbering has locality (experiment: at 2 km the rings get
 *                  THICKER -- nodes x 1.27 / 1.54 instead of x 1.25 / 1.51 -- and the own nodes of a patch are no longer contiguous: 4.51-4.77 against 4.30-4.42 ms of
 *                  sub-steps); 0 (default) = only where the numbering has none
 *   "pair_move"    single rank, k_substep_pair with 512 threads, no "um_ring": the launch applies the mesh move of its two sub-steps (FE.cpp:10543-10550) to its
 *                  own nodes itself -- M_UM and M_UT read and written once per launch, the additions in the order the deferred flush makes them -- so the
 *                  step needs no ring of one velocity buffer per sub-step (120 x 11.7 MB at 2 km) and no k_move_ring: 2 km 5.27 -> 5.17 ms per step, the
 *                  same bits.  -1 (default) = wherever that kernel runs on one rank, 0 = never (one flush per step from the ring), 1 = as -1
 *   "pair_flow"    single rank, where k_substep_pair runs with 512 threads and the whole step fits the velocity ring: 1 = every pair of sub-steps of a step in ONE
 *                  data-flow launch (k_substep_flow) whose workgroups take (pair, patch) tasks from queues and wait for the patches around theirs only
 *                  (per-patch counters; what patches hand each other is stored write-through and read past the L1) -- no launch drains the device 60
 *                  times a step; the same bits.  MEASURED SLOWER at 2 km (7.5 against 5.05 ms of sub-steps: the write-through traffic and the
 *                  software hand-over between tasks cost more than the part-empty last round of a launch), so 0 / -1 (default) = one launch per pair
 *   "pair_threads" single rank: threads of a k_substep_pair workgroup, 512 (default: two workgroups per CU, patches of ~430 nodes at 2 km) or 256 (four per CU,
 *                  patches of ~180 nodes: measured slower, 5.67 against 5.30 ms of sub-steps at 2 km -- the thicker rings cost more than four independent
 *                  workgroups per CU hide); set before nxs_dyn_set_mesh or the next step cuts the mesh again
 *   "um_ring"      apply M_UM/M_UT += dt*M_VT every n sub-steps from a ring of velocity buffers, 1..128;
 *                  0 = automatic (once per step on meshes that stream from HBM, every sub-step on cache-resident ones)
 *   "nt_mask"      non-temporal access classes of the fused kernel (1 sigma/damage, 2 UM/UT, 4 element constants); -1 = automatic
 *                  (default): 3 from 1 M local triangles on, where a sub-step streams more than the Infinity Cache holds, 0 below
 *   "pin_host"     1 = page-lock the caller's state / forcing vectors the first time they are seen (hipHostRegister), so the
 *                  per-step copies of a host that keeps its thermodynamics on the CPU run at PCIe speed; registrations are
 *                  dropped at set_mesh / destroy / pin_host 0.  Default 0: the library does not touch the caller's pages.
 *   "shape_mem"    the several-sub-steps kernel reads M_shape_coeff from a per-step 48-byte record (1, and -1 = automatic, the default)
 *                  or rebuilds it from the staged frozen coordinates every sub-step as the one-sub-step kernel always does (0)
 *   "trace_branches"  see nxs_dyn_get_branch_trace
 *   "work_arrays"  1 = the prep kernels also fill the one-array-per-quantity work vectors (M_shape_coeff, element mass, the per-step
 *                  element constants, rlmass, C_bu, grad_ssh, fcor) that only the fused = 0 kernels and nxs_dyn_debug_array read;
 *                  default 0: with fused != 0 the step writes its records only
 *   "halo_fused"   device-direct transport only: 1 = updateGhosts inside the fused sub-step kernel (default),
 *                  0 = separate push / pull kernels
 *   "resident_release"  "fused" 4 on several ranks: 1 (default) = a system-scope release fence in front of every sub-step's flags; 0 = none -- the flags publish
 *                  mailbox stores of other workgroups that were written through and drained before those workgroups took their tickets, which the publishing lane's
 *                  fence does not reach: 0.65 us per sub-step (a rank of eight of the 2 km mesh, looped back: 1.03 -> 0.95 ms of sub-steps).  The same bits on one
 *                  device; no run on several devices has told the two apart yet, hence the default; bench.py tries both and keeps 0 only if bit-identical
 *   "smooth_persist"  several ranks, device-direct mailboxes with the exchange inside the kernels: the 50 sweeps of the open-water smoother (FE.cpp:10578-10611) as ONE launch
 *                  of at most 128 persistent workgroups that meet at a barrier of their own between the sweeps (k_smooth_persist) instead of 50 launches of one sweep each
 *                  (0.12-0.16 ms per step saved on a rank of eight; a rank without ice-free own nodes whose exchanged nodes cannot change is done after one sweep); the same
 *                  bits.  -1 (default) / 1 = on, 0 = one launch per sweep
 *   "ipc_pad"      before nxs_dyn_ipc_export: the mailbox gets room for at least this many received nodes (profiling aid: a rank whose mailbox is connected
 *                  to itself stores its own, possibly longer, send segments into it)
 *   "ipc_delay", "halo_one_directional"   test doors of the exchange protocols, see NXS_DELAY_* above */
NXS_API int nxs_dyn_set_option(nxs_dyn_handle *h, const char *key, int64_t value);

/* Test door: copies a named internal work array (rlmass, node_mass, C_bu, grad_ssh, fcor, VTM, shape,
 * emass, ecbu, force, volume, expC) to the host so that parity tests can localise a difference. */
NXS_API int nxs_dyn_debug_array(nxs_dyn_handle *h, const char *name, double *out, int64_t n);

/* Test door: option "trace_branches" = 1 zeroes a per-element record and makes every following step run the one-kernel-per-loop
 * family with the record kept: 4 words per element -- a hash of the branch updateSigmaDamage took at every sub-step (damage
 * increment yes / no, FE.cpp:4229; skipped, conc <= 0.1, FE.cpp:4151), the number of damaging sub-steps, flag bits (|dcrit - 1| <
 * 1e-9 seen, |conc - 0.1| < 1e-12 seen, skipped) and the sub-steps seen.  The oracle keeps the same record (oracle/dyn_ref.h), so a
 * test can name the elements where the two implementations ever took different branches.  Results are unchanged by the option. */
NXS_API int nxs_dyn_get_branch_trace(nxs_dyn_handle *h, uint64_t *out, int64_t num_words);

/* Connectivity tables with the exact content and ordering of BamgConvertMeshx -> Mesh::WriteMesh
 * (contrib/bamg/src/Mesh.cpp:514-543, 798-865) for a mesh given as 1-based triangles.
 * Pass NULL outputs to query the widths first. Tables are doubles (NaN / 0 padded) like bamg's. */
NXS_API int nxs_mesh_connectivity(const int32_t *indices, int32_t num_nodes, int32_t num_elements,
                          int32_t *nec_width, double *nodal_element_connectivity,
                          int32_t *nc_width, double *nodal_connectivity);

/* bamgmesh->ElementConnectivity (contrib/bamg/src/Mesh.cpp:777-796): ec[3*e+j] = 1-based number of the triangle
 * across local edge j (vertices (j+1)%3,(j+2)%3) of triangle e, NaN on the boundary.  Host only. */
/* M_Cohesion of calcCohesion() (FE.cpp:3909-3914) from initIce's random field (FE.cpp:11459-11475): C_fix + C_alea * r(id),
 * r = boost::uniform_01<boost::minstd_rand>, one draw per global element in id order.  global_element_id: 1-based
 * (M_mesh.trianglesIdWithGhost()).  Host only. */
NXS_API int nxs_calc_cohesion(double C_fix, double C_alea, const int32_t *global_element_id, int64_t num_elements,
                              int64_t num_global_elements, double *cohesion);

NXS_API int nxs_mesh_element_connectivity(const int32_t *indices, int32_t num_nodes, int32_t num_elements,
                                  double *element_connectivity);

#ifdef __cplusplus
}
#endif
#endif /* NXS_DYN_H */
