/*
 * nxs_io.h -- host-side output writers: the two file formats the reference's
 * time loop produces from the arrays of the dynamics path, so that a host that drops libnxsdyn.so in
 * keeps its outputs.  Plain C ABI, host memory only, no GPU involved.
 *
 *   Exporter  core/src/exporter.cpp:32-189   field_*.bin/.dat, mesh_*.bin/.dat, restart files:
 *             binary  = per record  int32 count, then count values (int32 | float32 | float64);
 *             sidecar = one text line per record  "name type count min max".
 *   Moorings  model/gridoutput.cpp:805-1035  CF-1.6 NetCDF: dims time(unlimited), nv=2, x, y;
 *             time, time_bnds, longitude, latitude, one float variable per field with _FillValue,
 *             optional Polar_Stereographic_Grid mapping variable.  The reference writes NetCDF-4 through netcdf-cxx4;
 *             no NetCDF library exists in this image, so the file is written as NetCDF-4 through the HDF5 C library
 *             (resolved with dlopen at the first call: libhdf5 + libhdf5_hl >= 1.10, NXS_HDF5_LIBRARY / NXS_HDF5_HL_LIBRARY
 *             override the search) following the netCDF-4 on-disk conventions -- dimension scales, _Netcdf4Dimid, tracked
 *             creation order, chunked unlimited dimension -- or, without HDF5 or on request, by hand as NetCDF-3 classic
 *             (CDF-1), which every NetCDF reader opens too; same schema either way.
 */
#ifndef NXS_IO_H
#define NXS_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define NXS_IO_API __attribute__((visibility("default")))
#else
#define NXS_IO_API
#endif

typedef struct nxs_exporter nxs_exporter;

/* precision: "double" or "float" (output.exporter_precision, exporter.cpp:17-27). */
NXS_IO_API int nxs_exporter_open(const char *bin_path, const char *dat_path, const char *precision, nxs_exporter **out);
/* writeMesh (exporter.cpp:72-128): records Elements, id, Nodes_x, Nodes_y in that order. */
NXS_IO_API int nxs_exporter_write_mesh(nxs_exporter *e, const double *xnod, const double *ynod, const int32_t *idnod,
                                       int64_t num_nodes, const int32_t *elements, int64_t num_indices);
/* writeField (exporter.cpp:130-156): floating fields follow the exporter precision, except "Time" (double). */
NXS_IO_API int nxs_exporter_write_field(nxs_exporter *e, const char *name, const double *values, int64_t count);
NXS_IO_API int nxs_exporter_write_field_int(nxs_exporter *e, const char *name, const int32_t *values, int64_t count);
/* writeRecord (exporter.cpp:158-189) + close both files. */
NXS_IO_API int nxs_exporter_close(nxs_exporter *e);

/* Reading the files back: Exporter::readRecord + loadFile (core/src/exporter.cpp:183-222).  Records of type "double"
 * and "int" as the reference; "float" records (which the reference's loadFile rejects) are widened to double. */
typedef struct nxs_exporter_file nxs_exporter_file;
NXS_IO_API int nxs_exporter_load(const char *bin_path, const char *dat_path, nxs_exporter_file **out);
NXS_IO_API int nxs_exporter_file_num_records(const nxs_exporter_file *f);
NXS_IO_API int nxs_exporter_file_record(const nxs_exporter_file *f, int index, const char **name, const char **type, int64_t *count);
/* By name, as field_map_dbl[name] / field_map_int[name] in readRestart (the first record of that name). */
NXS_IO_API int nxs_exporter_file_get_double(const nxs_exporter_file *f, const char *name, double *out, int64_t count);
NXS_IO_API int nxs_exporter_file_get_int(const nxs_exporter_file *f, const char *name, int32_t *out, int64_t count);
NXS_IO_API int nxs_exporter_file_close(nxs_exporter_file *f);

/* FiniteElement::writeRestart (FE.cpp:9518-9695): <directory>/mesh_<name>.bin/.dat (writeMesh) and
 * <directory>/field_<name>.bin/.dat with the records, in this order: Misc_int {pcpt, M_flag_fix, mesh_adapt_step,
 * M_nb_regrid}, M_dirichlet_flags, Time, the element variables under their M_restart_names_elt names
 * (e.g. M_conc, M_thick, M_sigma_0..2, M_damage, ...), M_VT, M_UM, M_UT ([u | v], 2*num_nodes), PreviousNumbering.
 * Always double precision, as the reference (Exporter exporter("double"), FE.cpp:9590).  The directory must exist. */
NXS_IO_API int nxs_restart_write(const char *directory, const char *name_str, const double *xnod, const double *ynod,
                                 const int32_t *idnod, int64_t num_nodes, const int32_t *elements, int64_t num_indices,
                                 const int32_t misc_int[4], const int32_t *dirichlet_flags, int64_t num_dirichlet,
                                 double current_time, int32_t num_elt_vars, const char *const *elt_names,
                                 const double *const *elt_values, const double *VT, const double *UM, const double *UT,
                                 const double *previous_numbering);
/* FiniteElement::readRestart's file part (FE.cpp:9699-9790): loads the pair and checks that the records it needs exist. */
NXS_IO_API int nxs_restart_read(const char *directory, const char *name_str, nxs_exporter_file **mesh, nxs_exporter_file **field);

typedef struct nxs_mooring_var {
    const char *name, *standard_name, *long_name, *units, *cell_methods; /* gridoutput.hpp variable descriptors */
} nxs_mooring_var;

typedef struct nxs_mooring_proj { /* createProjectionVariable, gridoutput.cpp:943-980; NULL = none */
    double semi_major_axis, semi_minor_axis, lat0, lat_ts, rotation;
    int32_t false_easting;
} nxs_mooring_proj;

/* initNetCDF (gridoutput.cpp:805-940).  lon/lat: [nrows*ncols] floats, row-major (y, x).
 * averaging_period in days (0 = snapshots: "time: point ").  nxs_moorings_create = format NXS_NC_AUTO: NetCDF-4 as the reference
 * when the HDF5 library can be loaded, NetCDF-3 classic otherwise. */
enum { NXS_NC_AUTO = 0, NXS_NC_CLASSIC = 3, NXS_NC_NETCDF4 = 4 };
NXS_IO_API int nxs_moorings_create_format(const char *path, int32_t ncols, int32_t nrows, const float *lon, const float *lat,
                                          int32_t nvars, const nxs_mooring_var *vars, float miss_val, double averaging_period,
                                          const nxs_mooring_proj *proj, int32_t format);
/* NXS_NC_CLASSIC or NXS_NC_NETCDF4 from the file's magic bytes; negative on error */
NXS_IO_API int nxs_moorings_file_format(const char *path);
NXS_IO_API int nxs_moorings_create(const char *path, int32_t ncols, int32_t nrows, const float *lon, const float *lat,
                                   int32_t nvars, const nxs_mooring_var *vars, float miss_val, double averaging_period,
                                   const nxs_mooring_proj *proj);
/* appendNetCDF (gridoutput.cpp:984-1030): one more record; data[v] is [nrows*ncols] floats. timestamp in days
 * since 1900-01-01.  Works on either container (told apart by the magic bytes). */
NXS_IO_API int nxs_moorings_append(const char *path, double timestamp, double averaging_period, int32_t nvars,
                                   const float *const *data);

NXS_IO_API const char *nxs_io_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
