// nxs_io_nc4.inl -- the Moorings file as NetCDF-4, the container the reference writes (netCDF::NcFile(..., replace) of netcdf-cxx4:
// model/gridoutput.cpp:857), through the HDF5 C library.  Textually included by nxs_io.cpp.
//
// A NetCDF-4 file is an HDF5 file that follows the netCDF-4 conventions (netcdf-c, libhdf5/nc4hdf.c), all written here:
//   * groups and datasets track and index their creation order (dimension and variable ids are creation order);
//   * every dimension is a dataset that is an HDF5 DIMENSION SCALE carrying the attribute _Netcdf4Dimid: a coordinate variable
//     (time) is its own scale, named after itself; a dimension without variable (nv, x, y) is a float32 big-endian dataset of
//     the dimension's length without data, named "This is a netCDF dimension but not a netCDF variable.%10d";
//   * every variable has its scales attached (DIMENSION_LIST / REFERENCE_LIST through H5DSattach_scale);
//   * an unlimited dimension is a chunked dataset with an unlimited maximum extent, and so is every variable along it;
//   * text attributes are fixed-length null-terminated ASCII strings on a scalar dataspace, numeric ones 1-D arrays;
//     _FillValue is both an attribute and the dataset's fill value.
// No HDF5 header is needed: the handful of entry points is resolved with dlopen (libhdf5 + libhdf5_hl, version 1.10 or later:
// hid_t is 64 bits there), so the product has no link-time dependency on a library a host may keep elsewhere.
#include <dlfcn.h>

namespace nc4 {

typedef int64_t hid_t;
typedef unsigned long long hsize_t;
typedef int herr_t;

struct Api {
    void *lib = nullptr, *hl = nullptr;
    bool tried = false, ok = false;
    std::string why;
    herr_t (*H5open)();
    herr_t (*H5get_libversion)(unsigned *, unsigned *, unsigned *);
    herr_t (*H5Eset_auto2)(hid_t, void *, void *);
    hid_t (*H5Fcreate)(const char *, unsigned, hid_t, hid_t);
    hid_t (*H5Fopen)(const char *, unsigned, hid_t);
    herr_t (*H5Fclose)(hid_t);
    hid_t (*H5Pcreate)(hid_t);
    herr_t (*H5Pclose)(hid_t);
    herr_t (*H5Pset_link_creation_order)(hid_t, unsigned);
    herr_t (*H5Pset_attr_creation_order)(hid_t, unsigned);
    herr_t (*H5Pset_chunk)(hid_t, int, const hsize_t *);
    herr_t (*H5Pset_fill_value)(hid_t, hid_t, const void *);
    herr_t (*H5Pset_fill_time)(hid_t, int);
    hid_t (*H5Screate)(int);
    hid_t (*H5Screate_simple)(int, const hsize_t *, const hsize_t *);
    herr_t (*H5Sselect_hyperslab)(hid_t, int, const hsize_t *, const hsize_t *, const hsize_t *, const hsize_t *);
    int (*H5Sget_simple_extent_dims)(hid_t, hsize_t *, hsize_t *);
    herr_t (*H5Sclose)(hid_t);
    hid_t (*H5Dcreate2)(hid_t, const char *, hid_t, hid_t, hid_t, hid_t, hid_t);
    hid_t (*H5Dopen2)(hid_t, const char *, hid_t);
    hid_t (*H5Dget_space)(hid_t);
    herr_t (*H5Dset_extent)(hid_t, const hsize_t *);
    herr_t (*H5Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void *);
    herr_t (*H5Dclose)(hid_t);
    hid_t (*H5Acreate2)(hid_t, const char *, hid_t, hid_t, hid_t, hid_t);
    herr_t (*H5Awrite)(hid_t, hid_t, const void *);
    herr_t (*H5Aclose)(hid_t);
    hid_t (*H5Tcopy)(hid_t);
    herr_t (*H5Tset_size)(hid_t, size_t);
    herr_t (*H5Tset_strpad)(hid_t, int);
    herr_t (*H5Tset_cset)(hid_t, int);
    herr_t (*H5Tclose)(hid_t);
    int (*H5Lexists)(hid_t, const char *, hid_t);
    herr_t (*H5Literate)(hid_t, int, int, hsize_t *, herr_t (*)(hid_t, const char *, const void *, void *), void *);
    herr_t (*H5DSset_scale)(hid_t, const char *);
    herr_t (*H5DSattach_scale)(hid_t, hid_t, unsigned);
    hid_t P_FILE_CREATE, P_DATASET_CREATE, T_NATIVE_DOUBLE, T_NATIVE_FLOAT, T_NATIVE_INT, T_F32LE, T_F64LE, T_F32BE, T_I32LE, T_C_S1;
    unsigned ver[3] = {0, 0, 0};
};

inline Api &api() {
    static Api a;
    if (a.tried) return a;
    a.tried = true;
    const char *names[] = {"libhdf5.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"};
    const char *hl_names[] = {"libhdf5_hl.so", "libhdf5_hl.so.100", "libhdf5_hl.so.200", "libhdf5_serial_hl.so", "/opt/conda/lib/libhdf5_hl.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5_hl.so"};
    if (const char *env = getenv("NXS_HDF5_LIBRARY")) a.lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    for (const char *n : names) { if (a.lib) break; a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
    if (!a.lib) { a.why = "libhdf5 not found (set NXS_HDF5_LIBRARY)"; return a; }
    if (const char *env = getenv("NXS_HDF5_HL_LIBRARY")) a.hl = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    for (const char *n : hl_names) { if (a.hl) break; a.hl = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
    if (!a.hl) { a.why = "libhdf5_hl not found (set NXS_HDF5_HL_LIBRARY)"; return a; }
#define NC4_SYM(lib, field) *(void **)(&a.field) = dlsym(a.lib, #field); if (!a.field) { a.why = std::string("HDF5 lacks ") + #field; return a; }
    NC4_SYM(lib, H5open) NC4_SYM(lib, H5get_libversion) NC4_SYM(lib, H5Eset_auto2) NC4_SYM(lib, H5Fcreate) NC4_SYM(lib, H5Fopen) NC4_SYM(lib, H5Fclose)
    NC4_SYM(lib, H5Pcreate) NC4_SYM(lib, H5Pclose) NC4_SYM(lib, H5Pset_link_creation_order) NC4_SYM(lib, H5Pset_attr_creation_order) NC4_SYM(lib, H5Pset_chunk)
    NC4_SYM(lib, H5Pset_fill_value) NC4_SYM(lib, H5Pset_fill_time) NC4_SYM(lib, H5Screate) NC4_SYM(lib, H5Screate_simple) NC4_SYM(lib, H5Sselect_hyperslab)
    NC4_SYM(lib, H5Sget_simple_extent_dims) NC4_SYM(lib, H5Sclose) NC4_SYM(lib, H5Dcreate2) NC4_SYM(lib, H5Dopen2) NC4_SYM(lib, H5Dget_space) NC4_SYM(lib, H5Dset_extent)
    NC4_SYM(lib, H5Dwrite) NC4_SYM(lib, H5Dclose) NC4_SYM(lib, H5Acreate2) NC4_SYM(lib, H5Awrite) NC4_SYM(lib, H5Aclose) NC4_SYM(lib, H5Tcopy) NC4_SYM(lib, H5Tset_size)
    NC4_SYM(lib, H5Tset_strpad) NC4_SYM(lib, H5Tset_cset) NC4_SYM(lib, H5Tclose) NC4_SYM(lib, H5Lexists)
#undef NC4_SYM
    for (const char *n : {"H5Literate", "H5Literate1", "H5Literate2"}) { *(void **)(&a.H5Literate) = dlsym(a.lib, n); if (a.H5Literate) break; }
    if (!a.H5Literate) { a.why = "HDF5 lacks H5Literate"; return a; }
    *(void **)(&a.H5DSset_scale) = dlsym(a.hl, "H5DSset_scale");
    *(void **)(&a.H5DSattach_scale) = dlsym(a.hl, "H5DSattach_scale");
    if (!a.H5DSset_scale || !a.H5DSattach_scale) { a.why = "libhdf5_hl lacks the dimension-scale API"; return a; }
    if (a.H5open() < 0) { a.why = "H5open failed"; return a; }
    a.H5get_libversion(&a.ver[0], &a.ver[1], &a.ver[2]);
    if (a.ver[0] == 1 && a.ver[1] < 10) { a.why = "HDF5 older than 1.10 (32-bit hid_t)"; return a; }
    struct { hid_t *dst; const char *sym; } globals[] = {
        {&a.P_FILE_CREATE, "H5P_CLS_FILE_CREATE_ID_g"}, {&a.P_DATASET_CREATE, "H5P_CLS_DATASET_CREATE_ID_g"},
        {&a.T_NATIVE_DOUBLE, "H5T_NATIVE_DOUBLE_g"}, {&a.T_NATIVE_FLOAT, "H5T_NATIVE_FLOAT_g"}, {&a.T_NATIVE_INT, "H5T_NATIVE_INT_g"},
        {&a.T_F32LE, "H5T_IEEE_F32LE_g"}, {&a.T_F64LE, "H5T_IEEE_F64LE_g"}, {&a.T_F32BE, "H5T_IEEE_F32BE_g"}, {&a.T_I32LE, "H5T_STD_I32LE_g"},
        {&a.T_C_S1, "H5T_C_S1_g"}};
    for (auto &g : globals) {
        const hid_t *p = (const hid_t *)dlsym(a.lib, g.sym);
        if (!p) { a.why = std::string("HDF5 lacks ") + g.sym; return a; }
        *g.dst = *p;
    }
    a.H5Eset_auto2(0 /* H5E_DEFAULT */, nullptr, nullptr);  // errors come back as return codes; nothing is printed
    a.ok = true;
    return a;
}

inline bool available() { return api().ok; }

enum { ACC_RDWR = 1, ACC_TRUNC = 2, S_SCALAR = 0, CRT_ORDER = 1 | 2 /* tracked | indexed */, D_FILL_TIME_NEVER = 1, SELECT_SET = 0 };
const hsize_t UNLIMITED = (hsize_t)(-1);

struct Closer {  // handles opened while building a file, closed in reverse order whatever happens
    Api &a;
    std::vector<std::pair<char, hid_t>> ids;
    explicit Closer(Api &api_) : a(api_) {}
    hid_t keep(char kind, hid_t id) { if (id >= 0) ids.emplace_back(kind, id); return id; }
    ~Closer() {
        for (auto it = ids.rbegin(); it != ids.rend(); ++it)
            switch (it->first) {
                case 'F': a.H5Fclose(it->second); break; case 'P': a.H5Pclose(it->second); break; case 'S': a.H5Sclose(it->second); break;
                case 'D': a.H5Dclose(it->second); break; case 'A': a.H5Aclose(it->second); break; case 'T': a.H5Tclose(it->second); break;
            }
    }
};

inline int put_text_att(Api &a, hid_t obj, const std::string &name, const std::string &text) {
    Closer c(a);
    const hid_t t = c.keep('T', a.H5Tcopy(a.T_C_S1));
    if (t < 0 || a.H5Tset_size(t, std::max<size_t>(text.size(), 1)) < 0 || a.H5Tset_strpad(t, 0 /* NULLTERM */) < 0 || a.H5Tset_cset(t, 0 /* ASCII */) < 0) return -1;
    const hid_t s = c.keep('S', a.H5Screate(S_SCALAR));
    const hid_t at = c.keep('A', a.H5Acreate2(obj, name.c_str(), t, s, 0, 0));
    if (s < 0 || at < 0) return -1;
    std::string buf = text;
    if (buf.empty()) buf.push_back('\0');
    return a.H5Awrite(at, t, buf.data()) < 0 ? -1 : 0;
}

inline int put_num_att(Api &a, hid_t obj, const std::string &name, hid_t file_type, hid_t mem_type, const void *v) {
    Closer c(a);
    const hsize_t one = 1;
    const hid_t s = c.keep('S', a.H5Screate_simple(1, &one, nullptr));
    const hid_t at = c.keep('A', a.H5Acreate2(obj, name.c_str(), file_type, s, 0, 0));
    if (s < 0 || at < 0) return -1;
    return a.H5Awrite(at, mem_type, v) < 0 ? -1 : 0;
}

inline int put_atts(Api &a, hid_t obj, const std::vector<Att> &atts) {
    for (const Att &t : atts) {
        if (t.type == NC_CHAR) { if (put_text_att(a, obj, t.name, t.text)) return -1; }
        else { const float f = (float)t.nums[0]; if (put_num_att(a, obj, t.name, a.T_F32LE, a.T_NATIVE_FLOAT, &f)) return -1; }
    }
    return 0;
}

inline int create(const char *path, int32_t ncols, int32_t nrows, const float *lon, const float *lat, const Schema &S, float miss_val) {
    Api &a = api();
    if (!a.ok) return fail(NXS_ERR_INVALID, "NetCDF-4 output needs the HDF5 library: %s", a.why.c_str());
    Closer c(a);
    const hid_t fcpl = c.keep('P', a.H5Pcreate(a.P_FILE_CREATE));
    if (fcpl < 0 || a.H5Pset_link_creation_order(fcpl, CRT_ORDER) < 0 || a.H5Pset_attr_creation_order(fcpl, CRT_ORDER) < 0) return fail(NXS_ERR_INVALID, "HDF5: file creation properties");
    const hid_t file = c.keep('F', a.H5Fcreate(path, ACC_TRUNC, fcpl, 0));
    if (file < 0) return fail(NXS_ERR_INVALID, "cannot create %s", path);
    {
        char prov[96];
        snprintf(prov, sizeof prov, "version=2,nxs_io=1,hdf5=%u.%u.%u", a.ver[0], a.ver[1], a.ver[2]);
        if (put_text_att(a, file, "_NCProperties", prov)) return fail(NXS_ERR_INVALID, "HDF5: _NCProperties");
    }
    const hsize_t dimlen[4] = {0, 2, (hsize_t)ncols, (hsize_t)nrows};
    hid_t scale[4] = {-1, -1, -1, -1};
    auto new_dcpl = [&]() -> hid_t {
        const hid_t p = c.keep('P', a.H5Pcreate(a.P_DATASET_CREATE));
        if (p >= 0) a.H5Pset_attr_creation_order(p, CRT_ORDER);
        return p;
    };
    // datasets in the reference's creation order: [projection], time, (nv), time_bnds, (x), (y), longitude, latitude, the fields --
    // a dimension without variable is created where the reference calls addDim for it
    auto make_pure_dim = [&](int d) -> int {
        const hid_t p = new_dcpl();
        const hid_t s = c.keep('S', a.H5Screate_simple(1, &dimlen[d], &dimlen[d]));
        if (p < 0 || s < 0 || a.H5Pset_fill_time(p, D_FILL_TIME_NEVER) < 0) return -1;
        const hid_t ds = c.keep('D', a.H5Dcreate2(file, S.dims[d].first.c_str(), a.T_F32BE, s, 0, p, 0));
        if (ds < 0) return -1;
        char nm[96];
        snprintf(nm, sizeof nm, "This is a netCDF dimension but not a netCDF variable.%10d", (int)dimlen[d]);
        if (a.H5DSset_scale(ds, nm) < 0) return -1;
        const int id = d;
        if (put_num_att(a, ds, "_Netcdf4Dimid", a.T_I32LE, a.T_NATIVE_INT, &id)) return -1;
        scale[d] = ds;
        return 0;
    };
    std::vector<hid_t> var_ds(S.V.size(), -1);
    for (size_t i = 0; i < S.V.size(); ++i) {
        const Var &v = S.V[i];
        if (v.name == "time_bnds" && make_pure_dim(D_NV)) return fail(NXS_ERR_INVALID, "HDF5: dimension nv");
        if (v.name == "longitude" && (make_pure_dim(D_X) || make_pure_dim(D_Y))) return fail(NXS_ERR_INVALID, "HDF5: dimensions x, y");
        const int nd = (int)v.dims.size();
        hsize_t cur[3] = {1, 1, 1}, mx[3] = {1, 1, 1}, chunk[3] = {1, 1, 1};
        for (int k = 0; k < nd; ++k) {
            const bool unl = v.dims[k] == D_TIME;
            cur[k] = unl ? 0 : dimlen[v.dims[k]]; mx[k] = unl ? UNLIMITED : cur[k];
            chunk[k] = unl ? (nd == 1 ? 512 : 1) : dimlen[v.dims[k]];
        }
        const hid_t p = new_dcpl();
        const hid_t s = c.keep('S', nd == 0 ? a.H5Screate(S_SCALAR) : a.H5Screate_simple(nd, cur, mx));
        if (p < 0 || s < 0) return fail(NXS_ERR_INVALID, "HDF5: dataspace of %s", v.name.c_str());
        if (v.record && a.H5Pset_chunk(p, nd, chunk) < 0) return fail(NXS_ERR_INVALID, "HDF5: chunking of %s", v.name.c_str());
        const hid_t ft = v.type == NC_DOUBLE ? a.T_F64LE : (v.type == NC_INT ? a.T_I32LE : a.T_F32LE);
        bool has_fill = false;
        for (const Att &t : v.atts) has_fill = has_fill || t.name == "_FillValue";
        if (has_fill && a.H5Pset_fill_value(p, a.T_NATIVE_FLOAT, &miss_val) < 0) return fail(NXS_ERR_INVALID, "HDF5: fill value of %s", v.name.c_str());
        const hid_t ds = c.keep('D', a.H5Dcreate2(file, v.name.c_str(), ft, s, 0, p, 0));
        if (ds < 0) return fail(NXS_ERR_INVALID, "HDF5: cannot create variable %s", v.name.c_str());
        var_ds[i] = ds;
        if (v.name == "time") {  // a coordinate variable is the scale of its dimension
            if (a.H5DSset_scale(ds, "time") < 0) return fail(NXS_ERR_INVALID, "HDF5: time scale");
            const int id = D_TIME;
            if (put_num_att(a, ds, "_Netcdf4Dimid", a.T_I32LE, a.T_NATIVE_INT, &id)) return fail(NXS_ERR_INVALID, "HDF5: _Netcdf4Dimid");
            scale[D_TIME] = ds;
        }
        if (put_atts(a, ds, v.atts)) return fail(NXS_ERR_INVALID, "HDF5: attributes of %s", v.name.c_str());
        if (v.name == "Polar_Stereographic_Grid") { const int zero = 0; if (a.H5Dwrite(ds, a.T_NATIVE_INT, 0, 0, 0, &zero) < 0) return fail(NXS_ERR_INVALID, "HDF5: write"); }
        if (v.name == "longitude" || v.name == "latitude")
            if (a.H5Dwrite(ds, a.T_NATIVE_FLOAT, 0, 0, 0, v.name == "longitude" ? lon : lat) < 0) return fail(NXS_ERR_INVALID, "HDF5: write %s", v.name.c_str());
    }
    for (size_t i = 0; i < S.V.size(); ++i) {
        const Var &v = S.V[i];
        if (v.name == "time") continue;  // its own scale
        for (size_t k = 0; k < v.dims.size(); ++k)
            if (a.H5DSattach_scale(var_ds[i], scale[v.dims[k]], (unsigned)k) < 0) return fail(NXS_ERR_INVALID, "HDF5: attaching dimension %s to %s", S.dims[v.dims[k]].first.c_str(), v.name.c_str());
    }
    if (put_atts(a, file, S.gatts)) return fail(NXS_ERR_INVALID, "HDF5: global attributes");
    return NXS_OK;
}

// the fields of a Moorings file = its datasets of rank 3, in creation order (the order the caller listed them at create)
inline herr_t collect_name(hid_t, const char *name, const void *, void *op) {
    static_cast<std::vector<std::string> *>(op)->push_back(name);
    return 0;
}
inline int list_fields(Api &a, hid_t file, std::vector<std::string> &fields) {
    std::vector<std::string> all;
    hsize_t idx = 0;
    if (a.H5Literate(file, 1 /* H5_INDEX_CRT_ORDER */, 0 /* H5_ITER_INC */, &idx, collect_name, &all) < 0) return fail(NXS_ERR_INVALID, "HDF5: cannot list the file");
    for (const std::string &nm : all) {
        Closer c(a);
        const hid_t ds = c.keep('D', a.H5Dopen2(file, nm.c_str(), 0));
        if (ds < 0) continue;
        const hid_t s = c.keep('S', a.H5Dget_space(ds));
        hsize_t d[8];
        if (s >= 0 && a.H5Sget_simple_extent_dims(s, nullptr, nullptr) == 3 && a.H5Sget_simple_extent_dims(s, d, nullptr) == 3) fields.push_back(nm);
    }
    return NXS_OK;
}

// appendNetCDF (gridoutput.cpp:984-1030): one more record along time
inline int append(const char *path, double timestamp, double averaging_period, int32_t nvars, const float *const *data) {
    Api &a = api();
    if (!a.ok) return fail(NXS_ERR_INVALID, "%s is a NetCDF-4 file and the HDF5 library is not available: %s", path, a.why.c_str());
    Closer c(a);
    const hid_t file = c.keep('F', a.H5Fopen(path, ACC_RDWR, 0));
    if (file < 0) return fail(NXS_ERR_INVALID, "cannot open %s", path);
    const hid_t tds = c.keep('D', a.H5Dopen2(file, "time", 0));
    if (tds < 0) return fail(NXS_ERR_INVALID, "%s has no variable time", path);
    hsize_t n = 0;
    {
        const hid_t s = c.keep('S', a.H5Dget_space(tds));
        if (s < 0 || a.H5Sget_simple_extent_dims(s, &n, nullptr) != 1) return fail(NXS_ERR_INVALID, "time is not one-dimensional");
    }
    auto put_record = [&](hid_t ds, int nd, const hsize_t *rest, hid_t mem_type, const void *buf) -> int {
        hsize_t ext[3] = {n + 1, 1, 1}, start[3] = {n, 0, 0}, cnt[3] = {1, 1, 1};
        for (int k = 1; k < nd; ++k) ext[k] = cnt[k] = rest[k - 1];
        if (a.H5Dset_extent(ds, ext) < 0) return -1;
        Closer cc(a);
        const hid_t fs = cc.keep('S', a.H5Dget_space(ds)), ms = cc.keep('S', a.H5Screate_simple(nd, cnt, nullptr));
        if (fs < 0 || ms < 0 || a.H5Sselect_hyperslab(fs, SELECT_SET, start, nullptr, cnt, nullptr) < 0) return -1;
        return a.H5Dwrite(ds, mem_type, ms, fs, 0, buf) < 0 ? -1 : 0;
    };
    if (put_record(tds, 1, nullptr, a.T_NATIVE_DOUBLE, &timestamp)) return fail(NXS_ERR_INVALID, "HDF5: append to time");
    {
        const hid_t ds = c.keep('D', a.H5Dopen2(file, "time_bnds", 0));
        const double tb[2] = {timestamp - 0.5 * averaging_period, timestamp + 0.5 * averaging_period};
        const hsize_t two = 2;
        if (ds < 0 || put_record(ds, 2, &two, a.T_NATIVE_DOUBLE, tb)) return fail(NXS_ERR_INVALID, "HDF5: append to time_bnds");
    }
    // the fields: every dataset of rank 3, in creation order = the order the caller listed them at nxs_moorings_create
    hsize_t yx[2] = {0, 0};
    {
        const hid_t ds = c.keep('D', a.H5Dopen2(file, "longitude", 0));
        const hid_t s = ds >= 0 ? c.keep('S', a.H5Dget_space(ds)) : -1;
        if (s < 0 || a.H5Sget_simple_extent_dims(s, yx, nullptr) != 2) return fail(NXS_ERR_INVALID, "%s has no 2-D longitude", path);
    }
    std::vector<std::string> names;
    if (int rc = list_fields(a, file, names)) return rc;
    if ((int)names.size() != nvars) return fail(NXS_ERR_INVALID, "file holds %zu fields, caller passed %d", names.size(), nvars);
    for (int i = 0; i < nvars; ++i) {
        const hid_t ds = c.keep('D', a.H5Dopen2(file, names[i].c_str(), 0));
        if (ds < 0 || put_record(ds, 3, yx, a.T_NATIVE_FLOAT, data[i])) return fail(NXS_ERR_INVALID, "HDF5: append to %s", names[i].c_str());
    }
    return NXS_OK;
}

}  // namespace nc4
