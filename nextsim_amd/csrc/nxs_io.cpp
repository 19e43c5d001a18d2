// nxs_io.cpp -- host-side output writers of include/nxs_io.h.
//
//   Exporter : core/src/exporter.cpp:32-189 (binary records + "name type count min max" sidecar lines)
//   Moorings : model/gridoutput.cpp:805-1035 (CF-1.6 schema).  The reference writes it through netcdf-cxx4 as NetCDF-4; no
//              NetCDF library exists in this image, so the file is written here as NetCDF-4 through the HDF5 C library
//              (dlopen'ed: nxs_io_nc4.inl) or, where that is absent or on request, as NetCDF-3 classic (CDF-1) by hand:
//              big-endian header (dim_list, gatt_list, var_list), fixed-size variables, then interleaved records.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "nxs_dyn.h"
#include "nxs_guard.hpp"
#include "nxs_io.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// handler of every extern "C" function-try-block of this file (nxs_guard.hpp): status code + text, never an exception across the ABI
int entry_caught(const char *entry) noexcept {
    return nxs_guard::caught(entry, [](int code, const char *text) { (void)fail(code, "%s", text); });
}

// "%g" as boost::format applies it in exporter.cpp:88-94: integers print as integers, reals with %g
std::string g_int(long long v) { return std::to_string(v); }
std::string g_dbl(double v) {
    char b[64];
    snprintf(b, sizeof b, "%g", v);
    return b;
}

}  // namespace

struct nxs_exporter {
    FILE *bin = nullptr;
    std::string dat_path, precision;
    std::vector<std::string> records;
};

namespace {

template <typename T>
int write_container(nxs_exporter *e, const T *v, int64_t n, const std::string &prec) {  // exporter.cpp:30-61
    if (n > 0x7fffffffLL) return fail(NXS_ERR_INVALID, "record too long for the int32 length prefix");
    const int fsize = (int)n;
    if (fwrite(&fsize, sizeof fsize, 1, e->bin) != 1) return fail(NXS_ERR_INVALID, "write failed");
    if (prec == "float") {
        for (int64_t i = 0; i < n; ++i) {
            const float f = (float)v[i];
            if (fwrite(&f, sizeof f, 1, e->bin) != 1) return fail(NXS_ERR_INVALID, "write failed");
        }
    } else if (n > 0 && fwrite(v, sizeof(T), (size_t)n, e->bin) != (size_t)n) {
        return fail(NXS_ERR_INVALID, "write failed");
    }
    return NXS_OK;
}

}  // namespace

extern "C" {

const char *nxs_io_last_error(void) { return g_err.c_str(); }

int nxs_exporter_open(const char *bin_path, const char *dat_path, const char *precision, nxs_exporter **out) try {
    if (!bin_path || !dat_path || !precision || !out) return fail(NXS_ERR_INVALID, "NULL argument");
    const std::string prec = precision;
    if (prec != "float" && prec != "double") return fail(NXS_ERR_INVALID, "Exporter: Unknown precision: %s", precision);  // exporter.cpp:23-27
    nxs_exporter *e = new nxs_exporter();
    e->bin = fopen(bin_path, "wb");
    if (!e->bin) { delete e; return fail(NXS_ERR_INVALID, "cannot open %s", bin_path); }
    e->dat_path = dat_path;
    e->precision = prec;
    *out = e;
    return NXS_OK;
} catch (...) { return entry_caught("nxs_exporter_open"); }

int nxs_exporter_write_mesh(nxs_exporter *e, const double *xnod, const double *ynod, const int32_t *idnod, int64_t nn,
                            const int32_t *elements, int64_t n3) try {
    if (!e || !xnod || !ynod || !idnod || !elements || nn <= 0 || n3 <= 0) return fail(NXS_ERR_INVALID, "bad mesh arguments");
    int rc;
    if ((rc = write_container(e, elements, n3, "int"))) return rc;
    e->records.push_back("Elements int " + g_int(n3) + " " + g_int(*std::min_element(elements, elements + n3)) + " " +
                         g_int(*std::max_element(elements, elements + n3)));
    if ((rc = write_container(e, idnod, nn, "int"))) return rc;
    e->records.push_back("id int " + g_int(nn) + " " + g_int(*std::min_element(idnod, idnod + nn)) + " " + g_int(*std::max_element(idnod, idnod + nn)));
    // coordinates are std::vector<double> in the model: written as doubles whatever the field precision (exporter.cpp:104-106)
    if ((rc = write_container(e, xnod, nn, "double"))) return rc;
    e->records.push_back("Nodes_x double " + g_int(nn) + " " + g_dbl(*std::min_element(xnod, xnod + nn)) + " " + g_dbl(*std::max_element(xnod, xnod + nn)));
    if ((rc = write_container(e, ynod, nn, "double"))) return rc;
    e->records.push_back("Nodes_y double " + g_int(nn) + " " + g_dbl(*std::min_element(ynod, ynod + nn)) + " " + g_dbl(*std::max_element(ynod, ynod + nn)));
    return NXS_OK;
} catch (...) { return entry_caught("nxs_exporter_write_mesh"); }

int nxs_exporter_write_field(nxs_exporter *e, const char *name, const double *v, int64_t n) try {
    if (!e || !name || (n > 0 && !v) || n < 0) return fail(NXS_ERR_INVALID, "bad field arguments");
    std::string prec = e->precision;
    if (!strcmp(name, "Time")) prec = "double";  // exporter.cpp:143-145
    int rc = write_container(e, v, n, prec);
    if (rc) return rc;
    e->records.push_back(std::string(name) + " " + prec + " " + g_int(n) + " " + g_dbl(n > 0 ? *std::min_element(v, v + n) : 0.) + " " +
                         g_dbl(n > 0 ? *std::max_element(v, v + n) : 0.));
    return NXS_OK;
} catch (...) { return entry_caught("nxs_exporter_write_field"); }

int nxs_exporter_write_field_int(nxs_exporter *e, const char *name, const int32_t *v, int64_t n) try {
    if (!e || !name || (n > 0 && !v) || n < 0) return fail(NXS_ERR_INVALID, "bad field arguments");
    int rc = write_container(e, v, n, "int");
    if (rc) return rc;
    e->records.push_back(std::string(name) + " int " + g_int(n) + " " + g_int(n > 0 ? *std::min_element(v, v + n) : 0) + " " +
                         g_int(n > 0 ? *std::max_element(v, v + n) : 0));
    return NXS_OK;
} catch (...) { return entry_caught("nxs_exporter_write_field_int"); }

int nxs_exporter_close(nxs_exporter *e) try {
    if (!e) return NXS_OK;
    int rc = NXS_OK;
    if (e->bin && fclose(e->bin) != 0) rc = fail(NXS_ERR_INVALID, "close failed");
    FILE *dat = fopen(e->dat_path.c_str(), "w");
    if (!dat) rc = fail(NXS_ERR_INVALID, "cannot open %s", e->dat_path.c_str());
    else {
        for (const std::string &s : e->records) fprintf(dat, "%s\n", s.c_str());  // exporter.cpp:158-189
        fclose(dat);
    }
    delete e;
    return rc;
} catch (...) { return entry_caught("nxs_exporter_close"); }

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Reading the Exporter's files back (Exporter::readRecord / loadFile, core/src/exporter.cpp:183-222) and the
// restart file set built from them (FiniteElement::writeRestart / readRestart, FE.cpp:9518-9695, 9699-9925)
struct nxs_exporter_file {
    struct Rec { std::string name, type; std::vector<double> d; std::vector<int32_t> i; };
    std::vector<Rec> recs;
};

extern "C" {

int nxs_exporter_load(const char *bin_path, const char *dat_path, nxs_exporter_file **out) try {
    if (!bin_path || !dat_path || !out) return fail(NXS_ERR_INVALID, "NULL argument");
    FILE *dat = fopen(dat_path, "r");
    if (!dat) return fail(NXS_ERR_INVALID, "File not found: %s", dat_path);
    nxs_exporter_file *f = new nxs_exporter_file();
    char name[256], type[64], size[64], mn[64], mx[64];
    while (fscanf(dat, "%255s %63s %63s %63s %63s", name, type, size, mn, mx) == 5) {  // readRecord: five tokens per record
        nxs_exporter_file::Rec r;
        r.name = name; r.type = type;
        f->recs.push_back(std::move(r));
    }
    fclose(dat);
    FILE *bin = fopen(bin_path, "rb");
    if (!bin) { delete f; return fail(NXS_ERR_INVALID, "File not found: %s", bin_path); }
    int rc = NXS_OK;
    for (auto &r : f->recs) {  // loadFile: int32 length, then the payload in the type the record names
        int32_t reclen = 0;
        if (fread(&reclen, sizeof reclen, 1, bin) != 1 || reclen < 0) { rc = fail(NXS_ERR_INVALID, "%s: truncated before record %s", bin_path, r.name.c_str()); break; }
        size_t got = 0;
        if (r.type == "double") { r.d.resize(reclen); got = fread(r.d.data(), sizeof(double), reclen, bin); }
        else if (r.type == "int") { r.i.resize(reclen); got = fread(r.i.data(), sizeof(int32_t), reclen, bin); }
        else if (r.type == "float") {  // the reference's loadFile stops here ("unknown type in file"); fields exported in
            std::vector<float> t(reclen);  // single precision are widened instead
            got = fread(t.data(), sizeof(float), reclen, bin);
            r.d.assign(t.begin(), t.end());
        } else { rc = fail(NXS_ERR_INVALID, "unknown type in file"); break; }
        if (got != (size_t)reclen) { rc = fail(NXS_ERR_INVALID, "%s: record %s is truncated", bin_path, r.name.c_str()); break; }
    }
    fclose(bin);
    if (rc) { delete f; return rc; }
    *out = f;
    return NXS_OK;
} catch (...) { return entry_caught("nxs_exporter_load"); }

int nxs_exporter_file_num_records(const nxs_exporter_file *f) { return f ? (int)f->recs.size() : 0; }

int nxs_exporter_file_record(const nxs_exporter_file *f, int index, const char **name, const char **type, int64_t *count) try {
    if (!f || index < 0 || index >= (int)f->recs.size()) return fail(NXS_ERR_INVALID, "record index out of range");
    const auto &r = f->recs[index];
    if (name) *name = r.name.c_str();
    if (type) *type = r.type.c_str();
    if (count) *count = (int64_t)(r.type == "int" ? r.i.size() : r.d.size());
    return NXS_OK;
} catch (...) { return entry_caught("nxs_exporter_file_record"); }

static const nxs_exporter_file::Rec *find_rec(const nxs_exporter_file *f, const char *name) {
    if (!f || !name) return nullptr;
    for (const auto &r : f->recs) if (r.name == name) return &r;  // field_map.emplace keeps the FIRST record of a name
    return nullptr;
}

int nxs_exporter_file_get_double(const nxs_exporter_file *f, const char *name, double *out, int64_t count) try {
    const auto *r = find_rec(f, name);
    if (!r) return fail(NXS_ERR_INVALID, "no record named %s", name ? name : "(null)");
    if (r->type == "int") return fail(NXS_ERR_INVALID, "record %s holds integers", name);
    if ((int64_t)r->d.size() != count || (count > 0 && !out)) return fail(NXS_ERR_INVALID, "record %s has %zu values, not %lld", name, r->d.size(), (long long)count);
    std::copy(r->d.begin(), r->d.end(), out);
    return NXS_OK;
} catch (...) { return entry_caught("nxs_exporter_file_get_double"); }

int nxs_exporter_file_get_int(const nxs_exporter_file *f, const char *name, int32_t *out, int64_t count) try {
    const auto *r = find_rec(f, name);
    if (!r) return fail(NXS_ERR_INVALID, "no record named %s", name ? name : "(null)");
    if (r->type != "int") return fail(NXS_ERR_INVALID, "record %s holds reals", name);
    if ((int64_t)r->i.size() != count || (count > 0 && !out)) return fail(NXS_ERR_INVALID, "record %s has %zu values, not %lld", name, r->i.size(), (long long)count);
    std::copy(r->i.begin(), r->i.end(), out);
    return NXS_OK;
} catch (...) { return entry_caught("nxs_exporter_file_get_int"); }

int nxs_exporter_file_close(nxs_exporter_file *f) { delete f; return NXS_OK; }

int nxs_restart_write(const char *directory, const char *name_str, const double *xnod, const double *ynod, const int32_t *idnod,
                      int64_t num_nodes, const int32_t *elements, int64_t num_indices, const int32_t misc_int[4],
                      const int32_t *dirichlet_flags, int64_t num_dirichlet, double current_time, int32_t num_elt_vars,
                      const char *const *elt_names, const double *const *elt_values, const double *VT, const double *UM,
                      const double *UT, const double *previous_numbering) try {
    if (!directory || !name_str || !misc_int || !VT || !UM || !UT || !previous_numbering || (num_elt_vars > 0 && (!elt_names || !elt_values)))
        return fail(NXS_ERR_INVALID, "NULL argument");
    const std::string base = std::string(directory) + "/";
    const int64_t ne = num_indices / 3;
    nxs_exporter *e = nullptr;
    int rc;
    // mesh_<name>.bin/.dat: writeMesh, FE.cpp:9601-9620.  Restart files are always double precision (:9590)
    if ((rc = nxs_exporter_open((base + "mesh_" + name_str + ".bin").c_str(), (base + "mesh_" + name_str + ".dat").c_str(), "double", &e))) return rc;
    rc = nxs_exporter_write_mesh(e, xnod, ynod, idnod, num_nodes, elements, num_indices);
    int rc2 = nxs_exporter_close(e);
    if (rc || rc2) return rc ? rc : rc2;
    // field_<name>.bin/.dat in the reference's record order, FE.cpp:9633-9690
    if ((rc = nxs_exporter_open((base + "field_" + name_str + ".bin").c_str(), (base + "field_" + name_str + ".dat").c_str(), "double", &e))) return rc;
    rc = nxs_exporter_write_field_int(e, "Misc_int", misc_int, 4);
    if (!rc) rc = nxs_exporter_write_field_int(e, "M_dirichlet_flags", dirichlet_flags, num_dirichlet);
    if (!rc) rc = nxs_exporter_write_field(e, "Time", &current_time, 1);
    for (int j = 0; j < num_elt_vars && !rc; ++j) rc = nxs_exporter_write_field(e, elt_names[j], elt_values[j], ne);
    if (!rc) rc = nxs_exporter_write_field(e, "M_VT", VT, 2 * num_nodes);
    if (!rc) rc = nxs_exporter_write_field(e, "M_UM", UM, 2 * num_nodes);
    if (!rc) rc = nxs_exporter_write_field(e, "M_UT", UT, 2 * num_nodes);
    if (!rc) rc = nxs_exporter_write_field(e, "PreviousNumbering", previous_numbering, num_nodes);
    rc2 = nxs_exporter_close(e);
    return rc ? rc : rc2;
} catch (...) { return entry_caught("nxs_restart_write"); }

int nxs_restart_read(const char *directory, const char *name_str, nxs_exporter_file **mesh, nxs_exporter_file **field) try {
    if (!directory || !name_str || !mesh || !field) return fail(NXS_ERR_INVALID, "NULL argument");
    const std::string base = std::string(directory) + "/";
    int rc = nxs_exporter_load((base + "mesh_" + name_str + ".bin").c_str(), (base + "mesh_" + name_str + ".dat").c_str(), mesh);
    if (rc) return rc;
    rc = nxs_exporter_load((base + "field_" + name_str + ".bin").c_str(), (base + "field_" + name_str + ".dat").c_str(), field);
    if (rc) { nxs_exporter_file_close(*mesh); *mesh = nullptr; return rc; }
    // what readRestart requires of the pair (FE.cpp:9744-9747, 9781-9787, 9813)
    for (const char *need : {"Elements", "Nodes_x", "Nodes_y", "id"})
        if (!find_rec(*mesh, need)) rc = fail(NXS_ERR_INVALID, "restart mesh file lacks %s", need);
    for (const char *need : {"Time", "Misc_int", "M_dirichlet_flags", "M_VT", "M_UM", "M_UT", "PreviousNumbering"})
        if (!find_rec(*field, need)) rc = fail(NXS_ERR_INVALID, "restart field file lacks %s", need);
    if (rc) { nxs_exporter_file_close(*mesh); nxs_exporter_file_close(*field); *mesh = *field = nullptr; }
    return rc;
} catch (...) { return entry_caught("nxs_restart_read"); }

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// NetCDF-3 classic writer (just what the Moorings schema needs)
namespace {

enum { NC_CHAR = 2, NC_INT = 4, NC_FLOAT = 5, NC_DOUBLE = 6, NC_DIMENSION = 10, NC_VARIABLE = 11, NC_ATTRIBUTE = 12 };

struct Buf {
    std::vector<unsigned char> b;
    void u32(uint32_t v) { for (int s = 24; s >= 0; s -= 8) b.push_back((unsigned char)(v >> s)); }
    void bytes(const void *p, size_t n) { const unsigned char *c = (const unsigned char *)p; b.insert(b.end(), c, c + n); }
    void pad4() { while (b.size() % 4) b.push_back(0); }
    void name(const std::string &s) { u32((uint32_t)s.size()); bytes(s.data(), s.size()); pad4(); }
    void f32(float v) { uint32_t u; memcpy(&u, &v, 4); u32(u); }
    void f64(double v) { uint64_t u; memcpy(&u, &v, 8); u32((uint32_t)(u >> 32)); u32((uint32_t)u); }
};

struct Att {
    std::string name; int type; std::string text; std::vector<double> nums;  // nums stored as float (NC_FLOAT) / int
};
Att att_text(const std::string &n, const std::string &t) { return Att{n, NC_CHAR, t, {}}; }
Att att_float(const std::string &n, double v) { return Att{n, NC_FLOAT, "", {v}}; }

void put_atts(Buf &h, const std::vector<Att> &atts) {
    if (atts.empty()) { h.u32(0); h.u32(0); return; }
    h.u32(NC_ATTRIBUTE); h.u32((uint32_t)atts.size());
    for (const Att &a : atts) {
        h.name(a.name);
        h.u32((uint32_t)a.type);
        if (a.type == NC_CHAR) { h.u32((uint32_t)a.text.size()); h.bytes(a.text.data(), a.text.size()); h.pad4(); }
        else { h.u32((uint32_t)a.nums.size()); for (double v : a.nums) h.f32((float)v); }
    }
}

struct Var {
    std::string name; std::vector<int> dims; std::vector<Att> atts; int type; uint32_t vsize = 0; bool record = false;
    uint32_t begin = 0;
};

uint32_t type_size(int t) { return t == NC_DOUBLE ? 8 : (t == NC_CHAR ? 1 : 4); }

uint32_t rd32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// minimal header parser for append: returns numrecs, record size and the begin of every record variable
struct Parsed { uint32_t numrecs = 0, recsize = 0; std::vector<std::pair<std::string, uint32_t>> rec_begin; std::vector<uint32_t> rec_vsize; };

bool parse_header(const std::vector<unsigned char> &f, Parsed &out) {
    size_t p = 0;
    auto need = [&](size_t n) { return p + n <= f.size(); };
    if (!need(8) || memcmp(f.data(), "CDF\x01", 4)) return false;
    out.numrecs = rd32(&f[4]);
    p = 8;
    auto skip_name = [&]() -> std::string {
        if (!need(4)) return "";
        uint32_t n = rd32(&f[p]); p += 4;
        std::string s((const char *)&f[p], n);
        p += (n + 3) & ~3u;
        return s;
    };
    auto skip_atts = [&]() {
        uint32_t tag = rd32(&f[p]), n = rd32(&f[p + 4]); p += 8;
        if (tag == 0) return;
        for (uint32_t i = 0; i < n; ++i) {
            skip_name();
            uint32_t t = rd32(&f[p]), cnt = rd32(&f[p + 4]); p += 8;
            p += ((size_t)cnt * type_size((int)t) + 3) & ~(size_t)3;
        }
    };
    // dims
    uint32_t tag = rd32(&f[p]), nd = rd32(&f[p + 4]); p += 8;
    std::vector<uint32_t> dimlen;
    if (tag == NC_DIMENSION) for (uint32_t i = 0; i < nd; ++i) { skip_name(); dimlen.push_back(rd32(&f[p])); p += 4; }
    skip_atts();
    tag = rd32(&f[p]); uint32_t nv = rd32(&f[p + 4]); p += 8;
    if (tag != NC_VARIABLE) return false;
    for (uint32_t i = 0; i < nv; ++i) {
        std::string nm = skip_name();
        uint32_t ndims = rd32(&f[p]); p += 4;
        bool rec = false;
        for (uint32_t k = 0; k < ndims; ++k) { uint32_t id = rd32(&f[p]); p += 4; if (k == 0 && id < dimlen.size() && dimlen[id] == 0) rec = true; }
        skip_atts();
        p += 4;  // nc_type
        uint32_t vsize = rd32(&f[p]); p += 4;
        uint32_t begin = rd32(&f[p]); p += 4;
        if (rec) { out.rec_begin.emplace_back(nm, begin); out.rec_vsize.push_back(vsize); out.recsize += vsize; }
    }
    return true;
}

}  // namespace

namespace {

// The Moorings schema of initNetCDF (gridoutput.cpp:805-940), shared by the two container formats
struct Schema {
    std::vector<std::pair<std::string, uint32_t>> dims;  // time (0 = unlimited), nv, x, y: the reference's creation order
    std::vector<Var> V;
    std::vector<Att> gatts;
};
enum { D_TIME = 0, D_NV = 1, D_X = 2, D_Y = 3 };

int build_schema(Schema &S, int32_t ncols, int32_t nrows, int32_t nvars, const nxs_mooring_var *vars, float miss_val,
                 double averaging_period, const nxs_mooring_proj *proj) {
    S.dims = {{"time", 0}, {"nv", 2}, {"x", (uint32_t)ncols}, {"y", (uint32_t)nrows}};
    std::vector<Var> &V = S.V;
    std::string cm_time = "time: point ";
    if (averaging_period > 0) {  // gridoutput.cpp:887-894 (boost::format %1% of a double)
        char b[64]; snprintf(b, sizeof b, "%g", 24 * averaging_period);
        cm_time = std::string("time: mean (interval: ") + b + " hours) ";
    }
    if (proj) {  // gridoutput.cpp:943-980
        Var v; v.name = "Polar_Stereographic_Grid"; v.type = NC_INT;
        char p4[256];
        snprintf(p4, sizeof p4, "+proj=stere +a=%g +b=%g +lat_0=%g +lat_ts=%g +lon_0=%g", proj->semi_major_axis, proj->semi_minor_axis,
                 proj->lat0, proj->lat_ts, proj->rotation);
        v.atts = {att_text("grid_mapping_name", "polar_stereographic"), att_float("false_easting", proj->false_easting ? 1 : 0),
                  att_float("false_northing", proj->false_easting ? 1 : 0), att_float("semi_major_axis", proj->semi_major_axis),
                  att_float("semi_minor_axis", proj->semi_minor_axis), att_float("straight_vertical_longitude_from_pole", proj->rotation),
                  att_float("latitude_of_projection_origin", proj->lat0), att_float("standard_parallel", proj->lat_ts),
                  att_text("proj4_string", p4)};
        V.push_back(v);
    }
    {
        Var t; t.name = "time"; t.type = NC_DOUBLE; t.dims = {D_TIME}; t.record = true;
        t.atts = {att_text("standard_name", "time"), att_text("long_name", "simulation time"), att_text("units", "days since 1900-01-01 00:00:00"),
                  att_text("calendar", "standard"), att_text("bounds", "time_bnds")};
        V.push_back(t);
        Var tb; tb.name = "time_bnds"; tb.type = NC_DOUBLE; tb.dims = {D_TIME, D_NV}; tb.record = true;
        tb.atts = {att_text("units", "days since 1900-01-01 00:00:00")};
        V.push_back(tb);
        Var lo; lo.name = "longitude"; lo.type = NC_FLOAT; lo.dims = {D_Y, D_X};
        lo.atts = {att_text("standard_name", "longitude"), att_text("long_name", "longitude"), att_text("units", "degrees_east")};
        V.push_back(lo);
        Var la; la.name = "latitude"; la.type = NC_FLOAT; la.dims = {D_Y, D_X};
        la.atts = {att_text("standard_name", "latitude"), att_text("long_name", "latitude"), att_text("units", "degrees_north")};
        V.push_back(la);
    }
    for (int i = 0; i < nvars; ++i) {
        if (!vars[i].name) return fail(NXS_ERR_INVALID, "variable %d has no name", i);
        Var d; d.name = vars[i].name; d.type = NC_FLOAT; d.dims = {D_TIME, D_Y, D_X}; d.record = true;
        auto s = [](const char *c) { return std::string(c ? c : ""); };
        d.atts = {att_text("standard_name", s(vars[i].standard_name)), att_text("long_name", s(vars[i].long_name)),
                  att_text("coordinates", "latitude longitude"), att_text("units", s(vars[i].units)),
                  att_text("cell_methods", cm_time + s(vars[i].cell_methods)), att_float("_FillValue", miss_val)};
        V.push_back(d);
    }
    S.gatts = {att_text("Conventions", "CF-1.6"), att_text("institution", "NERSC, Jahnebakken 3, N-5007 Bergen, Norway"),
                                    att_text("source", "neXtSIM model fields")};
    return NXS_OK;
}

}  // namespace

#include "nxs_io_nc4.inl"

extern "C" {

static int create_classic(const char *path, int32_t ncols, int32_t nrows, const float *lon, const float *lat, Schema &S) {
    const auto &dims = S.dims;
    std::vector<Var> &V = S.V;
    const std::vector<Att> &gatts = S.gatts;
    // sizes
    for (Var &v : V) {
        uint64_t n = 1;
        for (size_t k = 0; k < v.dims.size(); ++k) if (!(k == 0 && v.record)) n *= dims[v.dims[k]].second;
        v.vsize = (uint32_t)((n * type_size(v.type) + 3) & ~(uint64_t)3);
    }
    auto build_header = [&](Buf &h) {
        h.bytes("CDF\x01", 4); h.u32(0);  // numrecs
        h.u32(NC_DIMENSION); h.u32((uint32_t)dims.size());
        for (auto &d : dims) { h.name(d.first); h.u32(d.second); }
        put_atts(h, gatts);
        h.u32(NC_VARIABLE); h.u32((uint32_t)V.size());
        for (Var &v : V) {
            h.name(v.name); h.u32((uint32_t)v.dims.size());
            for (int id : v.dims) h.u32((uint32_t)id);
            put_atts(h, v.atts);
            h.u32((uint32_t)v.type); h.u32(v.vsize); h.u32(v.begin);
        }
    };
    Buf probe; build_header(probe);
    uint32_t off = (uint32_t)probe.b.size();
    for (Var &v : V) if (!v.record) { v.begin = off; off += v.vsize; }
    for (Var &v : V) if (v.record) { v.begin = off; off += v.vsize; }
    Buf h; build_header(h);
    FILE *f = fopen(path, "wb");
    if (!f) return fail(NXS_ERR_INVALID, "cannot open %s", path);
    fwrite(h.b.data(), 1, h.b.size(), f);
    // fixed-size data in definition order: [projection int], longitude, latitude
    Buf d;
    for (Var &v : V) {
        if (v.record) continue;
        if (v.name == "Polar_Stereographic_Grid") d.u32(0);
        else { const float *src = (v.name == "longitude") ? lon : lat; for (int64_t i = 0; i < (int64_t)nrows * ncols; ++i) d.f32(src[i]); }
    }
    fwrite(d.b.data(), 1, d.b.size(), f);
    fclose(f);
    return NXS_OK;
}

static int append_classic(const char *path, double timestamp, double averaging_period, int32_t nvars, const float *const *data) {
    FILE *f = fopen(path, "rb+");
    if (!f) return fail(NXS_ERR_INVALID, "cannot open %s", path);
    std::vector<unsigned char> head(1 << 16);
    size_t got = fread(head.data(), 1, head.size(), f);
    head.resize(got);
    Parsed P;
    if (!parse_header(head, P)) { fclose(f); return fail(NXS_ERR_INVALID, "%s is not a NetCDF classic file written by nxs_moorings_create", path); }
    if ((int)P.rec_begin.size() != nvars + 2) { fclose(f); return fail(NXS_ERR_INVALID, "file holds %zu record variables, caller passed %d fields", P.rec_begin.size() - 2, nvars); }
    const uint64_t rec_off = (uint64_t)P.numrecs * P.recsize;
    for (size_t i = 0; i < P.rec_begin.size(); ++i) {
        Buf b;
        if (i == 0) b.f64(timestamp);                                                                 // time
        else if (i == 1) { b.f64(timestamp - 0.5 * averaging_period); b.f64(timestamp + 0.5 * averaging_period); }  // time_bnds
        else {
            const float *src = data[i - 2];
            const size_t n = P.rec_vsize[i] / 4;
            b.b.reserve(n * 4);
            for (size_t k = 0; k < n; ++k) b.f32(src[k]);
        }
        fseek(f, (long)(P.rec_begin[i].second + rec_off), SEEK_SET);
        fwrite(b.b.data(), 1, b.b.size(), f);
    }
    Buf n; n.u32(P.numrecs + 1);
    fseek(f, 4, SEEK_SET);
    fwrite(n.b.data(), 1, 4, f);
    fclose(f);
    return NXS_OK;
}


int nxs_moorings_file_format(const char *path) try {
    if (!path) return fail(NXS_ERR_INVALID, "NULL path");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(NXS_ERR_INVALID, "cannot open %s", path);
    unsigned char m[8] = {0};
    const size_t got = fread(m, 1, 8, f);
    fclose(f);
    if (got >= 4 && !memcmp(m, "CDF\x01", 4)) return NXS_NC_CLASSIC;
    if (got >= 8 && !memcmp(m, "\x89HDF\r\n\x1a\n", 8)) return NXS_NC_NETCDF4;
    return fail(NXS_ERR_INVALID, "%s is neither a NetCDF classic nor a NetCDF-4 (HDF5) file", path);
} catch (...) { return entry_caught("nxs_moorings_file_format"); }

int nxs_moorings_create_format(const char *path, int32_t ncols, int32_t nrows, const float *lon, const float *lat, int32_t nvars,
                               const nxs_mooring_var *vars, float miss_val, double averaging_period, const nxs_mooring_proj *proj,
                               int32_t format) try {
    if (!path || !lon || !lat || ncols < 1 || nrows < 1 || nvars < 0 || (nvars > 0 && !vars)) return fail(NXS_ERR_INVALID, "bad moorings arguments");
    if (format != NXS_NC_AUTO && format != NXS_NC_CLASSIC && format != NXS_NC_NETCDF4) return fail(NXS_ERR_INVALID, "unknown format %d", format);
    Schema S;
    if (int rc = build_schema(S, ncols, nrows, nvars, vars, miss_val, averaging_period, proj)) return rc;
    if (format == NXS_NC_AUTO) format = nc4::available() ? NXS_NC_NETCDF4 : NXS_NC_CLASSIC;
    if (format == NXS_NC_NETCDF4) return nc4::create(path, ncols, nrows, lon, lat, S, miss_val);
    return create_classic(path, ncols, nrows, lon, lat, S);
} catch (...) { return entry_caught("nxs_moorings_create_format"); }

int nxs_moorings_create(const char *path, int32_t ncols, int32_t nrows, const float *lon, const float *lat, int32_t nvars,
                        const nxs_mooring_var *vars, float miss_val, double averaging_period, const nxs_mooring_proj *proj) try {
    return nxs_moorings_create_format(path, ncols, nrows, lon, lat, nvars, vars, miss_val, averaging_period, proj, NXS_NC_AUTO);
} catch (...) { return entry_caught("nxs_moorings_create"); }

int nxs_moorings_append(const char *path, double timestamp, double averaging_period, int32_t nvars, const float *const *data) try {
    if (!path || nvars < 0 || (nvars > 0 && !data)) return fail(NXS_ERR_INVALID, "bad moorings arguments");
    const int fmt = nxs_moorings_file_format(path);
    if (fmt < 0) return fmt;
    return fmt == NXS_NC_NETCDF4 ? nc4::append(path, timestamp, averaging_period, nvars, data) : append_classic(path, timestamp, averaging_period, nvars, data);
} catch (...) { return entry_caught("nxs_moorings_append"); }

}  // extern "C"
