"""Output writers.  No reference binary can be built here (exporter.cpp needs
Boost, gridoutput.cpp needs netcdf-cxx4), so the writers are checked against the file formats themselves:
the Exporter's record layout (core/src/exporter.cpp:30-61, 130-189) by reading the bytes back, the Moorings
file with an independent NetCDF reader (scipy.io.netcdf_file) against the schema of
model/gridoutput.cpp:805-940."""
import os

import numpy as np
import pytest

from nextsim_amd import io as nio


def test_exporter_record_layout(tmp_path):
    rng = np.random.default_rng(0)
    x = rng.standard_normal(50) * 1e5; y = rng.standard_normal(50) * 1e5
    ids = np.arange(1, 51, dtype=np.int32); el = rng.integers(1, 51, 3 * 80).astype(np.int32)
    b, d = str(tmp_path / "mesh_1.bin"), str(tmp_path / "mesh_1.dat")
    e = nio.Exporter(b, d, "float"); e.writeMesh(x, y, ids, el); e.close()
    rec = nio.read_exported(b, d)
    assert list(rec) == ["Elements", "id", "Nodes_x", "Nodes_y"]               # exporter.cpp:82-128 order
    assert np.array_equal(rec["Elements"], el) and np.array_equal(rec["id"], ids)
    assert rec["Nodes_x"].dtype == np.float64 and np.array_equal(rec["Nodes_x"], x)   # coordinates stay double
    lines = open(d).read().splitlines()
    assert lines[0] == f"Elements int 240 {el.min()} {el.max()}"
    assert lines[2] == "Nodes_x double 50 %g %g" % (x.min(), x.max())

    conc = rng.random(80); t = np.array([42000.25])
    b, d = str(tmp_path / "field_1.bin"), str(tmp_path / "field_1.dat")
    e = nio.Exporter(b, d, "float")
    e.writeField(t, "Time"); e.writeField(conc, "M_conc"); e.writeField(np.array([3, 1, 7, 2], np.int32), "Misc_int")
    e.writeField(np.zeros(0), "Empty"); e.close()
    rec = nio.read_exported(b, d)
    assert rec["Time"].dtype == np.float64 and rec["Time"][0] == 42000.25        # Time is always double (:143-145)
    assert rec["M_conc"].dtype == np.float32 and np.array_equal(rec["M_conc"], conc.astype(np.float32))
    assert rec["Misc_int"].dtype == np.int32 and rec["Empty"].size == 0
    lines = open(d).read().splitlines()
    assert lines[1] == "M_conc float 80 %g %g" % (conc.min(), conc.max())        # min/max of the doubles
    assert lines[3] == "Empty float 0 0 0"
    # double precision round trip is exact (restart files)
    b, d = str(tmp_path / "r.bin"), str(tmp_path / "r.dat")
    e = nio.Exporter(b, d, "double"); e.writeField(conc, "M_conc"); e.close()
    assert np.array_equal(nio.read_exported(b, d)["M_conc"], conc)
    with pytest.raises(Exception):
        nio.Exporter(b, d, "half")


def test_moorings_netcdf_schema_and_records(tmp_path):
    from scipy.io import netcdf_file
    nrows, ncols = 7, 11
    lon = np.linspace(-180, 180, nrows * ncols, dtype=np.float32).reshape(nrows, ncols)
    lat = np.linspace(60, 90, nrows * ncols, dtype=np.float32).reshape(nrows, ncols)
    variables = [dict(name="sic", standard_name="sea_ice_area_fraction", long_name="Sea Ice Concentration", units="1", cell_methods="area: mean"),
                 dict(name="sit", standard_name="sea_ice_thickness", long_name="Sea Ice Thickness", units="m", cell_methods="area: mean")]
    proj = dict(semi_major_axis=6378273.0, semi_minor_axis=6356889.449, lat0=90.0, lat_ts=60.0, rotation=-45.0, false_easting=0)
    path = str(tmp_path / "Moorings.nc")
    nio.moorings_create(path, lon, lat, variables, miss_val=-1e14, averaging_period=0.125, proj=proj, format=nio.NC_CLASSIC)
    assert nio.moorings_file_format(path) == nio.NC_CLASSIC
    rng = np.random.default_rng(1)
    recs = []
    for k in range(3):
        f = [rng.random((nrows, ncols)).astype(np.float32), rng.random((nrows, ncols)).astype(np.float32)]
        f[1][0, 0] = -1e14
        nio.moorings_append(path, 42000.0 + 0.125 * k, f, averaging_period=0.125)
        recs.append(f)
    nc = netcdf_file(path, "r", mmap=False)
    assert nc.dimensions == {"time": None, "nv": 2, "x": ncols, "y": nrows}               # gridoutput.cpp:862-884
    assert nc.Conventions == b"CF-1.6" and nc.source == b"neXtSIM model fields"
    t = nc.variables["time"]
    assert t.units == b"days since 1900-01-01 00:00:00" and t.calendar == b"standard" and t.bounds == b"time_bnds"
    assert np.array_equal(t[:], 42000.0 + 0.125 * np.arange(3))
    assert np.array_equal(nc.variables["time_bnds"][:], np.stack([t[:] - 0.0625, t[:] + 0.0625], 1))
    assert nc.variables["longitude"].dimensions == ("y", "x") and np.array_equal(nc.variables["longitude"][:], lon)
    assert np.array_equal(nc.variables["latitude"][:], lat) and nc.variables["latitude"].units == b"degrees_north"
    sic = nc.variables["sic"]
    assert sic.dimensions == ("time", "y", "x") and sic.typecode() == "f"
    assert sic.standard_name == b"sea_ice_area_fraction" and sic.coordinates == b"latitude longitude"
    assert sic.cell_methods == b"time: mean (interval: 3 hours) area: mean"              # gridoutput.cpp:887-894
    assert np.float32(sic._FillValue) == np.float32(-1e14)
    for k in range(3):
        assert np.array_equal(sic[k], recs[k][0]) and np.array_equal(nc.variables["sit"][k], recs[k][1])
    pv = nc.variables["Polar_Stereographic_Grid"]
    assert pv.grid_mapping_name == b"polar_stereographic" and np.float32(pv.standard_parallel) == np.float32(60.0)
    assert pv.proj4_string.startswith(b"+proj=stere +a=6.37827e+06")
    nc.close()
    # snapshots: "time: point " and no projection variable
    path2 = str(tmp_path / "snap.nc")
    nio.moorings_create(path2, lon, lat, variables[:1], format=nio.NC_CLASSIC)
    nio.moorings_append(path2, 1.0, [recs[0][0]])
    nc = netcdf_file(path2, "r", mmap=False)
    assert nc.variables["sic"].cell_methods == b"time: point area: mean" and "Polar_Stereographic_Grid" not in nc.variables
    assert nc.variables["sic"].shape == (1, nrows, ncols)
    nc.close()
    with pytest.raises(Exception):
        nio.moorings_append(path2, 2.0, [recs[0][0], recs[0][1]])   # wrong number of fields


def _h5dump():
    import shutil
    for cand in (shutil.which("h5dump"), "/opt/conda/bin/h5dump"):
        if cand and os.path.exists(cand):
            return cand
    return None


@pytest.mark.skipif(_h5dump() is None, reason="no h5dump (HDF5 tools) on this box: no independent reader for NetCDF-4")
def test_moorings_netcdf4_as_the_reference_writes_it(tmp_path):
    """The default container is NetCDF-4 (gridoutput.cpp:857: netCDF::NcFile(..., replace)), written through the HDF5 C library
    following the netCDF-4 conventions.  Read back with h5dump -- HDF5's own tool, independent of the writer: dimensions as
    dimension scales with _Netcdf4Dimid and netCDF's naming, the scales attached to every variable, the unlimited time
    dimension, the attributes of the reference's schema, and the data, record by record."""
    import re, subprocess
    nrows, ncols = 5, 9
    lon = np.linspace(-180, 180, nrows * ncols, dtype=np.float32).reshape(nrows, ncols)
    lat = np.linspace(60, 90, nrows * ncols, dtype=np.float32).reshape(nrows, ncols)
    variables = [dict(name="sic", standard_name="sea_ice_area_fraction", long_name="Sea Ice Concentration", units="1", cell_methods="area: mean"),
                 dict(name="sit", standard_name="sea_ice_thickness", long_name="Sea Ice Thickness", units="m", cell_methods="area: mean")]
    proj = dict(semi_major_axis=6378273.0, semi_minor_axis=6356889.449, lat0=90.0, lat_ts=60.0, rotation=-45.0, false_easting=0)
    path = str(tmp_path / "Moorings.nc")
    nio.moorings_create(path, lon, lat, variables, miss_val=-1e14, averaging_period=0.125, proj=proj)     # NC_AUTO
    assert nio.moorings_file_format(path) == nio.NC_NETCDF4
    rng = np.random.default_rng(1)
    recs = []
    for k in range(3):
        f = [rng.random((nrows, ncols)).astype(np.float32), rng.random((nrows, ncols)).astype(np.float32)]
        nio.moorings_append(path, 42000.0 + 0.125 * k, f, averaging_period=0.125)
        recs.append(f)
    h5 = _h5dump()
    head = subprocess.check_output([h5, "-H", "-A", path], text=True)

    def block(name):        # the text of DATASET "name" { ... } (brace matching)
        i = head.index(f'DATASET "{name}"')
        depth, j = 0, head.index("{", i)
        for j in range(j, len(head)):
            depth += head[j] == "{"; depth -= head[j] == "}"
            if depth == 0:
                break
        return head[i:j + 1]

    order = re.findall(r'^   DATASET "([^"]+)"', head, re.M)
    # (h5dump lists by name; the creation order the netCDF library uses for ids is checked through _Netcdf4Dimid below)
    assert set(order) == {"Polar_Stereographic_Grid", "time", "nv", "time_bnds", "x", "y", "longitude", "latitude", "sic", "sit"}
    t = block("time")
    assert "H5S_UNLIMITED" in t and "H5T_IEEE_F64LE" in t and '"DIMENSION_SCALE"' in t and re.search(r'ATTRIBUTE "NAME".*?"time"', t, re.S)
    assert re.search(r'ATTRIBUTE "_Netcdf4Dimid".*?\(0\): 0', t, re.S) and "REFERENCE_LIST" in t
    for name, dimid, n in (("nv", 1, 2), ("x", 2, ncols), ("y", 3, nrows)):
        b = block(name)
        assert '"DIMENSION_SCALE"' in b and "H5T_IEEE_F32BE" in b
        assert f"This is a netCDF dimension but not a netCDF variable.{n:10d}" in b
        assert re.search(r'ATTRIBUTE "_Netcdf4Dimid".*?\(0\): %d' % dimid, b, re.S)
        assert f"( {n} ) / ( {n} )" in b
    sic = block("sic")
    assert f"( 3, {nrows}, {ncols} ) / ( H5S_UNLIMITED, {nrows}, {ncols} )" in sic and "H5T_IEEE_F32LE" in sic
    assert "DIMENSION_LIST" in sic and '"sea_ice_area_fraction"' in sic and '"latitude longitude"' in sic
    assert '"time: mean (interval: 3 hours) area: mean"' in sic and re.search(r'ATTRIBUTE "_FillValue".*?-1e\+14', sic, re.S)
    assert f"( 3, 2 ) / ( H5S_UNLIMITED, 2 )" in block("time_bnds")
    assert f"( {nrows}, {ncols} ) / ( {nrows}, {ncols} )" in block("longitude") and '"degrees_east"' in block("longitude")
    pv = block("Polar_Stereographic_Grid")
    assert '"polar_stereographic"' in pv and "+proj=stere +a=6.37827e+06" in pv and "H5T_STD_I32LE" in pv
    assert re.search(r'ATTRIBUTE "Conventions".*?"CF-1.6"', head, re.S) and '"neXtSIM model fields"' in head and "_NCProperties" in head

    def data(name, dtype, count):
        raw = subprocess.check_output([h5, "-d", "/" + name, "-b", "LE", "-o", str(tmp_path / "raw.bin"), path], text=True)
        return np.fromfile(tmp_path / "raw.bin", dtype)[:count]
    assert np.array_equal(data("time", "<f8", 3), 42000.0 + 0.125 * np.arange(3))
    assert np.array_equal(data("time_bnds", "<f8", 6).reshape(3, 2), np.stack([42000.0 + 0.125 * np.arange(3) - 0.0625, 42000.0 + 0.125 * np.arange(3) + 0.0625], 1))
    assert np.array_equal(data("longitude", "<f4", nrows * ncols).reshape(nrows, ncols), lon)
    assert np.array_equal(data("latitude", "<f4", nrows * ncols).reshape(nrows, ncols), lat)
    got = data("sic", "<f4", 3 * nrows * ncols).reshape(3, nrows, ncols)
    got2 = data("sit", "<f4", 3 * nrows * ncols).reshape(3, nrows, ncols)
    for k in range(3):
        assert np.array_equal(got[k], recs[k][0]) and np.array_equal(got2[k], recs[k][1])
    with pytest.raises(Exception):
        nio.moorings_append(path, 43000.0, [recs[0][0]])          # wrong number of fields
    # the same calls give the classic container on request, and append tells the two apart by itself
    p3 = str(tmp_path / "classic.nc")
    nio.moorings_create(p3, lon, lat, variables, format=nio.NC_CLASSIC)
    nio.moorings_append(p3, 1.0, recs[0])
    assert nio.moorings_file_format(p3) == nio.NC_CLASSIC and open(p3, "rb").read(4) == b"CDF\x01" and open(path, "rb").read(4) == b"\x89HDF"


def test_library_loader_equals_the_independent_reader(tmp_path):
    """nxs_exporter_load (readRecord + loadFile, exporter.cpp:183-222) against the pure-numpy reader."""
    rng = np.random.default_rng(1)
    b, d = str(tmp_path / "field_7.bin"), str(tmp_path / "field_7.dat")
    e = nio.Exporter(b, d, "float")
    e.writeField(np.array([1.5]), "Time"); e.writeField(rng.random(33), "M_thick"); e.writeField(np.arange(5, dtype=np.int32), "Misc_int")
    e.writeField(rng.random(4), "M_thick")          # a second record of the same name: the first one wins (emplace)
    e.close()
    ind = nio.read_exported(b, d)
    got = nio.load_exported(b, d)
    assert list(got) == ["Time", "M_thick", "Misc_int"]
    assert got["M_thick"].size == 33 and got["M_thick"].dtype == np.float64   # float records are widened
    assert np.array_equal(got["Misc_int"], ind["Misc_int"]) and got["Time"][0] == 1.5
    with pytest.raises(Exception):
        nio.load_exported(str(tmp_path / "missing.bin"), d)
    open(b, "r+b").truncate(20)                      # a truncated payload is an error, not garbage
    with pytest.raises(Exception):
        nio.load_exported(b, d)


def test_restart_files_round_trip_in_the_reference_layout(tmp_path):
    """writeRestart (FE.cpp:9518-9695): file names, record names and ORDER, double precision; then the pair is
    read back (readRestart's file part) and every array must come back bit for bit."""
    import cases
    gm, p, g, lms, fields = cases.make_case("tiny")
    lm, f = lms[0], fields[0]
    nn, ne = lm.num_nodes, lm.num_elements
    ids = np.arange(1, nn + 1, dtype=np.int32)
    names = ["M_conc", "M_thick", "M_snow_thick", "M_sigma_0", "M_sigma_1", "M_sigma_2", "M_damage", "M_ridge_ratio"]
    keys = ["conc", "thick", "snow_thick", "sigma0", "sigma1", "sigma2", "damage", "ridge_ratio"]
    elt = {n: f[k] for n, k in zip(names, keys)}
    dirichlet = np.flatnonzero(lm.mask_dirichlet[:nn]).astype(np.int32) + 1
    prev = np.arange(1, nn + 1, dtype=np.float64)
    misc = [17, 10000, 3, 2]
    nio.write_restart(tmp_path, "final", lm.coord_x, lm.coord_y, ids, lm.indices, misc, dirichlet, 42005.125, elt, f["VT"], f["UM"], f["UT"], prev)
    for fn in ("mesh_final.bin", "mesh_final.dat", "field_final.bin", "field_final.dat"):
        assert (tmp_path / fn).exists()
    order = [ln.split()[0] for ln in open(tmp_path / "field_final.dat")]
    assert order == ["Misc_int", "M_dirichlet_flags", "Time"] + names + ["M_VT", "M_UM", "M_UT", "PreviousNumbering"]
    types = [ln.split()[1] for ln in open(tmp_path / "field_final.dat")]
    assert types[:3] == ["int", "int", "double"] and set(types[3:]) == {"double"}
    assert [ln.split()[0] for ln in open(tmp_path / "mesh_final.dat")] == ["Elements", "id", "Nodes_x", "Nodes_y"]
    mesh, field = nio.read_restart(tmp_path, "final")
    assert np.array_equal(mesh["Elements"], lm.indices.ravel()) and np.array_equal(mesh["id"], ids)
    assert np.array_equal(mesh["Nodes_x"], lm.coord_x) and np.array_equal(mesh["Nodes_y"], lm.coord_y)
    assert list(field["Misc_int"]) == misc and field["Time"][0] == 42005.125
    assert np.array_equal(field["M_dirichlet_flags"], dirichlet)
    for n, k in zip(names, keys):
        assert np.array_equal(field[n], f[k]), n
    for n, k in (("M_VT", "VT"), ("M_UM", "UM"), ("M_UT", "UT")):
        assert field[n].size == 2 * nn and np.array_equal(field[n], f[k])
    assert np.array_equal(field["PreviousNumbering"], prev)
    # the independent reader sees the same bytes
    ind = nio.read_exported(str(tmp_path / "field_final.bin"), str(tmp_path / "field_final.dat"))
    assert all(np.array_equal(ind[k], field[k]) for k in field)
    # a pair that lacks what readRestart needs is refused
    e = nio.Exporter(str(tmp_path / "field_bad.bin"), str(tmp_path / "field_bad.dat"), "double"); e.writeField(np.zeros(3), "Time"); e.close()
    e = nio.Exporter(str(tmp_path / "mesh_bad.bin"), str(tmp_path / "mesh_bad.dat"), "double"); e.writeMesh(lm.coord_x, lm.coord_y, ids, lm.indices); e.close()
    with pytest.raises(Exception, match="lacks"):
        nio.read_restart(tmp_path, "bad")


def test_io_symbols_are_declared():
    import os, re
    from nextsim_amd import dynamics
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "nxs_io.h")).read()
    declared = set(re.findall(r"NXS_IO_API\s+(?:const\s+char\s*\*|int)\s*(nxs_\w+)\s*\(", text))
    assert declared == set(nio.IO_EXPORTS)
    L = dynamics.load_library()
    for n in declared:
        assert hasattr(L, n)
