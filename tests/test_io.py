"""Output writers.  No reference binary can be built here (exporter.cpp needs
Boost, gridoutput.cpp needs netcdf-cxx4), so the writers are checked against the file formats themselves:
the Exporter's record layout (core/src/exporter.cpp:30-61, 130-189) by reading the bytes back, the Moorings
file with an independent NetCDF reader (scipy.io.netcdf_file) against the schema of
model/gridoutput.cpp:805-940."""
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
    nio.moorings_create(path, lon, lat, variables, miss_val=-1e14, averaging_period=0.125, proj=proj)
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
    nio.moorings_create(path2, lon, lat, variables[:1])
    nio.moorings_append(path2, 1.0, [recs[0][0]])
    nc = netcdf_file(path2, "r", mmap=False)
    assert nc.variables["sic"].cell_methods == b"time: point area: mean" and "Polar_Stereographic_Grid" not in nc.variables
    assert nc.variables["sic"].shape == (1, nrows, ncols)
    nc.close()
    with pytest.raises(Exception):
        nio.moorings_append(path2, 2.0, [recs[0][0], recs[0][1]])   # wrong number of fields


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
