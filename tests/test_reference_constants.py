"""The numbers the path takes as literals, against the REFERENCE's: tests/golden/reference_constants.json holds what a g++-compiled
translation unit that #includes model/constants.hpp, model/enums.hpp and contrib/bamg/include/OppositeAngle.h printed, plus every
default_value( ) of model/options.cpp evaluated by the compiler (tests/golden/make_reference_constants.py, run in the build container;
only the JSON travels).  Compared bit for bit with: the constants oracle/dyn_ref.c computes with, the constants compiled into the
kernels (nxs_dyn_physical_constants), include/nxs_dyn.h's enums, and the three default-parameter sets (library, oracle, Python)."""
import ctypes as C
import json
import os
import re

import pytest

from nextsim_amd import _abi, dynamics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFC = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_constants.json")))


def ref(section, name):
    return float.fromhex(REFC[section][name]["hex"])


def opt(name):
    o = REFC["options"][name]
    return float.fromhex(o["hex"]) if o["type"] == "double" else o["value"]


# NXS_CONST_* order of include/nxs_dyn.h
EXPECTED_CONSTANTS = [ref("physical", "rhoi"), ref("physical", "rhow"), ref("physical", "rhos"), ref("physical", "rhoa"),
                      ref("physical", "gravity"), ref("physical", "omega"), float.fromhex(REFC["pi"]["hex"]), ref("members", "days_in_sec")]


def test_fixture_is_what_the_generator_describes():
    assert len(REFC["physical"]) >= 20 and len(REFC["options"]) >= 150 and "DynamicsType" in REFC["enums"]
    assert REFC["physical"]["rhoi"]["value"] == 917.0 and REFC["options"]["dynamics.substeps"]["value"] == 120


def test_library_constants_are_the_references():
    L = dynamics.load_library()
    out = (C.c_double * 8)()
    assert L.nxs_dyn_physical_constants(out, 8) == 0
    assert [float(v).hex() for v in out] == [v.hex() for v in EXPECTED_CONSTANTS]


def test_oracle_constants_are_the_references():
    from oracle import pyoracle as O
    L = O.lib()
    out = (C.c_double * 8)()
    L.ref_physical_constants.argtypes = [C.POINTER(C.c_double)]
    L.ref_physical_constants.restype = None
    L.ref_physical_constants(out)
    assert [float(v).hex() for v in out] == [v.hex() for v in EXPECTED_CONSTANTS]


def test_header_enums_are_the_references():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "nxs_dyn.h")).read(), flags=re.S)
    enums = {}
    for body in re.findall(r"enum\s*\{([^}]*)\}", text):
        nxt = 0
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            name, _, val = item.partition("=")
            nxt = int(val) if val.strip() else nxt
            enums[name.strip()] = nxt
            nxt += 1
    R = REFC["enums"]
    dyn = R["DynamicsType"]
    assert (enums["NXS_DYN_BBM"], enums["NXS_DYN_NO_MOTION"], enums["NXS_DYN_FREE_DRIFT"], enums["NXS_DYN_EVP"], enums["NXS_DYN_MEVP"]) == \
           (dyn["BBM"], dyn["NO_MOTION"], dyn["FREE_DRIFT"], dyn["EVP"], dyn["mEVP"])
    assert (enums["NXS_BASAL_NONE"], enums["NXS_BASAL_LEMIEUX"]) == (R["BasalStressType"]["NONE"], R["BasalStressType"]["LEMIEUX"])
    assert (enums["NXS_ICECAT_CLASSIC"], enums["NXS_ICECAT_YOUNG_ICE"]) == (R["IceCategoryType"]["CLASSIC"], R["IceCategoryType"]["YOUNG_ICE"])
    # the Python mirror of the header
    assert (_abi.NXS_DYN_BBM, _abi.NXS_DYN_EVP, _abi.NXS_DYN_MEVP) == (dyn["BBM"], dyn["EVP"], dyn["mEVP"])
    assert (_abi.NXS_BASAL_LEMIEUX, _abi.NXS_ICECAT_YOUNG_ICE) == (R["BasalStressType"]["LEMIEUX"], R["IceCategoryType"]["YOUNG_ICE"])


def expected_default_params():
    """nxs_dyn_params as model/options.cpp's defaults give it (the mapping of option names to members is include/nxs_dyn.h's)."""
    S, R = REFC["string_options"], REFC["enums"]
    pi = float.fromhex(REFC["pi"]["hex"])
    assert S["setup.dynamics-type"] == "bbm" and S["setup.basal_stress-type"] == "lemieux" and S["setup.atmosphere-type"] == "asr"
    return {
        "dtime_step": float(opt("simul.timestep")), "substeps": opt("dynamics.substeps"),
        "dynamics_type": R["DynamicsType"]["BBM"], "basal_stress_type": R["BasalStressType"]["LEMIEUX"],
        # thermo.newice_type == 4 -> YOUNG_ICE (FE.cpp:1206-1209)
        "ice_cat_type": R["IceCategoryType"]["YOUNG_ICE"] if opt("thermo.newice_type") == 4 else R["IceCategoryType"]["CLASSIC"],
        "newice_type": opt("thermo.newice_type"), "equal_ridging": opt("age.equal_ridging"), "use_young_ice_in_myi_reset": opt("age.include_young_ice"),
        "young": opt("dynamics.young"), "nu0": opt("dynamics.nu0"), "tan_phi": opt("dynamics.tan_phi"), "compr_strength": opt("dynamics.compr_strength"),
        "compaction_param": opt("dynamics.compaction_param"), "undamaged_time_relaxation_sigma": opt("dynamics.undamaged_time_relaxation_sigma"),
        "exponent_relaxation_sigma": opt("dynamics.exponent_relaxation_sigma"), "compression_factor": opt("dynamics.compression_factor"),
        "exponent_compression_factor": opt("dynamics.exponent_compression_factor"), "min_h": opt("dynamics.min_h"), "min_c": opt("dynamics.min_c"),
        "quad_drag_coef_water": opt("dynamics.quad_drag_coef_water"), "lin_drag_coef_water": opt("dynamics.lin_drag_coef_water"),
        "quad_drag_coef_air": opt("dynamics.ASR_quad_drag_coef_air"),    # setup.atmosphere-type = asr (FE.cpp:1286-1287)
        "lin_drag_coef_air": opt("dynamics.lin_drag_coef_air"),
        "ocean_turning_angle_rad": (pi / 180.) * opt("dynamics.oceanic_turning_angle"),   # FE.cpp:1167-1172
        "basal_k1": opt("dynamics.Lemieux_basal_k1"), "basal_k2": opt("dynamics.Lemieux_basal_k2"), "basal_Cb": opt("dynamics.Lemieux_basal_Cb"),
        "basal_u_0": opt("dynamics.Lemieux_basal_u_0"),
        "evp_e": opt("dynamics.evp.e"), "evp_Pstar": opt("dynamics.evp.Pstar"), "evp_C": float(opt("dynamics.evp.C")), "evp_dmin": opt("dynamics.evp.dmin"),
        "mevp_alpha": float(opt("dynamics.mevp.alpha")), "mevp_beta": float(opt("dynamics.mevp.beta")),
        "regrid_angle": opt("numerics.regrid_angle"), "reserved0": 0,
    }


def _as_dict(p):
    return {name: getattr(p, name) for name, _ in _abi.Params._fields_}


def _same(got, want):
    assert set(got) == set(want)
    for k, v in want.items():
        g = got[k]
        assert (float(g).hex() == float(v).hex()) if isinstance(v, float) else (g == v), (k, g, v)


def test_library_default_params_are_options_cpp():
    L = dynamics.load_library()
    p = _abi.Params()
    assert L.nxs_dyn_default_params(C.byref(p)) == 0
    _same(_as_dict(p), expected_default_params())


def test_python_default_params_are_options_cpp():
    from nextsim_amd.forcing import default_params
    _same(_as_dict(default_params()), expected_default_params())


def test_oracle_default_params_are_options_cpp():
    from oracle import pyoracle as O
    p = _abi.Params()
    O.lib().ref_default_params(C.byref(p))
    _same(_as_dict(p), expected_default_params())


def test_hard_coded_sweep_count_is_the_references_default():
    """Q9: FE.cpp:10580 hard-codes 50 sweeps whatever numerics.nit_ow says; the default of that option is the same number."""
    assert opt("numerics.nit_ow") == 50
    text = open(os.path.join(ROOT, "nextsim_amd", "csrc", "nxs_dyn.hip")).read()
    assert re.search(r"NXS_SMOOTH_SWEEPS == 50", text)


@pytest.mark.skipif(not os.path.isdir("/root/reference/model"), reason="the reference is only present in the build container")
def test_fixture_is_current(tmp_path):
    """In the build container: regenerating the fixture from the reference gives the committed file."""
    import subprocess
    import sys
    src = os.path.join(ROOT, "tests", "golden", "make_reference_constants.py")
    code = open(src).read().replace('HERE = os.path.dirname(os.path.abspath(__file__))', f'HERE = {str(tmp_path)!r}')
    subprocess.check_call([sys.executable, "-c", code])
    assert json.load(open(tmp_path / "reference_constants.json")) == REFC
