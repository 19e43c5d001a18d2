"""Shared builders for the tests: a (mesh, params, fields) case on one or several ranks."""
from __future__ import annotations

import numpy as np

from nextsim_amd import _abi, forcing as F, mesh as M

_cache = {}


def global_mesh(kind):
    if kind not in _cache:
        _cache[kind] = M.make_mesh(kind)
    return _cache[kind]


def make_case(kind="small", forcing_kind=None, nparts=1, alea=0.33, **param_over):
    gm = global_mesh(kind)
    forcing_kind = forcing_kind or ("toy" if kind in ("toy", "tiny") else "arctic")
    p = F.default_params(**param_over)
    p, C_fix, C_alea = F.scale_params_to_mesh(p, gm, alea_factor=alea)
    g = F.global_fields(gm, p, forcing_kind, C_fix, C_alea)
    lms = M.localize(gm, nparts)
    fields = [F.localize_fields(g, lm, gm.num_nodes) for lm in lms]
    return gm, p, g, lms, fields


def rel_err(a, b):
    """max |a-b| / max|b| (fields here have a natural scale; pointwise relative error is meaningless
    near zero crossings)."""
    scale = max(float(np.abs(b).max()), 1e-300)
    return float(np.abs(a - b).max()) / scale
