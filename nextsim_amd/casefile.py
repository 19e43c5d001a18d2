"""Flat binary dump of one rank's case (mesh, parameters, state, forcing) for hosts without Python:
read by examples/nextsim_toy.cpp.  Layout: b"NXSCASE1", u64 sizeof(params), params, then records
(u32 name length, name, u32 kind {0 f64, 1 i32, 2 u8}, u64 count, payload)."""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import _abi


def write_case(path: str, lm, params: _abi.Params, fields: dict) -> None:
    with open(path, "wb") as f:
        f.write(b"NXSCASE1")
        f.write(struct.pack("<Q", C.sizeof(_abi.Params)))
        f.write(bytes(params))

        def rec(name, arr):
            arr = np.ascontiguousarray(arr)
            kind = {np.dtype(np.float64): 0, np.dtype(np.int32): 1, np.dtype(np.uint8): 2}[arr.dtype]
            nb = name.encode()
            f.write(struct.pack("<I", len(nb))); f.write(nb); f.write(struct.pack("<IQ", kind, arr.size)); f.write(arr.tobytes())

        rec("sizes", np.array([lm.num_nodes, lm.num_elements, lm.local_ndof, lm.local_nelements], np.int32))
        rec("indices", lm.indices); rec("ghost_nodes", lm.ghost_nodes)
        rec("coord_x", lm.coord_x); rec("coord_y", lm.coord_y); rec("lat", lm.lat)
        rec("mask_dirichlet", lm.mask_dirichlet); rec("neumann_flags", lm.neumann_flags)
        # halo lists of initUpdateGhosts() (FE.hpp:615-618), for multi-rank hosts (examples/nextsim_mpi.cpp)
        rec("halo_rank", np.array([lm.rank, lm.nranks], np.int32))
        for k in ("send_procs", "send_offsets", "send_index", "recv_procs", "recv_offsets", "recv_index"):
            rec(k, np.ascontiguousarray(getattr(lm, k), np.int32))
        for k, v in fields.items():
            rec(k, np.asarray(v, np.float64))
