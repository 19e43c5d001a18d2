"""nextsim_amd -- the MI355X-native dynamics hot path of neXtSIM behind a C ABI.

Only what the path needs: csrc/ (HIP kernels + C ABI), dynamics.py (host mirror of the reference's
call surface), mesh.py / forcing.py (synthetic meshes, partitions and forcing for tests and bench).
"""
__all__ = ["dynamics", "mesh", "forcing"]
