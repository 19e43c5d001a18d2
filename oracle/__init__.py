"""CPU oracle -- test infrastructure only (see oracle/dyn_ref.h)."""
