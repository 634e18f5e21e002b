"""Translates a stack of reference forces plus per-particle parameters into the elementary
op list the device API runs (one op per force and source), reproducing the source
discovery loops of the reference force functions.

Per-particle double parameters are numpy columns with NaN = not set (or None = not set
anywhere); presence-only parameters are boolean/uint8 columns or None.
"""
import numpy as np

from ._lib import (COORD_JACOBI, COORD_PARTICLE, OP_CENTRAL, OP_GR, OP_GR_FULL, OP_GR_POTENTIAL, OP_J2, OP_J4, OP_MOF,
                   OP_RADIATION, OP_TIDES, Op)


class ForceError(ValueError):
    """A required effect parameter is missing (the reference reports reb_error and returns)."""


def _set_indices(col):
    if col is None:
        return np.zeros(0, dtype=np.int64)
    col = np.asarray(col)
    if col.dtype == np.float64:
        return np.nonzero(~np.isnan(col))[0]
    return np.nonzero(col)[0]


def build_ops(forces, pp, gravity_ignore_terms=0):
    """forces: list of (name, effect_params_dict) in EXECUTION order (the reference runs added
    forces in reverse order of rebx_add_force, src/core.c:868-880 + src/linkedlist.c:33-36).
    pp: dict of per-particle parameter columns (host numpy).  Returns a list of Op."""
    ops = []

    def op(kind, src=-1, iarg=0, a=0.0, b=0.0):
        ops.append(Op(kind, int(src), int(iarg), 0, float(a), float(b)))

    for name, ep in forces:
        if name == "radiation_forces":          # src/radiation_forces.c:100-117
            if "c" not in ep:
                raise ForceError("Need to set speed of light in radiation_forces effect.")
            srcs = _set_indices(pp.get("radiation_source"))
            for s in (srcs if len(srcs) else [0]):
                op(OP_RADIATION, s, 0, ep["c"])
        elif name == "gravitational_harmonics":  # src/gravitational_harmonics.c:87-97,125-140
            Req = pp.get("R_eq")
            for kind, key in ((OP_J2, "J2"), (OP_J4, "J4")):
                col = pp.get(key)
                if col is None or Req is None:
                    continue
                for s in np.nonzero(~np.isnan(col) & ~np.isnan(Req))[0]:
                    op(kind, s, 0, col[s], Req[s])
        elif name == "gr":                       # src/gr.c:173-188
            if "c" not in ep:
                raise ForceError("Need to set speed of light in gr effect.")
            op(OP_GR, 0, ep.get("max_iterations", 10), ep["c"])
        elif name == "gr_potential":             # src/gr_potential.c:89-98
            if "c" not in ep:
                raise ForceError("Need to set speed of light in gr effect.")
            op(OP_GR_POTENTIAL, 0, 0, ep["c"])
        elif name == "gr_full":                  # src/gr_full.c:233-249
            if "c" not in ep:
                raise ForceError("Need to set speed of light in gr effect.")
            op(OP_GR_FULL, -1, 1 if gravity_ignore_terms == 1 else 0, ep["c"])
        elif name == "central_force":            # src/central_force.c:85-95
            A, g = pp.get("Acentral"), pp.get("gammacentral")
            if A is not None and g is not None:
                for s in np.nonzero(~np.isnan(A) & ~np.isnan(g))[0]:
                    op(OP_CENTRAL, s, 0, A[s], g[s])
        elif name == "tides_precession":         # src/tides_precession.c:118-130
            srcs = _set_indices(pp.get("tides_primary"))
            for s in (srcs if len(srcs) else [0]):
                op(OP_TIDES, s)
        elif name == "modify_orbits_forces":     # src/modify_orbits_forces.c:120-129, src/rebxtools.c:49-68
            coords = ep.get("coordinates", COORD_JACOBI)
            src = -1
            if coords == COORD_PARTICLE:
                prim = _set_indices(pp.get("primary"))
                if len(prim) == 0:
                    raise ForceError("Coordinates set to REBX_COORDINATES_PARTICLE, but primary param was not found in any particle.")
                src = prim[0]
            op(OP_MOF, src, coords)
        else:
            raise ForceError("Force '%s' not found in REBOUNDx library." % name)
    return ops
