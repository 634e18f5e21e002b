"""Shared helpers for the GPU parity tests: run the same stack of forces through the CPU
checker and through librebxamd (device-resident C ABI) and compare accelerations."""
import json
import os

import numpy as np

import oracle

DYN = ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az")

# The parity bar for fp64 results (stated by the project's north star): <= 1e-12 relative per
# acceleration component.  "Relative" is taken against the component itself, floored at 1e-3
# of the largest component of the same particle's update so that a component that vanishes by
# symmetry (e.g. z in a planar orbit) is not held to digits it does not have.
TOL = 1e-12
FLOOR = 1e-3


def rel_err(got, ref, scale_rows=None):
    """got, ref: arrays (3, n).  Returns per-particle max relative error."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    big = np.abs(ref).max(axis=0)
    if scale_rows is not None:
        big = np.maximum(big, scale_rows)
    den = np.maximum(np.abs(ref), FLOOR * big)
    den = np.where(den == 0.0, 1.0, den)
    return (np.abs(got - ref) / den).max(axis=0)


def unfloored_rel_err(got, ref):
    """|got - ref| / |ref| per component with NO floor (components whose reference value is exactly
    zero are left out); per-particle maximum.  Reported beside rel_err: a component that nearly
    cancels shows here with the full size of the rounding of its terms."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        e = np.where(ref != 0.0, np.abs(got - ref) / np.abs(ref), 0.0)
    return e.max(axis=0)


def err_vs_exact(got, hi, lo):
    """|got - (hi + lo)| per component, for a binary128 value carried as a pair of doubles."""
    return np.abs((np.asarray(got, dtype=np.float64) - hi) - lo)


_REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.jsonl")


def record(label, **numbers):
    """Appends one line to gpurun_out/parity_report.jsonl (floored and un-floored error figures of a
    parity case, for the record; never fails a test)."""
    try:
        os.makedirs(os.path.dirname(_REPORT), exist_ok=True)
        with open(_REPORT, "a") as f:
            f.write(json.dumps({"case": label, **{k: float(v) for k, v in numbers.items()}}) + "\n")
    except OSError:
        pass


def ulp_diff(a, b):
    """Distance in units in the last place between two float64 arrays (same sign assumed
    for meaningful values; +0 and -0 count as equal)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia = np.where(ia < 0, np.int64(-2 ** 63) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2 ** 63) - ib, ib)
    return np.abs(ia - ib)


def run_oracle(cols, pp, forces, G=1.0, gravity_ignore_terms=0):
    sim = oracle.Sim(cols, G)
    oracle.apply_forces(sim, forces, pp, gravity_ignore_terms)
    return np.array(sim.acc)


def run_device(ctx, cols, pp, forces, G=1.0, mode=0, gravity_ignore_terms=0, n_head=None, tail_massless=None, device="cuda:0"):
    from reboundx_amd.device import DeviceParticles, apply_forces
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        if pp.get(k) is not None:
            host[k] = pp[k]
    dp = DeviceParticles.from_host(host, device=device, G=G, n_head=n_head, tail_massless=tail_massless)
    apply_forces(ctx, dp, forces, pp, mode, gravity_ignore_terms)
    return np.array(dp.acc_host()), dp
