#!/usr/bin/env python3
"""Writes the fixtures under tests/golden/ from data the reference itself holds.

Runs in the development container only (it reads /root/reference); the tests read the
committed JSON files, never the reference tree.

  reference_archives.json -- from reboundx/test/binaries/<force>.sa, the trajectory archives
      the reference's test_machine_independent.py replays: REBOUND SimulationArchive files
      (version 3.9.0) of the three-body `twoplanets` system integrated with WHFast,
      dt = P1/100, one force each.  Extracted: time and particle state (x y z vx vy vz m) of
      every snapshot.  File format as found in the files: a
      64-byte header, then records {uint32 type, uint32 pad, uint64 size, payload}; type 0 =
      time, type 85 = the particle array (128-byte records: x y z vx vy vz ax ay az m ...),
      type 9999 ends a snapshot and is followed by an 8-byte index blob; type 1 = G, type 3 = dt.
      All eleven snapshots (t = 0 ... 1e4) are extracted.
  notebook_kats.json -- numbers printed in the stored outputs of the reference's notebooks
      (ipython_examples/*.ipynb), with the cell they come from.

  reference_archives.json, key "integrate_force" -- from reboundx/test/binaries/test.sa, the archive the
      reference's test_simulationarchive.py writes (:44-47, :74-77, ...: one snapshot at t = 1000, one at
      t = 2000, both reached with exact_finish_time, i.e. a shortened last step) next to test.rebx
      (copied under tests/golden/rebx/): star + planet (m = 1e-4, a = 1, e = 0.2), dt = P/100, with
      the integrate_force operator over gr (c = 100) as the one pre-timestep step, dt_fraction 1 and
      no "integrator" parameter (the default scheme, Euler: src/integrate_force.c:40-44).  Also kept:
      the archive's integrator field (REBOUND's binary field 51; the WHFast archives above hold 1
      there, and so does this one) and gravity_ignore_terms (field 14).

Floats are stored as C99 hex strings so that the fixtures are exact.
"""
import json
import os
import struct

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
FORCES = ["tides_precession", "gr", "gr_potential", "gr_full", "radiation_forces", "central_force",
          "gravitational_harmonics", "modify_orbits_forces", "modify_mass", "modify_orbits_direct"]


def read_sa(path, max_snapshots):
    d = open(path, "rb").read()
    header = d[:64].split(b"\0")[0].decode()
    off, out = 64, []
    while off < len(d) and len(out) < max_snapshots:
        info = {}
        while True:
            t, _, s = struct.unpack_from("<IIQ", d, off)
            if t == 9999:
                off += 16 + 8
                break
            info[t] = (s, off + 16)
            off += 16 + s
        time = struct.unpack_from("<d", d, info[0][1])[0]
        p = np.frombuffer(d, dtype=np.float64, count=info[85][0] // 8, offset=info[85][1]).reshape(-1, 16)
        if not out:
            scalars = {"G": struct.unpack_from("<d", d, info[1][1])[0], "dt": struct.unpack_from("<d", d, info[3][1])[0],
                       "integrator": struct.unpack_from("<i", d, info[51][1])[0],
                       "gravity_ignore_terms": struct.unpack_from("<i", d, info[14][1])[0]}
        out.append((time, p))
    return header, scalars, out


def hexrow(a):
    return [float(v).hex() for v in a]


def main():
    os.makedirs(OUT, exist_ok=True)
    arch = {"source": "reboundx/test/binaries/<force>.sa (reference test fixtures), all snapshots",
            "columns": ["x", "y", "z", "vx", "vy", "vz", "m"], "forces": {}}
    whfast_field = None
    for name in FORCES:
        header, scalars, snaps = read_sa(os.path.join(REF, "reboundx", "test", "binaries", name + ".sa"), 99)
        assert whfast_field in (None, scalars["integrator"])
        whfast_field = scalars["integrator"]
        arch["header"] = header
        arch["G"], arch["dt"] = float(scalars["G"]).hex(), float(scalars["dt"]).hex()
        arch["forces"][name] = [{"t": float(t).hex(), "particles": [hexrow(np.concatenate([row[:6], row[9:10]])) for row in p]}
                                for t, p in snaps]
    header, scalars, snaps = read_sa(os.path.join(REF, "reboundx", "test", "binaries", "test.sa"), 99)
    assert scalars["integrator"] == whfast_field and len(snaps) == 2
    arch["integrate_force"] = {
        "source": "reboundx/test/binaries/test.sa + test.rebx (written by reboundx/test/test_simulationarchive.py)",
        "G": float(scalars["G"]).hex(), "dt": float(scalars["dt"]).hex(), "integrator_field": scalars["integrator"],
        "integrator": "whfast (the field holds the value the WHFast archives of test_machine_independent.py hold)",
        "gravity_ignore_terms": scalars["gravity_ignore_terms"],
        "snapshots": [{"t": float(t).hex(), "particles": [hexrow(np.concatenate([row[:6], row[9:10]])) for row in p]} for t, p in snaps]}
    with open(os.path.join(OUT, "reference_archives.json"), "w") as f:
        json.dump(arch, f, indent=1)

    kats = {
        "rad_calc_beta": {"source": "ipython_examples/Radiation_Forces_Circumplanetary_Dust.ipynb cell 11",
                          "args": {"G": 6.674e-11, "c": 3.e8, "source_mass": 1.99e30, "source_luminosity": 3.85e26,
                                   "radius": 1.e-5, "density": 1000., "Q_pr": 1.}, "beta": "0.05767021830984005"},
        "gr_pomega": {"source": "ipython_examples/GeneralRelativity.ipynb cells 1-9: star m=1, planet m=1.66013e-07 a=0.387098 "
                                "e=0.205630, COM frame; 10 time units of gravity only, then 100 with gr, c=10065.32012038219",
                      "pomega_t110": "0.0000332591781488"},
        "gr_pomega_from_t0": {"source": "ipython_examples/SavingAndLoadingSimulations.ipynb cells 1-12: same system, gr from t=0",
                              "pomega_t10": "3.418184768435803e-06", "pomega_t20": "6.317055505806844e-06"},
        "j2_pomega": {"source": "ipython_examples/J2.ipynb cells 1-6: star J2=0.1 J4=0.2 R_eq=1/200, planet m=1e-5 a=1 e=0.01; "
                                "gravity only to t=1, then harmonics to t=1e4", "pomega_t1e4": "0.037037080331452"},
    }
    with open(os.path.join(OUT, "notebook_kats.json"), "w") as f:
        json.dump(kats, f, indent=1)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
