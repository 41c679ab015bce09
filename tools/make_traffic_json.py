#!/usr/bin/env python
"""profiles/traffic.json from a PMC summary (stamped with the kernel source's hash, the DSI shape and the events per launch:
bench.py's roofline.traffic quotes it only when all three match the run) -- (tools/rocpd_summary.py output of the FETCH_SIZE and
WRITE_SIZE passes of tools/profile_round.sh).  Fabric-side traffic of the voting kernel per launch
= fetch_correction * FETCH_SIZE + WRITE_SIZE (KiB -> bytes), with the FETCH_SIZE correction
calibrated on k_fuse2<2>, whose traffic is known exactly (reads two volumes, writes one).
NX NY NZ = the grid k_fuse2<2> ran on (bench.py's stream_kernels: 512 512 200).
Usage: make_traffic_json.py profiles/rNN_pmc_counters.txt NX NY NZ > profiles/traffic.json"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha16():
    """bench.py quotes this file's traffic only for the kernel source it was measured on."""
    path = os.path.join(ROOT, "dvs_mcemvs_amd", "csrc", "dsi_kernels.hip")
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def main():
    path, nx, ny, nz = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    vals = {}
    for line in open(path):
        f = line.split()
        if len(f) >= 5 and f[-4] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals.setdefault(" ".join(f[:-4]), {})[f[-4]] = float(f[-2])
    vote = next(k for k in vals if k.startswith("k_vote_"))
    cal = vals["k_fuse2<2>"]
    vol = 4 * nx * ny * nz
    fc = 2.0 * vol / (cal["FETCH_SIZE"] * 1024.0)
    wc = 1.0 * vol / (cal["WRITE_SIZE"] * 1024.0)
    v = vals[vote]
    out = {
        "kernel": vote,
        "config": "346x260x100, one camera of configs[1] per launch (default bench.py workload)",
        "kernel_source_sha16": kernel_source_sha16(), "dims": [346, 260, 100], "events_per_launch": 9999360,
        "FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
        "calibration": {"kernel": "k_fuse2<2> (reads 2 volumes, writes 1; %d bytes each)" % vol,
                        "FETCH_SIZE_KiB": cal["FETCH_SIZE"], "expected_read_bytes": 2 * vol,
                        "fetch_correction": fc, "WRITE_SIZE_KiB": cal["WRITE_SIZE"],
                        "expected_write_bytes": vol, "write_correction": wc},
        "note": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes read "
                "(confirmed by the calibration above); WRITE_SIZE is 1:1. Counters sit at the L2's fabric "
                "side, so Infinity-Cache hits are included (upper bound on HBM). "
                "hbm_bytes_per_launch = 2*FETCH_SIZE + WRITE_SIZE.",
        "hbm_bytes_per_launch": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0,
    }
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
