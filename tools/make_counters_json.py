#!/usr/bin/env python
"""profiles/counters.json: per workload, what the PMC passes of the round say about the dominant (voting) kernel --
HBM-side traffic per launch and the shares that explain where its time goes -- stamped with the hash of the kernel source
they were measured on.  bench.py quotes them (roofline.traffic, roofline.counters) only when that hash is the running
source's and the kernel name matches, so a line never carries a stale counter as if it were a measurement.

  lds_busy_frac       SQ_LDS_IDX_ACTIVE / (CUs x kernel cycles)      share of the kernel's time the LDS pipe is indexing
  bank_conflict_frac  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE       share of those cycles that are bank-conflict cycles
  parked_frac         SQ_WAIT_ANY / SQ_WAVE_CYCLES                   share of wave-cycles spent waiting (waitcnt, barrier)
  wait_inst_lds_frac  SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES              ... of them, waiting for an LDS instruction to issue
  hbm_bytes_per_launch  (2 x FETCH_SIZE + WRITE_SIZE) KiB            (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes)
Kernel cycles = average duration under the profiler x the NOMINAL 2.4 GHz (the clock actually held is ~3 % lower, so
lds_busy_frac is a slight underestimate).

Usage: make_counters_json.py NAME=profiles/rNN_<name>_pmc_counters.txt[:events_per_launch] ... > profiles/counters.json
(the files are tools/rocpd_summary.py outputs of separate --pmc passes, tools/profile_round.sh / profile_workloads.sh)"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CUS, CLK = 256, 2.4e9


def kernel_source_sha16():
    path = os.path.join(ROOT, "dvs_mcemvs_amd", "csrc", "dsi_kernels.hip")
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def parse(path):
    """{kernel: {counter: (n, avg_value, avg_dur_us)}}"""
    vals = {}
    for line in open(path):
        f = line.split()
        if len(f) < 5:
            continue
        try:
            n, avg, dur = int(f[-3]), float(f[-2]), float(f[-1])
        except ValueError:
            continue
        kern, counter = " ".join(f[:-4]), f[-4]
        if not counter.isupper() or kern in ("kernel",):
            continue
        vals.setdefault(kern, {})[counter] = (n, avg, dur)
    return vals


def block(path, events_per_launch):
    vals = parse(path)
    votes = {k: v for k, v in vals.items() if "k_vote_" in k}
    if not votes:
        return None

    def weight(v):
        n, _, dur = next(iter(v.values()))
        return n * dur
    kern = max(votes, key=lambda k: weight(votes[k]))
    v = votes[kern]
    g = lambda c: v[c][1] if c in v else None
    dur_us = next(iter(v.values()))[2]
    cycles = dur_us * 1e-6 * CLK
    out = {"kernel": kern, "kernel_avg_us_profiled": dur_us, "launches_profiled": next(iter(v.values()))[0],
           "events_per_launch": events_per_launch, "source": os.path.relpath(path, ROOT) if os.path.isabs(path) else path}
    if g("SQ_LDS_IDX_ACTIVE") is not None:
        out["lds_busy_frac"] = g("SQ_LDS_IDX_ACTIVE") / (CUS * cycles)
        if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
            out["bank_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    if g("SQ_WAVE_CYCLES"):
        if g("SQ_WAIT_ANY") is not None:
            out["parked_frac"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
        if g("SQ_WAIT_INST_LDS") is not None:
            out["wait_inst_lds_frac"] = g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES")
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        out["FETCH_SIZE_KiB"], out["WRITE_SIZE_KiB"] = g("FETCH_SIZE"), g("WRITE_SIZE")
        out["hbm_bytes_per_launch"] = (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0
    out["raw"] = {c: v[c][1] for c in sorted(v)}
    return out


def main():
    workloads = {}
    for arg in sys.argv[1:]:
        name, rest = arg.split("=", 1)
        path, _, ev = rest.partition(":")
        b = block(path, int(ev) if ev else None)
        if b:
            workloads[name] = b
    json.dump({"kernel_source_sha16": kernel_source_sha16(), "clock_hz_nominal": CLK, "cus": CUS,
               "note": "per launch of the named kernel; separate rocprofv3 --pmc passes (never combined with other trace "
                       "domains); counters sit at the L2's fabric side for the traffic (Infinity-Cache hits included)",
               "workloads": workloads}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
