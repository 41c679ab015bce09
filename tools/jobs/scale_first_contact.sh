#!/bin/bash
# First contact with a multi-GPU node.  RCCL has never run with more than one rank in
# any round -- every lease so far had one GPU.  This script is what to run on the first node that has more:
#   1. the two tests that turn such a run into a CORRECTNESS run of the collectives,
#   2. bench.py --gpus {1,2,4,8} for the three workloads (stereo with both forms of the temporal collective),
#   3. a table: ranks RCCL itself reports, the collective's bus bandwidth alone, whole-job rate, scaling efficiency,
#   4. ONE record per workload in the shape of the driver's SCALE_rNN.json ($OUT/SCALE_<workload>.json and all of them
#      in $OUT/SCALE_first_contact.json): per N the bench line, rccl_ranks, value, efficiency = value_N / (N x value_1),
#      collective.busbw_GBps, plus the verdict of the collectives' correctness tests -- a curve, not a log
#      (VERDICT r05 item 6).
# On a one-GPU box it exits 0 after printing the expected refusals (bench.py exits 2 for N > devices: no smaller job).
# Usage: bash tools/jobs/scale_first_contact.sh [out_dir] [max_gpus]      (DSI_LAUNCH_NO_ENV_DEFAULTS=1 or
#        NCCL_SOCKET_IFNAME=... / HSA_ENABLE_IPC_MODE_LEGACY=... in the environment override launch.py's two defaults;
#        the JSON lines carry what the ranks ran with: launch_env)
set -u
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out/scale_first_contact}
MAXG=${2:-8}
mkdir -p "$OUT"
python - <<'PY' > "$OUT/devices.txt" 2>&1
import __graft_entry__ as g
g.build()
import dvs_mcemvs_amd as d
print(d.device_count())
PY
NDEV=$(tail -1 "$OUT/devices.txt")
echo "== devices visible: $NDEV"

echo "== 1. collectives over every device (correctness: all-reduce sum / min / max, sharded and reduce-scattered arg-max)"
python -m pytest tests/test_gpu_multirank.py -m gpu -q -x -k "every_device or bench_gpus_2 or communicator" \
    > "$OUT/tests.txt" 2>&1
TESTS_RC=$?
echo "   pytest exit $TESTS_RC : $(grep -E 'passed|failed|error' "$OUT/tests.txt" | tail -1)"
echo "$TESTS_RC" > "$OUT/tests.rc"

run_line() {   # name, gpus, extra flags...
    local name=$1 n=$2
    shift 2
    local f="$OUT/${name}_n${n}.json" e="$OUT/${name}_n${n}.err"
    timeout 1800 python bench.py --gpus "$n" "$@" > "$f" 2> "$e"
    local rc=$?
    if [ "$rc" -ne 0 ]; then
        if [ "$n" -gt "$NDEV" ] && [ "$rc" -eq 2 ]; then
            echo "   $name --gpus $n: refused as expected ($NDEV device(s)): $(grep -m1 'launch:' "$e")"
        else
            echo "   $name --gpus $n: FAILED rc=$rc: $(tail -2 "$e" | tr '\n' ' ')"
        fi
        rm -f "$f"
    else
        echo "   $name --gpus $n: ok"
    fi
}

echo "== 2. bench lines"
for n in 1 2 4 8; do
    [ "$n" -gt "$MAXG" ] && continue
    run_line stereo_allreduce "$n" --steps 20 --warmup 5 --no-cpu --no-extra --no-host-fed --temporal-collective allreduce
    [ "$n" -gt 1 ] && run_line stereo_reduce_scatter "$n" --steps 20 --warmup 5 --no-cpu --no-extra --no-host-fed --temporal-collective reduce_scatter
    run_line windows "$n" --workload windows --no-cpu --no-extra --no-host-fed
    run_line cameras4 "$n" --workload cameras4 --no-cpu --no-extra --no-host-fed
done

echo "== 3. summary"
python - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(out, "*_n*.json"))):
    name, n = os.path.basename(f)[:-5].rsplit("_n", 1)
    try:
        j = json.loads([ln for ln in open(f).read().splitlines() if ln.startswith("{")][-1])
    except Exception as e:
        print("unreadable", f, e)
        continue
    rows.setdefault(name, {})[int(n)] = j
print("%-24s %3s %10s %12s %10s %10s %12s %s" % ("workload", "N", "rccl_ranks", "value", "unit", "ms/step", "busbw GB/s", "value / (N x value at N=1)"))
for name, by_n in rows.items():
    base_name = "stereo_allreduce" if name.startswith("stereo") else name
    base = rows.get(base_name, {}).get(1)
    for n in sorted(by_n):
        j = by_n[n]
        coll = j.get("collective") or {}
        # whole-job rate against N times the one-GPU rate (weak scaling: N times the work in the same time; strong: the same
        # work in 1/N of the time -- either way the ideal is N x value_1)
        eff = "%.3f" % (j["value"] / (n * base["value"])) if base and base.get("value") else ""
        print("%-24s %3d %10s %12.1f %10s %10.3f %12s %s" % (name, n, j.get("rccl_ranks"), j["value"], j["unit"], j["ms_per_step"],
                                                            ("%.1f" % coll["busbw_GBps"]) if coll.get("busbw_GBps") else "-", eff))
        if n > 1 and j.get("launch_env"):
            print("    launch_env:", json.dumps(j["launch_env"]))

# 4. the records
try:
    tests_rc = int(open(os.path.join(out, "tests.rc")).read().strip())
except Exception:
    tests_rc = None
tests_tail = [ln.strip() for ln in open(os.path.join(out, "tests.txt")).read().splitlines() if ln.strip()][-1:] if os.path.exists(os.path.join(out, "tests.txt")) else []
try:
    ndev = int(open(os.path.join(out, "devices.txt")).read().split()[-1])
except Exception:
    ndev = None
records = {}
for name, by_n in rows.items():
    base_name = "stereo_allreduce" if name.startswith("stereo") else name
    base = rows.get(base_name, {}).get(1)
    runs = []
    for n in sorted(by_n):
        j = by_n[n]
        coll = j.get("collective") or {}
        runs.append({"n_gpus": n, "rccl_ranks": j.get("rccl_ranks"), "value": j["value"], "unit": j["unit"],
                     "ms_per_step": j["ms_per_step"], "scaling": j.get("scaling"),
                     "efficiency": (j["value"] / (n * base["value"])) if base and base.get("value") else None,
                     "collective": {"bytes": coll.get("bytes"), "ms_alone": coll.get("ms_alone"), "busbw_GBps": coll.get("busbw_GBps")},
                     "parsed": j})
    records[name] = {"skipped": False, "metric": runs[0]["parsed"].get("metric") if runs else None, "workload": name,
                     "devices_visible": ndev, "runs": runs,
                     "collectives_test": {"name": "tests/test_gpu_multirank.py -k 'every_device or bench_gpus_2 or communicator'",
                                          "rc": tests_rc, "passed": tests_rc == 0, "summary": tests_tail[0] if tests_tail else None},
                     "rccl_with_more_than_one_rank": any((r["rccl_ranks"] or 0) > 1 for r in runs)}
    with open(os.path.join(out, "SCALE_%s.json" % name), "w") as f:
        json.dump(records[name], f, indent=1)
with open(os.path.join(out, "SCALE_first_contact.json"), "w") as f:
    json.dump(records, f, indent=1)
print("records:", ", ".join("SCALE_%s.json" % k for k in records), "+ SCALE_first_contact.json")
PY
echo "== done: lines and logs in $OUT"
