#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 2400 python - <<'PY'
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import __graft_entry__ as g; g.build()
import dvs_mcemvs_amd as d
import test_gpu_fused_vote as t
ctx = d.Context(0)
bad = 0
for seed in range(0, 2000):
    try:
        t.test_fused_fuzz_over_shapes_bands_and_mappings.__wrapped__(ctx, seed) if hasattr(t.test_fused_fuzz_over_shapes_bands_and_mappings, "__wrapped__") else t.test_fused_fuzz_over_shapes_bands_and_mappings(ctx, seed)
    except AssertionError as e:
        bad += 1
        print("FAIL", seed, str(e)[:300])
    except Exception as e:
        bad += 1
        print("ERROR", seed, repr(e)[:300])
print("fuzz (1-3 cameras, own calibrations) seeds 0..1999: %d failures" % bad)
PY
