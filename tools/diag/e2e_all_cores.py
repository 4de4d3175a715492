"""bench.py's two end-to-end legs alone (same-cores pair + the node-level pair with the mixed farm).
usage: python tools/diag/e2e_all_cores.py [germline|somatic|both]"""
import argparse
import json
import sys

sys.path.insert(0, ".")
import bench

which = sys.argv[1] if len(sys.argv) > 1 else "both"
args = argparse.Namespace(e2e_bp=16000000, e2e_segment_bp=2000000, e2e_somatic_bp=3200000, e2e_somatic_segment_bp=400000, e2e_max_procs_per_gpu=8)
for mode in ("germline", "somatic"):
    if which not in (mode, "both"):
        continue
    out = bench.e2e_leg(args, 0, 1, 0, lambda: None, lambda v: v, with_reference=True, mode=mode)
    for k in ("procs_note", "hook_seconds_note", "workload"):
        out.pop(k, None)
    out.get("all_cores", {}).pop("note", None)
    print(mode, json.dumps(out), flush=True)
