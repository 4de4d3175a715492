"""bench.py's somatic end-to-end leg alone: usage: python tools/diag/e2e_somatic_farm.py [BP=3200000] [SEGMENT=400000] [MAX_PROCS=8]"""
import argparse
import json
import sys

sys.path.insert(0, ".")
import bench

bp = int(sys.argv[1]) if len(sys.argv) > 1 else 3200000
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
procs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
args = argparse.Namespace(e2e_somatic_bp=bp, e2e_somatic_segment_bp=seg, e2e_max_procs_per_gpu=procs)
out = bench.e2e_leg(args, 0, 1, 0, lambda: None, lambda v: v, with_reference=True, mode="somatic")
out.pop("procs_note", None)
out.pop("hook_seconds_note", None)
print(json.dumps(out))
