import argparse, json, sys
sys.path.insert(0, ".")
import bench
args = argparse.Namespace(e2e_somatic_bp=3200000, e2e_somatic_segment_bp=400000, e2e_max_procs_per_gpu=8)
out = bench.e2e_leg(args, 0, 1, 0, lambda: None, lambda v: v, with_reference=True, mode="somatic")
out.pop("procs_note", None); out.pop("hook_seconds_note", None)
print(json.dumps(out))
