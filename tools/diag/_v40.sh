#!/bin/bash
# the push in two halves: the GPU suite, then the end-to-end legs with it and without
O=gpurun_out/r06_v40; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python bench.py --only e2e > $O/e2e_async.json 2> $O/e2e_async.err; echo "async rc=$?"
STRELKA_AMD_PUSH_ASYNC=0 python bench.py --only e2e --no-cpu-baseline > $O/e2e_sync.json 2> $O/e2e_sync.err; echo "sync rc=$?"
STRELKA_AMD_BROKER_LAZY_KICK=1 python bench.py --only e2e --no-cpu-baseline > $O/e2e_lazy_kick.json 2> $O/e2e_lazy_kick.err; echo "lazy rc=$?"
python - <<'PY'
import json
for name in ("async", "sync", "lazy_kick"):
    try:
        d = json.loads(open("gpurun_out/r06_v40/e2e_%s.json" % name).read().strip().splitlines()[-1])
    except Exception as e:
        print(name, "unreadable", e); continue
    for k in ("e2e", "e2e_somatic", "e2e_box", "e2e_somatic_box"):
        v = d.get(k)
        if v: print(name, k, "wall", v.get("amd_wall_s"), "speedup", v.get("speedup"), "identical", v.get("identical"), "psum", v.get("process_seconds_sum"), (v.get("hook_seconds") or {}).get("pileup_abi"), [ (r.get("callers"), r.get("procs"), round(r.get("wall_s"),2)) for r in v.get("runs", [])])
PY
