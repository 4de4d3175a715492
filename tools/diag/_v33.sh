#!/bin/bash
# F5 with the read queue: parity tests, then the a5 step under grid / waves variants
O=gpurun_out/r06_v33; mkdir -p $O
python -m pytest tests/test_device_enumeration.py tests/test_read_realign.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
one() { env "$@" python bench.py --only a5 --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['kernel_ms'])"; }
{
one SK_F5_GRID=0
one SK_F5_GRID=384
one SK_F5_GRID=512
one SK_F5_GRID=768
one SK_F5_GRID=1024
one SK_F5_GRID=8192
one SK_F5_WAVES=4
one SK_F5_WAVES=16
one SK_F5_WAVES=2
} 2>&1 | tee $O/f5_variants.txt
