#!/bin/bash
# the -m gpu parity suite on fresh synthetic inputs: seeds shifted by 1..N (default 8); run on the GPU box
N=${1:-8}
fail=0
for k in $(seq 1 $N); do
  SK_TEST_SEED_OFFSET=$((1000 * k)) timeout 600 python -m pytest tests -m gpu -q -x --deselect tests/test_full_size.py -p no:cacheprovider 2>&1 | tail -1 | sed "s/^/offset $((1000 * k)): /"
  [ ${PIPESTATUS[0]} -ne 0 ] && fail=1
done
exit $fail
