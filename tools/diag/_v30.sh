O=gpurun_out/r06_v30; mkdir -p $O
for k in 1 2 3 4; do SK_TEST_SEED_OFFSET=$((k*1000)) timeout 900 python -m pytest tests -m gpu -q -k "not at_bench_configuration" --deselect tests/test_full_size.py -p no:cacheprovider > $O/pytest_shifted_$k.txt 2>&1; echo "offset $((k*1000)): $(tail -1 $O/pytest_shifted_$k.txt)"; grep FAILED $O/pytest_shifted_$k.txt | head -5; done
