#!/bin/bash
# the GPU suite on shifted seed sets at the head (F5's read queue, the push in two halves), and as a broker client
O=gpurun_out/r06_v44; mkdir -p $O
for k in 5 6 7; do SK_TEST_SEED_OFFSET=$((k*1000)) timeout 900 python -m pytest tests -m gpu -q -k "not at_bench_configuration" --deselect tests/test_full_size.py -p no:cacheprovider > $O/pytest_shifted_$k.txt 2>&1; echo "offset $((k*1000)): $(tail -1 $O/pytest_shifted_$k.txt)"; grep FAILED $O/pytest_shifted_$k.txt | head -5; done
STRELKA_AMD_BROKER=1 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu_as_broker_client.txt 2>&1; echo "as broker client: $(tail -1 $O/pytest_gpu_as_broker_client.txt)"; grep FAILED $O/pytest_gpu_as_broker_client.txt | head
