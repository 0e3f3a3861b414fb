O=gpurun_out/r02j; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -m gpu -k "matches_golden or tower_step" > $O/golden.log 2>&1
grep "worst grad\|tower gradients\|passed\|failed" $O/golden.log
