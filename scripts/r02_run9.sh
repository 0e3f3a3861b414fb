O=gpurun_out/r02i; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -m gpu -k "matches_golden" > $O/golden.log 2>&1
grep "worst grad" $O/golden.log
timeout 1800 python -m pytest tests -q -m gpu --deselect "tests/test_parity_gpu.py::test_training_step_matches_golden" > $O/rest.log 2>&1; tail -3 $O/rest.log
