#!/bin/bash
# samples sclk / power while the bench runs (DVFS check)
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clocks.log &
python scripts/dev_bench.py 64 cnn_L3_melspec2 60 2>&1 | tail -9
wait
sort gpurun_out/clocks.log | uniq -c | sort -rn | head -12
