# PMC of the final mixed-precision kernels; tests of the first-layer kernel after the modulo hoist
set -x
O=gpurun_out/r05ag; mkdir -p $O
export L3_DEBUG_KNOBS=1
timeout -k 10 600 python -m pytest tests -q -s -m gpu -x -k "forms_its_own or first_layer_weight_gradient or training_step_matches_golden" > $O/tests_a.log 2>&1; echo "tests_a rc=$?"
grep -a "passed\|failed" $O/tests_a.log | tail -2
bash scripts/pmc_bf16.sh $O/pmc 128 > $O/pmc.log 2>&1
python scripts/pmc_ratios.py $O/pmc/summary.txt halo wgrad_bf16 first_wgrad first_fwd | cut -c1-260
