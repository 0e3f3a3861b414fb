mkdir -p gpurun_out/r05y
PMC_WORKLOAD="scripts/step_profile.py 128 cnn_L3_melspec2 1 bf16 (live head)" bash scripts/pmc_bf16.sh gpurun_out/r05y/pmc_bf16 128 > gpurun_out/r05y/pmc_bf16.log 2>&1
ls gpurun_out/r05y/pmc_bf16 | head
