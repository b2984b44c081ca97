mkdir -p gpurun_out/r3j
bash tools/pmc_prepass.sh gpurun_out/r3j/pmc > gpurun_out/r3j/pmc_prepass_c3.txt 2>&1; cat gpurun_out/r3j/pmc_prepass_c3.txt
