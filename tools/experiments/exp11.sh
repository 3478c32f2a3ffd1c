set -x
mkdir -p gpurun_out/e11
bash tools/pmc_mfma.sh gpurun_out/e11/pmc --workload c4small > gpurun_out/e11/pmc_mfma.txt 2>&1; cat gpurun_out/e11/pmc_mfma.txt | tail -40
