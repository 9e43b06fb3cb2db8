mkdir -p gpurun_out/r05a
bash tools/pmc_tile.sh gpurun_out/r05a/pmc_runs > gpurun_out/r05a/pmc_runs.txt 2>&1
MTH_TILE_RUNS=0 bash tools/pmc_tile.sh gpurun_out/r05a/pmc_old > gpurun_out/r05a/pmc_old.txt 2>&1
tail -12 gpurun_out/r05a/pmc_runs.txt; tail -12 gpurun_out/r05a/pmc_old.txt
