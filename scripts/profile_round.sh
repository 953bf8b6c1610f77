#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the bench command, the two PMC passes (never combined with other trace domains), the un-profiled
#   bench line, the per-layer sweep.  Outputs land in gpurun_out/; copy what is to be judged into profiles/.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:-r01}
python scripts/layer_sweep.py > gpurun_out/${TAG}_layer_sweep.txt 2>&1
python scripts/layer_sweep.py --math fp32 > gpurun_out/${TAG}_layer_sweep_direct_kernels.txt 2>&1
python bench.py --steps 4 --warmup 1 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_prof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_pmc_write.log 2>&1
python scripts/pmc_traffic.py gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_summary.log 2>&1
find gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write -name "*counter_collection.csv" -delete
rocprofv3 --kernel-trace --pmc MfmaUtil GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_pmc_mfma -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_pmc_mfma.log 2>&1
python scripts/pmc_mfma_util.py gpurun_out/${TAG}_pmc_mfma gpurun_out/${TAG}_pmc_mfma_util.json > gpurun_out/${TAG}_pmc_mfma_summary.log 2>&1
find gpurun_out/${TAG}_pmc_mfma -name "*.csv" -delete
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_pmc_lds -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_pmc_lds.log 2>&1
python scripts/pmc_lds.py gpurun_out/${TAG}_pmc_lds gpurun_out/${TAG}_pmc_lds.json > gpurun_out/${TAG}_pmc_lds_summary.log 2>&1
find gpurun_out/${TAG}_pmc_lds -name "*.csv" -delete
find gpurun_out/${TAG}_prof -name "*kernel_trace.csv" -delete
cp gpurun_out/${TAG}_prof/*/*kernel_stats.csv gpurun_out/${TAG}_bench_kernel_stats.csv
echo profiled
