#!/bin/bash
# Copy the summaries of gpurun_out/prof (made by tools/collect_profiles.cmd on the GPU box) into profiles/.
cd "$(dirname "$0")/.."
newest() { ls -t gpurun_out/prof/$1/vm/*$2 | head -1; }
clean() { grep -v "rocprofv3\|output_stream\|tool.cpp\|amdgpu.ids" "$1"; }
cp gpurun_out/prof/bench.json profiles/r01_bench.json
cp $(newest bench_trace kernel_stats.csv) profiles/r01_bench_kernel_stats.csv
cp $(newest hvp_trace kernel_stats.csv) profiles/r01_hvp_kernel_stats.csv
cp $(newest large_trace kernel_stats.csv) profiles/r01_large_d_kernel_stats.csv
cp $(newest rw_trace kernel_stats.csv) profiles/r01_reweight_kernel_stats.csv
clean gpurun_out/prof/large.log > profiles/r01_large_d_stages.txt
clean gpurun_out/prof/reweight.log > profiles/r01_reweight_stages.txt
clean gpurun_out/prof/hvp.log > profiles/r01_hvp_stages.txt
{ echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-hvp"
  echo "# units: KB per dispatch (mean over dispatches). gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of the bytes of 16-B/lane streaming reads -> x2 for k_trsm_rows_small and for k_glm_hess_tile (16-B/lane LDS-DMA rows)"
  python3 tools/pmc_summary.py $(newest pmc_fetch counter_collection.csv) k_
  python3 tools/pmc_summary.py $(newest pmc_write counter_collection.csv) k_; } > profiles/r01_pmc_hbm_traffic.txt
{ echo "# rocprofv3 --pmc FETCH_SIZE, python3 tools/time_hvp.py (N x D = 100000x20000, 1000000x1000, 4000000x100); KB per dispatch, x2 gfx950 correction for 16-B/lane reads"
  python3 tools/pmc_summary.py $(newest hvp_fetch counter_collection.csv) k_; } > profiles/r01_pmc_hvp_traffic.txt
{ echo "# rocprofv3 --pmc SQ_* (one pass, 8 counters), python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-hvp"
  python3 tools/pmc_summary.py $(newest pmc_sq counter_collection.csv) k_; } > profiles/r01_pmc_sq_bench.txt
