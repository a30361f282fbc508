#!/bin/bash
# Copy the summaries of gpurun_out/prof (made by tools/collect_profiles.cmd on the GPU box) into profiles/.
# usage: ROUND=r02 bash tools/refresh_profiles.sh
cd "$(dirname "$0")/.."
P=${ROUND:-r03}
newest() { ls -t gpurun_out/prof/$1/*/*$2 | head -1; }
clean() { grep -v "rocprofv3\|output_stream\|tool.cpp\|amdgpu.ids" "$1"; }
SIDE="--no-cpu-baseline --no-hvp --no-block --no-fp32 --no-c3"
cp gpurun_out/prof/bench.json profiles/${P}_bench.json
cp $(newest bench_trace kernel_stats.csv) profiles/${P}_bench_kernel_stats.csv
cp $(newest hvp_trace kernel_stats.csv) profiles/${P}_hvp_kernel_stats.csv
cp $(newest large_trace kernel_stats.csv) profiles/${P}_large_d_kernel_stats.csv
cp $(newest rw_trace kernel_stats.csv) profiles/${P}_reweight_kernel_stats.csv
cp $(newest block_trace kernel_stats.csv) profiles/${P}_block_kernel_stats.csv
clean gpurun_out/prof/large.log > profiles/${P}_large_d_stages.txt
clean gpurun_out/prof/reweight.log > profiles/${P}_reweight_stages.txt
clean gpurun_out/prof/hvp.log > profiles/${P}_hvp_stages.txt
clean gpurun_out/prof/block.log > profiles/${P}_block_stages.txt
clean gpurun_out/prof/fp32_stages.txt > profiles/${P}_fp32_stages.txt
clean gpurun_out/prof/narrow_d_stages.txt > profiles/${P}_narrow_d_stages.txt
cp gpurun_out/prof/mfma_f64_4x4.txt profiles/${P}_mfma_f64_4x4_microbench.txt
cp gpurun_out/prof/mfma_f32_valu_overlap.txt profiles/${P}_mfma_f32_valu_overlap.txt
{ echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --steps 5 --warmup 2 $SIDE"
  echo "# units: KB per dispatch (mean / max over dispatches). gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of the bytes of 16-B/lane streaming reads -> x2 for k_trsm_rows_small and for k_glm_hess_tile (16-B/lane LDS-DMA rows)"
  python3 tools/pmc_summary.py $(newest pmc_fetch counter_collection.csv) k_
  python3 tools/pmc_summary.py $(newest pmc_write counter_collection.csv) k_; } > profiles/${P}_pmc_hbm_traffic.txt
{ echo "# rocprofv3 --pmc FETCH_SIZE, python3 tools/time_hvp.py (N x D = 100000x20000, 1000000x1000, 4000000x100); KB per dispatch, x2 gfx950 correction for 16-B/lane reads"
  python3 tools/pmc_summary.py $(newest hvp_fetch counter_collection.csv) k_; } > profiles/${P}_pmc_hvp_traffic.txt
{ echo "# rocprofv3 --pmc SQ_* (one pass, 8 counters), python3 bench.py --steps 5 --warmup 2 $SIDE"
  python3 tools/pmc_summary.py $(newest pmc_sq counter_collection.csv) k_; } > profiles/${P}_pmc_sq_bench.txt
{ echo "# rocprofv3 --pmc SQ_* (one pass), python3 tools/time_hess32.py 100  (fp32 storage mode, N = 10^6, D = 100)"
  python3 tools/pmc_summary.py $(newest pmc_sq32 counter_collection.csv) k_; } > profiles/${P}_pmc_sq_fp32.txt
python3 tools/make_traffic_json.py $(newest pmc_fetch counter_collection.csv) $(newest pmc_write counter_collection.csv) ${P} > profiles/${P}_traffic.json
cp $(newest mid_trace kernel_stats.csv) profiles/${P}_mid_d_kernel_stats.csv
clean gpurun_out/prof/mid.log > profiles/${P}_mid_d_stages.txt
clean gpurun_out/prof/hvp_multi.txt > profiles/${P}_hvp_multi_stages.txt
{ echo "# rocprofv3 --pmc FETCH_SIZE and --pmc TCC_HIT_sum TCC_MISS_sum (separate passes), python3 tools/time_syrk_only.py 1250000 1000"
  echo "# FETCH_SIZE in KB per dispatch, x2 on gfx950 for 16-B/lane LDS-DMA reads; round 2 (tile-major workgroup order): 4.09e7 KB raw = 82 GB, hit rate 2 %"
  python3 tools/pmc_summary.py $(newest syrk_fetch counter_collection.csv) k_syrk_lds
  python3 tools/pmc_summary.py $(newest syrk_hit counter_collection.csv) k_syrk_lds; } > profiles/${P}_pmc_syrk_l2.txt
clean gpurun_out/prof/hess_tile_d.txt > profiles/${P}_hess_tile_d_stages.txt
clean gpurun_out/prof/d257_768.txt > profiles/${P}_d257_768_stages.txt
