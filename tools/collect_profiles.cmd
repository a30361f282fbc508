#!/bin/bash
# Run on the GPU box (bash tools/collect_profiles.cmd): every measurement that profiles/ holds.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
set -o pipefail
SIDE="--no-cpu-baseline --no-hvp --no-block --no-fp32 --no-c3"
timeout -k 10 400 python3 $R/bench.py > $O/bench.log 2>&1 || exit 1
tail -1 $O/bench.log > $O/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python3 $R/bench.py --no-cpu-baseline --no-hvp --no-block --no-c3 > $O/bench_trace.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 $SIDE > $O/pmc_fetch.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 $SIDE > $O/pmc_write.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 5 --warmup 2 $SIDE > $O/pmc_sq.log 2>&1 || exit 5
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_sq32 -- python3 $R/tools/time_hess32.py 100 > $O/pmc_sq32.log 2>&1 || exit 5
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/hvp_trace -- python3 $R/tools/time_hvp.py > $O/hvp.log 2>&1 || exit 6
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/hvp_fetch -- python3 $R/tools/time_hvp.py > $O/hvp_fetch.log 2>&1 || exit 7
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/large_trace -- python3 $R/tools/time_large.py 1250000 1000 > $O/large.log 2>&1 || exit 8
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rw_trace -- python3 $R/tools/time_reweight.py 1000000 100 64 256 1024 > $O/reweight.log 2>&1 || exit 9
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/block_trace -- python3 $R/tools/prof_block.py > $O/block.log 2>&1 || exit 10
timeout -k 10 120 $R/tools/micro/mfma_f64_4x4 > $O/mfma_f64_4x4.txt 2>&1 || exit 11
timeout -k 10 120 $R/tools/micro/mfma_f32_valu_overlap > $O/mfma_f32_valu_overlap.txt 2>&1 || exit 12
timeout -k 10 120 python3 $R/tools/time_hess32.py 100 > $O/fp32_stages.txt 2>&1 || exit 13
timeout -k 10 200 python3 $R/tools/time_stages.py 4000000 4 8 20 32 48 64 80 > $O/narrow_d_stages.txt 2>&1 || exit 14
for D in 8 20 64; do timeout -k 10 120 python3 $R/tools/time_hess32.py $D >> $O/fp32_stages.txt 2>&1 || exit 15; done
timeout -k 10 200 python3 $R/tools/time_hvp32.py 100000 20000 >> $O/fp32_stages.txt 2>&1 || exit 15
timeout -k 10 200 python3 $R/tools/time_hess_only.py 1000000 81 88 96 100 104 112 128 > $O/hess_tile_d.txt 2>&1 || exit 15
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mid_trace -- python3 $R/tools/sweep_mid.py 2000000 128 129 144 160 192 224 256 257 320 384 512 > $O/mid_profiled.log 2>&1 || exit 16
timeout -k 10 300 python3 $R/tools/sweep_mid.py 2000000 128 129 144 160 192 224 256 257 320 384 512 > $O/mid.log 2>&1 || exit 16
timeout -k 10 300 python3 $R/tools/sweep_mid.py 2000000 257 288 320 352 384 385 400 512 640 768 > $O/d257_768.txt 2>&1 || exit 16
timeout -k 10 300 python3 $R/tools/time_hvp_multi.py 100000 20000 4 8 16 32 64 128 > $O/hvp_multi.txt 2>&1 || exit 17
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/syrk_fetch -- python3 $R/tools/time_syrk_only.py 1250000 1000 > $O/syrk_fetch.log 2>&1 || exit 18
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/syrk_hit -- python3 $R/tools/time_syrk_only.py 1250000 1000 > $O/syrk_hit.log 2>&1 || exit 19
echo collected
