#!/bin/bash
# Collects the measurements that profiles/ holds for a round (run on the GPU box through gpurun):
#   1. bench.py JSON lines for the three workloads (with the CPU baseline for the default one)
#   2. rocprofv3 --kernel-trace --stats of the default bench command
#   3. separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of the same command (HBM traffic)
#   4. separate --pmc passes for the instruction mix (SQ_INSTS_*) and the busy counters (TA, VALU)
# usage: tools/collect_profiles.sh <tag>      (writes under gpurun_out/<tag>/)
set -e
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python bench.py --steps 10 --warmup 2 > "$out/bench_1080p_2k.json"
python bench.py --batch 256 --steps 20 --warmup 3 --no-cpu-baseline --no-pcie > "$out/bench_1080p_2k_b256.json"
python bench.py --workload c2 --steps 20 --warmup 3 --no-pcie > "$out/bench_c2.json"
python bench.py --workload c3 --steps 10 --warmup 2 --no-pcie > "$out/bench_c3.json"
rocprofv3 --kernel-trace --stats -d "$out/stats" -o stats --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pcie > "$out/bench_under_rocprof.json"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch" -o fetch --output-format csv -- python bench.py --batch 256 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write" -o write --output-format csv -- python bench.py --batch 256 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > /dev/null
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d "$out/pmc_insts" -o insts --output-format csv -- python bench.py --batch 256 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > /dev/null
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU -d "$out/pmc_busy" -o busy --output-format csv -- python bench.py --batch 256 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > /dev/null
python tools/pmc_summary.py "$out/pmc_insts/insts_counter_collection.csv" > "$out/pmc_insts.csv"
python tools/pmc_summary.py "$out/pmc_busy/busy_counter_collection.csv" > "$out/pmc_busy.csv"
python tools/pmc_summary.py "$out/pmc_fetch/fetch_counter_collection.csv" > "$out/pmc_fetch.csv"
python tools/pmc_summary.py "$out/pmc_write/write_counter_collection.csv" > "$out/pmc_write.csv"
ls -la "$out"
