#!/bin/bash
# Collects the measurements that profiles/ holds for a round (run on the GPU box through gpurun):
#   1. bench.py JSON lines for every workload (with the CPU baselines: one thread and all granted cores)
#   2. rocprofv3 --kernel-trace --stats of the default bench command, of c3 and of c4
#   3. separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of the default command (HBM traffic)
#   4. separate --pmc passes for the instruction mix (SQ_INSTS_*) and the busy counters (TA, VALU)
#   5. matrix-core counters: FP64 MFMA of gn_accumulate_kernel (c4), int8 MFMA of match_mfma_kernel (c3)
#   6. tools/microbench: sustained FP64 MFMA rate, issue rates of the VALU / LDS instructions the kernels lean on
# usage: tools/collect_profiles.sh <tag>      (writes under gpurun_out/<tag>/)
set -e
tag=${1:-r05}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python bench.py --steps 10 --warmup 2 > "$out/bench_1080p_2k.json"
echo "default done"
python bench.py --batch 256 --steps 20 --warmup 3 --no-cpu-baseline --no-pcie > "$out/bench_1080p_2k_b256.json"
python bench.py --workload c1 --steps 20 --warmup 3 --no-pcie > "$out/bench_c1.json"
python bench.py --workload c2 --steps 20 --warmup 3 --no-pcie > "$out/bench_c2.json"
python bench.py --workload c3 --steps 10 --warmup 2 > "$out/bench_c3.json"
echo "c3 done"
python bench.py --workload c3 --with-pose --steps 10 --warmup 2 --no-pcie --no-cpu-baseline > "$out/bench_c3_pose.json"
echo "c3 + pose done"
python bench.py --workload c3p --steps 5 --warmup 2 > "$out/bench_c3p.json"
python bench.py --workload c3p --with-pose --steps 5 --warmup 2 --no-pcie --no-cpu-baseline > "$out/bench_c3p_pose.json"
echo "c3p done"
python bench.py --workload c4 --steps 20 --warmup 3 > "$out/bench_c4.json"
echo "c4 done"
# the reference-signature calls one frame at a time, and the facade's sequence interface (round 5)
ls tests/golden/ref_data/images/*.png > /tmp/kitti_imgs.txt
R=tests/golden/ref_data
./tools/bin/facade_latency $R/feature_detector.yml $R/feature_matcher.yml $R/images/0000000000.png $R/images/0000000001.png 50 > "$out/facade_latency.json"
./tools/bin/test_frame_batch $R/feature_detector.yml $R/feature_matcher.yml 25 $(cat /tmp/kitti_imgs.txt) > "$out/frame_batch_kitti_250.txt"
./tools/bin/test_frame_batch $R/feature_detector.yml $R/feature_matcher.yml 1 synth:1080x1920:6800:3:24:512 > "$out/frame_batch_1080p_512.txt"
echo "facade done"
./tools/microbench/mfma_f64_peak > "$out/mfma_f64_peak.txt"
./tools/microbench/op_rates > "$out/op_rates.txt"
# --no-extras: exactly ONE program under every rocprofv3 invocation (the default workload would otherwise start the
# c3 / c3p / c4 child benches inside the profiled process, and they would write into the same output directory)
P="--no-cpu-baseline --no-pcie --no-extras"
rocprofv3 --kernel-trace --stats -d "$out/stats" -o stats --output-format csv -- python bench.py --steps 10 --warmup 2 $P > "$out/bench_under_rocprof.json"
rocprofv3 --kernel-trace --stats -d "$out/stats_c3" -o stats --output-format csv -- python bench.py --workload c3 --steps 5 --warmup 2 $P > "$out/bench_c3_under_rocprof.json"
rocprofv3 --kernel-trace --stats -d "$out/stats_c4" -o stats --output-format csv -- python bench.py --workload c4 --steps 20 --warmup 3 $P > "$out/bench_c4_under_rocprof.json"
rocprofv3 --kernel-trace --stats -d "$out/stats_c2" -o stats --output-format csv -- python bench.py --workload c2 --steps 10 --warmup 2 $P > "$out/bench_c2_under_rocprof.json"
rocprofv3 --kernel-trace --stats -d "$out/stats_c3p" -o stats --output-format csv -- python bench.py --workload c3p --steps 4 --warmup 2 $P > "$out/bench_c3p_under_rocprof.json"
rocprofv3 --kernel-trace --stats -d "$out/stats_c3_pose" -o stats --output-format csv -- python bench.py --workload c3 --with-pose --steps 5 --warmup 2 $P > "$out/bench_c3_pose_under_rocprof.json"
rocprofv3 --kernel-trace --stats -d "$out/stats_facade" -o stats --output-format csv -- ./tools/bin/facade_latency $R/feature_detector.yml $R/feature_matcher.yml $R/images/0000000000.png $R/images/0000000001.png 50 > "$out/facade_latency_under_rocprof.json"
echo "stats done"
B="python bench.py --batch 256 --steps 3 --warmup 1 $P"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch" -o fetch --output-format csv -- $B > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write" -o write --output-format csv -- $B > /dev/null
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d "$out/pmc_insts" -o insts --output-format csv -- $B > /dev/null
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU -d "$out/pmc_busy" -o busy --output-format csv -- $B > /dev/null
echo "pmc default done"
G="python bench.py --workload c4 --steps 4 --warmup 1 $P"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE -d "$out/pmc_gn" -o gn --output-format csv -- $G > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_gn_fetch" -o gnf --output-format csv -- $G > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_gn_write" -o gnw --output-format csv -- $G > /dev/null
M="python bench.py --workload c3 --batch 128 --steps 2 --warmup 1 $P"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE -d "$out/pmc_match" -o mm --output-format csv -- $M > /dev/null
# the pose stage (c3 --with-pose at 64 frames per launch) and the pyramid (c3p at 64 frames per launch): instruction mix, busy
# counters and the FP64 matrix-core counters of the refit / slamx_pnp_dlt_batch kernels; HBM traffic of the pyramid
Q="python bench.py --workload c3 --with-pose --batch 64 --steps 2 --warmup 1 $P"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d "$out/pmc_pose" -o pose --output-format csv -- $Q > /dev/null
Y="python bench.py --workload c3p --batch 64 --steps 2 --warmup 1 $P"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d "$out/pmc_c3p" -o c3p --output-format csv -- $Y > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_c3p_fetch" -o c3pf --output-format csv -- $Y > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_c3p_write" -o c3pw --output-format csv -- $Y > /dev/null
# the sort at c3 (80k keys per frame) in its ONE-launch form at 128 frames per launch, as rounds 3 / 4 measured it (launches of
# at most 128 frames would take the two-launch form since round 5: switched off for this pass)
SLAMX_SORT_SPLIT_MAX_FRAMES=0 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d "$out/pmc_sort_c3" -o sortc3 --output-format csv -- python bench.py --workload c3 --batch 128 --steps 2 --warmup 1 $P > /dev/null
echo "pmc mfma / pose / pyramid / sort done"
for n in insts busy fetch write gn gn_fetch gn_write match pose c3p c3p_fetch c3p_write sort_c3; do
    f=$(find "$out/pmc_$n" -name "*counter_collection.csv" | head -1)
    python tools/pmc_summary.py "$f" > "$out/pmc_$n.csv"
done
for n in stats stats_c3 stats_c4 stats_c2 stats_c3p stats_c3_pose stats_facade; do
    f=$(find "$out/$n" -name "*kernel_stats.csv" | head -1)
    cp "$f" "$out/${n}_kernel_stats.csv"
done
ls -la "$out"
