#!/bin/bash
# a long differential-fuzz campaign (run on the GPU box): stops at the first failure, keeps its trace
out=gpurun_out/campaign
mkdir -p $out
n=0
for seed in $(seq ${DET0:-200} $(( ${DET0:-200} + 11 ))); do
  FUZZ_TRACE=$out/trace_det_$seed.txt timeout -k 10 200 python tests/fuzz/fuzz_detector.py 1500 $seed > $out/det_$seed.log 2>&1 || { echo "DETECTOR seed $seed FAILED"; tail -4 $out/det_$seed.log; tail -3 $out/trace_det_$seed.txt; exit 1; }
  rm -f $out/trace_det_$seed.txt; n=$((n+1500))
done
echo "detector: $n cases identical"
m=0
for seed in $(seq ${MAT0:-300} $(( ${MAT0:-300} + 5 ))); do
  timeout -k 10 300 python tests/fuzz/fuzz_matcher.py 2500 $seed > $out/match_$seed.log 2>&1 || { echo "MATCHER seed $seed FAILED"; tail -4 $out/match_$seed.log; exit 1; }
  m=$((m+2500))
done
echo "matcher: $m cases identical"
for seed in $(seq ${POSE0:-400} $(( ${POSE0:-400} + 7 ))); do
  timeout -k 10 200 python tests/fuzz/fuzz_pose.py 250 $seed > $out/pose_$seed.log 2>&1 || { echo "POSE seed $seed FAILED"; tail -4 $out/pose_$seed.log; exit 1; }
  tail -1 $out/pose_$seed.log
done
for seed in $(seq ${PYR0:-500} $(( ${PYR0:-500} + 9 ))); do
  timeout -k 10 200 python tests/fuzz/fuzz_pyramid.py 400 $seed > $out/pyr_$seed.log 2>&1 || { echo "PYRAMID seed $seed FAILED"; tail -4 $out/pyr_$seed.log; exit 1; }
done
echo "pyramid: 10 x 400 cases identical"
for seed in $(seq ${BAT0:-600} $(( ${BAT0:-600} + 4 ))); do
  timeout -k 10 250 python tests/fuzz/fuzz_batch.py 400 $seed > $out/batch_$seed.log 2>&1 || { echo "BATCH seed $seed FAILED"; tail -4 $out/batch_$seed.log; exit 1; }
  tail -1 $out/batch_$seed.log
done
