#!/bin/bash
# A long differential-fuzz campaign (run on the GPU box): stops at the first failure and keeps its trace.
#   BUDGET=<seconds> tests/fuzz/campaign.sh      (default 1000: one 20-minute GPU call with margin for the box's start-up)
# The campaign has a WALL BUDGET: a leg is only started when the time the previous leg of its kind took (an initial guess
# for the first) still fits, so a call's time limit never cuts a leg short in the middle (round 4: two campaigns ended
# that way with empty logs).  Legs that did not start are listed at the end; totals count finished legs only.
out=gpurun_out/campaign
mkdir -p $out
budget=${BUDGET:-1000}
t_start=$(date +%s)
skipped=""
# run_leg <kind> <script> <cases> <seed> <timeout> <first-guess seconds>
declare -A last
run_leg() {
  local kind=$1 script=$2 cases=$3 seed=$4 limit=$5 guess=$6
  local est=${last[$kind]:-$guess}
  local now=$(( $(date +%s) - t_start ))
  if (( now + est + est / 4 > budget )); then
    skipped="$skipped $kind:$seed"
    return 0
  fi
  local t0=$(date +%s)
  FUZZ_TRACE=$out/trace_${kind}_$seed.txt timeout -k 10 $limit python tests/fuzz/$script $cases $seed > $out/${kind}_$seed.log 2>&1 || {
    echo "$kind seed $seed FAILED"; tail -4 $out/${kind}_$seed.log; tail -3 $out/trace_${kind}_$seed.txt 2>/dev/null; exit 1; }
  rm -f $out/trace_${kind}_$seed.txt
  last[$kind]=$(( $(date +%s) - t0 ))
  done_cases[$kind]=$(( ${done_cases[$kind]:-0} + cases ))
  tail -1 $out/${kind}_$seed.log
}
declare -A done_cases
for i in $(seq 0 $(( ${DET_LEGS:-12} - 1 ))); do run_leg det fuzz_detector.py 1500 $(( ${DET0:-200} + i )) 300 45; done
for i in $(seq 0 $(( ${MAT_LEGS:-6} - 1 ))); do run_leg match fuzz_matcher.py 2500 $(( ${MAT0:-300} + i )) 300 40; done
for i in $(seq 0 $(( ${POSE_LEGS:-8} - 1 ))); do run_leg pose fuzz_pose.py 250 $(( ${POSE0:-400} + i )) 300 40; done
for i in $(seq 0 $(( ${PYR_LEGS:-10} - 1 ))); do run_leg pyr fuzz_pyramid.py 400 $(( ${PYR0:-500} + i )) 300 40; done
for i in $(seq 0 $(( ${BAT_LEGS:-5} - 1 ))); do run_leg batch fuzz_batch.py 400 $(( ${BAT0:-600} + i )) 300 60; done
for i in $(seq 0 $(( ${GN_LEGS:-2} - 1 ))); do run_leg gn fuzz_gn.py 300 $(( ${GN0:-700} + i )) 300 60; done
for i in $(seq 0 $(( ${SORT_LEGS:-3} - 1 ))); do run_leg sort fuzz_sort.py 600 $(( ${SORT0:-800} + i )) 300 60; done
echo "campaign: $(( $(date +%s) - t_start )) s of $budget;" $(for k in "${!done_cases[@]}"; do echo -n " $k: ${done_cases[$k]} cases identical;"; done)
[ -n "$skipped" ] && echo "not started (budget):$skipped"
exit 0
