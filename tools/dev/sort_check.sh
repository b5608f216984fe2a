# sort parity tests + the sort stage's time in the three workloads it dominates (development aid)
mkdir -p gpurun_out/sort
timeout -k 10 500 python -m pytest tests/test_parity_gpu.py -x -q -k "sort or detect or golden or full_size" > gpurun_out/sort/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/sort/pytest.log
for w in 1080p_2k c2 c3; do
  timeout -k 10 200 python bench.py --workload $w --steps 16 --warmup 2 --no-cpu-baseline --no-pcie --no-extras > gpurun_out/sort/bench_$w.json 2>gpurun_out/sort/bench_$w.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/sort/bench_$w.json").read().strip().splitlines()[-1])
s=d["stage_ms_per_step"]
print("$w", "value %.0f" % d["value"], "ms/step %.3f" % d["ms_per_step"], "sort %.4f nms %.4f moments %.4f blur %.4f fast %.4f match %.4f" % (s["sort"], s["nms_build"], s["moments"], s["blur"], s["fast"], s["match"]))
PY
done
