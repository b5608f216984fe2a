mkdir -p gpurun_out/sort
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "sort" > gpurun_out/sort/pytest_split.log 2>&1; rc=$?; echo "pytest sort rc=$rc"; tail -3 gpurun_out/sort/pytest_split.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_facade_gpu.py -x -q -k "not sort" > gpurun_out/sort/pytest_split2.log 2>&1; rc=$?; echo "pytest rest rc=$rc"; tail -3 gpurun_out/sort/pytest_split2.log
[ $rc -ne 0 ] && exit 1
ls tests/golden/ref_data/images/*.png > /tmp/imgs.txt
tools/bin/facade_latency tests/golden/ref_data/feature_detector.yml tests/golden/ref_data/feature_matcher.yml tests/golden/ref_data/images/0000000000.png tests/golden/ref_data/images/0000000001.png 50 | tail -1
SLAMX_SORT_SPLIT_MAX_FRAMES=0 tools/bin/facade_latency tests/golden/ref_data/feature_detector.yml tests/golden/ref_data/feature_matcher.yml tests/golden/ref_data/images/0000000000.png tests/golden/ref_data/images/0000000001.png 50 | tail -1
for sm in 128 0; do
SLAMX_SORT_SPLIT_MAX_FRAMES=$sm timeout -k 10 200 python bench.py --workload c1 --steps 20 --warmup 3 --no-cpu-baseline --no-pcie --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('split_max $sm c1 value %.0f ms/step %.4f sort %.4f nms %.4f' % (d['value'], d['ms_per_step'], s['sort'], s['nms_build']))"
done
