# timeline of the sort's queue phase for one frame of the default workload inside a FULL batch (512 frames, every CU busy), and
# for a lone KITTI frame (tools/dev/exp/libslamx_CLOCKS.so: order.hip built with -DSLAMX_SORT_CLOCKS)
cp slam-cin0051_amd/libslamx.so /tmp/libslamx_full.so
cp tools/dev/exp/libslamx_CLOCKS.so slam-cin0051_amd/libslamx.so
python bench.py --workload 1080p_2k --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-extras 2>&1 | grep -v "^{" | tail -40
echo "---- lone KITTI frame"
tools/bin/facade_latency tests/golden/ref_data/feature_detector.yml tests/golden/ref_data/feature_matcher.yml tests/golden/ref_data/images/0000000000.png tests/golden/ref_data/images/0000000001.png 1 2>&1 | grep -v "^{" | tail -40
cp /tmp/libslamx_full.so slam-cin0051_amd/libslamx.so
