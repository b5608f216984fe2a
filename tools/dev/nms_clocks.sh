# phase clocks of the greedy NMS for frame 0 (nms.hip built with -DSLAMX_NMS_CLOCKS into tools/dev/exp/libslamx_NMSCLK.so)
cp slam-cin0051_amd/libslamx.so /tmp/libslamx_full.so
cp tools/dev/exp/libslamx_NMSCLK.so slam-cin0051_amd/libslamx.so
for w in 1080p_2k c3; do
python bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-extras 2>&1 | grep "^nms frame" | tail -2
done
cp /tmp/libslamx_full.so slam-cin0051_amd/libslamx.so
