# phase clocks of describe_tile_kernel for a few regions of frame 0 (describe.hip built with -DSLAMX_DESC_CLOCKS into tools/dev/exp/libslamx_DESCCLK.so)
cp slam-cin0051_amd/libslamx.so /tmp/libslamx_full.so
cp tools/dev/exp/libslamx_DESCCLK.so slam-cin0051_amd/libslamx.so
for w in c3 c2; do
echo "== $w"
python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-extras 2>&1 | grep "^describe tile" | sort -u | head -12
done
cp /tmp/libslamx_full.so slam-cin0051_amd/libslamx.so
