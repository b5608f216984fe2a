cp slam-cin0051_amd/libslamx.so /tmp/libslamx_full.so
cp tools/dev/exp/libslamx_CLOCKS.so slam-cin0051_amd/libslamx.so
tools/bin/facade_latency tests/golden/ref_data/feature_detector.yml tests/golden/ref_data/feature_matcher.yml tests/golden/ref_data/images/0000000000.png tests/golden/ref_data/images/0000000001.png 1 2>&1 | grep -v "piece wg" | tail -60
cp /tmp/libslamx_full.so slam-cin0051_amd/libslamx.so
