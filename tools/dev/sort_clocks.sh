cp slam-cin0051_amd/libslamx.so /tmp/libslamx_full.so
cp tools/dev/exp/libslamx_CLOCKS.so slam-cin0051_amd/libslamx.so
SLAMX_SORT_SPLIT_SMALL=1 tools/bin/facade_latency tests/golden/ref_data/feature_detector.yml tests/golden/ref_data/feature_matcher.yml tests/golden/ref_data/images/0000000000.png tests/golden/ref_data/images/0000000001.png 1 2>&1 | grep "partition of" > /tmp/parts.txt
python3 - <<'PY'
import re
best={}
for l in open('/tmp/parts.txt'):
    m=re.search(r'partition of (\d+) keys: (\d+) cycles',l)
    if m:
        n,c=int(m.group(1)),int(m.group(2)); k=n//300
        if k not in best or c<best[k][1]: best[k]=(n,c)
for k in sorted(best): print("%5d keys: %6d cycles = %.1f per key" % (best[k][0],best[k][1],best[k][1]/best[k][0]))
PY
grep "^split" /tmp/parts.txt | tail -2
cp /tmp/libslamx_full.so slam-cin0051_amd/libslamx.so
