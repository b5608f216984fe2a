# timing-only variants of the sort (tools/dev/exp/libslamx_*.so: results are WRONG with them): what each phase costs
cp slam-cin0051_amd/libslamx.so /tmp/libslamx_full.so
for v in full SKIP_LEAF SKIP_SMALL SKIP_QUEUE; do
  if [ $v = full ]; then cp /tmp/libslamx_full.so slam-cin0051_amd/libslamx.so; else cp tools/dev/exp/libslamx_$v.so slam-cin0051_amd/libslamx.so; fi
  for w in 1080p_2k c2 c3; do
    timeout -k 10 200 python bench.py --workload $w --steps 16 --warmup 2 --no-cpu-baseline --no-pcie --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('$v $w sort %.4f' % s['sort'])"
  done
done
cp /tmp/libslamx_full.so slam-cin0051_amd/libslamx.so
