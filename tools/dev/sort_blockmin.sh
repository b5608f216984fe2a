for bm in 2048 3072 4096 6144 8192 16384; do for w in 1080p_2k c2 c3; do
SLAMX_SORT_BLOCK_MIN=$bm timeout -k 10 200 python bench.py --workload $w --steps 16 --warmup 2 --no-cpu-baseline --no-pcie --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('bm $bm $w value %.0f sort %.4f' % (d['value'], s['sort']))"
done; done
