cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sm in 128 0; do
rm -rf /tmp/fprof$sm
SLAMX_SORT_SPLIT_MAX_FRAMES=$sm rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fprof$sm -- $R/tools/bin/facade_latency $R/tests/golden/ref_data/feature_detector.yml $R/tests/golden/ref_data/feature_matcher.yml $R/tests/golden/ref_data/images/0000000000.png $R/tests/golden/ref_data/images/0000000001.png 50 > /tmp/fprof$sm.log 2>&1
echo "== split_max $sm"
f=$(find /tmp/fprof$sm -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-52s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"].split("(")[0][-52:], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
