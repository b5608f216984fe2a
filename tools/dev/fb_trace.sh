# timeline of the pipelined FrameBatch run: kernels and copies with timestamps (rocprofv3), summarised by tools/dev/fb_trace.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ls $R/tests/golden/ref_data/images/*.png > /tmp/imgs.txt
rm -rf /tmp/fbtrace
FB_CHUNK=${FB_CHUNK:-32} FB_THREADS=${FB_THREADS:-4} rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/fbtrace -- $R/tools/bin/test_frame_batch $R/tests/golden/ref_data/feature_detector.yml $R/tests/golden/ref_data/feature_matcher.yml 25 $(cat /tmp/imgs.txt) > /tmp/fbtrace.log 2>&1
tail -5 /tmp/fbtrace.log
find /tmp/fbtrace -name "*.csv" | head
python3 $R/tools/dev/fb_trace.py /tmp/fbtrace
