ls tests/golden/ref_data/images/*.png > /tmp/imgs.txt
run() { echo "== $*"; env "$@" tools/bin/test_frame_batch tests/golden/ref_data/feature_detector.yml tests/golden/ref_data/feature_matcher.yml 25 $(cat /tmp/imgs.txt) 2>&1 | grep -A1 "pipelined"; }
run FB_CHUNK=0
run FB_CHUNK=0
run FB_SIZES=16,32,32,32,32,32,32,26,16
run FB_SIZES=8,24,32,32,32,32,32,32,18,8
run FB_SIZES=16,48,48,48,48,26,16
run FB_SIZES=24,40,40,40,40,40,26
run FB_SIZES=32,32,32,32,32,32,32,16,10
run FB_SIZES=20,36
run FB_CHUNK=40
run FB_CHUNK=48
