#!/bin/bash
# A/B of the sort kernel's counters: tools/dev/exp/libslamx_OLD.so against the tree's library
out=gpurun_out/pmc_ab; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
B="python bench.py --batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-extras"
cp slam-cin0051_amd/libslamx.so /tmp/libslamx_new.so
for v in new old; do
  if [ $v = old ]; then cp tools/dev/exp/libslamx_OLD.so slam-cin0051_amd/libslamx.so; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $out/a_$v -o a --output-format csv -- $B > /dev/null 2>$out/a_$v.err
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_BUSY_CYCLES -d $out/b_$v -o b --output-format csv -- $B > /dev/null 2>$out/b_$v.err
  for p in a b; do
    f=$(find $out/${p}_$v -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" | grep -i "sort_kernel\|^kernel\|Kernel" > $out/${p}_$v.csv
  done
done
cp /tmp/libslamx_new.so slam-cin0051_amd/libslamx.so
head -50 $out/*.csv
