#!/usr/bin/env python3
"""Static instruction histogram of one kernel in a -save-temps .s file:  tools/isa_histogram.py file.s kernel_substring [top]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
m = re.search(r'^(_Z\w*%s\w*):' % re.escape(name), s, re.M)
if not m:
    sys.exit("kernel not found")
body = s[m.end():]
body = body[:body.index('.end_amdhsa_kernel')]
ops = collections.Counter()
for line in body.splitlines():
    mm = re.match(r'^\s*([sv]_[a-z0-9_]+|ds_[a-z0-9_]+|buffer_[a-z0-9_]+|global_[a-z0-9_]+|flat_[a-z0-9_]+|scratch_[a-z0-9_]+)\b', line)
    if mm:
        ops[mm.group(1)] += 1
print(m.group(1), sum(ops.values()), "instructions")
for k, v in ops.most_common(top):
    print("  %-28s %d" % (k, v))
