#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in a `hipcc -save-temps` assembly file (the .amdhsa metadata)."""
import re
import sys

txt = open(sys.argv[1]).read()
for blk in txt.split("  - .agpr_count:")[1:]:
    def g(k):
        m = re.search(r"\.%s:\s+(\S+)" % k, blk)
        return m.group(1) if m else "?"
    name = g("name")
    name = re.sub(r"^_ZN5slamx12_GLOBAL__N_1\d+", "", name)[:48]
    print("%-50s vgpr %4s agpr %4s sgpr %4s lds %7s scratch %5s" % (name, g("vgpr_count"), blk.split()[0], g("sgpr_count"),
                                                                   g("group_segment_fixed_size"), g("private_segment_fixed_size")))
