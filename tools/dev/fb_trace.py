"""Summarise a rocprofv3 kernel + memory-copy trace of tools/bin/test_frame_batch: the last window of activity (the last
pipelined pass): per chunk when its upload and its kernels ran."""
import csv, glob, os, sys
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0][:40]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "?")[:40]))
ev.sort()
# windows: gaps > 3 ms separate passes
wins, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur) > 3_000_000:
        wins.append(cur); cur = [e]
    else:
        cur.append(e)
wins.append(cur)
print(len(ev), "events,", len(wins), "windows")
for w in wins[-6:]:
    t0 = w[0][0]
    k = [e for e in w if e[2] == "K"]; c = [e for e in w if e[2] == "C"]
    print("window: %.3f ms, %d kernels (busy %.3f ms), %d copies (busy %.3f ms)" % ((max(e[1] for e in w) - t0) / 1e6, len(k), sum(e[1] - e[0] for e in k) / 1e6, len(c), sum(e[1] - e[0] for e in c) / 1e6))
w = wins[-2] if len(wins[-1]) < 50 else wins[-1]
t0 = w[0][0]
for e in w:
    if e[2] == "C" or "fast_detect" in e[3] or "emit" in e[3] or "pack" in e[3] or "sort_kernel" in e[3]:
        print("%8.3f %8.3f %s %s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, e[2], e[3]))
