#!/usr/bin/env python3
"""Copies the summaries tools/collect_profiles.sh left under gpurun_out/<tag>/ into profiles/<round>_* (the tracked
names profiles/README.md lists; bench.py reads <round>_pmc_hbm_traffic_b256.csv for roofline.traffic).
usage: tools/install_profiles.py <tag> <round>      e.g.  tools/install_profiles.py r02f r02"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def merged_traffic(fetch_csv, write_csv, out_csv):
    def col(path):  # kernel,launches,VALUE -- kernel names hold commas (template arguments) and are not quoted
        rows = [line.rsplit(",", 2) for line in open(path).read().splitlines()[1:]]
        return {r[0]: r[2] for r in rows if len(r) == 3}

    f, w = col(fetch_csv), col(write_csv)
    with open(out_csv, "w") as out:
        out.write("kernel,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch\n")
        for k in f:
            out.write("%s,%s,%s\n" % (k, f[k], w.get(k, "")))


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    plain = {
        "bench_1080p_2k.json": "bench_1080p_2k.json", "bench_1080p_2k_b256.json": "bench_1080p_2k_b256.json",
        "bench_c1.json": "bench_c1.json", "bench_c2.json": "bench_c2.json", "bench_c3.json": "bench_c3.json", "bench_c3p.json": "bench_c3p.json",
        "bench_c4.json": "bench_c4.json", "bench_under_rocprof.json": "bench_under_rocprof.json",
        "bench_c3_under_rocprof.json": "bench_c3_under_rocprof.json", "bench_c4_under_rocprof.json": "bench_c4_under_rocprof.json",
        "stats_kernel_stats.csv": "bench_1080p_2k_kernel_stats.csv", "stats_c3_kernel_stats.csv": "bench_c3_kernel_stats.csv",
        "stats_c4_kernel_stats.csv": "bench_c4_kernel_stats.csv", "pmc_insts.csv": "pmc_instruction_mix_b256.csv",
        "pmc_busy.csv": "pmc_busy_b256.csv", "pmc_gn.csv": "pmc_gn_mfma_f64_c4.csv", "pmc_match.csv": "pmc_match_mfma_i8_c3_b128.csv",
        "mfma_f64_peak.txt": "mfma_f64_peak_microbench.txt", "op_rates.txt": "op_rates_microbench.txt",
        "bench_c3_pose.json": "bench_c3_pose.json", "bench_c3p_pose.json": "bench_c3p_pose.json",
        "bench_c2_under_rocprof.json": "bench_c2_under_rocprof.json", "bench_c3p_under_rocprof.json": "bench_c3p_under_rocprof.json",
        "bench_c3_pose_under_rocprof.json": "bench_c3_pose_under_rocprof.json",
        "stats_c2_kernel_stats.csv": "bench_c2_kernel_stats.csv", "stats_c3p_kernel_stats.csv": "bench_c3p_kernel_stats.csv",
        "stats_c3_pose_kernel_stats.csv": "bench_c3_pose_kernel_stats.csv",
        "pmc_pose.csv": "pmc_pose_stage_c3_b64.csv", "pmc_c3p.csv": "pmc_instruction_mix_c3p_b64.csv",
        "pmc_sort_c3.csv": "pmc_sort_c3_b128.csv", "facade_latency.json": "facade_latency.json",
        "facade_latency_under_rocprof.json": "facade_latency_under_rocprof.json",
        "stats_facade_kernel_stats.csv": "facade_latency_kernel_stats.csv",
        "frame_batch_kitti_250.txt": "frame_batch_kitti_250.txt", "frame_batch_1080p_512.txt": "frame_batch_1080p_512.txt",
    }
    for a, b in plain.items():
        if os.path.exists(os.path.join(src, a)):
            shutil.copyfile(os.path.join(src, a), os.path.join(dst, "%s_%s" % (rnd, b)))
        else:
            print("missing", a)
    merged_traffic(os.path.join(src, "pmc_fetch.csv"), os.path.join(src, "pmc_write.csv"),
                   os.path.join(dst, "%s_pmc_hbm_traffic_b256.csv" % rnd))
    merged_traffic(os.path.join(src, "pmc_gn_fetch.csv"), os.path.join(src, "pmc_gn_write.csv"),
                   os.path.join(dst, "%s_pmc_gn_hbm_traffic_c4.csv" % rnd))
    if os.path.exists(os.path.join(src, "pmc_c3p_fetch.csv")):
        merged_traffic(os.path.join(src, "pmc_c3p_fetch.csv"), os.path.join(src, "pmc_c3p_write.csv"),
                       os.path.join(dst, "%s_pmc_hbm_traffic_c3p_b64.csv" % rnd))


if __name__ == "__main__":
    main()
