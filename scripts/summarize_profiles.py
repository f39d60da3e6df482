#!/usr/bin/env python3
"""Turn the rocprofv3 (rocpd sqlite) outputs of scripts/collect_profiles.sh into the committed summaries:

  profiles/<tag>_kernel_stats.csv   per-kernel calls / total / average duration (the --kernel-trace --stats pass)
  profiles/<tag>_pmc_hbm.csv/.json  FETCH_SIZE / WRITE_SIZE per dispatch with the gfx950 corrections of
                                    MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies 128-B requests at 64 B
                                    -> bytes = 2 * FETCH_SIZE(KiB) * 1024; WRITE_SIZE is exact
  profiles/<tag>_pmc_mfma.csv       SQ_VALU_MFMA_BUSY_CYCLES etc. per dispatch (when that pass succeeded)

usage: python scripts/summarize_profiles.py [tag] [--bins N --nt NT]
"""
import csv
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(db, q):
    con = sqlite3.connect(db)
    try:
        return con.execute(q).fetchall()
    finally:
        con.close()


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r03"
    bins = int(sys.argv[sys.argv.index("--bins") + 1]) if "--bins" in sys.argv else 32768      # bench.py's default batch
    nt = int(sys.argv[sys.argv.index("--nt") + 1]) if "--nt" in sys.argv else 30
    src = os.path.join(ROOT, "gpurun_out", "prof", tag)
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)

    q = "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc"
    st = rows(os.path.join(src, "stats", "stats_results.db"), q)
    if os.path.exists(os.path.join(src, "stats_r", "stats_r_results.db")):        # the realistic-mix leg (k_sos_stream)
        st = sorted(st + [r for r in rows(os.path.join(src, "stats_r", "stats_r_results.db"), q) if "k_sos_stream" in r[0] or "k_profile" in r[0]],
                    key=lambda r: -r[2])
    tot = sum(r[2] for r in st)
    with open(os.path.join(out, "%s_kernel_stats.csv" % tag), "w", newline="") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-mix  (k_sos_os, 32768 bins per launch)\n")
        f.write("#                                and ... bench.py --steps 3 --warmup 1 --no-cpu --workload realistic --bins 4096  (k_sos_stream, k_profile); durations in ns\n")
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "average_ns", "min_ns", "max_ns", "percent"])
        for r in st:
            w.writerow([r[0], r[1], r[2], "%.1f" % r[3], r[4], r[5], "%.2f" % (100.0 * r[2] / tot)])
    print("kernel stats:", st[0][0][:60], "avg %.3f ms over %d calls" % (st[0][3] / 1e6, st[0][1]))

    summary = dict(tag=tag, bins_per_gpu=bins, nt=nt, command="python3 bench.py --steps 3 --warmup 1 --no-cpu --no-mix")
    lines = []
    for pas, counter, factor in (("fetch", "FETCH_SIZE", 2.0), ("write", "WRITE_SIZE", 1.0)):
        qq = "select name, count(*), avg(counter_value) from pmc_events where counter_name = '%s' group by name order by avg(counter_value) desc" % counter
        r = rows(os.path.join(src, pas, "%s_results.db" % pas), qq)
        dbr = os.path.join(src, pas + "_r", "%s_r_results.db" % pas)
        if os.path.exists(dbr):
            r = r + [x for x in rows(dbr, qq) if "k_sos_stream" in x[0]]
        for name, n, mean in r:
            lines.append([name, counter, n, "%.3f" % mean, "%.4e" % (factor * mean * 1024.0)])
            if "k_sos_os" in name:
                summary["k_sos_os_%s_bytes" % pas] = factor * mean * 1024.0
            if "k_sos_stream" in name:
                summary["k_sos_stream_%s_bytes" % pas] = factor * mean * 1024.0
    with open(os.path.join(out, "%s_pmc_hbm.csv" % tag), "w", newline="") as f:
        f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu\n")
        f.write("# derived counters are in KiB per dispatch (mean over dispatches); gfx950 correction per MI355X_MICROARCH.md (HBM section):\n")
        f.write("# FETCH_SIZE tallies 128-B requests at 64 B -> bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact -> bytes = WRITE_SIZE * 1024\n")
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean_value_KiB", "corrected_bytes_per_dispatch"])
        w.writerows(lines)
    if "k_sos_os_fetch_bytes" in summary and "k_sos_os_write_bytes" in summary:
        summary["k_sos_os_bytes_per_launch"] = summary["k_sos_os_fetch_bytes"] + summary["k_sos_os_write_bytes"]
    with open(os.path.join(out, "%s_pmc_hbm.json" % tag), "w") as f:
        json.dump(summary, f, indent=1)
    print("hbm:", summary)
    if "k_sos_stream_fetch_bytes" in summary and "k_sos_stream_write_bytes" in summary:
        # the realistic-mix leg of the same command (bench.py run_realistic): read by bench.pmc_traffic("_realistic")
        mix = dict(tag=tag, bins_per_gpu=4096, command="python3 bench.py --steps 3 --warmup 1 --no-cpu --workload realistic --bins 4096",
                   workload="realistic_mix",
                   k_sos_os_bytes_per_launch=summary["k_sos_stream_fetch_bytes"] + summary["k_sos_stream_write_bytes"],
                   fetch_bytes=summary["k_sos_stream_fetch_bytes"], write_bytes=summary["k_sos_stream_write_bytes"])
        with open(os.path.join(out, "%s_pmc_hbm_realistic.json" % tag), "w") as f:
            json.dump(mix, f, indent=1)
    aux = os.path.join(src, "aux", "aux_results.db")
    if os.path.exists(aux):
        st = rows(aux, "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc")
        with open(os.path.join(out, "%s_aux_kernel_stats.csv" % tag), "w", newline="") as f:
            f.write("# rocprofv3 --kernel-trace --stats -- python3 scripts/aux_kernels.py   (durations in ns; N = 41, OS_NB = 80, 4096 bins)\n")
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "average_ns", "min_ns", "max_ns"])
            for r in st:
                w.writerow([r[0], r[1], r[2], "%.1f" % r[3], r[4], r[5]])

    hyp = os.path.join(src, "hyper", "hyper_results.db")
    if os.path.exists(hyp):
        st = rows(hyp, "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc")
        with open(os.path.join(out, "%s_hyperspectral_kernel_stats.csv" % tag), "w", newline="") as f:
            f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload hyperspectral   (run_sos.sos_spectrum, 2496 wavelengths, 5732 CKD bins; durations in ns)\n")
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "average_ns", "min_ns", "max_ns"])
            for r in st[:40]:
                w.writerow([r[0], r[1], r[2], "%.1f" % r[3], r[4], r[5]])

    db = os.path.join(src, "mfma", "mfma_results.db")
    if os.path.exists(db):
        r = rows(db, "select name, counter_name, count(*), avg(counter_value) from pmc_events group by name, counter_name order by name")
        with open(os.path.join(out, "%s_pmc_mfma.csv" % tag), "w", newline="") as f:
            f.write("# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 3 --warmup 1 --no-cpu\n")
            w = csv.writer(f)
            w.writerow(["kernel", "counter", "dispatches", "mean_value"])
            for row in r:
                w.writerow([row[0], row[1], row[2], "%.4e" % row[3]])
                if "k_sos_os" in row[0]:
                    print("mfma:", row[1], "%.4e" % row[3])


if __name__ == "__main__":
    main()
