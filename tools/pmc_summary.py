"""Turns rocprofv3 counter-collection CSVs (one --pmc pass each, --kernel-trace only) into the per-kernel summary that
bench.py reads (profiles/rNN_pmc_fetch_write_size_per_kernel.csv) or into a plain table for other counters.

  python tools/pmc_summary.py fetch_write <fetch_counter_collection.csv> <write_counter_collection.csv> <out.csv> "<command>" [windows]
      (windows = BA windows the profiled command solved: adds the row k_ba_service_per_window = the resident grid's totals / windows)
  python tools/pmc_summary.py table <counter_collection.csv> <out.txt> "<command>"
  python tools/pmc_summary.py grid <counter_collection.csv> <log with the bench JSON line of that pass> <out.txt> "<command>"
      (the resident solver grid: its counters summed over its dispatches next to the windows it solved, their workgroups and the
      shader cycles they were resident -- what bench.py's roofline.mfma_busy divides by)

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB per dispatch; the summary keeps them as reported (the gfx950
correction -- FETCH_SIZE counts a wide coalesced read at half its bytes -- is applied by the reader, per access pattern)."""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    for pre in ("void ",):
        if name.startswith(pre):
            name = name[len(pre):]
    return name.split("<")[0].strip()


def collect(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    mode = sys.argv[1]
    if mode == "fetch_write":
        fetch, write, out, cmd = sys.argv[2:6]
        windows = int(sys.argv[6]) if len(sys.argv) > 6 else 0
        F, W = collect(fetch), collect(write)
        with open(out, "w") as o:
            o.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), %s\n" % cmd)
            o.write("# per-dispatch averages in KB as reported (FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, "
                    "MI355X_MICROARCH.md HBM section)\n")
            o.write("kernel,calls,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg\n")
            for k in sorted(set(F) | set(W)):
                fv, wv = F.get(k, {}).get("FETCH_SIZE", []), W.get(k, {}).get("WRITE_SIZE", [])
                o.write("%s,%d,%.2f,%.2f\n" % (k, max(len(fv), len(wv)), sum(fv) / max(len(fv), 1), sum(wv) / max(len(wv), 1)))
                if k == "k_ba_service" and windows > 0:
                    # the resident grid is a handful of dispatches that span the run: per window = its totals / the windows solved
                    o.write("k_ba_service_per_window,%d,%.2f,%.2f\n" % (windows, sum(fv) / windows, sum(wv) / windows))
    elif mode == "grid":
        import json
        src, log, out, cmd = sys.argv[2:6]
        A = collect(src)
        line = [ln for ln in open(log, errors="replace").read().splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
        r = json.loads(line)["roofline"]
        names = sorted({c for c in A.get("k_ba_service", {})})
        with open(out, "w") as o:
            o.write("# rocprofv3 --pmc %s --kernel-trace, %s\n" % (" ".join(names), cmd))
            o.write("# k_ba_service: counters SUMMED over its dispatches of the pass; windows / resident_cycles / workgroups_per_window from the "
                    "JSON line of the same pass (roofline.resident_windows, .resident_cycles, .workgroups_per_window)\n")
            o.write("kernel,dispatches," + ",".join(names) + ",windows,resident_cycles,workgroups_per_window\n")
            g = A.get("k_ba_service", {})
            o.write("k_ba_service,%d,%s,%d,%.0f,%d\n" % (max([len(v) for v in g.values()] or [0]), ",".join("%.1f" % sum(g[c]) for c in names),
                                                     int(r.get("resident_windows", 0)), float(r.get("resident_cycles", 0)), int(r.get("workgroups_per_window", 0))))
    else:
        src, out, cmd = sys.argv[2:5]
        A = collect(src)
        names = sorted({c for k in A for c in A[k]})
        with open(out, "w") as o:
            o.write("# rocprofv3 --pmc %s --kernel-trace, %s\n# per-dispatch averages\n" % (" ".join(names), cmd))
            o.write("kernel,calls," + ",".join(names) + "\n")
            for k in sorted(A):
                o.write("%s,%d,%s\n" % (k, max(len(v) for v in A[k].values()),
                                        ",".join("%.1f" % (sum(A[k].get(c, [0])) / max(len(A[k].get(c, [])), 1)) for c in names)))


if __name__ == "__main__":
    main()
