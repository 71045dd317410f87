#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace of bench.py into per-kernel statistics of the TIMED graph
replays only (start-up tuning, warm-up, capture and the eager roofline pass are cut away).

    python tools/profile_summary.py <kernel_trace.csv> <steps> <out.md> [<out.csv>]

An iteration ends with the generator's Adam step, the third adam_apply_kernel of the iteration; the
trace ends with 3 eager iterations (bench.py's roofline pass; the replays of the contraction-only graph that
follow them hold no Adam kernels) preceded by <steps> timed replays."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("gan::", "")
    return re.sub(r"\(.*$", "", name)


def main():
    trace, steps, out_md = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "adam_apply_kernel" in r["Kernel_Name"]]
    iter_ends = ends[2::3] if len(ends) % 3 == 0 else ends[len(ends) % 3 + 2::3]
    last = iter_ends[-4]                    # end of the last timed replay
    first = iter_ends[-4 - steps] + 1       # first kernel of the first timed replay
    sel = rows[first:last + 1]
    wall = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6
    agg = collections.OrderedDict()
    for r in sel:
        d = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0])
        d[0] += 1
        d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    total = sum(d[1] for d in agg.values())
    contraction = sum(d[1] for k, d in agg.items() if k.startswith(("igemm_kernel", "igemm_split_kernel", "reduce_slabs", "tprop_thin")))
    lines = ["# Kernel statistics of the timed region (%d graph replays)" % steps, "",
             "Source: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps %d --warmup 2 --no-cpu-baseline --no-exact-f32-time`; "
             "only the kernels of the %d timed replays are counted (%d launches, %.1f per iteration)."
             % (steps, steps, len(sel), len(sel) / steps), "",
             "Wall time of the region %.3f ms = %.3f ms per iteration; summed kernel time %.3f ms per iteration; "
             "contraction family (igemm_kernel + igemm_split_kernel + reduce_slabs + tprop_thin) %.3f ms per iteration."
             % (wall, wall / steps, total / 1e3 / steps, contraction / 1e3 / steps), "",
             "| kernel | launches / iteration | us / iteration | avg us | % of kernel time |", "|---|---|---|---|---|"]
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("| `%s` | %.1f | %.1f | %.2f | %.1f |" % (k, n / steps, us / steps, us / n, 100 * us / total))
    open(out_md, "w").write("\n".join(lines) + "\n")
    if len(sys.argv) > 4:
        with open(sys.argv[4], "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "launches_per_iteration", "us_per_iteration", "avg_us", "percent"])
            for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                w.writerow([k, "%.2f" % (n / steps), "%.2f" % (us / steps), "%.3f" % (us / n), "%.2f" % (100 * us / total)])
    print("\n".join(lines[:6]))


if __name__ == "__main__":
    main()
