#!/usr/bin/env python
"""Per-kernel statistics of the region of a rocprofv3 --kernel-trace that lies between the last two marker
triples (three 3-element `fill_kernel` launches in a row, as tools/model_bench.py issues around its timed loop).

    python tools/trace_summary.py <kernel_trace.csv> <steps> <title> <out.md>

Rows are (kernel, grid size): one kernel template serves several layer shapes, the grid tells them apart."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name.replace("gan::", ""))


def main():
    trace, steps, title, out_md = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    grid = lambda r: int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    is_mark = [("fill_kernel" in r["Kernel_Name"] and grid(r) <= 256) for r in rows]
    marks = [i for i in range(len(rows) - 2) if is_mark[i] and is_mark[i + 1] and is_mark[i + 2] and (i == 0 or not is_mark[i - 1])]
    assert len(marks) >= 2, "no marker pair in the trace"
    sel = rows[marks[-2] + 3:marks[-1]]
    wall = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6
    agg = collections.OrderedDict()
    for r in sel:
        d = agg.setdefault((short(r["Kernel_Name"]), grid(r)), [0, 0.0])
        d[0] += 1
        d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    total = sum(d[1] for d in agg.values())
    fam = sum(d[1] for (k, _), d in agg.items() if k.startswith(("igemm_kernel", "igemm_split_kernel", "reduce_slabs", "tprop_thin")))
    lines = ["# %s: kernel statistics of the timed region (%d iterations)" % (title, steps), "",
             "Wall time of the region %.3f ms = %.3f ms per iteration (under the profiler); summed kernel time %.3f ms per "
             "iteration, %d launches per iteration; contraction family %.3f ms per iteration."
             % (wall, wall / steps, total / 1e3 / steps, len(sel) // steps, fam / 1e3 / steps), "",
             "| kernel | grid (threads) | launches / iteration | us / iteration | avg us | % of kernel time |", "|---|---|---|---|---|---|"]
    for (k, g), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        lines.append("| `%s` | %d | %.1f | %.1f | %.2f | %.1f |" % (k, g, n / steps, us / steps, us / n, 100 * us / total))
    open(out_md, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:16]))


if __name__ == "__main__":
    main()
