#!/usr/bin/env python
"""Summarise one `rocprofv3 --pmc` pass over tools/traffic_probe.py into matrix-pipe utilisation per kernel of the
contraction family.

    python tools/pmc_summary.py <pass_dir> <replays> <out.md> [<meta.json of traffic_probe.py --meta>]

Counters of the pass: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE.  SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the
SIMDs (32 per v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the cycles the dispatch was
active, so kernel cycles = GRBM_GUI_ACTIVE / 8 and

    MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)      (share of the dispatch's own cycles)
    effective clock = GRBM_GUI_ACTIVE / 8 / duration                               (reads high below ~0.3 ms)

Only the launches between the last two marker triples (the replays of the contraction-only graph) are counted."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name.replace("gan::", ""))


def main():
    pass_dir, replays, out_md = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    meta = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else {}
    files = sorted(glob.glob(os.path.join(pass_dir, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    assert files, "no counter_collection.csv under %s" % pass_dir
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(files[-1])):
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "c": {},
                                                    "t0": int(r.get("Start_Timestamp", 0) or 0), "t1": int(r.get("End_Timestamp", 0) or 0)})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = [disp[k] for k in sorted(disp)]
    is_mark = [("fill_kernel" in r["name"] and r["grid"] <= 256) for r in rows]
    marks = [i for i in range(len(rows) - 2) if is_mark[i] and is_mark[i + 1] and is_mark[i + 2] and (i == 0 or not is_mark[i - 1])]
    assert len(marks) == 4, "expected 4 marker triples, found %d" % len(marks)
    sel = rows[marks[2] + 3:marks[3]]
    agg = collections.OrderedDict()
    for r in sel:
        a = agg.setdefault(short(r["name"]), {"n": 0, "ns": 0.0, "c": collections.Counter()})
        a["n"] += 1
        a["ns"] += r["t1"] - r["t0"]
        a["c"].update(r["c"])
    lines = ["# Matrix-pipe utilisation of the contraction kernels, DCGAN-64 batch 64 (SQ counters)", "",
             "Plan set `%s` (%s), library sources `%s`." % (meta.get("plans_sha16"), meta.get("plans_file"), meta.get("source_sha16")), "",
             "`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
             "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -- python3 tools/traffic_probe.py` (the program directly behind "
             "`--`), summarised by `tools/pmc_summary.py` over %d replays of the graph that holds only the contraction calls of "
             "one iteration (the launches `roofline.achieved` is timed on).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x "
             "GRBM_GUI_ACTIVE / 8): the share of the dispatch's own cycles in which a SIMD's matrix pipe is busy; the clock column is "
             "GRBM_GUI_ACTIVE / 8 / duration (reads high on dispatches this short; the in-kernel figure is in r03_summary.md); wave "
             "cycles split into issuing / waiting on an instruction / other (SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY over "
             "SQ_WAVE_CYCLES)." % replays, "",
             "| kernel | launches / iteration | avg us (under counter collection) | MFMA busy | busy x clock / 2.4 GHz | clock GHz | wave cycles issuing / waiting / other | VALU instr per MFMA |",
             "|---|---|---|---|---|---|---|---|"]
    tot_busy = tot_cyc = 0.0
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
        c = a["c"]
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if not k.startswith(("igemm_split_kernel", "igemm_kernel")) or cyc <= 0:
            continue
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc)
        ghz = cyc / a["ns"] if a["ns"] > 0 else float("nan")
        wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        act, wait = c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, c.get("SQ_WAIT_INST_ANY", 0.0) / wc
        n_mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 32.0 if "split" in k else c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 64.0
        valu_per = c.get("SQ_INSTS_VALU", 0.0) / max(n_mfma, 1.0)
        lines.append("| `%s` | %.1f | %.1f | %.3f | %.3f | %.2f | %.2f / %.2f / %.2f | %.1f |" % (
            k, a["n"] / replays, a["ns"] / a["n"] / 1e3, busy, busy * ghz / 2.4, ghz, act, wait, max(0.0, 1 - act - wait), valu_per))
        if "split" in k:
            tot_busy += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            tot_cyc += 1024.0 * cyc
    lines += ["", "All `igemm_split_kernel` launches together: MFMA busy **%.3f** of their own cycles." % (tot_busy / max(tot_cyc, 1.0))]
    open(out_md, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-14:]))


if __name__ == "__main__":
    main()
