#!/usr/bin/env python
"""Summarise the `rocprofv3 --pmc` passes over tools/traffic_probe.py into HBM-side bytes per iteration of the
contraction family.

    python tools/traffic_summary.py <fetch_pass_dir> <write_pass_dir> <replays> <out.md> <out.json> [<meta.json of traffic_probe.py --meta>]

Reads: FETCH_SIZE (KiB) from the first pass, WRITE_SIZE (KiB) and TCC_WRITE_sum (write requests arriving at L2)
from the second.  Corrections: FETCH_SIZE counts a 128-byte request as 64 bytes for 16-byte-per-lane streaming
reads on gfx950, so it is doubled (the 512 MiB calibration copy in the same pass shows the factor).
WRITE_SIZE is taken as it is: the calibration copies (512 MiB and 16 MiB, the second well below the 32 MiB the
L2s hold) are both reported exactly.  TCC_WRITE_sum x the bytes per request the calibration shows (64) is printed
beside it as a cross-check: it counts write requests arriving at L2, which over-counts kernels whose stores
are narrower than a 64-byte request (the 12-byte pixel stores of the thin transposed kernel)."""
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


def load(pass_dir):
    files = glob.glob(os.path.join(pass_dir, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv under %s" % pass_dir
    files = sorted(files, key=os.path.getmtime)[-1:]      # one process per pass: the newest file is the pass
    disp = collections.OrderedDict()
    for f in files:
        for r in csv.DictReader(open(f)):
            d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "c": {}})
            d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [disp[k] for k in sorted(disp)]


def regions(rows):
    """Dispatch index ranges between marker triples (three tiny fill_kernel launches in a row)."""
    is_mark = [("fill_kernel" in r["name"] and r["grid"] <= 256) for r in rows]
    marks = [i for i in range(len(rows) - 2) if is_mark[i] and is_mark[i + 1] and is_mark[i + 2]
             and (i == 0 or not is_mark[i - 1])]
    assert len(marks) == 4, "expected 4 marker triples, found %d" % len(marks)
    return (marks[0] + 3, marks[1]), (marks[2] + 3, marks[3])


def family(name):
    return name.startswith(("igemm_kernel", "igemm_split_kernel", "reduce_slabs", "tprop_thin", "colsum", "linear_thin"))


def main():
    fetch_dir, write_dir, replays, out_md, out_json = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    meta = json.load(open(sys.argv[6])) if len(sys.argv) > 6 else {}
    fr, wr = load(fetch_dir), load(write_dir)
    (fc0, fc1), (fk0, fk1) = regions(fr)
    (wc0, wc1), (wk0, wk1) = regions(wr)
    cal_bytes = [512 * 2 ** 20, 16 * 2 ** 20]
    fcal, wcal = fr[fc0:fc1], wr[wc0:wc1]
    assert len(fcal) >= 2 and len(wcal) >= 2, "calibration region should start with the two copies"
    gather_cal = [r for r in fcal[2:] if "igemm" in r["name"]]      # the two 1x1 convolutions
    fetch_factor = cal_bytes[0] / (fcal[0]["c"]["FETCH_SIZE"] * 1024)
    bytes_per_wreq = cal_bytes[0] / wcal[0]["c"]["TCC_WRITE_sum"]
    lines = ["# HBM-side traffic of the contraction family, DCGAN-64 batch 64", "",
             "Plan set `%s` (%s), library sources `%s`." % (meta.get("plans_sha16"), meta.get("plans_file"), meta.get("source_sha16")), "",
             "`rocprofv3 --pmc FETCH_SIZE -- python3 tools/traffic_probe.py` and, in a run of its own, "
             "`rocprofv3 --pmc WRITE_SIZE TCC_WRITE_sum -- python3 tools/traffic_probe.py`; summarised by "
             "`tools/traffic_summary.py` over %d replays of the graph that holds only the contraction calls of one "
             "iteration (the launches `roofline.achieved` is timed on)." % replays, "",
             "Calibration (device-to-device copies of known size, 16 bytes per lane, same passes):", "",
             "| copy | FETCH_SIZE reports | x factor | WRITE_SIZE reports | TCC_WRITE_sum | bytes per write request |", "|---|---|---|---|---|---|"]
    for i, b in enumerate(cal_bytes):
        lines.append("| %d MiB | %.1f MiB | %.3f | %.1f MiB | %d | %.1f |" % (
            b >> 20, fcal[i]["c"]["FETCH_SIZE"] / 1024, b / (fcal[i]["c"]["FETCH_SIZE"] * 1024),
            wcal[i]["c"]["WRITE_SIZE"] / 1024, wcal[i]["c"]["TCC_WRITE_sum"], b / wcal[i]["c"]["TCC_WRITE_sum"]))
    fetch_factor_used = 2.0
    lines += ["", "FETCH_SIZE is doubled (the guide's gfx950 correction; the large copy shows x%.3f).  WRITE_SIZE is used as "
              "reported (both copies come out exact, the 16 MiB one as %.1f MiB); TCC_WRITE_sum x %.1f bytes is the "
              "cross-check column." % (fetch_factor, wcal[1]["c"]["WRITE_SIZE"] / 1024, bytes_per_wreq), ""]
    if gather_cal:
        lines += ["Gather calibration (1x1 convolution, every input value read once, 64 bytes per pixel and 16-deep slice; "
                  "268.4 MB of 64-channel input, then 134.2 MB of 16-channel input):", "", "| kernel | FETCH_SIZE reports | x factor |", "|---|---|---|"]
        for r, expect in zip(gather_cal, (268.435456e6, 134.217728e6)):
            raw = r["c"]["FETCH_SIZE"] * 1024
            lines.append("| `%s` grid %d | %.1f MB | %.3f |" % (short(r["name"]), r["grid"], raw / 1e6, expect / raw))
        lines.append("")
    agg = collections.OrderedDict()
    for r in fr[fk0:fk1]:
        k = short(r["name"])
        a = agg.setdefault(k, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += r["c"]["FETCH_SIZE"] * 1024 * fetch_factor_used
    for r in wr[wk0:wk1]:
        k = short(r["name"])
        a = agg.setdefault(k, [0, 0.0, 0.0, 0.0])
        a[2] += r["c"]["WRITE_SIZE"] * 1024
        a[3] += r["c"]["TCC_WRITE_sum"] * bytes_per_wreq
    lines += ["| kernel | launches / iteration | read MB / iteration | written MB / iteration | (L2 write requests x 64 B, MB) |", "|---|---|---|---|---|"]
    tot_r = tot_w = 0.0
    n_launch = 0
    for k, (n, rb, wb, wraw) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        lines.append("| `%s` | %.1f | %.1f | %.1f | %.1f |" % (k, n / replays, rb / replays / 1e6, wb / replays / 1e6, wraw / replays / 1e6))
        if family(k):
            tot_r += rb / replays
            tot_w += wb / replays
            n_launch += n
    lines += ["", "Contraction family per iteration: **%.1f MB read + %.1f MB written = %.1f MB** over %.1f launches."
              % (tot_r / 1e6, tot_w / 1e6, (tot_r + tot_w) / 1e6, n_launch / replays)]
    open(out_md, "w").write("\n".join(lines) + "\n")
    json.dump({"plans_sha16": meta.get("plans_sha16"), "source_sha16": meta.get("source_sha16"),
               "plans_file": meta.get("plans_file"), "read_bytes_per_iteration": tot_r, "written_bytes_per_iteration": tot_w,
               "launches_per_iteration": n_launch / replays, "fetch_factor_measured": fetch_factor,
               "bytes_per_write_request": bytes_per_wreq,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE TCC_WRITE_sum over tools/traffic_probe.py, tools/traffic_summary.py"},
              open(out_json, "w"), indent=1)
    print("\n".join(lines[-12:]))


if __name__ == "__main__":
    main()
