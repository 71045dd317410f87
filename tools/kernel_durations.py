#!/usr/bin/env python
"""Durations of the launches of kernels whose name contains <pattern> in a rocprofv3 kernel trace, in launch order
(the last <count> of them), with their grid sizes.

    python tools/kernel_durations.py <kernel_trace.csv> <pattern> [count]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
count = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for r in rows[-count:]:
    print("%8.2f us  grid %s  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", "?"),
                                     r["Kernel_Name"][:70]))
