#!/usr/bin/env python3
import csv, glob, sys
from collections import defaultdict
tot = defaultdict(float); cnt = defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0][-60:]
        tot[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[n] += 1
for n, t in sorted(tot.items(), key=lambda kv: -kv[1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{t * 1e-6:9.1f} ms {cnt[n]:6d} {t / cnt[n] * 1e-3:9.1f} us  {n}")
