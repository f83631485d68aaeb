#!/usr/bin/env python3
"""Host timeline of ONE training step from a rocprofv3 --hip-trace CSV (the calls between two hipMemsetAsync of the gradient buffer):
    python tools/r05_hip_api_timeline.py <dir> [which step from the end]"""
import csv, glob, sys
d = sys.argv[1]; back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in csv.DictReader(open(f))
               if not r["Function"].startswith("__hip")), key=lambda x: x[0])
ms = [i for i, r in enumerate(rows) if r[2] == "hipMemsetAsync"]
a, b = ms[-back - 1], ms[-back]
t0 = rows[a][0]
prev_end = t0
busy = 0
for s, e, n in rows[a:b]:
    if n in ("hipGetDevice", "hipSetDevice", "hipGetLastError", "hipDevicePrimaryCtxGetState"):
        busy += e - s
        continue
    print(f"{(s - t0) / 1000:9.1f} us  +gap {(s - prev_end) / 1000:7.1f}  dur {(e - s) / 1000:7.1f}  {n}")
    prev_end = e
    busy += e - s
print(f"step: {(rows[b][0] - t0) / 1000:.1f} us between the two memsets, {busy / 1000:.1f} us inside HIP calls")
