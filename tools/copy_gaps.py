#!/usr/bin/env python3
"""Reads a rocprofv3 --memory-copy-trace CSV (…_memory_copy_trace.csv) and prints, for the host-to-device copies
larger than 1 MB: count, bytes, the rate while a copy is in flight, the union of the busy intervals against the
whole span (link utilisation), and the distribution of the idle gaps.   usage: tools/copy_gaps.py trace.csv"""
import csv
import sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        try:
            b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        except (KeyError, ValueError):
            continue
        size = int(r.get("Size", r.get("Bytes", 0)) or 0)
        d = r.get("Direction", r.get("Name", ""))
        if size >= (1 << 20) and "HOST_TO_DEVICE" in d.upper().replace(" ", "_"):
            rows.append((b, e, size))
rows.sort()
if not rows:
    sys.exit("no large host-to-device copies in the trace")
# skip the warm-up third
rows = rows[len(rows) // 3:]
span = rows[-1][1] - rows[0][0]
busy, cur_b, cur_e, gaps = 0, rows[0][0], rows[0][1], []
for b, e, _ in rows[1:]:
    if b > cur_e:
        busy += cur_e - cur_b
        gaps.append(b - cur_e)
        cur_b, cur_e = b, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_b
tot = sum(s for _, _, s in rows)
dur = sorted((e - b) for b, e, _ in rows)
gaps.sort()
print(f"{len(rows)} copies, {tot / 1e6:.1f} MB in {span / 1e3:.1f} us: {tot / span:.2f} GB/s over the span, "
      f"{tot / busy:.2f} GB/s while any copy is in flight; link busy {busy / span:.3f} of the span")
print(f"copy duration us: p10 {dur[len(dur) // 10] / 1e3:.1f} p50 {dur[len(dur) // 2] / 1e3:.1f} p90 {dur[len(dur) * 9 // 10] / 1e3:.1f}")
if gaps:
    big = [g for g in gaps if g > 20000]
    print(f"{len(gaps)} idle gaps, total {sum(gaps) / 1e3:.1f} us; > 20 us: {len(big)}, total {sum(big) / 1e3:.1f} us, "
          f"largest {gaps[-1] / 1e3:.1f} us")
