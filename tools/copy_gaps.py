#!/usr/bin/env python3
"""Reads a rocprofv3 --memory-copy-trace CSV (..._memory_copy_trace.csv; this rocprofv3 lists direction, stream and
the two timestamps of every copy, no size) and prints for the large host-to-device copies (longer than 20 us) of the
steady state (last two thirds): how many, their duration, the fraction of the span during which at least one copy
was in flight (link utilisation), and the idle gaps.
    usage: tools/copy_gaps.py trace.csv [bytes per copy]"""
import csv
import statistics
import sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e - b > 20000 and "HOST_TO_DEVICE" in r["Direction"]:
            rows.append((b, e, r["Stream_Id"]))
rows.sort()
if len(rows) < 6:
    sys.exit("no large host-to-device copies in the trace")
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
dur = sorted((e - b) / 1e3 for b, e, _ in rows)
print(f"{len(rows)} copies on streams {sorted(set(s for _, _, s in rows))} in {span / 1e3:.1f} us; duration us p10 "
      f"{dur[len(dur) // 10]:.1f} p50 {statistics.median(dur):.1f} p90 {dur[len(dur) * 9 // 10]:.1f}; "
      f"link busy {busy / span:.3f} of the span")
if len(sys.argv) > 2:
    nbytes = float(sys.argv[2])
    print(f"{len(rows) * nbytes / span:.2f} GB/s over the span, {len(rows) * nbytes / busy:.2f} GB/s while any copy is in flight")
if gaps:
    big = [g for g in gaps if g > 20000]
    print(f"{len(gaps)} idle gaps, total {sum(gaps) / 1e3:.1f} us ({sum(gaps) / span:.3f} of the span); longer than 20 us: "
          f"{len(big)}, total {sum(big) / 1e3:.1f} us, largest {max(gaps) / 1e3:.1f} us")
