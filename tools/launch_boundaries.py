"""What the launch boundaries of one generation cost: from a rocprofv3 --kernel-trace CSV of the bench command, over the
last N generations (anchor: the generation's first kernel), per boundary `kernel A -> kernel B` the idle gap between A's
end and B's start (mean / median / min / max, us, and the mean as % of the generation), the kernels' own durations, and
the sum.  This is the measurement behind DESIGN.md section 8's statement about north_star's "BPTT unroll plus weight-delta
accumulation as one fused kernel": the chain -> delta boundary is the gap printed for k_chain_persist -> k_delta_direct.
Usage: launch_boundaries.py kernel_trace.csv [anchor-substring=k_fwd_fused] [generations=40]"""
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_fwd_fused"
ngen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
idx = idx[-(ngen + 2):-1]  # the last ngen whole generations (the very last one may be cut)
short = lambda n: n.split("(")[0].split("<")[0][:40]
gaps, durs, walls = {}, {}, []
order = []
for a, b in zip(idx[:-1], idx[1:]):
    gen = rows[a:b + 1]  # ... up to and including the next generation's first kernel
    walls.append((int(gen[-1]["Start_Timestamp"]) - int(gen[0]["Start_Timestamp"])) / 1e3)
    for k, (r, nxt) in enumerate(zip(gen[:-1], gen[1:])):
        key = (k, short(r["Kernel_Name"]), short(nxt["Kernel_Name"]))
        if key not in gaps:
            order.append(key)
        gaps.setdefault(key, []).append((int(nxt["Start_Timestamp"]) - int(r["End_Timestamp"])) / 1e3)
        durs.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
wall = statistics.mean(walls)
print("%d generations, %.2f us each (start of %s to the next one's)" % (len(walls), wall, anchor))
print("%-34s %9s   %-34s %s" % ("kernel", "dur (us)", "-> next kernel", "gap: mean  median  min  max (us)   % of generation"))
tot_gap = tot_dur = 0.0
for key in order:
    if len(gaps[key]) < len(walls) // 2:
        continue  # (a launch that is not part of every generation)
    g, d = gaps[key], durs[key]
    tot_gap += statistics.mean(g)
    tot_dur += statistics.mean(d)
    print("%-34s %9.2f   %-34s %9.2f %7.2f %5.2f %5.2f   %5.2f %%" % (
        key[1], statistics.mean(d), "-> " + key[2], statistics.mean(g), statistics.median(g), min(g), max(g),
        100.0 * statistics.mean(g) / wall))
print("kernels %.2f us + boundaries %.2f us (%.2f %% of the generation)" % (tot_dur, tot_gap, 100.0 * tot_gap / wall))
