"""Summarise a rocprofv3 kernel trace of bench.py: per-queue first-kernel delays after the fork points and
lone-kernel time. usage: python tools/trace_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'lists_episode_kernel' in r['Kernel_Name']]
s, e = idx[-2], idx[-1]
t0 = int(rows[s]['Start_Timestamp'])
qs = sorted(set(r['Queue_Id'] for r in rows[s:e]))
ev = [((int(r['Start_Timestamp']) - t0) / 1000, (int(r['End_Timestamp']) - t0) / 1000, qs.index(r['Queue_Id']),
       r['Kernel_Name'].replace('refil::', '').replace('void ', '').split('(')[0][:36]) for r in rows[s:e]]
print("step wall us", (int(rows[e]['Start_Timestamp']) - t0) / 1000, "queues", len(qs))
for q in range(len(qs)):
    mine = [x for x in ev if x[2] == q]
    gaps = [(b[0] - a[1], a[3], b[3], a[1]) for a, b in zip(mine[:-1], mine[1:]) if b[0] - a[1] > 30]
    print(f"queue {q}: {len(mine)} kernels, busy {sum(x[1]-x[0] for x in mine):.0f} us; gaps > 30us:")
    for g in gaps:
        print(f"    {g[0]:7.0f} us idle at t={g[3]:7.0f} between {g[1]} -> {g[2]}")
pts = sorted(set([a for a, b, q, n in ev] + [b for a, b, q, n in ev]))
t = [0, 0, 0]
for a, b in zip(pts[:-1], pts[1:]):
    mid = (a + b) / 2
    c = sum(1 for x, y, q, n in ev if x <= mid < y)
    t[min(c, 2)] += b - a
print("us with 0/1/2+ kernels in flight:", [round(x) for x in t])
