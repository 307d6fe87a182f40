"""Timeline of ONE learner step from a rocprofv3 --kernel-trace CSV of bench.py: every dispatch between two
step starts (lists_episode_kernel / prep kernel) with its start offset, duration, queue and (shortened) name.
usage: python tools/trace_step.py <kernel_trace.csv> [--sum]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'lists_episode_kernel' in r['Kernel_Name']]       # first kernel of a step with row lists
if len(idx) < 2:
    idx = [i for i, r in enumerate(rows) if 'prep_kernel' in r['Kernel_Name'] or 'prep_rows_kernel' in r['Kernel_Name']]
# which step: --step K (K-th step start of the run), default the one at 40 % of the run (inside bench.py's timed region;
# the last steps of a bench run are its serialised profiling passes)
k = int(sys.argv[sys.argv.index('--step') + 1]) if '--step' in sys.argv else int(0.4 * len(idx))
s, e = idx[k], idx[k + 1]
t0 = int(rows[s]['Start_Timestamp'])
qs = sorted(set(r['Queue_Id'] for r in rows[s:e]))


def short(n):
    return n.replace('refil::', '').replace('void ', '').split('(')[0][:58]


print(f"step wall {(int(rows[e]['Start_Timestamp']) - t0) / 1000:.0f} us, {e - s} dispatches, {len(qs)} queues")
agg = {}
for r in rows[s:e]:
    a, b = (int(r['Start_Timestamp']) - t0) / 1000, (int(r['End_Timestamp']) - t0) / 1000
    n = short(r['Kernel_Name'])
    if '--sum' not in sys.argv:
        print(f"{a:8.1f} {b - a:7.1f}  q{qs.index(r['Queue_Id'])}  grid={r.get('Grid_Size_X', '?'):>7}x{r.get('Grid_Size_Y', '?')}x{r.get('Grid_Size_Z', '?')} wg={r.get('Workgroup_Size_X', '?')} {n}")
    x = agg.setdefault(n, [0, 0.0])
    x[0] += 1; x[1] += b - a
print("--- per symbol (this step) ---")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:8.1f} us  x{c:<3d} {n}")
print(f"sum {sum(t for c, t in agg.values()):.0f} us")
