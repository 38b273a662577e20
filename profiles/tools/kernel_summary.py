import csv, collections, sys, glob
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    n = n.split('(')[0]
    g = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))
    agg[(n, g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"total {tot/1e3/steps:.3f} ms/step over {steps} steps")
byk = collections.defaultdict(float)
for (n, g), v in agg.items():
    byk[n] += sum(v)
for n, t in sorted(byk.items(), key=lambda kv: -kv[1])[:14]:
    print(f"  {t/1e3/steps:7.3f} ms/step {100*t/tot:5.1f}%  {n[:70]}")
print("by (kernel, grid):")
for k in sorted(agg, key=lambda k: -sum(agg[k]))[:22]:
    v = agg[k]
    print(f"  {sum(v)/1e3/steps:7.3f} ms/step n/step={len(v)/steps:5.1f} avg={sum(v)/len(v):8.1f}us grid={k[1]:6d} {k[0][:50]}")
# ---- idle-gap analysis over the last steady-state steps (delimited by set_step_state_kernel, the first launch of a step):
#      where the GPU waits between kernels (eager launch gaps, cross-stream event waits, graph node dispatch)
recs = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]) for r in rows))
marks = [s0 for s0, e0, n in recs if n.startswith('set_step_state_kernel')]
NST = min(20, len(marks) - 1)
if NST >= 2:
    t_lo, t_hi = marks[-NST - 1], marks[-1]
    win = [r for r in recs if t_lo <= r[0] < t_hi]
    busy_until, gaps, overlap, busy = None, collections.defaultdict(lambda: [0.0, 0]), 0.0, 0.0
    for s0, e0, n in win:
        if busy_until is not None:
            if s0 > busy_until:
                gaps[n][0] += (s0 - busy_until) / 1e3
                gaps[n][1] += 1
            else:
                overlap += (min(e0, busy_until) - s0) / 1e3
        busy_until = e0 if busy_until is None else max(busy_until, e0)
    tg = sum(v[0] for v in gaps.values())
    print(f"steady state, last {NST} steps: {(t_hi - t_lo)/1e6/NST:.3f} ms/step wall, {len(win)/NST:.0f} kernels/step, "
          f"sum of kernel durations {sum(e0 - s0 for s0, e0, n in win)/1e6/NST:.3f} ms/step")
    print(f"idle gaps (no kernel running): {tg/1e3/NST:.3f} ms/step in {sum(v[1] for v in gaps.values())/NST:.0f} gaps/step; "
          f"kernel overlap (two kernels at once): {overlap/1e3/NST:.3f} ms/step")
    print("largest gap totals by the kernel that FOLLOWS the gap:")
    for n, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"  {t/1e3/NST:7.3f} ms/step  {c/NST:5.1f}/step  avg {t/max(c,1):6.1f} us  before {n[:60]}")
