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
