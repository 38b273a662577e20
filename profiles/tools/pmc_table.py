import csv, collections, glob, sys
def load(d):
    f = glob.glob(f'gpurun_out/pmc_{d}/runc/*_counter_collection.csv')[0]
    return list(csv.DictReader(open(f)))
def key(r):
    n = r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    return (n, int(r['Grid_Size'])//int(r['Workgroup_Size']))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ['FETCH_SIZE','WRITE_SIZE','SQ_WAVE_CYCLES','TCC_HIT_sum']:
    for r in load(d):
        agg[key(r)][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[key(r)]['dur_'+d].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
rows=[]
for k,v in agg.items():
    if 'FETCH_SIZE' not in v: continue
    n=len(v['FETCH_SIZE'])
    dur=sum(v['dur_FETCH_SIZE'])/n
    fetch=sum(v['FETCH_SIZE'])/n*1024*2/1e6   # KB -> bytes, x2 gfx950 correction (wide coalesced reads), MB
    write=sum(v.get('WRITE_SIZE',[0]))/max(1,len(v.get('WRITE_SIZE',[0])))*1024/1e6
    hit=sum(v.get('TCC_HIT_sum',[0])); miss=sum(v.get('TCC_MISS_sum',[0]))
    wc=sum(v.get('SQ_WAVE_CYCLES',[0])); wa=sum(v.get('SQ_WAIT_ANY',[0])); wi=sum(v.get('SQ_WAIT_INST_ANY',[0])); ac=sum(v.get('SQ_ACTIVE_INST_ANY',[0])); mf=sum(v.get('SQ_VALU_MFMA_BUSY_CYCLES',[0]))
    rows.append((dur*n, k, n, dur, fetch, write, hit/(hit+miss+1e-9), wa/(wc+1e-9), ac/(wc+1e-9), mf))
rows.sort(reverse=True)
print(f"{'kernel':42s} {'grid':>6s} {'n':>4s} {'us':>7s} {'fetchMB':>8s} {'writeMB':>8s} {'TB/s':>6s} {'L2hit':>6s} {'wait%':>6s} {'act%':>6s}")
for tot,k,n,dur,fe,wr,hr,wa,ac,mf in rows[:22]:
    print(f"{k[0][:42]:42s} {k[1]:6d} {n:4d} {dur:7.1f} {fe:8.1f} {wr:8.1f} {(fe+wr)/dur:6.2f} {hr:6.2f} {100*wa:6.1f} {100*ac:6.1f}")
