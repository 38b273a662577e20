"""Static instruction mix of one kernel from `hipcc -S --cuda-device-only` output (MFMA / VALU / SALU / LDS / VMEM / waits and the
most frequent vector opcodes): `python profiles/tools/isa_mix.py <file.s> <mangled-name substring>`.  Used in round 3 to find the
vector-instruction fat in the hot epilogues (817 -> 543 VALU per wave and chunk in eg_ffn_chain)."""
import re, collections, sys
txt = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
st = next(i for i, l in enumerate(txt) if re.match(r"^_Z\S*:", l) and pat in l.split(":")[0])
end = next(i for i in range(st, len(txt)) if "s_endpgm" in txt[i])
c = collections.Counter(); ops = collections.Counter()
for l in txt[st + 1:end]:
    ls = l.strip()
    if not ls or ls.startswith((";", ".")) or re.match(r"\.?LBB", ls): continue
    op = ls.split()[0]
    if op.startswith("v_mfma"): c["mfma"] += 1
    elif op.startswith("v_"): c["valu"] += 1; ops[op] += 1
    elif op.startswith("s_waitcnt"): c["wait"] += 1
    elif op.startswith("s_"): c["salu"] += 1
    elif op.startswith("ds_"): c["ds"] += 1
    elif op.startswith(("global_", "buffer_", "scratch_")): c["vmem"] += 1
print(pat, dict(c)); print(ops.most_common(40))
