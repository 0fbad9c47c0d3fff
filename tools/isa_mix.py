"""Static instruction mix of kernels in a hipcc -S listing:  python tools/isa_mix.py file.s [name-substring ...]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
pats = sys.argv[2:]
starts = [(i, l.split(":")[0]) for i, l in enumerate(txt) if re.match(r"^_Z\S+:\s", l)]
for k, (i, name) in enumerate(starts):
    if pats and not any(p in name for p in pats):
        continue
    end = starts[k + 1][0] if k + 1 < len(starts) else len(txt)
    c = collections.Counter()
    for l in txt[i + 1:end]:
        l = l.strip()
        if not l or l[0] in ".;" or l.endswith(":"):
            continue
        op = l.split()[0]
        if op == "s_endpgm":
            c["_end"] += 1
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_pk_"):
            c["valu_pk"] += 1
        elif op.startswith("v_cvt"):
            c["valu_cvt"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("global_load"):
            c["gload"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
            c["vmem_other"] += 1
    print(name, dict(sorted(c.items())))
