"""Blocks of one kernel in program order with VALU counts and the loop they belong to.
usage: asm_loop.py file.s mangled_kernel_name"""
import sys, re, collections
lines = open(sys.argv[1]).read().splitlines()
name = sys.argv[2]
i0 = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end"))
label, hdr, cur = "entry", "", []
tot = collections.Counter()
def flush():
    if not cur: return
    c = collections.Counter(cur)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    top = ", ".join(f"{v} {k}" for k, v in c.most_common(6) if k != "s_nop")
    print(f"{label:12s} {hdr:28s} n={len(cur):4d} valu={valu:4d}  {top}")
    tot[hdr] += valu
for l in lines[i0 + 1:i1]:
    m = re.match(r"^(\.LBB\S+):\s*;?\s*(.*)", l)
    m2 = re.match(r"^; %bb\.(\d+):\s*;?\s*(.*)", l)
    if m or m2:
        flush(); cur = []
        label = m.group(1) if m else "bb." + m2.group(1)
        h = (m or m2).group(2)
        hm = re.search(r"Header=(\S+) Depth=(\d+)", h)
        hdr = f"{hm.group(1)} d{hm.group(2)}" if hm else ("loop-header" if "Loop Header" in h else "")
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    cur.append(t.split()[0])
flush()
print(dict(tot))
