"""Basic blocks of one kernel in a device .s file: size and instruction mix of the largest ones.
usage: asm_blocks.py file.s mangled_kernel_name [nblocks]"""
import sys, re, collections
lines = open(sys.argv[1]).read().splitlines()
name = sys.argv[2]
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 3
i0 = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
blocks, cur, label = [], [], "entry"
for l in lines[i0 + 1:i1]:
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        blocks.append((label, cur)); cur, label = [], m.group(1); continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    cur.append(t.split()[0])
blocks.append((label, cur))
blocks.sort(key=lambda b: -len(b[1]))
for label, ins in blocks[:nb]:
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print(f"== {label}: {len(ins)} instructions, {valu} VALU")
    for k, v in c.most_common(40): print(f"   {v:5d} {k}")
