#!/usr/bin/env python3
"""usage: tools/loop_isa.py /tmp/engine.s <kernel symbol substring> [marker]  -- instruction counts between the PLO_MARK comments of a kernel
(`; LIFTOVER LOOP BEGIN` ... up to the next marker), by class, per basic block; the text of the region goes to stdout with --dump"""
import re
import sys

path, kern = sys.argv[1], sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else "LIFTOVER LOOP"
dump = "--dump" in sys.argv
lines = open(path).read().splitlines()
# the kernel's body: from its label to .Lfunc_end
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kern in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
marks = [i for i, l in enumerate(body) if l.strip().startswith(";") and "LOOP" in l and ("BEGIN" in l or "END" in l)]
def cls(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "VMEM"
    return "other"
for a, b in zip(marks, marks[1:] + [len(body)]):
    name = body[a].strip()
    if marker not in name or "BEGIN" not in name:
        continue
    region = body[a:b]
    tot = {}
    blocks = 0
    for l in region:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            if t.startswith(".LBB"):
                blocks += 1
            continue
        op = t.split()[0]
        c = cls(op)
        tot[c] = tot.get(c, 0) + 1
    print(name, "->", body[b].strip() if b < len(body) else "end", "| lines", len(region), "| blocks", blocks, "|", "  ".join(f"{k} {v}" for k, v in sorted(tot.items())), "| total", sum(tot.values()))
    if dump:
        print("\n".join(region))
