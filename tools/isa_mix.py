#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -save-temps assembly file, per basic block:
    python tools/isa_mix.py <file.s> <substring of the mangled kernel name> [--blocks]
Counts by class (VALU / transcendental / MFMA / LDS / VMEM / SALU) -- static counts; loops are listed as blocks with their labels."""
import collections
import re
import sys

path, sub = sys.argv[1], sys.argv[2]
blocks = "--blocks" in sys.argv
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sub in l.split(":")[0] and ":" in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
TRANS = ("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(TRANS): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_"): return "salu"
    return "other"


tot = collections.Counter()
cur, cur_name, per_block = collections.Counter(), "entry", []
ops_all = collections.Counter()
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith((";", ".")) and not t.endswith(":"):
        continue
    if t.endswith(":") or re.match(r"^\.LBB\d+_\d+:", t):
        if sum(cur.values()):
            per_block.append((cur_name, cur))
        cur, cur_name = collections.Counter(), t.split(":")[0]
        continue
    op = t.split()[0]
    c = cls(op)
    cur[c] += 1
    tot[c] += 1
    ops_all[op] += 1
if sum(cur.values()):
    per_block.append((cur_name, cur))
print(lines[start].split(":")[0])
print("  total:", dict(tot))
print("  top ops:", ops_all.most_common(30))
if blocks:
    for name, c in per_block:
        if sum(c.values()) >= 40:
            print(f"  {name:14s}", dict(c))
