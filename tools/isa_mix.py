#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -save-temps .s file: isa_mix.py file.s mangled_name"""
import sys
src, name = sys.argv[1], sys.argv[2]
lines = open(src, errors="ignore").read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(name + ":")][0]
lines = lines[start + 1:]
end = [i for i, l in enumerate(lines) if "s_endpgm" in l][0]
cur, stats, order = "entry", {"entry": {}}, ["entry"]
for l in lines[:end]:
    t = l.split(";")[0].strip()
    if not t or t.startswith("."):
        if t.startswith(".LBB") and t.endswith(":"):
            pass
        else:
            continue
    if t.endswith(":"):
        cur = t[:-1]; order.append(cur); stats[cur] = {}
        continue
    op = t.split()[0]
    k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
         "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "nop" if op.startswith("s_nop") else
         "wait" if op.startswith("s_waitcnt") else "barrier" if op.startswith("s_barrier") else "salu" if op.startswith("s_") else "other")
    stats[cur][k] = stats[cur].get(k, 0) + 1
for b in order:
    if sum(stats[b].values()) > 15:
        print(b, stats[b])
