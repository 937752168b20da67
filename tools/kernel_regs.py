#!/usr/bin/env python3
"""List VGPR/SGPR/spill/scratch/LDS of every kernel in a hipcc -save-temps .s file (amdhsa metadata)."""
import re, sys
txt = open(sys.argv[1], errors="ignore").read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if pat in name:
        print(f"{name:70s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} sgpr {g('sgpr_count'):>4s}")
