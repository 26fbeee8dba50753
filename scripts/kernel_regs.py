#!/usr/bin/env python3
"""Register / LDS / scratch figures of the kernels in a `hipcc --save-temps` assembly file (gfx950 .s).
usage: kernel_regs.py file.s [substring ...]"""
import re
import sys

txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    if pats and not any(p in name for p in pats):
        continue
    def val(key):
        k = re.search(r"\.amdhsa_%s\s+(\S+)" % key, body)
        return k.group(1) if k else "?"
    c = re.search(r"; codeLenInByte = (\d+).*?; NumVgprs: (\d+)\s*; NumAgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+).*?; LDSByteSize: (\d+)",
                  txt[m.end():m.end() + 4000], re.S)
    if c:
        print(f"{name[:110]}\n    vgpr {c.group(2)} agpr {c.group(3)} scratch {c.group(4)} B occupancy {c.group(5)} lds {c.group(6)} B code {c.group(1)} B")
    else:
        print(name, "next_free_vgpr", val("next_free_vgpr"), "scratch", val("private_segment_fixed_size"), "lds", val("group_segment_fixed_size"))
