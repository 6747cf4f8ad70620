#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in a hipcc --save-temps assembly file (the .amdhsa metadata):
   hipcc -O3 --offload-arch=gfx950 --save-temps -c x.hip && python tools/kernel_resources.py x-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re
import sys

text = open(sys.argv[1]).read()
for blk in re.split(r"\n  - \.agpr_count:", text)[1:]:
    blk = "  - .agpr_count:" + blk
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    name = re.sub(r"^_ZN3h2v\d+", "", name)[:28]
    print(f"{name:30s} vgpr {g('vgpr_count'):>4s}  agpr {g('agpr_count'):>3s}  sgpr {g('sgpr_count'):>4s}  spill {g('vgpr_spill_count'):>3s}  scratch {g('private_segment_fixed_size'):>5s}  lds {g('group_segment_fixed_size'):>6s}")
