"""Per-kernel register / scratch / LDS figures from a `hipcc -S --cuda-device-only` dump: python tools/kernel_regs.py file.s"""
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:\s+(\d+)(.*?)\.wavefront_size", txt, re.S):
    blk = m.group(0)
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    print(f"{name[:70]:<70} vgpr {g('vgpr_count'):>4} agpr {m.group(1):>4} spill {g('vgpr_spill_count'):>4} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6} sgpr {g('sgpr_count'):>4}")
