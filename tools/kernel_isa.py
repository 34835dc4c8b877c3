#!/usr/bin/env python3
"""Register / instruction census of the kernels of a built library whose symbol contains <substring>:
vgpr / agpr / sgpr counts, scratch, LDS from the code-object notes; instruction mix from the disassembly.
usage: tools/kernel_isa.py <substring> [lib path] [--dump]   (runs in the build container: no GPU needed)"""
import os, re, sys, tempfile, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_superpoint_amd as ssp
sub = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ssp.hipbuild.LIB
with tempfile.TemporaryDirectory(prefix="ssp_isa_", dir="/tmp") as tmp:
    dis, notes = ssp.hipbuild.disassemble(lib, tmp)
meta = {}
for blk in notes.split("- .agpr_count")[1:]:
    m = re.search(r"\.name:\s+(\S+)", blk)
    if m and sub in m.group(1):
        get = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
        meta[m.group(1)] = dict(agpr=re.match(r":\s+(\d+)", blk).group(1), vgpr=get("vgpr_count"), sgpr=get("sgpr_count"),
                                scratch=get("private_segment_fixed_size"), lds=get("group_segment_fixed_size"),
                                spill=get("vgpr_spill_count"))
parts = re.split(r"^[0-9a-f]{16} <([^>]+)>:\n", dis, flags=re.M)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1]
    if sub not in name:
        continue
    cnt = collections.Counter()
    for line in body.splitlines():
        m = re.match(r"\s+(\S+)", line)
        if m:
            op = m.group(1)
            key = ("mfma" if "mfma" in op else "exp/log/rcp" if re.match(r"v_(exp|log|rcp|rsq|sqrt)", op) else "v_pk" if op.startswith("v_pk") else
                   "valu" if op.startswith("v_") else "ds" if op.startswith("ds_") else "vmem" if re.match(r"(buffer|global|flat|scratch)_", op) else
                   "smem" if op.startswith("s_load") or op.startswith("s_buffer") else "waitcnt" if op == "s_waitcnt" else "salu/other")
            cnt[key] += 1
    print(name, meta.get(name, {}))
    print("   ", dict(cnt), "total", sum(cnt.values()))
    if "--dump" in sys.argv:
        print(body)
