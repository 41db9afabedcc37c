#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in the saved ISA (make -C tiger_hlm_gpu_amd/csrc asm).
usage: tools/isa_blocks.py <mangled-name-substring> [build/...s]   (development aid for DESIGN.md's instruction counts)"""
import re, sys
name = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else "build/hlm_abi-hip-amdgcn-amd-amdhsa-gfx950.s"
on = False
blocks = []
cur = None
for line in open(path):
    if not on:
        if re.match(r"^_Z\S*:", line) and name in line:
            on = True
            cur = dict(label="entry", valu=0, salu=0, lds=0, vmem=0, smem=0, br=[], mov=0, n=0, wait=0)
            blocks.append(cur)
        continue
    s = line.strip()
    if s.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB\S+):", s)
    if m:
        cur = dict(label=m.group(1), valu=0, salu=0, lds=0, vmem=0, smem=0, br=[], mov=0, n=0, wait=0)
        blocks.append(cur)
        continue
    if not s or s.startswith(";") or s.startswith("."):
        continue
    op = s.split()[0]
    cur["n"] += 1
    if op.startswith("v_"):
        cur["valu"] += 1
        if op.startswith("v_mov") or op.startswith("v_accvgpr"):
            cur["mov"] += 1
    elif op.startswith("ds_"):
        cur["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cur["vmem"] += 1
    elif op.startswith(("s_load", "s_buffer_load")):
        cur["smem"] += 1
    elif op.startswith(("s_cbranch", "s_branch")):
        cur["br"].append(s.split()[-1])
        cur["salu"] += 1
    elif op.startswith(("s_waitcnt", "s_nop")):
        cur["wait"] += 1
    else:
        cur["salu"] += 1
tot = dict(valu=0, salu=0, lds=0, vmem=0, smem=0, n=0, mov=0)
for b in blocks:
    for k in tot: tot[k] += b[k]
    print("%-12s n=%4d valu=%4d (mov %3d) salu=%3d lds=%2d vmem=%2d smem=%2d wait=%2d -> %s" % (
        b["label"], b["n"], b["valu"], b["mov"], b["salu"], b["lds"], b["vmem"], b["smem"], b["wait"], ",".join(b["br"])))
print("total", tot)
