#!/usr/bin/env python3
"""After `make -C tiger_hlm_gpu_amd/csrc asm`: every instance of rk45_queue_kernel must keep the loop over queue items
wave-uniform -- and, since round 4, the loops over an item's passes and over the pieces of a run inside it (item loop
-> piece loop -> attempt loop; the group list of an item is built in an inner loop of its own).  A compiler that threads lanes != 0 past the draw (seen once, with `if (lane == 0)` around the
atomics) shows up as one loop level more and as an exec-masked back-edge of the item loop; both are checked here from
the assembler's own loop comments, and the piece loop's back-edges must be scalar branches too.
usage: tools/check_queue_isa.py [build/...gfx950.s]"""
import glob
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else glob.glob(os.path.join(root, "build", "*gfx950.s"))[0]
text = open(path).read().splitlines()
bad, seen = [], 0
i = 0
while i < len(text):
    m = re.match(r"^(_ZN3hlm17rk45_queue_kernel\w+):", text[i])
    if not m:
        i += 1
        continue
    name, j = m.group(1), i
    while "s_endpgm" not in text[j]:
        j += 1
    body = text[i:j]
    seen += 1
    depth3 = [k for k, ln in enumerate(body) if re.search(r"Depth[= ]5", ln)]   # (one level more than the four the source has)
    atom = [k for k, ln in enumerate(body) if "global_atomic_add" in ln]
    # the item loop: the Depth=1 header; its latch must be a scalar or unconditional branch back to the header
    hdr = None
    for k, ln in enumerate(body):
        if "=>This Loop Header: Depth=1" in ln:
            hdr = re.match(r"^(\.LBB\d+_\d+):", ln).group(1)
            break
    back = [ln.strip() for ln in body if hdr and re.search(r"s_c?branch\w*\s+%s$" % re.escape(hdr), ln.strip())]
    masked = [b for b in back if "execz" in b or "execnz" in b or "vcc" in b]
    # the pass and the piece loop: every loop header of depth 2 or 3 that HAS an inner loop (not an "Inner Loop
    # Header"); scalar back-edges only
    phdr = []
    for k, ln in enumerate(body):
        if "=>  This Loop Header: Depth=2" in ln or "=>    This Loop Header: Depth=3" in ln:   # its label: on the same line or a comment line or two above
            for u in range(k, max(k - 5, -1), -1):
                mm = re.match(r"^(\.LBB\d+_\d+):", body[u])
                if mm:
                    phdr.append(mm.group(1))
                    break
    pback = [ln.strip() for ln in body for h in phdr if re.search(r"s_c?branch\w*\s+%s$" % re.escape(h), ln.strip())]
    pmasked = [b for b in pback if "execz" in b or "execnz" in b or "vcc" in b]
    counting = bool(re.search(r"ELb[01]ELb[01]ELb1ELb[01]EEEv", name))   # the COUNT instances (template arguments FIXED, RETILE, COUNT, HYBRID) add their 7 path counters at the end of an item
    ok = not depth3 and 1 <= len(atom) <= (2 + 7 if counting else 2) and hdr is not None and back and not masked and len(phdr) >= 1 and len(pback) >= len(phdr) and not pmasked
    print(("ok  " if ok else "BAD ") + name, "loops<=4" if not depth3 else "DEPTH 5", f"draws={len(atom)}", "item-loop latch:", back,
          "pass / piece loop latches:", pback)
    if not ok:
        bad.append(name)
    i = j
print(f"{seen} instances, {len(bad)} bad")
sys.exit(1 if bad or seen == 0 else 0)
