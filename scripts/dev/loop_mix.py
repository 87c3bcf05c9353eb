#!/usr/bin/env python3
"""Instruction mix of the biggest backward-branch loop of a gfx950 .s file (development aid): scripts/dev/loop_mix.py k.s"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().splitlines()
# keep only the LAST kernel body in the file (the template instantiation)
labels = {}
ins = []
for i, l in enumerate(lines):
    m = re.match(r"^(\.?[A-Za-z_][\w.$]*):", l)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    t = l.strip()
    if not t or t.startswith((";", ".", "//")):
        continue
    ins.append(t)
best = None
for idx, t in enumerate(ins):
    m = re.match(r"s_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)", t)
    if m:
        lab = m.group(1) or m.group(2)
        if lab in labels and labels[lab] < idx:
            span = idx - labels[lab]
            if best is None or span > best[0]:
                best = (span, labels[lab], idx, lab)
print("biggest loop:", best)
span, a, b, lab = best
c = Counter()
nopc = 0
for t in ins[a:b + 1]:
    op = t.split()[0]
    if op.startswith("v_mfma"):
        c["mfma"] += 1
    elif op.startswith("v_readlane") or op.startswith("v_writelane"):
        c["lane"] += 1
    elif op.startswith("v_"):
        c["valu"] += 1
    elif op == "s_nop":
        c["s_nop"] += 1
        nopc += int(t.split()[1]) + 1
    elif op.startswith("s_waitcnt"):
        c["waitcnt"] += 1
    elif op.startswith("s_barrier"):
        c["barrier"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
    elif op.startswith("ds_"):
        c["ds"] += 1
    elif op.startswith("buffer_") or op.startswith("global_"):
        c["vmem"] += 1
    else:
        c["other:" + op] += 1
print(dict(c), "nop cycles", nopc)
