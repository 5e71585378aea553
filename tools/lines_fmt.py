#!/usr/bin/env python3
"""format the output of tools/lines.sh: per site wave instructions, lanes per instruction, 128-B lines, share"""
import sys, re
NAMES = ["fast LV", "fast NZ", "fast IC row", "fast W third", "fast IC reverse store", "rescan LV", "rescan IC row", "P2 role W rows",
         "P2 staged row", "P2b NZ rows", "P3a LV", "P3a NZ", "plain W[x][w]", "plain W[w][b]", "heavy W[x][w]", "heavy W[w][b]",
         "plain IC", "plain IC reverse", "general W", "pending IC", "pending IC reverse", "heavy IC", "heavy IC reverse",
         "P4 W writes", "items overflow", "colpass K/Z row bits", "colpass IC (b,x)", "colpass IC (x,b)", "colpass IC (rs,b)",
         "colpass IC (b,rs)", "colpass bitmap rows", "colpass W[b][w]"]
chains = []
for l in sys.stdin:
    m = re.match(r"lines chain (\d+) n=(\d+) iters=(\d+) total_lines=(\d+)", l)
    if m:
        chains.append((tuple(int(x) for x in m.groups()), {}))
        continue
    m = re.match(r"lines site (\d+) (insts|lanes|lines) (\d+)", l)
    if m and chains:
        chains[-1][1].setdefault(int(m.group(1)), {})[m.group(2)] = int(m.group(3))
for (c, n, it, tot), sites in chains[: int(sys.argv[1]) if len(sys.argv) > 1 else 2]:
    print(f"chain {c} n={n} iterations={it}: {tot} lines = {tot * 128 / 1e6:.1f} MB requested by the S5 loop")
    for q in sorted(sites):
        v = sites[q]
        if v.get("insts"):
            print(f"  {NAMES[q]:22s} insts {v['insts']:9d}  lanes/inst {v['lanes'] / v['insts']:5.1f}  lines {v['lines']:9d}  lines/inst {v['lines'] / v['insts']:5.2f}  share {100.0 * v['lines'] / tot:5.1f} %")
