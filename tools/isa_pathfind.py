#!/usr/bin/env python
"""Is there a path in a kernel's control-flow graph from a load to a label (the loop's head) on which no `s_waitcnt vmcnt(0)` is met?

hipcc's structurised control flow has edges that no wave takes at run time (an `s_cbranch_execz` around a uniform branch, a `break` that joins the
latch block before it leaves); the wait-count pass honours them, and a load "still in flight" along one of them puts s_waitcnt vmcnt(0) at the
loop's head -- where it waits for the previous iteration's STORES.  This walks the assembly of ONE kernel (tools/isa_phases.sh
writes one file per kernel) breadth-first from the line of a load and prints the labels, branches, loads and waits of the first such path.

    python tools/isa_pathfind.py <kernel.s> <line of the load> <label of the loop head, e.g. .LBB9_17>"""
import collections
import re
import sys

K, start, target = sys.argv[1], int(sys.argv[2]), sys.argv[3]
L = open(K).read().split("\n")
lab = {}
for i, l in enumerate(L):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        lab[m.group(1)] = i
tgt = lab[target]


def succ(i):
    m = re.search(r"\s(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", L[i])
    out = []
    if m:
        out.append(lab[m.group(2)])
        if m.group(1) == "s_branch":
            return out
    if "s_endpgm" in L[i]:
        return out
    out.append(i + 1)
    return out


prev = {start - 1: None}
q = collections.deque([start - 1])
while q:
    i = q.popleft()
    if i == tgt:
        path = []
        while i is not None:
            path.append(i)
            i = prev[i]
        for p in reversed(path):
            l = L[p]
            if re.match(r"^\.LBB", l) or "s_cbranch" in l or "s_branch" in l or "vmcnt" in l or "global_" in l:
                print(p + 1, l.strip()[:100])
        sys.exit(0)
    if re.search(r"s_waitcnt.*vmcnt\(0\)", L[i]) and i != start - 1:
        continue
    for s in succ(i):
        if s not in prev and s < len(L):
            prev[s] = i
            q.append(s)
print("no path: every way from line %d to %s meets s_waitcnt vmcnt(0)" % (start, target))
