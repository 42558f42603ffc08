#!/usr/bin/env python3
"""Developer tool: static instruction counts of one kernel per source line / per enclosing function, from an assembly listing made
with -gline-tables-only (the .loc directives). Usage: isa_by_line.py listing.s kernels.hip [--lines]
Static counts are not time, but the phase bodies of the stage scheduler are mostly straight-line code."""
import collections
import re
import sys

listing, source = sys.argv[1], sys.argv[2]
src = open(source).read().split("\n")
# map a line to the function whose body contains it: crude -- the last line at or above it that looks like a definition
defs = []
pat = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:DEV|__device__|__global__|static|inline|[\w:<>\*&\s]+?)\s+([\w:]+)\s*\([^;]*$")
for i, text in enumerate(src, 1):
    if text.startswith(("DEV ", "template", "__device__", "__global__")) or re.match(r"^    DEV ", text):
        m = re.search(r"([A-Za-z_]\w*)\s*\(", text.replace("__launch_bounds__(", "_lb("))
        if m and "(" in text:
            defs.append((i, m.group(1)))
def func_of(line):
    name = "?"
    for i, n in defs:
        if i <= line:
            name = n
        else:
            break
    return name

cur = 0
valu = collections.Counter()
salu = collections.Counter()
mem = collections.Counter()
for text in open(listing):
    t = text.strip()
    m = re.match(r"\.loc\s+\d+\s+(\d+)", t)
    if m:
        cur = int(m.group(1))
        continue
    if t.startswith("v_"):
        valu[cur] += 1
    elif t.startswith("s_") and not t.startswith(("s_waitcnt", "s_nop", "s_branch", "s_cbranch")):
        salu[cur] += 1
    elif t.startswith(("global_", "ds_", "scratch_", "flat_")):
        mem[cur] += 1
byf = collections.Counter()
for line, n in valu.items():
    byf[func_of(line)] += n
total = sum(valu.values())
print("VALU %d  SALU %d  MEM %d" % (total, sum(salu.values()), sum(mem.values())))
for f, n in byf.most_common(60):
    print("%6d  %5.1f %%  %s" % (n, 100.0 * n / total, f))
if "--lines" in sys.argv:
    for line, n in valu.most_common(80):
        print("%6d  L%-5d %s" % (n, line, src[line - 1].strip()[:110]))
