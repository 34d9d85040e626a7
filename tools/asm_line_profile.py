"""Static profile of a kernel's day loop by source line: compile with -gline-tables-only -save-temps,
then   python tools/asm_line_profile.py <file.s> <mangled kernel name> [bucket]
Attributes every instruction of the largest loop to the innermost-inlined source location given by
the preceding .loc directive and prints instruction counts per (file, line-range bucket)."""
import collections
import re
import sys

path, name = sys.argv[1], sys.argv[2]
L = open(path).read().split("\n")
files = {}
for l in L:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
start = next(i for i, l in enumerate(L) if l.startswith(name + ":"))
end = next(i for i in range(start, len(L)) if L[i].startswith(".Lfunc_end"))
f = L[start:end]
labels = {}
for i, l in enumerate(f):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
best = None
for i, l in enumerate(f):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i and (best is None or i - labels[t] > best[1] - best[0]):
            best = (labels[t], i)
lo, hi = best
cur = None
# find the .loc in force at loop start
for l in f[:lo]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
cnt = collections.Counter()
valu = collections.Counter()
for l in f[lo:hi + 1]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    s = l.strip()
    if not s or s.startswith((";", ".")) or s.endswith(":"):
        continue
    cnt[cur] += 1
    if s.startswith("v_"):
        valu[cur] += 1
byline = collections.defaultdict(lambda: [0, 0])
for (fi, ln), c in cnt.items():
    byline[(files.get(fi, str(fi)), ln)][0] += c
    byline[(files.get(fi, str(fi)), ln)][1] += valu[(fi, ln)]
tot = sum(c for c, _ in byline.values())
print("loop instructions", tot)
for (fn, ln), (c, v) in sorted(byline.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    print(f"{fn}:{ln}\t{c}\t{v}")
