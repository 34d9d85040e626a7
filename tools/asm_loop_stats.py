"""Instruction mix of the day loop of a kernel in the -save-temps .s (largest backward branch)."""
import collections
import re
import sys

path, name = sys.argv[1], sys.argv[2]
L = open(path).read().split("\n")
start = next(i for i, l in enumerate(L) if l.startswith(name + ":"))
end = next(i for i in range(start, len(L)) if L[i].startswith(".Lfunc_end"))
f = L[start:end]
labels = {}
for i, l in enumerate(f):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
back = []
for i, l in enumerate(f):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            back.append((labels[t], i, t))


def isinstr(l):
    l = l.strip()
    return l and not l.startswith((";", ".")) and not l.endswith(":")


lo, hi, _ = max(back, key=lambda x: x[1] - x[0])
body = [l.strip() for l in f[lo:hi + 1] if isinstr(l)]
ops = collections.Counter(l.split()[0] for l in body)
f64 = sum(v for k, v in ops.items() if "f64" in k)
valu = sum(v for k, v in ops.items() if k.startswith("v_"))
print(f"{name[:50]}: loop instr {len(body)}  f64 {f64}  valu {valu}  salu "
      f"{sum(v for k, v in ops.items() if k.startswith('s_'))}  accvgpr "
      f"{sum(v for k, v in ops.items() if 'accvgpr' in k)}  scratch "
      f"{sum(v for k, v in ops.items() if 'scratch' in k)}  prologue "
      f"{len([l for l in f[:lo] if isinstr(l)])}")
print(ops.most_common(30))
