"""Static instruction counts per basic block of one kernel in a hipcc -S listing (VALU / SALU / LDS / VMEM): the blocks of
k_probe_mz's step loop are the ones with an LDS-DMA (`dma`).   python tools/isa_blocks.py dev.s 'k_probe_mzILi20ELi17ELi1'"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r"^(_ZN\S*" + re.escape(sys.argv[2]) + r"\S*):", s, re.M)
a = m.start()
body = s[a:s.index(".Lfunc_end", a)]
blocks, cur = [], None
for ln in body.split("\n"):
    mm = re.match(r"^(\.LBB\d+_\d+):", ln)
    if mm:
        cur = [mm.group(1), []]
        blocks.append(cur)
        continue
    t = ln.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    if cur is None:
        cur = ["entry", []]
        blocks.append(cur)
    cur[1].append(t.split()[0])


def cls(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "VMEM"
    return "other"


tot = collections.Counter()
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 60
for nm, ops in blocks:
    c = collections.Counter(cls(o) for o in ops)
    tot += c
    if len(ops) >= minlen:
        print(nm, len(ops), dict(c), "dma" if any("load_lds" in o for o in ops) else "")
print("total", dict(tot))
