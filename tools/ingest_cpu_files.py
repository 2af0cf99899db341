"""Texts for tools/ingest_cpu_asan.sh: regular FASTA / FASTQ of several shapes, the irregular forms the device parser must refuse
(multi-line FASTQ, a short last record, mixed files ...) and randomly damaged copies.  usage: ingest_cpu_files.py <dir> <mutated copies>"""
import os
import random
import sys

d, n_mut = sys.argv[1], int(sys.argv[2])
rnd = random.Random(3)
ACGT = bytes(b"ACGT"[i & 3] for i in range(256))


def fq(n, lo, hi, crlf=False, tail=True):
    nl = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n):
        ln = rnd.randint(lo, hi)
        out.append(b"@r%d some comment" % i + nl + rnd.randbytes(ln).translate(ACGT) + nl + b"+" + nl + b"#" * ln + nl)
    t = b"".join(out)
    return t if tail else t[:-len(nl)]


def fa(n, lo, hi, width=0):
    out = []
    for i in range(n):
        ln = rnd.randint(lo, hi)
        s = rnd.randbytes(ln).translate(ACGT)
        if width:
            s = b"\n".join(s[a:a + width] for a in range(0, ln, width))
        out.append(b">c%d desc\n" % i + s + b"\n")
    return b"".join(out)


files = {
    "fq_small": fq(150, 1, 300), "fq_long": fq(4, 20000, 40000), "fq_crlf": fq(50, 10, 300, True), "fq_notail": fq(40, 10, 300, tail=False),
    "fa_one": fa(10, 100, 4000), "fa_wrap": fa(10, 100, 4000, 60), "fa_empty_seq": b">a\n\n>b\nACGT\n>c\n",
    "fq_one": b"@a\nACGT\n+\n####\n", "fq_at_qual": b"@a\nACGT\n+\n@@@@\n@b\nAC\n+\n##\n", "fq_multiline": b"@a\nAC\nGT\n+\n##\n##\n",
    "fq_short_last": b"@a\nACGT\n+\n####\n@b\nAC\n", "fa_then_fq": b">a\nACGT\n@b\nAC\n+\n##\n", "blank_lines": b"@a\nACGT\n+\n####\n\n\n",
    "only_header": b">x", "only_nl": b"@\n", "fq_empty_read": b"@a\n\n+\n\n@b\nA\n+\n#\n", "name_tabs": b"@r1\tx y\nACGT\n+\n####\n>c\n",
    "fa_blank_mid": b">a\nAC\n\nGT\n>b\nA\n", "long_name": b"@" + b"n" * 5000 + b" c\nACGT\n+\n####\n",
    "fq_cut_at_tiles": fq(200, 50, 120)[:16384 * 2],
}
base = fq(60, 5, 200)
for j in range(n_mut):
    b = bytearray(base)
    for _ in range(rnd.randint(1, 4)):
        m, p = rnd.randrange(4), rnd.randrange(len(b))
        if m == 0:
            b[p] = rnd.choice(b"\n@+>\r #A")
        elif m == 1:
            del b[p:p + rnd.randint(1, 50)]
        elif m == 2:
            b[p:p] = bytes(rnd.choice(b"\n@+>\rACGT") for _ in range(rnd.randint(1, 20)))
        elif p > 10:
            b = b[:p]
    files["mut%03d" % j] = bytes(b)
for k, v in files.items():
    with open(os.path.join(d, k + ".txt"), "wb") as f:
        f.write(v)
