#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE'S OWN
prebuilt binaries (the exact files GAVISUNK's Snakemake rules execute:
/root/reference/workflow/scripts/{kmerpos_annot3,rlen,diag_filter_v3,diag_filter_step2},
tagONT.smk:36,73,92,110) on inputs built here.

Runs only in the build container (the reference tree does not exist on the GPU box and
never enters this repository: only inputs and the outputs the binaries printed are kept).
Re-running must reproduce the committed files byte for byte: `python tests/golden/make_golden.py --check`.

Cases:
  kat1   tagger quirks Q1 (deflated pos) / Q2 (carry across reads), FASTA + FASTQ(.gz)
  kat2   encoding edge cases: lower case, multi-line FASTA, N->A, duplicate loc key (last wins)
  kat2_missing  k-mer in db but not in loc -> exit code 1
  short  reads shorter than k (Q6: one 'A'-padded window each)
  kat3-5 diag_filter_v3 / diag_filter_step2: band strictness, fai restriction, tie order, median truncation
  multi_k{20,21,25,31}  3 contigs with repeats, reads on both strands with errors
  amy    cfg 1: real AMY-locus assemblies (.test/data/AMY.hap{1,2}.fa.gz), simulated ONT reads
  tie    many reads x random contig-name pairs with exactly tied scores (pins Nim Table slot order)
"""
from __future__ import annotations

import argparse
import gzip
import hashlib
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from gavisunk_amd import synth  # noqa: E402
from oracle import sunkdef

REF = "/root/reference"
BIN = os.path.join(REF, "workflow", "scripts")
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def rc(s: str) -> str:
    return "".join(COMP[c] for c in reversed(s))


def canon(s: str) -> str:
    return min(s, rc(s))


def run(cmd, stdout=None):
    p = subprocess.run(cmd, stdout=stdout, stderr=subprocess.PIPE)
    return p.returncode, p.stderr.decode(errors="replace")


def gz_write(path: str, data: bytes) -> None:
    # mtime=0 so that regenerated .gz files are byte-identical
    with open(path, "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, filename="") as f:
        f.write(data)


class Case:
    def __init__(self, out_root: str, name: str):
        self.dir = os.path.join(out_root, name)
        os.makedirs(self.dir, exist_ok=True)
        self.name = name
        self.meta = {}

    def p(self, f: str) -> str:
        return os.path.join(self.dir, f)

    def write(self, f: str, text: str) -> str:
        with open(self.p(f), "w") as fh:
            fh.write(text)
        return self.p(f)

    def tag(self, reads: str, db: str = "db", loc: str = "loc", out: str = "expected.sunkpos"):
        rcode, err = run([os.path.join(BIN, "kmerpos_annot3"), self.p(reads), self.p(db), self.p(loc), self.p(out)])
        self.meta.setdefault("tag", []).append({"reads": reads, "db": db, "loc": loc, "out": out, "rc": rcode,
                                                "stderr_tail": err.strip().splitlines()[-1:] if rcode else []})
        return rcode

    def rlen(self, reads: str, out: str = "expected.rlen"):
        rcode, _ = run([os.path.join(BIN, "rlen"), self.p(reads), self.p(out)])
        assert rcode == 0
        self.meta.setdefault("rlen", []).append({"reads": reads, "out": out})

    def diag(self, sunkpos: str, fai: str, out: str = "expected.diag", out2: str = "expected.diag2"):
        with open(self.p(out), "wb") as fh:
            rcode, err = run([os.path.join(BIN, "diag_filter_v3"), self.p(sunkpos), self.p(fai)], stdout=fh)
        assert rcode == 0, err
        with open(self.p(out2), "wb") as fh:
            rcode, err = run([os.path.join(BIN, "diag_filter_step2"), self.p(sunkpos), self.p(out)], stdout=fh)
        assert rcode == 0, err
        self.meta.setdefault("diag", []).append({"sunkpos": sunkpos, "fai": fai, "out": out, "out2": out2})

    def tag_chunked(self, reads: str, nchunks: int, out: str, db: str = "db", loc: str = "loc"):
        """What the workflow does with a haplotype's reads (tagONT.smk:2-37,112-131): cut them into `nchunks` files record by
        record (`rustybam fastq-split` deals records round-robin -- third-party, not in this image: believed), run
        the reference tagger on every file, concatenate the outputs in scatter order."""
        data = gzip.open(self.p(reads), "rb").read() if reads.endswith(".gz") else open(self.p(reads), "rb").read()
        lines = data.split(b"\n")
        recs, cur = [], []
        fastq = data[:1] == b"@"
        i = 0
        while i < len(lines):
            if fastq:
                if lines[i][:1] == b"@" and i + 3 < len(lines) + 1:
                    recs.append(b"\n".join(lines[i:i + 4]) + b"\n")
                    i += 4
                    continue
                i += 1
            else:
                if lines[i][:1] == b">":
                    if cur:
                        recs.append(b"\n".join(cur) + b"\n")
                    cur = [lines[i]]
                elif cur and lines[i]:
                    cur.append(lines[i])
                i += 1
        if cur and not fastq:
            recs.append(b"\n".join(cur) + b"\n")
        parts = []
        with tempfile.TemporaryDirectory() as t:
            for c in range(nchunks):
                f = os.path.join(t, f"chunk{c}.{'fq' if fastq else 'fa'}")
                open(f, "wb").write(b"".join(recs[c::nchunks]))
                o = os.path.join(t, f"chunk{c}.sunkpos")
                rcode, err = run([os.path.join(BIN, "kmerpos_annot3"), f, self.p(db), self.p(loc), o])
                assert rcode == 0, err
                parts.append(open(o, "rb").read())
        open(self.p(out), "wb").write(b"".join(parts))
        self.meta.setdefault("tag_chunked", []).append({"reads": reads, "db": db, "loc": loc, "nchunks": nchunks, "out": out,
                                                        "records": len(recs)})

    def finish(self):
        with open(self.p("case.json"), "w") as fh:
            json.dump(self.meta, fh, indent=1, sort_keys=True)
            fh.write("\n")


# ----------------------------------------------------------------------------------
def case_kat1(root):
    c = Case(root, "kat1")
    random.seed(1)
    k = 20
    asm = "".join(random.choice("ACGT") for _ in range(3000))
    starts = list(range(100, 140)) + list(range(500, 530)) + list(range(1000, 1003)) + list(range(2000, 2050))
    groups, g, prev = [], None, None
    for s in starts:
        if prev is None or s - prev > k:
            g = s
        groups.append(g)
        prev = s
    c.write("db", "".join(canon(asm[s:s + k]) + "\n" for s in starts))
    c.write("loc", "".join(f"ctg1\t{s}\t{canon(asm[s:s + k])}\t{g}\n" for s, g in zip(starts, groups)))
    reads = [("r1 desc", asm[50:1100]), ("r2", rc(asm[900:2100])),
             ("r3", asm[90:110].lower() + "N" + asm[111:600]), ("r4", asm[1500:2030]),
             ("r5", asm[2010:2100]), ("r6", asm[100:119]), ("r7", asm[100:120])]
    c.write("reads.fa", "".join(f">{n}\n{s}\n" for n, s in reads))
    fq = "".join(f"@{n}\n{s}\n+\n{'#' * len(s)}\n" for n, s in reads)
    c.write("reads.fq", fq)
    gz_write(c.p("reads.fq.gz"), fq.encode())
    assert c.tag("reads.fa") == 0
    assert c.tag("reads.fq", out="expected_fq.sunkpos") == 0
    assert c.tag("reads.fq.gz", out="expected_fqgz.sunkpos") == 0
    c.rlen("reads.fa")
    c.tag_chunked("reads.fa", 2, "expected_chunks2.sunkpos")  # r4 and r5 part ways: r5 no longer vanishes
    c.tag_chunked("reads.fa", 3, "expected_chunks3.sunkpos")
    c.meta["k"] = k
    c.finish()


def case_kat2(root):
    c = Case(root, "kat2")
    random.seed(3)
    k = 20
    asm = "".join(random.choice("ACGT") for _ in range(400))
    A, B, C = canon(asm[10:30]), canon(asm[100:120]), canon(asm[200:220])
    c.write("db", f"{A}\n{B}\n{C}\n")
    c.write("loc", f"c1\t10\t{A}\t10\nc1\t100\t{B}\t100\nc1\t777\t{B}\t700\nc1\t200\t{C}\t200\n")
    base = asm[0:250]
    win = list(base[200:220])
    ia = next(i for i, ch in enumerate(win) if ch == "A")
    ina = next(i for i, ch in enumerate(win) if ch != "A")
    n_at_a = base[:200 + ia] + "N" + base[200 + ia + 1:]
    n_at_na = base[:200 + ina] + "N" + base[200 + ina + 1:]
    low = base.lower()
    fa = f">lower some description\n{low}\n"
    fa += ">multi\n" + low[:80] + "\n" + low[80:170] + "\n" + low[170:] + "\n"
    fa += f">N_at_A\n{n_at_a}\n>N_at_nonA\n{n_at_na}\n"
    fa += ">A_then_C_then_A\n" + asm[5:35] + asm[195:225] + asm[5:35] + "\n"
    c.write("reads.fa", fa)
    assert c.tag("reads.fa") == 0
    c.rlen("reads.fa")
    c.meta["k"] = k
    c.finish()
    # missing key: db holds B and C but the loc does not
    m = Case(root, "kat2_missing")
    shutil.copy(c.p("db"), m.p("db"))
    shutil.copy(c.p("reads.fa"), m.p("reads.fa"))
    m.write("loc", f"c1\t10\t{A}\t10\n")
    assert m.tag("reads.fa") == 1
    m.meta["k"] = k
    m.finish()


def case_short(root):
    """Q6: reads shorter than k (the empty read included) still yield ONE window, padded with
    'A' -- found while building these fixtures; SURVEY.md says otherwise (its KAT-1 r6 simply
    does not complete to a SUNK)."""
    c = Case(root, "short")
    k = 20
    rnd = random.Random(5)
    body = "".join(rnd.choice("CGT") for _ in range(400))
    kms = [body[j * 30:j * 30 + k - j] + "A" * j for j in range(0, 8)]
    kms += ["C" + "A" * 19, "A" * 20, "GT" + "A" * 18]
    c.write("db", "".join(canon(x) + "\n" for x in kms))
    c.write("loc", "".join(f"c{j}\t{j * 100}\t{canon(x)}\t{j * 100}\n" for j, x in enumerate(kms)))
    longread = "".join(rnd.choice("CGT") for _ in range(5000))
    recs = []
    for rep in range(2):
        for j in range(1, 8):
            recs.append((f"long{rep}_{j}", longread))
            recs.append((f"s{rep}_{j}", kms[j][:k - j]))
    recs += [("one", "C"), ("L2", longread), ("empty", ""), ("two", "GT"), ("empty2", ""), ("empty3", ""),
             ("one2", "C"), ("n1", "N"), ("x", kms[6][:6] + "A" * 14 + "C"), ("last_empty", "")]
    c.write("reads.fa", "".join(f">{n}\n{s}\n" for n, s in recs))
    c.write("reads.fq", "".join(f"@{n}\n{s}\n+\n{'#' * len(s)}\n" for n, s in recs))
    assert c.tag("reads.fa") == 0
    assert c.tag("reads.fq", out="expected_fq.sunkpos") == 0
    c.rlen("reads.fa")
    c.meta["k"] = k
    c.finish()


def case_diag(root):
    c = Case(root, "kat3")
    c.write("fai", "cA\t100000\t4\t60\t61\ncB\t100000\t4\t60\t61\ncC\t100000\t4\t60\t61\n")
    rows = """x1 0 cA 1000 1000|x1 100 cA 3599 3599
x2 0 cA 0 0|x2 1 cA 5000 5000
x3 0 cB 10 10|x3 50 cB 60 60|x3 100 cA 10 10|x3 150 cA 60 60
x4 0 cC 10 10|x4 5 cC 15 10|x4 50 cC 60 60|x4 100 cA 10 10|x4 150 cA 60 60
x5 0 cZ 10 10|x5 50 cZ 60 60
x6 1000 cA 500 500|x6 2001 cA 1500 1500
x7 5 cA 9 9
x8 100 cA 900 900|x8 200 cA 800 800|x8 300 cA 700 700"""
    txt = ""
    for line in rows.splitlines():
        for r in line.split("|"):
            txt += "\t".join(r.split()) + "\n"
    c.write("in.sunkpos", txt)
    c.diag("in.sunkpos", "fai")
    c.finish()

    c = Case(root, "kat4")
    c.write("fai", "cA\t1\ncB\t1\ncC\t1\n")
    txt = ""
    for rn, (a, b) in (("y1", ("cA", "cB")), ("y2", ("cB", "cC")), ("y3", ("cC", "cA")),
                       ("y4", ("cB", "cA")), ("y5", ("cC", "cB")), ("y6", ("cA", "cC"))):
        txt += f"{rn}\t0\t{a}\t10\t10\n{rn}\t50\t{a}\t60\t60\n{rn}\t100\t{b}\t10\t10\n{rn}\t150\t{b}\t60\t60\n"
    c.write("in.sunkpos", txt)
    c.diag("in.sunkpos", "fai")
    c.finish()

    c = Case(root, "kat5")
    c.write("fai", "cA\t1\n")
    c.write("in.sunkpos", "z1\t10000\tcA\t5000\t1\nz1\t20000\tcA\t15001\t2\nz1\t30000\tcA\t30000\t3\nz1\t40000\tcA\t40000\t4\n"
                          "z2\t5000\tcA\t10000\t1\nz2\t15001\tcA\t20000\t2\nz2\t30000\tcA\t30000\t3\nz2\t40000\tcA\t40000\t4\n"
                          "z3\t10\tcA\t100\t1\nz3\t20\tcA\t111\t2\nz3\t30\tcA\t120\t3\n")
    c.diag("in.sunkpos", "fai")
    c.finish()


def case_tie(root):
    """Exactly tied contigs under many different names: the winner is decided by the slot
    order of Nim's Table[string, ...] (diag_filter_v3.nim:76), i.e. by Nim's string hash."""
    c = Case(root, "tie")
    rnd = random.Random(11)
    names = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY", "h1tg000001l", "h2tg000019l", "ctg5M",
             "AMY_CHM13", "AMY_CHM1", "haplotype1-0000001", "haplotype2-0000002", "cA", "cB", "cC",
             "contig_1", "contig_2", "scaffold_10", "ptg000001l", "ptg000002l", "a", "b", "zz", "NC_060925.1",
             "JAHKSE010000012.1", "chr1_MATERNAL", "chr1_PATERNAL", "h1#tig1", "x" * 40]
    c.write("fai", "".join(f"{n}\t1\n" for n in names))
    txt = ""
    for i in range(400):
        m = rnd.choice([2, 2, 2, 3, 4, 6])
        pick = rnd.sample(names, m)
        p = 0
        for n in pick:
            txt += f"t{i}\t{p}\t{n}\t10\t10\nt{i}\t{p + 50}\t{n}\t60\t60\n"
            p += 100
    # one read that touches more contigs than 2/3 of the table's 64 initial slots (forces a grow),
    # followed by more tied reads (capacity persists after clear())
    big = names[:44]
    p = 0
    for n in big:
        txt += f"tbig\t{p}\t{n}\t10\t10\ntbig\t{p + 50}\t{n}\t60\t60\n"
        p += 100
    for i in range(400, 500):
        pick = rnd.sample(names, 2)
        p = 0
        for n in pick:
            txt += f"t{i}\t{p}\t{n}\t10\t10\nt{i}\t{p + 50}\t{n}\t60\t60\n"
            p += 100
    c.write("in.sunkpos", txt)
    c.diag("in.sunkpos", "fai")
    c.finish()


def _multi_genome(rng, n_contigs=3, clen=8000):
    fam = rng.integers(0, 4, 600, dtype=np.uint8)
    contigs = []
    for i in range(n_contigs):
        s = rng.integers(0, 4, clen, dtype=np.uint8)
        for _ in range(3):
            p = int(rng.integers(0, clen - 600))
            s[p:p + 600] = fam
        contigs.append(s)
    # contig 2 carries an inverted copy of a piece of contig 0 (reverse-strand SUNK-free zone)
    contigs[2][5000:6500] = synth.revcomp_codes(contigs[0][3000:4500])
    return contigs


def case_multi(root, k):
    c = Case(root, f"multi_k{k}")
    rng = np.random.default_rng(100 + k)
    contigs = _multi_genome(rng)
    names = ["ctgA", "ctgB#2", "ctgC"]
    sunks = sunkdef.define_sunks([(n, synth.codes_to_ascii(s)) for n, s in zip(names, contigs)], k)
    sunks.write_db(c.p("db"))
    sunks.write_loc(c.p("loc"))
    c.write("fai", "".join(f"{n}\t{len(s)}\t0\t60\t61\n" for n, s in zip(names[:2], contigs[:2])))
    lens = synth.lognormal_lengths(rng, 110, median=1500, sigma=0.6, lo=10, hi=7000)
    reads = synth.simulate_reads(rng, contigs, lens, sub=0.01, dele=0.01, ins=0.01)
    # chimeric reads (two loci), an error-free read, an empty read, reads shorter than k
    reads.append(np.concatenate([contigs[0][100:1500], contigs[1][5000:7000]]))
    reads.append(contigs[1][0:8000].copy())
    reads.append(np.zeros(0, np.uint8))
    reads.append(contigs[0][40:40 + k - 1].copy())
    reads.append(contigs[0][40:40 + k].copy())
    reads.append(synth.revcomp_codes(contigs[2][0:8000]))
    rnames = [f"read{i}/{i % 7} len={len(r)}" for i, r in enumerate(reads)]
    asc = [bytearray(synth.codes_to_ascii(r)) for r in reads]
    for i, a in enumerate(asc):  # sprinkle N / lower case / IUPAC
        if len(a) > 50 and i % 3 == 0:
            for p in rng.integers(0, len(a), 3):
                a[int(p)] = ord("N")
        if i % 5 == 0:
            asc[i] = bytearray(bytes(a).lower())
        if len(a) > 50 and i % 11 == 0:
            a[int(rng.integers(0, len(a)))] = ord("R")
    synth.write_fasta(c.p("reads.fa"), rnames, [bytes(a) for a in asc], width=70)
    buf = b"".join(b"@" + n.encode() + b"\n" + bytes(a) + b"\n+\n" + b"#" * len(a) + b"\n" for n, a in zip(rnames, asc))
    gz_write(c.p("reads.fq.gz"), buf)
    assert c.tag("reads.fa") == 0
    assert c.tag("reads.fq.gz", out="expected_fqgz.sunkpos") == 0
    c.rlen("reads.fa")
    c.tag_chunked("reads.fa", 4, "expected_chunks4.sunkpos")
    c.diag("expected.sunkpos", "fai")
    for f in ("db", "loc"):
        gz_write(c.p(f + ".gz"), open(c.p(f), "rb").read())
        os.remove(c.p(f))
    c.meta["k"] = k
    c.meta["n_sunks"] = len(sunks)
    c.meta["note"] = "db/loc are stored gzip-compressed (db.gz, loc.gz); tests decompress to a tmp dir"
    c.finish()


def case_amy(root):
    """cfg 1 (BASELINE.json configs[0]): the reference's own AMY-locus test assemblies, exact
    SUNKs at k=20 over hap1+hap2, reads simulated with the length distribution of the
    (absent) `.test/data/AMY_hap1.ONT.fa.gz` taken from its surviving .fai."""
    c = Case(root, "amy")
    k = 20
    h1 = synth.read_fasta(os.path.join(REF, ".test/data/AMY.hap1.fa.gz"))
    h2 = synth.read_fasta(os.path.join(REF, ".test/data/AMY.hap2.fa.gz"))
    contigs = [(n, s.upper()) for n, s in h1 + h2]
    sunks = sunkdef.define_sunks(contigs, k)
    sunks.write_db(c.p("db"))
    sunks.write_loc(c.p("loc"))
    for f in ("db", "loc"):
        gz_write(c.p(f + ".gz"), open(c.p(f), "rb").read())
    c.write("fai", open(os.path.join(REF, ".test/data/AMY.hap1.fa.gz.fai")).read())
    rng = np.random.default_rng(7)
    fai_lens = np.array([int(l.split("\t")[1]) for l in open(os.path.join(REF, ".test/data/AMY_hap1.ONT.fa.gz.fai"))])
    lens = np.minimum(rng.choice(fai_lens, 36), 120000)
    codes = [synth.ascii_to_codes(s) for _, s in contigs]
    reads = synth.simulate_reads(rng, codes, lens)
    names = [f"amy{i}" for i in range(len(reads))]
    buf = b"".join(b"@" + n.encode() + b"\n" + synth.codes_to_ascii(r) + b"\n+\n" + b"#" * len(r) + b"\n"
                   for n, r in zip(names, reads))
    gz_write(c.p("reads.fq.gz"), buf)
    assert c.tag("reads.fq.gz") == 0
    c.rlen("reads.fq.gz")
    c.tag_chunked("reads.fq.gz", 10, "expected_chunks10.sunkpos")  # the workflow's default nchunks (Snakefile:23-24)
    c.diag("expected.sunkpos", "fai")
    os.remove(c.p("db"))
    os.remove(c.p("loc"))
    c.meta.update(k=k, n_sunks=len(sunks), n_reads=len(reads), bases=int(sum(len(r) for r in reads)),
                  note="db/loc are stored gzip-compressed (db.gz, loc.gz); tests decompress to a tmp dir")
    c.finish()


def build(root):
    case_kat1(root)
    case_kat2(root)
    case_short(root)
    case_diag(root)
    case_tie(root)
    for k in (20, 21, 25, 31):
        case_multi(root, k)
    case_amy(root)


def digest(root):
    out = {}
    for d, _, files in sorted(os.walk(root)):
        for f in sorted(files):
            if f.endswith((".py", ".pyc")) or f == "MD5SUMS":
                continue
            p = os.path.join(d, f)
            if os.path.relpath(p, root).startswith(("pair_", "_shim", "config2_full", "badsunks")):  # made by make_golden_pair.py / _config2.py / _badsunks.py
                continue
            out[os.path.relpath(p, root)] = hashlib.md5(open(p, "rb").read()).hexdigest()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true", help="regenerate into a temp dir and compare with the committed files")
    a = ap.parse_args()
    if not os.path.isdir(BIN):
        sys.exit("reference tree not present: goldens can only be regenerated in the build container")
    if a.check:
        with tempfile.TemporaryDirectory() as t:
            build(t)
            new, old = digest(t), digest(HERE)
        bad = [f for f in sorted(set(new) | set(old)) if new.get(f) != old.get(f)]
        print("golden fixtures reproduce" if not bad else f"MISMATCH: {bad}")
        sys.exit(1 if bad else 0)
    build(HERE)
    with open(os.path.join(HERE, "MD5SUMS"), "w") as fh:
        for f, m in digest(HERE).items():
            fh.write(f"{m}  {f}\n")


if __name__ == "__main__":
    main()
