#!/usr/bin/env python3
"""Goldens for the bad-SUNK caller: runs the reference's own workflow/scripts/badsunks_AR.py,
unmodified, where it lies under /root/reference, through tests/golden/_shim/pandas13_names.py (two pandas-1.3 API behaviours
restored under the image's pandas 2.x; seaborn stand-in: the script's histogram is never saved).  The counting, the mode, the
limit and the four queries (badsunks_AR.py:20-103) are the reference's code running for real.
    make_golden_badsunks.py            writes tests/golden/badsunks/<case>/{f1,f2,s1,s2,expected,stdout,case.json}
    make_golden_badsunks.py --check    re-runs the script on the committed inputs and compares"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SCRIPT = "/root/reference/workflow/scripts/badsunks_AR.py"
RUNNER = os.path.join(HERE, "_shim", "pandas13_names.py")
OUT = os.path.join(HERE, "badsunks")


def synthetic(seed, own, other, n_reads=600):
    rng = np.random.default_rng(seed)
    rows = []
    for r in range(n_reads):
        for _ in range(int(rng.integers(1, 30))):
            c = own[int(rng.integers(0, len(own)))] if rng.random() < 0.8 else other[int(rng.integers(0, len(other)))]
            g = int(rng.zipf(1.4)) % 400 * 1000
            rows.append(f"r{r}\t{int(rng.integers(0, 50000))}\t{c}\t{g + 5}\t{g}\n")
    return "".join(rows)


def coverage_like(seed, own, other, depth, n_sunks=400):
    """every SUNK seen by about `depth` reads (Poisson), a few collapsed repeats far above it, singletons, and the other
    haplotype's SUNKs at half the depth: the shape the rule is made for (a clear mode, both tails populated)"""
    rng = np.random.default_rng(seed)
    rows = []
    for s in range(n_sunks):
        c = own[s % len(own)] if s % 5 else other[s % len(other)]
        lam = depth if s % 5 else depth / 2
        n = int(rng.poisson(lam))
        if s % 97 == 0:
            n = int(depth * rng.integers(3, 9))
        if s % 53 == 0:
            n = 1
        for _ in range(n):
            rows.append(f"read{int(rng.integers(0, 5000))}\t{int(rng.integers(0, 90000))}\t{c}\t{s * 700 + 11}\t{s * 700}\n")
    order = rng.permutation(len(rows))
    return "".join(rows[i] for i in order)


def cases():
    amy = os.path.join(HERE, "amy")
    amy_sp, amy_fai = open(os.path.join(amy, "expected.sunkpos")).read(), open(os.path.join(amy, "fai")).read()
    return {
        "synth": ("h1a\t10\nh1b\t10\n", "h2a\t10\n", synthetic(1, ["h1a", "h1b"], ["h2a"]), synthetic(2, ["h2a"], ["h1a", "h1b"])),
        "coverage": ("h1_ctg1\t100\t5\t60\t61\nh1_ctg2\t100\t5\t60\t61\n", "h2_ctg1\t100\t5\t60\t61\n",
                     coverage_like(3, ["h1_ctg1", "h1_ctg2"], ["h2_ctg1"], 30), coverage_like(4, ["h2_ctg1"], ["h1_ctg1", "h1_ctg2"], 22)),
        "amy": (amy_fai, "AMY_CHM1\t467195\t10\t60\t61\n", amy_sp, amy_sp),
        "colon_first_part_in_fai": ("chr1\t10\n", "chr2\t10\n", synthetic(5, ["chr1:1-5000"], ["chr2:7-9"], 200), synthetic(6, ["chr2:7-9"], ["chr1:1-5000"], 200)),
        "colon_whole_name_in_fai": ("chr1:1-5000\t10\n", "chr2:7-9\t10\n", synthetic(5, ["chr1:1-5000"], ["chr2:7-9"], 200),
                                    synthetic(6, ["chr2:7-9"], ["chr1:1-5000"], 200)),
    }


def run_script(d):
    out = os.path.join(d, "out.tmp")
    if os.path.exists(out):
        os.remove(out)
    env = dict(os.environ, PYTHONPATH=os.path.join(HERE, "_shim"), MPLBACKEND="Agg")
    p = subprocess.run([sys.executable, RUNNER, REF_SCRIPT] + [os.path.join(d, n) for n in ("f1", "f2", "s1", "s2")] + [out],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    got = sorted(open(out).read().splitlines()) if p.returncode == 0 else None
    if os.path.exists(out):
        os.remove(out)
    return p.returncode, got, p.stdout.decode(), p.stderr.decode()


def main():
    check = "--check" in sys.argv
    ok = True
    for name, (f1, f2, s1, s2) in cases().items():
        d = os.path.join(OUT, name)
        if not check:
            os.makedirs(d, exist_ok=True)
            for n, t in (("f1", f1), ("f2", f2), ("s1", s1), ("s2", s2)):
                open(os.path.join(d, n), "w").write(t)
        rc, got, so, se = run_script(d)
        meta = {"rc": rc, "n_bad": len(got) if got is not None else None, "stderr_last_line": se.strip().splitlines()[-1] if rc else "",
                "note": "badsunks_AR.py run in place under pandas 2.x through _shim/pandas13_names.py; `expected` = its output set, sorted"}
        if check:
            want = json.load(open(os.path.join(d, "case.json")))
            exp = sorted(open(os.path.join(d, "expected")).read().splitlines()) if want["rc"] == 0 else None
            same = want["rc"] == rc and exp == got and (rc != 0 or open(os.path.join(d, "stdout")).read() == so)
            ok = ok and same
            print(name, "reproduces" if same else "DIFFERS")
        else:
            json.dump(meta, open(os.path.join(d, "case.json"), "w"), indent=1, sort_keys=True)
            open(os.path.join(d, "expected"), "w").write("".join(e + "\n" for e in got) if got is not None else "")
            open(os.path.join(d, "stdout"), "w").write(so)
            print(name, meta)
    if check:
        print("badsunks goldens reproduce" if ok else "badsunks goldens DIFFER")
        return 0 if ok else 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
